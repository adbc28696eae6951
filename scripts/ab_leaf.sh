cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for v in W1 W2 W4 W4U6; do
  for rep in 1 2; do
    FWGPU_LIB=$GRAFT_REPO_ROOT/firewheel_amd/csrc/var_$v.so timeout 120 python bench.py --steps 60 --warmup 5 --lean 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('$v', 'leaf_us=%.1f GB/s=%.0f step_ms=%.4f ctl_us=%.1f up_us=%.1f' % (r['avg_launch_us'], r['achieved'], d['ms_per_step'], r['other_kernels_us_per_step']['k_voice_control'], r['other_kernels_us_per_step']['upper_sums+graph_out']))
"
  done
done
