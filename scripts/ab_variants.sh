# usage (GPU box): bash scripts/ab_variants.sh "<bench args>" var_A.so var_B.so ...   — same-box A/B of library variants built into
# firewheel_amd/csrc/ (FWGPU_LIB), two interleaved repetitions each, with the bench line's parity_check against the oracle
args=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    FWGPU_LIB=$GRAFT_REPO_ROOT/firewheel_amd/csrc/$v timeout 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-other-configs --no-realtime $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('$v', 'kernel_us=%.1f GB/s=%.0f frac=%.3f step_ms=%.4f value=%.4g parity=%s' % (r['avg_launch_us'], r['achieved'], r['frac'], d['ms_per_step'], d['value'], (d.get('parity_check') or {}).get('bit_exact')))
"
  done
done
