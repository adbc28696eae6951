# round 6: same-box A/B of k_leaf_rs builds (make OBJDIR=_obj_X OUT=libfwgpu_X.so EXTRA=-D...), resampler-source bench + the deep parity check
for rep in 1 2; do
  for l in libfwgpu.so $(cd firewheel_amd/csrc && ls libfwgpu_r*.so 2>/dev/null); do
    FWGPU_LIB=$PWD/firewheel_amd/csrc/$l timeout 300 python bench.py --rs-source --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-realtime --contexts 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
pc = d.get('parity_check') or {}
print('$l', 'us=%.1f frac=%.3f value=%.3e step_ms=%.4f whole=%.3f parity=%s deep=%s' % (r['avg_launch_us'], r['frac'], d['value'], d['ms_per_step'], r['whole_step_frac'], pc.get('bit_exact'), (pc.get('deep') or {}).get('bit_exact')))
"
  done
done
