# round 6: same-box A/B of k_leaf_rs's pair convolution (frames 2l, 2l+1 share their window reads) against the frame-per-lane one
# (FWGPU_CHAIN_SKIP=512 takes the old path in the same binary): resampler-source bench + the deep parity check
for rep in 1 2; do
  for skip in 0 512; do
    FWGPU_CHAIN_SKIP=$skip timeout 300 python bench.py --rs-source --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-realtime --contexts 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
r = d['roofline']
pc = d.get('parity_check') or {}
print('skip=$skip', 'us=%.1f frac=%.3f value=%.3e step_ms=%.4f whole=%.3f parity=%s deep=%s' % (r['avg_launch_us'], r['frac'], d['value'], d['ms_per_step'], r['whole_step_frac'], pc.get('bit_exact'), (pc.get('deep') or {}).get('bit_exact')))
"
  done
done
