#!/bin/bash
# A/B of k_leaf_rs builds on ONE box: libfwgpu.so against firewheel_amd/csrc/var_*.so (FWGPU_LIB), resampler-source bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  for v in default $(ls firewheel_amd/csrc/var_*.so 2>/dev/null); do
    if [ $v = default ]; then L=$GRAFT_REPO_ROOT/firewheel_amd/csrc/libfwgpu.so; else L=$GRAFT_REPO_ROOT/$v; fi
    FWGPU_LIB=$L timeout 200 python bench.py --rs-source --steps 20 --warmup 3 --lean 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('$v', 'kernel', r.get('kernel'), 'us=%.1f value=%.3e step_ms=%.4f parity=%s' % (r['avg_launch_us'], d['value'], d['ms_per_step'], (d.get('parity_check') or {}).get('bit_exact')))
"
  done
done 2>&1 | tee gpurun_out/ab_rs.txt
