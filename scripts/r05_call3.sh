#!/bin/bash
# round 5, third GPU call: vertical fusion in the level executor (whole GPU tier + config 2 on the levels alone, fused / unfused),
# k_leaf_rs variants on one box, the 8-process line again (exchanges waited for per step)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05/suite_b.log 2>&1; echo "suite rc $?" >> gpurun_out/r05/suite_b.log; tail -4 gpurun_out/r05/suite_b.log
for f in 1 0 1 0; do
  FWGPU_LEVEL_FUSE=$f timeout 200 python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('levels fuse=$f value=%.3e step_ms=%.3f' % (d['value'], d['ms_per_step']))
"
done 2>&1 | tee gpurun_out/r05/ab_levels.txt
for rep in 1 2; do
  for v in libfwgpu.so libfwgpu_rsB.so libfwgpu_rsC.so libfwgpu_rs0.so; do
    FWGPU_LIB=$PWD/firewheel_amd/csrc/$v timeout 200 python bench.py --rs-source --steps 20 --warmup 3 --lean 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('$v', 'kernel', r.get('kernel'), 'us=%.1f value=%.3e step_ms=%.4f whole=%.3f idle=%.1f' % (r['avg_launch_us'], d['value'], d['ms_per_step'], r.get('whole_step_frac') or 0, r.get('idle_us_per_step') or 0))
"
  done
done 2>&1 | tee gpurun_out/r05/ab_rs2.txt
FWGPU_BENCH_PROGRESS=1 timeout 300 python bench.py --gpus 8 --share-device --steps 10 --warmup 2 > gpurun_out/r05/n8_line.json 2> gpurun_out/r05/n8_line.err
echo "bench n8 rc $?"; grep "^\[bench" gpurun_out/r05/n8_line.err | grep "rank 0" | tail -8; tail -c 600 gpurun_out/r05/n8_line.json
