#!/bin/bash
# round 5, second GPU call: the register-window k_leaf_rs (parity: whole GPU tier; A/B against the round-4 kernel), the 8-process line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05/suite_a.log 2>&1; echo "suite rc $?" >> gpurun_out/r05/suite_a.log; tail -4 gpurun_out/r05/suite_a.log
for rep in 1 2; do
  for v in libfwgpu.so libfwgpu_rs0.so; do
    FWGPU_LIB=$PWD/firewheel_amd/csrc/$v timeout 200 python bench.py --rs-source --steps 20 --warmup 3 --lean 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('$v', 'kernel', r.get('kernel'), 'us=%.1f value=%.3e step_ms=%.4f whole=%.3f idle=%.1f parity=%s deep=%s' % (r['avg_launch_us'], d['value'], d['ms_per_step'], r.get('whole_step_frac') or 0, r.get('idle_us_per_step') or 0, (d.get('parity_check') or {}).get('bit_exact'), ((d.get('parity_check') or {}).get('deep') or {}).get('bit_exact')))
"
  done
done 2>&1 | tee gpurun_out/r05/ab_rs.txt
for rr in 0.55,0.55 0.8,0.8 1.0,1.0 1.1,1.1 1.5,1.5 1.9,1.9; do
  for v in libfwgpu.so libfwgpu_rs0.so; do
    FWGPU_BENCH_RS_RATIO=$rr FWGPU_LIB=$PWD/firewheel_amd/csrc/$v timeout 200 python bench.py --rs-source --steps 10 --warmup 2 --lean --no-parity-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('ratio $rr $v us=%.1f step_ms=%.4f' % (r['avg_launch_us'], d['ms_per_step']))
"
  done
done 2>&1 | tee gpurun_out/r05/ab_rs_ratio.txt
FWGPU_BENCH_PROGRESS=1 timeout 420 python bench.py --gpus 8 --share-device --steps 10 --warmup 2 > gpurun_out/r05/n8_line.json 2> gpurun_out/r05/n8_line.err
echo "bench n8 rc $?"; grep "^\[bench" gpurun_out/r05/n8_line.err | tail -12
FWGPU_POISON=2 timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05/suite_poison2.log 2>&1; echo "suite rc $?" >> gpurun_out/r05/suite_poison2.log; tail -3 gpurun_out/r05/suite_poison2.log
