#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_rt_resident.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05/rt_all.log 2>&1; echo "rc $?"; tail -2 gpurun_out/r05/rt_all.log
timeout 200 python bench.py --workload cfg5 --contexts 1 --no-cpu-baseline --no-parity-check --no-other-configs --steps 4 --warmup 1 2>gpurun_out/r05/cfg5_rt.err | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('cfg5 realtime_us_per_callback', d.get('realtime_us_per_callback'), d.get('realtime_us_per_callback_from_python'), 'value %.3e' % d['value'])
"
