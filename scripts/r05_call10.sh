#!/bin/bash
# round 5: level executor, 32 blocks per wave on very wide levels + early exit of links rendered upstream; FZ_U = 8 as a variant
cd "$(dirname "$0")/.."
for v in libfwgpu.so libfwgpu_fz8.so libfwgpu.so libfwgpu_fz8.so; do
  FWGPU_LIB=$PWD/firewheel_amd/csrc/$v timeout 100 python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$v value=%.3e step_ms=%.3f' % (d['value'], d['ms_per_step']))
"
done
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
FWGPU_LIB=$PWD/firewheel_amd/csrc/libfwgpu_fz8.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "generic or hybrid" 2>&1 | tail -2
