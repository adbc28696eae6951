#!/bin/bash
# round 5: blocks per wave on a very wide level (config 2 on the level executor alone)
cd "$(dirname "$0")/.."
for b in 8 16 32 8 32; do
  FWGPU_LEVEL_BPW_WIDE=$b timeout 100 python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('bpw_wide=$b value=%.3e step_ms=%.3f' % (d['value'], d['ms_per_step']))
"
done
FWGPU_LEVEL_BPW_WIDE=32 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "generic or hybrid or levels" 2>&1 | tail -2
