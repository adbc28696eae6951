#!/bin/bash
cd "$(dirname "$0")/.."
for pf in 1 0; do
FWGPU_RT_PREFETCH=$pf timeout 200 python bench.py --workload cfg5 --contexts 1 --no-cpu-baseline --no-parity-check --no-other-configs --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('prefetch=$pf cfg5 realtime_us_per_callback', d.get('realtime_us_per_callback'), d.get('realtime_path'))
"
done
