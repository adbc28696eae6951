#!/bin/bash
cd "$(dirname "$0")/.."
timeout 400 python -m pytest tests/test_rt_resident.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -12
timeout 100 python - <<'PY'
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, bench, firewheel_amd as fa, torch
# config 5's shard, one block per callback through the headless stream
import argparse
PY
timeout 200 python bench.py --workload cfg5 --contexts 1 --no-cpu-baseline --no-parity-check --no-other-configs --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('cfg5 realtime_us_per_callback', d.get('realtime_us_per_callback'), 'value %.3e' % d['value'])
"
