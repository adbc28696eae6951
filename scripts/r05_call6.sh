#!/bin/bash
# round 5, sixth GPU call: the begin / end pair over registered ordinary memory (where does an `end` spend its time?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_gpu_benched_shapes.py -m gpu -x -q -p no:cacheprovider -k "begin_end" 2>&1 | tail -3
FWGPU_HOST_PROF=1 timeout 200 python bench.py --no-other-configs --contexts 1 --no-cpu-baseline --no-parity-check > gpurun_out/r05/line_quick.json 2> gpurun_out/r05/line_quick.err; echo "quick rc $?"
grep "process_interleaved_end" gpurun_out/r05/line_quick.err | tail -3
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05/line_quick.json").read().strip().splitlines()[-1])
    print("value %.4g  host_buffers %s" % (d["value"], json.dumps(d.get("value_host_buffers"))[:500]))
except Exception as ex:
    print("quick line:", repr(ex))
PY
