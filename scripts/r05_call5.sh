#!/bin/bash
# round 5, fifth GPU call: the begin / end pair without pinned staging; the 8-process line with config 5 at 16 blocks per step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_benched_shapes.py -m gpu -x -q -p no:cacheprovider -k "begin_end" 2>&1 | tail -3
timeout 200 python bench.py --no-other-configs --contexts 1 --no-cpu-baseline > gpurun_out/r05/line_quick.json 2> gpurun_out/r05/line_quick.err; echo "quick rc $?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05/line_quick.json").read().strip().splitlines()[-1])
    print("value %.4g  host_buffers %s" % (d["value"], json.dumps(d.get("value_host_buffers"))[:700]))
except Exception as ex:
    print("quick line:", repr(ex))
PY
FWGPU_BENCH_PROGRESS=1 timeout 300 python bench.py --gpus 8 --share-device --steps 10 --warmup 2 > gpurun_out/r05/n8_line.json 2> gpurun_out/r05/n8_line.err
echo "bench n8 rc $?"; grep "^\[bench" gpurun_out/r05/n8_line.err | grep "rank 0" | tail -6; grep -v "^W0927\|amdgpu.ids\|socket.cpp\|Gloo\|^\[bench" gpurun_out/r05/n8_line.err | tail -5; tail -c 1500 gpurun_out/r05/n8_line.json
