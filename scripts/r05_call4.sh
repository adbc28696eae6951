#!/bin/bash
# round 5, fourth GPU call: whole GPU tier (mono-adapter voices, begin / end pair, path counters), the 8-process lines, the pipelined
# host-buffer figure
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05/suite_c.log 2>&1; echo "suite rc $?" >> gpurun_out/r05/suite_c.log; tail -15 gpurun_out/r05/suite_c.log
timeout 200 python bench.py --no-other-configs --contexts 1 --no-cpu-baseline > gpurun_out/r05/line_quick.json 2> gpurun_out/r05/line_quick.err; echo "quick rc $?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05/line_quick.json").read().strip().splitlines()[-1])
    print("value %.4g  host_buffers %s" % (d["value"], json.dumps(d.get("value_host_buffers"))[:600]))
    print("rt", d.get("realtime_us_per_callback"))
except Exception as ex:
    print("quick line:", repr(ex))
PY
FWGPU_BENCH_PROGRESS=1 timeout 200 python bench.py --gpus 8 --share-device --steps 10 --warmup 2 --no-other-configs > gpurun_out/r05/n8_cfg2_line.json 2> gpurun_out/r05/n8_cfg2_line.err
echo "bench n8 cfg2 rc $?"; tail -c 400 gpurun_out/r05/n8_cfg2_line.json
FWGPU_BENCH_PROGRESS=1 timeout 200 python bench.py --workload cfg5 --gpus 8 --share-device --reduce-every 1 --steps 6 --warmup 2 --no-other-configs > gpurun_out/r05/n8_cfg5_line.json 2> gpurun_out/r05/n8_cfg5_line.err
echo "bench n8 cfg5 rc $?"; grep "^\[bench" gpurun_out/r05/n8_cfg5_line.err | grep "rank 0" | tail -8; tail -c 600 gpurun_out/r05/n8_cfg5_line.json
