#!/bin/bash
# round 5, first GPU call: BASELINE configs[4] whole (8 x 8 192 voices as 8 processes on one device) — the test and the bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_bus_exchange.py -m gpu -x -q -k "config5" > gpurun_out/r05/test_cfg5_world8.log 2>&1
echo "test rc $?" >> gpurun_out/r05/test_cfg5_world8.log
timeout 1200 python bench.py --gpus 8 --share-device > gpurun_out/r05/n8_line.json 2> gpurun_out/r05/n8_line.err
echo "bench rc $?" >> gpurun_out/r05/n8_line.err
tail -3 gpurun_out/r05/test_cfg5_world8.log; tail -c 600 gpurun_out/r05/n8_line.err; tail -c 1500 gpurun_out/r05/n8_line.json
