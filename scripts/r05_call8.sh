#!/bin/bash
# round 5: the pipelined host-buffer call with the graph-output kernel writing straight into mapped host staging
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_gpu_benched_shapes.py -m gpu -x -q -p no:cacheprovider -k "begin_end" 2>&1 | tail -2
run() {  # label, env...
  label=$1; shift
  env "$@" FWGPU_HOST_PROF=1 timeout 100 python bench.py --host-buffers --host-async --lean --contexts 1 --steps 30 --warmup 3 2> gpurun_out/r05/async_$label.err | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$label step_ms=%.4f value=%.3e' % (d['ms_per_step'], d['value']))
"
  grep "process_interleaved_end" gpurun_out/r05/async_$label.err | tail -1
}
for rep in 1 2; do
timeout 100 python bench.py --lean --contexts 1 --steps 30 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('device-resident step_ms=%.4f value=%.3e' % (d['ms_per_step'], d['value']))
"
timeout 100 python bench.py --host-buffers --lean --contexts 1 --steps 30 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('sync step_ms=%.4f value=%.3e' % (d['ms_per_step'], d['value']))
"
run mapped FWGPU_ASYNC_MODE=3
run sdma_ctx FWGPU_ASYNC_MODE=2
done
