#!/bin/bash
# round 5: where does the pipelined host-buffer call lose its overlap?  copy engine vs blit kernel, own stream vs the ctx stream
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
run() {  # label, env...
  label=$1; shift
  env "$@" FWGPU_HOST_PROF=1 timeout 100 python bench.py --host-buffers --host-async --lean --contexts 1 --steps 30 --warmup 3 2> gpurun_out/r05/async_$label.err | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$label step_ms=%.4f value=%.3e' % (d['ms_per_step'], d['value']))
"
  grep "process_interleaved_end" gpurun_out/r05/async_$label.err | tail -1
}
timeout 100 python bench.py --host-buffers --lean --contexts 1 --steps 30 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('sync step_ms=%.4f value=%.3e' % (d['ms_per_step'], d['value']))
"
run sdma_own FWGPU_ASYNC_MODE=0
run blit_own FWGPU_ASYNC_MODE=0 HSA_ENABLE_SDMA=0
run sdma_ctx FWGPU_ASYNC_MODE=2
run blit_ctx FWGPU_ASYNC_MODE=2 HSA_ENABLE_SDMA=0
