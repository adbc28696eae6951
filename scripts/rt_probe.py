"""Where does a realtime callback's time go?  (run on the GPU box)  Prints, for the config-2 graph with one block per call:
the whole synchronous callback (fwgpu_stream_callback), the asynchronous call alone (launch cost on the host), and — when run
under `rocprofv3 --kernel-trace --stats` — leaves the kernel durations in the trace."""
import sys
import time

import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import firewheel_amd as fa  # noqa: E402


class A:
    force_generic = master = voice_fx = rs_source = False
    taps = 65536


wl, V, B = (sys.argv[1] if len(sys.argv) > 1 else "cfg2"), int(sys.argv[2]) if len(sys.argv) > 2 else 1024, int(sys.argv[3]) if len(sys.argv) > 3 else 256
src = torch.empty((V, 2, 8192), dtype=torch.float32, device="cuda").uniform_(-1, 1)
cx, g, s, v = bench.make_gpu(fa, wl, V, B, 4, 32, src, 8192, "f32", 0, A, torch.cuda.current_stream().cuda_stream, 0)
print(wl, V, B, "callback us %.2f (native loop), %.2f (driven from Python)" % bench.realtime_probe(cx, B, 3000))
out = torch.empty(B * 2, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
n = 3000
for _ in range(100):
    cx.process_blocks_device(1, out.data_ptr(), 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    cx.process_blocks_device(1, out.data_ptr(), 2)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("async call (host launch cost) us %.2f; drained after %.2f us per call" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
# one call + sync, device output
for _ in range(100):
    cx.process_blocks_device(1, out.data_ptr(), 2)
    cx.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    cx.process_blocks_device(1, out.data_ptr(), 2)
    cx.synchronize()
t1 = time.perf_counter()
print("call + hipStreamSynchronize, output left in HBM us %.2f" % ((t1 - t0) / n * 1e6))
cx.close()
