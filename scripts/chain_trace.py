"""Role timelines of k_chain (profiling builds only).

Build a tracing variant of the library and run config 3 on it:
    cd firewheel_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DFW_CHAIN_TRACE \
        -shared -o libfwgpu_trace.so -x hip fwgpu_kernels.hip fwgpu_*.cpp
    FWGPU_LIB=firewheel_amd/csrc/libfwgpu_trace.so python scripts/chain_trace.py
Prints, for steps 8..23 of workgroup 0, the clock64() deltas of each role: S3a, S1, issue, barrier wait.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import firewheel_amd as fa  # noqa: E402

V, B, K, F = 4096, 512, 16, 65536
radix = int(os.environ.get("RADIX", "32"))
stream = torch.cuda.current_stream().cuda_stream
src = torch.empty((V, 2, F), dtype=torch.float32, device="cuda").uniform_(-1, 1)


class _Args(object):
    force_generic = master = voice_fx = rs_source = False
    taps = 65536


cx, _, samplers, _ = bench.make_gpu(fa, "cfg3", V, B, K, radix, src, F, "f32", 0, _Args, stream, 0)
out = torch.empty(K * B * 2, dtype=torch.float32, device="cuda")
for _ in range(3):
    cx.process_blocks_device(K, out.data_ptr(), 2)
torch.cuda.synchronize()
buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
f = cx.L.fwgpu_debug_read_trace
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p]
assert f(cx.c, buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.reshape(64, 16, 8).astype(np.int64)
base = t[8, 0, 0]
print("clock64 ticks; wave 2 = serial (S2), 11 = mixer (S3b), 6/10 idle, the rest = workers")
print("SIMD of waves 0..9 (HW_ID bits 5:4):", [int((t[8, w, 7] >> 4) & 3) for w in range(12)])
print("step | serial: work wait | worker1: issue S3a S1 wait | worker8: issue S3a S1 wait | mixer: work wait | step len")
for s in range(8, 28):
    ser = t[s, 2]
    w1 = t[s, 0]
    w8 = t[s, 9]
    mx = t[s, 11]
    step_len = t[s + 1, 2, 0] - t[s, 2, 0]
    print("   S3a detail w1: lds-wait %d math+stores+ldswr %d tail %d | w8: %d %d %d" % (
        w1[5] - w1[2], w1[6] - w1[5], w1[3] - w1[6], w8[5] - w8[2], w8[6] - w8[5], w8[3] - w8[6]))
    print("%4d | %6d %6d | %6d %6d %6d %6d | %6d %6d %6d %6d | %6d %6d | %6d" % (
        s, ser[3] - ser[0], ser[4] - ser[3],
        w1[1] - w1[0], w1[2] - w1[1], w1[3] - w1[2], w1[4] - w1[3],
        w8[1] - w8[0], w8[2] - w8[1], w8[3] - w8[2], w8[4] - w8[3],
        mx[3] - mx[0], mx[4] - mx[3], step_len))

# steady-call loop (k_chain's branch-free worker steps): slots 0 top, 2 after S1 + source request, 3 before / 4 after barrier
steady, general = cx.plan_chain_stats()
if steady:
    print("steady-call loop ran (%d workgroup launches; %d general)" % (steady, general))
    print("step | serial: work wait | worker1: S1+req S3a+req wait | worker8: S1+req S3a+req wait | mixer: work wait | step len")
    tot = np.zeros(11)
    for s in range(8, 40):
        ser, w1, w8, mx = t[s, 2], t[s, 0], t[s, 9], t[s, 11]
        row = [ser[3] - ser[0], ser[4] - ser[3], w1[2] - w1[0], w1[3] - w1[2], w1[4] - w1[3],
               w8[2] - w8[0], w8[3] - w8[2], w8[4] - w8[3], mx[3] - mx[0], mx[4] - mx[3], t[s + 1, 2, 0] - t[s, 2, 0]]
        tot += np.array(row, dtype=np.float64)
        if s < 20:
            print("%4d | %6d %6d | %6d %6d %6d | %6d %6d %6d | %6d %6d | %6d" % tuple([s] + row))
    print("mean | %6d %6d | %6d %6d %6d | %6d %6d %6d | %6d %6d | %6d" % tuple((tot / 32).astype(int)))
    print("per worker wave (mean over steps 8..39): wave simd | S1+req  S3a+req  barrier-wait")
    for w in (0, 1, 3, 4, 5, 7, 8, 9):
        a = np.array([[t[s, w, 2] - t[s, w, 0], t[s, w, 3] - t[s, w, 2], t[s, w, 4] - t[s, w, 3]] for s in range(8, 40)
                      if 0 < t[s, w, 4] - t[s, w, 0] < 20000], dtype=np.float64)
        print("   wave %2d simd %d | %6d %6d %6d" % ((w, int((t[8, w, 7] >> 4) & 3)) + tuple(a.mean(axis=0).astype(int))))
