"""debugging aid (needs a library built with `make -C firewheel_amd/csrc EXTRA=-DFW_DEBUG_BUS`): the same three one-block callbacks with
and without the resident realtime kernel, outputs and LEAF BUSES compared — how round 4 found that the garbage frames of the resident
kernel were pointer bits in lanes 12-15 of every 16 of a leaf bus (the inline-asm store hazard, k_leaf.hip.h bus_store_pair)."""
import sys, os, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import scenarios, fwapi
from fwapi import LOOP_FULL, GpuEngine
def run(persist):
    os.environ["FWGPU_RT_PERSIST"] = "1" if persist else "0"
    g = GpuEngine(max_block_frames=128, max_batch=4)
    voices = scenarios.build_voice_bank(g, 20, radix=8, src_frames=900)
    for vc in voices:
        g.sampler_set_loop_range(vc["sampler"], LOOP_FULL); g.sampler_play(vc["sampler"])
    outs = [np.asarray(g.process_interleaved(128)).copy() for _ in range(3)]
    L = g.cx.L
    L.fwgpu_debug_read_bus.restype = C.c_int
    L.fwgpu_debug_read_bus.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint64]
    buf = np.zeros(64 * 128, np.float32)
    nb = L.fwgpu_debug_read_bus(g.cx.c, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size)
    return outs, buf.reshape(-1, 128)[:nb], g.cx.rt_resident_stats()
o1, b1, s1 = run(True)
o0, b0, s0 = run(False)
print("stats", s1, s0)
for i in range(3):
    d = np.nonzero(o1[i].view(np.uint32) != o0[i].view(np.uint32))[0]
    print("callback", i, "out differs at", d[:20], len(d))
for r in range(b1.shape[0]):
    d = np.nonzero(b1[r].view(np.uint32) != b0[r].view(np.uint32))[0]
    if len(d): print("bus row", r, "differs at frames", d[:24], len(d), b1[r][d[:4]], b0[r][d[:4]])
print("bus rows", b1.shape)
