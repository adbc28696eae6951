#!/usr/bin/env python3
"""placement_probe.py — is k_leaf_sum's 257 / 291 us split a property of the CONTEXT (where its buffers landed)?
N contexts alive at once on the same sources and the same stream, each timed in turn, three rounds.
usage (GPU box): python scripts/placement_probe.py [n_contexts [decoy_gb [before-src]]]
PROBE_SOAK / PROBE_STAGGER / PROBE_SWAP: see below.  What it found: the state follows the BUS buffer — where the
leaf buses sit relative to the samples; scripts/ubench/bus_place.hip reproduces it without the library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import time

import torch

import bench
import firewheel_amd as fa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
decoy_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0  # held while the contexts are created
decoy_first = len(sys.argv) > 3 and sys.argv[3] == "before-src"
V, B, K, F, _ = bench.DEFAULTS["cfg2"]
args = argparse.Namespace(radix=32, master=False, voice_fx=False, rs_source=False, force_generic=False, taps=65536)
decoy = torch.empty(int(decoy_gb * (1 << 30)), dtype=torch.uint8, device="cuda") if decoy_gb and decoy_first else None
# PROBE_STAGGER="0,1056": one set of sources per value — that many floats of padding between consecutive voices' buffers —
# and the contexts take them in turn (where the 2048 lockstep streams sit relative to one another in HBM)
staggers = [int(x) for x in os.environ.get("PROBE_STAGGER", "0").split(",")]
srcs = []
for st in staggers:
    pitch = 2 * F + st
    t = torch.empty(V * pitch, dtype=torch.float32, device="cuda").as_strided((V, 2, F), (pitch, F, 1))
    t.uniform_(-1.0, 1.0)
    srcs.append(t)
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty(K * B * 2, dtype=torch.float32, device="cuda")
if decoy_gb and not decoy_first:
    decoy = torch.empty(int(decoy_gb * (1 << 30)), dtype=torch.uint8, device="cuda")
# PROBE_SOAK="16:200,64:50": hold that many buffers of that many MiB while the contexts are created — soaks up the VRAM heap's
# small free blocks so that the contexts' own buffers are cut from large ones
soak = [torch.empty(int(mb) << 20, dtype=torch.uint8, device="cuda")
        for mb, cnt in (x.split(":") for x in os.environ.get("PROBE_SOAK", "").split(",") if x) for _ in range(int(cnt))]
ctxs = []
made = []
for i in range(n):
    t0 = time.perf_counter()
    cx, g, samplers, volumes = bench.make_gpu(fa, "cfg2", V, B, K, 32, srcs[i % len(srcs)], F, "f32", 0, args, stream, 0)
    ctxs.append(cx)
    made.append(round((time.perf_counter() - t0) * 1e3))
print("context build ms:", made, flush=True)
for rnd in range(3):
    row = []
    for cx in ctxs:
        for _ in range(5):
            cx.process_blocks_device(K, out.data_ptr(), 2)
        torch.cuda.synchronize()
        cx.timing_reset()
        cx.timing_enable(True)
        for _ in range(20):
            cx.process_blocks_device(K, out.data_ptr(), 2)
        torch.cuda.synchronize()
        cx.timing_enable(False)
        ms, cnt = cx.timing_read(0)
        row.append(round(ms / cnt * 1e3, 1))
    print("decoy %.0f GB %s," % (decoy_gb, "before src" if decoy_first else "after src") + (" stagger %s," % staggers if len(staggers) > 1 else "") + (" soak %s," % os.environ["PROBE_SOAK"] if soak else "") + " round", rnd,
          "k_leaf_sum us per context:", row, flush=True)


def leaf_us(cx):
    for _ in range(5):
        cx.process_blocks_device(K, out.data_ptr(), 2)
    torch.cuda.synchronize()
    cx.timing_reset()
    cx.timing_enable(True)
    for _ in range(20):
        cx.process_blocks_device(K, out.data_ptr(), 2)
    torch.cuda.synchronize()
    cx.timing_enable(False)
    ms, cnt = cx.timing_read(0)
    return round(ms / cnt * 1e3, 1)


# PROBE_SWAP=1 (library built with make EXTRA=-DFW_PROBE): exchange one device table at a time between the slowest and the
# fastest context — both hold the same graph, so every table is interchangeable — and see which one the state follows
if os.environ.get("PROBE_SWAP"):
    import ctypes as C

    L = ctxs[0].L
    L.fwgpu_probe_swap.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.fwgpu_probe_swap.restype = C.c_int
    slow = max(range(n), key=lambda i: row[i])
    fast = min(range(n), key=lambda i: row[i])
    print("slowest context %d (%.1f us), fastest %d (%.1f us)" % (slow, row[slow], fast, row[fast]), flush=True)
    names = ["bus", "refs", "gain sets", "leaf descs", "sample table", "voice tables", "ramp buffer"]
    for which, name in enumerate(names):
        assert L.fwgpu_probe_swap(ctxs[slow].c, ctxs[fast].c, which) == 0
        a, b = leaf_us(ctxs[slow]), leaf_us(ctxs[fast])
        assert L.fwgpu_probe_swap(ctxs[slow].c, ctxs[fast].c, which) == 0
        print("  %-12s exchanged: was-slow %.1f us, was-fast %.1f us" % (name, a, b), flush=True)
    print("  back in place:          was-slow %.1f us, was-fast %.1f us" % (leaf_us(ctxs[slow]), leaf_us(ctxs[fast])), flush=True)
