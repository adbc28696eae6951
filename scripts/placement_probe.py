#!/usr/bin/env python3
"""placement_probe.py — is k_leaf_sum's 257 / 291 us split a property of the CONTEXT (where its buffers landed)?
N contexts alive at once on the same sources and the same stream, each timed in turn, three rounds.
usage (GPU box): python scripts/placement_probe.py [n_contexts]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse

import torch

import bench
import firewheel_amd as fa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
decoy_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0  # held while the contexts are created
decoy_first = len(sys.argv) > 3 and sys.argv[3] == "before-src"
V, B, K, F, _ = bench.DEFAULTS["cfg2"]
args = argparse.Namespace(radix=32, master=False, voice_fx=False, rs_source=False, force_generic=False, taps=65536)
decoy = torch.empty(int(decoy_gb * (1 << 30)), dtype=torch.uint8, device="cuda") if decoy_gb and decoy_first else None
src = torch.empty((V, 2, F), dtype=torch.float32, device="cuda")
src.uniform_(-1.0, 1.0)
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty(K * B * 2, dtype=torch.float32, device="cuda")
if decoy_gb and not decoy_first:
    decoy = torch.empty(int(decoy_gb * (1 << 30)), dtype=torch.uint8, device="cuda")
ctxs = []
for i in range(n):
    cx, g, samplers, volumes = bench.make_gpu(fa, "cfg2", V, B, K, 32, src, F, "f32", 0, args, stream, 0)
    ctxs.append(cx)
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
    print("decoy %.0f GB %s, round" % (decoy_gb, "before src" if decoy_first else "after src"), rnd, "k_leaf_sum us per context:", row, flush=True)
