"""debugging: which voice shapes of the chain grammar leave lazy-capable records (one bank per shape, test_chain_grammar.run_quiet, the lazy counter per call)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import fwapi  # noqa: E402
import test_chain_grammar as t  # noqa: E402
from fwapi import GpuEngine  # noqa: E402

mbf = 128
for sh in t.ACCEPTED + ["v", "vp", "mvp", "pc"]:
    e = GpuEngine(max_block_frames=mbf, max_batch=8)
    _, marks = t.run_quiet(e, [sh] * 10, mbf, fwapi.PLANAR_F32)
    print(sh, [m[0] for m in marks], flush=True)
