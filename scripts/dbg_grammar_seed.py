"""debugging aid (GPU box): python scripts/dbg_grammar_seed.py SEED [SEED...] — replays tests/test_chain_grammar.py's fuzz seed, and when the
HIP path differs from the oracle, replays it once per voice with every other voice stopped, to name the voice, its shape and its messages"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import fwapi  # noqa: E402
import scenarios  # noqa: E402
import test_chain_grammar as t  # noqa: E402

fwapi.build_oracle()
for seed in [int(x) for x in sys.argv[1:]]:
    mbf = [128, 64, 256][seed % 3]
    max_batch = int(os.environ.get("DBG_MAX_BATCH", [64, 1, 3, 8][seed % 4]))

    def both(only=None, generic=False, log=None):
        o = scenarios.TaggedOracle(fwapi.OracleEngine(max_block_frames=mbf))
        g = fwapi.GpuEngine(max_block_frames=mbf, max_batch=max_batch, force_generic=generic)
        a, b = t.fuzz_grammar(o, seed, only, log), t.fuzz_grammar(g, seed, only)
        bad = np.nonzero(fwapi.bits(np.asarray(a)) != fwapi.bits(np.asarray(b)))[0]
        return bad, g.cx.plan_kind(), np.asarray(a), np.asarray(b)

    log = []
    bad, plan, a, b = both(log=log)
    print("seed %d mbf %d K<=%d plan %d: %d of %d differ%s" % (seed, mbf, max_batch, plan, bad.size, a.size, (", first at block %d" % (bad[0] // (2 * mbf))) if bad.size else ""))
    if not bad.size:
        continue
    print("  generic executor differs: %d" % both(generic=True)[0].size)
    shapes = log[0][1]
    print("  delays", log[0][3], "radix", log[0][5], "n", len(shapes))
    for i, sh in enumerate(shapes):
        bd, _, aa, bb = both(only=i)
        if bd.size:
            ev = [x for x in log[1:] if x[2] == i]
            print("  voice %d shape %r: %d differ, first at sample %d (block %d, frame %d, ch %d): oracle %r gpu %r; events (call, k, voice, what, at): %s" % (
                i, sh, bd.size, bd[0], bd[0] // (2 * mbf), (bd[0] % (2 * mbf)) // 2, bd[0] % 2, aa[bd[0]], bb[bd[0]], ev))
