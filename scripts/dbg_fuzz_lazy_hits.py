"""debugging: how many seeds of test_chain_grammar.fuzz_grammar reach calls without a control kernel (chain plan + lazy records)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_chain_grammar as t  # noqa: E402
from fwapi import GpuEngine  # noqa: E402

hits = chain = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for seed in range(N):
    g = GpuEngine(max_block_frames=[128, 64, 256][seed % 3], max_batch=[64, 1, 3, 8][seed % 4])
    t.fuzz_grammar(g, seed)
    lazy = g.cx.lazy_stats()[0]
    chain += g.cx.plan_kind() == 2
    hits += lazy > 0
print("%d seeds: %d on the chain plan, %d with lazy calls" % (N, chain, hits))
