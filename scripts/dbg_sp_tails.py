"""debug: where does the spatialiser bank differ from the oracle (tests/test_gpu_parity.py, spatialiser tails test)?"""
import inspect, os, sys, textwrap
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fwapi, scenarios
import test_gpu_parity as T
src = inspect.getsource(T.test_spatialiser_waves_keep_their_histories_in_lds_across_consecutive_blocks)
body = src[src.index('    rng = np.random.default_rng(5)\n\n    def run(e):'):src.index('    rng = np.random.default_rng(5)\n    ro = run(')]
n_leaves, K = int(sys.argv[1]), int(sys.argv[2])
ns = dict(np=np, fwapi=fwapi, scenarios=scenarios, PLANAR_F32=fwapi.PLANAR_F32, n_leaves=n_leaves, K=K)
exec(textwrap.dedent(body), ns)
ro = ns['run'](T.oracle(max_block_frames=256))
ns['rng'] = np.random.default_rng(5)
g = T.GpuEngine(max_block_frames=256, max_batch=K)
rg = ns['run'](g)
a = ro.view(np.uint32).reshape(-1, 256 * 2); b = rg.view(np.uint32).reshape(-1, 256 * 2)
bad = np.nonzero((a != b).any(axis=1))[0]
print("blocks total", a.shape[0], "bad blocks", len(bad), "first", bad[:40])
if len(bad):
    k = bad[0]
    fr = np.nonzero(a[k] != b[k])[0]
    print("block", k, "bad samples", len(fr), "frames", sorted(set(fr // 2))[:20], "max abs diff", float(np.max(np.abs(ro.reshape(-1, 512)[k] - rg.reshape(-1, 512)[k]))))
    # pattern of bad blocks relative to call starts: calls of 3, K, K, 5
    starts = [0, 3, 3 + K, 3 + 2 * K]
    for kk in bad[:40]:
        c = max(i for i, s0 in enumerate(starts) if kk >= s0)
        print(" call", c, "block-in-call", kk - starts[c])
