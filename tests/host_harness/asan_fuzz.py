"""Host-side fuzz of libfwgpu's HOST half under AddressSanitizer + UBSan (CPU tier; tests/test_host_logic.py runs a few
seeds, `python tests/host_harness/asan_fuzz.py N` run by hand does more — 3000 seeds are clean).  The graph / message /
edit generators are the GPU fuzz families' own (tests/test_fuzz_gpu.py), driven on the host-only harness: no audio is
computed, the point is every plan build, group packing, message sort and buffer (re)allocation of the host translation units.
Must be started with LD_PRELOAD=libasan.so:libubsan.so (see run_sanitised)."""
import sys, os, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import fwapi
import firewheel_amd._lib as flib
# load the sanitised build in place of the regular harness
L = C.CDLL(os.environ['FWGPU_HOSTONLY_ASAN_SO'])
for name, (res, args) in flib.SIGNATURES.items():
    f = getattr(L, name); f.restype = res; f.argtypes = args
L.fwh_launch_count.restype = C.c_ulonglong; L.fwh_launch_count.argtypes = [C.c_int]; L.fwh_launch_reset.restype = None
fwapi._hostonly_lib = L
import test_fuzz_gpu as F
n = int(sys.argv[1])
for seed in range(n):
    pick = np.random.default_rng(10_000 + seed)
    mbf = int(pick.choice([64, 128, 256]))
    for mb in (1, 5, 64):
        F.fuzz_run(fwapi.HostOnlyEngine(max_block_frames=mbf, max_batch=mb), seed)
    F.fuzz_run(fwapi.HostOnlyEngine(max_block_frames=mbf, force_generic=True), seed)
    F.fuzz_dag(fwapi.HostOnlyEngine(max_block_frames=int(pick.choice([32, 64, 100, 128, 256])), max_batch=int(pick.choice([1, 2, 5, 64]))), seed)
    n_in = 1 + seed % 4
    F.fuzz_stream(fwapi.HostOnlyEngine(max_block_frames=mbf, num_graph_inputs=n_in, max_batch=int(pick.choice([1, 3, 64]))), seed, n_in)
L.fwh_violation.restype = C.c_char_p
assert L.fwh_violation() == b"", L.fwh_violation()
L.fwh_portints_selftest.restype = C.c_int
assert L.fwh_portints_selftest() == 0   # (the planner's port-list container: copies / moves across its inline / heap border, under ASan)
print("ok", n)
