"""Hostile-caller fuzz of the C ABI's host half under ASan + UBSan (CPU tier, host-only harness): random entry points with
random — mostly invalid — arguments (stale and made-up node / edge / sample ids, out-of-range ports and kinds, NaN and
huge parameters, far-future at_block, zero and odd frame counts), interleaved with valid graph building so that the
calls hit real state.  Every call must return (0 / a handle / a negative error) — never crash, never trip a sanitizer.
Start with LD_PRELOAD=libasan.so:libubsan.so and FWGPU_HOSTONLY_ASAN_SO (tests/test_host_logic.py does)."""
import ctypes as C
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import firewheel_amd._lib as flib  # noqa: E402

L = C.CDLL(os.environ["FWGPU_HOSTONLY_ASAN_SO"])
for name, (res, args) in flib.SIGNATURES.items():
    f = getattr(L, name)
    f.restype = res
    f.argtypes = args
fp = C.POINTER(C.c_float)
WEIRD_F = [0.0, -0.0, 1.0, -1.0, 100.0, 1e-30, 1e30, -1e30, float("inf"), float("-inf"), float("nan"), 48000.0, 0.5]
WEIRD_U = [0, 1, 2, 3, 63, 64, 65, 127, 128, 255, 256, 1 << 16, (1 << 31) - 1, 1 << 31, (1 << 32) - 1]


def run(seed):
    rng = random.Random(seed)
    mbf = rng.choice([1, 7, 64, 100, 256])
    c = L.fwgpu_ctx_create(0, rng.choice([8000, 44100, 48000, 192000]), mbf, rng.randint(0, 4), rng.randint(0, 4), None)
    assert c
    gin, gout = rng.randint(0, 4), rng.randint(0, 4)
    L.fwgpu_ctx_destroy(c)
    c = L.fwgpu_ctx_create(0, 48000, mbf, gin, gout, None)
    nodes = [L.fwgpu_graph_in_node(c), L.fwgpu_graph_out_node(c)]
    ports = {nodes[0]: (0, gin, 0), nodes[1]: (gout, 0, 0)}  # id -> (n_in, n_out, kind)
    edges, samples = [], []
    # shapes a kind activates with (volume.rs:63-65, sum.rs:27-29, ...): mostly used, so that updates succeed and the
    # later calls meet installed plans; the rest of the time anything goes
    SHAPES = {0: [(1, 1), (2, 2), (0, 2), (2, 0)], 1: [(0, 1), (0, 2)], 2: [(1, 1), (2, 2)], 3: [(4, 2), (6, 2), (2, 1), (8, 2), (64, 2)],
              4: [(0, 1), (0, 2)], 5: [(2, 2), (1, 1)], 6: [(1, 2)], 7: [(2, 1)], 8: [(2, 2)], 9: [(2, 2)],
              10: [(2, 2), (1, 1)], 11: [(2, 2), (1, 1)], 12: [(2, 2), (1, 1)], 13: [(0, 2), (0, 1)], 14: [(1, 2), (2, 2)]}
    keep = []

    def node():
        r = rng.random()
        if nodes and r < 0.8:
            return rng.choice(nodes)
        if r < 0.9:
            return rng.choice(nodes) ^ (rng.randint(1, 5) << 32) if nodes else 0  # stale generation
        return rng.choice([-1, 0, 5, 1 << 40, (1 << 63) - 1, -(1 << 62)])

    def u():
        return rng.choice(WEIRD_U) if rng.random() < 0.3 else rng.randint(0, 5)

    for step in range(rng.randint(40, 160)):
        op = rng.randint(0, 21)
        if op <= 3:
            kind = rng.randint(-1, 16) if rng.random() < 0.1 else rng.randint(0, 14)
            if kind in SHAPES and rng.random() < 0.85:
                n_in, n_out = rng.choice(SHAPES[kind])
            else:
                n_in, n_out = rng.choice([(2, 2), (0, 2), (1, 1), (4, 2), (5, 2), (2, 3), (u() % 70, u() % 70)])
            npar = rng.randint(0, 5)
            par = (C.c_float * max(npar, 1))(*[rng.choice(WEIRD_F + [float(s) for s in samples[:3]]) for _ in range(max(npar, 1))])
            r = L.fwgpu_add_node(c, kind, n_in, n_out, par, npar)
            if r >= 0:
                nodes.append(r)
                ports[r] = (n_in, n_out, kind)
        elif op <= 7:
            a, b = node(), node()
            sp = rng.randrange(ports[a][1]) if a in ports and ports[a][1] and rng.random() < 0.85 else u()
            dp = rng.randrange(ports[b][0]) if b in ports and ports[b][0] and rng.random() < 0.85 else u()
            r = L.fwgpu_connect(c, a, sp, b, dp, rng.randint(0, 1))
            if r >= 0:
                edges.append(r)
        elif op == 8:
            L.fwgpu_disconnect(c, node(), u(), node(), u())
        elif op == 9:
            L.fwgpu_disconnect_edge(c, rng.choice(edges) if edges and rng.random() < 0.7 else rng.choice([-1, 0, 1 << 35, 77]))
        elif op == 10:
            n = node()
            if L.fwgpu_remove_node(c, n) == 0 and n in nodes:
                nodes.remove(n)
        elif op <= 12:
            L.fwgpu_update(c)
            L.fwgpu_cycle_detected(c)
            L.fwgpu_plan_kind(c)
            L.fwgpu_plan_num_levels(c)
            buf = (C.c_int * 64)()
            L.fwgpu_plan_node_inputs_clear(c, node(), buf, rng.choice([0, 1, 64]))
            L.fwgpu_plan_node_level(c, node())
        elif op == 13:
            fmt, ch = rng.randint(-1, 6), rng.choice([0, 1, 2, 3, 6])
            frames = rng.choice([0, 1, 3, mbf, 5 * mbf + 3])
            data = (C.c_float * max(frames * max(ch, 1), 1))()
            keep.append(data)
            r = L.fwgpu_sample_create(c, fmt, ch, frames, C.cast(data, C.c_void_p))
            if r >= 0:
                samples.append(r)
        elif op == 14:
            L.fwgpu_sample_destroy(c, rng.choice(samples) if samples and rng.random() < 0.6 else rng.randint(-2, 40))
        elif op == 15:
            L.fwgpu_node_set_param(c, node(), rng.choice([0, 0, 0, 1, 2, 3, 4]) if rng.random() < 0.8 else rng.randint(-1, 6),
                                   rng.choice(WEIRD_F) if rng.random() < 0.5 else rng.uniform(-2, 200), rng.choice(WEIRD_U) if rng.random() < 0.3 else rng.randint(0, 3))
        elif op == 16:
            smp_nodes = [x for x in nodes if ports.get(x, (0, 0, -1))[2] == 4]
            n = rng.choice(smp_nodes) if smp_nodes and rng.random() < 0.7 else node()
            at = rng.choice(WEIRD_U) if rng.random() < 0.4 else rng.randint(0, 3)
            which = rng.randint(0, 5)
            if which == 0:
                L.fwgpu_sampler_set_sample(c, n, rng.choice(samples) if samples and rng.random() < 0.7 else rng.randint(-3, 50), rng.randint(0, 1), at)
            elif which == 1:
                L.fwgpu_sampler_play(c, n, at)
            elif which == 2:
                L.fwgpu_sampler_pause(c, n, at)
            elif which == 3:
                L.fwgpu_sampler_stop(c, n, at)
            elif which == 4:
                L.fwgpu_sampler_set_playhead_secs(c, n, rng.choice(WEIRD_F), at)
            else:
                L.fwgpu_sampler_set_loop_range(c, n, rng.randint(-1, 3), rng.choice(WEIRD_F), rng.choice(WEIRD_F), at)
        elif op == 17:
            L.fwgpu_set_max_batch(c, rng.choice([0, 1, 2, 5, 64, 300]))
            L.fwgpu_set_force_generic(c, rng.randint(0, 1))
        elif op <= 19:
            frames = rng.choice([0, 1, mbf - 1, mbf, mbf + 1, 3 * mbf, 7 * mbf + 2])
            n_in, n_out = rng.randint(0, 5), rng.randint(0, 5)
            inp = (C.c_float * max(frames * n_in, 1))()
            out = (C.c_float * max(frames * n_out, 1))()
            L.fwgpu_process_interleaved(c, inp if rng.random() < 0.8 else None, out, n_in, n_out, frames, 0.0, 0)
        elif op == 20:
            k = rng.choice([0, 1, 3, 70])
            n_out = rng.randint(0, 4)
            out = (C.c_float * max(k * mbf * max(n_out, 1), 1))()
            L.fwgpu_process_blocks_device(c, k, out, n_out)
            L.fwgpu_synchronize(c)
        else:
            n = node()
            frames = rng.choice([0, 1, mbf, mbf + 1])
            n_in, n_out = rng.randint(0, 3), rng.randint(0, 3)
            if n in ports and rng.random() < 0.8:
                n_in, n_out = ports[n][0], ports[n][1]
            bufs = [(C.c_float * max(frames, 1))() for _ in range(n_in + n_out)]
            ins = (fp * max(n_in, 1))(*[C.cast(b, fp) for b in bufs[:n_in]])
            outs = (fp * max(n_out, 1))(*[C.cast(b, fp) for b in bufs[n_in:]])
            om = C.c_uint64(0)
            L.fwgpu_node_process(c, n, frames, ins, n_in, outs, n_out, rng.getrandbits(64), C.byref(om), 0.0, 0)
    L.fwgpu_last_error(c)
    L.fwgpu_ctx_destroy(c)


if __name__ == "__main__":
    n = int(sys.argv[1])
    for seed in range(n):
        run(seed)
    L.fwh_violation.restype = C.c_char_p
    assert L.fwh_violation() == b"", L.fwh_violation()
    print("ok", n)
