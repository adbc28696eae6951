#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json on MI355X: stereo voice-samples/sec @ block=256, 48 kHz.

Workloads (BASELINE.json `configs`, concrete graphs from SURVEY.md §8d; built through the reference-shaped
edit API add_node / connect / update):

  cfg2 (default, the headline, configs[1]): 1024 stereo voices, sampler -> gain (VolumeNode) -> pan -> radix-32
        SumNode tree (32 + 1) -> graph_out, block = 256.  HBM-bound, 8 B per stereo voice-sample (k_leaf_sum).
  cfg3 (configs[2]): 4096 voices, sampler -> biquad LPF -> delay (feedback) -> gain -> sum tree (128 + 4 + 1),
        block = 512.  HBM-bound, 24 B per stereo voice-sample (k_chain).
  cfg4 (configs[3]): 256 voices, sampler -> 65536-tap stereo FIR -> sum tree, block = 256.  f32-MFMA-bound,
        262144 flop per stereo voice-sample (k_fir_gemm).
  cfg5 (configs[4], one GPU's shard): 8192 voices of the cfg2 chain, tree 256 + 8 + 1, block = 1024; with N > 1
        ranks the step ends with the mix-bus reduction over RCCL.
  cfg1 (configs[0], plumbing): beep -> gain -> stereo out, 750 one-block callbacks through the headless stream (the
        process_interleaved call pattern of firewheel-cpal's callback); reported under `other_configs`, never `value`.

Sources are planar f32, resident in HBM, each voice looping over its own buffer so every block streams fresh
HBM.  One "step" = one fwgpu_process_blocks_device call of `--blocks-per-step` consecutive blocks (the K-block
throughput mode); the interleaved mix bus stays in HBM.  With N > 1 every rank runs the same
shard (weak scaling, one process per GPU) and the step ends with the mix-bus collective.  `--gpus N` without a
launcher around it starts the N ranks itself (torch.distributed.run, 127.0.0.1).

Prints ONE compact JSON line (rank 0, <= 8 KiB, strict JSON); the whole record goes to gpurun_out/bench_full.json (named in the
line as `full`) and to stderr.  `roofline` times the dominant kernel with HIP events on the stream it runs on;
`cpu_baseline` times the oracle (C++ restatement of the reference's single-threaded executor) on the SAME graph and the
SAME source data, copied back from HBM; `parity_check` renders one call of the benched launch shape on a fresh context
and compares chosen blocks of it, bit for bit, with the oracle; `other_configs` carries configs 1, 3, 4 and 5 (one
shard) in short form; `realtime_us_per_callback` is the one-block-per-call latency of the headline graph.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md:35 (spec; 6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TF = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md:41 (dense f32 MFMA)

# workload -> (voices/GPU, block, blocks per step, source frames per voice, steps).  Blocks per step = the batch one
# fwgpu_process_blocks_device call renders (throughput mode: 768 x 256 frames = 4.1 s of audio per call for config 2);
# the fixed cost of a call (control kernel + upper sums, ~10 us) is amortised over it.
DEFAULTS = {
    "cfg2": (1024, 256, 768, 262144, 200),
    "cfg3": (4096, 512, 64, 65536, 60),
    "cfg4": (256, 256, 16, 65536, 30),
    "cfg5": (8192, 1024, 64, 65536, 60),
}
# node kinds of include/fwgpu.h (both engines take raw kinds)
K_BEEP, K_VOLUME, K_SUM, K_SAMPLER, K_HARD_CLIP, K_PAN, K_WIDTH, K_BIQUAD, K_DELAY, K_FIR, K_RESAMPLER, K_SPATIAL = 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14
PLANAR_F32, INTERLEAVED_I16 = 5, 0
# --rs-source: the resamplers' ratios are U(lo, hi); the driver line's is (0.5, 1.5).  FWGPU_BENCH_RS_RATIO=lo,hi: kernel experiments only
RS_RATIO = tuple(float(x) for x in os.environ.get("FWGPU_BENCH_RS_RATIO", "0.5,1.5").split(","))


# ------------------------------------------------------------------------------------------------ the two engines
class GpuSide(object):
    """the reference-shaped edit surface over the product's C ABI (firewheel_amd.FirewheelGpuCtx)"""

    def __init__(self, cx):
        from firewheel_amd.graph import _RawNode

        self.cx, self._raw = cx, _RawNode
        self.sample_rate = cx.sample_rate

    def add(self, kind, n_in, n_out, params=()):
        return self.cx.add_node(n_in, n_out, self._raw(kind, list(params)))

    def connect_stereo(self, a, b, port0=0):
        self.cx.connect(a, 0, b, port0, False)
        self.cx.connect(a, 1, b, port0 + 1, False)

    def out_node(self):
        return self.cx.graph_out_node()

    def update(self):
        self.cx.update()

    def start(self, sampler, sample, play=True):
        L, c = self.cx.L, self.cx.c
        self.cx._check(L.fwgpu_sampler_set_sample(c, sampler, sample, 0, 0))
        self.cx._check(L.fwgpu_sampler_set_loop_range(c, sampler, 1, 0.0, 0.0, 0))
        if play:
            self.cx._check(L.fwgpu_sampler_play(c, sampler, 0))

    def set_param(self, node, param, value, at_block=0):
        self.cx._check(self.cx.L.fwgpu_node_set_param(self.cx.c, node, param, value, at_block))


class OracleSide(object):
    """the same surface over the CPU oracle (tests/fwapi.OracleEngine) — cpu_baseline and parity_check only"""

    def __init__(self, block):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import fwapi

        self.e = fwapi.OracleEngine(max_block_frames=block)
        self.sample_rate = self.e.sample_rate

    def add(self, kind, n_in, n_out, params=()):
        return self.e.add_node(kind, n_in, n_out, list(params))

    def connect_stereo(self, a, b, port0=0):
        self.e.connect_stereo(a, b, port0)

    def out_node(self):
        return self.e.graph_out_node

    def update(self):
        self.e.update()

    def start(self, sampler, sample, play=True):
        self.e.sampler_set_sample(sampler, sample)
        self.e.sampler_set_loop_range(sampler, 1)
        if play:
            self.e.sampler_play(sampler)

    def set_param(self, node, param, value, at_block=0):
        self.e.set_param(node, param, value)


# ------------------------------------------------------------------------------------------------ the graphs
def sum_tree(e, ends, radix, master=False, connect_out=True):
    """radix-`radix` SumNode tree over `ends`; returns the node that feeds graph_out (left unconnected with connect_out=False:
    a shard of a whole graph whose top node is the mix-bus SumNode)"""
    level = ends
    first_level = None
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.add(K_SUM, 2 * len(grp), 2)
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            nxt.append(m)
        level = nxt
        first_level = first_level or nxt
        if len(level) == 1:
            break
    cur = level[0]
    if master == "send":  # --send: every fourth leaf bus is ALSO tapped into a send bus -> return gain -> width -> limiter that
        # joins the root in a two-port sum: buses consumed twice — no fused shape as a whole, the voice banks inside still
        # are (hybrid plan).  (No IIR on the return: a bus biquad / delay is one serial recurrence over the 4.1 s of a
        # 768-block call — ~2 ms on one lane whatever surrounds it, DESIGN.md §3.2 — and would be all this line measures.)
        taps = first_level[::4][:32]  # (a SumNode takes at most 64 channels)
        send = e.add(K_SUM, 2 * len(taps), 2)
        for p, n in enumerate(taps):
            e.connect_stereo(n, send, 2 * p)
        ret = e.add(K_VOLUME, 2, 2, [45.0])
        wid = e.add(K_WIDTH, 2, 2, [1.4])
        lim = e.add(K_HARD_CLIP, 2, 2, [-1.0])
        e.connect_stereo(send, ret)
        e.connect_stereo(ret, wid)
        e.connect_stereo(wid, lim)
        mix = e.add(K_SUM, 4, 2)
        e.connect_stereo(cur, mix, 0)
        e.connect_stereo(lim, mix, 2)
        cur = mix
    elif master:  # --master: a master volume + limiter between the root SumNode and graph_out (--master-iir: low-pass + delay)
        for kind, params in (((K_BIQUAD, [0.0, 9000.0, 0.707]), (K_DELAY, [0.030, 0.2, 0.3])) if master == "iir" else
                             ((K_VOLUME, [70.0]), (K_HARD_CLIP, [-1.0]))):
            m = e.add(kind, 2, 2, params)
            e.connect_stereo(cur, m)
            cur = m
    if connect_out:
        e.connect_stereo(cur, e.out_node())
    return cur


def graph_bank(e, voices, radix, seed=0, master=False, extra=(), rs_samples=None, connect_out=True):
    """cfg2 / cfg5 voice: sampler -> gain -> pan [-> width -> hard clip with --voice-fx].  Returns (samplers, volumes).
    rs_samples: the voices' sources are SPEC resamplers (looping, ratio U(0.5, 1.5)) on these sample ids instead of samplers."""
    import numpy as np

    rng = np.random.default_rng(1234 + seed)
    rng_ratio = np.random.default_rng(777 + seed)
    ends, samplers, volumes = [], [], []
    for v in range(voices):
        if rs_samples is not None:
            s = e.add(K_RESAMPLER, 0, 2, [float(rs_samples[v]), float(rng_ratio.uniform(*RS_RATIO)), 1.0, 1.0])
        else:
            s = e.add(K_SAMPLER, 0, 2, [100.0])
        vol = e.add(K_VOLUME, 2, 2, [float(rng.uniform(10, 100))])
        pan = e.add(K_PAN, 2, 2, [float(rng.uniform(-1, 1))])
        e.connect_stereo(s, vol)
        e.connect_stereo(vol, pan)
        cur = pan
        for kind, params in extra:
            n = e.add(kind, 2, 2, params)
            e.connect_stereo(cur, n)
            cur = n
        samplers.append(s)
        volumes.append(vol)
        ends.append(cur)
    root = sum_tree(e, ends, radix, master, connect_out)
    if connect_out:
        e.update()
    return samplers, volumes, root


def graph_chain(e, voices, radix, seed=0, master=False, connect_out=True, reordered=False):
    """cfg3 voice: sampler -> biquad LPF (cutoff U(200, 8000) Hz, Q 0.707) -> delay (U(10, 250) ms, feedback 0.3,
    mix 0.5) -> gain.  reordered (round 6, the chain plan's wider grammar): sampler -> gain -> biquad LPF -> biquad
    (a peaking band-pass beside it: an EQ cascade) -> delay -> pan."""
    import numpy as np

    rng = np.random.default_rng(4321 + seed)
    ends, samplers = [], []
    # (FWGPU_BENCH_CHAIN_SHAPE: kernel experiments only — v gain, B / b the two biquads, D delay, p pan, in any accepted order)
    shape = os.environ.get("FWGPU_BENCH_CHAIN_SHAPE", "vBbDp")
    for v in range(voices):
        s = e.add(K_SAMPLER, 0, 2, [100.0])
        if reordered:
            par = dict(B=(K_BIQUAD, [0.0, float(rng.uniform(200, 8000)), 0.707]), D=(K_DELAY, [float(rng.uniform(0.010, 0.250)), 0.3, 0.5]),
                       v=(K_VOLUME, [float(rng.uniform(10, 100))]), b=(K_BIQUAD, [2.0, float(rng.uniform(300, 5000)), 1.2]),
                       p=(K_PAN, [float(rng.uniform(-1, 1))]))  # (every voice draws all five, whatever the shape: one stream of parameters)
            cur = s
            for t in shape:
                n = e.add(par[t][0], 2, 2, par[t][1])
                e.connect_stereo(cur, n)
                cur = n
            samplers.append(s)
            ends.append(cur)
            continue
        bq = e.add(K_BIQUAD, 2, 2, [0.0, float(rng.uniform(200, 8000)), 0.707])
        dl = e.add(K_DELAY, 2, 2, [float(rng.uniform(0.010, 0.250)), 0.3, 0.5])
        vol = e.add(K_VOLUME, 2, 2, [float(rng.uniform(10, 100))])
        e.connect_stereo(s, bq)
        e.connect_stereo(bq, dl)
        e.connect_stereo(dl, vol)
        samplers.append(s)
        ends.append(vol)
    root = sum_tree(e, ends, radix, master, connect_out)
    if connect_out:
        e.update()
    return samplers, [], root


def reverb_ir(taps):
    import numpy as np

    n = np.arange(taps, dtype=np.float64)
    rng = np.random.default_rng(4)
    h = (rng.uniform(-1, 1, size=(2, taps)) * np.exp(-n / 16384.0)[None, :])
    return (h / np.abs(h).sum(axis=1, keepdims=True)).astype(np.float32)


def graph_reverb(e, voices, radix, ir_sample, connect_out=True):
    """cfg4: V x (sampler -> `taps`-tap stereo FIR convolution) -> radix sum tree -> out."""
    ends, samplers = [], []
    for v in range(voices):
        s = e.add(K_SAMPLER, 0, 2, [100.0])
        f = e.add(K_FIR, 2, 2, [float(ir_sample)])
        e.connect_stereo(s, f)
        samplers.append(s)
        ends.append(f)
    root = sum_tree(e, ends, radix, False, connect_out)
    if connect_out:
        e.update()
    return samplers, [], root


def build_graph(e, wl, voices, radix, seed, master, ir_sample=None, voice_fx=False, rs_samples=None, connect_out=True):
    """-> (samplers, volumes, root).  connect_out=False: the shard's tree is left unconnected (and not compiled)"""
    if wl == "cfg4":
        return graph_reverb(e, voices, radix, ir_sample, connect_out)
    if wl == "cfg3":
        return graph_chain(e, voices, radix, seed, master, connect_out, reordered=voice_fx == "reordered")
    extra = ((K_WIDTH, [1.3]), (K_HARD_CLIP, [-3.0])) if voice_fx is True else ()
    if voice_fx == "spatial":  # a SPEC 3D spatialiser at the end of every voice (2 -> 2: mono sum, ITD, distance + pan gains)
        extra = ((K_SPATIAL, [2.0, 0.5, -3.0]),)
    return graph_bank(e, voices, radix, seed, master, extra, rs_samples, connect_out)


def want_plan(wl, force_generic):
    return 0 if (force_generic or wl == "cfg4") else (2 if wl == "cfg3" else 1)


def make_gpu(fa, wl, V, B, K, radix, src, F, sfmt, seed, args, stream, device):
    """a fresh device context with the workload's graph on the sources `src` (HBM), every voice looping and playing"""
    cx = fa.FirewheelGpuCtx(48000, B, 0, 2, device=device, stream=stream)
    cx.set_max_batch(K)
    if args.force_generic:
        cx.set_force_generic(True)
    g = GpuSide(cx)
    ir = cx.new_sample(PLANAR_F32, 2, reverb_ir(args.taps)) if wl == "cfg4" else None
    fmt = PLANAR_F32 if sfmt == "f32" else INTERLEAVED_I16
    rs = getattr(args, "rs_source", False) and wl in ("cfg2", "cfg5")
    ids = [cx.new_sample_device(fmt, 2, F, src[v].data_ptr()) for v in range(V)] if rs else None
    samplers, volumes, _ = build_graph(g, wl, V, radix, seed, "send" if getattr(args, "send", False) else ("iir" if getattr(args, "master_iir", False) else args.master), ir,
                                       "reordered" if (getattr(args, "chain_reordered", False) and wl == "cfg3") else
                                       ("spatial" if getattr(args, "voice_spatial", False) else args.voice_fx), ids)
    if not rs:
        for v, s in enumerate(samplers):
            smp = cx.new_sample_device(fmt, 2, F, src[v].data_ptr())
            g.start(s, smp)
    generic = args.force_generic  # (round 3: a sampler -> volume -> spatialiser voice is a voice-bank shape, SK_SPATIAL)
    want = 3 if (getattr(args, "send", False) and not generic and wl in ("cfg2", "cfg3", "cfg5")) else want_plan(wl, generic)
    # (cfg4: the samplers in front of the FIR nodes are solo voices of the hybrid plan from 8 voices on — round 4; FWGPU_SOLO=0: levels only)
    ok = cx.plan_kind() == want or (wl == "cfg4" and not generic and cx.plan_kind() == 3)
    assert ok, "expected launch plan %d, got %d" % (want, cx.plan_kind())
    return cx, g, samplers, volumes


def make_oracle(wl, V, B, radix, seed, args, host_src, fmt=PLANAR_F32):
    """the oracle on the same graph; host_src[v] = that voice's sample data as the engine's format wants it"""
    o = OracleSide(B)
    ir = o.e.new_sample(PLANAR_F32, 2, reverb_ir(args.taps)) if wl == "cfg4" else None
    rs = getattr(args, "rs_source", False) and wl in ("cfg2", "cfg5")
    ids = [o.e.new_sample(fmt, 2, host_src[v]) for v in range(V)] if rs else None
    samplers, volumes, _ = build_graph(o, wl, V, radix, seed, "send" if getattr(args, "send", False) else ("iir" if getattr(args, "master_iir", False) else args.master), ir,
                                       "reordered" if (getattr(args, "chain_reordered", False) and wl == "cfg3") else
                                       ("spatial" if getattr(args, "voice_spatial", False) else args.voice_fx), ids)
    if not rs:
        for v, s in enumerate(samplers):
            o.start(s, o.e.new_sample(fmt, 2, host_src[v]))
    return o, samplers, volumes


def make_oracle_whole(wl, world, V, B, radix, args, host_srcs):
    """The WHOLE graph of an N-rank run as the reference would express it: the N shards (rank r's graph is built with seed r,
    as its process builds it) under ONE top-level N-port stereo SumNode (nodes/sum.rs:41-136) -> graph_out.  host_srcs[r][v] =
    voice v of rank r's sample data."""
    o = OracleSide(B)
    roots, starts = [], []
    for r in range(world):
        samplers, _, root = build_graph(o, wl, V, radix, r, args.master, None, args.voice_fx, None, connect_out=False)
        roots.append(root)
        starts.append(samplers)
    top = o.add(K_SUM, 2 * world, 2)
    for p, root in enumerate(roots):
        o.connect_stereo(root, top, 2 * p)
    o.connect_stereo(top, o.out_node())
    o.update()
    for r in range(world):
        for v, smp in enumerate(starts[r]):
            o.start(smp, o.e.new_sample(PLANAR_F32, 2, host_srcs[r][v]))
    return o


# ------------------------------------------------------------------------------------------------ CPU baseline
def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline(wl, V, B, radix, seed, args, src, F, target_secs):
    """Oracle on the SAME graph and the SAME source data as the timed GPU run (copied back from HBM).  `value` is the
    faithful figure: ONE thread, like the reference's audio thread (DESIGN_DOC.md:48).  `all_cores` is the generous one
    (SURVEY §8d): the voices split over host cores, one oracle engine per thread, no mix-bus exchange charged."""
    note = "same graph, same %d-frame looping sources as the GPU run (copied from HBM)" % F
    voices = V
    if wl == "cfg4":
        voices = min(V, 32)  # bounded sample: the scalar direct-form convolution is ~1e9 fmaf per voice-block
        note = "the first %d of the GPU run's %d voices (scalar 65536-tap convolution: 27 ms per voice-block), same sources" % (voices, V)
    Fh = F
    if voices * F * 8 > 6 * 2 ** 30:  # keep the host copy bounded
        Fh = int(6 * 2 ** 30 // (voices * 8) // B * B)
        note = "same graph; sources = the first %d of the GPU run's %d frames per voice (host copy bounded to 6 GiB)" % (Fh, F)
    host = src[:voices, :, :Fh].cpu().numpy()
    chunk = {"cfg3": 2, "cfg4": 1}.get(wl, 16)
    o, _, _ = make_oracle(wl, voices, B, radix, seed, args, host)
    if wl != "cfg4":
        o.e.process_blocks(4)  # warm-up
    n_blocks, t = 0, 0.0
    t0 = time.perf_counter()
    while t < target_secs:
        o.e.process_blocks(chunk)
        n_blocks += chunk
        t = time.perf_counter() - t0
    del o
    ncpu = host_cores()
    all_cores = None
    T = min(ncpu, voices)  # every host core gets a slice of the voices: one engine + one NATIVE thread each (libfw_oracle's own
    if T > 1:              # std::thread loop — round 2 drove 32 Python threads through ctypes and left 7/8 of the box idle)
        import fwapi

        per = max(1, voices // T)
        # the voice parameters of a slice differ from the whole graph's (the seeded stream restarts): a throughput figure
        engines = [make_oracle(wl, per, B, radix, 1 + i, args, host[i * per:(i + 1) * per])[0] for i in range(T)]
        ch = chunk * 8
        secs = max(2.0, target_secs / 3.0)
        done, tb = fwapi.oracle_process_parallel([x.e for x in engines], ch * B, secs)
        all_cores = {"value": per * B * ch * sum(done) / tb, "unit": "voice-samples/s", "cores": T,
                     "sample": "%d native threads x %d voices, %.1f s, calls of %d blocks; no mix-bus exchange charged" % (T, per, tb, ch)}
        del engines
    return {
        "value": voices * B * n_blocks / t,
        "unit": "voice-samples/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d blocks of the %d-voice %s graph (block=%d), %.1f s on 1 of %d host cores; %s; oracle = C++ restatement of "
                  "the reference's single-threaded executor (the Rust build is not available: no cargo/rustc)"
                  % (n_blocks, voices, wl, B, t, ncpu, note),
        "all_cores": all_cores,
    }


# ------------------------------------------------------------------------------------------------ parity check
def parity_check(fa, torch, wl, V, B, K, radix, seed, args, src, F, sfmt, stream, device):
    """One call of the benched launch shape (all voices, K blocks) on a FRESH context, chosen blocks of it compared bit for
    bit with the oracle on the same graph and the same source frames.  cfg2 / cfg5 voices carry no history but the
    playhead, so the oracle is handed the source frames of blocks {0, 1, K/2, K-1} back to back and renders those four;
    cfg3 (filter + delay state) is checked on the call's first two blocks, cfg4 on its first block."""
    import numpy as np

    t0 = time.perf_counter()
    cx, g, samplers, _ = make_gpu(fa, wl, V, B, K, radix, src, F, sfmt, seed, args, stream, device)
    out = torch.empty(K * B * 2, dtype=torch.float32, device=src.device)
    if src.is_cuda:
        torch.cuda.synchronize()
    cx.process_blocks_device(K, out.data_ptr(), 2)
    cx.synchronize()
    rs = getattr(args, "rs_source", False) and wl in ("cfg2", "cfg5")
    if rs:
        blocks = [0, 1]  # a resampled voice's block does not start on a source-block boundary: prefix only, whole samples
    elif getattr(args, "voice_spatial", False) or getattr(args, "master_iir", False):
        blocks = [0, 1, 2, 3][:K]  # a spatialiser carries 64 frames of history from block to block: a contiguous prefix
    elif wl in ("cfg2", "cfg5") and K >= 4 and F >= K * B and sfmt == "f32":
        blocks = [0, 1, K // 2, K - 1]
    elif wl == "cfg4":
        blocks = [0]
    else:
        blocks = [0, 1] if K >= 2 else [0]
    got = torch.cat([out[b * B * 2:(b + 1) * B * 2] for b in blocks]).cpu().numpy()
    if rs:  # (looping resamplers wrap their 16-tap window around the sample's end: the oracle needs whole samples)
        host = src.cpu().numpy()
        fmt = PLANAR_F32 if sfmt == "f32" else INTERLEAVED_I16
    elif sfmt == "f32":
        host = torch.cat([src[:, :, b * B:(b + 1) * B] for b in blocks], dim=2).cpu().numpy()
        fmt = PLANAR_F32
    else:
        host = torch.cat([src[:, b * B:(b + 1) * B, :] for b in blocks], dim=1).cpu().numpy()
        fmt = INTERLEAVED_I16
    o, _, _ = make_oracle(wl, V, B, radix, seed, args, host, fmt)
    ref = o.e.process_blocks(len(blocks))
    same = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
    res = {"bit_exact": same, "blocks": len(blocks), "block_indices": blocks, "voices": V, "blocks_per_call": K,
           "samples_compared": int(got.size), "launch_plan": cx.plan_kind(),
           "against": "oracle (C++ restatement of the reference), same graph, same source frames",
           "secs": None}
    if not same:
        bad = np.nonzero(got.view(np.uint32) != ref.view(np.uint32))[0]
        res["mismatches"] = int(bad.size)
        res["first_mismatch"] = [int(bad[0]), float(got[bad[0]]), float(ref[bad[0]])]
    if not bool(np.any(ref)):
        res["bit_exact"] = False
        res["error"] = "the reference output is all zeros: nothing was compared"
    cx.close()
    res["voices_checked"], res["blocks_checked"] = V, len(blocks)
    carried = rs or getattr(args, "voice_spatial", False)  # resampler positions / spatialiser histories carry from block to block too
    if (wl in ("cfg3", "cfg4") or (carried and wl == "cfg2" and sfmt == "f32")) and not (getattr(args, "send", False) or args.master or getattr(args, "master_iir", False)) and K >= 2:
        # filter / delay / FIR history carries from block to block, so blocks deep inside the call need their whole prefix from
        # the oracle: a SLICE of the voices (same launch shape, same K), every block of the call
        try:
            res["deep"] = parity_deep(fa, torch, wl, B, K, radix, seed, args, src, F, stream, device, sfmt)
            res["bit_exact"] = bool(res["bit_exact"] and res["deep"]["bit_exact"])
        except Exception as ex:  # noqa: BLE001
            res["deep"] = {"error": repr(ex)}
    if wl in ("cfg3", "cfg4"):
        # (say what this in-line check is and is not)
        res["scope"] = ("a smoke check inside the bench run: blocks %s of all %d voices + `deep` = a SLICE of the voices over every block of one call; the "
                        "full-size comparisons are GPU tests (tests/test_gpu_benched_shapes.py: config 3 at 4 096 voices x 64 blocks, config 4 at "
                        "65 536 taps) — those compare a PREFIX of the voices against the oracle too: the scalar oracle needs ~1 s per voice-second") % (blocks, V)
    res["secs"] = round(time.perf_counter() - t0, 2)
    return res


def parity_deep(fa, torch, wl, B, K, radix, seed, args, src, F, stream, device, sfmt="f32"):
    """cfg3: 256 voices x ALL K = 64 blocks of one call (biquad + delay state through the whole call: blocks 31 and 63 are as
    checked as block 0); cfg4: 8 voices x all K = 16 blocks of one call at 65 536 taps (block 15 reads 15 blocks of FIR history
    written by the call itself; ~3.5 s of scalar oracle)."""
    import numpy as np

    t0 = time.perf_counter()
    rs = getattr(args, "rs_source", False) and wl == "cfg2"
    # (cfg2 with resampler sources / spatialiser stages: 64 voices — two full leaves — x all 768 blocks of one call; VERDICT r3: the
    #  fresh-context check looked at blocks {0, 1} of 768 only)
    Vd = min(src.shape[0], 256 if wl == "cfg3" else (64 if wl == "cfg2" else 8))
    if getattr(args, "voice_spatial", False) and wl == "cfg2":
        Vd = src.shape[0]  # every leaf: only then does a spatialiser wave take several consecutive blocks and keep its histories in LDS
    assert F >= K * B
    cx, g, samplers, _ = make_gpu(fa, wl, Vd, B, K, radix, src, F, sfmt, seed, args, stream, device)
    out = torch.empty(K * B * 2, dtype=torch.float32, device=src.device)
    torch.cuda.synchronize()
    cx.process_blocks_device(K, out.data_ptr(), 2)
    cx.synchronize()
    got = out.cpu().numpy()
    # (a looping resampler reads past K x B source frames at ratios > 1 and wraps its window around the sample's end: whole samples)
    if sfmt == "i16":  # interleaved stereo PCM: [voice][frame][channel]
        o, _, _ = make_oracle(wl, Vd, B, radix, seed, args, src[:Vd, :K * B, :].cpu().numpy(), INTERLEAVED_I16)
    else:
        o, _, _ = make_oracle(wl, Vd, B, radix, seed, args, (src[:Vd] if rs else src[:Vd, :, :K * B]).cpu().numpy())
    ref = o.e.process_blocks(K)
    same = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32))) and bool(np.any(ref))
    per_block = (got.view(np.uint32).reshape(K, -1) == ref.view(np.uint32).reshape(K, -1)).all(axis=1)
    res = {"bit_exact": same, "voices_checked": Vd, "blocks_checked": K, "block_indices": "0..%d (every block of the call)" % (K - 1),
           "blocks_per_call": K, "samples_compared": int(got.size), "launch_plan": cx.plan_kind(),
           "first_bad_block": None if same else int(np.argmin(per_block)), "secs": None}
    cx.close()
    res["secs"] = round(time.perf_counter() - t0, 2)
    return res


def shard_sources(torch, shard, r, V, F, dev, stagger=0):
    """rank r's synthetic sources: uniform(-1, 1) f32, the generator seeded by the shard's first GLOBAL voice id — the same call
    on any rank of the same hardware gives the same bytes, which is how rank 0 hands the oracle every shard's inputs"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(shard.voice_seed(r * V))
    pitch = 2 * F + stagger
    src = torch.empty(V * pitch, dtype=torch.float32, device=dev).as_strided((V, 2, F), (pitch, F, 1))
    src.uniform_(-1.0, 1.0, generator=gen)
    return src


def parity_check_multi(env, args, wl, V, B, K, F, src, stream, device, mode):
    """N > 1: one call of the benched launch shape on a fresh context on EVERY rank, the partial buses (+ silence flags) through
    the run's own mix-bus reduction, and rank 0's reduced bus compared with the oracle's WHOLE graph — the N shards under one
    N-port SumNode (nodes/sum.rs:41-136) — on blocks {0, 1, K/2, K-1} (cfg3: {0, 1}).  Bit-exact in the rank-ordered modes
    (exchange, ordered); the all-reduce re-associates the f32 sum for N > 2, so there the bar is |got - ref| <= 1e-6 x the sum
    of the shards' |partial| per sample.  Collective: every rank calls it."""
    import numpy as np

    torch, fa, shard, dist = env["torch"], env["fa"], env["shard"], env["dist"]
    rank, world, dev, hostonly = env["rank"], env["world"], env["dev"], env["hostonly"]
    t0 = time.perf_counter()
    progress(env, "%s: parity (whole graph, %d ranks, mode %s)" % (wl, world, mode))
    cx, g, samplers, _ = make_gpu(fa, wl, V, B, K, args.radix, src, F, "f32", rank, args, stream, device)
    n = K * B * 2
    part = [torch.zeros(n, dtype=torch.float32, device=dev)]
    sil = [torch.zeros(K * 2, dtype=torch.uint8, device=dev)]
    red = [torch.zeros(n, dtype=torch.float32, device=dev)]
    reducer, used, note = make_reducer(env, args, cx, part, sil, B, mode=mode, reds=red)
    if not hostonly:
        torch.cuda.synchronize()
    cx.process_blocks_device_flags(K, part[0].data_ptr(), 2, sil[0].data_ptr())
    cx.synchronize()
    mine = part[0].clone()  # (the all-reduce works in place)
    reducer.submit(0)
    out = reducer.wait(0)
    reducer.wait_all()
    cx.synchronize()
    if not hostonly:
        torch.cuda.synchronize()
    blocks = sorted(set([0, 1, K // 2, K - 1])) if (wl in ("cfg2", "cfg5") and K >= 4 and F >= K * B) else ([0, 1] if K >= 2 else [0])
    pick = lambda t: torch.cat([t[b * B * 2:(b + 1) * B * 2] for b in blocks]).cpu().numpy()
    got = pick(out)
    parts = [None] * world
    dist.all_gather_object(parts, pick(mine))  # the shards' own partial buses on the compared blocks: the tolerance's scale
    flags = [None] * world
    dist.all_gather_object(flags, sil[0].cpu().numpy().reshape(K, 2)[blocks])
    progress(env, "%s: parity: device side done, the oracle's whole graph next (rank 0)" % wl)
    res = None
    if rank == 0:
        host_srcs = []
        for r in range(world):
            sr = src if r == 0 else shard_sources(torch, shard, r, V, F, dev, args.src_stagger)
            if r == 0:  # the regeneration is what it claims to be: rank 0's own shard comes out identical
                again = shard_sources(torch, shard, 0, V, F, dev, args.src_stagger)
                assert bool(torch.equal(again, src)), "source regeneration is not deterministic"
                del again
            host_srcs.append(torch.cat([sr[:, :, b * B:(b + 1) * B] for b in blocks], dim=2).cpu().numpy())
            del sr
        o = make_oracle_whole(wl, world, V, B, args.radix, args, host_srcs)
        ref = o.e.process_blocks(len(blocks))
        same = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
        scale = np.sum([np.abs(p) for p in parts], axis=0)
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        tol_ok = bool(np.all(err <= 1e-6 * scale + 1e-30))
        res = {"bit_exact": same, "within_tolerance": tol_ok, "tolerance": "1e-6 x sum over ranks of |partial bus| per sample",
               "max_abs_err": float(err.max()), "max_err_over_scale": float(np.max(err / np.maximum(scale, 1e-30))),
               "ranks": world, "bus_reduce": used, "bus_reduce_fallback": note, "blocks": len(blocks), "block_indices": blocks,
               "voices": V * world, "voices_per_rank": V, "blocks_per_call": K, "samples_compared": int(got.size),
               "silent_flags_seen": int(sum(int(f.sum()) for f in flags)), "launch_plan": cx.plan_kind(),
               "against": "oracle (C++ restatement of the reference): the WHOLE graph, %d shards under one %d-port SumNode, same source frames" % (world, world),
               "expected": "bit_exact" if used in ("exchange", "ordered", "ordered_abi") or world <= 2 else "within_tolerance"}
        if hostonly:
            res = {"skipped": "host-only harness: no audio computed", "ranks": world, "bus_reduce": used, "oracle_whole_graph_nonzero": bool(np.any(ref))}
        elif not bool(np.any(ref)):
            res["bit_exact"] = res["within_tolerance"] = False
            res["error"] = "the reference output is all zeros: nothing was compared"
        res["secs"] = round(time.perf_counter() - t0, 2)
    if hasattr(reducer, "close"):
        reducer.close(dist)
    else:
        dist.barrier()
    cx.close()
    return res


# ------------------------------------------------------------------------------------------------ realtime / cfg1
def realtime_probe(cx, B, callbacks=1500):
    """one max_block_frames block per fwgpu_stream_callback, output to pageable host memory, synchronous — the call
    pattern of the reference's backend callback (firewheel-cpal/src/lib.rs:378-449).  Returns microseconds per callback:
    (the backend thread's loop inside the library — fwgpu_stream_run: what a native host pays —, the same callbacks driven
    one by one from Python through ctypes)."""
    st = cx.open_stream(0, 2)
    st.run(B, 50, 0.0)
    _, dt_native = st.run(B, callbacks, 50 * B / 48000.0)
    t = (50 + callbacks) * B / 48000.0
    t0 = time.perf_counter()
    for _ in range(callbacks):
        t += B / 48000.0
        st.callback(B, t)
    dt = time.perf_counter() - t0
    st.close()
    return dt_native / callbacks * 1e6, dt / callbacks * 1e6


def run_cfg1(fa, stream, device):
    """BASELINE configs[0]: beep_test — sine -> gain -> stereo out, block 256 @ 48 kHz, 750 callbacks (4 s) of one block
    each through the headless stream (examples/beep_test/src/main.rs:10-52, cpal/lib.rs:429-437), checked against the
    oracle within BeepTest's libm tolerance (2e-6 absolute: ocml sinf vs glibc sinf)."""
    import numpy as np

    B, n = 256, 750

    def build(e):
        beep = e.add(K_BEEP, 0, 2, [440.0, -12.0, 1.0])
        vol = e.add(K_VOLUME, 2, 2, [80.0])
        e.connect_stereo(beep, vol)
        e.connect_stereo(vol, e.out_node())
        e.update()

    cx = fa.FirewheelGpuCtx(48000, B, 0, 2, device=device, stream=stream)
    build(GpuSide(cx))
    st = cx.open_stream(0, 2)
    outs = []
    t0 = time.perf_counter()
    for i in range(n):
        o, _ = st.callback(B, (i + 1) * B / 48000.0)
        outs.append(o)
    dt = time.perf_counter() - t0
    got = np.concatenate(outs)
    o = OracleSide(B)
    build(o)
    ref = np.concatenate([o.e.process_blocks(1) for _ in range(n)])
    err = float(np.max(np.abs(got - ref)))
    cbs, unders, _ = st.stats()
    st.close()
    cx.close()
    return {"workload": "cfg1 beep_test: sine 440 Hz -> gain -> stereo out, block=256, %d callbacks of one block (headless stream)" % n,
            "callbacks": cbs, "underflows": unders, "us_per_callback": dt / n * 1e6, "realtime_factor": (n * B / 48000.0) / dt,
            "max_abs_err_vs_oracle": err, "within_tolerance": bool(err <= 2e-6), "tolerance": 2e-6}


# ------------------------------------------------------------------------------------------------ measured run
def pmc_traffic(kernel, V, B, K, name):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
    gfx950 corrections per the microarch guide) — quoted only when collected on this exact workload."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic*.json")), reverse=True):
        try:
            pm = json.load(open(path))
            w = pm["workload"]
            if (w["voices"], w["block"], w["blocks_per_step"]) == (V, B, K) and kernel in pm and w.get("name", name) == name:
                return pm[kernel]["traffic_bytes"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


class ExchangeFailed(RuntimeError):
    """raised on EVERY rank when some rank's exchange timed out during a run"""


REDUCE_DESC = {"exchange": "one-shot exchange over peer-mapped slots (fwgpu_bus_exchange, rank-ordered: bit-exact)",
               "ordered": "RCCL all-gather + rank-ordered sum kernel (bit-exact)", "allreduce": "RCCL all-reduce",
               "ordered_abi": "RCCL all-gather + rank-ordered sum through the C ABI (fwgpu_bus_allgather_ordered: bit-exact)",
               "allreduce_abi": "RCCL all-reduce through the C ABI (fwgpu_bus_allreduce_rccl)", None: "none"}


def make_reducer(env, args, cx, outs, sils, B, mode=None, reds=None):
    """the mix-bus reduction of an N > 1 run -> (reducer, mode used, fallback note).  `exchange` (default) needs dmabuf IPC and
    peer access between the devices; when any rank cannot set it up, ALL ranks fall back to the RCCL all-reduce and the line
    says so."""
    torch, shard, dist = env["torch"], env["shard"], env["dist"]
    mode = mode or args.bus_reduce
    note = None
    if mode == "exchange":
        reds = reds or [torch.empty_like(o) for o in outs]
        try:
            return shard.ExchangeReducer(dist, outs, cx, reds, sils, B, 2), "exchange", None
        except RuntimeError as ex:
            if env.get("share_device"):
                raise  # ranks sharing one device have no RCCL to fall back to
            note = "%s -> fell back to the RCCL all-reduce" % ex
            mode = "allreduce"
    if env.get("share_device"):
        raise SystemExit("bench.py --share-device: RCCL refuses two ranks on one device; only --bus-reduce exchange runs there")
    if mode in ("ordered_abi", "allreduce_abi"):  # libfwgpu's own RCCL calls (what a host bound to include/fwgpu.h gets); gloo carries the id
        if env["hostonly"]:
            raise SystemExit("bench.py: the C ABI's RCCL modes need librccl (tests/test_rccl_abi.py covers them on the host harness)")
        reds = reds or [torch.empty_like(o) for o in outs]
        r = shard.AbiRcclReducer(dist, outs, cx, mode, reds if mode == "ordered_abi" else None, sils, B, 2)
        env["rccl_ranks_seen"] = dist.get_world_size()
        return r, mode, note
    grp = rccl_group(env)
    if mode == "ordered":
        return shard.BusReducer(dist, outs, "ordered", group=grp, cx=cx, sils=sils, frames=B, n_ch=2), "ordered", note
    return shard.BusReducer(dist, outs, "allreduce", group=grp, cx=cx), "allreduce", note


def rccl_group(env):
    """the RCCL (backend "nccl") group over all ranks, created on first use; the host-only harness has no device: gloo's world"""
    if env["hostonly"]:
        env["rccl_ranks_seen"] = env["dist"].get_world_size()
        return None
    if env["rccl"] is None:
        torch, dist = env["torch"], env["dist"]
        env["rccl"] = dist.new_group(backend="nccl", device_id=torch.device("cuda", env["local_rank"]))
        env["rccl_ranks_seen"] = dist.get_world_size(env["rccl"])
    return env["rccl"]


_T0 = time.perf_counter()


def progress(env, msg):
    """FWGPU_BENCH_PROGRESS=1: where a run is, on stderr (a run of N ranks that stalls says where)"""
    if os.environ.get("FWGPU_BENCH_PROGRESS"):
        sys.stderr.write("[bench %7.1f s] rank %d/%d: %s\n" % (time.perf_counter() - _T0, env["rank"], env["world"], msg))
        sys.stderr.flush()


OTHER_WARM_MS = 100  # other_configs entries: untimed steps until the clocks have settled (run_workload)


def run_workload(env, args, wl, V, B, K, F, steps, warmup, full=True, parity_multi=False, rt_probe=0, warm_ms=0.0):
    """times `steps` steps of one workload; returns the fields of its bench line (rank 0) — `full`: with the CPU baseline,
    the parity check and the realtime probe.  warm_ms > 0: behind the `warmup` steps, more untimed steps until that many milliseconds of
    stepping have gone by (round 6: the device's clocks take tens of milliseconds of load to settle — a step of the LDS- / VALU-bound
    resampler kernel takes 0.68 ms right behind 5 warm-up steps, 0.56 after 20, 0.49 after 80; the HBM-bound headline kernel is steady
    after 5 — so the `other_configs` entries, whose step counts are this file's choice, are timed warm; the headline keeps the
    driver's W)"""
    torch, fa, shard, dist = env["torch"], env["fa"], env["shard"], env["dist"]
    rank, world, device, dev = env["rank"], env["world"], env["device"], env["dev"]
    hostonly = env["hostonly"]

    def sync():
        if not hostonly:
            torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream if not hostonly else None
    # synthetic sources, generated in HBM: uniform(-1,1) f32, the stream keyed by the shard's first GLOBAL voice id
    sfmt = args.source_format if (wl in ("cfg2", "cfg5") or (wl == "cfg3" and getattr(args, "chain_reordered", False))) else "f32"
    if sfmt == "i16":  # interleaved stereo PCM, [voice][frame][channel]
        gen = torch.Generator(device=dev)
        gen.manual_seed(shard.voice_seed(rank * V))
        src = torch.randint(-32768, 32768, (V, F, 2), dtype=torch.int16, device=dev, generator=gen)
    else:
        # (--src-stagger: floats between consecutive voices' buffers — where the samples sit in HBM relative to one another)
        src = shard_sources(torch, shard, rank, V, F, dev, args.src_stagger)
    progress(env, "%s: %d voices, block %d, K %d, %d + %d steps: sources made" % (wl, V, B, K, warmup, steps))
    build_err = None
    try:
        cx, g, samplers, volumes = make_gpu(fa, wl, V, B, K, args.radix, src, F, sfmt, rank, args, stream, device)
    except Exception as ex:  # noqa: BLE001
        if dist is None:
            raise
        build_err = repr(ex)
    if dist is not None:
        # every rank leaves together: a rank that could not build its shard (round 5: 8 ranks x config 5 at 64 blocks per step is
        # 360 GB of tables on ONE shared device — out of memory on one rank, the other seven waiting in the next collective for ever)
        errs = [None] * world
        dist.all_gather_object(errs, build_err)
        if any(errs):
            raise SystemExit("bench.py: %s" % "; ".join("rank %d could not build its shard: %s" % (r, e) for r, e in enumerate(errs) if e))
    progress(env, "%s: graph built, plan %d" % (wl, cx.plan_kind()))
    variant = args.variant if wl in ("cfg2", "cfg5") else "A"
    playing = 1.0
    changes = {}
    if variant == "C":
        for s in samplers[::4]:
            cx._check(cx.L.fwgpu_sampler_pause(cx.c, s, 0))
        playing = 1.0 - len(samplers[::4]) / float(len(samplers))
    elif variant == "B":  # voice v changes its gain once, at block b_v of timed step s_v
        import numpy as np

        rng = np.random.default_rng(99 + rank)
        for v, vol in enumerate(volumes):  # (keyed by the step of the TIMED region: step_no counts from its start)
            changes.setdefault(int(rng.integers(0, steps)), []).append(
                (vol, float(rng.uniform(10, 100)), int(rng.integers(0, K))))
    # two bus buffers: with N > 1 the reduction of step i overlaps the compute of step i+1 — the mix bus is a sink, nothing
    # in a shard reads it back — and each buffer holds the buses of R consecutive steps, reduced by ONE exchange / collective
    # (the hand-over between the compute stream and RCCL's stream costs ~10 us of idle GPU per event on this stack).
    # Every step also reports its per-(block, channel) silence flags: the top-level SumNode skips silent ports (sum.rs:122-124).
    R = max(1, args.reduce_every) if dist is not None else 1
    step_elems = K * B * 2
    outs = [torch.empty(R * step_elems, dtype=torch.float32, device=dev) for _ in range(2)]
    sils = [torch.zeros(R * K * 2, dtype=torch.uint8, device=dev) for _ in range(2)] if dist is not None else None
    reducer, reduce_mode, reduce_note = (None, None, None)
    if dist is not None:
        progress(env, "%s: opening the mix-bus reduction (%s)" % (wl, args.bus_reduce))
        reducer, reduce_mode, reduce_note = make_reducer(env, args, cx, outs, sils, B)
        progress(env, "%s: reduction ready (%s)" % (wl, reduce_mode))
    step_no = [-1]  # (-1: warming up; the timed region counts from 0)
    slot = [0]  # bus slot counter: buffer (slot // R) % 2, slice slot % R
    xfail = []
    host_out = None
    if args.host_buffers:
        import numpy as np

        host_out = np.empty(K * B * 2, dtype=np.float32)

    # (variant B's messages go out as ONE foreign call per step — fwgpu_node_set_params; a hundred ctypes calls were ~0.15 ms of Python
    #  per step, which a native host does not pay)
    import ctypes as C

    bulk = {}
    for sno, lst in changes.items():
        n = len(lst)
        bulk[sno] = (n, (C.c_int64 * n)(*[x[0] for x in lst]), (C.c_int * n)(*([0] * n)), (C.c_float * n)(*[x[1] for x in lst]),
                     (C.c_uint32 * n)(*[x[2] for x in lst]))
    set_params_raw = cx.L.fwgpu_node_set_params
    ctx_ptr = cx.c

    def step():
        b = (slot[0] // R) % 2
        r = slot[0] % R
        m = bulk.get(step_no[0]) if step_no[0] >= 0 else None
        if m is not None and set_params_raw(ctx_ptr, m[0], m[1], m[2], m[3], m[4]) < 0:
            raise RuntimeError("fwgpu_node_set_params failed")
        if step_no[0] >= 0:
            step_no[0] += 1
        slot[0] += 1
        if reducer is not None and r == 0:
            reducer.wait(b)  # the collective that last used this buffer (2R steps ago)
        if args.host_buffers:  # the literal process_interleaved boundary: pageable host output, synchronous
            import ctypes as C

            if getattr(args, "host_async", False):
                # ... or the same call split in two (fwgpu_process_interleaved_begin / _end): step n + 1 is begun before step n is
                # ended, so the copy back and the host's memcpy of step n overlap the rendering of step n + 1
                t = cx.L.fwgpu_process_interleaved_begin(cx.c, None, 0, 2, K * B, 0.0, 0)
                assert t >= 0, t
                if pend_async[0] is not None:
                    rc = cx.L.fwgpu_process_interleaved_end(cx.c, pend_async[0], host_out.ctypes.data_as(C.POINTER(C.c_float)))
                    assert rc == 0, rc
                pend_async[0] = t
                return
            rc = cx.L.fwgpu_process_interleaved(cx.c, None, host_out.ctypes.data_as(C.POINTER(C.c_float)), 0, 2, K * B, 0.0, 0)
            assert rc == 0, rc
            return
        if reducer is not None:
            cx.process_blocks_device_flags(K, outs[b].data_ptr() + r * step_elems * 4, 2, sils[b].data_ptr() + r * K * 2)
        else:
            cx.process_blocks_device(K, outs[b].data_ptr() + r * step_elems * 4, 2)
        if reducer is not None and r == R - 1:  # the mix bus: one exchange per R steps over R x K x 2 x block f32
            reducer.submit(b)
            if env.get("share_device"):
                # N processes time-sharing ONE device (no scaling figure comes out of this mode): every exchange is waited for before
                # the next step is queued, as tests/test_bus_exchange.py does it — with 8 ranks' launch queues running ahead of one
                # another on one device, config 5's every-step exchange did not come back within minutes (round 5)
                try:
                    reducer.wait_all()
                except fa.FwgpuError as ex:
                    xfail.append(repr(ex))

    pend_async = [None]

    def finish_reductions():
        """submit the partly filled buffer (steps % R != 0), then wait for every collective"""
        if pend_async[0] is not None:  # (host-buffer steps split in two: the last step's other half)
            import ctypes as C

            rc = cx.L.fwgpu_process_interleaved_end(cx.c, pend_async[0], host_out.ctypes.data_as(C.POINTER(C.c_float)))
            assert rc == 0, rc
            pend_async[0] = None
        if reducer is None:
            return
        if slot[0] % R != 0:
            reducer.submit((slot[0] // R) % 2)
            slot[0] += R - slot[0] % R
        try:
            reducer.wait_all()
        except fa.FwgpuError as ex:  # exchange: a peer did not arrive in time.  Kept until the ranks can agree on it (below):
            xfail.append(repr(ex))    # raising here would leave the others inside the next barrier

    for _ in range(warmup):
        step()
    warm_steps = warmup
    if warm_ms > 0 and not hostonly and dist is None:
        sync()
        tw = time.perf_counter()
        while (time.perf_counter() - tw) * 1e3 < warm_ms and warm_steps < 5000:
            for _ in range(4):
                step()
            warm_steps += 4
            sync()
    finish_reductions()
    timing = not args.no_kernel_timing and not hostonly
    sync()
    step_no[0] = 0
    progress(env, "%s: warm" % wl)
    if dist is not None:
        dist.barrier()
    sync()
    lazy0 = cx.lazy_stats() if hasattr(cx, "lazy_stats") else (0, 0)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt_enqueue = time.perf_counter() - t0  # the host's share: message calls + launches of all the steps (the device runs behind them)
    finish_reductions()  # every bus of the timed region is fully reduced before the clock stops
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    progress(env, "%s: timed region done, %.3f ms per step" % (wl, dt / steps * 1e3))
    lazy1 = cx.lazy_stats() if hasattr(cx, "lazy_stats") else (0, 0)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if env.get("ctrl_cpu") else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        errs = [None] * world
        dist.all_gather_object(errs, xfail[0] if xfail else None)
        if any(errs):  # every rank leaves together; main() runs the workload again over the RCCL all-reduce
            if hasattr(reducer, "close"):
                reducer.close(dist)
            cx.close()
            raise ExchangeFailed("; ".join("rank %d: %s" % (r, e) for r, e in enumerate(errs) if e))

    roofline = None
    if timing:
        # Per-kernel HIP events, on the ctx stream, in a SEPARATE pass of the same steps right after the timed region:
        # an event record between two dependent kernels costs ~10 us of idle GPU on this stack (rocprofv3 trace), which
        # would inflate ms_per_step by 25 % on config 2 if the events sat inside the timed region.  Kernel durations
        # themselves are unaffected (profiles/*_kernel_stats.csv agrees).
        ev_steps = min(steps, 20)
        cx.timing_reset()
        cx.timing_enable(True)
        for _ in range(ev_steps):
            step()
        finish_reductions()
        sync()
        cx.timing_enable(False)
        dom_ms, dom_n = cx.timing_read(0)   # k_leaf_sum (plan 1) / k_chain (plan 2)
        ctl_ms, ctl_n = cx.timing_read(1)   # k_voice_control
        up_ms, up_n = cx.timing_read(2)     # k_bus_sum levels + k_graph_out
        gen_ms, gen_n = cx.timing_read(3)   # generic executor: all level kernels of one block (cfg4: + FIR GEMM)
        fir_ms, fir_n = cx.timing_read(4)   # k_fir_gemm alone
        if wl == "cfg4" and fir_n:
            # one k_fir_gemm launch = the FIR bank of the step's K blocks: K x (512 rows x 256 cols x 65791 positions)
            flops = 2.0 * 2 * args.taps * V * B * K  # direct-form definition: 2 ch x 2 flop x T per voice-sample
            avg_s = fir_ms / fir_n / 1e3
            ach = flops / avg_s / 1e12
            traffic, traffic_src = pmc_traffic("k_fir_gemm", V, B, K, wl)
            step_us = dt / steps * 1e6
            roofline = {"bound": "mfma", "kernel": "k_fir_gemm", "achieved": ach,
                        "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF, "traffic": traffic,
                        "traffic_source": traffic_src, "algorithmic_flops_per_launch": flops, "avg_launch_us": avg_s * 1e6,
                        "launches": fir_n, "blocks_per_launch": K, "timing": "HIP events, separate pass after the timed region",
                        "whole_block_us_all_kernels": gen_ms / max(gen_n, 1) / K * 1e3,
                        # the WHOLE step against the same peak: the step's algorithmic flops / its wall time in the timed region
                        "whole_step_frac": flops * (fir_n / float(ev_steps)) / (step_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TF,
                        "other_kernels_us_per_step": {"levels+reduce (all but k_fir_gemm)": (gen_ms - fir_ms) / ev_steps * 1e3,
                                                      # (hybrid plan: the samplers in front of the FIR nodes are solo voices)
                                                      "k_voice_control + solo-voice leaves": (ctl_ms + dom_ms) / ev_steps * 1e3},
                        "idle_us_per_step": step_us - (gen_ms + ctl_ms + dom_ms) / ev_steps * 1e3}
        elif dom_n:
            # source L+R once (f32: 8 B, i16: 4 B) (+ delay ring read + write)
            per_vs = (20.0 if sfmt == "i16" else 24.0) if wl == "cfg3" else (4.0 if sfmt == "i16" else 8.0)  # (cfg3: source + ring read + ring write)
            kernel = "k_chain" if wl == "cfg3" else "k_leaf_sum"
            if getattr(args, "rs_source", False) and wl in ("cfg2", "cfg5"):
                kernel = "k_leaf_rs"  # (+ k_leaf_sum_wl over its work list, timed together: fwgpu_kernels.hip launch_leaf_sum)
            k_launch = min(K, 64) if wl == "cfg3" else K  # the chain plan renders at most 64 blocks per k_chain launch
            alg_bytes = V * B * k_launch * per_vs * playing  # paused voices (variant C) fetch nothing
            avg_s = dom_ms / dom_n / 1e3
            ach = alg_bytes / avg_s / 1e9
            # (PMC passes ran on f32 sources; a profile is quoted for the workload it was collected on: plain / --voice-fx / --rs-source)
            prof_name = (wl + ("_voicefx" if args.voice_fx else "") + ("_rs" if getattr(args, "rs_source", False) else "") +
                         ("_spatial" if getattr(args, "voice_spatial", False) else "") + ("_reordered" if getattr(args, "chain_reordered", False) else ""))
            plain = not (args.master or getattr(args, "master_iir", False) or variant != "A" or args.force_generic or getattr(args, "send", False))
            traffic, traffic_src = pmc_traffic(kernel, V, B, K, prof_name) if (sfmt == "f32" or getattr(args, "chain_reordered", False)) and plain else (None, None)
            step_us = dt / steps * 1e6
            others = {"k_voice_control": ctl_ms / ev_steps * 1e3, "upper_sums+graph_out": up_ms / ev_steps * 1e3}
            if gen_n:
                others["level executor (hybrid plan / master chain)"] = gen_ms / ev_steps * 1e3
            frac = ach / HBM_PEAK_GBS
            roofline = {
                "bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": frac, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_voice_sample": per_vs, "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_us": avg_s * 1e6, "launches": dom_n, "launches_per_step": dom_n / float(ev_steps),
                "timing": "HIP events, separate pass after the timed region",
                # north_star's "throughput as achieved fraction of the HBM roofline": the step's algorithmic bytes / its wall time in
                # the TIMED region (every kernel, every gap) against the same 8 TB/s — `frac` above is the dominant kernel alone
                "whole_step_frac": V * B * K * per_vs * playing / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "other_kernels_us_per_step": others,
                # what is left of the timed step once every kernel's own duration is taken out: launch gaps, cross-stream waits
                # (control kernels of calls WITH messages run a call ahead on their own stream and are then not in the step at all)
                "idle_us_per_step": step_us - (dom_ms + ctl_ms + up_ms + gen_ms) / ev_steps * 1e3,
                # k_leaf_sum's two HBM placement states (profiles/PLACEMENT.md), fixed per context at allocation
                "placement_state": (("fast" if frac >= 0.76 else "slow") if kernel == "k_leaf_sum" and sfmt == "f32" and playing == 1.0 else None),
            }
    if roofline is None and timing and args.force_generic and wl in ("cfg2", "cfg5"):
        # the level executor alone: no dominant kernel — the step against the same 8 B per voice-sample the fused plan is priced at
        # (what the levels really move is 24 B per voice-sample of pool traffic since the vertical fusion: DESIGN.md section 3.1)
        step_us = dt / steps * 1e6
        ach = V * B * K * 8.0 / (step_us * 1e-6) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_level (sampler / volume / pan / sum levels)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "whole_step_frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_voice_sample": 8.0,
                    "pool_traffic_bytes_per_voice_sample": 24.0, "avg_launch_us": gen_ms / max(gen_n, 1) * 1e3 if gen_n else None,
                    "timing": "the timed region's wall clock (all level launches of a step)"}
    step_dist = None
    if timing and world == 1 and steps >= 5:
        # SURVEY 8d's per-step statistics: a distribution INSIDE the context (the line's `contexts` entry is one across contexts).
        # One event per step boundary in a pass of its own: a boundary event costs a few us of idle stream, which is inside
        # every sample — so the median here sits slightly above ms_per_step of the timed region, which has no events in it.
        n_d = min(steps, 20)
        ctx_stream = torch.cuda.ExternalStream(cx.hip_stream(), device=dev)  # the stream the kernels are launched on (fwgpu_hip_stream)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_d + 1)]
        evs[0].record(ctx_stream)
        for i in range(n_d):
            step()
            evs[i + 1].record(ctx_stream)
        finish_reductions()
        sync()
        us = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n_d))
        pick = lambda q: us[min(n_d - 1, int(round(q * (n_d - 1))))]
        step_dist = {"n": n_d, "min": us[0], "p10": pick(0.10), "median": pick(0.50), "p90": pick(0.90), "max": us[-1], "unit": "us per step",
                     "per_block_us": {"p10": pick(0.10) / K, "median": pick(0.50) / K, "p90": pick(0.90) / K},
                     "how": "one stream event per step boundary, a separate pass after the timed region (the events' own gaps are inside the samples)"}
    own = None
    if (full and rank == 0 and world == 1 and not hostonly and not args.no_parity_check and wl in ("cfg2", "cfg5") and sfmt == "f32" and variant == "A"
            and not args.host_buffers and F % B == 0 and K >= 4 and
            not (args.master or getattr(args, "master_iir", False) or getattr(args, "send", False) or getattr(args, "rs_source", False) or
                 getattr(args, "voice_spatial", False) or args.voice_fx or args.force_generic)):
        # The TIMED context's own output (parity ran on a fresh context only): one more call on the context that was just
        # timed, blocks {0, 1, K/2, K-1} of it against the oracle.  Every voice loops over its whole F-frame buffer and has played
        # step_no x K blocks since frame 0, so block b of this call reads source frames [(p + b*B) mod F, +B) with p = calls x K x B
        # mod F (F is a multiple of B: no wrap inside a block) — the oracle, starting at frame 0, is handed exactly those frames.
        import numpy as np

        t_own = time.perf_counter()
        p0 = (step_no[0] * K * B) % F
        step()
        finish_reductions()
        sync()
        b_used = ((slot[0] - 1) // R) % 2
        r_used = (slot[0] - 1) % R
        got_all = outs[b_used][r_used * step_elems:(r_used + 1) * step_elems]
        blocks = [0, 1, K // 2, K - 1]
        got = torch.cat([got_all[b * B * 2:(b + 1) * B * 2] for b in blocks]).cpu().numpy()
        host = torch.cat([src[:, :, (p0 + b * B) % F:(p0 + b * B) % F + B] for b in blocks], dim=2).cpu().numpy()
        o, _, _ = make_oracle(wl, V, B, args.radix, rank, args, host)
        ref = o.e.process_blocks(len(blocks))
        same = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32))) and bool(np.any(ref))
        own = {"bit_exact": same, "what": "one more call on the context that was just timed (after %d calls of %d blocks), compared with the oracle"
               % (step_no[0] - 1, K), "block_indices": blocks, "voices": V, "source_frame_of_block_0": p0,
               "samples_compared": int(got.size), "secs": round(time.perf_counter() - t_own, 2)}
        del o
    res = None
    if rank == 0:
        total = float(V) * B * K * steps * world
        name, cus, hbm = cx.device_info()
        desc = {
            "cfg2": "cfg2: %d stereo voices/GPU, sampler->gain->pan->radix-%d sum tree" % (V, args.radix),
            "cfg3": "cfg3: %d stereo voices/GPU, sampler->biquad LPF->delay(fb)->gain->radix-%d sum tree" % (V, args.radix),
            "cfg4": "cfg4: %d stereo voices/GPU, sampler->%d-tap stereo FIR (f32 MFMA Toeplitz GEMM)->radix-%d sum tree"
                    % (V, args.taps, args.radix),
            "cfg5": "cfg5 shard: %d stereo voices/GPU, sampler->gain->pan->radix-%d sum tree" % (V, args.radix),
        }[wl]
        if args.voice_fx and wl in ("cfg2", "cfg5"):
            desc += " + StereoWidth + HardClip in every voice"
        if args.rs_source and wl in ("cfg2", "cfg5"):
            desc = desc.replace("sampler->", "resampler(ratio U(0.5,1.5), looping)->")
        if getattr(args, "voice_spatial", False) and wl in ("cfg2", "cfg5"):
            desc = desc.replace("->pan->", "->pan->3D spatialiser(ITD + distance + equal-power gains)->")
        if getattr(args, "chain_reordered", False) and wl == "cfg3":
            desc = "cfg3_reordered: %d stereo voices/GPU, sampler->gain->biquad LPF->biquad BP->delay(fb)->pan->radix-%d sum tree" % (V, args.radix)
        if getattr(args, "send", False) and wl in ("cfg2", "cfg3", "cfg5"):
            desc += " + every 4th leaf bus tapped into a send -> gain -> width -> limiter return (hybrid plan)"
        res = {
            "value": total / dt,
            "ms_per_step": dt / steps * 1e3,
            "steps": steps,
            "config": {
                "workload": "%s, block=%d @48kHz, %s sources in HBM (%d frames/voice, looping)"
                            % (desc, B, "planar f32" if sfmt == "f32" else "interleaved stereo i16", F),
                "voices_per_gpu": V, "block": B, "blocks_per_step": K, "variant": variant, "master_chain": bool(args.master),
                "voice_fx": bool(args.voice_fx), "launch_plan": cx.plan_kind(),
                "parallelism": "voice-shard x%d%s" % (world, (" + mix-bus %s" % REDUCE_DESC[reduce_mode]) if world > 1 else ""),
                "bus_reduce": reduce_mode, "bus_reduce_fallback": reduce_note,
                "realtime_factor": (total / dt) / (48000.0 * V * world),
                # launch batches of the timed region rendered without / with a control kernel (include/fwgpu.h fwgpu_lazy_stats): a
                # message-free step of a plan whose every voice is steady and plain needs no per-block state machine pass
                "batches_without_control_kernel": lazy1[0] - lazy0[0], "batches_with_control_kernel": lazy1[1] - lazy0[1],
                # untimed steps in front of the timed region: the W asked for + (warm_ms > 0) the ones that brought the clocks up
                "warmup_steps_run": warm_steps, "warm_ms": warm_ms,
                # how long the host spent inside a step's calls (message calls + the process call: its launches AND, for a call with
                # messages, its wait for the staging buffers of the call before — so a figure near ms_per_step means "paced by the device")
                "host_enqueue_ms_per_step": dt_enqueue / steps * 1e3,
                "device": name, "compute_units": cus,
            },
            "roofline": roofline,
            "step_time_distribution": step_dist,
        }
        if own is not None:
            res["parity_check_timed_context"] = own
    if reducer is not None and hasattr(reducer, "close"):
        if res is not None:  # how far the ranks ran apart: the longest rank 0's reduce kernels waited for each rank's arrival
            res["config"]["bus_exchange_max_wait_us"] = reducer.x.wait_stats() if hasattr(reducer, "x") else None
        reducer.close(dist)
    if world > 1 and (full or parity_multi) and not args.no_parity_check and wl in ("cfg2", "cfg5", "cfg3") and sfmt == "f32":
        pc = parity_check_multi(env, args, wl, V, B, K, F, src, stream, device, reduce_mode)  # collective: every rank takes part
        if rank == 0:
            res["parity_check"] = pc
    if rt_probe and not full and rank == 0 and world == 1 and not hostonly and not args.no_realtime:
        # the reference's callback is one block per call for EVERY graph (firewheel-cpal/src/lib.rs:429-437): this
        # config's graph through the headless stream, one block per callback, host buffers, back to back
        try:
            us = realtime_probe(cx, B, callbacks=rt_probe)[0]
            res["realtime_us_per_callback"] = us
            res["realtime_block_period_us"] = B / 48000.0 * 1e6
            res["realtime_frac_of_block_period"] = us / (B / 48000.0 * 1e6)
            if hasattr(cx, "rt_path_stats"):
                res["realtime_path"] = dict(zip(("resident_kernel", "one_launch", "fused_launch_sequence", "level_executor"), cx.rt_path_stats()))
        except Exception as ex:  # noqa: BLE001
            res["realtime_us_per_callback"] = {"error": repr(ex)}
    if full and rank == 0 and world == 1 and not hostonly:
        if not args.no_realtime and wl != "cfg4":
            p0 = cx.rt_path_stats() if hasattr(cx, "rt_path_stats") else None
            res["realtime_us_per_callback"], res["realtime_us_per_callback_from_python"] = realtime_probe(cx, B)
            if p0 is not None:
                res["realtime_path"] = dict(zip(("resident_kernel", "one_launch", "fused_launch_sequence", "level_executor"),
                                                [a - b for a, b in zip(cx.rt_path_stats(), p0)]))
        cx.close()
        if not args.no_parity_check:
            res["parity_check"] = parity_check(fa, torch, wl, V, B, K, args.radix, rank, args, src, F, sfmt, stream, device)
        if not args.no_cpu_baseline and sfmt == "f32":
            res["cpu_baseline"] = cpu_baseline(wl, V, B, args.radix, rank, args, src, F, args.cpu_secs)
    else:
        cx.close()
    del src, outs
    if not hostonly:
        torch.cuda.empty_cache()
    return res


def context_summary(runs, mid):
    """the fresh contexts one entry was timed in: every run's step time, kernel time, roofline fractions and placement state"""
    rf = [(r["roofline"] or {}) for r in runs]
    ms = [r["ms_per_step"] for r in runs]
    return {"n": len(runs), "ms_per_step_runs": ms, "kernel_us_runs": [x.get("avg_launch_us") for x in rf],
            "roofline_frac_runs": [x.get("frac") for x in rf], "whole_step_frac_runs": [x.get("whole_step_frac") for x in rf],
            "placement_state_runs": [x.get("placement_state") for x in rf],
            "median": ms[mid], "min": min(ms), "max": max(ms),
            "reported": "the median context (run %d of %d): value, ms_per_step and roofline are its own" % (mid, len(runs)),
            "why": "k_leaf_sum runs in one of two HBM placement states fixed per context at allocation (DESIGN.md section 7): "
                   "`fast` = kernel-only roofline fraction >= 0.76, else `slow`"}


def other_configs(env, args):
    """configs 1, 3, 4, 5 (one shard) and the hybrid-plan variant of config 2 in short form, each on the driver-run line next to the headline: a few steps at the
    config's own size, its roofline, and the same parity check against the oracle (~45 s together)."""
    out = {}
    t0 = time.perf_counter()
    try:
        out["cfg1"] = run_cfg1(env["fa"], env["torch"].cuda.current_stream().cuda_stream, env["device"])
    except Exception as ex:  # a broken side config must not take the headline line with it
        out["cfg1"] = {"error": repr(ex)}
    import copy

    # (cfg2_sends: the headline graph with sends off its leaf buses — no fused shape as a whole: the hybrid plan)
    # (cfg2_rs / cfg2_spatial: the headline graph with every voice's source a SPEC resampler (ratio U(0.5, 1.5), looping) / every
    #  voice ending in a SPEC spatialiser — the other two north-star node families on the voice-bank plan; cfg2_variantB: 64 voices
    #  with a volume glide every ~20 blocks, the reference's automation case)
    # (cfg2_i16: SURVEY 8d's 4 B row — the headline graph on interleaved 16-bit PCM sources, core/sample_resource.rs:338-340)
    # (cfg2_levels: --force-generic, no roofline object: five level kernels, none of them dominant)
    # cfg3 / cfg5: three fresh contexts each, the median reported and all three listed with their placement state — one context is
    # a coin toss between k_leaf_sum's two HBM placement states (the profile and the line disagreed by 12 % on cfg5)
    # (cfg3_reordered, round 6: config 3's voices in an order the chain plan refused before — a gain in FRONT of the filter, two biquads,
    #  interleaved 16-bit sources: 20 B per voice-sample)
    for name, steps, n_ctx in (("cfg3", 12, 3), ("cfg3_reordered", 12, 1), ("cfg5", 12, 3), ("cfg4", 6, 1), ("cfg2_sends", 12, 1), ("cfg2_rs", 10, 1),
                               ("cfg2_spatial", 10, 1), ("cfg2_variantB", 10, 1), ("cfg2_i16", 12, 1), ("cfg2_levels", 4, 1)):
        wl = name.split("_")[0]
        V, B, K, F, _ = DEFAULTS[wl]
        try:
            wargs = args
            sfmt = "f32"
            if name != wl:
                wargs = copy.copy(args)
                wargs.send = name == "cfg2_sends"
                wargs.rs_source = name == "cfg2_rs"
                wargs.voice_spatial = name == "cfg2_spatial"
                if name == "cfg2_variantB":
                    wargs.variant = "B"
                if name == "cfg2_i16":
                    wargs.source_format = sfmt = "i16"
                wargs.chain_reordered = name == "cfg3_reordered"
                if name == "cfg3_reordered":
                    wargs.source_format = sfmt = "i16"
                if name == "cfg2_levels":  # the headline graph on the level executor alone: what a graph no fused plan takes runs at
                    wargs.force_generic = True
            # (the BASELINE configs themselves also report their one-block-per-callback latency: the last context of each)
            rt = 200 if name in ("cfg3", "cfg4", "cfg5") else 0
            runs = [run_workload(env, wargs, wl, V, B, K, F, steps, 3, full=False, rt_probe=rt if i == n_ctx - 1 else 0, warm_ms=OTHER_WARM_MS)
                    for i in range(n_ctx)]
            order = sorted(range(n_ctx), key=lambda i: runs[i]["ms_per_step"])
            r = runs[order[n_ctx // 2]]
            cfg = r["config"]
            ent = {"workload": cfg["workload"], "value": r["value"], "unit": "voice-samples/s", "ms_per_step": r["ms_per_step"],
                   "steps": steps, "warmup_steps_run": cfg.get("warmup_steps_run"), "warm_ms": cfg.get("warm_ms"), "blocks_per_step": K, "launch_plan": cfg["launch_plan"], "realtime_factor": cfg["realtime_factor"],
                   "roofline": r["roofline"]}
            for key in ("realtime_us_per_callback", "realtime_block_period_us", "realtime_frac_of_block_period", "realtime_path"):
                if key in runs[-1]:
                    ent[key] = runs[-1][key]
            if n_ctx > 1:
                ent["contexts"] = context_summary(runs, order[n_ctx // 2])
            if not args.no_parity_check:
                torch = env["torch"]
                gen = torch.Generator(device=env["dev"])
                gen.manual_seed(env["shard"].voice_seed(0))
                if sfmt == "i16":
                    src = torch.randint(-32768, 32768, (V, F, 2), dtype=torch.int16, device=env["dev"], generator=gen)
                else:
                    src = torch.empty((V, 2, F), dtype=torch.float32, device=env["dev"])
                    src.uniform_(-1.0, 1.0, generator=gen)
                ent["parity_check"] = parity_check(env["fa"], torch, wl, V, B, K, args.radix, 0, wargs, src, F, sfmt,
                                                   torch.cuda.current_stream().cuda_stream, env["device"])
                del src
                torch.cuda.empty_cache()
            out[name] = ent
        except Exception as ex:
            out[name] = {"error": repr(ex)}
    # VERDICT r4 #4 / SURVEY H1: what a host that can only tolerate K blocks of batching gets — the headline graph with K blocks per
    # device call (fwgpu_process_blocks_device, output left in HBM), a fresh context per K, ~3 000 blocks each; K x 5.33 ms is the
    # latency the batching adds.  (K = 1 here is the multi-launch device call; the host-buffer callback edge is
    # realtime_us_per_callback.)
    try:
        V, B, _, F, _ = DEFAULTS["cfg2"]
        sweep = {}
        for Kx in (1, 4, 16, 64, 256, 768):
            n = max(4, min(1500, 3072 // Kx))
            r = run_workload(env, args, "cfg2", V, B, Kx, F, n, 3, full=False)
            sweep[str(Kx)] = {"value": r["value"], "ms_per_call": r["ms_per_step"], "calls": n, "batching_latency_ms": Kx * B / 48000.0 * 1e3,
                              "whole_step_frac": (r["roofline"] or {}).get("whole_step_frac")}
        out["k_sweep"] = {"workload": "cfg2 (the headline graph), K blocks per fwgpu_process_blocks_device call", "unit": "voice-samples/s", "by_K": sweep}
    except Exception as ex:  # noqa: BLE001
        out["k_sweep"] = {"error": repr(ex)}
    out["secs"] = round(time.perf_counter() - t0, 1)
    return out


def other_configs_multi(env, args):
    """N > 1, next to the headline: (1) BASELINE configs[4] as it is written — 8 192 voices per GPU, block 1024, the mix-bus
    reduction after EVERY step — with its own roofline and parity check against the oracle's whole graph; (2) the headline
    workload again under the other mix-bus reductions (the RCCL all-reduce north_star names; all-gather + rank-ordered sum), each
    with its own parity check.  Every rank runs them (they are collective); rank 0 returns the dict."""
    import copy

    out = {"other_configs": {}, "bus_reduce_modes": {}}
    t0 = time.perf_counter()

    def short(r):
        cfg = r["config"]
        return {"workload": cfg["workload"], "value": None if env["hostonly"] else r["value"], "unit": "voice-samples/s", "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                "blocks_per_step": cfg["blocks_per_step"], "parallelism": cfg["parallelism"], "bus_reduce": cfg["bus_reduce"],
                "bus_reduce_fallback": cfg["bus_reduce_fallback"], "bus_exchange_max_wait_us": cfg.get("bus_exchange_max_wait_us"),
                "roofline": r["roofline"], "parity_check": r.get("parity_check")}

    hostonly = env["hostonly"]
    V, B, K, F, _ = (64, 64, 4, 1024, 0) if hostonly else DEFAULTS["cfg5"]
    if env.get("share_device") and env["world"] > 2 and not hostonly:
        K = 16  # (N ranks' tables on ONE device: at 64 blocks per step config 5 is ~45 GB per rank — 8 ranks do not fit 288 GB)
    a5 = copy.copy(args)
    a5.reduce_every = 1  # "collective every step"
    try:
        r = run_workload(env, a5, "cfg5", V, B, K, F, 3 if hostonly else 12, 1 if hostonly else 3, full=False, parity_multi=True)
        if env["rank"] == 0:
            out["other_configs"]["cfg5"] = short(r)
    except Exception as ex:  # (an exception on one rank only would leave the others in a collective: the run then times out loudly)
        out["other_configs"]["cfg5"] = {"error": repr(ex)}
    if not env.get("share_device"):
        dV, dB, dK, dF, _ = (64, 64, 4, 1024, 0) if hostonly else DEFAULTS["cfg2"]
        # (the two RCCL reductions through torch.distributed, the same two through libfwgpu's own C ABI — what a Rust host gets —, the exchange)
        for mode in ("allreduce", "ordered", "exchange") + (() if hostonly else ("allreduce_abi", "ordered_abi")):
            if mode == args.bus_reduce:
                continue
            am = copy.copy(args)
            am.bus_reduce = mode
            try:
                r = run_workload(env, am, "cfg2", dV, dB, dK, dF, 3 if hostonly else 10, 1 if hostonly else 3, full=False, parity_multi=True)
                if env["rank"] == 0:
                    out["bus_reduce_modes"][mode] = short(r)
            except Exception as ex:
                out["bus_reduce_modes"][mode] = {"error": repr(ex)}
    out["other_configs"]["secs"] = round(time.perf_counter() - t0, 1)
    return out


# ------------------------------------------------------------------------------------------------ the line
LINE_BYTES_MAX = 8192  # the driver holds a bounded tail of stdout (round 5's 22.5 KB line came back `parsed: null`): the LAST stdout
#                        line is this compact object; everything else goes to a side file (`full`, named in the line) and to stderr


def _r(x, sig=6):
    """floats to `sig` significant digits; non-finite floats (which strict JSON parsers refuse) to None"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return dict((k, _r(v, sig)) for k, v in x.items())
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return dict((k, d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None) if isinstance(d, dict) else d


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "whole_step_frac", "traffic", "traffic_source",
             "algorithmic_bytes_per_voice_sample", "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "avg_launch_us",
             "placement_state", "pool_traffic_bytes_per_voice_sample")


def _parity_short(pc):
    if not isinstance(pc, dict):
        return pc
    out = _pick(pc, ("bit_exact", "within_tolerance", "expected", "blocks", "voices", "ranks", "bus_reduce", "samples_compared", "skipped",
                     "oracle_whole_graph_nonzero", "mismatches", "error", "max_err_over_scale"))
    if "against" in pc:
        out["against"] = "oracle (C++ restatement of the reference)"
    if isinstance(pc.get("deep"), dict):
        out["deep"] = _pick(pc["deep"], ("bit_exact", "voices_checked", "blocks_checked", "error"))
    return out


def _cfg_short(ent):
    """other_configs / bus_reduce_modes entries: {value, frac, whole_step_frac, bit_exact} + what names the run"""
    if not isinstance(ent, dict) or "error" in ent:
        return _pick(ent, ("error",)) if isinstance(ent, dict) else ent
    rf = ent.get("roofline") or {}
    pc = ent.get("parity_check")
    out = _pick(ent, ("value", "ms_per_step", "warmup_steps_run", "bus_reduce", "bus_reduce_fallback", "realtime_us_per_callback", "us_per_callback",
                      "within_tolerance", "max_abs_err_vs_oracle"))
    if "bus_reduce_fallback" in out:
        out["bus_reduce_fallback"] = _short(out["bus_reduce_fallback"], 120)
    if "value" in ent:
        out["value"] = ent["value"]  # (None on the host-only harness: kept, it is what says "no measurement")
    if rf:
        out.update(_pick(rf, ("kernel", "frac", "whole_step_frac", "avg_launch_us")))
    if isinstance(pc, dict):
        out["bit_exact"] = pc.get("bit_exact")
        out["parity_check"] = _pick(pc, ("ranks", "bus_reduce", "within_tolerance", "expected", "skipped", "oracle_whole_graph_nonzero", "error"))
        if isinstance(pc.get("deep"), dict):
            out["parity_check"]["deep_bit_exact"] = pc["deep"].get("bit_exact")
        if not out["parity_check"]:
            del out["parity_check"]
    return out


def compact_line(line, full_path):
    """the driver's line: the contract fields + roofline + cpu_baseline + parity, short forms of the rest"""
    out = _pick(line, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    out["vs_baseline"] = line.get("vs_baseline")
    for k in ("value", "ms_per_step"):
        out.setdefault(k, line.get(k))
    out.update(_pick(line, ("dtype", "data")))
    cfg = line.get("config") or {}
    out["config"] = _pick(cfg, ("workload", "voices_per_gpu", "block", "blocks_per_step", "variant", "launch_plan", "parallelism", "bus_reduce",
                                "realtime_factor", "batches_without_control_kernel", "batches_with_control_kernel", "warmup_steps_run", "warm_ms", "device", "compute_units"))
    out["config"]["bus_reduce_fallback"] = _short(cfg.get("bus_reduce_fallback"), 160)
    out["roofline"] = _pick(line.get("roofline"), ROOF_KEYS) if line.get("roofline") else None
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        out["cpu_baseline"]["sample"] = _short(cb.get("sample"), 200)
        if isinstance(cb.get("all_cores"), dict):
            out["cpu_baseline"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
    else:
        out["cpu_baseline"] = cb
    out["parity_check"] = _parity_short(line.get("parity_check"))
    if isinstance(line.get("parity_check_timed_context"), dict):
        out["parity_check_timed_context"] = _pick(line["parity_check_timed_context"], ("bit_exact", "samples_compared"))
    out.update(_pick(line, ("realtime_us_per_callback", "realtime_path", "virtual_ranks_on_one_device", "invalid")))
    if "note" in line:
        out["note"] = _short(line["note"], 200)
    vh = line.get("value_host_buffers")
    if isinstance(vh, dict):
        out["value_host_buffers"] = _pick(vh, ("value", "error"))
        if isinstance(vh.get("pipelined"), dict):
            out["value_host_buffers"]["pipelined"] = vh["pipelined"].get("value")
            out["value_host_buffers"]["pipelined_frac_of_value"] = vh["pipelined"].get("frac_of_value")
        out["value_host_buffers"]["what"] = "fwgpu_process_interleaved on host buffers: including D2H (SURVEY 8d)"
    if isinstance(line.get("contexts"), dict):
        c = line["contexts"]
        out["contexts"] = {"n": c.get("n"), "reported": "median", "ms_per_step_runs": _r(c.get("ms_per_step_runs"), 4),
                           "roofline_frac_runs": _r(c.get("roofline_frac_runs"), 3)}
    out["ranks_seen"] = line.get("ranks_seen")
    out["rccl_ranks_seen"] = line.get("rccl_ranks_seen")
    oc = line.get("other_configs")
    if isinstance(oc, dict):
        o2 = {}
        for name, ent in oc.items():
            if name == "k_sweep" and isinstance(ent, dict) and "by_K" in ent:
                o2[name] = dict((k, _r(v.get("whole_step_frac"), 3)) for k, v in ent["by_K"].items())
            elif name != "secs":
                o2[name] = _cfg_short(ent)
        out["other_configs"] = o2
    if isinstance(line.get("bus_reduce_modes"), dict):
        out["bus_reduce_modes"] = dict((m, _cfg_short(e)) for m, e in line["bus_reduce_modes"].items())
    out["full"] = full_path
    out = _r(out)
    text = json.dumps(out, allow_nan=False, separators=(",", ":"))
    if len(text) > LINE_BYTES_MAX:  # never again: drop the optional short forms, largest first, until it fits
        for k in ("bus_reduce_modes", "other_configs", "contexts", "value_host_buffers", "realtime_path", "parity_check_timed_context"):
            if k in out:
                out[k] = {"dropped": "line over %d bytes: see `full`" % LINE_BYTES_MAX}
                text = json.dumps(out, allow_nan=False, separators=(",", ":"))
                if len(text) <= LINE_BYTES_MAX:
                    break
    return text


def write_full(line, world):
    """the whole record (contexts, step distributions, the K sweep, every side config's roofline object) -> a side file + stderr"""
    text = json.dumps(_r(line, 9), allow_nan=False)
    path = None
    for d in (os.path.join(ROOT, "gpurun_out"), os.environ.get("TMPDIR", "/tmp")):
        try:
            os.makedirs(d, exist_ok=True)
            p = os.path.join(d, "bench_full.json" if world == 1 else "bench_full_n%d.json" % world)
            with open(p, "w") as f:
                f.write(text + "\n")
            path = os.path.relpath(p, ROOT) if p.startswith(ROOT) else p
            break
        except OSError:
            continue
    sys.stderr.write("[bench.py] full record (%d bytes) -> %s\n%s\n" % (len(text), path, text))
    sys.stderr.flush()
    return path



# ------------------------------------------------------------------------------------------------ launcher
def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here, one process per GPU, exactly as the
    driver would (torch.distributed.run, rendezvous on 127.0.0.1).  Rank 0's JSON line is the only thing on stdout."""
    import socket

    hostonly = bool(os.environ.get("FWGPU_BENCH_HOSTONLY"))
    if not hostonly:
        import torch

        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n < (1 if args.share_device else args.gpus):
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible on this node" % (args.gpus, n))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--warm-ms", type=float, default=0.0,
                    help="behind the --warmup steps, more untimed steps until this many milliseconds of stepping have gone by (the device's "
                         "clocks take tens of milliseconds of load to settle; profile collection and --workload runs; the default line's "
                         "headline keeps exactly --warmup steps, its other_configs entries use %d ms)" % OTHER_WARM_MS)
    ap.add_argument("--repeat-first", type=int, default=0,
                    help="diagnostic: run the headline workload this many times in the same process (fresh allocations each time) "
                         "before the reported run; their step / kernel times go to `repeats_before`")
    ap.add_argument("--workload", choices=sorted(DEFAULTS), default="cfg2",
                    help="cfg2 = the headline (BASELINE configs[1]); cfg3 / cfg4 / cfg5 = configs[2..4]")
    ap.add_argument("--voices", type=int, default=None, help="voices per GPU")
    ap.add_argument("--block", type=int, default=None)
    ap.add_argument("--radix", type=int, default=32)
    ap.add_argument("--blocks-per-step", type=int, default=None)
    ap.add_argument("--src-frames", type=int, default=None, help="source frames per voice (2 ch f32)")
    ap.add_argument("--src-stagger", type=int, default=0, help="f32 sources: floats of padding between consecutive voices' buffers")
    ap.add_argument("--cpu-secs", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-realtime", action="store_true")
    ap.add_argument("--lean", action="store_true",
                    help="profiling runs: the timed region only (no CPU baseline, parity check, other configs, realtime probe)")
    ap.add_argument("--taps", type=int, default=65536)
    ap.add_argument("--source-format", choices=["f32", "i16"], default="f32",
                    help="cfg2/cfg5 only: planar f32 sources (the headline) or interleaved stereo i16 (4 B per voice-sample)")
    ap.add_argument("--host-buffers", action="store_true",
                    help="time fwgpu_process_interleaved on HOST buffers instead (PCIe-inclusive; DESIGN.md §7 note, "
                         "never the headline)")
    ap.add_argument("--host-async", action="store_true",
                    help="with --host-buffers: the call split in two (fwgpu_process_interleaved_begin / _end), step n + 1 begun before step n is ended")
    ap.add_argument("--master", action="store_true",
                    help="put a master VolumeNode + HardClipNode between the root SumNode and graph_out (the fused plans "
                         "then run that chain with the generic node kernel on the mix bus)")
    ap.add_argument("--master-iir", action="store_true",
                    help="a master low-pass + delay between the root SumNode and graph_out: one serial recurrence over the whole call")
    ap.add_argument("--send", action="store_true",
                    help="cfg2/cfg5: every fourth leaf bus also feeds a send -> gain -> width -> limiter return mixed with the root "
                         "(buses consumed twice: the hybrid plan — voice banks on the voice-bank kernels, the rest on the level executor)")
    ap.add_argument("--voice-spatial", action="store_true",
                    help="cfg2/cfg5: a SPEC 3D spatialiser node at the end of every voice (the voice-bank plan's last stage, k_leaf_sum<.., SP>)")
    ap.add_argument("--voice-fx", action="store_true",
                    help="cfg2/cfg5: a StereoWidthNode + HardClipNode at the end of every voice chain")
    ap.add_argument("--rs-source", action="store_true",
                    help="cfg2/cfg5: the voices' sources are SPEC resamplers (looping, ratio U(0.5, 1.5)) instead of samplers")
    ap.add_argument("--chain-reordered", action="store_true",
                    help="cfg3: the voices as sampler -> gain -> biquad -> biquad -> delay -> pan (the chain plan's round-6 grammar); "
                         "with --source-format i16 on interleaved 16-bit sources")
    ap.add_argument("--force-generic", action="store_true",
                    help="run the workload on the generic level-batched executor (plan 0) instead of its fused plan")
    ap.add_argument("--variant", choices=["A", "B", "C"], default="A",
                    help="cfg2/cfg5 (SURVEY 8d): A steady; B one gain change per voice at a seeded block of the run "
                         "(smoother ramps, message path inside the timed region); C every 4th voice paused (silence masks)")
    ap.add_argument("--reduce-every", type=int, default=4,
                    help="N>1: steps whose mix buses share one collective (the reduction of R steps overlaps the next R)")
    ap.add_argument("--bus-reduce", choices=["exchange", "allreduce", "ordered", "allreduce_abi", "ordered_abi"], default="exchange",
                    help="N>1, the mix bus: libfwgpu's one-shot exchange over peer-mapped slots (rank-ordered, bit-exact; falls back to "
                         "the all-reduce when IPC / peer access is unavailable), the RCCL all-reduce (north_star's named path; "
                         "re-associates the sum for N > 2), or RCCL all-gather + rank-ordered sum kernel (bit-exact)")
    ap.add_argument("--share-device", action="store_true",
                    help="N>1 on a ONE-GPU box: all ranks on device 0 (separate processes, hipIpc between them, gloo as the control "
                         "plane because RCCL refuses two ranks on one device).  Proves the N > 1 path end to end; the aggregate is "
                         "NOT a scaling figure and the line says so")
    ap.add_argument("--contexts", type=int, default=0,
                    help="time the workload in this many FRESH contexts inside the one run and report the median context (default: 5 for "
                         "the default single-GPU line, 1 otherwise)")
    ap.add_argument("--force-other-configs", action="store_true", help="tests: emit other_configs / bus_reduce_modes for a non-default shape too")
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline = args.no_parity_check = args.no_other_configs = args.no_realtime = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    dV, dB, dK, dF, dS = DEFAULTS[args.workload]
    V = args.voices or dV
    B = args.block or dB
    K = args.blocks_per_step or dK
    F = args.src_frames or dF
    steps = args.steps or dS
    wl = args.workload
    default_shape = (wl == "cfg2" and (V, B, K, F) == (dV, dB, dK, dF) and args.source_format == "f32" and args.variant == "A" and
                     not (args.master or args.master_iir or args.voice_fx or args.voice_spatial or args.send or args.rs_source or args.force_generic or args.host_buffers or args.chain_reordered))

    # stdout carries exactly ONE line (the JSON, rank 0): everything else that writes to fd 1 — RCCL's version banner
    # and warnings come from C stdio, flushed whenever — is sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FWGPU_BENCH_HOSTONLY (CPU test tier only, tests/test_bench_launch.py): the orchestration of this file — launcher,
    # ranks, reducer, the one JSON line — on the host-only harness library and gloo.  No audio is computed, and the line
    # says so instead of carrying a value.
    hostonly = bool(os.environ.get("FWGPU_BENCH_HOSTONLY"))
    if not hostonly:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product has no CPU path)")
        if args.share_device:
            local_rank = 0
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE is %d: launch as many ranks as GPUs (or let --gpus N start them)"
                         % (args.gpus, world))
    dist = None
    ranks_seen = 1
    if world > 1 or os.environ.get("FWGPU_BENCH_FORCE_DIST"):  # the env var: exercise the RCCL path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # The CONTROL plane (barriers, the max over ranks, the exchange handles, parity gathers) is gloo on host tensors; the
        # mix bus itself moves through libfwgpu's exchange (hipIpc / xGMI stores) or, in the two RCCL modes, through an RCCL
        # group created when one of them is first used (rccl_group below) — so the default path does not depend on RCCL coming up.
        dist.init_process_group(backend="gloo")
        ranks_seen = dist.get_world_size()

    import firewheel_amd as fa
    from firewheel_amd import shard

    env = {"torch": torch, "fa": fa, "shard": shard, "dist": dist, "rank": rank, "world": world, "device": 0 if hostonly else local_rank,
           "dev": "cpu" if hostonly else "cuda", "hostonly": hostonly, "ctrl_cpu": True,
           "share_device": bool(args.share_device and not hostonly), "local_rank": local_rank, "rccl": None, "rccl_ranks_seen": 0}
    repeats = []
    for _ in range(max(0, args.repeat_first)):  # diagnostic: the same run, same process, fresh allocations each time
        r0 = run_workload(env, args, wl, V, B, K, F, steps, args.warmup, full=False, warm_ms=args.warm_ms)
        repeats.append({"ms_per_step": r0["ms_per_step"], "kernel_us": (r0["roofline"] or {}).get("avg_launch_us")})
    # The headline kernel runs in one of two HBM placement states, fixed per context when its buffers are allocated (DESIGN.md
    # §7): one context is a coin toss.  So the default line times the same workload in `--contexts` FRESH contexts (fresh bus /
    # table allocations each) inside this one run and reports the MEDIAN context; all of them are in the line.
    n_ctx = args.contexts if args.contexts else (5 if (world == 1 and default_shape and not hostonly) else 1)
    ctx_runs = []
    for _ in range(max(0, n_ctx - 1)):
        r0 = run_workload(env, args, wl, V, B, K, F, steps, args.warmup, full=False, warm_ms=args.warm_ms)
        ctx_runs.append(r0)
    runtime_fallback = None
    try:
        res = run_workload(env, args, wl, V, B, K, F, steps, args.warmup, full=True, warm_ms=args.warm_ms)
    except ExchangeFailed as ex:  # (collective: raised on every rank)
        if env["share_device"]:
            raise
        runtime_fallback = "exchange failed during the run (%s) -> the workload was run again over the RCCL all-reduce" % ex
        args.bus_reduce = "allreduce"
        res = run_workload(env, args, wl, V, B, K, F, steps, args.warmup, full=True, warm_ms=args.warm_ms)
        if res is not None:
            res["config"]["bus_reduce_fallback"] = runtime_fallback
    line = None
    contexts = None
    if rank == 0 and ctx_runs:
        ctx_runs.append(res)
        order = sorted(range(len(ctx_runs)), key=lambda i: ctx_runs[i]["ms_per_step"])
        mid = order[len(order) // 2]
        contexts = context_summary(ctx_runs, mid)
        for k in ("value", "ms_per_step", "roofline", "step_time_distribution"):
            res[k] = ctx_runs[mid].get(k)
    if rank == 0:
        line = {
            "metric": "stereo voice-samples/sec @ block=256, 48kHz; % HBM roofline; 1/2/4/8 GPU",
            "value": res["value"],
            "unit": "voice-samples/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.host_buffers else "synthetic; output delivered to HOST buffers (PCIe-inclusive)",
            "config": res["config"],
            "roofline": res["roofline"],
            "step_time_distribution": res.get("step_time_distribution"),
            "cpu_baseline": res.get("cpu_baseline"),
            "parity_check": res.get("parity_check"),
            "parity_check_timed_context": res.get("parity_check_timed_context"),
            "realtime_us_per_callback": res.get("realtime_us_per_callback"),
            "realtime_us_per_callback_from_python": res.get("realtime_us_per_callback_from_python"),
            "realtime_path": res.get("realtime_path"),
            "ranks_seen": ranks_seen,
        }
        if args.share_device and world > 1:
            line["virtual_ranks_on_one_device"] = True
            line["note"] = ("--share-device: %d processes on ONE GPU (hipIpc between them, gloo control plane) — proves the N > 1 path; "
                            "`value` is the aggregate of ranks that time-share one device, not a scaling figure" % world)
        if contexts:
            line["contexts"] = contexts
        if repeats:
            line["repeats_before"] = repeats
        if hostonly:
            line["value"] = None
            line["invalid"] = "host-only harness (FWGPU_BENCH_HOSTONLY): orchestration test, no audio computed, not a measurement"
        if world == 1 and default_shape and not hostonly and not args.no_realtime:
            # SURVEY §8d "including the D2H of the mix bus": the literal process_interleaved boundary — synchronous, K x 2 KiB of
            # interleaved output over PCIe into pageable host memory per call.  Never `value`.
            import copy

            ah = copy.copy(args)
            ah.host_buffers = True
            try:
                rh = run_workload(env, ah, wl, V, B, K, F, 10, 2, full=False)
                line["value_host_buffers"] = {"value": rh["value"], "unit": "voice-samples/s", "ms_per_step": rh["ms_per_step"], "steps": 10,
                                              "what": "fwgpu_process_interleaved: host output buffers, synchronous, PCIe-inclusive (%d KiB D2H per call)" % (K * B * 8 // 1024)}
                ah.host_async = True
                ra = run_workload(env, ah, wl, V, B, K, F, 20, 3, full=False)
                line["value_host_buffers"]["pipelined"] = {
                    "value": ra["value"], "ms_per_step": ra["ms_per_step"], "steps": 20, "frac_of_value": ra["value"] / res["value"],
                    "what": "fwgpu_process_interleaved_begin / _end: call n + 1 begun before call n is ended — the graph-output kernel writes into "
                            "mapped host staging, the host's wait + memcpy of call n overlap the rendering of call n + 1; the frames still land "
                            "in pageable host memory"}
            except Exception as ex:  # noqa: BLE001
                line["value_host_buffers"] = {"error": repr(ex)}
        if world == 1 and default_shape and not args.no_other_configs and not hostonly:
            line["other_configs"] = other_configs(env, args)
    if world > 1 and (default_shape or args.force_other_configs) and not args.no_other_configs:
        extra = other_configs_multi(env, args)  # collective: every rank runs them, rank 0 keeps the results
        if rank == 0:
            line.update(extra)
    if rank == 0:
        line["rccl_ranks_seen"] = env["rccl_ranks_seen"]  # ranks of the RCCL group, if a run of this line created one (the two RCCL reductions)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        full_path = write_full(line, world)  # (stderr first: the compact line is the last thing any captured stream ends with)
        os.write(json_fd, (compact_line(line, full_path) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
