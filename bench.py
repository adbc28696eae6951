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

Sources are planar f32, resident in HBM, each voice looping over its own buffer so every block streams fresh
HBM.  One "step" = one fwgpu_process_blocks_device call of `--blocks-per-step` consecutive blocks (the K-block
throughput mode); the interleaved mix bus stays in HBM.  With N > 1 every rank runs the same
shard (weak scaling, one process per GPU) and the step ends with the mix-bus collective.

Prints ONE JSON line (rank 0).  `roofline` times the dominant kernel with HIP events on the stream it runs on;
`cpu_baseline` times the oracle (C++ restatement of the reference's single-threaded executor) on a bounded
sample of the same workload.
"""
import argparse
import json
import os
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


MASTER = [False]  # --master: a master volume + limiter between the root SumNode and graph_out


def sum_tree(cx, fa, ends, radix):
    level = ends
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = cx.add_node(2 * len(grp), 2, fa.SumNode())
            for p, n in enumerate(grp):
                cx.connect(n, 0, m, 2 * p, False)
                cx.connect(n, 1, m, 2 * p + 1, False)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    cur = level[0]
    if MASTER[0]:
        for node in (fa.VolumeNode(70.0), fa.HardClipNode(-1.0)):
            m = cx.add_node(2, 2, node)
            cx.connect(cur, 0, m, 0, False)
            cx.connect(cur, 1, m, 1, False)
            cur = m
    cx.connect(cur, 0, cx.graph_out_node(), 0, False)
    cx.connect(cur, 1, cx.graph_out_node(), 1, False)


def start_voices(cx, fa, samplers, src, frames_per_voice, fmt="f32"):
    elem = 4 if fmt == "f32" else 2
    sfmt = fa.SampleFormat.PLANAR_F32 if fmt == "f32" else fa.SampleFormat.INTERLEAVED_I16
    for v, s in enumerate(samplers):
        ptr = src.data_ptr() + v * 2 * frames_per_voice * elem
        smp = cx.new_sample_device(sfmt, 2, frames_per_voice, ptr)
        node = cx.node(s)
        node.set_sample(smp, False)
        node.set_loop_range(fa.LoopRange.Full())
        node.play()


def build_bank(cx, fa, voices, radix, seed=0, volumes_out=None):
    """cfg2 / cfg5 chain: sampler -> gain -> pan."""
    import numpy as np

    rng = np.random.default_rng(1234 + seed)
    ends, samplers = [], []
    for v in range(voices):
        s = cx.add_node(0, 2, fa.SamplerNode(100.0))
        vol = cx.add_node(2, 2, fa.VolumeNode(float(rng.uniform(10, 100))))
        pan = cx.add_node(2, 2, fa.StereoPanNode(float(rng.uniform(-1, 1))))
        for c in (0, 1):
            cx.connect(s, c, vol, c, False)
            cx.connect(vol, c, pan, c, False)
        samplers.append(s)
        ends.append(pan)
        if volumes_out is not None:
            volumes_out.append(vol)
    sum_tree(cx, fa, ends, radix)
    cx.update()
    return samplers


def build_chain_bank(cx, fa, voices, radix, seed=0):
    """cfg3 chain: sampler -> biquad LPF (cutoff U(200, 8000) Hz, Q 0.707) -> delay (U(10, 250) ms, feedback 0.3,
    mix 0.5) -> gain."""
    import numpy as np

    rng = np.random.default_rng(4321 + seed)
    ends, samplers = [], []
    for v in range(voices):
        s = cx.add_node(0, 2, fa.SamplerNode(100.0))
        bq = cx.add_node(2, 2, fa.BiquadNode(fa.BiquadNode.LOWPASS, float(rng.uniform(200, 8000)), 0.707))
        dl = cx.add_node(2, 2, fa.DelayNode(float(rng.uniform(0.010, 0.250)), 0.3, 0.5))
        vol = cx.add_node(2, 2, fa.VolumeNode(float(rng.uniform(10, 100))))
        for c in (0, 1):
            cx.connect(s, c, bq, c, False)
            cx.connect(bq, c, dl, c, False)
            cx.connect(dl, c, vol, c, False)
        samplers.append(s)
        ends.append(vol)
    sum_tree(cx, fa, ends, radix)
    cx.update()
    return samplers


def reverb_ir(taps):
    import numpy as np

    n = np.arange(taps, dtype=np.float64)
    rng = np.random.default_rng(4)
    h = (rng.uniform(-1, 1, size=(2, taps)) * np.exp(-n / 16384.0)[None, :])
    return (h / np.abs(h).sum(axis=1, keepdims=True)).astype(np.float32)


def build_reverb_bank(cx, fa, voices, radix, taps):
    """cfg4: V x (sampler -> `taps`-tap stereo FIR convolution) -> radix sum tree -> out."""
    ir = cx.new_sample(fa.SampleFormat.PLANAR_F32, 2, reverb_ir(taps))
    ends, samplers = [], []
    for v in range(voices):
        s = cx.add_node(0, 2, fa.SamplerNode(100.0))
        f = cx.add_node(2, 2, fa.FirReverbNode(ir))
        for c in (0, 1):
            cx.connect(s, c, f, c, False)
        samplers.append(s)
        ends.append(f)
    sum_tree(cx, fa, ends, radix)
    cx.update()
    return samplers


def cpu_engine(workload, voices, block, radix, taps, src_frames, seed=0):
    """one oracle engine running `voices` voices of the workload's graph shape; returns (engine, voices, chunk)"""
    import fwapi
    import scenarios

    e = fwapi.OracleEngine(max_block_frames=block)
    chunk = 16
    if workload == "cfg3":
        vs = scenarios.build_chain_bank(e, voices, radix=radix, src_frames=src_frames, min_delay_frames=480,
                                        max_delay_frames=12000, seed=seed)
        chunk = 2
    elif workload == "cfg4":
        ir = e.new_sample(fwapi.PLANAR_F32, 2, reverb_ir(taps))
        m = e.sum(voices)
        vs = []
        for v in range(voices):
            s = e.sampler(100.0)
            f = e.fir(ir)
            e.connect_stereo(s, f)
            e.connect_stereo(f, m, 2 * v)
            vs.append(dict(sampler=s))
        e.connect_stereo(m, e.graph_out_node)
        e.update()
        for v, vc in enumerate(vs):
            e.sampler_set_sample(vc["sampler"],
                                 e.new_sample(fwapi.PLANAR_F32, 2, scenarios.voice_source(seed * 100000 + v, src_frames)))
        chunk = 1
    else:
        vs = scenarios.build_voice_bank(e, voices, radix=radix, src_frames=src_frames, seed=seed)
    for vc in vs:
        e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_FULL)
        e.sampler_play(vc["sampler"])
    return e, chunk


def cpu_baseline(workload, voices, block, radix, taps, target_secs):
    """Oracle on the same graph shape.  `value` is the faithful figure: ONE thread, like the reference's audio thread
    (DESIGN_DOC.md:48).  `all_cores` is the generous one: the voices split over every host core, one
    oracle engine per thread, no mix-bus exchange charged."""
    import threading

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    src_frames = 16384
    if workload == "cfg4":
        voices = min(voices, 32)  # bounded sample: the scalar direct-form convolution is ~1e9 fmaf per voice-block
    e, chunk = cpu_engine(workload, voices, block, radix, taps, src_frames)
    if workload != "cfg4":
        e.process_blocks(4)  # warm-up
    n_blocks, t = 0, 0.0
    t0 = time.perf_counter()
    while t < target_secs:
        e.process_blocks(chunk)
        n_blocks += chunk
        t = time.perf_counter() - t0
    del e
    ncpu = os.cpu_count() or 1
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    all_cores = None
    T = min(ncpu, voices, 32)  # python threads around ctypes calls: beyond ~32 the GIL hand-offs between calls dominate
    if T > 1:
        per = max(1, voices // T)
        engines = [cpu_engine(workload, per, block, radix, taps, src_frames, seed=1 + i) for i in range(T)]
        counts = [0] * T
        go = threading.Event()
        deadline = [0.0]

        def run(i):  # ctypes releases the GIL for the whole of a process call
            eng, ch = engines[i]
            ch *= 8  # long calls: the GIL is only held between them
            go.wait()
            while time.perf_counter() < deadline[0]:
                eng.process_blocks(ch)
                counts[i] += ch

        th = [threading.Thread(target=run, args=(i,)) for i in range(T)]
        for x in th:
            x.start()
        secs = max(2.0, target_secs / 3.0)
        ta = time.perf_counter()
        deadline[0] = ta + secs
        go.set()
        for x in th:
            x.join()
        tb = time.perf_counter() - ta
        all_cores = {"value": per * block * sum(counts) / tb, "unit": "voice-samples/s", "cores": T,
                     "sample": "%d threads x %d voices, %.1f s" % (T, per, tb)}
    return {
        "value": voices * block * n_blocks / t,
        "unit": "voice-samples/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d blocks of a %d-voice %s graph (block=%d, %d-frame looping sources), %.1f s on 1 of %d host cores; "
                  "oracle = C++ restatement of the reference's single-threaded executor (the Rust build is not "
                  "available: no cargo/rustc)" % (n_blocks, voices, workload, block, src_frames, t, ncpu),
        "all_cores": all_cores,
    }


def pmc_traffic(kernel, V, B, K):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
    gfx950 corrections per the microarch guide) — quoted only when collected on this exact workload."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic*.json")), reverse=True):
        try:
            pm = json.load(open(path))
            w = pm["workload"]
            if (w["voices"], w["block"], w["blocks_per_step"]) == (V, B, K) and kernel in pm:
                return pm[kernel]["traffic_bytes"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(DEFAULTS), default="cfg2",
                    help="cfg2 = the headline (BASELINE configs[1]); cfg3 / cfg4 / cfg5 = configs[2..4]")
    ap.add_argument("--voices", type=int, default=None, help="voices per GPU")
    ap.add_argument("--block", type=int, default=None)
    ap.add_argument("--radix", type=int, default=32)
    ap.add_argument("--blocks-per-step", type=int, default=None)
    ap.add_argument("--src-frames", type=int, default=None, help="source frames per voice (2 ch f32)")
    ap.add_argument("--cpu-secs", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--taps", type=int, default=65536)
    ap.add_argument("--source-format", choices=["f32", "i16"], default="f32",
                    help="cfg2/cfg5 only: planar f32 sources (the headline) or interleaved stereo i16 (4 B per voice-sample)")
    ap.add_argument("--host-buffers", action="store_true",
                    help="time fwgpu_process_interleaved on HOST buffers instead (PCIe-inclusive; DESIGN.md §7 note, "
                         "never the headline)")
    ap.add_argument("--master", action="store_true",
                    help="put a master VolumeNode + HardClipNode between the root SumNode and graph_out (the fused plans "
                         "then run that chain with the generic node kernel on the mix bus)")
    ap.add_argument("--force-generic", action="store_true",
                    help="run the workload on the generic level-batched executor (plan 0) instead of its fused plan")
    ap.add_argument("--variant", choices=["A", "B", "C"], default="A",
                    help="cfg2/cfg5 (SURVEY 8d): A steady; B one gain change per voice at a seeded block of the run "
                         "(smoother ramps, message path inside the timed region); C every 4th voice paused (silence masks)")
    ap.add_argument("--reduce-every", type=int, default=4,
                    help="N>1: steps whose mix buses share one collective (the reduction of R steps overlaps the next R)")
    ap.add_argument("--bus-reduce", choices=["allreduce", "ordered"], default="allreduce",
                    help="N>1: RCCL all-reduce (named path) or all-gather + rank-ordered sum (bit-exact)")
    args = ap.parse_args()
    MASTER[0] = args.master
    dV, dB, dK, dF, dS = DEFAULTS[args.workload]
    V = args.voices or dV
    B = args.block or dB
    K = args.blocks_per_step or dK
    F = args.src_frames or dF
    steps = args.steps or dS
    wl = args.workload

    # stdout carries exactly ONE line (the JSON, rank 0): everything else that writes to fd 1 — RCCL's version banner
    # and warnings come from C stdio, flushed whenever — is sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("FWGPU_BENCH_FORCE_DIST"):  # the env var: exercise the RCCL path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    import firewheel_amd as fa
    from firewheel_amd import shard

    stream = torch.cuda.current_stream().cuda_stream
    cx = fa.FirewheelGpuCtx(48000, B, 0, 2, device=local_rank, stream=stream)
    cx.set_max_batch(K)
    if args.force_generic:
        cx.set_force_generic(True)
    # synthetic sources, generated in HBM: uniform(-1,1) f32, seed offset by global voice id
    g = torch.Generator(device="cuda")
    g.manual_seed(shard.voice_seed(rank * V))  # stream keyed by the shard's first GLOBAL voice id
    sfmt = args.source_format if wl in ("cfg2", "cfg5") else "f32"
    if sfmt == "i16":  # interleaved stereo PCM, [voice][frame][channel]
        src = torch.randint(-32768, 32768, (V, F, 2), dtype=torch.int16, device="cuda", generator=g)
    else:
        src = torch.empty((V, 2, F), dtype=torch.float32, device="cuda")
        src.uniform_(-1.0, 1.0, generator=g)
    if wl == "cfg4":
        samplers = build_reverb_bank(cx, fa, V, args.radix, args.taps)
        want_plan = 0
    elif wl == "cfg3":
        samplers = build_chain_bank(cx, fa, V, args.radix, seed=rank)
        want_plan = 2
    else:
        volumes = []
        samplers = build_bank(cx, fa, V, args.radix, seed=rank, volumes_out=volumes)
        want_plan = 1
    start_voices(cx, fa, samplers, src, F, sfmt)
    variant = args.variant if wl in ("cfg2", "cfg5") else "A"
    playing = 1.0
    changes = {}
    if variant == "C":
        for s in samplers[::4]:
            cx.node(s).pause()
        playing = 1.0 - len(samplers[::4]) / float(len(samplers))
    elif variant == "B":  # voice v changes its gain once, at block b_v of timed step s_v
        import numpy as np

        rng = np.random.default_rng(99 + rank)
        for v, vol in enumerate(volumes):
            changes.setdefault(args.warmup + int(rng.integers(0, steps)), []).append(
                (vol, float(rng.uniform(10, 100)), int(rng.integers(0, K))))
    if args.force_generic:
        want_plan = 0
    assert cx.plan_kind() == want_plan, "expected launch plan %d, got %d" % (want_plan, cx.plan_kind())
    # two bus buffers: with N > 1 the reduction of step i (RCCL, its own stream) overlaps the compute of step i+1 —
    # the mix bus is a sink, nothing in a shard reads it back
    # ... and each buffer holds the buses of R consecutive steps, reduced by ONE collective: the hand-over between the
    # compute stream and RCCL's stream costs ~10 us of idle GPU per event on this stack (measured: 20 us per step with a
    # collective per step), so it is paid once per R steps
    R = max(1, args.reduce_every) if dist is not None else 1
    step_elems = K * B * 2
    outs = [torch.empty(R * step_elems, dtype=torch.float32, device="cuda") for _ in range(2)]
    reducer = shard.BusReducer(dist, outs, args.bus_reduce) if dist is not None else None
    step_no = [0]
    slot = [0]  # bus slot counter: buffer (slot // R) % 2, slice slot % R
    host_out = None
    if args.host_buffers:
        import numpy as np

        host_out = np.empty(K * B * 2, dtype=np.float32)

    def step():
        b = (slot[0] // R) % 2
        r = slot[0] % R
        for vol, pct, at in changes.get(step_no[0], ()):
            cx.node(vol).set_percent_volume(pct, at_block=at)
        step_no[0] += 1
        slot[0] += 1
        if reducer is not None and r == 0:
            reducer.wait(b)  # the collective that last used this buffer (2R steps ago)
        if args.host_buffers:  # the literal process_interleaved boundary: pageable host output, synchronous
            import ctypes as C

            rc = cx.L.fwgpu_process_interleaved(cx.c, None, host_out.ctypes.data_as(C.POINTER(C.c_float)), 0, 2, K * B, 0.0, 0)
            assert rc == 0, rc
            return
        cx.process_blocks_device(K, outs[b].data_ptr() + r * step_elems * 4, 2)
        if reducer is not None and r == R - 1:  # the mix bus: one collective per R steps over R x K x 2 x block f32
            reducer.submit(b)

    def finish_reductions():
        """submit the partly filled buffer (steps % R != 0), then wait for every collective"""
        if reducer is None:
            return
        if slot[0] % R != 0:
            reducer.submit((slot[0] // R) % 2)
            slot[0] += R - slot[0] % R
        reducer.wait_all()

    for _ in range(args.warmup):
        step()
    finish_reductions()
    timing = not args.no_kernel_timing
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    finish_reductions()  # every bus of the timed region is fully reduced before the clock stops
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

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
        torch.cuda.synchronize()
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
            traffic, traffic_src = pmc_traffic("k_fir_gemm", V, B, K)
            roofline = {"bound": "mfma", "kernel": "k_fir_gemm", "achieved": ach,
                        "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF, "traffic": traffic,
                        "traffic_source": traffic_src, "algorithmic_flops_per_launch": flops, "avg_launch_us": avg_s * 1e6,
                        "launches": fir_n, "blocks_per_launch": K, "timing": "HIP events, separate pass after the timed region",
                        "whole_block_us_all_kernels": gen_ms / max(gen_n, 1) / K * 1e3}
        elif dom_n:
            # source L+R once (f32: 8 B, i16: 4 B) (+ delay ring read + write)
            per_vs = 24.0 if wl == "cfg3" else (4.0 if sfmt == "i16" else 8.0)
            kernel = "k_chain" if wl == "cfg3" else "k_leaf_sum"
            k_launch = min(K, 64) if wl == "cfg3" else K  # the chain plan renders at most 64 blocks per k_chain launch
            alg_bytes = V * B * k_launch * per_vs * playing  # paused voices (variant C) fetch nothing
            avg_s = dom_ms / dom_n / 1e3
            ach = alg_bytes / avg_s / 1e9
            traffic, traffic_src = pmc_traffic(kernel, V, B, K) if sfmt == "f32" else (None, None)  # PMC passes ran on f32 sources
            roofline = {
                "bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_voice_sample": per_vs, "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_us": avg_s * 1e6, "launches": dom_n, "timing": "HIP events, separate pass after the timed region",
                "other_kernels_us_per_step": {"k_voice_control": ctl_ms / max(ctl_n, 1) * 1e3,
                                              "upper_sums+graph_out": up_ms / max(up_n, 1) * 1e3},
            }

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
        line = {
            "metric": "stereo voice-samples/sec @ block=256, 48kHz; % HBM roofline; 1/2/4/8 GPU",
            "value": total / dt,
            "unit": "voice-samples/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.host_buffers else "synthetic; output delivered to HOST buffers (PCIe-inclusive)",
            "config": {
                "workload": "%s, block=%d @48kHz, %s sources in HBM (%d frames/voice, looping)"
                            % (desc, B, "planar f32" if sfmt == "f32" else "interleaved stereo i16", F),
                "voices_per_gpu": V, "block": B, "blocks_per_step": K, "variant": variant, "master_chain": bool(args.master), "parallelism": "voice-shard x%d%s" %
                (world, (" + RCCL mix-bus %s" % args.bus_reduce) if world > 1 else ""),
                "realtime_factor": (total / dt) / (48000.0 * V * world),
                "device": name, "compute_units": cus,
            },
            "roofline": roofline,
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl, V, B, args.radix, args.taps, args.cpu_secs)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
