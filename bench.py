#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json on MI355X: stereo voice-samples/sec @ block=256, 48 kHz.

Workload at N=1 (BASELINE.json configs[1]): 1024 stereo voices, each sampler -> gain (VolumeNode) -> pan
-> radix-32 SumNode tree (32 + 1) -> graph_out, block = 256 frames, planar f32 sources resident in HBM,
every voice looping over its own 2 MiB-per-channel source so each block streams fresh HBM (total source
2 GiB >> the 256 MiB Infinity Cache).  One "step" = one fwgpu_process_blocks_device call of
`--blocks-per-step` consecutive blocks (the K-block throughput mode, DESIGN.md §launch plan); the output
(interleaved mix bus) stays in HBM.  With N > 1 every rank runs the same shard (weak scaling, one process per
GPU) and the step ends with the mix-bus all-reduce over RCCL.

Prints ONE JSON line (rank 0).  `roofline` times the dominant kernel (k_leaf_sum) with HIP events on the
stream it runs on; `cpu_baseline` times the oracle (C++ restatement of the reference's single-threaded
executor) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md:35 (spec; 6.29 TB/s measured float4 copy)


def build_bank(cx, fa, voices, radix, src, frames_per_voice, seed=0):
    """cfg2 graph through the reference-shaped API (AudioGraph::add_node / connect)."""
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
    cx.connect(level[0], 0, cx.graph_out_node(), 0, False)
    cx.connect(level[0], 1, cx.graph_out_node(), 1, False)
    cx.update()
    elem = 4
    for v, s in enumerate(samplers):
        ptr = src.data_ptr() + v * 2 * frames_per_voice * elem
        smp = cx.new_sample_device(fa.SampleFormat.PLANAR_F32, 2, frames_per_voice, ptr)
        node = cx.node(s)
        node.set_sample(smp, False)
        node.set_loop_range(fa.LoopRange.Full())
        node.play()
    return samplers


def build_reverb_bank(cx, fa, voices, radix, src, frames_per_voice, taps, torch):
    """cfg4: V x (sampler -> 65536-tap stereo FIR convolution) -> radix sum tree -> out."""
    import numpy as np

    n = np.arange(taps, dtype=np.float64)
    rng = np.random.default_rng(4)
    h = (rng.uniform(-1, 1, size=(2, taps)) * np.exp(-n / 16384.0)[None, :])
    h = (h / np.abs(h).sum(axis=1, keepdims=True)).astype(np.float32)
    ir = cx.new_sample(fa.SampleFormat.PLANAR_F32, 2, h)
    from firewheel_amd.graph import _RawNode

    ends, samplers = [], []
    for v in range(voices):
        s = cx.add_node(0, 2, fa.SamplerNode(100.0))
        f = cx.add_node(2, 2, _RawNode(12, [float(ir)]))
        for c in (0, 1):
            cx.connect(s, c, f, c, False)
        samplers.append(s)
        ends.append(f)
    level = ends
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = cx.add_node(2 * len(grp), 2, fa.SumNode())
            for p, nd in enumerate(grp):
                cx.connect(nd, 0, m, 2 * p, False)
                cx.connect(nd, 1, m, 2 * p + 1, False)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    cx.connect(level[0], 0, cx.graph_out_node(), 0, False)
    cx.connect(level[0], 1, cx.graph_out_node(), 1, False)
    cx.update()
    for v, s in enumerate(samplers):
        ptr = src.data_ptr() + v * 2 * frames_per_voice * 4
        smp = cx.new_sample_device(fa.SampleFormat.PLANAR_F32, 2, frames_per_voice, ptr)
        node = cx.node(s)
        node.set_sample(smp, False)
        node.set_loop_range(fa.LoopRange.Full())
        node.play()
    return samplers


def cpu_baseline(voices, block, radix, target_secs):
    """Oracle (single thread, like the reference's audio thread: DESIGN_DOC.md:48) on the same graph shape."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np

    import fwapi
    import scenarios

    e = fwapi.OracleEngine(max_block_frames=block)
    src_frames = 16384
    vs = scenarios.build_voice_bank(e, voices, radix=radix, src_frames=src_frames)
    for vc in vs:
        e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_FULL)
        e.sampler_play(vc["sampler"])
    e.process_blocks(4)  # warm-up
    n_blocks, t = 0, 0.0
    chunk = 16
    t0 = time.perf_counter()
    while t < target_secs:
        e.process_blocks(chunk)
        n_blocks += chunk
        t = time.perf_counter() - t0
    return {
        "value": voices * block * n_blocks / t,
        "unit": "voice-samples/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d blocks of the same %d-voice cfg2 graph (block=%d, %d-frame looping sources), %.1f s on 1 of %d host cores"
                  % (n_blocks, voices, block, src_frames, t, os.cpu_count() or 0),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--voices", type=int, default=1024, help="voices per GPU")
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--radix", type=int, default=32)
    ap.add_argument("--blocks-per-step", type=int, default=256)
    ap.add_argument("--src-frames", type=int, default=262144, help="source frames per voice (2 ch f32)")
    ap.add_argument("--cpu-secs", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--workload", choices=["cfg2", "cfg4"], default="cfg2",
                    help="cfg2 = the headline (BASELINE configs[1]); cfg4 = 256-voice 65536-tap FIR reverb (MFMA)")
    ap.add_argument("--taps", type=int, default=65536)
    ap.add_argument("--bus-reduce", choices=["allreduce", "ordered"], default="allreduce",
                    help="N>1: RCCL all-reduce (named path) or all-gather + rank-ordered sum (bit-exact)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    import firewheel_amd as fa
    from firewheel_amd import shard

    if args.workload == "cfg4":
        if args.voices == 1024:
            args.voices = 256
        if args.blocks_per_step == 256:
            args.blocks_per_step = 4
        args.src_frames = min(args.src_frames, 65536)
        args.no_cpu_baseline = True
    V, B, K = args.voices, args.block, args.blocks_per_step
    stream = torch.cuda.current_stream().cuda_stream
    cx = fa.FirewheelGpuCtx(48000, B, 0, 2, device=local_rank, stream=stream)
    cx.set_max_batch(K)
    # synthetic sources, generated in HBM: uniform(-1,1) f32, seed offset by global voice id
    g = torch.Generator(device="cuda")
    g.manual_seed(shard.voice_seed(rank * V))  # stream keyed by the shard's first GLOBAL voice id
    src = torch.empty((V, 2, args.src_frames), dtype=torch.float32, device="cuda")
    src.uniform_(-1.0, 1.0, generator=g)
    if args.workload == "cfg4":
        build_reverb_bank(cx, fa, V, args.radix, src, args.src_frames, args.taps, torch)
    else:
        build_bank(cx, fa, V, args.radix, src, args.src_frames, seed=rank)
        assert cx.plan_kind() == 1, "fused voice-bank plan was not selected"
    out = torch.empty(K * B * 2, dtype=torch.float32, device="cuda")

    def step():
        cx.process_blocks_device(K, out.data_ptr(), 2)
        if dist is not None:  # the mix bus: one collective per step over K x 2 x block f32 (K x 2 KiB)
            if args.bus_reduce == "allreduce":
                shard.reduce_bus_allreduce(out, dist)
            else:
                shard.reduce_bus_ordered(out, dist)

    for _ in range(args.warmup):
        step()
    timing = not args.no_kernel_timing
    if timing:
        cx.timing_reset()
        cx.timing_enable(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
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
        cx.timing_enable(False)
        leaf_ms, leaf_n = cx.timing_read(0)
        ctl_ms, ctl_n = cx.timing_read(1)
        up_ms, up_n = cx.timing_read(2)
        alg_bytes = V * B * K * 8.0  # 8 B per stereo voice-sample (L+R f32 source read once)
        gen_ms, gen_n = cx.timing_read(3)
        if args.workload == "cfg4" and gen_n:
            # all level kernels + the FIR GEMM of one block; the GEMM dominates (profiles/r01_cfg4_kernel_stats.csv)
            flops = 2.0 * 2 * args.taps * V * B  # direct-form definition: 2 ch x 2 flop x T per voice-sample
            avg_s = gen_ms / gen_n / 1e3
            ach = flops / avg_s / 1e12
            roofline = {"bound": "mfma", "kernel": "k_fir_gemm (+ level kernels of the block)", "achieved": ach,
                        "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None,
                        "algorithmic_flops_per_block": flops, "avg_block_us": avg_s * 1e6, "blocks": gen_n}
        if leaf_n:
            avg_s = leaf_ms / leaf_n / 1e3
            ach = alg_bytes / avg_s / 1e9
            # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
            # runs, gfx950 FETCH_SIZE x2 correction) — only quoted when it was collected on this exact workload
            traffic, traffic_src = None, None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
                w = pm["workload"]
                if (w["voices"], w["block"], w["blocks_per_step"]) == (V, B, K):
                    traffic = pm["k_leaf_sum"]["traffic_bytes"]
                    traffic_src = "profiles/r01_pmc_hbm_traffic.json"
            except Exception:
                pass
            roofline = {
                "bound": "hbm", "kernel": "k_leaf_sum", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": avg_s * 1e6, "launches": leaf_n,
                "other_kernels_us_per_step": {"k_voice_control": ctl_ms / max(ctl_n, 1) * 1e3,
                                              "upper_sums+graph_out": up_ms / max(up_n, 1) * 1e3},
            }

    if rank == 0:
        total = float(V) * B * K * args.steps * world
        name, cus, hbm = cx.device_info()
        line = {
            "metric": "stereo voice-samples/sec @ block=256, 48kHz",
            "value": total / dt,
            "unit": "voice-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("cfg2: %d stereo voices/GPU, sampler->gain->pan->radix-%d sum tree, block=%d @48kHz, "
                             "planar f32 sources in HBM (%d frames/voice, looping)" % (V, args.radix, B, args.src_frames))
                if args.workload == "cfg2" else
                ("cfg4: %d stereo voices/GPU, sampler->%d-tap stereo FIR (f32 MFMA Toeplitz GEMM)->radix-%d sum tree, "
                 "block=%d @48kHz" % (V, args.taps, args.radix, B)),
                "voices_per_gpu": V, "block": B, "blocks_per_step": K, "parallelism": "voice-shard x%d%s" %
                (world, " + RCCL mix-bus all-reduce" if world > 1 else ""),
                "realtime_factor": (total / dt) / (48000.0 * V * world),
                "device": name, "compute_units": cus,
            },
            "roofline": roofline,
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(V, B, args.radix, args.cpu_secs)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
