"""GPU tier: seeded random graphs + random control traffic, every launch plan against the oracle, bit for bit.

Each seed draws a voice bank (1-70 voices, random tree radix, random per-voice chain: gain stages and / or biquad and /
or delay, random sample formats and lengths, mono and stereo), an optional master chain on the mix bus, and several
rounds of messages (play / pause / stop / gains / pans / playheads / loop ranges, tagged at random blocks of the next
call) between calls of random length — the things the hand-written scenarios combine by design, combined by chance.
The same draw runs on the oracle and on the GPU (fused plan if the graph qualifies, generic executor if not, and the
generic executor again when forced), through the synchronous host-buffer call or the asynchronous device call."""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before libfwgpu is loaded: one process, one copy of the HIP runtime — torch brings its own)

import fwapi
import scenarios
from fwapi import (INTERLEAVED_F32, INTERLEAVED_I16, INTERLEAVED_U16, LOOP_FULL, LOOP_NONE, LOOP_RANGE_SECS, PLANAR_F32,
                   PLANAR_I16, PLANAR_U16, GpuEngine)
from test_gpu_parity import assert_bits_equal, oracle

pytestmark = pytest.mark.gpu


def encode(data, fmt):
    """(channels, frames) f32 -> the raw array of sample format `fmt`"""
    if fmt in (PLANAR_I16, INTERLEAVED_I16):
        raw = np.round(data * 32767).astype(np.int16)
    elif fmt in (PLANAR_U16, INTERLEAVED_U16):
        raw = np.round((data + 1) * 32767.5).astype(np.uint16)
    else:
        raw = data
    return raw.T.copy() if fmt <= INTERLEAVED_F32 else raw


def fuzz_run(e, seed):
    rng = np.random.default_rng(seed)
    mbf = e.max_block_frames
    n_voices = int(rng.integers(1, 71)) if rng.random() < 0.85 else int(rng.integers(71, 260))  # several leaves and levels
    radix = int(rng.choice([2, 3, 8, 32]))
    # 0 gains only, 1 + biquad, 2 + delay, 3 + both (each present in ~90 % of the voices), 4 = every voice has both and
    # delays of >= 3 k_chain tiles: calls without pending messages then take k_chain's steady-call loop
    shape = int(rng.integers(0, 5))
    tile = 128 if mbf % 128 == 0 else 64
    f32_only = rng.random() < 0.6  # (chain voices on other formats take k_chain's per-element fetch: valid, slower)
    spare_port = rng.random() < 0.5  # leaf SumNodes keep an unconnected port: voices are plugged in / out between calls
    voices, ends = [], []
    rng_m = np.random.default_rng(seed + 7_000_003)  # (a stream of its own: the graphs of the seeds run before round 5 keep their other draws)
    for v in range(n_voices):
        # round 5: one voice in eight is a ONE-output sampler behind the reference's MonoToStereoNode — a voice of the fused plans when its
        # chain is gains only, a refused shape (solo prefix + levels) behind a filter / delay / spatialiser
        mono_adapter = rng_m.random() < 0.125
        s = e.sampler(float(rng.uniform(30, 100)), n_out=1) if mono_adapter else e.sampler(float(rng.uniform(30, 100)))
        cur = s
        if mono_adapter:
            cur = e.add_node(fwapi.MONO_TO_STEREO, 1, 2)
            e.connect(s, 0, cur, 0)
        vc = dict(sampler=s, gains=[], pans=[], bq=None, dl=None)
        if shape == 4 or (shape in (1, 3) and rng.random() < 0.9):
            vc["bq"] = e.biquad(int(rng.integers(0, 3)), float(rng.uniform(200, 8000)), float(rng.uniform(0.5, 3.0)))
            e.connect_stereo(cur, vc["bq"])
            cur = vc["bq"]
        if shape == 4 or (shape in (2, 3) and rng.random() < 0.9):
            d_lo = 3 * tile if shape == 4 else 64
            vc["dl"] = e.delay(int(rng.integers(d_lo, d_lo + 640)) / float(e.sample_rate), feedback=float(rng.uniform(0, 0.6)),
                               mix=float(rng.uniform(0, 1)))
            e.connect_stereo(cur, vc["dl"])
            cur = vc["dl"]
        for _ in range(int(rng.integers(0, 4))):
            if rng.random() < 0.6:
                g = e.volume(float(rng.uniform(10, 120)))
                vc["gains"].append(g)
            else:
                g = e.pan(float(rng.uniform(-1, 1)))
                vc["pans"].append(g)
            e.connect_stereo(cur, g)
            cur = g
        if shape == 0 and rng.random() < 0.5:
            # dry voices: up to two more stages that are not plain gains — a stereo width and / or a hard clip anywhere in
            # the chain's tail (the voice-bank plan's stage programs; at most 5 chain stages per voice)
            for _ in range(int(rng.integers(1, 3))):
                if rng.random() < 0.5:
                    g = e.width(float(rng.uniform(0, 2)))
                    vc["pans"].append(g)  # (param 0, range [-1, 1] is inside the width's [0, inf) after its clamp)
                else:
                    g = e.hard_clip(float(rng.uniform(-18, 0)))
                e.connect_stereo(cur, g)
                cur = g
        if shape == 0 and rng.random() < 0.25:
            # ... and a 3D spatialiser as the voice's last node (the voice-bank plan's SK_SPATIAL stage: ear delays, 64-frame
            # history across blocks and calls; param 0 = x moves it left / right: gain glides + delay switches)
            g = e.spatial(float(rng.uniform(-4, 4)), float(rng.uniform(-1, 1)), float(rng.uniform(-4, 4)), n_in=2)
            vc["pans"].append(g)
            e.connect_stereo(cur, g)
            cur = g
        voices.append(vc)
        ends.append(cur)
    level = ends
    free_ports = []  # (leaf SumNode, first channel of the unconnected stereo port)
    first = True
    leaf_sums = []
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            spare = first and spare_port and len(grp) < 32
            m = e.sum(len(grp) + (1 if spare else 0))
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            if spare:
                free_ports.append((m, 2 * len(grp)))
            if first:
                leaf_sums.append(m)
            nxt.append(m)
        level = nxt
        first = False
        if len(level) == 1:
            break
    # master chain: (constructor, automatable parameter id or None, value range)
    kinds = [(lambda e: e.volume(float(rng.uniform(40, 130))), 0, (10.0, 130.0)),
             (lambda e: e.hard_clip(float(rng.uniform(-12, 0))), None, None),
             (lambda e: e.pan(float(rng.uniform(-1, 1))), 0, (-1.0, 1.0)),
             (lambda e: e.width(float(rng.uniform(0, 2))), 0, (0.0, 2.0)),
             (lambda e: e.biquad(0, float(rng.uniform(2000, 12000)), 0.707), 1, (500.0, 12000.0)),
             (lambda e: e.delay(int(rng.integers(20, 500)) / float(e.sample_rate), feedback=0.3, mix=0.3), 2, (0.0, 1.0))]
    chosen = [kinds[int(i)] for i in rng.integers(0, len(kinds), size=int(rng.integers(0, 4)))]
    top = level[0]
    cross = [fp for fp in free_ports if fp[0] != leaf_sums[0]]
    if cross and rng.random() < 0.3:
        # a bus into a voice mixer: the first leaf's bus, through a gain, into the spare LAST port of another leaf — that leaf is
        # voices on its leading ports and a bus behind them (the hybrid plan splits it: the voice-bank kernels sum the leading
        # ports, the SumNode continues on the levels), and the first leaf's bus is consumed twice
        port = cross[int(rng.integers(0, len(cross)))]
        free_ports.remove(port)
        xg = e.volume(float(rng.uniform(20, 90)))
        e.connect_stereo(leaf_sums[0], xg)
        e.connect_stereo(xg, port[0], port[1])
    if rng.random() < 0.35:
        # a send: one leaf bus is ALSO tapped into a return (a gain, sometimes a delay behind it) that joins the root in a
        # two-port sum — a bus consumed twice is no fused shape; the voice banks inside still are (hybrid plan: k_leaf_sum for
        # dry banks, k_chain when some bank holds a biquad / delay)
        tap = leaf_sums[int(rng.integers(0, len(leaf_sums)))]
        ret = e.volume(float(rng.uniform(20, 90)))
        e.connect_stereo(tap, ret)
        if rng.random() < 0.5:
            dly = e.delay(int(rng.integers(20, 300)) / float(e.sample_rate), feedback=0.25, mix=1.0)
            e.connect_stereo(ret, dly)
            ret = dly
        mix = e.sum(2)
        e.connect_stereo(top, mix, 0)
        e.connect_stereo(ret, mix, 2)
        top = mix
    m_nodes = scenarios.connect_through_master(e, top, [c[0] for c in chosen])
    e.update()
    fmts = [PLANAR_F32] if f32_only else [PLANAR_F32, PLANAR_I16, PLANAR_U16, INTERLEAVED_F32, INTERLEAVED_I16, INTERLEAVED_U16]
    for v, vc in enumerate(voices):
        ch = 1 if rng.random() < 0.2 else 2
        frames = int(rng.integers(mbf + 8, 6 * mbf))  # loops are never shorter than a block (Q8)
        vc["frames"] = frames
        data = scenarios.voice_source(seed * 1000 + v, frames, ch)
        fmt = int(rng.choice(fmts))
        e.sampler_set_sample(vc["sampler"], e.new_sample(fmt, ch, encode(data, fmt)))
        if rng.random() < 0.7:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        if rng.random() < 0.85:
            e.sampler_play(vc["sampler"])
    outs = []
    plugged = []  # voices added after the first compile: (sampler, last node, (sum, port))
    for rnd in range(int(rng.integers(3, 7))):
        k = int(rng.choice([1, 2, 3, 5, 9, 24, 70]))
        if rnd > 0 and rng.random() < 0.4:
            # a graph edit between calls (graph/graph.rs:201-231, :268-299, :396-477 + recompile): plug a new dry voice
            # into a spare port, or pull one out again — node state of everything else carries over
            if plugged and rng.random() < 0.5:
                smp, last, port = plugged.pop()
                e.remove_node(last)
                if last != smp:
                    e.remove_node(smp)
                free_ports.append(port)
                e.update()
            elif free_ports:
                port = free_ports.pop()
                smp = e.sampler(float(rng.uniform(40, 100)))
                last = smp
                if rng.random() < 0.7:
                    last = e.volume(float(rng.uniform(20, 100)))
                    e.connect_stereo(smp, last)
                e.connect_stereo(last, port[0], port[1])
                e.update()
                frames = int(rng.integers(mbf + 8, 4 * mbf))
                e.sampler_set_sample(smp, e.new_sample(PLANAR_F32, 2, scenarios.voice_source(seed * 1000 + 900 + rnd, frames, 2)))
                e.sampler_set_loop_range(smp, LOOP_FULL)
                e.sampler_play(smp)
                plugged.append((smp, last, port))
        if rnd > 0:
            for vc in voices:
                if rng.random() > 0.35:
                    continue
                at = int(rng.integers(0, k + (3 if rng.random() < 0.1 else 0)))  # sometimes a block of a LATER call
                what = int(rng.integers(0, 10))
                sr = float(e.sample_rate)
                if what == 0:
                    e.sampler_play(vc["sampler"], at_block=at)
                elif what == 1:
                    e.sampler_pause(vc["sampler"], at_block=at)
                elif what == 2:
                    e.sampler_stop(vc["sampler"], at_block=at)
                elif what == 3 and vc["gains"]:
                    e.set_param(vc["gains"][int(rng.integers(0, len(vc["gains"])))], 0, float(rng.choice([0.0, 25.0, 90.0, 140.0])),
                                at_block=at)
                elif what == 4 and vc["pans"]:
                    e.set_param(vc["pans"][int(rng.integers(0, len(vc["pans"])))], 0, float(rng.uniform(-1, 1)), at_block=at)
                elif what == 5:
                    e.set_param(vc["sampler"], 0, float(rng.choice([0.0, 50.0, 100.0])), at_block=at)
                elif what == 6:
                    e.sampler_set_playhead_secs(vc["sampler"], float(rng.integers(0, vc["frames"] - 1)) / sr, at_block=at)
                elif what == 7:
                    mode = int(rng.choice([LOOP_NONE, LOOP_FULL, LOOP_RANGE_SECS]))
                    vc["ranged"] = vc.get("ranged", False) or mode == LOOP_RANGE_SECS
                    lo = int(rng.integers(0, vc["frames"] - mbf - 4))
                    hi = int(rng.integers(lo + mbf + 2, vc["frames"]))  # range >= a block, inside the sample (Q8)
                    e.sampler_set_loop_range(vc["sampler"], mode, lo / sr, hi / sr, at_block=at)
                elif what == 8 and vc["dl"] is not None:
                    e.set_param(vc["dl"], int(rng.integers(1, 3)), float(rng.uniform(0, 0.9)), at_block=at)
                elif what == 9 and not vc.get("ranged", False):
                    # swap the sample under the voice (SetSample, sampler.rs:67-79), sometimes stopping it.  Not for a
                    # voice that ever had a RangeSecs loop, and later ranges / playheads stay inside BOTH samples:
                    # messages of one call apply at different blocks, and a range outside the sample panics upstream (Q8)
                    ch = 1 if rng.random() < 0.3 else 2
                    new_frames = int(rng.integers(mbf + 8, 5 * mbf))
                    vc["frames"], new_frames = min(vc["frames"], new_frames), new_frames
                    fmt = int(rng.choice(fmts))
                    data = scenarios.voice_source(seed * 1000 + 500 + int(rng.integers(0, 400)), new_frames, ch)
                    e.sampler_set_sample(vc["sampler"], e.new_sample(fmt, ch, encode(data, fmt)), bool(rng.random() < 0.3),
                                         at_block=at)
            for m, (_, pid, rng_v) in zip(m_nodes, chosen):
                if pid is not None and rng.random() < 0.25:
                    e.set_param(m, pid, float(rng.uniform(*rng_v)), at_block=int(rng.integers(0, k)))
        outs.append(np.asarray(e.process_blocks(k)))
    return np.concatenate(outs)


class AsyncEngine(GpuEngine):
    async_device = True


def test_spatialiser_history_crosses_calls_that_are_not_waited_for():
    """the scenario of test_gpu_parity's spatial bank through fwgpu_process_blocks_device with nothing between the calls: the next
    call's control kernel must not read the 64-frame histories before the render kernel of this call has left them behind"""
    for mbf, batch in ((128, 64), (64, 16)):
        g = AsyncEngine(max_block_frames=mbf, max_batch=batch)
        o = oracle(max_block_frames=mbf)
        assert_bits_equal(scenarios.scenario_spatial_bank(o), scenarios.scenario_spatial_bank(g), "spatial bank, async calls, block %d" % mbf)
        assert g.cx.plan_kind() == 1


# (91, 196, 384: spatialiser voices across calls with the control kernel running a call ahead — the history hand-over raced; found
#  by a 500-seed run in round 3, in the suite since)
@pytest.mark.parametrize("seed", sorted(set(range(int(os.environ.get("FWGPU_FUZZ_SEEDS", "80")))) | {91, 196, 384}))
def test_random_graph_and_messages_every_plan_bit_exact(seed):
    pick = np.random.default_rng(10_000 + seed)
    mbf = int(pick.choice([64, 128, 256]))
    o = oracle(max_block_frames=mbf)
    scheds = []  # the CompiledSchedule of every update() of the run (graph edits recompile)

    def recording_update():
        o.e.update()
        scheds.append((o.e.schedule(), o.e.num_buffers()))

    o.update = recording_update
    want = fuzz_run(o, seed)
    assert np.all(np.isfinite(want))
    cls = AsyncEngine if pick.random() < 0.5 else GpuEngine
    g = cls(max_block_frames=mbf, max_batch=int(pick.choice([1, 2, 5, 64])))
    assert_bits_equal(want, fuzz_run(g, seed), "seed %d plan %d %s" % (seed, g.cx.plan_kind(), cls.__name__))
    g2 = GpuEngine(max_block_frames=mbf, force_generic=True, max_batch=int(pick.choice([1, 3, 64])))
    assert_bits_equal(want, fuzz_run(g2, seed), "seed %d generic" % seed)
    # (FWGPU_LAZY_ADOPT, the test mode that delays every adoption to the next process call, also delays the reuse of a removed node's
    #  slot by one adoption — by design, fwgpu_ctx.h `limbo` — so node ids stop matching the oracle's allocator after a removal)
    if pick.random() < 0.3 and not os.environ.get("FWGPU_LAZY_ADOPT"):
        # level A of INTEGRATION.md: every schedule of the run comes from the reference's compiler (restated by the oracle)
        # through fwgpu_schedule_upload — node ids are the same on both sides, also after removals and slot reuse; the
        # fused plans must be recognised on the imported schedules too
        g3 = GpuEngine(max_block_frames=mbf, max_batch=int(pick.choice([2, 64])))
        todo = list(scheds)
        g3.update = lambda: g3.cx.schedule_upload(*todo.pop(0))
        assert_bits_equal(want, fuzz_run(g3, seed), "seed %d imported schedules" % seed)
        assert not todo and g3.cx.plan_kind() == g.cx.plan_kind()


def fuzz_dag(e, seed):
    """a random DAG over every node kind the generic executor has (no beep: libm tolerance): sampler / resampler sources,
    stereo 2->2 processors, mono detours (stereo->mono -> 1->1 volume -> spatialiser or mono->stereo), 2/3/4/n-port sums,
    small FIR convolutions, one-to-many edges, dangling nodes and unconnected ports; then message traffic."""
    rng = np.random.default_rng(50_000 + seed)
    mbf = e.max_block_frames
    sr = float(e.sample_rate)
    sigs = []      # stereo signals: (node, first output port)
    ctl = []       # (node, [(param id, lo, hi)]) for automation
    samplers = []
    for v in range(int(rng.integers(1, 7))):
        frames = int(rng.integers(mbf + 8, 5 * mbf))
        ch = 1 if rng.random() < 0.3 else 2
        fmt = int(rng.choice([PLANAR_F32, PLANAR_I16, INTERLEAVED_F32, INTERLEAVED_U16]))
        smp = e.new_sample(fmt, ch, encode(scenarios.voice_source(seed * 977 + v, frames, ch), fmt))
        if rng.random() < 0.6:
            s = e.sampler(float(rng.uniform(40, 100)))
            samplers.append((s, smp, frames))
            ctl.append((s, [(0, 0.0, 110.0)]))
            sigs.append((s, 0))
        else:
            rs = e.resampler(smp, float(rng.choice([1.0, 0.5, 1.37, 2.0, 44100.0 / 48000.0])), loop=bool(rng.random() < 0.6), n_out=2)
            ctl.append((rs, [(1, 0.3, 2.5), (3, 0.0, 1.0), (4, 0.0, float(frames - 1))]))
            sigs.append((rs, 0))
    ir = None
    for step in range(int(rng.integers(2, 14))):
        kind = int(rng.integers(0, 11))
        src = sigs[int(rng.integers(0, len(sigs)))]
        if kind == 0:
            n = e.volume(float(rng.uniform(10, 130)))
            ctl.append((n, [(0, 0.0, 130.0)]))
        elif kind == 1:
            n = e.pan(float(rng.uniform(-1, 1)))
            ctl.append((n, [(0, -1.0, 1.0)]))
        elif kind == 2:
            n = e.width(float(rng.uniform(0, 2)))
            ctl.append((n, [(0, 0.0, 2.0)]))
        elif kind == 3:
            n = e.hard_clip(float(rng.uniform(-18, 0)))
        elif kind == 4:
            n = e.biquad(int(rng.integers(0, 3)), float(rng.uniform(100, 10000)), float(rng.uniform(0.5, 4)))
            ctl.append((n, [(1, 100.0, 10000.0), (2, 0.5, 4.0)]))
        elif kind == 5:
            n = e.delay(int(rng.integers(1, 900)) / sr, feedback=float(rng.uniform(0, 0.7)), mix=float(rng.uniform(0, 1)))
            ctl.append((n, [(1, 0.0, 0.8), (2, 0.0, 1.0)]))
        elif kind == 6:  # mono detour: stereo -> mono -> gain -> spatialiser (or mono -> stereo)
            m = e.add_node(fwapi.STEREO_TO_MONO, 2, 1)
            e.connect_stereo(src[0], m, 0, src[1])
            g = e.volume(float(rng.uniform(30, 100)), ch=1)
            e.connect(m, 0, g, 0)
            ctl.append((g, [(0, 0.0, 120.0)]))
            if rng.random() < 0.5:
                n = e.spatial(float(rng.uniform(-4, 4)), float(rng.uniform(-1, 1)), float(rng.uniform(-4, 4)), n_in=1)
                ctl.append((n, [(0, -4.0, 4.0), (1, -1.0, 1.0), (2, -4.0, 4.0)]))
            else:
                n = e.add_node(fwapi.MONO_TO_STEREO, 1, 2)
            e.connect(g, 0, n, 0)
            sigs.append((n, 0))
            continue
        elif kind == 7:  # sum of 2..6 signals (2/3/4-port paths and the n-port path), sometimes with an unconnected port
            k = int(rng.integers(2, 7))
            n = e.sum(k)
            for p in range(k):
                if rng.random() < 0.85:
                    a = sigs[int(rng.integers(0, len(sigs)))]
                    e.connect_stereo(a[0], n, 2 * p, a[1])
            sigs.append((n, 0))
            continue
        elif kind == 8:
            if ir is None:
                ir = e.new_sample(PLANAR_F32, 2, scenarios.reverb_ir(seed, int(rng.integers(30, 400)), 2))
            n = e.fir(ir)
        elif kind == 9:  # stereo spatialiser
            n = e.spatial(float(rng.uniform(-4, 4)), float(rng.uniform(-1, 1)), float(rng.uniform(-4, 4)), n_in=2)
            ctl.append((n, [(0, -4.0, 4.0), (2, -4.0, 4.0)]))
        else:
            continue  # (a step that adds nothing: graphs of every size)
        e.connect_stereo(src[0], n, 0, src[1])
        sigs.append((n, 0))
    # master: the last signal, or a sum of a few, to graph_out
    if rng.random() < 0.5 and len(sigs) > 1:
        k = int(rng.integers(2, 5))
        m = e.sum(k)
        for p in range(k):
            a = sigs[-1 - p] if p < len(sigs) else sigs[0]
            e.connect_stereo(a[0], m, 2 * p, a[1])
        e.connect_stereo(m, e.graph_out_node)
    else:
        e.connect_stereo(sigs[-1][0], e.graph_out_node, 0, sigs[-1][1])
    e.update()
    for s, smp, frames in samplers:
        e.sampler_set_sample(s, smp)
        if rng.random() < 0.7:
            e.sampler_set_loop_range(s, LOOP_FULL)
        if rng.random() < 0.9:
            e.sampler_play(s)
    outs = []
    for rnd in range(int(rng.integers(2, 6))):
        k = int(rng.choice([1, 2, 4, 7, 19]))
        if rnd > 0:
            for node, params in ctl:
                if rng.random() < 0.3:
                    pid, lo, hi = params[int(rng.integers(0, len(params)))]
                    e.set_param(node, pid, float(rng.uniform(lo, hi)), at_block=int(rng.integers(0, k)))
            for s, smp, frames in samplers:
                r = rng.random()
                at = int(rng.integers(0, k))
                if r < 0.1:
                    e.sampler_pause(s, at_block=at)
                elif r < 0.25:
                    e.sampler_play(s, at_block=at)
                elif r < 0.3:
                    e.sampler_stop(s, at_block=at)
                elif r < 0.4:
                    e.sampler_set_playhead_secs(s, float(rng.integers(0, frames - 1)) / sr, at_block=at)
        outs.append(np.asarray(e.process_blocks(k)))
    return np.concatenate(outs)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FWGPU_FUZZ_SEEDS", "60"))))
def test_random_dag_generic_executor_bit_exact(seed):
    pick = np.random.default_rng(70_000 + seed)
    mbf = int(pick.choice([32, 64, 100, 128, 256]))
    o = oracle(max_block_frames=mbf)
    want = fuzz_dag(o, seed)
    assert np.all(np.isfinite(want))
    cls = AsyncEngine if pick.random() < 0.5 else GpuEngine
    g = cls(max_block_frames=mbf, max_batch=int(pick.choice([1, 2, 5, 64])))
    imported = pick.random() < 0.3
    if imported:
        # level A of INTEGRATION.md: keep the reference's own compiler (restated by the oracle: its order, its LIFO buffer
        # assignment) and hand the CompiledSchedule to fwgpu_schedule_upload; node ids are the same on both sides
        sched, nbuf = o.e.schedule(), o.e.num_buffers()
        g.update = lambda: g.cx.schedule_upload(sched, nbuf)
    assert_bits_equal(want, fuzz_dag(g, seed), "dag seed %d plan %d %s%s" % (seed, g.cx.plan_kind(), cls.__name__,
                                                                             " imported schedule" if imported else ""))


def fuzz_stream(e, seed, n_in):
    """an effects rack on the stream inputs (graph_in -> random processors / mixes -> graph_out), driven with calls of
    ARBITRARY length — whole blocks plus a partial one (processor.rs:95-96), also calls shorter than a block — and
    parameter changes between calls.  No sampler (the reference's sampler asserts frames == max_block_frames: Q5)."""
    rng = np.random.default_rng(90_000 + seed)
    mbf = e.max_block_frames
    sr = float(e.sample_rate)
    gin = e.graph_in_node
    sigs = [(gin, 2 * p) for p in range(n_in // 2)]
    if n_in % 2:  # an odd last input goes through mono -> stereo
        m = e.add_node(fwapi.MONO_TO_STEREO, 1, 2)
        e.connect(gin, n_in - 1, m, 0)
        sigs.append((m, 0))
    ctl = []
    for step in range(int(rng.integers(1, 10))):
        kind = int(rng.integers(0, 8))
        src = sigs[int(rng.integers(0, len(sigs)))]
        if kind == 0:
            n = e.volume(float(rng.uniform(10, 130)))
            ctl.append((n, [(0, 0.0, 130.0)]))
        elif kind == 1:
            n = e.pan(float(rng.uniform(-1, 1)))
            ctl.append((n, [(0, -1.0, 1.0)]))
        elif kind == 2:
            n = e.width(float(rng.uniform(0, 2)))
            ctl.append((n, [(0, 0.0, 2.0)]))
        elif kind == 3:
            n = e.hard_clip(float(rng.uniform(-18, 0)))
        elif kind == 4:
            n = e.biquad(int(rng.integers(0, 3)), float(rng.uniform(100, 10000)), float(rng.uniform(0.5, 4)))
            ctl.append((n, [(1, 100.0, 10000.0), (2, 0.5, 4.0)]))
        elif kind == 5:
            n = e.delay(int(rng.integers(1, 700)) / sr, feedback=float(rng.uniform(0, 0.7)), mix=float(rng.uniform(0, 1)))
            ctl.append((n, [(1, 0.0, 0.8), (2, 0.0, 1.0)]))
        elif kind == 6:
            k = int(rng.integers(2, 5))
            n = e.sum(k)
            for p in range(k):
                a = sigs[int(rng.integers(0, len(sigs)))]
                e.connect_stereo(a[0], n, 2 * p, a[1])
            sigs.append((n, 0))
            continue
        else:
            n = e.spatial(float(rng.uniform(-4, 4)), float(rng.uniform(-1, 1)), float(rng.uniform(-4, 4)), n_in=2)
            ctl.append((n, [(0, -4.0, 4.0), (2, -4.0, 4.0)]))
        e.connect_stereo(src[0], n, 0, src[1])
        sigs.append((n, 0))
    e.connect_stereo(sigs[-1][0], e.graph_out_node, 0, sigs[-1][1])
    e.update()
    outs = []
    for rnd in range(int(rng.integers(3, 9))):
        frames = int(rng.choice([1, 3, mbf - 1, mbf, mbf + 1, 2 * mbf + 5, 3 * mbf, int(rng.integers(1, 5 * mbf))]))
        if rnd > 0:
            for node, params in ctl:
                if rng.random() < 0.35:
                    pid, lo, hi = params[int(rng.integers(0, len(params)))]
                    e.set_param(node, pid, float(rng.uniform(lo, hi)))
        if rng.random() < 0.15:
            inp = np.zeros(frames * n_in, dtype=np.float32)  # digital silence on the inputs
        else:
            inp = fwapi.xorshift_uniform(seed * 31 + rnd, frames * n_in)
        outs.append(np.asarray(e.process_interleaved(frames, 2, inp=inp, n_in_ch=n_in)))
    return np.concatenate(outs)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FWGPU_FUZZ_SEEDS", "60"))))
def test_random_effects_rack_on_stream_inputs_any_call_length(seed):
    pick = np.random.default_rng(95_000 + seed)
    mbf = int(pick.choice([16, 64, 100, 256]))
    n_in = int(pick.choice([1, 2, 3, 4]))
    want = fuzz_stream(oracle(max_block_frames=mbf, num_graph_inputs=n_in), seed, n_in)
    assert np.all(np.isfinite(want))
    g = GpuEngine(max_block_frames=mbf, num_graph_inputs=n_in, max_batch=int(pick.choice([1, 3, 64])))
    assert_bits_equal(want, fuzz_stream(g, seed, n_in), "stream seed %d" % seed)
