"""Round 6 — the chain plan's grammar.

Reference: any edge is legal (crates/firewheel-graph/src/graph.rs:396-477), so a voice may put its gain in FRONT of its filter,
cascade two biquads (an EQ), or run its delay line into a filter.  The chain plan (k_chain) took `sampler -> [biquad] -> [delay] ->
gains` only; every other order fell to the level executor at a quarter of the speed.  It now takes

    sampler -> G* -> F1 [-> G* -> F2 [-> G* -> F3]] -> G*     G = volume | pan | hard clip (<= 3),  F1 F2 F3 = B | BB | D | BD | BBD | DB | DBB

CPU tier: the planner accepts exactly that (host-only harness: the real host code on a fake HIP runtime whose launch stubs validate
every table the kernels would read).  GPU tier: every accepted shape bit for bit against the oracle — steady calls, source pauses
(the stages in front of the filters see the sampler's silence flag and reset, the ones behind never do), muted pre-gains, gain
glides on both sides of the filters, coefficient / feedback messages for every filter of the chain, every K batching.
"""
import os

import numpy as np
import pytest

import fwapi
import scenarios
from fwapi import LOOP_FULL, GpuEngine, HostOnlyEngine, OracleEngine

# token: v volume, p pan, B biquad, D delay.  One string = one voice chain behind its sampler.
# a leading m: a ONE-output sampler behind the reference's MonoToStereoNode (mono_to_stereo.rs:33-50) — round 6: also in front of filters
# (c = hard clip, w = stereo width)
ACCEPTED = ["vB", "pBv", "vBD", "vBDv", "vpBDp", "BB", "BBv", "vBB", "vBBDp", "BBD", "DB", "DBv", "vDB", "DBB", "pDBBv", "vD", "vDp", "mBD", "mvBBp", "mDBv",
            "BvB", "BvD", "BpBvD", "BBvD", "vBvDv", "DvB", "DpBvB", "cB", "BcD", "BDc", "vBcBD", "DcBv", "mBvBc"]  # gains between filters, clips anywhere
REFUSED = ["BDB", "DBD", "BBB", "DD", "wB", "Bw", "BwD", "vBvDvp"]  # filter orders k_chain has no pipeline for, a width beside a filter, four gain stages


def build_voice(e, shape, rng, delay_frames):
    mono = shape.startswith("m")
    s = e.sampler(100.0, n_out=1) if mono else e.sampler(100.0)
    cur = s
    if mono:
        cur = e.add_node(fwapi.MONO_TO_STEREO, 1, 2)
        e.connect(s, 0, cur, 0)
        shape = shape[1:]
    nodes = dict(sampler=s, vols=[], pans=[], bqs=[], dls=[])
    for t in shape:
        if t == "v":
            n = e.volume(float(rng.uniform(30, 100)))
            nodes["vols"].append(n)
        elif t == "p":
            n = e.pan(float(rng.uniform(-1, 1)))
            nodes["pans"].append(n)
        elif t == "B":
            n = e.biquad(int(rng.integers(0, 3)), float(rng.uniform(200, 8000)), float(rng.choice([0.707, 1.8])))
            nodes["bqs"].append(n)
        elif t == "D":
            n = e.delay(delay_frames / float(e.sample_rate), feedback=float(rng.choice([0.0, 0.45])), mix=0.5)
            nodes["dls"].append(n)
        elif t == "w":
            n = e.width(1.3)
        elif t == "c":
            n = e.hard_clip(-3.0)
        else:
            raise ValueError(t)
        e.connect_stereo(cur, n)
        cur = n
    nodes["end"] = cur
    return nodes


def build_bank(e, shapes, radix=32, src_frames=1500, seed=0, delays=(64, 129, 300, 384, 700, 1000), fmt=fwapi.PLANAR_F32):
    rng = np.random.default_rng(900 + seed)
    voices = [build_voice(e, sh, rng, delays[i % len(delays)]) for i, sh in enumerate(shapes)]
    level = [v["end"] for v in voices]
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.sum(max(2, len(grp)))
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    e.connect_stereo(level[0], e.graph_out_node)
    e.update()
    for i, vc in enumerate(voices):
        data = scenarios.voice_source(seed * 1000 + 77 + i, src_frames, 2)
        if fmt == fwapi.INTERLEAVED_I16:
            data = np.round(data * 32767).astype(np.int16).T.copy()
        elif fmt == fwapi.INTERLEAVED_U16:
            data = np.round((data + 1.0) * 32767.5).astype(np.uint16).T.copy()
        vc["sample"] = e.new_sample(fmt, 2, data)
        e.sampler_set_sample(vc["sampler"], vc["sample"])
    return voices


def run_events(e, shapes, calls=(3, 5, 2, 4, 6, 3), **kw):
    """steady calls, then everything that can happen to such a voice, every kind of message landing inside a call"""
    voices = build_bank(e, shapes, **kw)
    for vc in voices:
        e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        e.sampler_play(vc["sampler"])
    out = [e.process_blocks(calls[0])]           # start-up: smoothers settle
    out.append(e.process_blocks(calls[1]))        # steady
    for i, vc in enumerate(voices):               # glides on both sides of the filters, filter messages, one block into the call
        for j, n in enumerate(vc["vols"]):
            if (i + j) % 2 == 0:
                e.set_param(n, 0, 20.0 + 7.0 * ((i + j) % 5), at_block=1 + (i % 2))
        for n in vc["pans"]:
            if i % 3 == 0:
                e.set_param(n, 0, -0.5, at_block=1)
        for j, n in enumerate(vc["bqs"]):
            if (i + j) % 2 == 1:
                e.set_param(n, 1, 500.0 + 300.0 * j + 100.0 * (i % 7), at_block=1)
        for n in vc["dls"]:
            if i % 4 == 1:
                e.set_param(n, 1, 0.3, at_block=0)
    out.append(e.process_blocks(calls[2]))
    out.append(e.process_blocks(calls[3]))        # the glides settle, then steady again
    for i, vc in enumerate(voices):               # source pauses (the stages in front reset), a muted pre-gain, a muted post-gain
        if i % 3 == 0:
            e.sampler_pause(vc["sampler"], at_block=1)
        if i % 3 == 1 and vc["vols"]:
            e.set_param(vc["vols"][0], 0, 0.0, at_block=0)
        if i % 3 == 2 and len(vc["vols"]) > 1:
            e.set_param(vc["vols"][-1], 0, 0.0, at_block=2)
    out.append(e.process_blocks(calls[4]))
    for i, vc in enumerate(voices):               # ... and back
        if i % 3 == 0:
            e.sampler_play(vc["sampler"], at_block=0)
        if i % 3 == 1 and vc["vols"]:
            e.set_param(vc["vols"][0], 0, 80.0, at_block=1)
    out.append(e.process_blocks(calls[5]))
    out.append(e.process_blocks(calls[1]))
    return np.concatenate(out)


# ------------------------------------------------------------------------------------------------ CPU tier: the planner
@pytest.mark.parametrize("shape", ACCEPTED)
def test_planner_takes_the_shape_on_the_chain_plan(shape):
    e = HostOnlyEngine(max_block_frames=128, max_batch=8)
    build_bank(e, [shape] * 5)
    assert e.cx.plan_kind() == 2, shape
    assert e.cx.plan_fused_voices() == 5
    for vc in range(3):
        e.process_blocks(3)
    assert e.violation() == ""
    la = e.launches()
    assert la["chain"] >= 3 and la["level"] == 0, la


@pytest.mark.parametrize("shape", REFUSED)
def test_planner_refuses_the_shape_as_a_whole_and_keeps_its_prefix(shape):
    """not lost: the hybrid plan renders the longest acceptable prefix of each voice as a solo voice, the level executor the rest"""
    e = HostOnlyEngine(max_block_frames=128, max_batch=8)
    build_bank(e, [shape] * 9)
    assert e.cx.plan_kind() in (0, 3), shape
    e.process_blocks(3)
    assert e.violation() == ""
    if e.cx.plan_kind() == 3:
        assert e.cx.plan_fused_voices() == 9


def test_planner_mixed_bank_of_every_accepted_shape_and_events_on_the_host_harness():
    e = HostOnlyEngine(max_block_frames=128, max_batch=4)
    run_events(e, ACCEPTED * 2, radix=8)
    assert e.cx.plan_kind() == 2
    assert e.violation() == ""


# ------------------------------------------------------------------------------------------------ GPU tier: parity
def both(shapes, mbf=128, max_batch=None, force_generic=False, **kw):
    o = scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf))  # (messages carry at_block, like the C ABI's)
    g = GpuEngine(max_block_frames=mbf, max_batch=max_batch, force_generic=force_generic)
    ro = run_events(o, shapes, **kw)
    rg = run_events(g, shapes, **kw)
    return ro, rg, g


def assert_bits(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    bad = np.nonzero(fwapi.bits(a) != fwapi.bits(b))[0]
    assert bad.size == 0, "%s: %d of %d samples differ, first at %d (block %d): %r vs %r" % (what, bad.size, a.size, bad[0], bad[0] // (2 * 128), a[bad[0]], b[bad[0]])
    assert np.any(a != 0), what + ": nothing sounded"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ACCEPTED)
@pytest.mark.parametrize("max_batch", [64, 2])
def test_every_accepted_shape_is_bit_exact_on_the_chain_plan(shape, max_batch):
    ro, rg, g = both([shape] * 7, max_batch=max_batch, radix=4)
    assert g.cx.plan_kind() == 2, shape
    assert_bits(ro, rg, "%s K<=%d" % (shape, max_batch))


@pytest.mark.gpu
@pytest.mark.parametrize("max_batch", [64, 3, 1])
@pytest.mark.parametrize("mbf", [128, 64])
def test_a_mixed_bank_of_every_accepted_shape_is_bit_exact(max_batch, mbf):
    """voices of different shapes share workgroups (<= 32 rows each): per-lane stage positions, one instantiation for the whole plan;
    mbf 64 = the small-tile instantiations"""
    shapes = (ACCEPTED * 3)[:45]
    ro, rg, g = both(shapes, mbf=mbf, max_batch=max_batch, radix=16, seed=3)
    assert g.cx.plan_kind() == 2
    assert_bits(ro, rg, "mixed bank K<=%d mbf %d" % (max_batch, mbf))
    steady, general = g.cx.plan_chain_stats()  # (every workgroup of this bank holds a delay shorter than three tiles: the general loop)
    assert general > 0, (steady, general)


@pytest.mark.gpu
@pytest.mark.parametrize("shapes", [["vBDv", "BBDp", "vBBD", "pBD"], ["DB", "vDBp", "DBB", "pDBBv"], ["vBDv", "DBv", "BBD", "vDBB"]])
def test_every_accepted_order_takes_the_steady_call_loop(shapes):
    """gain-before-filter, two-biquad and delay-first voices with delays >= 3 tiles run k_chain's branch-free steady-call loop on
    message-free calls (delay-first lanes request their ring slots two tiles AHEAD of S1 instead of behind it)"""
    shapes = shapes * 8
    ro, rg, g = both(shapes, max_batch=16, radix=32, delays=(384, 500, 777, 1000), calls=(4, 16, 2, 12, 6, 3))
    assert g.cx.plan_kind() == 2
    assert_bits(ro, rg, "steady-call loop shapes")
    steady, general = g.cx.plan_chain_stats()
    assert steady >= 2 and general > 0, (steady, general)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["vBDv", "BBv", "DBp", "vDBB"])
def test_the_level_executor_twin_of_the_new_shapes(shape):
    ro, rg, g = both([shape] * 6, force_generic=True, radix=3)
    assert g.cx.plan_kind() == 0
    assert_bits(ro, rg, shape + " on the level executor")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", REFUSED)
def test_refused_shapes_stay_bit_exact_on_the_hybrid_plan(shape):
    ro, rg, g = both([shape] * 9, radix=3)
    assert g.cx.plan_kind() in (0, 3)
    assert_bits(ro, rg, shape + " (refused as a whole)")


@pytest.mark.gpu
def test_state_of_both_biquads_and_the_pre_gain_survives_plan_switches():
    """chain plan -> level executor -> chain plan mid-stream: both filters' histories, the ring, the smoothers in front carry over"""
    def run(e):
        voices = build_bank(e, ["vBBDp", "pDBBv", "vBD"] * 3, radix=3)
        for vc in voices:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
            e.sampler_play(vc["sampler"])
        e.set_param(voices[0]["vols"][0], 0, 15.0, at_block=2)
        a = e.process_blocks(4)
        extra = e.sum(2)  # a dangling node: no fused plan covers the graph any more
        e.update()
        b = e.process_blocks(3)
        e.remove_node(extra)
        e.update()
        c = e.process_blocks(5)
        return np.concatenate([a, b, c])

    o, g = scenarios.TaggedOracle(OracleEngine(max_block_frames=128)), GpuEngine(max_block_frames=128)
    ro, rg = run(o), run(g)
    assert g.cx.plan_kind() == 2
    assert_bits(ro, rg, "plan switches")


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["INTERLEAVED_I16", "INTERLEAVED_U16"])
@pytest.mark.parametrize("max_batch", [16, 1])
def test_interleaved_16_bit_sources_are_fetched_by_the_chain_kernel_itself(fmt, max_batch):
    """16-bit PCM (core/sample_resource.rs:338-345) as a compact source class of the chain plan: one dwordx4 per quad like planar f32,
    the channel's half-word converted in S1 — both loops (the sample length is no multiple of the block: loops wrap inside blocks),
    message-free calls on the steady-call loop"""
    shapes = ["vBDv", "BD", "BBDp", "vB", "D"] * 7 + ["vDB", "DBB"]
    ro, rg, g = both(shapes, max_batch=max_batch, radix=32, delays=(384, 500, 777, 1000), calls=(4, 16, 2, 12, 6, 3),
                     fmt=getattr(fwapi, fmt), src_frames=1501)
    assert g.cx.plan_kind() == 2
    assert_bits(ro, rg, "16-bit sources K<=%d" % max_batch)
    steady, general = g.cx.plan_chain_stats()
    assert general > 0 and (steady > 0 or max_batch == 1), (steady, general)


# ------------------------------------------------------------------------------------------------ a seeded fuzz family for the grammar
DRY = ["v", "vp", "", "pv"]
FUZZ_SEEDS = int(__import__("os").environ.get("FWGPU_FUZZ_SEEDS", "40"))


def fuzz_grammar(e, seed, only=None, log=None):
    """(only = i: debugging — every voice but i stays stopped, same random draws; log: list that receives (call, voice, what, at_block))
    a bank of voices of random shapes — accepted, refused and dry ones side by side —, random source formats and lengths (loops wrap
    inside blocks), random delays from one tile up, then calls of random length with every kind of message at random blocks"""
    rng = np.random.default_rng(10_000 + seed)
    pool = ACCEPTED * 3 + DRY + (REFUSED if seed % 3 == 0 else [])
    n = int(rng.integers(1, 60))
    shapes = [pool[int(rng.integers(0, len(pool)))] for _ in range(n)]
    radix = int(rng.choice([2, 5, 16, 32]))
    fmts = [fwapi.PLANAR_F32] * 3 + [fwapi.INTERLEAVED_I16, fwapi.INTERLEAVED_U16, fwapi.PLANAR_I16, fwapi.INTERLEAVED_F32]
    delays = tuple(int(x) for x in rng.integers(64, 1300, size=7))
    mbf = e.max_block_frames
    voices = [build_voice(e, sh, np.random.default_rng(seed * 977 + i), delays[i % 7]) for i, sh in enumerate(shapes)]
    level = [v["end"] for v in voices]
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.sum(max(2, len(grp)))
            for p, nd in enumerate(grp):
                e.connect_stereo(nd, m, 2 * p)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    e.connect_stereo(level[0], e.graph_out_node)
    e.update()
    for i, vc in enumerate(voices):
        fmt = fmts[int(rng.integers(0, len(fmts)))]
        ch = 1 if rng.random() < 0.15 else 2
        if seed % 4 == 3:  # (every fourth seed: only what k_chain fetches compactly — planar f32, interleaved stereo 16-bit — so that the whole
            fmt = fmts[int(rng.integers(0, 5))]  # bank can leave lazy records)
            ch = 2 if fmt != fwapi.PLANAR_F32 else ch
        frames = int(rng.integers(mbf + 40, 6 * mbf))
        if seed % 2 == 1:
            frames = mbf * int(rng.integers(2, 7))  # odd seeds: loops a whole number of blocks long — the quiet calls at the end can go lazy
        data = scenarios.voice_source(seed * 5000 + i, frames, ch)
        if fmt in (fwapi.INTERLEAVED_I16, fwapi.PLANAR_I16):
            data = np.round(data * 32767).astype(np.int16)
        elif fmt == fwapi.INTERLEAVED_U16:
            data = np.round((data + 1.0) * 32767.5).astype(np.uint16)
        if fmt in (fwapi.INTERLEAVED_I16, fwapi.INTERLEAVED_U16, fwapi.INTERLEAVED_F32):
            data = data.T.copy()
        e.sampler_set_sample(vc["sampler"], e.new_sample(fmt, ch, data))
        if rng.random() < 0.85:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        if rng.random() < 0.9 and (only is None or only == i):
            e.sampler_play(vc["sampler"])
    if log is not None:
        log.append(("shapes", shapes, "delays", delays, "radix", radix))
    out = []
    for call in range(int(rng.integers(3, 6))):
        k = int(rng.integers(1, 9))
        # this call's messages, sent in block order (include/fwgpu.h: a node's messages go out in non-decreasing at_block order — a biquad's
        # cutoff / Q are folded into coefficients when SENT; seeds 26 and 208 of the first version of this fuzz sent them out of order)
        msgs = sorted(((int(rng.integers(0, n)), int(rng.integers(0, k)), int(rng.integers(0, 7)), rng.random(4)) for _ in range(int(rng.integers(0, 1 + n // 2)))),
                      key=lambda m: m[1])
        for vi, at, what, u in msgs:
            vc = voices[vi]
            if log is not None:
                log.append((call, k, vi, what, at))
            if what == 0 and vc["vols"]:
                e.set_param(vc["vols"][int(u[0] * len(vc["vols"]))], 0, [0.0, 25.0, 60.0, 110.0][int(u[1] * 4)], at_block=at)
            elif what == 1 and vc["pans"]:
                e.set_param(vc["pans"][int(u[0] * len(vc["pans"]))], 0, float(2.0 * u[1] - 1.0), at_block=at)
            elif what == 2 and vc["bqs"]:
                cut = u[2] < 0.5
                e.set_param(vc["bqs"][int(u[0] * len(vc["bqs"]))], 1 if cut else 2, float(150.0 + 8850.0 * u[1] if cut else 0.6 + 2.4 * u[1]), at_block=at)
            elif what == 3 and vc["dls"]:
                e.set_param(vc["dls"][0], 1 + int(u[0] * 2), float(0.7 * u[1]), at_block=at)
            elif what == 4:
                e.sampler_pause(vc["sampler"], at_block=at)
            elif what == 5:
                if only is None or only == vi:
                    e.sampler_play(vc["sampler"], at_block=at)
            else:
                e.sampler_stop(vc["sampler"], at_block=at)
        out.append(e.process_blocks(k))
    # quiet calls: when every voice has settled (and loops in whole blocks) these run without the control kernel (lazy records); one more
    # message in the middle of them: filter coefficients / delay parameters changed INSIDE a control call, read back from node state
    for k in (int(rng.integers(2, 9)), int(rng.integers(1, 9)), int(rng.integers(2, 20))) + ((24, 24, 24, 7) if seed % 4 == 3 else ()):
        out.append(e.process_blocks(k))  # (every fourth seed: long enough for a glide to 0.0 to settle)
    vi = int(rng.integers(0, n))
    if voices[vi]["bqs"]:
        e.set_param(voices[vi]["bqs"][0], 1, float(rng.uniform(200.0, 6000.0)), at_block=1)
    if voices[vi]["dls"]:
        e.set_param(voices[vi]["dls"][0], 2, float(rng.uniform(0.1, 0.6)), at_block=0)
    for k in (3, int(rng.integers(2, 6)), int(rng.integers(2, 12)), int(rng.integers(1, 9))):
        out.append(e.process_blocks(k))
    return np.concatenate(out)


# ------------------------------------------------------------------------------------------------ lazy calls (no control kernel)
def run_quiet(e, shapes, mbf, fmt, seed=0):
    """Loops a whole number of blocks long (the lazy records' condition), long message-free stretches, and in between everything a
    later lazy call must not miss: filter coefficients and delay parameters changed INSIDE a control call (the ChainStart record holds the
    values at that call's start, node state the ones at its end: a lazy call reads the state), one-shots running out (the horizon),
    pauses, a muted stage between two filters, a one-block call, a call longer than a batch."""
    voices = build_bank(e, shapes, radix=8, src_frames=mbf * 6, fmt=fmt, seed=seed, delays=(64, 129, 300, 384, 700, 1000, 2048))
    for i, vc in enumerate(voices):
        if i % 7 != 3:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)   # i % 7 == 3: one-shots, 6 blocks long
        if i % 10 != 9:
            e.sampler_play(vc["sampler"])                        # i % 10 == 9: never started
    outs, marks = [], []
    for c, k in enumerate([3, 2, 2, 3, 4, 1, 4, 4, 9, 2, 2, 3, 3, 5, 2, 2, 2, 6] + [8] * 12 + [4, 4]):
        if c == 8:
            for i, vc in enumerate(voices):
                for j, n in enumerate(vc["bqs"]):
                    if (i + j) % 2 == 0:
                        e.set_param(n, 1, 400.0 + 250.0 * j + 90.0 * (i % 9), at_block=1 + (i % 3))
                for n in vc["dls"]:
                    if i % 2 == 1:
                        e.set_param(n, 1 + (i % 4) // 2, 0.15 + 0.05 * (i % 5), at_block=2)
        if c == 12:
            for i, vc in enumerate(voices):
                if i % 5 == 0:
                    e.sampler_pause(vc["sampler"], at_block=1)
                if i % 5 == 1 and len(vc["vols"]) > 1:
                    e.set_param(vc["vols"][1], 0, 0.0, at_block=0)   # (between two filters in the shapes that have a mid stage)
        if c == 16:
            for i, vc in enumerate(voices):
                if i % 5 == 0:
                    e.sampler_play(vc["sampler"], at_block=0)
        outs.append(np.asarray(e.process_blocks(k)))
        if hasattr(e, "cx"):
            marks.append(e.cx.lazy_stats())
    return np.concatenate(outs), marks




def test_chain_plan_lazy_calls_on_the_host_harness():
    """the host's side of it on the fake device (whose control launches always report an unbounded horizon): the launch stubs check the
    block offset every lazy k_chain launch names and that the flush names the voice table"""
    e = HostOnlyEngine(max_block_frames=128, max_batch=8)
    _, marks = run_quiet(e, (ACCEPTED + ["v", "vp", "mvp", "pc"])[:41], 128, fwapi.PLANAR_F32)
    assert e.cx.plan_kind() == 2 and e.violation() == "", e.violation()
    assert marks[-1][0] >= 8, marks


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [fwapi.PLANAR_F32, fwapi.INTERLEAVED_I16, fwapi.INTERLEAVED_U16])
@pytest.mark.parametrize("mbf,max_batch", [(128, 8), (64, 3), (256, 64)])
def test_chain_plan_calls_without_a_control_kernel_are_bit_exact_and_happen(fmt, mbf, max_batch):
    """Round 6: lazy records for chain plans.  k_chain derives every block's record from the voice's LazyRec, takes coefficients,
    delay parameters and the delay position from node state, and k_lazy_flush moves the delay positions along with the playheads."""
    shapes = (ACCEPTED + ["v", "vp", "mvp", "pc"])[:41]   # every accepted shape, dry voices beside them
    if fmt != fwapi.PLANAR_F32:
        # (channel 0 of an interleaved sample behind the mono adapter is fetched frame by frame — no compact class, k_control.hip.h
        #  mono_adapt — so such a voice leaves no lazy record and holds the whole plan on the control path: correct, and not this test)
        shapes = [sh for sh in shapes if not sh.startswith("m")]
    ro, _ = run_quiet(scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf)), shapes, mbf, fmt)
    g = GpuEngine(max_block_frames=mbf, max_batch=max_batch)
    rg, marks = run_quiet(g, shapes, mbf, fmt)
    assert g.cx.plan_kind() == 2
    assert_bits(ro, rg, "quiet calls fmt %d mbf %d K<=%d" % (fmt, mbf, max_batch))
    lazy = [m[0] for m in marks]
    if os.environ.get("FWGPU_LAZY") == "0":
        assert lazy[-1] == 0
        return
    # calls 0 and 1: messages and the call after them; the one-shots end in block 6 (call 2): the horizon keeps control in until then
    assert lazy[2] == 0, marks
    assert lazy[7] > lazy[3], marks            # quiet calls 4, 6, 7 (call 5 is ONE block)
    assert lazy[9] == lazy[8], marks           # the messages of call 8 (that call is a control call) and the call after it
    assert lazy[11] > lazy[9], marks           # ... then lazy again, with the coefficients and delay parameters of call 8's END
    assert lazy[13] == lazy[12], marks         # pauses and mutes: control for as long as their glides last ...
    assert lazy[-1] > lazy[17], marks          # ... and lazy once more when the last of them has settled (a glide to 0.0 takes ~45 blocks of 128)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_fuzz_chain_grammar_bit_exact(seed):
    mbf = [128, 64, 256][seed % 3]
    max_batch = [64, 1, 3, 8][seed % 4]
    o = scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf))
    g = GpuEngine(max_block_frames=mbf, max_batch=max_batch)
    ro, rg = fuzz_grammar(o, seed), fuzz_grammar(g, seed)
    a, b = np.asarray(ro), np.asarray(rg)
    bad = np.nonzero(fwapi.bits(a) != fwapi.bits(b))[0]
    assert bad.size == 0, "seed %d (plan %d, mbf %d, K<=%d): %d of %d samples differ, first at %d (block %d)" % (
        seed, g.cx.plan_kind(), mbf, max_batch, bad.size, a.size, bad[0], bad[0] // (2 * mbf))


def test_fuzz_chain_grammar_on_the_host_harness():
    """the same graphs and message streams through the host half on the fake runtime: every table the kernels would read is validated"""
    for seed in range(12):
        e = HostOnlyEngine(max_block_frames=[128, 64, 256][seed % 3], max_batch=[64, 1, 3, 8][seed % 4])
        fuzz_grammar(e, seed)
        assert e.violation() == "", (seed, e.violation())
