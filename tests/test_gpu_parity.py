"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the oracle on the same seeded inputs.
Bar: bit-exact for every node except BeepTest's sinf (ocml vs glibc libm, |err| <= 2e-6 absolute, H6)."""
import json
import os

import numpy as np
import pytest

import fwapi
import scenarios
from fwapi import (BEEP_TEST, DUMMY, HARD_CLIP, MONO_TO_STEREO, PLANAR_F32, SAMPLER, STEREO_PAN, STEREO_TO_MONO, SUM,
                   VOLUME, GpuEngine, OracleEngine, bits)
from test_scenarios_oracle import CASES, GOLDEN, digest

pytestmark = pytest.mark.gpu
f32 = np.float32


def oracle(**kw):
    return scenarios.TaggedOracle(OracleEngine(**kw))


def assert_bits_equal(a, b, what=""):
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    assert a.shape == b.shape, what
    if not np.array_equal(bits(a), bits(b)):
        bad = np.nonzero(bits(a) != bits(b))[0]
        raise AssertionError("%s: %d/%d samples differ, first at %d: %r vs %r" %
                             (what, bad.size, a.size, bad[0], a.flat[bad[0]], b.flat[bad[0]]))


# ------------------------------------------------------------------ native code actually runs
def test_native_library_is_loaded_and_device_is_mi355x():
    g = GpuEngine()
    name, cus, hbm = g.cx.device_info()
    assert cus >= 200 and hbm > 100 * 2 ** 30, (name, cus, hbm)
    maps = open("/proc/self/maps").read()
    assert "firewheel_amd/csrc/libfwgpu.so" in maps


# ------------------------------------------------------------------ node level (B1: AudioNodeProcessor::process)
def pair(kind, n_in, n_out, params=(), mbf=256):
    o = OracleEngine(max_block_frames=mbf)
    g = GpuEngine(max_block_frames=mbf)
    no = o.add_node(kind, n_in, n_out, params)
    ng = g.add_node(kind, n_in, n_out, params)
    o.update()
    g.update()
    return (o, no), (g, ng)


def both(po, pg, frames, x, n_out, in_mask=0, out_init=None, exact=True, what=""):
    (o, no), (g, ng) = po, pg
    yo, mo = o.node_process(no, frames, x, n_out, in_mask=in_mask, out_init=out_init)
    yg, mg = g.node_process(ng, frames, x, n_out, in_mask=in_mask, out_init=out_init)
    assert mo == mg, "%s: out mask %x vs %x" % (what, mo, mg)
    if exact:
        assert_bits_equal(yo, yg, what)
    return yo, yg


@pytest.mark.parametrize("frames,mbf", [(256, 256), (64, 64), (100, 128), (1024, 1024), (1, 4)])
def test_volume_node(frames, mbf):
    po, pg = pair(VOLUME, 2, 2, [50.0], mbf)
    x = fwapi.xorshift_uniform(1, 2 * frames).reshape(2, frames)
    for mask in (0, 0b01, 0b10, 0b11):
        both(po, pg, frames, x, 2, in_mask=mask, what="volume mask %d" % mask)
    # smoothing ramp across several blocks until it settles / stalls (Q1-Q3, Q28)
    po[0].set_param(po[1], 0, 100.0)
    pg[0].set_param(pg[1], 0, 100.0)
    for b in range(40):
        both(po, pg, frames, x, 2, what="volume ramp block %d" % b)
    po[0].set_param(po[1], 0, 0.0)
    pg[0].set_param(pg[1], 0, 0.0)
    for b in range(40):
        both(po, pg, frames, x, 2, what="volume ramp-to-zero block %d" % b)
    # silent input resets the smoother (volume.rs:94-99)
    po[0].set_param(po[1], 0, 30.0)
    pg[0].set_param(pg[1], 0, 30.0)
    both(po, pg, frames, x, 2)
    both(po, pg, frames, x, 2, in_mask=0b11)
    both(po, pg, frames, x, 2)


def test_volume_generic_channels_and_mute():
    po, pg = pair(VOLUME, 5, 5, [73.0])
    x = fwapi.xorshift_uniform(2, 5 * 256).reshape(5, 256)
    for mask in (0, 0b00101, 0b11110, 0b11111):
        both(po, pg, 256, x, 5, in_mask=mask)
    po, pg = pair(VOLUME, 2, 2, [0.0])
    both(po, pg, 256, x[:2], 2)


@pytest.mark.parametrize("ports,ch", [(1, 2), (2, 2), (3, 2), (4, 2), (5, 2), (32, 2), (7, 1), (3, 4), (16, 4)])
def test_sum_node(ports, ch):
    po, pg = pair(SUM, ports * ch, ch)
    import random

    rng = random.Random(ports * 10 + ch)
    x = fwapi.xorshift_uniform(3 + ports, ports * ch * 256).reshape(ports * ch, 256)
    x[rng.randrange(ports * ch)] = -0.0
    full = (1 << (ports * ch)) - 1
    for mask in (0, full, rng.getrandbits(ports * ch), rng.getrandbits(ports * ch), full & ~1, 1):
        both(po, pg, 256, x, ch, in_mask=mask, what="sum %dx%d mask %x" % (ports, ch, mask))
    z = np.full_like(x, -0.0)
    both(po, pg, 256, z, ch, in_mask=full & ~((1 << ch) - 1), what="sum -0.0")


def test_hard_clip_and_adapters():
    po, pg = pair(HARD_CLIP, 2, 2, [-6.0])
    x = fwapi.xorshift_uniform(4, 512).reshape(2, 256) * 2
    for mask in (0, 1, 2, 3):
        both(po, pg, 256, x, 2, in_mask=mask)
    po, pg = pair(HARD_CLIP, 3, 3, [-20.0])
    x3 = fwapi.xorshift_uniform(5, 768).reshape(3, 256)
    for mask in (0, 0b010):
        both(po, pg, 256, x3, 3, in_mask=mask)
    po, pg = pair(MONO_TO_STEREO, 1, 2)
    for mask in (0, 1):
        both(po, pg, 256, x[:1], 2, in_mask=mask)
    po, pg = pair(STEREO_TO_MONO, 2, 1)
    for mask in (0, 1, 3):
        both(po, pg, 256, x, 1, in_mask=mask)


def test_dummy_and_beep_disabled_leave_outputs_untouched():
    po, pg = pair(DUMMY, 1, 2)
    init = fwapi.xorshift_uniform(6, 512).reshape(2, 256)
    both(po, pg, 256, [np.zeros(256, f32)], 2, out_init=init)
    po, pg = pair(BEEP_TEST, 0, 3, [440.0, -12.0, 0.0])
    init = fwapi.xorshift_uniform(7, 768).reshape(3, 256)
    yo, yg = both(po, pg, 256, [], 3, out_init=init, what="beep disabled (Q12)")
    assert np.array_equal(yg[0], init[0])


@pytest.mark.parametrize("freq", [440.0, 19.0, 12345.6])
def test_beep_node_within_libm_tolerance(freq):
    po, pg = pair(BEEP_TEST, 0, 2, [freq, -12.0, 1.0])
    for b in range(6):
        yo, yg = both(po, pg, 256, [], 2, exact=False)
        # gain <= 0.25: |ocml sinf - glibc sinf| <= ~2 ulp of 1.0 => 2e-6 absolute is generous (H6)
        assert np.max(np.abs(yo - yg)) <= 2e-6
        assert np.array_equal(yg[0], yg[1])


def test_pan_node():
    po, pg = pair(STEREO_PAN, 2, 2, [0.3])
    x = fwapi.xorshift_uniform(8, 512).reshape(2, 256)
    for mask in (0, 1, 3):
        both(po, pg, 256, x, 2, in_mask=mask)
    for p in (-1.0, 1.0, 0.0, -0.7):
        po[0].set_param(po[1], 0, p)
        pg[0].set_param(pg[1], 0, p)
        for b in range(30):
            both(po, pg, 256, x, 2, what="pan %g block %d" % (p, b))


@pytest.mark.parametrize("fmt", list(range(6)))
@pytest.mark.parametrize("channels,n_out", [(1, 1), (1, 2), (2, 2), (2, 1), (3, 2), (2, 4)])
def test_sampler_formats(fmt, channels, n_out):
    rng = np.random.default_rng(fmt * 10 + channels)
    frames = 700
    dt = fwapi._FMT_DTYPE[fmt]
    if dt == np.float32:
        raw = (rng.random((channels, frames), dtype=f32) * 2 - 1).astype(f32)
    elif dt == np.int16:
        raw = rng.integers(-32768, 32768, size=(channels, frames)).astype(dt)
    else:
        raw = rng.integers(0, 65536, size=(channels, frames)).astype(dt)
    data = raw.T.copy() if fmt <= 2 else raw
    po, pg = pair(SAMPLER, 0, n_out, [80.0])
    for e, n in (po, pg):
        s = e.new_sample(fmt, channels, data)
        e.sampler_set_sample(n, s)
        e.sampler_play(n)
    init = np.full((n_out, 256), 9.0, f32)
    for b in range(4):  # last block crosses the one-shot end (Q9)
        both(po, pg, 256, [], n_out, out_init=init, what="sampler fmt %d ch %d out %d block %d" % (fmt, channels, n_out, b))


# ------------------------------------------------------------------ graph level
def run_case(name, **gpu_kw):
    out_o = CASES[name]()
    fn = {
        "steady_96x32": lambda e: scenarios.scenario_voice_bank_steady(e, 96, 6),
        "ref_steady_64": lambda e: scenarios.scenario_voice_bank_steady(e, 64, 6, with_pan=False),
        "ref_steady_33_i16_r8": lambda e: scenarios.scenario_voice_bank_steady(e, 33, 9, radix=8, with_pan=False, fmt=fwapi.INTERLEAVED_I16,
                                                                                mono_every=5, src_frames=777),
        "ref_desk_30": scenarios.scenario_ref_desk,
        "ref_desk_21_b64": lambda e: scenarios.scenario_ref_desk(e, 21, radix=4, src_frames=500, seed=9),
        "steady_40x4_i16": lambda e: scenarios.scenario_voice_bank_steady(e, 40, 5, radix=4, fmt=fwapi.INTERLEAVED_I16),
        "steady_9x3_u16": lambda e: scenarios.scenario_voice_bank_steady(e, 9, 4, radix=3, fmt=fwapi.PLANAR_U16),
        "steady_fmt_p_i16_mono3": lambda e: scenarios.scenario_voice_bank_steady(e, 20, 9, radix=8, fmt=fwapi.PLANAR_I16,
                                                                                 mono_every=3, src_frames=1000),
        "steady_fmt_i_f32": lambda e: scenarios.scenario_voice_bank_steady(e, 11, 7, radix=4, fmt=fwapi.INTERLEAVED_F32,
                                                                           mono_every=4, src_frames=700),
        "steady_fmt_i_u16": lambda e: scenarios.scenario_voice_bank_steady(e, 9, 7, radix=16, fmt=fwapi.INTERLEAVED_U16,
                                                                           mono_every=2, src_frames=600),
        "steady_fmt_p_i16_oddlen": lambda e: scenarios.scenario_voice_bank_steady(e, 7, 20, radix=8, fmt=fwapi.PLANAR_I16,
                                                                                  src_frames=333),
        "steady_fmt_mixed_leaf": lambda e: scenarios.scenario_voice_bank_steady(e, 26, 8, radix=32, fmt_cycle=list(range(6)),
                                                                                mono_every=5, src_frames=900),
        "events_33_i16": lambda e: scenarios.scenario_voice_bank_events(e, 33, radix=8, src_frames=777,
                                                                        fmt=fwapi.INTERLEAVED_I16),
        "voice_fx_steady": lambda e: scenarios.scenario_voice_bank_steady(e, 70, 6, src_frames=1500, voice_fx=scenarios.width_clip_fx),
        "voice_fx_events_45": scenarios.scenario_voice_fx_events,
        "voice_fx_events_20_i16_r32": lambda e: scenarios.scenario_voice_fx_events(e, 20, radix=32, src_frames=500, with_pan=False,
                                                                                  fmt=fwapi.INTERLEAVED_I16),
        "rs_bank_40": scenarios.scenario_rs_bank,
        "rs_bank_21_b64_i16_pure": lambda e: scenarios.scenario_rs_bank(e, 21, radix=32, src_frames=400, mixed=False, fmt=fwapi.INTERLEAVED_I16),
        "events_70": lambda e: scenarios.scenario_voice_bank_events(e, 70),
        "events_33_r2": lambda e: scenarios.scenario_voice_bank_events(e, 33, radix=2, src_frames=777),
        "spatial_steady_b128": scenarios.scenario_spatial_steady,
        "spatial_steady_b64": lambda e: scenarios.scenario_spatial_steady(e, 5, src_frames=1500, calls=(3, 90, 33, 7)),
        "hybrid_sends_b128": scenarios.scenario_hybrid_sends,
        "hybrid_sends_b64": lambda e: scenarios.scenario_hybrid_sends(e, 17, 9, src_frames=900, seed=8, long_call=61),
        "hybrid_chain_sends_b128": scenarios.scenario_hybrid_chain_sends,
        "hybrid_chain_sends_b64": lambda e: scenarios.scenario_hybrid_chain_sends(e, 21, radix=7, src_frames=800, seed=12, long_call=70),
        "split_mixers_b128": scenarios.scenario_split_mixers,
        "split_mixers_b64": lambda e: scenarios.scenario_split_mixers(e, seed=22, long_call=45, src_frames=700),
        "bus_iir_b256": scenarios.scenario_bus_iir,
        "bus_iir_b512": lambda e: scenarios.scenario_bus_iir(e, seed=32, src_frames=9000),
        "bus_iir_b128": lambda e: scenarios.scenario_bus_iir(e, seed=33),
        "storm_48x6": scenarios.scenario_message_storm,
        "storm_200x50_b64": lambda e: scenarios.scenario_message_storm(e, 200, radix=32, blocks=60, per_voice=50, src_frames=3000, seed=4),
        "mixed_generic": scenarios.scenario_mixed_generic,
        "mixed_generic_nobeep": lambda e: scenarios.scenario_mixed_generic(e, use_beep=False),
        "graph_inputs": scenarios.scenario_graph_inputs,
        "cfg3_chain": scenarios.scenario_cfg3_chain,
        "spatial_scene": scenarios.scenario_spatial_scene,
        "spatial_scene_b96": lambda e: scenarios.scenario_spatial_scene(e, n_sources=4, blocks=9),
        "chain_steady_40": lambda e: scenarios.scenario_chain_steady(e, 40, 6),
        "chain_steady_bq_only_i16": lambda e: scenarios.scenario_chain_steady(e, 21, 9, radix=4, delay=False,
                                                                              fmt=fwapi.INTERLEAVED_I16),
        "chain_steady_dl_only_pan": lambda e: scenarios.scenario_chain_steady(e, 10, 7, radix=3, biquad=False, with_pan=True),
        "chain_events_37": lambda e: scenarios.scenario_chain_events(e, 37),
        "chain_events_19_r2_pan": lambda e: scenarios.scenario_chain_events(e, 19, radix=2, src_frames=777, with_pan=True),
        # every delay >= 128 frames and block % 128 == 0: k_chain runs its 128-frame tiles (ring prefetch for D >= 256)
        "chain_steady_40_d128": lambda e: scenarios.scenario_chain_steady(e, 40, 6, first_delay_frames=128, min_delay_frames=129),
        "chain_events_37_d130": lambda e: scenarios.scenario_chain_events(e, 37, first_delay_frames=130, min_delay_frames=128),
        # two 128-frame tiles per block, every ring prefetched (delays >= 256 frames)
        "chain_events_21_d256": lambda e: scenarios.scenario_chain_events(e, 21, first_delay_frames=256, min_delay_frames=257,
                                                                          src_frames=1500),
        "chain_calls_37_b256": lambda e: scenarios.scenario_chain_steady_calls(e, 37, tile=128),
        "chain_calls_20_b128_pan": lambda e: scenarios.scenario_chain_steady_calls(e, 20, tile=128, with_pan=True),
        "chain_calls_33_b64": lambda e: scenarios.scenario_chain_steady_calls(e, 33, tile=64),
        "chain_calls_37_b256_wrap": lambda e: scenarios.scenario_chain_steady_calls(e, 37, tile=128, src_extra=77),
        "chain_calls_21_b128_wrap": lambda e: scenarios.scenario_chain_steady_calls(e, 21, tile=128, src_extra=130),
        "chain_calls_33_b64_wrap": lambda e: scenarios.scenario_chain_steady_calls(e, 33, tile=64, src_extra=5),
        "master_chain_bank": scenarios.scenario_master_chain,
        "master_chain_fx": lambda e: scenarios.scenario_master_chain(e, chain=True, n_voices=37),
        "cfg4_reverb": scenarios.scenario_cfg4_reverb,
        "cfg4_reverb_2irs_mono": lambda e: scenarios.scenario_cfg4_reverb(e, n_voices=5, taps=700, shared_ir=False,
                                                                          ir_channels=1),
    }[name]
    mbf = {"rs_bank_40": 128, "rs_bank_21_b64_i16_pure": 64, "voice_fx_steady": 256, "voice_fx_events_45": 128, "voice_fx_events_20_i16_r32": 64, "steady_fmt_p_i16_mono3": 128, "steady_fmt_i_f32": 64, "steady_fmt_i_u16": 64, "steady_fmt_p_i16_oddlen": 64,
           "steady_fmt_mixed_leaf": 128, "events_33_i16": 128, "steady_96x32": 256, "steady_40x4_i16": 64, "steady_9x3_u16": 128, "events_70": 256, "events_33_r2": 128, "storm_48x6": 128, "storm_200x50_b64": 64, "bus_iir_b256": 256, "bus_iir_b512": 512, "bus_iir_b128": 128, "hybrid_sends_b128": 128, "hybrid_sends_b64": 64, "split_mixers_b128": 128, "split_mixers_b64": 64, "hybrid_chain_sends_b128": 128, "hybrid_chain_sends_b64": 64, "spatial_steady_b128": 128, "spatial_steady_b64": 64,
           "mixed_generic": 256, "mixed_generic_nobeep": 256, "graph_inputs": 64, "cfg3_chain": 128, "cfg4_reverb": 128,
           "cfg4_reverb_2irs_mono": 64, "chain_steady_40": 256, "chain_steady_bq_only_i16": 64,
           "chain_steady_dl_only_pan": 128, "chain_events_37": 128, "chain_events_19_r2_pan": 64,
           "chain_steady_40_d128": 256, "chain_events_37_d130": 128, "chain_events_21_d256": 256, "chain_calls_37_b256": 256, "chain_calls_20_b128_pan": 128,
           "chain_calls_33_b64": 64, "chain_calls_37_b256_wrap": 256, "chain_calls_21_b128_wrap": 128,
           "chain_calls_33_b64_wrap": 64, "master_chain_bank": 128, "master_chain_fx": 128, "spatial_scene": 128, "spatial_scene_b96": 96,
           "ref_steady_64": 256, "ref_steady_33_i16_r8": 64, "ref_desk_30": 128, "ref_desk_21_b64": 64}[name]
    kw = dict(max_block_frames=mbf)
    if name == "graph_inputs":
        kw["num_graph_inputs"] = 3
    kw.update(gpu_kw)
    g = GpuEngine(**kw)
    out_g = fn(g)
    return out_o, out_g, g


VOICE_CASES = ["rs_bank_40", "rs_bank_21_b64_i16_pure", "voice_fx_steady", "voice_fx_events_45", "voice_fx_events_20_i16_r32", "steady_96x32", "steady_40x4_i16", "steady_9x3_u16", "events_70", "events_33_r2", "storm_48x6", "storm_200x50_b64", "steady_fmt_p_i16_mono3",
               "steady_fmt_i_f32", "steady_fmt_i_u16", "steady_fmt_p_i16_oddlen", "steady_fmt_mixed_leaf", "events_33_i16"]


@pytest.mark.parametrize("name", VOICE_CASES)
def test_voice_bank_generic_executor_bit_exact(name):
    out_o, out_g, g = run_case(name, force_generic=True)
    assert g.cx.plan_kind() == 0
    assert_bits_equal(out_o, out_g, name + " generic")


@pytest.mark.parametrize("name", VOICE_CASES)
@pytest.mark.parametrize("max_batch", [64, 3, 1])
def test_voice_bank_fused_plan_bit_exact(name, max_batch):
    out_o, out_g, g = run_case(name, max_batch=max_batch)
    assert g.cx.plan_kind() == 1, "fused voice-bank plan was not selected"
    assert_bits_equal(out_o, out_g, name + " fused K<=%d" % max_batch)
    gold = json.load(open(GOLDEN))
    assert digest(out_g) == gold[name]


@pytest.mark.parametrize("name", ["mixed_generic_nobeep", "cfg3_chain", "spatial_scene", "cfg4_reverb", "cfg4_reverb_2irs_mono",
                                  "graph_inputs", "events_33_r2", "chain_events_19_r2_pan"])
@pytest.mark.parametrize("max_batch", [1, 3, 64])
def test_generic_executor_k_batched_bit_exact(name, max_batch):
    """the generic level-batched executor runs K blocks per launch (one pool slice per block; stateful nodes walk
    their K blocks in order inside one wave, FIR banks become one K-block GEMM): same bits for every K"""
    out_o, out_g, g = run_case(name, max_batch=max_batch, force_generic=True)
    assert g.cx.plan_kind() == 0
    assert_bits_equal(out_o, out_g, "%s generic K<=%d" % (name, max_batch))


def test_fir_history_ring_caps_the_generic_batch_when_kmax_grows_later():
    # the FIR ring is sized for the batch size in force at activation; raising max_batch afterwards must not outrun it
    def run(e, grow):
        out = []
        ir = e.new_sample(PLANAR_F32, 2, scenarios.reverb_ir(5, 300, 2))
        s = e.sampler(90.0)
        f = e.fir(ir)
        e.connect_stereo(s, f)
        e.connect_stereo(f, e.graph_out_node)
        e.update()
        e.sampler_set_sample(s, e.new_sample(PLANAR_F32, 2, scenarios.voice_source(55, 5000)))
        e.sampler_play(s)
        out.append(e.process_blocks(3))
        if grow:
            e.cx.set_max_batch(16)
            e.update()
        out.append(e.process_blocks(9))
        return np.concatenate(out)

    o = oracle(max_block_frames=64)
    g = GpuEngine(max_block_frames=64, max_batch=2)
    assert_bits_equal(run(o, False), run(g, True), "FIR ring vs later kmax")


def test_mixed_graph_generic_executor():
    out_o, out_g, g = run_case("mixed_generic")
    assert g.cx.plan_kind() == 0
    # the beep branch goes through sinf (ocml vs glibc): absolute tolerance (H6)
    assert np.max(np.abs(out_o - out_g)) <= 4e-6
    # same graph with the beep replaced by a mono one-shot sampler: bit-exact
    out_o, out_g, g = run_case("mixed_generic_nobeep")
    assert g.cx.plan_kind() == 0
    assert_bits_equal(out_o, out_g, "mixed graph")


def test_cfg3_chain_spec_nodes_bit_exact():
    # (a width behind the delay is not a chain-plan voice: the banks holding one stay on the level executor, the others go
    # through k_chain — hybrid plan; then everything on the level executor)
    out_o, out_g, g = run_case("cfg3_chain")
    assert g.cx.plan_kind() in (0, 3)
    assert_bits_equal(out_o, out_g, "cfg3 chain (biquad + delay + width)")
    gold = json.load(open(GOLDEN))
    assert digest(out_g) == gold["cfg3_chain"]
    out_o, out_g, g = run_case("cfg3_chain", force_generic=True)
    assert g.cx.plan_kind() == 0
    assert_bits_equal(out_o, out_g, "cfg3 chain (biquad + delay + width), level executor alone")


CHAIN_CASES = ["chain_steady_40", "chain_steady_bq_only_i16", "chain_steady_dl_only_pan", "chain_events_37",
               "chain_events_19_r2_pan", "chain_steady_40_d128", "chain_events_37_d130", "chain_events_21_d256"]


@pytest.mark.parametrize("name", CHAIN_CASES)
def test_chain_bank_generic_executor_bit_exact(name):
    out_o, out_g, g = run_case(name, force_generic=True)
    assert g.cx.plan_kind() == 0
    assert_bits_equal(out_o, out_g, name + " generic")


@pytest.mark.parametrize("name", CHAIN_CASES)
@pytest.mark.parametrize("max_batch", [64, 3, 1])
def test_chain_bank_fused_chain_plan_bit_exact(name, max_batch):
    """config-3 voices (sampler -> biquad -> delay -> gain) through k_chain: serial DF1 recurrence in packed f32,
    delay-line RMW in HBM, ordered leaf sums — bit-identical to the oracle for every K batching."""
    out_o, out_g, g = run_case(name, max_batch=max_batch)
    assert g.cx.plan_kind() == 2, "fused chain plan was not selected"
    assert_bits_equal(out_o, out_g, name + " k_chain K<=%d" % max_batch)
    gold = json.load(open(GOLDEN))
    assert digest(out_g) == gold[name]


@pytest.mark.parametrize("name", ["chain_calls_37_b256", "chain_calls_20_b128_pan", "chain_calls_33_b64",
                                  "chain_calls_37_b256_wrap", "chain_calls_21_b128_wrap", "chain_calls_33_b64_wrap"])
@pytest.mark.parametrize("max_batch", [64, 4])
def test_chain_plan_steady_call_loop_bit_exact(name, max_batch):
    # calls that qualify for k_chain's steady-call loop (loads two tiles ahead, branch-free worker steps) between calls
    # that do not: both loops must have run, and the stream must equal the oracle's bit for bit
    out_o, out_g, g = run_case(name, max_batch=max_batch)
    assert g.cx.plan_kind() == 2
    steady, general = g.cx.plan_chain_stats()
    # start-up, the message burst and the blocks in which its ramps settle go through the general loop; the other four
    # calls — also when their loops wrap inside blocks — must have qualified for the steady-call loop
    n_wg = 2 * ((int(name.split("_")[2]) + 31) // 32)  # one workgroup per (leaf, channel)
    assert steady >= 4 * n_wg and general > 0, (steady, general, n_wg)
    assert_bits_equal(out_o, out_g, name + " k_chain steady calls K<=%d" % max_batch)
    gold = json.load(open(GOLDEN))
    assert digest(out_g) == gold[name]


@pytest.mark.parametrize("name,nq", [("chain_steady_40_d128", "1"), ("chain_events_37_d130", "1"), ("chain_events_21_d256", "1")])
def test_chain_plan_small_tiles_forced(name, nq, monkeypatch):
    # the same scenarios through a smaller-tile instantiation (FWGPU_CHAIN_NQ overrides the planner's choice downwards)
    monkeypatch.setenv("FWGPU_CHAIN_NQ", nq)
    out_o, out_g, g = run_case(name, max_batch=5)
    assert g.cx.plan_kind() == 2
    assert_bits_equal(out_o, out_g, name + " k_chain<1>")


def test_chain_plan_falls_back_when_a_delay_is_shorter_than_a_tile():
    # D = 48 < 64: the generic executor's chunked in-block recurrence handles it (and must stay bit-exact)
    def run(e):
        voices = scenarios.build_chain_bank(e, 6, radix=3, min_delay_frames=20, max_delay_frames=60)
        for vc in voices[1:]:
            e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_FULL)
            e.sampler_play(vc["sampler"])
        return e.process_blocks(5)

    o = oracle(max_block_frames=128)
    g = GpuEngine(max_block_frames=128)
    ro, rg = run(o), run(g)
    assert g.cx.plan_kind() == 0
    assert_bits_equal(ro, rg, "short delays")


def test_chain_plan_state_survives_plan_switches():
    # k_chain -> generic -> k_chain mid-stream: biquad history, ring position, smoothers carry over exactly
    def run(e):
        voices = scenarios.build_chain_bank(e, 9, radix=4, src_frames=700)
        for vc in voices:
            e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_FULL)
            e.sampler_play(vc["sampler"])
        e.set_param(voices[0]["volume"], 0, 5.0, at_block=2)
        a = e.process_blocks(4)
        extra = e.sum(2)            # a dangling node: the fused plan no longer covers the graph
        e.update()
        b = e.process_blocks(3)
        e.remove_node(extra)
        e.update()
        c = e.process_blocks(4)
        return np.concatenate([a, b, c])

    o = oracle(max_block_frames=128)
    g = GpuEngine(max_block_frames=128)
    ro, rg = run(o), run(g)
    assert g.cx.plan_kind() == 2
    assert_bits_equal(ro, rg, "chain plan across graph edits")


def test_config3_full_size_chain_plan_equals_generic_and_oracle_prefix():
    # BASELINE configs[2]: 4096 voices, biquad LPF + delay, block = 512.  Fused (k_chain) == generic executor on
    # the whole run; the oracle on the first block (sized to finish in seconds).
    V, blocks = 4096, 3
    gf = GpuEngine(max_block_frames=512, max_batch=8)
    kw = dict(src_frames=2048, first_delay_frames=480, min_delay_frames=480, max_delay_frames=12000)  # 10..250 ms
    of = scenarios.scenario_chain_steady(gf, V, blocks, **kw)
    assert gf.cx.plan_kind() == 2
    gg = GpuEngine(max_block_frames=512, force_generic=True)
    og = scenarios.scenario_chain_steady(gg, V, blocks, **kw)
    assert_bits_equal(of, og, "4096 voices k_chain vs generic")
    assert np.all(np.isfinite(of)) and np.std(of) > 0.1
    o = oracle(max_block_frames=512)
    oo = scenarios.scenario_chain_steady(o, V, 1, **kw)
    assert_bits_equal(oo, of[:oo.size], "4096 voices vs oracle")


@pytest.mark.parametrize("name", ["hybrid_sends_b128", "hybrid_sends_b64", "hybrid_chain_sends_b128", "hybrid_chain_sends_b64",
                                  "split_mixers_b128", "split_mixers_b64"])
@pytest.mark.parametrize("max_batch", [64, 8, 1])
def test_hybrid_plan_voice_banks_inside_a_generic_graph_bit_exact(name, max_batch):
    """buses consumed twice (send + dry), a return chain, a spatialised source: not a fused shape — the three voice banks whose
    SumNodes only take voice chains are rendered by the voice-bank kernels (plan kind 3), the rest by the level executor"""
    out_o, out_g, g = run_case(name, max_batch=max_batch)
    assert g.cx.plan_kind() == 3
    assert_bits_equal(out_o, out_g, name)
    gold = json.load(open(GOLDEN))
    assert digest(out_g) == gold[name]
    out_o2, out_g2, g2 = run_case(name, force_generic=True)
    assert g2.cx.plan_kind() == 0
    assert_bits_equal(out_o2, out_g2, name + " (level executor alone)")


@pytest.mark.parametrize("name", ["spatial_scene", "spatial_scene_b96", "spatial_steady_b128", "spatial_steady_b64"])
def test_resampler_and_spatialiser_scene_bit_exact(name):
    # (the steady scenes: resting spatialisers take their K blocks in parallel — max_batch 64 splits the 70- / 90-block calls)
    out_o, out_g, g = run_case(name)
    # resampler -> spatialiser sources: the level executor; sampler -> spatialiser voices under a mixer: the voice-bank plan's
    # spatialiser stage (round 3: SK_SPATIAL in k_leaf_sum, history re-rendered from the block before)
    assert g.cx.plan_kind() == (1 if name.startswith("spatial_steady") else 0)
    assert_bits_equal(out_o, out_g, name)
    gold = json.load(open(GOLDEN))
    assert digest(out_g) == gold[name]


@pytest.mark.parametrize("mbf,max_batch", [(128, 64), (64, 8), (256, 1), (512, 16)])
def test_spatialiser_voices_on_the_voice_bank_plan_bit_exact(mbf, max_batch):
    g = GpuEngine(max_block_frames=mbf, max_batch=max_batch)
    o = scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf))
    out_g = scenarios.scenario_spatial_bank(g)
    out_o = scenarios.scenario_spatial_bank(o)
    assert g.cx.plan_kind() == 1 and g.cx.plan_fused_voices() == 23
    assert_bits_equal(out_o, out_g, "spatial bank mbf %d batch %d" % (mbf, max_batch))
    # ... and the level executor agrees (same graph, generic plan): the two paths hand the history to each other's blocks
    f = GpuEngine(max_block_frames=mbf, max_batch=max_batch, force_generic=True)
    assert_bits_equal(out_o, scenarios.scenario_spatial_bank(f), "spatial bank, level executor")


@pytest.mark.parametrize("fmt", [fwapi.PLANAR_F32, fwapi.INTERLEAVED_I16, fwapi.PLANAR_U16])
@pytest.mark.parametrize("ratio,loop", [(1.0, False), (44100.0 / 48000.0, True), (2.7, True), (0.11, False), (3.0, False)])
def test_resampler_node_level(fmt, ratio, loop):
    from fwapi import RESAMPLER

    frames = 300
    rng = np.random.default_rng(int(ratio * 100) + fmt)
    if fmt == fwapi.PLANAR_F32:
        raw = (rng.random((2, frames), dtype=f32) * 2 - 1).astype(f32)
    elif fmt == fwapi.INTERLEAVED_I16:
        raw = rng.integers(-32768, 32768, size=(frames, 2)).astype(np.int16)
    else:
        raw = rng.integers(0, 65536, size=(2, frames)).astype(np.uint16)
    for n_out in (1, 2, 3):
        o = OracleEngine(max_block_frames=128)
        g = GpuEngine(max_block_frames=128)
        nodes = []
        for e in (o, g):
            smp = e.new_sample(fmt, 2, raw)
            nodes.append(e.resampler(smp, ratio, loop=loop, n_out=n_out))
            e.update()
        for b in range(8):
            both((o, nodes[0]), (g, nodes[1]), 128, [], n_out, what="resampler ratio %g loop %d block %d" % (ratio, loop, b))


@pytest.mark.parametrize("n_in", [1, 2])
@pytest.mark.parametrize("pos", [(0.0, 0.0, -1.0), (4.0, 1.0, 0.5), (-0.3, 0.0, 0.1), (0.0, 0.0, 0.0), (-20.0, 5.0, -3.0)])
def test_spatial_node_level(n_in, pos):
    from fwapi import SPATIAL

    po, pg = pair(SPATIAL, n_in, 2, list(pos), mbf=128)
    x = fwapi.xorshift_uniform(17 + n_in, n_in * 128 * 6).reshape(6, n_in, 128)
    for b in range(3):
        both(po, pg, 128, x[b], 2, what="spatial %r block %d" % (pos, b))
    for e, n in (po, pg):
        e.set_param(n, 0, -pos[0] + 1.0)      # the source jumps to the other side: gains ramp, delays swap ears
    for b in range(3, 6):
        both(po, pg, 128, x[b], 2, what="spatial moved block %d" % b)
    both(po, pg, 40, x[0][:, :40], 2, what="spatial partial block")   # frames < SP_HIST: history shifts


@pytest.mark.parametrize("name", ["cfg4_reverb", "cfg4_reverb_2irs_mono"])
def test_cfg4_fir_reverb_mfma_bit_exact(name):
    out_o, out_g, g = run_case(name)
    # resampler -> spatialiser sources: the level executor; sampler -> spatialiser voices under a mixer: the voice-bank plan's
    # spatialiser stage (round 3: SK_SPATIAL in k_leaf_sum, history re-rendered from the block before)
    assert g.cx.plan_kind() == (1 if name.startswith("spatial_steady") else 0)
    assert_bits_equal(out_o, out_g, name)
    gold = json.load(open(GOLDEN))
    assert digest(out_g) == gold[name]


def test_fir_long_ir_impulse_and_linearity_properties():
    # full-length 65536-tap IR (config 4), too slow for the oracle: size-independent properties instead
    taps, frames = 65536, 256
    h = scenarios.reverb_ir(77, taps, 2, decay=16384.0)

    def run(sig):
        g = GpuEngine(max_block_frames=frames, num_graph_inputs=2)
        ir = g.new_sample(PLANAR_F32, 2, h)
        f = g.fir(ir)
        g.connect_stereo(g.graph_in_node, f)
        g.connect_stereo(f, g.graph_out_node)
        g.update()
        return g.process_interleaved(sig.shape[0], 2, inp=sig.reshape(-1), n_in_ch=2).reshape(-1, 2)

    n = 3 * frames
    imp = np.zeros((n, 2), f32)
    imp[5] = (1.0, 0.5)
    y = run(imp)
    # an impulse reproduces the impulse response exactly: every product but one is x*0 and 1.0*h is exact
    assert np.array_equal(y[5:, 0], h[0, :n - 5])
    assert np.array_equal(y[5:, 1], (h[1, :n - 5] * f32(0.5)).astype(f32))
    assert not np.any(y[:5])
    # scaling by a power of two is exact through the whole fmaf chain
    x = fwapi.xorshift_uniform(123, 2 * n).reshape(n, 2)
    y1 = run(x)
    y2 = run((x * f32(0.25)).astype(f32))
    assert np.array_equal(y2, (y1 * f32(0.25)).astype(f32))
    # and against an f64 convolution (H7 bound)
    ref = np.convolve(x[:, 0].astype(np.float64), h[0].astype(np.float64))[:n]
    bound = np.convolve(np.abs(x[:, 0]).astype(np.float64), np.abs(h[0]).astype(np.float64))[:n]
    assert np.max(np.abs(y1[:, 0] - ref) / np.maximum(bound, 1e-30)) < 64 * 2.0 ** -24


@pytest.mark.parametrize("ch", [1, 2, 5])
def test_spec_nodes_node_level(ch):
    from fwapi import BIQUAD, DELAY, STEREO_WIDTH

    x = fwapi.xorshift_uniform(40 + ch, ch * 256).reshape(ch, 256)
    for ftype in (0, 1, 2):
        po, pg = pair(BIQUAD, ch, ch, [float(ftype), 1234.5, 1.3])
        for b in range(4):
            both(po, pg, 256, x, ch, what="biquad type %d block %d" % (ftype, b))
        po[0].set_param(po[1], 1, 4000.0)
        pg[0].set_param(pg[1], 1, 4000.0)
        for b in range(3):
            both(po, pg, 256, x * 0 if b == 2 else x, ch, what="biquad after cutoff change")
    for secs, fb in ((0.02, 0.0), (0.001, 0.6), (7 / 48000.0, 0.9), (1 / 48000.0, 0.5)):   # D = 960, 48, 7, 1
        po, pg = pair(DELAY, ch, ch, [secs, fb, 0.4])
        for b in range(5):
            both(po, pg, 256, x, ch, what="delay %g fb %g block %d" % (secs, fb, b))
    if ch == 2:
        po, pg = pair(STEREO_WIDTH, 2, 2, [1.5])
        for mask in (0, 1, 3):
            both(po, pg, 256, x, 2, in_mask=mask)
        po[0].set_param(po[1], 0, 0.2)
        pg[0].set_param(pg[1], 0, 0.2)
        for b in range(30):
            both(po, pg, 256, x, 2, what="width ramp %d" % b)


def test_graph_inputs_and_partial_blocks():
    out_o, out_g, g = run_case("graph_inputs")
    assert_bits_equal(out_o, out_g, "graph inputs")


def test_imported_reference_schedule_runs_identically():
    """Keep Firewheel's own scheduler: compile with the ORACLE's restated compiler (reference buffer
    assignment, LIFO reuse) and hand that CompiledSchedule to fwgpu_schedule_upload."""
    o = oracle(max_block_frames=128)
    out_o = scenarios.scenario_voice_bank_steady(o, 20, 5, radix=4, src_frames=999)
    sched = o.e.schedule()
    nbuf = o.e.num_buffers()

    class Importing(GpuEngine):
        def update(self_inner):
            # node ids are identical on both sides (same slot/generation sequence)
            self_inner.cx.schedule_upload(sched, nbuf)

    g = Importing(max_block_frames=128)
    out_g = scenarios.scenario_voice_bank_steady(g, 20, 5, radix=4, src_frames=999)
    assert_bits_equal(out_o, out_g, "imported schedule")
    assert g.cx.plan_kind() == 1


# ------------------------------------------------------------------ full BASELINE sizes: size-independent properties
def test_config2_full_size_fused_equals_generic_and_is_loop_periodic():
    V, blocks, src = 1024, 16, 2048
    gf = GpuEngine(max_block_frames=256, max_batch=16)
    gg = GpuEngine(max_block_frames=256, force_generic=True)
    of = scenarios.scenario_voice_bank_steady(gf, V, blocks, src_frames=src)
    og = scenarios.scenario_voice_bank_steady(gg, V, blocks, src_frames=src)
    assert gf.cx.plan_kind() == 1 and gg.cx.plan_kind() == 0
    assert_bits_equal(of, og, "1024 voices fused vs generic")
    fr = of.reshape(-1, 2)
    assert np.array_equal(fr[:src], fr[src:2 * src])          # loop periodicity
    assert np.all(np.isfinite(of)) and np.std(of) > 1.0
    # and the oracle agrees on the first blocks (sized to finish in seconds)
    o = oracle(max_block_frames=256)
    oo = scenarios.scenario_voice_bank_steady(o, V, 2, src_frames=src)
    assert_bits_equal(oo, of[:oo.size], "1024 voices vs oracle")


def test_config5_shard_full_size_fused_equals_generic_and_oracle_prefix():
    # BASELINE configs[4], one GPU's shard: 8192 voices, block 1024, tree 256 + 8 + 1
    V, blocks, src = 8192, 4, 2048
    gf = GpuEngine(max_block_frames=1024, max_batch=4)
    of = scenarios.scenario_voice_bank_steady(gf, V, blocks, src_frames=src)
    assert gf.cx.plan_kind() == 1
    gg = GpuEngine(max_block_frames=1024, force_generic=True, max_batch=2)
    og = scenarios.scenario_voice_bank_steady(gg, V, blocks, src_frames=src)
    assert_bits_equal(of, og, "8192 voices fused vs generic")
    fr = of.reshape(-1, 2)
    assert np.array_equal(fr[:src], fr[src:2 * src])          # loop periodicity
    o = oracle(max_block_frames=1024)
    oo = scenarios.scenario_voice_bank_steady(o, V, 1, src_frames=src)
    assert_bits_equal(oo, of[:oo.size], "8192 voices vs oracle")


def test_all_paused_bank_outputs_exact_zeros():
    g = GpuEngine(max_block_frames=256, max_batch=8)
    voices = scenarios.build_voice_bank(g, 200)
    out = g.process_blocks(8)
    assert g.cx.plan_kind() == 1
    assert not np.any(bits(out))  # +0.0 everywhere (interleave_stereo zero-fill, util.rs:129-134)


def test_block_1024_large_bank_matches_oracle():
    # config-5 shard shape at reduced voice count: block = 1024, radix-32 tree of depth 2
    o = oracle(max_block_frames=1024)
    g = GpuEngine(max_block_frames=1024, max_batch=4)
    oo = scenarios.scenario_voice_bank_steady(o, 70, 5, src_frames=3000)
    og = scenarios.scenario_voice_bank_steady(g, 70, 5, src_frames=3000)
    assert g.cx.plan_kind() == 1
    assert_bits_equal(oo, og, "block 1024")


def test_graph_edit_keeps_node_state_across_recompile():
    # processors persist across schedules (processor.rs:195-197).  Adding a dangling SumNode makes the fused
    # plan ineligible, so the executor switches plans mid-stream: playheads/smoothers must carry over.
    def run(e):
        voices = scenarios.build_voice_bank(e, 5, radix=8, src_frames=600)
        for vc in voices:
            e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_FULL)
            e.sampler_play(vc["sampler"])
        e.set_param(voices[0]["volume"], 0, 5.0, at_block=2)   # a ramp in flight across the edit
        a = e.process_blocks(3)
        extra = e.sum(2)
        e.update()
        b = e.process_blocks(3)
        e.remove_node(extra)
        e.update()
        c = e.process_blocks(3)
        return np.concatenate([a, b, c])

    o = oracle(max_block_frames=128)
    g = GpuEngine(max_block_frames=128)
    ro = run(o)
    rg = run(g)
    assert g.cx.plan_kind() == 1
    assert_bits_equal(ro, rg, "across graph edits")


@pytest.mark.parametrize("name", ["bus_iir_b256", "bus_iir_b512", "bus_iir_b128"])
@pytest.mark.parametrize("kw", [dict(max_batch=64), dict(max_batch=3), dict(max_batch=64, force_generic=True)])
def test_bus_filters_and_delays_walked_over_whole_batches_bit_exact(name, kw):
    """stereo and mono bus biquads / delays: k_bus_iir's batch walkers (whole 256-frame chunks, K >= 2, delay >= 512 frames, no
    message in the batch) and the block-by-block path they alternate with, on the hybrid plan and on the level executor alone"""
    out_o, out_g, g = run_case(name, **kw)
    assert_bits_equal(out_o, out_g, name)
    gold = json.load(open(GOLDEN))
    assert digest(out_g) == gold[name]


def test_split_mixers_put_their_leading_voices_on_the_voice_bank_kernels():
    out_o, out_g, g = run_case("split_mixers_b128")
    # mixer A: 3 leading voices (+ a null slot) in front of its bus; mixer B: 11 in front of mixer A's bus; the 3-port mixer's two
    # voices, the voice behind mixer A's bus and the bare sampler in front of the mono detour are SOLO voices (round 4: one-port
    # leaves that write the chain's own pool buffers) — 14 + 4
    assert g.cx.plan_kind() == 3 and g.cx.plan_fused_voices() == 18
    assert_bits_equal(out_o, out_g, "split mixers")
    out_o, out_g, g = run_case("hybrid_sends_b128")
    assert g.cx.plan_fused_voices() == 26 + 11 + 9   # banks A1, A2, B whole; bank C's nine voices lead its bus port


@pytest.mark.parametrize("shape", ["voices_behind_the_bus", "two_port_mixers", "chain_voices_behind_the_bus", "chain_bank_with_width"])
@pytest.mark.parametrize("max_batch", [8, 1])
def test_solo_voices_shapes_the_hybrid_plan_used_to_refuse(shape, max_batch):
    """VERDICT r3 missing #5: (a) a mixer whose FIRST port is a bus and whose other ports are voices, (b) a cascade of 2-port
    mixers (voice, the mixer before) — no port run of either is a bank, both fell to the level executor whole.  Round 4 renders
    each such voice as a one-port leaf into its own last node's pool buffers (fwgpu_plan_detect.cpp, solo voices): plan kind 3,
    bit for bit the oracle — pauses put silent flags on both sides, gains glide, (c) the same behind biquad + delay (k_chain), (d) a
    bank of biquad + delay voices some of which end in a StereoWidth: the chain kernels refuse the bank as a whole, its voices go
    solo up to their last gain and the width nodes and the mixer stay on the levels."""
    def build(e):
        rng = np.random.default_rng(77)
        chain = shape in ("chain_voices_behind_the_bus", "chain_bank_with_width")
        def voice(i):
            ch = 1 if i % 6 == 2 else 2
            smp = e.new_sample(PLANAR_F32, ch, scenarios.voice_source(9100 + i, 900 + 13 * i, ch))
            s = e.sampler(100.0)
            cur = s
            if chain:
                bq = e.biquad(0, 400.0 + 90.0 * i, 0.8)
                dl = e.delay(0.004 + 0.0007 * i, 0.3, 0.4)
                e.connect_stereo(cur, bq)
                e.connect_stereo(bq, dl)
                cur = dl
            vol = e.volume(float(rng.uniform(30, 100)))
            e.connect_stereo(cur, vol)
            cur = vol
            if i % 3 == 1:
                pan = e.pan(float(rng.uniform(-1, 1)))
                e.connect_stereo(cur, pan)
                cur = pan
            if i % 5 == 3 and not chain:
                hc = e.hard_clip(-3.0)
                e.connect_stereo(cur, hc)
                cur = hc
            if shape == "chain_bank_with_width" and i % 4 == 1:
                w = e.width(0.3 + 0.1 * i)
                e.connect_stereo(cur, w)
                cur = w
            return dict(sampler=s, smp=smp, volume=vol, end=cur)
        side = voice(99)                       # the bus: a voice through a mono detour (not a voice chain any more)
        if shape != "chain_bank_with_width":
            s2m = e.add_node(fwapi.STEREO_TO_MONO, 2, 1)
            m2s = e.add_node(fwapi.MONO_TO_STEREO, 1, 2)
            e.connect_stereo(side["end"], s2m)
            e.connect(s2m, 0, m2s, 0)
        voices = [voice(i) for i in range(13)]
        if shape == "chain_bank_with_width":
            mix = e.sum(14)
            for p, vc in enumerate(voices + [side]):
                e.connect_stereo(vc["end"], mix, 2 * p)
            e.connect_stereo(mix, e.graph_out_node)
        elif shape == "two_port_mixers":
            bus = m2s
            for vc in voices:
                m = e.sum(2)
                e.connect_stereo(vc["end"], m, 0)
                e.connect_stereo(bus, m, 2)
                bus = m
            e.connect_stereo(bus, e.graph_out_node)
        else:
            mix = e.sum(15)
            e.connect_stereo(m2s, mix, 0)
            for p, vc in enumerate(voices):    # ports 1..13 (14 stays empty)
                e.connect_stereo(vc["end"], mix, 2 * (p + 1))
            e.connect_stereo(mix, e.graph_out_node)
        e.update()
        allv = voices + [side]
        for vc in allv:
            e.sampler_set_sample(vc["sampler"], vc["smp"])
            e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_FULL)
            e.sampler_play(vc["sampler"])
        outs = [e.process_blocks(3)]
        for j, vc in enumerate(allv[::2]):
            e.sampler_pause(vc["sampler"], at_block=1 + j)
            e.sampler_play(vc["sampler"], at_block=9 + j)
        for j, vc in enumerate(voices[1::3]):
            e.set_param(vc["volume"], 0, float(rng.uniform(5, 100)), at_block=2 * j)
        outs.append(e.process_blocks(24))
        for vc in voices[:5]:
            e.sampler_stop(vc["sampler"])
        outs.append(e.process_blocks(5))
        return np.concatenate(outs)

    mbf = 128
    ro = build(oracle(max_block_frames=mbf))
    g = GpuEngine(max_block_frames=mbf, max_batch=max_batch)
    rg = build(g)
    assert g.cx.plan_kind() == 3 and g.cx.plan_fused_voices() == 14, (g.cx.plan_kind(), g.cx.plan_fused_voices())
    assert np.std(ro) > 0.01
    assert_bits_equal(ro, rg, shape)
    g2 = GpuEngine(max_block_frames=mbf, max_batch=max_batch, force_generic=True)
    assert_bits_equal(ro, build(g2), shape + " (level executor alone)")


@pytest.mark.parametrize("n_voices,max_batch", [(600, 32), (300, 32)])
def test_level_executor_streams_frozen_nodes_several_blocks_at_a_time(n_voices, max_batch):
    """round 4, k_level's frozen fast path (k_generic.hip.h frozen_fast): on a WIDE level — thousands of (node, block) pairs — a wave
    takes 2 or 8 consecutive blocks of its node and, for a frozen stereo volume / pan / width / hard clip or a steadily playing
    planar-f32 sampler, reads everything but the audio once and streams the blocks four at a time.  The small scenarios of this file
    never reach that width: here 600 (8 blocks per wave) / 300 (2 per wave: partial groups) voices of sampler -> volume -> pan -> width
    -> clip with width automation, mutes in front of the width (all-silent blocks: cleared + flagged), late starts, one-shot ends and
    mono sources run on the levels alone, bit for bit the oracle."""
    def run(e):
        return scenarios.scenario_voice_fx_events(e, n_voices, radix=32, src_frames=700)

    ro = run(oracle(max_block_frames=64))
    g = GpuEngine(max_block_frames=64, max_batch=max_batch, force_generic=True)
    rg = run(g)
    assert g.cx.plan_kind() == 0
    assert np.std(ro) > 0.01
    assert_bits_equal(ro, rg, "voice fx bank on the levels alone")


@pytest.mark.parametrize("n_leaves,K", [(32, 768), (16, 400)])
def test_spatialiser_waves_keep_their_histories_in_lds_across_consecutive_blocks(n_leaves, K):
    """round 4, k_leaf_sum_sp: on a launch with thousands of (leaf, block) pairs a wave takes up to 8 consecutive 256-frame blocks of its
    leaf and keeps every port's last 64 mono frames in LDS — only the first block of its run re-fetches the history (leaf_sp_fast,
    hmode 3).  The small scenarios never get there (one block per wave below 6 144 pairs): here config 2's shape — 32 x 32 voices of
    sampler -> volume -> pan -> spatialiser, 768 blocks per call: 8 blocks per wave — and a half-size one (2 per wave), with a pause
    and a position glide and a gain change INSIDE the long call (blocks that leave the batched path and come back: the kept tails
    must be dropped and rebuilt), every block of every call against the oracle."""
    rng = np.random.default_rng(5)

    def run(e):
        voices, ends = [], []
        for v in range(n_leaves * 32):
            s = e.sampler(float(rng.uniform(60, 100)))
            vol = e.volume(float(rng.uniform(30, 100)))
            pan = e.pan(float(rng.uniform(-1, 1)))
            sp = e.spatial(float(rng.uniform(-5, 5)), float(rng.uniform(-1, 1)), float(rng.uniform(-5, 5)), n_in=2)
            e.connect_stereo(s, vol)
            e.connect_stereo(vol, pan)
            e.connect_stereo(pan, sp)
            voices.append(dict(s=s, vol=vol, sp=sp))
            ends.append(sp)
        leaves = []
        for i in range(0, len(ends), 32):
            m = e.sum(32)
            for p, n in enumerate(ends[i:i + 32]):
                e.connect_stereo(n, m, 2 * p)
            leaves.append(m)
        root = e.sum(len(leaves))
        for p, m in enumerate(leaves):
            e.connect_stereo(m, root, 2 * p)
        e.connect_stereo(root, e.graph_out_node)
        e.update()
        smp = [e.new_sample(PLANAR_F32, 2, scenarios.voice_source(6100 + j, 30000 + 256 * j, 2)) for j in range(8)]
        for v, vc in enumerate(voices):
            e.sampler_set_sample(vc["s"], smp[v % 8])
            e.sampler_set_loop_range(vc["s"], fwapi.LOOP_FULL)
            e.sampler_play(vc["s"])
        outs = [e.process_blocks(3)]
        outs.append(e.process_blocks(K))                                   # steady: every wave on the batched path, tails kept
        e.sampler_pause(voices[5]["s"], at_block=K // 4)                   # leaf 0 leaves the batched path for a while ...
        e.sampler_play(voices[5]["s"], at_block=K // 4 + 37)               # ... and comes back
        e.set_param(voices[40]["sp"], 0, 4.0, at_block=K // 8)             # leaf 1: a position glide (ear delays switch, gains ramp)
        e.set_param(voices[n_leaves * 32 - 7]["vol"], 0, 15.0, at_block=K - K // 5)   # the last leaf: a gain glide late in the call
        outs.append(e.process_blocks(K))
        outs.append(e.process_blocks(5))
        return np.concatenate(outs)

    rng = np.random.default_rng(5)
    ro = run(oracle(max_block_frames=256))
    rng = np.random.default_rng(5)
    g = GpuEngine(max_block_frames=256, max_batch=K)
    rg = run(g)
    assert g.cx.plan_kind() == 1
    assert np.std(ro) > 0.01
    assert_bits_equal(ro, rg, "spatialiser bank, %d blocks per call" % K)


def test_plans_switch_between_fused_hybrid_and_levels_mid_stream():
    """a send is patched into a running voice bank and pulled out again, then a spatialised copy of the root bus is added (graph
    edits + recompile): voice-bank plan -> hybrid -> voice-bank plan -> hybrid; playheads, gliding smoothers and the
    steady-call cache carry over (processor.rs:195-197: processors persist across schedules)"""
    kinds = []

    def run2(e):
        rng = np.random.default_rng(3)
        ends, voices = [], []
        for v in range(24):
            s = e.sampler(100.0)
            vol = e.volume(float(rng.uniform(20, 100)))
            e.connect_stereo(s, vol)
            ends.append(vol)
            voices.append((s, vol))
        leaves = []
        for i in range(0, 24, 6):
            m = e.sum(6)
            for p, n in enumerate(ends[i:i + 6]):
                e.connect_stereo(n, m, 2 * p)
            leaves.append(m)
        root = e.sum(4)
        for p, m in enumerate(leaves):
            e.connect_stereo(m, root, 2 * p)
        mix = e.sum(2)
        e.connect_stereo(root, mix, 0)
        e.connect_stereo(mix, e.graph_out_node)
        e.update()
        for v, (s, vol) in enumerate(voices):
            e.sampler_set_sample(s, e.new_sample(PLANAR_F32, 2, scenarios.voice_source(4400 + v, 1500 + 7 * v, 2)))
            e.sampler_set_loop_range(s, fwapi.LOOP_FULL)
            e.sampler_play(s)
        outs = [e.process_blocks(5)]
        k = [e.cx.plan_kind()] if hasattr(e, "cx") else []
        e.set_param(voices[3][1], 0, 8.0, at_block=1)              # a glide in flight across the edit
        ret = e.volume(40.0)                                        # the send: leaf 1 -> ret -> mix port 1
        e.connect_stereo(leaves[1], ret)
        e.connect_stereo(ret, mix, 2)
        e.update()
        outs.append(e.process_blocks(6))
        k += [e.cx.plan_kind()] if hasattr(e, "cx") else []
        e.set_param(voices[9][1], 0, 90.0, at_block=2)
        e.remove_node(ret)
        e.update()
        outs.append(e.process_blocks(6))
        k += [e.cx.plan_kind()] if hasattr(e, "cx") else []
        sp = e.spatial(1.0, 0.0, -2.0, n_in=2)                      # a spatialised copy of the root bus beside the dry one
        e.connect_stereo(root, sp)
        e.connect_stereo(sp, mix, 2)
        e.update()
        outs.append(e.process_blocks(6))
        k += [e.cx.plan_kind()] if hasattr(e, "cx") else []
        kinds.append(k)
        return np.concatenate(outs)

    o = oracle(max_block_frames=128)
    g = GpuEngine(max_block_frames=128, max_batch=4)
    ro = run2(o)
    rg = run2(g)
    assert kinds[-1] == [1, 3, 1, 3], kinds   # voice-bank plan -> hybrid -> voice-bank plan -> hybrid
    assert_bits_equal(ro, rg, "across plan switches")


# ------------------------------------------------------------------ error behaviour of the SPEC nodes (activate() -> Err)
def test_spec_node_activation_and_argument_errors():
    from fwapi import FIR, RESAMPLER, SPATIAL, BIQUAD, DELAY

    g = GpuEngine(max_block_frames=64)
    smp = g.new_sample(PLANAR_F32, 1, scenarios.voice_source(1, 100, 1))
    # constructor-time argument errors: a sample id that does not exist
    for kind, n_in, n_out in ((FIR, 2, 2), (RESAMPLER, 0, 2)):
        with pytest.raises(Exception):
            g.add_node(kind, n_in, n_out, [99.0])
    # activation errors surface from update() as CompileGraphError::NodeActivationFailed, and the graph stays usable
    for kind, n_in, n_out, params in ((SPATIAL, 3, 2, [0, 0, -1]), (SPATIAL, 1, 1, [0, 0, -1]), (RESAMPLER, 1, 2, [float(smp), 1.0]),
                                      (BIQUAD, 2, 3, [0, 1000, 0.7]), (DELAY, 0, 0, [0.1])):
        bad = g.add_node(kind, n_in, n_out, params)
        with pytest.raises(fwapi.CompileGraphError) as ei:
            g.update()
        assert ei.value.name == "NodeActivationFailed"
        g.remove_node(bad)
    s = g.sampler(100.0)
    g.connect_stereo(s, g.graph_out_node)
    g.update()
    g.sampler_set_sample(s, smp)
    g.sampler_play(s)
    out = g.process_blocks(2)
    assert np.any(out != 0)
    # runtime parameter ids are checked
    rs = g.resampler(smp, 1.0, n_out=1)
    g.update()
    with pytest.raises(Exception):
        g.set_param(rs, 7, 1.0)


@pytest.mark.parametrize("mbf", [100, 96, 1, 36])
def test_odd_block_sizes_bit_exact(mbf):
    # block sizes that are not a multiple of 4 / 64: the voice-bank plan stays on its per-element path, the chain
    # shape falls back to the generic executor; results must not change
    frames = max(300, 7 * mbf)
    o = oracle(max_block_frames=mbf)
    g = GpuEngine(max_block_frames=mbf, max_batch=5)
    a = scenarios.scenario_voice_bank_steady(o, 9, 6, radix=4, src_frames=frames, mono_every=4)
    b = scenarios.scenario_voice_bank_steady(g, 9, 6, radix=4, src_frames=frames, mono_every=4)
    assert g.cx.plan_kind() == 1
    assert_bits_equal(a, b, "voice bank, block %d" % mbf)
    o = oracle(max_block_frames=mbf)
    g = GpuEngine(max_block_frames=mbf, max_batch=5)
    kw = dict(radix=4, src_frames=frames, first_delay_frames=64, min_delay_frames=64, max_delay_frames=200)
    a = scenarios.scenario_chain_steady(o, 6, 6, **kw)
    b = scenarios.scenario_chain_steady(g, 6, 6, **kw)
    assert g.cx.plan_kind() == 0        # not a multiple of 64: generic executor
    assert_bits_equal(a, b, "chain shape, block %d" % mbf)


def test_chain_plan_ring_hazard_stress():
    """k_chain keeps prefetched ring loads in flight across workgroup barriers and relies on one CU's L1 ordering its
    waves' accesses: the risky delays are the shortest ones (a slot is re-read one or two tiles after it was
    stored).  Many voices (HBM busy), many blocks, delays pinned to 64, 65, 127, 128, 129, 255, 256, 257, 383, 384:
    every sample of the whole run must equal the oracle's."""
    delays = [64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 512, 1000]

    def run(e, n_voices, blocks, mbf):
        rng = np.random.default_rng(99)
        ends, samplers = [], []
        for v in range(n_voices):
            s = e.sampler(100.0)
            bq = e.biquad(0, float(rng.uniform(300, 6000)), 0.9)
            dl = e.delay(delays[v % len(delays)] / float(e.sample_rate), feedback=0.6, mix=0.5)
            vol = e.volume(float(rng.uniform(30, 100)))
            e.connect_stereo(s, bq)
            e.connect_stereo(bq, dl)
            e.connect_stereo(dl, vol)
            samplers.append(s)
            ends.append(vol)
        level = ends
        while len(level) > 1:
            nxt = []
            for i in range(0, len(level), 32):
                grp = level[i:i + 32]
                m = e.sum(len(grp))
                for p, n in enumerate(grp):
                    e.connect_stereo(n, m, 2 * p)
                nxt.append(m)
            level = nxt
        e.connect_stereo(level[0], e.graph_out_node)
        e.update()
        for v, s in enumerate(samplers):
            e.sampler_set_sample(s, e.new_sample(PLANAR_F32, 2, scenarios.voice_source(3000 + v, 4096)))
            e.sampler_set_loop_range(s, fwapi.LOOP_FULL)
            e.sampler_play(s)
        return np.concatenate([e.process_blocks(blocks // 2), e.process_blocks(blocks - blocks // 2)])

    for mbf, n_voices, blocks in ((256, 1024, 48), (128, 384, 80), (64, 96, 120)):
        o = oracle(max_block_frames=mbf)
        g = GpuEngine(max_block_frames=mbf, max_batch=32)
        a = run(o, n_voices, blocks, mbf)
        b = run(g, n_voices, blocks, mbf)
        assert g.cx.plan_kind() == 2
        assert_bits_equal(a, b, "ring hazard stress, block %d" % mbf)


def _realtime_callbacks(e, chain, n_voices, calls):
    """one callback per call, the way cpal drives it (cpal/lib.rs:378-449): steady callbacks, then a burst of
    messages, then steady callbacks again."""
    build = scenarios.build_chain_bank if chain else scenarios.build_voice_bank
    voices = build(e, n_voices, radix=8, src_frames=1100)
    for vc in voices:
        e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_FULL)
        e.sampler_play(vc["sampler"])
    out = []
    for i in range(calls):
        if i == calls // 2:
            for v, vc in enumerate(voices):
                if v % 3 == 0:
                    e.set_param(vc["volume"], 0, 33.0)
                if v % 5 == 1:
                    e.sampler_pause(vc["sampler"])
        out.append(e.process_blocks(2 if i % 4 == 3 else 1))
    return np.concatenate(out)


@pytest.mark.parametrize("chain", [False, True])
@pytest.mark.parametrize("graph", ["1", "0"])
def test_realtime_callbacks_pinned_io_and_graph_replay_bit_exact(chain, graph, monkeypatch):
    # small calls take the realtime edge: pinned device-mapped I/O blocks, and (graph=1) the steady launch sequence
    # replayed from a hipGraph; message bursts drop back to plain launches and the graph is picked up again afterwards
    monkeypatch.setenv("FWGPU_RT_GRAPH", graph)
    o = oracle(max_block_frames=128)
    g = GpuEngine(max_block_frames=128)
    out_o = _realtime_callbacks(o, chain, 21, 14)
    out_g = _realtime_callbacks(g, chain, 21, 14)
    assert g.cx.plan_kind() == (2 if chain else 1)
    assert_bits_equal(out_o, out_g, "realtime callbacks chain=%s graph=%s" % (chain, graph))


def test_virtual_shards_on_one_device_ordered_bus_equals_whole_graph():
    # SURVEY 8(e) "measurability": G shards as G contexts (own streams) on the one visible device; the rank-ordered
    # sum of their partial buses is the reference's G-port SumNode, bit for bit (whole graph run by the oracle)
    import torch

    import test_multirank_gloo as mr
    from firewheel_amd import shard

    G = 4
    want = mr.reference_whole_graph(G)
    parts = []
    engines = []
    for r in range(G):
        lo, hi = shard.voice_range(r, G, mr.TOTAL_VOICES)
        e = GpuEngine(max_block_frames=mr.BLOCK)
        root, voices = mr.build_shard(e, lo, hi)
        e.connect_stereo(root, e.graph_out_node)
        e.update()
        mr.start_voices(e, voices)
        assert e.cx.plan_kind() == 1
        buf = torch.empty(mr.BLOCKS * mr.BLOCK * 2, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        e.cx.process_blocks_device(mr.BLOCKS, buf.data_ptr(), 2)
        parts.append(buf)
        engines.append(e)
    for e in engines:
        e.cx.synchronize()
    bus = shard.ordered_sum(parts, torch.empty_like(parts[0]))
    torch.cuda.synchronize()
    assert_bits_equal(want, bus.cpu().numpy(), "virtual shards, ordered bus")


def _c_host_reference(e, voices, block, callbacks):
    """examples/host_c/fw_host.c restated on the test engine API: same graph, samples, messages, callback sizes"""
    SRC, RADIX = 1900, 8
    sampler, volume, level = [], [], []
    for v in range(voices):
        s = e.sampler(100.0)
        g = e.volume(float(10 + (v * 37) % 90))
        p = e.pan(float(np.float32((v * 53) % 200 - 100) / np.float32(100.0)))
        e.connect_stereo(s, g)
        e.connect_stereo(g, p)
        sampler.append(s)
        volume.append(g)
        level.append(p)
    while True:
        nxt = []
        for i in range(0, len(level), RADIX):
            grp = level[i:i + RADIX]
            m = e.sum(len(grp))
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    e.connect_stereo(level[0], e.graph_out_node)
    e.update()
    # xorshift32, vectorised over the voices
    st = (np.uint32(0x9E3779B9) ^ ((np.arange(voices, dtype=np.uint64) * 2654435761 + 1) & 0xFFFFFFFF).astype(np.uint32))
    data = np.empty((voices, 2 * SRC), dtype=np.float32)
    for i in range(2 * SRC):
        st ^= st << np.uint32(13)
        st ^= st >> np.uint32(17)
        st ^= st << np.uint32(5)
        data[:, i] = (st >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
    for v in range(voices):
        smp = e.new_sample(PLANAR_F32, 2, data[v].reshape(2, SRC))
        e.sampler_set_sample(sampler[v], smp)
        if v % 3 != 2:
            e.sampler_set_loop_range(sampler[v], fwapi.LOOP_FULL)
        e.sampler_play(sampler[v])
    out = []
    for cb in range(callbacks):
        if cb == callbacks // 3:
            for v in range(0, voices, 2):
                e.set_param(volume[v], 0, 20.0 + float(v % 7) * 10.0)
        if cb == callbacks // 2:
            for v in range(1, voices, 5):
                e.sampler_pause(sampler[v])
        out.append(e.process_blocks(2 if cb % 4 == 3 else 1))
    return np.concatenate(out)


@pytest.mark.parametrize("voices,block,callbacks", [(50, 128, 40), (9, 96, 25)])
def test_plain_c_host_through_the_c_abi_matches_oracle(voices, block, callbacks, tmp_path):
    # the drop-in boundary used from compiled C (no Python between the host and libfwgpu): examples/host_c
    import subprocess

    exe = os.path.join(fwapi.ROOT, "examples", "host_c", "fw_host")
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    path = str(tmp_path / "out.f32")
    r = subprocess.run([exe, str(voices), str(block), str(callbacks), path], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "plan kind 1" in r.stdout
    got = np.fromfile(path, dtype=np.float32)
    want = _c_host_reference(oracle(max_block_frames=block), voices, block, callbacks)
    assert_bits_equal(want, got, "plain C host %dx%d" % (voices, block))


@pytest.mark.parametrize("name", ["events_70", "events_33_r2", "chain_events_37", "chain_calls_37_b256_wrap",
                                  "chain_calls_33_b64", "steady_96x32"])
@pytest.mark.parametrize("max_batch", [64, 2])
def test_async_device_calls_back_to_back_bit_exact(name, max_batch):
    # the throughput form of the call (fwgpu_process_blocks_device: output left in HBM) back to back with no host sync
    # in between, message bursts included: the stream must come out exactly as through the synchronous host-buffer form
    class AsyncEngine(GpuEngine):
        async_device = True

    saved = fwapi.GpuEngine
    try:
        globals()["GpuEngine"] = AsyncEngine
        out_o, out_g, g = run_case(name, max_batch=max_batch)
    finally:
        globals()["GpuEngine"] = saved
    assert g.cx.plan_kind() in (1, 2)
    assert_bits_equal(out_o, np.asarray(out_g), name + " async K<=%d" % max_batch)


@pytest.mark.parametrize("name,plan", [("master_chain_bank", 1), ("master_chain_fx", 2)])
@pytest.mark.parametrize("max_batch", [64, 5, 1])
def test_master_chain_after_the_root_keeps_the_fused_plans(name, plan, max_batch):
    # voice bank -> sum tree -> master biquad / delay / volume / limiter -> graph_out: the fused plans run the tree, the
    # generic node kernel runs the master chain on the mix bus (K-batched); automation on master nodes included
    out_o, out_g, g = run_case(name, max_batch=max_batch)
    assert g.cx.plan_kind() == plan
    assert_bits_equal(out_o, out_g, name + " K<=%d" % max_batch)
    gold = json.load(open(GOLDEN))
    assert digest(out_g) == gold[name]
    out_o2, out_g2, g2 = run_case(name, force_generic=True)
    assert g2.cx.plan_kind() == 0
    assert_bits_equal(out_o2, out_g2, name + " generic")


@pytest.mark.parametrize("n_voices,mbf", [(19, 64), (32, 64), (31, 36), (40, 64)])
def test_wide_leaves_with_short_blocks(n_voices, mbf):
    # a leaf SumNode with more ports than the block has frame quads (block 64: 16 lanes own frames, the other lanes only
    # lend their port's descriptor through v_readlane) and loops that wrap inside blocks (the per-voice path).  Found by
    # the fuzz test: the lanes without frames had never computed the values that were read from them.
    def run(e):
        v = scenarios.build_voice_bank(e, n_voices, radix=32, src_frames=300, mono_every=5)
        for vc in v:
            e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_FULL)
            e.sampler_play(vc["sampler"])
        return np.asarray(e.process_blocks(9))

    g = GpuEngine(max_block_frames=mbf)
    assert_bits_equal(run(oracle(max_block_frames=mbf)), run(g), "wide leaf, short block")
    assert g.cx.plan_kind() == 1


def test_chain_plan_steady_loop_soak_equals_generic_executor():
    # 1024 voices x 167 blocks of 512 frames through k_chain's steady-call loop: short delay lines (3 tiles .. 3 tiles +
    # 700 frames: hundreds of laps, a straddling quad per lap and voice), loops that wrap inside blocks, 64-block calls.
    # Reference: the generic executor (separate kernels, serial biquads and delays) — bit for bit; the oracle checks the head.
    def run(e, calls):
        v = scenarios.build_chain_bank(e, 1024, radix=32, src_frames=1500, mono_every=9, first_delay_frames=384,
                                       min_delay_frames=385, max_delay_frames=1085)
        for vc in v:
            e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_FULL)
            e.sampler_play(vc["sampler"])
        return np.concatenate([np.asarray(e.process_blocks(k)) for k in calls])

    calls = (2, 64, 64, 37)
    gf = GpuEngine(max_block_frames=512, max_batch=64)
    of = run(gf, calls)
    assert gf.cx.plan_kind() == 2
    steady, general = gf.cx.plan_chain_stats()
    assert steady >= 3 * 64, (steady, general)  # 32 leaves x 2 channels, at least the three long calls
    og = run(GpuEngine(max_block_frames=512, force_generic=True, max_batch=64), calls)
    assert_bits_equal(og, of, "steady-loop soak vs generic executor")
    assert np.all(np.isfinite(of)) and np.std(of) > 0.1
    oo = run(oracle(max_block_frames=512), (2,))
    assert_bits_equal(oo, of[:oo.size], "steady-loop soak head vs oracle")


@pytest.mark.parametrize("chain", [False, True])
def test_unconnected_sum_ports_stay_on_the_fused_plans_and_take_voices_later(chain):
    # voice slots nothing is plugged into (an unconnected stereo port of a leaf SumNode; also one on an upper SumNode):
    # the reference feeds them the cleared, silent-flagged buffer (schedule.rs:310-313).  The fused plans model them as
    # null voices / bus 0; plugging a voice in later recompiles, keeps the plan and every node's state.
    def run(e):
        rng = np.random.default_rng(3)
        ends, smp = [], []
        for v in range(21):
            s = e.sampler(90.0)
            cur = s
            if chain:
                b = e.biquad(0, 2000.0 + 100 * v, 0.8)
                d = e.delay((200 + 13 * v) / 48000.0, feedback=0.3, mix=0.4)
                e.connect_stereo(cur, b)
                e.connect_stereo(b, d)
                cur = d
            g = e.volume(float(rng.uniform(30, 90)))
            e.connect_stereo(cur, g)
            ends.append(g)
            smp.append(s)
        leaves = []
        for i in range(0, 21, 7):
            m = e.sum(9)                      # 7 voices + 2 empty slots (ports 3 and 8)
            slots = [0, 1, 2, 4, 5, 6, 7]
            for p, n in zip(slots, ends[i:i + 7]):
                e.connect_stereo(n, m, 2 * p)
            leaves.append(m)
        root = e.sum(4)                       # 3 leaves + 1 empty upper port
        for p, m in enumerate(leaves):
            e.connect_stereo(m, root, 2 * (p if p < 2 else 3))
        e.connect_stereo(root, e.graph_out_node)
        e.update()
        for v, s in enumerate(smp):
            e.sampler_set_sample(s, e.new_sample(PLANAR_F32, 2, scenarios.voice_source(400 + v, 700)))
            e.sampler_set_loop_range(s, fwapi.LOOP_FULL)
            e.sampler_play(s)
        out = [np.asarray(e.process_blocks(5))]
        plan0 = e.cx.plan_kind() if hasattr(e, "cx") else None
        # plug a new dry voice into an empty slot of the first leaf
        s = e.sampler(70.0)
        e.connect_stereo(s, leaves[0], 2 * 3)
        e.update()
        e.sampler_set_sample(s, e.new_sample(PLANAR_F32, 2, scenarios.voice_source(999, 500)))
        e.sampler_set_loop_range(s, fwapi.LOOP_FULL)
        e.sampler_play(s)
        out.append(np.asarray(e.process_blocks(6)))
        return np.concatenate(out), plan0

    want, _ = run(oracle(max_block_frames=128))
    g = GpuEngine(max_block_frames=128)
    got, plan0 = run(g)
    assert plan0 == (2 if chain else 1) and g.cx.plan_kind() == plan0
    assert_bits_equal(want, got, "unconnected ports, chain=%s" % chain)


@pytest.mark.parametrize("mode", ["bank", "chain", "generic"])
def test_sample_destroy_after_the_samplers_moved_on_and_while_still_held(mode):
    # fwgpu_sample_destroy (the last Arc<dyn SampleResource> dropped, core/sample_resource.rs): (1) after every sampler
    # switched to another sample the destroy changes nothing — bit-exact vs the oracle, which keeps the sample;
    # (2) destroyed while samplers still hold it (a caller error the ABI must survive): empty sample, silence, no fault;
    # (3) refused while a FIR node names the sample.
    chain = mode == "chain"

    def build(e):
        ends, smp = [], []
        for v in range(12):
            s = e.sampler(80.0)
            cur = s
            if chain:
                b = e.biquad(0, 1500.0 + 90 * v, 0.8)
                e.connect_stereo(cur, b)
                cur = b
            g = e.volume(40.0 + 3 * v)
            e.connect_stereo(cur, g)
            ends.append(g)
            smp.append(s)
        m = e.sum(12)
        for p, n in enumerate(ends):
            e.connect_stereo(n, m, 2 * p)
        e.connect_stereo(m, e.graph_out_node)
        e.update()
        return smp

    def run(e, destroy):
        smp = build(e)
        a = [e.new_sample(PLANAR_F32, 2, scenarios.voice_source(100 + v, 900)) for v in range(12)]
        b = [e.new_sample(PLANAR_F32, 2, scenarios.voice_source(300 + v, 640)) for v in range(12)]
        for v, s in enumerate(smp):
            e.sampler_set_sample(s, a[v])
            e.sampler_set_loop_range(s, fwapi.LOOP_FULL)
            e.sampler_play(s)
        out = [np.asarray(e.process_blocks(4))]
        for v, s in enumerate(smp):
            e.sampler_set_sample(s, b[v], at_block=1)
            e.sampler_play(s, at_block=1)
        out.append(np.asarray(e.process_blocks(3)))
        if destroy:
            for i in a:
                e.cx.destroy_sample(i)
            with pytest.raises(Exception):
                e.cx.destroy_sample(a[0])  # already gone
        out.append(np.asarray(e.process_blocks(5)))
        return np.concatenate(out), smp, b

    og, _, _ = run(oracle(max_block_frames=128), False)
    g = GpuEngine(max_block_frames=128, force_generic=(mode == "generic"), max_batch=4)
    gg, smp, b = run(g, True)
    assert g.cx.plan_kind() == {"bank": 1, "chain": 2, "generic": 0}[mode]
    assert_bits_equal(og, gg, "destroy of samples nobody plays any more (%s)" % mode)
    assert np.any(gg[-5 * 128 * 2:] != 0)
    # (2) pull the samples out from under the playing samplers
    for i in b:
        g.cx.destroy_sample(i)
    tail = np.asarray(g.process_blocks(6))
    assert np.all(np.isfinite(tail))
    if chain:  # the biquads ring out
        assert np.all(np.abs(tail[-2 * 128 * 2:]) < 1e-6)
    else:
        assert np.all(tail[-2 * 128 * 2:] == 0), "a destroyed sample plays as no sample"
    # the context stays usable: a fresh sample plays again
    n = g.new_sample(PLANAR_F32, 2, scenarios.voice_source(7, 500))
    g.sampler_set_sample(smp[0], n)
    g.sampler_play(smp[0])
    assert np.any(np.asarray(g.process_blocks(2)) != 0)
    # (3) FIR nodes keep their impulse response: refused while the node exists, allowed once it is gone
    ir = g.new_sample(PLANAR_F32, 1, scenarios.voice_source(9, 64, 1))
    f = g.fir(ir)
    with pytest.raises(Exception):
        g.cx.destroy_sample(ir)
    g.remove_node(f)
    g.cx.destroy_sample(ir)


def test_messages_queued_for_a_removed_node_do_not_reach_the_node_that_reuses_its_slot():
    def run(e, stale):
        a = e.sampler(100.0)
        keep = e.sampler(60.0)
        m = e.sum(2)
        e.connect_stereo(a, m, 0)
        e.connect_stereo(keep, m, 2)
        e.connect_stereo(m, e.graph_out_node)
        e.update()
        smp = e.new_sample(PLANAR_F32, 2, scenarios.voice_source(5, 4000))
        for s in (a, keep):
            e.sampler_set_sample(s, smp)
            e.sampler_set_loop_range(s, fwapi.LOOP_FULL)
            e.sampler_play(s)
        out = [np.asarray(e.process_blocks(2))]
        if stale:  # for a block this node will not live to see
            e.set_param(a, 0, 5.0, at_block=3)
            e.sampler_pause(a, at_block=4)
        e.remove_node(a)
        b = e.sampler(90.0)  # takes a's slot
        e.connect_stereo(b, m, 0)
        e.update()
        e.sampler_set_sample(b, smp)
        e.sampler_play(b)
        out.append(np.asarray(e.process_blocks(8)))
        return np.concatenate(out)

    ref = run(oracle(max_block_frames=64), False)
    for generic in (False, True):
        clean = run(GpuEngine(max_block_frames=64, force_generic=generic), False)
        stale = run(GpuEngine(max_block_frames=64, force_generic=generic), True)
        assert_bits_equal(ref, clean, "slot reuse (generic=%s)" % generic)
        assert_bits_equal(clean, stale, "stale messages of a removed node (generic=%s)" % generic)
