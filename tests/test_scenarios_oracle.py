"""CPU tier: run every parity scenario on the oracle alone and pin the results to committed golden
digests (tests/golden/oracle_digests.json, made by tests/golden/make_golden.py from the oracle).  The
reference cannot run in the build image, so these digests pin the ORACLE against drift; they are not
reference outputs."""
import hashlib
import json
import os

import numpy as np
import pytest

import fwapi
import scenarios

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_digests.json")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def oracle(**kw):
    return scenarios.TaggedOracle(fwapi.OracleEngine(**kw))


CASES = {
    "steady_96x32": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=256), 96, 6),
    "steady_40x4_i16": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=64), 40, 5, radix=4,
                                                                     fmt=fwapi.INTERLEAVED_I16),
    "steady_9x3_u16": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=128), 9, 4, radix=3,
                                                                    fmt=fwapi.PLANAR_U16),
    # every sample format through the compact fast path of the leaf kernel (and its fallbacks)
    "steady_fmt_p_i16_mono3": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=128), 20, 9, radix=8,
                                                                            fmt=fwapi.PLANAR_I16, mono_every=3, src_frames=1000),
    "steady_fmt_i_f32": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=64), 11, 7, radix=4,
                                                                      fmt=fwapi.INTERLEAVED_F32, mono_every=4, src_frames=700),
    "steady_fmt_i_u16": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=64), 9, 7, radix=16,
                                                                      fmt=fwapi.INTERLEAVED_U16, mono_every=2, src_frames=600),
    "steady_fmt_p_i16_oddlen": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=64), 7, 20, radix=8,
                                                                             fmt=fwapi.PLANAR_I16, src_frames=333),
    "steady_fmt_mixed_leaf": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=128), 26, 8, radix=32,
                                                                           fmt_cycle=list(range(6)), mono_every=5, src_frames=900),
    # reference kinds only: what rust/firewheel-gpu/tests/reference_digests.rs can replay on the real firewheel-graph
    "ref_steady_64": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=256), 64, 6, with_pan=False),
    "ref_steady_33_i16_r8": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=64), 33, 9, radix=8, with_pan=False,
                                                                          fmt=fwapi.INTERLEAVED_I16, mono_every=5, src_frames=777),
    "ref_desk_30": lambda: scenarios.scenario_ref_desk(oracle(max_block_frames=128)),
    "ref_desk_21_b64": lambda: scenarios.scenario_ref_desk(oracle(max_block_frames=64), 21, radix=4, src_frames=500, seed=9),
    "events_33_i16": lambda: scenarios.scenario_voice_bank_events(oracle(max_block_frames=128), 33, radix=8, src_frames=777,
                                                                   fmt=fwapi.INTERLEAVED_I16),
    "events_70": lambda: scenarios.scenario_voice_bank_events(oracle(max_block_frames=256), 70),
    "events_33_r2": lambda: scenarios.scenario_voice_bank_events(oracle(max_block_frames=128), 33, radix=2, src_frames=777),
    # 300 and 10 000 messages inside one call (the control kernel's wave-wide message search: one round / 64-ary rounds)
    "spatial_steady_b128": lambda: scenarios.scenario_spatial_steady(oracle(max_block_frames=128)),
    "spatial_steady_b64": lambda: scenarios.scenario_spatial_steady(oracle(max_block_frames=64), 5, src_frames=1500, calls=(3, 90, 33, 7)),
    # voice banks with sends, a return chain and a spatialised source around them: the hybrid plan
    "hybrid_sends_b128": lambda: scenarios.scenario_hybrid_sends(oracle(max_block_frames=128)),
    "hybrid_sends_b64": lambda: scenarios.scenario_hybrid_sends(oracle(max_block_frames=64), 17, 9, src_frames=900, seed=8, long_call=61),
    "hybrid_chain_sends_b128": lambda: scenarios.scenario_hybrid_chain_sends(oracle(max_block_frames=128)),
    "hybrid_chain_sends_b64": lambda: scenarios.scenario_hybrid_chain_sends(oracle(max_block_frames=64), 21, radix=7, src_frames=800, seed=12,
                                                                             long_call=70),
    "split_mixers_b128": lambda: scenarios.scenario_split_mixers(oracle(max_block_frames=128)),
    "split_mixers_b64": lambda: scenarios.scenario_split_mixers(oracle(max_block_frames=64), seed=22, long_call=45, src_frames=700),
    "bus_iir_b256": lambda: scenarios.scenario_bus_iir(oracle(max_block_frames=256)),
    "bus_iir_b512": lambda: scenarios.scenario_bus_iir(oracle(max_block_frames=512), seed=32, src_frames=9000),
    "bus_iir_b128": lambda: scenarios.scenario_bus_iir(oracle(max_block_frames=128), seed=33),
    "storm_48x6": lambda: scenarios.scenario_message_storm(oracle(max_block_frames=128)),
    "storm_200x50_b64": lambda: scenarios.scenario_message_storm(oracle(max_block_frames=64), 200, radix=32, blocks=60, per_voice=50,
                                                                 src_frames=3000, seed=4),
    # width / hard-clip stages at the end of the voice chains (the voice-bank plan's stage programs)
    "voice_fx_steady": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=256), 70, 6, src_frames=1500,
                                                                     voice_fx=scenarios.width_clip_fx),
    "voice_fx_events_45": lambda: scenarios.scenario_voice_fx_events(oracle(max_block_frames=128)),
    "voice_fx_events_20_i16_r32": lambda: scenarios.scenario_voice_fx_events(oracle(max_block_frames=64), 20, radix=32, src_frames=500,
                                                                              with_pan=False, fmt=fwapi.INTERLEAVED_I16),
    # voices whose source is the SPEC resampler, on the voice-bank plan
    "rs_bank_40": lambda: scenarios.scenario_rs_bank(oracle(max_block_frames=128)),
    "rs_bank_21_b64_i16_pure": lambda: scenarios.scenario_rs_bank(oracle(max_block_frames=64), 21, radix=32, src_frames=400, mixed=False,
                                                                  fmt=fwapi.INTERLEAVED_I16),
    "mixed_generic": lambda: scenarios.scenario_mixed_generic(oracle(max_block_frames=256)),
    "mixed_generic_nobeep": lambda: scenarios.scenario_mixed_generic(oracle(max_block_frames=256), use_beep=False),
    "cfg3_chain": lambda: scenarios.scenario_cfg3_chain(oracle(max_block_frames=128)),
    "cfg4_reverb": lambda: scenarios.scenario_cfg4_reverb(oracle(max_block_frames=128)),
    "cfg4_reverb_2irs_mono": lambda: scenarios.scenario_cfg4_reverb(oracle(max_block_frames=64), n_voices=5, taps=700,
                                                                     shared_ir=False, ir_channels=1),
    "graph_inputs": lambda: scenarios.scenario_graph_inputs(oracle(max_block_frames=64, num_graph_inputs=3)),
    "spatial_scene": lambda: scenarios.scenario_spatial_scene(oracle(max_block_frames=128)),
    "spatial_scene_b96": lambda: scenarios.scenario_spatial_scene(oracle(max_block_frames=96), n_sources=4, blocks=9),
    "chain_steady_40": lambda: scenarios.scenario_chain_steady(oracle(max_block_frames=256), 40, 6),
    "chain_steady_bq_only_i16": lambda: scenarios.scenario_chain_steady(oracle(max_block_frames=64), 21, 9, radix=4, delay=False,
                                                                          fmt=fwapi.INTERLEAVED_I16),
    "chain_steady_dl_only_pan": lambda: scenarios.scenario_chain_steady(oracle(max_block_frames=128), 10, 7, radix=3,
                                                                          biquad=False, with_pan=True),
    "chain_events_37": lambda: scenarios.scenario_chain_events(oracle(max_block_frames=128), 37),
    "chain_steady_40_d128": lambda: scenarios.scenario_chain_steady(oracle(max_block_frames=256), 40, 6, first_delay_frames=128,
                                                                      min_delay_frames=129),
    "chain_events_37_d130": lambda: scenarios.scenario_chain_events(oracle(max_block_frames=128), 37, first_delay_frames=130,
                                                                      min_delay_frames=128),
    "chain_events_21_d256": lambda: scenarios.scenario_chain_events(oracle(max_block_frames=256), 21, first_delay_frames=256,
                                                                      min_delay_frames=257, src_frames=1500),
    "chain_events_19_r2_pan": lambda: scenarios.scenario_chain_events(oracle(max_block_frames=64), 19, radix=2, src_frames=777,
                                                                        with_pan=True),
    "master_chain_bank": lambda: scenarios.scenario_master_chain(oracle(max_block_frames=128)),
    "master_chain_fx": lambda: scenarios.scenario_master_chain(oracle(max_block_frames=128), chain=True, n_voices=37),
    "chain_calls_37_b256": lambda: scenarios.scenario_chain_steady_calls(oracle(max_block_frames=256), 37, tile=128),
    "chain_calls_20_b128_pan": lambda: scenarios.scenario_chain_steady_calls(oracle(max_block_frames=128), 20, tile=128,
                                                                             with_pan=True),
    "chain_calls_33_b64": lambda: scenarios.scenario_chain_steady_calls(oracle(max_block_frames=64), 33, tile=64),
    "chain_calls_37_b256_wrap": lambda: scenarios.scenario_chain_steady_calls(oracle(max_block_frames=256), 37, tile=128,
                                                                              src_extra=77),
    "chain_calls_21_b128_wrap": lambda: scenarios.scenario_chain_steady_calls(oracle(max_block_frames=128), 21, tile=128,
                                                                              src_extra=130),
    "chain_calls_33_b64_wrap": lambda: scenarios.scenario_chain_steady_calls(oracle(max_block_frames=64), 33, tile=64,
                                                                             src_extra=5),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden_digest(name):
    out = CASES[name]()
    assert np.all(np.isfinite(out))
    assert np.any(out != 0)
    gold = json.load(open(GOLDEN))
    assert digest(out) == gold[name], "oracle output drifted from the committed digest"


def test_steady_bank_is_loop_periodic():
    # size-independent property: constant gains + full loops => output repeats with the loop period
    e = oracle(max_block_frames=256)
    out = scenarios.scenario_voice_bank_steady(e, 40, 16, src_frames=1024)
    frames = out.reshape(-1, 2)
    assert np.array_equal(frames[:1024], frames[1024:2048])
    assert np.array_equal(frames[:1024], frames[3072:4096])


# ------------------------------------------------------------------ SPEC nodes: the f32 spec vs an f64 evaluation
def test_biquad_spec_matches_f64_lfilter():
    from scipy.signal import lfilter

    L = fwapi.oracle_lib()
    for ftype, fc, q in [(0, 1000.0, 0.707), (1, 300.0, 1.5), (2, 5000.0, 4.0), (0, 15000.0, 0.5)]:
        e = fwapi.OracleEngine(max_block_frames=256)
        n = e.biquad(ftype, fc, q, ch=1)
        e.update()
        x = fwapi.xorshift_uniform(31 + ftype, 2048)
        y = np.concatenate([e.node_process(n, 256, [x[i:i + 256]], 1)[0][0] for i in range(0, 2048, 256)])
        # RBJ cookbook in f64
        w0 = 2 * np.pi * fc / 48000.0
        cw, al = np.cos(w0), np.sin(w0) / (2 * q)
        b = {0: [(1 - cw) / 2, 1 - cw, (1 - cw) / 2], 1: [(1 + cw) / 2, -(1 + cw), (1 + cw) / 2], 2: [al, 0, -al]}[ftype]
        a = [1 + al, -2 * cw, 1 - al]
        ref = lfilter(np.array(b) / a[0], np.array(a) / a[0], x.astype(np.float64))
        # f32 DF1 vs f64: error grows with the pole radius; 2e-5 relative to the signal peak covers Q = 4
        assert np.max(np.abs(y - ref)) <= 2e-5 * max(1.0, np.max(np.abs(ref))), (ftype, fc, q)


def test_delay_and_width_spec_against_numpy():
    e = fwapi.OracleEngine(max_block_frames=64)
    d = e.delay(10.0 / 48000.0, feedback=0.5, mix=0.25, ch=1)   # D = 10 frames < block: in-block recurrence
    w = e.width(0.0)
    e.update()
    x = fwapi.xorshift_uniform(5, 192)
    y = np.concatenate([e.node_process(d, 64, [x[i:i + 64]], 1)[0][0] for i in range(0, 192, 64)])
    ring = np.zeros(10, np.float32)
    exp = np.empty(192, np.float32)
    f32 = np.float32
    for i in range(192):
        dd = ring[i % 10]
        ring[i % 10] = f32(x[i] + f32(dd * f32(0.5)))
        exp[i] = f32(f32(x[i] * f32(0.75)) + f32(dd * f32(0.25)))
    assert np.array_equal(y, exp)
    st = fwapi.xorshift_uniform(6, 128).reshape(2, 64)
    yw, _ = e.node_process(w, 64, st, 2)
    mono = ((st[0] + st[1]).astype(f32) * f32(0.5)).astype(f32)
    assert np.array_equal(yw[0], mono) and np.array_equal(yw[1], mono)    # width 0 = mono


def test_fir_spec_matches_f64_convolution():
    # H7: the f32 segment-ordered fmaf chain vs an f64 convolution, error relative to sum|h x|
    taps, frames, blocks = 9000, 128, 4
    e = fwapi.OracleEngine(max_block_frames=frames)
    h = scenarios.reverb_ir(9, taps, 1)
    ir = e.new_sample(fwapi.PLANAR_F32, 1, h)
    n = e.fir(ir, ch=1)
    e.update()
    x = fwapi.xorshift_uniform(10, frames * blocks)
    y = np.concatenate([e.node_process(n, frames, [x[i:i + frames]], 1)[0][0] for i in range(0, x.size, frames)])
    ref = np.convolve(x.astype(np.float64), h[0].astype(np.float64))[:x.size]
    bound = np.convolve(np.abs(x).astype(np.float64), np.abs(h[0]).astype(np.float64))[:x.size]
    assert np.max(np.abs(y - ref) / np.maximum(bound, 1e-30)) < 16 * 2.0 ** -24


def rs_table_f64():
    """independent numpy evaluation of the SPEC filter bank: Kaiser(beta 8)-windowed sinc, cutoff 0.9"""
    P, T, fc, beta = 32, 16, 0.9, 8.0
    h = np.zeros((P, T))
    for ph in range(P):
        t = (np.arange(T) - (T // 2 - 1)) - ph / P
        r = t / (T / 2)
        w = np.where(np.abs(r) < 1, np.i0(beta * np.sqrt(np.clip(1 - r * r, 0, None))) / np.i0(beta), 0.0)
        row = fc * np.sinc(fc * t) * w
        h[ph] = row / row.sum()
    return h


def rs_reference(x, ratio, n_out, loop=False):
    """f64 evaluation at the SPEC's 32.32 fixed-point positions"""
    h = rs_table_f64()
    step = int(round(float(np.float32(ratio)) * 2 ** 32))   # the node takes its ratio as an f32
    y = np.zeros(n_out)
    L = x.size
    for i in range(n_out):
        p = i * step
        if loop:
            p %= L << 32
        idx, ph = p >> 32, (p >> 27) & 31
        j = idx - 7 + np.arange(16)
        xs = x[j % L] if loop else np.where((j >= 0) & (j < L), x[np.clip(j, 0, L - 1)], 0.0)
        y[i] = np.dot(h[ph], xs.astype(np.float64))
    return y


def test_resampler_spec_reproduces_a_sine_at_the_converted_rate():
    # 44.1 kHz source played into the 48 kHz stream (ratio 44100/48000): a 1 kHz sine stays a 1 kHz sine
    e = fwapi.OracleEngine(max_block_frames=256)
    n_src = 6000
    t = np.arange(n_src) / 44100.0
    x = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    rs = e.resampler(e.new_sample(fwapi.PLANAR_F32, 1, x[None, :]), 44100.0 / 48000.0, n_out=1)
    e.update()
    y = np.concatenate([e.node_process(rs, 256, [], 1)[0][0] for _ in range(16)])
    n = np.arange(y.size)
    ref = np.sin(2 * np.pi * 1000.0 * n / 48000.0)
    # positions are quantised to 1/32 source sample (RS_PHASES): phase error <= 2*pi*1000/44100/32 = 4.5e-3
    assert np.max(np.abs(y[64:] - ref[64:])) < 8e-3
    # against the independent f64 evaluation of the same table at the same positions: f32 rounding only
    assert np.max(np.abs(y - rs_reference(x, 44100.0 / 48000.0, y.size))) < 2e-6


def test_resampler_fixed_point_positions_loop_and_end():
    x = fwapi.xorshift_uniform(3, 500)
    for ratio, loop in ((1.0, False), (0.37, False), (2.25, True), (1.0 / 3.0, True)):
        e = fwapi.OracleEngine(max_block_frames=64)
        rs = e.resampler(e.new_sample(fwapi.PLANAR_F32, 1, x[None, :]), ratio, loop=loop, n_out=1)
        e.update()
        y = np.concatenate([e.node_process(rs, 64, [], 1)[0][0] for _ in range(12)])
        ref = rs_reference(x, ratio, y.size, loop)
        if not loop:   # stops at the first block boundary after the window has left the sample
            end = next((b * 64 for b in range(1, 13) if (b * 64 * int(round(float(np.float32(ratio)) * 2 ** 32)) >> 32) >= 500 + 8), y.size)
            assert not np.any(y[end:])
            ref[end:] = 0
        assert np.max(np.abs(y - ref)) < 2e-6, (ratio, loop)
    # looping with ratio 0.5 over 256 frames is periodic with 512 output frames: exact 32.32 arithmetic, no drift
    e2 = fwapi.OracleEngine(max_block_frames=64)
    rs2 = e2.resampler(e2.new_sample(fwapi.PLANAR_F32, 1, x[None, :256]), 0.5, loop=True, n_out=1)
    e2.update()
    z = np.concatenate([e2.node_process(rs2, 64, [], 1)[0][0] for _ in range(24)])
    assert np.array_equal(z[:512], z[512:1024])


def test_spatial_spec_against_numpy():
    f32 = np.float32
    e = fwapi.OracleEngine(max_block_frames=64)
    sp = e.spatial(2.0, 0.0, -2.0, n_in=1)        # 45 degrees to the right, distance 2.83
    e.update()
    x = fwapi.xorshift_uniform(11, 192)
    y = np.concatenate([np.stack(e.node_process(sp, 64, [x[i:i + 64]], 2)[0], axis=1) for i in range(0, 192, 64)])
    d = np.sqrt(8.0)
    s = 2.0 / d
    th = (s + 1) * np.pi / 4
    gl, gr = f32(np.cos(th) / d), f32(np.sin(th) / d)
    dl = int(round(s * round(0.00066 * 48000)))
    xl = np.concatenate([np.zeros(dl, f32), x])[:192]
    assert dl == 23
    assert np.array_equal(y[:, 0], (xl * gl).astype(f32))       # left ear: later and quieter
    assert np.array_equal(y[:, 1], (x * gr).astype(f32))        # right ear: no delay
    assert gr > gl
