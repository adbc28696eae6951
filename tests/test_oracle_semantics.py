"""Hand-derived known-answer tests for the oracle (SURVEY.md Appendix A, Q1-Q22 + table A.1).
The reference ships no numeric tests, so each expectation below is derived from the cited Rust
lines with independent numpy float32 arithmetic."""
import ctypes as C

import numpy as np
import pytest

import fwapi
from fwapi import (BEEP_TEST, DUMMY, HARD_CLIP, MONO_TO_STEREO, PLANAR_F32, PLANAR_I16, PLANAR_U16, INTERLEAVED_F32,
                   INTERLEAVED_I16, INTERLEAVED_U16, SAMPLER, STEREO_TO_MONO, SUM, VOLUME, LOOP_FULL, LOOP_RANGE_SECS,
                   LOOP_NONE, OracleEngine)

f32 = np.float32
L = fwapi.oracle_lib()
INACTIVE, ACTIVE, DEACTIVATING = 0, 1, 2


def coeffs(sr=48000):
    # smoother.rs:99-100 with smooth_secs = 10.0/1000.0 (:21)
    # glibc expf (what Rust's f32::exp calls on linux-gnu) is correctly rounded here; numpy's SIMD
    # float32 exp is not (1 ulp off), so round the f64 result instead.
    import math
    x = f32(-1.0) / (f32(f32(10.0) / f32(1000.0)) * f32(sr))
    b = f32(math.exp(float(x)))
    return f32(1.0) - b, b


def ramp(inp, last, n, sr=48000):
    a, b = coeffs(sr)
    out = np.empty(n, f32)
    x = f32(inp) * a
    prev = f32(last)
    for i in range(n):
        prev = f32(x + f32(prev * b))
        out[i] = prev
    return out


class Smoother:
    def __init__(self, val, sr=48000, mbf=256):
        self.p = L.fwo_smoother_new(val, sr, mbf)
        self.mbf = mbf

    def set(self, v):
        L.fwo_smoother_set(self.p, v)

    def reset(self, v):
        L.fwo_smoother_reset(self.p, v)

    def process(self, frames):
        out = np.zeros(self.mbf, f32)
        st = C.c_int()
        n = L.fwo_smoother_process(self.p, frames, fwapi._fptr(out), self.mbf, C.byref(st))
        return out[:n], st.value

    def state(self):
        v = [C.c_float() for _ in range(4)]
        st = C.c_int()
        L.fwo_smoother_state(self.p, *[C.byref(x) for x in v], C.byref(st))
        return dict(input=v[0].value, last_output=v[1].value, a=v[2].value, b=v[3].value, status=st.value)


def test_smoother_coefficients():
    s = Smoother(0.5)
    a, b = coeffs()
    st = s.state()
    assert f32(st["a"]) == a and f32(st["b"]) == b
    # glibc expf(-1/480) in f32; pin the constant of record
    assert abs(float(b) - np.exp(-1.0 / 480.0)) < 1e-7


def test_smoother_table_A1():
    s = Smoother(1.0, mbf=256)
    # Inactive: process returns the FULL buffer (Q4), constant
    v, st = s.process(64)
    assert st == INACTIVE and len(v) == 256 and np.all(v == f32(1.0))
    # set(same) is a no-op; set(new) activates
    s.set(1.0)
    assert s.state()["status"] == INACTIVE
    s.set(0.0)
    assert s.state()["status"] == ACTIVE
    v, st = s.process(100)
    exp = ramp(0.0, 1.0, 100)
    assert st == ACTIVE and len(v) == 100 and np.array_equal(v, exp)
    assert f32(s.state()["last_output"]) == exp[-1]
    # continue until it settles; Q1: the test is on output[0]
    last = exp[-1]
    for _ in range(100):
        y = ramp(0.0, last, 256)
        v, st = s.process(256)
        if abs(f32(0.0) - y[0]) < f32(0.00001):
            # Q2: the ramp is discarded, the whole buffer is the target; status Deactivating
            assert st == DEACTIVATING and np.all(v == f32(0.0)) and len(v) == 256
            assert f32(s.state()["last_output"]) == f32(0.0)
            break
        assert st == ACTIVE and np.array_equal(v, y)
        last = y[-1]
    else:
        raise AssertionError("never settled")
    # Q3: Deactivating never becomes Inactive through process()
    for _ in range(3):
        v, st = s.process(256)
        assert st == DEACTIVATING and len(v) == 256 and np.all(v == f32(0.0))
    # reset() from Deactivating -> Inactive
    s.reset(0.0)
    assert s.state()["status"] == INACTIVE
    # reset() while Inactive with a new value refills
    s.reset(0.75)
    v, st = s.process(16)
    assert st == INACTIVE and np.all(v == f32(0.75))
    # frames == 0 while Active returns the full buffer and leaves state alone
    s.set(0.1)
    v, st = s.process(0)
    assert st == ACTIVE and len(v) == 256


def test_smoother_can_stall_above_epsilon_and_stay_active_forever():
    # Q28 (found while pinning the oracle): the unfused f32 recurrence reaches a fixed point
    # |input - y| = 1.07e-5 > settle_epsilon for 1.0 -> 0.25, so the smoother never leaves Active.
    s = Smoother(1.0, mbf=256)
    s.set(0.25)
    last = f32(1.0)
    for _ in range(60):
        y = ramp(0.25, last, 256)
        v, st = s.process(256)
        assert st == ACTIVE and np.array_equal(v, y)
        last = y[-1]
    assert y[0] == y[-1] and abs(float(y[0]) - 0.25) > 1e-5


def test_smoother_partial_then_inactive_never_exposes_stale():
    s = Smoother(0.0, mbf=64)
    s.set(1.0)
    v, st = s.process(10)
    assert len(v) == 10 and st == ACTIVE
    s.reset(1.0)  # Active -> Inactive fills everything
    v, st = s.process(64)
    assert np.all(v == f32(1.0)) and st == INACTIVE


def test_silence_mask():
    assert L.fwo_mask_new_all_silent(0) == 0
    assert L.fwo_mask_new_all_silent(2) == 3
    assert L.fwo_mask_new_all_silent(63) == (1 << 63) - 1
    assert L.fwo_mask_new_all_silent(64) == (1 << 64) - 1
    assert L.fwo_mask_all_silent(0b11, 2) == 1 and L.fwo_mask_all_silent(0b01, 2) == 0
    assert L.fwo_mask_all_silent(0, 0) == 1  # empty mask == empty mask
    assert L.fwo_mask_any_silent(0b10, 2) == 1 and L.fwo_mask_any_silent(0b100, 2) == 0
    assert L.fwo_mask_all_silent((1 << 64) - 1, 64) == 1 and L.fwo_mask_any_silent(1 << 63, 64) == 1


def test_scalar_helpers():
    assert f32(L.fwo_percent_volume_to_raw_gain(50.0)) == f32(f32(50.0) * f32(0.01)) ** 2  # Q22 square
    assert L.fwo_percent_volume_to_raw_gain(-3.0) == 0.0
    assert f32(L.fwo_percent_volume_to_raw_gain(200.0)) == f32(f32(200.0) * f32(0.01)) ** 2  # no upper clamp
    assert L.fwo_db_to_gain_clamped(-100.0) == 0.0 and L.fwo_db_to_gain_clamped(-120.0) == 0.0
    assert abs(L.fwo_db_to_gain(-6.0) - 10 ** (-0.3)) < 1e-7
    assert L.fwo_gain_to_db_clamped(0.00001) == -100.0
    assert f32(L.fwo_pcm_i16_to_f32(32767)) == f32(1.0)
    assert f32(L.fwo_pcm_i16_to_f32(-32768)) == f32(f32(-32768.0) * f32(f32(1.0) / f32(32767.0)))
    assert f32(L.fwo_pcm_u16_to_f32(0)) == f32(-1.0)
    assert f32(L.fwo_pcm_u16_to_f32(65535)) == f32(f32(f32(65535.0) * f32(f32(2.0) / f32(65535.0))) - f32(1.0))


def test_interleave_helpers():
    l = np.arange(8, dtype=f32) + 1
    r = -(np.arange(8, dtype=f32) + 1)
    out = np.full(16, np.nan, f32)
    L.fwo_interleave_stereo(fwapi._fptr(l), fwapi._fptr(r), fwapi._fptr(out), 16, 1, 0b01)
    assert np.array_equal(out[0::2], l) and np.array_equal(out[1::2], r)  # Q20: copies both unless BOTH silent
    L.fwo_interleave_stereo(fwapi._fptr(l), fwapi._fptr(r), fwapi._fptr(out), 16, 1, 0b11)
    assert np.all(out == 0)
    # generic interleave zero-fills then skips silent channels (Q20)
    chans = [l, r, l * 2]
    out = np.full(24, np.nan, f32)
    L.fwo_interleave(fwapi._ptr_array(chans), 3, 8, fwapi._fptr(out), 24, 3, 1, 0b010)
    assert np.array_equal(out[0::3], l) and np.all(out[1::3] == 0) and np.array_equal(out[2::3], l * 2)
    # deinterleave scans the DESTINATION's old contents for the mask (Q11); extra channels zeroed + flagged
    src = np.arange(16, dtype=f32)
    d0 = np.zeros(8, f32)
    d1 = np.ones(8, f32)
    d2 = np.ones(8, f32)
    m = L.fwo_deinterleave(fwapi._ptr_array([d0, d1, d2]), 3, 8, fwapi._fptr(src), 16, 2, 1)
    assert np.array_equal(d0, src[0::2]) and np.array_equal(d1, src[1::2]) and np.all(d2 == 0)
    assert m == 0b101  # ch0 old contents were zero -> "silent" although it now holds data; ch2 extra


# --------------------------------------------------------------------------------- node-level (B1) checks
def mk(kind, n_in, n_out, params=(), mbf=64):
    e = OracleEngine(max_block_frames=mbf)
    n = e.add_node(kind, n_in, n_out, params)
    e.update()
    return e, n


def test_volume_paths():
    e, n = mk(VOLUME, 2, 2, [50.0])
    g = f32(L.fwo_percent_volume_to_raw_gain(50.0))
    x = fwapi.xorshift_uniform(1, 128).reshape(2, 64)
    y, om = e.node_process(n, 64, x, 2, in_mask=0)
    assert np.array_equal(y, x * g) and om == 0
    # Q15: stereo fast path multiplies even a silent channel, mask passthrough
    y, om = e.node_process(n, 64, x, 2, in_mask=0b01)
    assert np.array_equal(y, x * g) and om == 0b01
    # all silent -> clear + all-silent mask
    y, om = e.node_process(n, 64, x, 2, in_mask=0b11)
    assert np.all(y == 0) and om == 0b11
    # gain change -> smoothing from the block it is seen
    e.set_param(n, 0, 100.0)
    y, om = e.node_process(n, 64, x, 2)
    r = ramp(1.0, g, 64)
    assert np.array_equal(y, x * r)
    # generic path (3 ch) zero-fills silent channels
    e3, n3 = mk(VOLUME, 3, 3, [100.0])
    x3 = fwapi.xorshift_uniform(2, 192).reshape(3, 64)
    y, om = e3.node_process(n3, 64, x3, 3, in_mask=0b010)
    assert np.array_equal(y[0], x3[0]) and np.all(y[1] == 0) and np.array_equal(y[2], x3[2]) and om == 0b010
    # muted fast path only while Inactive
    em, nm = mk(VOLUME, 2, 2, [0.0])
    y, om = em.node_process(nm, 64, x, 2)
    assert np.all(y == 0) and om == 0b11


def test_volume_q3_mute_never_fires_after_settle():
    # Q3: after a change settles the smoother stays Deactivating, so the mute shortcut is dead
    e, n = mk(VOLUME, 2, 2, [100.0])
    x = np.ones((2, 64), f32)
    e.set_param(n, 0, 0.0)
    for _ in range(400):
        y, om = e.node_process(n, 64, x, 2)
    assert np.all(y == 0.0) and om == 0  # zeros by multiplication, but NOT flagged silent


def test_volume_activation_error():
    e = OracleEngine()
    e.add_node(VOLUME, 2, 1, [100.0])
    with pytest.raises(fwapi.CompileGraphError) as ei:
        e.update()
    assert ei.value.name == "NodeActivationFailed"


def test_sum_paths():
    x = fwapi.xorshift_uniform(3, 64 * 12).reshape(12, 64)
    # copy path (Q14)
    e, n = mk(SUM, 2, 2)
    y, om = e.node_process(n, 64, x[:2], 2, in_mask=0b10)
    assert np.array_equal(y, x[:2]) and om == 0b10
    # 2/3/4-port: left-assoc, mask ignored, out mask 0 (Q13)
    for ports in (2, 3, 4):
        e, n = mk(SUM, 2 * ports, 2)
        y, om = e.node_process(n, 64, x[:2 * ports], 2, in_mask=0b0100)
        for c in range(2):
            acc = x[c].copy()
            for p in range(1, ports):
                acc = (acc + x[2 * p + c]).astype(f32)
            assert np.array_equal(y[c], acc)
        assert om == 0
    # n-port: port 0 copied even if silent; silent ports >= 1 skipped; sequential order
    e, n = mk(SUM, 12, 2)
    mask = 0b000011001101  # ch0 (port0 L) silent, ch2,ch3 (port1) silent, ch6,7 (port3) silent
    y, om = e.node_process(n, 64, x, 2, in_mask=mask)
    for c in range(2):
        acc = x[c].copy()
        for p in range(1, 6):
            if mask >> (2 * p + c) & 1:
                continue
            acc = (acc + x[2 * p + c]).astype(f32)
        assert np.array_equal(y[c], acc)
    assert om == 0
    # sign of zero: skipped -0.0 port keeps -0.0
    z = np.full((10, 8), -0.0, f32)
    e, n = mk(SUM, 10, 2, mbf=8)
    y, om = e.node_process(n, 8, z, 2, in_mask=0b1111111100)
    assert np.all(np.signbit(y))
    y, om = e.node_process(n, 8, z, 2, in_mask=(1 << 10) - 1)
    assert not np.any(np.signbit(y)) and om == 0b11
    # 1 mono out, 5 ports
    e, n = mk(SUM, 5, 1)
    y, om = e.node_process(n, 64, x[:5], 1)
    acc = x[0].copy()
    for p in range(1, 5):
        acc = (acc + x[p]).astype(f32)
    assert np.array_equal(y[0], acc)


def test_hard_clip_paths():
    t = f32(L.fwo_db_to_gain_clamped(-6.0))
    e, n = mk(HARD_CLIP, 2, 2, [-6.0])
    x = fwapi.xorshift_uniform(4, 128).reshape(2, 64)
    y, om = e.node_process(n, 64, x, 2, out_mask=0)
    assert np.array_equal(y, np.clip(x, -t, t)) and om == 0
    # Q16: a silent channel -> generic path: zero-fill + mask passthrough
    y, om = e.node_process(n, 64, x, 2, in_mask=0b10)
    assert np.array_equal(y[0], np.clip(x[0], -t, t)) and np.all(y[1] == 0) and om == 0b10


def test_mono_stereo():
    e, n = mk(MONO_TO_STEREO, 1, 2)
    x = fwapi.xorshift_uniform(5, 64).reshape(1, 64)
    y, om = e.node_process(n, 64, x, 2)
    assert np.array_equal(y[0], x[0]) and np.array_equal(y[1], x[0]) and om == 0
    y, om = e.node_process(n, 64, x, 2, in_mask=1)
    assert np.all(y == 0) and om == 0b11
    e, n = mk(STEREO_TO_MONO, 2, 1)
    x = fwapi.xorshift_uniform(6, 128).reshape(2, 64)
    y, om = e.node_process(n, 64, x, 1)
    assert np.array_equal(y[0], ((x[0] + x[1]).astype(f32) * f32(0.5)).astype(f32)) and om == 0
    y, om = e.node_process(n, 64, x, 1, in_mask=0b11)
    assert np.all(y == 0) and om == 0b1


def test_beep_test():
    e, n = mk(BEEP_TEST, 0, 2, [440.0, -12.0, 1.0])
    y, om = e.node_process(n, 64, [], 2)
    inc = f32(440.0) / f32(48000.0)
    g = f32(L.fwo_db_to_gain_clamped(-12.0))
    ph = f32(0)
    exp = np.empty(64, f32)
    for i in range(64):
        exp[i] = f32(np.sin(f32(ph * f32(2 * np.pi)), dtype=f32)) * g
        t = f32(ph + inc)
        ph = f32(t - np.trunc(t))
    assert np.allclose(y[0], exp, rtol=0, atol=2e-7) and np.array_equal(y[0], y[1]) and om == 0
    # Q12: disabled -> channel 0 left untouched, others cleared, mask = all_silent(n-1)
    e.set_param(n, 0, 0.0)
    init = np.full((2, 64), 7.0, f32)
    y, om = e.node_process(n, 64, [], 2, out_init=init)
    assert np.all(y[0] == 7.0) and np.all(y[1] == 0) and om == 0b1


def test_dummy_does_not_write():
    e, n = mk(DUMMY, 1, 1)
    init = np.full((1, 64), 3.0, f32)
    y, om = e.node_process(n, 64, [np.zeros(64, f32)], 1, out_init=init)
    assert np.all(y == 3.0) and om == 0


# --------------------------------------------------------------------------------- sampler
def sampler_engine(data, fmt=PLANAR_F32, channels=2, n_out=2, mbf=64, percent=100.0):
    e = OracleEngine(max_block_frames=mbf)
    s = e.sampler(percent, n_out)
    for c in range(min(n_out, 2)):
        e.connect(s, c, e.graph_out_node, c)
    e.update()
    smp = e.new_sample(fmt, channels, data)
    e.sampler_set_sample(s, smp)
    return e, s


def test_sampler_oneshot_and_end():
    data = fwapi.xorshift_uniform(7, 2 * 150).reshape(2, 150)
    e, s = sampler_engine(data)
    assert np.all(e.process_interleaved(64) == 0)  # not playing
    e.sampler_play(s)
    out = np.concatenate([e.process_interleaved(64) for _ in range(4)])
    exp = np.zeros((2, 256), f32)
    exp[:, :150] = data
    assert np.array_equal(out[0::2], exp[0]) and np.array_equal(out[1::2], exp[1])
    # Q9: after the tail block playing=false, playhead=0 -> silence until Play, then from 0 again
    e.sampler_play(s)
    out = e.process_interleaved(64)
    assert np.array_equal(out[0::2], data[0, :64])


def test_sampler_loop_and_messages():
    data = fwapi.xorshift_uniform(8, 2 * 200).reshape(2, 200)
    e, s = sampler_engine(data)
    e.sampler_set_loop_range(s, LOOP_FULL)
    e.sampler_play(s)
    out = np.concatenate([e.process_interleaved(64) for _ in range(8)])
    idx = np.arange(512) % 200
    assert np.array_equal(out[0::2], data[0, idx]) and np.array_equal(out[1::2], data[1, idx])
    # Stop returns to loop start; SetPlayheadSecs rounds to nearest frame
    e.sampler_stop(s)
    assert np.all(e.process_interleaved(64) == 0)
    e.sampler_set_playhead_secs(s, 10.4 / 48000.0)
    e.sampler_play(s)
    out = e.process_interleaved(64)
    assert np.array_equal(out[0::2], data[0, 10:74])
    # Q7: SetLoopRange with the playhead INSIDE the new range snaps to range start
    e.sampler_set_loop_range(s, LOOP_RANGE_SECS, 70.0 / 48000.0, 170.0 / 48000.0)  # playhead is 74
    out = e.process_interleaved(64)
    assert np.array_equal(out[0::2], data[0, 70:134])
    # playhead outside the new range is left alone and plays through to the range end, then wraps
    e.sampler_set_loop_range(s, LOOP_RANGE_SECS, 150.0 / 48000.0, 190.0 / 48000.0)  # playhead 134 < 150
    out = e.process_interleaved(64)
    exp = np.concatenate([data[0, 134:190], data[0, 150:158]])
    assert np.array_equal(out[0::2], exp)


def test_sampler_gain_and_mute_and_pause():
    data = np.ones((2, 4096), f32)
    e, s = sampler_engine(data, percent=50.0)
    g = f32(L.fwo_percent_volume_to_raw_gain(50.0))
    e.sampler_play(s)
    out = e.process_interleaved(64)
    assert np.all(out == g)
    # Q6: smoother only advances on playing blocks
    e.set_param(s, 0, 100.0)
    e.sampler_pause(s)
    assert np.all(e.process_interleaved(64) == 0)
    e.sampler_play(s)
    out = e.process_interleaved(64)
    assert np.array_equal(out[0::2], ramp(1.0, g, 64))
    # muted sampler (Inactive smoother, gain < 1e-5) clears
    e2, s2 = sampler_engine(data, percent=0.0)
    e2.sampler_play(s2)
    assert np.all(e2.process_interleaved(64) == 0)


@pytest.mark.parametrize("fmt", [INTERLEAVED_I16, INTERLEAVED_U16, INTERLEAVED_F32, PLANAR_I16, PLANAR_U16, PLANAR_F32])
@pytest.mark.parametrize("channels,n_out", [(1, 1), (1, 2), (2, 2), (2, 1), (3, 2), (2, 4)])
def test_sampler_formats(fmt, channels, n_out):
    rng = np.random.default_rng(fmt * 10 + channels)
    frames = 100
    if fmt in (INTERLEAVED_I16, PLANAR_I16):
        raw = rng.integers(-32768, 32768, size=(channels, frames)).astype(np.int16)
        conv = (raw.astype(f32) * f32(f32(1.0) / f32(32767.0))).astype(f32)
    elif fmt in (INTERLEAVED_U16, PLANAR_U16):
        raw = rng.integers(0, 65536, size=(channels, frames)).astype(np.uint16)
        conv = ((raw.astype(f32) * f32(f32(2.0) / f32(65535.0))).astype(f32) - f32(1.0)).astype(f32)
    else:
        raw = rng.random((channels, frames), dtype=f32) * 2 - 1
        conv = raw
    data = raw.T.copy() if fmt <= INTERLEAVED_F32 else raw
    e = OracleEngine(max_block_frames=64)
    s = e.sampler(100.0, n_out)
    e.update()
    smp = e.new_sample(fmt, channels, data)
    e.sampler_set_sample(s, smp)
    e.sampler_play(s)
    init = np.full((n_out, 64), 9.0, f32)
    y, om = e.node_process(s, 64, [], n_out, out_init=init)
    exp = np.full((n_out, 64), 9.0, f32)
    exp_mask = 0
    if fmt <= INTERLEAVED_F32 and channels == 2 and n_out < 2:
        # interleaved stereo fast path needs >= 2 buffers, else generic path fills min(ch, bufs)
        pass
    for c in range(min(channels, n_out)):
        exp[c] = conv[c, :64]
    if n_out > channels:
        if n_out == 2 and channels == 1:
            exp[1] = exp[0]
        else:
            for c in range(channels, n_out):
                exp[c] = 0
                exp_mask |= 1 << c
    assert np.array_equal(y, exp) and om == exp_mask


def test_graph_end_to_end_small():
    # 3 voices: sampler -> volume -> sum(3) -> out; checks routing + masks + interleave through process_interleaved
    e = OracleEngine(max_block_frames=64)
    data = [fwapi.xorshift_uniform(20 + v, 2 * 640).reshape(2, 640) for v in range(3)]
    m = e.sum(3)
    vs = []
    for v in range(3):
        s = e.sampler(100.0)
        vol = e.volume(50.0 + 10 * v)
        e.connect_stereo(s, vol)
        e.connect_stereo(vol, m, 2 * v)
        vs.append((s, vol))
    e.connect_stereo(m, e.graph_out_node)
    e.update()
    for v, (s, vol) in enumerate(vs):
        e.sampler_set_sample(s, e.new_sample(PLANAR_F32, 2, data[v]))
        e.sampler_set_loop_range(s, LOOP_FULL)
        if v != 1:
            e.sampler_play(s)
    out = e.process_interleaved(640)  # 10 sub-blocks (processor.rs:95-96)
    g = [f32(L.fwo_percent_volume_to_raw_gain(50.0 + 10 * v)) for v in range(3)]
    for c in range(2):
        # voice 1 is paused -> silent; 3-port path ignores masks and adds its zeros
        acc = (data[0][c] * g[0]).astype(f32)
        acc = (acc + np.zeros(640, f32)).astype(f32)
        acc = (acc + (data[2][c] * g[2]).astype(f32)).astype(f32)
        assert np.array_equal(out[c::2], acc)
    # frames == 0 -> zero output (Q19)
    assert e.process_interleaved(0).size == 0
