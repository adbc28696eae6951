"""A SECOND, independent CPU restatement of the hot path's graph level, in numpy float32 (test infrastructure).

Why it exists: the Rust reference cannot be built here, so the oracle (oracle/fw_oracle.cpp) is pinned by source reading
only, and the SPEC nodes' control math is the same text in product and oracle.  This model was
written straight from the reference's .rs files and from the SPEC table of DESIGN.md §6 — it shares NO code with
fw_oracle.cpp (no ctypes, no C++; its own graph walk, its own smoother, its own sample fetch, its own fused-multiply-add)
— and `tests/test_refmodel_differential.py` fuzzes the two against each other, bit for bit, over the GPU fuzz families'
own generators.  tests/golden/refmodel_digests.json is generated from THIS model (make_golden_refmodel.py) and checked
against the oracle and, on the GPU tier, against the HIP path.

Restated (file:line = BillyDM/firewheel @ 2024-10-16):
  core/param/smoother.rs:93-205        ParamSmoother (new / reset / set / process / set_and_process)
  core/param/range.rs:32-35            percent_volume_to_raw_gain
  core/util.rs:7-27,44-147,165-175     dB helpers, (de)interleave, interleave_stereo, clear_all_outputs
  core/sample_resource.rs:28-456       the six sample formats, fill_buffers_* and the PCM conversions
  basic_nodes/volume.rs:84-145, sum.rs:41-136, sampler.rs:241-277,323-561, hard_clip.rs:51-95,
  mono_to_stereo.rs:33-50, stereo_to_mono.rs:33-56, dummy.rs
  graph/processor.rs:61-165,214-248    process_interleaved, process_block
  graph/graph/compiler/schedule.rs:213-344   prepare_graph_inputs / process / read_graph_outputs (silence flags)
SPEC nodes (not in the reference): StereoPan, StereoWidth, Biquad, Delay, FIR reverb, Resampler, Spatial.
BeepTest asks the platform libm for sinf, as it does for powf (see platform_powf).  Not modelled: the reference's buffer REUSE (every output port owns a buffer here: results differ only
where the reference exposes stale data, Q12 / a19, which is outside the parity domain).
"""
import math

import numpy as np

import fwapi
from fwapi import (BEEP_TEST, BIQUAD, DELAY, DUMMY, HARD_CLIP, MONO_TO_STEREO, RESAMPLER, SAMPLER, SPATIAL, STEREO_PAN, STEREO_TO_MONO, STEREO_WIDTH, SUM,
                   VOLUME)

f32 = np.float32
F0 = f32(0.0)


def fma32(a, b, c):
    """correctly rounded f32 fused multiply-add on numpy arrays (Rust: f32::mul_add).  a*b is exact in f64; the f64 sum
    is taken with its rounding error (TwoSum) and forced to ROUND-TO-ODD, after which the final rounding to f32 is the
    single rounding of the exact value (53 >= 2*24 + 2)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    p = a * b
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    bits = s.view(np.int64) if s.ndim else np.array(s).view(np.int64)
    inexact = (err != 0.0) & ((bits & 1) == 0) & np.isfinite(s)
    toward = np.where(err > 0.0, np.inf, -np.inf)
    s = np.where(inexact, np.nextafter(s, toward), s)
    return s.astype(f32)


# ------------------------------------------------------------------------------------------ core/param/smoother.rs
INACTIVE, ACTIVE, DEACTIVATING = 0, 1, 2


class ParamSmoother(object):
    def __init__(self, val, sample_rate, max_block_frames):  # :93-112, config defaults :18-25
        smooth_secs = f32(10.0) / f32(1000.0)
        x = f32(-1.0) / (smooth_secs * f32(sample_rate))
        self.b = f32(math.exp(float(x)))  # f32::exp
        self.a = f32(1.0) - self.b
        self.status = INACTIVE
        self.input = f32(val)
        self.output = np.full(max_block_frames, f32(val), dtype=f32)
        self.last_output = f32(val)
        self.settle_epsilon = f32(0.00001)

    def is_active(self):
        return self.status != INACTIVE

    def reset(self, val):  # :115-129
        val = f32(val)
        if self.is_active():
            self.status = INACTIVE
            self.input = val
            self.last_output = val
            self.output[:] = val
        elif self.input != val:
            self.input = val
            self.last_output = val
            self.output[:] = val

    def set(self, val):  # :133-140
        val = f32(val)
        if self.input == val:
            return
        self.input = val
        self.status = ACTIVE

    def process(self, frames):  # :159-194 -> (values, is_smoothing)
        frames = min(frames, len(self.output))
        if self.status != ACTIVE or frames == 0 or len(self.output) == 0:
            return self.output, self.status != INACTIVE
        inp = self.input * self.a
        out = self.output
        prev = inp + (self.last_output * self.b)
        out[0] = prev
        b = self.b
        for i in range(1, frames):
            prev = inp + (prev * b)
            out[i] = prev
        self.last_output = out[frames - 1]
        if self.status == ACTIVE:
            if abs(self.input - out[0]) < self.settle_epsilon:
                self.reset(self.input)
                self.status = DEACTIVATING
        elif self.status == DEACTIVATING:
            self.status = INACTIVE
        return self.output[:frames], self.status != INACTIVE

    def set_and_process(self, val, frames):  # :202-205
        self.set(val)
        return self.process(frames)


def percent_volume_to_raw_gain(p):  # core/param/range.rs:32-35
    n = max(f32(p), F0) * (f32(1.0) / f32(100.0))
    return f32(n * n)


_LIBM = None


def platform_powf(x, y):
    """f32::powf is the platform libm's powf (Rust lowers it to the LLVM intrinsic, which calls libm).  glibc's powf is
    NOT correctly rounded everywhere: 10^(0.05 * -15.95945 dB) is 0.50003 ulp away from what it returns (found by the
    differential fuzz, seed 230: a correctly rounded pow put this model one ulp below the oracle on a hard-clip
    threshold).  The reference's value is therefore "whatever libm the host links" — Q29 in DESIGN.md §4 — and this model
    asks the same libm, as a Rust build on this machine would."""
    global _LIBM
    if _LIBM is None:
        import ctypes
        import ctypes.util

        _LIBM = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _LIBM.powf.restype = ctypes.c_float
        _LIBM.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    return f32(_LIBM.powf(float(f32(x)), float(f32(y))))


def db_to_gain_clamped_neg_100_db(db):  # core/util.rs:7-9,21-27
    db = f32(db)
    if db <= f32(-100.0):
        return F0
    return platform_powf(10.0, f32(0.05) * db)  # 10.0f32.powf(0.05 * db)


def clear_all_outputs(frames, outputs):  # core/util.rs:165-175
    for o in outputs:
        o[:frames] = F0
    return (1 << len(outputs)) - 1 if len(outputs) < 64 else (1 << 64) - 1


def all_silent(mask, n):  # core/silence_mask.rs
    full = (1 << n) - 1
    return (mask & full) == full


# ------------------------------------------------------------------------------------------ core/sample_resource.rs
class Sample(object):
    """the six resources of core/sample_resource.rs:28-335: raw data + how fill_buffers reads it"""

    def __init__(self, fmt, channels, data):
        self.fmt, self.channels = fmt, channels
        a = np.ascontiguousarray(np.asarray(data, dtype={0: np.int16, 1: np.uint16, 2: np.float32, 3: np.int16, 4: np.uint16,
                                                         5: np.float32}[fmt]))
        self.frames = a.size // channels
        self.interleaved = fmt <= 2
        self.data = a.reshape(self.frames, channels) if self.interleaved else a.reshape(channels, self.frames)

    def convert(self, x):
        if self.fmt in (0, 3):    # pcm_i16_to_f32 :338-340
            return x.astype(f32) * (f32(1.0) / f32(32767.0))
        if self.fmt in (1, 4):    # pcm_u16_to_f32 :343-345
            return (x.astype(f32) * (f32(2.0) / f32(65535.0))) - f32(1.0)
        return x

    def fill_buffers(self, buffers, lo, hi, start_frame):
        """fill_buffers_interleaved :348-401 / fill_buffers_deinterleaved(_f32) :404-456: which buffers get which channel"""
        n = hi - lo
        ch = self.channels
        if self.interleaved:
            if ch == 1:
                filled = 1
            elif ch == 2 and len(buffers) >= 2:
                filled = 2
            else:
                filled = min(ch, len(buffers))
            for c in range(filled):
                buffers[c][lo:hi] = self.convert(self.data[start_frame:start_frame + n, c])
        else:
            filled = min(ch, len(buffers))  # (the stereo fast path needs both buffers and fills both: same thing)
            for c in range(filled):
                buffers[c][lo:hi] = self.convert(self.data[c, start_frame:start_frame + n])


# ------------------------------------------------------------------------------------------ nodes
class Node(object):
    kind = DUMMY

    def __init__(self, eng, n_in, n_out, params):
        self.n_in, self.n_out = n_in, n_out

    def set_param(self, param, value):
        raise AssertionError("no runtime params")

    def process(self, frames, ins, outs, in_mask):  # basic_nodes/dummy.rs:33-42: writes nothing, out mask NONE_SILENT
        return 0


class VolumeNode(Node):  # basic_nodes/volume.rs
    kind = VOLUME

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        pv = max(f32(params[0] if len(params) else 100.0), F0)
        self.raw_gain = percent_volume_to_raw_gain(pv)
        self.smoother = ParamSmoother(self.raw_gain, eng.sample_rate, eng.max_block_frames)

    def set_param(self, param, value):
        assert param == 0
        self.raw_gain = percent_volume_to_raw_gain(value)  # :28-34 (the atomic store)

    def process(self, frames, ins, outs, in_mask):  # :84-145
        raw_gain = self.raw_gain
        if all_silent(in_mask, len(ins)):
            self.smoother.reset(raw_gain)
            return clear_all_outputs(frames, outs)
        gain, smoothing = self.smoother.set_and_process(raw_gain, frames)
        if not smoothing and gain[0] < f32(0.00001):
            return clear_all_outputs(frames, outs)
        g = gain[:frames]
        if len(ins) == 2 and len(outs) == 2:
            outs[0][:frames] = ins[0][:frames] * g
            outs[1][:frames] = ins[1][:frames] * g
            return in_mask
        for i in range(min(len(ins), len(outs))):
            if (in_mask >> i) & 1:
                outs[i][:frames] = F0
            else:
                outs[i][:frames] = ins[i][:frames] * g
        return in_mask


def platform_sinf(x):
    """f32::sin is the platform libm's sinf (Q29, like powf above)"""
    platform_powf(1.0, 1.0)  # (loads libm)
    if not hasattr(_LIBM, "_sinf_ready"):
        import ctypes

        _LIBM.sinf.restype = ctypes.c_float
        _LIBM.sinf.argtypes = [ctypes.c_float]
        _LIBM._sinf_ready = True
    return f32(_LIBM.sinf(float(f32(x))))


class BeepTestNode(Node):  # basic_nodes/beep_test.rs
    kind = BEEP_TEST
    TAU = f32(6.283185307179586)  # std::f32::consts::TAU

    def __init__(self, eng, n_in, n_out, params):  # :14-24, :48-61
        Node.__init__(self, eng, n_in, n_out, params)
        freq = f32(params[0] if len(params) > 0 else 440.0)
        freq = min(max(freq, f32(20.0)), f32(20000.0)) if not np.isnan(freq) else freq  # f32::clamp: NaN stays NaN
        gain = db_to_gain_clamped_neg_100_db(params[1] if len(params) > 1 else -12.0)
        self.gain = min(max(gain, F0), f32(1.0)) if not np.isnan(gain) else gain
        self.enabled = bool(params[2] != 0.0) if len(params) > 2 else True
        self.phasor = F0
        self.phasor_inc = f32(freq / f32(eng.sample_rate))  # freq_hz / sample_rate as f32

    def set_param(self, param, value):  # :30-32 (the atomic store)
        assert param == 0
        self.enabled = value != 0.0

    def process(self, frames, ins, outs, in_mask):  # :71-97
        if not outs:
            return 0
        if not self.enabled:  # :83-86 — clears outputs[1..] ONLY and reports them through the mask's low bits (Q12)
            return clear_all_outputs(frames, outs[1:])
        out1 = outs[0]
        ph, inc, gain = self.phasor, self.phasor_inc, self.gain
        with np.errstate(all="ignore"):
            for i in range(frames):  # :88-91
                out1[i] = f32(platform_sinf(f32(ph * self.TAU)) * gain)
                t = f32(ph + inc)
                ph = f32(t - np.trunc(t))  # f32::fract
        self.phasor = ph
        for o in outs[1:]:  # :93-95
            o[:frames] = out1[:frames]
        return 0


class SumNode(Node):  # basic_nodes/sum.rs
    kind = SUM

    def process(self, frames, ins, outs, in_mask):  # :41-136
        n_inputs, n_outputs = len(ins), len(outs)
        if all_silent(in_mask, n_inputs):
            return clear_all_outputs(frames, outs)
        if n_inputs == n_outputs:
            for o, i in zip(outs, ins):
                o[:frames] = i[:frames]
            return in_mask
        n = n_inputs // n_outputs
        for ch in range(n_outputs):
            if n in (2, 3, 4):  # :67-109: in1 + in2 (+ in3 (+ in4)), left to right, silent ports included
                acc = ins[ch][:frames] + ins[n_outputs + ch][:frames]
                for p in range(2, n):
                    acc = acc + ins[n_outputs * p + ch][:frames]
                outs[ch][:frames] = acc
            else:  # :111-133
                acc = ins[ch][:frames].copy()
                for p in range(1, n):
                    ic = n_outputs * p + ch
                    if (in_mask >> ic) & 1:
                        continue
                    acc += ins[ic][:frames]
                outs[ch][:frames] = acc
        return 0


class HardClipNode(Node):  # basic_nodes/hard_clip.rs
    kind = HARD_CLIP

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        self.t = db_to_gain_clamped_neg_100_db(params[0] if len(params) else 0.0)

    def process(self, frames, ins, outs, in_mask):  # :51-95
        t = self.t
        if len(ins) == 2 and len(outs) == 2 and (in_mask & 3) == 0:
            for c in (0, 1):
                outs[c][:frames] = np.maximum(np.minimum(ins[c][:frames], t), -t)
            return 0
        for i in range(min(len(ins), len(outs))):
            if (in_mask >> i) & 1:
                outs[i][:frames] = F0
            else:
                outs[i][:frames] = np.maximum(np.minimum(ins[i][:frames], t), -t)
        return in_mask


class MonoToStereoNode(Node):  # basic_nodes/mono_to_stereo.rs:33-50
    kind = MONO_TO_STEREO

    def process(self, frames, ins, outs, in_mask):
        if in_mask & 1:
            return clear_all_outputs(frames, outs)
        outs[0][:frames] = ins[0][:frames]
        outs[1][:frames] = ins[0][:frames]
        return 0


class StereoToMonoNode(Node):  # basic_nodes/stereo_to_mono.rs:33-56
    kind = STEREO_TO_MONO

    def process(self, frames, ins, outs, in_mask):
        if all_silent(in_mask, 2) or len(ins) < 2 or not outs:
            return clear_all_outputs(frames, outs)
        outs[0][:frames] = (ins[0][:frames] + ins[1][:frames]) * f32(0.5)
        return 0


class SamplerNode(Node):  # basic_nodes/sampler.rs
    kind = SAMPLER

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        self.eng = eng
        pv = max(f32(params[0] if len(params) else 100.0), F0)
        self.raw_gain = percent_volume_to_raw_gain(pv)
        self.gain_smoother = ParamSmoother(self.raw_gain, eng.sample_rate, eng.max_block_frames)  # :302-319
        self.playing = False
        self.playhead = 0
        self.loop_range = None  # (start, end, full_range)
        self.sample = None
        self.msgs = []

    def set_param(self, param, value):
        assert param == 0
        self.raw_gain = percent_volume_to_raw_gain(value)  # :171-177

    def _loop_new(self, mode, start_secs, end_secs):  # ProcLoopRange::new :241-263
        sr = float(self.eng.sample_rate)
        if mode == 1:
            return [0, self.sample.frames if self.sample is not None else 0, True]
        return [rust_round_u64(start_secs * sr), rust_round_u64(end_secs * sr), False]

    def process(self, frames, ins, outs, in_mask):  # :323-561
        for m in self.msgs:  # :331-414
            what = m[0]
            if what == "sample":
                self.sample = m[1]
                if self.loop_range is not None and self.loop_range[2]:  # update_sample :265-277
                    self.loop_range[0], self.loop_range[1] = 0, self.sample.frames
                if m[2]:
                    self.playhead = self.loop_range[0] if self.loop_range is not None else 0
                    self.playing = False
            elif what == "play":
                self.playing = True
            elif what == "pause":
                self.playing = False
            elif what == "stop":
                self.playhead = self.loop_range[0] if self.loop_range is not None else 0
                self.playing = False
            elif what == "playhead":
                self.playhead = rust_round_u64(m[1] * float(self.eng.sample_rate))
            elif what == "loop":
                self.loop_range = None if m[1] == 0 else self._loop_new(m[1], m[2], m[3])
                if self.loop_range is not None and self.loop_range[0] <= self.playhead < self.loop_range[1]:
                    self.playhead = self.loop_range[0]
        self.msgs = []
        if self.sample is None or not self.playing:  # :416-430
            return clear_all_outputs(frames, outs)
        sample = self.sample
        gain, smoothing = self.gain_smoother.set_and_process(self.raw_gain, frames)  # :432-433
        assert len(gain) == frames  # :435 (Q5)
        if not smoothing and gain[0] < f32(0.00001):  # :437-443
            return clear_all_outputs(frames, outs)
        out_mask = 0
        if self.loop_range is not None:  # :445-484
            start, end = self.loop_range[0], self.loop_range[1]
            if self.playhead >= end:
                self.playhead = start
            first = min(frames, end - self.playhead)
            sample.fill_buffers(outs, 0, first, self.playhead)
            if first < frames:
                self.playhead = start
                second = frames - first
                sample.fill_buffers(outs, first, frames, self.playhead)
                self.playhead += second
            else:
                self.playhead += frames
        else:  # :485-517
            if self.playhead >= sample.frames:
                self.playing = False
                return clear_all_outputs(frames, outs)
            copy = min(frames, sample.frames - self.playhead)
            sample.fill_buffers(outs, 0, copy, self.playhead)
            if copy < frames:
                self.playing = False
                self.playhead = 0
                for o in outs:
                    o[copy:frames] = F0
            else:
                self.playhead += frames
        sc = sample.channels
        g = gain[:frames]
        if len(outs) >= 2 and sc == 2:  # :522-533
            outs[0][:frames] *= g
            outs[1][:frames] *= g
        else:  # :534-543
            for c in range(min(len(outs), sc)):
                outs[c][:frames] *= g
        if len(outs) > sc:  # :545-559
            if len(outs) == 2 and sc == 1:
                outs[1][:frames] = outs[0][:frames]
            else:
                for i in range(sc, len(outs)):
                    outs[i][:frames] = F0
                    out_mask |= 1 << i
        return out_mask


def rust_round_u64(x):
    """`x.round() as u64`: round half away from zero, saturating, NaN -> 0"""
    if x != x or x <= 0.0:
        return 0
    r = math.floor(x + 0.5) if x < 4.5e15 else x
    return int(min(r, 18446744073709551615.0))


# ------------------------------------------------------------------------------------------ SPEC nodes
def pan_gains(pan):
    p = min(max(float(f32(pan)), -1.0), 1.0)
    if p <= -1.0:
        return f32(1.0), F0
    if p >= 1.0:
        return F0, f32(1.0)
    theta = (p + 1.0) * (math.pi / 4.0)
    return f32(math.cos(theta)), f32(math.sin(theta))


class PanNode(Node):
    kind = STEREO_PAN

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        self.gl, self.gr = pan_gains(params[0] if len(params) else 0.0)
        self.sl = ParamSmoother(self.gl, eng.sample_rate, eng.max_block_frames)
        self.sr_ = ParamSmoother(self.gr, eng.sample_rate, eng.max_block_frames)

    def set_param(self, param, value):
        assert param == 0
        self.gl, self.gr = pan_gains(value)

    def process(self, frames, ins, outs, in_mask):
        if all_silent(in_mask, len(ins)):
            self.sl.reset(self.gl)
            self.sr_.reset(self.gr)
            return clear_all_outputs(frames, outs)
        gl, _ = self.sl.set_and_process(self.gl, frames)
        gr, _ = self.sr_.set_and_process(self.gr, frames)
        outs[0][:frames] = ins[0][:frames] * gl[:frames]
        outs[1][:frames] = ins[1][:frames] * gr[:frames]
        return in_mask


class WidthNode(Node):
    kind = STEREO_WIDTH

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        self.w = max(f32(params[0] if len(params) else 1.0), F0)
        self.s = ParamSmoother(self.w, eng.sample_rate, eng.max_block_frames)

    def set_param(self, param, value):
        assert param == 0
        self.w = max(f32(value), F0)

    def process(self, frames, ins, outs, in_mask):
        if all_silent(in_mask, len(ins)):
            self.s.reset(self.w)
            return clear_all_outputs(frames, outs)
        w, _ = self.s.set_and_process(self.w, frames)
        l, r = ins[0][:frames], ins[1][:frames]
        m = (l + r) * f32(0.5)
        sd = ((l - r) * f32(0.5)) * w[:frames]
        outs[0][:frames] = m + sd
        outs[1][:frames] = m - sd
        return 0


def rbj_coefs(ftype, cutoff_hz, q, sample_rate):
    """RBJ cookbook in f64, a0-normalised, rounded to f32: 0 low-pass, 1 high-pass, 2 band-pass (constant 0 dB peak gain);
    cutoff clamped to [1 Hz, 0.49 fs], Q >= 1e-3"""
    fs = float(sample_rate)
    f0 = min(max(float(f32(cutoff_hz)), 1.0), 0.49 * fs)
    Q = max(float(f32(q)), 1e-3)
    w0 = 2.0 * math.pi * f0 / fs
    cw, alpha = math.cos(w0), math.sin(w0) / (2.0 * Q)
    a0, a1, a2 = 1.0 + alpha, -2.0 * cw, 1.0 - alpha
    if ftype == 1:
        b0, b1, b2 = (1.0 + cw) * 0.5, -(1.0 + cw), (1.0 + cw) * 0.5
    elif ftype == 2:
        b0, b1, b2 = alpha, 0.0, -alpha
    else:
        b0, b1, b2 = (1.0 - cw) * 0.5, 1.0 - cw, (1.0 - cw) * 0.5
    return [f32(b0 / a0), f32(b1 / a0), f32(b2 / a0), f32(a1 / a0), f32(a2 / a0)]


class BiquadNode(Node):
    kind = BIQUAD

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        self.sr = eng.sample_rate
        self.ftype = int(params[0]) if len(params) > 0 else 0
        self.cutoff = f32(params[1]) if len(params) > 1 else f32(1000.0)
        self.q = f32(params[2]) if len(params) > 2 else f32(0.70710678)
        self.co = rbj_coefs(self.ftype, self.cutoff, self.q, self.sr)
        self.nch = min(n_in, n_out)
        self.st = np.zeros((self.nch, 4), dtype=f32)  # x1 x2 y1 y2 per channel

    def set_param(self, param, value):
        if param == 1:
            self.cutoff = f32(value)
        else:
            assert param == 2
            self.q = f32(value)
        self.co = rbj_coefs(self.ftype, self.cutoff, self.q, self.sr)

    @staticmethod
    def process_batch(items, frames):
        """items: [(node, ins, outs)] — every biquad of one schedule level, their channels side by side, ONE time loop.
        ff = ((b0*x) + (b1*x1)) + (b2*x2);  y = fma(-a1, y1, fma(-a2, y2, ff))"""
        rows = [(n, c, ins[c], outs[c]) for (n, ins, outs) in items for c in range(n.nch)]
        if not rows:
            return
        co = np.array([r[0].co for r in rows], dtype=f32)
        st = np.array([r[0].st[r[1]] for r in rows], dtype=f32)
        x = np.stack([r[2][:frames] for r in rows]).astype(f32)
        b0, b1, b2, na1, na2 = co[:, 0], co[:, 1], co[:, 2], -co[:, 3], -co[:, 4]
        # feed-forward half for the whole block at once (x1, x2 are the inputs shifted by one / two frames)
        xp1 = np.concatenate([st[:, 0:1], x[:, :-1]], axis=1)
        xp2 = np.concatenate([st[:, 1:2], st[:, 0:1], x[:, :-2]], axis=1) if frames >= 2 else st[:, 1:2].copy()
        ff = ((b0[:, None] * x) + (b1[:, None] * xp1)) + (b2[:, None] * xp2)
        y = np.empty_like(x)
        y1, y2 = st[:, 2].copy(), st[:, 3].copy()
        for i in range(frames):
            t = fma32(na2, y2, ff[:, i])
            yi = fma32(na1, y1, t)
            y[:, i] = yi
            y2, y1 = y1, yi
        for k, (n, c, _, out) in enumerate(rows):
            out[:frames] = y[k]
            n.st[c, 0] = x[k, frames - 1]
            n.st[c, 1] = x[k, frames - 2] if frames >= 2 else st[k, 0]
            n.st[c, 2] = y1[k]
            n.st[c, 3] = y2[k]

    def process(self, frames, ins, outs, in_mask):
        BiquadNode.process_batch([(self, ins, outs)], frames)
        return 0


class DelayNode(Node):
    kind = DELAY

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        secs = float(f32(params[0])) if len(params) > 0 else 0.1
        d = round_half_away(secs * float(eng.sample_rate))
        self.D = int(min(max(d, 1.0), 16777216.0))
        self.fb = min(max(f32(params[1]) if len(params) > 1 else F0, F0), f32(0.999))
        self.mix = min(max(f32(params[2]) if len(params) > 2 else f32(0.5), F0), f32(1.0))
        self.dry = f32(1.0) - self.mix
        self.nch = min(n_in, n_out)
        self.ring = np.zeros((self.nch, self.D), dtype=f32)
        self.pos = 0

    def set_param(self, param, value):
        if param == 1:
            self.fb = min(max(f32(value), F0), f32(0.999))
        else:
            assert param == 2
            self.mix = min(max(f32(value), F0), f32(1.0))
            self.dry = f32(1.0) - self.mix

    def process(self, frames, ins, outs, in_mask):
        D = self.D
        for c in range(self.nch):
            x, out, ring = ins[c], outs[c], self.ring[c]
            done = 0
            while done < frames:  # frames of one chunk touch distinct ring slots: d = ring[p]; ring[p] = x + d*fb
                n = min(D, frames - done)
                idx = (self.pos + done + np.arange(n)) % D
                d = ring[idx]
                xs = x[done:done + n]
                ring[idx] = xs + (d * self.fb)
                out[done:done + n] = (xs * self.dry) + (d * self.mix)
                done += n
        self.pos = (self.pos + frames) % D
        return 0


def round_half_away(x):
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


SP_HIST = 64


def spatial_params(x, y, z, sample_rate):
    x, y, z = float(f32(x)), float(f32(y)), float(f32(z))
    d = math.sqrt(x * x + y * y + z * z)
    att = 1.0 / max(d, 1.0)
    s = 0.0 if d < 1e-9 else x / d
    theta = (s + 1.0) * (math.pi / 4.0)
    itd = min(round_half_away(0.00066 * float(sample_rate)), SP_HIST - 1)
    return (f32(math.cos(theta) * att), f32(math.sin(theta) * att), int(round_half_away(max(0.0, s) * itd)),
            int(round_half_away(max(0.0, -s) * itd)))


class SpatialNode(Node):
    kind = SPATIAL

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        self.sr = eng.sample_rate
        self.xyz = [f32(params[0]) if len(params) > 0 else F0, f32(params[1]) if len(params) > 1 else F0,
                    f32(params[2]) if len(params) > 2 else f32(-1.0)]
        self.gl, self.gr, self.dl, self.dr = spatial_params(*self.xyz, sample_rate=self.sr)
        self.sl = ParamSmoother(self.gl, eng.sample_rate, eng.max_block_frames)
        self.sr_ = ParamSmoother(self.gr, eng.sample_rate, eng.max_block_frames)
        self.hist = np.zeros(SP_HIST, dtype=f32)

    def set_param(self, param, value):
        self.xyz[param] = f32(value)
        self.gl, self.gr, self.dl, self.dr = spatial_params(*self.xyz, sample_rate=self.sr)

    def process(self, frames, ins, outs, in_mask):
        m = ins[0][:frames].copy() if self.n_in < 2 else (ins[0][:frames] + ins[1][:frames]) * f32(0.5)
        gl, _ = self.sl.set_and_process(self.gl, frames)
        gr, _ = self.sr_.set_and_process(self.gr, frames)
        ext = np.concatenate([self.hist, m])  # ext[SP_HIST + i] = m[i]
        i = np.arange(frames)
        outs[0][:frames] = ext[SP_HIST + i - self.dl] * gl[:frames]
        outs[1][:frames] = ext[SP_HIST + i - self.dr] * gr[:frames]
        self.hist = ext[-SP_HIST:].copy()
        return 0


# ------------------------------------------------------------------------------------------ SPEC resampling source
# 32 phases x 16 taps of a Kaiser-windowed sinc (cutoff 0.9, beta 8), each row normalised to unit sum;
# the source position is a 32.32 fixed-point frame index advanced by `step` per output frame; out[n] = sum_k
# h[phase][k] * s[idx - 7 + k] as an ascending fused chain from +0.0, phase = top 5 fraction bits; outside [0, len) a
# one-shot reads 0, a loop wraps.  Written from that description with numpy / scipy (np.sinc, scipy.special.i0: not the
# power series the C++ sides use).
RS_PHASES, RS_TAPS = 32, 16


class FirNode(Node):
    """SPEC (summation order §3.4): y[n] = sum_k h[k] x[n-k] evaluated over the block's window of
    W = T-1+frames input samples (the last T-1 of the previous blocks, then this block's) in ascending WINDOW position: the
    taps that do not reach output n count as +0.0; positions are cut into segments of 4096, each segment is one fma chain
    from +0.0, the segment sums are added in order, and +0.0 is added at the end.  Channel c convolves with channel
    min(c, C_h - 1) of the impulse response; silence masks are ignored (the tail keeps ringing), the out mask stays clear."""
    kind = fwapi.FIR
    SEG = 4096

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        ir = eng.samples[int(params[0])]
        self.T = ir.frames
        tmp = [np.zeros(self.T, dtype=f32) for _ in range(ir.channels)]
        ir.fill_buffers(tmp, 0, self.T, 0)
        self.nch = min(n_in, n_out)
        self.h = [tmp[min(c, ir.channels - 1)] for c in range(self.nch)]
        self.hist = [np.zeros(self.T - 1, dtype=f32) for _ in range(self.nch)]

    def process(self, frames, ins, outs, in_mask):
        T, W = self.T, self.T - 1 + frames
        i = np.arange(frames)
        for c in range(min(self.nch, len(ins), len(outs))):
            win = np.concatenate([self.hist[c], ins[c][:frames]]).astype(f32)
            h = self.h[c]
            total = None
            with np.errstate(all="ignore"):
                for s0 in range(0, W, self.SEG):
                    acc = np.zeros(frames, dtype=f32)
                    for m in range(s0, min(W, s0 + self.SEG)):
                        d = m - i                                   # window position m is x[n - k] with k = T-1 - (m - n)
                        ok = (d >= 0) & (d <= T - 1)
                        hv = np.where(ok, h[np.clip(T - 1 - d, 0, T - 1)], F0)
                        acc = fma32(np.full(frames, win[m], dtype=f32), hv, acc)
                    total = acc if total is None else (total + acc).astype(f32)
                outs[c][:frames] = (total + F0).astype(f32)
            if T > 1:
                self.hist[c] = win[W - (T - 1):].copy()
        return 0


def resampler_table():
    from scipy.special import i0

    fc, beta = 0.9, 8.0
    k = np.arange(RS_TAPS, dtype=np.float64)[None, :]
    ph = np.arange(RS_PHASES, dtype=np.float64)[:, None]
    t = (k - (RS_TAPS // 2 - 1)) - ph / RS_PHASES
    r = t / (RS_TAPS / 2.0)
    w = np.where(np.abs(r) >= 1.0, 0.0, i0(beta * np.sqrt(np.clip(1.0 - r * r, 0.0, None))) / i0(beta))
    row = fc * np.sinc(fc * t) * w
    return (row / row.sum(axis=1, keepdims=True)).astype(f32)


def resampler_step(ratio):
    r = float(f32(ratio))
    if not r >= 1.0 / 256.0:
        r = 1.0 / 256.0
    r = min(r, 256.0)
    return int(round_half_away(r * 4294967296.0))


class ResamplerNode(Node):
    kind = RESAMPLER
    _table = None

    def __init__(self, eng, n_in, n_out, params):
        Node.__init__(self, eng, n_in, n_out, params)
        assert n_in == 0 and n_out >= 1
        p = list(params) + [None] * 4
        self.src = eng.samples[int(p[0])]
        self.step = resampler_step(1.0 if p[1] is None else p[1])
        self.loop = bool(p[2]) if p[2] is not None else False
        self.playing_ctl = (p[3] != 0.0) if p[3] is not None else True
        self.seek = None
        self.pos = 0
        if ResamplerNode._table is None:
            ResamplerNode._table = resampler_table()
        self.h = ResamplerNode._table

    def set_param(self, param, value):  # 1 = ratio, 3 = playing, 4 = seek to a source frame
        if param == 1:
            self.step = resampler_step(value)
        elif param == 3:
            self.playing_ctl = f32(value) != F0
        elif param == 4:
            self.seek = int(max(float(f32(value)), 0.0))  # (truncation, as a float -> u64 cast)
        else:
            raise AssertionError("resampler param %d" % param)

    def _channel(self, c, j):
        """source channel c at frame indices j (int64 array, all inside the sample), converted to f32"""
        s = self.src
        return s.convert(s.data[j, c] if s.interleaved else s.data[c, j])

    def process(self, frames, ins, outs, in_mask):
        if self.seek is not None:
            self.pos = (self.seek << 32) & ((1 << 64) - 1)
            self.seek = None
        n = self.src.frames
        if not self.playing_ctl or n == 0:
            return clear_all_outputs(frames, outs)
        sch = self.src.channels
        nfill = min(self.n_out, sch)
        p = [(self.pos + i * self.step) & ((1 << 64) - 1) for i in range(frames)]
        idx = np.array([q >> 32 for q in p], dtype=np.int64)
        ph = np.array([(q >> 27) & (RS_PHASES - 1) for q in p], dtype=np.int64)
        mask = 0
        for c in range(nfill):
            acc = np.zeros(frames, dtype=f32)
            for k in range(RS_TAPS):
                j = idx - (RS_TAPS // 2 - 1) + k
                if self.loop:
                    x = self._channel(c, j % n)
                else:
                    inside = (j >= 0) & (j < n)
                    x = np.where(inside, self._channel(c, np.where(inside, j, 0)), F0).astype(f32)
                acc = fma32(self.h[ph, k], x, acc)
            outs[c][:frames] = acc
        if self.n_out > sch:  # like the sampler: a mono source feeds both outputs of a stereo node, anything else is zero + flagged
            if self.n_out == 2 and sch == 1:
                outs[1][:frames] = outs[0][:frames]
            else:
                for c in range(sch, self.n_out):
                    outs[c][:frames] = F0
                    mask |= 1 << c
        self.pos = (self.pos + frames * self.step) & ((1 << 64) - 1)
        if self.loop:
            self.pos %= n << 32
        elif (self.pos >> 32) >= n + RS_TAPS // 2:
            self.playing_ctl = False  # ran off the end: silent from the next block on
        return mask


NODE_CLASSES = {DUMMY: Node, BEEP_TEST: BeepTestNode, VOLUME: VolumeNode, SUM: SumNode, SAMPLER: SamplerNode, HARD_CLIP: HardClipNode,
                MONO_TO_STEREO: MonoToStereoNode, STEREO_TO_MONO: StereoToMonoNode, STEREO_PAN: PanNode, STEREO_WIDTH: WidthNode,
                BIQUAD: BiquadNode, DELAY: DelayNode, fwapi.FIR: FirNode, SPATIAL: SpatialNode, RESAMPLER: ResamplerNode}


# ------------------------------------------------------------------------------------------ graph + processor
class RefEngine(fwapi.Engine):
    """fwapi.Engine surface (the one OracleEngine / GpuEngine have) over the numpy model; wrap it in
    scenarios.TaggedOracle for messages tagged with a block."""

    backend = "refmodel"

    def __init__(self, sample_rate=48000, max_block_frames=256, num_graph_inputs=0, num_graph_outputs=2):
        self.sample_rate, self.max_block_frames = sample_rate, max_block_frames
        self.nodes = {}      # id -> Node
        self.in_edge = {}    # id -> [None | (src id, src port)] per input port
        self.next_id = 0
        self.samples = []
        self.graph_in_node = self._add(Node(self, 0, num_graph_inputs, ()))
        self.graph_out_node = self._add(Node(self, num_graph_outputs, 0, ()))
        self.plan = None

    def _add(self, node):
        nid = self.next_id
        self.next_id += 1
        self.nodes[nid] = node
        self.in_edge[nid] = [None] * node.n_in
        self.plan = None
        return nid

    def add_node(self, kind, n_in, n_out, params=()):
        return self._add(NODE_CLASSES[kind](self, n_in, n_out, [float(p) for p in params]))

    def remove_node(self, node):
        assert node not in (self.graph_in_node, self.graph_out_node)
        del self.nodes[node]
        del self.in_edge[node]
        for ports in self.in_edge.values():
            for p, e in enumerate(ports):
                if e is not None and e[0] == node:
                    ports[p] = None
        self.plan = None
        return 0

    def connect(self, src, sp, dst, dp, check_for_cycles=False):
        assert self.in_edge[dst][dp] is None and sp < self.nodes[src].n_out
        self.in_edge[dst][dp] = (src, sp)
        self.plan = None
        return 0

    def update(self):
        """levelised schedule: a node runs after everything it reads (any topological order gives the same audio: every
        output port owns its buffer); buffer 0 is the cleared, silent-flagged buffer unconnected inputs read (should_clear)"""
        level = {}

        def lv(n):
            if n not in level:
                level[n] = 0  # (no cycles in the tests' graphs)
                level[n] = 1 + max([lv(e[0]) for e in self.in_edge[n] if e is not None] or [-1])
            return level[n]

        order = sorted(self.nodes, key=lambda n: (lv(n), n))
        buf_of, nbuf = {}, 1
        for n in order:
            for p in range(self.nodes[n].n_out):
                buf_of[(n, p)] = nbuf
                nbuf += 1
        old = getattr(self, "pool", None)
        self.pool = np.zeros((nbuf, self.max_block_frames), dtype=f32)  # a new schedule starts from zeroed buffers (schedule.rs:202-203)
        self.flags = np.zeros(nbuf, dtype=bool)
        del old
        self.plan = [(n, lv(n), [0 if e is None else buf_of[e] for e in self.in_edge[n]],
                      [buf_of[(n, p)] for p in range(self.nodes[n].n_out)]) for n in order]

    # ---- samples + messages (immediate: seen at the top of the next block, like the reference's rings / atomics)
    def new_sample(self, fmt, channels, data):
        self.samples.append(Sample(fmt, channels, data))
        return len(self.samples) - 1

    def set_param(self, node, param, value, at_block=0):
        assert at_block == 0
        self.nodes[node].set_param(param, f32(value))

    def sampler_set_sample(self, node, sample, stop_playback=False, at_block=0):
        self.nodes[node].msgs.append(("sample", self.samples[sample], bool(stop_playback)))

    def sampler_play(self, node, at_block=0):
        self.nodes[node].msgs.append(("play",))

    def sampler_pause(self, node, at_block=0):
        self.nodes[node].msgs.append(("pause",))

    def sampler_stop(self, node, at_block=0):
        self.nodes[node].msgs.append(("stop",))

    def sampler_set_playhead_secs(self, node, secs, at_block=0):
        self.nodes[node].msgs.append(("playhead", float(secs)))

    def sampler_set_loop_range(self, node, mode, start=0.0, end=0.0, at_block=0):
        self.nodes[node].msgs.append(("loop", mode, float(start), float(end)))

    # ---- graph/processor.rs:61-165
    def process_interleaved(self, frames, n_out_ch=2, inp=None, n_in_ch=0, t=0.0, status=0):
        out = np.zeros(frames * n_out_ch, dtype=f32)
        if self.plan is None or frames == 0:  # :86-89
            return out
        inp = np.zeros(frames * n_in_ch, dtype=f32) if inp is None else np.asarray(inp, dtype=f32)
        done = 0
        while done < frames:
            bf = min(frames - done, self.max_block_frames)
            self._block(bf, inp[done * n_in_ch:(done + bf) * n_in_ch], n_in_ch, out[done * n_out_ch:(done + bf) * n_out_ch], n_out_ch)
            done += bf
        return out

    def process_blocks(self, k, n_out_ch=2):
        return self.process_interleaved(k * self.max_block_frames, n_out_ch)

    def _block(self, frames, inp, n_in_ch, out, n_out_ch):
        pool, flags = self.pool, self.flags
        pool[0, :] = F0
        flags[0] = True
        # prepare_graph_inputs (schedule.rs:213-253) + deinterleave (util.rs:44-87)
        gin = self.plan[0]
        assert gin[0] == self.graph_in_node
        fill = min(n_in_ch, len(gin[3]))
        for i in range(fill):
            pool[gin[3][i], :frames] = inp[i::n_in_ch][:frames]
        for b in gin[3][fill:]:
            pool[b, :frames] = F0
        # schedule.rs:289-344 — graph_in is a scheduled (Dummy) node: its out mask NONE_SILENT overwrites the flags
        # prepare_graph_inputs computed (Q10); graph_out is a Dummy with inputs only
        i = 0
        plan = self.plan
        while i < len(plan):
            lvl = plan[i][1]
            batch = []
            while i < len(plan) and plan[i][1] == lvl:
                nid, _, in_bufs, out_bufs = plan[i]
                i += 1
                node = self.nodes[nid]
                ins = [pool[b] for b in in_bufs]
                outs = [pool[b] for b in out_bufs]
                in_mask = 0
                for p, b in enumerate(in_bufs):
                    if flags[b]:
                        in_mask |= 1 << p
                if node.kind == BIQUAD:
                    batch.append((node, ins, outs))
                    om = 0
                else:
                    om = node.process(frames, ins, outs, in_mask)
                for p, b in enumerate(out_bufs):
                    flags[b] = bool((om >> p) & 1)
            if batch:
                BiquadNode.process_batch(batch, frames)
        # read_graph_outputs (schedule.rs:255-287) + interleave / interleave_stereo (util.rs:90-147)
        gout = [p for p in plan if p[0] == self.graph_out_node][0]
        n_read = min(n_out_ch, len(gout[2]))
        mask = [bool(flags[gout[2][c]]) for c in range(n_read)]
        if n_read == 2 and n_out_ch == 2:
            if mask[0] and mask[1]:
                out[:] = F0
            else:
                out[0::2] = pool[gout[2][0], :frames]
                out[1::2] = pool[gout[2][1], :frames]
            return
        out[:] = F0
        for c in range(n_read):
            if not mask[c]:
                out[c::n_out_ch] = pool[gout[2][c], :frames]
