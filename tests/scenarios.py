"""Deterministic graph scenarios run identically on the oracle and on the GPU engines (fwapi.Engine).
Each returns the interleaved output of every process call, concatenated."""
import numpy as np

import fwapi
from fwapi import (BEEP_TEST, DUMMY, HARD_CLIP, INTERLEAVED_F32, INTERLEAVED_I16, INTERLEAVED_U16, LOOP_FULL, LOOP_NONE,
                   LOOP_RANGE_SECS, MONO_TO_STEREO, PLANAR_F32, PLANAR_I16, PLANAR_U16, SAMPLER, STEREO_PAN,
                   STEREO_TO_MONO, SUM, VOLUME)


def voice_source(seed, frames, channels=2):
    return fwapi.xorshift_uniform(0xF1EE0000 + seed, channels * frames).reshape(channels, frames)


def connect_through_master(e, root, master):
    """root -> master[0] -> master[1] ... -> graph_out; `master` = callables creating stereo 2->2 nodes.  Returns the nodes."""
    made = []
    cur = root
    for mk in master:
        n = mk(e)
        e.connect_stereo(cur, n)
        made.append(n)
        cur = n
    e.connect_stereo(cur, e.graph_out_node)
    return made


def build_voice_bank(e, n_voices, radix=32, src_frames=4096, with_pan=True, with_volume=True, seed=0, fmt=PLANAR_F32,
                     mono_every=0, fmt_cycle=None, master=(), voice_fx=None):
    """config-2 shape: V x (sampler -> volume -> pan) -> radix-`radix` SumNode tree -> graph_out.
    voice_fx(e, v, rng) -> list of extra stereo 2->2 nodes appended to voice v's chain (width / hard clip / ...)."""
    rng = np.random.default_rng(1234 + seed)
    voices = []
    ends = []
    for v in range(n_voices):
        s = e.sampler(100.0)
        cur = s
        vol = pan = None
        if with_volume:
            vol = e.volume(float(rng.uniform(10, 100)))
            e.connect_stereo(cur, vol)
            cur = vol
        if with_pan:
            pan = e.pan(float(rng.uniform(-1, 1)))
            e.connect_stereo(cur, pan)
            cur = pan
        fx = []
        if voice_fx is not None:
            for n in voice_fx(e, v, rng):
                e.connect_stereo(cur, n)
                cur = n
                fx.append(n)
        voices.append(dict(sampler=s, volume=vol, pan=pan, fx=fx))
        ends.append(cur)
    # sum tree
    level = ends
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.sum(len(grp))
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    voices[0]["master"] = connect_through_master(e, level[0], master)
    e.update()
    for v, vc in enumerate(voices):
        ch = 1 if (mono_every and v % mono_every == 0) else 2
        data = voice_source(seed * 100000 + v, src_frames, ch)
        vfmt = fmt_cycle[v % len(fmt_cycle)] if fmt_cycle else fmt   # per-voice formats: mixed leaves
        if vfmt in (PLANAR_I16, INTERLEAVED_I16):
            raw = np.round(data * 32767).astype(np.int16)
        elif vfmt in (PLANAR_U16, INTERLEAVED_U16):
            raw = np.round((data + 1) * 32767.5).astype(np.uint16)
        else:
            raw = data
        if vfmt <= INTERLEAVED_F32:
            raw = raw.T.copy()
        vc["sample"] = e.new_sample(vfmt, ch, raw)
        vc["frames"] = src_frames
        e.sampler_set_sample(vc["sampler"], vc["sample"])
    return voices


def scenario_voice_bank_steady(e, n_voices=96, blocks=6, radix=32, **kw):
    """steady state: all voices looping, constant gains."""
    voices = build_voice_bank(e, n_voices, radix=radix, **kw)
    for vc in voices:
        e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        e.sampler_play(vc["sampler"])
    return e.process_blocks(blocks)


def scenario_voice_bank_events(e, n_voices=70, radix=32, mbf=None, src_frames=1000, fmt=PLANAR_F32):
    """everything at once: loop wraps inside blocks (src_frames not a multiple of the block), one-shot ends,
    paused voices (silence masks), gain / pan changes (smoother ramps that settle, and ones that stall),
    mute -> all-silent chains, messages tagged at later blocks of a multi-block call."""
    voices = build_voice_bank(e, n_voices, radix=radix, src_frames=src_frames, mono_every=7, fmt=fmt)
    outs = []
    for v, vc in enumerate(voices):
        if v % 5 != 3:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)   # v%5==3: one-shot
        if v % 4 != 1:
            e.sampler_play(vc["sampler"])                         # v%4==1: never started yet
    outs.append(e.process_blocks(3))
    # gain changes seen at block 0 and block 2 of the next call; pan change; mute; start paused voices late
    for v, vc in enumerate(voices):
        if v % 3 == 0:
            e.set_param(vc["volume"], 0, 25.0 if v % 2 else 90.0, at_block=0)
        if v % 6 == 2:
            e.set_param(vc["pan"], 0, -0.5, at_block=2)
        if v % 10 == 4:
            e.set_param(vc["sampler"], 0, 0.0, at_block=1)        # ramps to 0 -> settles -> (Q3) stays unflagged
        if v % 4 == 1:
            e.sampler_play(vc["sampler"], at_block=1)
        if v % 11 == 5:
            e.sampler_pause(vc["sampler"], at_block=3)
    outs.append(e.process_blocks(5))
    # long tail so ramps settle / stall, one-shots end, loops wrap several times
    outs.append(e.process_blocks(28))
    # stop + restart some, set playhead, change loop range (Q7), second gain change
    for v, vc in enumerate(voices):
        if v % 9 == 0:
            e.sampler_stop(vc["sampler"])
        if v % 9 == 1:
            e.sampler_set_playhead_secs(vc["sampler"], 300.25 / e.sample_rate)
        if v % 9 == 2:
            # keep the range inside the sample: the reference panics on an out-of-range slice (Q8)
            e.sampler_set_loop_range(vc["sampler"], LOOP_RANGE_SECS, 100.0 / e.sample_rate,
                                     (src_frames - 100.0) / e.sample_rate)
        if v % 9 == 3:
            e.sampler_play(vc["sampler"])
        if v % 3 == 0:
            e.set_param(vc["volume"], 0, 100.0)
    outs.append(e.process_blocks(8))
    return np.concatenate(outs)


def scenario_ref_desk(e, n_voices=30, radix=8, src_frames=1100, seed=5):
    """REFERENCE KINDS ONLY (sampler, volume, hard clip, mono<->stereo, sum, beep: crates/firewheel-graph/src/basic_nodes/) — the
    scenario the Rust replay on the real firewheel-graph can run (rust/firewheel-gpu/tests/reference_digests.rs): a small desk of
    sampler -> volume [-> hard clip] voices in all six sample formats, mono samplers through MonoToStereo, a beep, a radix sum
    tree, a master volume; loop wraps inside blocks, one-shot ends, pauses (silence masks), gain changes that settle and that stall,
    a fade to zero, sample swaps, playhead jumps, loop ranges (Q7) — whole blocks only (Q5), ranges inside the samples (Q8)."""
    fmts = [PLANAR_F32, INTERLEAVED_I16, PLANAR_U16, INTERLEAVED_F32, PLANAR_I16, INTERLEAVED_U16]
    voices, ends = [], []
    for v in range(n_voices):
        mono = v % 6 == 4
        s = e.sampler(100.0 if v % 4 else 70.0, n_out=1 if mono else 2)
        cur = s
        if mono:
            m2s = e.add_node(MONO_TO_STEREO, 1, 2)
            e.connect(s, 0, m2s, 0)
            cur = m2s
        vol = e.volume(30.0 + 2.0 * v)
        e.connect_stereo(cur, vol)
        cur = vol
        if v % 3 == 1:
            clip = e.hard_clip(-6.0 - v * 0.25)
            e.connect_stereo(cur, clip)
            cur = clip
        voices.append(dict(sampler=s, volume=vol, mono=mono))
        ends.append(cur)
    beep = e.beep(330.0, -18.0, True, n_out=2)
    ends.append(beep)
    level = ends
    while len(level) > 1:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.sum(len(grp))
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            nxt.append(m)
        level = nxt
    master = e.volume(80.0)
    e.connect_stereo(level[0], master)
    e.connect_stereo(master, e.graph_out_node)
    e.update()
    samples = []
    for v, vc in enumerate(voices):
        ch = 1 if (vc["mono"] or v % 7 == 2) else 2
        vfmt = fmts[v % 6]
        data = voice_source(seed * 100000 + v, src_frames + 13 * v, ch)
        if vfmt in (PLANAR_I16, INTERLEAVED_I16):
            raw = np.round(data * 32767).astype(np.int16)
        elif vfmt in (PLANAR_U16, INTERLEAVED_U16):
            raw = np.round((data + 1) * 32767.5).astype(np.uint16)
        else:
            raw = data
        if vfmt <= INTERLEAVED_F32:
            raw = raw.T.copy()
        samples.append(e.new_sample(vfmt, ch, raw))
        vc["frames"] = src_frames + 13 * v
        e.sampler_set_sample(vc["sampler"], samples[-1])
        if v % 5 != 3:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)   # v % 5 == 3: one-shots
        if v % 4 != 1:
            e.sampler_play(vc["sampler"])                         # v % 4 == 1: started later
    outs = [e.process_blocks(3)]
    for v, vc in enumerate(voices):
        if v % 3 == 0:
            e.set_param(vc["volume"], 0, 25.0 if v % 2 else 90.0, at_block=0)
        if v % 10 == 4:
            e.set_param(vc["sampler"], 0, 0.0, at_block=1)        # fades to 0, settles (Q3)
        if v % 4 == 1:
            e.sampler_play(vc["sampler"], at_block=1)
        if v % 11 == 5:
            e.sampler_pause(vc["sampler"], at_block=3)
    e.set_param(master, 0, 55.0, at_block=2)
    outs.append(e.process_blocks(5))
    outs.append(e.process_blocks(26))                              # ramps settle or stall (Q28), one-shots end, loops wrap
    for v, vc in enumerate(voices):
        if v % 9 == 0:
            e.sampler_stop(vc["sampler"])
        if v % 9 == 1:
            e.sampler_set_playhead_secs(vc["sampler"], 300.25 / e.sample_rate)
        if v % 9 == 2:
            e.sampler_set_loop_range(vc["sampler"], LOOP_RANGE_SECS, 100.0 / e.sample_rate, (vc["frames"] - 100.0) / e.sample_rate)
        if v % 9 == 3:
            e.sampler_play(vc["sampler"])
        if v % 9 == 4 and not vc["mono"]:                          # another sample, playback kept / stopped
            e.sampler_set_sample(vc["sampler"], samples[(v + 6) % n_voices], stop_playback=(v % 2 == 0))
        if v % 9 == 5:
            e.set_param(vc["volume"], 0, 0.0)                      # mute: the chain behind it goes silent once it has settled
    outs.append(e.process_blocks(4))
    for v, vc in enumerate(voices):
        if v % 9 == 0 or (v % 9 == 4 and v % 2 == 0):
            e.sampler_play(vc["sampler"], at_block=1)
        if v % 9 == 5:
            e.set_param(vc["volume"], 0, 65.0, at_block=2)
    e.set_param(master, 0, 100.0)
    outs.append(e.process_blocks(7))
    return np.concatenate(outs)


def scenario_message_storm(e, n_voices=48, radix=8, blocks=40, per_voice=6, src_frames=5000, seed=3):
    """hundreds to thousands of messages inside ONE call: every voice gets `per_voice` gain / pan / pause / play messages at
    random blocks (several per block on some voices).  The control kernel finds a voice's messages in the sorted list of the
    call by a wave-wide search — one round for <= 128 messages, 64-ary rounds beyond — and walks them with per-node cursors."""
    rng = np.random.default_rng(seed)
    voices = build_voice_bank(e, n_voices, radix=radix, src_frames=src_frames, mono_every=5)
    for vc in voices:
        e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        e.sampler_play(vc["sampler"])
    outs = [e.process_blocks(2)]
    for vc in voices:
        for _ in range(per_voice):
            at = int(rng.integers(0, blocks))
            kind = int(rng.integers(0, 6))
            if kind <= 2:
                e.set_param(vc["volume"], 0, float(rng.uniform(0.0, 100.0)), at_block=at)
            elif kind == 3:
                e.set_param(vc["pan"], 0, float(rng.uniform(-1.0, 1.0)), at_block=at)
            elif kind == 4:
                e.sampler_pause(vc["sampler"], at_block=at)
            else:
                e.sampler_play(vc["sampler"], at_block=at)
    outs.append(e.process_blocks(blocks))
    outs.append(e.process_blocks(3))
    return np.concatenate(outs)


def width_clip_fx(e, v, rng, limit=3):
    """per-voice tail of the stage-program tests: width and / or hard clip in varying order and number"""
    wv, cv = float(rng.uniform(0.0, 2.0)), float(rng.uniform(-20.0, -2.0))
    shape = ["wc", "cw", "w", "c", "cwC", ""][v % 6][:limit]  # (only the nodes a voice uses are created: nothing dangles)
    return [e.width(wv) if k == "w" else e.hard_clip(cv if k == "c" else -1.0) for k in shape]


def scenario_voice_fx_events(e, n_voices=45, radix=8, src_frames=1100, with_pan=True, fmt=PLANAR_F32):
    """the voice-bank plan's stage programs: every voice ends in width / hard-clip stages (sampler -> volume -> pan -> ...),
    with width automation (ramps that settle and stall), negative widths (clamped), mutes in front of the width (silence
    passes width and clip: the smoother resets), late starts, one-shots that end, mono sources"""
    voices = build_voice_bank(e, n_voices, radix=radix, src_frames=src_frames, mono_every=7, with_pan=with_pan, fmt=fmt,
                              voice_fx=width_clip_fx)
    outs = []
    for v, vc in enumerate(voices):
        if v % 5 != 3:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        if v % 4 != 1:
            e.sampler_play(vc["sampler"])
    outs.append(e.process_blocks(3))
    for v, vc in enumerate(voices):
        widths = [n for n in vc["fx"][:2] if v % 6 in (0, 2) and n == vc["fx"][0]] + ([vc["fx"][1]] if v % 6 in (1, 4) else [])
        for w in widths:
            e.set_param(w, 0, [0.0, 1.7, -0.5, 0.9][v % 4], at_block=v % 3)
        if v % 3 == 0:
            e.set_param(vc["volume"], 0, 25.0 if v % 2 else 120.0, at_block=1)
        if v % 10 == 4:
            e.set_param(vc["volume"], 0, 0.0, at_block=1)          # mute in front of the width / clip
        if v % 4 == 1:
            e.sampler_play(vc["sampler"], at_block=2)
        if v % 11 == 5:
            e.sampler_pause(vc["sampler"], at_block=3)
    outs.append(e.process_blocks(6))
    outs.append(e.process_blocks(25))
    for v, vc in enumerate(voices):
        if v % 10 == 4:
            e.set_param(vc["volume"], 0, 70.0)
        if v % 6 == 0:
            e.set_param(vc["fx"][0], 0, 1.0)
        if v % 9 == 0:
            e.sampler_stop(vc["sampler"])
        if v % 9 == 3:
            e.sampler_play(vc["sampler"])
    outs.append(e.process_blocks(9))
    return np.concatenate(outs)


def scenario_rs_bank(e, n_voices=40, radix=8, src_frames=900, mixed=True, fmt=PLANAR_F32):
    """voices whose SOURCE is the SPEC resampler (varispeed / rate conversion) -> gain -> pan [-> width / clip] -> sum tree:
    looping and one-shot sources (one-shots run out inside the run), mono and stereo, ratios below and above 1, some starting
    paused; ratio changes, seeks and pause / resume tagged at later blocks; gain and width automation on top.  With `mixed`
    every third voice is an ordinary sampler voice under the same leaves."""
    from fwapi import INTERLEAVED_I16 as I16
    rng = np.random.default_rng(4242)
    ratios = [1.0, 44100.0 / 48000.0, 1.5, 0.37, 2.25, 0.999, 1.0 / 3.0, 3.7]
    voices, ends = [], []
    for v in range(n_voices):
        ch = 1 if v % 5 == 2 else 2
        data = voice_source(8100 + v, src_frames, ch)
        vfmt = I16 if (fmt != PLANAR_F32 and v % 2) else PLANAR_F32
        raw = np.round(data * 32767).astype(np.int16).T.copy() if vfmt == I16 else data
        smp = e.new_sample(vfmt, ch, raw)
        is_rs = not (mixed and v % 3 == 2)
        if is_rs:
            src = e.resampler(smp, ratios[v % len(ratios)], loop=(v % 4 != 1), playing=(v % 7 != 3), n_out=2)
        else:
            src = e.sampler(90.0)
        cur = src
        vol = e.volume(float(rng.uniform(20, 110)))
        e.connect_stereo(cur, vol)
        cur = vol
        pan = None
        if v % 2 == 0:
            pan = e.pan(float(rng.uniform(-1, 1)))
            e.connect_stereo(cur, pan)
            cur = pan
        fx = []
        for n in width_clip_fx(e, v, rng, limit=2):
            e.connect_stereo(cur, n)
            cur = n
            fx.append(n)
        voices.append(dict(src=src, is_rs=is_rs, volume=vol, pan=pan, fx=fx, sample=smp))
        ends.append(cur)
    level = ends
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.sum(len(grp))
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    e.connect_stereo(level[0], e.graph_out_node)
    e.update()
    for vc in voices:
        if not vc["is_rs"]:
            e.sampler_set_sample(vc["src"], vc["sample"])
            e.sampler_set_loop_range(vc["src"], LOOP_FULL)
            e.sampler_play(vc["src"])
    outs = [e.process_blocks(4)]
    for v, vc in enumerate(voices):
        if vc["is_rs"]:
            if v % 6 == 0:
                e.set_param(vc["src"], 1, [0.5, 1.25, 2.0][v % 3], at_block=1)     # ratio
            if v % 6 == 4:
                e.set_param(vc["src"], 4, float(v * 7 % src_frames), at_block=2)   # seek
            if v % 7 == 3:
                e.set_param(vc["src"], 3, 1.0, at_block=1)                         # the paused ones start
            if v % 9 == 5:
                e.set_param(vc["src"], 3, 0.0, at_block=0)                         # pause ...
                e.set_param(vc["src"], 3, 1.0, at_block=3)                         # ... and resume
        if v % 4 == 0:
            e.set_param(vc["volume"], 0, 35.0 + v, at_block=2)
        if v % 10 == 7:
            e.set_param(vc["volume"], 0, 0.0, at_block=1)                          # mute behind a resampler
    outs.append(e.process_blocks(5))
    outs.append(e.process_blocks(23))                                              # ramps settle, one-shots run out
    for v, vc in enumerate(voices):
        if v % 10 == 7:
            e.set_param(vc["volume"], 0, 80.0)
        if vc["is_rs"] and v % 4 == 1:
            e.set_param(vc["src"], 4, 0.0)                                         # restart a finished one-shot ...
            e.set_param(vc["src"], 3, 1.0)
    outs.append(e.process_blocks(9))
    return np.concatenate(outs)


class TaggedOracle(object):
    """Wraps an OracleEngine so that messages carry at_block like the GPU ABI: they are queued and delivered
    just before the tagged block of the next process_blocks call (the reference's rings are polled per block)."""

    def __init__(self, eng):
        self.e = eng
        self.q = []
        self.backend = "oracle"
        self.sample_rate = eng.sample_rate
        self.max_block_frames = eng.max_block_frames

    def __getattr__(self, name):
        return getattr(self.e, name)

    def _defer(self, at_block, fn, *a):
        # a message for block 0 is delivered right away — unless one carried over from the previous call (tagged beyond
        # its last block) is still waiting for block 0: messages reach a node in the order they were sent
        if at_block == 0 and not any(at == 0 for at, _, _ in self.q):
            fn(*a)
        else:
            self.q.append((at_block, fn, a))

    def set_param(self, node, param, value, at_block=0):
        self._defer(at_block, self.e.set_param, node, param, value)

    def sampler_set_sample(self, node, sample, stop_playback=False, at_block=0):
        self._defer(at_block, self.e.sampler_set_sample, node, sample, stop_playback)

    def sampler_play(self, node, at_block=0):
        self._defer(at_block, self.e.sampler_play, node)

    def sampler_pause(self, node, at_block=0):
        self._defer(at_block, self.e.sampler_pause, node)

    def sampler_stop(self, node, at_block=0):
        self._defer(at_block, self.e.sampler_stop, node)

    def sampler_set_playhead_secs(self, node, secs, at_block=0):
        self._defer(at_block, self.e.sampler_set_playhead_secs, node, secs)

    def sampler_set_loop_range(self, node, mode, start=0.0, end=0.0, at_block=0):
        self._defer(at_block, self.e.sampler_set_loop_range, node, mode, start, end)

    def process_blocks(self, k, n_out_ch=2):
        outs = []
        for b in range(k):
            keep = []
            for at, fn, a in self.q:
                if at == b:
                    fn(*a)
                elif at > b:
                    keep.append((at, fn, a))
            self.q = keep
            outs.append(self.e.process_blocks(1, n_out_ch))
        self.q = [(at - k, fn, a) for at, fn, a in self.q]
        return np.concatenate(outs)

    def process_interleaved(self, frames, *a, **kw):
        assert not self.q
        return self.e.process_interleaved(frames, *a, **kw)


def scenario_hybrid_sends(e, n_a=26, n_b=11, src_frames=2100, seed=5, long_call=40):
    """a mixing-desk shape that is NOT a fused plan as a whole: two voice banks (A: sampler -> volume -> pan, two SumNodes, one
    with an empty slot; B: with width / hard-clip stages, resampler sources among them) whose buses are consumed TWICE — dry
    into the master sum and as a send into a delay -> biquad return — plus a resampler -> spatialiser source, a bank whose
    SumNode also takes a non-voice input (stays on the level executor), a master volume.  The voice banks are rendered by the
    voice-bank kernels into their SumNodes' pool buffers (hybrid plan, kind 3), everything else by the level executor.
    Gain / pan / width changes, pauses and restarts tagged across a long call, a master fade."""
    rng = np.random.default_rng(seed)

    def voice(v, fx):
        ch = 1 if v % 6 == 2 else 2
        data = voice_source(seed * 1000 + v, src_frames + 13 * v, ch)
        smp = e.new_sample(PLANAR_F32, ch, data)
        rs = fx and v % 4 == 1
        s = e.resampler(smp, float(rng.uniform(0.6, 1.7)), loop=True, n_out=2) if rs else e.sampler(100.0)
        vol = e.volume(float(rng.uniform(20, 100)))
        pan = e.pan(float(rng.uniform(-1, 1)))
        e.connect_stereo(s, vol)
        e.connect_stereo(vol, pan)
        cur = pan
        extra = []
        if fx:
            for n in width_clip_fx(e, v, rng, limit=2):
                e.connect_stereo(cur, n)
                cur = n
                extra.append(n)
        return dict(src=s, rs=rs, smp=smp, volume=vol, pan=pan, fx=extra, end=cur)

    a = [voice(v, False) for v in range(n_a)]
    b = [voice(100 + v, True) for v in range(n_b)]
    half = n_a // 2
    sum_a1 = e.sum(half + 1)              # one empty voice slot (the last port)
    sum_a2 = e.sum(n_a - half)
    sum_b = e.sum(n_b)
    for p, vc in enumerate(a[:half]):
        e.connect_stereo(vc["end"], sum_a1, 2 * p)
    for p, vc in enumerate(a[half:]):
        e.connect_stereo(vc["end"], sum_a2, 2 * p)
    for p, vc in enumerate(b):
        e.connect_stereo(vc["end"], sum_b, 2 * p)
    # a bank the voice-bank kernels cannot take: its SumNode also sums a bus
    c = [voice(200 + v, False) for v in range(9)]
    sum_c = e.sum(10)
    for p, vc in enumerate(c):
        e.connect_stereo(vc["end"], sum_c, 2 * p)
    e.connect_stereo(sum_a2, sum_c, 18)    # (sum_a2 is consumed here, by the master and by the send)
    # send / return
    send = e.sum(2)
    e.connect_stereo(sum_a2, send, 0)
    e.connect_stereo(sum_b, send, 2)
    dl = e.delay(0.011, 0.35, 1.0)
    bq = e.biquad(0, 2500.0)
    e.connect_stereo(send, dl)
    e.connect_stereo(dl, bq)
    # a moving source
    sp_smp = e.new_sample(PLANAR_F32, 1, voice_source(seed * 1000 + 999, 1700, 1))
    rs = e.resampler(sp_smp, 0.93, loop=True, n_out=1)
    sp = e.spatial(1.5, 0.2, -2.0, n_in=1)
    e.connect(rs, 0, sp, 0)
    master = e.sum(6)
    for p, n in enumerate([sum_a1, sum_a2, sum_b, sum_c, bq, sp]):
        e.connect_stereo(n, master, 2 * p)
    mvol = e.volume(80.0)
    e.connect_stereo(master, mvol)
    e.connect_stereo(mvol, e.graph_out_node)
    e.update()
    for vc in a + b + c:
        if not vc["rs"]:
            e.sampler_set_sample(vc["src"], vc["smp"])
            e.sampler_set_loop_range(vc["src"], LOOP_FULL)
            e.sampler_play(vc["src"])
    outs = [e.process_blocks(3)]
    for vc in (a + b + c)[::3]:
        e.set_param(vc["volume"], 0, float(rng.uniform(0, 100)), at_block=int(rng.integers(0, long_call)))
    for vc in (a + b)[1::4]:
        e.set_param(vc["pan"], 0, float(rng.uniform(-1, 1)), at_block=int(rng.integers(0, long_call)))
    for vc in a[2::5]:
        e.sampler_pause(vc["src"], at_block=int(rng.integers(0, long_call // 2)))
        e.sampler_play(vc["src"], at_block=int(rng.integers(long_call // 2, long_call)))
    for vc in b:
        if vc["rs"]:
            e.set_param(vc["src"], 1, float(rng.uniform(0.5, 2.0)), at_block=int(rng.integers(0, long_call)))
    e.set_param(mvol, 0, 35.0, at_block=long_call // 3)
    e.set_param(sp, 0, -2.0, at_block=4)
    outs.append(e.process_blocks(long_call))
    e.set_param(dl, 1, 0.1)
    outs.append(e.process_blocks(7))
    outs.append(e.process_blocks(2))
    return np.concatenate(outs)


def scenario_bus_iir(e, seed=31, src_frames=5000):
    """filters and delays on buses, long calls: a stereo master chain low-pass -> delay (960 frames) -> high-pass behind a small
    voice bank, and a mono strip sampler -> biquad -> delay (700 frames) -> biquad -> mono-to-stereo beside it.  In a batch of
    K >= 2 whole 256-frame chunks such nodes are walked by k_bus_iir over all K blocks (state in registers, next chunk
    prefetched); a coefficient / feedback message inside a batch sends that batch through the block-by-block path; a short delay
    (300 frames) never takes the walker."""
    rng = np.random.default_rng(seed)
    voices = []
    ends = []
    for v in range(9):
        s = e.sampler(100.0)
        vol = e.volume(float(rng.uniform(30, 100)))
        e.connect_stereo(s, vol)
        voices.append((s, vol))
        ends.append(vol)
    bank = e.sum(9)
    for p, n in enumerate(ends):
        e.connect_stereo(n, bank, 2 * p)
    lp = e.biquad(0, 3000.0, 0.8)
    dl = e.delay(960.0 / e.sample_rate, feedback=0.35, mix=0.4)
    hp = e.biquad(1, 200.0, 0.7)
    short = e.delay(300.0 / e.sample_rate, feedback=0.2, mix=0.5)
    for a, b in ((bank, lp), (lp, dl), (dl, hp), (hp, short)):
        e.connect_stereo(a, b)
    ms = e.sampler(80.0, n_out=1)
    mbq = e.biquad(2, 1200.0, 2.0, ch=1)
    mdl = e.delay(700.0 / e.sample_rate, feedback=0.5, mix=0.6, ch=1)
    mbq2 = e.biquad(0, 5000.0, 0.7, ch=1)
    m2s = e.add_node(MONO_TO_STEREO, 1, 2)
    e.connect(ms, 0, mbq, 0)
    e.connect(mbq, 0, mdl, 0)
    e.connect(mdl, 0, mbq2, 0)
    e.connect(mbq2, 0, m2s, 0)
    mix = e.sum(2)
    e.connect_stereo(short, mix, 0)
    e.connect_stereo(m2s, mix, 2)
    e.connect_stereo(mix, e.graph_out_node)
    e.update()
    for v, (s, vol) in enumerate(voices):
        e.sampler_set_sample(s, e.new_sample(PLANAR_F32, 2, voice_source(seed * 1000 + v, src_frames + 29 * v, 2)))
        e.sampler_set_loop_range(s, LOOP_FULL)
        e.sampler_play(s)
    e.sampler_set_sample(ms, e.new_sample(PLANAR_F32, 1, voice_source(seed * 1000 + 77, src_frames, 1)))
    e.sampler_set_loop_range(ms, LOOP_FULL)
    e.sampler_play(ms)
    outs = [e.process_blocks(1), e.process_blocks(7)]          # one block (no walker), then a batch
    e.set_param(lp, 1, 800.0, at_block=3)                      # a filter sweep inside the next batch: block by block
    e.set_param(mdl, 1, 0.2, at_block=1)
    outs.append(e.process_blocks(6))
    outs.append(e.process_blocks(9))                           # walkers again, from the state the other path left
    e.set_param(voices[2][1], 0, 10.0, at_block=2)             # a voice message does not concern the bus nodes
    outs.append(e.process_blocks(5))
    return np.concatenate(outs)


def scenario_split_mixers(e, seed=21, long_call=30, src_frames=1300):
    """mixers that take voices on their leading ports AND buses behind them (the usual master section: voices + a reverb return
    + a sub-mix): the voice-bank kernels sum the leading voice ports into a partial bus — the reference's accumulator at that
    point — and the SumNode itself continues on the level executor with (partial, the rest) on the path of its FULL port count
    (Q13: 2 / 3 / 4 ports add silent inputs, any other count skips them).  Pauses make ports silent on both sides of the split
    (whole leading groups too), a 5-port mixer and a 12-port one, a voice port BEHIND the bus (stays on the levels), a null
    slot among the leading ports, a mono-to-stereo detour as the non-voice input, a send off the first mixer."""
    rng = np.random.default_rng(seed)

    def voice(i, fx=False):
        ch = 1 if i % 5 == 1 else 2
        smp = e.new_sample(PLANAR_F32, ch, voice_source(seed * 1000 + i, src_frames + 17 * i, ch))
        s = e.sampler(100.0)
        vol = e.volume(float(rng.uniform(20, 100)))
        e.connect_stereo(s, vol)
        cur = vol
        if fx:
            hc = e.hard_clip(-4.0)
            e.connect_stereo(cur, hc)
            cur = hc
        return dict(sampler=s, smp=smp, volume=vol, end=cur)

    # mixer A: 4 leading voices (one port of them empty), then a bus, = 6 ports... leading: v0 v1 - v2, then detour, then a voice
    va = [voice(i) for i in range(4)]
    side = e.sampler(60.0)
    side_smp = e.new_sample(PLANAR_F32, 2, voice_source(seed * 1000 + 500, 900, 2))
    s2m = e.add_node(STEREO_TO_MONO, 2, 1)
    m2s = e.add_node(MONO_TO_STEREO, 1, 2)
    e.connect_stereo(side, s2m)
    e.connect(s2m, 0, m2s, 0)
    mix_a = e.sum(7)
    for p, vc in zip((0, 1, 3), va[:3]):      # port 2 stays empty: a null voice among the leading ports
        e.connect_stereo(vc["end"], mix_a, 2 * p)
    e.connect_stereo(m2s, mix_a, 8)            # port 4: the bus
    e.connect_stereo(va[3]["end"], mix_a, 10)  # port 5: a voice BEHIND the bus (level executor); port 6 empty
    # mixer B: 11 leading voices (some with a hard clip), then mixer A's bus as port 11  (12 ports)
    vb = [voice(10 + i, fx=(i % 4 == 2)) for i in range(11)]
    mix_b = e.sum(12)
    for p, vc in enumerate(vb):
        e.connect_stereo(vc["end"], mix_b, 2 * p)
    e.connect_stereo(mix_a, mix_b, 22)
    # a 3-port mixer with a bus (never split) and the master
    vc3 = [voice(30 + i) for i in range(2)]
    mix_c = e.sum(3)
    for p, vc in enumerate(vc3):
        e.connect_stereo(vc["end"], mix_c, 2 * p)
    e.connect_stereo(mix_a, mix_c, 4)          # mixer A's bus is consumed twice
    master = e.sum(2)
    e.connect_stereo(mix_b, master, 0)
    e.connect_stereo(mix_c, master, 2)
    e.connect_stereo(master, e.graph_out_node)
    e.update()
    allv = va + vb + vc3
    for vc in allv:
        e.sampler_set_sample(vc["sampler"], vc["smp"])
        e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        e.sampler_play(vc["sampler"])
    e.sampler_set_sample(side, side_smp)
    e.sampler_set_loop_range(side, LOOP_FULL)
    e.sampler_play(side)
    outs = [e.process_blocks(3)]
    # silence on both sides of the splits: every leading voice of mixer A paused for a while, the bus paused, single voices
    for vc in va[:3]:
        e.sampler_pause(vc["sampler"], at_block=4)
        e.sampler_play(vc["sampler"], at_block=11)
    e.sampler_pause(side, at_block=8)
    e.sampler_play(side, at_block=19)
    for vc in vb[::3]:
        e.sampler_pause(vc["sampler"], at_block=int(rng.integers(0, long_call // 2)))
        e.sampler_play(vc["sampler"], at_block=int(rng.integers(long_call // 2, long_call)))
    for vc in allv[1::4]:
        e.set_param(vc["volume"], 0, float(rng.choice([0.0, 30.0, 110.0])), at_block=int(rng.integers(0, long_call)))
    outs.append(e.process_blocks(long_call))
    for vc in vb:
        e.sampler_pause(vc["sampler"])         # the whole leading group of mixer B silent: the partial bus is flagged silent
    outs.append(e.process_blocks(4))
    for vc in vb[:5]:
        e.sampler_play(vc["sampler"], at_block=1)
    outs.append(e.process_blocks(5))
    return np.concatenate(outs)


def scenario_hybrid_chain_sends(e, n_voices=30, radix=6, src_frames=2300, seed=9, long_call=50):
    """banks of filtered voices (sampler -> biquad -> delay -> volume [-> pan]; some dry, one bank with a hard clip in a voice,
    which the chain plan does not take) whose buses are consumed twice: into the master sum and into a send -> width return.
    The banks the chain plan can take are rendered by k_chain into their SumNodes' pool buffers, the rest by the level executor
    (hybrid plan with chain banks).  Filter sweeps, delay feedback changes, gain glides, pauses across a long call."""
    rng = np.random.default_rng(seed)
    voices = []
    for v in range(n_voices):
        s = e.sampler(100.0)
        cur = s
        bq = dl = None
        if v % 5 != 4:
            bq = e.biquad(int(v % 3), float(rng.uniform(300, 6000)), float(rng.uniform(0.5, 2.0)))
            e.connect_stereo(cur, bq)
            cur = bq
        if v % 4 != 3:
            dl = e.delay(int(rng.integers(70, 700)) / float(e.sample_rate), feedback=float(rng.uniform(0, 0.5)), mix=float(rng.uniform(0.2, 0.9)))
            e.connect_stereo(cur, dl)
            cur = dl
        vol = e.volume(float(rng.uniform(20, 100)))
        e.connect_stereo(cur, vol)
        cur = vol
        if v % 2:
            pan = e.pan(float(rng.uniform(-1, 1)))
            e.connect_stereo(cur, pan)
            cur = pan
        if v == 2 * radix + 1:  # one voice of the third bank ends in a hard clip: that bank stays on the level executor
            hc = e.hard_clip(-6.0)
            e.connect_stereo(cur, hc)
            cur = hc
        voices.append(dict(sampler=s, bq=bq, dl=dl, volume=vol, end=cur))
    banks = []
    for i in range(0, n_voices, radix):
        grp = voices[i:i + radix]
        m = e.sum(len(grp))
        for p, vc in enumerate(grp):
            e.connect_stereo(vc["end"], m, 2 * p)
        banks.append(m)
    send = e.sum(2)
    e.connect_stereo(banks[0], send, 0)
    e.connect_stereo(banks[1], send, 2)
    wid = e.width(1.6)
    e.connect_stereo(send, wid)
    master = e.sum(len(banks) + 1)
    for p, m in enumerate(banks + [wid]):
        e.connect_stereo(m, master, 2 * p)
    e.connect_stereo(master, e.graph_out_node)
    e.update()
    for v, vc in enumerate(voices):
        smp = e.new_sample(PLANAR_F32, 1 if v % 7 == 3 else 2, voice_source(seed * 1000 + v, src_frames + 11 * v, 1 if v % 7 == 3 else 2))
        e.sampler_set_sample(vc["sampler"], smp)
        e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        e.sampler_play(vc["sampler"])
    outs = [e.process_blocks(4)]
    for vc in voices[::3]:
        e.set_param(vc["volume"], 0, float(rng.uniform(0, 100)), at_block=int(rng.integers(0, long_call)))
    for vc in voices[1::4]:
        if vc["bq"] is not None:
            e.set_param(vc["bq"], 1, float(rng.uniform(200, 9000)), at_block=int(rng.integers(0, long_call)))
        if vc["dl"] is not None:
            e.set_param(vc["dl"], 1, float(rng.uniform(0, 0.8)), at_block=int(rng.integers(0, long_call)))
    for vc in voices[2::6]:
        e.sampler_pause(vc["sampler"], at_block=int(rng.integers(0, long_call // 2)))
        e.sampler_play(vc["sampler"], at_block=int(rng.integers(long_call // 2, long_call)))
    e.set_param(wid, 0, 0.4, at_block=7)
    outs.append(e.process_blocks(long_call))
    outs.append(e.process_blocks(9))
    outs.append(e.process_blocks(9))
    return np.concatenate(outs)


def scenario_mixed_generic(e, use_beep=True):
    """a graph the fused plan does not cover: beep (or a mono sampler) + sampler through clip / mono<->stereo /
    2,3,4-port sums, dangling ports, one-to-many edges.  A disabled BeepTest with consumers is outside the
    parity domain (Q12: its channel 0 exposes whatever the reference's reused buffer held), so the beep stays on."""
    s = e.sampler(80.0)
    data = voice_source(77, 2000)
    if use_beep:
        beep = e.beep(440.0, -12.0, True, n_out=1)
    else:
        beep = e.sampler(30.0, n_out=1)
    m2s = e.add_node(MONO_TO_STEREO, 1, 2)
    clip = e.hard_clip(-9.0)
    s2m = e.add_node(STEREO_TO_MONO, 2, 1)
    m2s2 = e.add_node(MONO_TO_STEREO, 1, 2)
    vol3 = e.volume(70.0, ch=3)                  # generic (non-stereo) volume path with one dangling input
    sum2 = e.sum(2)
    sum3 = e.sum(3)
    sum4 = e.sum(4)
    e.connect(beep, 0, m2s, 0)
    e.connect_stereo(s, clip)
    e.connect_stereo(clip, s2m)
    e.connect(s2m, 0, m2s2, 0)
    e.connect_stereo(m2s, sum2, 0)
    e.connect_stereo(m2s2, sum2, 2)
    e.connect(s, 0, vol3, 0)                     # one-to-many from the sampler
    e.connect(s, 1, vol3, 2)                     # vol3 input 1 unconnected (should_clear)
    e.connect_stereo(sum2, sum3, 0)
    e.connect(vol3, 0, sum3, 2)
    e.connect(vol3, 2, sum3, 3)                  # port 2 (inputs 4,5) unconnected
    e.connect_stereo(sum3, sum4, 0)
    e.connect_stereo(clip, sum4, 4)              # ports 1 and 3 unconnected
    e.connect_stereo(sum4, e.graph_out_node)
    e.update()
    smp = e.new_sample(PLANAR_F32, 2, data)
    e.sampler_set_sample(s, smp)
    e.sampler_set_loop_range(s, LOOP_FULL)
    e.sampler_play(s)
    if not use_beep:
        e.sampler_set_sample(beep, e.new_sample(PLANAR_F32, 1, voice_source(78, 1500, 1)))
        e.sampler_play(beep)                     # one-shot: ends inside the run
    out1 = e.process_blocks(4)
    e.set_param(vol3, 0, 20.0)
    out2 = e.process_blocks(4)
    return np.concatenate([out1, out2])


def scenario_graph_inputs(e):
    """graph with stream inputs: in(2) -> volume -> out, plus an extra graph input left unconnected upstream."""
    vol = e.volume(60.0)
    e.connect_stereo(e.graph_in_node, vol)
    e.connect_stereo(vol, e.graph_out_node)
    e.update()
    mbf = e.max_block_frames
    frames = 3 * mbf + 17            # last sub-block is partial (processor.rs:95-96)
    inp = fwapi.xorshift_uniform(5150, frames * 2)
    out = e.process_interleaved(frames, 2, inp=inp, n_in_ch=2)
    e.set_param(vol, 0, 10.0)
    out2 = e.process_interleaved(frames, 2, inp=inp, n_in_ch=2)
    return np.concatenate([out, out2])


def scenario_cfg3_chain(e, n_voices=12, blocks=10, radix=4, src_frames=3000):
    """config-3 shape: V x (sampler -> biquad LPF -> delay -> gain) -> sum tree, plus a width node on
    the bus.  SPEC nodes: runs on the generic executor."""
    rng = np.random.default_rng(77)
    ends, voices = [], []
    for v in range(n_voices):
        s = e.sampler(100.0)
        bq = e.biquad(v % 3, float(rng.uniform(200, 8000)), 0.707 if v % 2 else 2.5)
        dl = e.delay(float(rng.uniform(0.0002, 0.02)), feedback=0.0 if v % 3 == 0 else 0.4, mix=0.5)
        vol = e.volume(float(rng.uniform(20, 100)))
        e.connect_stereo(s, bq)
        e.connect_stereo(bq, dl)
        e.connect_stereo(dl, vol)
        ends.append(vol)
        voices.append(dict(sampler=s, biquad=bq, delay=dl, volume=vol))
    level = ends
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.sum(len(grp))
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    w = e.width(1.6)
    e.connect_stereo(level[0], w)
    e.connect_stereo(w, e.graph_out_node)
    e.update()
    for v, vc in enumerate(voices):
        e.sampler_set_sample(vc["sampler"], e.new_sample(PLANAR_F32, 2, voice_source(900 + v, src_frames)))
        e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        e.sampler_play(vc["sampler"])
    out1 = e.process_blocks(blocks // 2)
    e.set_param(w, 0, 0.3)                       # width ramps
    e.set_param(voices[1]["biquad"], 1, 500.0)   # cutoff change -> new coefficients at the next block
    e.set_param(voices[2]["delay"], 1, 0.7)      # feedback
    e.set_param(voices[2]["delay"], 2, 0.9)      # mix
    e.sampler_pause(voices[0]["sampler"])        # the filter/delay tails keep ringing on zeros
    out2 = e.process_blocks(blocks - blocks // 2)
    return np.concatenate([out1, out2])


def build_chain_bank(e, n_voices, radix=32, src_frames=3000, biquad=True, delay=True, with_pan=False, seed=0,
                     fmt=PLANAR_F32, mono_every=0, min_delay_frames=64, max_delay_frames=900, first_delay_frames=64,
                     master=()):
    """config-3 shape: V x (sampler -> biquad LPF -> delay -> gain [-> pan]) -> radix sum tree -> out.
    The shape the fused chain plan (k_chain) accepts; delays are >= one 64-frame tile."""
    rng = np.random.default_rng(4321 + seed)
    voices, ends = [], []
    for v in range(n_voices):
        s = e.sampler(100.0)
        cur = s
        bq = dl = pan = None
        if biquad:
            bq = e.biquad(v % 3, float(rng.uniform(200, 8000)), 0.707 if v % 2 else 2.5)
            e.connect_stereo(cur, bq)
            cur = bq
        if delay:
            d_frames = first_delay_frames if v == 0 else int(rng.integers(min_delay_frames, max_delay_frames))
            dl = e.delay(d_frames / float(e.sample_rate), feedback=0.0 if v % 3 == 0 else 0.45, mix=0.5)
            e.connect_stereo(cur, dl)
            cur = dl
        vol = e.volume(float(rng.uniform(20, 100)))
        e.connect_stereo(cur, vol)
        cur = vol
        if with_pan:
            pan = e.pan(float(rng.uniform(-1, 1)))
            e.connect_stereo(cur, pan)
            cur = pan
        voices.append(dict(sampler=s, biquad=bq, delay=dl, volume=vol, pan=pan))
        ends.append(cur)
    level = ends
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.sum(len(grp))
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    voices[0]["master"] = connect_through_master(e, level[0], master)
    e.update()
    for v, vc in enumerate(voices):
        ch = 1 if (mono_every and v % mono_every == 0) else 2
        data = voice_source(seed * 100000 + 5000 + v, src_frames, ch)
        if fmt in (PLANAR_I16, INTERLEAVED_I16):
            raw = np.round(data * 32767).astype(np.int16)
        else:
            raw = data
        if fmt <= INTERLEAVED_F32:
            raw = raw.T.copy()
        vc["sample"] = e.new_sample(fmt, ch, raw)
        e.sampler_set_sample(vc["sampler"], vc["sample"])
    return voices


def scenario_chain_steady(e, n_voices=40, blocks=6, radix=32, **kw):
    voices = build_chain_bank(e, n_voices, radix=radix, **kw)
    for vc in voices:
        e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        e.sampler_play(vc["sampler"])
    return e.process_blocks(blocks)


def scenario_chain_events(e, n_voices=37, radix=32, src_frames=1000, **kw):
    """the chain plan under everything the control plane can do: loop wraps inside blocks, one-shot ends (the
    filter / delay tails keep ringing on zeros), voices that never start, gain ramps that settle and stall, a
    mute after the delay (silent port, tails still advance), coefficient and feedback/mix changes tagged at later
    blocks of a multi-block call, mono sources."""
    voices = build_chain_bank(e, n_voices, radix=radix, src_frames=src_frames, mono_every=7, **kw)
    outs = []
    for v, vc in enumerate(voices):
        if v % 5 != 3:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)   # v%5==3: one-shot, ends inside the run
        if v % 4 != 1:
            e.sampler_play(vc["sampler"])                         # v%4==1: started late
    outs.append(e.process_blocks(3))
    for v, vc in enumerate(voices):
        if v % 3 == 0:
            e.set_param(vc["volume"], 0, 25.0 if v % 2 else 90.0, at_block=0)
        if v % 6 == 2 and vc["biquad"] is not None:
            e.set_param(vc["biquad"], 1, 700.0 + 10 * v, at_block=2)       # cutoff -> new coefficients at block 2
        if v % 6 == 4 and vc["delay"] is not None:
            e.set_param(vc["delay"], 1, 0.8, at_block=1)                   # feedback
            e.set_param(vc["delay"], 2, 0.9, at_block=3)                   # mix (and dry)
        if v % 10 == 4:
            e.set_param(vc["sampler"], 0, 0.0, at_block=1)                 # sampler gain ramps to 0, then mutes
        if v % 10 == 7:
            e.set_param(vc["volume"], 0, 0.0, at_block=1)                  # post-delay mute: silent port
        if v % 4 == 1:
            e.sampler_play(vc["sampler"], at_block=1)
        if v % 11 == 5:
            e.sampler_pause(vc["sampler"], at_block=3)
        if vc["pan"] is not None and v % 5 == 0:
            e.set_param(vc["pan"], 0, -0.5, at_block=2)
    outs.append(e.process_blocks(5))
    outs.append(e.process_blocks(24))
    for v, vc in enumerate(voices):
        if v % 9 == 0:
            e.sampler_stop(vc["sampler"])
        if v % 9 == 1:
            e.sampler_set_playhead_secs(vc["sampler"], 300.25 / e.sample_rate)
        if v % 9 == 3:
            e.sampler_play(vc["sampler"])
        if v % 10 == 7:
            e.set_param(vc["volume"], 0, 60.0)
        if v % 3 == 0:
            e.set_param(vc["volume"], 0, 100.0)
        if vc["biquad"] is not None and v % 8 == 1:
            e.set_param(vc["biquad"], 2, 3.0)                              # Q
    outs.append(e.process_blocks(7))
    return np.concatenate(outs)


def scenario_chain_steady_calls(e, n_voices=37, tile=128, with_pan=False, src_extra=0):
    """calls k_chain's steady-call loop accepts (every voice one descriptor shape for the whole call, delays >= 3
    tiles, no message pending) between calls it does not: start-up messages, a burst of pauses / gain changes / a
    mute whose ramps take a few blocks to settle.  Sources are a whole number of blocks long (loops wrap on block
    boundaries), delay lengths are mostly not multiples of 4 (quads straddle the ring end once per lap), some voices
    never start (cleared source through biquad + delay), some samples are mono."""
    mbf = e.max_block_frames
    # src_extra != 0: the loops wrap INSIDE blocks (a different frame for every lap), still a steady call
    voices = build_chain_bank(e, n_voices, radix=32, src_frames=6 * mbf + src_extra, mono_every=5, with_pan=with_pan,
                              first_delay_frames=3 * tile, min_delay_frames=3 * tile + 1, max_delay_frames=3 * tile + 500)
    outs = []
    for v, vc in enumerate(voices):
        e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        if v % 6 != 4:
            e.sampler_play(vc["sampler"])           # v%6==4 never starts
    outs.append(e.process_blocks(2))                # start-up messages: general loop
    outs.append(e.process_blocks(7))                # steady
    outs.append(e.process_blocks(1))                # steady, one block
    for v, vc in enumerate(voices):
        if v % 7 == 3:
            e.sampler_pause(vc["sampler"])
        if v % 4 == 0:
            e.set_param(vc["volume"], 0, 35.0 + v)
        if v % 9 == 5:
            e.set_param(vc["volume"], 0, 0.0)       # ramps to 0, then the port is muted
    outs.append(e.process_blocks(3))                # messages
    # a 10 ms smoother needs ~10 time constants (4 800 frames) to come within settle_epsilon of its target
    outs.append(e.process_blocks((5200 + mbf - 1) // mbf))
    outs.append(e.process_blocks(11))               # steady again, with paused and muted voices
    outs.append(e.process_blocks(5))
    return np.concatenate(outs)


def scenario_master_chain(e, chain=False, n_voices=45):
    """the usual application shape: a voice bank under a sum tree, then a master chain on the mix bus (filter, delay,
    master volume, limiter) before graph_out — with automation on the master volume and the delay mix while voices
    start, pause and change gain underneath."""
    master = (lambda e: e.biquad(0, 9000.0, 0.707), lambda e: e.delay(333 / float(e.sample_rate), feedback=0.25, mix=0.2),
              lambda e: e.volume(80.0), lambda e: e.hard_clip(-3.0))
    if chain:
        voices = build_chain_bank(e, n_voices, radix=32, src_frames=1300, mono_every=6, master=master)
    else:
        voices = build_voice_bank(e, n_voices, radix=8, src_frames=1300, mono_every=6, master=master)
    m_bq, m_dl, m_vol, m_clip = voices[0]["master"]
    outs = []
    for v, vc in enumerate(voices):
        e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        if v % 5 != 2:
            e.sampler_play(vc["sampler"])
    outs.append(e.process_blocks(3))
    outs.append(e.process_blocks(4))                  # steady
    e.set_param(m_vol, 0, 140.0, at_block=1)          # master volume up: the limiter starts to clip
    e.set_param(m_dl, 2, 0.6, at_block=2)             # master delay mix
    for v, vc in enumerate(voices):
        if v % 5 == 2:
            e.sampler_play(vc["sampler"], at_block=1)
        if v % 7 == 3:
            e.set_param(vc["volume"], 0, 15.0, at_block=2)
    outs.append(e.process_blocks(6))
    outs.append(e.process_blocks(9))
    e.set_param(m_vol, 0, 0.0)                        # master fader down: ramps to silence, then a muted bus
    outs.append(e.process_blocks(30))
    e.set_param(m_vol, 0, 70.0)
    outs.append(e.process_blocks(5))
    return np.concatenate(outs)


def scenario_spatial_scene(e, n_sources=7, blocks=12, src_frames=2500):
    """moving sources: resampler (varispeed, some looping, one i16, one mono-to-stereo) -> spatialiser -> sum -> out.
    Exercises the 32.32 position arithmetic (loop wrap, one-shot end), the ITD history across blocks, gain ramps
    from position changes, ratio changes and seeks tagged at later blocks.  SPEC nodes: generic executor."""
    from fwapi import RESAMPLER, SPATIAL
    rng = np.random.default_rng(31)
    m = e.sum(n_sources + 1)
    srcs = []
    for v in range(n_sources):
        ch = 1 if v % 3 else 2
        data = voice_source(7000 + v, src_frames, ch)
        if v == 2:
            smp = e.new_sample(PLANAR_I16, ch, np.round(data * 32767).astype(np.int16))
        else:
            smp = e.new_sample(PLANAR_F32, ch, data)
        ratio = [1.0, 44100.0 / 48000.0, 1.5, 0.37, 2.25, 0.999, 1.0 / 3.0][v % 7]
        rs = e.resampler(smp, ratio, loop=(v % 2 == 0), n_out=ch)
        sp = e.spatial(float(rng.uniform(-5, 5)), float(rng.uniform(-1, 1)), float(rng.uniform(-5, 5)), n_in=ch)
        for c in range(ch):
            e.connect(rs, c, sp, c)
        e.connect_stereo(sp, m, 2 * v)
        srcs.append(dict(rs=rs, sp=sp))
    direct = e.resampler(e.new_sample(PLANAR_F32, 1, voice_source(7100, 900, 1)), 0.8, loop=True, n_out=2)  # mono -> L,R
    e.connect_stereo(direct, m, 2 * n_sources)
    e.connect_stereo(m, e.graph_out_node)
    e.update()
    out1 = e.process_blocks(blocks // 3)
    e.set_param(srcs[0]["sp"], 0, 3.0)                     # move right: gains ramp, ITD switches at the block start
    e.set_param(srcs[0]["sp"], 2, 0.5, at_block=1)
    e.set_param(srcs[1]["sp"], 0, -0.2, at_block=2)
    e.set_param(srcs[1]["rs"], 1, 0.5, at_block=1)         # ratio change
    e.set_param(srcs[3]["rs"], 4, 100.0, at_block=2)       # seek
    e.set_param(srcs[-1]["rs"], 3, 0.0, at_block=1)        # pause ...
    e.set_param(srcs[-1]["rs"], 3, 1.0, at_block=3)        # ... and resume
    out2 = e.process_blocks(blocks - blocks // 3)
    return np.concatenate([out1, out2])


def scenario_spatial_steady(e, n_sources=9, src_frames=4000, calls=(8, 70, 20, 20, 5)):
    """sampler -> spatialiser voices that sit still for whole calls (the generic executor runs a resting spatialiser's blocks in
    parallel: the ITD history of block b is the tail of block b-1's input), moved once between the first two calls so that the
    gain glides start, settle inside the long call and the following calls are at rest again; one source is mono, one is
    paused and resumed (an all-zero history)."""
    rng = np.random.default_rng(77)
    m = e.sum(n_sources)
    nodes = []
    for v in range(n_sources):
        ch = 1 if v == 3 else 2
        s = e.sampler(100.0)
        sp = e.spatial(float(rng.uniform(-5, 5)), float(rng.uniform(-1, 1)), float(rng.uniform(-5, 5)), n_in=2)
        e.connect_stereo(s, sp)
        e.connect_stereo(sp, m, 2 * v)
        nodes.append((s, sp, ch))
    e.connect_stereo(m, e.graph_out_node)
    e.update()
    for v, (s, sp, ch) in enumerate(nodes):
        smp = e.new_sample(PLANAR_F32, ch, voice_source(8100 + v, src_frames + 37 * v, ch))
        e.sampler_set_sample(s, smp)
        e.sampler_set_loop_range(s, LOOP_FULL)
        e.sampler_play(s)
    outs = [e.process_blocks(calls[0])]
    e.set_param(nodes[0][1], 0, 4.0)
    e.set_param(nodes[1][1], 2, -0.3, at_block=2)
    e.sampler_pause(nodes[2][0], at_block=5)
    outs.append(e.process_blocks(calls[1]))
    e.sampler_play(nodes[2][0], at_block=3)
    for n in calls[2:]:
        outs.append(e.process_blocks(n))
    return np.concatenate(outs)


def scenario_spatial_bank(e, n_voices=23, radix=8, src_frames=1700):
    """sampler -> [volume] -> [pan | width | clip] -> SPATIALISER voices next to plain ones under radix-8 mixers — the voice-bank
    plan's spatialiser stage: ear delays and gains moved by position messages (glides + delay switches at block starts), voices
    that stop (one-shots end: the last 63 frames still sound, then cleared input with live smoothers), a voice muted upstream
    (its spatialiser is fed cleared zeros), pauses and seeks (the history is what WAS played, not what precedes the new
    position), mono samples, 16-bit sources, calls of 1 .. 40 blocks."""
    rng = np.random.default_rng(4242)
    voices, ends = [], []
    for v in range(n_voices):
        s = e.sampler(float(rng.uniform(60, 100)))
        cur = s
        vol = None
        if v % 3 != 1:
            vol = e.volume(float(rng.uniform(30, 100)))
            e.connect_stereo(cur, vol)
            cur = vol
        mid = None
        if v % 4 == 0:
            mid = e.pan(float(rng.uniform(-1, 1)))
        elif v % 4 == 2:
            mid = e.width(float(rng.uniform(0.2, 1.8)))
        elif v % 7 == 3:
            mid = e.hard_clip(-6.0)
        if mid is not None:
            e.connect_stereo(cur, mid)
            cur = mid
        sp = None
        if v % 5 != 4:  # every fifth voice stays dry
            sp = e.spatial(float(rng.uniform(-5, 5)), float(rng.uniform(-1, 1)), float(rng.uniform(-5, 5)), n_in=2)
            e.connect_stereo(cur, sp)
            cur = sp
        voices.append(dict(s=s, vol=vol, mid=mid, sp=sp))
        ends.append(cur)
    level = ends
    while True:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.sum(len(grp))
            for p, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * p)
            nxt.append(m)
        level = nxt
        if len(level) == 1:
            break
    e.connect_stereo(level[0], e.graph_out_node)
    e.update()
    for v, vc in enumerate(voices):
        ch = 1 if v % 6 == 5 else 2
        data = voice_source(9000 + v, src_frames + 41 * v, ch)
        if v % 8 == 3:
            smp = e.new_sample(PLANAR_I16, ch, np.round(data * 32767).astype(np.int16))
        else:
            smp = e.new_sample(PLANAR_F32, ch, data)
        e.sampler_set_sample(vc["s"], smp)
        if v % 3 != 2:
            e.sampler_set_loop_range(vc["s"], LOOP_FULL)  # (the others are one-shots: they end inside the run)
        if v != 7:
            e.sampler_play(vc["s"])
    outs = [e.process_blocks(2)]
    sps = [vc for vc in voices if vc["sp"] is not None]
    e.set_param(sps[0]["sp"], 0, 4.5)                       # hard right: both gains glide, the ear delays switch
    e.set_param(sps[1]["sp"], 0, -3.0, at_block=1)
    e.set_param(sps[2]["sp"], 2, 0.25, at_block=2)          # z: distance -> both gains
    e.sampler_play(voices[7]["s"], at_block=1)              # a late starter: zero history
    outs.append(e.process_blocks(5))
    outs.append(e.process_blocks(1))
    muted = next(vc for vc in sps if vc["vol"] is not None)
    e.set_param(muted["vol"], 0, 0.0)                       # fades out upstream: ramp, then zeros that are not flagged ...
    e.sampler_pause(sps[3]["s"], at_block=3)                # ... a paused source: cleared input, the tail of the history sounds
    outs.append(e.process_blocks(40))
    e.sampler_play(sps[3]["s"], at_block=1)
    e.sampler_set_playhead_secs(sps[4]["s"], 0.01, at_block=2)   # a seek: the history is the audio before the jump
    e.set_param(sps[0]["sp"], 0, -4.5, at_block=4)
    outs.append(e.process_blocks(9))
    e.set_param(muted["vol"], 0, 80.0)
    outs.append(e.process_blocks(3))
    outs.append(e.process_blocks(17))
    return np.concatenate(outs)


def reverb_ir(seed, taps, channels=2, decay=None):
    """exponentially decaying seeded noise, L1-normalised per channel."""
    decay = decay or taps / 4.0
    h = fwapi.xorshift_uniform(seed, channels * taps).reshape(channels, taps)
    h = h * np.exp(-np.arange(taps, dtype=np.float32) / np.float32(decay))[None, :]
    h = h / np.sum(np.abs(h), axis=1, keepdims=True)
    return h.astype(np.float32)


def scenario_cfg4_reverb(e, n_voices=6, taps=5000, blocks=5, shared_ir=True, ir_channels=2, fir_ch=2):
    """config-4 shape: V x (sampler -> FIR convolution) -> sum -> out.  Taps > FIR_SEG exercises the split-K order."""
    irs = []
    n_irs = 1 if shared_ir else 2
    for k in range(n_irs):
        irs.append(e.new_sample(PLANAR_F32, ir_channels, reverb_ir(4000 + k, taps - 37 * k, ir_channels)))
    m = e.sum(n_voices)
    voices = []
    for v in range(n_voices):
        s = e.sampler(80.0)
        f = e.fir(irs[v % n_irs], ch=fir_ch)
        e.connect_stereo(s, f)
        e.connect_stereo(f, m, 2 * v)
        voices.append(s)
    e.connect_stereo(m, e.graph_out_node)
    e.update()
    for v, s in enumerate(voices):
        e.sampler_set_sample(s, e.new_sample(PLANAR_F32, 2, voice_source(1200 + v, 1500)))
        if v % 2 == 0:
            e.sampler_set_loop_range(s, LOOP_FULL)
        e.sampler_play(s)
    return e.process_blocks(blocks)
