"""GPU tier (-m gpu): the resident realtime kernel (k_rt_persist, include/fwgpu.h `fwgpu_rt_resident_stats`) — one-block
callbacks on the voice-bank plan are served through a doorbell by a kernel that stays on the device; everything that is not a
steady callback ends it first; its watchdog ends it when nobody calls.  Every test compares with the oracle bit for bit: the
kernel runs the same device functions as the one-launch edge, so a difference means the hand-over lost or repeated a block."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import scenarios
from fwapi import LOOP_FULL, LOOP_RANGE_SECS, GpuEngine, OracleEngine
from test_gpu_parity import assert_bits_equal

pytestmark = pytest.mark.gpu
MBF = 256


def _bank(e, n_voices=96):
    voices = scenarios.build_voice_bank(e, n_voices, radix=32, src_frames=5000, mono_every=7)
    for v, vc in enumerate(voices):
        if v % 5 != 3:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)  # (v % 5 == 3: one-shots, they end inside the run)
        if v % 9 != 4:
            e.sampler_play(vc["sampler"])
    return voices


def _run(e, script):
    """script(e, voices, i) is called before callback i; returns the concatenated one-block callbacks"""
    voices = _bank(e)
    outs = []
    for i in range(script.n):
        script(e, voices, i)
        outs.append(np.asarray(e.process_interleaved(MBF)))
    return np.concatenate(outs)


class Steady:
    n = 40

    def __call__(self, e, voices, i):
        pass


class Traffic:
    """messages, a call of another size and a graph edit between the callbacks"""
    n = 60

    def __call__(self, e, voices, i):
        if i == 7:
            e.set_param(voices[2]["volume"], 0, 35.0)          # a glide: ~21 callbacks that are not steady for that voice
        if i == 19:
            e.sampler_pause(voices[5]["sampler"])
        if i == 23:
            e.sampler_play(voices[5]["sampler"])
        if i == 31:
            extra = e.process_interleaved(3 * MBF)              # not one block: the launch sequence, the resident kernel ended first
            self.extra = np.asarray(extra)
        if i == 40:
            e.set_param(voices[11]["pan"], 0, -0.6)
        if i == 48:                                             # a graph edit: a new plan, adopted by the next callback
            e.remove_node(voices[20]["pan"])                    # (its mixer port is left unconnected: silence from there)
            e.update()


def test_consecutive_callbacks_ride_the_doorbell_and_match_the_oracle():
    g, o = GpuEngine(max_block_frames=MBF), OracleEngine(max_block_frames=MBF)
    out_g, out_o = _run(g, Steady()), _run(o, Steady())
    assert g.cx.plan_kind() == 1
    assert_bits_equal(out_o, out_g, "steady callbacks")
    launches, doorbells = g.cx.rt_resident_stats()
    if os.environ.get("FWGPU_RT_PERSIST") == "0":
        assert (launches, doorbells) == (0, 0)
    else:
        # the one-shots end inside the run (a voice that needs its state machines is still a steady CALLBACK: no message on the
        # device); what may cut the run short is the watchdog, when the test machine stalls for its whole idle time
        # (the first callback carries the play / set-sample messages: an ordinary launch)
        assert launches >= 1 and Steady.n - 2 <= launches + doorbells <= Steady.n, (launches, doorbells)
        if int(os.environ.get("FWGPU_RT_IDLE_MS", "20")) >= 20:
            assert launches <= 3, (launches, doorbells)


def test_messages_other_sizes_and_edits_end_the_kernel_and_nothing_is_lost():
    sg, so = Traffic(), Traffic()
    g, o = GpuEngine(max_block_frames=MBF), OracleEngine(max_block_frames=MBF)
    out_g, out_o = _run(g, sg), _run(o, so)
    assert_bits_equal(out_o, out_g, "callbacks with traffic")
    assert_bits_equal(so.extra, sg.extra, "the three-block call in between")
    launches, doorbells = g.cx.rt_resident_stats()
    if os.environ.get("FWGPU_RT_PERSIST") != "0":
        assert launches >= 4 and doorbells >= 20, (launches, doorbells)   # ended and launched again around every event


@pytest.mark.parametrize("idle_ms,pause_ms", [(1, 0.7), (1, 1.0), (1, 1.4), (2, 2.0)])
def test_watchdog_ends_the_kernel_and_a_late_doorbell_loses_no_block(idle_ms, pause_ms):
    """pauses around the watchdog's length between callbacks: the kernel ends by itself, sometimes while the next doorbell is on its
    way (the host then finds alive == 0 with the completion flag unchanged and renders the block with an ordinary launch)"""
    code = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np
from fwapi import GpuEngine, OracleEngine
from test_rt_resident import _bank, MBF
from test_gpu_parity import assert_bits_equal
g, o = GpuEngine(max_block_frames=MBF), OracleEngine(max_block_frames=MBF)
_bank(g), _bank(o)
rng = np.random.default_rng(5)
outs_g, outs_o = [], []
for i in range(150):
    outs_g.append(np.asarray(g.process_interleaved(MBF)))
    outs_o.append(np.asarray(o.process_interleaved(MBF)))
    t = %f * 1e-3 * float(rng.uniform(0.8, 1.2)) if i %% 3 else 0.0
    end = time.perf_counter() + t
    while time.perf_counter() < end:
        pass
assert_bits_equal(np.concatenate(outs_o), np.concatenate(outs_g), "callbacks around the watchdog")
launches, doorbells = g.cx.rt_resident_stats()
assert launches >= %d and launches + doorbells <= 150, (launches, doorbells)
print("OK", launches, doorbells)
''' % (os.path.dirname(os.path.abspath(__file__)), pause_ms, 20 if pause_ms > 1.3 * idle_ms else 1)
    if os.environ.get("FWGPU_RT_PERSIST") == "0":
        pytest.skip("the resident kernel is switched off")
    env = dict(os.environ, FWGPU_RT_IDLE_MS=str(idle_ms))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_destroy_and_idle_leave_no_kernel_behind():
    g = GpuEngine(max_block_frames=MBF)
    _bank(g)
    for _ in range(5):
        g.process_interleaved(MBF)
    t0 = time.perf_counter()
    g.cx.synchronize()     # the ctx stream: the resident kernel runs on its own, this does not wait for it
    assert time.perf_counter() - t0 < 0.015
    g.cx.close()           # ends it (doorbell | quit) and waits
    import torch

    t0 = time.perf_counter()
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.010, "a resident kernel outlived its context"


# ------------------------------------------------------------------ more shapes through the resident kernel
def _varied_bank(e, n_voices, fx, src_frames):
    """loops of odd lengths (wraps at every offset inside a block), one-shots that end inside the run, mono samples, voices that
    never play (silent ports), a voice muted by a fade, loop ranges; a last leaf of 4 ports (the unmasked SumNode path)"""
    def voice_fx(e_, v, rng):
        if not fx:
            return []
        return [e_.width(float(rng.uniform(0.3, 1.7)))] if v % 3 == 0 else ([e_.hard_clip(-4.0)] if v % 3 == 1 else [])

    voices = scenarios.build_voice_bank(e, n_voices, radix=32, src_frames=src_frames, mono_every=5, voice_fx=voice_fx if fx else None)
    for v, vc in enumerate(voices):
        if v % 7 == 3:
            pass                                                   # a one-shot: ends after src_frames
        elif v % 7 == 5:
            e.sampler_set_loop_range(vc["sampler"], LOOP_RANGE_SECS, (13 + v) / 48000.0, (src_frames - 29 - v) / 48000.0)
        else:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        if v % 11 != 6:
            e.sampler_play(vc["sampler"])                          # v % 11 == 6: never plays
        if v == 9:
            e.set_param(vc["volume"], 0, 0.0)                      # fades to silence: muted from then on
    return voices


@pytest.mark.parametrize("mbf,n_voices,fx,src_frames,n_cb", [(64, 100, False, 701, 260), (64, 132, True, 997, 200), (128, 260, False, 1500, 120),
                                                              (256, 1024, False, 4099, 60)])
def test_varied_banks_through_the_resident_kernel_match_the_oracle(mbf, n_voices, fx, src_frames, n_cb):
    """(the scenario that showed a resident kernel with a second rendering path compiled in writing garbage from its ORDINARY
    block — scripts/experiments/r03_rt_sliced_kernel.patch)"""
    g, o = GpuEngine(max_block_frames=mbf), OracleEngine(max_block_frames=mbf)
    outs_g, outs_o = [], []
    for e, outs in ((g, outs_g), (o, outs_o)):
        _varied_bank(e, n_voices, fx, src_frames)
        for _ in range(n_cb):
            outs.append(np.asarray(e.process_interleaved(mbf)))
    assert g.cx.plan_kind() == 1
    assert_bits_equal(np.concatenate(outs_o), np.concatenate(outs_g), "varied bank %d voices block %d" % (n_voices, mbf))
    if os.environ.get("FWGPU_RT_PERSIST") != "0":
        launches, doorbells = g.cx.rt_resident_stats()
        assert doorbells >= n_cb // 2, (launches, doorbells)
