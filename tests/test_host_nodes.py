"""Custom (host) nodes inside a device-resident graph — SURVEY §8b: "unknown/custom Rust nodes fall back to B1 on host with
explicit D2H/H2D of just their buffers"; VERDICT r2 missing #2.

The reference's schedule loop calls ANY `dyn AudioNodeProcessor` (graph/processor.rs:226-247).  A node libfwgpu has no kernel
for is a FWGPU_HOST_NODE: the plan is cut at its level, its input buffers go to pinned host memory, the caller's process
function runs on the audio thread once per block, its outputs go back.  The oracle gets the SAME Python process functions as
custom processors (KIND_CUSTOM): GPU == oracle bit for bit means the cut moves the right buffers, flags and blocks.

CPU tier: the host half of it (plan cut, staging, callback cadence, masks, error returns) on the host-only harness."""
import numpy as np
import pytest

import fwapi
import scenarios
from fwapi import GpuEngine, OracleEngine


# ---------------------------------------------------------------------------------------------- custom processors (test doubles)
class OnePoleGain(object):
    """stereo 2 -> 2, stateful: y[n] = a * x[n] + (1 - a) * y[n-1] per channel (f32, unfused), then x gain.  All inputs silent ->
    clears its outputs, resets its state and says so in the out mask (what the reference's own nodes do)."""

    def __init__(self, a=0.25, gain=2.0):
        self.a, self.b, self.g = np.float32(a), np.float32(1.0 - a), np.float32(gain)
        self.y = np.zeros(2, dtype=np.float32)
        self.calls = []

    def __call__(self, frames, ins, outs, in_mask, t, status):
        self.calls.append((frames, in_mask, t, status))
        if in_mask & 3 == 3:
            for o in outs:
                o[:] = 0.0
            self.y[:] = 0
            return 3
        for c in range(2):
            x = ins[c]
            y = self.y[c]
            out = outs[c]
            for i in range(frames):
                y = np.float32(np.float32(self.a * x[i]) + np.float32(self.b * y))
                out[i] = np.float32(y * self.g)
            self.y[c] = y
        return 0


class MidSideSplit(object):
    """2 -> 3: mid, side, and a constant-silent third output it flags; stateless"""

    def __call__(self, frames, ins, outs, in_mask, t, status):
        l, r = ins
        outs[0][:] = (l + r) * np.float32(0.5)
        outs[1][:] = (l - r) * np.float32(0.5)
        outs[2][:] = 0.0
        return 4


class Source(object):
    """0 -> 2: a generator (no inputs at all): a counter ramp, the second channel negated"""

    def __init__(self):
        self.n = 0

    def __call__(self, frames, ins, outs, in_mask, t, status):
        ramp = ((np.arange(self.n, self.n + frames) % 97).astype(np.float32) - np.float32(48.0)) * np.float32(1.0 / 64.0)
        outs[0][:] = ramp
        outs[1][:] = -ramp
        self.n += frames
        return 0


def desk_with_host_nodes(e, n_a=12, n_b=9, src_frames=1100):
    """bank A (voices -> SumNode) -> HOST one-pole+gain -> volume ┐
       bank B (voices -> SumNode) --------------------------------┼-> 4-port mixer -> hard clip -> out
       HOST source (no inputs) -> HOST mid/side split (2 -> 3) -> mid+side into the mixer's 3rd port (third output dangling)
    Two host nodes at different levels, one of them a source, one stateful; banks that stay fusable (hybrid plan)."""
    rng = np.random.default_rng(77)
    procs = dict(pole=OnePoleGain(), ms=MidSideSplit(), src=Source())

    def bank(n, seed):
        ends, voices = [], []
        for v in range(n):
            s = e.sampler(100.0)
            vol = e.volume(float(rng.uniform(20, 100)))
            pan = e.pan(float(rng.uniform(-1, 1)))
            e.connect_stereo(s, vol)
            e.connect_stereo(vol, pan)
            ends.append(pan)
            voices.append((s, seed * 1000 + v))
        m = e.sum(n)
        for p, x in enumerate(ends):
            e.connect_stereo(x, m, 2 * p)
        return m, voices

    ma, va = bank(n_a, 1)
    mb, vb = bank(n_b, 2)
    pole = e.host_node(2, 2, procs["pole"])
    e.connect_stereo(ma, pole)
    vol = e.volume(70.0)
    e.connect_stereo(pole, vol)
    src = e.host_node(0, 2, procs["src"])
    ms = e.host_node(2, 3, procs["ms"])
    e.connect_stereo(src, ms)
    mix = e.sum(4)
    e.connect_stereo(vol, mix, 0)
    e.connect_stereo(mb, mix, 2)
    e.connect_stereo(ms, mix, 4)  # mid -> L, side -> R of port 2; port 3 stays unconnected
    clip = e.hard_clip(-1.0)
    e.connect_stereo(mix, clip)
    e.connect_stereo(clip, e.graph_out_node)
    e.update()
    for s, seed in va + vb:
        e.sampler_set_sample(s, e.new_sample(fwapi.PLANAR_F32, 2, scenarios.voice_source(seed, src_frames)))
        e.sampler_set_loop_range(s, fwapi.LOOP_FULL)
    return dict(va=va, vb=vb, procs=procs, vol=vol, pole=pole)


def run_desk(e, calls=(3, 7, 1, 12)):
    d = desk_with_host_nodes(e)
    outs = []
    # call 0: nothing plays -> bank A's bus is silent: the host node sees in_mask 3, clears, flags
    outs.append(e.process_blocks(calls[0]))
    for s, _ in d["va"] + d["vb"]:
        e.sampler_play(s)
    outs.append(e.process_blocks(calls[1]))
    e.set_param(d["vol"], 0, 35.0)
    outs.append(e.process_blocks(calls[2]))
    for s, _ in d["va"]:
        e.sampler_pause(s)
    outs.append(e.process_blocks(calls[3]))
    return np.concatenate(outs), d


def ragged_rack(e):
    """no sampler anywhere (a sampler panics on a block shorter than max_block_frames in the reference, Q5): a host source and a
    beep through a stateful host node, driven by process_interleaved calls with ragged tails and real ProcInfo"""
    procs = dict(pole=OnePoleGain(a=0.5, gain=0.5), src=Source())
    src = e.host_node(0, 2, procs["src"])
    beep = e.beep(660.0, -6.0, True, 2)
    mix = e.sum(2)
    e.connect_stereo(src, mix, 0)
    e.connect_stereo(beep, mix, 2)
    pole = e.host_node(2, 2, procs["pole"])
    e.connect_stereo(mix, pole)
    vol = e.volume(80.0)
    e.connect_stereo(pole, vol)
    e.connect_stereo(vol, e.graph_out_node)
    e.update()
    outs = [e.process_interleaved(2 * e.max_block_frames + 37, 2, t=1.5, status=2), e.process_interleaved(5, 2, t=2.5),
            e.process_interleaved(7 * e.max_block_frames, 2, t=3.0, status=1), e.process_interleaved(e.max_block_frames - 1, 2, t=4.0)]
    return np.concatenate(outs), procs


@pytest.mark.gpu
@pytest.mark.parametrize("force_generic,max_batch", [(False, 8), (False, 1), (True, 4), (False, 64)])
def test_host_nodes_inside_a_device_graph_equal_the_oracle_with_the_same_custom_processors(force_generic, max_batch):
    g = GpuEngine(max_block_frames=64, force_generic=force_generic, max_batch=max_batch)
    o = OracleEngine(max_block_frames=64)
    out_g, dg = run_desk(g)
    out_o, do = run_desk(o)
    assert np.array_equal(out_g.view(np.uint32), out_o.view(np.uint32))
    assert np.any(out_o)
    n_host, n_cb = g.cx.plan_host_nodes()
    assert n_host == 3 and n_cb > 0
    if not force_generic:
        assert g.cx.plan_kind() == 3 and g.cx.plan_fused_voices() == 21  # the banks stay on the fused kernels
    # the callbacks saw the same blocks in the same order with the same ProcInfo (frames, in mask, stream time, status)
    assert dg["procs"]["pole"].calls == do["procs"]["pole"].calls
    assert dg["procs"]["pole"].calls[0][1] == 3
    assert dg["procs"]["src"].n == do["procs"]["src"].n


@pytest.mark.gpu
@pytest.mark.parametrize("max_batch", [1, 4])
def test_host_nodes_ragged_calls_and_proc_info(max_batch):
    g = GpuEngine(max_block_frames=64, max_batch=max_batch)
    o = OracleEngine(max_block_frames=64)
    out_g, pg = ragged_rack(g)
    out_o, po = ragged_rack(o)
    # BeepTest's sinf: ocml vs glibc, 2e-6 absolute (DESIGN H6); the host nodes' own arithmetic is the same Python on both sides
    assert np.max(np.abs(out_g - out_o)) <= 4e-6 and np.any(out_o)
    assert [c[:2] + c[2:] for c in pg["pole"].calls] == [c[:2] + c[2:] for c in po["pole"].calls]
    assert [c[0] for c in pg["pole"].calls] == [64, 64, 37, 5] + [64] * 7 + [63]
    assert pg["pole"].calls[2][2:] == (1.5, 2) and pg["pole"].calls[4][2:] == (3.0, 1)


@pytest.mark.gpu
def test_host_node_in_an_imported_schedule():
    # keep Firewheel's own scheduler: the oracle's CompiledSchedule (reference buffer indices) handed to fwgpu_schedule_upload
    o = OracleEngine(max_block_frames=64)
    out_o, _ = run_desk(o)
    g = GpuEngine(max_block_frames=64, max_batch=4)
    o2 = OracleEngine(max_block_frames=64)
    # build the same graph on both, then replace the GPU's own plan by the oracle's schedule, ids mapped by creation order
    d2 = desk_with_host_nodes(o2)
    dg = desk_with_host_nodes(g)
    sched = o2.schedule()
    ids_o = sorted({s["id"] for s in sched})
    # creation order is identical on both engines: map by rank of the id's slot
    slot = lambda i: i & 0xffffffff
    by_slot_g = {}
    for nid in list(g.cx._nodes) + [g.graph_in_node, g.graph_out_node]:
        by_slot_g[slot(nid)] = nid
    mapped = [dict(id=by_slot_g[slot(s["id"])], **{"in": s["in"], "out": s["out"]}) for s in sched]
    assert len(ids_o) == len(mapped)
    g.cx.schedule_upload(mapped, o2.num_buffers())
    outs = []
    outs.append(g.process_blocks(3))
    for s, _ in dg["va"] + dg["vb"]:
        g.sampler_play(s)
    outs.append(g.process_blocks(7))
    g.set_param(dg["vol"], 0, 35.0)
    outs.append(g.process_blocks(1))
    for s, _ in dg["va"]:
        g.sampler_pause(s)
    outs.append(g.process_blocks(12))
    assert np.array_equal(np.concatenate(outs).view(np.uint32), out_o.view(np.uint32))


# ---------------------------------------------------------------------------------------------- CPU tier: the host half
def test_host_node_plan_cut_and_callback_cadence_on_the_host_harness():
    from firewheel_amd import FwgpuError

    e = fwapi.HostOnlyEngine(max_block_frames=64, max_batch=8)
    d = desk_with_host_nodes(e)
    n_host, n_cb = e.cx.plan_host_nodes()
    assert (n_host, n_cb) == (3, 0)
    assert e.cx.plan_kind() == 3  # hybrid: the banks on the fused kernels, the rest (and the cut) on the levels
    e.process_blocks(20)  # 20 blocks = batches of 8 + 8 + 4: every host node is called once per block
    assert e.cx.plan_host_nodes() == (3, 60)
    calls = d["procs"]["pole"].calls
    assert [c[0] for c in calls] == [64] * 20
    assert d["procs"]["src"].n == 20 * 64
    e.process_interleaved(64 + 10, 2, t=0.25, status=1)  # a ragged call: one whole block + 10 frames
    assert [c[0] for c in d["procs"]["pole"].calls[-2:]] == [64, 10] and d["procs"]["pole"].calls[-1][2:] == (0.25, 1)
    assert e.violation() == ""
    # B1 on a host node is refused; a host node without a function fails activation and leaves the old plan in place
    with pytest.raises(FwgpuError, match="lives in the caller"):
        e.cx.node_process(d["pole"], 64, [np.zeros(64, np.float32)] * 2, [np.zeros(64, np.float32)] * 2)
    from firewheel_amd.graph import _RawNode

    bare = e.cx.add_node(2, 2, _RawNode(15, []))
    with pytest.raises(Exception, match="without a process function"):
        e.cx.update()
    e.cx.remove_node(bare)
    e.cx.update()
    e.process_blocks(2)
    assert e.cx.plan_host_nodes()[0] == 3


def test_host_node_set_process_argument_checks():
    from firewheel_amd import FwgpuError
    from firewheel_amd import _lib as flib

    e = fwapi.HostOnlyEngine(max_block_frames=64)
    v = e.volume(50.0)
    cb = flib.host_process_adapter(lambda *a: 0)
    L = e.cx.L
    assert L.fwgpu_host_node_set_process(e.cx.c, v, cb, None) < 0 and b"not a host node" in L.fwgpu_last_error(e.cx.c)
    assert L.fwgpu_host_node_set_process(e.cx.c, 12345678, cb, None) < 0
    assert L.fwgpu_host_node_set_process(None, v, cb, None) < 0
