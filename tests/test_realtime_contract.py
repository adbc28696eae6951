"""CPU tier: the boundary's threading / realtime contract (include/fwgpu.h) on the host-only harness —
the lock-free control->audio message ring under ThreadSanitizer, the audio thread's host-heap allocation count across
20 000 callbacks, ReturnSample / sample_retired, the headless stream's underflow bookkeeping against the oracle's
restatement of firewheel-cpal's DataCallback, ext-pool slice reuse and activation rollback under injected failures.
No audio is computed here (fake HIP runtime + launch stubs)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import fwapi
from fwapi import PLANAR_F32, HostOnlyEngine, OracleEngine, hostonly_lib

HARNESS = os.path.join(fwapi.ROOT, "tests", "host_harness")


def _build_driver(out, extra):
    srcs = [os.path.join(HARNESS, "rt_driver.cpp"), os.path.join(HARNESS, "launch_stubs.cpp")] + fwapi.host_sources()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-Wno-unused-function", "-I", os.path.join(HARNESS, "fakehip"),
                           "-I", os.path.join(fwapi.ROOT, "include"), "-o", out] + extra + srcs)


def test_message_ring_is_race_free_and_loses_nothing_under_tsan(tmp_path):
    """two control threads push gain / pan / sampler messages while the audio thread runs 3000 one-block callbacks and a
    third party polls the return ring: ThreadSanitizer must stay silent and every message must be applied exactly once"""
    tsan = subprocess.check_output(["gcc", "-print-file-name=libtsan.so"]).decode().strip()
    if not os.path.isabs(tsan):
        pytest.skip("no libtsan in this toolchain")
    exe = str(tmp_path / "rt_tsan")
    _build_driver(exe, ["-fsanitize=thread"])
    r = subprocess.run([exe, "tsan"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    assert r.returncode == 0 and "tsan-run ok" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.stdout[-2000:], r.stderr[-4000:])


def test_graph_edits_race_callbacks_under_tsan(tmp_path):
    """VERDICT r2 missing #3 / ADVICE r2: an editor thread adds and removes voice chains and calls fwgpu_update 590 times while the
    audio thread runs callbacks and a third thread sends messages.  fwgpu_update builds the new plan off to the side; the
    next callback adopts it (graph/processor.rs:167-206).  ThreadSanitizer must stay silent and every call must succeed."""
    tsan = subprocess.check_output(["gcc", "-print-file-name=libtsan.so"]).decode().strip()
    if not os.path.isabs(tsan):
        pytest.skip("no libtsan in this toolchain")
    exe = str(tmp_path / "rt_tsan_edits")
    _build_driver(exe, ["-fsanitize=thread"])
    r = subprocess.run([exe, "edits"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    assert r.returncode == 0 and "edits-run ok" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.stdout[-2000:], r.stderr[-6000:])


def test_audio_gate_steps_back_for_a_waiting_control_call_only_for_a_bounded_time(tmp_path):
    """AudioGate lets a control call that waits at ControlGate go first — for at most gate_defer_ns.  A waiter
    that raised the counter and never arrives (descheduled) costs a callback the bound, not a scheduler quantum; a waiter that does
    arrive still gets in beside back-to-back callbacks."""
    exe = str(tmp_path / "rt_gate")
    _build_driver(exe, [])
    r = subprocess.run([exe, "gate"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gate-run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_adopting_a_plan_does_not_touch_the_host_allocator_on_the_audio_thread(tmp_path):
    exe = str(tmp_path / "rt_alloc_edits")
    _build_driver(exe, ["-DCOUNT_ALLOCS"])
    r = subprocess.run([exe, "edits"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "edits-run ok" in r.stdout and "audio-thread allocations 0" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_audio_thread_never_touches_the_host_allocator_once_warm(tmp_path):
    """malloc / calloc / realloc of the whole process are counted per thread: 10 000 steady callbacks, 10 000 more while a
    control thread sends messages, and a failing call — zero allocations on the audio thread, zero device / pinned ones"""
    exe = str(tmp_path / "rt_alloc")
    _build_driver(exe, ["-DCOUNT_ALLOCS"])
    r = subprocess.run([exe, "alloc"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "alloc-run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "steady 0, with messages 0, failing call 0, device/pinned 0" in r.stdout, r.stdout


@pytest.mark.skipif(bool(os.environ.get("FWGPU_LAZY_ADOPT")), reason="asserts what an update does when the audio side is idle (it adopts at once); "
                    "in lazy mode the old processors are dropped by the next process call")
def test_returned_samples_and_sample_retired():
    """ProcessorToNodeMsg::ReturnSample (sampler.rs:339-343,563-571): a SetSample that replaces a held sample hands the
    old one back once the call that applied it is done; a removed sampler hands its sample back at the next schedule"""
    e = HostOnlyEngine(max_block_frames=64, max_batch=8)
    a = e.new_sample(PLANAR_F32, 2, np.zeros((2, 500), np.float32))
    b = e.new_sample(PLANAR_F32, 2, np.zeros((2, 500), np.float32))
    s1, s2 = e.sampler(), e.sampler()
    m = e.sum(2)
    e.connect_stereo(s1, m, 0)
    e.connect_stereo(s2, m, 2)
    e.connect_stereo(m, e.graph_out_node)
    e.update()
    cx = e.cx
    assert cx.sample_retired(a) and cx.sample_retired(b)          # nobody has asked for them yet
    e.sampler_set_sample(s1, a)
    e.sampler_set_sample(s2, a)
    assert not cx.sample_retired(a)                                # the queued messages hold references
    e.process_blocks(2)
    assert cx.poll_returned_samples() == [] and not cx.sample_retired(a)
    e.sampler_set_sample(s1, b, at_block=3)                        # takes effect in block 3 of the next calls
    e.process_blocks(2)
    assert cx.poll_returned_samples() == []                        # not yet applied
    e.process_blocks(2)
    assert cx.poll_returned_samples() == [(s1, a)]                 # s1 let go of `a` ...
    assert not cx.sample_retired(a)                                # ... s2 still holds it
    assert not cx.sample_retired(b)
    e.sampler_set_sample(s2, b)
    e.sampler_set_sample(s2, a)                                    # b in, then straight out again
    e.process_blocks(1)
    assert cx.poll_returned_samples() == [(s2, a), (s2, b)]
    assert not cx.sample_retired(a) and not cx.sample_retired(b)
    # a message that never reaches its sampler (the node is removed first) gives its reference back
    e.sampler_set_sample(s2, b, at_block=50)
    e.remove_node(s2)
    e.update()                                                     # the processor is dropped with the old schedule
    assert cx.sample_retired(a)                                    # s2 held `a`
    assert not cx.sample_retired(b)                                # s1 holds `b`
    e.remove_node(s1)
    e.update()
    assert cx.sample_retired(b)
    cx.destroy_sample(a)
    cx.destroy_sample(b)
    # FIR / resampler nodes name their sample for life
    ir = e.new_sample(PLANAR_F32, 1, np.ones(16, np.float32))
    f = e.fir(ir)
    assert not cx.sample_retired(ir)
    e.remove_node(f)
    assert cx.sample_retired(ir)


def test_b1_node_process_retires_only_its_own_nodes_messages():
    """fwgpu_node_process consumes ONE node's block: messages queued for other nodes stay queued (they were
    deleted unapplied).  Counted through the returns: a SetSample for node B must still be applied after A processed."""
    e = HostOnlyEngine(max_block_frames=64)
    smp = [e.new_sample(PLANAR_F32, 2, np.zeros((2, 100), np.float32)) for _ in range(3)]
    a, b = e.sampler(), e.sampler()
    m = e.sum(2)
    e.connect_stereo(a, m, 0)
    e.connect_stereo(b, m, 2)
    e.connect_stereo(m, e.graph_out_node)
    e.update()
    for n in (a, b):
        e.sampler_set_sample(n, smp[0])
    outs = [np.zeros(64, np.float32), np.zeros(64, np.float32)]
    e.cx.node_process(a, 64, [], outs)
    e.cx.node_process(b, 64, [], outs)
    e.sampler_set_sample(b, smp[1])                 # block 0 of b's next process
    e.sampler_set_sample(b, smp[2], at_block=1)     # block 1 of b's
    e.cx.node_process(a, 64, [], outs)              # a processes twice: b's queue must not move
    e.cx.node_process(a, 64, [], outs)
    assert e.cx.poll_returned_samples() == []
    e.cx.node_process(b, 64, [], outs)
    assert e.cx.poll_returned_samples() == [(b, smp[0])]
    e.cx.node_process(b, 64, [], outs)
    assert e.cx.poll_returned_samples() == [(b, smp[1])]


def test_headless_stream_matches_the_reference_callback_bookkeeping():
    """fwgpu_stream_callback vs the oracle's restatement of DataCallback::callback (firewheel-cpal/src/lib.rs:378-449): the
    same stream_time_secs and OUTPUT_UNDERFLOW decision for every callback of a jittery, occasionally late clock"""
    L = fwapi.oracle_lib()
    for block, seed in ((256, 1), (64, 2), (1024, 3)):
        rng = np.random.default_rng(seed)
        e = HostOnlyEngine(max_block_frames=block)
        vol = e.volume(50.0)
        e.connect_stereo(vol, e.graph_out_node)
        e.update()
        st = e.cx.open_stream(0, 2)
        o = OracleEngine(max_block_frames=block)
        ost = L.fwo_stream_new(o.c, 48000, 0, 2)
        period = block / 48000.0
        t = 100.0 + float(rng.uniform(0, 5))
        underflows = 0
        out = np.zeros(block * 2, np.float32)
        for i in range(400):
            late = rng.random() < 0.1
            t += period * (float(rng.uniform(1.25, 3.0)) if late else float(rng.uniform(0.9, 1.15)))
            _, status = st.callback(block, t)
            ot = C.c_double()
            ostatus = L.fwo_stream_callback(ost, out.ctypes.data_as(C.POINTER(C.c_float)), block, t, C.byref(ot))
            assert status == ostatus, (i, status, ostatus)
            assert e.cx.proc_info()[0] == ot.value, (i, e.cx.proc_info()[0], ot.value)   # bit-identical f64
            assert e.cx.proc_info()[1] == status
            underflows += 1 if status & 2 else 0
        assert underflows > 5
        cbs, unders, last = st.stats()
        assert (cbs, unders) == (400, underflows) and e.cx.proc_info()[2] == underflows
        L.fwo_stream_free(ost)
        st.close()


def test_stream_run_is_the_callback_loop():
    """fwgpu_stream_run(n) = n fwgpu_stream_callback calls at instants first + i * period: the same callback count, the same
    stream time at the end, no underflow (a clock that is never late), the last block in the output buffer"""
    for block in (64, 256):
        e1, e2 = HostOnlyEngine(max_block_frames=block), HostOnlyEngine(max_block_frames=block)
        sts = []
        for e in (e1, e2):
            vol = e.volume(50.0)
            e.connect_stereo(vol, e.graph_out_node)
            e.update()
            sts.append(e.cx.open_stream(0, 2))
        period = block / 48000.0
        out, secs = sts[0].run(block, 37, 12.5)
        assert secs > 0.0 and out.shape == (block * 2,) and not np.isnan(out).any()
        for i in range(37):
            sts[1].callback(block, 12.5 + i * period)
        assert sts[0].stats() == sts[1].stats() and sts[0].stats()[:2] == (37, 0)
        assert e1.cx.proc_info()[:2] == e2.cx.proc_info()[:2]
        out2, _ = sts[0].run(block, 3, 12.5 + 37 * period)   # the clock carries on: still no underflow
        assert sts[0].stats()[:2] == (40, 0)
        for st in sts:
            st.close()


def test_stream_without_a_schedule_outputs_silence():
    e = HostOnlyEngine(max_block_frames=64)
    st = e.cx.open_stream(0, 2)
    out, status = st.callback(64, 1.0)          # cpal/lib.rs:442-445: no processor yet -> output.fill(0.0)
    assert status == 0 and not out.any()


def test_ext_pool_slices_are_recycled():
    """a host that keeps spawning and retiring effect voices (delay rings) must not grow the ext state pool without bound
    (ADVICE r1): the slice of a removed node is reused by the next node that needs the same amount"""
    e = HostOnlyEngine(max_block_frames=64, max_batch=4)
    m = e.sum(4)
    e.connect_stereo(m, e.graph_out_node)
    smp = e.new_sample(PLANAR_F32, 2, np.zeros((2, 500), np.float32))
    live = []
    used = []
    for it in range(40):
        s = e.sampler()
        d = e.delay(0.05, 0.3, 0.5)      # 2400-frame stereo ring
        e.connect_stereo(s, d)
        port = it % 4
        if len(live) == 4:
            old_s, old_d = live.pop(0)
            e.remove_node(old_s)
            e.remove_node(old_d)
        e.connect_stereo(d, m, 2 * port)
        live.append((s, d))
        e.update()
        e.sampler_set_sample(s, smp)
        e.process_blocks(3)
        used.append(e.cx.ext_pool_floats()[0])
        assert e.violation() == ""
    # four rings alive at most, plus one in hand-over: a removed node's slice becomes reusable once the plan WITHOUT the node
    # has been adopted (until then a process call may still be running the node) — the edit that removes voice n and adds
    # voice n + 4 in one go therefore takes a fifth slice; from then on every edit reuses the one freed by the edit before
    assert used[4] == used[-1] and used[4] == 5 * used[0], used
    assert used[0] < used[3]


def test_activation_failures_roll_back_completely():
    """install_plan is two-phase: whichever device allocation fails while new nodes are being activated — ext
    pool growth, the scatter uploads, an impulse-response conversion — the ctx keeps no half-activated node, and the next
    fwgpu_update activates the same nodes correctly.  Failures are injected into the fake runtime's hipMalloc."""
    L = hostonly_lib()
    L.fwh_fail_alloc.argtypes = [C.c_longlong]
    L.fwh_fail_alloc.restype = None
    L.fwh_violation_reset()
    n_failed = 0
    for nth in range(1, 40):
        e = HostOnlyEngine(max_block_frames=64, max_batch=4)
        m = e.sum(3)
        e.connect_stereo(m, e.graph_out_node)
        s = e.sampler()
        e.connect_stereo(s, m, 0)
        e.update()
        e.process_blocks(2)
        kind0 = e.cx.plan_kind()
        # second batch of nodes: a biquad + delay voice and a FIR voice with a fresh impulse response
        ir = e.new_sample(PLANAR_F32, 2, np.ones((2, 300), np.float32))
        s2, bq, dl = e.sampler(), e.biquad(0, 1000.0), e.delay(0.01, 0.2, 0.5)
        e.connect_stereo(s2, bq)
        e.connect_stereo(bq, dl)
        e.connect_stereo(dl, m, 2)
        s3, f = e.sampler(), e.fir(ir)
        e.connect_stereo(s3, f)
        e.connect_stereo(f, m, 4)
        used0 = e.cx.ext_pool_floats()[0]
        L.fwh_fail_alloc(nth)
        failed = False
        try:
            e.update()
        except Exception:
            failed = True
        L.fwh_fail_alloc(0)
        if failed:
            # nothing leaked into the bookkeeping: either the old plan is still installed (activation failed before the
            # device tables were touched) or the ctx has no plan at all (it outputs silence) — never a mixture
            assert e.cx.plan_kind() in (kind0, -1)
            e.process_blocks(2)
            e.update()                      # same nodes, this time for real
        assert e.cx.plan_kind() == 0        # (a FIR voice: generic executor)
        assert e.cx.ext_pool_floats()[0] > used0
        e.process_blocks(3)
        assert e.violation() == "", (nth, e.violation())
        n_failed += 1 if failed else 0
        if not failed and nth > 25:
            break                           # past the last allocation of a successful update
    else:
        raise AssertionError("the update never ran out of allocations to fail")
    assert n_failed >= 5, n_failed          # activation-phase and table-phase failures were both hit


def test_schedule_upload_rejects_duplicates_and_misplaced_io():
    from fwapi import DUMMY

    e = HostOnlyEngine(max_block_frames=64)
    a = e.add_node(DUMMY, 1, 1)
    gi, go = e.graph_in_node, e.graph_out_node
    first = {"id": gi, "in": [], "out": []}
    node = {"id": a, "in": [(0, True)], "out": [0]}
    last = {"id": go, "in": [(0, False), (1, True)], "out": []}
    e.cx.schedule_upload([first, node, last], 2)
    with pytest.raises(Exception):
        e.cx.schedule_upload([first, node, node, last], 2)       # two waves would share one NodeState
    with pytest.raises(Exception):
        e.cx.schedule_upload([first, node, dict(first), last], 2)  # graph_in in the middle
    e.cx.schedule_upload([first, node, last], 2)
