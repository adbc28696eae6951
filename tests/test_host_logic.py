"""CPU tier: the HOST half of libfwgpu (fwgpu_abi / fwgpu_run / fwgpu_plan_* / fwgpu_control_math / fwgpu_graph .cpp) on the host-only harness of tests/host_harness —
a fake HIP runtime and no-op kernel launches, so NO audio is computed here (the parity tests proper are the GPU tier).
What runs is the product's own graph editing, planning, plan selection, batching, message bookkeeping and error
conventions behind the real C ABI entry points."""
import ctypes as C

import numpy as np
import pytest

import fwapi
import scenarios
from fwapi import (DUMMY, PLANAR_F32, SAMPLER, SUM, VOLUME, CompileGraphError, HostOnlyEngine, OracleEngine, hostonly_lib)

ERR_INVALID, ERR_QUEUE_FULL = -20, -21


@pytest.fixture(autouse=True)
def no_descriptor_violation():
    """the launch stubs of the harness validate every launch: table extents as the kernels index them, the descriptor
    invariants the kernels rely on (leaves / chain groups tile the voice rows, sorted messages, ...): none may trip"""
    L = hostonly_lib()
    L.fwh_violation_reset()
    yield
    assert L.fwh_violation() == b"", L.fwh_violation()


def bank(e, n_voices=12, radix=4, chain=False, clip_in_voice=False, master=False):
    """sampler -> [biquad -> delay] -> volume -> pan -> sum tree -> [master volume] -> graph_out; returns the samplers"""
    ends, smp = [], []
    for v in range(n_voices):
        s = e.sampler(90.0)
        cur = s
        if chain:
            b = e.biquad(0, 1000.0 + 10 * v, 0.7)
            d = e.delay(0.01 + 0.001 * v, feedback=0.2, mix=0.5)
            e.connect_stereo(cur, b)
            e.connect_stereo(b, d)
            cur = d
        if clip_in_voice and v == 3:
            h = e.hard_clip(-3.0)
            e.connect_stereo(cur, h)
            cur = h
        g = e.volume(50.0 + v)
        p = e.pan(0.1 * (v % 5) - 0.2)
        e.connect_stereo(cur, g)
        e.connect_stereo(g, p)
        ends.append(p)
        smp.append(s)
    level = ends
    while len(level) > 1:
        nxt = []
        for i in range(0, len(level), radix):
            grp = level[i:i + radix]
            m = e.sum(max(len(grp), 2))
            for k, n in enumerate(grp):
                e.connect_stereo(n, m, 2 * k)
            nxt.append(m)
        level = nxt
    top = level[0]
    if master:
        mv = e.volume(80.0)
        e.connect_stereo(top, mv)
        top = mv
    e.connect_stereo(top, e.graph_out_node)
    e.update()
    return smp


# a hard clip (or a width) among a dry voice's stages rides the voice-bank plan as a stage program; round 6: k_chain clips per channel
# too, so a hard clip behind a biquad / delay stays on the chain plan (a stereo WIDTH there still sends its bank to the level executor)
@pytest.mark.parametrize("kw,plan", [({}, 1), ({"chain": True}, 2), ({"clip_in_voice": True}, 1), ({"chain": True, "clip_in_voice": True}, 2),
                                     ({"master": True}, 1),
                                     ({"chain": True, "master": True}, 2), ({"radix": 32, "n_voices": 70}, 1)])
def test_plan_selection(kw, plan):
    e = HostOnlyEngine(max_block_frames=128)
    bank(e, **kw)
    assert e.cx.plan_kind() == plan
    e.cx.set_force_generic(True)
    assert e.cx.plan_kind() == 0
    e.cx.set_force_generic(False)
    assert e.cx.plan_kind() == plan


def send_graph(e, n_voices=12, radix=4, return_node=True, dangling_bank=False):
    """bank() with one leaf bus ALSO tapped into a return gain that joins the root in a two-port sum"""
    ends = []
    for v in range(n_voices):
        s = e.sampler(90.0)
        g = e.volume(50.0 + v)
        e.connect_stereo(s, g)
        ends.append(g)
    leaves = []
    for i in range(0, n_voices, radix):
        m = e.sum(max(len(ends[i:i + radix]), 2))
        for k, n in enumerate(ends[i:i + radix]):
            e.connect_stereo(n, m, 2 * k)
        leaves.append(m)
    root = e.sum(len(leaves))
    for k, m in enumerate(leaves):
        e.connect_stereo(m, root, 2 * k)
    ret = e.volume(30.0)
    e.connect_stereo(leaves[0], ret)
    mix = e.sum(2)
    e.connect_stereo(root, mix, 0)
    e.connect_stereo(ret, mix, 2)
    e.connect_stereo(mix, e.graph_out_node)
    e.update()
    return leaves, ret


def test_hybrid_plan_selection_and_launch_sequence():
    """a bus consumed twice is no fused shape; the leaf SumNodes whose ports are all dry voice chains are rendered by the
    voice-bank kernels (plan kind 3): one control + one leaf launch per batch, then the level executor WITHOUT those nodes
    (the stubs check that no level list of a hybrid batch names a sampler); a partial block runs all of it on the levels"""
    e = HostOnlyEngine(max_block_frames=64, max_batch=16)
    send_graph(e)
    assert e.cx.plan_kind() == 3
    e.reset_launches()
    e.process_blocks(40)  # 16 + 16 + 8
    c = e.launches()
    assert (c["voice_control"], c["leaf_sum"], c["root_out"], c["chain"]) == (3, 3, 0, 0) and c["level"] >= 2 * 3  # root + return level, mix level
    lv_hybrid = c["level"]
    e.cx.set_force_generic(True)
    assert e.cx.plan_kind() == 0
    e.reset_launches()
    e.process_blocks(40)
    c = e.launches()
    assert c["voice_control"] == 0 and c["leaf_sum"] == 0 and c["level"] > lv_hybrid  # the voice levels are back
    e.cx.set_force_generic(False)
    e.reset_launches()
    e.process_interleaved(64 * 3 + 10)  # 3 whole blocks (hybrid) + a 10-frame tail (levels only)
    c = e.launches()
    assert (c["voice_control"], c["leaf_sum"]) == (1, 1)
    assert e.cx.plan_fused_voices() == 12
    # fewer than 8 voices in fusable banks: not worth the two launches
    e2 = HostOnlyEngine(max_block_frames=64)
    send_graph(e2, n_voices=6, radix=3)
    assert e2.cx.plan_kind() == 0 and e2.cx.plan_fused_voices() == 0


def test_hybrid_plan_splits_a_mixer_that_also_takes_a_bus():
    """a mixer SumNode with 10 voices on its leading ports and a return on its last: the voices are summed by the voice-bank
    kernels into a partial bus, the node stays on the levels as the continuation (partial, return).  A 3-port mixer is not
    split (its path is spelled out port by port in the reference), nor one whose FIRST port is the bus — round 4: the voices of
    those two shapes are SOLO voices (one-port leaves writing the chain's own pool buffers), and so is the bare sampler in front of
    the detour: the plan is hybrid either way"""
    def mixer(e, n_voices, bus_port_first=False):
        ends = []
        for v in range(n_voices):
            s = e.sampler(90.0)
            g = e.volume(40.0 + v)
            e.connect_stereo(s, g)
            ends.append(g)
        side = e.sampler(50.0)          # something that is not a voice chain of this mixer: a sampler through a 1-in detour
        s2m = e.add_node(fwapi.STEREO_TO_MONO, 2, 1)
        m2s = e.add_node(fwapi.MONO_TO_STEREO, 1, 2)
        e.connect_stereo(side, s2m)
        e.connect(s2m, 0, m2s, 0)
        mix = e.sum(n_voices + 1)
        ports = list(range(n_voices + 1))
        bus_port = 0 if bus_port_first else n_voices
        vp = [p for p in ports if p != bus_port]
        for p, n in zip(vp, ends):
            e.connect_stereo(n, mix, 2 * p)
        e.connect_stereo(m2s, mix, 2 * bus_port)
        e.connect_stereo(mix, e.graph_out_node)
        e.update()
        return e

    e = mixer(HostOnlyEngine(max_block_frames=64, max_batch=8), 10)
    assert e.cx.plan_kind() == 3 and e.cx.plan_fused_voices() == 10 + 1             # (+ the detour's sampler, solo)
    e.reset_launches()
    e.process_blocks(8)
    c = e.launches()
    assert (c["voice_control"], c["leaf_sum"]) == (1, 1) and c["level"] >= 3   # detour levels + the continuation
    assert e.violation() == "", e.violation()
    for e2, nv in ((mixer(HostOnlyEngine(max_block_frames=64, max_batch=8), 10, bus_port_first=True), 11),   # voices BEHIND the bus: ten solo leaves
                   (mixer(HostOnlyEngine(max_block_frames=64, max_batch=8), 7), 8)):                       # an 8-port mixer... (split: 7 + 1 solo)
        assert e2.cx.plan_kind() == 3 and e2.cx.plan_fused_voices() == nv, (e2.cx.plan_kind(), e2.cx.plan_fused_voices())
        e2.reset_launches()
        e2.process_blocks(8)
        c = e2.launches()
        assert (c["voice_control"], c["leaf_sum"]) == (1, 1) and c["level"] >= 3
        assert e2.violation() == "", e2.violation()
    assert mixer(HostOnlyEngine(max_block_frames=64), 2).cx.plan_kind() == 0          # a 3-port sum is never split, and 3 solo voices are not worth the launches


def test_imported_reference_schedule_selects_the_same_plan_and_levels():
    o = OracleEngine(max_block_frames=128)
    bank(o, chain=True)
    e = HostOnlyEngine(max_block_frames=128)
    e.update = lambda: e.cx.schedule_upload(o.schedule(), o.num_buffers())
    bank(e, chain=True)
    n = HostOnlyEngine(max_block_frames=128)
    bank(n, chain=True)
    assert e.cx.plan_kind() == n.cx.plan_kind() == 2
    assert e.cx.plan_num_levels() == n.cx.plan_num_levels()
    for s in o.schedule():
        assert e.cx.plan_node_level(s["id"]) == n.cx.plan_node_level(s["id"])
        assert e.cx.plan_node_inputs_clear(s["id"]) == [c for _, c in s["in"]] == n.cx.plan_node_inputs_clear(s["id"])


def test_batching_of_a_call_into_launch_sequences():
    # voice-bank plan: one control + one leaf launch per batch of max_batch blocks; the chain plan caps a batch at 64
    e = HostOnlyEngine(max_block_frames=64, max_batch=16)
    bank(e, n_voices=16, radix=4)  # 4 leaves + root
    e.reset_launches()
    e.process_blocks(40)  # 16 + 16 + 8
    c = e.launches()
    assert (c["voice_control"], c["leaf_sum"], c["root_out"], c["chain"], c["level"]) == (3, 3, 3, 0, 0)
    e2 = HostOnlyEngine(max_block_frames=64, max_batch=200)
    bank(e2, n_voices=16, radix=4, chain=True)
    e2.reset_launches()
    e2.process_blocks(130)  # 64 + 64 + 2
    c = e2.launches()
    assert (c["voice_control"], c["chain"], c["leaf_sum"]) == (3, 3, 0)
    # generic executor: one launch per level (and kernel set) per batch; a partial last block is its own batch
    e3 = HostOnlyEngine(max_block_frames=64, max_batch=8, force_generic=True)
    bank(e3, n_voices=4, radix=4)
    e3.reset_launches()
    e3.process_interleaved(64 * 8 + 10)
    c = e3.launches()
    assert c["voice_control"] == 0 and c["level"] >= 2 * 4  # >= 4 node levels x 2 batches


def test_null_handle_is_an_error_return_on_every_entry_point():
    import firewheel_amd._lib as flib

    L = hostonly_lib()
    for name, (res, args) in flib.SIGNATURES.items():
        if not args or args[0] is not C.c_void_p or name == "fwgpu_ctx_create":
            continue
        zeros = [None] + [a() if not hasattr(a, "contents") and a is not C.c_void_p and a is not C.c_char_p else None for a in args[1:]]
        r = getattr(L, name)(*zeros)
        if name in ("fwgpu_ctx_destroy", "fwgpu_stream_close", "fwgpu_bus_exchange_close"):
            continue
        if name == "fwgpu_rccl_comm_destroy":
            assert r == 0  # (destroying nothing is fine, like free(NULL))
            continue
        if name in ("fwgpu_stream_open", "fwgpu_bus_exchange_open", "fwgpu_hip_stream", "fwgpu_rccl_comm_create"):
            assert r is None  # a null stream handle, like fwgpu_ctx_create
            continue
        if name == "fwgpu_last_error":
            assert r == b"null ctx"
            continue
        assert r is not None and r < 0, (name, r)


def test_sampler_message_ring_capacity_and_reset():
    e = HostOnlyEngine(max_block_frames=64)
    s = e.sampler(100.0)
    e.connect_stereo(s, e.graph_out_node)
    e.update()
    L, c = e.cx.L, e.cx.c
    for i in range(128):  # nodes/sampler.rs:14 CHANNEL_CAPACITY
        assert L.fwgpu_sampler_pause(c, s, 0) == 0
    assert L.fwgpu_sampler_play(c, s, 0) == ERR_QUEUE_FULL
    assert b"ring full" in L.fwgpu_last_error(c)
    assert L.fwgpu_node_set_param(c, s, 0, 50.0, 0) == 0  # the gain is an atomic, not a ring message
    e.process_blocks(1)  # the audio thread drains the ring at the next block
    assert L.fwgpu_sampler_play(c, s, 0) == 0


def test_argument_errors_of_the_edit_and_message_calls():
    e = HostOnlyEngine(max_block_frames=64)
    L, c = e.cx.L, e.cx.c
    fp = C.POINTER(C.c_float)
    assert L.fwgpu_add_node(c, 99, 1, 1, None, 0) == ERR_INVALID
    assert L.fwgpu_add_node(c, VOLUME, 65, 65, None, 0) == ERR_INVALID          # core/node.rs:62,69
    assert L.fwgpu_add_node(c, VOLUME, 2, 2, None, 3) == ERR_INVALID            # params missing
    assert L.fwgpu_add_node(c, fwapi.FIR, 2, 2, (C.c_float * 1)(5.0), 1) == ERR_INVALID  # no such impulse response
    v = e.volume(50.0)
    s = e.sampler(100.0)
    assert L.fwgpu_node_set_param(c, v, 3, 1.0, 0) == ERR_INVALID               # unknown param id
    assert L.fwgpu_node_set_param(c, 12345 << 32 | 7, 0, 1.0, 0) == ERR_INVALID  # unknown node
    assert L.fwgpu_sampler_play(c, v, 0) == ERR_INVALID                          # not a sampler
    assert L.fwgpu_sampler_set_sample(c, s, 3, 0, 0) == ERR_INVALID              # unknown sample
    assert L.fwgpu_sampler_set_loop_range(c, s, 7, 0.0, 0.0, 0) == ERR_INVALID
    assert L.fwgpu_remove_node(c, e.graph_in_node) < 0 and L.fwgpu_remove_node(c, e.graph_out_node) < 0
    assert L.fwgpu_set_max_batch(c, 0) == ERR_INVALID
    assert L.fwgpu_sample_create(c, 9, 2, 10, None) == ERR_INVALID               # bad format
    assert L.fwgpu_sample_create(c, PLANAR_F32, 0, 10, None) == ERR_INVALID      # no channels
    assert L.fwgpu_sample_create(c, PLANAR_F32, 2, 10, None) == ERR_INVALID      # null data
    assert L.fwgpu_sample_create(c, PLANAR_F32, 2, 1 << 62, None) == ERR_INVALID  # size overflow
    assert L.fwgpu_process_blocks_device(c, 1, None, 2) == ERR_INVALID           # no schedule yet
    out = (C.c_float * 128)()
    assert L.fwgpu_process_interleaved(c, None, out, 0, 2, 64, 0.0, 0) == 0      # processor.rs:86-89: no schedule -> zeros
    assert L.fwgpu_process_interleaved(c, None, None, 0, 2, 64, 0.0, 0) == ERR_INVALID
    assert L.fwgpu_process_interleaved(c, None, out, 0, 65, 1, 0.0, 0) == ERR_INVALID  # processor.rs:43-44


def test_activation_failure_aborts_the_compile_and_leaves_the_graph_usable():
    e = HostOnlyEngine(max_block_frames=64)
    s = e.sampler(100.0)
    e.connect_stereo(s, e.graph_out_node)
    e.update()
    bad = e.add_node(VOLUME, 2, 3)  # volume.rs:63-65: n_in == n_out
    with pytest.raises(CompileGraphError) as ei:
        e.update()
    assert ei.value.name == "NodeActivationFailed"
    assert e.cx.plan_kind() >= 0  # the previous schedule stays installed (graph.rs:603-609 rollback)
    e.remove_node(bad)
    e.update()
    odd = e.add_node(SUM, 5, 2)  # sum.rs:27-29: inputs divisible by outputs
    with pytest.raises(CompileGraphError):
        e.update()
    e.remove_node(odd)
    e.update()


def test_schedule_upload_validation():
    e = HostOnlyEngine(max_block_frames=64)
    a = e.add_node(DUMMY, 1, 1)
    gi, go = e.graph_in_node, e.graph_out_node
    ok = [{"id": gi, "in": [], "out": []}, {"id": a, "in": [(0, True)], "out": [0]}, {"id": go, "in": [(0, False), (1, True)], "out": []}]
    e.cx.schedule_upload(ok, 2)
    cases = {
        "unknown node": [ok[0], {"id": 77 << 32 | 9, "in": [], "out": []}, ok[2]],
        "port counts": [ok[0], {"id": a, "in": [], "out": [0]}, ok[2]],
        "unwritten buffer": [ok[0], {"id": a, "in": [(1, False)], "out": [0]}, ok[2]],
        "buffer index": [ok[0], {"id": a, "in": [(0, True)], "out": [5]}, ok[2]],
        "graph_in first": [ok[1], ok[0], ok[2]],
        "too short": [ok[0]],
    }
    for what, sched in cases.items():
        with pytest.raises(Exception):
            e.cx.schedule_upload(sched, 2)
    e.cx.schedule_upload(ok, 2)  # still usable


def test_sample_destroy_contract_on_the_host_side():
    e = HostOnlyEngine(max_block_frames=64)
    ir = e.new_sample(PLANAR_F32, 1, np.ones(16, dtype=np.float32))
    f = e.fir(ir)
    with pytest.raises(Exception):
        e.cx.destroy_sample(ir)       # named by a FIR node
    with pytest.raises(Exception):
        e.cx.destroy_sample(42)       # unknown
    e.remove_node(f)
    e.cx.destroy_sample(ir)
    with pytest.raises(Exception):
        e.cx.destroy_sample(ir)       # already destroyed
    with pytest.raises(Exception):
        e.fir(ir)                     # and no longer usable as an impulse response
    assert e.new_sample(PLANAR_F32, 1, np.ones(4, dtype=np.float32)) == ir + 1  # ids are never reused


def test_graph_edits_keep_the_plan_and_grow_buffers():
    # plug voices into spare leaf ports one at a time: every update recompiles, the fused plan stays
    e = HostOnlyEngine(max_block_frames=64, max_batch=4)
    leaf = e.sum(8)
    e.connect_stereo(leaf, e.graph_out_node)
    for v in range(8):
        s = e.sampler(80.0)
        g = e.volume(60.0)
        e.connect_stereo(s, g)
        e.connect_stereo(g, leaf, 2 * v)
        e.update()
        assert e.cx.plan_kind() == 1
        e.process_blocks(3)
    e.cx.set_max_batch(64)
    e.update()
    e.reset_launches()
    e.process_blocks(64)
    assert e.launches()["leaf_sum"] == 1


def test_host_half_fuzz_under_address_and_ub_sanitizers(tmp_path):
    # the GPU fuzz families' generators (random banks / chains / DAGs / effects racks, messages, graph edits, calls of
    # arbitrary length) against the host half built with -fsanitize=address,undefined: any out-of-bounds plan table, group
    # packing or message bookkeeping bug aborts the subprocess
    import os
    import subprocess
    import sys

    def lib(name):
        p = subprocess.check_output(["gcc", "-print-file-name=" + name]).decode().strip()
        return p if os.path.isabs(p) and os.path.exists(p) else None

    asan, ubsan = lib("libasan.so"), lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtimes in this toolchain")
    d = os.path.join(fwapi.ROOT, "tests", "host_harness")
    csrc = os.path.join(fwapi.ROOT, "firewheel_amd", "csrc")
    so = str(tmp_path / "_hostonly_asan.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-Wno-unused-function",
                           "-I", os.path.join(d, "fakehip"), "-I", os.path.join(fwapi.ROOT, "include"), "-o", so,
                           os.path.join(d, "launch_stubs.cpp")] + fwapi.host_sources())
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0", FWGPU_HOSTONLY_ASAN_SO=so)
    r = subprocess.run([sys.executable, os.path.join(d, "asan_fuzz.py"), "30"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok 30" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    # ... and a hostile caller: random entry points with mostly invalid arguments (stale ids, bad ports / kinds, NaN and huge
    # parameters, far-future at_block, odd frame counts) — error returns only, no crash, no sanitizer report
    r = subprocess.run([sys.executable, os.path.join(d, "api_fuzz.py"), "150"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok 150" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("kw", [{}, {"chain": True}, {"clip_in_voice": True}, {"master": True}])
def test_process_calls_allocate_no_device_or_pinned_memory_once_warm(kw):
    # SURVEY 8(b) realtime rules: "no locks/allocs in fwgpu_process_block*" — every device / pinned buffer is sized by
    # fwgpu_update (control thread) or by the first call of a given size; steady callbacks, message bursts of a size seen
    # before and K-batched calls allocate nothing (counted in the fake runtime's hipMalloc / hipHostMalloc)
    e = HostOnlyEngine(max_block_frames=64, max_batch=8)
    smp = bank(e, **kw)
    s = e.new_sample(PLANAR_F32, 2, scenarios.voice_source(1, 900))
    for x in smp:
        e.sampler_set_sample(x, s)
        e.sampler_play(x)

    def traffic(at):
        for x in smp:
            e.set_param(x, 0, 70.0, at_block=at)
            e.sampler_pause(x, at_block=at)
            e.sampler_play(x, at_block=at + 1)

    # warm-up: one call of each shape, with the largest message burst
    traffic(0)
    e.process_blocks(1)
    traffic(2)
    e.process_blocks(20)
    e.process_interleaved(64 * 3 + 5)
    before = hostonly_lib().fwh_alloc_count()
    for i in range(30):
        if i % 7 == 3:
            traffic(i % 5)
        if i % 3 == 0:
            e.process_blocks(1)          # a realtime callback
        elif i % 3 == 1:
            e.process_blocks(20)         # K-batched: 8 + 8 + 4
        else:
            e.process_interleaved(64 * 3 + 5)  # a partial last block
    assert hostonly_lib().fwh_alloc_count() == before


def test_control_dispatch_order_and_cache_carry_on_the_host_side():
    """round 3, host half only (the harness stubs CHECK the tables, tests/host_harness/launch_stubs.cpp): every call with messages
    hands k_voice_control a dispatch order that is a permutation with the voices those messages go to in front; a call without
    messages (and none in the call before) hands it none; an adoption passes the steady-cache carry tables of the OLD plan with
    every index inside its table."""
    import ctypes as C

    from fwapi import HostOnlyEngine, hostonly_lib

    L = hostonly_lib()
    L.fwh_ctl_orders.restype = C.c_ulonglong
    L.fwh_violation_reset()
    e = HostOnlyEngine(max_block_frames=128, max_batch=8)
    voices = scenarios.build_voice_bank(e, 70, radix=32, src_frames=900)
    for vc in voices:
        e.sampler_play(vc["sampler"])
    n0 = L.fwh_ctl_orders()
    e.process_blocks(8)                       # play / set-sample messages for every voice
    assert L.fwh_ctl_orders() == n0 + 1
    e.process_blocks(8)                       # none now, but the call before had them: their glides may continue
    e.process_blocks(8)                       # none, none: identity order, nothing uploaded
    assert L.fwh_ctl_orders() == n0 + 2
    e.set_param(voices[33]["volume"], 0, 20.0, at_block=5)
    e.set_param(voices[2]["pan"], 0, 0.5, at_block=1)
    e.process_blocks(8)
    assert L.fwh_ctl_orders() == n0 + 3
    # an edit that leaves every voice as it was, and one that removes a voice: both adoptions carry caches (stub: bounds)
    x = e.volume(50.0)
    e.remove_node(x)
    e.update()
    e.process_blocks(2)
    e.remove_node(voices[5]["pan"])
    e.update()
    e.set_param(voices[6]["volume"], 0, 70.0)
    e.process_blocks(3)
    assert e.violation() == "", e.violation()


def test_plan_builds_go_out_in_pieces_only_while_a_stream_is_live(monkeypatch):
    """round 3, host half (fwgpu_plan_install.cpp, quiet_window / audio_live): a build with no process call in the last 200 ms
    uploads every table whole; a build right after a process call cuts its uploads into pieces of at most FWGPU_UP_PIECE bytes
    (set to 128 KiB here; the default is 256), each issued when the gate word says no process call is in flight.  The fake HIP counts the asynchronous
    host-to-device copies and remembers the largest."""
    import ctypes as C
    import time

    from fwapi import HostOnlyEngine, hostonly_lib

    L = hostonly_lib()
    for f in (L.fwh_h2d_count, L.fwh_h2d_max):
        f.restype = C.c_ulonglong
    monkeypatch.setenv("FWGPU_UP_PIECE", str(128 * 1024))
    monkeypatch.setenv("FWGPU_UP_DIFF", "0")                                  # (every table whole, every build: the test below covers the diff)
    monkeypatch.setenv("FWGPU_BUILD_ONE_KERNEL", "0")                         # (the calls one by one: the default is one kernel per build, below)
    e = HostOnlyEngine(max_block_frames=256, max_batch=8)
    L.fwh_h2d_reset()
    voices = scenarios.build_voice_bank(e, 3000, radix=32, src_frames=600)   # (build_voice_bank ends in update(): no stream yet)
    whole_n, whole_max = L.fwh_h2d_count(), L.fwh_h2d_max()
    assert whole_max > 128 * 1024, whole_max                                 # some table of 3 000 voices is larger than a piece
    for vc in voices[:8]:
        e.sampler_play(vc["sampler"])
    e.process_blocks(2)                                                       # a stream is live now
    x = e.volume(40.0)
    e.remove_node(x)
    L.fwh_h2d_reset()
    e.update()                                                                # the same tables again, within 200 ms of the call
    live_n, live_max = L.fwh_h2d_count(), L.fwh_h2d_max()
    assert live_max <= 128 * 1024, live_max
    assert live_n > whole_n, (live_n, whole_n)
    time.sleep(0.35)                                                          # the stream has gone quiet: whole again
    x = e.volume(41.0)
    e.remove_node(x)
    L.fwh_h2d_reset()
    e.update()
    assert L.fwh_h2d_max() > 128 * 1024, L.fwh_h2d_max()                      # (the first build also carried the new nodes' states)
    e.process_blocks(1)
    assert e.violation() == "", e.violation()


def test_a_rebuild_uploads_only_the_chunks_of_its_tables_that_changed(monkeypatch):
    """round 3, host half (up() in fwgpu_plan_install.cpp): the tables the device only reads keep a host copy per plan image; a
    build copies the 4 KiB chunks that differ from what the SAME image got two edits ago (the two images alternate).  Replacing one
    voice of a 2 000-voice bank: from the third edit on an update sends a fraction of what the first build sent; the harness stubs
    keep checking every table a launch is handed (a stale chunk would be an index out of its table)."""
    import ctypes as C

    from fwapi import HostOnlyEngine, hostonly_lib

    L = hostonly_lib()
    L.fwh_h2d_total.restype = C.c_ulonglong
    L.fwh_build_applies.restype = C.c_ulonglong
    L.fwh_h2d_count.restype = C.c_ulonglong
    L.fwh_violation_reset()
    for k in ("FWGPU_UP_PIECE", "FWGPU_UP_DIFF", "FWGPU_BUILD_ONE_KERNEL", "FWGPU_QUIET_WAIT_US"):
        monkeypatch.delenv(k, raising=False)                  # (the defaults are what is tested)
    e = HostOnlyEngine(max_block_frames=256, max_batch=8)
    L.fwh_h2d_reset()
    a0 = L.fwh_build_applies()
    voices = scenarios.build_voice_bank(e, 2000, radix=32, src_frames=600)
    first = L.fwh_h2d_total()
    # ... and everything a build does on the device — the changed chunks, the cleared state, the flags — is ONE launch
    # (k_build_apply over a job list), not a runtime call per table
    assert L.fwh_build_applies() == a0 + 1 and L.fwh_h2d_count() == 0, (L.fwh_build_applies() - a0, L.fwh_h2d_count())
    for vc in voices[:40]:
        e.sampler_play(vc["sampler"])
    e.process_blocks(2)
    sent = []
    for k in range(5):
        vc = voices[(k * 977 + 13) % len(voices)]
        x = e.volume(30.0 + k)                     # a node more in front of one voice's pan: the schedule shifts behind it
        e.remove_node(x)
        e.set_param(vc["volume"], 0, 60.0 + k)
        y = e.volume(10.0 + k)
        e.connect_stereo(voices[(k * 31 + 7) % len(voices)]["sampler"], y)
        L.fwh_h2d_reset()
        a1 = L.fwh_build_applies()
        e.update()
        sent.append(L.fwh_h2d_total())
        assert 1 <= L.fwh_build_applies() - a1 <= 12      # (a stream is live: groups of a few microseconds each, not one long kernel)
        e.process_blocks(3)
        assert e.violation() == "", e.violation()
    assert min(sent[2:]) < 0.5 * first, (first, sent)


def test_quiet_window_look_ahead_rule():
    """fwgpu_ctx.h, quiet_next_call_is_due: when a group of a build's device work, no call being in flight, still waits because the
    stream's next call is about to begin (times in ns; margin 60 us)."""
    import ctypes as C

    from fwapi import hostonly_lib

    f = hostonly_lib().fwh_quiet_next_call_is_due
    f.restype = C.c_int
    f.argtypes = [C.c_ulonglong] * 5
    us, M = 1000, 60_000
    start, period, dur = 10_000 * us, 1000 * us, 80 * us              # a callback every millisecond, 80 us long
    assert f(start + 100 * us, start, period, dur, M) == 0             # just over: 900 us of room
    assert f(start + 939 * us, start, period, dur, M) == 0             # 61 us before the next one: still room for a group
    assert f(start + 941 * us, start, period, dur, M) == 1             # 59 us before it: wait for it to come and go
    assert f(start + 1000 * us, start, period, dur, M) == 1            # due now
    assert f(start + 1059 * us, start, period, dur, M) == 1            # a little late: still expected
    assert f(start + 1061 * us, start, period, dur, M) == 0            # overdue by more than the margin: the stream may have stopped
    assert f(start + 500 * us, start, 0, dur, M) == 0                  # no rhythm known yet
    assert f(start + 70 * us, start, 85 * us, 80 * us, M) == 0         # back to back: no gaps to use, the plain rule applies
    assert f(start + 195 * us, start, 199 * us, 80 * us, M) == 0       # gaps shorter than two margins: likewise
    assert f(start + 195 * us, start, 201 * us, 80 * us, M) == 1       # ... just long enough
    assert f(start + 10 * us, start, 300_000 * us, dur, M) == 0        # a "period" of 300 ms is a pause, not a rhythm


def test_port_lists_container_selftest():
    """fwgpu_graph.h, PortInts — the planner's per-node port lists (inline up to 4 ints, heap beyond): copies, moves, growth across
    the inline / heap border, conversion to a vector, life inside a reallocating vector (C++ self-test in the harness)."""
    f = fwapi.hostonly_lib().fwh_portints_selftest
    f.restype = C.c_int
    assert f() == 0


def test_a_build_larger_than_the_upload_arena_goes_out_in_flushes_and_stays_whole(monkeypatch):
    """fwgpu_plan_install.cpp, arena_room / build_apply: 24 000 voices (72 000 nodes) are ~16 MB of tables against an 8 MB pinned
    arena — the pending jobs are applied and waited for whenever the arena fills, and nothing is lost on the way (the harness stubs
    check every table of every launch); the same image's next build then finds its tables unchanged and sends almost nothing."""
    import ctypes as C

    from fwapi import HostOnlyEngine, hostonly_lib

    for k in ("FWGPU_UP_PIECE", "FWGPU_UP_DIFF", "FWGPU_BUILD_ONE_KERNEL", "FWGPU_QUIET_WAIT_US", "FWGPU_BUILD_STREAM"):
        monkeypatch.delenv(k, raising=False)
    L = hostonly_lib()
    L.fwh_build_applies.restype = C.c_ulonglong
    L.fwh_h2d_total.restype = C.c_ulonglong
    L.fwh_violation_reset()
    e = HostOnlyEngine(max_block_frames=128, max_batch=4)
    ends = []
    for v in range(24000):
        s, vol, pan = e.sampler(100.0), e.volume(50.0), e.pan(0.1)
        e.connect_stereo(s, vol)
        e.connect_stereo(vol, pan)
        ends.append(pan)
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
    scenarios.connect_through_master(e, level[0], ())
    L.fwh_h2d_reset()
    a0 = L.fwh_build_applies()
    e.update()
    first = L.fwh_h2d_total()
    assert first > 12 << 20 and L.fwh_build_applies() - a0 >= 2, (first, L.fwh_build_applies() - a0)   # more than one arena's worth
    assert e.cx.plan_kind() == 1 and e.cx.plan_fused_voices() == 24000
    e.process_blocks(4)
    assert e.violation() == "", e.violation()
    sent = []
    for k in range(3):                            # edits that leave every table as it was (a node nobody is connected to comes and goes)
        x = e.volume(3.0)
        e.remove_node(x)
        L.fwh_h2d_reset()
        e.update()
        sent.append(L.fwh_h2d_total())
        e.process_blocks(4)
        assert e.violation() == "", e.violation()
    assert sent[0] > 4 << 20 and max(sent[1:]) < 1 << 20, sent   # the OTHER image's first build is whole; after that: chunks


def test_lazy_records_host_bookkeeping():
    """Round 4, lazy records (fwgpu_types.h LazyRec): a message-free multi-block call of a plain voice-bank plan is rendered without a
    control launch once the host has seen the last control launch's horizon — and node state is flushed before anything else
    touches it.  The stubs model a device on which every voice ends every control launch steady and check the book-keeping: the
    block offset a lazy launch names, "no control / realtime launch over unflushed blocks", "a flush names exactly them"."""
    e = HostOnlyEngine(max_block_frames=64, max_batch=8)
    smp = bank(e)
    assert e.cx.plan_kind() == 1
    for s in smp:
        e.sampler_play(s)
    e.reset_launches()
    e.process_blocks(8)                      # messages: control launch (+ publish)
    assert e.launches()["voice_control"] == 1 and e.cx.lazy_stats() == (0, 1)
    e.process_blocks(8)                      # the call before had messages (glides may continue): control again
    e.process_blocks(8)                      # quiet, and the publish has been seen: lazy
    e.process_blocks(20)                     # 8 + 8 + 4: three lazy batches, block offsets 8, 16, 24 (checked by the stub)
    lazy, ctl = e.cx.lazy_stats()
    assert (lazy, ctl) == (4, 2), (lazy, ctl)
    assert e.launches()["voice_control"] == 2 and e.violation() == ""
    e.set_param(smp[0], 0, 50.0)             # a message: flush (28 blocks), then control
    e.process_blocks(3)
    assert e.cx.lazy_stats() == (4, 3) and e.violation() == ""
    e.process_blocks(2)                      # hot_prev: control
    e.process_blocks(2)                      # lazy again
    assert e.cx.lazy_stats() == (5, 4)
    e.process_interleaved(64)                # a one-block realtime-sized call: flush first, the one-launch kernel, LazyRecs spent
    assert e.violation() == ""
    e.process_blocks(4)                      # so this one runs control
    assert e.cx.lazy_stats() == (5, 5)
    e.process_blocks(4)
    assert e.cx.lazy_stats() == (6, 5)
    e.cx.set_max_batch(4)                    # a recompile (same graph, new tables): the adoption flushes the 4 lazy blocks against the OLD
    e.update()                               # plan's LazyRecs, the new plan starts with a control launch
    assert e.cx.plan_kind() == 1
    e.process_blocks(4)
    e.process_blocks(4)
    lazy, ctl = e.cx.lazy_stats()
    assert ctl == 6 and lazy == 7 and e.violation() == "", (lazy, ctl, e.violation())
    # chain plans (round 6): the same rule — k_chain derives its records from the LazyRecs, and the flush names the voices, whose delay
    # lines' positions it moves (the stubs check both: launch_chain's block offset, launch_lazy_flush's voice table)
    e2 = HostOnlyEngine(max_block_frames=64, max_batch=8)
    smp2 = bank(e2, chain=True)
    assert e2.cx.plan_kind() == 2
    for s_ in smp2:
        e2.sampler_play(s_)
    for _ in range(4):
        e2.process_blocks(8)                 # control, control (hot_prev), lazy, lazy
    assert e2.cx.lazy_stats() == (2, 2) and e2.violation() == "", (e2.cx.lazy_stats(), e2.violation())
    e2.process_blocks(20)                    # three lazy batches: offsets 16, 24, 32
    assert e2.cx.lazy_stats() == (5, 2) and e2.violation() == "", (e2.cx.lazy_stats(), e2.violation())
    e2.set_param(smp2[1], 0, 40.0)           # a message: flush (36 blocks, with the voice table), control
    e2.process_blocks(4)
    assert e2.cx.lazy_stats() == (5, 3) and e2.violation() == "", (e2.cx.lazy_stats(), e2.violation())


def test_a_list_of_messages_in_one_call():
    """fwgpu_node_set_params (include/fwgpu.h): n messages in order through one foreign call; it stops at the first message that fails,
    returns that error, and the ones in front of it stay sent."""
    import ctypes as C

    e = HostOnlyEngine(max_block_frames=64, max_batch=8)
    smp = bank(e)
    L = e.cx.L
    for s_ in smp:
        e.sampler_play(s_)
    e.process_blocks(2)
    n = 5
    nodes = (C.c_int64 * n)(*[smp[i] for i in range(n)])
    params = (C.c_int * n)(*([0] * n))
    values = (C.c_float * n)(*[10.0 * (i + 1) for i in range(n)])
    at = (C.c_uint32 * n)(*[i for i in range(n)])
    seen = hostonly_lib().fwh_cmds_seen
    seen.restype = C.c_ulonglong
    applied0 = seen()
    assert L.fwgpu_node_set_params(e.cx.c, n, nodes, params, values, at) == 0
    assert L.fwgpu_node_set_params(e.cx.c, 0, None, None, None, None) == 0
    assert L.fwgpu_node_set_params(e.cx.c, 2, None, params, values, at) < 0          # a null list
    bad = (C.c_int64 * 3)(smp[0], 1 << 40, smp[1])                                   # the second node does not exist
    assert L.fwgpu_node_set_params(e.cx.c, 3, bad, params, values, at) < 0
    assert b"unknown node" in L.fwgpu_last_error(e.cx.c)
    e.process_blocks(8)                                                               # 5 + 1 messages reach the control kernel (the stub counts them)
    assert e.violation() == "", e.violation()
    assert seen() - applied0 == 6, seen() - applied0


def test_process_interleaved_begin_end_ticket_rules_on_the_host_harness():
    """fwgpu_process_interleaved_begin / _end (include/fwgpu.h): at most two calls in flight, tickets ended in the order they were
    begun, `output` filled on every return of `end`, the no-schedule case (processor.rs:86-89) is silence through this pair too."""
    from firewheel_amd._lib import FwgpuError

    e = HostOnlyEngine(max_block_frames=64, max_batch=8)
    t = e.cx.process_interleaved_begin(None, 0, 2, 64 * 3)       # no plan yet: zeros
    out = e.cx.process_interleaved_end(t)
    assert out.shape == (64 * 3 * 2,) and not out.any()
    s = e.sampler()
    e.connect_stereo(s, e.graph_out_node)
    e.update()
    a = e.cx.process_interleaved_begin(None, 0, 2, 64 * 5)
    b = e.cx.process_interleaved_begin(None, 0, 2, 64 * 2)
    assert b[0] == a[0] + 1
    with pytest.raises(FwgpuError):
        e.cx.process_interleaved_begin(None, 0, 2, 64)             # a third call in flight
    with pytest.raises(FwgpuError):
        e.cx.process_interleaved_end(b)                            # not the oldest
    assert e.cx.process_interleaved_end(a).shape == (64 * 5 * 2,)
    with pytest.raises(FwgpuError):
        e.cx.process_interleaved_end(a)                            # ended already
    c = e.cx.process_interleaved_begin(None, 0, 2, 64 * 9)         # slot of `a` again, larger: grows
    assert e.cx.process_interleaved_end(b).shape == (64 * 2 * 2,)
    assert e.cx.process_interleaved_end(c).shape == (64 * 9 * 2,)
    assert np.asarray(e.process_interleaved(64)).shape == (128,)   # the synchronous call beside it
    # a ticket nobody ends must not wedge `begin` for ever — cancel abandons every ticket up to and including the one named
    a = e.cx.process_interleaved_begin(None, 0, 2, 64 * 2)
    b = e.cx.process_interleaved_begin(None, 0, 2, 64 * 2)
    with pytest.raises(FwgpuError):
        e.cx.process_interleaved_begin(None, 0, 2, 64)
    e.cx.process_interleaved_cancel(b)                              # sweeps `a` too
    with pytest.raises(FwgpuError):
        e.cx.process_interleaved_end(a)                            # gone
    with pytest.raises(FwgpuError):
        e.cx.process_interleaved_cancel(b)                          # nothing in flight under that id
    c = e.cx.process_interleaved_begin(None, 0, 2, 64 * 4)
    d = e.cx.process_interleaved_begin(None, 0, 2, 64)
    e.cx.process_interleaved_cancel(c)                              # the oldest only
    assert e.cx.process_interleaved_end(d).shape == (64 * 2,)
    # `end` with a null buffer for a ticket with frames is refused and leaves the ticket in flight (include/fwgpu.h)
    f = e.cx.process_interleaved_begin(None, 0, 2, 64)
    assert e.cx.L.fwgpu_process_interleaved_end(e.cx.c, f[0], None) < 0
    assert e.cx.process_interleaved_end(f).shape == (64 * 2,)
    assert e.violation() == ""


def test_a_one_voice_replace_uploads_a_few_chunks_not_the_tables(monkeypatch):
    """Round 6 (contract: context.rs:93-137, swap at processor.rs:167-206).  examples/host_c/fw_edit_race's edit on the
    host-only harness: config 3's bank — 4 096 voices of sampler -> biquad -> delay -> gain under 32-port mixers — with one voice after
    another (scattered: (e * 977 + 13) % 4096) removed and rebuilt into its mixer port.  build_plan's canonical order (fwgpu_graph.cpp)
    keeps every other voice's entries where they were, so from the third edit on (both plan images have been built once) an edit
    sends <= 64 KB of the 1.1 MB of tables; with the reference's order (FWGPU_PLAN_ORDER=reference) it sent 0.4-0.6 MB."""
    import ctypes as C

    from fwapi import HostOnlyEngine, hostonly_lib

    for k in ("FWGPU_UP_PIECE", "FWGPU_UP_DIFF", "FWGPU_BUILD_ONE_KERNEL", "FWGPU_QUIET_WAIT_US", "FWGPU_BUILD_STREAM", "FWGPU_PLAN_ORDER"):
        monkeypatch.delenv(k, raising=False)
    L = hostonly_lib()
    L.fwh_h2d_total.restype = C.c_ulonglong

    def run(n_edits):
        e = HostOnlyEngine(max_block_frames=512, max_batch=8)

        def voice(v):
            s, b, d, g = e.sampler(90.0), e.biquad(0, 1000.0 + v, 0.7), e.delay(0.01, feedback=0.2, mix=0.5), e.volume(60.0)
            e.connect_stereo(s, b)
            e.connect_stereo(b, d)
            e.connect_stereo(d, g)
            return [s, b, d, g]

        N = 4096
        voices = [voice(v) for v in range(N)]
        leaves = []
        for i in range(0, N, 32):
            m = e.sum(32)
            for k in range(32):
                e.connect_stereo(voices[i + k][3], m, 2 * k)
            leaves.append(m)
        tops = []
        for i in range(0, len(leaves), 32):
            m = e.sum(32)
            for k, n in enumerate(leaves[i:i + 32]):
                e.connect_stereo(n, m, 2 * k)
            tops.append(m)
        root = e.sum(len(tops))
        for k, n in enumerate(tops):
            e.connect_stereo(n, root, 2 * k)
        e.connect_stereo(root, e.graph_out_node)
        e.update()
        assert e.cx.plan_kind() == 2 and e.cx.plan_fused_voices() == N
        e.process_blocks(2)
        sent = []
        for it in range(n_edits):
            v = (it * 977 + 13) % N
            for n in voices[v]:
                e.remove_node(n)
            voices[v] = voice(N + it)
            e.connect_stereo(voices[v][3], leaves[v // 32], 2 * (v % 32))
            L.fwh_h2d_reset()
            e.update()
            sent.append(L.fwh_h2d_total())
            e.process_blocks(1)
            assert e.violation() == "", e.violation()
        return sent

    sent = run(7)
    assert max(sent[2:]) <= 64 << 10, sent
    monkeypatch.setenv("FWGPU_PLAN_ORDER", "reference")
    ref = run(5)
    assert min(ref[2:]) > 200 << 10, ref   # (what the canonical order is for)


def test_resampler_plan_lazy_calls_on_the_host_harness():
    """Round 6: lazy records for resampler plans, the host's side — the launch stubs check the block offset every lazy k_leaf_rs launch
    names, that its templates are the LazyRecs' copy, and that the flush names no voice table (no delay lines to move).  (The fake
    device reports every voice steady with an unbounded horizon.)"""
    from fwapi import HostOnlyEngine
    from test_gpu_benched_shapes import rs_quiet_run

    e = HostOnlyEngine(max_block_frames=256, max_batch=8)
    _, marks = rs_quiet_run(e, 256)
    assert e.cx.plan_kind() == 1 and e.violation() == "", e.violation()
    assert marks[-1][0] >= 6, marks
