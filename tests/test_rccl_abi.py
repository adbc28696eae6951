"""The mix bus over RCCL behind the C ABI (include/fwgpu.h "the mix bus over RCCL"; the node being computed:
nodes/sum.rs:111-133).  CPU tier: the host harness library (real host code, fake HIP runtime) against tests/host_harness/fakerccl.cpp —
N ranks as threads of this process, each with its own context and communicator: the dlopen path, the unique-id hand-over, the
in-place all-reduce, the all-gather's slot layout (16-byte flag slots, in-place send).  GPU tier: the REAL librccl at world 1 on one
MI355X (a communicator of one rank is legal): the all-reduce leaves the bus as it was, all-gather + ordered sum reproduces it bit for
bit and reports its silence flags — what can be checked without a second GPU; N > 1 on hardware stays unmeasured."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

import fwapi

ROOT = fwapi.ROOT
ID_BYTES = 128


def fake_rccl():
    d = os.path.join(ROOT, "tests", "host_harness")
    so, src = os.path.join(d, "_fakerccl.so"), os.path.join(d, "fakerccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wall", "-o", so, src, "-lpthread"])
    return so


def make_ctx(L):
    c = L.fwgpu_ctx_create(0, 48000, 64, 0, 2, None)
    assert c
    return c


@pytest.mark.parametrize("world", [1, 2, 5, 8])
def test_allreduce_and_allgather_through_the_abi_on_the_host_harness(world):
    os.environ["FWGPU_RCCL_LIB"] = fake_rccl()  # (read when the harness library first loads an RCCL: once per process)
    L = fwapi.hostonly_lib()
    uid = (C.c_uint8 * ID_BYTES)()
    assert L.fwgpu_rccl_unique_id(uid) == 0, L.fwgpu_rccl_last_error()
    n = 4 * 64 * 2  # 4 blocks x 64 frames x 2 channels
    rng = np.random.default_rng(5)
    parts = [rng.standard_normal(n).astype(np.float32) for _ in range(world)]
    sil = [np.zeros(4 * 2, dtype=np.uint8) for _ in range(world)]
    sil[world - 1][3] = 1
    expect = parts[0].copy()
    for r in range(1, world):
        expect = expect + parts[r]  # rank order, one f32 rounding per add: the fake ring's order
    got, errs = [None] * world, []

    def rank(r):
        try:
            cx = make_ctx(L)
            m = L.fwgpu_rccl_comm_create(cx, uid, world, r)
            assert m, L.fwgpu_last_error(cx)
            w, k = C.c_uint32(), C.c_uint32()
            assert L.fwgpu_rccl_comm_info(m, C.byref(w), C.byref(k)) == 0 and (w.value, k.value) == (world, r)
            bus = parts[r].copy()
            assert L.fwgpu_bus_allreduce_rccl(m, bus.ctypes.data_as(C.c_void_p), n) == 0, L.fwgpu_last_error(cx)
            got[r] = bus
            out = np.full(n, np.nan, dtype=np.float32)
            osil = np.full(8, 9, dtype=np.uint8)
            for _ in range(2):  # (second call: the scratch is reused)
                rc = L.fwgpu_bus_allgather_ordered(m, parts[r].ctypes.data_as(C.c_void_p), sil[r].ctypes.data_as(C.c_void_p),
                                                   out.ctypes.data_as(C.c_void_p), osil.ctypes.data_as(C.c_void_p), n, 64, 2)
                assert rc == 0, L.fwgpu_last_error(cx)
            assert L.fwgpu_bus_allgather_ordered(m, parts[r].ctypes.data_as(C.c_void_p), None, out.ctypes.data_as(C.c_void_p), None, n + 2, 64, 2) < 0
            assert L.fwgpu_rccl_comm_destroy(m) == 0
            L.fwgpu_ctx_destroy(cx)
        except BaseException as ex:  # noqa: BLE001
            errs.append((r, repr(ex)))

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errs, errs
    assert L.fwh_violation() in (b"", None), L.fwh_violation()
    for r in range(world):
        assert np.array_equal(got[r].view(np.uint32), expect.view(np.uint32)), r


def test_bad_arguments_are_refused_without_a_collective():
    os.environ["FWGPU_RCCL_LIB"] = fake_rccl()
    L = fwapi.hostonly_lib()
    cx = make_ctx(L)
    uid = (C.c_uint8 * ID_BYTES)()
    assert L.fwgpu_rccl_unique_id(uid) == 0
    assert not L.fwgpu_rccl_comm_create(cx, uid, 0, 0)
    assert not L.fwgpu_rccl_comm_create(cx, uid, 2, 2)
    assert not L.fwgpu_rccl_comm_create(cx, None, 1, 0)
    assert L.fwgpu_rccl_unique_id(None) < 0
    assert L.fwgpu_bus_allreduce_rccl(None, None, 4) < 0
    assert L.fwgpu_rccl_comm_destroy(None) == 0
    L.fwgpu_ctx_destroy(cx)


@pytest.mark.gpu
def test_real_rccl_at_world_one_on_the_device():
    import torch

    import firewheel_amd as fa

    L = fa.load_library()
    cx = fa.FirewheelGpuCtx(48000, 256, 0, 2)
    uid = (C.c_uint8 * ID_BYTES)()
    assert L.fwgpu_rccl_unique_id(uid) == 0, L.fwgpu_rccl_last_error()
    m = L.fwgpu_rccl_comm_create(cx.c, uid, 1, 0)
    assert m, L.fwgpu_last_error(cx.c)
    n = 8 * 256 * 2
    bus = torch.randn(n, device="cuda", dtype=torch.float32)
    bus.view(8, 256, 2)[2, :, 1] = 0.0  # block 2, channel 1 is silent: cleared AND flagged, as fwgpu_process_blocks_device_flags leaves it
    ref = bus.clone()
    sil = torch.zeros(8 * 2, dtype=torch.uint8, device="cuda")
    sil[5] = 1  # a one-port sum copies the port and passes its flag on (sum.rs:58-65)
    torch.cuda.synchronize()
    assert L.fwgpu_bus_allreduce_rccl(m, C.c_void_p(bus.data_ptr()), n) == 0, L.fwgpu_last_error(cx.c)
    out = torch.full((n,), float("nan"), device="cuda")
    osil = torch.full((16,), 7, dtype=torch.uint8, device="cuda")
    assert L.fwgpu_bus_allgather_ordered(m, C.c_void_p(ref.data_ptr()), C.c_void_p(sil.data_ptr()), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(osil.data_ptr()), n, 256, 2) == 0, L.fwgpu_last_error(cx.c)
    cx.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(bus.view(torch.int32), ref.view(torch.int32))
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    assert osil.cpu().tolist() == sil.cpu().tolist()
    assert L.fwgpu_rccl_comm_destroy(m) == 0
    cx.close()
