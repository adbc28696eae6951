"""Voice sharding + mix-bus reduction for multi-GPU runs (BASELINE config 5).

Voices are independent until the sum tree, so the shard unit is a contiguous voice range per rank; each rank
owns its voices' samples, node state and sub-tree and produces one stereo partial bus per block.  The only
exchange step is the top-level sum of the R partial buses — in the reference that is one SumNode with R
stereo ports (nodes/sum.rs:111-133, sequential in port order).  Two reductions are provided:

* `reduce_bus_allreduce`  — one `all_reduce(SUM)` (RCCL ring/tree order: equal to the reference up to f32
  re-association; exact for 2 ranks because a+b is commutative);
* `reduce_bus_ordered`    — `all_gather` + accumulation in rank order: bit-identical to the reference's R-port
  SumNode on every rank.
Both work on any torch tensor/device, which is how the world_size-2 gloo test exercises them on CPU.

`BusReducer` pipelines either of them behind the compute of the next step; `ExchangeReducer` is the third way and the
default of bench.py: no collective library at all — libfwgpu's own one-shot exchange over peer-mapped slots
(fwgpu_bus_exchange_*, SURVEY §8e path 2), torch.distributed only carries the 128-byte handles once.  The ordered and the
exchange reductions take the shards' per-(block, channel) SILENCE FLAGS along (fwgpu_process_blocks_device_flags), because
the reference's n-port SumNode skips silent ports (sum.rs:122-124): with them the reduced bus is the single-process graph's,
bit for bit, sign of zero included.
"""


def voice_range(rank, world, total_voices):
    """contiguous split, first `total % world` ranks get one extra voice"""
    q, r = divmod(total_voices, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def voice_seed(global_voice):
    """seed of a voice's synthetic source: depends on the GLOBAL voice id only, so the inputs are the same for
    every GPU count."""
    return 0xF1EE0000 + int(global_voice)


def reduce_bus_allreduce(bus, dist, group=None):
    dist.all_reduce(bus, op=dist.ReduceOp.SUM, group=group)
    return bus


def ordered_sum(parts, out, cx=None, sils=None, frames=0, n_ch=2):
    """the R-port SumNode over the partial buses, in port (= rank) order, into `out` (may be parts[0]).  On the device
    this is ONE kernel of libfwgpu on the ctx stream (fwgpu_bus_sum_ordered[_flags]: every part's quad in flight before the
    first add); host tensors (the gloo tests) take the same sum through torch.  `sils[r]` = rank r's silence flags, uint8
    [blocks * n_ch] (or None: no port is ever silent); `frames` = frames per block.
    NOTE (device path): the kernel runs on the CTX stream — the parts must be complete on that stream, which they are when
    the ctx was created on the torch stream the gather was waited on (bench.py does that); a ctx with its own stream needs
    cx.synchronize() / an event between the two."""
    if cx is not None and out.is_cuda:
        if sils is None:
            cx.bus_sum_ordered([p.data_ptr() for p in parts], out.data_ptr(), out.numel())
        else:
            cx.bus_sum_ordered([p.data_ptr() for p in parts], out.data_ptr(), out.numel(), [s.data_ptr() for s in sils], None, frames, n_ch)
        return out
    world = len(parts)
    if sils is None or frames <= 0:
        if out is not parts[0]:
            out.copy_(parts[0])         # sum.rs:117 out = in0
        for p in parts[1:]:             # sum.rs:119-131 out += in_p, port order
            out += p
        return out
    import torch

    per = frames * n_ch
    blocks = (out.numel() + per - 1) // per
    pad = blocks * per - out.numel()
    v = [torch.nn.functional.pad(p, (0, pad)).view(blocks, frames, n_ch) for p in parts]
    s = [x.view(blocks, n_ch).bool() for x in sils]
    all_sil = torch.stack(s).all(dim=0).all(dim=1)                     # sum.rs:52-56, per block
    acc = v[0].clone()                                                 # :117 (copied even if silent)
    for r in range(1, world):
        if world in (2, 3, 4):                                         # :67-110 unmasked
            acc = acc + v[r]
        else:                                                          # :122-124 silent ports skipped
            acc = torch.where(s[r][:, None, :], acc, acc + v[r])
    acc = torch.where(all_sil[:, None, None], torch.zeros_like(acc), acc)
    out.copy_(acc.reshape(-1)[:out.numel()])
    return out


def reduce_bus_ordered(bus, dist, group=None):
    import torch

    world = dist.get_world_size(group)
    parts = [torch.empty_like(bus) for _ in range(world)]
    dist.all_gather(parts, bus, group=group)
    return ordered_sum(parts, bus)


class BusReducer(object):
    """The mix bus is a sink (nothing in a shard reads it back), so the reduction of step i can run while step i+1
    computes: each step writes its partial bus into the next of `bufs` (>= 2 buffers), `submit(i)` starts the
    collective on that buffer asynchronously (RCCL's own stream on a GPU) and `wait(i)` is called only right before
    the buffer is overwritten again — or read.  `mode` as in bench.py: "allreduce" or "ordered" (bit-exact)."""

    def __init__(self, dist, bufs, mode="allreduce", group=None, cx=None, sils=None, frames=0, n_ch=2):
        """sils (ordered mode): one uint8 tensor per bus buffer, the shard's silence flags [blocks * n_ch] of what the buffer
        holds; gathered next to the bus and handed to the sum (sum.rs:122-124).  frames = frames per block."""
        import torch

        self.dist, self.bufs, self.mode, self.group, self.cx = dist, list(bufs), mode, group, cx
        self.sils, self.frames, self.n_ch = (list(sils) if sils is not None else None), frames, n_ch
        self.works = [None] * len(self.bufs)
        self.parts = None
        if mode == "ordered":
            world = dist.get_world_size(group)
            # one flat gather buffer per bus buffer: rank r's bus lands in slot r (the slots are what the sum kernel reads)
            self.flat = [torch.empty(world * b.numel(), dtype=b.dtype, device=b.device) for b in self.bufs]
            self.parts = [[f[r * b.numel():(r + 1) * b.numel()] for r in range(world)] for f, b in zip(self.flat, self.bufs)]
            if self.sils is not None:
                self.sflat = [torch.empty(world * x.numel(), dtype=x.dtype, device=x.device) for x in self.sils]
                self.sparts = [[f[r * x.numel():(r + 1) * x.numel()] for r in range(world)] for f, x in zip(self.sflat, self.sils)]

    def submit(self, i):
        assert self.works[i] is None, "buffer %d is still being reduced" % i
        if self.mode == "allreduce":
            self.works[i] = [self.dist.all_reduce(self.bufs[i], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)]
        else:
            if self.bufs[i].is_cuda:
                w = [self.dist.all_gather_into_tensor(self.flat[i], self.bufs[i], group=self.group, async_op=True)]
                if self.sils is not None:
                    w.append(self.dist.all_gather_into_tensor(self.sflat[i], self.sils[i], group=self.group, async_op=True))
            else:  # gloo (CPU tests)
                w = [self.dist.all_gather(self.parts[i], self.bufs[i], group=self.group, async_op=True)]
                if self.sils is not None:
                    w.append(self.dist.all_gather(self.sparts[i], self.sils[i], group=self.group, async_op=True))
            self.works[i] = w

    def wait(self, i):
        w = self.works[i]
        if w is None:
            return self.bufs[i]
        for x in w:
            x.wait()  # on a GPU: the current stream waits for the collective, the host does not block
        self.works[i] = None
        if self.mode == "ordered":
            ordered_sum(self.parts[i], self.bufs[i], self.cx, self.sparts[i] if self.sils is not None else None, self.frames, self.n_ch)
        return self.bufs[i]

    def wait_all(self):
        for i in range(len(self.bufs)):
            self.wait(i)


def exchange_handles(dist, handle, group=None):
    """every rank's exchange handle, in rank order (128 bytes each, once per exchange: any side channel would do)"""
    world = dist.get_world_size(group)
    got = [None] * world
    dist.all_gather_object(got, bytes(handle), group=group)
    return got


class ExchangeReducer(object):
    """BusReducer's interface over libfwgpu's own exchange (fwgpu_bus_exchange_*): `submit(i)` stores buffer i (+ its silence
    flags) into this rank's slot on every rank and adds the R slots that arrive here, in rank order, into `outs[i]` — two
    kernels on the ctx stream, no collective library, no host round trip; `wait(i)` has nothing to wait for on the host (the
    device waits inside the reduce kernel).  torch.distributed is only the side channel for the handles."""

    mode = "exchange"

    def __init__(self, dist, bufs, cx, outs, sils=None, frames=0, n_ch=2, group=None, timeout_ms=None):
        """Collective over the group: every rank either gets a connected exchange or ALL of them raise (a rank that cannot
        open or map — no dmabuf IPC, no peer access — must not leave the others waiting inside a barrier)."""
        self.bufs, self.outs, self.sils, self.frames, self.n_ch, self.cx = list(bufs), list(outs), sils, frames, n_ch, cx
        n = max(b.numel() for b in self.bufs)
        nsil = max(x.numel() for x in sils) if sils is not None else 0
        self.x, err = None, None
        try:
            self.x = cx.open_bus_exchange(dist.get_rank(group), dist.get_world_size(group), n, nsil)
            if timeout_ms:
                self.x.set_timeout_ms(timeout_ms)
            handle = self.x.export()
        except Exception as ex:  # noqa: BLE001 — whatever it is, the peers must hear about it
            handle, err = None, repr(ex)
        handles = exchange_handles(dist, handle if handle is not None else b"", group)
        if err is None and all(len(h) > 0 for h in handles):
            try:
                self.x.connect_all(handles)
            except Exception as ex:  # noqa: BLE001
                err = repr(ex)
        elif err is None:
            err = "a peer could not open its exchange"
        errs = [None] * dist.get_world_size(group)
        dist.all_gather_object(errs, err, group=group)  # (also the barrier: every rank has mapped every region before the first store)
        bad = [(r, e) for r, e in enumerate(errs) if e]
        if bad:
            if self.x is not None:
                self.x.close()
                self.x = None
            raise RuntimeError("bus exchange unavailable: " + "; ".join("rank %d: %s" % be for be in bad))

    def submit(self, i):
        b = self.bufs[i]
        if self.sils is not None:
            s = self.sils[i]
            self.x.step(b.data_ptr(), self.outs[i].data_ptr(), b.numel(), s.data_ptr(), None, s.numel() // self.n_ch, self.frames, self.n_ch)
        else:
            self.x.step(b.data_ptr(), self.outs[i].data_ptr(), b.numel())

    def wait(self, i):
        return self.outs[i]

    def wait_all(self):
        self.x.status()  # waits for the stream; raises if a peer did not arrive within the time budget

    def close(self, dist=None, group=None):
        self.cx.synchronize()
        if dist is not None:
            dist.barrier(group=group)  # nobody unmaps a region a peer may still be storing into
        self.x.close()


class AbiRcclReducer(object):
    """BusReducer's interface over libfwgpu's OWN RCCL calls (include/fwgpu.h "the mix bus over RCCL": fwgpu_bus_allreduce_rccl /
    fwgpu_bus_allgather_ordered — librccl dlopen'ed by the library, a communicator built from a unique id): what a Rust or C host
    bound to the header gets.  torch.distributed is only the side channel that carries rank 0's 128-byte id.  mode "allreduce_abi"
    reduces buffer i in place; "ordered_abi" all-gathers buses + silence flags and adds them in rank order into `outs[i]`
    (bit-exact).  Both run on the ctx stream: `wait` has nothing to wait for on the host."""

    def __init__(self, dist, bufs, cx, mode="allreduce_abi", outs=None, sils=None, frames=0, n_ch=2, group=None):
        import ctypes as C

        assert mode in ("allreduce_abi", "ordered_abi")
        self.mode, self.bufs, self.cx, self.sils, self.frames, self.n_ch = mode, list(bufs), cx, sils, frames, n_ch
        self.outs = list(outs) if outs is not None else self.bufs
        L = cx.L
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = (C.c_uint8 * 128)()
        err = None
        if rank == 0 and L.fwgpu_rccl_unique_id(uid) != 0:
            err = (L.fwgpu_rccl_last_error() or b"").decode(errors="replace")
        box = [bytes(uid) if err is None else None]
        dist.broadcast_object_list(box, src=0, group=group)
        self.m = None
        if box[0] is not None:
            got = (C.c_uint8 * 128).from_buffer_copy(box[0])
            self.m = L.fwgpu_rccl_comm_create(cx.c, got, world, rank)  # collective
            if not self.m:
                err = (L.fwgpu_last_error(cx.c) or b"").decode(errors="replace")
        else:
            err = err or "rank 0 could not make an RCCL unique id"
        errs = [None] * world
        dist.all_gather_object(errs, err, group=group)
        bad = [(r, e) for r, e in enumerate(errs) if e]
        if bad:
            self.close()
            raise RuntimeError("RCCL through the C ABI unavailable: " + "; ".join("rank %d: %s" % be for be in bad))

    def submit(self, i):
        import ctypes as C

        L, b = self.cx.L, self.bufs[i]
        if self.mode == "allreduce_abi":
            self.cx._check(L.fwgpu_bus_allreduce_rccl(self.m, C.c_void_p(b.data_ptr()), b.numel()))
        else:
            s = self.sils[i] if self.sils is not None else None
            self.cx._check(L.fwgpu_bus_allgather_ordered(self.m, C.c_void_p(b.data_ptr()), C.c_void_p(s.data_ptr()) if s is not None else None,
                                                         C.c_void_p(self.outs[i].data_ptr()), None, b.numel(), self.frames, self.n_ch))

    def wait(self, i):
        return self.outs[i]

    def wait_all(self):
        self.cx.synchronize()

    def close(self, dist=None, group=None):
        if self.m:
            self.cx.synchronize()
            if dist is not None:
                dist.barrier(group=group)
            self.cx.L.fwgpu_rccl_comm_destroy(self.m)
            self.m = None
