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


def ordered_sum(parts, out, cx=None):
    """the R-port SumNode over the partial buses, in port (= rank) order, into `out` (may be parts[0]).  On the device
    this is ONE kernel of libfwgpu on the ctx stream (fwgpu_bus_sum_ordered: every part's quad in flight before the
    first add); host tensors (the gloo tests) take the same sum through torch."""
    if cx is not None and out.is_cuda:
        cx.bus_sum_ordered([p.data_ptr() for p in parts], out.data_ptr(), out.numel())
        return out
    if out is not parts[0]:
        out.copy_(parts[0])         # sum.rs:117 out = in0
    for p in parts[1:]:             # sum.rs:119-131 out += in_p, port order
        out += p
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

    def __init__(self, dist, bufs, mode="allreduce", group=None, cx=None):
        import torch

        self.dist, self.bufs, self.mode, self.group, self.cx = dist, list(bufs), mode, group, cx
        self.works = [None] * len(self.bufs)
        self.parts = None
        if mode == "ordered":
            world = dist.get_world_size(group)
            # one flat gather buffer per bus buffer: rank r's bus lands in slot r (the slots are what the sum kernel reads)
            self.flat = [torch.empty(world * b.numel(), dtype=b.dtype, device=b.device) for b in self.bufs]
            self.parts = [[f[r * b.numel():(r + 1) * b.numel()] for r in range(world)] for f, b in zip(self.flat, self.bufs)]

    def submit(self, i):
        assert self.works[i] is None, "buffer %d is still being reduced" % i
        if self.mode == "allreduce":
            self.works[i] = self.dist.all_reduce(self.bufs[i], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            if self.bufs[i].is_cuda:
                self.works[i] = self.dist.all_gather_into_tensor(self.flat[i], self.bufs[i], group=self.group, async_op=True)
            else:  # gloo (CPU tests)
                self.works[i] = self.dist.all_gather(self.parts[i], self.bufs[i], group=self.group, async_op=True)

    def wait(self, i):
        w = self.works[i]
        if w is None:
            return self.bufs[i]
        w.wait()  # on a GPU: the current stream waits for the collective, the host does not block
        self.works[i] = None
        if self.mode == "ordered":
            ordered_sum(self.parts[i], self.bufs[i], self.cx)
        return self.bufs[i]

    def wait_all(self):
        for i in range(len(self.bufs)):
            self.wait(i)
