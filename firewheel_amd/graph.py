"""Host-side mirror of the reference interface for the accelerated path, over the C ABI.

Names, argument meaning and error behaviour follow the reference (BillyDM/firewheel @ 2024-10-16):
  * `FirewheelGpuCtx`   ~ FirewheelGraphCtx + FirewheelProcessor (graph/context.rs:29-254, graph/processor.rs:18-248)
  * graph methods       ~ AudioGraph (graph/graph.rs:198-580): add_node, remove_node, connect, disconnect, ...
  * node classes        ~ basic_nodes (nodes/*.rs): constructors take the same arguments; the control-half
                          setters (set_percent_volume, play, ...) become messages with an `at_block` tag.
All arithmetic happens in libfwgpu's HIP kernels; this file only marshals arguments.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FwgpuError


class SampleFormat:  # core/sample_resource.rs:28-335
    INTERLEAVED_I16, INTERLEAVED_U16, INTERLEAVED_F32, PLANAR_I16, PLANAR_U16, PLANAR_F32 = range(6)
    DTYPE = {0: np.int16, 1: np.uint16, 2: np.float32, 3: np.int16, 4: np.uint16, 5: np.float32}


_ADD_EDGE = {-1: "SrcNodeNotFound", -2: "DstNodeNotFound", -3: "InPortOutOfRange", -4: "OutPortOutOfRange",
             -5: "EdgeAlreadyExists", -6: "InputPortAlreadyConnected", -7: "CycleDetected"}
_COMPILE = {-10: "CycleDetected", -11: "ManyToOneError", -12: "NodeActivationFailed"}


class AddEdgeError(Exception):  # graph/graph/error.rs
    def __init__(self, code):
        super().__init__(_ADD_EDGE.get(code, str(code)))
        self.code = code
        self.name = _ADD_EDGE.get(code, str(code))


class CompileGraphError(Exception):  # graph/graph/error.rs
    def __init__(self, code, msg=""):
        super().__init__("%s: %s" % (_COMPILE.get(code, str(code)), msg))
        self.code = code
        self.name = _COMPILE.get(code, str(code))


class LoopRange:  # nodes/sampler.rs:16-19
    NONE, FULL, RANGE_SECS = 0, 1, 2

    def __init__(self, mode, start=0.0, end=0.0):
        self.mode, self.start, self.end = mode, start, end

    @staticmethod
    def Full():
        return LoopRange(LoopRange.FULL)

    @staticmethod
    def RangeSecs(start, end):
        return LoopRange(LoopRange.RANGE_SECS, start, end)


# ------------------------------------------------------------------------------------------- nodes
class _Node(object):
    KIND = 0
    cx = None
    id = None

    def params(self):
        return []

    def _bind(self, cx, node_id):
        self.cx, self.id = cx, node_id

    def _set(self, param, value, at_block=0):
        self.cx._check(self.cx.L.fwgpu_node_set_param(self.cx.c, self.id, param, value, at_block))


class DummyAudioNode(_Node):  # nodes/dummy.rs
    KIND = 0


class BeepTestNode(_Node):  # nodes/beep_test.rs:14-33
    KIND = 1

    def __init__(self, freq_hz, gain_db, enabled):
        self.freq_hz, self.gain_db, self._enabled = freq_hz, gain_db, enabled

    def params(self):
        return [self.freq_hz, self.gain_db, 1.0 if self._enabled else 0.0]

    def set_enabled(self, enabled, at_block=0):
        self._enabled = enabled
        self._set(0, 1.0 if enabled else 0.0, at_block)


class VolumeNode(_Node):  # nodes/volume.rs:14-39
    KIND = 2

    def __init__(self, percent_volume):
        self.percent_volume = max(percent_volume, 0.0)

    def params(self):
        return [self.percent_volume]

    def set_percent_volume(self, percent_volume, at_block=0):
        self.percent_volume = max(percent_volume, 0.0)
        self._set(0, percent_volume, at_block)


class SumNode(_Node):  # nodes/sum.rs
    KIND = 3


class SamplerNode(_Node):  # nodes/sampler.rs:46-182
    KIND = 4

    def __init__(self, percent_volume):
        self.percent_volume = max(percent_volume, 0.0)
        self.playing = False

    def params(self):
        return [self.percent_volume]

    def set_percent_volume(self, percent_volume, at_block=0):
        self.percent_volume = max(percent_volume, 0.0)
        self._set(0, percent_volume, at_block)

    def set_sample(self, sample, stop_playback, at_block=0):
        self.cx._check(self.cx.L.fwgpu_sampler_set_sample(self.cx.c, self.id, sample, int(stop_playback), at_block))

    def play(self, at_block=0):  # sampler.rs:82-97: only sends when not already playing
        if not self.playing:
            self.cx._check(self.cx.L.fwgpu_sampler_play(self.cx.c, self.id, at_block))
            self.playing = True

    def pause(self, at_block=0):
        if self.playing:
            self.cx._check(self.cx.L.fwgpu_sampler_pause(self.cx.c, self.id, at_block))
            self.playing = False

    def stop(self, at_block=0):
        if self.playing:
            self.cx._check(self.cx.L.fwgpu_sampler_stop(self.cx.c, self.id, at_block))
            self.playing = False

    def is_playing(self):  # sampler.rs:162-164: the control half's own flag, not the processor's
        return self.playing

    def set_playhead(self, playhead_secs, at_block=0):
        self.cx._check(self.cx.L.fwgpu_sampler_set_playhead_secs(self.cx.c, self.id, playhead_secs, at_block))

    def set_loop_range(self, loop_range, at_block=0):
        lr = loop_range or LoopRange(LoopRange.NONE)
        self.cx._check(self.cx.L.fwgpu_sampler_set_loop_range(self.cx.c, self.id, lr.mode, lr.start, lr.end, at_block))


class HardClipNode(_Node):  # nodes/hard_clip.rs:7-13
    KIND = 5

    def __init__(self, threshold_db):
        self.threshold_db = threshold_db

    def params(self):
        return [self.threshold_db]


class MonoToStereoNode(_Node):  # nodes/mono_to_stereo.rs
    KIND = 6


class StereoToMonoNode(_Node):  # nodes/stereo_to_mono.rs
    KIND = 7


class StereoPanNode(_Node):  # SPEC node: constant-power pan, pan in [-1, 1]
    KIND = 8

    def __init__(self, pan):
        self.pan = pan

    def params(self):
        return [self.pan]

    def set_pan(self, pan, at_block=0):
        self.pan = pan
        self._set(0, pan, at_block)


class StereoWidthNode(_Node):  # SPEC node: mid/side width, w >= 0 (1 = unchanged, 0 = mono), smoothed
    KIND = 9

    def __init__(self, width):
        self.width = width

    def params(self):
        return [self.width]

    def set_width(self, width, at_block=0):
        self.width = width
        self._set(0, width, at_block)


class BiquadNode(_Node):  # SPEC node: RBJ biquad, Direct Form I in f32; coefficients computed on the control side (f64)
    KIND = 10
    LOWPASS, HIGHPASS, BANDPASS = 0, 1, 2

    def __init__(self, filter_type, cutoff_hz, q=0.70710678):
        self.filter_type, self.cutoff_hz, self.q = filter_type, cutoff_hz, q

    def params(self):
        return [float(self.filter_type), self.cutoff_hz, self.q]

    def set_cutoff_hz(self, cutoff_hz, at_block=0):
        self.cutoff_hz = cutoff_hz
        self._set(1, cutoff_hz, at_block)

    def set_q(self, q, at_block=0):
        self.q = q
        self._set(2, q, at_block)


class DelayNode(_Node):  # SPEC node: integer-sample delay line (fixed length) with feedback and dry/wet mix
    KIND = 11

    def __init__(self, delay_secs, feedback=0.0, mix=0.5):
        self.delay_secs, self.feedback, self.mix = delay_secs, feedback, mix

    def params(self):
        return [self.delay_secs, self.feedback, self.mix]

    def set_feedback(self, feedback, at_block=0):
        self.feedback = feedback
        self._set(1, feedback, at_block)

    def set_mix(self, mix, at_block=0):
        self.mix = mix
        self._set(2, mix, at_block)


class FirReverbNode(_Node):  # SPEC node: convolution with an impulse-response sample (f32 MFMA Toeplitz GEMM)
    KIND = 12

    def __init__(self, impulse_response_sample):
        self.ir = impulse_response_sample

    def params(self):
        return [float(self.ir)]


class ResamplerNode(_Node):  # SPEC node: resampling source — polyphase windowed sinc, 32.32 fixed-point position
    KIND = 13

    def __init__(self, sample, ratio=1.0, loop=False, playing=True):
        self.sample, self.ratio, self.loop, self.playing = sample, ratio, loop, playing

    def params(self):
        return [float(self.sample), self.ratio, 1.0 if self.loop else 0.0, 1.0 if self.playing else 0.0]

    def set_ratio(self, ratio, at_block=0):
        """source frames consumed per output frame (= source rate / stream rate x playback speed)"""
        self.ratio = ratio
        self._set(1, ratio, at_block)

    def set_playing(self, playing, at_block=0):
        self.playing = playing
        self._set(3, 1.0 if playing else 0.0, at_block)

    def seek_frames(self, frame, at_block=0):
        self._set(4, float(frame), at_block)


class SpatialNode(_Node):  # SPEC node: 3D spatialiser — distance gain, equal-power pan, per-ear delay
    KIND = 14

    def __init__(self, x, y, z):
        self.pos = [x, y, z]

    def params(self):
        return list(self.pos)

    def set_position(self, x, y, z, at_block=0):
        """source position relative to the listener: +x right, +y up, -z forward"""
        self.pos = [x, y, z]
        for i, v in enumerate(self.pos):
            self._set(i, v, at_block)


class HostNode(_Node):
    """any other `dyn AudioNodeProcessor` (graph/processor.rs:243) inside a device-resident graph: `process` runs on the host,
    on the audio thread, once per block — process(frames, inputs, outputs, in_silence_mask, stream_time_secs, stream_status)
    -> out_silence_mask, inputs / outputs being float32 numpy views (FWGPU_HOST_NODE, include/fwgpu.h)"""
    KIND = 15

    def __init__(self, process):
        self.process = process
        self._cb = _lib.host_process_adapter(process)

    def _bind(self, cx, node_id):
        super()._bind(cx, node_id)
        cx._check(cx.L.fwgpu_host_node_set_process(cx.c, node_id, self._cb, None))


class _RawNode(_Node):
    def __init__(self, kind, params):
        self.KIND = kind
        self._p = list(params)

    def params(self):
        return self._p


# ------------------------------------------------------------------------------------------- context
def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class FirewheelGpuCtx(object):
    """Device-resident graph context.  `stream` is an optional hipStream_t (int) to run on."""

    def __init__(self, sample_rate=48000, max_block_frames=256, num_graph_inputs=0, num_graph_outputs=2, device=0,
                 stream=None):
        self.L = _lib.load_library()
        self.sample_rate = sample_rate
        self.max_block_frames = max_block_frames
        self.c = self.L.fwgpu_ctx_create(device, sample_rate, max_block_frames, num_graph_inputs, num_graph_outputs,
                                         C.c_void_p(stream) if stream else None)
        if not self.c:
            raise FwgpuError(-30, self.L.fwgpu_create_error().decode())
        self._nodes = {}
        self._limbo = []  # removed HostNodes (their ctypes thunks) until a plan without them is the active one
        self._keep = []

    def close(self):
        if getattr(self, "c", None):
            self.L.fwgpu_ctx_destroy(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise FwgpuError(rc, self.L.fwgpu_last_error(self.c).decode())
        return rc

    # ---- AudioGraph (graph/graph.rs)
    @property
    def graph(self):
        return self

    graph_mut = graph

    def graph_in_node(self):
        return self.L.fwgpu_graph_in_node(self.c)

    def graph_out_node(self):
        return self.L.fwgpu_graph_out_node(self.c)

    def add_node(self, num_inputs, num_outputs, node):
        p = np.asarray(node.params(), dtype=np.float32)
        nid = self.L.fwgpu_add_node(self.c, node.KIND, num_inputs, num_outputs, _fptr(p), len(p))
        self._check(nid)
        node._bind(self, nid)
        self._nodes[nid] = node
        return nid

    def node(self, node_id):
        return self._nodes.get(node_id)

    node_mut = node

    def remove_node(self, node_id):
        self._check(self.L.fwgpu_remove_node(self.c, node_id))
        n = self._nodes.pop(node_id, None)
        if isinstance(n, HostNode):
            # the running plan (and a pending one) still call the node's ctypes thunk until a plan WITHOUT it is the active one
            # (fwgpu_abi.cpp: "the running plan may hold it until the next plan is adopted"): a thunk freed here is a segfault in
            # the next process call.  It waits in limbo for an update that has been adopted (_reap_limbo).
            self._limbo.append([n, False])

    def _reap_limbo(self, updated=False):
        """drop removed host nodes once a plan built AFTER their removal is the active one: an update has returned since
        (entry[1]) and none is waiting for adoption (fwgpu_plan_pending)"""
        if updated:
            for ent in self._limbo:
                ent[1] = True
        if self._limbo and not self.L.fwgpu_plan_pending(self.c):
            self._limbo = [ent for ent in self._limbo if not ent[1]]

    def connect(self, src_node, src_port, dst_node, dst_port, check_for_cycles=False):
        r = self.L.fwgpu_connect(self.c, src_node, src_port, dst_node, dst_port, 1 if check_for_cycles else 0)
        if r < 0:
            raise AddEdgeError(r)
        return r

    def disconnect(self, src_node, src_port, dst_node, dst_port):
        return bool(self.L.fwgpu_disconnect(self.c, src_node, src_port, dst_node, dst_port))

    def disconnect_by_edge_id(self, edge_id):
        return bool(self.L.fwgpu_disconnect_edge(self.c, edge_id))

    def cycle_detected(self):
        return bool(self.L.fwgpu_cycle_detected(self.c))

    # ---- FirewheelGraphCtx::update (graph/context.rs:93-137)
    def update(self):
        r = self.L.fwgpu_update(self.c)
        if r in _COMPILE:
            raise CompileGraphError(r, self.L.fwgpu_last_error(self.c).decode())
        self._check(r)
        self._reap_limbo(updated=True)

    def plan_pending(self):
        """True while a plan built by update() / schedule_upload() waits for a process call to adopt it (include/fwgpu.h)"""
        return bool(self._check(self.L.fwgpu_plan_pending(self.c)))

    def schedule_upload(self, sched, num_buffers):
        """sched: list of dicts {"id", "in": [(buffer_index, should_clear)], "out": [buffer_index]} in schedule order
        (the reference's CompiledSchedule, schedule.rs:12-30)."""
        arr = (_lib.SchedNode * len(sched))()
        keep = []
        for i, s in enumerate(sched):
            ib = (C.c_uint32 * max(len(s["in"]), 1))(*[b for b, _ in s["in"]])
            ic = (C.c_uint8 * max(len(s["in"]), 1))(*[1 if c else 0 for _, c in s["in"]])
            ob = (C.c_uint32 * max(len(s["out"]), 1))(*s["out"])
            keep += [ib, ic, ob]
            arr[i].node = s["id"]
            arr[i].num_inputs = len(s["in"])
            arr[i].num_outputs = len(s["out"])
            arr[i].in_buffer_index = ib
            arr[i].in_should_clear = ic
            arr[i].out_buffer_index = ob
        r = self.L.fwgpu_schedule_upload(self.c, arr, len(sched), num_buffers)
        if r in _COMPILE:
            raise CompileGraphError(r, self.L.fwgpu_last_error(self.c).decode())
        self._check(r)
        self._reap_limbo(updated=True)

    # ---- plan introspection
    def plan_kind(self):
        return self.L.fwgpu_plan_kind(self.c)

    def plan_fused_voices(self):
        """voices of the installed plan that the fused kernels render (plan 3: the banks')"""
        return self.L.fwgpu_plan_fused_voices(self.c)

    def plan_num_levels(self):
        return self.L.fwgpu_plan_num_levels(self.c)

    def plan_node_level(self, node_id):
        return self.L.fwgpu_plan_node_level(self.c, node_id)

    def plan_node_inputs_clear(self, node_id):
        buf = (C.c_int * 64)()
        n = self._check(self.L.fwgpu_plan_node_inputs_clear(self.c, node_id, buf, 64))
        return [bool(buf[i]) for i in range(n)]

    def rt_resident_stats(self):
        """(resident realtime kernels launched, callbacks served through the doorbell) — include/fwgpu.h"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.fwgpu_rt_resident_stats(self.c, C.byref(a), C.byref(b)))
        return a.value, b.value

    def rt_path_stats(self):
        """one-block launch batches by path: (resident kernel, one k_rt_block launch, the fused plans' launch sequence, the level
        executor) — include/fwgpu.h fwgpu_rt_path_stats"""
        a = (C.c_uint64 * 4)()
        self._check(self.L.fwgpu_rt_path_stats(self.c, a))
        return tuple(int(x) for x in a)

    def hip_stream(self):
        """the hipStream_t (int) the ctx's process calls launch on — include/fwgpu.h fwgpu_hip_stream"""
        return int(self.L.fwgpu_hip_stream(self.c) or 0)

    def lazy_stats(self):
        """(launch batches rendered without a control kernel, with one) — include/fwgpu.h fwgpu_lazy_stats"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.fwgpu_lazy_stats(self.c, C.byref(a), C.byref(b)))
        return a.value, b.value

    def plan_handover_stats(self):
        """(plans adopted so far, those adopted by a process call, the longest one of those held up its call in ns)"""
        a, b, m = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.fwgpu_plan_handover_stats(self.c, C.byref(a), C.byref(b), C.byref(m)))
        return a.value, b.value, m.value

    def update_phase(self):
        """which part of fwgpu_update the control thread is in right now (0 none, 1 graph compile, 21..28 table sections, 3 the
        build's device work); any thread may ask"""
        return int(self.L.fwgpu_update_phase(self.c))

    def plan_chain_stats(self):
        """(steady, general): k_chain workgroup launches that ran the steady-call loop / the general loop"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.fwgpu_plan_chain_stats(self.c, C.byref(a), C.byref(b)))
        return a.value, b.value

    def plan_host_nodes(self):
        """(host nodes in the installed plan, callbacks run since it was installed)"""
        n = C.c_uint64()
        return self._check(self.L.fwgpu_plan_host_nodes(self.c, C.byref(n))), n.value

    def set_max_batch(self, k):
        self._check(self.L.fwgpu_set_max_batch(self.c, k))

    def set_force_generic(self, on):
        self._check(self.L.fwgpu_set_force_generic(self.c, 1 if on else 0))

    # ---- SampleResource (core/sample_resource.rs)
    def new_sample(self, fmt, channels, data):
        a = np.ascontiguousarray(np.asarray(data, dtype=SampleFormat.DTYPE[fmt]))
        frames = a.size // channels
        return self._check(self.L.fwgpu_sample_create(self.c, fmt, channels, frames, a.ctypes.data_as(C.c_void_p)))

    def new_sample_device(self, fmt, channels, frames, device_ptr):
        return self._check(self.L.fwgpu_sample_create_device(self.c, fmt, channels, frames, C.c_void_p(device_ptr)))

    def destroy_sample(self, sample):
        """Release a sample's HBM (the last Arc<dyn SampleResource> dropped).  Refused while a FIR / resampler node
        names it; a sampler that still holds the id sees an empty sample from the next call on."""
        self._check(self.L.fwgpu_sample_destroy(self.c, sample))

    def poll_returned_samples(self, cap=64):
        """ProcessorToNodeMsg::ReturnSample (nodes/sampler.rs:339-343): [(node id, sample id)] a completed process call
        swapped out since the last poll — what SamplerNode::update() drains (sampler.rs:222-231)."""
        nodes, samples = (C.c_int64 * cap)(), (C.c_int * cap)()
        n = self._check(self.L.fwgpu_poll_returned_samples(self.c, nodes, samples, cap))
        return [(nodes[i], samples[i]) for i in range(n)]

    def sample_retired(self, sample):
        """True once no sampler holds the sample, no queued message names it and the device is done reading it."""
        return bool(self._check(self.L.fwgpu_sample_retired(self.c, sample)))

    def ext_pool_floats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.fwgpu_ext_pool_floats(self.c, C.byref(a), C.byref(b)))
        return a.value, b.value

    def proc_info(self):
        """(stream_time_secs, stream_status, output_underflows, input_overflows) — ProcInfo of the last call (core/node.rs:111-132)"""
        t, st, u, o = C.c_double(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        self._check(self.L.fwgpu_proc_info(self.c, C.byref(t), C.byref(st), C.byref(u), C.byref(o)))
        return t.value, st.value, u.value, o.value

    def open_stream(self, num_in_channels=0, num_out_channels=2):
        return HeadlessStream(self, num_in_channels, num_out_channels)

    # ---- FirewheelProcessor::process_interleaved (graph/processor.rs:61-165)
    def process_interleaved(self, input, num_in_channels, num_out_channels, frames, stream_time_secs=0.0,
                            stream_status=0):
        out = np.full(frames * num_out_channels, np.nan, dtype=np.float32)
        if input is None:
            input = np.zeros(max(frames * num_in_channels, 1), dtype=np.float32)
        inp = np.ascontiguousarray(input, dtype=np.float32)
        self._check(self.L.fwgpu_process_interleaved(self.c, _fptr(inp), _fptr(out), num_in_channels, num_out_channels,
                                                     frames, stream_time_secs, stream_status))
        if self._limbo:  # (this call may have adopted the plan that no longer names a removed host node)
            self._reap_limbo()
        return out

    def process_interleaved_begin(self, input, num_in_channels, num_out_channels, frames, stream_time_secs=0.0, stream_status=0):
        """fwgpu_process_interleaved_begin: the call up to its last launch; returns the ticket `process_interleaved_end` takes"""
        if input is None:
            input = np.zeros(max(frames * num_in_channels, 1), dtype=np.float32)
        inp = np.ascontiguousarray(input, dtype=np.float32)
        t = self.L.fwgpu_process_interleaved_begin(self.c, _fptr(inp), num_in_channels, num_out_channels, frames, stream_time_secs, stream_status)
        self._check(t)
        return int(t), frames * num_out_channels

    def process_interleaved_end(self, ticket, out=None):
        """... and its other half: waits for the ticket's copy back, returns the frames (or fills `out`)"""
        t, n = ticket
        if out is None:
            out = np.full(n, np.nan, dtype=np.float32)
        self._check(self.L.fwgpu_process_interleaved_end(self.c, t, _fptr(out)))
        if self._limbo:
            self._reap_limbo()
        return out

    def process_interleaved_cancel(self, ticket):
        """abandon every ticket in flight up to and including this one (no frames copied)"""
        self._check(self.L.fwgpu_process_interleaved_cancel(self.c, ticket[0]))

    def process_blocks_device(self, num_blocks, device_out_ptr, num_out_channels=2):
        self._check(self.L.fwgpu_process_blocks_device(self.c, num_blocks, C.c_void_p(device_out_ptr), num_out_channels))

    def process_blocks_device_flags(self, num_blocks, device_out_ptr, num_out_channels, device_silence_ptr):
        """... also writing, per (block, channel), whether that graph-output channel was flagged silent (u8, device memory)"""
        self._check(self.L.fwgpu_process_blocks_device_flags(self.c, num_blocks, C.c_void_p(device_out_ptr), num_out_channels,
                                                             C.c_void_p(device_silence_ptr) if device_silence_ptr else None))

    def process_blocks_device_io(self, num_blocks, device_in_ptr, num_in_channels, device_out_ptr, num_out_channels=2, device_silence_ptr=None):
        """... for graphs with stream inputs: interleaved input frames in device memory (fwgpu_process_blocks_device_io)"""
        self._check(self.L.fwgpu_process_blocks_device_io(self.c, num_blocks, C.c_void_p(device_in_ptr) if device_in_ptr else None, num_in_channels,
                                                          C.c_void_p(device_out_ptr), num_out_channels,
                                                          C.c_void_p(device_silence_ptr) if device_silence_ptr else None))

    def bus_sum_ordered(self, part_ptrs, out_ptr, n_floats, silence_ptrs=None, out_silence_ptr=None, frames_per_block=0, n_channels=2):
        """the top-level R-port SumNode over the shards' partial buses (device pointers), rank order, on the ctx stream;
        silence_ptrs: per part, its [blocks][channels] silence flags as process_blocks_device_flags wrote them (sum.rs:122-124)"""
        arr = (C.c_void_p * len(part_ptrs))(*[C.c_void_p(p) for p in part_ptrs])
        if silence_ptrs is None:
            self._check(self.L.fwgpu_bus_sum_ordered(self.c, arr, len(part_ptrs), C.c_void_p(out_ptr), n_floats))
            return
        sil = (C.c_void_p * len(part_ptrs))(*[C.c_void_p(p) if p else None for p in silence_ptrs])
        self._check(self.L.fwgpu_bus_sum_ordered_flags(self.c, arr, sil, len(part_ptrs), C.c_void_p(out_ptr),
                                                       C.c_void_p(out_silence_ptr) if out_silence_ptr else None, n_floats,
                                                       frames_per_block, n_channels))

    def open_bus_exchange(self, rank, world, max_floats, max_silence_bytes=0):
        return BusExchange(self, rank, world, max_floats, max_silence_bytes)

    def synchronize(self):
        self._check(self.L.fwgpu_synchronize(self.c))

    # ---- AudioNodeProcessor::process for one node on host buffers (core/node.rs:37-53)
    def node_process(self, node_id, frames, inputs, outputs, in_silence_mask=0, out_silence_mask=0):
        ins = [np.ascontiguousarray(x, dtype=np.float32) for x in inputs]
        for o in outputs:
            assert o.dtype == np.float32 and o.flags["C_CONTIGUOUS"]
        it = (C.POINTER(C.c_float) * max(len(ins), 1))(*[_fptr(a) for a in ins])
        ot = (C.POINTER(C.c_float) * max(len(outputs), 1))(*[_fptr(a) for a in outputs])
        om = C.c_uint64(out_silence_mask)
        self._check(self.L.fwgpu_node_process(self.c, node_id, frames, it, len(ins), ot, len(outputs), in_silence_mask,
                                              C.byref(om), 0.0, 0))
        return om.value

    # ---- measurement hooks
    def timing_enable(self, on=True):
        self._check(self.L.fwgpu_timing_enable(self.c, 1 if on else 0))

    def timing_read(self, which=0):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.L.fwgpu_timing_read(self.c, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_reset(self):
        self._check(self.L.fwgpu_timing_reset(self.c))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int(), C.c_uint64()
        self._check(self.L.fwgpu_device_info(self.c, name, 256, C.byref(cus), C.byref(hbm)))
        return name.value.decode(), cus.value, hbm.value


class BusExchange(object):
    """The multi-GPU mix bus behind the C ABI (fwgpu_bus_exchange_*, SURVEY §8e path 2): every rank stores its partial bus
    into its slot on every rank (peer-mapped over xGMI, or by pointer inside one process), then adds the R slots it holds in
    rank order — the top-level R-port SumNode (nodes/sum.rs:41-136), bit-identical to the single-process graph."""

    def __init__(self, cx, rank, world, max_floats, max_silence_bytes=0):
        self.cx, self.rank, self.world = cx, rank, world
        self.x = cx.L.fwgpu_bus_exchange_open(cx.c, rank, world, max_floats, max_silence_bytes)
        if not self.x:
            raise FwgpuError(-30, cx.L.fwgpu_last_error(cx.c).decode())

    def export(self):
        """this rank's handle (bytes) — carry it to the peers through any side channel"""
        buf = C.create_string_buffer(_lib.EXCHANGE_HANDLE_BYTES)
        self.cx._check(self.cx.L.fwgpu_bus_exchange_export(self.x, buf))
        return buf.raw

    def connect(self, peer_rank, handle):
        buf = C.create_string_buffer(bytes(handle), _lib.EXCHANGE_HANDLE_BYTES)
        self.cx._check(self.cx.L.fwgpu_bus_exchange_connect(self.x, peer_rank, buf))

    def connect_all(self, handles):
        for r, h in enumerate(handles):
            if r != self.rank:
                self.connect(r, h)

    def set_timeout_ms(self, ms):
        self.cx._check(self.cx.L.fwgpu_bus_exchange_set_timeout_ms(self.x, ms))

    def push(self, part_ptr, n_floats, silence_ptr=None, n_blocks=0, n_channels=2):
        self.cx._check(self.cx.L.fwgpu_bus_exchange_push(self.x, C.c_void_p(part_ptr), C.c_void_p(silence_ptr) if silence_ptr else None,
                                                         n_floats, n_blocks, n_channels))

    def reduce(self, out_ptr, n_floats, out_silence_ptr=None, n_blocks=0, frames_per_block=0, n_channels=2, have_silence=False):
        self.cx._check(self.cx.L.fwgpu_bus_exchange_reduce(self.x, C.c_void_p(out_ptr), C.c_void_p(out_silence_ptr) if out_silence_ptr else None,
                                                           n_floats, n_blocks, frames_per_block, n_channels, 1 if have_silence else 0))

    def step(self, part_ptr, out_ptr, n_floats, silence_ptr=None, out_silence_ptr=None, n_blocks=0, frames_per_block=0, n_channels=2):
        self.cx._check(self.cx.L.fwgpu_bus_exchange_step(self.x, C.c_void_p(part_ptr), C.c_void_p(silence_ptr) if silence_ptr else None,
                                                         C.c_void_p(out_ptr), C.c_void_p(out_silence_ptr) if out_silence_ptr else None,
                                                         n_floats, n_blocks, frames_per_block, n_channels))

    def status(self):
        """(reduces issued, 0 or the first step a peer did not arrive for) — waits for the ctx stream; raises on a failed step"""
        a, b = C.c_uint64(), C.c_uint64()
        rc = self.cx.L.fwgpu_bus_exchange_status(self.x, C.byref(a), C.byref(b))
        if rc < 0:
            raise FwgpuError(rc, "%s (step %d)" % (self.cx.L.fwgpu_last_error(self.cx.c).decode(), b.value))
        return a.value, b.value

    def wait_stats(self, reset=False):
        """[us] the longest a reduce of this rank has waited for each rank's arrival so far (waits for the ctx stream)"""
        buf = (C.c_uint64 * self.world)()
        self.cx._check(self.cx.L.fwgpu_bus_exchange_wait_stats(self.x, buf, self.world, 1 if reset else 0))
        return [int(v) for v in buf]

    def close(self):
        if getattr(self, "x", None) and getattr(self.cx, "c", None):
            self.cx.L.fwgpu_bus_exchange_close(self.x)
        self.x = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HeadlessStream(object):
    """DataCallback (firewheel-cpal/src/lib.rs:362-449) without a device: `callback(frames, instant)` once per device
    period, `instant` = the backend's clock at that callback in seconds (cpal: info.timestamp().callback)."""

    OUTPUT_UNDERFLOW = 2  # StreamStatus (core/node.rs:120-132)
    INPUT_OVERFLOW = 1

    def __init__(self, cx, num_in_channels, num_out_channels):
        self.cx = cx
        self.num_out_channels = num_out_channels
        self.s = cx.L.fwgpu_stream_open(cx.c, num_in_channels, num_out_channels)
        if not self.s:
            raise FwgpuError(-20, "fwgpu_stream_open failed")

    def callback(self, frames, instant_secs):
        """returns (interleaved output, StreamStatus bits handed to process_interleaved)"""
        out = np.full(frames * self.num_out_channels, np.nan, dtype=np.float32)
        st = self.cx._check(self.cx.L.fwgpu_stream_callback(self.s, _fptr(out), frames, instant_secs))
        return out, st

    def run(self, frames, n_callbacks, first_instant_secs=0.0):
        """the backend thread's loop (fwgpu_stream_run): n callbacks back to back inside the library.  Returns (the last
        block, seconds the loop took)"""
        out = np.full(frames * self.num_out_channels, np.nan, dtype=np.float32)
        dt = C.c_double()
        self.cx._check(self.cx.L.fwgpu_stream_run(self.s, _fptr(out), frames, n_callbacks, first_instant_secs, C.byref(dt)))
        return out, dt.value

    def stats(self):
        a, b, t = C.c_uint64(), C.c_uint64(), C.c_double()
        self.cx._check(self.cx.L.fwgpu_stream_stats(self.s, C.byref(a), C.byref(b), C.byref(t)))
        return a.value, b.value, t.value

    def close(self):
        if self.s:
            self.cx.L.fwgpu_stream_close(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
