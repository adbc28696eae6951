"""Shared test harness: one Python `Engine` interface over two C-ABI back ends.

* `OracleBackend`  -> oracle/_build/libfw_oracle.so  (CPU restatement of the reference; the checker)
* `GpuBackend`     -> firewheel_amd/csrc/libfwgpu.so (the product: HIP kernels behind the C ABI)

Both expose the reference's edit/process surface (graph/graph.rs add_node/connect/...,
graph/processor.rs process_interleaved, nodes/sampler.rs messages) so a parity test is the same
script run twice.  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# node kinds (oracle/fw_oracle.hpp NodeKind == include/fwgpu.h fwgpu_node_kind)
DUMMY, BEEP_TEST, VOLUME, SUM, SAMPLER, HARD_CLIP, MONO_TO_STEREO, STEREO_TO_MONO, STEREO_PAN = range(9)
STEREO_WIDTH, BIQUAD, DELAY, FIR, RESAMPLER, SPATIAL = 9, 10, 11, 12, 13, 14
# sample formats
INTERLEAVED_I16, INTERLEAVED_U16, INTERLEAVED_F32, PLANAR_I16, PLANAR_U16, PLANAR_F32 = range(6)
_FMT_DTYPE = {0: np.int16, 1: np.uint16, 2: np.float32, 3: np.int16, 4: np.uint16, 5: np.float32}
# loop range modes (nodes/sampler.rs:16-19 + Option)
LOOP_NONE, LOOP_FULL, LOOP_RANGE_SECS = 0, 1, 2

ADD_EDGE_ERRORS = {
    -1: "SrcNodeNotFound", -2: "DstNodeNotFound", -3: "InPortOutOfRange", -4: "OutPortOutOfRange",
    -5: "EdgeAlreadyExists", -6: "InputPortAlreadyConnected", -7: "CycleDetected",
}
COMPILE_ERRORS = {-10: "CycleDetected", -11: "ManyToOneError", -12: "NodeActivationFailed"}


class AddEdgeError(Exception):
    def __init__(self, code):
        super().__init__(ADD_EDGE_ERRORS.get(code, str(code)))
        self.code = code
        self.name = ADD_EDGE_ERRORS.get(code, str(code))


class CompileGraphError(Exception):
    def __init__(self, code, msg=""):
        super().__init__("%s: %s" % (COMPILE_ERRORS.get(code, str(code)), msg))
        self.code = code
        self.name = COMPILE_ERRORS.get(code, str(code))


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return os.path.join(ROOT, "oracle", "_build", "libfw_oracle.so")


_oracle_lib = None


def oracle_lib():
    global _oracle_lib
    if _oracle_lib is None:
        path = os.environ.get("FWO_LIB") or os.path.join(ROOT, "oracle", "_build", "libfw_oracle.so")  # FWO_LIB: a sanitised build
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        i64, u32, u64, f32, f64, vp, ci = C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double, C.c_void_p, C.c_int
        fp = C.POINTER(C.c_float)
        sig = {
            "fwo_ctx_new": (vp, [u32, u32, u32, u32]),
            "fwo_ctx_free": (None, [vp]),
            "fwo_last_error": (C.c_char_p, [vp]),
            "fwo_graph_in_node": (i64, [vp]),
            "fwo_graph_out_node": (i64, [vp]),
            "fwo_add_node": (i64, [vp, ci, u32, u32, fp, ci]),
            "fwo_remove_node": (ci, [vp, i64]),
            "fwo_connect": (i64, [vp, i64, u32, i64, u32, ci]),
            "fwo_disconnect": (ci, [vp, i64, u32, i64, u32]),
            "fwo_disconnect_edge": (ci, [vp, i64]),
            "fwo_cycle_detected": (ci, [vp]),
            "fwo_update": (ci, [vp]),
            "fwo_sched_len": (ci, [vp]),
            "fwo_sched_num_buffers": (ci, [vp]),
            "fwo_sched_node": (i64, [vp, ci]),
            "fwo_sched_in": (ci, [vp, ci, C.POINTER(ci), C.POINTER(ci), ci]),
            "fwo_sched_out": (ci, [vp, ci, C.POINTER(ci), ci]),
            "fwo_sample_new": (ci, [vp, ci, u32, u64, vp]),
            "fwo_set_param": (ci, [vp, i64, ci, f32]),
            "fwo_sampler_set_sample": (ci, [vp, i64, ci, ci]),
            "fwo_sampler_play": (ci, [vp, i64]),
            "fwo_sampler_pause": (ci, [vp, i64]),
            "fwo_sampler_stop": (ci, [vp, i64]),
            "fwo_sampler_set_playhead_secs": (ci, [vp, i64, f64]),
            "fwo_sampler_set_loop_range": (ci, [vp, i64, ci, f64, f64]),
            "fwo_process_interleaved": (ci, [vp, fp, fp, u32, u32, u64, f64, u32]),
            "fwo_process_interleaved_masks": (ci, [vp, fp, fp, u32, u32, u64, f64, u32, C.POINTER(u64), u32]),
            "fwo_custom_node_set_process": (ci, [vp, i64, vp, vp]),
            "fwo_process_parallel": (f64, [C.POINTER(vp), ci, u32, u64, f64, C.POINTER(u64)]),
            "fwo_node_process": (ci, [vp, i64, u64, C.POINTER(fp), u32, C.POINTER(fp), u32, u64, C.POINTER(u64), f64, u32]),
            "fwo_stream_new": (vp, [vp, u32, u32, u32]),
            "fwo_stream_free": (None, [vp]),
            "fwo_stream_callback": (ci, [vp, fp, u64, f64, C.POINTER(f64)]),
            "fwo_smoother_new": (vp, [f32, u32, u32]),
            "fwo_smoother_free": (None, [vp]),
            "fwo_smoother_set": (None, [vp, f32]),
            "fwo_smoother_reset": (None, [vp, f32]),
            "fwo_smoother_process": (ci, [vp, u32, fp, u32, C.POINTER(ci)]),
            "fwo_smoother_state": (None, [vp, fp, fp, fp, fp, C.POINTER(ci)]),
            "fwo_mask_new_all_silent": (u64, [u64]),
            "fwo_mask_any_silent": (ci, [u64, u64]),
            "fwo_mask_all_silent": (ci, [u64, u64]),
            "fwo_db_to_gain": (f32, [f32]),
            "fwo_db_to_gain_clamped": (f32, [f32]),
            "fwo_gain_to_db_clamped": (f32, [f32]),
            "fwo_percent_volume_to_raw_gain": (f32, [f32]),
            "fwo_pcm_i16_to_f32": (f32, [C.c_int16]),
            "fwo_pcm_u16_to_f32": (f32, [C.c_uint16]),
            "fwo_pan_to_gains": (None, [f32, fp, fp]),
            "fwo_deinterleave": (u64, [C.POINTER(fp), u32, u32, fp, u32, u32, ci]),
            "fwo_interleave": (None, [C.POINTER(fp), u32, u32, fp, u32, u32, ci, u64]),
            "fwo_interleave_stereo": (None, [fp, fp, fp, u32, ci, u64]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _oracle_lib = L
    return _oracle_lib


def oracle_process_parallel(engines, frames_per_call, secs, n_out_ch=2):
    """one native thread per OracleEngine (libfw_oracle's own std::thread loop: no Python between the calls) for `secs`
    seconds -> (process calls completed per engine, wall seconds)"""
    L = oracle_lib()
    n = len(engines)
    arr = (C.c_void_p * n)(*[e.c for e in engines])
    done = (C.c_uint64 * n)()
    dt = L.fwo_process_parallel(arr, n, n_out_ch, frames_per_call, secs, done)
    return [int(x) for x in done], dt


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ptr_array(arrs):
    t = C.POINTER(C.c_float) * max(len(arrs), 1)
    return t(*[_fptr(a) for a in arrs])


class Engine(object):
    """Common surface.  Subclasses bind `self.L`/prefix and fill the small differences."""

    backend = "?"

    def add_node(self, kind, n_in, n_out, params=()):
        raise NotImplementedError

    # ---- helpers mirroring the reference node constructors (nodes/*.rs)
    def volume(self, percent, ch=2):
        return self.add_node(VOLUME, ch, ch, [percent])

    def sampler(self, percent=100.0, n_out=2):
        return self.add_node(SAMPLER, 0, n_out, [percent])

    def sum(self, ports, ch=2):
        return self.add_node(SUM, ports * ch, ch)

    def beep(self, freq=440.0, gain_db=-12.0, enabled=True, n_out=2):
        return self.add_node(BEEP_TEST, 0, n_out, [freq, gain_db, 1.0 if enabled else 0.0])

    def hard_clip(self, threshold_db, ch=2):
        return self.add_node(HARD_CLIP, ch, ch, [threshold_db])

    def pan(self, pan):
        return self.add_node(STEREO_PAN, 2, 2, [pan])

    def width(self, w):
        return self.add_node(STEREO_WIDTH, 2, 2, [w])

    def biquad(self, ftype, cutoff_hz, q=0.70710678, ch=2):
        return self.add_node(BIQUAD, ch, ch, [float(ftype), cutoff_hz, q])

    def delay(self, secs, feedback=0.0, mix=0.5, ch=2):
        return self.add_node(DELAY, ch, ch, [secs, feedback, mix])

    def fir(self, ir_sample, ch=2):
        return self.add_node(FIR, ch, ch, [float(ir_sample)])

    def resampler(self, sample, ratio=1.0, loop=False, playing=True, n_out=2):
        return self.add_node(RESAMPLER, 0, n_out, [float(sample), ratio, 1.0 if loop else 0.0, 1.0 if playing else 0.0])

    def spatial(self, x, y, z, n_in=1):
        return self.add_node(SPATIAL, n_in, 2, [x, y, z])

    def connect_stereo(self, src, dst, dst_port0=0, src_port0=0):
        self.connect(src, src_port0, dst, dst_port0)
        self.connect(src, src_port0 + 1, dst, dst_port0 + 1)


class OracleEngine(Engine):
    backend = "oracle"

    def __init__(self, sample_rate=48000, max_block_frames=256, num_graph_inputs=0, num_graph_outputs=2):
        self.L = oracle_lib()
        self.sample_rate = sample_rate
        self.max_block_frames = max_block_frames
        self.c = self.L.fwo_ctx_new(sample_rate, max_block_frames, num_graph_inputs, num_graph_outputs)
        self._keep = []

    def __del__(self):
        try:
            self.L.fwo_ctx_free(self.c)
        except Exception:
            pass

    @property
    def graph_in_node(self):
        return self.L.fwo_graph_in_node(self.c)

    @property
    def graph_out_node(self):
        return self.L.fwo_graph_out_node(self.c)

    def add_node(self, kind, n_in, n_out, params=()):
        p = np.asarray(list(params), dtype=np.float32)
        return self.L.fwo_add_node(self.c, kind, n_in, n_out, _fptr(p), len(p))

    def host_node(self, n_in, n_out, process):
        """a custom node: `process(frames, inputs, outputs, in_mask, stream_time, status) -> out_mask` (KIND_CUSTOM)"""
        import firewheel_amd._lib as flib

        cb = flib.host_process_adapter(process)
        self._keep_cbs = getattr(self, "_keep_cbs", []) + [cb]
        n = self.add_node(15, n_in, n_out)
        assert self.L.fwo_custom_node_set_process(self.c, n, C.cast(cb, C.c_void_p), None) == 0
        return n

    def remove_node(self, node):
        return self.L.fwo_remove_node(self.c, node)

    def connect(self, src, sp, dst, dp, check_for_cycles=False):
        r = self.L.fwo_connect(self.c, src, sp, dst, dp, 1 if check_for_cycles else 0)
        if r < 0:
            raise AddEdgeError(r)
        return r

    def disconnect(self, src, sp, dst, dp):
        return bool(self.L.fwo_disconnect(self.c, src, sp, dst, dp))

    def disconnect_by_edge_id(self, e):
        return bool(self.L.fwo_disconnect_edge(self.c, e))

    def cycle_detected(self):
        return bool(self.L.fwo_cycle_detected(self.c))

    def update(self):
        r = self.L.fwo_update(self.c)
        if r < 0:
            raise CompileGraphError(r, self.L.fwo_last_error(self.c).decode())

    # schedule introspection -> list of dicts like ScheduledNode (schedule.rs:12-20)
    def schedule(self):
        n = self.L.fwo_sched_len(self.c)
        out = []
        buf = (C.c_int * 64)()
        clr = (C.c_int * 64)()
        for i in range(n):
            ni = self.L.fwo_sched_in(self.c, i, buf, clr, 64)
            ins = [(buf[k], bool(clr[k])) for k in range(ni)]
            no = self.L.fwo_sched_out(self.c, i, buf, 64)
            outs = [buf[k] for k in range(no)]
            out.append({"id": self.L.fwo_sched_node(self.c, i), "in": ins, "out": outs})
        return out

    def num_buffers(self):
        return self.L.fwo_sched_num_buffers(self.c)

    def new_sample(self, fmt, channels, data):
        """interleaved: data shape (frames*channels,) or (frames, channels); planar: (channels, frames)."""
        a = np.ascontiguousarray(np.asarray(data, dtype=_FMT_DTYPE[fmt]))
        frames = a.size // channels
        self._keep.append(a)
        return self.L.fwo_sample_new(self.c, fmt, channels, frames, a.ctypes.data_as(C.c_void_p))

    def set_param(self, node, param, value, at_block=0):
        assert at_block == 0
        r = self.L.fwo_set_param(self.c, node, param, value)
        assert r == 0, r

    def sampler_set_sample(self, node, sample, stop_playback=False, at_block=0):
        assert self.L.fwo_sampler_set_sample(self.c, node, sample, int(stop_playback)) == 0

    def sampler_play(self, node, at_block=0):
        assert self.L.fwo_sampler_play(self.c, node) == 0

    def sampler_pause(self, node, at_block=0):
        assert self.L.fwo_sampler_pause(self.c, node) == 0

    def sampler_stop(self, node, at_block=0):
        assert self.L.fwo_sampler_stop(self.c, node) == 0

    def sampler_set_playhead_secs(self, node, secs, at_block=0):
        assert self.L.fwo_sampler_set_playhead_secs(self.c, node, secs) == 0

    def sampler_set_loop_range(self, node, mode, start=0.0, end=0.0, at_block=0):
        assert self.L.fwo_sampler_set_loop_range(self.c, node, mode, start, end) == 0

    def process_interleaved(self, frames, n_out_ch=2, inp=None, n_in_ch=0, t=0.0, status=0):
        out = np.full(frames * n_out_ch, np.nan, dtype=np.float32)
        if inp is None:
            inp = np.zeros(frames * n_in_ch, dtype=np.float32)
        inp = np.ascontiguousarray(inp, dtype=np.float32)
        r = self.L.fwo_process_interleaved(self.c, _fptr(inp), _fptr(out), n_in_ch, n_out_ch, frames, t, status)
        assert r == 0
        return out

    def process_blocks(self, k, n_out_ch=2):
        """K consecutive max_block_frames blocks (what one K-block device launch computes)."""
        return self.process_interleaved(k * self.max_block_frames, n_out_ch)

    def process_blocks_flags(self, k, n_out_ch=2):
        """... and, per (block, channel), whether read_graph_outputs saw that channel flagged silent (schedule.rs:255-287):
        (interleaved output, uint8 [k][n_out_ch])"""
        frames = k * self.max_block_frames
        out = np.full(frames * n_out_ch, np.nan, dtype=np.float32)
        inp = np.zeros(1, dtype=np.float32)
        masks = (C.c_uint64 * k)()
        r = self.L.fwo_process_interleaved_masks(self.c, _fptr(inp), _fptr(out), 0, n_out_ch, frames, 0.0, 0, masks, k)
        assert r == k, r
        fl = np.array([[(masks[b] >> c) & 1 for c in range(n_out_ch)] for b in range(k)], dtype=np.uint8)
        return out, fl

    def node_process(self, node, frames, inputs, n_out, in_mask=0, out_mask=0, out_init=None):
        """B1: one node's process() on caller buffers.  Returns (outputs[n_out, frames], out_mask)."""
        ins = [np.ascontiguousarray(x, dtype=np.float32) for x in inputs]
        outs = [np.full(frames, np.nan, dtype=np.float32) if out_init is None else np.array(out_init[c], dtype=np.float32)
                for c in range(n_out)]
        om = C.c_uint64(out_mask)
        r = self.L.fwo_node_process(self.c, node, frames, _ptr_array(ins), len(ins), _ptr_array(outs), n_out,
                                    in_mask, C.byref(om), 0.0, 0)
        assert r == 0
        return np.stack(outs) if outs else np.zeros((0, frames), np.float32), om.value


_planner_lib = None


def planner_lib():
    """The product's host planner (fwgpu_graph.cpp, no HIP in it) compiled with g++ behind tests/planner_harness."""
    global _planner_lib
    if _planner_lib is None:
        d = os.path.join(ROOT, "tests", "planner_harness")
        so = os.path.join(d, "_planner.so")
        srcs = [os.path.join(d, "planner_harness.cpp"), os.path.join(ROOT, "firewheel_amd", "csrc", "fwgpu_graph.cpp")]
        deps = srcs + [os.path.join(ROOT, "firewheel_amd", "csrc", h) for h in ("fwgpu_graph.h", "fwgpu_types.h")]
        if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in deps):
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wall", "-o", so] + srcs)
        L = C.CDLL(so)
        i64, u32, vp, ci = C.c_int64, C.c_uint32, C.c_void_p, C.c_int
        ip = C.POINTER(ci)
        sig = {
            "fwp_new": (vp, [u32, u32]), "fwp_free": (None, [vp]), "fwp_last_error": (C.c_char_p, [vp]),
            "fwp_graph_in_node": (i64, [vp]), "fwp_graph_out_node": (i64, [vp]),
            "fwp_add_node": (i64, [vp, ci, u32, u32]), "fwp_remove_node": (ci, [vp, i64]),
            "fwp_connect": (i64, [vp, i64, u32, i64, u32, ci]), "fwp_disconnect": (ci, [vp, i64, u32, i64, u32]),
            "fwp_disconnect_edge": (ci, [vp, i64]), "fwp_cycle_detected": (ci, [vp]), "fwp_update": (ci, [vp]),
            "fwp_set_canonical_order": (None, [vp, ci]),
            "fwp_sched_len": (ci, [vp]), "fwp_sched_num_buffers": (ci, [vp]), "fwp_sched_num_levels": (ci, [vp]),
            "fwp_sched_node": (i64, [vp, ci]), "fwp_sched_level": (ci, [vp, ci]),
            "fwp_sched_in": (ci, [vp, ci, ip, ip, ci]), "fwp_sched_out": (ci, [vp, ci, ip, ci]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _planner_lib = L
    return _planner_lib


class PlannerEngine(Engine):
    """Graph-editing + compile surface of OracleEngine on the PRODUCT's host planner (CPU, no device)."""

    backend = "planner"

    def __init__(self, sample_rate=48000, max_block_frames=256, num_graph_inputs=0, num_graph_outputs=2):
        self.L = planner_lib()
        self.c = self.L.fwp_new(num_graph_inputs, num_graph_outputs)

    def __del__(self):
        try:
            self.L.fwp_free(self.c)
        except Exception:
            pass

    @property
    def graph_in_node(self):
        return self.L.fwp_graph_in_node(self.c)

    @property
    def graph_out_node(self):
        return self.L.fwp_graph_out_node(self.c)

    def add_node(self, kind, n_in, n_out, params=()):
        return self.L.fwp_add_node(self.c, kind, n_in, n_out)

    def remove_node(self, node):
        return self.L.fwp_remove_node(self.c, node)

    def connect(self, src, sp, dst, dp, check_for_cycles=False):
        r = self.L.fwp_connect(self.c, src, sp, dst, dp, 1 if check_for_cycles else 0)
        if r < 0:
            raise AddEdgeError(r)
        return r

    def disconnect(self, src, sp, dst, dp):
        return self.L.fwp_disconnect(self.c, src, sp, dst, dp)

    def disconnect_by_edge_id(self, e):
        return self.L.fwp_disconnect_edge(self.c, e)

    def set_canonical_order(self, on):
        """build_plan's order of the plan's tables: True (the default) = level by level, by where a node is connected; False = the
        reference's Kahn order (compiler.rs:232-300)"""
        self.L.fwp_set_canonical_order(self.c, 1 if on else 0)

    def cycle_detected(self):
        return bool(self.L.fwp_cycle_detected(self.c))

    def update(self):
        r = self.L.fwp_update(self.c)
        if r < 0:
            raise CompileGraphError(r, self.L.fwp_last_error(self.c).decode())

    def schedule(self):
        out = []
        buf = (C.c_int * 64)()
        clr = (C.c_int * 64)()
        for i in range(self.L.fwp_sched_len(self.c)):
            ni = self.L.fwp_sched_in(self.c, i, buf, clr, 64)
            ins = [(buf[k], bool(clr[k])) for k in range(ni)]
            no = self.L.fwp_sched_out(self.c, i, buf, 64)
            out.append({"id": self.L.fwp_sched_node(self.c, i), "in": ins, "out": [buf[k] for k in range(no)],
                        "level": self.L.fwp_sched_level(self.c, i)})
        return out

    def num_buffers(self):
        return self.L.fwp_sched_num_buffers(self.c)

    def num_levels(self):
        return self.L.fwp_sched_num_levels(self.c)


def xorshift_uniform(seed, n):
    """Deterministic uniform [-1, 1) f32 stream, LANGUAGE-NEUTRAL (round 4: the scenarios are exported as JSON for a Rust test on the
    real firewheel-graph, tests/golden/scenarios/ — numpy's PCG64 could not be restated there): element i of stream `seed` is
        x = (seed + (i + 1) * 0x9E3779B9) mod 2^32          (counter-based: no state, any element on its own)
        x ^= x >> 16;  x *= 0x85EBCA6B;  x ^= x >> 13;  x *= 0xC2B2AE35;  x ^= x >> 16     (murmur3's 32-bit finaliser, mod 2^32)
        value = f32(x >> 8) * 2^-23 - 1                      (exact in f32: 24 bits, a multiple of 2^-23 in [-1, 1))
    (The name is historical: SURVEY 8d's "seeded xorshift".)"""
    i = np.arange(1, n + 1, dtype=np.uint64)
    x = (np.uint64(int(seed) & 0xFFFFFFFF) + i * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    out = ((x >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)).astype(np.float32)
    _GEN_LOG.append((int(seed) & 0xFFFFFFFF, int(n), out))
    if len(_GEN_LOG) > 4096:
        del _GEN_LOG[:2048]
    return out


# the streams handed out lately: (seed, n, array) — tests/golden/make_scenarios_json.py finds the recipe of a sample's data among them
_GEN_LOG = []


class _DeviceResult(object):
    """interleaved output of an asynchronous process call; converts to numpy (after a sync) on first use"""

    def __init__(self, cx, buf):
        self.cx, self.buf = cx, buf

    def __array__(self, dtype=None, copy=None):
        self.cx.synchronize()
        a = self.buf.cpu().numpy()
        return a if dtype is None else a.astype(dtype)


class GpuEngine(Engine):
    """Same surface as OracleEngine, through the product's C ABI (firewheel_amd -> libfwgpu.so)."""

    backend = "gpu"

    def __init__(self, sample_rate=48000, max_block_frames=256, num_graph_inputs=0, num_graph_outputs=2,
                 force_generic=False, max_batch=None, stream=None):
        import firewheel_amd as fa

        self.fa = fa
        self.sample_rate = sample_rate
        self.max_block_frames = max_block_frames
        self.cx = fa.FirewheelGpuCtx(sample_rate, max_block_frames, num_graph_inputs, num_graph_outputs, stream=stream)
        if force_generic:
            self.cx.set_force_generic(True)
        if max_batch:
            self.cx.set_max_batch(max_batch)

    @property
    def graph_in_node(self):
        return self.cx.graph_in_node()

    @property
    def graph_out_node(self):
        return self.cx.graph_out_node()

    def add_node(self, kind, n_in, n_out, params=()):
        from firewheel_amd.graph import _RawNode

        return self.cx.add_node(n_in, n_out, _RawNode(kind, params))

    def host_node(self, n_in, n_out, process):
        """FWGPU_HOST_NODE: the caller's own AudioNodeProcessor::process, run on the host inside the device-resident graph"""
        return self.cx.add_node(n_in, n_out, self.fa.HostNode(process))

    def remove_node(self, node):
        self.cx.remove_node(node)
        return 0

    def connect(self, src, sp, dst, dp, check_for_cycles=False):
        try:
            return self.cx.connect(src, sp, dst, dp, check_for_cycles)
        except self.fa.AddEdgeError as e:
            raise AddEdgeError(e.code)

    def disconnect(self, src, sp, dst, dp):
        return self.cx.disconnect(src, sp, dst, dp)

    def disconnect_by_edge_id(self, e):
        return self.cx.disconnect_by_edge_id(e)

    def cycle_detected(self):
        return self.cx.cycle_detected()

    def update(self):
        try:
            self.cx.update()
        except self.fa.CompileGraphError as e:
            raise CompileGraphError(e.code, str(e))

    def new_sample(self, fmt, channels, data):
        return self.cx.new_sample(fmt, channels, data)

    def _chk(self, rc):
        self.cx._check(rc)

    def set_param(self, node, param, value, at_block=0):
        self._chk(self.cx.L.fwgpu_node_set_param(self.cx.c, node, param, value, at_block))

    def sampler_set_sample(self, node, sample, stop_playback=False, at_block=0):
        self._chk(self.cx.L.fwgpu_sampler_set_sample(self.cx.c, node, sample, int(stop_playback), at_block))

    def sampler_play(self, node, at_block=0):
        self._chk(self.cx.L.fwgpu_sampler_play(self.cx.c, node, at_block))

    def sampler_pause(self, node, at_block=0):
        self._chk(self.cx.L.fwgpu_sampler_pause(self.cx.c, node, at_block))

    def sampler_stop(self, node, at_block=0):
        self._chk(self.cx.L.fwgpu_sampler_stop(self.cx.c, node, at_block))

    def sampler_set_playhead_secs(self, node, secs, at_block=0):
        self._chk(self.cx.L.fwgpu_sampler_set_playhead_secs(self.cx.c, node, secs, at_block))

    def sampler_set_loop_range(self, node, mode, start=0.0, end=0.0, at_block=0):
        self._chk(self.cx.L.fwgpu_sampler_set_loop_range(self.cx.c, node, mode, start, end, at_block))

    def process_interleaved(self, frames, n_out_ch=2, inp=None, n_in_ch=0, t=0.0, status=0):
        return self.cx.process_interleaved(inp, n_in_ch, n_out_ch, frames, t, status)

    def process_blocks(self, k, n_out_ch=2):
        if getattr(self, "async_device", False):
            # the throughput form of the call: output left in HBM, NO host sync between calls; read back when the
            # result is first looked at
            import torch

            buf = torch.empty(k * self.max_block_frames * n_out_ch, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()  # the allocator's stream is not the ctx stream
            self.cx.process_blocks_device(k, buf.data_ptr(), n_out_ch)
            return _DeviceResult(self.cx, buf)
        return self.process_interleaved(k * self.max_block_frames, n_out_ch)

    def process_blocks_flags(self, k, n_out_ch=2):
        """(interleaved output, uint8 [k][n_out_ch] graph-output silence flags) through fwgpu_process_blocks_device_flags"""
        import torch

        buf = torch.full((k * self.max_block_frames * n_out_ch,), float("nan"), dtype=torch.float32, device="cuda")
        fl = torch.full((k * n_out_ch,), 77, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        self.cx.process_blocks_device_flags(k, buf.data_ptr(), n_out_ch, fl.data_ptr())
        self.cx.synchronize()
        return buf.cpu().numpy(), fl.cpu().numpy().reshape(k, n_out_ch)

    def node_process(self, node, frames, inputs, n_out, in_mask=0, out_mask=0, out_init=None):
        outs = [np.full(frames, np.nan, dtype=np.float32) if out_init is None else np.array(out_init[c], dtype=np.float32)
                for c in range(n_out)]
        om = self.cx.node_process(node, frames, inputs, outs, in_mask, out_mask)
        return (np.stack(outs) if outs else np.zeros((0, frames), np.float32)), om


_hostonly_lib = None


def host_sources():
    """libfwgpu's host translation units, as its own Makefile names them (everything but the device TU)"""
    csrc = os.path.join(ROOT, "firewheel_amd", "csrc")
    import re

    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = re.search(r"^SRCS\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    return [os.path.join(csrc, x) for x in srcs if x.endswith(".cpp")]


def hostonly_lib():
    """The HOST half of libfwgpu (every .cpp of its Makefile: fwgpu_abi / _run / _plan_* / _control_math / _graph) built with g++ against tests/host_harness: a fake HIP
    runtime (host memory, inert streams) and no-op kernel launches.  Computes no audio — it lets the CPU tier test graph
    editing, planning, plan selection, batching, message bookkeeping and the error conventions of the C ABI."""
    global _hostonly_lib
    if _hostonly_lib is None:
        import firewheel_amd._lib as flib

        d = os.path.join(ROOT, "tests", "host_harness")
        csrc = os.path.join(ROOT, "firewheel_amd", "csrc")
        so = os.path.join(d, "_hostonly.so")
        srcs = [os.path.join(d, "launch_stubs.cpp")] + host_sources()
        deps = srcs + [os.path.join(csrc, h) for h in ("fwgpu_graph.h", "fwgpu_types.h", "fwgpu_launch.h", "fwgpu_ctx.h")] + [
            os.path.join(ROOT, "include", "fwgpu.h"), os.path.join(d, "fakehip", "hip", "hip_runtime_api.h")]
        if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in deps):
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wall", "-Wno-unused-function",
                                   "-I", os.path.join(d, "fakehip"), "-I", os.path.join(ROOT, "include"), "-o", so] + srcs)
        L = C.CDLL(so)
        for name, (res, args) in flib.SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        L.fwh_launch_count.restype = C.c_ulonglong
        L.fwh_launch_count.argtypes = [C.c_int]
        L.fwh_launch_reset.restype = None
        L.fwh_violation.restype = C.c_char_p
        L.fwh_violation_reset.restype = None
        L.fwh_alloc_count.restype = C.c_ulonglong
        _hostonly_lib = L
    return _hostonly_lib


def hostonly_ctx(**kw):
    """firewheel_amd.FirewheelGpuCtx (the typed host mirror bench.py uses) on the host-only harness library"""
    import firewheel_amd as fa
    import firewheel_amd._lib as flib

    saved = flib._lib
    flib._lib = hostonly_lib()
    try:
        return fa.FirewheelGpuCtx(**kw)
    finally:
        flib._lib = saved


class HostOnlyEngine(GpuEngine):
    """GpuEngine's surface on the host-only harness library (CPU tier; outputs are meaningless, only the host logic runs)."""

    backend = "hostonly"

    def __init__(self, sample_rate=48000, max_block_frames=256, num_graph_inputs=0, num_graph_outputs=2,
                 force_generic=False, max_batch=None):
        import firewheel_amd as fa
        import firewheel_amd._lib as flib

        self.fa = fa
        self.sample_rate = sample_rate
        self.max_block_frames = max_block_frames
        saved = flib._lib
        flib._lib = hostonly_lib()
        try:
            self.cx = fa.FirewheelGpuCtx(sample_rate, max_block_frames, num_graph_inputs, num_graph_outputs)
        finally:
            flib._lib = saved
        if force_generic:
            self.cx.set_force_generic(True)
        if max_batch:
            self.cx.set_max_batch(max_batch)

    def launches(self):
        """kernel launches since the last reset: dict by kind"""
        L = hostonly_lib()
        names = ("level", "voice_control", "leaf_sum", "chain", "bus_sum", "root_out", "fir", "other")
        return {n: int(L.fwh_launch_count(i)) for i, n in enumerate(names)}

    def reset_launches(self):
        hostonly_lib().fwh_launch_reset()

    def violation(self):
        """first descriptor invariant / table extent the launch stubs found violated since the last reset ('' = none)"""
        return hostonly_lib().fwh_violation().decode()


def make_engine(backend, **kw):
    if backend == "oracle":
        kw.pop("force_generic", None)
        kw.pop("max_batch", None)
        return OracleEngine(**kw)
    return GpuEngine(**kw)


def bits(a):
    """bit pattern view for exact comparisons (distinguishes -0.0 / NaN payloads)."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
