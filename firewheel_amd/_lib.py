"""ctypes binding of include/fwgpu.h.  Fails loudly when the HIP library is missing."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# FWGPU_LIB: kernel-variant experiments only (gpurun A/B runs); the product is csrc/libfwgpu.so
LIB_PATH = os.environ.get("FWGPU_LIB") or os.path.join(CSRC, "libfwgpu.so")


class FwgpuError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("fwgpu error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


def build_library(force=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", CSRC])
    return LIB_PATH


EXCHANGE_HANDLE_BYTES = 128  # FWGPU_EXCHANGE_HANDLE_BYTES
i64, u32, u64, f32, f64, vp, ci = C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double, C.c_void_p, C.c_int
fp = C.POINTER(C.c_float)


class SchedNode(C.Structure):
    _fields_ = [("node", i64), ("num_inputs", u32), ("num_outputs", u32), ("in_buffer_index", C.POINTER(u32)),
                ("in_should_clear", C.POINTER(C.c_uint8)), ("out_buffer_index", C.POINTER(u32))]


# fwgpu_host_process_fn: AudioNodeProcessor::process + ProcInfo as a C callback (FWGPU_HOST_NODE)
HOST_PROCESS_FN = C.CFUNCTYPE(None, vp, u64, C.POINTER(fp), u32, C.POINTER(fp), u32, u64, C.POINTER(u64), f64, u32)


def host_process_adapter(py_fn):
    """wrap `py_fn(frames, inputs, outputs, in_silence_mask, stream_time_secs, stream_status) -> out_silence_mask` (inputs /
    outputs: lists of float32 numpy views of the caller's buffers, outputs writable) as a fwgpu_host_process_fn"""
    import numpy as np

    def tramp(user, frames, inputs, n_in, outputs, n_out, in_mask, out_mask, t, status):
        ins = [np.ctypeslib.as_array(inputs[i], shape=(frames,)) for i in range(n_in)]
        outs = [np.ctypeslib.as_array(outputs[i], shape=(frames,)) for i in range(n_out)]
        m = py_fn(int(frames), ins, outs, int(in_mask), float(t), int(status))
        out_mask[0] = int(m or 0)

    return HOST_PROCESS_FN(tramp)


# every symbol include/fwgpu.h declares: (restype, argtypes)
SIGNATURES = {
    "fwgpu_ctx_create": (vp, [ci, u32, u32, u32, u32, vp]),
    "fwgpu_ctx_destroy": (None, [vp]),
    "fwgpu_last_error": (C.c_char_p, [vp]),
    "fwgpu_create_error": (C.c_char_p, []),
    "fwgpu_graph_in_node": (i64, [vp]),
    "fwgpu_graph_out_node": (i64, [vp]),
    "fwgpu_add_node": (i64, [vp, ci, u32, u32, fp, ci]),
    "fwgpu_remove_node": (ci, [vp, i64]),
    "fwgpu_connect": (i64, [vp, i64, u32, i64, u32, ci]),
    "fwgpu_disconnect": (ci, [vp, i64, u32, i64, u32]),
    "fwgpu_disconnect_edge": (ci, [vp, i64]),
    "fwgpu_cycle_detected": (ci, [vp]),
    "fwgpu_update": (ci, [vp]),
    "fwgpu_host_node_set_process": (ci, [vp, i64, HOST_PROCESS_FN, vp]),
    "fwgpu_plan_host_nodes": (ci, [vp, C.POINTER(u64)]),
    "fwgpu_schedule_upload": (ci, [vp, C.POINTER(SchedNode), u32, u32]),
    "fwgpu_plan_kind": (ci, [vp]),
    "fwgpu_plan_fused_voices": (ci, [vp]),
    "fwgpu_plan_num_levels": (ci, [vp]),
    "fwgpu_plan_node_level": (ci, [vp, i64]),
    "fwgpu_plan_node_inputs_clear": (ci, [vp, i64, C.POINTER(ci), ci]),
    "fwgpu_plan_handover_stats": (ci, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
    "fwgpu_plan_pending": (ci, [vp]),
    "fwgpu_lazy_stats": (ci, [vp, C.POINTER(u64), C.POINTER(u64)]),
    "fwgpu_hip_stream": (vp, [vp]),
    "fwgpu_update_phase": (ci, [vp]),
    "fwgpu_rt_resident_stats": (ci, [vp, C.POINTER(u64), C.POINTER(u64)]),
    "fwgpu_rt_path_stats": (ci, [vp, C.POINTER(u64)]),
    "fwgpu_plan_chain_stats": (ci, [vp, C.POINTER(u64), C.POINTER(u64)]),
    "fwgpu_set_max_batch": (ci, [vp, u32]),
    "fwgpu_set_force_generic": (ci, [vp, ci]),
    "fwgpu_sample_create": (ci, [vp, ci, u32, u64, vp]),
    "fwgpu_sample_create_device": (ci, [vp, ci, u32, u64, vp]),
    "fwgpu_sample_destroy": (ci, [vp, ci]),
    "fwgpu_poll_returned_samples": (ci, [vp, C.POINTER(i64), C.POINTER(ci), ci]),
    "fwgpu_sample_retired": (ci, [vp, ci]),
    "fwgpu_ext_pool_floats": (ci, [vp, C.POINTER(u64), C.POINTER(u64)]),
    "fwgpu_proc_info": (ci, [vp, C.POINTER(f64), C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]),
    "fwgpu_stream_open": (vp, [vp, u32, u32]),
    "fwgpu_stream_close": (None, [vp]),
    "fwgpu_stream_callback": (ci, [vp, fp, u64, f64]),
    "fwgpu_stream_stats": (ci, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(f64)]),
    "fwgpu_stream_run": (ci, [vp, fp, u64, u32, f64, C.POINTER(f64)]),
    "fwgpu_node_set_param": (ci, [vp, i64, ci, f32, u32]),
    "fwgpu_node_set_params": (ci, [vp, u32, C.POINTER(i64), C.POINTER(ci), C.POINTER(f32), C.POINTER(u32)]),
    "fwgpu_sampler_set_sample": (ci, [vp, i64, ci, ci, u32]),
    "fwgpu_sampler_play": (ci, [vp, i64, u32]),
    "fwgpu_sampler_pause": (ci, [vp, i64, u32]),
    "fwgpu_sampler_stop": (ci, [vp, i64, u32]),
    "fwgpu_sampler_set_playhead_secs": (ci, [vp, i64, f64, u32]),
    "fwgpu_sampler_set_loop_range": (ci, [vp, i64, ci, f64, f64, u32]),
    "fwgpu_process_interleaved": (ci, [vp, fp, fp, u32, u32, u64, f64, u32]),
    "fwgpu_process_interleaved_begin": (i64, [vp, fp, u32, u32, u64, f64, u32]),
    "fwgpu_process_interleaved_end": (ci, [vp, i64, fp]),
    "fwgpu_process_interleaved_cancel": (ci, [vp, i64]),
    "fwgpu_process_blocks_device": (ci, [vp, u32, vp, u32]),
    "fwgpu_process_blocks_device_flags": (ci, [vp, u32, vp, u32, vp]),
    "fwgpu_process_blocks_device_io": (ci, [vp, u32, vp, u32, vp, u32, vp]),
    "fwgpu_rccl_unique_id": (ci, [vp]),
    "fwgpu_rccl_comm_create": (vp, [vp, vp, u32, u32]),
    "fwgpu_rccl_comm_destroy": (ci, [vp]),
    "fwgpu_rccl_comm_info": (ci, [vp, C.POINTER(u32), C.POINTER(u32)]),
    "fwgpu_bus_allreduce_rccl": (ci, [vp, vp, u64]),
    "fwgpu_bus_allgather_ordered": (ci, [vp, vp, vp, vp, vp, u64, u32, u32]),
    "fwgpu_rccl_last_error": (C.c_char_p, []),
    "fwgpu_bus_sum_ordered": (ci, [vp, C.POINTER(vp), u32, vp, u64]),
    "fwgpu_bus_sum_ordered_flags": (ci, [vp, C.POINTER(vp), C.POINTER(vp), u32, vp, vp, u64, u32, u32]),
    "fwgpu_bus_exchange_open": (vp, [vp, u32, u32, u64, u32]),
    "fwgpu_bus_exchange_close": (None, [vp]),
    "fwgpu_bus_exchange_export": (ci, [vp, vp]),
    "fwgpu_bus_exchange_connect": (ci, [vp, u32, vp]),
    "fwgpu_bus_exchange_set_timeout_ms": (ci, [vp, u32]),
    "fwgpu_bus_exchange_push": (ci, [vp, vp, vp, u64, u32, u32]),
    "fwgpu_bus_exchange_reduce": (ci, [vp, vp, vp, u64, u32, u32, u32, ci]),
    "fwgpu_bus_exchange_step": (ci, [vp, vp, vp, vp, vp, u64, u32, u32, u32]),
    "fwgpu_bus_exchange_status": (ci, [vp, C.POINTER(u64), C.POINTER(u64)]),
    "fwgpu_bus_exchange_wait_stats": (ci, [vp, C.POINTER(u64), u32, ci]),
    "fwgpu_synchronize": (ci, [vp]),
    "fwgpu_node_process": (ci, [vp, i64, u64, C.POINTER(fp), u32, C.POINTER(fp), u32, u64, C.POINTER(u64), f64, u32]),
    "fwgpu_timing_enable": (ci, [vp, ci]),
    "fwgpu_timing_read": (ci, [vp, ci, C.POINTER(f64), C.POINTER(u64)]),
    "fwgpu_timing_reset": (ci, [vp]),
    "fwgpu_device_info": (ci, [vp, C.c_char_p, ci, C.POINTER(ci), C.POINTER(u64)]),
}

_lib = None


def load_library():
    """dlopen libfwgpu.so and bind every declared symbol.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FwgpuError(-30, "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)  # AttributeError here = header/library drift
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib
