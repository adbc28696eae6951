// ffi.rs — GENERATED from include/fwgpu.h by scripts/gen_rust_ffi.py (do not edit; tests/test_abi.py keeps it in sync).
// The raw C ABI of libfwgpu, the MI355X executor behind Firewheel's AudioNodeProcessor / FirewheelProcessor.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct fwgpu_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct fwgpu_stream {
    _private: [u8; 0],
}
#[repr(C)]
pub struct fwgpu_bus_exchange {
    _private: [u8; 0],
}
pub const FWGPU_EXCHANGE_HANDLE_BYTES: usize = 128;
#[repr(C)]
pub struct fwgpu_rccl_comm {
    _private: [u8; 0],
}
pub const FWGPU_RCCL_UNIQUE_ID_BYTES: usize = 128;
/// AudioNodeProcessor::process + ProcInfo (core/node.rs:37-53,94-118) as the C callback of a FWGPU_HOST_NODE
pub type fwgpu_host_process_fn = Option<
    unsafe extern "C" fn(
        user: *mut c_void,
        frames: u64,
        inputs: *const *const f32,
        num_inputs: u32,
        outputs: *const *mut f32,
        num_outputs: u32,
        in_silence_mask: u64,
        out_silence_mask: *mut u64,
        stream_time_secs: f64,
        stream_status: u32,
    ),
>;
/// one ScheduledNode of Firewheel's CompiledSchedule (graph/graph/compiler/schedule.rs:12-30)
#[repr(C)]
pub struct fwgpu_sched_node {
    pub node: i64,
    pub num_inputs: u32,
    pub num_outputs: u32,
    pub in_buffer_index: *const u32,
    pub in_should_clear: *const u8,
    pub out_buffer_index: *const u32,
}

// enum fwgpu_node_kind
pub const FWGPU_DUMMY: c_int = 0;
pub const FWGPU_BEEP_TEST: c_int = 1;
pub const FWGPU_VOLUME: c_int = 2;
pub const FWGPU_SUM: c_int = 3;
pub const FWGPU_SAMPLER: c_int = 4;
pub const FWGPU_HARD_CLIP: c_int = 5;
pub const FWGPU_MONO_TO_STEREO: c_int = 6;
pub const FWGPU_STEREO_TO_MONO: c_int = 7;
pub const FWGPU_STEREO_PAN: c_int = 8;
pub const FWGPU_STEREO_WIDTH: c_int = 9;
pub const FWGPU_BIQUAD: c_int = 10;
pub const FWGPU_DELAY: c_int = 11;
pub const FWGPU_FIR: c_int = 12;
pub const FWGPU_RESAMPLER: c_int = 13;
pub const FWGPU_SPATIAL: c_int = 14;
pub const FWGPU_HOST_NODE: c_int = 15;

// enum fwgpu_sample_format
pub const FWGPU_INTERLEAVED_I16: c_int = 0;
pub const FWGPU_INTERLEAVED_U16: c_int = 1;
pub const FWGPU_INTERLEAVED_F32: c_int = 2;
pub const FWGPU_PLANAR_I16: c_int = 3;
pub const FWGPU_PLANAR_U16: c_int = 4;
pub const FWGPU_PLANAR_F32: c_int = 5;

// enum fwgpu_error
pub const FWGPU_OK: c_int = 0;
pub const FWGPU_ERR_SRC_NODE_NOT_FOUND: c_int = -1;
pub const FWGPU_ERR_DST_NODE_NOT_FOUND: c_int = -2;
pub const FWGPU_ERR_IN_PORT_OUT_OF_RANGE: c_int = -3;
pub const FWGPU_ERR_OUT_PORT_OUT_OF_RANGE: c_int = -4;
pub const FWGPU_ERR_EDGE_ALREADY_EXISTS: c_int = -5;
pub const FWGPU_ERR_INPUT_PORT_ALREADY_CONNECTED: c_int = -6;
pub const FWGPU_ERR_CYCLE_DETECTED: c_int = -7;
pub const FWGPU_ERR_COMPILE_CYCLE: c_int = -10;
pub const FWGPU_ERR_COMPILE_MANY_TO_ONE: c_int = -11;
pub const FWGPU_ERR_NODE_ACTIVATION_FAILED: c_int = -12;
pub const FWGPU_ERR_INVALID: c_int = -20;
pub const FWGPU_ERR_QUEUE_FULL: c_int = -21;
pub const FWGPU_ERR_DEVICE: c_int = -30;

#[link(name = "fwgpu")]
extern "C" {
    pub fn fwgpu_ctx_create(device: c_int, sample_rate: u32, max_block_frames: u32, num_graph_inputs: u32, num_graph_outputs: u32, hip_stream: *mut c_void) -> *mut fwgpu_ctx;
    pub fn fwgpu_ctx_destroy(ctx: *mut fwgpu_ctx);
    pub fn fwgpu_last_error(ctx: *mut fwgpu_ctx) -> *const c_char;
    pub fn fwgpu_create_error() -> *const c_char;
    pub fn fwgpu_graph_in_node(ctx: *mut fwgpu_ctx) -> i64;
    pub fn fwgpu_graph_out_node(ctx: *mut fwgpu_ctx) -> i64;
    pub fn fwgpu_add_node(ctx: *mut fwgpu_ctx, kind: c_int, num_inputs: u32, num_outputs: u32, params: *const f32, n_params: c_int) -> i64;
    pub fn fwgpu_remove_node(ctx: *mut fwgpu_ctx, node: i64) -> c_int;
    pub fn fwgpu_connect(ctx: *mut fwgpu_ctx, src_node: i64, src_port: u32, dst_node: i64, dst_port: u32, check_for_cycles: c_int) -> i64;
    pub fn fwgpu_disconnect(ctx: *mut fwgpu_ctx, src_node: i64, src_port: u32, dst_node: i64, dst_port: u32) -> c_int;
    pub fn fwgpu_disconnect_edge(ctx: *mut fwgpu_ctx, edge: i64) -> c_int;
    pub fn fwgpu_cycle_detected(ctx: *mut fwgpu_ctx) -> c_int;
    pub fn fwgpu_update(ctx: *mut fwgpu_ctx) -> c_int;
    pub fn fwgpu_host_node_set_process(ctx: *mut fwgpu_ctx, node: i64, r#fn: fwgpu_host_process_fn, user: *mut c_void) -> c_int;
    pub fn fwgpu_plan_host_nodes(ctx: *mut fwgpu_ctx, callbacks_run: *mut u64) -> c_int;
    pub fn fwgpu_schedule_upload(ctx: *mut fwgpu_ctx, nodes: *const fwgpu_sched_node, n_nodes: u32, num_buffers: u32) -> c_int;
    pub fn fwgpu_plan_kind(ctx: *mut fwgpu_ctx) -> c_int;
    pub fn fwgpu_plan_fused_voices(ctx: *mut fwgpu_ctx) -> c_int;
    pub fn fwgpu_plan_num_levels(ctx: *mut fwgpu_ctx) -> c_int;
    pub fn fwgpu_plan_node_level(ctx: *mut fwgpu_ctx, node: i64) -> c_int;
    pub fn fwgpu_plan_node_inputs_clear(ctx: *mut fwgpu_ctx, node: i64, should_clear: *mut c_int, cap: c_int) -> c_int;
    pub fn fwgpu_plan_chain_stats(ctx: *mut fwgpu_ctx, steady_workgroups: *mut u64, general_workgroups: *mut u64) -> c_int;
    pub fn fwgpu_plan_handover_stats(ctx: *mut fwgpu_ctx, adoptions: *mut u64, audio_adoptions: *mut u64, max_adopt_ns: *mut u64) -> c_int;
    pub fn fwgpu_plan_pending(ctx: *mut fwgpu_ctx) -> c_int;
    pub fn fwgpu_lazy_stats(ctx: *mut fwgpu_ctx, lazy_batches: *mut u64, control_batches: *mut u64) -> c_int;
    pub fn fwgpu_hip_stream(ctx: *mut fwgpu_ctx) -> *mut c_void;
    pub fn fwgpu_update_phase(ctx: *mut fwgpu_ctx) -> c_int;
    pub fn fwgpu_rt_resident_stats(ctx: *mut fwgpu_ctx, launches: *mut u64, doorbells: *mut u64) -> c_int;
    pub fn fwgpu_rt_path_stats(ctx: *mut fwgpu_ctx, paths: *mut u64) -> c_int;
    pub fn fwgpu_set_max_batch(ctx: *mut fwgpu_ctx, max_blocks: u32) -> c_int;
    pub fn fwgpu_set_force_generic(ctx: *mut fwgpu_ctx, on: c_int) -> c_int;
    pub fn fwgpu_ext_pool_floats(ctx: *mut fwgpu_ctx, in_use: *mut u64, capacity: *mut u64) -> c_int;
    pub fn fwgpu_sample_create(ctx: *mut fwgpu_ctx, format: c_int, channels: u32, frames: u64, data: *const c_void) -> c_int;
    pub fn fwgpu_sample_create_device(ctx: *mut fwgpu_ctx, format: c_int, channels: u32, frames: u64, device_data: *const c_void) -> c_int;
    pub fn fwgpu_sample_destroy(ctx: *mut fwgpu_ctx, sample: c_int) -> c_int;
    pub fn fwgpu_poll_returned_samples(ctx: *mut fwgpu_ctx, nodes: *mut i64, samples: *mut c_int, cap: c_int) -> c_int;
    pub fn fwgpu_sample_retired(ctx: *mut fwgpu_ctx, sample: c_int) -> c_int;
    pub fn fwgpu_node_set_param(ctx: *mut fwgpu_ctx, node: i64, param: c_int, value: f32, at_block: u32) -> c_int;
    pub fn fwgpu_node_set_params(ctx: *mut fwgpu_ctx, n: u32, nodes: *const i64, params: *const c_int, values: *const f32, at_blocks: *const u32) -> c_int;
    pub fn fwgpu_sampler_set_sample(ctx: *mut fwgpu_ctx, node: i64, sample: c_int, stop_playback: c_int, at_block: u32) -> c_int;
    pub fn fwgpu_sampler_play(ctx: *mut fwgpu_ctx, node: i64, at_block: u32) -> c_int;
    pub fn fwgpu_sampler_pause(ctx: *mut fwgpu_ctx, node: i64, at_block: u32) -> c_int;
    pub fn fwgpu_sampler_stop(ctx: *mut fwgpu_ctx, node: i64, at_block: u32) -> c_int;
    pub fn fwgpu_sampler_set_playhead_secs(ctx: *mut fwgpu_ctx, node: i64, playhead_secs: f64, at_block: u32) -> c_int;
    pub fn fwgpu_sampler_set_loop_range(ctx: *mut fwgpu_ctx, node: i64, mode: c_int, start_secs: f64, end_secs: f64, at_block: u32) -> c_int;
    pub fn fwgpu_process_interleaved(ctx: *mut fwgpu_ctx, input: *const f32, output: *mut f32, num_in_channels: u32, num_out_channels: u32, frames: u64, stream_time_secs: f64, stream_status: u32) -> c_int;
    pub fn fwgpu_process_interleaved_begin(ctx: *mut fwgpu_ctx, input: *const f32, num_in_channels: u32, num_out_channels: u32, frames: u64, stream_time_secs: f64, stream_status: u32) -> i64;
    pub fn fwgpu_process_interleaved_end(ctx: *mut fwgpu_ctx, ticket: i64, output: *mut f32) -> c_int;
    pub fn fwgpu_process_interleaved_cancel(ctx: *mut fwgpu_ctx, ticket: i64) -> c_int;
    pub fn fwgpu_process_blocks_device(ctx: *mut fwgpu_ctx, num_blocks: u32, d_output: *mut f32, num_out_channels: u32) -> c_int;
    pub fn fwgpu_process_blocks_device_flags(ctx: *mut fwgpu_ctx, num_blocks: u32, d_output: *mut f32, num_out_channels: u32, d_silence: *mut u8) -> c_int;
    pub fn fwgpu_process_blocks_device_io(ctx: *mut fwgpu_ctx, num_blocks: u32, d_input: *const f32, num_in_channels: u32, d_output: *mut f32, num_out_channels: u32, d_silence: *mut u8) -> c_int;
    pub fn fwgpu_bus_sum_ordered(ctx: *mut fwgpu_ctx, d_parts: *const *const f32, n_parts: u32, d_out: *mut f32, n_floats: u64) -> c_int;
    pub fn fwgpu_bus_sum_ordered_flags(ctx: *mut fwgpu_ctx, d_parts: *const *const f32, d_silence: *const *const u8, n_parts: u32, d_out: *mut f32, d_out_silence: *mut u8, n_floats: u64, frames_per_block: u32, n_channels: u32) -> c_int;
    pub fn fwgpu_bus_exchange_open(ctx: *mut fwgpu_ctx, rank: u32, world: u32, max_floats: u64, max_silence_bytes: u32) -> *mut fwgpu_bus_exchange;
    pub fn fwgpu_bus_exchange_close(ex: *mut fwgpu_bus_exchange);
    pub fn fwgpu_bus_exchange_export(ex: *mut fwgpu_bus_exchange, handle: *mut c_void) -> c_int;
    pub fn fwgpu_bus_exchange_connect(ex: *mut fwgpu_bus_exchange, peer_rank: u32, handle: *const c_void) -> c_int;
    pub fn fwgpu_bus_exchange_set_timeout_ms(ex: *mut fwgpu_bus_exchange, ms: u32) -> c_int;
    pub fn fwgpu_bus_exchange_push(ex: *mut fwgpu_bus_exchange, d_partial: *const f32, d_silence: *const u8, n_floats: u64, n_blocks: u32, n_channels: u32) -> c_int;
    pub fn fwgpu_bus_exchange_reduce(ex: *mut fwgpu_bus_exchange, d_out: *mut f32, d_out_silence: *mut u8, n_floats: u64, n_blocks: u32, frames_per_block: u32, n_channels: u32, have_silence: c_int) -> c_int;
    pub fn fwgpu_bus_exchange_step(ex: *mut fwgpu_bus_exchange, d_partial: *const f32, d_silence: *const u8, d_out: *mut f32, d_out_silence: *mut u8, n_floats: u64, n_blocks: u32, frames_per_block: u32, n_channels: u32) -> c_int;
    pub fn fwgpu_bus_exchange_status(ex: *mut fwgpu_bus_exchange, steps: *mut u64, failed_step: *mut u64) -> c_int;
    pub fn fwgpu_bus_exchange_wait_stats(ex: *mut fwgpu_bus_exchange, max_wait_us: *mut u64, cap: u32, reset: c_int) -> c_int;
    pub fn fwgpu_rccl_unique_id(id: *mut u8) -> c_int;
    pub fn fwgpu_rccl_comm_create(ctx: *mut fwgpu_ctx, id: *const u8, world: u32, rank: u32) -> *mut fwgpu_rccl_comm;
    pub fn fwgpu_rccl_comm_destroy(comm: *mut fwgpu_rccl_comm) -> c_int;
    pub fn fwgpu_rccl_comm_info(comm: *mut fwgpu_rccl_comm, world: *mut u32, rank: *mut u32) -> c_int;
    pub fn fwgpu_bus_allreduce_rccl(comm: *mut fwgpu_rccl_comm, d_bus: *mut f32, n_floats: u64) -> c_int;
    pub fn fwgpu_bus_allgather_ordered(comm: *mut fwgpu_rccl_comm, d_bus: *const f32, d_silence: *const u8, d_out: *mut f32, d_out_silence: *mut u8, n_floats: u64, frames_per_block: u32, n_channels: u32) -> c_int;
    pub fn fwgpu_rccl_last_error() -> *const c_char;
    pub fn fwgpu_synchronize(ctx: *mut fwgpu_ctx) -> c_int;
    pub fn fwgpu_proc_info(ctx: *mut fwgpu_ctx, stream_time_secs: *mut f64, stream_status: *mut u32, output_underflows: *mut u64, input_overflows: *mut u64) -> c_int;
    pub fn fwgpu_stream_open(ctx: *mut fwgpu_ctx, num_in_channels: u32, num_out_channels: u32) -> *mut fwgpu_stream;
    pub fn fwgpu_stream_close(s: *mut fwgpu_stream);
    pub fn fwgpu_stream_callback(s: *mut fwgpu_stream, output: *mut f32, frames: u64, callback_instant_secs: f64) -> c_int;
    pub fn fwgpu_stream_stats(s: *mut fwgpu_stream, callbacks: *mut u64, underflows: *mut u64, last_stream_time_secs: *mut f64) -> c_int;
    pub fn fwgpu_stream_run(s: *mut fwgpu_stream, output: *mut f32, frames: u64, n_callbacks: u32, first_instant_secs: f64, elapsed_secs: *mut f64) -> c_int;
    pub fn fwgpu_node_process(ctx: *mut fwgpu_ctx, node: i64, frames: u64, inputs: *const *const f32, num_inputs: u32, outputs: *const *mut f32, num_outputs: u32, in_silence_mask: u64, out_silence_mask: *mut u64, stream_time_secs: f64, stream_status: u32) -> c_int;
    pub fn fwgpu_timing_enable(ctx: *mut fwgpu_ctx, on: c_int) -> c_int;
    pub fn fwgpu_timing_read(ctx: *mut fwgpu_ctx, which: c_int, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn fwgpu_timing_reset(ctx: *mut fwgpu_ctx) -> c_int;
    pub fn fwgpu_device_info(ctx: *mut fwgpu_ctx, name: *mut c_char, name_cap: c_int, compute_units: *mut c_int, hbm_bytes: *mut u64) -> c_int;
}
