//! firewheel-gpu — the reference-side binding of libfwgpu (include/fwgpu.h).
//!
//! Two levels, as in INTEGRATION.md:
//! * **B2** ([`GpuContext`] + [`GpuProcessor`]): the whole `FirewheelProcessor::process_interleaved`
//!   (firewheel-graph/src/processor.rs:61-165) runs on the device.  Firewheel keeps its `AudioGraph` and its scheduler;
//!   after every successful compile the `CompiledSchedule` is handed over with [`GpuContext::upload_schedule`].
//! * **B1** ([`nodes`]): each built-in node as an `AudioNode` whose processor half forwards
//!   `AudioNodeProcessor::process` (firewheel-core/src/node.rs:37-53) to `fwgpu_node_process` — the literal trait
//!   drop-in, for graphs that mix GPU nodes with custom Rust nodes.
//!
//! SOURCE ONLY here: never compiled in the repository that ships it (no Rust toolchain in its build image).
pub mod exchange;
pub mod ffi;
pub mod host_node;
pub mod nodes;
pub mod sample;
pub mod stream;

use std::ffi::CStr;
use std::fmt;
use std::ptr::NonNull;
use std::sync::{Arc, Mutex, MutexGuard};

/// Error of a libfwgpu call: the negative code of `enum fwgpu_error` + `fwgpu_last_error`.
#[derive(Debug, Clone)]
pub struct GpuError {
    pub code: i32,
    pub message: String,
}
impl fmt::Display for GpuError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "fwgpu error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for GpuError {}

/// Owner of one `fwgpu_ctx`.  Shared (`Arc`) between the control side (graph edits, node handles, sample handles) and the
/// one audio-side [`GpuProcessor`].  The C ABI's threading contract (fwgpu.h) is what makes that sound:
/// * every CONTROL call — messages, graph edits, `update`, sample-table calls — may run while a process call is in flight
///   (`fwgpu_update` builds the new plan off to the side and the next process call adopts it, the way Firewheel hands a
///   schedule over through its ring, graph/processor.rs:167-206), but the control side is ONE thread at a time.  Rust nodes
///   and handles can live on any thread, so every control call of this crate goes through `control()`: a mutex the audio
///   thread never touches.
/// * the AUDIO calls belong to the single [`GpuProcessor`] (`Send`, not `Sync`, `&mut self`).
pub struct GpuContext {
    raw: NonNull<ffi::fwgpu_ctx>,
    control: Mutex<()>,
    /// processors of host nodes that were removed while a plan could still call them (host_node.rs)
    pub(crate) graveyard: Mutex<Vec<host_node::Grave>>,
    /// successful `update` / `upload_schedule` calls so far.  Written and read under `control` only where it decides anything: a
    /// grave is stamped with the value at its removal, and is freed once a LATER compile — one that ran entirely after the removal,
    /// both under the lock — has been adopted (marking graves after the lock was released let a removal slip in between
    /// another thread's `fwgpu_update` and its mark, and freed a processor the adopted plan still called)
    pub(crate) compile_gen: std::sync::atomic::AtomicU64,
    /// the graph's one global user context (`ProcInfo::cx`, core/node.rs:117-118; processor.rs:21 owns one per graph): every host
    /// node's `process` is lent THIS box.  Written by `set_user_cx` before the stream starts, dereferenced by the audio thread only.
    user_cx: std::cell::UnsafeCell<Box<dyn std::any::Any + Send>>,
    pub sample_rate: u32,
    pub max_block_frames: u32,
}
// sound because of the two rules above: control calls are serialised by `control`, audio calls by `&mut GpuProcessor`
unsafe impl Send for GpuContext {}
unsafe impl Sync for GpuContext {}

impl GpuContext {
    /// `FirewheelGraphCtx::new` + `activate` (graph/context.rs:35-82) for the device executor.
    pub fn new(
        device: i32,
        sample_rate: u32,
        max_block_frames: u32,
        num_graph_inputs: u32,
        num_graph_outputs: u32,
    ) -> Result<Arc<Self>, GpuError> {
        let raw = unsafe {
            ffi::fwgpu_ctx_create(
                device,
                sample_rate,
                max_block_frames,
                num_graph_inputs,
                num_graph_outputs,
                std::ptr::null_mut(),
            )
        };
        match NonNull::new(raw) {
            Some(raw) => Ok(Arc::new(Self { raw, control: Mutex::new(()), graveyard: Mutex::new(Vec::new()), compile_gen: std::sync::atomic::AtomicU64::new(0), user_cx: std::cell::UnsafeCell::new(Box::new(())), sample_rate, max_block_frames })),
            None => Err(GpuError {
                code: ffi::FWGPU_ERR_DEVICE,
                message: unsafe { CStr::from_ptr(ffi::fwgpu_create_error()) }.to_string_lossy().into_owned(),
            }),
        }
    }

    pub fn as_ptr(&self) -> *mut ffi::fwgpu_ctx {
        self.raw.as_ptr()
    }

    /// The `user_cx` of `FirewheelGraphCtx::activate` (graph/context.rs:53-82 hands it to the processor, processor.rs:41): ONE per
    /// graph, shared by every custom node (the default is `Box::new(())`).
    ///
    /// # Safety
    /// The old box is dropped here while host-node trampolines lend `&mut` of it to `process()` on the audio thread, which the control
    /// lock does not exclude.  The caller must guarantee that no process call of this context is running or can start during the call
    /// — "before the stream starts", as in the reference, where `activate` consumes the context.  (this was a safe method
    /// on a `Sync` type whose contract lived in a comment.)
    pub unsafe fn set_user_cx(&self, user_cx: Box<dyn std::any::Any + Send>) {
        let _g = self.control();
        *self.user_cx.get() = user_cx;
    }
    pub(crate) fn user_cx_ptr(&self) -> *mut Box<dyn std::any::Any + Send> {
        self.user_cx.get()
    }

    /// The control side's lock: hold it around every control call made through `as_ptr()` directly (messages, samples).
    /// Never taken by the audio thread; a poisoned lock is still a lock (the C side keeps its own invariants).
    pub fn control(&self) -> MutexGuard<'_, ()> {
        self.control.lock().unwrap_or_else(|e| e.into_inner())
    }

    pub(crate) fn check(&self, rc: i64) -> Result<i64, GpuError> {
        if rc >= 0 {
            return Ok(rc);
        }
        Err(GpuError {
            code: rc as i32,
            message: unsafe { CStr::from_ptr(ffi::fwgpu_last_error(self.as_ptr())) }.to_string_lossy().into_owned(),
        })
    }

    /// `AudioGraph::add_node` (graph/graph.rs:201-231) for a built-in node kind; `params` are its constructor arguments.
    pub fn add_node(&self, kind: i32, num_inputs: u32, num_outputs: u32, params: &[f32]) -> Result<i64, GpuError> {
        let _g = self.control();
        self.check(unsafe {
            ffi::fwgpu_add_node(self.as_ptr(), kind, num_inputs, num_outputs, params.as_ptr(), params.len() as i32)
        })
    }
    pub fn remove_node(&self, node: i64) -> Result<(), GpuError> {
        let _g = self.control();
        self.check(unsafe { ffi::fwgpu_remove_node(self.as_ptr(), node) } as i64).map(|_| ())
    }
    /// `AudioGraph::connect` (graph/graph.rs:396-477); the error codes are the `AddEdgeError` variants.
    pub fn connect(&self, src: i64, src_port: u32, dst: i64, dst_port: u32, check_for_cycles: bool) -> Result<i64, GpuError> {
        let _g = self.control();
        self.check(unsafe { ffi::fwgpu_connect(self.as_ptr(), src, src_port, dst, dst_port, check_for_cycles as i32) })
    }
    pub fn graph_in_node(&self) -> i64 {
        unsafe { ffi::fwgpu_graph_in_node(self.as_ptr()) }
    }
    pub fn graph_out_node(&self) -> i64 {
        unsafe { ffi::fwgpu_graph_out_node(self.as_ptr()) }
    }
    /// `FirewheelGraphCtx::update` (graph/context.rs:93-137) when the graph is mirrored with `add_node` / `connect`.
    pub fn update(&self) -> Result<(), GpuError> {
        let r = {
            let _g = self.control();
            let r = self.check(unsafe { ffi::fwgpu_update(self.as_ptr()) } as i64).map(|_| ());
            if r.is_ok() {
                self.compile_gen.fetch_add(1, std::sync::atomic::Ordering::AcqRel); // still under the lock: see `compile_gen`
            }
            r
        };
        self.reap();
        r
    }
    /// true while a plan built by `update` / `upload_schedule` waits for a process call to adopt it (`fwgpu_plan_pending`)
    pub fn plan_pending(&self) -> bool {
        unsafe { ffi::fwgpu_plan_pending(self.as_ptr()) != 0 }
    }
    /// `fwgpu_node_set_params`: a list of (node, param, value, at_block) messages in one foreign call, in order; stops at the first
    /// one that fails.  (A Rust host pays nanoseconds per `fwgpu_node_set_param` call anyway: this is for symmetry with the C ABI.)
    pub fn set_params(&self, nodes: &[i64], params: &[i32], values: &[f32], at_blocks: &[u32]) -> Result<(), GpuError> {
        let n = nodes.len();
        assert!(params.len() == n && values.len() == n && at_blocks.len() == n);
        self.check(unsafe { ffi::fwgpu_node_set_params(self.as_ptr(), n as u32, nodes.as_ptr(), params.as_ptr(), values.as_ptr(), at_blocks.as_ptr()) } as i64)
            .map(|_| ())
    }
    /// (launch batches rendered without a control kernel, with one) — `fwgpu_lazy_stats`: a message-free call of a plain voice-bank
    /// plan derives its block records from per-voice records the last control kernel left behind.
    pub fn lazy_stats(&self) -> (u64, u64) {
        let (mut a, mut b) = (0u64, 0u64);
        unsafe { ffi::fwgpu_lazy_stats(self.as_ptr(), &mut a, &mut b) };
        (a, b)
    }
    /// The `hipStream_t` every process call launches on (`fwgpu_hip_stream`), for hosts that order their own device work against it.
    pub fn hip_stream(&self) -> *mut std::os::raw::c_void {
        unsafe { ffi::fwgpu_hip_stream(self.as_ptr()) }
    }
    /// Blocks one launch sequence may cover (a realtime host never needs more than one; an offline bounce wants many).
    pub fn set_max_batch(&self, blocks: u32) -> Result<(), GpuError> {
        let _g = self.control();
        self.check(unsafe { ffi::fwgpu_set_max_batch(self.as_ptr(), blocks) } as i64).map(|_| ())
    }

    /// Hand Firewheel's own `CompiledSchedule` over (keep the Rust scheduler): one entry per `ScheduledNode`
    /// (graph/graph/compiler/schedule.rs:12-30) in schedule order, `ids` maps Firewheel `NodeID`s to the ids
    /// `add_node` returned.  Called from `FirewheelGraphCtx::update` right after `graph.compile()` succeeded
    /// (graph/context.rs:118-126).
    pub fn upload_schedule<'a, I>(&self, schedule: I, num_buffers: usize) -> Result<(), GpuError>
    where
        I: IntoIterator<Item = ScheduledNodeView<'a>>,
    {
        let views: Vec<ScheduledNodeView<'a>> = schedule.into_iter().collect();
        let nodes: Vec<ffi::fwgpu_sched_node> = views
            .iter()
            .map(|v| ffi::fwgpu_sched_node {
                node: v.node,
                num_inputs: v.in_buffer_index.len() as u32,
                num_outputs: v.out_buffer_index.len() as u32,
                in_buffer_index: v.in_buffer_index.as_ptr(),
                in_should_clear: v.in_should_clear.as_ptr(),
                out_buffer_index: v.out_buffer_index.as_ptr(),
            })
            .collect();
        let r = {
            let _g = self.control();
            let r = self
                .check(unsafe {
                    ffi::fwgpu_schedule_upload(self.as_ptr(), nodes.as_ptr(), nodes.len() as u32, num_buffers as u32)
                } as i64)
                .map(|_| ());
            if r.is_ok() {
                self.compile_gen.fetch_add(1, std::sync::atomic::Ordering::AcqRel); // still under the lock: see `compile_gen`
            }
            r
        };
        self.reap();
        r
    }

    /// 0 = generic level-batched executor, 1 = fused voice-bank plan, 2 = fused chain plan, 3 = hybrid (voice banks on the fused
    /// kernels inside a graph the level executor runs).
    pub fn plan_kind(&self) -> i32 {
        unsafe { ffi::fwgpu_plan_kind(self.as_ptr()) }
    }

    /// (plans adopted so far, those a process call adopted, the longest one of those held its call up) — the schedule hand-over
    /// of graph/processor.rs:167-206 as libfwgpu does it
    pub fn plan_handover_stats(&self) -> (u64, u64, std::time::Duration) {
        let (mut a, mut b, mut ns) = (0u64, 0u64, 0u64);
        unsafe { ffi::fwgpu_plan_handover_stats(self.as_ptr(), &mut a, &mut b, &mut ns) };
        (a, b, std::time::Duration::from_nanos(ns))
    }

    /// Voices of the installed plan that the fused kernels render (plan 3: the voice banks' and the split mixers' leading ones).
    pub fn plan_fused_voices(&self) -> i32 {
        unsafe { ffi::fwgpu_plan_fused_voices(self.as_ptr()) }
    }
}

impl Drop for GpuContext {
    fn drop(&mut self) {
        // the ctx first (it ends every kernel and no audio call can be in flight: the GpuProcessor holds an Arc); the graveyard's
        // processors are fields and go after it
        unsafe { ffi::fwgpu_ctx_destroy(self.as_ptr()) }
    }
}

/// One begun, not yet ended `process_interleaved_begin` call (not `Clone`: `process_interleaved_end` consumes it).  A ticket that is
/// dropped instead — an error path, an unwinding panic — cancels itself and every older ticket still in flight
/// (`fwgpu_process_interleaved_cancel`): the two slots never stay busy behind a lost ticket.  `!Send`: like
/// `GpuProcessor`'s calls it belongs to the one audio thread.
pub struct Ticket {
    id: i64,
    floats: usize,
    cx: Arc<GpuContext>,
    ended: bool,
    _audio_thread: std::marker::PhantomData<*mut ()>,
}
impl Drop for Ticket {
    fn drop(&mut self) {
        if !self.ended {
            // Err = an older drop already swept it: nothing is left in flight under this id
            let _ = unsafe { ffi::fwgpu_process_interleaved_cancel(self.cx.as_ptr(), self.id) };
        }
    }
}

/// Borrowed view of one `ScheduledNode`: `InBufferAssignment { buffer_index, should_clear }` split into two slices.
pub struct ScheduledNodeView<'a> {
    pub node: i64,
    pub in_buffer_index: &'a [u32],
    pub in_should_clear: &'a [u8],
    pub out_buffer_index: &'a [u32],
}

/// The audio-thread half: what `FirewheelProcessor` is to `FirewheelGraphCtx`.  `Send`, not `Sync` — one audio thread.
pub struct GpuProcessor {
    cx: Arc<GpuContext>,
}
impl GpuProcessor {
    pub fn new(cx: Arc<GpuContext>) -> Self {
        Self { cx }
    }

    /// `FirewheelProcessor::process_interleaved` (graph/processor.rs:61-165), same arguments.  `output` is filled on
    /// every return (zeros on error: core/node.rs:41-42 — libfwgpu does that itself).
    pub fn process_interleaved(
        &mut self,
        input: &[f32],
        output: &mut [f32],
        num_in_channels: usize,
        num_out_channels: usize,
        frames: usize,
        stream_time_secs: f64,
        stream_status: firewheel_core::node::StreamStatus,
    ) -> firewheel_graph::processor::FirewheelProcessorStatus {
        // a safe API must not let a short slice through to C: real checks, in release builds too
        assert!(output.len() >= frames * num_out_channels, "output slice shorter than frames * num_out_channels");
        assert!(input.is_empty() || input.len() >= frames * num_in_channels, "input slice shorter than frames * num_in_channels");
        assert!(num_in_channels <= 64 && num_out_channels <= 64);
        let rc = unsafe {
            ffi::fwgpu_process_interleaved(
                self.cx.as_ptr(),
                if input.is_empty() { std::ptr::null() } else { input.as_ptr() },
                output.as_mut_ptr(),
                num_in_channels as u32,
                num_out_channels as u32,
                frames as u64,
                stream_time_secs,
                stream_status.bits(),
            )
        };
        if rc < 0 {
            log::error!("fwgpu_process_interleaved failed: {}", rc);
        }
        firewheel_graph::processor::FirewheelProcessorStatus::Ok
    }

    /// The same call split in two, for hosts that render ahead (`fwgpu_process_interleaved_begin` / `_end`, include/fwgpu.h): `begin`
    /// returns a ticket as soon as the call's launches are queued; `end` hands that ticket's frames to `output`.  Begin call n + 1 before
    /// ending call n and the host's share of call n overlaps the rendering of call n + 1.  At most two tickets in flight, ended in order
    /// — which the type enforces: a `Ticket` is consumed by `end`, and the processor hands out the next one only while fewer than two
    /// are alive (the C side checks the order again).
    pub fn process_interleaved_begin(
        &mut self,
        input: &[f32],
        num_in_channels: usize,
        num_out_channels: usize,
        frames: usize,
        stream_time_secs: f64,
        stream_status: firewheel_core::node::StreamStatus,
    ) -> Result<Ticket, GpuError> {
        assert!(input.is_empty() || input.len() >= frames * num_in_channels, "input slice shorter than frames * num_in_channels");
        assert!(num_in_channels <= 64 && num_out_channels <= 64);
        let t = unsafe {
            ffi::fwgpu_process_interleaved_begin(
                self.cx.as_ptr(),
                if input.is_empty() { std::ptr::null() } else { input.as_ptr() },
                num_in_channels as u32,
                num_out_channels as u32,
                frames as u64,
                stream_time_secs,
                stream_status.bits(),
            )
        };
        self.cx.check(t).map(|id| Ticket { id, floats: frames * num_out_channels, cx: Arc::clone(&self.cx), ended: false, _audio_thread: std::marker::PhantomData })
    }

    /// ... its other half.  `output` is filled on every return for the oldest ticket in flight (zeros when the device failed); a
    /// ticket that is not the oldest one is refused (`FWGPU_ERR_INVALID`), `output` untouched, and — being consumed here — cancelled
    /// together with the older ones by its `Drop`.
    pub fn process_interleaved_end(&mut self, mut ticket: Ticket, output: &mut [f32]) -> Result<(), GpuError> {
        assert!(output.len() >= ticket.floats, "output slice shorter than the ticket's frames * num_out_channels");
        let rc = unsafe { ffi::fwgpu_process_interleaved_end(self.cx.as_ptr(), ticket.id, output.as_mut_ptr()) };
        // (ended = the C side released the slot: success, or a device failure on the oldest ticket — not a refused ticket)
        ticket.ended = rc != ffi::FWGPU_ERR_INVALID;
        self.cx.check(rc as i64).map(|_| ())
    }
}
