//! Custom nodes inside a device-resident graph (level B2 with the reference's open node set).
//!
//! Firewheel's schedule loop calls ANY `dyn AudioNodeProcessor` (firewheel-graph/src/processor.rs:226-247).  A node libfwgpu
//! has no kernel for stays what it is — a Rust processor — and is registered as a `FWGPU_HOST_NODE`: the device plan is cut
//! at its level, its input buffers (and only those) come to pinned host memory, `process()` runs on the audio thread once per
//! block in block order, its outputs go back (include/fwgpu.h, "custom nodes inside a device-resident graph").
use std::os::raw::c_void;
use std::sync::Arc;

use arrayvec::ArrayVec;
use firewheel_core::node::{AudioNodeProcessor, ProcInfo, StreamStatus};
use firewheel_core::SilenceMask;

use crate::{ffi, GpuContext, GpuError};

/// Owner of a custom processor inside the device graph.  Dropping it removes the node (if it still exists) and hands the processor
/// to the context's graveyard, which frees it only once a plan WITHOUT the node is the active one: the grave carries the number of
/// compiles that had succeeded when the node left the graph (both read and bumped under the control lock), a compile AFTER that
/// one builds a plan without the node, `fwgpu_plan_pending() == 0` says that plan has been adopted — until
/// then the audio thread may still be inside, or about to enter, the trampoline with the raw `user` pointer.  That is Firewheel's
/// own rule (a removed node's processor is dropped when the old schedule comes back through the ring, processor.rs:182-188), made
/// part of the type instead of the documentation: dropping the handle at any time, from any thread, with the stream running, is
/// sound.
///
/// `drop` does NOT recompile: the reference only ever compiles in `update()` (graph/context.rs:93-137), and a handle
/// dropped in the middle of a batch of edits must not publish the half-edited graph to the audio thread, swallow its compile
/// error, or take the control mutex a second time.  The plan that still calls the node keeps running until the host updates.
pub struct HostNodeHandle {
    pub node: i64,
    state: Option<Box<HostState>>,
    cx: Arc<GpuContext>,
}
impl Drop for HostNodeHandle {
    fn drop(&mut self) {
        // ONE critical section for "the graph stops naming the node" and "which compile was the last one before that": a compile
        // on another thread is either entirely before (its plan may call the node: the grave waits for a later one) or entirely
        // after.  Err from the removal = the host removed it already (remove_node): the stamp is then merely later than needed.
        let _g = self.cx.control();
        let _ = unsafe { ffi::fwgpu_remove_node(self.cx.as_ptr(), self.node) };
        if let Some(st) = self.state.take() {
            let gen = self.cx.compile_gen.load(std::sync::atomic::Ordering::Acquire);
            self.cx.bury(st, gen);
        }
    }
}
/// what the C side's `user` pointer names: the processor and where the graph's ONE global user context lives.  The reference's
/// processor owns one `Box<dyn Any + Send>` per graph and lends it to every node's `process` (processor.rs:21,224,240;
/// core/node.rs:117-118): custom nodes may share state through it.  Here it belongs to the `GpuContext` (`set_user_cx`, before the
/// stream starts); only the audio thread ever dereferences the pointer, one host node at a time, exactly as the reference's
/// schedule loop does.
pub(crate) struct HostState {
    processor: Box<dyn AudioNodeProcessor>,
    user_cx: *mut Box<dyn std::any::Any + Send>,
}
unsafe impl Send for HostNodeHandle {}
unsafe impl Send for HostState {}

/// A processor whose node is gone from the graph but which a plan may still call: (state, compiles that had succeeded at its removal).
pub(crate) struct Grave {
    _state: Box<HostState>,
    removed_at_gen: u64,
}
impl GpuContext {
    /// (called with the control lock held; lock order control -> graveyard everywhere)
    pub(crate) fn bury(&self, st: Box<HostState>, removed_at_gen: u64) {
        let mut g = self.graveyard.lock().unwrap_or_else(|e| e.into_inner());
        g.push(Grave { _state: st, removed_at_gen });
    }
    /// Free the processors no plan can call any more: a compile that began after their removal has succeeded (`compile_gen` is past
    /// the grave's stamp) AND no built plan is waiting for adoption — so the newest plan, which does not name them, is the one the
    /// audio thread runs.  Called after every `update` / `upload_schedule`, by `release_removed_host_nodes` (a plan built earlier may
    /// have been adopted meanwhile), and by `GpuContext::drop` (everything, as fields).  The generation is read BEFORE the pending
    /// flag: a compile that lands in between only makes this pass keep a grave one call longer.
    pub(crate) fn reap(&self) {
        let mut g = self.graveyard.lock().unwrap_or_else(|e| e.into_inner());
        if g.is_empty() {
            return;
        }
        let gen = self.compile_gen.load(std::sync::atomic::Ordering::Acquire);
        if unsafe { ffi::fwgpu_plan_pending(self.as_ptr()) } == 0 {
            g.retain(|gr| gr.removed_at_gen >= gen);
        }
    }
    /// Frees the processors of removed host nodes that no plan can call any more, without recompiling (a host that wants the memory
    /// back before its next `update`).
    pub fn release_removed_host_nodes(&self) {
        self.reap();
    }
}

unsafe extern "C" fn trampoline(
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
) {
    let st = &mut *(user as *mut HostState);
    let frames = frames as usize;
    let ins: ArrayVec<&[f32], 64> =
        (0..num_inputs as usize).map(|i| std::slice::from_raw_parts(*inputs.add(i), frames)).collect();
    let mut outs: ArrayVec<&mut [f32], 64> =
        (0..num_outputs as usize).map(|i| std::slice::from_raw_parts_mut(*outputs.add(i), frames)).collect();
    let mut out_mask = SilenceMask(*out_silence_mask);
    let info = ProcInfo {
        in_silence_mask: SilenceMask(in_silence_mask),
        out_silence_mask: &mut out_mask,
        stream_time_secs,
        stream_status: StreamStatus::from_bits_truncate(stream_status),
        cx: &mut *st.user_cx, // the graph's one context, as processor.rs:224,240 lends it
    }; // core/node.rs:94-118
    st.processor.process(frames, &ins, &mut outs, info);
    *out_silence_mask = out_mask.0;
}

impl GpuContext {
    /// Register `processor` (what `AudioNode::activate` returned for a node libfwgpu does not implement) with `num_inputs` /
    /// `num_outputs` ports.  The returned id is used like any other node id: in `connect`, or in the entry of
    /// `upload_schedule` that stands for the node.  Takes effect with the next `update` / `upload_schedule`.
    pub fn add_host_node(
        self: &Arc<Self>,
        num_inputs: u32,
        num_outputs: u32,
        processor: Box<dyn AudioNodeProcessor>,
    ) -> Result<HostNodeHandle, GpuError> {
        assert!(num_inputs <= 64 && num_outputs <= 64);
        let mut boxed = Box::new(HostState { processor, user_cx: self.user_cx_ptr() });
        let user = &mut *boxed as *mut HostState as *mut c_void;
        let _g = self.control();
        let node = self.check(unsafe {
            ffi::fwgpu_add_node(self.as_ptr(), ffi::FWGPU_HOST_NODE, num_inputs, num_outputs, std::ptr::null(), 0)
        })?;
        if let Err(e) = self.check(unsafe { ffi::fwgpu_host_node_set_process(self.as_ptr(), node, Some(trampoline), user) } as i64) {
            unsafe { ffi::fwgpu_remove_node(self.as_ptr(), node) };
            return Err(e);
        }
        Ok(HostNodeHandle { node, state: Some(boxed), cx: Arc::clone(self) })
    }
}
