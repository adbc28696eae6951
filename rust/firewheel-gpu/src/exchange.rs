//! The multi-GPU mix bus for a host without torch / RCCL: libfwgpu's one-shot exchange over peer-mapped slots
//! (include/fwgpu.h "multi-GPU mix bus"; SURVEY §8e path 2).
//!
//! Voices shard across GPUs — one process (or one `GpuContext`) per device, each with its voices' samples, state and the lower
//! levels of the sum tree — and meet only in the top-level `SumNode` over the R partial buses
//! (firewheel-graph/src/basic_nodes/sum.rs:41-136).  Each rank opens a [`BusExchange`], publishes its 128-byte handle through
//! whatever channel the host has (a file, a pipe, MPI), connects the peers' handles, and from then on calls [`BusExchange::step`]
//! once per process call: the partial bus (and its per-block silence flags) go to every rank's slot over xGMI, the R slots that
//! arrive here are added in rank order — every rank ends with the bits of the single-process graph.
use std::ptr::NonNull;
use std::sync::Arc;

use crate::{ffi, GpuContext, GpuError};

pub const HANDLE_BYTES: usize = ffi::FWGPU_EXCHANGE_HANDLE_BYTES;

pub struct BusExchange {
    cx: Arc<GpuContext>,
    raw: NonNull<ffi::fwgpu_bus_exchange>,
    pub rank: u32,
    pub world: u32,
}
unsafe impl Send for BusExchange {}

impl BusExchange {
    /// `max_floats`: the longest interleaved bus of one step (blocks x frames x channels); `max_silence_bytes`: blocks x channels.
    pub fn open(cx: Arc<GpuContext>, rank: u32, world: u32, max_floats: u64, max_silence_bytes: u32) -> Result<Self, GpuError> {
        let _g = cx.control();
        let raw = unsafe { ffi::fwgpu_bus_exchange_open(cx.as_ptr(), rank, world, max_floats, max_silence_bytes) };
        drop(_g);
        match NonNull::new(raw) {
            Some(raw) => Ok(Self { cx, raw, rank, world }),
            None => Err(cx.check(ffi::FWGPU_ERR_DEVICE as i64).unwrap_err()),
        }
    }
    pub fn export(&self) -> Result<[u8; HANDLE_BYTES], GpuError> {
        let mut h = [0u8; HANDLE_BYTES];
        self.cx.check(unsafe { ffi::fwgpu_bus_exchange_export(self.raw.as_ptr(), h.as_mut_ptr() as *mut _) } as i64)?;
        Ok(h)
    }
    pub fn connect(&mut self, peer_rank: u32, handle: &[u8; HANDLE_BYTES]) -> Result<(), GpuError> {
        self.cx
            .check(unsafe { ffi::fwgpu_bus_exchange_connect(self.raw.as_ptr(), peer_rank, handle.as_ptr() as *const _) } as i64)
            .map(|_| ())
    }
    pub fn set_timeout_ms(&mut self, ms: u32) {
        unsafe { ffi::fwgpu_bus_exchange_set_timeout_ms(self.raw.as_ptr(), ms) };
    }
    /// AUDIO side, asynchronous on the context's stream: push this rank's partial bus, wait on the device for the peers', sum in
    /// rank order into `d_out`.  All pointers are device memory (`d_partial` / `d_out` 16-byte aligned; `d_silence` as
    /// `fwgpu_process_blocks_device_flags` wrote it, or null).
    ///
    /// # Safety
    /// The pointers must be valid device allocations of at least `n_floats` floats / `n_blocks * n_channels` bytes that stay
    /// alive until the context's stream has passed this step.
    pub unsafe fn step(
        &mut self,
        d_partial: *const f32,
        d_silence: *const u8,
        d_out: *mut f32,
        d_out_silence: *mut u8,
        n_floats: u64,
        n_blocks: u32,
        frames_per_block: u32,
        n_channels: u32,
    ) -> Result<(), GpuError> {
        self.cx
            .check(ffi::fwgpu_bus_exchange_step(
                self.raw.as_ptr(),
                d_partial,
                d_silence,
                d_out,
                d_out_silence,
                n_floats,
                n_blocks,
                frames_per_block,
                n_channels,
            ) as i64)
            .map(|_| ())
    }
    /// Control side: waits for the stream; `Err` when a peer did not arrive within the time budget (its bus was replaced by zeros).
    pub fn status(&self) -> Result<u64, GpuError> {
        let (mut steps, mut failed) = (0u64, 0u64);
        self.cx.check(unsafe { ffi::fwgpu_bus_exchange_status(self.raw.as_ptr(), &mut steps, &mut failed) } as i64)?;
        Ok(steps)
    }
    /// The longest a reduce has waited for each rank's arrival so far, in microseconds.
    pub fn wait_stats(&self, reset: bool) -> Vec<u64> {
        let mut v = vec![0u64; self.world as usize];
        unsafe { ffi::fwgpu_bus_exchange_wait_stats(self.raw.as_ptr(), v.as_mut_ptr(), self.world, reset as i32) };
        v
    }
}
impl Drop for BusExchange {
    fn drop(&mut self) {
        unsafe { ffi::fwgpu_bus_exchange_close(self.raw.as_ptr()) }
    }
}

/// The same step over RCCL (include/fwgpu.h "the mix bus over RCCL"; north_star's named path): libfwgpu `dlopen`s librccl on first
/// use.  Rank 0 makes the unique id ([`RcclComm::unique_id`]), the host carries its 128 bytes to every rank, every rank builds its
/// communicator on its context's device (collective).
pub const RCCL_ID_BYTES: usize = ffi::FWGPU_RCCL_UNIQUE_ID_BYTES;

pub struct RcclComm {
    cx: Arc<GpuContext>,
    raw: NonNull<ffi::fwgpu_rccl_comm>,
    pub rank: u32,
    pub world: u32,
}
unsafe impl Send for RcclComm {}

impl RcclComm {
    pub fn unique_id() -> Result<[u8; RCCL_ID_BYTES], GpuError> {
        let mut id = [0u8; RCCL_ID_BYTES];
        let rc = unsafe { ffi::fwgpu_rccl_unique_id(id.as_mut_ptr()) };
        if rc < 0 {
            let message = unsafe { std::ffi::CStr::from_ptr(ffi::fwgpu_rccl_last_error()) }.to_string_lossy().into_owned();
            return Err(GpuError { code: rc, message });
        }
        Ok(id)
    }
    /// Collective: every rank of `id` calls it (ncclCommInitRank).
    pub fn create(cx: Arc<GpuContext>, id: &[u8; RCCL_ID_BYTES], world: u32, rank: u32) -> Result<Self, GpuError> {
        let _g = cx.control();
        let raw = unsafe { ffi::fwgpu_rccl_comm_create(cx.as_ptr(), id.as_ptr(), world, rank) };
        drop(_g);
        match NonNull::new(raw) {
            Some(raw) => Ok(Self { cx, raw, rank, world }),
            None => Err(cx.check(ffi::FWGPU_ERR_DEVICE as i64).unwrap_err()),
        }
    }
    /// AUDIO side, asynchronous on the context's stream: ncclAllReduce(sum) in place.  Re-associates the f32 sum for more than two
    /// ranks (within 1e-6 relative of the single-process graph, not its bits).
    ///
    /// # Safety
    /// `d_bus` must be a device allocation of at least `n_floats` floats that stays alive until the stream has passed the call.
    pub unsafe fn allreduce(&mut self, d_bus: *mut f32, n_floats: u64) -> Result<(), GpuError> {
        self.cx.check(ffi::fwgpu_bus_allreduce_rccl(self.raw.as_ptr(), d_bus, n_floats) as i64).map(|_| ())
    }
    /// AUDIO side: ncclAllGather of buses and silence flags + the rank-ordered sum (sum.rs:111-133's order and silent-port rule):
    /// bit-identical to the single-process graph on every rank.
    ///
    /// # Safety
    /// As [`BusExchange::step`]: valid device pointers of the stated sizes until the stream has passed the call.
    pub unsafe fn allgather_ordered(
        &mut self,
        d_bus: *const f32,
        d_silence: *const u8,
        d_out: *mut f32,
        d_out_silence: *mut u8,
        n_floats: u64,
        frames_per_block: u32,
        n_channels: u32,
    ) -> Result<(), GpuError> {
        self.cx
            .check(ffi::fwgpu_bus_allgather_ordered(self.raw.as_ptr(), d_bus, d_silence, d_out, d_out_silence, n_floats, frames_per_block, n_channels) as i64)
            .map(|_| ())
    }
}
impl Drop for RcclComm {
    fn drop(&mut self) {
        let _g = self.cx.control();
        unsafe { ffi::fwgpu_rccl_comm_destroy(self.raw.as_ptr()) };
    }
}
