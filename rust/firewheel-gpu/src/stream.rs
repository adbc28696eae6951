//! Headless stream: `DataCallback::callback` of firewheel-cpal (src/lib.rs:378-449) without a device — the "dummy
//! backend" the reference leaves as `todo!()` (lib.rs:150,168,222).  The caller plays cpal: one `callback` per device
//! period with the instant of that callback on its own clock; stream time and OUTPUT_UNDERFLOW detection are the
//! reference's, evaluated inside libfwgpu (`fwgpu_stream_callback`).
use std::ptr::NonNull;
use std::sync::Arc;

use firewheel_core::node::StreamStatus;

use crate::{ffi, GpuContext, GpuError};

pub struct HeadlessStream {
    cx: Arc<GpuContext>,
    raw: NonNull<ffi::fwgpu_stream>,
    num_out_channels: usize,
}
unsafe impl Send for HeadlessStream {}

impl HeadlessStream {
    pub fn open(cx: Arc<GpuContext>, num_in_channels: u32, num_out_channels: u32) -> Result<Self, GpuError> {
        if num_out_channels == 0 || num_out_channels > 64 || num_in_channels > 64 {
            // (callback / run divide by the channel count: refuse here instead of panicking there — ADVICE r2)
            return Err(GpuError { code: ffi::FWGPU_ERR_INVALID, message: "a stream has 1..=64 output channels".into() });
        }
        let raw = unsafe { ffi::fwgpu_stream_open(cx.as_ptr(), num_in_channels, num_out_channels) };
        match NonNull::new(raw) {
            Some(raw) => Ok(Self { cx, raw, num_out_channels: num_out_channels as usize }),
            None => Err(GpuError { code: ffi::FWGPU_ERR_INVALID, message: "fwgpu_stream_open failed".into() }),
        }
    }

    /// One device period.  `output.len() / num_out_channels` frames are rendered; returns the `StreamStatus` the block
    /// was processed with (OUTPUT_UNDERFLOW when this callback came later than the previous one predicted).
    pub fn callback(&mut self, output: &mut [f32], callback_instant_secs: f64) -> Result<StreamStatus, GpuError> {
        let frames = output.len() / self.num_out_channels;
        let rc = unsafe {
            ffi::fwgpu_stream_callback(self.raw.as_ptr(), output.as_mut_ptr(), frames as u64, callback_instant_secs)
        };
        self.cx.check(rc as i64).map(|bits| StreamStatus::from_bits_truncate(bits as u32))
    }

    /// The backend thread's loop without a device: `n_callbacks` periods back to back, callback `i` at
    /// `first_instant_secs + i * period` (a clock that is never late).  `output` holds the last block afterwards.
    /// Returns the wall time of the loop — `elapsed / n_callbacks` is what one callback costs the audio thread.
    pub fn run(&mut self, output: &mut [f32], n_callbacks: u32, first_instant_secs: f64) -> Result<std::time::Duration, GpuError> {
        let frames = output.len() / self.num_out_channels;
        let mut secs = 0f64;
        let rc = unsafe {
            ffi::fwgpu_stream_run(self.raw.as_ptr(), output.as_mut_ptr(), frames as u64, n_callbacks, first_instant_secs, &mut secs)
        };
        self.cx.check(rc as i64).map(|_| std::time::Duration::from_secs_f64(secs))
    }

    /// (callbacks so far, underflows so far, stream time of the last callback)
    pub fn stats(&self) -> (u64, u64, f64) {
        let (mut c, mut u, mut t) = (0u64, 0u64, 0f64);
        unsafe { ffi::fwgpu_stream_stats(self.raw.as_ptr(), &mut c, &mut u, &mut t) };
        (c, u, t)
    }
}
impl Drop for HeadlessStream {
    fn drop(&mut self) {
        unsafe { ffi::fwgpu_stream_close(self.raw.as_ptr()) }
    }
}
