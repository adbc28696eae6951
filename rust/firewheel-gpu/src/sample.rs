//! Sample resources on the device.  The reference moves `S: SampleResource` values (usually `Arc`s) into the sampler's
//! processor and gets the old one back through `ProcessorToNodeMsg::ReturnSample` (basic_nodes/sampler.rs:339-343,
//! 563-571), dropping it on the control thread in `SamplerNode::update`.  Here a sample is uploaded once
//! (`fwgpu_sample_create`) and named by id; [`GpuSample`] is the handle the `Arc` owners share, and the HBM goes when
//! the last handle is gone AND the device has let go of the id (`fwgpu_sample_retired`).
use std::sync::{Arc, Mutex};

use firewheel_core::sample_resource::{
    InterleavedResourceF32, InterleavedResourceI16, InterleavedResourceU16, SampleResource,
};

use crate::{ffi, GpuContext, GpuError};

struct Inner {
    cx: Arc<GpuContext>,
    id: i32,
    graveyard: Arc<Graveyard>,
}
impl Drop for Inner {
    fn drop(&mut self) {
        // the last owner is gone; the device may still be reading (a SetSample that replaces it is queued or in flight):
        // park the id, `Graveyard::collect` destroys it once fwgpu_sample_retired says so
        self.graveyard.ids.lock().unwrap().push(self.id);
    }
}

/// Ids waiting for the device to finish with them.  One per context; `collect` is called from the host's update loop
/// (where `FirewheelGraphCtx::update` runs, graph/context.rs:93) — never from the audio thread.
#[derive(Default)]
pub struct Graveyard {
    ids: Mutex<Vec<i32>>,
}
impl Graveyard {
    pub fn collect(&self, cx: &GpuContext) {
        let mut ids = self.ids.lock().unwrap();
        ids.retain(|&id| {
            let _g = cx.control();
            let retired = unsafe { ffi::fwgpu_sample_retired(cx.as_ptr(), id) } == 1;
            if retired {
                unsafe { ffi::fwgpu_sample_destroy(cx.as_ptr(), id) };
            }
            !retired
        });
    }
}

#[derive(Clone)]
pub struct GpuSample(Arc<Inner>);

impl GpuSample {
    pub fn id(&self) -> i32 {
        self.0.id
    }

    fn create(
        cx: &Arc<GpuContext>,
        graveyard: &Arc<Graveyard>,
        format: i32,
        channels: u32,
        frames: u64,
        data: *const std::ffi::c_void,
    ) -> Result<Self, GpuError> {
        let _g = cx.control();
        let id = cx.check(unsafe { ffi::fwgpu_sample_create(cx.as_ptr(), format, channels, frames, data) } as i64)? as i32;
        drop(_g);
        Ok(Self(Arc::new(Inner { cx: Arc::clone(cx), id, graveyard: Arc::clone(graveyard) })))
    }

    /// `InterleavedResourceI16` (core/sample_resource.rs:28-60)
    pub fn from_interleaved_i16(cx: &Arc<GpuContext>, g: &Arc<Graveyard>, r: &InterleavedResourceI16) -> Result<Self, GpuError> {
        Self::create(cx, g, ffi::FWGPU_INTERLEAVED_I16, r.channels.get() as u32, r.len_frames(), r.data.as_ptr().cast())
    }
    /// `InterleavedResourceU16` (core/sample_resource.rs:89-121)
    pub fn from_interleaved_u16(cx: &Arc<GpuContext>, g: &Arc<Graveyard>, r: &InterleavedResourceU16) -> Result<Self, GpuError> {
        Self::create(cx, g, ffi::FWGPU_INTERLEAVED_U16, r.channels.get() as u32, r.len_frames(), r.data.as_ptr().cast())
    }
    /// `InterleavedResourceF32` (core/sample_resource.rs:150-182)
    pub fn from_interleaved_f32(cx: &Arc<GpuContext>, g: &Arc<Graveyard>, r: &InterleavedResourceF32) -> Result<Self, GpuError> {
        Self::create(cx, g, ffi::FWGPU_INTERLEAVED_F32, r.channels.get() as u32, r.len_frames(), r.data.as_ptr().cast())
    }
    /// `Vec<Vec<f32>>` (core/sample_resource.rs:283-335): the planes are concatenated for the upload
    pub fn from_planar_f32(cx: &Arc<GpuContext>, g: &Arc<Graveyard>, planes: &[Vec<f32>]) -> Result<Self, GpuError> {
        let frames = planes.first().map(|p| p.len()).unwrap_or(0);
        let mut flat = Vec::with_capacity(frames * planes.len());
        for p in planes {
            debug_assert_eq!(p.len(), frames);
            flat.extend_from_slice(p);
        }
        Self::create(cx, g, ffi::FWGPU_PLANAR_F32, planes.len() as u32, frames as u64, flat.as_ptr().cast())
    }
    /// `Vec<Vec<i16>>` (core/sample_resource.rs:211-245)
    pub fn from_planar_i16(cx: &Arc<GpuContext>, g: &Arc<Graveyard>, planes: &[Vec<i16>]) -> Result<Self, GpuError> {
        let frames = planes.first().map(|p| p.len()).unwrap_or(0);
        let flat: Vec<i16> = planes.iter().flat_map(|p| p.iter().copied()).collect();
        Self::create(cx, g, ffi::FWGPU_PLANAR_I16, planes.len() as u32, frames as u64, flat.as_ptr().cast())
    }
    /// `Vec<Vec<u16>>` (core/sample_resource.rs:247-281)
    pub fn from_planar_u16(cx: &Arc<GpuContext>, g: &Arc<Graveyard>, planes: &[Vec<u16>]) -> Result<Self, GpuError> {
        let frames = planes.first().map(|p| p.len()).unwrap_or(0);
        let flat: Vec<u16> = planes.iter().flat_map(|p| p.iter().copied()).collect();
        Self::create(cx, g, ffi::FWGPU_PLANAR_U16, planes.len() as u32, frames as u64, flat.as_ptr().cast())
    }
}
