//! Level B1: Firewheel's built-in nodes (firewheel-graph/src/basic_nodes/*.rs) — and the north-star nodes the reference
//! lists as TODO (README.md:14-19) — as `AudioNode`s whose processor half runs on the device.
//!
//! Each wrapper has the reference node's constructor and control-half setters.  `activate()` registers the node with
//! the shared [`GpuContext`] (`fwgpu_add_node` + `fwgpu_update`) — on Firewheel's control thread, WHILE the audio thread may
//! be inside `process()` of nodes that are already active (graph.rs:586-612): libfwgpu builds the new plan off to the side
//! and the audio thread's next `fwgpu_node_process` adopts it, and that call looks the node up in the ACTIVE plan, never in
//! the graph being edited — and returns a [`GpuNodeProcessor`], whose `process()`
//! is one `fwgpu_node_process` call: the same device function the whole-graph executor runs for that node, on the
//! caller's buffers, with `in_silence_mask` / `out_silence_mask` honoured (core/node.rs:94-118).  Setters become
//! messages (`fwgpu_node_set_param`, `fwgpu_sampler_*`) with `at_block = 0`: a per-block caller gets the reference's
//! "polled at the top of process()" semantics exactly (volume.rs:92, sampler.rs:331).
use std::error::Error;
use std::sync::Arc;

use arrayvec::ArrayVec;
use firewheel_core::node::{AudioNode, AudioNodeInfo, AudioNodeProcessor, ProcInfo};
use firewheel_core::SilenceMask;

use crate::sample::GpuSample;
use crate::{ffi, GpuContext, GpuError};

/// `AudioNodeProcessor` of every GPU node: forwards the block to `fwgpu_node_process`.
pub struct GpuNodeProcessor {
    cx: Arc<GpuContext>,
    node: i64,
}
impl AudioNodeProcessor for GpuNodeProcessor {
    fn process(&mut self, frames: usize, inputs: &[&[f32]], outputs: &mut [&mut [f32]], proc_info: ProcInfo) {
        let ins: ArrayVec<*const f32, 64> = inputs.iter().map(|s| s.as_ptr()).collect();
        let outs: ArrayVec<*mut f32, 64> = outputs.iter_mut().map(|s| s.as_mut_ptr()).collect();
        let mut out_mask: u64 = proc_info.out_silence_mask.0; // 0 on entry (graph/processor.rs:233)
        let rc = unsafe {
            ffi::fwgpu_node_process(
                self.cx.as_ptr(),
                self.node,
                frames as u64,
                ins.as_ptr(),
                ins.len() as u32,
                outs.as_ptr(),
                outs.len() as u32,
                proc_info.in_silence_mask.0,
                &mut out_mask,
                proc_info.stream_time_secs,
                proc_info.stream_status.bits(),
            )
        };
        if rc < 0 {
            // "all output buffers MUST be filled" (core/node.rs:41-42)
            firewheel_core::util::clear_all_outputs(frames, outputs, proc_info.out_silence_mask);
            return;
        }
        *proc_info.out_silence_mask = SilenceMask(out_mask);
    }
}

/// What every wrapper shares: the context, the device node id once activated.
struct Binding {
    cx: Arc<GpuContext>,
    node: Option<i64>,
}
impl Binding {
    fn new(cx: &Arc<GpuContext>) -> Self {
        Self { cx: Arc::clone(cx), node: None }
    }
    fn activate(
        &mut self,
        kind: i32,
        num_inputs: usize,
        num_outputs: usize,
        params: &[f32],
    ) -> Result<Box<dyn AudioNodeProcessor>, Box<dyn Error>> {
        let node = self.cx.add_node(kind, num_inputs as u32, num_outputs as u32, params)?;
        // activation checks (volume.rs:63-65, sum.rs:27-29, hard_clip.rs:37-39) run in fwgpu_update and come back as
        // FWGPU_ERR_NODE_ACTIVATION_FAILED + the reference's message
        if let Err(e) = self.cx.update() {
            let _ = self.cx.remove_node(node);
            return Err(Box::new(e));
        }
        self.node = Some(node);
        Ok(Box::new(GpuNodeProcessor { cx: Arc::clone(&self.cx), node }))
    }
    fn deactivate(&mut self) {
        if let Some(node) = self.node.take() {
            let _ = self.cx.remove_node(node);
        }
    }
    fn set_param(&self, param: i32, value: f32) -> Result<(), GpuError> {
        match self.node {
            Some(node) => {
                let _g = self.cx.control();
                self.cx.check(unsafe { ffi::fwgpu_node_set_param(self.cx.as_ptr(), node, param, value, 0) } as i64).map(|_| ())
            }
            None => Ok(()), // not activated yet: the constructor argument carries the value (activate reads it)
        }
    }
}

fn io(min_in: u32, max_in: u32, min_out: u32, max_out: u32, updates: bool) -> AudioNodeInfo {
    AudioNodeInfo {
        num_min_supported_inputs: min_in,
        num_max_supported_inputs: max_in,
        num_min_supported_outputs: min_out,
        num_max_supported_outputs: max_out,
        updates,
    }
}

// ------------------------------------------------------------------------------------------------ VolumeNode
/// basic_nodes/volume.rs:8-77
pub struct GpuVolumeNode {
    b: Binding,
    percent_volume: f32,
}
impl GpuVolumeNode {
    pub fn new(cx: &Arc<GpuContext>, percent_volume: f32) -> Self {
        Self { b: Binding::new(cx), percent_volume: percent_volume.max(0.0) }
    }
    pub fn percent_volume(&self) -> f32 {
        self.percent_volume
    }
    /// volume.rs:28-34 — the `Arc<AtomicF32>` store becomes a message in libfwgpu's lock-free ring
    pub fn set_percent_volume(&mut self, percent_volume: f32) {
        self.percent_volume = percent_volume.max(0.0);
        let _ = self.b.set_param(0, percent_volume);
    }
}
impl AudioNode for GpuVolumeNode {
    fn debug_name(&self) -> &'static str {
        "volume"
    }
    fn info(&self) -> AudioNodeInfo {
        io(1, 64, 1, 64, false)
    }
    fn activate(&mut self, _sr: u32, _mbf: usize, num_inputs: usize, num_outputs: usize) -> Result<Box<dyn AudioNodeProcessor>, Box<dyn Error>> {
        self.b.activate(ffi::FWGPU_VOLUME, num_inputs, num_outputs, &[self.percent_volume])
    }
    fn deactivate(&mut self, _p: Option<Box<dyn AudioNodeProcessor>>) {
        self.b.deactivate()
    }
}

// ------------------------------------------------------------------------------------------------ SumNode
/// basic_nodes/sum.rs:3-35
pub struct GpuSumNode {
    b: Binding,
}
impl GpuSumNode {
    pub fn new(cx: &Arc<GpuContext>) -> Self {
        Self { b: Binding::new(cx) }
    }
}
impl AudioNode for GpuSumNode {
    fn debug_name(&self) -> &'static str {
        "sum"
    }
    fn info(&self) -> AudioNodeInfo {
        io(1, 64, 1, 64, false)
    }
    fn activate(&mut self, _sr: u32, _mbf: usize, num_inputs: usize, num_outputs: usize) -> Result<Box<dyn AudioNodeProcessor>, Box<dyn Error>> {
        self.b.activate(ffi::FWGPU_SUM, num_inputs, num_outputs, &[])
    }
    fn deactivate(&mut self, _p: Option<Box<dyn AudioNodeProcessor>>) {
        self.b.deactivate()
    }
}

// ------------------------------------------------------------------------------------------------ SamplerNode
/// basic_nodes/sampler.rs:16-19
#[derive(Debug, Clone, PartialEq)]
pub enum LoopRange {
    Full,
    RangeSecs(std::ops::Range<f64>),
}

/// basic_nodes/sampler.rs:46-233.  `set_sample` takes a [`GpuSample`] (the device-resident counterpart of
/// `S: SampleResource`); the sample the processor hands back (`ReturnSample`, :339-343) is dropped in `update()`.
pub struct GpuSamplerNode {
    b: Binding,
    percent_volume: f32,
    playing: bool,
    held: Vec<GpuSample>, // handles the device may still be reading: released when it hands the id back
}
impl GpuSamplerNode {
    pub fn new(cx: &Arc<GpuContext>, percent_volume: f32) -> Self {
        Self { b: Binding::new(cx), percent_volume: percent_volume.max(0.0), playing: false, held: Vec::new() }
    }
    fn msg(&self, f: impl FnOnce(*mut ffi::fwgpu_ctx, i64) -> i32) -> Result<(), ()> {
        let node = self.b.node.ok_or(())?; // not activated: sampler.rs:68-70 returns Err(())
        let _g = self.b.cx.control();
        if f(self.b.cx.as_ptr(), node) < 0 {
            return Err(()); // ring full (FWGPU_ERR_QUEUE_FULL): sampler.rs:72-78 `.map_err(|_| ())`
        }
        Ok(())
    }
    /// sampler.rs:67-79
    pub fn set_sample(&mut self, sample: GpuSample, stop_playback: bool) -> Result<(), ()> {
        let id = sample.id();
        self.msg(|c, n| unsafe { ffi::fwgpu_sampler_set_sample(c, n, id, stop_playback as i32, 0) })?;
        self.held.push(sample);
        if stop_playback {
            self.playing = false;
        }
        Ok(())
    }
    /// sampler.rs:82-97
    pub fn play(&mut self) -> Result<(), ()> {
        self.msg(|c, n| unsafe { ffi::fwgpu_sampler_play(c, n, 0) })?;
        self.playing = true;
        Ok(())
    }
    /// sampler.rs:100-115
    pub fn pause(&mut self) -> Result<(), ()> {
        self.msg(|c, n| unsafe { ffi::fwgpu_sampler_pause(c, n, 0) })?;
        self.playing = false;
        Ok(())
    }
    /// sampler.rs:118-133
    pub fn stop(&mut self) -> Result<(), ()> {
        self.msg(|c, n| unsafe { ffi::fwgpu_sampler_stop(c, n, 0) })?;
        self.playing = false;
        Ok(())
    }
    /// sampler.rs:136-147
    pub fn set_playhead(&mut self, playhead_secs: f64) -> Result<(), ()> {
        self.msg(|c, n| unsafe { ffi::fwgpu_sampler_set_playhead_secs(c, n, playhead_secs, 0) })
    }
    /// sampler.rs:150-161
    pub fn set_loop_range(&mut self, loop_range: Option<LoopRange>) -> Result<(), ()> {
        let (mode, s, e) = match loop_range {
            None => (0, 0.0, 0.0),
            Some(LoopRange::Full) => (1, 0.0, 0.0),
            Some(LoopRange::RangeSecs(r)) => (2, r.start, r.end),
        };
        self.msg(|c, n| unsafe { ffi::fwgpu_sampler_set_loop_range(c, n, mode, s, e, 0) })
    }
    pub fn is_playing(&self) -> bool {
        self.playing
    }
    pub fn percent_volume(&self) -> f32 {
        self.percent_volume
    }
    /// sampler.rs:171-177
    pub fn set_percent_volume(&mut self, percent_volume: f32) {
        self.percent_volume = percent_volume.max(0.0);
        let _ = self.b.set_param(0, percent_volume);
    }
}
impl AudioNode for GpuSamplerNode {
    fn debug_name(&self) -> &'static str {
        "sampler"
    }
    fn info(&self) -> AudioNodeInfo {
        io(0, 0, 1, 64, true) // sampler.rs:188-195: updates = true
    }
    fn activate(&mut self, _sr: u32, _mbf: usize, num_inputs: usize, num_outputs: usize) -> Result<Box<dyn AudioNodeProcessor>, Box<dyn Error>> {
        self.b.activate(ffi::FWGPU_SAMPLER, num_inputs, num_outputs, &[self.percent_volume])
    }
    fn deactivate(&mut self, _p: Option<Box<dyn AudioNodeProcessor>>) {
        self.b.deactivate();
        self.held.clear(); // the ids go to the graveyard; they are destroyed once fwgpu_sample_retired agrees
    }
    /// sampler.rs:222-231: drain `ReturnSample`.  The ring is per context in libfwgpu, so the owner of the context
    /// usually polls once and routes by node id; a single-sampler host can let the node do it.
    fn update(&mut self) {
        let Some(node) = self.b.node else { return };
        let mut nodes = [0i64; 16];
        let mut samples = [0i32; 16];
        let _g = self.b.cx.control();
        let n = unsafe { ffi::fwgpu_poll_returned_samples(self.b.cx.as_ptr(), nodes.as_mut_ptr(), samples.as_mut_ptr(), 16) };
        for i in 0..n.max(0) as usize {
            if nodes[i] == node {
                if let Some(pos) = self.held.iter().position(|s| s.id() == samples[i]) {
                    self.held.swap_remove(pos); // drop on the control thread, like `ReturnSample(_smp) => {}`
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ BeepTestNode
/// basic_nodes/beep_test.rs:8-62
pub struct GpuBeepTestNode {
    b: Binding,
    freq_hz: f32,
    gain_db: f32,
    enabled: bool,
}
impl GpuBeepTestNode {
    pub fn new(cx: &Arc<GpuContext>, freq_hz: f32, gain_db: f32, enabled: bool) -> Self {
        Self { b: Binding::new(cx), freq_hz, gain_db, enabled }
    }
    pub fn enabled(&self) -> bool {
        self.enabled
    }
    /// beep_test.rs:30-32
    pub fn set_enabled(&mut self, enabled: bool) {
        self.enabled = enabled;
        let _ = self.b.set_param(0, if enabled { 1.0 } else { 0.0 });
    }
}
impl AudioNode for GpuBeepTestNode {
    fn debug_name(&self) -> &'static str {
        "beep_test"
    }
    fn info(&self) -> AudioNodeInfo {
        io(0, 0, 1, 64, false)
    }
    fn activate(&mut self, _sr: u32, _mbf: usize, num_inputs: usize, num_outputs: usize) -> Result<Box<dyn AudioNodeProcessor>, Box<dyn Error>> {
        // the clamps of beep_test.rs:16-17 are applied by fwgpu_add_node
        self.b.activate(ffi::FWGPU_BEEP_TEST, num_inputs, num_outputs, &[self.freq_hz, self.gain_db, self.enabled as i32 as f32])
    }
    fn deactivate(&mut self, _p: Option<Box<dyn AudioNodeProcessor>>) {
        self.b.deactivate()
    }
}

// ------------------------------------------------------------------------------------------------ parameterless / one-shot nodes
macro_rules! simple_node {
    ($name:ident, $debug:expr, $kind:expr, $info:expr, $doc:expr, [$($field:ident : $ty:ty),*]) => {
        #[doc = $doc]
        pub struct $name {
            b: Binding,
            $(pub $field: $ty,)*
        }
        impl $name {
            pub fn new(cx: &Arc<GpuContext> $(, $field: $ty)*) -> Self {
                Self { b: Binding::new(cx) $(, $field)* }
            }
        }
        impl AudioNode for $name {
            fn debug_name(&self) -> &'static str {
                $debug
            }
            fn info(&self) -> AudioNodeInfo {
                $info
            }
            fn activate(&mut self, _sr: u32, _mbf: usize, num_inputs: usize, num_outputs: usize) -> Result<Box<dyn AudioNodeProcessor>, Box<dyn Error>> {
                self.b.activate($kind, num_inputs, num_outputs, &[$(self.$field as f32),*])
            }
            fn deactivate(&mut self, _p: Option<Box<dyn AudioNodeProcessor>>) {
                self.b.deactivate()
            }
        }
    };
}
simple_node!(GpuHardClipNode, "hard_clip", ffi::FWGPU_HARD_CLIP, io(1, 64, 1, 64, false), "basic_nodes/hard_clip.rs:3-41", [threshold_db: f32]);
simple_node!(GpuMonoToStereoNode, "mono_to_stereo", ffi::FWGPU_MONO_TO_STEREO, io(1, 1, 2, 2, false), "basic_nodes/mono_to_stereo.rs", []);
simple_node!(GpuStereoToMonoNode, "stereo_to_mono", ffi::FWGPU_STEREO_TO_MONO, io(2, 2, 1, 1, false), "basic_nodes/stereo_to_mono.rs", []);
simple_node!(GpuDummyNode, "dummy", ffi::FWGPU_DUMMY, io(0, 64, 0, 64, false), "basic_nodes/dummy.rs", []);

// ------------------------------------------------------------------------------------------------ north-star nodes (README.md:14-19; SPEC in the fwgpu repository's DESIGN.md §6)
macro_rules! param_node {
    ($name:ident, $debug:expr, $kind:expr, $info:expr, $doc:expr, [$($field:ident),*], {$($setter:ident => ($param:expr, $sfield:ident)),*}) => {
        #[doc = $doc]
        pub struct $name {
            b: Binding,
            $($field: f32,)*
        }
        impl $name {
            pub fn new(cx: &Arc<GpuContext> $(, $field: f32)*) -> Self {
                Self { b: Binding::new(cx) $(, $field)* }
            }
            $(pub fn $setter(&mut self, value: f32) {
                self.$sfield = value;
                let _ = self.b.set_param($param, value);
            })*
        }
        impl AudioNode for $name {
            fn debug_name(&self) -> &'static str {
                $debug
            }
            fn info(&self) -> AudioNodeInfo {
                $info
            }
            fn activate(&mut self, _sr: u32, _mbf: usize, num_inputs: usize, num_outputs: usize) -> Result<Box<dyn AudioNodeProcessor>, Box<dyn Error>> {
                self.b.activate($kind, num_inputs, num_outputs, &[$(self.$field),*])
            }
            fn deactivate(&mut self, _p: Option<Box<dyn AudioNodeProcessor>>) {
                self.b.deactivate()
            }
        }
    };
}
param_node!(GpuStereoPanNode, "stereo_pan", ffi::FWGPU_STEREO_PAN, io(2, 2, 2, 2, false),
            "equal-power stereo pan, pan in [-1, 1], one ParamSmoother per channel", [pan], {set_pan => (0, pan)});
param_node!(GpuStereoWidthNode, "stereo_width", ffi::FWGPU_STEREO_WIDTH, io(2, 2, 2, 2, false),
            "mid/side width, one smoothed parameter (0 = mono, 1 = unchanged)", [width], {set_width => (0, width)});
param_node!(GpuBiquadNode, "biquad", ffi::FWGPU_BIQUAD, io(1, 64, 1, 64, false),
            "RBJ biquad (filter_type 0 = low-pass, 1 = high-pass, 2 = band-pass), Direct Form I in f32",
            [filter_type, cutoff_hz, q], {set_cutoff_hz => (1, cutoff_hz), set_q => (2, q)});
param_node!(GpuDelayNode, "delay", ffi::FWGPU_DELAY, io(1, 64, 1, 64, false),
            "integer-sample delay line with feedback and dry/wet mix; the delay time is fixed at construction",
            [delay_secs, feedback, mix], {set_feedback => (1, feedback), set_mix => (2, mix)});
param_node!(GpuSpatialNode, "spatial", ffi::FWGPU_SPATIAL, io(1, 2, 2, 2, false),
            "3D spatialiser: inverse-distance gain, equal-power pan from the direction cosine, per-ear delay (listener at the origin, -z forward)",
            [x, y, z], {set_x => (0, x), set_y => (1, y), set_z => (2, z)});

/// FIR / convolution reverb: the impulse response is a [`GpuSample`] named at construction (kept alive by the node).
pub struct GpuFirReverbNode {
    b: Binding,
    ir: GpuSample,
}
impl GpuFirReverbNode {
    pub fn new(cx: &Arc<GpuContext>, impulse_response: GpuSample) -> Self {
        Self { b: Binding::new(cx), ir: impulse_response }
    }
}
impl AudioNode for GpuFirReverbNode {
    fn debug_name(&self) -> &'static str {
        "fir_reverb"
    }
    fn info(&self) -> AudioNodeInfo {
        io(1, 64, 1, 64, false)
    }
    fn activate(&mut self, _sr: u32, _mbf: usize, num_inputs: usize, num_outputs: usize) -> Result<Box<dyn AudioNodeProcessor>, Box<dyn Error>> {
        // FIR banks run at graph level (one MFMA GEMM per schedule level): use this node under GpuProcessor (B2);
        // fwgpu_node_process refuses it
        self.b.activate(ffi::FWGPU_FIR, num_inputs, num_outputs, &[self.ir.id() as f32])
    }
    fn deactivate(&mut self, _p: Option<Box<dyn AudioNodeProcessor>>) {
        self.b.deactivate()
    }
}

/// Polyphase resampling source (varispeed / rate conversion of a sample): 32.32 fixed-point position, 32 x 16 Kaiser bank.
pub struct GpuResamplerNode {
    b: Binding,
    source: GpuSample,
    ratio: f32,
    looping: bool,
    playing: bool,
}
impl GpuResamplerNode {
    pub fn new(cx: &Arc<GpuContext>, source: GpuSample, ratio: f32, looping: bool, playing: bool) -> Self {
        Self { b: Binding::new(cx), source, ratio, looping, playing }
    }
    pub fn set_ratio(&mut self, ratio: f32) {
        self.ratio = ratio;
        let _ = self.b.set_param(1, ratio);
    }
    pub fn set_playing(&mut self, playing: bool) {
        self.playing = playing;
        let _ = self.b.set_param(3, playing as i32 as f32);
    }
    pub fn seek(&mut self, source_frame: f32) {
        let _ = self.b.set_param(4, source_frame);
    }
}
impl AudioNode for GpuResamplerNode {
    fn debug_name(&self) -> &'static str {
        "resampler"
    }
    fn info(&self) -> AudioNodeInfo {
        io(0, 0, 1, 64, false)
    }
    fn activate(&mut self, _sr: u32, _mbf: usize, num_inputs: usize, num_outputs: usize) -> Result<Box<dyn AudioNodeProcessor>, Box<dyn Error>> {
        self.b.activate(
            ffi::FWGPU_RESAMPLER,
            num_inputs,
            num_outputs,
            &[self.source.id() as f32, self.ratio, self.looping as i32 as f32, self.playing as i32 as f32],
        )
    }
    fn deactivate(&mut self, _p: Option<Box<dyn AudioNodeProcessor>>) {
        self.b.deactivate()
    }
}
