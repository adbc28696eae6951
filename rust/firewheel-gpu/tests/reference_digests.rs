//! The parity scenarios on the REAL reference: `firewheel-graph`'s `AudioGraph` + `FirewheelProcessor::process_interleaved`
//! (crates/firewheel-graph/src/processor.rs:61-165), driven by the language-neutral documents of
//! `tests/golden/scenarios/*.json` (format: tests/scenario_json.py of the fwgpu repository).
//!
//! For every document whose nodes are all reference nodes (`"reference_kinds_only": true`) this test builds the graph with
//! `AudioGraph::add_node` / `connect`, sends the recorded messages through the nodes' own setters, calls
//! `process_interleaved` for every recorded process op, hashes the interleaved f32 output (sha256, little endian) and writes
//!     tests/golden/reference_digests.json = { scenario: { "calls": [sha256 per process call], "sha256_calls": sha256 of all } }
//! next to the documents.  It FAILS if a digest differs from the one recorded in the document (which is the oracle's): that is
//! the comparison that pins — or unpins — the oracle.  The repository's CPU test tier then requires the committed file to
//! equal the oracle's digests (tests/test_scenario_json.py::test_reference_digests_equal_the_oracles_when_present).
//!
//! Needs NO GPU and NOT libfwgpu (build with FWGPU_NO_LINK=1: build.rs then links nothing): only firewheel-core / firewheel-graph.  `FWGPU_SCENARIOS=<dir>` names the documents
//! (default: ../../tests/golden/scenarios relative to this crate, i.e. the fwgpu repository this crate ships in).
//! One command: scripts/pin_parity.sh.
//!
//! SOURCE ONLY in the repository that ships it: written against BillyDM/firewheel @ 2024-10-16, never compiled there (no Rust
//! toolchain in its build image).
use std::any::Any;
use std::collections::BTreeMap;
use std::num::NonZeroUsize;
use std::ops::Range;
use std::path::PathBuf;
use std::sync::Arc;

use firewheel_core::node::StreamStatus;
use firewheel_core::sample_resource::{
    InterleavedResourceF32, InterleavedResourceI16, InterleavedResourceU16, SampleResource,
};
use firewheel_graph::basic_nodes::beep_test::BeepTestNode;
use firewheel_graph::basic_nodes::sampler::{LoopRange, SamplerNode};
use firewheel_graph::basic_nodes::{DummyAudioNode, HardClipNode, MonoToStereoNode, StereoToMonoNode, SumNode, VolumeNode};
use firewheel_graph::graph::{AudioGraphConfig, EdgeID, NodeID};
use firewheel_graph::{FirewheelGraphCtx, UpdateStatus};
use serde_json::Value;
use sha2::{Digest, Sha256};

/// One sample of any of the reference's six resource types (core/sample_resource.rs:28-335) behind one `S` for `SamplerNode<S>`.
#[derive(Clone)]
enum AnySample {
    II16(Arc<InterleavedResourceI16>),
    IU16(Arc<InterleavedResourceU16>),
    IF32(Arc<InterleavedResourceF32>),
    PI16(Arc<Vec<Vec<i16>>>),
    PU16(Arc<Vec<Vec<u16>>>),
    PF32(Arc<Vec<Vec<f32>>>),
}
impl SampleResource for AnySample {
    fn num_channels(&self) -> NonZeroUsize {
        match self {
            AnySample::II16(s) => s.num_channels(),
            AnySample::IU16(s) => s.num_channels(),
            AnySample::IF32(s) => s.num_channels(),
            AnySample::PI16(s) => s.num_channels(),
            AnySample::PU16(s) => s.num_channels(),
            AnySample::PF32(s) => s.num_channels(),
        }
    }
    fn len_frames(&self) -> u64 {
        match self {
            AnySample::II16(s) => s.len_frames(),
            AnySample::IU16(s) => s.len_frames(),
            AnySample::IF32(s) => s.len_frames(),
            AnySample::PI16(s) => s.len_frames(),
            AnySample::PU16(s) => s.len_frames(),
            AnySample::PF32(s) => s.len_frames(),
        }
    }
    fn fill_buffers(&self, buffers: &mut [&mut [f32]], buffer_range: Range<usize>, start_frame: u64) {
        match self {
            AnySample::II16(s) => s.fill_buffers(buffers, buffer_range, start_frame),
            AnySample::IU16(s) => s.fill_buffers(buffers, buffer_range, start_frame),
            AnySample::IF32(s) => s.fill_buffers(buffers, buffer_range, start_frame),
            AnySample::PI16(s) => s.fill_buffers(buffers, buffer_range, start_frame),
            AnySample::PU16(s) => s.fill_buffers(buffers, buffer_range, start_frame),
            AnySample::PF32(s) => s.fill_buffers(buffers, buffer_range, start_frame),
        }
    }
}

// ---- the documents' generator (tests/fwapi.py xorshift_uniform; spelled out in tests/scenario_json.py)
fn fmix32_stream(seed: u32, count: usize) -> Vec<f32> {
    (0..count)
        .map(|i| {
            let mut x = seed.wrapping_add(((i as u32).wrapping_add(1)).wrapping_mul(0x9E37_79B9));
            x ^= x >> 16;
            x = x.wrapping_mul(0x85EB_CA6B);
            x ^= x >> 13;
            x = x.wrapping_mul(0xC2B2_AE35);
            x ^= x >> 16;
            (x >> 8) as f32 * (1.0f32 / 8_388_608.0) - 1.0
        })
        .collect()
}
enum Data {
    F32(Vec<f32>),
    I16(Vec<i16>),
    U16(Vec<u16>),
}
/// a data record -> CHANNEL-MAJOR values (generator) or the raw bytes as stored (already in the sample's own layout)
fn data_of(rec: &Value) -> (Data, bool) {
    if let Some(b64) = rec.get("raw_b64") {
        use base64::Engine;
        let bytes = base64::engine::general_purpose::STANDARD.decode(b64.as_str().unwrap()).unwrap();
        let d = match rec["dtype"].as_str().unwrap() {
            "f32" => Data::F32(bytes.chunks_exact(4).map(|c| f32::from_le_bytes([c[0], c[1], c[2], c[3]])).collect()),
            "i16" => Data::I16(bytes.chunks_exact(2).map(|c| i16::from_le_bytes([c[0], c[1]])).collect()),
            _ => Data::U16(bytes.chunks_exact(2).map(|c| u16::from_le_bytes([c[0], c[1]])).collect()),
        };
        return (d, true);
    }
    assert_eq!(rec["gen"], "fmix32");
    let x = fmix32_stream(rec["seed"].as_u64().unwrap() as u32, rec["count"].as_u64().unwrap() as usize);
    let d = match rec["quant"].as_str().unwrap() {
        // numpy's np.round = round half to even, in f32 arithmetic
        "i16" => Data::I16(x.iter().map(|v| (v * 32767.0f32).round_ties_even() as i16).collect()),
        "u16" => Data::U16(x.iter().map(|v| ((v + 1.0f32) * 32767.5f32).round_ties_even() as u16).collect()),
        _ => Data::F32(x),
    };
    (d, false)
}
fn planar<T: Copy>(v: &[T], channels: usize, stored_interleaved: bool) -> Vec<Vec<T>> {
    let frames = v.len() / channels;
    (0..channels)
        .map(|c| (0..frames).map(|f| if stored_interleaved { v[f * channels + c] } else { v[c * frames + f] }).collect())
        .collect()
}
fn interleaved<T: Copy>(v: &[T], channels: usize, stored_interleaved: bool) -> Vec<T> {
    if stored_interleaved {
        return v.to_vec();
    }
    let frames = v.len() / channels;
    (0..frames).flat_map(|f| (0..channels).map(move |c| (f, c))).map(|(f, c)| v[c * frames + f]).collect()
}
fn make_sample(fmt: u64, channels: usize, rec: &Value) -> AnySample {
    let (d, raw) = data_of(rec);
    let ch = NonZeroUsize::new(channels).unwrap();
    // raw bytes are in the format's own layout; generated streams are channel-major
    let il_stored = raw && fmt <= 2;
    match (fmt, d) {
        (0, Data::I16(v)) => AnySample::II16(Arc::new(InterleavedResourceI16 { data: interleaved(&v, channels, il_stored), channels: ch })),
        (1, Data::U16(v)) => AnySample::IU16(Arc::new(InterleavedResourceU16 { data: interleaved(&v, channels, il_stored), channels: ch })),
        (2, Data::F32(v)) => AnySample::IF32(Arc::new(InterleavedResourceF32 { data: interleaved(&v, channels, il_stored), channels: ch })),
        (3, Data::I16(v)) => AnySample::PI16(Arc::new(planar(&v, channels, false))),
        (4, Data::U16(v)) => AnySample::PU16(Arc::new(planar(&v, channels, false))),
        (5, Data::F32(v)) => AnySample::PF32(Arc::new(planar(&v, channels, false))),
        (f, _) => panic!("sample format {f} with data of another type"),
    }
}

fn sha_f32(v: &[f32]) -> String {
    let mut h = Sha256::new();
    for x in v {
        h.update(x.to_le_bytes());
    }
    h.finalize().iter().map(|b| format!("{b:02x}")).collect()
}

type Sampler = SamplerNode<AnySample>;

/// Replays one document; returns the digest of every process call.
fn replay(doc: &Value) -> Vec<String> {
    let sample_rate = doc["sample_rate"].as_u64().unwrap() as u32;
    let mbf = doc["max_block_frames"].as_u64().unwrap() as usize;
    let n_in = doc["num_graph_inputs"].as_u64().unwrap() as usize;
    let n_out = doc["num_graph_outputs"].as_u64().unwrap() as usize;
    let mut cx = FirewheelGraphCtx::new(AudioGraphConfig {
        num_graph_inputs: n_in,
        num_graph_outputs: n_out,
        ..Default::default()
    });
    // graph/context.rs:46-82: the processor half, driven by this thread in place of a backend callback
    let user_cx: Box<dyn Any + Send> = Box::new(());
    let mut processor = cx.activate(sample_rate, n_in, n_out, mbf, user_cx).expect("activate");
    let (g_in, g_out) = (cx.graph.graph_in_node(), cx.graph.graph_out_node());
    let mut nodes: Vec<NodeID> = Vec::new();
    let mut edges: Vec<Option<EdgeID>> = Vec::new();
    let mut samples: Vec<AnySample> = Vec::new();
    let mut calls = Vec::new();
    let node = |i: i64, nodes: &Vec<NodeID>| -> NodeID {
        match i {
            -1 => g_in,
            -2 => g_out,
            i => nodes[i as usize],
        }
    };
    for op in doc["ops"].as_array().unwrap() {
        let a = op.as_array().unwrap();
        match a[0].as_str().unwrap() {
            "add_node" => {
                let (kind, ni, no) = (a[1].as_u64().unwrap(), a[2].as_u64().unwrap() as usize, a[3].as_u64().unwrap() as usize);
                let p: Vec<f32> = a[4].as_array().unwrap().iter().map(|v| v.as_f64().unwrap() as f32).collect();
                // kinds = include/fwgpu.h FWGPU_KIND_*; constructor arguments = the node's `new` (basic_nodes/*.rs)
                let id = match kind {
                    0 => cx.graph.add_node(ni, no, DummyAudioNode),
                    1 => cx.graph.add_node(ni, no, BeepTestNode::new(p[0], p[1], p[2] != 0.0)),
                    2 => cx.graph.add_node(ni, no, VolumeNode::new(p[0])),
                    3 => cx.graph.add_node(ni, no, SumNode),
                    4 => cx.graph.add_node(ni, no, Sampler::new(p[0])),
                    5 => cx.graph.add_node(ni, no, HardClipNode::new(p[0])),
                    6 => cx.graph.add_node(ni, no, MonoToStereoNode),
                    7 => cx.graph.add_node(ni, no, StereoToMonoNode),
                    k => panic!("node kind {k} is not a reference node"),
                };
                nodes.push(id);
            }
            "remove_node" => {
                let _ = cx.graph.remove_node(node(a[1].as_i64().unwrap(), &nodes));
            }
            "connect" => {
                let r = cx.graph.connect(
                    node(a[1].as_i64().unwrap(), &nodes),
                    a[2].as_u64().unwrap() as usize,
                    node(a[3].as_i64().unwrap(), &nodes),
                    a[4].as_u64().unwrap() as usize,
                    a[5].as_bool().unwrap(),
                );
                assert_eq!(r.is_err(), a[6].as_i64().unwrap() != 0, "connect outcome differs from the recorded one: {op}");
                edges.push(r.ok());
            }
            "disconnect" => {
                cx.graph.disconnect(
                    node(a[1].as_i64().unwrap(), &nodes),
                    a[2].as_u64().unwrap() as usize,
                    node(a[3].as_i64().unwrap(), &nodes),
                    a[4].as_u64().unwrap() as usize,
                );
            }
            "disconnect_edge" => {
                if let Some(e) = edges[a[1].as_u64().unwrap() as usize] {
                    cx.graph.disconnect_by_edge_id(e);
                }
            }
            "update" => {
                // FirewheelGraphCtx::update compiles and sends the schedule; the processor picks it up at its next call
                match cx.update() {
                    UpdateStatus::Active { graph_error } => assert_eq!(graph_error.is_some(), a[1].as_i64().unwrap() != 0, "{op}"),
                    _ => panic!("context not active"),
                }
            }
            "new_sample" => samples.push(make_sample(a[1].as_u64().unwrap(), a[2].as_u64().unwrap() as usize, &a[4])),
            "set_param" => {
                let id = node(a[1].as_i64().unwrap(), &nodes);
                let v = a[3].as_f64().unwrap() as f32;
                let n = cx.graph.node_mut(id).expect("node");
                // param 0 of the three reference nodes that have one (include/fwgpu.h, fwgpu_node_set_param)
                if let Some(x) = n.downcast_mut::<VolumeNode>() {
                    x.set_percent_volume(v);
                } else if let Some(x) = n.downcast_mut::<Sampler>() {
                    x.set_percent_volume(v);
                } else if let Some(x) = n.downcast_mut::<BeepTestNode>() {
                    x.set_enabled(v != 0.0);
                } else {
                    panic!("set_param on a node without parameters: {op}");
                }
            }
            m @ ("set_sample" | "play" | "pause" | "stop" | "set_playhead_secs" | "set_loop_range") => {
                let id = node(a[1].as_i64().unwrap(), &nodes);
                let s = cx.graph.node_mut(id).expect("node").downcast_mut::<Sampler>().expect("a sampler");
                // the reference drops a message when its ring is full (`Result<(), ()>`): the documents never send that many
                match m {
                    "set_sample" => s.set_sample(samples[a[2].as_u64().unwrap() as usize].clone(), a[3].as_bool().unwrap()).unwrap(),
                    "play" => s.play().unwrap(),
                    "pause" => s.pause().unwrap(),
                    "stop" => s.stop().unwrap(),
                    "set_playhead_secs" => s.set_playhead(a[2].as_f64().unwrap()).unwrap(),
                    _ => {
                        let r = match a[2].as_u64().unwrap() {
                            0 => None,
                            1 => Some(LoopRange::Full),
                            _ => Some(LoopRange::RangeSecs(a[3].as_f64().unwrap()..a[4].as_f64().unwrap())),
                        };
                        s.set_loop_range(r).unwrap()
                    }
                }
            }
            "process" => {
                let frames = a[1].as_u64().unwrap() as usize;
                let (ci, co) = (a[2].as_u64().unwrap() as usize, a[3].as_u64().unwrap() as usize);
                let input: Vec<f32> = if a[4].is_null() {
                    vec![0.0; frames * ci]
                } else {
                    match data_of(&a[4]).0 {
                        Data::F32(v) => v,
                        _ => panic!("stream input must be f32"),
                    }
                };
                let mut out = vec![f32::NAN; frames * co];
                processor.process_interleaved(
                    &input,
                    &mut out,
                    ci,
                    co,
                    frames,
                    a[5].as_f64().unwrap(),
                    StreamStatus::from_bits_truncate(a[6].as_u64().unwrap() as u32),
                );
                let d = sha_f32(&out);
                assert_eq!(d, a[7].as_str().unwrap(), "{}: process call {} differs from the oracle's recorded digest", doc["name"], calls.len());
                calls.push((d, out));
            }
            other => panic!("unknown op {other}"),
        }
    }
    calls.into_iter().map(|(d, _)| d).collect()
}

#[test]
fn reference_reproduces_the_recorded_digests_of_every_replayable_scenario() {
    let dir = std::env::var("FWGPU_SCENARIOS")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("../../tests/golden/scenarios"));
    let mut out = BTreeMap::new();
    let mut names: Vec<_> = std::fs::read_dir(&dir).expect("scenario directory").filter_map(|e| e.ok()).map(|e| e.path()).collect();
    names.sort();
    for p in names {
        if p.extension().and_then(|e| e.to_str()) != Some("json") || p.file_name().unwrap() == "index.json" {
            continue;
        }
        let doc: Value = serde_json::from_str(&std::fs::read_to_string(&p).unwrap()).unwrap();
        if !doc["reference_kinds_only"].as_bool().unwrap() {
            continue; // SPEC nodes (pan, width, biquad, delay, FIR, resampler, spatialiser): nothing in the reference to run
        }
        let calls = replay(&doc);
        // the digest over all calls' samples in order: re-hash is not possible from digests, so replay keeps it simple — the
        // per-call digests all matched (asserted above), hence the concatenation's digest is the document's
        out.insert(
            doc["name"].as_str().unwrap().to_string(),
            serde_json::json!({ "calls": calls, "sha256_calls": doc["sha256_calls"] }),
        );
    }
    assert!(out.len() >= 6, "expected the reference-only scenarios among the documents, found {}", out.len());
    let dst = dir.join("../reference_digests.json");
    std::fs::write(&dst, serde_json::to_string_pretty(&out).unwrap() + "\n").unwrap();
    eprintln!("wrote {} ({} scenarios): the reference reproduces the oracle bit for bit on all of them", dst.display(), out.len());
}
