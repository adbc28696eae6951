/* fwgpu.h — C ABI of libfwgpu: the MI355X (gfx950) per-block DSP executor for Firewheel audio graphs.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no C++/torch types.
 * Each entry point names the reference interface (BillyDM/firewheel @ 2024-10-16) it replaces:
 *   core/  = crates/firewheel-core/src/      graph/ = crates/firewheel-graph/src/
 *   nodes/ = crates/firewheel-graph/src/basic_nodes/
 *
 * Conventions (replacing Rust's Result/panic, SURVEY §8b "error convention"):
 *   - functions return int: 0 = ok, negative = error; handles are int64 (>= 0) or negative error.
 *   - nothing throws or aborts across the ABI; fwgpu_last_error() returns the message.
 *   - threading: a ctx has an AUDIO side — fwgpu_process_interleaved / _process_blocks_device[_flags] /
 *     _node_process / _stream_callback, one thread at a time — and a CONTROL side, everything else, ALSO one thread at a
 *     time (a host with several control threads serialises them itself: a message call looks its node up in the graph a
 *     graph call may be growing).  Every control call may run WHILE a process call is in flight:
 *       * messages (fwgpu_node_set_param, fwgpu_sampler_*) go through a lock-free ring the audio side drains at the start
 *         of its next call (the reference's Arc<AtomicF32> gains and rtrb rings: nodes/volume.rs:10,28-34,
 *         nodes/sampler.rs:14,171-177,205-208);
 *       * graph calls (add_node / remove_node / connect / disconnect / host_node_set_process / update / schedule_upload):
 *         update BUILDS the new plan off to the side on the calling thread and the next process call adopts it at its
 *         start, as the reference hands a new schedule over through a ring (graph/processor.rs:167-206;
 *         fwgpu_plan_handover_stats) — the audio side neither locks nor allocates nor waits for the build;
 *       * sample-table and configuration calls (sample_create / sample_destroy / set_force_generic) upload their data first
 *         and then hold process calls at their ENTRY for the few microseconds the table entry takes (they wait for a
 *         running call to finish first).  fwgpu_set_max_batch takes effect with the next update.
 *   - once warm, a process call touches neither the host allocator nor the device allocator; a failing call writes its
 *     message into a fixed buffer.  fwgpu_process_interleaved fills `output` on EVERY return (zeros on error:
 *     core/node.rs:41-42).
 *   - the library fails loudly (ctx_create returns NULL) when no HIP device / kernel image is usable;
 *     there is no CPU fallback anywhere behind this ABI.
 */
#ifndef FWGPU_H
#define FWGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fwgpu_ctx fwgpu_ctx;

/* node kinds — nodes/mod.rs:1-15 plus the north-star nodes the reference lists as TODO (README.md:14-19) */
enum fwgpu_node_kind {
    FWGPU_DUMMY = 0,          /* nodes/dummy.rs */
    FWGPU_BEEP_TEST = 1,      /* nodes/beep_test.rs   params: freq_hz, gain_db, enabled */
    FWGPU_VOLUME = 2,         /* nodes/volume.rs      params: percent_volume */
    FWGPU_SUM = 3,            /* nodes/sum.rs */
    FWGPU_SAMPLER = 4,        /* nodes/sampler.rs     params: percent_volume */
    FWGPU_HARD_CLIP = 5,      /* nodes/hard_clip.rs   params: threshold_db */
    FWGPU_MONO_TO_STEREO = 6, /* nodes/mono_to_stereo.rs */
    FWGPU_STEREO_TO_MONO = 7, /* nodes/stereo_to_mono.rs */
    FWGPU_STEREO_PAN = 8,     /* SPEC (not in reference) params: pan in [-1,1] */
    FWGPU_STEREO_WIDTH = 9,   /* SPEC                 params: width */
    FWGPU_BIQUAD = 10,        /* SPEC                 params: type, cutoff_hz, q */
    FWGPU_DELAY = 11,         /* SPEC                 params: delay_secs, feedback, mix */
    FWGPU_FIR = 12,           /* SPEC convolution     params: impulse-response sample id (fwgpu_sample_create) */
    FWGPU_RESAMPLER = 13,     /* SPEC polyphase resampling source (0 inputs)  params: sample id, ratio, loop, playing */
    FWGPU_SPATIAL = 14,       /* SPEC 3D spatialiser (1|2 in, 2 out)          params: x, y, z of the source */
    FWGPU_HOST_NODE = 15      /* any other `dyn AudioNodeProcessor` (graph/processor.rs:243): runs on the HOST, see below */
};

/* sample formats — core/sample_resource.rs:28-335 */
enum fwgpu_sample_format {
    FWGPU_INTERLEAVED_I16 = 0,
    FWGPU_INTERLEAVED_U16 = 1,
    FWGPU_INTERLEAVED_F32 = 2,
    FWGPU_PLANAR_I16 = 3, /* data = channels planes of `frames` items, concatenated */
    FWGPU_PLANAR_U16 = 4,
    FWGPU_PLANAR_F32 = 5
};

/* error codes: AddEdgeError / CompileGraphError variants (graph/graph/error.rs) */
enum fwgpu_error {
    FWGPU_OK = 0,
    FWGPU_ERR_SRC_NODE_NOT_FOUND = -1,
    FWGPU_ERR_DST_NODE_NOT_FOUND = -2,
    FWGPU_ERR_IN_PORT_OUT_OF_RANGE = -3,
    FWGPU_ERR_OUT_PORT_OUT_OF_RANGE = -4,
    FWGPU_ERR_EDGE_ALREADY_EXISTS = -5,
    FWGPU_ERR_INPUT_PORT_ALREADY_CONNECTED = -6,
    FWGPU_ERR_CYCLE_DETECTED = -7,
    FWGPU_ERR_COMPILE_CYCLE = -10,
    FWGPU_ERR_COMPILE_MANY_TO_ONE = -11,
    FWGPU_ERR_NODE_ACTIVATION_FAILED = -12,
    FWGPU_ERR_INVALID = -20,   /* bad handle / argument */
    FWGPU_ERR_QUEUE_FULL = -21,/* sampler message ring full (nodes/sampler.rs:14 CHANNEL_CAPACITY) */
    FWGPU_ERR_DEVICE = -30     /* HIP error; see fwgpu_last_error */
};

/* ---- context: FirewheelGraphCtx::new + activate (graph/context.rs:35-82) and the device-resident
 * FirewheelProcessor (graph/processor.rs:34-55).  `hip_stream` may be NULL (the ctx creates its own
 * stream) or a hipStream_t the caller owns (e.g. torch's current stream). */
fwgpu_ctx* fwgpu_ctx_create(int device, uint32_t sample_rate, uint32_t max_block_frames,
                            uint32_t num_graph_inputs, uint32_t num_graph_outputs, void* hip_stream);
void fwgpu_ctx_destroy(fwgpu_ctx* ctx);
const char* fwgpu_last_error(fwgpu_ctx* ctx);
/* message of the last failed fwgpu_ctx_create (which has no ctx to ask) */
const char* fwgpu_create_error(void);

/* ---- graph edit API: AudioGraph (graph/graph.rs) */
int64_t fwgpu_graph_in_node(fwgpu_ctx* ctx);  /* graph.rs:189 */
int64_t fwgpu_graph_out_node(fwgpu_ctx* ctx); /* graph.rs:194 */
/* graph.rs:201-231 add_node(num_inputs, num_outputs, node); `params` are the node constructor args */
int64_t fwgpu_add_node(fwgpu_ctx* ctx, int kind, uint32_t num_inputs, uint32_t num_outputs,
                       const float* params, int n_params);
int fwgpu_remove_node(fwgpu_ctx* ctx, int64_t node); /* graph.rs:268-299 */
/* graph.rs:396-477; returns the edge id or an AddEdgeError code */
int64_t fwgpu_connect(fwgpu_ctx* ctx, int64_t src_node, uint32_t src_port, int64_t dst_node,
                      uint32_t dst_port, int check_for_cycles);
int fwgpu_disconnect(fwgpu_ctx* ctx, int64_t src_node, uint32_t src_port, int64_t dst_node,
                     uint32_t dst_port);                 /* graph.rs:483-501; 1 = removed, 0 = no such edge */
int fwgpu_disconnect_edge(fwgpu_ctx* ctx, int64_t edge); /* graph.rs:507-524 */
int fwgpu_cycle_detected(fwgpu_ctx* ctx);                /* graph.rs:573-580 */
/* FirewheelGraphCtx::update (graph/context.rs:93-137): recompile when dirty, activate new nodes
 * (AudioNode::activate, core/node.rs:12-18), build + upload the device launch plan. */
int fwgpu_update(fwgpu_ctx* ctx);

/* ---- custom nodes inside a device-resident graph ("unknown/custom Rust nodes fall back to B1 on host with explicit
 * D2H/H2D of just their buffers").  The reference calls ANY `dyn AudioNodeProcessor` from its schedule loop
 * (graph/processor.rs:226-247); a node this library has no kernel for is added as FWGPU_HOST_NODE with its port counts and
 * given its process function — `AudioNodeProcessor::process` + `ProcInfo` (core/node.rs:37-53,94-118) as a C callback:
 * inputs[i] / outputs[i] are `frames` floats, in_silence_mask bit i = input i is silent, *out_silence_mask arrives 0
 * (processor.rs:233), every output must be filled.  The launch plan is CUT at such a node's level: the levels above it run on
 * the device, its input buffers (and only those) are copied to pinned host memory, the callback runs on the AUDIO thread —
 * once per block, in block order, like the reference — its outputs go back, the levels below continue.  A process call of K
 * blocks crosses the boundary twice per host level, not twice per block.  Voice banks of the graph stay on the fused kernels
 * (hybrid plan).  The function must be set before the fwgpu_update / fwgpu_schedule_upload that activates the node; it must
 * not call back into this ctx. */
typedef void (*fwgpu_host_process_fn)(void* user, uint64_t frames, const float* const* inputs, uint32_t num_inputs,
                                      float* const* outputs, uint32_t num_outputs, uint64_t in_silence_mask,
                                      uint64_t* out_silence_mask, double stream_time_secs, uint32_t stream_status);
int fwgpu_host_node_set_process(fwgpu_ctx* ctx, int64_t node, fwgpu_host_process_fn fn, void* user);
/* how many host nodes the installed plan holds / how often their callbacks have run since the plan was installed */
int fwgpu_plan_host_nodes(fwgpu_ctx* ctx, uint64_t* callbacks_run);

/* ---- external schedule import: keep Firewheel's own Rust scheduler and hand its CompiledSchedule over
 * (graph/graph/compiler/schedule.rs:12-30,105-126,166-173).  Buffer indices are the reference's; the
 * library reconstructs the edges, renames buffers (no reuse inside a level) and levelises. */
typedef struct fwgpu_sched_node {
    int64_t node;                 /* id returned by fwgpu_add_node / graph_in / graph_out */
    uint32_t num_inputs;
    uint32_t num_outputs;
    const uint32_t* in_buffer_index;  /* [num_inputs]  InBufferAssignment.buffer_index */
    const uint8_t* in_should_clear;   /* [num_inputs]  InBufferAssignment.should_clear */
    const uint32_t* out_buffer_index; /* [num_outputs] OutBufferAssignment.buffer_index */
} fwgpu_sched_node;
int fwgpu_schedule_upload(fwgpu_ctx* ctx, const fwgpu_sched_node* nodes, uint32_t n_nodes,
                          uint32_t num_buffers);

/* ---- introspection of the device launch plan (tests) */
/* 0 = generic level-batched executor, 1 = fused voice-bank plan (k_leaf_sum),
 * 2 = fused chain plan (voices with a biquad / delay: k_chain), 3 = hybrid: voice banks inside a graph that is not a fused
 * shape as a whole (sends, bus effects, anything) are rendered by the voice-bank kernels, the rest by the level executor */
int fwgpu_plan_kind(fwgpu_ctx* ctx);
/* how many voices (source -> stages chains) of the installed plan the fused kernels render — all of them on plans 1 and 2,
 * the banks' on plan 3, 0 on plan 0; -1 without a plan */
int fwgpu_plan_fused_voices(fwgpu_ctx* ctx);
int fwgpu_plan_num_levels(fwgpu_ctx* ctx);
/* level of a node in the plan (graph_in = 0); negative if unknown */
int fwgpu_plan_node_level(fwgpu_ctx* ctx, int64_t node);
/* per input port: 1 if the port is unconnected (the reference's should_clear), else 0.  Returns n. */
int fwgpu_plan_node_inputs_clear(fwgpu_ctx* ctx, int64_t node, int* should_clear, int cap);
/* chain plan (kind 2) only: k_chain workgroup launches since the plan was installed that ran the steady-call loop
 * (every voice of the leaf steady for the whole call, delays >= 3 tiles, no message pending) / the general loop */
int fwgpu_plan_chain_stats(fwgpu_ctx* ctx, uint64_t* steady_workgroups, uint64_t* general_workgroups);
/* Plan hand-over (graph/context.rs:93-137 + graph/processor.rs:167-206).  fwgpu_update / fwgpu_schedule_upload BUILD the new
 * plan on the calling (control) thread, off to the side, while process calls keep running on the current one; the next process
 * call adopts it at its start — a swap of descriptors plus a handful of asynchronous launches for the nodes it activates, no
 * allocation, no wait for the device — and the old plan travels back to the control side, whose next update reuses its
 * buffers.  (When no process call is in flight the updating thread adopts the plan itself, right away.)  *adoptions = plans
 * adopted so far, *audio_adoptions = those a process call adopted, *max_adopt_ns = the longest one of THOSE held up its process
 * call (host nanoseconds).  Any pointer may be NULL. */
int fwgpu_plan_handover_stats(fwgpu_ctx* ctx, uint64_t* adoptions, uint64_t* audio_adoptions, uint64_t* max_adopt_ns);
/* 1 while a plan built by fwgpu_update / fwgpu_schedule_upload waits for a process call to adopt it, else 0.  Once an update has
 * returned and this reads 0, the plan it built is the active one: nothing the update removed — a host node's callback and its
 * `user` pointer above all — will be called again, and its owner may free it (the reference drops a removed node's processor when
 * the old schedule comes back through the ring, graph/processor.rs:182-188; rust/firewheel-gpu's HostNodeHandle and the Python
 * wrapper keep removed host nodes in limbo until this says so).  Any thread may ask. */
int fwgpu_plan_pending(fwgpu_ctx* ctx);
/* Voice-bank and chain plans, diagnostics: launch batches of process calls rendered WITHOUT a control kernel (*lazy_batches) and with one.
 * A message-free call of a plan whose every voice the last control kernel left steady and plain (silent, or a planar-f32 source
 * or interleaved 16-bit source that never wraps inside a block) needs no per-block state machine pass — the reference's processors do nothing in such a block
 * but advance a playhead (nodes/sampler.rs:445-484) — and the leaf kernel derives each block's record from one per-voice
 * record; the host skips the control kernel once it has SEEN (pinned memory) that the last one found every voice so.  A host that
 * never waits for the device between calls sees no difference but the time.  FWGPU_LAZY=0 switches it off.  Either may be NULL. */
int fwgpu_lazy_stats(fwgpu_ctx* ctx, uint64_t* lazy_batches, uint64_t* control_batches);
/* The hipStream_t every process call of this ctx launches on (the caller's, or the one fwgpu_ctx_create made): for callers that
 * order their own device work or events against the engine's (bench.py's per-step time distribution).  NULL ctx -> NULL. */
void* fwgpu_hip_stream(fwgpu_ctx* ctx);
/* Diagnostics for the same hand-over: which part of fwgpu_update / fwgpu_schedule_upload the control thread is in right now —
 * 0 none, 1 compiling the graph (host only: graph/compiler.rs), 21..28 the sections of the plan build that upload tables
 * (23 node tables, 26 buffer pool, 27 voice tables, 28 staging areas), 3 waiting for the last upload.  Any thread may ask;
 * examples/host_c/fw_edit_race.c tags every callback with it to say WHEN a build reaches the audio side. */
int fwgpu_update_phase(fwgpu_ctx* ctx);
/* Realtime edge, resident kernel (cpal/lib.rs:378-449: a backend thread that is woken per block, never torn down between blocks).
 * The first steady one-block fwgpu_process_interleaved call of a run of them — voice-bank plan, stereo stream, no message pending —
 * launches a kernel that stays resident and is handed every following callback through a doorbell word in pinned host memory: no
 * launch call on the audio thread, no grid dispatch.  Anything else that touches the device state (a call with a message, a call of
 * another size, a plan adoption, fwgpu_node_process, destroy) ends it first; its own WATCHDOG ends it after `idle_ms` without a
 * callback (default 20; FWGPU_RT_IDLE_MS), so it can never hold a device that nobody feeds — the next callback launches a new one.
 * FWGPU_RT_PERSIST=0 switches it off (every callback is then one k_rt_block launch).  *launches = resident kernels launched so far,
 * *doorbells = callbacks served without a launch.  Any pointer may be NULL. */
int fwgpu_rt_resident_stats(fwgpu_ctx* ctx, uint64_t* launches, uint64_t* doorbells);
/* Which path the ONE-BLOCK calls of this context took so far (cpal/lib.rs:429-437 calls once per block for every graph; only some
 * plans have the one-launch edge): paths[0] = the resident kernel's doorbell (no launch), paths[1] = one k_rt_block launch,
 * paths[2] = the fused plans' ordinary launch sequence (a chain plan, a spatialiser / master chain, a tree of more than two levels,
 * a call that carried a message), paths[3] = the level executor (generic / hybrid plans).  A host reads it to see that its
 * callbacks run where it expects them to. */
int fwgpu_rt_path_stats(fwgpu_ctx* ctx, uint64_t* paths);  /* paths: room for 4 */
/* K = the most blocks one fused launch sequence processes (default 64); sizes the K-batched descriptor,
 * ramp and bus buffers at the next fwgpu_update. */
int fwgpu_set_max_batch(fwgpu_ctx* ctx, uint32_t max_blocks);
/* force the generic executor even when the fused plan matches (parity tests) */
int fwgpu_set_force_generic(fwgpu_ctx* ctx, int on);
/* floats of the per-node extended state pool (delay rings, FIR history, filter state) handed out / allocated: removing
 * a node returns its slice for reuse, so a host that spawns and retires effect voices sees `in_use` level off */
int fwgpu_ext_pool_floats(fwgpu_ctx* ctx, uint64_t* in_use, uint64_t* capacity);

/* ---- sample resources: SampleResource (core/sample_resource.rs:4-26), uploaded once, HBM-resident.
 * Returns a sample id >= 0. */
int fwgpu_sample_create(fwgpu_ctx* ctx, int format, uint32_t channels, uint64_t frames, const void* data);
/* same, from memory that is already on this device (bench: generate in HBM, skip PCIe) */
int fwgpu_sample_create_device(fwgpu_ctx* ctx, int format, uint32_t channels, uint64_t frames,
                               const void* device_data);
/* Releases the sample's HBM (for an id from fwgpu_sample_create; a _device sample only loses its table entry).  The
 * reference keeps a sample alive through the Arc every processor holds; here the CALLER guarantees that no sampler
 * will read the id any more (the Rust shim calls this when the last Arc drops).  Waits for the work
 * in flight on the context's stream.  Ids are never reused.  A sampler that still holds the id plays an EMPTY
 * (0-frame) sample from the next process call on — silence, never a stale pointer; FIR / resampler nodes name their
 * sample at construction, so the call is refused (FWGPU_ERR_INVALID) while such a node exists. */
int fwgpu_sample_destroy(fwgpu_ctx* ctx, int sample);
/* ProcessorToNodeMsg::ReturnSample (nodes/sampler.rs:339-343,563-571): when a SetSample message replaces the sample a
 * sampler holds — or a sampler node is removed and its processor dropped — the processor hands the old one back to the
 * node, which is when the host may let go of it.  Here: every sample that a process call COMPLETED on the device has
 * swapped out since the last poll, oldest first, as (node id, sample id) pairs.  Control side, never blocks (it asks
 * the completion events, it does not wait for them).  Returns the number of pairs written (<= cap). */
int fwgpu_poll_returned_samples(fwgpu_ctx* ctx, int64_t* nodes, int* samples, int cap);
/* 1 when no sampler holds `sample`, no queued SetSample message names it, every process call that read it has completed
 * on the device and no FIR / resampler node was built on it — i.e. fwgpu_sample_destroy is safe and changes no sound;
 * 0 otherwise.  (The reference gets this from Arc's count reaching zero after the ReturnSample above.) */
int fwgpu_sample_retired(fwgpu_ctx* ctx, int sample);

/* ---- control -> audio messages.  `at_block` = index of the max_block_frames-sized block, counted from
 * the start of the NEXT process call, before which the message is seen (the reference's rings/atomics
 * are polled at block start: nodes/sampler.rs:331, nodes/volume.rs:92).  Messages to one node apply in (at_block, send) order.
 * Send a node's messages in NON-DECREASING at_block order: the reference has no tag at all — "message, then process" is the only
 * order it knows — and parameters the control side folds into derived state (a biquad's cutoff and Q into its five coefficients,
 * computed in f64 with the host's libm) are folded WHEN THE MESSAGE IS SENT, with the node's other parameters as last sent; out
 * of block order they take effect as sent, not as tagged (found by tests/test_chain_grammar.py's fuzz, round 6). */
/* VolumeNode::set_percent_volume (volume.rs:28-34) / SamplerNode::set_percent_volume (sampler.rs:171-177):
 * param 0.  BeepTestNode::set_enabled (beep_test.rs:30-32): param 0.  SPEC nodes: StereoPan 0 = pan;
 * StereoWidth 0 = width; Biquad 1 = cutoff_hz, 2 = q; Delay 1 = feedback, 2 = mix; Resampler 1 = ratio,
 * 3 = playing, 4 = seek (source frame); Spatial 0/1/2 = x/y/z. */
int fwgpu_node_set_param(fwgpu_ctx* ctx, int64_t node, int param, float value, uint32_t at_block);
/* The same for `n` messages in one call, in order (hosts behind a foreign-function interface pay per call: bench.py's variant B
 * issues a hundred per step).  Stops at the first message that fails and returns its (negative) error; the ones in front of it
 * stay sent.  0 = all sent. */
int fwgpu_node_set_params(fwgpu_ctx* ctx, uint32_t n, const int64_t* nodes, const int* params, const float* values, const uint32_t* at_blocks);
int fwgpu_sampler_set_sample(fwgpu_ctx* ctx, int64_t node, int sample, int stop_playback,
                             uint32_t at_block);                                  /* sampler.rs:67-79 */
int fwgpu_sampler_play(fwgpu_ctx* ctx, int64_t node, uint32_t at_block);  /* sampler.rs:82-97 */
int fwgpu_sampler_pause(fwgpu_ctx* ctx, int64_t node, uint32_t at_block); /* sampler.rs:100-115 */
int fwgpu_sampler_stop(fwgpu_ctx* ctx, int64_t node, uint32_t at_block);  /* sampler.rs:118-133 */
int fwgpu_sampler_set_playhead_secs(fwgpu_ctx* ctx, int64_t node, double playhead_secs,
                                    uint32_t at_block);                           /* sampler.rs:136-147 */
/* mode: 0 = None, 1 = LoopRange::Full, 2 = LoopRange::RangeSecs(start..end)  (sampler.rs:150-161) */
int fwgpu_sampler_set_loop_range(fwgpu_ctx* ctx, int64_t node, int mode, double start_secs,
                                 double end_secs, uint32_t at_block);

/* ---- processing */
/* FirewheelProcessor::process_interleaved (graph/processor.rs:61-165): host buffers, splits `frames`
 * into max_block_frames blocks, returns after the output has landed in `output`. */
int fwgpu_process_interleaved(fwgpu_ctx* ctx, const float* input, float* output, uint32_t num_in_channels,
                              uint32_t num_out_channels, uint64_t frames, double stream_time_secs,
                              uint32_t stream_status);
/* The same call split in two, for hosts that render ahead (a bounce, an offline render) and keep the host-buffer boundary: `begin` is
 * process_interleaved up to its last launch — the graph-output kernel writes the interleaved frames straight into one of TWO page-locked,
 * device-mapped host staging blocks — and returns a ticket (>= 0); `end` waits for that ticket's last launch and hands the frames to
 * `output`.  With begin(n + 1) called before end(n) the host's share of call n (the wait, the memcpy) overlaps the rendering of call
 * n + 1 and no copy engine or blit kernel sits between two renders: config 2 runs at 0.90 of the device-resident rate this way, 0.84
 * through the synchronous call.  At most two calls in flight; tickets are ended in order.  `end` on the oldest
 * ticket in flight fills `output` on every return (zeros when the device failed, core/node.rs:41-42); `end` on anything else — not a
 * ticket, not the oldest one, a null `output` for a ticket with frames — returns FWGPU_ERR_INVALID, touches nothing (it does not know
 * a size to fill) and leaves the tickets in flight as they were.  `cancel` abandons every ticket in flight up to and including
 * `ticket`: waits for their launches, releases their slots, copies nothing — for hosts that unwind.  Stream INPUTS (`input` non-null)
 * are copied host-to-device by `begin` itself on the ctx stream, behind the previous ticket's render: from pageable memory that copy
 * is synchronous to the host, so a graph with stream inputs gets the output-side overlap only (hand device-resident inputs to
 * fwgpu_process_blocks_device_io for the rest).  Audio-side calls, like process_interleaved itself. */
int64_t fwgpu_process_interleaved_begin(fwgpu_ctx* ctx, const float* input, uint32_t num_in_channels, uint32_t num_out_channels,
                                        uint64_t frames, double stream_time_secs, uint32_t stream_status);
int fwgpu_process_interleaved_end(fwgpu_ctx* ctx, int64_t ticket, float* output);
int fwgpu_process_interleaved_cancel(fwgpu_ctx* ctx, int64_t ticket);
/* Throughput form of the same call: `num_blocks` full blocks, interleaved output written to DEVICE memory
 * `d_output` [num_blocks*max_block_frames*num_out_channels] on the ctx stream, asynchronously (no host
 * sync).  Graph inputs read zeros (fwgpu_process_blocks_device_io takes them from device memory). */
int fwgpu_process_blocks_device(fwgpu_ctx* ctx, uint32_t num_blocks, float* d_output,
                                uint32_t num_out_channels);
/* The same call, also reporting the silence mask read_graph_outputs sees (graph/graph/compiler/schedule.rs:255-287) for every
 * block: d_silence[block * num_out_channels + c] = 1 when graph-output channel c was flagged silent for that block, else 0
 * (device memory, num_blocks * num_out_channels bytes; may be NULL).  These are the flags a shard's partial mix bus carries
 * into the top-level SumNode of a voice-sharded graph (below). */
int fwgpu_process_blocks_device_flags(fwgpu_ctx* ctx, uint32_t num_blocks, float* d_output, uint32_t num_out_channels,
                                      uint8_t* d_silence);
/* ... for graphs WITH stream inputs (an effects rack, a send bus fed from outside): `d_input` = num_blocks * max_block_frames *
 * num_in_channels interleaved frames in DEVICE memory, read on the ctx stream (prepare_graph_inputs + deinterleave,
 * schedule.rs:213-253, util.rs:44-87: channels beyond num_graph_inputs are ignored, missing ones read zeros); NULL / 0 channels = no
 * input.  Asynchronous like the two calls above; the caller keeps `d_input` alive until the stream has passed it. */
int fwgpu_process_blocks_device_io(fwgpu_ctx* ctx, uint32_t num_blocks, const float* d_input, uint32_t num_in_channels, float* d_output,
                                   uint32_t num_out_channels, uint8_t* d_silence);

/* ---- multi-GPU mix bus.  Voices shard across ranks with no exchange until the mix bus; the one exchange step
 * is the top-level SumNode over the R partial buses (nodes/sum.rs:41-136: all inputs silent -> cleared; 2 / 3 / 4 ports ->
 * in1 + in2 (+ in3 (+ in4)); any other count -> out = in0; out += in_p skipping SILENT ports; port = rank order). */
/* That node as one kernel on the ctx stream: d_parts[r] = rank r's interleaved bus of `n_floats` floats in memory this
 * device can read (its own, peer-mapped over xGMI, or the slots of an all-gather), 16-byte aligned; d_out may alias
 * d_parts[0].  Every rank that runs it over the same parts ends up with the bits of the single-process graph (an all-reduce
 * does not: ring order re-associates the f32 sum for R > 2).  Asynchronous; R <= 64.  This form treats no port as silent. */
int fwgpu_bus_sum_ordered(fwgpu_ctx* ctx, const float* const* d_parts, uint32_t n_parts, float* d_out, uint64_t n_floats);
/* ... with the ports' silence flags (sum.rs:52-56,122-124): d_silence[r] = rank r's flags as fwgpu_process_blocks_device_flags
 * wrote them ([blocks][n_channels], or NULL = never silent); the buses are [block][frames_per_block][n_channels] interleaved.
 * d_out_silence (may be NULL) receives the node's out-mask per (block, channel). */
int fwgpu_bus_sum_ordered_flags(fwgpu_ctx* ctx, const float* const* d_parts, const uint8_t* const* d_silence, uint32_t n_parts,
                                float* d_out, uint8_t* d_out_silence, uint64_t n_floats, uint32_t frames_per_block,
                                uint32_t n_channels);
/* The exchange itself, for a host without torch / RCCL (SURVEY §8e path 2: one-shot all-to-all over peer-mapped slots — xGMI
 * is point-to-point, every GPU of a node has a direct link to every other, so each rank STORES its partial bus into its slot
 * on every rank and then each rank adds the R slots it holds in rank order: one link hop instead of the 2(R-1) steps of a
 * ring, and bit-identical to the single-process graph on every rank).
 *   open     (control side) one region of fine-grained HBM on the ctx's device: R slots of max_floats floats + max_silence_bytes
 *            flags, twice (step parity), and R arrival words.  Every rank of an exchange passes the same world and sizes.
 *   export   this rank's FWGPU_EXCHANGE_HANDLE_BYTES-byte handle; the host carries it to the peers by whatever channel it
 *            has (a file, a pipe, MPI_Allgather, torch.distributed.all_gather_object ...).
 *   connect  map a peer's region from its handle: by pointer when the peer lives in this process (virtual shards, several
 *            devices under one host thread: hipDeviceEnablePeerAccess), else hipIpcOpenMemHandle (dmabuf).
 *   push     (audio side, asynchronous on the ctx stream) store d_partial / d_silence into slot `rank` of every region,
 *            release at system scope, raise this rank's arrival word everywhere.
 *   reduce   (audio side, asynchronous) wait ON THE DEVICE for all R arrival words of this step, then the SumNode above
 *            over the R slots -> d_out (+ d_out_silence).  The wait is bounded (set_timeout_ms, default 3000): a peer that
 *            never arrives leaves a ZERO bus and an error fwgpu_bus_exchange_status reports — never a hung device.
 *   step     = push + reduce.  A step of rank g needs every rank's push of the same step: all ranks call step the same
 *            number of times; push and reduce of one exchange must stay on its ctx's stream, in that order (two data
 *            parities make push(s+1) safe while peers still read step s).
 * A reduce with have_silence = 0 treats no port as silent. */
#define FWGPU_EXCHANGE_HANDLE_BYTES 128
typedef struct fwgpu_bus_exchange fwgpu_bus_exchange;
fwgpu_bus_exchange* fwgpu_bus_exchange_open(fwgpu_ctx* ctx, uint32_t rank, uint32_t world, uint64_t max_floats,
                                            uint32_t max_silence_bytes); /* NULL on error: fwgpu_last_error(ctx) */
void fwgpu_bus_exchange_close(fwgpu_bus_exchange* ex);
int fwgpu_bus_exchange_export(fwgpu_bus_exchange* ex, void* handle);
int fwgpu_bus_exchange_connect(fwgpu_bus_exchange* ex, uint32_t peer_rank, const void* handle);
int fwgpu_bus_exchange_set_timeout_ms(fwgpu_bus_exchange* ex, uint32_t ms);
int fwgpu_bus_exchange_push(fwgpu_bus_exchange* ex, const float* d_partial, const uint8_t* d_silence, uint64_t n_floats,
                            uint32_t n_blocks, uint32_t n_channels);
int fwgpu_bus_exchange_reduce(fwgpu_bus_exchange* ex, float* d_out, uint8_t* d_out_silence, uint64_t n_floats, uint32_t n_blocks,
                              uint32_t frames_per_block, uint32_t n_channels, int have_silence);
int fwgpu_bus_exchange_step(fwgpu_bus_exchange* ex, const float* d_partial, const uint8_t* d_silence, float* d_out,
                            uint8_t* d_out_silence, uint64_t n_floats, uint32_t n_blocks, uint32_t frames_per_block,
                            uint32_t n_channels);
/* control side: waits for the ctx stream, then *steps = reduces issued so far, *failed_step = 0 or the first step whose wait
 * ran out of time (then the return value is FWGPU_ERR_DEVICE). */
int fwgpu_bus_exchange_status(fwgpu_bus_exchange* ex, uint64_t* steps, uint64_t* failed_step);
/* control side: waits for the ctx stream; max_wait_us[p] = the longest any reduce so far sat waiting for rank p's arrival, in
 * microseconds (how far the ranks run apart: the slowest rank reads ~0 for everybody, the others read their lead over it).
 * Returns the world size; reset != 0 clears the maxima. */
int fwgpu_bus_exchange_wait_stats(fwgpu_bus_exchange* ex, uint64_t* max_wait_us, uint32_t cap, int reset);
/* ---- the mix bus over RCCL (north_star: "a single RCCL all-reduce over xGMI for the final mix bus"; the node being computed is the
 * top-level R-port SumNode, nodes/sum.rs:111-133).  librccl is dlopen'ed by the first of these calls, never linked: a host that does
 * not shard does not load it.  One process per GPU; rank 0 makes the unique id, the host carries its FWGPU_RCCL_UNIQUE_ID_BYTES bytes
 * to every rank (the side channel that carries the exchange's handles), every rank creates its communicator on its ctx's device.
 *   unique_id          rank 0: ncclGetUniqueId.
 *   comm_create        collective (ncclCommInitRank): every rank of the id calls it; NULL on error (fwgpu_last_error of the ctx).
 *                      Control side; the communicator belongs to the ctx and must be destroyed before it.
 *   allreduce_rccl     ncclAllReduce(sum) IN PLACE on the ctx stream, asynchronous.  The ring re-associates the f32 sum for more
 *                      than two ranks: within 1e-6 relative of the single-process graph, not its bits; silence flags play no part
 *                      (a silent shard's bus is cleared zeros, and x + 0 = x).
 *   allgather_ordered  ncclAllGather of the partial buses and of their per-(block, channel) silence flags (as
 *                      fwgpu_process_blocks_device_flags wrote them; NULL = never silent), then fwgpu_bus_sum_ordered_flags over
 *                      the gathered slots: sum.rs's port order and silent-port rule, bit-identical to the single-process graph on
 *                      every rank.  n_floats a multiple of 4; d_out 16-byte aligned, may alias d_bus; d_out_silence may be NULL.
 * Audio-side calls like the process calls they follow (same stream).  fwgpu_rccl_last_error: the text of the last failure of a
 * call that has no ctx to report through (unique_id). */
#define FWGPU_RCCL_UNIQUE_ID_BYTES 128
typedef struct fwgpu_rccl_comm fwgpu_rccl_comm;
int fwgpu_rccl_unique_id(uint8_t* id);
fwgpu_rccl_comm* fwgpu_rccl_comm_create(fwgpu_ctx* ctx, const uint8_t* id, uint32_t world, uint32_t rank);
int fwgpu_rccl_comm_destroy(fwgpu_rccl_comm* comm);
int fwgpu_rccl_comm_info(fwgpu_rccl_comm* comm, uint32_t* world, uint32_t* rank);
int fwgpu_bus_allreduce_rccl(fwgpu_rccl_comm* comm, float* d_bus, uint64_t n_floats);
int fwgpu_bus_allgather_ordered(fwgpu_rccl_comm* comm, const float* d_bus, const uint8_t* d_silence, float* d_out,
                                uint8_t* d_out_silence, uint64_t n_floats, uint32_t frames_per_block, uint32_t n_channels);
const char* fwgpu_rccl_last_error(void);
int fwgpu_synchronize(fwgpu_ctx* ctx);
/* ProcInfo::stream_time_secs / stream_status (core/node.rs:111-132) of the most recent fwgpu_process_interleaved call —
 * what a custom node run through fwgpu_node_process inside that call would be handed — and how often the backend has
 * reported StreamStatus::OUTPUT_UNDERFLOW (bit 1) / INPUT_OVERFLOW (bit 0) so far.  Any pointer may be NULL. */
int fwgpu_proc_info(fwgpu_ctx* ctx, double* stream_time_secs, uint32_t* stream_status, uint64_t* output_underflows,
                    uint64_t* input_overflows);

/* ---- headless stream: the reference's audio-backend callback without a device (firewheel-cpal/src/lib.rs:378-449
 * DataCallback::callback; its non-cpal "dummy backend" is todo!() at lib.rs:150,168,222).  The caller plays the backend:
 * it calls fwgpu_stream_callback once per device period with the instant of that callback on ITS clock (cpal's
 * info.timestamp().callback).  The stream-time and underflow bookkeeping is the reference's, line for line: the first
 * callback's instant is ignored (stream time 0), the second one anchors the clock, and from then on a callback that
 * arrives later than the previous one's stream time + 1.2 periods is an OUTPUT_UNDERFLOW, passed on as stream_status. */
typedef struct fwgpu_stream fwgpu_stream;
fwgpu_stream* fwgpu_stream_open(fwgpu_ctx* ctx, uint32_t num_in_channels, uint32_t num_out_channels);
void fwgpu_stream_close(fwgpu_stream* s);
/* returns the StreamStatus bits handed to process_interleaved (>= 0) or a negative error; `output` is always filled */
int fwgpu_stream_callback(fwgpu_stream* s, float* output, uint64_t frames, double callback_instant_secs);
int fwgpu_stream_stats(fwgpu_stream* s, uint64_t* callbacks, uint64_t* underflows, double* last_stream_time_secs);
/* The backend thread's loop without the device: `n_callbacks` callbacks of `frames` frames back to back, callback i at
 * instant first_instant_secs + i * frames / sample_rate — a stream that never underruns, driven as fast as the engine
 * answers.  `output` (frames x num_out_channels) is overwritten by every callback and holds the last block on return.
 * *elapsed_secs (may be NULL) = wall time of the loop on the monotonic clock: elapsed / n_callbacks is what one callback
 * costs the audio thread, with nothing of the caller's language runtime inside the loop.  Returns 0 or the first error. */
int fwgpu_stream_run(fwgpu_stream* s, float* output, uint64_t frames, uint32_t n_callbacks, double first_instant_secs,
                     double* elapsed_secs);

/* AudioNodeProcessor::process (core/node.rs:37-53) for ONE activated node on caller (host) buffers —
 * the literal per-node drop-in for graph/processor.rs:243.  inputs/outputs: arrays of `frames` floats.
 * out_silence_mask is pre-set by the caller (the reference passes 0). */
int fwgpu_node_process(fwgpu_ctx* ctx, int64_t node, uint64_t frames, const float* const* inputs,
                       uint32_t num_inputs, float* const* outputs, uint32_t num_outputs,
                       uint64_t in_silence_mask, uint64_t* out_silence_mask, double stream_time_secs,
                       uint32_t stream_status);

/* ---- measurement hooks (bench.py): HIP-event timing of the dominant kernel on the ctx stream */
int fwgpu_timing_enable(fwgpu_ctx* ctx, int on);
/* which: 0 = fused leaf kernel (k_leaf_sum / k_chain, dominant), 1 = control kernel, 2 = upper/out kernels,
 * 3 = all kernels of one generic-executor block, 4 = k_fir_gemm alone (FIR banks).
 * Returns accumulated milliseconds and launch count since the last reset. */
int fwgpu_timing_read(fwgpu_ctx* ctx, int which, double* total_ms, uint64_t* launches);
int fwgpu_timing_reset(fwgpu_ctx* ctx);
/* device facts for the bench line */
int fwgpu_device_info(fwgpu_ctx* ctx, char* name, int name_cap, int* compute_units, uint64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* FWGPU_H */
