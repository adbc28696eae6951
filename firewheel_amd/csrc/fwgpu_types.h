// fwgpu_types.h — POD structs shared by the host planner and the HIP kernels (HBM-resident layout).
#pragma once
#include <stdint.h>

namespace fwgpu {

enum : int {
    K_DUMMY = 0, K_BEEP = 1, K_VOLUME = 2, K_SUM = 3, K_SAMPLER = 4, K_HARD_CLIP = 5,
    K_MONO_TO_STEREO = 6, K_STEREO_TO_MONO = 7, K_PAN = 8, K_WIDTH = 9, K_BIQUAD = 10, K_DELAY = 11,
    K_FIR = 12, K_RESAMPLER = 13, K_SPATIAL = 14,
    K_HOST = 15,  // a node the library does not implement: the caller's own AudioNodeProcessor::process, run on the host
    K_LAST = K_HOST,
};
#define RS_PHASES 32  // SPEC resampler: polyphase windowed-sinc table [RS_PHASES][RS_TAPS], 32.32 fixed-point position
#define RS_TAPS 16
// realtime edge, resident kernel (k_rt_persist): the mailbox in pinned, device-mapped host memory
struct RtMailbox {
    unsigned long long doorbell;  // host -> device: the sequence number to render next, or that number | RT_QUIT_BIT
    unsigned long long hold;      // host (CONTROL side) -> device: > 0 while a control call is about to free device memory or wait for the
                                  // device (hipFree synchronises with every stream — a kernel that never ends would hold it forever):
                                  // the kernel ends as if its watchdog had fired, and the audio side launches no new one (RtHold)
    unsigned long long pad0[6];
    unsigned long long alive;     // device -> host: 1 while the kernel takes doorbells, 0 once it has decided to end
    unsigned long long pad1[7];
};
#define RT_QUIT_BIT (1ull << 63)
#define RT_PREFETCH_BIT (1ull << 62)  // device-internal (`go` word): "no doorbell for 10 us: fetch the next block's sources now"
#define LEAF_WPB_MAX 4  // k_leaf.hip.h: at most this many 256-frame pieces (waves) per block
#define SP_HIST 64    // SPEC spatialiser: mono history frames (>= the largest per-ear delay + 1)

enum : int { FMT_I_I16 = 0, FMT_I_U16 = 1, FMT_I_F32 = 2, FMT_P_I16 = 3, FMT_P_U16 = 4, FMT_P_F32 = 5 };

// core/param/smoother.rs:72-88 without the output Vec: whenever status != Active the reference's buffer
// is constant == input (every path that leaves Active fills it), so only the scalars are state.
enum : int { SM_INACTIVE = 0, SM_ACTIVE = 1, SM_DEACTIVATING = 2 };
struct Smoother {
    int status;
    float input;
    float last;  // last_output
    float a, b, eps;
};

// One state record per node (audio half of every node kind; 128 B, indexed by node slot).
struct NodeState {
    float p0, p1;  // the reference's atomics: VOLUME/SAMPLER raw_gain; PAN gl,gr targets; HARD_CLIP threshold
    Smoother s0, s1;
    // nodes/sampler.rs:280-292
    int playing;
    int has_loop;
    int full_range;
    int sample;  // sample-table index, -1 = None
    uint64_t playhead;
    uint64_t loop_start, loop_end;
    // nodes/beep_test.rs:64-69
    float phasor, phasor_inc, gain;
    int enabled;
    uint32_t sample_rate;
    // SPEC nodes with per-channel state: offset/length (floats) of this node's slice of the ext pool
    //   BIQUAD: ext = [b0 b1 b2 a1 a2][x1 x2 y1 y2] x channels      DELAY: ext = ring[channels][D]
    //   DELAY also uses p0 = feedback, p1 = mix, gain = dry (1-mix), playhead = ring position, loop_end = D
    //   FIR: ext = mirrored history ring[channels][2R]; playhead = ring position, loop_end = R, loop_start = T,
    //        sample = impulse-response sample id
    //   RESAMPLER: sample = source, playhead = 32.32 source position, loop_start = 32.32 step, has_loop, playing
    //   SPATIAL: p0/p1 = ear gain targets (s0/s1 smooth them), playing = left-ear delay, has_loop = right-ear delay
    //        (frames), ext = the last SP_HIST mono samples
    uint32_t ext_off;
    uint32_t ext_len;
    int pad[1];
};
static_assert(sizeof(NodeState) == 128, "NodeState layout");

struct SampleDesc {  // core/sample_resource.rs:4-26; data is HBM-resident
    const void* data;
    uint64_t frames;
    int channels;
    int format;
};

// Static description of one scheduled node (graph/graph/compiler/schedule.rs:12-20 after buffer renaming).
struct NodeDesc {
    int kind;
    int n_in, n_out;
    int in_off, out_off;  // offsets into the port tables
    int state;            // NodeState index
    int aux0;             // SUM: num_in_ports (low 16 bits); high 16 bits, when set: the port count whose path the node takes
                          //      (sum.rs:67-133: 2 / 3 / 4 ports add unmasked, any other count skips silent ports) — the
                          //      continuation of a SumNode whose leading voice ports were summed by the voice-bank kernels
    int is_graph_io;      // 1 = graph_in, 2 = graph_out (Dummy nodes the executor treats as I/O edges)
};

// control -> audio messages (nodes/sampler.rs:21-28 + the atomics), sorted by (state, block, seq).
enum : int {
    CMD_SET_P0 = 0, CMD_SET_P1 = 1, CMD_SET_ENABLED = 2, CMD_SET_GAIN = 3,
    CMD_SET_COEFS = 4,  // biquad: f0,i0,i1 (as float bits) = b0,b1,b2; d0 bits = (a1,a2)
    CMD_SMP_SET_SAMPLE = 10, CMD_SMP_PLAY = 11, CMD_SMP_PAUSE = 12, CMD_SMP_STOP = 13,
    CMD_SMP_SET_PLAYHEAD = 14, CMD_SMP_SET_LOOP = 15,
    CMD_RS_STEP = 20,  // resampler: d0 bits = u64 32.32 step
    CMD_RS_SEEK = 21,  // resampler: d0 bits = u64 source frame
    CMD_SP_ITD = 22,   // spatialiser: i0 / i1 = left / right ear delay in frames
};
struct Cmd {
    int state;
    uint32_t block;
    int type;
    int i0;
    float f0;
    int i1;
    double d0, d1;
};
static_assert(sizeof(Cmd) == 40, "Cmd layout");

// ---------------------------------------------------------------- fused voice-bank plan
#define FW_MAX_STAGES 6    // sampler gain + up to 5 chain nodes (volume / pan / width / hard clip) in any order
#define FW_CHAIN_STAGES 4  // what k_chain keeps in registers: sampler gain + up to 3 gain stages behind the biquad / delay
// what a chain stage does to the voice's two channels (4 bits per stage in FusedView::progs[voice], stage 1 in bits 0..3);
// the stage's values sit in the gain set: g[j][0], g[j][1]
enum : uint32_t {
    SK_GAIN = 0,   // volume / pan:  L *= g0;  R *= g1                                   (volume.rs:123-126)
    SK_WIDTH = 1,  // stereo width:  m = (L+R)*0.5; s = ((L-R)*0.5)*g0;  L = m+s; R = m-s (SPEC)
    SK_CLIP = 2,   // hard clip:     x = max(min(x, g0), -g0) on both channels            (hard_clip.rs:70-76)
    SK_SPATIAL = 3,  // 3D spatialiser (SPEC), LAST stage only: m = (L+R)*0.5; L = m[i-dL]*g0; R = m[i-dR]*g1 with the
                     // per-ear delays of the block's record (VB_SP_SHIFT) and the 64-frame mono history of the block before
};

// static per voice chain.  Voice-bank plan: source -> [volume|pan|width|hard clip|spatialiser]* -> leaf sum port.  Chain plan (round 6
// grammar): sampler -> G* -> F1 [-> G* -> F2 [-> G* -> F3]] -> G* -> leaf sum port, G = volume | pan | hard clip (<= 3 in all), F1 F2 F3 one of
//   biquad, biquad biquad, delay, biquad delay, biquad biquad delay  (fx_order 0: filters first)   or
//   delay biquad, delay biquad biquad                                (fx_order 1: the delay line first)
struct VoiceDesc {
    int sampler_state;
    int n_stages;                      // gain-like stages of the chain, in schedule order: the first n_pre sit between the source and the
    int stage_kind[FW_MAX_STAGES - 1]; //   biquad / delay (they see the source's silence flag), the rest behind them (they never see one)
    int stage_state[FW_MAX_STAGES - 1];
    int bq_state;                      // the (first) biquad of the chain, -1 = none (k_chain plan)
    int dl_state;                      // the delay line, -1 = none
    int src_kind;                      // 0 = SamplerNode, 1 = SPEC resampling source (sampler_state = its state; no gain of its own),
                                       // 2 = a ONE-output SamplerNode behind a MonoToStereoNode: channel 0 on both outputs (round 5)
    int sp_ext_off;                    // a SPEC spatialiser as the last stage: ext-pool offset of its SP_HIST-frame mono history; -1 = none
    int n_pre;                         // chain plan: gain stages in FRONT of the first filter (0 for a dry voice)
    int bq2_state;                     // chain plan: a second biquad behind the first (an EQ cascade), -1 = none
    int fx_order;                      // chain plan: 0 = biquad(s) then delay, 1 = delay then biquad(s)
    int n_mid;                         // chain plan: gain stages BETWEEN the filters: bits 0..7 between the first and the second, 8..15 between
                                       //   the second and the third (the stages behind n_pre, in schedule order)
};
static_assert(sizeof(VoiceDesc) == 80, "VoiceDesc layout");

// per (block, voice) record written by the control kernels, read by the leaf kernel (80 B)
enum : uint32_t {
    VB_SILENT = 1u,       // chain output is cleared + flagged silent for this block
    VB_WRAP = 2u,         // loop wrap inside the block: frames [n1, frames) come from off1
    VB_TAIL_ZERO = 4u,    // one-shot end inside the block: frames [n1, frames) are 0.0
    VB_MONO = 8u,         // 1-channel sample duplicated to both outputs (sampler.rs:546-551)
    VB_SIMPLE = 16u,      // contiguous planar-f32 source + constant gains: src_l/src_r valid, fast path
                          //   (k_chain plan: also a VB_SRC_ZERO block with constant gains; no full VoiceBlk either way)
    VB_SRC_ZERO = 32u,    // the sampler's output is cleared this block (differs from VB_SILENT only when a biquad /
                          //   delay sits between the sampler and the gain stages: their tails keep ringing)
    VB_RESAMPLE = 64u,    // resampling source (SPEC): off0 = 32.32 position of frame 0, off1 = 32.32 step, n1 = loops;
                          //   every such block carries a full VoiceBlk (the 16-tap polyphase fetch is the leaf kernel's slow path)
    VB_RS_LEAN = 128u,    // a resampler block of a STEADY voice (VoiceRef::flags_gset only; round 4): its full descriptor is the voice's
                          //   template FusedView::rs_tmpl[voice] — written once per call — with off0 = the 32.32 position carried in
                          //   VoiceRef::src_l; no VoiceBlk row is written for it (they were 75 MB per 768-block call of 1 024 voices,
                          //   and what the control kernel's 42 us went into)
    VB_FMT_SHIFT = 24,    // VB_RESAMPLE blocks: bits 24..26 = the sample's format (FMT_*), src_l = its data, pad = its frames (< 2^31)
                          //   — the leaf kernel needs no second dependent load for the sample table
    VB_RAMP_SHIFT = 8,    // bit (VB_RAMP_SHIFT + 2*stage + ch): that gain is a per-frame ramp (12 bits: 8..19)
    VB_RAMP_MASK = 0xfffu << 8,
    VB_SP_SHIFT = 20,     // voices whose last stage is a spatialiser: bits 20..25 = left-ear delay, 26..31 = right-ear delay (frames,
                          //   <= 63) in force for this block — in VoiceBlk::flags AND VoiceRef::flags_gset (such a voice's source is
                          //   never a resampler, so the VB_FMT bits are free)
    VB_SP_MASK = 0xfffu << 20,
};
struct VoiceBlk {
    uint32_t flags;
    uint32_t n1;         // frames taken from off0
    const float* src_l;  // frame 0 of channel 0 / 1 when the block's frames are contiguous planar f32
    const float* src_r;
    uint64_t off0;       // source frame of frame 0
    uint64_t off1;       // source frame of frame n1 when VB_WRAP
    int sample;          // sample-table index
    uint32_t pad;
    float g[FW_MAX_STAGES][2];  // constant gains per stage and channel (used when the ramp bit is clear)
};
static_assert(sizeof(VoiceBlk) == 96, "VoiceBlk layout");

// Compact per (block, voice) record (16 B): all the leaf kernel needs for silent and VB_SIMPLE blocks.  Only
// blocks that are neither (ramps, loop wrap, one-shot tail, non-planar-f32 sources) also get a full VoiceBlk.
// VB_SIMPLE source classes (bits 16..18 of VoiceRef::flags_gset): how the leaf kernel fetches 4 consecutive frames
enum : uint32_t {
    SF_P_F32 = 0,  // planar f32 (also interleaved mono): dwordx4 per channel
    SF_P_I16 = 1,  // planar i16 (also interleaved mono), 4-byte aligned: dwordx2 per channel
    SF_P_U16 = 2,
    SF_I_I16 = 3,  // interleaved stereo i16: ONE dwordx4 holds both channels of 4 frames
    SF_I_U16 = 4,
    SF_I_F32 = 5,  // interleaved stereo f32: two dwordx4
    SF_NONE = 7,   // not eligible for the compact fast path
};
struct VoiceRef {
    const float* src_l;  // VB_SIMPLE: address of channel 0 of the block's first frame (typed by the source class)
    uint32_t r_delta;    // VB_SIMPLE: channel-1 offset in source ELEMENTS (0 for a mono sample)
    uint32_t flags_gset; // bits 0..7 VB_* flags, bits 8..15 gain-set index, bits 16..18 source class (SF_*)
};
static_assert(sizeof(VoiceRef) == 16, "VoiceRef layout");
#define FW_GSETS 4  // distinct constant-gain sets a voice may use inside one call before falling back to VoiceBlk
struct GainSet {
    float g[FW_MAX_STAGES][2];
};

// Per voice, across calls: "this voice ended the last call steady" + the descriptor its blocks share.
// Valid only while epoch == FusedView::epoch (the host bumps it on every plan install / sample-table change).
struct VoiceCache {
    uint32_t epoch;
    int mode;
    uint32_t flags;
    int sample;
    GainSet g;
};
static_assert(sizeof(VoiceCache) == 64, "VoiceCache layout");

// (host side, here so that the test harness can reach it) quiet_window's look-ahead rule, no call being in flight at `now` (all steady_clock ns): the last call began at `start`, `period`
// after the one before, and took `dur`.  true: the next call is due within `margin` — wait for it to come and go.  false: go now
// (room before the next call; or no rhythm known; or a stream without gaps of 2 x margin to use; or the call is overdue by more
// than the margin: a stream that stopped must not hold a build up).
inline bool quiet_next_call_is_due(uint64_t now, uint64_t start, uint64_t period, uint64_t dur, uint64_t margin) {
    if (!period || period > 200000000ull || period < dur + 2 * margin) return false;
    const uint64_t since = now - start;
    return !(since + margin < period || since > period + margin);
}
// plan build: one piece of the build's device work (k_build_apply).  src != nullptr: copy row_bytes from pinned host memory
// (rows = 1); src == nullptr: rows x row_bytes at `pitch` bytes set to `value`, byte 0 of every row to `head` if head >= 0
struct BuildJob {
    void* dst;
    const void* src;
    unsigned long long row_bytes, pitch;
    uint32_t rows, value;
    int head, pad_;
};
static_assert(sizeof(BuildJob) == 48, "BuildJob layout");

// plan adoption: which steady caches travel from the old plan to the new one (k_adopt_init / carry_cache_voice); n_new = 0: none
// Round 4 — "lazy" records.  A voice that ends a call STEADY (constant gains, nothing but the playhead moving) and whose blocks
// are all plain ones (silent, or a contiguous planar-f32 source that never wraps inside a block) has records that are a pure
// function of (this record, block index): the control kernel writes one of these per voice at the end of a call, and the NEXT
// calls — as long as they carry no message, the plan has not changed and the host has SEEN that every voice of the plan left
// one behind (FusedView::horizon -> pinned memory) — are rendered without a control kernel at all: the leaf wave's lane p loads
// port p's LazyRec (descriptor AND gains: one round trip where records + gain sets were two) and computes the block's source
// address itself.  Node state is brought up to date (k_lazy_flush) before anything else reads it.
struct LazyRec {
    uint64_t base;        // BYTE address of source frame 0 of the record: mode 1: the loop start; mode 2: the sample's start; mode 0: unused
    uint64_t off0;        // mode 2: playhead (frames) at the record's block 0
    uint64_t loop_start;  // mode 1 (k_lazy_flush rebuilds the playhead from it)
    uint32_t r_delta;     // the record's r_delta (channel-1 offset in elements; 0 = mono)
    uint32_t flags_gset;  // the record's flags_gset (gain-set index bits unused: the gains are `g` below)
    uint32_t q;           // mode 1: loop length in BLOCKS
    uint32_t r0b;         // mode 1: playhead offset from the loop start at the record's block 0, in blocks
    uint32_t frames;      // block size the block counts refer to
    int mode;             // 0 nothing moves, 1 looping playhead, 2 one-shot playhead (TailJob::mode); -1 = not lazy-capable
    int sampler_state;    // the voice's sampler state slot (-1: a null voice)
    uint32_t bpf;         // bytes per source frame at `base` (the record's source class: planar f32 4, planar 16-bit 2, interleaved
                          // stereo 16-bit 4, interleaved stereo f32 8)
    uint32_t pad[2];
    GainSet g;
    uint32_t pad2[4];
};
static_assert(sizeof(LazyRec) == 128, "LazyRec layout");

struct VoiceDesc;
struct CarryArgs {
    VoiceCache* new_cache;
    const VoiceDesc* new_voices;
    int n_new;
    const VoiceCache* old_cache;
    const VoiceDesc* old_voices;
    const int* old_slot_voice;
    int n_old_slots;
    uint32_t old_epoch, new_epoch;
};

// ---------------------------------------------------------------- FIR convolution bank (MFMA GEMM)
#define FIR_SEG 4096  // window positions per split-K segment — part of the numeric SPEC (summation order)
#ifndef FIR_KC
#define FIR_KC 128    // window positions staged per LDS chunk (64 or 128; not part of the numeric SPEC)
#endif
struct FirRow {  // one GEMM row = one channel of one FIR node
    int state;
    int ch;
    int in_buf;
    int out_buf;
};

#define CH_FAST_KMAX 64  // blocks per k_chain launch its steady-call loop keeps per-block source addresses for (LDS);
                         // the host never batches more blocks than this into one chain-plan launch

// the root SumNode of a fused plan, passed to k_root_out by value (kernel arguments: scalar loads, no dependent fetch)
struct RootArgs {
    int n_in, ports;  // ports = n_in / 2 stereo ports (<= 32)
    int in_buf[64];   // bus buffer of input channel i
    const int* in_tab;  // the same table in device memory (for per-lane indexing)
};

// k_chain plan: what the two channel workgroups of a voice share, as it stood at the START of the current call
// (written by k_voice_control, which also advances the node state to the end of the call; read-only for k_chain)
struct ChainStart {
    uint32_t pos;       // delay ring position
    float fb, mix, dry; // delay feedback / wet / dry
    float co[5];        // biquad b0 b1 b2 a1 a2
    float co2[5];       // the second biquad's (VoiceDesc::bq2_state)
    uint32_t pad[2];
};
static_assert(sizeof(ChainStart) == 64, "ChainStart layout");

// the top-level SumNode over the partial mix buses of R voice shards (nodes/sum.rs:111-133), passed by value
#define FW_MAX_BUS_PARTS 64
struct BusParts {
    int n;
    const float* part[FW_MAX_BUS_PARTS];  // part[r] = rank r's interleaved bus (peer-mapped or all-gathered), same length each
};

// one-shot mix-bus exchange over peer-mapped slots (k_exchange.hip.h): what every rank of an exchange agrees on ...
struct ExchangeGeom {
    int world, rank;
    uint64_t max_floats;  // floats of one slot's bus
    uint64_t slot_bytes;  // max_floats * 4 + silence bytes, padded to 256
};
// ... and where each rank's region is mapped in THIS process (base[rank] = its own), passed by value
struct ExchangePeers {
    char* base[FW_MAX_BUS_PARTS];
};

#define CH_GROUP_LEAVES 8
// k_chain workgroup = up to CH_GROUP_LEAVES consecutive leaf SumNodes with at most 32 voices together (their voices are
// consecutive): a tree of small leaves (a bus per instrument) fills the workgroup's 32 voice rows like one wide leaf
struct ChainGroup {
    int first_voice, n_voices;
    int n_leaves;
    int uniform_ports;     // 32 / 16 / 8 / 4: the 32 rows are full leaves of exactly that many ports; 0 otherwise
    uint32_t start_mask;   // bit r: voice row r is port 0 of a leaf
    uint32_t masked_rows;  // bit r: row r belongs to a leaf on the n-port path (ports not 2, 3 or 4): silent ports skipped
    int out_buf[CH_GROUP_LEAVES];  // per leaf: compact bus-buffer id of output channel 0
    int row0[CH_GROUP_LEAVES];     // per leaf: its first voice row
    int ports[CH_GROUP_LEAVES];
};

struct LeafDesc {  // a SumNode whose ports are all voice chains (nodes/sum.rs)
    int first_voice;
    int ports;     // num_in_ports
    int out_buf;   // compact bus-buffer id of output channel 0 (channel 1 = +1)
    int pad;       // when set: the port count of the WHOLE SumNode (this leaf is its leading voice ports): decides the
                   // masked / unmasked path (Q13) instead of `ports`
};

}  // namespace fwgpu
