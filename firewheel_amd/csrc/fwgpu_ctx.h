// fwgpu_ctx.h — internal to libfwgpu's host half: the context object and what its translation units share.
//   fwgpu_control_math.cpp  control-half scalar math (what the reference's control thread computes) + initial node state
//   fwgpu_plan_detect.cpp   launch-plan selection: does the compiled schedule match a fused plan?
//   fwgpu_plan_install.cpp  node activation + upload of the launch plan's device tables
//   fwgpu_run.cpp           per-call work: message upload / retirement, kernel sequences of each plan, timing
//   fwgpu_abi.cpp           the C ABI of include/fwgpu.h
#pragma once
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/fwgpu.h"
#include "fwgpu_graph.h"
#include "fwgpu_launch.h"
#include "fwgpu_msgq.h"

namespace fwgpu {


struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // host copy of what a plan build last uploaded here (fwgpu_plan_install.cpp, up()): only for tables the device never writes;
    // the next build of this image uploads the 4 KiB chunks that differ.  Dropped whenever the device memory is.
    std::vector<uint8_t> shadow;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap), shadow(std::move(o.shadow)) {
        o.p = nullptr;
        o.cap = 0;
        o.shadow.clear();
    }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            cap = o.cap;
            shadow = std::move(o.shadow);
            o.p = nullptr;
            o.cap = 0;
            o.shadow.clear();
        }
        return *this;
    }
    ~DevBuf() { release(); }  // whatever fwgpu_ctx_destroy's list misses still goes with the ctx
    // FWGPU_POISON=1 (tests): fresh device memory is NOT zero in general (a long-lived process hands out what earlier contexts
    // left behind) — fill new allocations with a byte pattern so that a read of something nobody wrote shows up at once.
    // FWGPU_POISON_ONLY=<name>: only the buffers of that name (bisecting which table is read before it is written).
    hipError_t ensure_n(const char* name, size_t bytes) {
        const void* before = p;
        const size_t cap0 = cap;
        hipError_t e = ensure(bytes);
        static const int poison = getenv("FWGPU_POISON") ? atoi(getenv("FWGPU_POISON")) : 0;
        static const char* only = getenv("FWGPU_POISON_ONLY");
        // FWGPU_POISON=2: the audio buffers a build no longer clears — the leaf / mix buses and the level executor's pool —
        // are filled with NaNs at EVERY build, recycled images included: a kernel path that skips a write before a reader (an
        // early-out leaf, an unmasked 2/3/4-port sum, a steady chain group) then mixes NaNs into the output instead of stale audio
        // that may happen to be right.  (The image being built is not the active one: nothing reads it meanwhile.)
        const bool audio_buf = !strcmp(name, "d_bus") || !strcmp(name, "d_pool");
        if (e == hipSuccess && poison && ((p != before || cap != cap0) || (poison >= 2 && audio_buf)) && (!only || !only[0] || !strcmp(only, name))) {
            e = hipMemset(p, (poison >= 2 && audio_buf) ? 0xFF : 0xCB, cap);
            if (e == hipSuccess) e = hipDeviceSynchronize();  // (the fill is in place before any stream writes real data)
        }
        return e;
    }
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        const bool regrow = p != nullptr;  // a buffer that grows once tends to grow again (graph edits add a few nodes
                                           // at a time; a 2 GB pool costs ~250 ms to free + allocate): leave headroom
        shadow.clear();  // (fresh memory holds none of it)
        if (p) {
            hipError_t e = hipFree(p);
            if (e != hipSuccess) return e;
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes < 256 ? 256 : bytes;
        // (round 6: tables up to 16 MB get an eighth of headroom from the start — slot-indexed tables grow by a few entries with every
        //  one of a graph's first edits (a removed node's slot is reusable only when no plan holds it), and a regrow is a hipFree +
        //  hipMalloc on the control thread: the free waits for every stream, and the callback that runs beside it took +35 us in
        //  fw_edit_race's paced runs — one or two edits of thirty, scripts/r06_edit_paced_ab.sh.  Pools keep their exact size.)
        if (!regrow && want <= ((size_t)16 << 20)) want += want / 8 > 4096 ? want / 8 : 4096;
        hipError_t e = hipErrorOutOfMemory;
        if (regrow) {
            const size_t roomy = want + want / 4;
            e = hipMalloc(&p, roomy);
            if (e == hipSuccess) cap = roomy;
            else (void)hipGetLastError();
        }
        if (e != hipSuccess) {
            e = hipMalloc(&p, want);
            if (e == hipSuccess) cap = want;
        }
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        shadow.clear();
    }
    template <class T>
    T* as() const { return (T*)p; }
};

struct SampleRec {
    bool alive = false;
    bool owned = true;
    void* d_data = nullptr;
    SampleDesc desc{};
};

constexpr size_t RT_IO_BYTES = 256 * 1024;  // realtime path: calls whose interleaved in/out blocks fit (e.g. 16 x 1024 stereo)

// which instantiation of the node kernel runs a kind (mirrors kind_set in k_generic.hip.h)
inline int host_kind_set(int kind) {
    if (kind == K_SAMPLER) return 2;
    return (kind == K_BEEP || kind == K_BIQUAD || kind == K_DELAY || kind == K_RESAMPLER || kind == K_SPATIAL) ? 1 : 0;
}
// a level's launch bits: the kernel set of the kind, plus bit 3 for a biquad / delay (a bus one goes to the batch walkers)
inline int host_kind_bits(int kind) { return (1 << host_kind_set(kind)) | ((kind == K_BIQUAD || kind == K_DELAY) ? 8 : 0); }

// A statistics counter the audio thread bumps and the control thread reads (fwgpu_rt_path_stats, fwgpu_lazy_stats,
// fwgpu_rt_resident_stats): relaxed atomics — a plain uint64_t there is a data race by the letter (the race test runs under
// ThreadSanitizer).  One writer, so ++ is a load + store, not a locked read-modify-write.
struct StatCounter {
    std::atomic<uint64_t> v{0};
    void operator++(int) { v.store(v.load(std::memory_order_relaxed) + 1, std::memory_order_relaxed); }
    operator uint64_t() const { return v.load(std::memory_order_relaxed); }
};

struct TimerCat {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    double acc_ms = 0.0;
    uint64_t launches = 0;
};

}  // namespace fwgpu

using namespace fwgpu;  // (internal header: only libfwgpu's own host translation units include it)

namespace fwgpu {

// ---------------------------------------------------------------------------------------------------------------------
// PlanImage: everything a launch plan IS on the device and on the host — the tables install_plan builds and the process calls
// read.  A ctx holds the ACTIVE image as its base class (so `c->d_nodes` is the active plan's node table); fwgpu_update /
// fwgpu_schedule_upload build the NEXT image off to the side on the control thread, while process calls keep running on the
// active one, and publish it; it is adopted — a swap of this struct's members, no allocation, no wait — at the start of the
// next process call (graph/processor.rs:167-206: "NewSchedule" applied at block start), the old image goes back to the control
// side through a ring and is reused as the next build target (graph/context.rs:93-137).
struct PlanImage {
    Plan plan;
    bool have_plan = false;
    uint64_t gen = 0;        // images are numbered in build order
    uint32_t kmax = 64;      // blocks per launch this image's buffers are sized for

    // generic plan
    DevBuf d_nodes, d_in_buf, d_out_buf, d_level_nodes, d_pool, d_flags, d_gin_bufs, d_gout_bufs;
    std::vector<int> level_off, level_cnt;
    std::vector<int> level_kinds;  // bit s: the level holds node kinds of kernel set s (host_kind_set)
    int n_gout_bufs = 0, n_gin_bufs = 0;
    std::vector<int> slot_index;   // node slot -> index into plan.nodes, -1 = not in this plan (B1: fwgpu_node_process)

    // fused plan
    bool fused = false;
    uint32_t generic_k = 1;  // blocks per generic-executor batch: kmax, capped by what the FIR history rings hold
    bool fused_fx = false;  // the fused plan's leaves run k_chain (biquad / delay in the voice chains)
    int chain_nq = 1;       // k_chain tile size / 64 frames (bits 0..1); bit 2: some voice holds two biquads (the instantiation with a second recurrence stage); bit 3: some voice has a gain stage between two filters or a hard clip
    int n_voices = 0, n_leaves = 0, n_bus = 1, ramp_slots = 0;
    int n_groups = 0;  // k_chain workgroups (groups of consecutive leaves)
    DevBuf d_groups;
    bool ctl_ahead_on = false;       // control-ahead mode: the installed plan qualifies
    DevBuf d_blks2, d_refs2, d_gsets2, d_ramps2;
    bool fused_rs = false;    // the plan has resampler-sourced voices
    bool fused_prog = false;  // the voice-bank plan carries stage programs (k_leaf_sum<true>)
    bool fused_sp = false;    // ... and spatialiser stages (k_leaf_sum_sp)
    // control-kernel dispatch order (FusedView::ctl_order), rebuilt by upload_cmds for every call that has messages
    std::vector<int> slot_voice;        // node state slot -> voice of the voice-bank plan (-1: none)
    DevBuf d_slot_voice;                // ... on the device, for the plan that replaces this one (k_carry_cache)
    std::vector<uint8_t> ctl_mark;      // [n_voices] scratch
    std::vector<int> hot_prev, hot_now; // voices with messages in the call before / in this one (capacity reserved at build)
    int* h_ctl_order = nullptr;         // pinned [h_ctl_order_cap >= n_voices]
    size_t h_ctl_order_cap = 0;
    DevBuf d_ctl_order;
    bool ctl_order_live = false;        // the device copy holds this call's order (else: identity, nothing uploaded)
    DevBuf d_rs_wl;           // resampler plans: the work list between k_leaf_rs and k_leaf_sum_wl (FusedView::rs_wl)
    DevBuf d_lazy;            // plain voice-bank plans (no resampler source, no spatialiser stage, no chain voices): [n_voices] LazyRec —
                              // what every steady voice leaves behind for the calls after a control run (fwgpu_types.h)
    bool lazy_capable = false;
    DevBuf d_rs_tmpl;         // resampler plans: [2][n_voices] VoiceBlk — the template a steady resampler voice's VB_RS_LEAN blocks of a call
                              // share (FusedView::rs_tmpl); two copies: the control-ahead mode writes call n+1's while call n renders
    DevBuf d_progs, d_hist;   // d_hist: [n_voices][SP_HIST] mono histories the spatialiser voices enter the call with
    DevBuf d_voices, d_leaves, d_blks, d_refs, d_gsets, d_cache, d_ramps, d_bus, d_bus_flags, d_chain_start, d_chain_dummy, d_chain_stats;
    DevBuf d_up_nodes, d_up_in, d_up_out, d_up_level_nodes, d_root_bufs;
    std::vector<int> up_level_off, up_level_cnt;
    int n_tail = 0;  // master chain after the root SumNode (generic node kernel on the mix bus)
    std::vector<int> tail_kinds;  // kernel-set bit per master node
    DevBuf d_tail_nodes, d_tail_in, d_tail_out, d_tail_idx, d_tail_frozen;
    DevBuf d_frozen_ph;  // ... and its playhead snapshots
    DevBuf d_frozen;  // generic plan: k_frozen_scan's verdict per plan node, valid for the batch in flight
    DevBuf d_rt_tree, d_rt_tree_sync;  // one-launch realtime kernels: [parent of leaf][parent of upper node][children of upper node]; a counter per upper node
    int rt_tree_leaves = 0, rt_tree_up = 0;  // ... their extents (0: no tree — the edge takes the launch sequence)
    DevBuf d_chain_done;  // ... and, per node, chain_words words of "this block was rendered by the wave upstream" bits (vertical fusion)
    int chain_words = 0;
    int up_root_node = -1;  // index (in the upper-tree node table) of the root SumNode when it is alone on the last level
    RootArgs root_args;     // that node's port table, handed to k_root_out in its kernel arguments

    // FIR banks (generic executor): rows grouped by (level, impulse-response channel)
    struct FirGroup {  // one GEMM launch: every FIR row of a level with the same tap count
        int level, row_off, n_rows, tile_off;
        uint32_t T;
    };
    std::vector<FirGroup> fir_groups;
    DevBuf d_fir_rows, d_fir_tiles, d_fir_partials;

    // hybrid plan (kind 3): voice-bank groups inside a graph the level executor runs — the level lists without the nodes
    // the fused kernels render
    int n_fused_real = 0;  // voices (not null slots) the fused kernels render under the installed plan
    bool hybrid = false;
    bool hybrid_fx = false;  // ... and the banks hold biquad / delay voices: k_chain renders them
    DevBuf d_hlevel_nodes;
    std::vector<int> hlevel_off, hlevel_cnt, hlevel_kinds;

    // host nodes (K_HOST): what the audio side needs to call them, per plan level, and their pinned staging area
    struct HostCall {
        int node_idx = 0;  // index into the plan's node table (its in / out buffer ids sit in d_in_buf / d_out_buf)
        int n_in = 0, n_out = 0, in_off = 0, out_off = 0;
        fwgpu_host_process_fn fn = nullptr;
        void* user = nullptr;
        size_t stage_off = 0;  // float offset of this node's [K][n_in + n_out][stride] block in the pinned staging area
        size_t flag_off = 0;   // byte offset of its [K][n_in + n_out] silence flags
    };
    std::vector<std::vector<HostCall>> host_levels;  // per plan level
    int n_host_nodes = 0;
    uint64_t host_callbacks = 0;
    float *h_host_stage = nullptr, *d_host_stage = nullptr;  // pinned, device-mapped: the host nodes' inputs and outputs
    uint8_t *h_host_flags = nullptr, *d_host_flags = nullptr;
    size_t host_stage_floats = 0, host_flag_bytes = 0;
    std::vector<const float*> host_in_ptrs;  // scratch for the callback's pointer tables (sized at install: no allocation per call)
    std::vector<float*> host_out_ptrs;

    // the steady realtime launch sequence of THIS plan kept as a hipGraph (FWGPU_RT_GRAPH=1)
    struct RtGraph {
        hipGraphExec_t exec = nullptr;
        uint32_t epoch = 0, K = 0;
        int n_out_ch = 0;
        const float* d_out = nullptr;
    } rt_graph;

    // ---- what adopting this image does to the state that OUTLIVES plans (node states, ext pool, sampler bookkeeping): built on
    // the control thread, applied at adoption — asynchronously, on the ctx stream — so that the control thread never writes
    // into buffers a running plan reads
    DevBuf grow_states;            // a larger node-state array (old contents copied in at adoption), or empty
    size_t grow_states_cap = 0;
    DevBuf grow_ext;               // a larger ext pool
    size_t grow_ext_cap = 0;
    DevBuf d_state_inits;          // StateInitHost records of the nodes this image activates
    int n_state_inits = 0;
    DevBuf d_ext_jobs;             // AdoptExtJobHost records: recycled slices zeroed, biquad coefficients set (k_adopt_init)
    int n_ext_jobs = 0;
    struct IrConv {                // impulse responses to convert to f32 into the ext pool
        int sample, ch;
        uint32_t off, T;
    };
    std::vector<IrConv> ir_convs;
    std::vector<std::pair<uint32_t, int64_t>> activated;  // (slot, node id) of the nodes this image activates
    std::vector<uint32_t> dropped_samplers;                // removed sampler nodes: their processor "drops" at this swap
    std::vector<uint32_t> removed_slots;                   // removed nodes: messages still queued for them go with them
    size_t slots_cap = 0;                                  // cur_sample / slot_ids must hold this many slots
    std::vector<int> grow_cur_sample;                      // pre-sized replacements (empty when the current ones are large enough)
    std::vector<int64_t> grow_slot_ids;
    hipEvent_t retired_ev = nullptr;  // recorded on the ctx stream when this image stops being the active one

    PlanImage() = default;
    PlanImage(const PlanImage&) = delete;
    PlanImage& operator=(const PlanImage&) = delete;
    PlanImage(PlanImage&&) = default;
    PlanImage& operator=(PlanImage&&) = default;
    void release_device();  // frees what the image owns (control side / ctx destruction)
};

}  // namespace fwgpu

struct fwgpu_ctx : fwgpu::PlanImage {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t sample_rate = 48000;
    uint32_t mbf = 256;
    int stride = 256;
    uint32_t n_gin = 0, n_gout = 2;
    // fwgpu_last_error(): fixed buffers, one for the process calls (audio thread) and one for everything else, so that a
    // failing call never touches the host allocator and the two sides never write the same bytes
    char err_ctl[256] = {0};
    char err_audio[256] = {0};
    std::atomic<int> err_last{0};  // 0 = err_ctl, 1 = err_audio was written last

    HostGraph graph;
    bool force_generic = false;
    uint32_t kmax_req = 64;  // fwgpu_set_max_batch: takes effect when the next plan is installed

    // ---- plan hand-over (control -> audio), graph/processor.rs:167-206 + graph/context.rs:93-137
    // `gate`: 0 = nobody inside, 1 = a process call is running, 2 = the control thread is adopting an image (only ever for the
    // microseconds of the member swap, and only when it found the gate at 0).  A process call takes the gate for its whole
    // duration; the control thread BUILDS outside the gate and only tries it to adopt when the audio side is idle.
    std::atomic<int> gate{0};
    std::atomic<int> gate_ctl_waiting{0};  // control calls waiting at ControlGate: the audio side lets them in before its next call
    // fwgpu_process_interleaved_begin / _end: two output slots (device + registered host staging), a copy stream, the events that order them
    struct AsyncOut {
        DevBuf d[2];
        void* h[2] = {nullptr, nullptr};  // page-locked (hipHostRegister, mapped) ordinary memory: what the graph-output kernel writes, the host memcpy's source
        float* h_dev[2] = {nullptr, nullptr};  // ... as the device sees it
        size_t hcap[2] = {0, 0};
        hipEvent_t ev_copy[2] = {nullptr, nullptr};
        uint64_t prof_wait_ns = 0, prof_copy_ns = 0, prof_calls = 0;
        size_t bytes[2] = {0, 0};
        bool busy[2] = {false, false}, zeros[2] = {false, false};
        uint64_t ret_ticket[2] = {0, 0};
        hipEvent_t ev_render[2] = {nullptr, nullptr};
        hipStream_t copy_stream = nullptr;
        int64_t next = 0, done = 0;  // tickets handed out / ended
    } ao;
    bool level_fuse = true;                        // FWGPU_LEVEL_FUSE=0: the level executor without vertical fusion (A/B, bisecting)
    StatCounter rt_path[4];                        // one-block launch batches by path (fwgpu_rt_path_stats); audio side writes
    uint64_t gate_defer_ns = 20000;               // ... for at most this long per process call (FWGPU_GATE_DEFER_US; 0: never steps back)
    std::atomic<uint64_t> gate_defer_expired{0};  // process calls that stopped waiting for a waiter that did not come
    std::atomic<fwgpu::PlanImage*> pending{nullptr};  // built and published, waiting for the next process call
    fwgpu::PlanImage* spare = nullptr;                // control side: a retired image, the next build target (its buffers are reused)
    static constexpr uint32_t RETIRE_CAP = 8;
    fwgpu::PlanImage* retired[RETIRE_CAP] = {nullptr};   // audio -> control SPSC ring of images that stopped being active
    std::atomic<uint32_t> retired_head{0}, retired_tail{0};
    std::atomic<uint64_t> adopted_gen{0};             // generation of the active image (written by whoever adopts)
    uint64_t build_gen = 0;                           // control side: generation of the last image built
    hipStream_t up_stream = nullptr;                  // control side: uploads of the image being built
    // control-side mirror of what the introspection calls report: the LATEST BUILT plan (adopted or still pending)
    struct PlanInfo {
        bool have_plan = false;
        int kind = 0, fused_voices = 0, n_host_nodes = 0;
        Plan plan;
    } info;
    // control side: what becomes reusable only once the image that stops using it has been adopted — node slots, ext slices
    // and impulse-response copies of removed nodes (tag = generation of the first image built without them)
    struct Limbo {
        uint64_t gen;
        int what;  // 0 = node slot, 1 = ext slice (off, rounded len)
        uint32_t a, b;
    };
    std::vector<Limbo> limbo;
    std::vector<uint32_t> limbo_slots;     // HostGraph::remove_node parks freed slots here (graph.limbo points at it)
    std::vector<uint32_t> pending_removed; // slots removed since the last build (the next image carries them)
    size_t ctl_states_cap = 0, ctl_ext_cap = 0;  // control side's view of the capacities once every built image is adopted
    std::vector<Cmd> early_msgs;           // messages for nodes no built plan holds yet: released when their plan is published

    // device state that outlives plans
    DevBuf d_states;
    size_t states_cap = 0;
    DevBuf d_ext;  // per-node extended state (floats): biquad coefficients + history, delay rings
    size_t ext_cap = 0, ext_used = 0;
    std::vector<SampleRec> samples;
    DevBuf d_samples;
    std::vector<SampleDesc> h_sample_tab;  // host copy of d_samples, rebuilt by the calls that change it (control side)
    bool samples_dirty = true;

    uint32_t epoch = 1;  // invalidates every VoiceCache when bumped (plan adoption, sample-table change)
    // Control kernel one batch AHEAD (FWGPU_CTL_AHEAD, voice-bank plan without a master chain, batches of more than one block):
    // k_voice_control of batch b+1 runs on its own high-priority stream under the render kernels of batch b.  What it writes and
    // the render kernels read exists twice (parity = batch number & 1); what orders the two streams is one event per parity and
    // direction.  The kernels do not know: they get pointers.
    bool ctl_ahead = true;           // wanted (default since round 3; FWGPU_CTL_AHEAD=0 switches it off)
    // Lazy records (round 4, fwgpu_types.h LazyRec): a message-free call of a plan whose every voice was left steady AND plain by
    // the last control kernel — the host knows because that kernel's horizon has arrived in pinned memory — is rendered without a
    // control kernel: the leaf waves compute their records from the LazyRecs.  Node state then lags by `lazy_pending` blocks until
    // the next thing that reads it (a control kernel, the realtime kernels, the level executor, an adoption) — lazy_flush first.
    // FWGPU_LAZY=0 switches it off.
    bool lazy_on = true;
    DevBuf d_lazy_horizon;                       // one u64, re-armed to ~0 by k_lazy_publish
    unsigned long long *h_lazy_pub = nullptr, *d_lazy_pub = nullptr;  // pinned {horizon, seq of the control launch it belongs to}
    uint64_t ctl_launch_seq = 0;                 // control kernels launched (with a publish behind each)
    uint64_t abs_blk = 0;                        // blocks the fused voice-bank plan has rendered (any plan image: it only orders)
    uint64_t lazy_base_blk = 0;                  // abs_blk right behind the last control run: the LazyRecs' block 0
    uint64_t lazy_pending = 0;                   // blocks rendered from the LazyRecs that node state has not seen yet
    uint32_t lazy_epoch = 0;                     // epoch the LazyRecs were made under
    bool lazy_valid = false;                     // nothing but lazy calls has moved the voices since they were made
    bool lazy_this_call = false;
    StatCounter lazy_calls, ctl_calls;           // fused batches rendered without / with a control kernel (fwgpu_lazy_stats)
    int ctl_ahead_mode = 2;          // 1 = every qualifying call (round 3), 2 = only calls with messages / continuing glides (round 4)
    hipStream_t ctl_stream = nullptr;
    hipEvent_t ev_ctl[2] = {nullptr, nullptr}, ev_render[2] = {nullptr, nullptr}, ev_join = nullptr;
    uint64_t ahead_seq = 0;          // batches launched in ahead mode since the streams were last joined
    bool streams_split = false;      // ctl_stream may hold work the main stream has not waited for
    bool ahead_this_call = false;    // the process call in progress runs in ahead mode
    bool cmds_on_ctl = false;        // ... and its message upload goes to the control stream

    std::map<std::pair<int, int>, uint32_t> ir_cache;  // (sample id, channel) -> ext offset of h[T] as f32
    std::map<std::pair<int, int>, uint32_t> ir_len;    // ... and its length in floats (freed with the last FIR user)
    std::map<size_t, std::vector<uint32_t>> ext_free;  // ext slices of removed nodes, by 64-rounded size (floats)

    // host nodes: the control side's registry of process functions (by node slot); an image gets copies
    struct HostProc {
        fwgpu_host_process_fn fn = nullptr;
        void* user = nullptr;
    };
    std::vector<HostProc> host_procs;

    // messages.  Setters push into `ring` from any thread; the audio thread drains it at the start of a process call
    // into `cmds` (sorted by (node, block), arrival order inside a block).  Every buffer on this path has its final
    // size from fwgpu_ctx_create on: a process call never allocates for messages.
    static constexpr uint32_t RING_CAP = 1u << 15;  // messages in flight between two process calls
    static constexpr size_t CMD_CAP = 1u << 16;     // messages waiting for their block (drained, not yet applied)
    MsgRing ring;
    std::atomic<uint64_t> drain_epoch{1};  // bumped by every drain: the producers' view of "the ring was emptied"
    std::vector<Cmd> cmds, cmds_scratch;   // reserve(CMD_CAP) once; never grown
    DevBuf d_cmds;
    int n_cmds_dev = 0;
    Cmd* h_cmds = nullptr;  // pinned staging for the async upload [CMD_CAP]
    hipEvent_t cmds_copied = nullptr;
    // ProcessorToNodeMsg::ReturnSample (sampler.rs:339-343): which sample every sampler holds, as the messages retired
    // so far leave it (audio thread), and the swapped-out ones on their way back to the control side
    std::vector<int> cur_sample;     // [node slot] -> sample id or -1; sized at adoption
    std::vector<int64_t> slot_ids;   // [node slot] -> node id of the activated node
    RetRing returns;
    static constexpr uint32_t RET_EVENTS = 64;
    hipEvent_t ret_events[RET_EVENTS] = {nullptr};
    std::atomic<uint32_t> ret_event_ticket[RET_EVENTS] = {};  // ticket + 1 the slot's event was last recorded for (0 = never)
    uint32_t ret_ticket = 0;         // audio thread: process calls that returned a sample so far
    std::atomic<uint32_t> ret_done_ticket{0};  // tickets below this belong to calls the audio thread has SEEN complete (sync / flag)
    bool ret_this_call = false;
    // control side of the same: reference counts per sample id = SetSample messages sent - samples handed back
    std::vector<int64_t> sample_refs;
    std::vector<RetItem> ret_ready;  // completed returns not yet handed to fwgpu_poll_returned_samples
    std::vector<uint32_t> dropped_samplers_ctl;  // control side: removed sampler nodes since the last build

    // staging + B1 scratch
    DevBuf d_in_stage, d_out_stage, d_scratch_pool, d_scratch_flags, d_scratch_tab, d_mask;
    DevBuf d_trace;  // FW_CHAIN_TRACE builds only

    // realtime edge (cpal/lib.rs:378-449 — one callback = a few hundred frames): pinned, device-mapped I/O blocks the
    // kernels read / write directly (no copy-engine hop)
    float *h_rt_in = nullptr, *h_rt_out = nullptr, *d_rt_in = nullptr, *d_rt_out = nullptr;
    // completion flag of realtime-sized calls: a word in pinned, device-mapped host memory the last kernel of the call sets to
    // the call's sequence number; the audio thread polls it instead of paying a blocking stream sync's wake-up
    unsigned long long *h_rt_flag = nullptr, *d_rt_flag = nullptr;
    unsigned long long rt_seq = 0;         // sequence number of the last realtime call
    unsigned long long rt_signal_seq = 0;  // != 0 while run_blocks should arrange for the flag to be raised
    bool rt_signalled = false;
    bool rt_last_batch = false;  // the fused batch being launched ends the call
    int rt_persist_max_leaves = 64;  // the resident kernel takes trees of at most this many leaf workgroups (fwgpu_run.cpp)
    bool rt_one_launch = true;  // one-block calls on the voice-bank plan: control + leaf + root in ONE kernel (FWGPU_RT_ONE_LAUNCH=0: off)
    DevBuf d_rt_sync;           // its workgroup counter
    // the resident realtime kernel (k_rt_persist, k_rt.hip.h): launched by the first steady one-block callback of a run of them, fed
    // through `h_rt_mb->doorbell` from then on, ended by rt_persist_stop before anything else touches the device state it owns
    // control side: pinned staging arena of the plan build's uploads (fwgpu_plan_install.cpp `up`)
    char* h_up = nullptr;
    size_t h_up_cap = 0, h_up_used = 0;
    // the build's GPU work goes out in pieces, each in a window with no process call in flight (fwgpu_plan_install.cpp, quiet_window):
    // how long a piece waits for such a window (0 = the old behaviour: everything at once), and the size of an upload piece
    Plan spare_plan;  // control side: a retired plan's node array, reused by the next fwgpu_update
    // FWGPU_UPDATE_PROF=1: host nanoseconds of the control thread per update phase, printed when the ctx is destroyed
    bool update_prof = false;
    bool update_prof_tables = false;  // FWGPU_UPDATE_PROF=2: one stderr line per table a build uploads (bytes, bytes that differed)
    uint64_t phase_ns[32] = {0}, phase_t0 = 0, phase_updates = 0;
    uint64_t prof_groups = 0, prof_jobs = 0, prof_copy_bytes = 0, prof_fill_bytes = 0;  // build_apply's launches / jobs / bytes
    std::atomic<int> update_phase{0};  // fwgpu_update_phase: 0 none, 1 graph compile, 21..28 the sections of build_image, 3 waiting for the uploads
    std::atomic<uint64_t> last_audio_ns{0};  // steady_clock at the end of the last process call (0: none yet)
    std::atomic<uint64_t> cb_start_ns{0}, cb_period_ns{0}, cb_dur_ns{0};  // the last call's start, its distance to the one before, its length
    uint32_t quiet_wait_us = 30;    // FWGPU_QUIET_WAIT_US (100 when the build runs on its own stream)
    uint32_t up_piece = 256u << 10; // FWGPU_UP_PIECE (bytes): ~10 us of copy kernel per group (round 4: 128 -> 256 KiB, same p99 beside a saturated stream, half the groups)
    bool up_diff = true;            // FWGPU_UP_DIFF=0: every table uploaded whole, every build
    bool build_one_kernel = true;   // FWGPU_BUILD_ONE_KERNEL=0: the build's copies / fills as separate runtime calls
    std::vector<BuildJob> build_jobs;  // control thread: what build_apply will launch (fwgpu_plan_install.cpp)
    bool build_on_audio_stream = true;   // a build's job groups go into the AUDIO stream (round 4 default; FWGPU_BUILD_STREAM=own: the
                                         // build's own low-priority stream, round 3) — see build_apply
    hipEvent_t ev_build = nullptr;
    char* h_jobs = nullptr;            // ... and the pinned memory the launched lists travel in
    size_t h_jobs_cap = 0, h_jobs_used = 0;
    bool rt_persist = true;        // FWGPU_RT_PERSIST=0: every callback is its own launch (k_rt_block)
    uint32_t rt_idle_ms = 20;      // its watchdog: no doorbell for this long and it ends by itself (FWGPU_RT_IDLE_MS)
    RtMailbox *h_rt_mb = nullptr, *d_rt_mb = nullptr;
    hipStream_t rt_stream = nullptr;
    hipEvent_t rt_ev = nullptr;
    struct RtResident {
        bool launched = false;
        uint64_t epoch = 0;
        float* d_out = nullptr;
        const void* blks = nullptr;       // (the FusedView the kernel was launched with: any difference means a new launch)
        unsigned long long next_seq = 0;  // the doorbell value it waits for
        StatCounter launches, doorbells;
        uint64_t held = 0;                // launches not made because a control call held the device (RtHold)
    } rtp;
    bool rt_use_graph = false;  // FWGPU_RT_GRAPH=1: measured 5 us SLOWER per callback than 4 plain launches on ROCm 7.2
    DevBuf d_rs_table;  // SPEC resampler filter bank [RS_PHASES][RS_TAPS]

    // fwgpu_process_blocks_device_flags: where the call in progress reports, per (block, channel), whether that graph-output
    // channel was flagged silent (device memory of the caller; null = not asked for)
    uint8_t* out_sil = nullptr;

    // ProcInfo of the call in progress (core/node.rs:111-118) + what the backend reported so far (StreamStatus bits)
    double proc_stream_time = 0.0;
    uint32_t proc_stream_status = 0;
    uint64_t n_underflows = 0, n_overflows = 0;

    // FWGPU_HOST_PROF=1 (experiments): host nanoseconds spent inside process calls / inside the HIP launch calls they make,
    // printed to stderr when the ctx is destroyed
    bool host_prof = false;
    uint64_t hp_calls = 0, hp_call_ns = 0, hp_launch_ns = 0, hp_launches = 0;
    uint64_t hp_hist[16] = {0};
    uint64_t adopt_ns_max = 0, adoptions = 0, audio_adoptions = 0;  // the longest an adoption held up a process call (host nanoseconds)

    // timing
    bool timing = false;
    TimerCat timers[5];  // 0 fused leaf kernel, 1 control kernel, 2 upper sums + out, 3 generic block, 4 k_fir_gemm alone

    fwgpu_ctx(uint32_t gin, uint32_t gout) : graph(gin, gout) {}
};

namespace fwgpu {

// every change of fwgpu_update_phase goes through here (the phase that ends is charged with the time since the last mark)
inline void phase_mark(fwgpu_ctx* c, int p) {
    if (c->update_prof) {
        const uint64_t now = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        const int was = c->update_phase.load(std::memory_order_relaxed);
        if (was > 0 && was < 32) c->phase_ns[was] += now - c->phase_t0;
        c->phase_t0 = now;
        if (p == 0 && c->phase_updates++ == 0) {  // the first update (allocations, every table whole) is not what an edit costs
            for (uint64_t& v : c->phase_ns) v = 0;
            c->prof_groups = c->prof_jobs = c->prof_copy_bytes = c->prof_fill_bytes = 0;
        }
    }
    c->update_phase.store(p, std::memory_order_relaxed);
}

int fail(fwgpu_ctx* c, int code, const char* msg);  // no allocation: the message is copied into a fixed buffer
inline int fail(fwgpu_ctx* c, int code, const std::string& msg) { return fail(c, code, msg.c_str()); }
// process entry points bracket themselves with this: fail() then writes the audio thread's buffer
struct AudioCallScope {
    AudioCallScope();
    ~AudioCallScope();
};
int hipfail(fwgpu_ctx* c, hipError_t e, const char* what);
#define HIPC(c, x)                                        \
    do {                                                  \
        hipError_t e__ = (x);                             \
        if (e__ != hipSuccess) return hipfail(c, e__, #x); \
    } while (0)
#define LCHK(c, x)                                                        \
    do {                                                                  \
        int e__ = (x);                                                    \
        if (e__ != 0) return hipfail(c, (hipError_t)e__, "kernel launch " #x); \
    } while (0)

// ---- fwgpu_control_math.cpp
float percent_volume_to_raw_gain(float p);
float db_to_gain_clamped_neg_100_db(float db);
void pan_to_gains(float pan, float* gl, float* gr);
Smoother make_smoother(float val, uint32_t sample_rate);
void biquad_coefs(int type, float cutoff_hz, float q, uint32_t sample_rate, float co[5]);
void resampler_table(float* h);
uint64_t resampler_step(float ratio);
void spatial_params(float x, float y, float z, uint32_t sample_rate, float* gl, float* gr, int* dl, int* dr);
uint32_t delay_frames(float secs, uint32_t sample_rate);
NodeState make_state(int kind, const float* params, int n_params, uint32_t sample_rate);

// ---- fwgpu_plan_detect.cpp
struct FusedBuild {
    std::vector<VoiceDesc> voices;
    std::vector<LeafDesc> leaves;
    std::vector<NodeDesc> up_nodes;
    std::vector<int> up_in, up_out;
    std::vector<std::vector<int>> up_levels;  // indices into up_nodes per level
    int root_buf[2];
    // master chain: stereo 2->2 nodes between the root SumNode and graph_out (volume, hard clip, pan, width, biquad,
    // delay), run by the generic node kernel on the mix bus, one launch each, nearest the root first
    std::vector<NodeDesc> tail_nodes;
    std::vector<int> tail_in, tail_out;
    int n_bus = 1;
    int max_stages = 0;
    std::vector<uint32_t> progs;  // per voice: its chain stages' kinds (SK_*), 4 bits each
    bool has_prog = false;        // a width / hard-clip stage somewhere: the leaf kernel's program instantiation
    bool has_rs = false;          // a resampler-sourced voice somewhere
    bool has_sp = false;          // a voice whose last stage is a spatialiser somewhere
    bool has_width = false;       // a stereo-width stage somewhere (the chain plan's kernels render none)
    bool has_fx = false;  // some chain holds a biquad / delay: the k_chain plan
    uint64_t min_delay = ~0ull;  // shortest delay line among the chains (frames)
    std::vector<int> covered;    // hybrid plan: plan indices of the nodes the fused kernels render (voice chains + their SumNode)
    // hybrid plan: SumNodes that ALSO take other inputs behind their leading voice ports.  The voice-bank kernels sum the
    // leading ports into a partial bus (the reference's accumulator at that point), the node itself stays on the levels as
    // a continuation: (partial, the other ports...) on the path of its full port count
    struct Split {
        int sum;    // plan index of the SumNode
        int leaf;   // index into `leaves`
        int lead;   // leading voice ports taken by the leaf
    };
    std::vector<Split> splits;
};
bool detect_fused(const Plan& plan, const HostGraph& graph, uint32_t mbf, FusedBuild& fb);
// The hybrid plan (kind 3): the graph as a whole is not a fused shape, but it holds voice banks that are — SumNodes whose
// every port is a dry voice chain.  Those groups are rendered by the voice-bank kernels straight into the SumNode's pool
// buffers; everything else runs on the level executor, which finds the groups' outputs where it expects them.
bool detect_hybrid(const Plan& plan, const HostGraph& graph, uint32_t mbf, FusedBuild& fb);

// ---- fwgpu_plan_install.cpp
int install_plan(fwgpu_ctx* c, Plan& plan);           // control side: build the next image off to the side, publish it
void adopt_image(fwgpu_ctx* c, PlanImage* n, bool on_audio_thread);         // whoever holds the gate: make `n` the active image, retire the old one
void collect_retired(fwgpu_ctx* c);                   // control side: reuse / free the images the audio side is done with
// A process call holds the gate for its whole duration and picks up a pending image at its start (graph/processor.rs:167-206:
// poll_messages applies NewSchedule before the block).  It only ever spins while the control thread is inside the
// microseconds of an adoption it started because it found the audio side idle.
struct AudioGate {
    fwgpu_ctx* c;
    explicit AudioGate(fwgpu_ctx* ctx) : c(ctx) {
        // (a control call waiting at ControlGate goes first: a stream of back-to-back callbacks holds the gate ~100 % of the time, and a
        //  waiter that has to catch the instant between two of them starved for tens of milliseconds — 80 sample_create calls beside
        //  fwgpu_stream_run took 4.1 s, r04.  The audio side waits here for the waiter's few microseconds instead.)
        //  The deference is BOUNDED: a waiter that raised the counter and was descheduled before it took the gate
        //  must not cost the realtime thread a scheduler quantum.  The audio side steps back for at most gate_defer_ns per call —
        //  a running waiter needs a fraction of a microsecond to win the CAS — then takes the free gate as it always did; the
        //  waiter gets the same head start at the next callback.  (A control thread INSIDE its critical section still holds the
        //  gate for its few microseconds: that is mutual exclusion, not deference.)
        uint64_t defer_t0 = 0;
        bool defer = c->gate_defer_ns != 0;
        for (;;) {
            int expected = 0;
            const bool waiter = defer && c->gate_ctl_waiting.load(std::memory_order_acquire) != 0;
            if (!waiter && c->gate.compare_exchange_weak(expected, 1, std::memory_order_acquire)) break;
            if (waiter) {
                const uint64_t now = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
                if (!defer_t0) defer_t0 = now;
                else if (now - defer_t0 > c->gate_defer_ns) {
                    defer = false;
                    c->gate_defer_expired.fetch_add(1, std::memory_order_relaxed);
                }
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
        // the stream's rhythm, for quiet_window: when this call began and how long after the one before
        const uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
        const uint64_t prev = c->cb_start_ns.load(std::memory_order_relaxed);
        if (prev) c->cb_period_ns.store(t - prev, std::memory_order_relaxed);
        c->cb_start_ns.store(t, std::memory_order_relaxed);
        if (PlanImage* img = c->pending.exchange(nullptr, std::memory_order_acq_rel)) adopt_image(c, img, true);
    }
    ~AudioGate() {
        // when the audio side last ran (quiet_window: a control thread cuts its GPU work into pieces only while a stream is live)
        const uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
        c->last_audio_ns.store(t, std::memory_order_relaxed);
        c->cb_dur_ns.store(t - c->cb_start_ns.load(std::memory_order_relaxed), std::memory_order_relaxed);
        c->gate.store(0, std::memory_order_release);
    }
};
// The control-side calls that change what a process call reads OUTSIDE a plan image — the sample table, max_batch /
// force_generic — take the gate exclusively for the few microseconds of the change (the expensive part, uploading sample
// data, happens before).  A process call that starts meanwhile waits at its entry; the control thread waits for a running one
// to finish.  (fwgpu_update never does this: it builds outside the gate.)
struct ControlGate {
    fwgpu_ctx* c;
    explicit ControlGate(fwgpu_ctx* ctx) : c(ctx) {
        c->gate_ctl_waiting.fetch_add(1, std::memory_order_acq_rel);
        int expected = 0;
        for (unsigned spins = 1; !c->gate.compare_exchange_weak(expected, 2, std::memory_order_acquire); ++spins) {
            expected = 0;
            if ((spins & 1023u) == 0) std::this_thread::yield();
#if defined(__x86_64__) || defined(__i386__)
            else __builtin_ia32_pause();
#endif
        }
        c->gate_ctl_waiting.fetch_sub(1, std::memory_order_acq_rel);
    }
    ~ControlGate() { c->gate.store(0, std::memory_order_release); }
};

// A control call that may free device memory, free pinned memory or wait for the device holds this for its duration: hipFree /
// hipHostFree / hipDeviceSynchronize wait for every stream of the device, and the resident realtime kernel (k_rt_persist on
// rt_stream) ends only when it is told to or after 20 ms without a doorbell — a steady stream of callbacks keeps it alive for
// ever, and the control call with it.  Raising RtMailbox::hold makes the kernel end at its next poll (it finishes
// the block it is rendering) and keeps the audio side from launching another: its callbacks go out as ordinary launches
// (k_rt_block, +4 us each) until the last holder lets go.  Dekker pair with rt_persist_launch: [hold++ ; read alive] here,
// [alive = 1 ; read hold] there, both sequentially consistent — either the launch sees the hold, or this sees the kernel.
struct RtHold {
    fwgpu_ctx* c;
    explicit RtHold(fwgpu_ctx* ctx) : c(ctx) {
        if (!c || !c->h_rt_mb) return;
        __atomic_fetch_add(&c->h_rt_mb->hold, 1ull, __ATOMIC_SEQ_CST);
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(10 * (long)c->rt_idle_ms + 50);
        for (unsigned spins = 1; __atomic_load_n(&c->h_rt_mb->alive, __ATOMIC_SEQ_CST); ++spins) {
            if ((spins & 255u) == 0) {
                if (std::chrono::steady_clock::now() > give_up) break;  // (a device that does not answer: the call's own HIP errors name it)
                std::this_thread::yield();
            }
        }
    }
    ~RtHold() {
        if (c && c->h_rt_mb) __atomic_fetch_sub(&c->h_rt_mb->hold, 1ull, __ATOMIC_SEQ_CST);
    }
    RtHold(const RtHold&) = delete;
    RtHold& operator=(const RtHold&) = delete;
};

// ---- fwgpu_plan_install.cpp
// control side, before a piece of GPU work that is not the audio path's: returns when no process call is in flight, or after
// c->quiet_wait_us
void quiet_window(fwgpu_ctx* c);
// is a stream live (a process call within the last 200 ms)?  If not, the control side's GPU work goes out whole.
inline bool audio_live(const fwgpu_ctx* c) {
    if (!c->quiet_wait_us) return false;
    if (c->gate.load(std::memory_order_relaxed) == 1) return true;
    const uint64_t last = c->last_audio_ns.load(std::memory_order_relaxed);
    if (!last) return false;
    return (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() - last < 200000000ull;
}

// ---- fwgpu_run.cpp
int upload(fwgpu_ctx* c, DevBuf& b, const void* src, size_t bytes);
int upload_sample_table(fwgpu_ctx* c);
int join_streams(fwgpu_ctx* c);  // control-ahead mode: both streams wait for each other's work so far (no host wait)
void drain_ring(fwgpu_ctx* c);  // ring -> cmds (consumer side: the audio thread, or an edit call that does not overlap it)
int upload_cmds(fwgpu_ctx* c, bool drained = false);  // drained: the caller has emptied the ring into cmds already
int rt_persist_stop(fwgpu_ctx* c);
int lazy_flush(fwgpu_ctx* c);  // audio side (or whoever holds the gate): node state brought up to date after lazily rendered blocks; the LazyRecs end here
int rt_block_relaunch(fwgpu_ctx* c, float* d_out, unsigned long long seq);  // after the watchdog race: the same block through k_rt_block  // ends the resident realtime kernel, if one is running, and waits for it
void finish_returns(fwgpu_ctx* c);  // end of a process call: completion event for the samples it handed back
void retire_cmds(fwgpu_ctx* c, uint32_t nblocks);
void retire_cmds_node(fwgpu_ctx* c, int slot);
void timer_begin(fwgpu_ctx* c, int cat, hipEvent_t* e0, hipEvent_t* e1);
void timer_end(fwgpu_ctx* c, hipEvent_t e1);
void timer_drain(fwgpu_ctx* c);
DevView generic_view(fwgpu_ctx* c, int frames);
int run_generic_batch(fwgpu_ctx* c, int K, int frames, uint32_t cmd_block, const float* d_in, int n_in_ch, float* d_out,
                      int n_out_ch);
int run_fused_batch(fwgpu_ctx* c, int K, uint32_t cmd_block0, float* d_out, int n_out_ch);
int run_blocks(fwgpu_ctx* c, uint64_t frames, const float* d_in, int n_in_ch, float* d_out, int n_out_ch,
               bool stable_out = false);

}  // namespace fwgpu
