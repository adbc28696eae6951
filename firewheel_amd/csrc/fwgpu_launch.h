// fwgpu_launch.h — kernel argument views + launch wrappers (implemented in fwgpu_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "fwgpu_types.h"

namespace fwgpu {

// What k_level sees: the renamed-buffer pool of one plan.  `pool`/`flags` hold K independent blocks
// (pool_blk_stride floats / flags_blk_stride bytes apart) for the K-batched upper sum tree; the generic
// executor uses K = 1.  Buffer 0 is the constant zero buffer (flag always 1) = every unconnected input.
struct DevView {
    const NodeDesc* nodes;
    const int* in_buf;
    const int* out_buf;
    NodeState* states;
    const SampleDesc* samples;
    float* ext;  // per-node extended state (biquad coefficients + history, delay rings)
    const float* rs_table;  // SPEC resampler filter bank [RS_PHASES][RS_TAPS]
    float* pool;
    uint8_t* flags;
    size_t pool_blk_stride;
    size_t flags_blk_stride;
    int stride;  // floats per channel-block (multiple of 64)
    int frames;  // frames in this block (<= max_block_frames)
    const Cmd* cmds;
    int n_cmds;
    // per node of `nodes` (or nullptr): 1 = a gain-like stateful node whose control state cannot move during this
    // batch — k_level then runs its blocks in parallel (k_frozen_scan, launched once per batch before the levels)
    const uint8_t* frozen;
    const unsigned long long* frozen_playhead;  // per node: a frozen playing sampler's playhead at the start of the batch
    // vertical fusion of frozen 1:1 chains (round 5, k_generic.hip.h fz_links): per node `chain_words` words of block bits — set by the
    // wave that rendered a downstream node's block in registers, read by that node's own wave a level later.  nullptr: no fusion.
    uint32_t* chain_done = nullptr;  // (initialised here: the upper-tree / master-chain / realtime views are filled field by field)
    int chain_words = 0;
};

// The compact records are tiled: 32 voices x 8 blocks per 4 KiB tile, [voice % 32][block % 8].  A voice's 8 consecutive blocks
// are one 128-B line — what its control wave stores (full lines) and what k_chain walks — and the 32 records a leaf wave
// loads for ONE block sit in one page, 128 B apart.  (Voice-major rows put them 12 KiB apart at 768 blocks per call: 32 pages
// per wave, and k_leaf_sum ran 257 or 291 us depending on where the context's buffers had landed — DESIGN.md §7.)
#define FW_REF_TILE_VOICES 32
#define FW_REF_TILE_BLOCKS 8
#ifdef __HIPCC__
#define FW_HOST_DEVICE __host__ __device__
#else
#define FW_HOST_DEVICE  // (the host-logic test harness compiles the host translation units with plain g++)
#endif
static inline FW_HOST_DEVICE size_t ref_index(int voice, int k, int kgroups) {
    return ((((size_t)(voice / FW_REF_TILE_VOICES) * (size_t)kgroups + (size_t)(k / FW_REF_TILE_BLOCKS)) * FW_REF_TILE_VOICES +
             (size_t)(voice % FW_REF_TILE_VOICES)) * FW_REF_TILE_BLOCKS) + (size_t)(k % FW_REF_TILE_BLOCKS);
}
static inline size_t ref_count(size_t n_voices, size_t kmax) {
    const size_t vt = (n_voices + FW_REF_TILE_VOICES - 1) / FW_REF_TILE_VOICES, kg = (kmax + FW_REF_TILE_BLOCKS - 1) / FW_REF_TILE_BLOCKS;
    return vt * kg * FW_REF_TILE_VOICES * FW_REF_TILE_BLOCKS;
}
struct FusedView {
    const VoiceDesc* voices;
    const LeafDesc* leaves;
    NodeState* states;
    const SampleDesc* samples;
    VoiceRef* refs;   // tiled, see ref_index(): ref_kgroups = ceil(max blocks per call / 8)
    int ref_kgroups;
    GainSet* gsets;   // [n_voices][FW_GSETS], valid for the current call
    VoiceCache* cache;  // [n_voices]
    uint32_t epoch;     // >= 1
    VoiceBlk* blks;   // [K][n_voices], written only for blocks that are neither silent nor VB_SIMPLE
    LazyRec* lazy;      // [n_voices] written by k_voice_control (nullptr: the plan does not do lazy records), read by the lazy leaf kernel
    unsigned long long* horizon;  // k_voice_control: atomicMin of the absolute block up to which every voice's LazyRec holds (0: some voice has none)
    uint64_t abs_blk_end;         // k_voice_control: absolute block index right behind this call (the LazyRecs' block 0)
    uint64_t lazy_blk0;           // lazy leaf kernel: this call's first block, counted from the LazyRecs' block 0
    int lazy_chain = 0;           // round 6: this k_chain launch derives its block records from the LazyRecs too (no control kernel ran)
    int lazy_rs = 0;              // round 6: ... and this k_leaf_rs launch (rs_tmpl then points at the LazyRecs' templates)
    VoiceBlk* lazy_tmpl = nullptr;  // resampler plans with lazy records: [n_voices] the template of a voice's LazyRec (control kernel: written with the record)
    VoiceBlk* rs_tmpl;  // has_rs: [n_voices] the descriptor a steady resampler voice's VB_RS_LEAN blocks of this call share (all but off0)
    const float* rs_table;  // SPEC resampler filter bank [RS_PHASES][RS_TAPS] (voices whose source is a resampler)
    const uint32_t* progs;  // [n_voices] stage programs (SK_*, 4 bits per chain stage); nullptr / all 0 on gains-only plans
    int has_prog;           // some voice's program is not 0 (or its source is a resampler): k_leaf_sum<true>
    int has_rs;             // some voice's source is a resampler: the program instantiation stages windows + filter bank in LDS
    int has_sp;             // some voice ends in a spatialiser stage: k_leaf_sum_sp
    int sp_hist_in_render;  // control-ahead mode + spatialiser stages: k_sp_hist_copy makes the history copy, not k_voice_control
    const int* ctl_order;   // k_voice_control: wave w works voice ctl_order[w] (nullptr: w) — voices with messages first
    unsigned int* rs_wl;    // has_rs: work list k_leaf_rs leaves for k_leaf_sum_wl — [0] items, [1] workgroups done, then (leaf, block*4 + piece) pairs
    float* hist;            // [n_voices][SP_HIST]: the mono history each spatialiser voice enters THIS call with (copied from the ext
                            // pool by k_voice_control, so that the render waves of block 0 read it while those of the last block
                            // write the next call's into the pool)
    int n_gain_stages;  // 1 (sampler gain) + longest chain in the plan
    float* ramps;     // [K][n_voices][ramp_slots][stride], slot = 2*stage + channel
    int ramp_slots;
    float* bus;       // [K][n_bus_buffers][stride]
    uint8_t* bus_flags;
    size_t bus_blk_stride;
    size_t bus_flags_blk_stride;
    const Cmd* cmds;
    int n_cmds;
    int n_voices;
    int n_leaves;
    int stride;
    int frames;
    const ChainGroup* groups;  // k_chain plan: workgroups = groups of consecutive leaves
    int n_groups;
    int fx_plan;  // 1 = the chain plan (k_chain renders the leaves): descriptor conventions of k_voice_control
    float* ext;  // biquad coefficients + history, delay rings (k_chain plan)
    ChainStart* chain_start;  // [n_voices] (k_chain plan)
    float* chain_dummy;       // k_chain's steady-call loop: where lanes with nothing to fetch / store point (>= 32 KiB)
    unsigned long long* chain_stats;  // [2] workgroups that ran the steady-call loop / the general loop
    // the one-launch realtime kernels' way up the mixer tree (k_rt.hip.h): per leaf / per upper-tree node the upper-tree node that
    // reads its bus (the root's own entry: -1), per upper-tree node its number of connected children, one arrival counter per node
    const int* rt_parent_leaf = nullptr;
    const int* rt_parent_up = nullptr;
    const int* rt_kids = nullptr;
    unsigned* rt_tree_sync = nullptr;
    int rt_root = -1;
    unsigned long long* trace;  // FW_CHAIN_TRACE builds only: per-step role timestamps of workgroup 0
    int dbg;     // FW_CHAIN_TRACE builds only (env FWGPU_CHAIN_SKIP): bit 0 skip S2, 1 skip S3b, 2 skip source loads,
                 // 3 skip ring RMW, 4 no ring prefetch
};

int launch_level(hipStream_t s, const DevView& v, const int* d_level_nodes, int n_nodes, int K, uint32_t cmd_block0,
                 int kinds = 7);
int launch_frozen_scan(hipStream_t s, const DevView& v, int n_nodes, uint32_t cmd_block0, int K, uint8_t* d_frozen,
                       unsigned long long* d_playhead_snap);
int launch_bus_sum(hipStream_t s, const DevView& v, const int* d_level_nodes, int n_nodes, int K, int n_out);
// the fused plans' root SumNode + read_graph_outputs + interleave_stereo in one launch (stereo output)
int launch_root_out(hipStream_t s, const DevView& v, const RootArgs& root, float* d_out, int K);
// FIR bank: rows sharing one impulse-response channel h[T] (f32 in the ext pool at h_off)
int launch_ir_convert(hipStream_t s, const SampleDesc* samples, int sample, int ch, float* dst, uint32_t T);
// gemm_begin / gemm_end: optional events recorded around the k_fir_gemm launch alone (bench roofline)
int launch_fir(hipStream_t s, const DevView& v, const FirRow* d_rows, int n_rows, const uint32_t* d_tile_h_off, uint32_t T,
               float* d_partials, size_t partial_cap_floats, int K = 1, hipEvent_t gemm_begin = nullptr,
               hipEvent_t gemm_end = nullptr);
int launch_single_node(hipStream_t s, const DevView& v, int node_idx);
int launch_scatter_states(hipStream_t s, NodeState* states, const void* d_inits, int n);
int launch_graph_in(hipStream_t s, float* pool, uint8_t* flags, int stride, size_t pool_blk_stride, size_t flags_blk_stride,
                    const int* d_bufs, int n_bufs, const float* d_interleaved, int n_in_ch, int frames, int K);
int launch_graph_out(hipStream_t s, const float* pool, const uint8_t* flags, int stride, size_t pool_blk_stride,
                     size_t flags_blk_stride, const int* d_bufs, int n_bufs, float* d_out, int n_out_ch, int frames, int K);
int launch_set_flags(hipStream_t s, uint8_t* flags, const int* d_bufs, int n, uint64_t mask);
int launch_get_flags(hipStream_t s, const uint8_t* flags, const int* d_bufs, int n, uint64_t* d_mask);
int launch_voice_control(hipStream_t s, const FusedView& fv, int K, uint32_t cmd_block0, bool beside_render = false);
int launch_leaf_sum(hipStream_t s, const FusedView& fv, int K);
// round 4, lazy records (fwgpu_types.h LazyRec): the leaf kernel that computes its records itself; the horizon word -> pinned memory;
// node state brought up to date after `blocks` lazily rendered blocks
int launch_leaf_sum_lazy(hipStream_t s, const FusedView& fv, int K);
int launch_lazy_publish(hipStream_t s, unsigned long long* d_horizon, unsigned long long* pinned_pub, unsigned long long seq);
int launch_lazy_flush(hipStream_t s, const LazyRec* lazy, NodeState* states, int n_voices, unsigned long long blocks,
                      const VoiceDesc* chain_voices = nullptr);  // chain plans: the voices (their delay lines' positions moved too)
int launch_sp_hist_copy(hipStream_t s, const FusedView& fv);
// realtime edge: control + leaf sums + root sum + interleave of ONE block in one launch (tree = leaves + root, stereo out);
// d_sync: one zero-initialised unsigned the workgroups count themselves in with
// d_done_flag (may be null): device view of a pinned host word that receives done_seq when the output block is complete
int launch_rt_persist(hipStream_t s, const FusedView& fv, const DevView& upv, const RootArgs& root, float* d_out, uint32_t cmd_block0,
                      unsigned* d_sync, unsigned long long* d_done_flag, RtMailbox* d_mb, unsigned long long* d_go, unsigned long long first_seq,
                      unsigned long long idle_ticks);
int launch_rt_block(hipStream_t s, const FusedView& fv, const DevView& upv, const RootArgs& root, float* d_out, uint32_t cmd_block0,
                    unsigned* d_sync, unsigned long long* d_done_flag, unsigned long long done_seq);
int launch_signal_done(hipStream_t s, unsigned long long* d_done_flag, unsigned long long done_seq);
// the R-port top-level SumNode over the shards' partial buses, rank order (16-byte aligned parts).  d_sil (may be null) =
// n_parts device pointers' worth of silence flags, each [n_blocks][n_ch] or null; d_out_sil (may be null) receives the node's
// out-mask per (block, channel)
int launch_bus_sum_ordered(hipStream_t s, const BusParts& bp, const uint8_t* const* sil, float* d_out, uint8_t* d_out_sil, size_t n_floats,
                           uint32_t n_blocks, uint32_t frames, uint32_t n_ch);
// one-shot exchange (k_exchange.hip.h): push this rank's bus + flags into its slot on every rank / wait for all arrivals and sum
int launch_bus_push(hipStream_t s, const ExchangePeers& peers, const ExchangeGeom& g, const float* d_part, const uint8_t* d_sil,
                    size_t n_floats, uint32_t n_sil, unsigned long long seq, unsigned* d_counter);
int launch_bus_reduce(hipStream_t s, char* base, const ExchangeGeom& g, float* d_out, uint8_t* d_out_sil, size_t n_floats, uint32_t n_sil,
                      uint32_t frames, uint32_t n_ch, unsigned long long seq, unsigned long long budget_ticks, unsigned long long* d_sync);
// host nodes: copy the pool buffers `bufs[0..n)` of K blocks (+ their silence flags) to / from the pinned staging area
// stage[k][j][stride], stage_flags[k][j]
int launch_host_gather(hipStream_t s, const float* pool, const uint8_t* flags, int stride, size_t pool_blk_stride, size_t flags_blk_stride,
                       const int* d_bufs, int n, int frames, int K, int row_pitch, float* d_stage, uint8_t* d_stage_flags);
int launch_host_scatter(hipStream_t s, float* pool, uint8_t* flags, int stride, size_t pool_blk_stride, size_t flags_blk_stride,
                        const int* d_bufs, int n, int frames, int K, int row_pitch, const float* d_stage, const uint8_t* d_stage_flags);
// plan adoption, one launch: ext-pool jobs (AdoptExtJobHost records: a new node's slice zeroed and / or its head floats set) and
// the initial NodeState records (StateInitHost) of the nodes the image activates
struct AdoptExtJobHost {
    uint32_t off, zero_len, n_head, pad;
    float head[8];
};
int launch_zero_rows(hipStream_t s, float* p, size_t pitch, int width, int rows);
int launch_set_row_heads(hipStream_t s, uint8_t* p, size_t pitch, int rows, uint8_t v);
int launch_build_apply(hipStream_t s, const BuildJob* jobs, int n_jobs);
int launch_adopt_init(hipStream_t s, float* ext, const void* d_jobs, int n_jobs, NodeState* states, const void* d_inits, int n_inits,
                      const CarryArgs& carry);
// per (block, channel) of a batch: was that graph-output channel flagged silent (mode: see k_out_flags)
int launch_out_flags(hipStream_t s, const uint8_t* flags, size_t flags_blk_stride, const int* d_bufs, int n_bufs, int mode, int n_out_ch, int K,
                     uint8_t* d_out);
// nq = tile size / 64 frames (1 or 2; 256-frame tiles measured slower: the serial stage then dominates the step):
// frames % (64*nq) == 0 and every delay line >= 64*nq frames
int launch_chain(hipStream_t s, const FusedView& fv, int K, uint32_t cmd_block0, int nq);

// initial floats of a node's ext-pool slice (biquad coefficients), consumed by k_scatter_ext
struct ExtInitHost {
    uint32_t off;  // float offset into the ext pool
    uint32_t n;    // floats to write (<= 6)
    float v[6];
};
int launch_scatter_ext(hipStream_t s, float* ext, const void* d_items, int n);

// host-side mirror of the StateInit record consumed by k_scatter_states
struct StateInitHost {
    int index;
    int pad;
    NodeState st;
};

}  // namespace fwgpu
