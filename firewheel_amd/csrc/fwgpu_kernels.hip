// fwgpu_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the Firewheel per-block DSP executor.
// ONE device translation unit; the kernels live in the k_*.hip.h files included below:
//   k_common.hip.h   SilenceMask, ParamSmoother, control->audio messages, sampler playhead logic, sample fetch
//   k_generic.hip.h  generic level-batched executor: k_level — one 64-lane wave per scheduled node, one launch per
//                    topological level for K blocks, bit-exact restatement of every node kind (nodes/*.rs + SPEC nodes)
//   k_control.hip.h  fused plans, control half: k_voice_control (per-voice per-block state machines, K blocks per launch)
//   k_leaf.hip.h     fused voice-bank plan: k_leaf_sum (HBM-streaming source fetch + gain stages + ordered radix-P sum
//                    in registers), k_bus_sum (upper sum tree), k_root_out (root sum + interleave)
//   k_chain.hip.h    fused chain plan: k_chain (sampler -> biquad -> delay -> gains -> leaf sum, LDS software pipeline)
//   k_fir.hip.h      FIR convolution bank: Toeplitz GEMM on the f32 matrix cores
//   k_rt.hip.h       realtime edge: one launch per callback for the voice-bank plan (control + leaf + root)
//   k_exchange.hip.h multi-GPU mix bus: one-shot exchange over peer-mapped slots + the rank-ordered top-level SumNode
// All plans share the node state in HBM.  Compiled with -ffp-contract=off: the reference (Rust) never fuses mul+add,
// and parity is bit-exact; the only fused multiply-adds are the ones a SPEC node asks for by name.
//
// Layout: planar f32, one channel-block = `stride` floats (multiple of 64 => every buffer is 256-B aligned,
// a wave's float4 access covers 1 KiB contiguous).  Reference citations: core/ nodes/ graph/ as in fwgpu.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fwgpu_launch.h"

namespace fwgpu {

#define WAVE 64
#define WPB 4  // waves (nodes) per workgroup in k_level / k_leaf_sum

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned vector (unaligned playheads)

__device__ __forceinline__ v4f splat(float x) { return (v4f){x, x, x, x}; }

#include "k_common.hip.h"
#include "k_generic.hip.h"
#include "k_control.hip.h"
#include "k_leaf.hip.h"
#include "k_chain.hip.h"
#include "k_fir.hip.h"
#include "k_rt.hip.h"
#include "k_exchange.hip.h"

// ------------------------------------------------------------------ launch wrappers (host side of this TU)
#define HIPCHK(x)                        \
    do {                                 \
        hipError_t e__ = (x);            \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// kinds: bit s = the level holds node kinds of set s (k_generic.hip.h: kind_set) — one launch per set present
int launch_level(hipStream_t s, const DevView& v, const int* d_level_nodes, int n_nodes, int K, uint32_t cmd_block0, int kinds) {
    if (n_nodes <= 0) return 0;
    if (K <= 0) return 0;
    // blocks per wave: enough (node, block) pairs to fill the chip several times over -> LEVEL_BPW; a narrow level -> 1
    // (FWGPU_LEVEL_BPW_WIDE: blocks per wave on a VERY wide level — hundreds of thousands of pairs: the per-wave prologue of a frozen
    //  node, five dependent round trips, is then amortised over more blocks; must divide 32: k_level's chain_done words)
    static const uint32_t bpw_wide = getenv("FWGPU_LEVEL_BPW_WIDE") ? (uint32_t)atoi(getenv("FWGPU_LEVEL_BPW_WIDE")) : (uint32_t)LEVEL_BPW_WIDE;
    const long long pairs = (long long)n_nodes * K;
    uint32_t bpw = pairs >= 16384 ? LEVEL_BPW : (pairs >= 8192 ? 2u : 1u);
    if (pairs >= 262144 && (bpw_wide == 8u || bpw_wide == 16u || bpw_wide == 32u)) bpw = bpw_wide;
    dim3 grid((n_nodes + WPB - 1) / WPB, (K + bpw - 1) / bpw);
    // kinds bit 3: the level holds a biquad / delay node — in a batch, the ones that are bus effects go to the walkers' kernel
    const uint32_t walkers = (kinds & 8) && K > 1 ? 1u : 0u;
    if (kinds & 1) hipLaunchKernelGGL(k_level<0>, grid, dim3(WAVE * WPB), 0, s, v, d_level_nodes, n_nodes, cmd_block0, (uint32_t)K, bpw, 0u);
    if (kinds & 2) hipLaunchKernelGGL(k_level<1>, grid, dim3(WAVE * WPB), 0, s, v, d_level_nodes, n_nodes, cmd_block0, (uint32_t)K, bpw, walkers);
    if (kinds & 4) hipLaunchKernelGGL(k_level<2>, grid, dim3(WAVE * WPB), 0, s, v, d_level_nodes, n_nodes, cmd_block0, (uint32_t)K, bpw, 0u);
    if (walkers)
        hipLaunchKernelGGL(k_bus_iir, dim3((n_nodes + WPB - 1) / WPB), dim3(WAVE * WPB), 0, s, v, d_level_nodes, n_nodes, cmd_block0, (uint32_t)K);
    return (int)hipGetLastError();
}
int launch_frozen_scan(hipStream_t s, const DevView& v, int n_nodes, uint32_t cmd_block0, int K, uint8_t* d_frozen,
                       unsigned long long* d_playhead_snap) {
    if (n_nodes <= 0) return 0;
    hipLaunchKernelGGL(k_frozen_scan, dim3((n_nodes + 255) / 256), dim3(256), 0, s, v, n_nodes, cmd_block0, (uint32_t)K, d_frozen,
                       d_playhead_snap);
    return (int)hipGetLastError();
}
int launch_bus_sum(hipStream_t s, const DevView& v, const int* d_level_nodes, int n_nodes, int K, int n_out) {
    if (n_nodes <= 0) return 0;
    dim3 grid(n_nodes, K, n_out);
    hipLaunchKernelGGL(k_bus_sum, grid, dim3(256), 0, s, v, d_level_nodes);
    return (int)hipGetLastError();
}
int launch_root_out(hipStream_t s, const DevView& v, const RootArgs& root, float* d_out, int K) {
    if (v.frames <= 0 || K <= 0) return 0;
    hipLaunchKernelGGL(k_root_out, dim3((v.frames + 255) / 256, K), dim3(256), 0, s, v, root, d_out);
    return (int)hipGetLastError();
}
int launch_ir_convert(hipStream_t s, const SampleDesc* samples, int sample, int ch, float* dst, uint32_t T) {
    hipLaunchKernelGGL(k_ir_convert, dim3((T + 255) / 256), dim3(256), 0, s, samples, sample, ch, dst, T);
    return (int)hipGetLastError();
}
int launch_fir(hipStream_t s, const DevView& v, const FirRow* d_rows, int n_rows, const uint32_t* d_tile_h_off, uint32_t T,
               float* d_partials, size_t partial_cap_floats, int K, hipEvent_t gemm_begin, hipEvent_t gemm_end) {
    if (n_rows <= 0 || v.frames <= 0 || K <= 0) return 0;
    const uint32_t W = T - 1u + (uint32_t)v.frames;
    const int n_segs = (int)((W + FIR_SEG - 1) / FIR_SEG);
    const int row_tiles = (n_rows + 31) / 32, n_rows_pad = row_tiles * 32;
    const int col_groups = (v.frames + 255) / 256, n_pad = col_groups * 256;
    if ((size_t)n_segs * n_rows_pad * n_pad * K > partial_cap_floats) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_fir_append, dim3(n_rows, K), dim3(256), 0, s, v, d_rows, n_rows);
    if (gemm_begin) (void)hipEventRecord(gemm_begin, s);
    hipLaunchKernelGGL(k_fir_gemm, dim3(row_tiles, n_segs, col_groups * K), dim3(256), 0, s, v, d_rows, n_rows, d_tile_h_off, T,
                       d_partials, n_rows_pad, n_pad, col_groups);
    if (gemm_end) (void)hipEventRecord(gemm_end, s);
    hipLaunchKernelGGL(k_fir_reduce, dim3(n_rows, K), dim3(256), 0, s, v, d_rows, n_rows, d_partials, n_segs, n_rows_pad,
                       n_pad);
    return (int)hipGetLastError();
}
int launch_single_node(hipStream_t s, const DevView& v, int node_idx) {
    hipLaunchKernelGGL(k_single_node, dim3(1), dim3(WAVE), 0, s, v, node_idx);
    return (int)hipGetLastError();
}
int launch_scatter_ext(hipStream_t s, float* ext, const void* d_items, int n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_scatter_ext, dim3((n + 63) / 64), dim3(64), 0, s, ext, (const ExtInitHost*)d_items, n);
    return (int)hipGetLastError();
}
int launch_scatter_states(hipStream_t s, NodeState* states, const void* d_inits, int n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_scatter_states, dim3((n + 63) / 64), dim3(64), 0, s, states, (const uint8_t*)d_inits, n);
    return (int)hipGetLastError();
}
int launch_graph_in(hipStream_t s, float* pool, uint8_t* flags, int stride, size_t pool_blk_stride, size_t flags_blk_stride,
                    const int* d_bufs, int n_bufs, const float* d_interleaved, int n_in_ch, int frames, int K) {
    if (n_bufs <= 0) return 0;
    dim3 grid((frames + 255) / 256, n_bufs, K);
    hipLaunchKernelGGL(k_graph_in, grid, dim3(256), 0, s, pool, flags, stride, pool_blk_stride, flags_blk_stride, d_bufs, n_bufs,
                       d_interleaved, n_in_ch, frames);
    return (int)hipGetLastError();
}
int launch_graph_out(hipStream_t s, const float* pool, const uint8_t* flags, int stride, size_t pool_blk_stride,
                     size_t flags_blk_stride, const int* d_bufs, int n_bufs, float* d_out, int n_out_ch, int frames, int K) {
    if (n_out_ch <= 0 || frames <= 0) return 0;
    dim3 grid((frames + 255) / 256, K);
    hipLaunchKernelGGL(k_graph_out, grid, dim3(256), 0, s, pool, flags, stride, pool_blk_stride, flags_blk_stride, d_bufs,
                       n_bufs, d_out, n_out_ch, frames);
    return (int)hipGetLastError();
}
int launch_set_flags(hipStream_t s, uint8_t* flags, const int* d_bufs, int n, uint64_t mask) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_set_flags, dim3(1), dim3(64), 0, s, flags, d_bufs, n, mask);
    return (int)hipGetLastError();
}
int launch_get_flags(hipStream_t s, const uint8_t* flags, const int* d_bufs, int n, uint64_t* d_mask) {
    hipLaunchKernelGGL(k_get_flags, dim3(1), dim3(64), 0, s, flags, d_bufs, n, d_mask);
    return (int)hipGetLastError();
}
int launch_voice_control(hipStream_t s, const FusedView& fv, int K, uint32_t cmd_block0, bool beside_render) {
    if (fv.n_voices <= 0) return 0;
    // one wave per voice.  beside_render (control-ahead mode): the 168-VGPR build in workgroups of ONE wave, so that the dispatcher
    // can place a control wave wherever one SIMD has room next to the render waves of the call before
    static const int occ = [] { const char* e = getenv("FWGPU_CTL_OCC"); return e ? atoi(e) : 0; }();  // experiments: 1 / 3 = force a build
    const bool small = occ ? occ == 3 : beside_render;
    const int wpb = beside_render ? 1 : 4;
    if (small) hipLaunchKernelGGL(k_voice_control_small, dim3((fv.n_voices + wpb - 1) / wpb), dim3(WAVE * wpb), 0, s, fv, K, cmd_block0);
    else hipLaunchKernelGGL(k_voice_control, dim3((fv.n_voices + wpb - 1) / wpb), dim3(WAVE * wpb), 0, s, fv, K, cmd_block0);
    return (int)hipGetLastError();
}
int launch_chain(hipStream_t s, const FusedView& fv, int K, uint32_t cmd_block0, int nq) {
    if (fv.n_groups <= 0 || K <= 0) return 0;
    // nq: bits 0..1 = tile size / 64 frames, bit 2 = the plan holds a voice with two biquads, bit 3 = ... with a gain stage between two
    // filters or a hard clip (fwgpu_ctx.h chain_nq)
    const bool bq2 = (nq & 4) != 0, sites = (nq & 8) != 0;
    nq &= 3;
    const dim3 grid(fv.n_groups, 2), block(CH_THREADS);
#define FW_CHAIN_LAUNCH(N, B, S) hipLaunchKernelGGL((k_chain<N, B, S>), grid, block, 0, s, fv, K, cmd_block0)
    if (nq == 2) {
        if (bq2) { if (sites) FW_CHAIN_LAUNCH(2, true, true); else FW_CHAIN_LAUNCH(2, true, false); }
        else { if (sites) FW_CHAIN_LAUNCH(2, false, true); else FW_CHAIN_LAUNCH(2, false, false); }
    } else {
        if (bq2) { if (sites) FW_CHAIN_LAUNCH(1, true, true); else FW_CHAIN_LAUNCH(1, true, false); }
        else { if (sites) FW_CHAIN_LAUNCH(1, false, true); else FW_CHAIN_LAUNCH(1, false, false); }
    }
#undef FW_CHAIN_LAUNCH
    return (int)hipGetLastError();
}
int launch_bus_sum_ordered(hipStream_t s, const BusParts& bp, const uint8_t* const* sil, float* d_out, uint8_t* d_out_sil, size_t n_floats,
                           uint32_t n_blocks, uint32_t frames, uint32_t n_ch) {
    if (bp.n <= 0 || n_floats == 0) return 0;
    const size_t n4 = n_floats / 4;
    const size_t threads = n4 ? n4 : 1;
    SilView sv;
    for (int p = 0; p < FW_MAX_BUS_PARTS; ++p) sv.sil[p] = (sil && p < bp.n) ? sil[p] : nullptr;
    if (!sil) n_blocks = 0;
    hipLaunchKernelGGL(k_bus_sum_ordered, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, bp, sv, d_out, d_out_sil, n4, n_floats,
                       n_blocks, frames, n_ch);
    return (int)hipGetLastError();
}
int launch_bus_push(hipStream_t s, const ExchangePeers& peers, const ExchangeGeom& g, const float* d_part, const uint8_t* d_sil,
                    size_t n_floats, uint32_t n_sil, unsigned long long seq, unsigned* d_counter) {
    const size_t n4 = n_floats / 4;
    const size_t threads = n4 ? n4 : 1;
    hipLaunchKernelGGL(k_bus_push, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, peers, g, d_part, d_sil, n_floats, n_sil, seq,
                       d_counter);
    return (int)hipGetLastError();
}
int launch_bus_reduce(hipStream_t s, char* base, const ExchangeGeom& g, float* d_out, uint8_t* d_out_sil, size_t n_floats, uint32_t n_sil,
                      uint32_t frames, uint32_t n_ch, unsigned long long seq, unsigned long long budget_ticks, unsigned long long* d_sync) {
    const size_t n4 = n_floats / 4;
    const size_t threads = n4 ? n4 : 1;
    hipLaunchKernelGGL(k_bus_wait, dim3(1), dim3(64), 0, s, base, g.world, seq, budget_ticks, d_sync);
    hipLaunchKernelGGL(k_bus_reduce, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, base, g, d_out, d_out_sil, n_floats, n_sil,
                       frames, n_ch, seq, (const unsigned long long*)d_sync);
    return (int)hipGetLastError();
}
int launch_host_gather(hipStream_t s, const float* pool, const uint8_t* flags, int stride, size_t pool_blk_stride, size_t flags_blk_stride,
                       const int* d_bufs, int n, int frames, int K, int row_pitch, float* d_stage, uint8_t* d_stage_flags) {
    if (n <= 0 || K <= 0 || frames <= 0) return 0;
    hipLaunchKernelGGL(k_host_gather, dim3((frames + 255) / 256, n, K), dim3(256), 0, s, pool, flags, stride, pool_blk_stride, flags_blk_stride,
                       d_bufs, frames, row_pitch, d_stage, d_stage_flags);
    return (int)hipGetLastError();
}
int launch_host_scatter(hipStream_t s, float* pool, uint8_t* flags, int stride, size_t pool_blk_stride, size_t flags_blk_stride,
                        const int* d_bufs, int n, int frames, int K, int row_pitch, const float* d_stage, const uint8_t* d_stage_flags) {
    if (n <= 0 || K <= 0 || frames <= 0) return 0;
    hipLaunchKernelGGL(k_host_scatter, dim3((frames + 255) / 256, n, K), dim3(256), 0, s, pool, flags, stride, pool_blk_stride, flags_blk_stride,
                       d_bufs, frames, row_pitch, d_stage, d_stage_flags);
    return (int)hipGetLastError();
}
int launch_zero_rows(hipStream_t s, float* p, size_t pitch, int width, int rows) {
    if (rows <= 0 || width <= 0) return 0;
    hipLaunchKernelGGL(k_zero_rows, dim3(rows < 1024 ? rows : 1024), dim3(256), 0, s, p, pitch, width, rows);
    return (int)hipGetLastError();
}
int launch_set_row_heads(hipStream_t s, uint8_t* p, size_t pitch, int rows, uint8_t v) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_set_row_heads, dim3((rows + 255) / 256), dim3(256), 0, s, p, pitch, rows, v);
    return (int)hipGetLastError();
}
int launch_build_apply(hipStream_t s, const BuildJob* jobs, int n_jobs) {
    if (n_jobs <= 0) return 0;
    hipLaunchKernelGGL(k_build_apply, dim3(32, n_jobs), dim3(256), 0, s, jobs);
    return (int)hipGetLastError();
}
int launch_adopt_init(hipStream_t s, float* ext, const void* d_jobs, int n_jobs, NodeState* states, const void* d_inits, int n_inits,
                      const CarryArgs& carry) {
    if (n_jobs <= 0 && n_inits <= 0 && carry.n_new <= 0) return 0;
    hipLaunchKernelGGL(k_adopt_init, dim3(16, n_jobs + 2), dim3(256), 0, s, ext, (const AdoptExtJob*)d_jobs, n_jobs, states, (const uint8_t*)d_inits,
                       n_inits, carry);
    return (int)hipGetLastError();
}
int launch_out_flags(hipStream_t s, const uint8_t* flags, size_t flags_blk_stride, const int* d_bufs, int n_bufs, int mode, int n_out_ch, int K,
                     uint8_t* d_out) {
    const int n = K * n_out_ch;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_out_flags, dim3((n + 255) / 256), dim3(256), 0, s, flags, flags_blk_stride, d_bufs, n_bufs, mode, n_out_ch, K, d_out);
    return (int)hipGetLastError();
}
int launch_rt_block(hipStream_t s, const FusedView& fv, const DevView& upv, const RootArgs& root, float* d_out, uint32_t cmd_block0,
                    unsigned* d_sync, unsigned long long* d_done_flag, unsigned long long done_seq) {
    if (fv.n_leaves <= 0) return 0;
    if (fv.has_rs)
        hipLaunchKernelGGL((k_rt_block<true, true>), dim3(fv.n_leaves), dim3(256), RS_LDS_BYTES(4), s, fv, upv, root, d_out, cmd_block0, d_sync,
                           d_done_flag, done_seq);
    else if (fv.has_prog)
        hipLaunchKernelGGL((k_rt_block<true, false>), dim3(fv.n_leaves), dim3(256), 0, s, fv, upv, root, d_out, cmd_block0, d_sync, d_done_flag,
                           done_seq);
    else
        hipLaunchKernelGGL((k_rt_block<false, false>), dim3(fv.n_leaves), dim3(256), 0, s, fv, upv, root, d_out, cmd_block0, d_sync, d_done_flag,
                           done_seq);
    return (int)hipGetLastError();
}
int launch_rt_persist(hipStream_t s, const FusedView& fv, const DevView& upv, const RootArgs& root, float* d_out, uint32_t cmd_block0,
                      unsigned* d_sync, unsigned long long* d_done_flag, RtMailbox* d_mb, unsigned long long* d_go, unsigned long long first_seq,
                      unsigned long long idle_ticks) {
    if (fv.n_leaves <= 0) return 0;
    static const int prefetch = [] { const char* e = getenv("FWGPU_RT_PREFETCH"); return e ? atoi(e) : 1; }();
    if (fv.has_rs)
        hipLaunchKernelGGL((k_rt_persist<true, true>), dim3(fv.n_leaves), dim3(256), RS_LDS_BYTES(4), s, fv, upv, root, d_out, cmd_block0, d_sync,
                           d_done_flag, d_mb, d_go, first_seq, idle_ticks, prefetch);
    else if (fv.has_prog)
        hipLaunchKernelGGL((k_rt_persist<true, false>), dim3(fv.n_leaves), dim3(256), 0, s, fv, upv, root, d_out, cmd_block0, d_sync, d_done_flag,
                           d_mb, d_go, first_seq, idle_ticks, prefetch);
    else
        hipLaunchKernelGGL((k_rt_persist<false, false>), dim3(fv.n_leaves), dim3(256), 0, s, fv, upv, root, d_out, cmd_block0, d_sync, d_done_flag,
                           d_mb, d_go, first_seq, idle_ticks, prefetch);
    return (int)hipGetLastError();
}
int launch_signal_done(hipStream_t s, unsigned long long* d_done_flag, unsigned long long done_seq) {
    hipLaunchKernelGGL(k_signal_done, dim3(1), dim3(1), 0, s, d_done_flag, done_seq);
    return (int)hipGetLastError();
}
int launch_sp_hist_copy(hipStream_t s, const FusedView& fv) {
    if (fv.n_voices <= 0) return 0;
    hipLaunchKernelGGL(k_sp_hist_copy, dim3((fv.n_voices * SP_HIST + 255) / 256), dim3(256), 0, s, fv);
    return (int)hipGetLastError();
}
int launch_leaf_sum_lazy(hipStream_t s, const FusedView& fv, int K) {
    if (fv.n_leaves <= 0) return 0;
    const int wpk = (fv.frames % (256 * LEAF_WPB) == 0) ? LEAF_WPB : (fv.frames % 512 == 0 && LEAF_WPB % 2 == 0) ? 2 : 1;
    const int bpw = LEAF_WPB / wpk;
    dim3 grid(fv.n_leaves, (K + bpw - 1) / bpw);
    if (fv.has_prog) hipLaunchKernelGGL((k_leaf_sum_lazy<true>), grid, dim3(WAVE * LEAF_WPB), 0, s, fv, K, wpk);
    else hipLaunchKernelGGL((k_leaf_sum_lazy<false>), grid, dim3(WAVE * LEAF_WPB), 0, s, fv, K, wpk);
    return (int)hipGetLastError();
}
int launch_lazy_publish(hipStream_t s, unsigned long long* d_horizon, unsigned long long* pinned_pub, unsigned long long seq) {
    hipLaunchKernelGGL(k_lazy_publish, dim3(1), dim3(1), 0, s, d_horizon, pinned_pub, seq);
    return (int)hipGetLastError();
}
int launch_lazy_flush(hipStream_t s, const LazyRec* lazy, NodeState* states, int n_voices, unsigned long long blocks, const VoiceDesc* chain_voices) {
    if (n_voices <= 0 || blocks == 0) return 0;
    hipLaunchKernelGGL(k_lazy_flush, dim3((n_voices + 255) / 256), dim3(256), 0, s, lazy, states, n_voices, blocks, chain_voices);
    return (int)hipGetLastError();
}
int launch_leaf_sum(hipStream_t s, const FusedView& fv, int K) {
    if (fv.n_leaves <= 0) return 0;
    // waves per block: long blocks are cut into 256-frame pieces so that every wave is one short streaming pass
    const int wpk = (fv.frames % (256 * LEAF_WPB) == 0) ? LEAF_WPB : (fv.frames % 512 == 0 && LEAF_WPB % 2 == 0) ? 2 : 1;
#if LEAF_MAP_BLOCKS
    const int bpw = LEAF_WPB / wpk;  // blocks per workgroup
    dim3 grid(fv.n_leaves, (K + bpw - 1) / bpw);
#else
    dim3 grid((fv.n_leaves + LEAF_WPB - 1) / LEAF_WPB, K);
#endif
    if (fv.has_sp && fv.has_rs) hipLaunchKernelGGL((k_leaf_sum<true, true, true>), grid, dim3(WAVE * LEAF_WPB), RS_LDS_BYTES(LEAF_WPB), s, fv, K, wpk);
    else if (fv.has_sp) {
        // one-piece blocks: a wave takes up to 8 consecutive blocks of its leaf (k_leaf.hip.h, leaf_kernel_body) while the launch still fills the chip
        int nb = 1;
        if (wpk == 1) {
            const long long pairs = (long long)fv.n_leaves * K;
            nb = (int)(pairs / 3072 < 1 ? 1 : (pairs / 3072 > 8 ? 8 : pairs / 3072));
        }
        const dim3 g2 = wpk == 1 ? dim3(fv.n_leaves, (K + LEAF_WPB * nb - 1) / (LEAF_WPB * nb)) : grid;
        hipLaunchKernelGGL(k_leaf_sum_sp, g2, dim3(WAVE * LEAF_WPB), 0, s, fv, K, wpk, nb);
    }
    else if (fv.has_rs) {  // a pair: the resampler-pure leaves (k_leaf_rs), then whatever it put on the work list
        hipLaunchKernelGGL(k_leaf_rs, grid, dim3(WAVE * LEAF_WPB), RS2_LDS_BYTES(LEAF_WPB), s, fv, K, wpk);
        // (a lazy call's records are resampler voices k_leaf_rs renders itself and silence — make_lazy, k_control.hip.h — so it leaves
        //  the work list empty: no second launch, which would read block records nobody wrote)
        if (fv.lazy_rs) return (int)hipGetLastError();
        const int items = fv.n_leaves * K * wpk;
        const int wgs = (items + LEAF_WPB - 1) / LEAF_WPB;
        hipLaunchKernelGGL(k_leaf_sum_wl, dim3(wgs < 768 ? wgs : 768), dim3(WAVE * LEAF_WPB), RS_LDS_BYTES(LEAF_WPB), s, fv, K, wpk);
    }
    else if (fv.has_prog) hipLaunchKernelGGL((k_leaf_sum<true, false>), grid, dim3(WAVE * LEAF_WPB), 0, s, fv, K, wpk);
    else hipLaunchKernelGGL((k_leaf_sum<false, false>), grid, dim3(WAVE * LEAF_WPB), 0, s, fv, K, wpk);
    return (int)hipGetLastError();
}

}  // namespace fwgpu

