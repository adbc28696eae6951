// TEST DOUBLE (see fakehip/hip/hip_runtime_api.h): the kernel launch wrappers of fwgpu_kernels.hip as counted no-ops.
// Nothing is computed; the harness only lets the host half of libfwgpu run on the CPU tier.
#include "../../firewheel_amd/csrc/fwgpu_launch.h"
#include "../../firewheel_amd/csrc/fwgpu_graph.h"  // PortInts (self-test below)

namespace {
unsigned long long g_launches[8];  // 0 level, 1 voice_control, 2 leaf_sum, 3 chain, 4 bus_sum, 5 root_out, 6 fir, 7 other
unsigned long long g_cmds_applied = 0;
unsigned long long g_ctl_orders = 0;  // control launches that carried a dispatch order (FusedView::ctl_order)
}
extern "C" unsigned long long fwh_cmds_seen(void) { return g_cmds_applied; }
extern "C" {
unsigned long long fwh_h2d_copies = 0, fwh_h2d_max_bytes = 0, fwh_h2d_bytes = 0;  // (fakehip's hipMemcpyAsync counts)
}
extern "C" void fwh_h2d_reset(void) {
    __atomic_store_n(&fwh_h2d_copies, 0ull, __ATOMIC_RELAXED);
    __atomic_store_n(&fwh_h2d_max_bytes, 0ull, __ATOMIC_RELAXED);
    __atomic_store_n(&fwh_h2d_bytes, 0ull, __ATOMIC_RELAXED);
}
namespace fwgpu { extern unsigned long long g_build_applies; }
extern "C" unsigned long long fwh_build_applies(void) { return fwgpu::g_build_applies; }
extern "C" int fwh_quiet_next_call_is_due(unsigned long long now, unsigned long long start, unsigned long long period, unsigned long long dur,
                                          unsigned long long margin) {
    return fwgpu::quiet_next_call_is_due(now, start, period, dur, margin) ? 1 : 0;
}
extern "C" unsigned long long fwh_h2d_total(void) { return __atomic_load_n(&fwh_h2d_bytes, __ATOMIC_RELAXED); }
extern "C" unsigned long long fwh_h2d_count(void) { return __atomic_load_n(&fwh_h2d_copies, __ATOMIC_RELAXED); }
extern "C" unsigned long long fwh_h2d_max(void) { return __atomic_load_n(&fwh_h2d_max_bytes, __ATOMIC_RELAXED); }
extern "C" unsigned long long fwh_ctl_orders(void) { return g_ctl_orders; }
extern "C" {
unsigned long long fwh_alloc_calls = 0;  // hipMalloc / hipHostMalloc calls of the fake runtime
unsigned long long fwh_alloc_count(void) { return fwh_alloc_calls; }
long long fwh_fail_alloc_in = 0;
void* fwh_foreign_maps[64] = {nullptr};
void fwh_fail_alloc(long long nth) { fwh_fail_alloc_in = nth; }
}
extern "C" unsigned long long fwh_launch_count(int which) { return which >= 0 && which < 8 ? g_launches[which] : 0; }
extern "C" void fwh_launch_reset(void) {
    for (auto& x : g_launches) x = 0;
}

// ---- what the stubs CHECK instead of computing: every table a kernel would index is touched at the extent the kernel
// indexes it (first and last byte: under ASan an under-sized allocation of the host side is a report), and the
// descriptor invariants the kernels rely on are asserted.  The first violation is kept for the tests (fwh_violation).
#include <mutex>
#include <stdio.h>

#include <map>
#include <string>
#include <vector>
namespace {
std::string g_violation;
void violation(const char* what, long a = 0, long b = 0) {
    if (!g_violation.empty()) return;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s (%ld, %ld)", what, a, b);
    g_violation = buf;
    fprintf(stderr, "host harness: descriptor invariant violated: %s\n", buf);
}
#define REQUIRE(cond, ...) \
    do {                   \
        if (!(cond)) violation(#cond, ##__VA_ARGS__); \
    } while (0)
thread_local volatile unsigned char g_sink;  // (per thread: the build thread touches too — launch_zero_rows)
void touch(const void* p, size_t bytes) {
    if (!bytes) return;
    if (!p) {
        violation("null table with a non-zero extent", (long)bytes);
        return;
    }
    g_sink = ((const volatile unsigned char*)p)[0];
    g_sink = ((const volatile unsigned char*)p)[bytes - 1];
}
}  // namespace
extern "C" const char* fwh_violation(void) { return g_violation.c_str(); }
extern "C" void fwh_violation_reset(void) { g_violation.clear(); }

namespace fwgpu {
namespace {
void check_generic_node(const DevView& v, int idx, int K) {
    touch(&v.nodes[idx], sizeof(NodeDesc));
    const NodeDesc nd = v.nodes[idx];
    REQUIRE(nd.n_in >= 0 && nd.n_in <= 64 && nd.n_out >= 0 && nd.n_out <= 64, nd.n_in, nd.n_out);
    touch(v.in_buf + nd.in_off, sizeof(int) * (size_t)nd.n_in);
    touch(v.out_buf + nd.out_off, sizeof(int) * (size_t)nd.n_out);
    for (int p = 0; p < nd.n_in + nd.n_out; ++p) {
        const int b = p < nd.n_in ? v.in_buf[nd.in_off + p] : v.out_buf[nd.out_off + p - nd.n_in];
        REQUIRE(b >= 0, b);
        REQUIRE(p < nd.n_in || b != 0, idx, p);  // nothing writes the constant zero buffer
        touch(v.pool + (size_t)(K - 1) * v.pool_blk_stride + (size_t)b * v.stride, sizeof(float) * (size_t)v.stride);
        touch(v.flags + (size_t)(K - 1) * v.flags_blk_stride + b, 1);
    }
    if (nd.state >= 0) touch(&v.states[nd.state], sizeof(NodeState));
    if (v.frozen) touch(v.frozen + idx, 1);
    if (v.frozen_playhead) touch(v.frozen_playhead + idx, 8);
    if (nd.kind == K_SUM) {  // aux0: port count (low half); high half, when set: the full port count of a split SumNode's continuation
        REQUIRE(nd.n_out > 0 && (nd.aux0 & 0xffff) * nd.n_out == nd.n_in, nd.aux0, nd.n_in);
        // (>=: a split that takes ONE leading voice port leaves a continuation of as many ports as the node has — partial bus + the rest)
        REQUIRE((nd.aux0 >> 16) == 0 || ((nd.aux0 >> 16) >= (nd.aux0 & 0xffff) && (nd.aux0 >> 16) <= 32), nd.aux0);
    } else if (nd.aux0 != 0) {
        // vertical fusion (k_generic.hip.h fz_links): the ONE consumer of this stereo node — a 2 -> 2 gain-like node that reads exactly
        // this node's two output buffers, channel for channel
        REQUIRE(nd.n_out == 2 && (nd.kind == K_SAMPLER || nd.kind == K_VOLUME || nd.kind == K_PAN || nd.kind == K_WIDTH || nd.kind == K_HARD_CLIP), nd.kind);
        const NodeDesc nx = v.nodes[nd.aux0 - 1];
        REQUIRE(nx.n_in == 2 && nx.n_out == 2 && (nx.kind == K_VOLUME || nx.kind == K_PAN || nx.kind == K_WIDTH || nx.kind == K_HARD_CLIP), nx.kind);
        REQUIRE(v.in_buf[nx.in_off] == v.out_buf[nd.out_off] && v.in_buf[nx.in_off + 1] == v.out_buf[nd.out_off + 1], idx, nd.aux0);
        if (v.chain_done) touch(v.chain_done + (size_t)(nd.aux0 - 1) * v.chain_words, 4 * (size_t)v.chain_words);
    }
    if (v.chain_done) REQUIRE(v.chain_words > 0 && K <= 32 * v.chain_words && v.frozen != nullptr, K, v.chain_words);
}
void check_view_common(const DevView& v, int K) {
    REQUIRE(K >= 1, K);
    REQUIRE(v.stride % 64 == 0 && v.frames >= 1 && v.frames <= v.stride, v.stride, v.frames);
    touch(v.cmds, sizeof(Cmd) * (size_t)v.n_cmds);
    for (int i = 1; i < v.n_cmds; ++i)  // sorted by (state, block): the device lookups are binary searches
        REQUIRE(v.cmds[i - 1].state < v.cmds[i].state || (v.cmds[i - 1].state == v.cmds[i].state && v.cmds[i - 1].block <= v.cmds[i].block), i);
}
void check_fused_common(const FusedView& fv, int K) {
    REQUIRE(K >= 1 && K <= fv.ref_kgroups * FW_REF_TILE_BLOCKS, K, fv.ref_kgroups);
    REQUIRE(fv.n_voices >= 1 && fv.n_leaves >= 1 && fv.stride % 64 == 0 && fv.frames >= 1 && fv.frames <= fv.stride, fv.n_voices, fv.frames);
    REQUIRE(fv.epoch >= 1);
    const size_t nv = (size_t)fv.n_voices;
    touch(fv.voices, sizeof(VoiceDesc) * nv);
    touch(fv.refs, sizeof(VoiceRef) * ref_count(nv, (size_t)fv.ref_kgroups * FW_REF_TILE_BLOCKS));  // tiled: ref_index()
    REQUIRE(ref_index(fv.n_voices - 1, K - 1, fv.ref_kgroups) < ref_count(nv, (size_t)fv.ref_kgroups * FW_REF_TILE_BLOCKS), K);
    touch(fv.gsets, sizeof(GainSet) * nv * FW_GSETS);
    touch(fv.cache, sizeof(VoiceCache) * nv);
    touch(fv.progs, sizeof(uint32_t) * nv);
    touch(fv.blks, sizeof(VoiceBlk) * nv * (size_t)K);
    touch(fv.ramps, sizeof(float) * nv * (size_t)K * (size_t)fv.ramp_slots * (size_t)fv.stride);
    touch(fv.cmds, sizeof(Cmd) * (size_t)fv.n_cmds);
    int max_stages = 0;
    for (int i = 0; i < fv.n_voices; ++i) {
        const VoiceDesc& vd = fv.voices[i];
        if (vd.sampler_state < 0) continue;  // a null voice: an unconnected leaf port
        REQUIRE(vd.n_stages >= 0 && vd.n_stages <= FW_MAX_STAGES - 1, i, vd.n_stages);
        max_stages = vd.n_stages > max_stages ? vd.n_stages : max_stages;
        touch(&fv.states[vd.sampler_state], sizeof(NodeState));
        for (int j = 0; j < vd.n_stages; ++j) {
            REQUIRE(vd.stage_kind[j] == K_VOLUME || vd.stage_kind[j] == K_PAN || vd.stage_kind[j] == K_WIDTH || vd.stage_kind[j] == K_HARD_CLIP ||
                        vd.stage_kind[j] == K_SPATIAL,
                    i, vd.stage_kind[j]);
            const uint32_t sk = (fv.progs[i] >> (4 * j)) & 15u;  // the leaf kernel's view of the same stage
            REQUIRE(sk == (vd.stage_kind[j] == K_WIDTH ? SK_WIDTH : vd.stage_kind[j] == K_HARD_CLIP ? SK_CLIP : vd.stage_kind[j] == K_SPATIAL ? SK_SPATIAL : SK_GAIN), i, (long)sk);
            if (vd.stage_kind[j] == K_SPATIAL) {  // the last stage of a dry sampler voice; its history slice and the call's scratch exist
                REQUIRE(j == vd.n_stages - 1 && vd.src_kind == 0 && vd.bq_state < 0 && vd.dl_state < 0 && fv.has_sp && fv.frames % 64 == 0, i, j);
                REQUIRE(vd.sp_ext_off >= 0, i, vd.sp_ext_off);
                touch(fv.ext + vd.sp_ext_off, sizeof(float) * SP_HIST);
                touch(fv.hist + (size_t)i * SP_HIST, sizeof(float) * SP_HIST);
            }
            REQUIRE(sk == SK_GAIN || (fv.has_prog && !fv.fx_plan) || (fv.fx_plan && sk == SK_CLIP), i, j);  // (k_chain clips per channel: round 6)
            touch(&fv.states[vd.stage_state[j]], sizeof(NodeState));
        }
        REQUIRE(fv.fx_plan || (vd.bq_state < 0 && vd.dl_state < 0 && vd.bq2_state < 0), i);
        // round 6 grammar: gain stages in front of the filters are volume / pan only and only in a voice that has a filter; a second biquad
        // needs a first; the delay-first order needs both kinds
        {
            const int m1 = vd.n_mid & 0xff, m2 = (vd.n_mid >> 8) & 0xff, nf = (vd.bq_state >= 0) + (vd.bq2_state >= 0) + (vd.dl_state >= 0);
            REQUIRE(vd.n_pre >= 0 && vd.n_pre + m1 + m2 <= vd.n_stages && (vd.n_pre + m1 + m2 == 0 || nf >= 1), i, vd.n_pre);
            REQUIRE((m1 == 0 || nf >= 2) && (m2 == 0 || nf >= 3) && (vd.n_mid >> 16) == 0, i, vd.n_mid);  // stages between filters need the filters
            if (nf) {
                REQUIRE(vd.n_stages <= FW_CHAIN_STAGES - 1, i, vd.n_stages);
                for (int j = 0; j < vd.n_stages; ++j) REQUIRE(vd.stage_kind[j] == K_VOLUME || vd.stage_kind[j] == K_PAN || vd.stage_kind[j] == K_HARD_CLIP, i, j);
            }
        }
        REQUIRE(vd.bq2_state < 0 || vd.bq_state >= 0, i, vd.bq2_state);
        REQUIRE((vd.fx_order == 0 || vd.fx_order == 1) && (vd.fx_order == 0 || (vd.bq_state >= 0 && vd.dl_state >= 0)), i, vd.fx_order);
        if (vd.bq2_state >= 0) touch(&fv.states[vd.bq2_state], sizeof(NodeState));
        // 0 sampler, 1 SPEC resampler (program instantiation, no chain plan), 2 a one-output sampler behind MonoToStereo: no spatialiser of its own (round 6: filters are fine)
        REQUIRE(vd.src_kind == 0 || (vd.src_kind == 1 && fv.has_prog && !fv.fx_plan && fv.rs_table != nullptr) ||
                    (vd.src_kind == 2 && vd.sp_ext_off < 0), i, vd.src_kind);
        if (vd.bq_state >= 0) touch(&fv.states[vd.bq_state], sizeof(NodeState));
        if (vd.dl_state >= 0) touch(&fv.states[vd.dl_state], sizeof(NodeState));
    }
    REQUIRE(fv.n_gain_stages >= 1 + max_stages && fv.n_gain_stages <= FW_MAX_STAGES, fv.n_gain_stages, max_stages);
    REQUIRE(fv.ramp_slots >= 2 * fv.n_gain_stages || fv.ramp_slots == 0, fv.ramp_slots, fv.n_gain_stages);
    // leaves tile the voice rows in order
    touch(fv.leaves, sizeof(LeafDesc) * (size_t)fv.n_leaves);
    int row = 0;
    for (int l = 0; l < fv.n_leaves; ++l) {
        const LeafDesc& ld = fv.leaves[l];
        REQUIRE(ld.first_voice == row && ld.ports >= 1 && ld.ports <= 32, l, ld.ports);
        row += ld.ports;
        REQUIRE(ld.out_buf >= 1, l, ld.out_buf);
        touch(fv.bus + (size_t)(K - 1) * fv.bus_blk_stride + (size_t)(ld.out_buf + 1) * fv.stride, sizeof(float) * (size_t)fv.stride);
        touch(fv.bus_flags + (size_t)(K - 1) * fv.bus_flags_blk_stride + ld.out_buf + 1, 1);
    }
    REQUIRE(row == fv.n_voices, row, fv.n_voices);
}
}  // namespace

// node states the voice-bank kernels own in the batch under way (filled by the control launch, dropped when the batch's
// graph_out / root launch comes by): in a hybrid batch no level list may name one of them — a node is rendered ONCE
static unsigned char g_fused_states[1 << 16];  // (a flat table: the stubs run on the audio thread of the allocation tests)
static const void* g_fused_ctx_states = nullptr;  // whose states they index
int launch_level(hipStream_t, const DevView& v, const int* d_level_nodes, int n_nodes, int K, uint32_t, int kinds) {
    g_launches[0]++;
    check_view_common(v, K);
    REQUIRE(kinds >= 0 && kinds <= 15, kinds);  // 0: a level of Dummy / graph I/O / FIR nodes only — nothing to launch; bit 3: holds a biquad / delay
    touch(d_level_nodes, sizeof(int) * (size_t)n_nodes);
    for (int i = 0; i < n_nodes; ++i) {
        check_generic_node(v, d_level_nodes[i], K);
        const NodeDesc& nd = v.nodes[d_level_nodes[i]];
        if (v.pool_blk_stride && (nd.kind == K_SAMPLER || nd.kind == K_RESAMPLER || nd.kind == K_VOLUME || nd.kind == K_PAN || nd.kind == K_WIDTH ||
                                  nd.kind == K_HARD_CLIP) && v.cmds != nullptr)
            REQUIRE(v.states != g_fused_ctx_states || !(nd.state >= 0 && nd.state < (1 << 16) && g_fused_states[nd.state]), d_level_nodes[i], nd.state);
    }
    return 0;
}
int launch_frozen_scan(hipStream_t, const DevView& v, int n_nodes, uint32_t, int K, uint8_t* d_frozen, unsigned long long* d_snap) {
    g_launches[7]++;
    REQUIRE(K >= 1, K);
    touch(v.nodes, sizeof(NodeDesc) * (size_t)n_nodes);
    touch(d_frozen, (size_t)n_nodes);
    touch(d_snap, 8 * (size_t)n_nodes);
    if (v.chain_done) {
        REQUIRE(v.chain_words > 0 && K <= 32 * v.chain_words, K, v.chain_words);
        touch(v.chain_done, 4 * (size_t)n_nodes * (size_t)v.chain_words);
    }
    return 0;
}
int launch_bus_sum(hipStream_t, const DevView& v, const int* d_level_nodes, int n_nodes, int K, int n_out) {
    g_launches[4]++;
    check_view_common(v, K);
    touch(d_level_nodes, sizeof(int) * (size_t)n_nodes);
    for (int i = 0; i < n_nodes; ++i) {
        check_generic_node(v, d_level_nodes[i], K);
        REQUIRE(v.nodes[d_level_nodes[i]].kind == K_SUM && v.nodes[d_level_nodes[i]].n_out <= n_out, i, n_out);
    }
    return 0;
}
int launch_root_out(hipStream_t, const DevView& v, const RootArgs& root, float* d_out, int K) {
    g_launches[5]++;
    g_fused_ctx_states = nullptr;  // the batch ends here
    check_view_common(v, K);
    REQUIRE(root.n_in >= 2 && root.n_in <= 64 && root.ports * 2 == root.n_in, root.n_in, root.ports);
    touch(root.in_tab, sizeof(int) * (size_t)root.n_in);
    for (int i = 0; i < root.n_in; ++i) {
        REQUIRE(root.in_tab[i] == root.in_buf[i] && root.in_buf[i] >= 0, i, root.in_buf[i]);
        touch(v.pool + (size_t)(K - 1) * v.pool_blk_stride + (size_t)root.in_buf[i] * v.stride, sizeof(float) * (size_t)v.stride);
        touch(v.flags + (size_t)(K - 1) * v.flags_blk_stride + root.in_buf[i], 1);
    }
    touch(d_out, sizeof(float) * 2 * (size_t)v.frames * (size_t)K);
    return 0;
}
int launch_ir_convert(hipStream_t, const SampleDesc* samples, int sample, int, float* dst, uint32_t T) {
    g_launches[7]++;
    touch(&samples[sample], sizeof(SampleDesc));
    touch(dst, sizeof(float) * (size_t)T);
    return 0;
}
int launch_fir(hipStream_t, const DevView& v, const FirRow* d_rows, int n_rows, const uint32_t* d_tile_h_off, uint32_t T, float*, size_t, int K,
               hipEvent_t, hipEvent_t) {
    g_launches[6]++;
    check_view_common(v, K);
    REQUIRE(n_rows >= 1 && n_rows % 32 == 0 && T >= 1, n_rows, (long)T);  // rows padded to 32-row tiles
    touch(d_rows, sizeof(FirRow) * (size_t)n_rows);
    touch(d_tile_h_off, sizeof(uint32_t) * (size_t)(n_rows / 32));
    for (int t = 0; t < n_rows / 32; ++t) touch(v.ext + d_tile_h_off[t], sizeof(float) * (size_t)T);
    return 0;
}
int launch_single_node(hipStream_t, const DevView& v, int node_idx) {
    g_launches[7]++;
    check_view_common(v, 1);
    check_generic_node(v, node_idx, 1);
    return 0;
}
int launch_scatter_states(hipStream_t, NodeState* states, const void* d_inits, int n) {
    g_launches[7]++;
    touch(d_inits, sizeof(StateInitHost) * (size_t)n);
    for (int i = 0; i < n; ++i) {  // plain data movement, done for real: the checks above read node state (delay lengths)
        const StateInitHost& it = ((const StateInitHost*)d_inits)[i];
        touch(&states[it.index], sizeof(NodeState));
        states[it.index] = it.st;
    }
    return 0;
}
int launch_graph_in(hipStream_t, float* pool, uint8_t* flags, int stride, size_t pbs, size_t fbs, const int* d_bufs, int n_bufs, const float* d_in,
                    int n_in_ch, int frames, int K) {
    g_launches[7]++;
    touch(d_bufs, sizeof(int) * (size_t)n_bufs);
    for (int i = 0; i < n_bufs; ++i) {
        touch(pool + (size_t)(K - 1) * pbs + (size_t)d_bufs[i] * stride, sizeof(float) * (size_t)stride);
        touch(flags + (size_t)(K - 1) * fbs + d_bufs[i], 1);
    }
    touch(d_in, sizeof(float) * (size_t)n_in_ch * (size_t)frames * (size_t)K);
    return 0;
}
int launch_graph_out(hipStream_t, const float* pool, const uint8_t* flags, int stride, size_t pbs, size_t fbs, const int* d_bufs, int n_bufs,
                     float* d_out, int n_out_ch, int frames, int K) {
    g_launches[7]++;
    g_fused_ctx_states = nullptr;  // the batch ends here
    touch(d_bufs, sizeof(int) * (size_t)n_bufs);
    for (int i = 0; i < n_bufs; ++i) {
        touch(pool + (size_t)(K - 1) * pbs + (size_t)d_bufs[i] * stride, sizeof(float) * (size_t)stride);
        touch(flags + (size_t)(K - 1) * fbs + d_bufs[i], 1);
    }
    touch(d_out, sizeof(float) * (size_t)n_out_ch * (size_t)frames * (size_t)K);
    return 0;
}
int launch_set_flags(hipStream_t, uint8_t* flags, const int* d_bufs, int n, uint64_t) {
    g_launches[7]++;
    touch(d_bufs, sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) touch(flags + d_bufs[i], 1);
    return 0;
}
int launch_get_flags(hipStream_t, const uint8_t* flags, const int* d_bufs, int n, uint64_t* d_mask) {
    g_launches[7]++;
    touch(d_bufs, sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) touch(flags + d_bufs[i], 1);
    touch(d_mask, 8);
    return 0;
}
// Lazy records (fwgpu_types.h LazyRec), as the harness can model them: every voice ends every control launch steady and plain
// (the publish stub reports an unbounded horizon).  What IS checked is the host's book-keeping: a lazy leaf launch names the block
// offset that follows from the launches since the control launch its LazyRecs came from; nothing reads or moves node state — a
// control launch, the realtime kernels — while lazily rendered blocks have not been flushed into it; a flush names exactly them.
struct LazyBook {
    unsigned long long since_ctl = 0, unflushed = 0;
    bool have_ctl = false;
    bool chain = false;  // the unflushed blocks were rendered by k_chain
};
static LazyBook g_lazy[64];
static const void* g_lazy_key[64];
static std::mutex g_lazy_mu;  // (launches come from the audio thread, frees from whichever thread retires an image)
static LazyBook& lazy_book(const void* key) {
    std::lock_guard<std::mutex> lk(g_lazy_mu);
    for (int i = 0; i < 64; ++i) {
        if (g_lazy_key[i] == key) return g_lazy[i];
        if (!g_lazy_key[i]) {
            g_lazy_key[i] = key;
            g_lazy[i] = LazyBook();
            return g_lazy[i];
        }
    }
    g_lazy_key[0] = key;  // (more than 64 plans alive in one process: start over)
    g_lazy[0] = LazyBook();
    return g_lazy[0];
}
// (a freed LazyRec table takes its book along: the next context's table may get the same address)
extern "C" void fwh_freed(const void* p) {
    std::lock_guard<std::mutex> lk(g_lazy_mu);
    for (int i = 0; i < 64; ++i)
        if (g_lazy_key[i] == p) g_lazy[i] = LazyBook();
}
unsigned long long g_lazy_launches = 0;
extern "C" unsigned long long fwh_lazy_launches(void) { return g_lazy_launches; }
int launch_leaf_sum_lazy(hipStream_t, const FusedView& fv, int K) {
    g_launches[2]++;
    g_lazy_launches++;
    check_fused_common(fv, K);
    REQUIRE(!fv.fx_plan && fv.lazy != nullptr && !fv.has_rs && !fv.has_sp, fv.has_rs, fv.has_sp);
    touch(fv.lazy, sizeof(LazyRec) * (size_t)fv.n_voices);
    LazyBook& b = lazy_book(fv.lazy);
    REQUIRE(b.have_ctl && fv.lazy_blk0 == b.since_ctl, (long long)fv.lazy_blk0, (long long)b.since_ctl);
    b.since_ctl += (unsigned long long)K;
    b.unflushed += (unsigned long long)K;
    b.chain = false;
    return 0;
}
int launch_lazy_publish(hipStream_t, unsigned long long* d_horizon, unsigned long long* pub, unsigned long long seq) {
    g_launches[7]++;
    touch(d_horizon, 8);
    touch(pub, 16);
    pub[0] = ~0ull;  // (the fake device is done when the launch returns)
    pub[1] = seq;
    return 0;
}
int launch_lazy_flush(hipStream_t, const LazyRec* lazy, NodeState* states, int n_voices, unsigned long long blocks, const VoiceDesc* chain_voices) {
    g_launches[7]++;
    touch(lazy, sizeof(LazyRec) * (size_t)n_voices);
    (void)states;
    LazyBook& b = lazy_book(lazy);
    // (a chain plan's flush moves the delay lines too: it names the voices — exactly when the lazy launches were k_chain's)
    REQUIRE((chain_voices != nullptr) == b.chain, (int)b.chain);
    if (chain_voices) touch(chain_voices, sizeof(VoiceDesc) * (size_t)n_voices);
    REQUIRE(blocks == b.unflushed && blocks > 0, (long long)blocks, (long long)b.unflushed);
    b.unflushed = 0;
    b.chain = false;
    b.have_ctl = false;  // (the LazyRecs are spent: the next lazy launch needs a control launch first)
    return 0;
}
int launch_voice_control(hipStream_t, const FusedView& fv, int K, uint32_t cmd_block0, bool) {
    g_launches[1]++;
    check_fused_common(fv, K);
    if (fv.lazy) {
        touch(fv.lazy, sizeof(LazyRec) * (size_t)fv.n_voices);
        touch(fv.horizon, 8);
        LazyBook& b = lazy_book(fv.lazy);
        REQUIRE(b.unflushed == 0, (long long)b.unflushed);  // node state is current when the state machines run
        b.since_ctl = 0;
        b.have_ctl = true;
    }
    for (int i = 0; i < fv.n_cmds; ++i) {  // messages the control kernel would APPLY in this launch (each exactly once in its life)
        if (i > 0) {
            const Cmd &a = fv.cmds[i - 1], &b = fv.cmds[i];
            REQUIRE(a.state < b.state || (a.state == b.state && a.block <= b.block), i);  // sorted by (node, block)
        }
        if (fv.cmds[i].block >= cmd_block0 && fv.cmds[i].block < cmd_block0 + (uint32_t)K) g_cmds_applied++;
    }
    if (fv.fx_plan) touch(fv.chain_start, sizeof(ChainStart) * (size_t)fv.n_voices);
    if (fv.ctl_order) {
        // the dispatch order (upload_cmds): a permutation of the voices, and every voice a message of this call goes to sits in
        // front of every voice that has none (and did not have one in the call before: those may sit in front as well)
        static int pos[1 << 16];   // (fixed: the stubs run on the audio thread of the allocation-counting tests)
        static char has[1 << 16];
        REQUIRE(fv.n_voices <= (1 << 16), fv.n_voices);
        for (int v = 0; v < fv.n_voices; ++v) {
            pos[v] = -1;
            has[v] = 0;
        }
        for (int w = 0; w < fv.n_voices; ++w) {
            const int v = fv.ctl_order[w];
            REQUIRE(v >= 0 && v < fv.n_voices && pos[(size_t)v] < 0, w, v);
            pos[(size_t)v] = w;
        }
        int n_has = 0;
        for (int i = 0; i < fv.n_cmds; ++i)
            for (int v = 0; v < fv.n_voices; ++v) {
                const VoiceDesc& vd = fv.voices[v];
                bool mine = vd.sampler_state == fv.cmds[i].state || vd.bq_state == fv.cmds[i].state || vd.dl_state == fv.cmds[i].state ||
                            vd.bq2_state == fv.cmds[i].state;
                for (int j = 0; j < vd.n_stages && j < FW_MAX_STAGES - 1; ++j) mine = mine || vd.stage_state[j] == fv.cmds[i].state;
                if (mine && fv.cmds[i].state >= 0 && !has[(size_t)v]) {
                    has[(size_t)v] = 1;
                    ++n_has;
                }
            }
        for (int v = 0; v < fv.n_voices; ++v)
            if (has[(size_t)v]) REQUIRE(pos[(size_t)v] < n_has, v, pos[(size_t)v]);
        g_ctl_orders++;
    }
    memset(g_fused_states, 0, sizeof(g_fused_states));
    g_fused_ctx_states = fv.states;
    for (int i = 0; i < fv.n_voices; ++i) {
        const VoiceDesc& vd = fv.voices[i];
        if (vd.sampler_state < 0 || vd.sampler_state >= (1 << 16)) continue;
        g_fused_states[vd.sampler_state] = 1;
        for (int j = 0; j < vd.n_stages; ++j)
            if (vd.stage_state[j] >= 0 && vd.stage_state[j] < (1 << 16)) g_fused_states[vd.stage_state[j]] = 1;
    }
    return 0;
}
int launch_bus_sum_ordered(hipStream_t, const BusParts& bp, const uint8_t* const* sil, float* d_out, uint8_t* d_out_sil, size_t n_floats,
                           uint32_t n_blocks, uint32_t frames, uint32_t n_ch) {
    g_launches[7]++;
    REQUIRE(bp.n >= 1 && bp.n <= FW_MAX_BUS_PARTS, bp.n);
    for (int r = 0; r < bp.n; ++r) touch(bp.part[r], n_floats * sizeof(float));
    touch(d_out, n_floats * sizeof(float));
    if (sil) {
        REQUIRE((size_t)n_blocks * frames * n_ch >= n_floats, (long)n_blocks, (long)frames);
        for (int r = 0; r < bp.n; ++r)
            if (sil[r]) touch(sil[r], (size_t)n_blocks * n_ch);
        if (d_out_sil) touch(d_out_sil, (size_t)n_blocks * n_ch);
    }
    return 0;
}
int launch_bus_push(hipStream_t, const ExchangePeers& peers, const ExchangeGeom& g, const float* d_part, const uint8_t* d_sil, size_t n_floats,
                    uint32_t n_sil, unsigned long long seq, unsigned* d_counter) {
    g_launches[7]++;
    REQUIRE(g.world >= 1 && g.world <= FW_MAX_BUS_PARTS && g.rank >= 0 && g.rank < g.world, g.world, g.rank);
    REQUIRE(n_floats <= g.max_floats && g.max_floats * 4 + n_sil <= g.slot_bytes, (long)n_floats, (long)n_sil);
    REQUIRE(seq >= 1, (long)seq);
    touch(d_part, n_floats * sizeof(float));
    if (n_sil) touch(d_sil, n_sil);
    touch(d_counter, 4);
    for (int p = 0; p < g.world; ++p) {  // this rank's slot in every region: the last byte of the second parity
        REQUIRE(peers.base[p] != nullptr, p);
        if (peers.base[p]) touch(peers.base[p], 4096 + (size_t)(2 * g.world) * g.slot_bytes);
    }
    return 0;
}
int launch_bus_reduce(hipStream_t, char* base, const ExchangeGeom& g, float* d_out, uint8_t* d_out_sil, size_t n_floats, uint32_t n_sil,
                      uint32_t frames, uint32_t n_ch, unsigned long long seq, unsigned long long budget_ticks, unsigned long long* d_sync) {
    g_launches[7]++;
    touch(d_sync, (8 + FW_MAX_BUS_PARTS) * 8);
    REQUIRE(n_floats <= g.max_floats && g.max_floats * 4 + n_sil <= g.slot_bytes, (long)n_floats, (long)n_sil);
    REQUIRE(budget_ticks > 0 && seq >= 1, (long)seq);
    if (n_sil) REQUIRE(n_ch > 0 && (size_t)(n_sil / n_ch) * frames * n_ch >= n_floats, (long)n_sil, (long)frames);
    touch(base, 4096 + (size_t)(2 * g.world) * g.slot_bytes);
    touch(d_out, n_floats * sizeof(float));
    if (d_out_sil) touch(d_out_sil, n_sil);
    return 0;
}
int launch_host_gather(hipStream_t, const float* pool, const uint8_t* flags, int stride, size_t pool_blk_stride, size_t flags_blk_stride,
                       const int* d_bufs, int n, int frames, int K, int row_pitch, float* d_stage, uint8_t* d_stage_flags) {
    g_launches[7]++;
    REQUIRE(n >= 1 && n <= 64 && K >= 1 && frames >= 1 && frames <= stride && row_pitch >= n, n, row_pitch);
    touch(d_bufs, sizeof(int) * (size_t)n);
    for (int j = 0; j < n; ++j) {
        touch(pool + (size_t)(K - 1) * pool_blk_stride + (size_t)d_bufs[j] * stride, sizeof(float) * (size_t)frames);
        touch(flags + (size_t)(K - 1) * flags_blk_stride + d_bufs[j], 1);
    }
    touch(d_stage, sizeof(float) * (((size_t)(K - 1) * row_pitch + (n - 1)) * stride + frames));
    touch(d_stage_flags, (size_t)(K - 1) * row_pitch + n);
    return 0;
}
int launch_host_scatter(hipStream_t, float* pool, uint8_t* flags, int stride, size_t pool_blk_stride, size_t flags_blk_stride, const int* d_bufs,
                        int n, int frames, int K, int row_pitch, const float* d_stage, const uint8_t* d_stage_flags) {
    g_launches[7]++;
    REQUIRE(n >= 1 && n <= 64 && K >= 1 && frames >= 1 && frames <= stride && row_pitch >= n, n, row_pitch);
    touch(d_bufs, sizeof(int) * (size_t)n);
    for (int j = 0; j < n; ++j) {
        REQUIRE(d_bufs[j] != 0, j);  // buffer 0 is the constant zero buffer: never an output
        touch(pool + (size_t)(K - 1) * pool_blk_stride + (size_t)d_bufs[j] * stride, sizeof(float) * (size_t)frames);
        touch(flags + (size_t)(K - 1) * flags_blk_stride + d_bufs[j], 1);
    }
    touch(d_stage, sizeof(float) * (((size_t)(K - 1) * row_pitch + (n - 1)) * stride + frames));
    touch(d_stage_flags, (size_t)(K - 1) * row_pitch + n);
    return 0;
}
int launch_zero_rows(hipStream_t, float* p, size_t pitch, int width, int rows) {
    for (int r = 0; r < rows; ++r) {
        touch(p + (size_t)r * pitch, sizeof(float) * (size_t)width);
        for (int i = 0; i < width; ++i) p[(size_t)r * pitch + i] = 0.f;
    }
    return 0;
}
int launch_set_row_heads(hipStream_t, uint8_t* p, size_t pitch, int rows, uint8_t v) {
    for (int r = 0; r < rows; ++r) p[(size_t)r * pitch] = v;
    return 0;
}
unsigned long long g_build_applies = 0;
int launch_build_apply(hipStream_t, const BuildJob* jobs, int n_jobs) {
    g_build_applies++;
    REQUIRE(jobs != nullptr && n_jobs > 0, (long)n_jobs);
    for (int i = 0; i < n_jobs; ++i) {
        const BuildJob& j = jobs[i];
        REQUIRE(j.dst != nullptr && j.row_bytes > 0 && j.rows > 0, (long)i);
        if (j.src) {
            REQUIRE(j.rows == 1, (long)i);
            memmove(j.dst, j.src, j.row_bytes);
            __atomic_fetch_add(&fwh_h2d_bytes, (unsigned long long)j.row_bytes, __ATOMIC_RELAXED);  // (counted like a copy call's)
        } else {
            for (uint32_t r = 0; r < j.rows; ++r) {
                memset((char*)j.dst + (size_t)r * j.pitch, (int)j.value, j.row_bytes);
                if (j.head >= 0) ((unsigned char*)j.dst)[(size_t)r * j.pitch] = (unsigned char)j.head;
            }
        }
    }
    return 0;
}
static int check_carry(const CarryArgs& a) {
    if (a.n_new <= 0) return 0;
    VoiceCache* new_cache = a.new_cache;
    const VoiceDesc* new_voices = a.new_voices;
    const int n_new = a.n_new;
    const VoiceCache* old_cache = a.old_cache;
    const VoiceDesc* old_voices = a.old_voices;
    const int* old_slot_voice = a.old_slot_voice;
    const int n_old_slots = a.n_old_slots;
    const uint32_t old_epoch = a.old_epoch, new_epoch = a.new_epoch;
    REQUIRE(new_cache && new_voices && old_cache && old_voices && old_slot_voice && new_epoch != old_epoch, n_new, n_old_slots);
    touch(new_cache, sizeof(VoiceCache) * (size_t)n_new);
    touch(new_voices, sizeof(VoiceDesc) * (size_t)n_new);
    touch(old_slot_voice, sizeof(int) * (size_t)n_old_slots);
    for (int v = 0; v < n_new; ++v) {  // (the harness runs no kernels: what matters is that every index stays inside its table)
        const int s = new_voices[v].sampler_state;
        if (s < 0 || s >= n_old_slots) continue;
        const int vo = old_slot_voice[s];
        if (vo < 0) continue;
        touch(old_voices + vo, sizeof(VoiceDesc));
        touch(old_cache + vo, sizeof(VoiceCache));
    }
    return 0;
}
int launch_adopt_init(hipStream_t, float* ext, const void* d_jobs, int n_jobs, NodeState* states, const void* d_inits, int n_inits,
                      const CarryArgs& carry) {
    if (check_carry(carry)) return -1;
    g_launches[7]++;
    const AdoptExtJobHost* jobs = (const AdoptExtJobHost*)d_jobs;
    touch(d_jobs, sizeof(AdoptExtJobHost) * (size_t)n_jobs);
    for (int i = 0; i < n_jobs; ++i) {
        REQUIRE(jobs[i].zero_len % 64 == 0 && jobs[i].n_head <= 8 && (jobs[i].zero_len || jobs[i].n_head), i);
        const size_t n = jobs[i].zero_len > jobs[i].n_head ? jobs[i].zero_len : jobs[i].n_head;
        touch(ext + jobs[i].off, sizeof(float) * n);
        for (size_t k = 0; k < n; ++k) ext[jobs[i].off + k] = k < jobs[i].n_head ? jobs[i].head[k] : 0.f;
    }
    const StateInitHost* in = (const StateInitHost*)d_inits;
    touch(d_inits, sizeof(StateInitHost) * (size_t)n_inits);
    for (int i = 0; i < n_inits; ++i) {
        REQUIRE(in[i].index >= 0, in[i].index);
        touch(&states[in[i].index], sizeof(NodeState));
        states[in[i].index] = in[i].st;
    }
    return 0;
}
int launch_out_flags(hipStream_t, const uint8_t* flags, size_t flags_blk_stride, const int* d_bufs, int n_bufs, int mode, int n_out_ch, int K,
                     uint8_t* d_out) {
    g_launches[7]++;
    REQUIRE(mode == 0 || mode == 1, mode);
    REQUIRE(n_bufs >= 0 && n_bufs <= 64 && K >= 1, n_bufs, K);
    touch(d_bufs, sizeof(int) * (size_t)n_bufs);
    for (int j = 0; j < n_bufs; ++j) touch(flags + (size_t)(K - 1) * flags_blk_stride + d_bufs[j], 1);
    touch(d_out, (size_t)K * n_out_ch);
    return 0;
}
int launch_signal_done(hipStream_t, unsigned long long* d_done_flag, unsigned long long done_seq) {
    g_launches[7]++;
    touch(d_done_flag, 8);
    if (d_done_flag) *d_done_flag = done_seq;  // (the fake device is done when the launch returns)
    return 0;
}
int launch_rt_block(hipStream_t, const FusedView& fv, const DevView& upv, const RootArgs& root, float* d_out, uint32_t cmd_block0,
                    unsigned* d_sync, unsigned long long* d_done_flag, unsigned long long done_seq) {
    if (d_done_flag) *d_done_flag = done_seq;
    g_launches[7]++;
    check_fused_common(fv, 1);
    REQUIRE(fv.lazy == nullptr);  // (the one-launch kernels run their own control and leave no LazyRecs)
    REQUIRE(!fv.fx_plan && root.ports >= 1 && root.ports <= 32 && root.n_in == 2 * root.ports, root.ports, root.n_in);
    touch(d_sync, sizeof(unsigned));
    touch(d_out, sizeof(float) * 2 * (size_t)upv.frames);
    {
        // the way up the mixer tree (k_rt.hip.h): every leaf's chain of consumers ends at the root, and every node on the way waits for
        // exactly the arrivals it will get (one per leaf / upper node that names it)
        REQUIRE(fv.rt_parent_leaf && fv.rt_parent_up && fv.rt_kids && fv.rt_tree_sync && fv.rt_root >= 0, fv.rt_root);
        std::map<int, int> arrivals;
        std::map<int, bool> seen;
        for (int i = 0; i < fv.n_leaves && fv.rt_parent_leaf; ++i) {
            touch(fv.rt_parent_leaf + i, sizeof(int));
            int node = fv.rt_parent_leaf[i];
            arrivals[node]++;
            for (int hops = 0; hops < 70; ++hops) {
                REQUIRE(node >= 0 && hops < 64, i, node);
                if (node < 0) break;
                touch(fv.rt_kids + node, sizeof(int));
                touch(fv.rt_tree_sync + node, sizeof(unsigned));
                REQUIRE(fv.rt_tree_sync[node] == 0u, node);  // (between callbacks every counter rests at 0)
                if (node == fv.rt_root) break;
                touch(fv.rt_parent_up + node, sizeof(int));
                {  // a mixer between the leaves and the root: rendered by bus_sum_node_wg from the upper tree's tables
                    REQUIRE(upv.nodes && upv.in_buf && upv.out_buf, node);
                    const NodeDesc nd = upv.nodes[node];
                    REQUIRE(nd.kind == K_SUM && nd.n_out == 2 && nd.aux0 * 2 == nd.n_in, node, nd.n_in);
                    touch(upv.in_buf + nd.in_off, sizeof(int) * (size_t)nd.n_in);
                    touch(upv.out_buf + nd.out_off, sizeof(int) * (size_t)nd.n_out);
                }
                if (!seen[node]) {
                    seen[node] = true;
                    arrivals[fv.rt_parent_up[node]]++;
                }
                node = fv.rt_parent_up[node];
            }
        }
        for (const auto& a : arrivals) REQUIRE(a.first >= 0 && fv.rt_kids[a.first] == a.second, a.first, a.second);
    }
    for (int i = 0; i < root.n_in; ++i) {
        touch(upv.pool + (size_t)root.in_buf[i] * upv.stride, sizeof(float) * (size_t)upv.frames);
        touch(upv.flags + root.in_buf[i], 1);
    }
    for (int i = 0; i < fv.n_cmds; ++i)
        if (fv.cmds[i].block == cmd_block0) g_cmds_applied++;
    return 0;
}
// the resident realtime kernel, as the harness can model it: it renders the block its doorbell already names, and its watchdog
// fires at once (alive = 0) — the next callback finds it gone and launches another (the relaunch path of fwgpu_run.cpp)
int launch_rt_persist(hipStream_t s, const FusedView& fv, const DevView& upv, const RootArgs& root, float* d_out, uint32_t cmd_block0,
                      unsigned* d_sync, unsigned long long* d_done_flag, RtMailbox* d_mb, unsigned long long* d_go, unsigned long long first_seq,
                      unsigned long long idle_ticks) {
    REQUIRE(d_mb && d_go && idle_ticks > 0 && d_mb->doorbell == first_seq && d_mb->alive == 1, first_seq, idle_ticks);
    touch(d_go, sizeof(unsigned long long));
    const int rc = launch_rt_block(s, fv, upv, root, d_out, cmd_block0, d_sync, d_done_flag, first_seq);
    d_mb->alive = 0;
    return rc;
}
int launch_sp_hist_copy(hipStream_t, const FusedView& fv) {
    REQUIRE(fv.has_sp && fv.sp_hist_in_render && fv.hist != nullptr, fv.has_sp);
    for (int v = 0; v < fv.n_voices; ++v)
        if (fv.voices[v].sp_ext_off >= 0) {
            touch(fv.ext + fv.voices[v].sp_ext_off, sizeof(float) * SP_HIST);
            touch(fv.hist + (size_t)v * SP_HIST, sizeof(float) * SP_HIST);
        }
    return 0;
}
int launch_leaf_sum(hipStream_t, const FusedView& fv, int K) {
    g_launches[2]++;
    check_fused_common(fv, K);
    REQUIRE(!fv.fx_plan);
    if (fv.lazy_rs) {  // a resampler plan's lazy call: k_leaf_rs takes records and templates from the LazyRecs (the book-keeping of the other lazy launches)
        g_lazy_launches++;
        REQUIRE(fv.has_rs && !fv.has_sp && fv.lazy != nullptr && fv.n_cmds == 0 && fv.lazy_tmpl != nullptr && fv.rs_tmpl == fv.lazy_tmpl, fv.has_rs, fv.n_cmds);
        touch(fv.lazy, sizeof(LazyRec) * (size_t)fv.n_voices);
        touch(fv.lazy_tmpl, sizeof(VoiceBlk) * (size_t)fv.n_voices);
        LazyBook& b = lazy_book(fv.lazy);
        REQUIRE(b.have_ctl && fv.lazy_blk0 == b.since_ctl, (long long)fv.lazy_blk0, (long long)b.since_ctl);
        b.since_ctl += (unsigned long long)K;
        b.unflushed += (unsigned long long)K;
        b.chain = false;
    } else if (fv.has_rs && fv.rs_tmpl) {
        touch(fv.rs_tmpl, sizeof(VoiceBlk) * (size_t)fv.n_voices);
    }
    return 0;
}
int launch_chain(hipStream_t, const FusedView& fv, int K, uint32_t, int nq) {
    g_launches[3]++;
    check_fused_common(fv, K);
    bool any_bq2 = false;
    for (int i = 0; i < fv.n_voices; ++i) any_bq2 = any_bq2 || (fv.voices[i].sampler_state >= 0 && fv.voices[i].bq2_state >= 0);
    REQUIRE(((nq & 4) != 0) == any_bq2, nq);  // bit 2: the instantiation with the second recurrence stage, exactly when some voice needs it
    bool any_sites = false;
    for (int i = 0; i < fv.n_voices; ++i) {
        const VoiceDesc& vd = fv.voices[i];
        if (vd.sampler_state < 0) continue;
        any_sites = any_sites || vd.n_mid != 0;
        for (int j = 0; j < vd.n_stages; ++j) any_sites = any_sites || vd.stage_kind[j] == K_HARD_CLIP;
    }
    REQUIRE(((nq & 8) != 0) == any_sites, nq);  // bit 3: the five-site stage logic, exactly when some voice needs it
    nq &= 3;
    REQUIRE(fv.fx_plan == 1 && K <= CH_FAST_KMAX && (nq == 1 || nq == 2) && fv.frames % (64 * nq) == 0, K, nq);
    if (fv.lazy_chain) {  // no control launch in front of this one: the same book-keeping as the leaf kernel's lazy launch
        g_lazy_launches++;
        REQUIRE(fv.lazy != nullptr && fv.n_cmds == 0, fv.n_cmds);
        touch(fv.lazy, sizeof(LazyRec) * (size_t)fv.n_voices);
        LazyBook& b = lazy_book(fv.lazy);
        REQUIRE(b.have_ctl && fv.lazy_blk0 == b.since_ctl, (long long)fv.lazy_blk0, (long long)b.since_ctl);
        b.since_ctl += (unsigned long long)K;
        b.unflushed += (unsigned long long)K;
        b.chain = true;
    } else if (fv.lazy) {  // (behind a control launch: node state was current, launch_voice_control checked it)
        REQUIRE(lazy_book(fv.lazy).since_ctl == 0);
    }
    touch(fv.chain_start, sizeof(ChainStart) * (size_t)fv.n_voices);
    touch(fv.chain_dummy, 32 * 1024);
    touch(fv.chain_stats, 16);
    touch(fv.groups, sizeof(ChainGroup) * (size_t)fv.n_groups);
    int row = 0, leaf = 0;
    for (int g = 0; g < fv.n_groups; ++g) {
        const ChainGroup& cg = fv.groups[g];
        REQUIRE(cg.first_voice == row && cg.n_voices >= 1 && cg.n_voices <= 32 && cg.n_leaves >= 1 && cg.n_leaves <= CH_GROUP_LEAVES, g, cg.n_voices);
        int r = 0;
        uint32_t starts = 0, masked = 0;
        for (int l = 0; l < cg.n_leaves; ++l, ++leaf) {
            REQUIRE(leaf < fv.n_leaves && cg.row0[l] == r && cg.ports[l] == fv.leaves[leaf].ports && cg.out_buf[l] == fv.leaves[leaf].out_buf &&
                        fv.leaves[leaf].first_voice == row + r, g, l);
            starts |= 1u << r;
            const int p = cg.ports[l];
            // (a leaf that leads a WIDER SumNode — the hybrid plan's split mixers — takes that node's path: LeafDesc::pad; first met by the
            //  harness in round 5, when mono-adapter voices behind a filter made split banks of chain voices)
            const int path = fv.leaves[leaf].pad ? fv.leaves[leaf].pad : p;
            if (!(path == 2 || path == 3 || path == 4))
                for (int q = 0; q < p; ++q) masked |= 1u << (r + q);
            r += p;
        }
        REQUIRE(r == cg.n_voices && cg.start_mask == starts && cg.masked_rows == masked, g, r);
        if (cg.uniform_ports) {
            bool ok = cg.n_voices == 32 && (cg.uniform_ports == 32 || cg.uniform_ports == 16 || cg.uniform_ports == 8 || cg.uniform_ports == 4);
            for (int l = 0; l < cg.n_leaves; ++l) ok = ok && cg.ports[l] == cg.uniform_ports;
            REQUIRE(ok, g, cg.uniform_ports);
        }
        row += cg.n_voices;
    }
    REQUIRE(row == fv.n_voices && leaf == fv.n_leaves, row, leaf);
    for (int i = 0; i < fv.n_voices; ++i) {  // every delay line holds at least one tile
        const VoiceDesc& vd = fv.voices[i];
        if (vd.sampler_state >= 0 && vd.dl_state >= 0) REQUIRE(fv.states[vd.dl_state].loop_end >= (uint64_t)(64 * nq), i, nq);
    }
    return 0;
}
int launch_scatter_ext(hipStream_t, float* ext, const void* d_items, int n) {
    g_launches[7]++;
    touch(d_items, sizeof(ExtInitHost) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const ExtInitHost& it = ((const ExtInitHost*)d_items)[i];
        REQUIRE(it.n <= 6, i, it.n);
        touch(ext + it.off, sizeof(float) * it.n);
    }
    return 0;
}
}  // namespace fwgpu

// PortInts (fwgpu_graph.h): the planner's port lists, inline up to 4 ints.  Returns 0, or the number of the check that failed.
extern "C" int fwh_portints_selftest(void) {
    using fwgpu::PortInts;
    auto eq = [](const PortInts& a, std::initializer_list<int> want) {
        if (a.size() != want.size()) return false;
        size_t i = 0;
        for (int w : want)
            if (a[i++] != w) return false;
        return true;
    };
    PortInts a;
    if (!a.empty() || a.size() != 0 || a.begin() != a.end()) return 1;
    a.assign(3, -1);
    if (!eq(a, {-1, -1, -1})) return 2;
    a[1] = 7;
    a.resize(4);                                   // still inline, the new element 0
    if (!eq(a, {-1, 7, -1, 0})) return 3;
    a.resize(6);                                   // to the heap, contents kept
    a[5] = 9;
    if (!eq(a, {-1, 7, -1, 0, 0, 9})) return 4;
    PortInts b = a;                                // copy of a heap list
    b[0] = 1;
    if (!eq(a, {-1, 7, -1, 0, 0, 9}) || !eq(b, {1, 7, -1, 0, 0, 9})) return 5;
    PortInts c2 = std::move(b);                    // move: the source is left empty and usable
    if (!eq(c2, {1, 7, -1, 0, 0, 9}) || !b.empty()) return 6;
    b.assign(2, 5);
    if (!eq(b, {5, 5})) return 7;
    c2 = b;                                        // a short list over a heap one (capacity reused)
    if (!eq(c2, {5, 5})) return 8;
    c2.resize(1);
    if (!eq(c2, {5})) return 9;
    a = std::move(c2);                             // move-assign over a heap list
    if (!eq(a, {5}) || !c2.empty()) return 10;
    PortInts d;
    d.assign(64, 3);                               // a wide SumNode
    d[63] = 4;
    std::vector<int> v = d;
    if (v.size() != 64 || v[0] != 3 || v[63] != 4) return 11;
    std::vector<int> tab{1};
    tab.insert(tab.end(), d.begin() + 62, d.end());
    if (tab.size() != 3 || tab[1] != 3 || tab[2] != 4) return 12;
    a = a;                                         // self-assignment
    if (!eq(a, {5})) return 13;
    std::vector<fwgpu::PlanNode> nodes(3);         // inside a vector that reallocates (moves its elements)
    nodes[0].in_buf.assign(2, 8);
    nodes[1].in_buf.assign(70, 6);
    for (int i = 0; i < 100; ++i) nodes.emplace_back();
    if (!eq(nodes[0].in_buf, {8, 8}) || nodes[1].in_buf.size() != 70 || nodes[1].in_buf[69] != 6) return 14;
    return 0;
}
