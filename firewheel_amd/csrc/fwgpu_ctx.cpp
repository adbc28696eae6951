// fwgpu_ctx.cpp — device-resident executor behind the C ABI (include/fwgpu.h).
//
// A ctx is the device counterpart of FirewheelGraphCtx + FirewheelProcessor (graph/context.rs,
// graph/processor.rs): the host half keeps the editable graph, the device half keeps node state, the
// buffer pool and the launch plan in HBM.  No CPU compute path exists: every process call is kernels.
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>
#include <tuple>
#include <utility>
#include <string>
#include <vector>

#include "../../include/fwgpu.h"
#include "fwgpu_graph.h"
#include "fwgpu_launch.h"

using namespace fwgpu;

namespace {

thread_local std::string g_create_error;  // fwgpu_create_error(): of the calling thread's last failed fwgpu_ctx_create

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            cap = o.cap;
            o.p = nullptr;
            o.cap = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }  // whatever fwgpu_ctx_destroy's list misses still goes with the ctx
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        const bool regrow = p != nullptr;  // a buffer that grows once tends to grow again (graph edits add a few nodes
                                           // at a time; a 2 GB pool costs ~250 ms to free + allocate): leave headroom
        if (p) {
            hipError_t e = hipFree(p);
            if (e != hipSuccess) return e;
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes < 256 ? 256 : bytes;
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
static int host_kind_set(int kind) {
    if (kind == K_SAMPLER) return 2;
    return (kind == K_BEEP || kind == K_BIQUAD || kind == K_DELAY || kind == K_RESAMPLER || kind == K_SPATIAL) ? 1 : 0;
}

struct TimerCat {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    double acc_ms = 0.0;
    uint64_t launches = 0;
};

}  // namespace

struct fwgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t sample_rate = 48000;
    uint32_t mbf = 256;
    int stride = 256;
    uint32_t n_gin = 0, n_gout = 2;
    std::string last_error;

    HostGraph graph;
    Plan plan;
    bool have_plan = false;
    bool force_generic = false;
    uint32_t kmax = 64;      // blocks per launch the installed plan's buffers are sized for
    uint32_t kmax_req = 64;  // fwgpu_set_max_batch: takes effect when the next plan is installed

    // device state
    DevBuf d_states;
    size_t states_cap = 0;
    DevBuf d_ext;  // per-node extended state (floats): biquad coefficients + history, delay rings
    size_t ext_cap = 0, ext_used = 0;
    std::vector<SampleRec> samples;
    DevBuf d_samples;
    bool samples_dirty = true;

    // generic plan
    DevBuf d_nodes, d_in_buf, d_out_buf, d_level_nodes, d_pool, d_flags, d_gin_bufs, d_gout_bufs;
    std::vector<int> level_off, level_cnt;
    std::vector<int> level_kinds;  // bit s: the level holds node kinds of kernel set s (host_kind_set)
    int n_gout_bufs = 0, n_gin_bufs = 0;

    // fused plan
    bool fused = false;
    uint32_t generic_k = 1;  // blocks per generic-executor batch: kmax, capped by what the FIR history rings hold
    bool fused_fx = false;  // the fused plan's leaves run k_chain (biquad / delay in the voice chains)
    int chain_nq = 1;       // k_chain tile size / 64 frames
    int n_voices = 0, n_leaves = 0, n_bus = 1, ramp_slots = 0;
    int n_groups = 0;  // k_chain workgroups (groups of consecutive leaves)
    DevBuf d_groups;
    uint32_t epoch = 1;  // invalidates every VoiceCache when bumped (plan install, sample-table change)
    DevBuf d_voices, d_leaves, d_blks, d_refs, d_gsets, d_cache, d_ramps, d_bus, d_bus_flags, d_chain_start, d_chain_dummy, d_chain_stats;
    DevBuf d_up_nodes, d_up_in, d_up_out, d_up_level_nodes, d_root_bufs;
    std::vector<int> up_level_off, up_level_cnt;
    int n_tail = 0;  // master chain after the root SumNode (generic node kernel on the mix bus)
    std::vector<int> tail_kinds;  // kernel-set bit per master node
    DevBuf d_tail_nodes, d_tail_in, d_tail_out, d_tail_idx, d_tail_frozen;
    DevBuf d_frozen_ph;  // ... and its playhead snapshots
    DevBuf d_frozen;  // generic plan: k_frozen_scan's verdict per plan node, valid for the batch in flight
    int up_root_node = -1;  // index (in the upper-tree node table) of the root SumNode when it is alone on the last level
    RootArgs root_args;     // that node's port table, handed to k_root_out in its kernel arguments

    // FIR banks (generic executor): rows grouped by (level, impulse-response channel)
    struct FirGroup {  // one GEMM launch: every FIR row of a level with the same tap count
        int level, row_off, n_rows, tile_off;
        uint32_t T;
    };
    std::vector<FirGroup> fir_groups;
    std::map<std::pair<int, int>, uint32_t> ir_cache;  // (sample id, channel) -> ext offset of h[T] as f32
    DevBuf d_fir_rows, d_fir_tiles, d_fir_partials;

    // messages
    std::vector<Cmd> cmds;
    bool ring_msgs_pending = false;  // some node's pending_msgs is non-zero
    DevBuf d_cmds;
    int n_cmds_dev = 0;
    Cmd* h_cmds = nullptr;  // pinned staging for the async upload
    size_t h_cmds_cap = 0;
    hipEvent_t cmds_copied = nullptr;

    // staging + B1 scratch
    DevBuf d_in_stage, d_out_stage, d_scratch_pool, d_scratch_flags, d_scratch_tab, d_mask;
    DevBuf d_trace;  // FW_CHAIN_TRACE builds only

    // realtime edge (cpal/lib.rs:378-449 — one callback = a few hundred frames): pinned, device-mapped I/O blocks the
    // kernels read / write directly (no copy-engine hop), and the steady fused launch sequence kept as a hipGraph
    float *h_rt_in = nullptr, *h_rt_out = nullptr, *d_rt_in = nullptr, *d_rt_out = nullptr;
    struct RtGraph {
        hipGraphExec_t exec = nullptr;
        uint32_t epoch = 0, K = 0;
        int n_out_ch = 0;
        const float* d_out = nullptr;
    } rt_graph;
    bool rt_use_graph = false;  // FWGPU_RT_GRAPH=1: measured 5 us SLOWER per callback than 4 plain launches on ROCm 7.2
    DevBuf d_rs_table;  // SPEC resampler filter bank [RS_PHASES][RS_TAPS]

    // timing
    bool timing = false;
    TimerCat timers[5];  // 0 fused leaf kernel, 1 control kernel, 2 upper sums + out, 3 generic block, 4 k_fir_gemm alone

    fwgpu_ctx(uint32_t gin, uint32_t gout) : graph(gin, gout) {}
};

namespace {

int fail(fwgpu_ctx* c, int code, const std::string& msg) {
    c->last_error = msg;
    return code;
}
int hipfail(fwgpu_ctx* c, hipError_t e, const char* what) {
    c->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return FWGPU_ERR_DEVICE;
}
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

// ---- control-half scalar math, done on the host exactly as the reference's control thread does it
float percent_volume_to_raw_gain(float p) {  // core/param/range.rs:32-35
    float n = fmaxf(p, 0.0f) * (1.0f / 100.0f);
    return n * n;
}
float db_to_gain_clamped_neg_100_db(float db) {  // core/util.rs:7-9,21-27
    if (db <= -100.0f) return 0.0f;
    return powf(10.0f, 0.05f * db);
}
void pan_to_gains(float pan, float* gl, float* gr) {  // SPEC: DESIGN.md "spec nodes / pan"
    float p = fminf(fmaxf(pan, -1.0f), 1.0f);
    if (p <= -1.0f) {
        *gl = 1.0f;
        *gr = 0.0f;
        return;
    }
    if (p >= 1.0f) {
        *gl = 0.0f;
        *gr = 1.0f;
        return;
    }
    double theta = ((double)p + 1.0) * (3.14159265358979323846 / 4.0);
    *gl = (float)cos(theta);
    *gr = (float)sin(theta);
}
Smoother make_smoother(float val, uint32_t sample_rate) {  // core/param/smoother.rs:93-112, defaults :18-25
    Smoother s;
    const float smooth_secs = 10.0f / 1000.0f;
    s.b = expf(-1.0f / (smooth_secs * (float)sample_rate));
    s.a = 1.0f - s.b;
    s.status = SM_INACTIVE;
    s.input = val;
    s.last = val;
    s.eps = 0.00001f;
    return s;
}

// SPEC biquad: RBJ cookbook, computed in f64 on the control side, normalised by a0, rounded to f32.
void biquad_coefs(int type, float cutoff_hz, float q, uint32_t sample_rate, float co[5]) {
    double fs = (double)sample_rate;
    double f0 = fmin(fmax((double)cutoff_hz, 1.0), 0.49 * fs);
    double Q = fmax((double)q, 1e-3);
    double w0 = 2.0 * 3.14159265358979323846 * f0 / fs;
    double cw = cos(w0), alpha = sin(w0) / (2.0 * Q);
    double b0, b1, b2, a0 = 1.0 + alpha, a1 = -2.0 * cw, a2 = 1.0 - alpha;
    if (type == 1) {  // high-pass
        b0 = (1.0 + cw) * 0.5;
        b1 = -(1.0 + cw);
        b2 = (1.0 + cw) * 0.5;
    } else if (type == 2) {  // band-pass, constant 0 dB peak gain
        b0 = alpha;
        b1 = 0.0;
        b2 = -alpha;
    } else {  // low-pass
        b0 = (1.0 - cw) * 0.5;
        b1 = 1.0 - cw;
        b2 = (1.0 - cw) * 0.5;
    }
    co[0] = (float)(b0 / a0);
    co[1] = (float)(b1 / a0);
    co[2] = (float)(b2 / a0);
    co[3] = (float)(a1 / a0);
    co[4] = (float)(a2 / a0);
}
// SPEC resampler: Kaiser-windowed sinc (beta 8, cutoff 0.9 x Nyquist), RS_PHASES x RS_TAPS, every
// phase normalised to unity DC gain in f64 and rounded to f32 — the control side builds the table once per ctx.
double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 64; ++k) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
    }
    return sum;
}
void resampler_table(float* h) {
    const double fc = 0.9, beta = 8.0, half = RS_TAPS / 2.0, pi = 3.14159265358979323846;
    const double i0b = bessel_i0(beta);
    for (int ph = 0; ph < RS_PHASES; ++ph) {
        double row[RS_TAPS], sum = 0.0;
        for (int k = 0; k < RS_TAPS; ++k) {
            double t = (double)(k - (RS_TAPS / 2 - 1)) - (double)ph / RS_PHASES;
            double x = pi * fc * t;
            double sinc = fabs(t) < 1e-12 ? 1.0 : sin(x) / x;
            double r = t / half;
            double w = fabs(r) >= 1.0 ? 0.0 : bessel_i0(beta * sqrt(1.0 - r * r)) / i0b;
            row[k] = fc * sinc * w;
            sum += row[k];
        }
        for (int k = 0; k < RS_TAPS; ++k) h[ph * RS_TAPS + k] = (float)(row[k] / sum);
    }
}
uint64_t resampler_step(float ratio) {  // source frames per output frame as 32.32 fixed point
    double r = (double)ratio;
    if (!(r >= 1.0 / 256.0)) r = 1.0 / 256.0;
    if (r > 256.0) r = 256.0;
    return (uint64_t)llround(r * 4294967296.0);
}
// SPEC spatialiser: listener at the origin (+x right, +y up, -z forward): inverse-distance gain (reference distance
// 1, rolloff 1), equal-power pan from the direction cosine to the right, per-ear delay up to 0.66 ms.
void spatial_params(float x, float y, float z, uint32_t sample_rate, float* gl, float* gr, int* dl, int* dr) {
    const double pi = 3.14159265358979323846;
    double d = sqrt((double)x * x + (double)y * y + (double)z * z);
    double att = 1.0 / fmax(d, 1.0);
    double s = d < 1e-9 ? 0.0 : (double)x / d;
    double theta = (s + 1.0) * (pi / 4.0);
    *gl = (float)(cos(theta) * att);
    *gr = (float)(sin(theta) * att);
    double itd_max = round(0.00066 * (double)sample_rate);
    if (itd_max > SP_HIST - 1) itd_max = SP_HIST - 1;
    *dl = (int)round(fmax(0.0, s) * itd_max);
    *dr = (int)round(fmax(0.0, -s) * itd_max);
}
uint32_t delay_frames(float secs, uint32_t sample_rate) {
    double d = round((double)secs * (double)sample_rate);
    if (!(d >= 1.0)) d = 1.0;
    if (d > 16777216.0) d = 16777216.0;
    return (uint32_t)d;
}

// AudioNode constructors + activate(): the initial audio-half state of each node kind.
NodeState make_state(int kind, const float* params, int n_params, uint32_t sample_rate) {
    auto p = [&](int i, float d) { return i < n_params ? params[i] : d; };
    NodeState s;
    memset(&s, 0, sizeof(s));
    s.sample = -1;
    s.sample_rate = sample_rate;
    s.s0 = make_smoother(0.f, sample_rate);
    s.s1 = make_smoother(0.f, sample_rate);
    switch (kind) {
        case K_VOLUME:   // volume.rs:15-22, :67-75
        case K_SAMPLER:  // sampler.rs:55-64, :302-319
            s.p0 = percent_volume_to_raw_gain(fmaxf(p(0, 100.0f), 0.0f));
            s.s0 = make_smoother(s.p0, sample_rate);
            break;
        case K_BEEP: {  // beep_test.rs:15-24, :55-60
            float f = p(0, 440.0f);
            if (f < 20.0f) f = 20.0f;
            if (f > 20000.0f) f = 20000.0f;
            float g = db_to_gain_clamped_neg_100_db(p(1, -12.0f));
            if (g < 0.0f) g = 0.0f;
            if (g > 1.0f) g = 1.0f;
            s.gain = g;
            s.enabled = p(2, 1.0f) != 0.0f ? 1 : 0;
            s.phasor = 0.0f;
            s.phasor_inc = f / (float)sample_rate;
            break;
        }
        case K_HARD_CLIP:  // hard_clip.rs:8-12
            s.p0 = db_to_gain_clamped_neg_100_db(p(0, 0.0f));
            break;
        case K_PAN:
            pan_to_gains(p(0, 0.0f), &s.p0, &s.p1);
            s.s0 = make_smoother(s.p0, sample_rate);
            s.s1 = make_smoother(s.p1, sample_rate);
            break;
        case K_WIDTH:
            s.p0 = fmaxf(p(0, 1.0f), 0.0f);
            s.s0 = make_smoother(s.p0, sample_rate);
            break;
        case K_BIQUAD:  // coefficients go to the ext pool at activation; keep the ctor args for that
            s.p0 = p(1, 1000.0f);  // cutoff
            s.p1 = p(2, 0.70710678f);  // Q
            s.enabled = (int)p(0, 0.0f);  // type
            break;
        case K_FIR:
            s.sample = (int)p(0, -1.0f);  // impulse-response sample id; T and the ring are set at activation
            break;
        case K_RESAMPLER:  // params: sample id, ratio, loop, playing
            s.sample = (int)p(0, -1.0f);
            s.loop_start = resampler_step(p(1, 1.0f));
            s.has_loop = p(2, 0.0f) != 0.0f ? 1 : 0;
            s.playing = p(3, 1.0f) != 0.0f ? 1 : 0;
            s.playhead = 0;
            break;
        case K_SPATIAL: {  // params: x, y, z of the source; the ctor args stay in phasor / phasor_inc / gain
            s.phasor = p(0, 0.0f);
            s.phasor_inc = p(1, 0.0f);
            s.gain = p(2, -1.0f);
            int dl, dr;
            spatial_params(s.phasor, s.phasor_inc, s.gain, sample_rate, &s.p0, &s.p1, &dl, &dr);
            s.s0 = make_smoother(s.p0, sample_rate);
            s.s1 = make_smoother(s.p1, sample_rate);
            s.playing = dl;
            s.has_loop = dr;
            break;
        }
        case K_DELAY: {
            float mix = fminf(fmaxf(p(2, 0.5f), 0.0f), 1.0f);
            s.p0 = fminf(fmaxf(p(1, 0.0f), 0.0f), 0.999f);  // feedback
            s.p1 = mix;
            s.gain = 1.0f - mix;  // dry
            s.loop_end = delay_frames(p(0, 0.1f), sample_rate);
            s.playhead = 0;
            break;
        }
        default:
            break;
    }
    return s;
}

int upload(fwgpu_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    HIPC(c, b.ensure(bytes));
    if (bytes) HIPC(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

int upload_sample_table(fwgpu_ctx* c) {
    if (!c->samples_dirty) return 0;
    c->samples_dirty = false;
    c->epoch++;  // cached steady descriptors hold sample indices / sizes
    std::vector<SampleDesc> tab(std::max<size_t>(c->samples.size(), 1));
    for (size_t i = 0; i < c->samples.size(); ++i) tab[i] = c->samples[i].desc;
    HIPC(c, hipStreamSynchronize(c->stream));
    return upload(c, c->d_samples, tab.data(), tab.size() * sizeof(SampleDesc));
}

// ---------------------------------------------------------------- fused voice-bank plan detection
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
    bool has_fx = false;  // some chain holds a biquad / delay: the k_chain plan
    uint64_t min_delay = ~0ull;  // shortest delay line among the chains (frames)
};

// `graph`: for the delay lengths (k_chain needs D >= one tile); `mbf` must then be a multiple of the tile
bool detect_fused(const Plan& plan, const HostGraph& graph, uint32_t mbf, FusedBuild& fb) {
    const int N = (int)plan.nodes.size();
    if (N < 3) return false;
    const PlanNode& gout = plan.nodes.back();
    if (gout.is_graph_io != 2 || gout.n_in != 2) return false;
    // consumer counts per (node, port)
    std::vector<std::vector<int>> cons(N);
    for (int i = 0; i < N; ++i) cons[i].assign(plan.nodes[i].n_out, 0);
    for (const PlanNode& n : plan.nodes)
        for (int p = 0; p < n.n_in; ++p)
            if (n.in_src_node[p] >= 0) cons[n.in_src_node[p]][n.in_src_port[p]]++;
    auto stereo_src = [&](const PlanNode& n, int port0, int& src) -> bool {
        int a = n.in_src_node[port0], b = n.in_src_node[port0 + 1];
        if (a < 0 || a != b) return false;
        if (n.in_src_port[port0] != 0 || n.in_src_port[port0 + 1] != 1) return false;
        if (plan.nodes[a].n_out != 2 || cons[a][0] != 1 || cons[a][1] != 1) return false;
        src = a;
        return true;
    };
    int root;
    if (!stereo_src(gout, 0, root)) return false;
    std::vector<char> covered(N, 0);
    covered[N - 1] = 1;
    std::vector<int> tail;  // plan indices, graph_out side first
    while (plan.nodes[root].kind != K_SUM) {
        const PlanNode& n = plan.nodes[root];
        const bool master_kind = n.kind == K_VOLUME || n.kind == K_HARD_CLIP || n.kind == K_PAN || n.kind == K_WIDTH ||
                                 n.kind == K_BIQUAD || n.kind == K_DELAY;
        if (!master_kind || n.n_in != 2 || n.n_out != 2 || covered[root] || tail.size() >= 16) return false;
        covered[root] = 1;
        tail.push_back(root);
        int src;
        if (!stereo_src(n, 0, src)) return false;
        root = src;
    }
    for (int i = 0; i < N; ++i)
        if (plan.nodes[i].is_graph_io == 1) {
            covered[i] = 1;
            for (int cnt : cons[i])
                if (cnt) return false;  // graph inputs feed the graph: generic executor
        }
    // walk the sum tree breadth-first
    struct SumRec {
        int node;
        bool leaf;
        std::vector<int> kids;  // plan indices (sum nodes) or chain ends
        int out_buf;
    };
    std::vector<SumRec> sums;
    std::map<int, int> sum_index;
    std::vector<int> work{root};
    while (!work.empty()) {
        int si = work.back();
        work.pop_back();
        const PlanNode& s = plan.nodes[si];
        if (s.kind != K_SUM || s.n_out != 2 || s.n_in < 2 || s.n_in % 2) return false;
        if (covered[si]) return false;
        covered[si] = 1;
        SumRec r;
        r.node = si;
        r.out_buf = 0;
        int n_sum = 0, n_chain = 0;
        for (int p = 0; p < s.n_in / 2; ++p) {
            int src;
            if (s.in_src_node[2 * p] < 0 && s.in_src_node[2 * p + 1] < 0) {
                // an unconnected stereo port (a voice slot nothing is plugged into): the reference feeds it the cleared,
                // silent-flagged buffer (schedule.rs:310-313) — a null kid: a null voice under a leaf, bus 0 above
                r.kids.push_back(-1);
                continue;
            }
            if (!stereo_src(s, 2 * p, src)) return false;
            r.kids.push_back(src);
            if (plan.nodes[src].kind == K_SUM) n_sum++;
            else n_chain++;
        }
        if (n_sum && n_chain) return false;
        r.leaf = n_sum == 0;  // (a SumNode with nothing plugged in at all is a leaf of null voices)
        if (!r.leaf)
            for (int k : r.kids)
                if (k >= 0) work.push_back(k);
        sum_index[si] = (int)sums.size();
        sums.push_back(r);
    }
    // leaves in plan order (deterministic), chains in port order
    std::vector<int> leaf_order;
    for (int i = 0; i < (int)sums.size(); ++i)
        if (sums[i].leaf) leaf_order.push_back(i);
    std::sort(leaf_order.begin(), leaf_order.end(), [&](int a, int b) { return sums[a].node < sums[b].node; });
    int next_bus = 1;
    for (int li : leaf_order) {
        SumRec& r = sums[li];
        LeafDesc ld;
        ld.first_voice = (int)fb.voices.size();
        ld.ports = (int)r.kids.size();
        ld.out_buf = next_bus;
        ld.pad = 0;
        r.out_buf = next_bus;
        next_bus += 2;
        for (int end : r.kids) {
            if (end < 0) {  // null voice: k_voice_control emits a constant silent, cleared-source record for it
                VoiceDesc vd;
                memset(&vd, 0, sizeof(vd));
                vd.sampler_state = vd.bq_state = vd.dl_state = -1;
                fb.voices.push_back(vd);
                continue;
            }
            // walk upstream: end -> ... -> sampler
            // accepted shape: sampler -> [biquad] -> [delay] -> (volume|pan)*
            std::vector<int> chain;
            int cur = end;
            int bq = -1, dl = -1;
            for (;;) {
                const PlanNode& n = plan.nodes[cur];
                if (covered[cur]) return false;
                if (n.kind == K_SAMPLER) {
                    if (n.n_in != 0 || n.n_out != 2) return false;
                    covered[cur] = 1;
                    break;
                }
                if (n.n_in != 2 || n.n_out != 2) return false;
                if (n.kind == K_VOLUME || n.kind == K_PAN) {
                    if (bq >= 0 || dl >= 0) return false;  // gain stages before the filter: generic executor
                    chain.push_back(cur);
                } else if (n.kind == K_DELAY) {
                    if (bq >= 0 || dl >= 0) return false;
                    if (graph.nodes[n.slot].init.loop_end < 64) return false;  // shorter than one k_chain tile
                    fb.min_delay = std::min<uint64_t>(fb.min_delay, graph.nodes[n.slot].init.loop_end);
                    dl = cur;
                } else if (n.kind == K_BIQUAD) {
                    if (bq >= 0) return false;
                    bq = cur;
                } else {
                    return false;
                }
                covered[cur] = 1;
                int src;
                if (!stereo_src(n, 0, src)) return false;
                cur = src;
            }
            if ((int)chain.size() > FW_MAX_STAGES - 1) return false;
            VoiceDesc vd;
            memset(&vd, 0, sizeof(vd));
            vd.bq_state = bq >= 0 ? (int)plan.nodes[bq].slot : -1;
            vd.dl_state = dl >= 0 ? (int)plan.nodes[dl].slot : -1;
            if (bq >= 0 || dl >= 0) fb.has_fx = true;
            vd.sampler_state = (int)plan.nodes[cur].slot;
            vd.n_stages = (int)chain.size();
            for (int j = 0; j < vd.n_stages; ++j) {  // schedule order: nearest the sampler first
                const PlanNode& n = plan.nodes[chain[chain.size() - 1 - j]];
                vd.stage_kind[j] = n.kind;
                vd.stage_state[j] = (int)n.slot;
            }
            fb.max_stages = std::max(fb.max_stages, vd.n_stages);
            fb.voices.push_back(vd);
        }
        fb.leaves.push_back(ld);
    }
    for (int i = 0; i < N; ++i)
        if (!covered[i]) return false;  // anything else in the graph: generic executor
    // upper sums: heights above the leaves, children's buses resolved bottom-up
    std::vector<int> height(sums.size(), -1);
    std::function<int(int)> h = [&](int i) -> int {
        if (height[i] >= 0) return height[i];
        if (sums[i].leaf) return height[i] = 0;
        int m = 0;
        for (int k : sums[i].kids)
            if (k >= 0) m = std::max(m, h(sum_index[k]) + 1);
        return height[i] = m;
    };
    int maxh = 0;
    for (int i = 0; i < (int)sums.size(); ++i) maxh = std::max(maxh, h(i));
    fb.up_levels.assign(maxh, std::vector<int>());
    for (int lv = 1; lv <= maxh; ++lv) {
        std::vector<int> at;
        for (int i = 0; i < (int)sums.size(); ++i)
            if (height[i] == lv) at.push_back(i);
        std::sort(at.begin(), at.end(), [&](int a, int b) { return sums[a].node < sums[b].node; });
        for (int i : at) {
            SumRec& r = sums[i];
            r.out_buf = next_bus;
            next_bus += 2;
            NodeDesc nd;
            memset(&nd, 0, sizeof(nd));
            nd.kind = K_SUM;
            nd.n_in = (int)r.kids.size() * 2;
            nd.n_out = 2;
            nd.in_off = (int)fb.up_in.size();
            nd.out_off = (int)fb.up_out.size();
            nd.state = 0;
            nd.aux0 = (int)r.kids.size();
            for (int k : r.kids) {
                const int cb = k >= 0 ? sums[sum_index[k]].out_buf : 0;  // unconnected: bus 0, the cleared + silent-flagged buffer
                fb.up_in.push_back(cb);
                fb.up_in.push_back(k >= 0 ? cb + 1 : 0);
            }
            fb.up_out.push_back(r.out_buf);
            fb.up_out.push_back(r.out_buf + 1);
            fb.up_levels[lv - 1].push_back((int)fb.up_nodes.size());
            fb.up_nodes.push_back(nd);
        }
    }
    int rb = sums[sum_index[root]].out_buf;
    for (int j = (int)tail.size() - 1; j >= 0; --j) {  // root side first
        const PlanNode& n = plan.nodes[tail[j]];
        NodeDesc nd;
        memset(&nd, 0, sizeof(nd));
        nd.kind = n.kind;
        nd.n_in = nd.n_out = 2;
        nd.in_off = (int)fb.tail_in.size();
        nd.out_off = (int)fb.tail_out.size();
        nd.state = (int)n.slot;
        fb.tail_in.push_back(rb);
        fb.tail_in.push_back(rb + 1);
        rb = next_bus;
        next_bus += 2;
        fb.tail_out.push_back(rb);
        fb.tail_out.push_back(rb + 1);
        fb.tail_nodes.push_back(nd);
    }
    fb.root_buf[0] = rb;
    fb.root_buf[1] = rb + 1;
    fb.n_bus = next_bus;
    if (fb.has_fx) {  // k_chain: whole tiles, one workgroup per leaf of <= 32 voices
        if (mbf % 64 != 0) return false;
        for (const LeafDesc& l : fb.leaves)
            if (l.ports > 32) return false;
    }
    return !fb.voices.empty();
}

int install_plan(fwgpu_ctx* c, Plan& plan) {
    HIPC(c, hipStreamSynchronize(c->stream));
    c->kmax = c->kmax_req;
    // 1. node state capacity (persists across recompiles: processor.rs:19,195-197)
    size_t need = c->graph.nodes.size();
    if (need > c->states_cap) {
        size_t cap = std::max<size_t>(need * 2, 64);
        DevBuf nb;
        HIPC(c, nb.ensure(cap * sizeof(NodeState)));
        HIPC(c, hipMemset(nb.p, 0, cap * sizeof(NodeState)));
        if (c->d_states.p && c->states_cap)
            HIPC(c, hipMemcpy(nb.p, c->d_states.p, c->states_cap * sizeof(NodeState), hipMemcpyDeviceToDevice));
        c->d_states = std::move(nb);
        c->states_cap = cap;
    }
    // 2. activate new nodes (graph.rs:594-612): scatter their initial states, carve their ext-pool slices
    {
        std::vector<StateInitHost> inits;
        std::vector<std::pair<size_t, std::vector<float>>> ext_inits;  // (offset, initial floats)
        std::vector<std::pair<int, int>> ir_requests;                  // impulse responses to convert to f32
        size_t ext_need = c->ext_used;
        for (uint32_t slot : c->graph.nodes_to_activate) {
            HostNode& n = c->graph.nodes[slot];
            if (!n.alive || n.activated) continue;
            uint32_t nch = n.n_in < n.n_out ? n.n_in : n.n_out;
            size_t len = 0;
            std::vector<float> head;
            if (n.kind == K_BIQUAD) {
                len = 5 + 4 * (size_t)nch;
                head.resize(5);
                biquad_coefs(n.init.enabled, n.init.p0, n.init.p1, c->sample_rate, head.data());
            } else if (n.kind == K_DELAY) {
                len = (size_t)nch * (size_t)n.init.loop_end;
            } else if (n.kind == K_SPATIAL) {
                len = SP_HIST;
            } else if (n.kind == K_FIR) {
                int ir = n.init.sample;
                if (ir < 0 || ir >= (int)c->samples.size() || !c->samples[ir].alive)
                    return fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, "FIR node: impulse-response sample was destroyed");
                uint64_t T = c->samples[ir].desc.frames;
                if (T == 0 || T > (1u << 24)) return fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, "FIR node: 1 <= taps <= 2^24");
                uint64_t R = T - 1 + (uint64_t)c->kmax * c->mbf;  // every block of a K-batch finds its whole window in the ring
                n.init.loop_start = T;
                n.init.loop_end = R;
                n.init.playhead = 0;
                len = (size_t)nch * 2 * (size_t)R;
                for (uint32_t ch = 0; ch < nch; ++ch) {
                    auto key = std::make_pair(ir, (int)std::min<uint32_t>(ch, (uint32_t)c->samples[ir].desc.channels - 1));
                    if (!c->ir_cache.count(key)) ir_requests.push_back(key);
                    c->ir_cache.emplace(key, 0u);  // offset assigned below, once the pool layout is final
                }
            }
            if (len) {
                n.init.ext_off = (uint32_t)ext_need;
                n.init.ext_len = (uint32_t)len;
                if (!head.empty()) ext_inits.emplace_back(ext_need, head);
                ext_need += (len + 63) / 64 * 64;
                if (ext_need > 0xffffffffull) return fail(c, FWGPU_ERR_INVALID, "ext state pool exceeds 2^32 floats");
            }
            StateInitHost si;
            si.index = (int)slot;
            si.pad = 0;
            si.st = n.init;
            inits.push_back(si);
            n.activated = true;
        }
        c->graph.nodes_to_activate.clear();
        for (auto& key : ir_requests) {  // one f32 copy of each impulse-response channel
            uint64_t T = c->samples[key.first].desc.frames;
            c->ir_cache[key] = (uint32_t)ext_need;
            ext_need += (T + 63) / 64 * 64;
            if (ext_need > 0xffffffffull) return fail(c, FWGPU_ERR_INVALID, "ext state pool exceeds 2^32 floats");
        }
        if (ext_need > c->ext_cap) {
            size_t cap = std::max<size_t>(ext_need * 2, 4096);
            DevBuf nb;
            HIPC(c, nb.ensure((cap + 256) * sizeof(float)));  // slack: vector loads may overhang the last slice
            HIPC(c, hipMemset(nb.p, 0, (cap + 256) * sizeof(float)));
            if (c->d_ext.p && c->ext_used)
                HIPC(c, hipMemcpy(nb.p, c->d_ext.p, c->ext_used * sizeof(float), hipMemcpyDeviceToDevice));
            c->d_ext = std::move(nb);
            c->ext_cap = cap;
        }
        c->ext_used = ext_need;
        if (!ext_inits.empty()) {  // one upload + one scatter launch, however many nodes were activated
            std::vector<ExtInitHost> items(ext_inits.size());
            for (size_t i = 0; i < ext_inits.size(); ++i) {
                items[i].off = (uint32_t)ext_inits[i].first;
                items[i].n = (uint32_t)std::min<size_t>(ext_inits[i].second.size(), 6);
                for (uint32_t j = 0; j < 6; ++j) items[i].v[j] = j < items[i].n ? ext_inits[i].second[j] : 0.f;
            }
            DevBuf tmp;
            int rc2 = upload(c, tmp, items.data(), items.size() * sizeof(ExtInitHost));
            if (rc2) return rc2;
            LCHK(c, launch_scatter_ext(c->stream, c->d_ext.as<float>(), tmp.p, (int)items.size()));
            HIPC(c, hipStreamSynchronize(c->stream));
            tmp.release();
        }
        if (!ir_requests.empty()) {
            int rc = upload_sample_table(c);
            if (rc) return rc;
            for (auto& key : ir_requests)
                LCHK(c, launch_ir_convert(c->stream, c->d_samples.as<SampleDesc>(), key.first, key.second,
                                          c->d_ext.as<float>() + c->ir_cache[key],
                                          (uint32_t)c->samples[key.first].desc.frames));
            HIPC(c, hipStreamSynchronize(c->stream));
        }
        if (!inits.empty()) {
            DevBuf tmp;
            int rc = upload(c, tmp, inits.data(), inits.size() * sizeof(StateInitHost));
            if (rc) return rc;
            LCHK(c, launch_scatter_states(c->stream, c->d_states.as<NodeState>(), tmp.p, (int)inits.size()));
            HIPC(c, hipStreamSynchronize(c->stream));
            tmp.release();
        }
    }
    // 3. node tables
    const int N = (int)plan.nodes.size();
    std::vector<NodeDesc> nd(N);
    std::vector<int> in_tab, out_tab;
    std::vector<std::vector<int>> levels(plan.num_levels);
    std::vector<int> gin_bufs, gout_bufs;
    for (int i = 0; i < N; ++i) {
        const PlanNode& p = plan.nodes[i];
        NodeDesc& d = nd[i];
        memset(&d, 0, sizeof(d));
        d.kind = p.kind;
        d.n_in = p.n_in;
        d.n_out = p.n_out;
        d.in_off = (int)in_tab.size();
        d.out_off = (int)out_tab.size();
        d.state = (int)p.slot;
        d.aux0 = (p.kind == K_SUM && p.n_out > 0) ? p.n_in / p.n_out : 0;
        d.is_graph_io = p.is_graph_io;
        in_tab.insert(in_tab.end(), p.in_buf.begin(), p.in_buf.end());
        out_tab.insert(out_tab.end(), p.out_buf.begin(), p.out_buf.end());
        if (p.is_graph_io == 1) gin_bufs = p.out_buf;
        else if (p.is_graph_io == 2) gout_bufs = p.in_buf;
        else levels[p.level].push_back(i);
    }
    if (in_tab.empty()) in_tab.push_back(0);
    if (out_tab.empty()) out_tab.push_back(0);
    int rc;
    if ((rc = upload(c, c->d_nodes, nd.data(), nd.size() * sizeof(NodeDesc)))) return rc;
    if ((rc = upload(c, c->d_in_buf, in_tab.data(), in_tab.size() * sizeof(int)))) return rc;
    if ((rc = upload(c, c->d_out_buf, out_tab.data(), out_tab.size() * sizeof(int)))) return rc;
    std::vector<int> flat;
    c->level_off.clear();
    c->level_cnt.clear();
    c->level_kinds.clear();
    for (auto& l : levels) {
        c->level_off.push_back((int)flat.size());
        c->level_cnt.push_back((int)l.size());
        flat.insert(flat.end(), l.begin(), l.end());
        int kinds = 0;
        for (int i : l) kinds |= 1 << host_kind_set(nd[i].kind);
        c->level_kinds.push_back(kinds);
    }
    if (flat.empty()) flat.push_back(0);
    if ((rc = upload(c, c->d_level_nodes, flat.data(), flat.size() * sizeof(int)))) return rc;
    c->n_gin_bufs = (int)gin_bufs.size();
    c->n_gout_bufs = (int)gout_bufs.size();
    if (gin_bufs.empty()) gin_bufs.push_back(0);
    if (gout_bufs.empty()) gout_bufs.push_back(0);
    if ((rc = upload(c, c->d_gin_bufs, gin_bufs.data(), gin_bufs.size() * sizeof(int)))) return rc;
    if ((rc = upload(c, c->d_gout_bufs, gout_bufs.data(), gout_bufs.size() * sizeof(int)))) return rc;
    // 3b. FIR banks: one GEMM per (level, impulse-response channel)
    {
        std::map<std::tuple<int, uint32_t, uint32_t>, std::vector<FirRow>> groups;
        for (int i = 0; i < N; ++i) {
            const PlanNode& p = plan.nodes[i];
            if (p.kind != K_FIR) continue;
            const HostNode& hn = c->graph.nodes[p.slot];
            int ir = hn.init.sample;
            uint32_t T = (uint32_t)hn.init.loop_start;
            int nch = std::min(p.n_in, p.n_out);
            for (int ch = 0; ch < nch; ++ch) {
                auto key = std::make_pair(ir, std::min(ch, c->samples[ir].desc.channels - 1));
                FirRow r;
                r.state = (int)p.slot;
                r.ch = ch;
                r.in_buf = p.in_buf[ch];
                r.out_buf = p.out_buf[ch];
                groups[std::make_tuple(p.level, c->ir_cache[key], T)].push_back(r);
            }
        }
        // one launch per (level, T); inside it rows are sorted by impulse-response channel and padded so that
        // every 32-row tile convolves with a single h (tile_h_off)
        std::vector<FirRow> flat_rows;
        std::vector<uint32_t> flat_tiles;
        c->fir_groups.clear();
        size_t partial_need = 0;
        std::map<std::pair<int, uint32_t>, std::vector<std::pair<uint32_t, std::vector<FirRow>*>>> launches;
        for (auto& g : groups)
            launches[std::make_pair(std::get<0>(g.first), std::get<2>(g.first))].emplace_back(std::get<1>(g.first), &g.second);
        for (auto& l : launches) {
            fwgpu_ctx::FirGroup fg;
            fg.level = l.first.first;
            fg.T = l.first.second;
            fg.row_off = (int)flat_rows.size();
            fg.tile_off = (int)flat_tiles.size();
            for (auto& part : l.second) {
                for (const FirRow& r : *part.second) flat_rows.push_back(r);
                while ((flat_rows.size() - fg.row_off) % 32) {
                    FirRow pad;
                    pad.state = -1;
                    pad.ch = pad.in_buf = pad.out_buf = 0;
                    flat_rows.push_back(pad);
                }
                while (flat_tiles.size() - fg.tile_off < (flat_rows.size() - fg.row_off) / 32) flat_tiles.push_back(part.first);
            }
            fg.n_rows = (int)flat_rows.size() - fg.row_off;
            c->fir_groups.push_back(fg);
            size_t W = (size_t)fg.T - 1 + c->mbf;
            size_t segs = (W + FIR_SEG - 1) / FIR_SEG;
            partial_need = std::max(partial_need, segs * (size_t)fg.n_rows * (size_t)((c->mbf + 255) / 256 * 256) * c->kmax);
        }
        if (!flat_rows.empty()) {
            if ((rc = upload(c, c->d_fir_rows, flat_rows.data(), flat_rows.size() * sizeof(FirRow)))) return rc;
            if ((rc = upload(c, c->d_fir_tiles, flat_tiles.data(), flat_tiles.size() * sizeof(uint32_t)))) return rc;
            HIPC(c, c->d_fir_partials.ensure(partial_need * sizeof(float)));
        }
    }
    // 4. buffer pool: a new schedule starts from zeroed buffers (schedule.rs:202-203); one slice per block of a
    //    generic K-batch.  generic_k: the FIR history rings were sized for the batch size in force when their node was
    //    activated — a later, larger kmax must not outrun them.
    c->generic_k = c->kmax;
    for (int i = 0; i < N; ++i) {
        if (plan.nodes[i].kind != K_FIR) continue;
        const HostNode& hn = c->graph.nodes[plan.nodes[i].slot];
        uint64_t room = (hn.init.loop_end - (hn.init.loop_start - 1)) / c->mbf;  // (R - (T-1)) / block
        c->generic_k = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(c->generic_k, room));
    }
    {
        const size_t Kg = c->generic_k;
        size_t pool_bytes = Kg * (size_t)plan.num_buffers * c->stride * sizeof(float);
        HIPC(c, c->d_pool.ensure(pool_bytes));
        HIPC(c, hipMemset(c->d_pool.p, 0, pool_bytes));
        std::vector<uint8_t> fl(Kg * (size_t)plan.num_buffers, 0);
        for (size_t k = 0; k < Kg; ++k) fl[k * (size_t)plan.num_buffers] = 1;  // buffer 0: constant zero, always flagged silent
        if ((rc = upload(c, c->d_flags, fl.data(), fl.size()))) return rc;
    }

    // 5. fused voice-bank plan
    c->fused = false;
    FusedBuild fb;
    c->fused_fx = false;
    if (!c->force_generic && detect_fused(plan, c->graph, c->mbf, fb)) {
        c->fused_fx = fb.has_fx;
        // k_chain tile = 64*nq frames: the larger tile needs whole tiles per block and every delay >= one tile
        c->chain_nq = (c->mbf % 128 == 0 && fb.min_delay >= 128) ? 2 : 1;
        if (const char* e = getenv("FWGPU_CHAIN_NQ")) {  // experiments: force the smaller tile
            if (atoi(e) == 1) c->chain_nq = 1;
        }
        c->n_voices = (int)fb.voices.size();
        c->n_leaves = (int)fb.leaves.size();
        c->n_bus = fb.n_bus;
        c->ramp_slots = 2 * (1 + fb.max_stages);
        if ((rc = upload(c, c->d_voices, fb.voices.data(), fb.voices.size() * sizeof(VoiceDesc)))) return rc;
        if ((rc = upload(c, c->d_leaves, fb.leaves.data(), fb.leaves.size() * sizeof(LeafDesc)))) return rc;
        c->n_groups = 0;
        if (c->fused_fx) {
            // k_chain workgroups: consecutive leaves packed greedily into groups of <= 32 voices / <= 8 leaves (the
            // voices of consecutive leaves are consecutive), so that a tree of small leaves fills the 32 voice rows
            std::vector<ChainGroup> groups;
            for (size_t l = 0; l < fb.leaves.size(); ++l) {
                const LeafDesc& ld = fb.leaves[l];
                if (groups.empty() || groups.back().n_voices + ld.ports > 32 || groups.back().n_leaves >= CH_GROUP_LEAVES) {
                    ChainGroup g;
                    memset(&g, 0, sizeof(g));
                    g.first_voice = ld.first_voice;
                    groups.push_back(g);
                }
                ChainGroup& g = groups.back();
                const int li = g.n_leaves++;
                g.out_buf[li] = ld.out_buf;
                g.row0[li] = g.n_voices;
                g.ports[li] = ld.ports;
                g.start_mask |= 1u << g.n_voices;
                if (!(ld.ports == 2 || ld.ports == 3 || ld.ports == 4))  // sum.rs:67-133 (Q13): the n-port path skips silent ports
                    g.masked_rows |= (ld.ports >= 32 ? 0xffffffffu : ((1u << ld.ports) - 1u)) << g.n_voices;
                g.n_voices += ld.ports;
            }
            for (ChainGroup& g : groups) {
                const int P = g.ports[0];
                bool uni = g.n_voices == 32 && (P == 32 || P == 16 || P == 8 || P == 4);
                for (int i = 0; i < g.n_leaves && uni; ++i) uni = g.ports[i] == P;
                g.uniform_ports = uni ? P : 0;
            }
            c->n_groups = (int)groups.size();
            if ((rc = upload(c, c->d_groups, groups.data(), groups.size() * sizeof(ChainGroup)))) return rc;
        }
        const size_t K = c->kmax;
        HIPC(c, c->d_blks.ensure(K * c->n_voices * sizeof(VoiceBlk)));
        HIPC(c, c->d_refs.ensure(K * c->n_voices * sizeof(VoiceRef)));
        HIPC(c, c->d_gsets.ensure((size_t)c->n_voices * FW_GSETS * sizeof(GainSet)));
        HIPC(c, c->d_chain_start.ensure((size_t)c->n_voices * sizeof(ChainStart)));
        HIPC(c, c->d_chain_dummy.ensure(64 * 1024));
        HIPC(c, c->d_chain_stats.ensure(2 * sizeof(unsigned long long)));
        HIPC(c, hipMemset(c->d_chain_stats.p, 0, 2 * sizeof(unsigned long long)));
        HIPC(c, hipMemset(c->d_chain_start.p, 0, (size_t)c->n_voices * sizeof(ChainStart)));
        HIPC(c, c->d_cache.ensure((size_t)c->n_voices * sizeof(VoiceCache)));
        HIPC(c, hipMemset(c->d_cache.p, 0, (size_t)c->n_voices * sizeof(VoiceCache)));
        c->epoch++;
        HIPC(c, c->d_ramps.ensure(K * c->n_voices * (size_t)c->ramp_slots * c->stride * sizeof(float)));
        size_t bus_bytes = K * (size_t)c->n_bus * c->stride * sizeof(float);
        HIPC(c, c->d_bus.ensure(bus_bytes));
        HIPC(c, hipMemset(c->d_bus.p, 0, bus_bytes));
        std::vector<uint8_t> bf(K * c->n_bus, 0);
        for (size_t k = 0; k < K; ++k) bf[k * c->n_bus] = 1;
        if ((rc = upload(c, c->d_bus_flags, bf.data(), bf.size()))) return rc;
        if (fb.up_nodes.empty()) {
            NodeDesc z;
            memset(&z, 0, sizeof(z));
            fb.up_nodes.push_back(z);
        }
        if (fb.up_in.empty()) fb.up_in.push_back(0);
        if (fb.up_out.empty()) fb.up_out.push_back(0);
        if ((rc = upload(c, c->d_up_nodes, fb.up_nodes.data(), fb.up_nodes.size() * sizeof(NodeDesc)))) return rc;
        if ((rc = upload(c, c->d_up_in, fb.up_in.data(), fb.up_in.size() * sizeof(int)))) return rc;
        if ((rc = upload(c, c->d_up_out, fb.up_out.data(), fb.up_out.size() * sizeof(int)))) return rc;
        std::vector<int> uflat;
        c->up_level_off.clear();
        c->up_level_cnt.clear();
        for (auto& l : fb.up_levels) {
            c->up_level_off.push_back((int)uflat.size());
            c->up_level_cnt.push_back((int)l.size());
            uflat.insert(uflat.end(), l.begin(), l.end());
        }
        c->up_root_node = (!fb.up_levels.empty() && fb.up_levels.back().size() == 1) ? fb.up_levels.back()[0] : -1;
        c->n_tail = (int)fb.tail_nodes.size();
        c->tail_kinds.clear();
        for (const NodeDesc& t : fb.tail_nodes) c->tail_kinds.push_back(1 << host_kind_set(t.kind));
        if (c->n_tail) {
            c->up_root_node = -1;  // the root's planar result feeds the master chain: no fused root + interleave
            std::vector<int> idx(c->n_tail);
            for (int i = 0; i < c->n_tail; ++i) idx[i] = i;
            if ((rc = upload(c, c->d_tail_nodes, fb.tail_nodes.data(), fb.tail_nodes.size() * sizeof(NodeDesc)))) return rc;
            if ((rc = upload(c, c->d_tail_in, fb.tail_in.data(), fb.tail_in.size() * sizeof(int)))) return rc;
            if ((rc = upload(c, c->d_tail_out, fb.tail_out.data(), fb.tail_out.size() * sizeof(int)))) return rc;
            if ((rc = upload(c, c->d_tail_idx, idx.data(), idx.size() * sizeof(int)))) return rc;
            HIPC(c, c->d_tail_frozen.ensure((size_t)c->n_tail * 16));  // (also a dummy playhead-snapshot area)
        }
        if (c->up_root_node >= 0) {
            const NodeDesc& rn = fb.up_nodes[c->up_root_node];
            if (rn.n_out == 2 && rn.n_in >= 2 && rn.n_in <= 64 && rn.n_in % 2 == 0) {
                memset(&c->root_args, 0, sizeof(c->root_args));
                c->root_args.n_in = rn.n_in;
                c->root_args.ports = rn.n_in / 2;
                for (int i = 0; i < rn.n_in; ++i) c->root_args.in_buf[i] = fb.up_in[rn.in_off + i];
                c->root_args.in_tab = c->d_up_in.as<int>() + rn.in_off;
            } else {
                c->up_root_node = -1;
            }
        }
        if (uflat.empty()) uflat.push_back(0);
        if ((rc = upload(c, c->d_up_level_nodes, uflat.data(), uflat.size() * sizeof(int)))) return rc;
        if ((rc = upload(c, c->d_root_bufs, fb.root_buf, sizeof(fb.root_buf)))) return rc;
        c->fused = true;
    }
    // k_frozen_scan's verdict tables (generic executor, K > 1): sized here, on the control thread — a process call never
    // allocates
    HIPC(c, c->d_frozen.ensure(plan.nodes.size()));
    HIPC(c, c->d_frozen_ph.ensure(plan.nodes.size() * sizeof(unsigned long long)));
    c->plan = plan;
    c->have_plan = true;
    c->graph.needs_compile = false;
    return 0;
}

// ---------------------------------------------------------------- messages
int upload_cmds(fwgpu_ctx* c) {
    std::stable_sort(c->cmds.begin(), c->cmds.end(), [](const Cmd& a, const Cmd& b) {
        if (a.state != b.state) return a.state < b.state;
        return a.block < b.block;
    });
    c->n_cmds_dev = (int)c->cmds.size();
    if (c->n_cmds_dev == 0) return 0;
    size_t bytes = c->cmds.size() * sizeof(Cmd);
    if (bytes > c->d_cmds.cap) {
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, c->d_cmds.ensure(bytes * 2));
    }
    if (!c->cmds_copied) HIPC(c, hipEventCreateWithFlags(&c->cmds_copied, hipEventDisableTiming));
    else HIPC(c, hipEventSynchronize(c->cmds_copied));  // the previous upload has left the pinned buffer
    if (c->cmds.size() > c->h_cmds_cap) {
        if (c->h_cmds) HIPC(c, hipHostFree(c->h_cmds));
        c->h_cmds = nullptr;
        c->h_cmds_cap = c->cmds.size() * 2;
        HIPC(c, hipHostMalloc((void**)&c->h_cmds, c->h_cmds_cap * sizeof(Cmd), hipHostMallocDefault));
    }
    memcpy(c->h_cmds, c->cmds.data(), bytes);
    HIPC(c, hipMemcpyAsync(c->d_cmds.p, c->h_cmds, bytes, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipEventRecord(c->cmds_copied, c->stream));
    return 0;
}
void retire_cmds(fwgpu_ctx* c, uint32_t nblocks) {
    size_t w = 0;  // in place: nothing is allocated on the process path
    for (const Cmd& m : c->cmds)
        if (m.block >= nblocks) {
            Cmd k = m;
            k.block -= nblocks;
            c->cmds[w++] = k;
        }
    c->cmds.resize(w);
    if (c->ring_msgs_pending) {  // (a steady realtime callback has none: do not walk thousands of nodes per block)
        for (HostNode& n : c->graph.nodes) n.pending_msgs = 0;
        c->ring_msgs_pending = false;
    }
}

// ---------------------------------------------------------------- timing helpers
void timer_begin(fwgpu_ctx* c, int cat, hipEvent_t* e0, hipEvent_t* e1) {
    *e0 = *e1 = nullptr;
    if (!c->timing) return;
    TimerCat& t = c->timers[cat];
    if (t.used == t.ev.size()) {
        if (t.ev.size() >= 8192) return;  // drained by timing_read
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        t.ev.emplace_back(a, b);
    }
    *e0 = t.ev[t.used].first;
    *e1 = t.ev[t.used].second;
    t.used++;
    t.launches++;
    (void)hipEventRecord(*e0, c->stream);
}
void timer_end(fwgpu_ctx* c, hipEvent_t e1) {
    if (e1) (void)hipEventRecord(e1, c->stream);
}
void timer_drain(fwgpu_ctx* c) {
    (void)hipStreamSynchronize(c->stream);
    for (TimerCat& t : c->timers) {
        for (size_t i = 0; i < t.used; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, t.ev[i].first, t.ev[i].second) == hipSuccess) t.acc_ms += ms;
        }
        t.used = 0;
    }
}

// ---------------------------------------------------------------- executors
DevView generic_view(fwgpu_ctx* c, int frames) {
    DevView v;
    v.nodes = c->d_nodes.as<NodeDesc>();
    v.in_buf = c->d_in_buf.as<int>();
    v.out_buf = c->d_out_buf.as<int>();
    v.states = c->d_states.as<NodeState>();
    v.samples = c->d_samples.as<SampleDesc>();
    v.ext = c->d_ext.as<float>();
    v.rs_table = c->d_rs_table.as<float>();
    v.pool = c->d_pool.as<float>();
    v.flags = c->d_flags.as<uint8_t>();
    v.pool_blk_stride = (size_t)c->plan.num_buffers * c->stride;  // one pool slice per block of a K-batch
    v.flags_blk_stride = (size_t)c->plan.num_buffers;
    v.stride = c->stride;
    v.frames = frames;
    v.cmds = c->d_cmds.as<Cmd>();
    v.n_cmds = c->n_cmds_dev;
    v.frozen = nullptr;
    v.frozen_playhead = nullptr;
    return v;
}

// K blocks of `frames` frames through the level-batched executor (schedule.rs:289-344 as one launch per level for
// all K blocks: each block has its own pool slice, a stateful node walks its K blocks in order inside one wave)
int run_generic_batch(fwgpu_ctx* c, int K, int frames, uint32_t cmd_block, const float* d_in, int n_in_ch, float* d_out,
                      int n_out_ch) {
    DevView v = generic_view(c, frames);
    // which gain-like stateful nodes cannot change during this batch (their blocks then run in parallel): decided once,
    // before the first level
    if (K > 1 && c->d_frozen.ensure((size_t)c->plan.nodes.size()) == hipSuccess &&
        c->d_frozen_ph.ensure((size_t)c->plan.nodes.size() * sizeof(unsigned long long)) == hipSuccess) {
        LCHK(c, launch_frozen_scan(c->stream, v, (int)c->plan.nodes.size(), cmd_block, K, c->d_frozen.as<uint8_t>(),
                                   c->d_frozen_ph.as<unsigned long long>()));
        v.frozen = c->d_frozen.as<uint8_t>();
        v.frozen_playhead = c->d_frozen_ph.as<unsigned long long>();
    }
    if (c->n_gin_bufs > 0)
        LCHK(c, launch_graph_in(c->stream, v.pool, v.flags, c->stride, v.pool_blk_stride, v.flags_blk_stride,
                                c->d_gin_bufs.as<int>(), c->n_gin_bufs, d_in, d_in ? n_in_ch : 0, frames, K));
    c->epoch++;  // node state moves outside the fused control kernel: cached steady descriptors are stale
    hipEvent_t e0, e1;
    timer_begin(c, 3, &e0, &e1);
    for (size_t l = 0; l < c->level_cnt.size(); ++l) {
        LCHK(c, launch_level(c->stream, v, c->d_level_nodes.as<int>() + c->level_off[l], c->level_cnt[l], K, cmd_block,
                             c->level_kinds[l]));
        for (const fwgpu_ctx::FirGroup& g : c->fir_groups)
            if (g.level == (int)l) {
                hipEvent_t g0 = nullptr, g1 = nullptr;
                if (c->timing) {  // the GEMM alone, on its own event pair (no record of its own: launch_fir does it)
                    TimerCat& t = c->timers[4];
                    if (t.used == t.ev.size() && t.ev.size() < 8192) {
                        hipEvent_t a, b;
                        if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) t.ev.emplace_back(a, b);
                    }
                    if (t.used < t.ev.size()) {
                        g0 = t.ev[t.used].first;
                        g1 = t.ev[t.used].second;
                        t.used++;
                        t.launches++;
                    }
                }
                LCHK(c, launch_fir(c->stream, v, c->d_fir_rows.as<FirRow>() + g.row_off, g.n_rows,
                                   c->d_fir_tiles.as<uint32_t>() + g.tile_off, g.T, c->d_fir_partials.as<float>(),
                                   c->d_fir_partials.cap / sizeof(float), K, g0, g1));
            }
    }
    timer_end(c, e1);
    LCHK(c, launch_graph_out(c->stream, v.pool, v.flags, c->stride, v.pool_blk_stride, v.flags_blk_stride,
                             c->d_gout_bufs.as<int>(), c->n_gout_bufs, d_out, n_out_ch, frames, K));
    return 0;
}

// K full blocks through the fused voice-bank plan
int run_fused_batch(fwgpu_ctx* c, int K, uint32_t cmd_block0, float* d_out, int n_out_ch) {
    FusedView fv;
    fv.voices = c->d_voices.as<VoiceDesc>();
    fv.leaves = c->d_leaves.as<LeafDesc>();
    fv.states = c->d_states.as<NodeState>();
    fv.samples = c->d_samples.as<SampleDesc>();
    fv.blks = c->d_blks.as<VoiceBlk>();
    fv.refs = c->d_refs.as<VoiceRef>();
    fv.refs_stride = (int)c->kmax;
    fv.gsets = c->d_gsets.as<GainSet>();
    fv.cache = c->d_cache.as<VoiceCache>();
    fv.epoch = c->epoch;
    fv.n_gain_stages = c->ramp_slots / 2;
    fv.ramps = c->d_ramps.as<float>();
    fv.ramp_slots = c->ramp_slots;
    fv.bus = c->d_bus.as<float>();
    fv.bus_flags = c->d_bus_flags.as<uint8_t>();
    fv.bus_blk_stride = (size_t)c->n_bus * c->stride;
    fv.bus_flags_blk_stride = (size_t)c->n_bus;
    fv.cmds = c->d_cmds.as<Cmd>();
    fv.n_cmds = c->n_cmds_dev;
    fv.n_voices = c->n_voices;
    fv.n_leaves = c->n_leaves;
    fv.stride = c->stride;
    fv.frames = (int)c->mbf;
    fv.fx_plan = c->fused_fx ? 1 : 0;
    fv.groups = c->d_groups.as<ChainGroup>();
    fv.n_groups = c->n_groups;
    fv.ext = c->d_ext.as<float>();
    fv.chain_start = c->d_chain_start.as<ChainStart>();
    fv.chain_dummy = c->d_chain_dummy.as<float>();
    fv.chain_stats = c->d_chain_stats.as<unsigned long long>();
    fv.trace = nullptr;
#ifdef FW_CHAIN_TRACE
    if (c->d_trace.ensure(64 * 16 * 8 * sizeof(unsigned long long)) == hipSuccess) fv.trace = c->d_trace.as<unsigned long long>();
#endif
    {
        static const int dbg = getenv("FWGPU_CHAIN_SKIP") ? atoi(getenv("FWGPU_CHAIN_SKIP")) : 0;
        fv.dbg = dbg;
    }
    hipEvent_t e0, e1;
    timer_begin(c, 1, &e0, &e1);
    LCHK(c, launch_voice_control(c->stream, fv, K, cmd_block0));
    timer_end(c, e1);
    timer_begin(c, 0, &e0, &e1);
    if (c->fused_fx) LCHK(c, launch_chain(c->stream, fv, K, cmd_block0, c->chain_nq));
    else LCHK(c, launch_leaf_sum(c->stream, fv, K));
    timer_end(c, e1);
    timer_begin(c, 2, &e0, &e1);
    if (!c->up_level_cnt.empty()) {
        DevView v;
        v.nodes = c->d_up_nodes.as<NodeDesc>();
        v.in_buf = c->d_up_in.as<int>();
        v.out_buf = c->d_up_out.as<int>();
        v.states = c->d_states.as<NodeState>();
        v.samples = c->d_samples.as<SampleDesc>();
        v.ext = c->d_ext.as<float>();
        v.rs_table = c->d_rs_table.as<float>();
        v.pool = fv.bus;
        v.flags = fv.bus_flags;
        v.pool_blk_stride = fv.bus_blk_stride;
        v.flags_blk_stride = fv.bus_flags_blk_stride;
        v.stride = c->stride;
        v.frames = (int)c->mbf;
        v.cmds = nullptr;
        v.n_cmds = 0;
        v.frozen = nullptr;
        v.frozen_playhead = nullptr;
        // the root SumNode is fused with read_graph_outputs + interleave_stereo when the stream is stereo
        const bool fuse_root = c->up_root_node >= 0 && n_out_ch == 2;
        const size_t n_levels = c->up_level_cnt.size() - (fuse_root ? 1 : 0);
        for (size_t l = 0; l < n_levels; ++l)
            LCHK(c, launch_bus_sum(c->stream, v, c->d_up_level_nodes.as<int>() + c->up_level_off[l], c->up_level_cnt[l], K, 2));
        if (fuse_root) {
            LCHK(c, launch_root_out(c->stream, v, c->root_args, d_out, K));
            timer_end(c, e1);
            return 0;
        }
    }
    if (c->n_tail) {  // master chain on the mix bus: the generic node kernel, K-batched, one launch per node
        DevView v;
        v.nodes = c->d_tail_nodes.as<NodeDesc>();
        v.in_buf = c->d_tail_in.as<int>();
        v.out_buf = c->d_tail_out.as<int>();
        v.states = c->d_states.as<NodeState>();
        v.samples = c->d_samples.as<SampleDesc>();
        v.ext = c->d_ext.as<float>();
        v.rs_table = c->d_rs_table.as<float>();
        v.pool = fv.bus;
        v.flags = fv.bus_flags;
        v.pool_blk_stride = fv.bus_blk_stride;
        v.flags_blk_stride = fv.bus_flags_blk_stride;
        v.stride = c->stride;
        v.frames = (int)c->mbf;
        v.cmds = fv.cmds;
        v.n_cmds = fv.n_cmds;
        v.frozen = nullptr;
        v.frozen_playhead = nullptr;  // (no sampler can sit in a master chain)
        if (K > 1) {
            LCHK(c, launch_frozen_scan(c->stream, v, c->n_tail, cmd_block0, K, c->d_tail_frozen.as<uint8_t>(),
                                       c->d_tail_frozen.as<unsigned long long>()));
            v.frozen = c->d_tail_frozen.as<uint8_t>();
        }
        for (int j = 0; j < c->n_tail; ++j)
            LCHK(c, launch_level(c->stream, v, c->d_tail_idx.as<int>() + j, 1, K, cmd_block0, c->tail_kinds[j]));
    }
    LCHK(c, launch_graph_out(c->stream, fv.bus, fv.bus_flags, c->stride, fv.bus_blk_stride, fv.bus_flags_blk_stride,
                             c->d_root_bufs.as<int>(), 2, d_out, n_out_ch, (int)c->mbf, K));
    timer_end(c, e1);
    return 0;
}

// all blocks of one call; d_in may be null.  frames may end in a partial block.
int run_blocks(fwgpu_ctx* c, uint64_t frames, const float* d_in, int n_in_ch, float* d_out, int n_out_ch,
               bool stable_out = false) {
    const uint32_t mbf = c->mbf;
    const uint32_t nblocks = (uint32_t)((frames + mbf - 1) / mbf);
    int rc = upload_sample_table(c);
    if (rc) return rc;
    rc = upload_cmds(c);
    if (rc) return rc;
    uint64_t done = 0;
    uint32_t blk = 0;
    const bool can_fuse = c->fused && !c->force_generic;
    // steady realtime call: no message on the device, one fused batch, the same output block as last time — every
    // kernel argument repeats (block counters and playheads live in device state), so the launch sequence is replayed
    // from a hipGraph instead of being re-issued kernel by kernel
    if (stable_out && c->rt_use_graph && can_fuse && !c->timing && c->n_cmds_dev == 0 && frames % mbf == 0 &&
        frames / mbf <= (c->fused_fx ? std::min<uint32_t>(c->kmax, CH_FAST_KMAX) : c->kmax)) {
        const uint32_t K = (uint32_t)(frames / mbf);
        fwgpu_ctx::RtGraph& g = c->rt_graph;
        if (!g.exec || g.epoch != c->epoch || g.K != K || g.d_out != d_out || g.n_out_ch != n_out_ch) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            g.exec = nullptr;
            HIPC(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            rc = run_fused_batch(c, (int)K, 0, d_out, n_out_ch);
            hipGraph_t graph = nullptr;
            hipError_t ce = hipStreamEndCapture(c->stream, &graph);
            if (rc || ce != hipSuccess) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc ? rc : hipfail(c, ce, "hipStreamEndCapture");
            }
            ce = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ce != hipSuccess) {
                g.exec = nullptr;
                return hipfail(c, ce, "hipGraphInstantiate");
            }
            g.epoch = c->epoch;
            g.K = K;
            g.d_out = d_out;
            g.n_out_ch = n_out_ch;
        }
        HIPC(c, hipGraphLaunch(g.exec, c->stream));
        retire_cmds(c, nblocks);
        return 0;
    }
    while (done < frames) {
        uint64_t left = frames - done;
        if (can_fuse && left >= mbf) {
            const uint32_t kcap = c->fused_fx ? std::min<uint32_t>(c->kmax, CH_FAST_KMAX) : c->kmax;
            uint32_t K = (uint32_t)std::min<uint64_t>(left / mbf, kcap);
            rc = run_fused_batch(c, (int)K, blk, d_out + done * n_out_ch, n_out_ch);
            if (rc) return rc;
            done += (uint64_t)K * mbf;
            blk += K;
            continue;
        }
        // generic executor: whole blocks in batches of generic_k, a trailing partial block on its own
        int bf = (int)std::min<uint64_t>(left, mbf);
        int K = bf == (int)mbf ? (int)std::min<uint64_t>(left / mbf, c->generic_k) : 1;
        rc = run_generic_batch(c, K, bf, blk, d_in ? d_in + done * n_in_ch : nullptr, n_in_ch, d_out + done * n_out_ch, n_out_ch);
        if (rc) return rc;
        done += (uint64_t)K * bf;
        blk += K;
    }
    retire_cmds(c, nblocks);
    return 0;
}

int push_cmd(fwgpu_ctx* c, int64_t node, int want_kind, Cmd m, bool counts_as_msg) {
    HostNode* n = c->graph.get(node);
    if (!n) return fail(c, FWGPU_ERR_INVALID, "unknown node id");
    if (want_kind >= 0 && n->kind != want_kind) return fail(c, FWGPU_ERR_INVALID, "node kind does not accept this message");
    if (counts_as_msg) {
        if (n->pending_msgs >= 128) return fail(c, FWGPU_ERR_QUEUE_FULL, "sampler message ring full");  // sampler.rs:14
        n->pending_msgs++;
        c->ring_msgs_pending = true;
    }
    m.state = (int)(node & 0xffffffff);
    c->cmds.push_back(m);
    return 0;
}

}  // namespace

// ================================================================= C ABI
extern "C" {

// every entry point that takes a context: a null handle is an error return, never a crash
#define NEED_CTX(c, ret) \
    do {                 \
        if (!(c)) return (ret); \
    } while (0)

const char* fwgpu_create_error(void) { return g_create_error.c_str(); }

fwgpu_ctx* fwgpu_ctx_create(int device, uint32_t sample_rate, uint32_t max_block_frames, uint32_t num_graph_inputs,
                            uint32_t num_graph_outputs, void* hip_stream) {
    g_create_error.clear();
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e) +
                         " (libfwgpu has no CPU fallback)";
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "device index out of range";
        return nullptr;
    }
    if (max_block_frames == 0 || num_graph_inputs > 64 || num_graph_outputs > 64) {
        g_create_error = "invalid arguments (max_block_frames > 0, <= 64 graph channels)";
        return nullptr;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_create_error = std::string("device is ") + prop.gcnArchName + "; libfwgpu ships gfx950 (MI355X) code only";
        return nullptr;
    }
    fwgpu_ctx* c = new fwgpu_ctx(num_graph_inputs, num_graph_outputs);
    c->device = device;
    c->sample_rate = sample_rate;
    c->mbf = max_block_frames;
    c->stride = (int)((max_block_frames + 63) / 64 * 64);
    c->n_gin = num_graph_inputs;
    c->n_gout = num_graph_outputs;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
            g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
            delete c;
            return nullptr;
        }
        c->own_stream = true;
    }
    {
        std::vector<float> tab(RS_PHASES * RS_TAPS);
        resampler_table(tab.data());
        if (upload(c, c->d_rs_table, tab.data(), tab.size() * sizeof(float)) != 0) {
            g_create_error = c->last_error;
            delete c;
            return nullptr;
        }
    }
    if (upload_sample_table(c) != 0 || c->d_mask.ensure(64) != hipSuccess) {
        g_create_error = c->last_error;
        delete c;
        return nullptr;
    }
    // realtime I/O blocks: allocated here, on the control thread (the process calls never allocate).  A failure is not
    // fatal: process_interleaved then always takes the staged-copy path.
    if (hipHostMalloc((void**)&c->h_rt_in, RT_IO_BYTES, hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc((void**)&c->h_rt_out, RT_IO_BYTES, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_rt_in, c->h_rt_in, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_rt_out, c->h_rt_out, 0) != hipSuccess) {
        if (c->h_rt_in) (void)hipHostFree(c->h_rt_in);
        if (c->h_rt_out) (void)hipHostFree(c->h_rt_out);
        c->h_rt_in = c->h_rt_out = nullptr;
        (void)hipGetLastError();
    }
    if (const char* e = getenv("FWGPU_RT_GRAPH")) c->rt_use_graph = atoi(e) != 0;
    return c;
}

void fwgpu_ctx_destroy(fwgpu_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (SampleRec& s : c->samples)
        if (s.alive && s.owned && s.d_data) (void)hipFree(s.d_data);
    DevBuf* bufs[] = {&c->d_states, &c->d_ext, &c->d_samples, &c->d_nodes, &c->d_in_buf, &c->d_out_buf, &c->d_level_nodes, &c->d_pool,
                      &c->d_flags, &c->d_gin_bufs, &c->d_gout_bufs, &c->d_voices, &c->d_leaves, &c->d_blks, &c->d_refs, &c->d_gsets, &c->d_cache, &c->d_ramps,
                      &c->d_bus, &c->d_bus_flags, &c->d_chain_start, &c->d_chain_dummy, &c->d_chain_stats, &c->d_groups, &c->d_up_nodes, &c->d_up_in, &c->d_up_out, &c->d_up_level_nodes,
                      &c->d_root_bufs, &c->d_tail_nodes, &c->d_tail_in, &c->d_tail_out, &c->d_tail_idx, &c->d_tail_frozen, &c->d_frozen, &c->d_frozen_ph, &c->d_fir_rows, &c->d_fir_tiles, &c->d_fir_partials, &c->d_cmds, &c->d_in_stage, &c->d_out_stage, &c->d_scratch_pool,
                      &c->d_scratch_flags, &c->d_scratch_tab, &c->d_mask, &c->d_trace, &c->d_rs_table};
    for (DevBuf* b : bufs) b->release();
    for (TimerCat& t : c->timers)
        for (auto& p : t.ev) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    if (c->rt_graph.exec) (void)hipGraphExecDestroy(c->rt_graph.exec);
    if (c->h_rt_in) (void)hipHostFree(c->h_rt_in);
    if (c->h_rt_out) (void)hipHostFree(c->h_rt_out);
    if (c->h_cmds) (void)hipHostFree(c->h_cmds);
    if (c->cmds_copied) (void)hipEventDestroy(c->cmds_copied);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* fwgpu_last_error(fwgpu_ctx* c) { return c ? c->last_error.c_str() : "null ctx"; }

int64_t fwgpu_graph_in_node(fwgpu_ctx* c) { return c ? c->graph.id_of(c->graph.graph_in_slot) : FWGPU_ERR_INVALID; }
int64_t fwgpu_graph_out_node(fwgpu_ctx* c) { return c ? c->graph.id_of(c->graph.graph_out_slot) : FWGPU_ERR_INVALID; }

int64_t fwgpu_add_node(fwgpu_ctx* c, int kind, uint32_t n_in, uint32_t n_out, const float* params, int n_params) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (kind < 0 || kind > K_SPATIAL) return fail(c, FWGPU_ERR_INVALID, "unsupported node kind");
    if (n_params < 0 || (n_params > 0 && !params)) return fail(c, FWGPU_ERR_INVALID, "params is null but n_params > 0");
    if (kind == K_FIR || kind == K_RESAMPLER) {
        int ir = n_params > 0 ? (int)params[0] : -1;
        if (ir < 0 || ir >= (int)c->samples.size() || !c->samples[ir].alive || c->samples[ir].desc.frames == 0)
            return fail(c, FWGPU_ERR_INVALID, kind == K_FIR ? "FIR node: params[0] must be the id of a non-empty impulse-response sample"
                                                            : "Resampler node: params[0] must be the id of a non-empty source sample");
        if (kind == K_RESAMPLER && c->samples[ir].desc.frames >= (1ull << 31))
            return fail(c, FWGPU_ERR_INVALID, "Resampler node: source longer than 2^31 frames");
    }
    if (n_in > 64 || n_out > 64) return fail(c, FWGPU_ERR_INVALID, "a node has at most 64 ports per side (core/node.rs:62,69)");
    NodeState st = make_state(kind, params, n_params, c->sample_rate);
    return c->graph.add_node(kind, n_in, n_out, st);
}
int fwgpu_remove_node(fwgpu_ctx* c, int64_t node) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    int rc = c->graph.remove_node(node);
    if (rc) return fail(c, rc, "remove_node: unknown node or graph in/out node");
    // messages still queued for a later block go with the node (its slot — the message key — may be reused by the next
    // fwgpu_add_node)
    const int slot = (int)(node & 0xffffffff);
    c->cmds.erase(std::remove_if(c->cmds.begin(), c->cmds.end(), [slot](const Cmd& m) { return m.state == slot; }), c->cmds.end());
    return 0;
}
int64_t fwgpu_connect(fwgpu_ctx* c, int64_t src, uint32_t sp, int64_t dst, uint32_t dp, int check) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return c->graph.connect(src, sp, dst, dp, check != 0);
}
int fwgpu_disconnect(fwgpu_ctx* c, int64_t src, uint32_t sp, int64_t dst, uint32_t dp) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return c->graph.disconnect(src, sp, dst, dp);
}
int fwgpu_disconnect_edge(fwgpu_ctx* c, int64_t e) { return c ? c->graph.disconnect_edge(e) : FWGPU_ERR_INVALID; }
int fwgpu_cycle_detected(fwgpu_ctx* c) { return c ? (c->graph.cycle_detected() ? 1 : 0) : FWGPU_ERR_INVALID; }

int fwgpu_update(fwgpu_ctx* c) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    (void)hipSetDevice(c->device);
    if (!c->graph.needs_compile && c->have_plan) return 0;
    Plan plan;
    std::string err;
    int rc = c->graph.build_plan(plan, err);
    if (rc) return fail(c, rc, err);
    return install_plan(c, plan);
}

int fwgpu_schedule_upload(fwgpu_ctx* c, const fwgpu_sched_node* sn, uint32_t n_nodes, uint32_t num_buffers) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    (void)hipSetDevice(c->device);
    if (n_nodes < 2 || !sn) return fail(c, FWGPU_ERR_INVALID, "a schedule holds at least graph_in and graph_out");
    for (uint32_t i = 0; i < n_nodes; ++i)
        if ((sn[i].num_inputs && (!sn[i].in_buffer_index || !sn[i].in_should_clear)) || (sn[i].num_outputs && !sn[i].out_buffer_index))
            return fail(c, FWGPU_ERR_INVALID, "schedule node with ports but null buffer tables");
    Plan plan;
    std::vector<std::pair<int, int>> last_writer(num_buffers, std::make_pair(-1, 0));
    for (uint32_t i = 0; i < n_nodes; ++i) {
        HostNode* hn = c->graph.get(sn[i].node);
        if (!hn) return fail(c, FWGPU_ERR_INVALID, "schedule names an unknown node");
        if (hn->n_in != sn[i].num_inputs || hn->n_out != sn[i].num_outputs)
            return fail(c, FWGPU_ERR_INVALID, "schedule port counts differ from add_node");
        std::string err;
        if (!check_activation(hn->kind, hn->n_in, hn->n_out, err)) return fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, err);
        PlanNode pn;
        pn.slot = (uint32_t)(sn[i].node & 0xffffffff);
        pn.kind = hn->kind;
        pn.n_in = (int)hn->n_in;
        pn.n_out = (int)hn->n_out;
        pn.level = 0;
        pn.is_graph_io = pn.slot == c->graph.graph_in_slot ? 1 : (pn.slot == c->graph.graph_out_slot ? 2 : 0);
        pn.in_src_node.assign(pn.n_in, -1);
        pn.in_src_port.assign(pn.n_in, 0);
        for (int p = 0; p < pn.n_in; ++p) {
            if (sn[i].in_should_clear[p]) continue;  // unconnected (InBufferAssignment.should_clear)
            uint32_t b = sn[i].in_buffer_index[p];
            if (b >= num_buffers || last_writer[b].first < 0)
                return fail(c, FWGPU_ERR_INVALID, "schedule input reads a buffer no earlier node wrote");
            pn.in_src_node[p] = last_writer[b].first;
            pn.in_src_port[p] = last_writer[b].second;
        }
        for (int p = 0; p < pn.n_out; ++p) {
            uint32_t b = sn[i].out_buffer_index[p];
            if (b >= num_buffers) return fail(c, FWGPU_ERR_INVALID, "schedule buffer index out of range");
            last_writer[b] = std::make_pair((int)i, p);
        }
        plan.nodes.push_back(pn);
    }
    if (plan.nodes.front().is_graph_io != 1 || plan.nodes.back().is_graph_io != 2)
        return fail(c, FWGPU_ERR_INVALID, "schedule must start with graph_in and end with graph_out");
    finalize_plan(plan);
    return install_plan(c, plan);
}

int fwgpu_plan_kind(fwgpu_ctx* c) { return c && c->have_plan ? (c->fused && !c->force_generic ? (c->fused_fx ? 2 : 1) : 0) : -1; }
int fwgpu_plan_num_levels(fwgpu_ctx* c) { return c && c->have_plan ? c->plan.num_levels : -1; }
int fwgpu_plan_node_level(fwgpu_ctx* c, int64_t node) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (!c->have_plan || !c->graph.get(node)) return -1;
    uint32_t slot = (uint32_t)(node & 0xffffffff);
    for (const PlanNode& p : c->plan.nodes)
        if (p.slot == slot) return p.level;
    return -1;
}
int fwgpu_plan_node_inputs_clear(fwgpu_ctx* c, int64_t node, int* should_clear, int cap) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (!c->have_plan || !c->graph.get(node)) return -1;
    uint32_t slot = (uint32_t)(node & 0xffffffff);
    for (const PlanNode& p : c->plan.nodes)
        if (p.slot == slot) {
            for (int i = 0; i < p.n_in && i < cap; ++i) should_clear[i] = p.in_buf[i] == 0 ? 1 : 0;
            return p.n_in;
        }
    return -1;
}
int fwgpu_set_force_generic(fwgpu_ctx* c, int on) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    c->force_generic = on != 0;
    return 0;
}
int fwgpu_plan_chain_stats(fwgpu_ctx* c, uint64_t* steady_workgroups, uint64_t* general_workgroups) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    (void)hipSetDevice(c->device);
    unsigned long long h[2] = {0, 0};
    if (c->d_chain_stats.p) {
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, hipMemcpy(h, c->d_chain_stats.p, sizeof(h), hipMemcpyDeviceToHost));
    }
    if (steady_workgroups) *steady_workgroups = h[0];
    if (general_workgroups) *general_workgroups = h[1];
    return 0;
}
int fwgpu_set_max_batch(fwgpu_ctx* c, uint32_t k) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (k == 0) return fail(c, FWGPU_ERR_INVALID, "max batch must be >= 1");
    c->kmax_req = k;
    c->graph.needs_compile = true;  // K-sized buffers are (re)allocated by the next fwgpu_update
    return 0;
}

static size_t fmt_elem_size(int fmt) { return (fmt == FMT_I_F32 || fmt == FMT_P_F32) ? 4 : 2; }

static int sample_add(fwgpu_ctx* c, int format, uint32_t channels, uint64_t frames, const void* data, bool on_device) {
    (void)hipSetDevice(c->device);
    if (format < 0 || format > FMT_P_F32 || channels == 0) return fail(c, FWGPU_ERR_INVALID, "bad sample format/channels");
    if (frames > (1ull << 40) / channels) return fail(c, FWGPU_ERR_INVALID, "sample too large (frames x channels > 2^40)");
    if (frames && !data) return fail(c, FWGPU_ERR_INVALID, "sample data is null");
    SampleRec r;
    r.alive = true;
    size_t bytes = (size_t)frames * channels * fmt_elem_size(format);
    if (on_device) {
        r.owned = false;
        r.d_data = (void*)data;
    } else {
        r.owned = true;
        HIPC(c, hipMalloc(&r.d_data, bytes + 256));  // slack: a wave's last dwordx4 may overhang the data
        hipError_t e = hipMemset((char*)r.d_data + bytes, 0, 256);
        if (e == hipSuccess && bytes) e = hipMemcpy(r.d_data, data, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(r.d_data);
            return hipfail(c, e, "sample upload");
        }
    }
    r.desc.data = r.d_data;
    r.desc.frames = frames;
    r.desc.channels = (int)channels;
    r.desc.format = format;
    c->samples.push_back(r);
    c->samples_dirty = true;  // table is re-uploaded lazily by the next process/update call
    return (int)c->samples.size() - 1;
}
int fwgpu_sample_create(fwgpu_ctx* c, int format, uint32_t channels, uint64_t frames, const void* data) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return sample_add(c, format, channels, frames, data, false);
}
int fwgpu_sample_create_device(fwgpu_ctx* c, int format, uint32_t channels, uint64_t frames, const void* device_data) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return sample_add(c, format, channels, frames, device_data, true);
}
int fwgpu_sample_destroy(fwgpu_ctx* c, int sample) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (sample < 0 || sample >= (int)c->samples.size() || !c->samples[sample].alive)
        return fail(c, FWGPU_ERR_INVALID, "unknown sample id");
    // FIR / resampler nodes name their sample at construction and keep it for life: refuse while one exists.  Samplers
    // pick samples by message, which the host does not track: see the contract in fwgpu.h.
    for (const HostNode& n : c->graph.nodes)
        if (n.alive && (n.kind == K_FIR || n.kind == K_RESAMPLER) && n.init.sample == sample)
            return fail(c, FWGPU_ERR_INVALID, "sample is in use by a FIR / resampler node");
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    SampleRec& r = c->samples[sample];
    if (r.owned && r.d_data) (void)hipFree(r.d_data);
    r.alive = false;
    r.d_data = nullptr;
    r.desc.data = nullptr;
    r.desc.frames = 0;
    c->samples_dirty = true;
    return 0;
}

int fwgpu_node_set_param(fwgpu_ctx* c, int64_t node, int param, float value, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    HostNode* n = c->graph.get(node);
    if (!n) return fail(c, FWGPU_ERR_INVALID, "unknown node id");
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    switch (n->kind) {
        case K_VOLUME:
        case K_SAMPLER:  // volume.rs:28-34, sampler.rs:171-177
            if (param != 0) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            m.type = CMD_SET_P0;
            m.f0 = percent_volume_to_raw_gain(value);
            return push_cmd(c, node, -1, m, false);
        case K_BEEP:  // beep_test.rs:30-32
            if (param != 0) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            m.type = CMD_SET_ENABLED;
            m.i0 = value != 0.0f;
            return push_cmd(c, node, -1, m, false);
        case K_PAN: {
            if (param != 0) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            float gl, gr;
            pan_to_gains(value, &gl, &gr);
            m.type = CMD_SET_P0;
            m.f0 = gl;
            int rc = push_cmd(c, node, -1, m, false);
            if (rc) return rc;
            m.type = CMD_SET_P1;
            m.f0 = gr;
            return push_cmd(c, node, -1, m, false);
        }
        case K_WIDTH:
            if (param != 0) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            m.type = CMD_SET_P0;
            m.f0 = fmaxf(value, 0.0f);
            return push_cmd(c, node, -1, m, false);
        case K_BIQUAD: {  // param 1 = cutoff_hz, 2 = Q: recompute the coefficients on the control side
            if (param != 1 && param != 2) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            if (param == 1) n->init.p0 = value;
            else n->init.p1 = value;
            float co[5];
            biquad_coefs(n->init.enabled, n->init.p0, n->init.p1, c->sample_rate, co);
            m.type = CMD_SET_COEFS;
            m.f0 = co[0];
            memcpy(&m.i0, &co[1], 4);
            memcpy(&m.i1, &co[2], 4);
            uint32_t lo, hi;
            memcpy(&lo, &co[3], 4);
            memcpy(&hi, &co[4], 4);
            uint64_t u = ((uint64_t)hi << 32) | lo;
            memcpy(&m.d0, &u, 8);
            return push_cmd(c, node, -1, m, false);
        }
        case K_DELAY: {  // param 1 = feedback, 2 = mix (the delay time is fixed at construction)
            if (param == 1) {
                m.type = CMD_SET_P0;
                m.f0 = fminf(fmaxf(value, 0.0f), 0.999f);
                return push_cmd(c, node, -1, m, false);
            }
            if (param != 2) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            float mix = fminf(fmaxf(value, 0.0f), 1.0f);
            m.type = CMD_SET_P1;
            m.f0 = mix;
            int rc = push_cmd(c, node, -1, m, false);
            if (rc) return rc;
            m.type = CMD_SET_GAIN;
            m.f0 = 1.0f - mix;
            return push_cmd(c, node, -1, m, false);
        }
        case K_RESAMPLER: {  // 1 = ratio (source frames per output frame), 3 = playing, 4 = seek to a source frame
            uint64_t u;
            if (param == 1) {
                m.type = CMD_RS_STEP;
                u = resampler_step(value);
                memcpy(&m.d0, &u, 8);
                return push_cmd(c, node, -1, m, false);
            }
            if (param == 3) {
                m.type = value != 0.0f ? CMD_SMP_PLAY : CMD_SMP_PAUSE;
                return push_cmd(c, node, -1, m, false);
            }
            if (param != 4) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            m.type = CMD_RS_SEEK;
            u = (uint64_t)fmaxf(value, 0.0f);
            memcpy(&m.d0, &u, 8);
            return push_cmd(c, node, -1, m, false);
        }
        case K_SPATIAL: {  // 0 / 1 / 2 = x / y / z of the source relative to the listener
            if (param < 0 || param > 2) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            if (param == 0) n->init.phasor = value;
            else if (param == 1) n->init.phasor_inc = value;
            else n->init.gain = value;
            float gl, gr;
            int dl, dr;
            spatial_params(n->init.phasor, n->init.phasor_inc, n->init.gain, c->sample_rate, &gl, &gr, &dl, &dr);
            m.type = CMD_SET_P0;
            m.f0 = gl;
            int rc = push_cmd(c, node, -1, m, false);
            if (rc) return rc;
            m.type = CMD_SET_P1;
            m.f0 = gr;
            if ((rc = push_cmd(c, node, -1, m, false))) return rc;
            m.type = CMD_SP_ITD;
            m.i0 = dl;
            m.i1 = dr;
            return push_cmd(c, node, -1, m, false);
        }
        default:
            return fail(c, FWGPU_ERR_INVALID, "node kind has no runtime params");
    }
}
int fwgpu_sampler_set_sample(fwgpu_ctx* c, int64_t node, int sample, int stop_playback, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (sample < 0 || sample >= (int)c->samples.size() || !c->samples[sample].alive)
        return fail(c, FWGPU_ERR_INVALID, "unknown sample id");
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    m.type = CMD_SMP_SET_SAMPLE;
    m.i0 = sample;
    m.i1 = stop_playback != 0;
    return push_cmd(c, node, K_SAMPLER, m, true);
}
static int simple_msg(fwgpu_ctx* c, int64_t node, int type, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    m.type = type;
    return push_cmd(c, node, K_SAMPLER, m, true);
}
int fwgpu_sampler_play(fwgpu_ctx* c, int64_t node, uint32_t b) { return simple_msg(c, node, CMD_SMP_PLAY, b); }
int fwgpu_sampler_pause(fwgpu_ctx* c, int64_t node, uint32_t b) { return simple_msg(c, node, CMD_SMP_PAUSE, b); }
int fwgpu_sampler_stop(fwgpu_ctx* c, int64_t node, uint32_t b) { return simple_msg(c, node, CMD_SMP_STOP, b); }
int fwgpu_sampler_set_playhead_secs(fwgpu_ctx* c, int64_t node, double secs, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    m.type = CMD_SMP_SET_PLAYHEAD;
    m.d0 = secs;
    return push_cmd(c, node, K_SAMPLER, m, true);
}
int fwgpu_sampler_set_loop_range(fwgpu_ctx* c, int64_t node, int mode, double start, double end, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (mode < 0 || mode > 2) return fail(c, FWGPU_ERR_INVALID, "loop mode must be 0, 1 or 2");
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    m.type = CMD_SMP_SET_LOOP;
    m.i0 = mode;
    m.d0 = start;
    m.d1 = end;
    return push_cmd(c, node, K_SAMPLER, m, true);
}

int fwgpu_process_interleaved(fwgpu_ctx* c, const float* input, float* output, uint32_t n_in_ch, uint32_t n_out_ch,
                              uint64_t frames, double, uint32_t) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    (void)hipSetDevice(c->device);
    if (n_in_ch > 64 || n_out_ch > 64) return fail(c, FWGPU_ERR_INVALID, "at most 64 stream channels per side (processor.rs:43-44)");
    if (frames > (1ull << 32)) return fail(c, FWGPU_ERR_INVALID, "more than 2^32 frames in one call");
    size_t out_bytes = (size_t)frames * n_out_ch * sizeof(float);
    if (out_bytes && !output) return fail(c, FWGPU_ERR_INVALID, "output is null");
    if (!c->have_plan || frames == 0) {  // processor.rs:86-89 (Q19)
        if (out_bytes) memset(output, 0, out_bytes);
        return 0;
    }
    const float* d_in = nullptr;
    const size_t in_bytes_rt = (n_in_ch > 0 && input) ? (size_t)frames * n_in_ch * sizeof(float) : 0;
    if (c->h_rt_out && out_bytes <= RT_IO_BYTES && in_bytes_rt <= RT_IO_BYTES) {
        // realtime-sized call: graph inputs are read from, and the interleaved output written to, pinned host blocks
        // mapped into the device — the only wait is the stream sync (SURVEY 8(b) "realtime rules")
        if (in_bytes_rt) {
            memcpy(c->h_rt_in, input, in_bytes_rt);
            d_in = c->d_rt_in;
        }
        int rc = run_blocks(c, frames, d_in, (int)n_in_ch, c->d_rt_out, (int)n_out_ch, true);
        if (rc) {
            if (out_bytes) memset(output, 0, out_bytes);
            return rc;
        }
        HIPC(c, hipStreamSynchronize(c->stream));
        if (out_bytes) memcpy(output, c->h_rt_out, out_bytes);
        return 0;
    }
    if (n_in_ch > 0 && input) {
        size_t in_bytes = (size_t)frames * n_in_ch * sizeof(float);
        if (in_bytes > c->d_in_stage.cap) {
            HIPC(c, hipStreamSynchronize(c->stream));
            HIPC(c, c->d_in_stage.ensure(in_bytes));
        }
        HIPC(c, hipMemcpyAsync(c->d_in_stage.p, input, in_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = c->d_in_stage.as<float>();
    }
    if (out_bytes > c->d_out_stage.cap) {
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, c->d_out_stage.ensure(out_bytes));
    }
    int rc = run_blocks(c, frames, d_in, (int)n_in_ch, c->d_out_stage.as<float>(), (int)n_out_ch);
    if (rc) {
        if (out_bytes) memset(output, 0, out_bytes);  // "all output buffers MUST be filled" (core/node.rs:41-42)
        return rc;
    }
    if (out_bytes) HIPC(c, hipMemcpyAsync(output, c->d_out_stage.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int fwgpu_process_blocks_device(fwgpu_ctx* c, uint32_t num_blocks, float* d_output, uint32_t n_out_ch) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    (void)hipSetDevice(c->device);
    if (!c->have_plan) return fail(c, FWGPU_ERR_INVALID, "no schedule: call fwgpu_update first");
    if (num_blocks == 0) return 0;
    if (n_out_ch > 64 || (n_out_ch && !d_output)) return fail(c, FWGPU_ERR_INVALID, "bad output (null, or more than 64 channels)");
    return run_blocks(c, (uint64_t)num_blocks * c->mbf, nullptr, 0, d_output, (int)n_out_ch);
}

int fwgpu_synchronize(fwgpu_ctx* c) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int fwgpu_node_process(fwgpu_ctx* c, int64_t node, uint64_t frames, const float* const* inputs, uint32_t n_in,
                       float* const* outputs, uint32_t n_out, uint64_t in_mask, uint64_t* out_mask, double, uint32_t) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    (void)hipSetDevice(c->device);
    HostNode* hn = c->graph.get(node);
    if (!hn || !hn->activated) return fail(c, FWGPU_ERR_INVALID, "node is not activated (call fwgpu_update)");
    if (hn->n_in != n_in || hn->n_out != n_out) return fail(c, FWGPU_ERR_INVALID, "port counts differ from add_node");
    if (frames > c->mbf) return fail(c, FWGPU_ERR_INVALID, "frames > max_block_frames");
    if (hn->kind == K_FIR) return fail(c, FWGPU_ERR_INVALID, "FIR banks run at graph level (fwgpu_process_interleaved), not per node");
    if (n_in + n_out == 0) return fail(c, FWGPU_ERR_INVALID, "node has no ports");
    if (frames == 0) {  // the reference never calls a node with an empty block (processor.rs:86-89 returns first): nothing to do
        if (out_mask) *out_mask = 0;
        return 0;
    }
    if ((n_in && !inputs) || (n_out && !outputs)) return fail(c, FWGPU_ERR_INVALID, "null channel table");
    for (uint32_t i = 0; frames && i < n_in; ++i)
        if (!inputs[i]) return fail(c, FWGPU_ERR_INVALID, "null input channel");
    for (uint32_t i = 0; frames && i < n_out; ++i)
        if (!outputs[i]) return fail(c, FWGPU_ERR_INVALID, "null output channel");
    const size_t stride = (size_t)c->stride;
    const int nb = 1 + (int)n_in + (int)n_out;
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, c->d_scratch_pool.ensure((size_t)nb * stride * sizeof(float)));
    HIPC(c, c->d_scratch_flags.ensure((size_t)nb));
    HIPC(c, hipMemsetAsync(c->d_scratch_pool.p, 0, stride * sizeof(float), c->stream));
    std::vector<uint8_t> fl(nb, 0);
    fl[0] = 1;
    for (uint32_t i = 0; i < n_in; ++i) {
        fl[1 + i] = (in_mask >> i) & 1ull;
        HIPC(c, hipMemcpyAsync(c->d_scratch_pool.as<float>() + (1 + i) * stride, inputs[i], frames * sizeof(float),
                               hipMemcpyHostToDevice, c->stream));
    }
    for (uint32_t i = 0; i < n_out; ++i)  // nodes that leave outputs untouched (dummy.rs, beep_test.rs:83-86)
        HIPC(c, hipMemcpyAsync(c->d_scratch_pool.as<float>() + (1 + n_in + i) * stride, outputs[i], frames * sizeof(float),
                               hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(c->d_scratch_flags.p, fl.data(), nb, hipMemcpyHostToDevice, c->stream));
    // temp tables: [NodeDesc][in ids][out ids]
    std::vector<int> tab(sizeof(NodeDesc) / sizeof(int) + n_in + n_out + 2, 0);
    NodeDesc nd;
    memset(&nd, 0, sizeof(nd));
    nd.kind = hn->kind;
    nd.n_in = (int)n_in;
    nd.n_out = (int)n_out;
    nd.in_off = 0;
    nd.out_off = 0;
    nd.state = (int)(node & 0xffffffff);
    nd.aux0 = (hn->kind == K_SUM && n_out) ? (int)(n_in / n_out) : 0;
    memcpy(tab.data(), &nd, sizeof(nd));
    int* ins = tab.data() + sizeof(NodeDesc) / sizeof(int);
    int* outs = ins + n_in + 1;
    for (uint32_t i = 0; i < n_in; ++i) ins[i] = 1 + (int)i;
    for (uint32_t i = 0; i < n_out; ++i) outs[i] = 1 + (int)n_in + (int)i;
    HIPC(c, c->d_scratch_tab.ensure(tab.size() * sizeof(int)));
    HIPC(c, hipMemcpyAsync(c->d_scratch_tab.p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    int rc = upload_sample_table(c);
    if (rc) return rc;
    rc = upload_cmds(c);
    if (rc) return rc;
    DevView v = generic_view(c, (int)frames);
    v.nodes = (const NodeDesc*)c->d_scratch_tab.p;
    v.in_buf = c->d_scratch_tab.as<int>() + sizeof(NodeDesc) / sizeof(int);
    v.out_buf = v.in_buf + n_in + 1;
    v.pool = c->d_scratch_pool.as<float>();
    v.flags = c->d_scratch_flags.as<uint8_t>();
    c->epoch++;
    LCHK(c, launch_single_node(c->stream, v, 0));
    for (uint32_t i = 0; i < n_out; ++i)
        HIPC(c, hipMemcpyAsync(outputs[i], c->d_scratch_pool.as<float>() + (1 + n_in + i) * stride, frames * sizeof(float),
                               hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(fl.data(), c->d_scratch_flags.p, nb, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    uint64_t om = 0;
    for (uint32_t i = 0; i < n_out; ++i)
        if (fl[1 + n_in + i]) om |= 1ull << i;
    if (out_mask) *out_mask = om;
    retire_cmds(c, 1);
    return 0;
}

int fwgpu_timing_enable(fwgpu_ctx* c, int on) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    c->timing = on != 0;
    return 0;
}
int fwgpu_timing_read(fwgpu_ctx* c, int which, double* total_ms, uint64_t* launches) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (which < 0 || which > 4) return fail(c, FWGPU_ERR_INVALID, "timer index");
    timer_drain(c);
    *total_ms = c->timers[which].acc_ms;
    *launches = c->timers[which].launches;
    return 0;
}
int fwgpu_timing_reset(fwgpu_ctx* c) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    timer_drain(c);
    for (TimerCat& t : c->timers) {
        t.acc_ms = 0.0;
        t.launches = 0;
    }
    return 0;
}
#ifdef FW_CHAIN_TRACE
// profiling builds only (scripts/chain_trace.py): timestamps [step 0..63][wave 0..15][slot 0..7] of workgroup 0
int fwgpu_debug_read_trace(fwgpu_ctx* c, unsigned long long* out) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    HIPC(c, hipStreamSynchronize(c->stream));
    if (!c->d_trace.p) return fail(c, FWGPU_ERR_INVALID, "no trace");
    HIPC(c, hipMemcpy(out, c->d_trace.p, 64 * 16 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif

int fwgpu_device_info(fwgpu_ctx* c, char* name, int name_cap, int* cus, uint64_t* hbm) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    hipDeviceProp_t prop;
    HIPC(c, hipGetDeviceProperties(&prop, c->device));
    if (name && name_cap > 0) {
        strncpy(name, prop.name[0] ? prop.name : prop.gcnArchName, (size_t)name_cap - 1);  // no marketing name: the ISA
        name[name_cap - 1] = 0;
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = (uint64_t)prop.totalGlobalMem;
    return 0;
}

}  // extern "C"
