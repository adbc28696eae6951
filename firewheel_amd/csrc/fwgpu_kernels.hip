// fwgpu_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the Firewheel per-block DSP executor.
//
// Two execution plans share the node state in HBM:
//   * generic level-batched executor: k_level — one 64-lane wave per scheduled node, one launch per
//     topological level, bit-exact restatement of every reference node (nodes/*.rs);
//   * fused voice-bank plan: k_voice_control (per-voice per-block state machines, K blocks per launch)
//     -> k_leaf_sum (HBM-streaming kernel: source fetch + gain stages + ordered radix-P sum in registers)
//     -> k_level over the upper sum tree (K-batched) -> k_graph_out.
// Compiled with -ffp-contract=off: the reference (Rust) never fuses mul+add, and parity is bit-exact.
//
// Layout: planar f32, one channel-block = `stride` floats (multiple of 64 => every buffer is 256-B aligned,
// a wave's float4 access covers 1 KiB contiguous).  Reference citations: core/ nodes/ graph/ as in fwgpu.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fwgpu_launch.h"

namespace fwgpu {

#define WAVE 64
#define WPB 4  // waves (nodes) per workgroup in k_level / k_leaf_sum

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned vector (unaligned playheads)

__device__ __forceinline__ v4f splat(float x) { return (v4f){x, x, x, x}; }

// ------------------------------------------------------------------ SilenceMask (core/silence_mask.rs:7-74)
__device__ __forceinline__ uint64_t mask_all_silent_bits(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
__device__ __forceinline__ bool mask_all(uint64_t m, int n) {
    uint64_t a = mask_all_silent_bits(n);
    return (m & a) == a;
}
__device__ __forceinline__ bool mask_any(uint64_t m, int n) { return (m & mask_all_silent_bits(n)) != 0; }
__device__ __forceinline__ bool mask_bit(uint64_t m, int i) { return (m >> i) & 1ull; }

// ------------------------------------------------------------------ ParamSmoother (core/param/smoother.rs)
struct GainRun {
    int ramp;    // 1: per-frame values follow out[i] = in_a + out[i-1]*b from prev
    float c;     // constant value when !ramp
    float in_a;  // input * a
    float b;
    float prev;  // running last_output
};

// set_and_process() up to the point where the per-frame ramp starts (smoother.rs:133-140,159-184).
// When status != Active the reference returns its (constant == input) buffer; when the first ramp sample is
// within settle_epsilon the reference discards the ramp, refills with `input` and goes Deactivating (Q1,Q2).
__device__ __forceinline__ GainRun smoother_begin(Smoother& s, float target, int frames) {
    if (!(s.input == target)) {  // set(): smoother.rs:134
        s.input = target;
        s.status = SM_ACTIVE;
    }
    GainRun r;
    r.ramp = 0;
    r.c = s.input;
    r.in_a = 0.f;
    r.b = s.b;
    r.prev = s.last;
    if (s.status != SM_ACTIVE || frames == 0) return r;  // :162-167
    float in_a = s.input * s.a;                          // :169
    float y0 = in_a + (s.last * s.b);                    // :171
    if (fabsf(s.input - y0) < s.eps) {                   // :181  (Q1: output[0])
        s.last = s.input;                                // reset(input) :116-122
        s.status = SM_DEACTIVATING;                      // :183
        return r;
    }
    r.ramp = 1;
    r.in_a = in_a;
    return r;
}
__device__ __forceinline__ void smoother_reset(Smoother& s, float val) {  // smoother.rs:115-129
    if (s.status != SM_INACTIVE) {
        s.status = SM_INACTIVE;
        s.input = val;
        s.last = val;
    } else if (!(s.input == val)) {
        s.input = val;
        s.last = val;
    }
}
__device__ __forceinline__ bool smoother_is_smoothing(const Smoother& s) { return s.status != SM_INACTIVE; }

// Advance the serial recurrence over `n` (<= 256) frames; lane L keeps frames 4L..4L+3 of the chunk.
// All 64 lanes run the same scalar chain (the recurrence is serial in time; smoother.rs:171-175).
__device__ __forceinline__ v4f ramp_chunk(GainRun& r, int n, int lane) {
    v4f g = splat(0.f);
    float prev = r.prev;
    const float in_a = r.in_a, b = r.b;
    int q = 0;
    for (; q * 4 + 4 <= n; ++q) {
        float v0 = in_a + (prev * b);
        float v1 = in_a + (v0 * b);
        float v2 = in_a + (v1 * b);
        float v3 = in_a + (v2 * b);
        if (q == lane) g = (v4f){v0, v1, v2, v3};
        prev = v3;
    }
    int rem = n - q * 4;
    if (rem > 0) {
        float v0 = in_a + (prev * b);
        float v1 = in_a + (v0 * b);
        float v2 = in_a + (v1 * b);
        if (q == lane) g = (v4f){v0, v1, v2, 0.f};
        prev = rem == 1 ? v0 : (rem == 2 ? v1 : v2);
    }
    r.prev = prev;
    return g;
}
__device__ __forceinline__ v4f gain_chunk(GainRun& r, int n, int lane) { return r.ramp ? ramp_chunk(r, n, lane) : splat(r.c); }

// ------------------------------------------------------------------ control -> audio messages
__device__ __forceinline__ uint64_t sat_round_u64(double x) {  // `(x).round() as u64` (saturating, NaN -> 0)
    double r = round(x);
    if (!(r == r)) return 0;
    if (r <= 0.0) return 0;
    if (r >= 18446744073709551615.0) return ~0ull;
    return (uint64_t)r;
}

// Apply every queued message for (state_idx, block) in order.  cmds are sorted by (state, block, seq).
// nodes/sampler.rs:331-414 (ring drained at the top of process()), volume.rs:92 (atomic load per block).
__device__ inline void apply_cmds(NodeState& s, int state_idx, uint32_t block, const Cmd* cmds, int n_cmds,
                                  const SampleDesc* samples, float* ext = nullptr, bool ext_write = false) {
    if (n_cmds == 0) return;
    int lo = 0, hi = n_cmds;  // lower bound of (state_idx, block)
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        const Cmd& c = cmds[mid];
        bool less = c.state < state_idx || (c.state == state_idx && c.block < block);
        if (less) lo = mid + 1;
        else hi = mid;
    }
    for (int i = lo; i < n_cmds; ++i) {
        Cmd c = cmds[i];
        if (c.state != state_idx || c.block != block) break;
        switch (c.type) {
            case CMD_SET_P0: s.p0 = c.f0; break;
            case CMD_SET_P1: s.p1 = c.f0; break;
            case CMD_SET_ENABLED: s.enabled = c.i0; break;
            case CMD_SET_GAIN: s.gain = c.f0; break;
            case CMD_SET_COEFS:  // biquad coefficients live at the head of the node's ext slice
                if (ext && ext_write) {
                    float* co = ext + s.ext_off;
                    co[0] = c.f0;
                    co[1] = __int_as_float(c.i0);
                    co[2] = __int_as_float(c.i1);
                    unsigned long long u = (unsigned long long)__double_as_longlong(c.d0);
                    co[3] = __int_as_float((int)(u & 0xffffffffull));
                    co[4] = __int_as_float((int)(u >> 32));
                }
                break;
            case CMD_SMP_SET_SAMPLE:  // sampler.rs:333-364
                s.sample = c.i0;
                if (s.has_loop && s.sample >= 0 && s.full_range) {  // update_sample :265-277
                    s.loop_start = 0;
                    s.loop_end = samples[s.sample].frames;
                }
                if (c.i1) {  // stop_playback
                    s.playhead = s.has_loop ? s.loop_start : 0;
                    s.playing = 0;
                }
                break;
            case CMD_SMP_PLAY: s.playing = 1; break;   // :365-371
            case CMD_SMP_PAUSE: s.playing = 0; break;  // :372-378
            case CMD_SMP_STOP:                         // :379-391
                s.playhead = s.has_loop ? s.loop_start : 0;
                s.playing = 0;
                break;
            case CMD_SMP_SET_PLAYHEAD:  // :392-399
                s.playhead = sat_round_u64(c.d0 * (double)s.sample_rate);
                break;
            case CMD_RS_STEP: s.loop_start = (uint64_t)__double_as_longlong(c.d0); break;
            case CMD_RS_SEEK: s.playhead = ((uint64_t)__double_as_longlong(c.d0)) << 32; break;
            case CMD_SP_ITD:
                s.playing = c.i0;
                s.has_loop = c.i1;
                break;
            case CMD_SMP_SET_LOOP:  // :400-412 + ProcLoopRange::new :241-263
                if (c.i0 == 0) {
                    s.has_loop = 0;
                } else {
                    s.has_loop = 1;
                    if (c.i0 == 1) {
                        s.loop_start = 0;
                        s.loop_end = s.sample >= 0 ? samples[s.sample].frames : 0;
                        s.full_range = 1;
                    } else {
                        s.loop_start = sat_round_u64(c.d0 * (double)s.sample_rate);
                        s.loop_end = sat_round_u64(c.d1 * (double)s.sample_rate);
                        s.full_range = 0;
                    }
                    if (s.playhead >= s.loop_start && s.playhead < s.loop_end) s.playhead = s.loop_start;  // Q7
                }
                break;
            default: break;
        }
    }
}

// ------------------------------------------------------------------ sampler playhead logic (shared by both plans)
struct Fetch {
    uint64_t off0, off1;
    uint32_t n1;
    int wrap, tail_zero;
};
// nodes/sampler.rs:445-517.  Returns false when the one-shot playhead is already past the end
// (":486-497": playing=false, clear).  Updates playhead/playing exactly as the reference does.
__device__ __forceinline__ bool sampler_advance(NodeState& s, uint64_t len, uint32_t frames, Fetch& f) {
    f.off0 = f.off1 = 0;
    f.n1 = frames;
    f.wrap = f.tail_zero = 0;
    if (s.has_loop) {
        if (s.playhead >= s.loop_end) s.playhead = s.loop_start;  // :446-453
        uint64_t left = s.loop_end - s.playhead;                  // :457-462
        uint32_t first = left < (uint64_t)frames ? (uint32_t)left : frames;
        f.off0 = s.playhead;
        f.n1 = first;
        if (first < frames) {  // :467-481 wraps once (Q8)
            s.playhead = s.loop_start;
            f.off1 = s.playhead;
            f.wrap = 1;
            s.playhead += (uint64_t)(frames - first);
        } else {
            s.playhead += (uint64_t)frames;
        }
        return true;
    }
    if (s.playhead >= len) {  // :486-497
        s.playing = 0;
        return false;
    }
    uint64_t left = len - s.playhead;
    uint32_t copy = left < (uint64_t)frames ? (uint32_t)left : frames;  // :499
    f.off0 = s.playhead;
    f.n1 = copy;
    if (copy < frames) {  // :503-513 (Q9)
        s.playing = 0;
        s.playhead = 0;
        f.tail_zero = 1;
    } else {
        s.playhead += (uint64_t)frames;
    }
    return true;
}

// core/sample_resource.rs:338-345 + fill_buffers_* :348-456 — one source element, converted.
__device__ __forceinline__ float sample_fetch(const SampleDesc& sd, int ch, uint64_t frame) {
    switch (sd.format) {
        case FMT_I_I16: return (float)((const int16_t*)sd.data)[frame * (uint64_t)sd.channels + ch] * (1.0f / 32767.0f);
        case FMT_I_U16:
            return ((float)((const uint16_t*)sd.data)[frame * (uint64_t)sd.channels + ch] * (2.0f / 65535.0f)) - 1.0f;
        case FMT_I_F32: return ((const float*)sd.data)[frame * (uint64_t)sd.channels + ch];
        case FMT_P_I16: return (float)((const int16_t*)sd.data)[(uint64_t)ch * sd.frames + frame] * (1.0f / 32767.0f);
        case FMT_P_U16:
            return ((float)((const uint16_t*)sd.data)[(uint64_t)ch * sd.frames + frame] * (2.0f / 65535.0f)) - 1.0f;
        default: return ((const float*)sd.data)[(uint64_t)ch * sd.frames + frame];
    }
}
// four consecutive output frames f..f+3 of channel ch under a Fetch (per-element path: any format, wrap, tail)
__device__ __forceinline__ v4f sample_fetch4(const SampleDesc& sd, int ch, const Fetch& f, uint32_t frame, uint32_t frames) {
    v4f x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t i = frame + j;
        float v = 0.f;
        if (i < frames) {
            if (i < f.n1) v = sample_fetch(sd, ch, f.off0 + i);
            else if (f.wrap) v = sample_fetch(sd, ch, f.off1 + (i - f.n1));
            else v = 0.f;  // tail_zero (sampler.rs:509-511)
        }
        x[j] = v;
    }
    return x;
}

// ------------------------------------------------------------------ generic executor: one wave per node
struct WaveIO {
    float* pool;
    uint8_t* flags;
    const int* in_buf;
    const int* out_buf;
    int stride;
    int lane;
    int frames;
    __device__ __forceinline__ const float* in(int i) const { return pool + (size_t)in_buf[i] * stride; }
    __device__ __forceinline__ float* out(int i) const { return pool + (size_t)out_buf[i] * stride; }
};

// core/util.rs:165-175
__device__ __forceinline__ uint64_t clear_all_outputs(const WaveIO& io, int first, int n_out) {
    for (int c = first; c < n_out; ++c) {
        float* o = io.out(c);
        for (int base = io.lane * 4; base < io.frames; base += 256) *(v4f*)(o + base) = splat(0.f);
    }
    return mask_all_silent_bits(n_out - first);
}

__device__ __forceinline__ float clipf(float x, float t) { return fmaxf(fminf(x, t), -t); }
__device__ __forceinline__ float beep_step(float ph, float inc) {  // beep_test.rs:90 (f32::fract)
    float t = ph + inc;
    return t - truncf(t);
}

// node kinds whose audio half carries state from block to block
__device__ __forceinline__ bool kind_is_stateful(int kind) {
    return kind == K_VOLUME || kind == K_SAMPLER || kind == K_BEEP || kind == K_PAN || kind == K_HARD_CLIP ||
           kind == K_WIDTH || kind == K_BIQUAD || kind == K_DELAY || kind == K_RESAMPLER || kind == K_SPATIAL;
}
__device__ void node_process_wave(const DevView& v, int node_idx, uint32_t blk, uint32_t cmd_block) {
    const NodeDesc nd = v.nodes[node_idx];
    if (nd.is_graph_io || nd.kind == K_FIR) return;  // I/O edges (k_graph_in/out); FIR banks run as MFMA GEMMs
    const int lane = threadIdx.x & (WAVE - 1);
    WaveIO io;
    io.pool = v.pool + (size_t)blk * v.pool_blk_stride;
    io.flags = v.flags + (size_t)blk * v.flags_blk_stride;
    io.in_buf = v.in_buf + nd.in_off;
    io.out_buf = v.out_buf + nd.out_off;
    io.stride = v.stride;
    io.lane = lane;
    io.frames = v.frames;
    const int frames = v.frames;

    // in_silence_mask from the per-buffer flags (schedule.rs:305-320); unconnected inputs read buffer 0,
    // the constant zero buffer whose flag is always set (== should_clear).
    bool fl = lane < nd.n_in ? (io.flags[io.in_buf[lane]] != 0) : false;
    const uint64_t in_mask = __ballot(fl);
    uint64_t out_mask = 0;  // processor.rs:233

    NodeState s;
    const bool stateful = kind_is_stateful(nd.kind);
    if (stateful) {
        s = v.states[nd.state];
        apply_cmds(s, nd.state, cmd_block, v.cmds, v.n_cmds, v.samples, v.ext, lane == 0);
        if (nd.kind == K_BIQUAD && v.n_cmds) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // lane 0's coefficient stores
    }

    switch (nd.kind) {
        case K_DUMMY:  // nodes/dummy.rs:33-42 — writes nothing
            break;

        case K_VOLUME: {  // nodes/volume.rs:84-145
            float raw = s.p0;
            if (mask_all(in_mask, nd.n_in)) {  // :94-100
                smoother_reset(s.s0, raw);
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            GainRun run = smoother_begin(s.s0, raw, frames);  // :102
            if (!smoother_is_smoothing(s.s0) && run.c < 0.00001f) {  // :104-108
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            out_mask = in_mask;  // :110
            const bool stereo = nd.n_in == 2 && nd.n_out == 2;
            const int nch = nd.n_in < nd.n_out ? nd.n_in : nd.n_out;
            for (int base = 0; base < frames; base += 256) {
                int n = frames - base < 256 ? frames - base : 256;
                v4f g = gain_chunk(run, n, lane);
                int f0 = base + lane * 4;
                if (f0 >= frames) continue;
                for (int c = 0; c < nch; ++c) {
                    v4f y;
                    if (!stereo && mask_bit(in_mask, c)) y = splat(0.f);  // :132-135 (Q15)
                    else y = *(const v4f*)(io.in(c) + f0) * g;            // :123-126, :140-142
                    *(v4f*)(io.out(c) + f0) = y;
                }
            }
            if (run.ramp) s.s0.last = run.prev;  // :177
            break;
        }

        case K_PAN: {  // SPEC node: volume.rs stereo path with one smoother per channel
            float tl = s.p0, tr = s.p1;
            if (mask_all(in_mask, nd.n_in)) {
                smoother_reset(s.s0, tl);
                smoother_reset(s.s1, tr);
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            GainRun rl = smoother_begin(s.s0, tl, frames);
            GainRun rr = smoother_begin(s.s1, tr, frames);
            out_mask = in_mask;
            for (int base = 0; base < frames; base += 256) {
                int n = frames - base < 256 ? frames - base : 256;
                v4f gl = gain_chunk(rl, n, lane);
                v4f gr = gain_chunk(rr, n, lane);
                int f0 = base + lane * 4;
                if (f0 >= frames) continue;
                *(v4f*)(io.out(0) + f0) = *(const v4f*)(io.in(0) + f0) * gl;
                *(v4f*)(io.out(1) + f0) = *(const v4f*)(io.in(1) + f0) * gr;
            }
            if (rl.ramp) s.s0.last = rl.prev;
            if (rr.ramp) s.s1.last = rr.prev;
            break;
        }

        case K_SUM: {  // nodes/sum.rs:41-136
            const int n_in = nd.n_in, n_out = nd.n_out, ports = nd.aux0;
            if (mask_all(in_mask, n_in)) {  // :52-56
                out_mask = clear_all_outputs(io, 0, n_out);
                break;
            }
            if (n_in == n_out) {  // :58-65 (Q14)
                for (int c = 0; c < n_out; ++c)
                    for (int f0 = lane * 4; f0 < frames; f0 += 256) *(v4f*)(io.out(c) + f0) = *(const v4f*)(io.in(c) + f0);
                out_mask = in_mask;
                break;
            }
            const bool masked = !(ports == 2 || ports == 3 || ports == 4);  // :67-133 (Q13)
            // lane i keeps the buffer id of input channel i; ids are broadcast with v_readlane so the
            // per-port loads are independent and can be in flight together (8 at a time)
            const int my_in = lane < n_in ? io.in_buf[lane] : 0;
            const uint64_t later_ports = mask_all_silent_bits(n_in) & ~mask_all_silent_bits(n_out);
            const bool any_skip = masked && (in_mask & later_ports) != 0;
            for (int c = 0; c < n_out; ++c) {
                for (int f0 = lane * 4; f0 < frames; f0 += 256) {
                    v4f acc = *(const v4f*)(io.pool + (size_t)__builtin_amdgcn_readlane(my_in, c) * io.stride + f0);
                    if (!any_skip) {
                        for (int p0 = 1; p0 < ports; p0 += 8) {
                            v4f x[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (p0 + u < ports)
                                    x[u] = *(const v4f*)(io.pool +
                                                         (size_t)__builtin_amdgcn_readlane(my_in, n_out * (p0 + u) + c) * io.stride + f0);
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (p0 + u < ports) acc = acc + x[u];  // left-assoc, port order (:78,92,107,129)
                        }
                    } else {
                        for (int p = 1; p < ports; ++p) {
                            int ic = n_out * p + c;
                            if (mask_bit(in_mask, ic)) continue;  // :122-124
                            acc = acc + *(const v4f*)(io.in(ic) + f0);
                        }
                    }
                    *(v4f*)(io.out(c) + f0) = acc;
                }
            }
            break;
        }

        case K_SAMPLER: {  // nodes/sampler.rs:323-561 (messages already applied above)
            if (s.sample < 0 || !s.playing) {  // :416-430
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            GainRun run = smoother_begin(s.s0, s.p0, frames);        // :432-433
            if (!smoother_is_smoothing(s.s0) && run.c < 0.00001f) {  // :437-443
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            const SampleDesc sd = v.samples[s.sample];
            Fetch ft;
            if (!sampler_advance(s, sd.frames, (uint32_t)frames, ft)) {  // :486-497
                if (run.ramp) {  // the smoother already ran this block (:433) — keep its state exact
                    for (int base = 0; base < frames; base += 256) {
                        int n = frames - base < 256 ? frames - base : 256;
                        (void)ramp_chunk(run, n, lane);
                    }
                    s.s0.last = run.prev;
                }
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            const int sch = sd.channels;
            const int nfill = nd.n_out < sch ? nd.n_out : sch;  // fill_buffers zip + gain zip (:535)
            for (int base = 0; base < frames; base += 256) {
                int n = frames - base < 256 ? frames - base : 256;
                v4f g = gain_chunk(run, n, lane);
                int f0 = base + lane * 4;
                if (f0 >= frames) continue;
                v4f first = splat(0.f);
                for (int c = 0; c < nfill; ++c) {
                    v4f x = sample_fetch4(sd, c, ft, (uint32_t)f0, (uint32_t)frames) * g;  // :521-543
                    if (c == 0) first = x;
                    *(v4f*)(io.out(c) + f0) = x;
                }
                if (nd.n_out > sch) {  // :545-559
                    if (nd.n_out == 2 && sch == 1) {
                        *(v4f*)(io.out(1) + f0) = first;
                    } else {
                        for (int c = sch; c < nd.n_out; ++c) *(v4f*)(io.out(c) + f0) = splat(0.f);
                    }
                }
            }
            if (nd.n_out > sch && !(nd.n_out == 2 && sch == 1))
                for (int c = sch; c < nd.n_out; ++c) out_mask |= (1ull << c);  // :556
            if (run.ramp) s.s0.last = run.prev;
            break;
        }

        case K_BEEP: {  // nodes/beep_test.rs:71-97
            if (nd.n_out == 0) break;
            if (!s.enabled) {  // :83-86 (Q12): channel 0 untouched, mask = new_all_silent(n-1)
                out_mask = clear_all_outputs(io, 1, nd.n_out);
                break;
            }
            const float TAU = 6.28318530717958647692528676655900577f;
            float ph = s.phasor;
            const float inc = s.phasor_inc;
            for (int base = 0; base < frames; base += 256) {
                int n = frames - base < 256 ? frames - base : 256;
                v4f p4 = splat(0.f);
                int q = 0;  // serial phasor (:90); lane keeps the four phases of its frames
                for (; q * 4 + 4 <= n; ++q) {
                    float a0 = ph;
                    ph = beep_step(ph, inc);
                    float a1 = ph;
                    ph = beep_step(ph, inc);
                    float a2 = ph;
                    ph = beep_step(ph, inc);
                    float a3 = ph;
                    ph = beep_step(ph, inc);
                    if (q == lane) p4 = (v4f){a0, a1, a2, a3};
                }
                int rem = n - q * 4;
                if (rem > 0) {
                    float a0 = ph;
                    ph = beep_step(ph, inc);
                    float a1 = ph;
                    if (rem > 1) ph = beep_step(ph, inc);
                    float a2 = ph;
                    if (rem > 2) ph = beep_step(ph, inc);
                    if (q == lane) p4 = (v4f){a0, a1, a2, 0.f};
                }
                int f0 = base + lane * 4;
                if (f0 >= frames) continue;
                v4f y;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = sinf(p4[j] * TAU) * s.gain;  // :89
                for (int c = 0; c < nd.n_out; ++c) *(v4f*)(io.out(c) + f0) = y;  // :93-95
            }
            s.phasor = ph;
            break;
        }

        case K_HARD_CLIP: {  // nodes/hard_clip.rs:51-95
            const float t = s.p0;
            const bool fast = nd.n_in == 2 && nd.n_out == 2 && !mask_any(in_mask, 2);  // :60-63 (Q16)
            const int nch = nd.n_in < nd.n_out ? nd.n_in : nd.n_out;
            for (int c = 0; c < nch; ++c) {
                const bool sil = !fast && mask_bit(in_mask, c);
                for (int f0 = lane * 4; f0 < frames; f0 += 256) {
                    v4f y = splat(0.f);
                    if (!sil) {
                        v4f x = *(const v4f*)(io.in(c) + f0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[j] = clipf(x[j], t);
                    }
                    *(v4f*)(io.out(c) + f0) = y;
                }
            }
            if (!fast) out_mask = in_mask;  // :93
            break;
        }

        case K_MONO_TO_STEREO: {  // nodes/mono_to_stereo.rs:33-50
            if (mask_bit(in_mask, 0)) {
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            for (int f0 = lane * 4; f0 < frames; f0 += 256) {
                v4f x = *(const v4f*)(io.in(0) + f0);
                *(v4f*)(io.out(0) + f0) = x;
                *(v4f*)(io.out(1) + f0) = x;
            }
            break;
        }

        case K_STEREO_TO_MONO: {  // nodes/stereo_to_mono.rs:33-56
            if (mask_all(in_mask, 2) || nd.n_in < 2 || nd.n_out == 0) {
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            for (int f0 = lane * 4; f0 < frames; f0 += 256) {
                v4f a = *(const v4f*)(io.in(0) + f0);
                v4f b = *(const v4f*)(io.in(1) + f0);
                *(v4f*)(io.out(0) + f0) = (a + b) * 0.5f;
            }
            break;
        }
        case K_WIDTH: {  // SPEC: mid/side width, one smoothed parameter
            if (mask_all(in_mask, nd.n_in)) {
                smoother_reset(s.s0, s.p0);
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            GainRun run = smoother_begin(s.s0, s.p0, frames);
            for (int base = 0; base < frames; base += 256) {
                int n = frames - base < 256 ? frames - base : 256;
                v4f w = gain_chunk(run, n, lane);
                int f0 = base + lane * 4;
                if (f0 >= frames) continue;
                v4f l = *(const v4f*)(io.in(0) + f0);
                v4f r = *(const v4f*)(io.in(1) + f0);
                v4f m = (l + r) * 0.5f;
                v4f sd = ((l - r) * 0.5f) * w;
                *(v4f*)(io.out(0) + f0) = m + sd;
                *(v4f*)(io.out(1) + f0) = m - sd;
            }
            if (run.ramp) s.s0.last = run.prev;
            break;
        }

        case K_BIQUAD: {  // SPEC: RBJ biquad, Direct Form I, f32 state: unfused feed-forward half, then
            // y = fma(-a1, y1, fma(-a2, y2, ff)) (one fma on the recurrence's critical path; SPEC: DESIGN.md §6).
            // Serial in time: lane c runs channel c (the generic executor's coverage path).
            float* ext = v.ext + s.ext_off;
            const int nch = nd.n_in < nd.n_out ? nd.n_in : nd.n_out;
            if (lane < nch) {
                const float b0 = ext[0], b1 = ext[1], b2 = ext[2], a1 = ext[3], a2 = ext[4];
                float* st = ext + 5 + 4 * lane;
                float x1 = st[0], x2 = st[1], y1 = st[2], y2 = st[3];
                const float* in = io.in(lane);
                float* out = io.out(lane);
                for (int i = 0; i < frames; ++i) {
                    float x = in[i];
                    float acc = b0 * x;
                    acc = acc + (b1 * x1);
                    acc = acc + (b2 * x2);
                    acc = __builtin_fmaf(-a2, y2, acc);
                    acc = __builtin_fmaf(-a1, y1, acc);
                    x2 = x1;
                    x1 = x;
                    y2 = y1;
                    y1 = acc;
                    out[i] = acc;
                }
                st[0] = x1;
                st[1] = x2;
                st[2] = y1;
                st[3] = y2;
            }
            break;
        }

        case K_DELAY: {  // SPEC: integer-sample delay line with feedback, ring per channel in the ext pool
            const uint32_t D = (uint32_t)s.loop_end;
            const uint32_t pos = (uint32_t)s.playhead;
            const float fb = s.p0, mix = s.p1, dry = s.gain;
            const int nch = nd.n_in < nd.n_out ? nd.n_in : nd.n_out;
            const uint32_t chunk = D < 64u ? D : 64u;  // frames inside one chunk touch distinct ring slots
            for (int c = 0; c < nch; ++c) {
                float* ring = v.ext + s.ext_off + (size_t)c * D;
                const float* in = io.in(c);
                float* out = io.out(c);
                for (uint32_t base = 0; base < (uint32_t)frames; base += chunk) {
                    uint32_t i = base + (uint32_t)lane;
                    if ((uint32_t)lane < chunk && i < (uint32_t)frames) {
                        uint32_t slot = (pos + i) % D;
                        float x = in[i];
                        float d = ring[slot];
                        ring[slot] = x + (d * fb);
                        out[i] = (x * dry) + (d * mix);
                    }
                    if (D < (uint32_t)frames) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // next chunk re-reads these slots
                }
            }
            s.playhead = (uint64_t)((pos + (uint32_t)frames) % D);
            break;
        }

        case K_RESAMPLER: {  // SPEC: resampling source, polyphase windowed sinc
            const SampleDesc sd = s.sample >= 0 ? v.samples[s.sample] : SampleDesc{nullptr, 0, 0, FMT_P_F32};
            if (!s.playing || s.sample < 0 || sd.frames == 0) {
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            const uint64_t step = s.loop_start, pos = s.playhead;
            const bool loop = s.has_loop != 0;
            const int64_t len = (int64_t)sd.frames;
            const int sch = sd.channels;
            const int nfill = nd.n_out < sch ? nd.n_out : sch;
            for (int i = lane; i < frames; i += WAVE) {  // every output frame is independent
                const uint64_t p = pos + (uint64_t)i * step;
                const int64_t idx = (int64_t)(p >> 32);
                const float* hp = v.rs_table + ((uint32_t)(p >> 27) & (RS_PHASES - 1)) * RS_TAPS;
                float first = 0.f;
                for (int c = 0; c < nfill; ++c) {
                    float acc = 0.f;
                    for (int k = 0; k < RS_TAPS; ++k) {  // ascending-tap fmaf chain from +0.0 (the SPEC order)
                        int64_t j = idx - (RS_TAPS / 2 - 1) + k;
                        float x = 0.f;
                        if (loop) {
                            j %= len;
                            if (j < 0) j += len;
                            x = sample_fetch(sd, c, (uint64_t)j);
                        } else if (j >= 0 && j < len) {
                            x = sample_fetch(sd, c, (uint64_t)j);
                        }
                        acc = __builtin_fmaf(hp[k], x, acc);
                    }
                    io.out(c)[i] = acc;
                    if (c == 0) first = acc;
                }
                if (nd.n_out > sch) {
                    if (nd.n_out == 2 && sch == 1) io.out(1)[i] = first;
                    else
                        for (int c = sch; c < nd.n_out; ++c) io.out(c)[i] = 0.f;
                }
            }
            if (nd.n_out > sch && !(nd.n_out == 2 && sch == 1))
                for (int c = sch; c < nd.n_out; ++c) out_mask |= (1ull << c);
            uint64_t np = pos + (uint64_t)frames * step;
            if (loop) np %= ((uint64_t)len << 32);
            else if ((np >> 32) >= (uint64_t)len + RS_TAPS / 2) s.playing = 0;
            s.playhead = np;
            break;
        }

        case K_SPATIAL: {  // SPEC: distance gain + equal-power pan + per-ear integer delay
            float* hist = v.ext + s.ext_off;
            const int dl = s.playing, dr = s.has_loop;
            const float hreg = hist[lane];  // lane l keeps hist[l] (SP_HIST == 64), hist[63] = newest
            GainRun rl = smoother_begin(s.s0, s.p0, frames);
            GainRun rr = smoother_begin(s.s1, s.p1, frames);
            const bool two = nd.n_in >= 2;
            auto mono = [&](int j) -> float {  // m[j], j >= 0
                return two ? (io.in(0)[j] + io.in(1)[j]) * 0.5f : io.in(0)[j];
            };
            for (int base = 0; base < frames; base += 256) {
                int n = frames - base < 256 ? frames - base : 256;
                v4f gl = gain_chunk(rl, n, lane);
                v4f gr = gain_chunk(rr, n, lane);
                int f0 = base + lane * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = f0 + e;
                    const int jl = i - dl, jr = i - dr;
                    // history lookups go through the wave (every lane takes part), current-block ones through memory
                    const float hl = __shfl(hreg, (SP_HIST + jl) & 63), hr = __shfl(hreg, (SP_HIST + jr) & 63);
                    if (i < frames) {
                        const float ml = jl >= 0 ? mono(jl) : hl;
                        const float mr = jr >= 0 ? mono(jr) : hr;
                        io.out(0)[i] = ml * gl[e];
                        io.out(1)[i] = mr * gr[e];
                    }
                }
            }
            if (rl.ramp) s.s0.last = rl.prev;
            if (rr.ramp) s.s1.last = rr.prev;
            // new history = the last SP_HIST samples of (hist ++ m[0..frames))
            const int j = frames - SP_HIST + lane;
            const float keep = __shfl(hreg, (SP_HIST + j) & 63);
            hist[lane] = j >= 0 ? mono(j) : keep;
            break;
        }

        default: break;
    }

    if (stateful && lane == 0) v.states[nd.state] = s;
    // schedule.rs:338-341: every output buffer's flag is overwritten with the node's out mask bit
    if (lane < nd.n_out) io.flags[io.out_buf[lane]] = mask_bit(out_mask, lane) ? 1 : 0;
}

// K blocks per launch (gridDim.y = K, one pool slice per block).  A node whose audio half carries state from block
// to block is run by ONE wave that walks its K blocks in order; stateless nodes take their K blocks in parallel.
__global__ __launch_bounds__(WAVE* WPB) void k_level(DevView v, const int* __restrict__ level_nodes, int n_nodes,
                                                      uint32_t cmd_block0) {
    int w = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (w >= n_nodes) return;
    const int node = level_nodes[w];
    if (kind_is_stateful(v.nodes[node].kind)) {
        if (blockIdx.y != 0) return;
        for (uint32_t b = 0; b < gridDim.y; ++b) node_process_wave(v, node, b, cmd_block0 + b);
    } else {
        node_process_wave(v, node, blockIdx.y, cmd_block0 + blockIdx.y);
    }
}

// B1: one node on scratch buffers (single wave)
__global__ __launch_bounds__(WAVE) void k_single_node(DevView v, int node_idx) { node_process_wave(v, node_idx, 0, 0); }

// ------------------------------------------------------------------ state init / graph I/O edges
struct StateInit {
    int index;
    int pad;
    NodeState st;
};
__global__ void k_scatter_states(NodeState* states, const uint8_t* __restrict__ inits, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const StateInit* in = (const StateInit*)inits + i;
    states[in->index] = in->st;
}

// processor.rs:99-115 + schedule.rs:213-253 + util.rs:44-87.  Q10: the graph_in Dummy node's out mask (0)
// overwrites whatever prepare_graph_inputs computed, so every graph-input buffer flag ends up false.
__global__ void k_graph_in(float* pool, uint8_t* flags, int stride, size_t pool_blk_stride, size_t flags_blk_stride,
                           const int* __restrict__ bufs, int n_bufs, const float* __restrict__ interleaved, int n_in_ch,
                           int frames) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    int c = blockIdx.y;
    const uint32_t blk = blockIdx.z;  // K-batched: one pool slice per block
    pool += (size_t)blk * pool_blk_stride;
    flags += (size_t)blk * flags_blk_stride;
    if (f < frames) {
        float x = c < n_in_ch ? interleaved[((size_t)blk * frames + f) * n_in_ch + c] : 0.f;  // extra graph inputs zero-filled
        pool[(size_t)bufs[c] * stride + f] = x;
    }
    if (f == 0) flags[bufs[c]] = 0;
}

// processor.rs:120-148 + schedule.rs:255-287 + util.rs:90-147.  K-batched: blockIdx.y = block.
__global__ void k_graph_out(const float* __restrict__ pool, const uint8_t* __restrict__ flags, int stride,
                            size_t pool_blk_stride, size_t flags_blk_stride, const int* __restrict__ bufs, int n_bufs,
                            float* __restrict__ out, int n_out_ch, int frames) {
    const uint32_t blk = blockIdx.y;
    const float* p = pool + (size_t)blk * pool_blk_stride;
    const uint8_t* fl = flags + (size_t)blk * flags_blk_stride;
    float* o = out + (size_t)blk * frames * n_out_ch;
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    int n_read = n_bufs < n_out_ch ? n_bufs : n_out_ch;  // read_output_len
    if (n_read == 2 && n_out_ch == 2) {                   // interleave_stereo (util.rs:123-147)
        bool both = fl[bufs[0]] && fl[bufs[1]];
        float2 y;
        y.x = both ? 0.f : p[(size_t)bufs[0] * stride + f];
        y.y = both ? 0.f : p[(size_t)bufs[1] * stride + f];
        *(float2*)(o + (size_t)f * 2) = y;
        return;
    }
    for (int c = 0; c < n_out_ch; ++c) {  // interleave (util.rs:90-120): zero-fill, skip silent channels
        float y = 0.f;
        if (c < n_read && !fl[bufs[c]]) y = p[(size_t)bufs[c] * stride + f];
        o[(size_t)f * n_out_ch + c] = y;
    }
}

__global__ void k_set_flags(uint8_t* flags, const int* __restrict__ bufs, int n, uint64_t mask) {
    int i = threadIdx.x;
    if (i < n) flags[bufs[i]] = (mask >> i) & 1ull;
}
__global__ void k_get_flags(const uint8_t* flags, const int* __restrict__ bufs, int n, uint64_t* mask) {
    bool f = (int)threadIdx.x < n ? flags[bufs[threadIdx.x]] != 0 : false;
    uint64_t m = __ballot(f);
    if (threadIdx.x == 0) *mask = m;
}

// ------------------------------------------------------------------ message lookups shared by the fused plans
// first command of (state, block) in the (state, block, seq)-sorted list
__device__ inline int chain_cmd_lower_bound(const Cmd* cmds, int n_cmds, int state_idx, uint32_t block) {
    int lo = 0, hi = n_cmds;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        const Cmd& c = cmds[mid];
        bool less = c.state < state_idx || (c.state == state_idx && c.block < block);
        if (less) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
// Both helpers return by value and are force-inlined: a by-reference out-parameter of a real call would pin the
// caller's loop-carried registers to scratch memory (a scratch load per step, draining vmcnt with it).
struct ChainCoefs {
    bool found;
    float b0, b1, b2, a1, a2;
};
// the last CMD_SET_COEFS for (state, block), if any
__device__ __forceinline__ ChainCoefs chain_find_coefs(const Cmd* cmds, int n_cmds, int state_idx, uint32_t block) {
    ChainCoefs r;
    r.found = false;
    r.b0 = r.b1 = r.b2 = r.a1 = r.a2 = 0.f;
    for (int i = chain_cmd_lower_bound(cmds, n_cmds, state_idx, block); i < n_cmds; ++i) {
        const Cmd c = cmds[i];
        if (c.state != state_idx || c.block != block) break;
        if (c.type != CMD_SET_COEFS) continue;
        r.b0 = c.f0;
        r.b1 = __int_as_float(c.i0);
        r.b2 = __int_as_float(c.i1);
        unsigned long long u = (unsigned long long)__double_as_longlong(c.d0);
        r.a1 = __int_as_float((int)(u & 0xffffffffull));
        r.a2 = __int_as_float((int)(u >> 32));
        r.found = true;
    }
    return r;
}
// delay parameters: fb (p0), mix (p1), dry (gain)
struct ChainDelay {
    float fb, mix, dry;
};
__device__ __forceinline__ ChainDelay chain_delay_cmds(const Cmd* cmds, int n_cmds, int state_idx, uint32_t block, ChainDelay p) {
    for (int i = chain_cmd_lower_bound(cmds, n_cmds, state_idx, block); i < n_cmds; ++i) {
        const Cmd c = cmds[i];
        if (c.state != state_idx || c.block != block) break;
        if (c.type == CMD_SET_P0) p.fb = c.f0;
        else if (c.type == CMD_SET_P1) p.mix = c.f0;
        else if (c.type == CMD_SET_GAIN) p.dry = c.f0;
    }
    return p;
}

// ------------------------------------------------------------------ fused voice-bank plan
// Control kernel (k_voice_control): one thread per voice runs the per-block state machines of its whole
// chain in schedule order (sampler -> stage nodes) and emits one VoiceBlk per block.  As soon as the voice
// is STEADY (no message left for it in this call, every smoother constant) the remaining blocks only differ
// by the playhead, and the thread finishes the call with a short descriptor-store loop.  Per-frame ramps
// (ParamSmoother Active) are materialised into `ramps` only for blocks where the values actually change.
struct StageRegs {  // the NodeState prefix (p0,p1,s0,s1) a gain stage needs
    float p0, p1;
    Smoother s0, s1;
};

// Serial ramp -> global memory; returns false (and writes nothing) when the recurrence is already at its
// f32 fixed point (Q28: an Active smoother can stall above settle_epsilon forever) — the block is constant.
__device__ __forceinline__ bool ramp_emit(GainRun& r, int frames, float* dst0, float* dst1, bool write) {
    float prev = r.prev;
    float v0 = r.in_a + (prev * r.b);
    if (v0 == prev) {  // fixed point: every later value equals prev, bit for bit
        r.c = prev;
        r.ramp = 0;
        return false;
    }
    for (int i = 0; i < frames; ++i) {
        prev = r.in_a + (prev * r.b);
        if (write) {
            dst0[i] = prev;
            if (dst1) dst1[i] = prev;
        }
    }
    r.prev = prev;
    return true;
}

// A smoother whose next set_and_process(target) returns the same constant and leaves its state untouched:
// not Active, or Active but stalled at the f32 fixed point above settle_epsilon (Q28).
__device__ __forceinline__ bool smoother_is_constant(const Smoother& s, float target) {
    if (!(s.input == target)) return false;
    if (s.status != SM_ACTIVE) return true;
    float y0 = (s.input * s.a) + (s.last * s.b);
    return y0 == s.last && !(fabsf(s.input - y0) < s.eps);
}

// source pointers of a block whose frames are contiguous planar f32 (the fast path of the leaf kernel)
__device__ __forceinline__ void blk_set_source(VoiceBlk& d, const SampleDesc& sd, int frames) {
    d.src_l = nullptr;
    d.src_r = nullptr;
    const bool contiguous = !(d.flags & (VB_WRAP | VB_TAIL_ZERO | VB_SILENT)) && sd.format == FMT_P_F32;
    if (contiguous) {
        d.src_l = (const float*)sd.data + d.off0;
        d.src_r = (d.flags & VB_MONO) ? d.src_l : d.src_l + sd.frames;
        // VB_SIMPLE blocks carry no full descriptor, so they must never need the per-element path (ragged tail)
        if ((d.flags >> VB_RAMP_SHIFT) == 0 && (frames & 3) == 0 && sd.frames < 0xffffffffull) d.flags |= VB_SIMPLE;
    }
}

// last block index (relative to this call) that still has a message for node `state_idx`; -1 if none
__device__ inline int last_cmd_block(const Cmd* cmds, int n_cmds, int state_idx, uint32_t cmd_block0) {
    if (n_cmds == 0) return -1;
    int lo = 0, hi = n_cmds;  // upper bound of state_idx
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (cmds[mid].state <= state_idx) lo = mid + 1;
        else hi = mid;
    }
    if (lo == 0 || cmds[lo - 1].state != state_idx) return -1;
    return (int)(cmds[lo - 1].block - cmd_block0);  // sorted by (state, block): the last one is the latest
}

// Everything the steady tail of a call needs: the descriptor all its blocks share and how the playhead moves.
struct TailJob {
    int mode;          // 0 = nothing moves, 1 = looping playhead, 2 = one-shot playhead
    uint32_t flags;    // VB_SILENT / VB_MONO of the shared descriptor
    int sample;
    GainSet g;
    uint64_t playhead, loop_start, loop_end;
};

// Writes the compact record (always) and the full descriptor (only when the leaf kernel will need it).
// `fx`: the voice has a biquad / delay (k_chain plan) — its source is needed even when the chain output is
// silent, and every block that is not VB_SIMPLE carries a full descriptor.
__device__ __forceinline__ void put_blk(const FusedView& fv, int vi, int kk, const VoiceBlk& d, uint32_t gset,
                                        uint64_t sample_frames, bool fx) {
    VoiceRef ref;
    ref.src_l = d.src_l;
    ref.r_delta = ((d.flags & VB_SIMPLE) && !(d.flags & (VB_MONO | VB_SRC_ZERO))) ? (uint32_t)sample_frames : 0u;
    ref.flags_gset = (d.flags & 0xffu) | (gset << 8);
    fv.refs[(size_t)vi * fv.refs_stride + kk] = ref;  // [voice][block]: the tail lanes store 1 KiB contiguous
    const bool need_full = fx ? !(d.flags & VB_SIMPLE) : !(d.flags & (VB_SIMPLE | VB_SILENT));
    if (need_full) fv.blks[(size_t)kk * fv.n_voices + vi] = d;
}

// Steady tail: blocks k_first .. K-1 share one descriptor; only the playhead moves, by +frames with a wrap at
// the loop end (nodes/sampler.rs:445-484) — closed form (base + j*frames) mod L, so the 64 lanes of the
// voice's wave fill 64 blocks at a time.  Returns the playhead the reference holds after block K-1.
__device__ __forceinline__ uint64_t steady_tail(const FusedView& fv, int vi, int lane, int k_first, int K, const TailJob& job,
                                                const SampleDesc& sd, uint32_t gset, bool simple_ok, bool fx) {
    const int frames = fv.frames;
    const uint64_t fr = (uint64_t)frames;
    VoiceBlk t;
    t.flags = job.flags;
    t.n1 = frames;
    t.src_l = t.src_r = nullptr;
    t.off0 = t.off1 = 0;
    t.sample = job.sample;
    t.pad = 0;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES; ++j) {
        t.g[j][0] = job.g.g[j][0];
        t.g[j][1] = job.g.g[j][1];
    }
    const bool no_src = (job.flags & VB_SRC_ZERO) || (!fx && (job.flags & VB_SILENT));
    const bool contiguous_f32 = !no_src && job.sample >= 0 && sd.format == FMT_P_F32;
    const uint64_t n = (uint64_t)(K - k_first);
    if (job.mode == 1) {
        // all quantities fit 32 bits whenever the loop does (the usual case): avoid 64-bit division
        const uint64_t L = job.loop_end - job.loop_start;
        const uint64_t base = job.playhead >= job.loop_end ? 0 : job.playhead - job.loop_start;
        uint64_t r, step, r_last;
        if (L <= 0xffffffffull && n * fr <= 0xffffffffull) {
            // everything fits 32 bits (the usual case): 32-bit remainders instead of 64-bit division
            const uint32_t l32 = (uint32_t)L;
            auto addmod = [&](uint32_t j) -> uint64_t {  // (base + j*fr) mod L, base < L
                uint64_t x = (uint64_t)((j * (uint32_t)fr) % l32) + base;
                return x >= L ? x - L : x;
            };
            r = addmod((uint32_t)lane);
            step = (uint64_t)((64u * (uint32_t)fr) % l32);
            r_last = addmod((uint32_t)(n - 1));
        } else {
            r = (base + (uint64_t)lane * fr) % L;
            step = (64ull * fr) % L;
            r_last = (base + (n - 1) * fr) % L;
        }
        for (int k2 = k_first + lane; k2 < K; k2 += WAVE) {
            const uint64_t left = L - r;
            t.flags = job.flags;
            t.off0 = job.loop_start + r;
            t.off1 = job.loop_start;
            t.src_l = t.src_r = nullptr;
            if (left < fr) {  // wraps inside the block
                t.n1 = (uint32_t)left;
                t.flags |= VB_WRAP;
            } else {
                t.n1 = frames;
                if (contiguous_f32) {
                    t.src_l = (const float*)sd.data + t.off0;
                    t.src_r = (t.flags & VB_MONO) ? t.src_l : t.src_l + sd.frames;
                    if (simple_ok) t.flags |= VB_SIMPLE;
                }
            }
            put_blk(fv, vi, k2, t, gset, sd.frames, fx);
            r += step;
            if (r >= L) r -= L;
        }
        const uint64_t left = L - r_last;
        return left < fr ? job.loop_start + (fr - left) : job.loop_start + r_last + fr;
    }
    if (job.mode == 2) {
        for (int k2 = k_first + lane; k2 < K; k2 += WAVE) {
            t.flags = job.flags;
            t.off0 = job.playhead + (uint64_t)(k2 - k_first) * fr;
            t.src_l = t.src_r = nullptr;
            if (contiguous_f32) {
                t.src_l = (const float*)sd.data + t.off0;
                t.src_r = (t.flags & VB_MONO) ? t.src_l : t.src_l + sd.frames;
                if (simple_ok) t.flags |= VB_SIMPLE;
            }
            put_blk(fv, vi, k2, t, gset, sd.frames, fx);
        }
        return job.playhead + n * fr;
    }
    // nothing moves (mode 0 <=> the sampler is frozen): with fx the block still runs (zeros in, constant gains)
    if (fx && simple_ok) t.flags |= VB_SIMPLE;
    for (int k2 = k_first + lane; k2 < K; k2 += WAVE) put_blk(fv, vi, k2, t, fx ? gset : 0u, sd.frames, fx);
    return job.playhead;
}

// One WAVE per voice: the state machines are run by all 64 lanes redundantly (wave-uniform; lane 0 stores),
// the steady tail is split across the lanes.  A voice that ended the previous call steady and has no message
// in this one skips the state machines altogether (VoiceCache): its whole call is a steady tail.
__global__ __launch_bounds__(256) void k_voice_control(FusedView fv, int K, uint32_t cmd_block0) {
    const int vi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (vi >= fv.n_voices) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const bool w0 = lane == 0;
    const VoiceDesc vd = fv.voices[vi];
    const int frames = fv.frames;
    const bool simple_frames = (frames & 3) == 0;
    const bool fx = vd.bq_state >= 0 || vd.dl_state >= 0;  // k_chain plan voice

    // ---- k_chain plan: what both channel workgroups of the voice's leaf share is owned HERE — the record holds the
    // values at the start of this call (k_chain replays the call's messages block by block from them), the node state
    // is advanced to the end of the call.  k_chain itself only reads the record.
    if (fx) {
        ChainStart cs;
        cs.pos = 0;
        cs.fb = 0.f;
        cs.mix = 0.f;
        cs.dry = 1.f;
        cs.co[0] = 1.f;
        cs.co[1] = cs.co[2] = cs.co[3] = cs.co[4] = 0.f;
        cs.pad[0] = cs.pad[1] = cs.pad[2] = 0;
        if (vd.dl_state >= 0) {
            NodeState* ds = &fv.states[vd.dl_state];
            const uint64_t D = ds->loop_end;
            cs.pos = (uint32_t)ds->playhead;
            ChainDelay p = ChainDelay{ds->p0, ds->p1, ds->gain};
            cs.fb = p.fb;
            cs.mix = p.mix;
            cs.dry = p.dry;
            if (fv.n_cmds) {
                for (int i = chain_cmd_lower_bound(fv.cmds, fv.n_cmds, vd.dl_state, cmd_block0); i < fv.n_cmds; ++i) {
                    const Cmd c = fv.cmds[i];
                    if (c.state != vd.dl_state || c.block >= cmd_block0 + (uint32_t)K) break;
                    if (c.type == CMD_SET_P0) p.fb = c.f0;
                    else if (c.type == CMD_SET_P1) p.mix = c.f0;
                    else if (c.type == CMD_SET_GAIN) p.dry = c.f0;
                }
            }
            if (w0) {
                ds->playhead = ((uint64_t)cs.pos + (uint64_t)K * (uint64_t)frames) % D;
                ds->p0 = p.fb;
                ds->p1 = p.mix;
                ds->gain = p.dry;
            }
        }
        if (vd.bq_state >= 0) {
            float* co = fv.ext + fv.states[vd.bq_state].ext_off;
#pragma unroll
            for (int j = 0; j < 5; ++j) cs.co[j] = co[j];
            if (fv.n_cmds) {
                bool found = false;
                float nc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
                for (int i = chain_cmd_lower_bound(fv.cmds, fv.n_cmds, vd.bq_state, cmd_block0); i < fv.n_cmds; ++i) {
                    const Cmd c = fv.cmds[i];
                    if (c.state != vd.bq_state || c.block >= cmd_block0 + (uint32_t)K) break;
                    if (c.type != CMD_SET_COEFS) continue;
                    nc[0] = c.f0;
                    nc[1] = __int_as_float(c.i0);
                    nc[2] = __int_as_float(c.i1);
                    unsigned long long u = (unsigned long long)__double_as_longlong(c.d0);
                    nc[3] = __int_as_float((int)(u & 0xffffffffull));
                    nc[4] = __int_as_float((int)(u >> 32));
                    found = true;
                }
                if (found && w0) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) co[j] = nc[j];
                }
            }
        }
        if (w0) fv.chain_start[vi] = cs;
    }

    int last_cmd = -1;
    if (fv.n_cmds) {
        last_cmd = last_cmd_block(fv.cmds, fv.n_cmds, vd.sampler_state, cmd_block0);
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
            if (j < vd.n_stages) {
                int l = last_cmd_block(fv.cmds, fv.n_cmds, vd.stage_state[j], cmd_block0);
                last_cmd = l > last_cmd ? l : last_cmd;
            }
    }
    GainSet* my_gsets = fv.gsets + (size_t)vi * FW_GSETS;

    // ---- fast path: still steady from the previous call
    {
        const VoiceCache vc = fv.cache[vi];
        if (vc.epoch == fv.epoch && last_cmd < 0) {
            TailJob job;
            job.mode = vc.mode;
            job.flags = vc.flags;
            job.sample = vc.sample;
            job.g = vc.g;
            job.playhead = job.loop_start = job.loop_end = 0;
            SampleDesc sd;
            sd.data = nullptr;
            sd.frames = 0;
            sd.channels = 2;
            sd.format = FMT_P_F32;
            bool ok = true;
            if (vc.mode != 0) {
                const NodeState* sp = &fv.states[vd.sampler_state];
                job.playhead = sp->playhead;
                job.loop_start = sp->loop_start;
                job.loop_end = sp->loop_end;
                sd = fv.samples[vc.sample];
                if (vc.mode == 2 && job.playhead + (uint64_t)K * (uint64_t)frames > sd.frames) ok = false;  // ends in this call
            }
            if (ok) {
                const bool no_src = (job.flags & VB_SRC_ZERO) || (!fx && (job.flags & VB_SILENT));
                const bool simple_ok = no_src ? (fx && simple_frames)
                                              : (job.sample >= 0 && sd.format == FMT_P_F32 && simple_frames &&
                                                 sd.frames < 0xffffffffull);
                if (simple_ok && w0) my_gsets[0] = job.g;
                uint64_t ph = steady_tail(fv, vi, lane, 0, K, job, sd, 0u, simple_ok, fx);
                if (w0 && vc.mode != 0) fv.states[vd.sampler_state].playhead = ph;
                return;
            }
        }
    }

    // ---- general path
    NodeState ss = fv.states[vd.sampler_state];
    StageRegs st[FW_MAX_STAGES - 1];
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
        if (j < vd.n_stages) st[j] = *(const StageRegs*)&fv.states[vd.stage_state[j]];

    // gain sets used so far in this call (the current one is mirrored in registers)
    int n_gsets = 0;
    GainSet cur_gs;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES; ++j) cur_gs.g[j][0] = cur_gs.g[j][1] = 0.f;
    // picks (or allocates) the gain set of a VB_SIMPLE block; wave-uniform.  Returns its index.
    auto pick_gset = [&](VoiceBlk& d) -> uint32_t {
        if (!(d.flags & VB_SIMPLE)) return 0u;
        bool same = n_gsets > 0;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) same = same && cur_gs.g[j][0] == d.g[j][0] && cur_gs.g[j][1] == d.g[j][1];
        if (!same) {
            if (n_gsets < FW_GSETS) {
#pragma unroll
                for (int j = 0; j < FW_MAX_STAGES; ++j) {
                    cur_gs.g[j][0] = d.g[j][0];
                    cur_gs.g[j][1] = d.g[j][1];
                }
                if (w0) my_gsets[n_gsets] = cur_gs;
                n_gsets++;
            } else {
                d.flags &= ~VB_SIMPLE;  // out of gain-set slots: use the full descriptor for this block
                return 0u;
            }
        }
        return (uint32_t)(n_gsets - 1);
    };
    int cached_sample = -1;
    SampleDesc sd;
    sd.data = nullptr;
    sd.frames = 0;
    sd.channels = 2;
    sd.format = FMT_P_F32;
    bool became_steady = false;

    for (int k = 0; k < K; ++k) {
        const uint32_t cb = cmd_block0 + k;
        VoiceBlk d;
        d.flags = 0;
        d.n1 = frames;
        d.src_l = d.src_r = nullptr;
        d.off0 = d.off1 = 0;
        d.sample = -1;
        d.pad = 0;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) d.g[j][0] = d.g[j][1] = 1.0f;
        float* ramp_base = fv.ramps + ((size_t)k * fv.n_voices + vi) * (size_t)fv.ramp_slots * (size_t)fv.stride;

        // ---- sampler (nodes/sampler.rs:323-561)
        apply_cmds(ss, vd.sampler_state, cb, fv.cmds, fv.n_cmds, fv.samples);
        bool silent = true;
        if (ss.sample >= 0 && ss.playing) {
            GainRun run = smoother_begin(ss.s0, ss.p0, frames);
            if (!(!smoother_is_smoothing(ss.s0) && run.c < 0.00001f)) {
                if (cached_sample != ss.sample) {
                    sd = fv.samples[ss.sample];
                    cached_sample = ss.sample;
                }
                Fetch ft;
                bool ok = sampler_advance(ss, sd.frames, (uint32_t)frames, ft);
                if (run.ramp) {
                    if (ramp_emit(run, frames, ramp_base, ramp_base + fv.stride, w0)) {
                        d.flags |= 3u << VB_RAMP_SHIFT;
                        ss.s0.last = run.prev;
                    }
                }
                if (ok) {
                    silent = false;
                    d.sample = ss.sample;
                    d.off0 = ft.off0;
                    d.off1 = ft.off1;
                    d.n1 = ft.n1;
                    if (ft.wrap) d.flags |= VB_WRAP;
                    if (ft.tail_zero) d.flags |= VB_TAIL_ZERO;
                    if (sd.channels == 1) d.flags |= VB_MONO;
                    d.g[0][0] = d.g[0][1] = run.c;
                }
            }
        }
        // a biquad / delay between the sampler and the gain stages never reports silence (SPEC nodes: out mask 0)
        const bool src_silent = silent;
        if (fx) silent = false;
        // ---- chain stages in schedule order
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j) {
            if (j >= vd.n_stages) break;
            StageRegs& r = st[j];
            if (fv.n_cmds) {  // messages for this node (only p0/p1 apply to gain stages)
                NodeState tmp;
                tmp.p0 = r.p0;
                tmp.p1 = r.p1;
                apply_cmds(tmp, vd.stage_state[j], cb, fv.cmds, fv.n_cmds, fv.samples);
                r.p0 = tmp.p0;
                r.p1 = tmp.p1;
            }
            float* rb = ramp_base + (size_t)(j + 1) * 2 * fv.stride;
            if (vd.stage_kind[j] == K_VOLUME) {  // nodes/volume.rs:84-145
                if (silent) {
                    smoother_reset(r.s0, r.p0);
                } else {
                    GainRun run = smoother_begin(r.s0, r.p0, frames);
                    if (!smoother_is_smoothing(r.s0) && run.c < 0.00001f) {
                        silent = true;
                    } else {
                        if (run.ramp && ramp_emit(run, frames, rb, rb + fv.stride, w0)) {
                            d.flags |= 3u << (VB_RAMP_SHIFT + 2 * (j + 1));
                            r.s0.last = run.prev;
                        }
                        d.g[j + 1][0] = d.g[j + 1][1] = run.c;
                    }
                }
            } else {  // K_PAN (SPEC)
                if (silent) {
                    smoother_reset(r.s0, r.p0);
                    smoother_reset(r.s1, r.p1);
                } else {
                    GainRun rl = smoother_begin(r.s0, r.p0, frames);
                    GainRun rr = smoother_begin(r.s1, r.p1, frames);
                    if (rl.ramp && ramp_emit(rl, frames, rb, nullptr, w0)) {
                        d.flags |= 1u << (VB_RAMP_SHIFT + 2 * (j + 1));
                        r.s0.last = rl.prev;
                    }
                    if (rr.ramp && ramp_emit(rr, frames, rb + fv.stride, nullptr, w0)) {
                        d.flags |= 2u << (VB_RAMP_SHIFT + 2 * (j + 1));
                        r.s1.last = rr.prev;
                    }
                    d.g[j + 1][0] = rl.c;
                    d.g[j + 1][1] = rr.c;
                }
            }
        }
        if (!src_silent && (fx || !silent)) blk_set_source(d, sd, frames);
        else if (src_silent && fx && (d.flags >> VB_RAMP_SHIFT) == 0 && simple_frames) d.flags |= VB_SIMPLE;
        if (src_silent) d.flags |= VB_SRC_ZERO;
        if (silent) d.flags |= VB_SILENT;
        {
            uint32_t gs = pick_gset(d);
            if (w0) put_blk(fv, vi, k, d, gs, sd.frames, fx);
        }

        // ---- steady from the next block on?
        if (k < last_cmd) continue;
        bool steady = true;
        bool upstream_silent = false;
        int mode = 0;
        if (ss.sample < 0 || !ss.playing) {
            upstream_silent = true;  // frozen sampler: nothing moves
        } else {
            if (!smoother_is_constant(ss.s0, ss.p0)) steady = false;
            else if (ss.s0.status == SM_INACTIVE && ss.s0.input < 0.00001f) upstream_silent = true;  // muted, frozen
            else if (ss.has_loop) {
                uint64_t L = ss.loop_end - ss.loop_start;
                if (ss.loop_end > ss.loop_start && L >= (uint64_t)frames && ss.playhead >= ss.loop_start &&
                    ss.loop_end <= sd.frames && cached_sample == ss.sample)
                    mode = 1;
                else steady = false;
            } else {
                uint64_t need = (uint64_t)(K - 1 - k) * (uint64_t)frames;
                if (cached_sample == ss.sample && ss.playhead + need <= sd.frames) mode = 2;
                else steady = false;  // the one-shot ends inside this call: stay on the exact path
            }
        }
        bool sil = upstream_silent && !fx;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j) {
            if (j >= vd.n_stages || !steady) break;
            const StageRegs& r = st[j];
            if (sil) {  // reset() every block: idempotent once applied
                if (!(r.s0.status == SM_INACTIVE && r.s0.input == r.p0)) steady = false;
                if (vd.stage_kind[j] == K_PAN && !(r.s1.status == SM_INACTIVE && r.s1.input == r.p1)) steady = false;
            } else if (vd.stage_kind[j] == K_VOLUME) {
                if (!smoother_is_constant(r.s0, r.p0)) steady = false;
                else if (r.s0.status == SM_INACTIVE && r.s0.input < 0.00001f) sil = true;
            } else {
                if (!smoother_is_constant(r.s0, r.p0) || !smoother_is_constant(r.s1, r.p1)) steady = false;
            }
        }
        if (!steady) continue;
        // ---- steady: the descriptor every later block shares.  Constant gains are `input` for a settled
        // smoother and `last` for one stalled at its f32 fixed point (Q28).
        TailJob job;
        job.mode = mode;
        job.flags = (sil ? VB_SILENT : 0u) | (upstream_silent ? VB_SRC_ZERO : 0u);
        job.sample = upstream_silent ? -1 : ss.sample;
        job.playhead = ss.playhead;
        job.loop_start = ss.loop_start;
        job.loop_end = ss.loop_end;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) job.g.g[j][0] = job.g.g[j][1] = 1.0f;
        if (!upstream_silent) {
            if (sd.channels == 1) job.flags |= VB_MONO;
            job.g.g[0][0] = job.g.g[0][1] = ss.s0.status == SM_ACTIVE ? ss.s0.last : ss.s0.input;
        }
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j) {
            if (j >= vd.n_stages) break;
            const StageRegs& r = st[j];
            job.g.g[j + 1][0] = r.s0.status == SM_ACTIVE ? r.s0.last : r.s0.input;
            job.g.g[j + 1][1] = vd.stage_kind[j] == K_PAN ? (r.s1.status == SM_ACTIVE ? r.s1.last : r.s1.input)
                                                          : job.g.g[j + 1][0];
        }
        became_steady = true;
        if (w0) {
            VoiceCache vc;
            vc.epoch = fv.epoch;
            vc.mode = mode;
            vc.flags = job.flags;
            vc.sample = job.sample;
            vc.g = job.g;
            fv.cache[vi] = vc;
        }
        if (k + 1 < K) {
            uint32_t tail_gs = 0;
            bool simple_ok = false;
            const bool tail_simple = fx ? (simple_frames && (upstream_silent || (sd.format == FMT_P_F32 && sd.frames < 0xffffffffull)))
                                        : (!sil && !upstream_silent && sd.format == FMT_P_F32 && simple_frames &&
                                           sd.frames < 0xffffffffull);
            if (tail_simple) {
                VoiceBlk probe;  // every non-wrapping tail block is VB_SIMPLE with the same gains: one gain set
                probe.flags = VB_SIMPLE;
#pragma unroll
                for (int j = 0; j < FW_MAX_STAGES; ++j) {
                    probe.g[j][0] = job.g.g[j][0];
                    probe.g[j][1] = job.g.g[j][1];
                }
                tail_gs = pick_gset(probe);
                simple_ok = (probe.flags & VB_SIMPLE) != 0;  // false when the voice ran out of gain-set slots
            }
            uint64_t ph = steady_tail(fv, vi, lane, k + 1, K, job, sd, tail_gs, simple_ok, fx);
            if (mode != 0) ss.playhead = ph;
        }
        break;
    }
    if (!w0) return;
    if (!became_steady) fv.cache[vi].epoch = 0;
    fv.states[vd.sampler_state] = ss;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
        if (j < vd.n_stages) *(StageRegs*)&fv.states[vd.stage_state[j]] = st[j];
}

// Leaf kernel: one wave per (leaf SumNode, block).  For each port in order: fetch the voice's source frames,
// run its gain stages in registers, and accumulate in the reference's summation order (nodes/sum.rs).
// HBM traffic = the source samples once (8 B per stereo voice-sample) + one partial-bus write per leaf.
__device__ __forceinline__ void voice_eval(const FusedView& fv, const VoiceBlk& d, uint32_t k, int voice, int f0, int frames,
                                           v4f& xl, v4f& xr) {
    const bool mono = d.flags & VB_MONO;
    if (d.src_l && f0 + 4 <= frames) {  // planar f32, contiguous: one dwordx4 per channel per lane
        xl = *(const v4f_u*)(d.src_l + f0);
        xr = mono ? xl : *(const v4f_u*)(d.src_r + f0);
    } else {
        const SampleDesc sd = fv.samples[d.sample];
        Fetch ft;
        ft.off0 = d.off0;
        ft.off1 = d.off1;
        ft.n1 = d.n1;
        ft.wrap = (d.flags & VB_WRAP) ? 1 : 0;
        ft.tail_zero = (d.flags & VB_TAIL_ZERO) ? 1 : 0;
        xl = sample_fetch4(sd, 0, ft, (uint32_t)f0, (uint32_t)frames);
        xr = mono ? xl : sample_fetch4(sd, 1, ft, (uint32_t)f0, (uint32_t)frames);
    }
    const uint32_t rbits = d.flags >> VB_RAMP_SHIFT;
    if (rbits == 0) {  // constant gains: sampler.rs:530-533 then volume.rs:123-126 / pan, one rounding each
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) {
            if (j >= fv.n_gain_stages) break;
            xl = xl * d.g[j][0];
            xr = xr * d.g[j][1];
        }
        // a mono sample is duplicated AFTER the sampler gain (sampler.rs:546-551); identical values either way
    } else {
        const float* rb = fv.ramps + ((size_t)k * fv.n_voices + voice) * (size_t)fv.ramp_slots * (size_t)fv.stride + f0;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) {
            if (j >= fv.n_gain_stages) break;
            v4f gl = (rbits >> (2 * j)) & 1u ? *(const v4f*)(rb + (size_t)(2 * j) * fv.stride) : splat(d.g[j][0]);
            v4f gr = (rbits >> (2 * j + 1)) & 1u ? *(const v4f*)(rb + (size_t)(2 * j + 1) * fv.stride) : splat(d.g[j][1]);
            xl = xl * gl;
            xr = xr * gr;
        }
    }
}

__device__ __forceinline__ const float* readlane_ptr(const float* p, int lane) {
    uint64_t u = (uint64_t)p;
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
    return (const float*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float readlane_f(float x, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}

#ifndef LEAF_U
#define LEAF_U 4  // voices whose source loads are in flight together (2*LEAF_U dwordx4 per lane)
#endif
#ifndef LEAF_NT
#define LEAF_NT 1  // non-temporal source loads: every source byte is read exactly once (+12 % measured)
#endif
#ifndef LEAF_WPB
#define LEAF_WPB 4  // waves (leaf, block work items) per workgroup
#endif
#ifndef LEAF_MAP_BLOCKS
#define LEAF_MAP_BLOCKS 1
#endif
// the pointers come out of v_readlane as integers: tell the compiler they are GLOBAL (global_load, not flat_load)
typedef const v4f_u __attribute__((address_space(1)))* gv4p;
__device__ __forceinline__ v4f gload4(const float* p) {
#if LEAF_NT
    return __builtin_nontemporal_load((gv4p)(uint64_t)p);
#else
    return *(gv4p)(uint64_t)p;
#endif
}

// lane p holds port p's VoiceRef + GainSet; every port is VB_SIMPLE (contiguous planar f32, constant gains)
template <int NG>
__device__ __forceinline__ void leaf_fast(const float* my_l, const float* my_r, const GainSet& my_g, int ports, int f0,
                                          v4f& accl, v4f& accr) {
    for (int p0 = 0; p0 < ports; p0 += LEAF_U) {
        v4f xl[LEAF_U], xr[LEAF_U];
#pragma unroll
        for (int u = 0; u < LEAF_U; ++u) {
            if (p0 + u < ports) {
                xl[u] = gload4(readlane_ptr(my_l, p0 + u) + f0);
                xr[u] = gload4(readlane_ptr(my_r, p0 + u) + f0);
            }
        }
#pragma unroll
        for (int u = 0; u < LEAF_U; ++u) {
            if (p0 + u < ports) {
                v4f a = xl[u], b = xr[u];
#pragma unroll
                for (int j = 0; j < NG; ++j) {  // sampler.rs:530-533, volume.rs:123-126, pan: one rounding each
                    a = a * readlane_f(my_g.g[j][0], p0 + u);
                    b = b * readlane_f(my_g.g[j][1], p0 + u);
                }
                if (p0 + u == 0) {
                    accl = a;
                    accr = b;
                } else {
                    accl = accl + a;
                    accr = accr + b;
                }
            }
        }
    }
}

__global__ __launch_bounds__(WAVE* LEAF_WPB) void k_leaf_sum(FusedView fv, int K) {
#if LEAF_MAP_BLOCKS
    // the waves of a workgroup take CONSECUTIVE blocks of one leaf: a steady voice's source is contiguous across
    // blocks, so the workgroup streams LEAF_WPB KiB per voice-channel instead of 1 KiB from LEAF_WPB x 32 places
    const int leaf = blockIdx.x;
    const uint32_t k = blockIdx.y * LEAF_WPB + (threadIdx.x >> 6);
    if (k >= (uint32_t)K) return;
#else
    const int leaf = blockIdx.x * LEAF_WPB + (threadIdx.x >> 6);
    if (leaf >= fv.n_leaves) return;
    const uint32_t k = blockIdx.y;
#endif
    const int lane = threadIdx.x & (WAVE - 1);
    const LeafDesc ld = fv.leaves[leaf];
    const int frames = fv.frames;
    const size_t row = (size_t)k * fv.n_voices + ld.first_voice;
    float* bus = fv.bus + (size_t)k * fv.bus_blk_stride;
    uint8_t* bflags = fv.bus_flags + (size_t)k * fv.bus_flags_blk_stride;
    float* outl = bus + (size_t)ld.out_buf * fv.stride;
    float* outr = outl + fv.stride;

    // lane p loads the compact record of port p (ports <= 32); in_silence_mask: both channels share one flag
    VoiceRef ref;
    ref.src_l = nullptr;
    ref.r_delta = 0;
    ref.flags_gset = VB_SILENT;
    if (lane < ld.ports) ref = fv.refs[(size_t)(ld.first_voice + lane) * fv.refs_stride + k];
    const uint32_t my_flags = ref.flags_gset & 0xffu;
    GainSet my_g;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES; ++j) my_g.g[j][0] = my_g.g[j][1] = 1.0f;
    if (my_flags & VB_SIMPLE) my_g = fv.gsets[(size_t)(ld.first_voice + lane) * FW_GSETS + (ref.flags_gset >> 8)];
    const float* my_l = ref.src_l;
    const float* my_r = ref.src_l + ref.r_delta;
    const uint64_t lanes_in = mask_all_silent_bits(ld.ports);
    const uint64_t silent_ports = __ballot((my_flags & VB_SILENT) != 0) & lanes_in;
    const uint64_t simple_ports = __ballot((my_flags & VB_SIMPLE) != 0) & lanes_in;
    const bool all_silent = silent_ports == lanes_in;
    const bool masked = !(ld.ports == 2 || ld.ports == 3 || ld.ports == 4);  // sum.rs:67-133 (Q13)
    const bool fast = simple_ports == lanes_in && (frames & 3) == 0;

    for (int f0 = lane * 4; f0 < frames; f0 += 256) {
        v4f accl = splat(0.f), accr = splat(0.f);
        if (fast) {
            switch (fv.n_gain_stages) {
                case 1: leaf_fast<1>(my_l, my_r, my_g, ld.ports, f0, accl, accr); break;
                case 2: leaf_fast<2>(my_l, my_r, my_g, ld.ports, f0, accl, accr); break;
                case 3: leaf_fast<3>(my_l, my_r, my_g, ld.ports, f0, accl, accr); break;
                default: leaf_fast<4>(my_l, my_r, my_g, ld.ports, f0, accl, accr); break;
            }
        } else if (!all_silent) {
            for (int p = 0; p < ld.ports; ++p) {
                const bool psil = (silent_ports >> p) & 1ull;
                v4f xl = splat(0.f), xr = splat(0.f);  // a silent chain's buffers hold cleared zeros
                if (!psil) {
                    if ((simple_ports >> p) & 1ull) {  // VB_SIMPLE implies frames % 4 == 0
                        xl = gload4(readlane_ptr(my_l, p) + f0);
                        xr = gload4(readlane_ptr(my_r, p) + f0);
#pragma unroll
                        for (int j = 0; j < FW_MAX_STAGES; ++j) {
                            if (j >= fv.n_gain_stages) break;
                            xl = xl * readlane_f(my_g.g[j][0], p);
                            xr = xr * readlane_f(my_g.g[j][1], p);
                        }
                    } else {
                        const VoiceBlk d = fv.blks[row + p];
                        voice_eval(fv, d, k, ld.first_voice + p, f0, frames, xl, xr);
                    }
                }
                if (p == 0) {  // sum.rs:117 copy_from_slice(port 0) — also when silent; 2/3/4-port: in1
                    accl = xl;
                    accr = xr;
                } else if (!(masked && psil)) {  // :122-124 skip silent ports (n-port path only)
                    accl = accl + xl;
                    accr = accr + xr;
                }
            }
        }
        *(v4f*)(outl + f0) = accl;  // all_silent: clear_all_outputs (sum.rs:52-56)
        *(v4f*)(outr + f0) = accr;
    }
    // out mask: all-silent -> both flagged; 1-port copy -> passthrough (sum.rs:58-65); else 0
    if (lane < 2) bflags[ld.out_buf + lane] = all_silent ? 1 : 0;
}

// ------------------------------------------------------------------ fused chain plan (config 3): k_chain
// Voices of the shape  sampler -> [biquad] -> [delay] -> [volume|pan]* -> leaf SumNode.  The biquad (SPEC: DF1 in
// f32, unfused feed-forward half + two fused feedback taps) is a serial recurrence in time, so time cannot be split across workgroups; what is
// parallel is the voices — and the two channels, which never meet before the mix bus.  One workgroup owns one
// (leaf SumNode of <= 32 voices, channel) for all K blocks of the call and walks time in tiles of TT = 64*NQ frames
// through a 4-stage software pipeline over LDS (one barrier per step):
//   S1  (8 worker waves, lane = (voice, 4*NQ frames)): source fetch + sampler gain; the non-recursive half of the
//       biquad  A[n] = ((b0*x[n]) + (b1*x[n-1])) + (b2*x[n-2])  -> LDS, one row per (voice, channel)
//   S2  (1 wave, lane = voice, raised priority): y[n] = fma(-a1, y[n-1], fma(-a2, y[n-2], A[n])), in place
//   S3a (the same worker lanes, two tiles later): delay-line read-modify-write in HBM, dry/wet mix, gain stages
//   S3b (1 wave): the leaf SumNode in the reference's port order (nodes/sum.rs:67-133) -> partial mix bus
// Every rounding is the one the oracle performs (products and sums separately, same order), so the result is
// bit-identical to the generic executor.  HBM traffic per stereo voice-sample: 8 B source + 8 B ring read + 8 B
// ring write = the 24 B of SURVEY §8d.
//
// Latency hiding: the HBM loads a step consumes were issued during the previous step, right after their registers
// were last used (ring slots of tile s-1 after S3a of tile s-2, source of tile s+1 after S1 of tile s), and stay in
// flight across the barrier — a workgroup-scope barrier on gfx950 does not drain vmcnt, and one CU's L1 handles its
// waves' accesses in issue order, which is also why a ring slot stored in step s is visible to the loads another
// wave issues in step s+1.  Ring loads are prefetched only when the delay is >= 2 tiles (the slots they read were
// stored at least one barrier earlier); shorter delays load in-step.
#define CH_NBUF 4
#ifndef CH_RING_NT
#define CH_RING_NT 0  // non-temporal delay-ring accesses: measured 1.75x SLOWER (624 vs 357 us on config 3)
#endif
#ifndef CH_SRC_NT
#define CH_SRC_NT 1   // non-temporal source loads (every source byte is read once)
#endif
#ifndef CH_WORKERS
#define CH_WORKERS 8
#endif
#ifdef FW_CHAIN_TRACE  // profiling builds: role timelines of workgroup 0 (scripts/chain_trace.py)
#define CH_TRACE(slot)                                                                          \
    do {                                                                                        \
        if (fv.trace && blockIdx.x == 0 && lane == 0 && s < 64) {                               \
            fv.trace[(s * 16 + wave) * 8 + (slot)] = clock64();                                 \
            if ((slot) == 0) fv.trace[(s * 16 + wave) * 8 + 7] = __builtin_amdgcn_s_getreg(63492); /* HW_ID */ \
        }                                                                                       \
    } while (0)
#else
#define CH_TRACE(slot) \
    do {               \
    } while (0)
#endif
#define CH_THREADS ((CH_WORKERS + 4) * WAVE)  // 8 workers + serial + mixer + 2 idle waves (see the role map in k_chain)
typedef float v2f __attribute__((ext_vector_type(2)));

// rotate right by one lane inside each 16-lane DPP row (lane 0 of a row receives lane 15's value)
__device__ __forceinline__ float row_ror1(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
}

struct ChainInfo {  // what a worker lane carries from S1 of a tile to S3a of the same tile (two steps later)
    uint32_t flags;                // VB_* of the tile's block, ramp bits included
    float g[FW_MAX_STAGES - 1];    // this channel's constant post-gain stages (1..)
};

template <int NQ>
__global__ __launch_bounds__(CH_THREADS) void k_chain(FusedView fv, int K, uint32_t cmd_block0) {
    constexpr int TT = 64 * NQ;        // frames per tile
    constexpr int PITCH = TT + 4;      // floats per voice row: + 4 -> the 32 S2 lanes' b128 reads are conflict-free
    constexpr int LF = 4 * NQ;         // frames per worker lane
    __shared__ float tile[CH_NBUF][32][PITCH];  // row = voice (this workgroup's channel)
    __shared__ uint32_t silf[CH_NBUF][32];      // chain output cleared + flagged silent (VB_SILENT) per voice
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const LeafDesc ld = fv.leaves[blockIdx.x];
    const int ch = blockIdx.y;  // L and R never meet before the mix bus: one workgroup per (leaf, channel)
    const int ports = ld.ports;
    const int frames = fv.frames;
    const int tpb = frames / TT;  // the plan guarantees frames % TT == 0
    const int n_tiles = K * tpb;
    // A workgroup's waves are dealt to the 4 SIMDs round-robin, so waves w, w+4, w+8 share a SIMD (measured: HW_ID).
    // The serial wave (2) gets a SIMD to itself — waves 6 and 10 only take part in the barriers — the mixer (11)
    // shares one with two workers, the other six workers fill the remaining two SIMDs.
    const bool is_serial = wave == 2;
    const bool is_mixer = wave == 11;
    const bool is_idle = wave == 6 || wave == 10;
    const bool is_worker = !is_serial && !is_mixer && !is_idle;
    const int widx = wave - (wave > 2 ? 1 : 0) - (wave > 6 ? 1 : 0);  // 0..7 among the worker waves 0,1,3,4,5,7,8,9

    // ---- per-role persistent registers; worker lane = (voice v, frames [LF*q, LF*q + LF) of every tile),
    //      serial lane = voice v.  Everything both channels share (delay position / feedback / mix, biquad
    //      coefficients) is read from the ChainStart record k_voice_control wrote for this call and never written
    //      here: the two workgroups of a leaf are not ordered against each other.
    const int wl = widx * WAVE + lane;
    const int v = is_worker ? (wl >> 4) : lane;
    const int q = wl & 15;
    const bool active = v < ports && (is_worker || (is_serial && lane < 32));
    const int voice = ld.first_voice + (active ? v : 0);
    const VoiceDesc vd = fv.voices[voice];
    const bool has_bq = active && vd.bq_state >= 0, has_dl = active && vd.dl_state >= 0;
    const ChainStart cs = fv.chain_start[voice];
    float b0 = cs.co[0], b1 = cs.co[1], b2 = cs.co[2], a1 = cs.co[3], a2 = cs.co[4];
    float* bq_st = nullptr;  // this channel's [x1 x2 y1 y2]
    float y1 = 0.f, y2 = 0.f;
    if (has_bq) {  // ext = [b0 b1 b2 a1 a2][x1 x2 y1 y2] x 2 channels
        bq_st = fv.ext + fv.states[vd.bq_state].ext_off + 5 + 4 * ch;
        if (is_serial) {
            y1 = bq_st[2];
            y2 = bq_st[3];
        }
    }
    uint32_t D = 1, pos = cs.pos;
    float fb = cs.fb, mix = cs.mix, dry = cs.dry;
    float* ring = nullptr;
    if (has_dl && is_worker) {
        const NodeState* ds = &fv.states[vd.dl_state];
        D = (uint32_t)ds->loop_end;
        ring = fv.ext + ds->ext_off + (size_t)ch * D;
    }
    const bool ring_pref = has_dl && D >= 2u * TT && !(fv.dbg & 16);
    const bool any_bq = __syncthreads_or(has_bq ? 1 : 0) != 0;

    // compute-side block registers (block of tile s) and issue-side ones (block of tile s+1, one step ahead)
    float g0 = 1.f;
    ChainInfo inf0, inf1, inf2;  // tiles s, s-1, s-2
    inf0.flags = inf1.flags = inf2.flags = VB_SRC_ZERO | VB_SIMPLE;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES - 1; ++j) inf0.g[j] = inf1.g[j] = inf2.g[j] = 1.f;
    VoiceRef ref_n;        // descriptor of the block that starts two tiles ahead (in flight)
    ref_n.src_l = nullptr;
    ref_n.r_delta = 0;
    ref_n.flags_gset = VB_SRC_ZERO | VB_SIMPLE;
    float gs_n[FW_MAX_STAGES];  // this channel's gains of the issue-side block (in flight)
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES; ++j) gs_n[j] = 1.f;
    const float* nb_src = nullptr;  // issue-side block: this channel's source of frame 0, VB_* flags
    uint32_t nb_flags = VB_SRC_ZERO | VB_SIMPLE;
    v4f xs[NQ];  // source of the tile S1 computes next (prefetched)
    v4f rg[NQ];  // ring slots of the tile S3a consumes next (prefetched when ring_pref)
#pragma unroll
    for (int j = 0; j < NQ; ++j) xs[j] = rg[j] = splat(0.f);
    // newest x quad (post sampler gain) of this lane; the q == 15 lane's copy is the biquad's x[n-1], x[n-2] state
    v4f prev_x = splat(0.f);
    if (has_bq && is_worker && q == 15) {
        prev_x[3] = bq_st[0];
        prev_x[2] = bq_st[1];
    }

    // role-local (block, tile-in-block) counters: S1 computes tile s, S2 s-1, S3a s-2, S3b s-3; loads issue for s+1
    int k1 = 0, t1 = 0, k2 = 0, t2 = 0, k3 = 0, t3 = 0, k4 = 0, t4 = 0, kla = 0, tla = 0;
    const uint64_t port_mask = mask_all_silent_bits(ports);
    const bool masked = !(ports == 2 || ports == 3 || ports == 4);  // sum.rs:67-133 (Q13)

    // issue the HBM loads of tile `la` (= the tile S1 computes in the next step); at a block start first adopt the
    // block's descriptor (in flight since the previous step) and request its gain set
    auto issue_source = [&]() {
        if (tla == 0) {
            nb_flags = ref_n.flags_gset & 0xffu;
            nb_src = ref_n.src_l + (ch ? ref_n.r_delta : 0u);  // r_delta = 0 for a mono sample (sampler.rs:546-551)
            if (nb_flags & VB_SIMPLE) {
                const GainSet* gs = &fv.gsets[(size_t)voice * FW_GSETS + (ref_n.flags_gset >> 8)];
#pragma unroll
                for (int j = 0; j < FW_MAX_STAGES; ++j) gs_n[j] = gs->g[j][ch];
            }
        }
        if ((nb_flags & VB_SIMPLE) && !(nb_flags & VB_SRC_ZERO) && !(fv.dbg & 4)) {
            const float* p = nb_src + tla * TT + LF * q;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
#if CH_SRC_NT
                xs[j] = gload4(p + 4 * j);
#else
                xs[j] = *(gv4p)(uint64_t)(p + 4 * j);
#endif
            }
        }
        if (++tla == tpb) {
            tla = 0;
            ++kla;
        }
        // the tile after that starts a block: request its descriptor now
        if (tla == 0 && kla < K) ref_n = fv.refs[(size_t)voice * fv.refs_stride + kla];
    };
    auto ring_slot = [&](int j) -> uint32_t {
        uint32_t sl = pos + (uint32_t)(LF * q + 4 * j);
        return sl >= D ? sl - D : sl;
    };
    auto load_ring = [&]() {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const uint32_t sl = ring_slot(j);
            if (sl + 4u <= D) {
#if CH_RING_NT
                rg[j] = __builtin_nontemporal_load((gv4p)(uint64_t)(ring + sl));  // every ring line is touched once per lap
#else
                rg[j] = *(const v4f_u*)(ring + sl);
#endif
            } else {  // the quad straddles the end of the ring
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t se = sl + (uint32_t)e;
                    if (se >= D) se -= D;
                    rg[j][e] = ring[se];
                }
            }
        }
    };
    if (is_worker && active) {  // prologue = the issue halves of steps -2 and -1
        ref_n = fv.refs[(size_t)voice * fv.refs_stride + 0];
        issue_source();
    }

    // One loop per role (same number of barriers in each) so that the register allocation of a role does not
    // carry the other roles' loop state.
    if (is_worker) {
        for (int s = 0; s < n_tiles + 3; ++s) {
            const bool do1 = active && s < n_tiles;
            const bool do3 = active && s >= 2 && s - 2 < n_tiles;
            const bool dl_on = has_dl && !(fv.dbg & 8);
            CH_TRACE(0);
            if (do1 && t1 == 0) {  // new block: adopt the issue-side descriptor (its gain set has landed)
                if (nb_flags & VB_SIMPLE) {
                    inf0.flags = nb_flags;
                    g0 = gs_n[0];
#pragma unroll
                    for (int j = 0; j < FW_MAX_STAGES - 1; ++j) inf0.g[j] = gs_n[j + 1];
                } else {
                    const VoiceBlk* d = &fv.blks[(size_t)k1 * fv.n_voices + voice];
                    inf0.flags = d->flags;
#pragma unroll
                    for (int j = 0; j < FW_MAX_STAGES - 1; ++j) inf0.g[j] = d->g[j + 1][ch];
                }
                if (has_bq && fv.n_cmds) {
                    const ChainCoefs co = chain_find_coefs(fv.cmds, fv.n_cmds, vd.bq_state, cmd_block0 + (uint32_t)k1);
                    if (co.found) {
                        b0 = co.b0;
                        b1 = co.b1;
                        b2 = co.b2;
                    }
                }
            }
            CH_TRACE(1);
            // S3a's LDS rows (tile s-2) are requested first and consumed after S1: the round trip hides behind S1's math
            v4f yv[NQ];
            if (do3) {
                const float* yrow = &tile[(s - 2) & (CH_NBUF - 1)][v][LF * q];
#pragma unroll
                for (int j = 0; j < NQ; ++j) yv[j] = *(const v4f*)(yrow + 4 * j);
            }
            // ================= S1 on tile s: sampler gain + the feed-forward half of the biquad -> LDS
            if (do1) {
                v4f x[NQ];
                if (!(inf0.flags & VB_SIMPLE)) {  // ramps, loop wrap, one-shot tail, non-planar-f32 source: full descriptor
                    const VoiceBlk* d = &fv.blks[(size_t)k1 * fv.n_voices + voice];  // read in place (no private copy)
                    const uint32_t dflags = d->flags;
                    const bool mono = dflags & VB_MONO;
                    const float* dsrc = (mono || ch == 0) ? d->src_l : d->src_r;
                    const float g0c = d->g[0][ch];
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const int f0 = t1 * TT + LF * q + 4 * j;
                        x[j] = splat(0.f);
                        if (!(dflags & VB_SRC_ZERO)) {
                            if (d->src_l) {
                                x[j] = *(const v4f_u*)(dsrc + f0);
                            } else {
                                const SampleDesc sd = fv.samples[d->sample];
                                Fetch ft;
                                ft.off0 = d->off0;
                                ft.off1 = d->off1;
                                ft.n1 = d->n1;
                                ft.wrap = (dflags & VB_WRAP) ? 1 : 0;
                                ft.tail_zero = (dflags & VB_TAIL_ZERO) ? 1 : 0;
                                x[j] = sample_fetch4(sd, mono ? 0 : ch, ft, (uint32_t)f0, (uint32_t)frames);
                            }
                            const uint32_t rb0 = dflags >> VB_RAMP_SHIFT;
                            const float* rb = fv.ramps + ((size_t)k1 * fv.n_voices + voice) * (size_t)fv.ramp_slots * (size_t)fv.stride + f0;
                            const v4f gv = (rb0 >> ch) & 1u ? *(const v4f*)(rb + (size_t)ch * fv.stride) : splat(g0c);
                            x[j] = x[j] * gv;  // sampler.rs:530-533
                        }
                    }
                } else if (inf0.flags & VB_SRC_ZERO) {
#pragma unroll
                    for (int j = 0; j < NQ; ++j) x[j] = splat(0.f);
                } else {
#pragma unroll
                    for (int j = 0; j < NQ; ++j) x[j] = xs[j] * g0;
                }
                float* row = &tile[s & (CH_NBUF - 1)][v][LF * q];
                if (has_bq) {
                    // x[n-1], x[n-2] of this lane's first frame: lane q-1's last quad of THIS tile, or for q == 0 lane
                    // 15's last quad of the PREVIOUS tile — one rotate inside the voice's 16-lane DPP row, no LDS
                    const bool q15 = q == 15;
                    float p1 = row_ror1(q15 ? prev_x[3] : x[NQ - 1][3]), p2 = row_ror1(q15 ? prev_x[2] : x[NQ - 1][2]);
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const v4f xc = x[j];
                        const v4f x1v = (v4f){p1, xc[0], xc[1], xc[2]}, x2v = (v4f){p2, p1, xc[0], xc[1]};
                        const v4f a = ((xc * b0) + (x1v * b1)) + (x2v * b2);  // ((b0*x) + (b1*x1)) + (b2*x2)
                        p1 = xc[3];
                        p2 = xc[2];
                        *(v4f*)(row + 4 * j) = a;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NQ; ++j) *(v4f*)(row + 4 * j) = x[j];
                }
                prev_x = x[NQ - 1];
                if (++t1 == tpb) {
                    t1 = 0;
                    ++k1;
                }
            }
            // The ring slots S3a consumes below have been in flight since the end of the previous step.  Touch them
            // HERE, before the next source loads are issued: the compiler then places its (conservative, vmcnt(0))
            // wait for them ahead of those loads instead of draining them right after their issue.
#pragma unroll
            for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(rg[j]));
            // source of tile s+1 (S1 of the next step)
            if (active && s + 1 < n_tiles) issue_source();
            CH_TRACE(2);
            // ================= S3a on tile s-2: delay RMW + gain stages, in place in LDS (its rows were requested above)
            if (do3) {
                if (dl_on) {
                    if (t3 == 0 && fv.n_cmds) {
                        const ChainDelay p = chain_delay_cmds(fv.cmds, fv.n_cmds, vd.dl_state, cmd_block0 + (uint32_t)k3,
                                                              ChainDelay{fb, mix, dry});
                        fb = p.fb;
                        mix = p.mix;
                        dry = p.dry;
                    }
                    if (!ring_pref) load_ring();
                }
                float* row = &tile[(s - 2) & (CH_NBUF - 1)][v][LF * q];
                const uint32_t rbits = inf2.flags >> VB_RAMP_SHIFT;
#ifdef FW_CHAIN_TRACE
#pragma unroll
                for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(yv[j]));
                CH_TRACE(5);
#endif
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    v4f y = yv[j];
                    if (dl_on) {
                        const v4f nv = y + (rg[j] * fb);  // ring[p] = x + (d*fb)
                        const uint32_t sl = ring_slot(j);
                        if (sl + 4u <= D) {
#if CH_RING_NT
                            __builtin_nontemporal_store(nv, (v4f_u __attribute__((address_space(1)))*)(uint64_t)(ring + sl));
#else
                            *(v4f_u*)(ring + sl) = nv;
#endif
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                uint32_t se = sl + (uint32_t)e;
                                if (se >= D) se -= D;
                                ring[se] = nv[e];
                            }
                        }
                        y = (y * dry) + (rg[j] * mix);  // out = (x*dry) + (d*mix)
                    }
                    if (inf2.flags & VB_SILENT) {  // muted gain stage / silent chain: cleared buffer
                        y = splat(0.f);
                    } else if (rbits == 0) {
#pragma unroll
                        for (int g = 0; g < FW_MAX_STAGES - 1; ++g) {
                            if (g + 1 >= fv.n_gain_stages) break;
                            y = y * inf2.g[g];
                        }
                    } else {
                        const int f0 = t3 * TT + LF * q + 4 * j;
                        const float* rb = fv.ramps + ((size_t)k3 * fv.n_voices + voice) * (size_t)fv.ramp_slots * (size_t)fv.stride + f0;
#pragma unroll
                        for (int g = 1; g < FW_MAX_STAGES; ++g) {
                            if (g >= fv.n_gain_stages) break;
                            const v4f gv = (rbits >> (2 * g + ch)) & 1u ? *(const v4f*)(rb + (size_t)(2 * g + ch) * fv.stride) : splat(inf2.g[g - 1]);
                            y = y * gv;
                        }
                    }
                    *(v4f*)(row + 4 * j) = y;
                }
                CH_TRACE(6);
                if (dl_on) {
                    pos += TT;
                    if (pos >= D) pos -= D;
                }
                if (q == 0) silf[(s - 2) & (CH_NBUF - 1)][v] = (inf2.flags & VB_SILENT) ? 1u : 0u;
                if (++t3 == tpb) {
                    t3 = 0;
                    ++k3;
                }
            }
            // ring slots of tile s-1 (S3a of the next step): issue now, after this step's ring stores
            if (ring_pref && dl_on && s >= 1 && s - 1 < n_tiles) load_ring();
            inf2 = inf1;
            inf1 = inf0;
            CH_TRACE(3);
            __syncthreads();
            CH_TRACE(4);
        }
    } else if (is_serial) {
        // the recurrence is the critical path of every step: its wave wins VALU arbitration on its SIMD
        __builtin_amdgcn_s_setprio(3);
        for (int s = 0; s < n_tiles + 3; ++s) {
            CH_TRACE(0);
            // ================= S2 on tile s-1: the recursive half of the biquad, lane = voice
            if (any_bq && s >= 1 && s - 1 < n_tiles && !(fv.dbg & 1)) {
                if (has_bq && t2 == 0 && fv.n_cmds) {
                    const ChainCoefs co = chain_find_coefs(fv.cmds, fv.n_cmds, vd.bq_state, cmd_block0 + (uint32_t)k2);
                    if (co.found) {
                        a1 = co.a1;
                        a2 = co.a2;
                    }
                }
                if (has_bq) {
                    float* row = &tile[(s - 1) & (CH_NBUF - 1)][v][0];
                    v4f cur[4], nxt[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) cur[u] = *(const v4f*)(row + 4 * u);
#pragma unroll 2
                    for (int c = 0; c < TT / 16; ++c) {
                        if (c + 1 < TT / 16) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) nxt[u] = *(const v4f*)(row + 16 * (c + 1) + 4 * u);
                        }
                        // y[n] = fma(-a1, y[n-1], t[n]), t[n] = fma(-a2, y[n-2], A[n]): t[n+1] only needs y[n-1], so it is
                        // issued BEFORE y[n] — the recurrence then advances at one fma latency per frame
                        v4f o[4];
                        float t = __builtin_fmaf(-a2, y2, cur[0][0]);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int nu = e == 3 ? u + 1 : u, ne = (e + 1) & 3;
                                const float tn = nu < 4 ? __builtin_fmaf(-a2, y1, cur[nu & 3][ne]) : 0.f;  // t of the next frame
                                const float y = __builtin_fmaf(-a1, y1, t);
                                __builtin_amdgcn_sched_barrier(0);
                                y2 = y1;
                                y1 = y;
                                t = tn;
                                o[u][e] = y;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) *(v4f*)(row + 16 * c + 4 * u) = o[u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
                    }
                }
                if (++t2 == tpb) {
                    t2 = 0;
                    ++k2;
                }
            }
            CH_TRACE(3);
            __syncthreads();
            CH_TRACE(4);
        }
    } else if (is_idle) {
        for (int s = 0; s < n_tiles + 3; ++s) __syncthreads();
    } else {
        for (int s = 0; s < n_tiles + 3; ++s) {
            CH_TRACE(0);
            // ================= S3b on tile s-3: the leaf SumNode of this channel, lane = frame quad, ports in order
            if (s >= 3 && lane < TT / 4 && !(fv.dbg & 2)) {
                const int buf = (s - 3) & (CH_NBUF - 1);
                // ONE LDS round trip: every port's row (row index clamped, so the reads are unconditional) and the
                // silence flags are requested together; the adds are masked
                const float* col = &tile[buf][0][4 * lane];
                v4f x[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) x[u] = *(const v4f*)(col + (size_t)(u < ports ? u : ports - 1) * PITCH);
                const uint64_t silent_ports = __ballot(lane < ports && silf[buf][lane & 31] != 0) & port_mask;
                const bool all_silent = silent_ports == port_mask;
                const uint64_t skip = masked ? silent_ports : 0ull;  // :122-124 (n-port path only)
                v4f acc = x[0];  // sum.rs:117 copy port 0 (also when silent: a cleared buffer)
#pragma unroll
                for (int u = 1; u < 32; ++u) {
                    const bool use = u < ports && !((skip >> u) & 1ull);
                    const v4f t = acc + x[u];
                    acc = use ? t : acc;
                }
                if (all_silent) acc = splat(0.f);  // sum.rs:52-56
                float* bus = fv.bus + (size_t)k4 * fv.bus_blk_stride + (size_t)(ld.out_buf + ch) * fv.stride + t4 * TT + 4 * lane;
                *(v4f*)bus = acc;
                if (t4 == 0 && lane == 0) fv.bus_flags[(size_t)k4 * fv.bus_flags_blk_stride + ld.out_buf + ch] = all_silent ? 1 : 0;
                if (++t4 == tpb) {
                    t4 = 0;
                    ++k4;
                }
            }
            CH_TRACE(3);
            __syncthreads();
            CH_TRACE(4);
        }
    }

    // ---- write this channel's biquad state back (everything shared was advanced by k_voice_control)
    if (is_worker && has_bq && q == 15) {
        bq_st[0] = prev_x[3];
        bq_st[1] = prev_x[2];
    }
    if (is_serial && has_bq) {
        bq_st[2] = y1;
        bq_st[3] = y2;
    }
}

// ------------------------------------------------------------------ FIR convolution bank on the matrix cores
// SPEC ("fir"): y[n] = sum_k h[k] x[n-k].  Per block the outputs of all rows that share one
// impulse response are ONE dense GEMM:  Y[rows x frames] = Xwin[rows x W] * H[W x frames],  W = T-1+frames,
// Xwin[r][m] = x_r[n0-(T-1)+m] (history then the current block), H[m][i] = h[T-1-(m-i)] for 0 <= m-i <= T-1 else 0
// (Toeplitz, generated on the fly from h).  v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain, so the
// summation order is fully defined: the window is cut in segments of FIR_SEG positions, each segment is one
// fused chain in ascending m starting from +0.0, segment partials are added in segment order.  The oracle
// evaluates exactly that order with fmaf, so GPU == oracle bit for bit; vs an f64 convolution the error is the
// usual ~sqrt(W) * 2^-24 * sum|h x| (H7).
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_ir_convert(const SampleDesc* __restrict__ samples, int sample, int ch, float* __restrict__ dst, uint32_t T) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const SampleDesc sd = samples[sample];
    int c = ch < sd.channels ? ch : 0;  // a mono impulse response serves every channel
    dst[i] = i < sd.frames ? sample_fetch(sd, c, i) : 0.f;
}

// append the blocks' input to each row's mirrored history ring (positions q and q+R hold the same sample);
// blockIdx.y = block of the K-batch (the ring holds T-1 + K*max_block_frames samples: every block's window is there)
__global__ void k_fir_append(DevView v, const FirRow* __restrict__ rows, int n_rows) {
    int r = blockIdx.x;
    if (r >= n_rows) return;
    const FirRow row = rows[r];
    if (row.state < 0) return;  // padding row (tiles are impulse-response-homogeneous)
    const uint32_t kb = blockIdx.y;
    const NodeState* s = &v.states[row.state];
    const uint32_t R = (uint32_t)s->loop_end, p = (uint32_t)s->playhead;
    float* ring = v.ext + s->ext_off + (size_t)row.ch * 2u * R;
    const float* in = v.pool + (size_t)kb * v.pool_blk_stride + (size_t)row.in_buf * v.stride;
    for (int f = threadIdx.x; f < v.frames; f += blockDim.x) {
        uint32_t q = (p + kb * (uint32_t)v.frames + (uint32_t)f) % R;
        float x = in[f];
        ring[q] = x;
        ring[q + R] = x;
    }
}

#define FIR_PITCH (FIR_KC + 1)  // LDS row pitch in floats: 65 -> the 32 rows of a column hit 32 different banks
__global__ __launch_bounds__(256) void k_fir_gemm(DevView v, const FirRow* __restrict__ rows, int n_rows,
                                                  const uint32_t* __restrict__ tile_h_off, uint32_t T,
                                                  float* __restrict__ partials, int n_rows_pad, int n_pad, int col_groups) {
    __shared__ float lds[2 * 32 * FIR_PITCH + 2 * (256 + FIR_KC)];
    float* As = lds;                          // [2][32][FIR_PITCH]
    float* Hw = lds + 2 * 32 * FIR_PITCH;     // [2][256 + FIR_KC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 32;
    const uint32_t seg = blockIdx.y;
    const uint32_t kb = blockIdx.z / (uint32_t)col_groups;               // block of the K-batch
    const int ib = (int)(blockIdx.z % (uint32_t)col_groups) * 256;       // first output frame of this column group
    const int frames = v.frames;
    const uint32_t W = T - 1u + (uint32_t)frames;
    const uint32_t m_begin = seg * FIR_SEG;
    const uint32_t m_end = m_begin + FIR_SEG < W ? m_begin + FIR_SEG : W;
    const float* h = v.ext + tile_h_off[blockIdx.x];  // every row of a tile convolves with the same h

    // loader role: thread t stages 8 consecutive window positions of row (t >> 3)
    constexpr int NA = FIR_KC / 8;  // floats per loader thread: 8 threads cover one row of the chunk
    const int lrow = tid >> 3, lcol = (tid & 7) * NA;
    const float* wptr = nullptr;
    if (row0 + lrow < n_rows && rows[row0 + lrow].state >= 0) {
        const FirRow row = rows[row0 + lrow];
        const NodeState* s = &v.states[row.state];
        const uint32_t R = (uint32_t)s->loop_end, p = (uint32_t)s->playhead;
        const uint32_t e2 = (p + (kb + 1u) * (uint32_t)frames - 1u) % R + R;  // block kb's newest sample, upper mirror
        wptr = v.ext + s->ext_off + (size_t)row.ch * 2u * R + (e2 + 1u - W);
    }
    // Staging loads are unconditional and vectorised (addresses clamped into the ext pool, which carries 256 floats
    // of slack) and only ISSUED here; the selects that zero what lies outside the segment / the impulse response
    // touch the loaded registers — and therefore wait for them — in store_chunk, one MFMA loop later.  (A branch
    // per element would serialise eight HBM round trips per chunk; a select next to the load would expose one.)
    v4f xa[NA / 4];
#pragma unroll
    for (int j = 0; j < NA / 4; ++j) xa[j] = splat(0.f);
    float hraw0 = 0.f, hraw1 = 0.f;
    uint32_t m0_staged = 0;
    const float* wsafe = wptr ? wptr : v.ext;
    // Hw[q] = h[k], k = ib + T-1 - m0 - (KC-1) + q  (0 outside [0, T))
    auto h_index = [&](uint32_t m0, int q) -> long long {
        return (long long)ib + (long long)T - 1 - (long long)m0 - (FIR_KC - 1) + q;
    };
    auto h_clamp = [&](long long k) -> long long { return k < 0 ? 0 : (k >= (long long)T ? (long long)T - 1 : k); };
    auto load_chunk = [&](uint32_t m0) {
        m0_staged = m0;
#pragma unroll
        for (int j = 0; j < NA / 4; ++j) xa[j] = *(const v4f_u*)(wsafe + m0 + (uint32_t)(lcol + 4 * j));
        hraw0 = __builtin_nontemporal_load(h + h_clamp(h_index(m0, tid)));
        if (wave < FIR_KC / 64) hraw1 = __builtin_nontemporal_load(h + h_clamp(h_index(m0, tid + 256)));  // q = 256 .. 256+KC-1
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * 32 * FIR_PITCH + lrow * FIR_PITCH + lcol;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const uint32_t m = m0_staged + (uint32_t)(lcol + j);
            a[j] = (wptr && m < m_end) ? xa[j >> 2][j & 3] : 0.f;
        }
        float* hw = Hw + buf * (256 + FIR_KC);
        const long long k0 = h_index(m0_staged, tid), k1 = h_index(m0_staged, tid + 256);
        hw[tid] = (k0 >= 0 && k0 < (long long)T) ? hraw0 : 0.f;
        if (tid < FIR_KC) hw[tid + 256] = (k1 >= 0 && k1 < (long long)T) ? hraw1 : 0.f;
    };

    v16f acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = 0.f;
    const int ct0 = wave * 2, ct1 = wave * 2 + 1;  // this wave's two 32-column tiles
    const int a_row = lane & 31, k_half = lane >> 5;

    const uint32_t n_chunks = m_end > m_begin ? (m_end - m_begin + FIR_KC - 1) / FIR_KC : 0;
    if (n_chunks) {
        load_chunk(m_begin);
        store_chunk(0);
    }
    __syncthreads();
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) load_chunk(m_begin + (c + 1) * FIR_KC);  // in flight during the MFMAs below
        const float* a = As + buf * 32 * FIR_PITCH + a_row * FIR_PITCH;
        const float* hw = Hw + buf * (256 + FIR_KC) + (FIR_KC - 1) + (lane & 31);
        // operands of step kk+2 are read while the MFMAs of step kk run (the matrix pipe takes 64 cycles each)
        float av_n = a[k_half], b0_n = hw[ct0 * 32 - k_half], b1_n = hw[ct1 * 32 - k_half];
#pragma unroll
        for (int kk = 0; kk < FIR_KC; kk += 2) {  // ascending m: the fmaf chain order of the SPEC
            const float av = av_n, b0 = b0_n, b1 = b1_n;
            if (kk + 2 < FIR_KC) {
                const int k = kk + 2 + k_half;
                av_n = a[k];
                b0_n = hw[ct0 * 32 - k];
                b1_n = hw[ct1 * 32 - k];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads above ahead of the MFMAs below
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
        }
        if (c + 1 < n_chunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // partials[seg][row][col]: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const size_t row_pitch = (size_t)gridDim.z / col_groups * n_pad;  // K * n_pad
    float* P = partials + ((size_t)seg * n_rows_pad + row0) * row_pitch + (size_t)kb * n_pad + ib;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        P[(size_t)rr * row_pitch + ct0 * 32 + (lane & 31)] = acc0[r];
        P[(size_t)rr * row_pitch + ct1 * 32 + (lane & 31)] = acc1[r];
    }
}

// segment partials added in segment order; writes the node outputs, clears their silence flags, advances the ring
__global__ void k_fir_reduce(DevView v, const FirRow* __restrict__ rows, int n_rows, const float* __restrict__ partials,
                             int n_segs, int n_rows_pad, int n_pad) {
    int r = blockIdx.x;
    if (r >= n_rows) return;
    const FirRow row = rows[r];
    if (row.state < 0) return;
    const uint32_t kb = blockIdx.y, K = gridDim.y;
    const size_t row_pitch = (size_t)K * n_pad;
    float* out = v.pool + (size_t)kb * v.pool_blk_stride + (size_t)row.out_buf * v.stride;
    for (int i = threadIdx.x; i < v.frames; i += blockDim.x) {
        float t = partials[(size_t)r * row_pitch + (size_t)kb * n_pad + i];
        for (int sgm = 1; sgm < n_segs; ++sgm) t = t + partials[((size_t)sgm * n_rows_pad + r) * row_pitch + (size_t)kb * n_pad + i];
        out[i] = t;
    }
    if (threadIdx.x == 0) {
        v.flags[(size_t)kb * v.flags_blk_stride + row.out_buf] = 0;
        if (row.ch == 0 && kb == 0) {
            NodeState* s = &v.states[row.state];
            s->playhead = (s->playhead + (uint64_t)K * (uint64_t)v.frames) % s->loop_end;
        }
    }
}

// Upper sum tree of the fused plan, K-batched: SumNode semantics (nodes/sum.rs:41-136) with one THREAD per
// frame (blockIdx = node, block, channel) so that a 1-node level still puts K * n_out * frames/64 waves in flight.
__global__ __launch_bounds__(256) void k_bus_sum(DevView v, const int* __restrict__ level_nodes) {
    const NodeDesc nd = v.nodes[level_nodes[blockIdx.x]];
    const uint32_t blk = blockIdx.y;
    const int c = blockIdx.z;
    const int lane = threadIdx.x & (WAVE - 1);
    float* pool = v.pool + (size_t)blk * v.pool_blk_stride;
    uint8_t* flags = v.flags + (size_t)blk * v.flags_blk_stride;
    const int* in_buf = v.in_buf + nd.in_off;
    const int* out_buf = v.out_buf + nd.out_off;
    const int n_in = nd.n_in, n_out = nd.n_out, ports = nd.aux0;
    const int my_in = lane < n_in ? in_buf[lane] : 0;
    const uint64_t in_mask = __ballot(lane < n_in ? flags[my_in] != 0 : false);
    float* out = pool + (size_t)out_buf[c] * v.stride;
    uint64_t out_mask = 0;
    if (mask_all(in_mask, n_in)) {  // :52-56
        for (int f = threadIdx.x; f < v.frames; f += blockDim.x) out[f] = 0.f;
        out_mask = mask_all_silent_bits(n_out);
    } else if (n_in == n_out) {  // :58-65
        const float* in = pool + (size_t)__builtin_amdgcn_readlane(my_in, c) * v.stride;
        for (int f = threadIdx.x; f < v.frames; f += blockDim.x) out[f] = in[f];
        out_mask = in_mask;
    } else {
        const bool masked = !(ports == 2 || ports == 3 || ports == 4);
        for (int f = threadIdx.x; f < v.frames; f += blockDim.x) {
            float acc = pool[(size_t)__builtin_amdgcn_readlane(my_in, c) * v.stride + f];
            for (int p0 = 1; p0 < ports; p0 += 8) {
                float x[8];
                bool use[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    use[u] = false;
                    if (p0 + u < ports) {
                        int ic = n_out * (p0 + u) + c;
                        use[u] = !(masked && mask_bit(in_mask, ic));  // :122-124
                        x[u] = pool[(size_t)__builtin_amdgcn_readlane(my_in, ic) * v.stride + f];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (use[u]) acc = acc + x[u];
            }
            out[f] = acc;
        }
    }
    if (c == 0 && (int)threadIdx.x < n_out) flags[out_buf[threadIdx.x]] = mask_bit(out_mask, threadIdx.x) ? 1 : 0;
}

// The root SumNode of the fused plans (stereo, its ports are bus buffers) fused with read_graph_outputs +
// interleave_stereo (schedule.rs:255-287, util.rs:123-147): one launch fewer per call and the root's planar result
// never goes to memory.  Same arithmetic as k_bus_sum followed by k_graph_out: all inputs silent -> the sum clears and
// flags both channels -> interleave_stereo zero-fills; n_in == n_out -> copy with mask passthrough; otherwise ports
// added in order (silent ports skipped on the n-port path only) and both flags are clear.
__global__ __launch_bounds__(256) void k_root_out(DevView v, int root_node, float* __restrict__ out) {
    const NodeDesc nd = v.nodes[root_node];
    const uint32_t blk = blockIdx.y;
    const int lane = threadIdx.x & (WAVE - 1);
    const float* pool = v.pool + (size_t)blk * v.pool_blk_stride;
    const uint8_t* flags = v.flags + (size_t)blk * v.flags_blk_stride;
    const int* in_buf = v.in_buf + nd.in_off;
    const int n_in = nd.n_in, ports = nd.aux0;
    const int my_in = lane < n_in ? in_buf[lane] : 0;
    const uint64_t in_mask = __ballot(lane < n_in ? flags[my_in] != 0 : false);
    float* o = out + (size_t)blk * v.frames * 2;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= v.frames) return;
    float2 y = make_float2(0.f, 0.f);
    if (mask_all(in_mask, n_in)) {
        // sum.rs:52-56 then util.rs:129-134
    } else if (n_in == 2) {  // sum.rs:58-65: copy, flags pass through; both silent was handled above
        y.x = pool[(size_t)__builtin_amdgcn_readlane(my_in, 0) * v.stride + f];
        y.y = pool[(size_t)__builtin_amdgcn_readlane(my_in, 1) * v.stride + f];
    } else {
        const bool masked = !(ports == 2 || ports == 3 || ports == 4);
        float accl = pool[(size_t)__builtin_amdgcn_readlane(my_in, 0) * v.stride + f];
        float accr = pool[(size_t)__builtin_amdgcn_readlane(my_in, 1) * v.stride + f];
        for (int p0 = 1; p0 < ports; p0 += 8) {
            float xl[8], xr[8];
            bool ul[8], ur[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ul[u] = ur[u] = false;
                if (p0 + u < ports) {
                    const int il = 2 * (p0 + u), ir = il + 1;
                    ul[u] = !(masked && mask_bit(in_mask, il));  // :122-124
                    ur[u] = !(masked && mask_bit(in_mask, ir));
                    xl[u] = pool[(size_t)__builtin_amdgcn_readlane(my_in, il) * v.stride + f];
                    xr[u] = pool[(size_t)__builtin_amdgcn_readlane(my_in, ir) * v.stride + f];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (ul[u]) accl = accl + xl[u];
                if (ur[u]) accr = accr + xr[u];
            }
        }
        y = make_float2(accl, accr);
    }
    *(float2*)(o + (size_t)f * 2) = y;
}

// ------------------------------------------------------------------ launch wrappers (host side of this TU)
#define HIPCHK(x)                        \
    do {                                 \
        hipError_t e__ = (x);            \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int launch_level(hipStream_t s, const DevView& v, const int* d_level_nodes, int n_nodes, int K, uint32_t cmd_block0) {
    if (n_nodes <= 0) return 0;
    dim3 grid((n_nodes + WPB - 1) / WPB, K);
    hipLaunchKernelGGL(k_level, grid, dim3(WAVE * WPB), 0, s, v, d_level_nodes, n_nodes, cmd_block0);
    return (int)hipGetLastError();
}
int launch_bus_sum(hipStream_t s, const DevView& v, const int* d_level_nodes, int n_nodes, int K, int n_out) {
    if (n_nodes <= 0) return 0;
    dim3 grid(n_nodes, K, n_out);
    hipLaunchKernelGGL(k_bus_sum, grid, dim3(256), 0, s, v, d_level_nodes);
    return (int)hipGetLastError();
}
int launch_root_out(hipStream_t s, const DevView& v, int root_node, float* d_out, int K) {
    if (v.frames <= 0 || K <= 0) return 0;
    hipLaunchKernelGGL(k_root_out, dim3((v.frames + 255) / 256, K), dim3(256), 0, s, v, root_node, d_out);
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
int launch_voice_control(hipStream_t s, const FusedView& fv, int K, uint32_t cmd_block0) {
    if (fv.n_voices <= 0) return 0;
    hipLaunchKernelGGL(k_voice_control, dim3((fv.n_voices + 3) / 4), dim3(256), 0, s, fv, K, cmd_block0);
    return (int)hipGetLastError();
}
int launch_chain(hipStream_t s, const FusedView& fv, int K, uint32_t cmd_block0, int nq) {
    if (fv.n_leaves <= 0 || K <= 0) return 0;
    if (nq == 2) hipLaunchKernelGGL(k_chain<2>, dim3(fv.n_leaves, 2), dim3(CH_THREADS), 0, s, fv, K, cmd_block0);
    else hipLaunchKernelGGL(k_chain<1>, dim3(fv.n_leaves, 2), dim3(CH_THREADS), 0, s, fv, K, cmd_block0);
    return (int)hipGetLastError();
}
int launch_leaf_sum(hipStream_t s, const FusedView& fv, int K) {
    if (fv.n_leaves <= 0) return 0;
#if LEAF_MAP_BLOCKS
    dim3 grid(fv.n_leaves, (K + LEAF_WPB - 1) / LEAF_WPB);
#else
    dim3 grid((fv.n_leaves + LEAF_WPB - 1) / LEAF_WPB, K);
#endif
    hipLaunchKernelGGL(k_leaf_sum, grid, dim3(WAVE * LEAF_WPB), 0, s, fv, K);
    return (int)hipGetLastError();
}

}  // namespace fwgpu
