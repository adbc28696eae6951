// k_common.hip.h — part of the single device translation unit fwgpu_kernels.hip (included inside namespace fwgpu).
// SilenceMask, ParamSmoother, control->audio messages, sampler playhead logic, sample fetch: shared by every plan.
#pragma once

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

// Apply every queued message for (state_idx, block) in order.  cmds are sorted by (state, block, seq).
// nodes/sampler.rs:331-414 (ring drained at the top of process()), volume.rs:92 (atomic load per block).
// ... from index `lo` (the lower bound of (state_idx, block), or anything in front of it that is not this node's);
// returns the index behind the last message applied — the node's cursor for its next block
__device__ inline int apply_cmds_from(NodeState& s, int state_idx, uint32_t block, const Cmd* cmds, int n_cmds, const SampleDesc* samples,
                                      int lo, float* ext = nullptr, bool ext_write = false) {
    int i = lo;
    for (; i < n_cmds; ++i) {
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
    return i;
}
__device__ inline void apply_cmds(NodeState& s, int state_idx, uint32_t block, const Cmd* cmds, int n_cmds,
                                  const SampleDesc* samples, float* ext = nullptr, bool ext_write = false) {
    if (n_cmds == 0) return;
    apply_cmds_from(s, state_idx, block, cmds, n_cmds, samples, chain_cmd_lower_bound(cmds, n_cmds, state_idx, block), ext, ext_write);
}

// The same lower bound by a whole wave (all 64 lanes active, arguments wave-uniform): 64 pivots per round instead of one —
// a binary search is a chain of log2(n) dependent global loads, ~0.4 us apiece, and the control kernel runs six of them per
// voice in every call that carries messages.  Up to 128 messages: one round trip; up to 8 320: two.
__device__ __forceinline__ long long cmd_key(int state_idx, uint32_t block) {
    return (long long)(((unsigned long long)(uint32_t)state_idx << 32) | (unsigned long long)block);  // (state, block), signed order
}
__device__ __forceinline__ long long cmd_key_at(const Cmd* cmds, int i) {
    const unsigned long long raw = *(const unsigned long long*)&cmds[i];  // state (low word), block (high word)
    return (long long)((raw << 32) | (raw >> 32));
}
__device__ __forceinline__ int wave_cmd_lower_bound(const Cmd* cmds, int n_cmds, long long key, int lane) {
    int lo = 0, hi = n_cmds;
    while (hi - lo > 2 * WAVE) {
        const long long span = hi - lo;
        const int idx = lo + (int)((span * (lane + 1)) / (WAVE + 1));  // ascending in the lane, all inside [lo, hi)
        const int c = __popcll(__ballot(cmd_key_at(cmds, idx) < key));  // sorted: exactly the first c pivots compare less
        const int nlo = c > 0 ? lo + (int)((span * c) / (WAVE + 1)) + 1 : lo;
        const int nhi = c < WAVE ? lo + (int)((span * (c + 1)) / (WAVE + 1)) : hi;
        lo = nlo;
        hi = nhi;
    }
    const int i0 = lo + lane, i1 = lo + WAVE + lane;
    const bool l0 = i0 < hi && cmd_key_at(cmds, i0) < key;
    const bool l1 = i1 < hi && cmd_key_at(cmds, i1) < key;
    return lo + __popcll(__ballot(l0)) + __popcll(__ballot(l1));
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
        // Outside the parity domain the reference dies deterministically: an empty / inverted range
        // underflows `end - playhead` (:457), a fill_buffers range past the sample is a slice panic (:465, :478).
        // Nothing may abort across the ABI and nothing may read HBM past the sample (the range comes unclamped from
        // SetLoopRange(RangeSecs) / SetSample), so such a block plays silence and leaves the playhead where it was.
        if (s.loop_start >= s.loop_end) return false;
        {
            const uint64_t ph = s.playhead >= s.loop_end ? s.loop_start : s.playhead;
            const uint64_t l = s.loop_end - ph;
            const uint64_t n1 = l < (uint64_t)frames ? l : (uint64_t)frames;
            if (ph > len || n1 > len - ph) return false;  // (no u64 overflow: RangeSecs saturates at 2^64-1)
            if (n1 < (uint64_t)frames && (s.loop_start > len || (uint64_t)frames - n1 > len - s.loop_start)) return false;
        }
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

