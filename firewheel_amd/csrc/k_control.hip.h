// k_control.hip.h — part of the single device translation unit fwgpu_kernels.hip (included inside namespace fwgpu).
// Fused plans, control half: message lookups and k_voice_control (per-voice per-block state machines, K blocks per launch).
#pragma once

// ------------------------------------------------------------------ message lookups shared by the fused plans
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

// -DFW_CTL_TRACE: lane 0 of a voice whose call contained a ramp continuation prints where its wave's time went (10 ns ticks)
#ifdef FW_CTL_TRACE
#define CTL_T(i) tr[i] = __builtin_amdgcn_s_memrealtime()
#else
#define CTL_T(i)
#endif

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

// The recurrence, repeated by the preprocessor rather than by an assembler .rept: the compiler sizes an inline-asm block
// by its line count, and with the repeat hidden from it the branch relaxation pass placed short branches across blocks
// that did not fit ("branch size exceeds simm16").
#define FW_REP4(x) x x x x
#define FW_REP16(x) FW_REP4(x) FW_REP4(x) FW_REP4(x) FW_REP4(x)
#define FW_REP64(x) FW_REP16(x) FW_REP16(x) FW_REP16(x) FW_REP16(x)
// one frame per EXEC step (lane i keeps y[i] in %0) ...
#define FW_RAMP_STEP1 "v_mul_f32 %1, %0, %4\n" "v_add_f32 %0, %3, %1\n" "s_lshl_b64 exec, exec, 1\n"
// ... and four: lane q keeps frames 4q .. 4q+3 in %0..%3, the chain's carry is %3
#define FW_RAMP_STEP4                                                                                          \
    "v_mul_f32 %4, %3, %7\n" "v_add_f32 %0, %6, %4\n" "v_mul_f32 %4, %0, %7\n" "v_add_f32 %1, %6, %4\n" \
    "v_mul_f32 %4, %1, %7\n" "v_add_f32 %2, %6, %4\n" "v_mul_f32 %4, %2, %7\n" "v_add_f32 %3, %6, %4\n" \
    "s_lshl_b64 exec, exec, 1\n"

// Serial ramp -> global memory; returns false (and writes nothing) when the recurrence is already at its
// f32 fixed point (Q28: an Active smoother can stall above settle_epsilon forever) — the block is constant.
// The smoother recurrence (core/param/smoother.rs:169-175: out[i] = in*a + out[i-1]*b, two roundings) is serial, so
// every lane of the voice's wave runs it redundantly and EXEC shrinks lane by lane as it goes, each lane dropping out
// with the frames it is to store: two dependent VALU ops per frame (the mul and the add, one rounding each, as
// smoother.rs:171-175) and nothing else on the vector unit.  (A compare + select that parks step i's value in lane i
// doubled the length of the chain; a single lane storing element by element made a ramp block cost ~6 us.)  One wave
// issues one instruction per 4 clocks, so what matters is instructions per frame: 256 frames at a time, lane q keeps frames
// 4q .. 4q+3 — 9 instructions per 4 frames (the EXEC shift is amortised over four) and one 16-byte store per lane; the
// rest of a block 64 frames at a time, lane i keeping frame i (3 per frame); the last < 64 frames with a select.
// Two copies in the library, behind calls (ramp_run_call, ramp_blocks): the 64-fold chains are ~3 KiB of straight-line
// code, and the control kernel has ten call sites.
__device__ __forceinline__ float ramp_run(float prev, const float in_a, const float b, int frames, float* dst0, float* dst1, int lane) {
    int i0 = 0;
    // (the whole wave is active here: the voice's control code is wave-uniform)
    const bool vec_ok = ((((uintptr_t)dst0) | ((uintptr_t)dst1)) & 15u) == 0;
    for (; vec_ok && i0 + 4 * WAVE <= frames; i0 += 4 * WAVE) {
        float d0, d1, d2, d3 = prev, t;
        unsigned long long saved_exec;
        asm volatile(
            "s_mov_b64 %5, exec\n"
            FW_REP64(FW_RAMP_STEP4)
            "s_mov_b64 exec, %5\n"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "+v"(d3), "=&v"(t), "=&s"(saved_exec)
            : "v"(in_a), "v"(b)
            : "scc");
        const v4f g = {d0, d1, d2, d3};
        *(v4f*)(dst0 + i0 + 4 * lane) = g;
        if (dst1) *(v4f*)(dst1 + i0 + 4 * lane) = g;
        prev = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d3), WAVE - 1));
    }
    for (; i0 < frames; i0 += WAVE) {
        float mine = 0.f;
        const int n = frames - i0 < WAVE ? frames - i0 : WAVE;
        if (n == WAVE) {
            float y = prev, t;
            unsigned long long saved_exec;
            asm volatile(
                "s_mov_b64 %2, exec\n"
                FW_REP64(FW_RAMP_STEP1)
                "s_mov_b64 exec, %2\n"
                : "+v"(y), "=&v"(t), "=&s"(saved_exec)
                : "v"(in_a), "v"(b)
                : "scc");
            mine = y;
            prev = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), WAVE - 1));
        } else {
            for (int i = 0; i < n; ++i) {
                prev = in_a + (prev * b);
                mine = i == lane ? prev : mine;
            }
        }
        if (lane < n) {
            dst0[i0 + lane] = mine;
            if (dst1) dst1[i0 + lane] = mine;
        }
    }
    return prev;
}
// (a function returns with its stores retired — s_waitcnt vmcnt(0) — which is why a whole glide is one call: ramp_blocks)
__device__ __attribute__((noinline)) float ramp_run_call(float prev, const float in_a, const float b, int frames, float* dst0, float* dst1, int lane) {
    return ramp_run(prev, in_a, b, frames, dst0, dst1, lane);
}
__device__ __forceinline__ bool ramp_emit(GainRun& r, int frames, float* dst0, float* dst1, int lane) {
    const float v0 = r.in_a + (r.prev * r.b);
    if (v0 == r.prev) {  // fixed point: every later value equals prev, bit for bit
        r.c = r.prev;
        r.ramp = 0;
        return false;
    }
    r.prev = ramp_run_call(r.prev, r.in_a, r.b, frames, dst0, dst1, lane);
    return true;
}

// One smoother's whole glide, block kk0 onwards, in a loop that holds nothing else: set_and_process() of every block
// (smoother_begin: settle test on the block's first value — smoother.rs:181, Q1 — then the fixed-point test, then the
// block's ramp) until the smoother settles, stalls or the call ends.  A smoother's recurrence needs only itself, so the
// continuation runs the gliding smoothers one after the other rather than block by block through the unrolled stage
// loops (hundreds of scalar instructions per block, 4 clocks each: ~1.8 us per block around a 1.0 us chain).
// `dst`: the smoother's ramp row in block kk0; `blk_stride`: floats between consecutive blocks' rows; `dual`: the row
// behind it gets the same values (a gain shared by both channels).  Returns {last, status, input, until} as bits —
// until = one past the last block that got a ramp (0: none).
typedef int ctl_v4i __attribute__((ext_vector_type(4)));
__device__ __attribute__((noinline)) ctl_v4i ramp_blocks(int status, float input, float last, const float a, const float b, const float eps,
                                                          const float target, const int frames, const int kk0, const int K, float* dst,
                                                          const size_t blk_stride, const int row_stride, const int dual, const int lane) {
    if (!(input == target)) {  // set(): smoother.rs:134
        input = target;
        status = SM_ACTIVE;
    }
    int until = 0;
    const float in_a = input * a;  // :169
    for (int kk = kk0; kk < K && status == SM_ACTIVE; ++kk, dst += blk_stride) {
        const float y0 = in_a + (last * b);  // :171
        if (fabsf(input - y0) < eps) {      // :181 — this block and every later one: the constant `input`
            last = input;
            status = SM_DEACTIVATING;
            break;
        }
        if (y0 == last) break;  // f32 fixed point above settle_epsilon (Q28): stalled for good, state untouched
        last = ramp_run(last, in_a, b, frames, dst, dual ? dst + row_stride : nullptr, lane);
        until = kk + 1;
    }
    return ctl_v4i{__float_as_int(last), status, __float_as_int(input), until};
}
__device__ __forceinline__ int ramp_glide(Smoother& s, float target, int frames, int kk0, int K, float* dst, size_t blk_stride, int row_stride,
                                          bool dual, int lane) {
    const ctl_v4i r = ramp_blocks(s.status, s.input, s.last, s.a, s.b, s.eps, target, frames, kk0, K, dst, blk_stride, row_stride, dual ? 1 : 0, lane);
    s.last = __int_as_float(r[0]);
    s.status = r[1];
    s.input = __int_as_float(r[2]);
    return r[3];
}

// A smoother whose next set_and_process(target) returns the same constant and leaves its state untouched:
// not Active, or Active but stalled at the f32 fixed point above settle_epsilon (Q28).
__device__ __forceinline__ bool smoother_is_constant(const Smoother& s, float target) {
    if (!(s.input == target)) return false;
    if (s.status != SM_ACTIVE) return true;
    float y0 = (s.input * s.a) + (s.last * s.b);
    return y0 == s.last && !(fabsf(s.input - y0) < s.eps);
}

// Source class of a block that starts at frame `off0` of sample `sd` and is contiguous in it (no wrap, no tail):
// which vector fetch the leaf kernel may use (SF_*), or SF_NONE.  16-bit planar data needs a 4-byte aligned start
// in both channels; interleaved data of more than two channels and anything in a k_chain voice other than planar
// f32 and interleaved stereo 16-bit PCM stay on the per-element path.
__device__ __forceinline__ uint32_t simple_class(const SampleDesc& sd, uint64_t off0, bool fx) {
    if (sd.frames >= 0xffffffffull) return SF_NONE;
    const bool mono = sd.channels == 1;
    switch (sd.format) {
        case FMT_P_F32: return SF_P_F32;
        case FMT_I_F32: return fx ? SF_NONE : (mono ? SF_P_F32 : (sd.channels == 2 ? SF_I_F32 : SF_NONE));
        case FMT_P_I16:
        case FMT_P_U16:
            if (fx || (off0 & 1) || (!mono && (sd.frames & 1))) return SF_NONE;
            return sd.format == FMT_P_I16 ? SF_P_I16 : SF_P_U16;
        case FMT_I_I16:
        case FMT_I_U16:
            // (round 6: k_chain fetches interleaved STEREO 16-bit PCM itself — 4 bytes per frame, one dwordx4 per quad like planar f32, its
            //  channel's half of every word converted in S1; mono and planar 16-bit data stay on the per-element path there)
            if (fx) return (!mono && sd.channels == 2) ? (sd.format == FMT_I_I16 ? SF_I_I16 : SF_I_U16) : SF_NONE;
            if (mono) return (off0 & 1) ? SF_NONE : (sd.format == FMT_I_I16 ? SF_P_I16 : SF_P_U16);
            return sd.channels == 2 ? (sd.format == FMT_I_I16 ? SF_I_I16 : SF_I_U16) : SF_NONE;
        default: return SF_NONE;
    }
}
// src_kind 2 — sampler(0 -> 1) -> MonoToStereoNode: the sampler fills min(outputs, channels) = 1 buffer with channel 0 (sampler.rs:521-543)
// and the adapter copies it to both of its outputs (mono_to_stereo.rs:33-50).  To everything below that IS a 1-channel sample: channel 0
// of a planar sample is a mono sample where it lies; an interleaved sample with more channels gets no compact class (format none of the
// above: frame-by-frame fetch of channel 0 by the render side, which reads the sample table itself) and VB_MONO like any mono sample.
__device__ __forceinline__ void mono_adapt(SampleDesc& sd) {
    if (sd.channels <= 1) return;
    if (!(sd.format == FMT_P_F32 || sd.format == FMT_P_I16 || sd.format == FMT_P_U16)) sd.format = 0x7f;
    sd.channels = 1;
}
// format-only part of the test above (a voice whose sample can never be fetched compactly needs no gain-set slot)
__device__ __forceinline__ bool simple_capable(const SampleDesc& sd, bool fx) {
    return simple_class(sd, 0, fx) != SF_NONE;
}

// source of a block whose frames are contiguous in the sample: direct pointers for planar f32 (voice_eval / k_chain)
// and the VB_SIMPLE verdict (the compact fast path of the leaf kernel)
__device__ __forceinline__ void blk_set_source(VoiceBlk& d, const SampleDesc& sd, int frames, bool fx) {
    d.src_l = nullptr;
    d.src_r = nullptr;
    const bool contiguous = !(d.flags & (VB_WRAP | VB_TAIL_ZERO | VB_SILENT));
    if (contiguous) {
        if (sd.format == FMT_P_F32) {
            d.src_l = (const float*)sd.data + d.off0;
            d.src_r = (d.flags & VB_MONO) ? d.src_l : d.src_l + sd.frames;
        }
        // VB_SIMPLE blocks carry no full descriptor, so they must never need the per-element path (ragged tail)
        if ((d.flags & VB_RAMP_MASK) == 0 && (frames & 3) == 0 && simple_class(sd, d.off0, fx) != SF_NONE) d.flags |= VB_SIMPLE;
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

// first block index (relative to this call, >= 0) that has a message for node `state_idx`; INT_MAX if none
__device__ inline int first_cmd_block(const Cmd* cmds, int n_cmds, int state_idx, uint32_t cmd_block0) {
    if (n_cmds == 0) return 0x7fffffff;
    const int i = chain_cmd_lower_bound(cmds, n_cmds, state_idx, cmd_block0);
    if (i >= n_cmds || cmds[i].state != state_idx) return 0x7fffffff;
    return (int)(cmds[i].block - cmd_block0);
}

// Both at once, by the whole wave (wave_cmd_lower_bound), plus the index of the node's first message of this call
struct CmdSpan {
    int first, last, cursor;
};
__device__ __forceinline__ CmdSpan wave_cmd_span(const Cmd* cmds, int n_cmds, int state_idx, uint32_t cmd_block0, int lane) {
    CmdSpan r;
    r.first = 0x7fffffff;
    r.last = -1;
    const int lo = wave_cmd_lower_bound(cmds, n_cmds, cmd_key(state_idx, cmd_block0), lane);
    const int up = wave_cmd_lower_bound(cmds, n_cmds, cmd_key(state_idx + 1, 0u), lane);
    r.cursor = lo;
    if (lo < up) {  // [lo, up): the node's messages at or after cmd_block0, in block order
        r.first = (int)(cmds[lo].block - cmd_block0);
        r.last = (int)(cmds[up - 1].block - cmd_block0);
    }
    return r;
}

// ... from keys the wave already holds: lane l has the keys of messages l and l + 64 (n_cmds <= 128)
__device__ __forceinline__ uint32_t held_block(long long key0, long long key1, int i) {  // i: wave-uniform
    const int lo0 = (int)(key0 & 0xffffffffll), lo1 = (int)(key1 & 0xffffffffll);
    const int l = __builtin_amdgcn_readfirstlane(i);
    return (uint32_t)(l < WAVE ? __builtin_amdgcn_readlane(lo0, l) : __builtin_amdgcn_readlane(lo1, l - WAVE));
}
__device__ __forceinline__ CmdSpan wave_cmd_span_held(long long key0, long long key1, int n_cmds, int state_idx, uint32_t cmd_block0, int lane) {
    CmdSpan r;
    r.first = 0x7fffffff;
    r.last = -1;
    const long long klo = cmd_key(state_idx, cmd_block0), kup = cmd_key(state_idx + 1, 0u);
    const bool v0 = lane < n_cmds, v1 = lane + WAVE < n_cmds;
    const int lo = __popcll(__ballot(v0 && key0 < klo)) + __popcll(__ballot(v1 && key1 < klo));
    const int up = __popcll(__ballot(v0 && key0 < kup)) + __popcll(__ballot(v1 && key1 < kup));
    r.cursor = lo;
    if (lo < up) {
        r.first = (int)(held_block(key0, key1, lo) - cmd_block0);
        r.last = (int)(held_block(key0, key1, up - 1) - cmd_block0);
    }
    return r;
}

// Everything the steady tail of a call needs: the descriptor all its blocks share and how the playhead moves.
struct TailJob {
    int mode;          // 0 = nothing moves, 1 = looping playhead, 2 = one-shot playhead, 3 = resampling source: playhead = 32.32
                       //     position, loop_start = 32.32 step, loop_end = (sample frames << 32) when it loops, else 0
    uint32_t flags;    // VB_SILENT / VB_MONO of the shared descriptor
    int sample;
    GainSet g;
    uint64_t playhead, loop_start, loop_end;
    // ramp continuation: slot s = 2*stage + channel carries a per-frame ramp (written to `ramps` by the caller) in the blocks
    // before ramp_until[s]; 0 everywhere on a plain steady tail
    int ramp_until[2 * FW_MAX_STAGES];
};
__device__ __forceinline__ uint32_t tail_ramp_bits(const TailJob& job, int k2) {
    uint32_t rb = 0;
#pragma unroll
    for (int sl = 0; sl < 2 * FW_MAX_STAGES; ++sl) rb |= (k2 < job.ramp_until[sl] ? 1u : 0u) << sl;
    return rb;
}

// Writes the compact record (always) and the full descriptor (only when the leaf kernel will need it).
// `fx`: the voice has a biquad / delay (k_chain plan) — its source is needed even when the chain output is
// silent, and every block that is not VB_SIMPLE carries a full descriptor.
__device__ __forceinline__ void put_blk(const FusedView& fv, int vi, int kk, const VoiceBlk& d, uint32_t gset,
                                        const SampleDesc& sd, bool fx) {
    VoiceRef ref;
    ref.src_l = d.src_l;
    ref.r_delta = 0u;
    uint32_t cls = SF_P_F32;
    if ((d.flags & VB_SIMPLE) && !(d.flags & VB_SRC_ZERO)) {
        cls = simple_class(sd, d.off0, fx);
        const bool mono = d.flags & VB_MONO;
        switch (cls) {
            case SF_P_F32:
                ref.src_l = (const float*)sd.data + d.off0;
                ref.r_delta = mono ? 0u : (uint32_t)sd.frames;
                break;
            case SF_P_I16:
            case SF_P_U16:
                ref.src_l = (const float*)((const int16_t*)sd.data + d.off0);
                ref.r_delta = mono ? 0u : (uint32_t)sd.frames;
                break;
            case SF_I_I16:
            case SF_I_U16:
                ref.src_l = (const float*)((const int16_t*)sd.data + 2 * d.off0);
                ref.r_delta = 1u;
                break;
            default:  // SF_I_F32
                ref.src_l = (const float*)sd.data + 2 * d.off0;
                ref.r_delta = 1u;
                break;
        }
    }
    ref.flags_gset = (d.flags & 0xffu) | (gset << 8) | (cls << 16);
    if (!(d.flags & VB_RESAMPLE)) ref.flags_gset |= d.flags & VB_SP_MASK;  // a spatialiser voice's per-ear delays (never a resampler's)
    fv.refs[ref_index(vi, kk, fv.ref_kgroups)] = ref;  // (tiled: the tail lanes store eight full 128-B lines)
    const bool need_full = fx ? !(d.flags & VB_SIMPLE) : !(d.flags & (VB_SIMPLE | VB_SILENT));
    if (need_full) fv.blks[(size_t)kk * fv.n_voices + vi] = d;
}

// a one-shot resampling source keeps playing through `n` more blocks (k_generic.hip.h K_RESAMPLER: it stops after the block
// that carries its position to sample length + RS_TAPS / 2)
__device__ __forceinline__ bool rs_survives(uint64_t pos, uint64_t step, uint64_t frames, uint64_t n, uint64_t len) {
    return ((pos + n * frames * step) >> 32) < len + RS_TAPS / 2;
}

// Steady tail: blocks k_first .. K-1 share one descriptor; only the playhead moves, by +frames with a wrap at
// the loop end (nodes/sampler.rs:445-484) — closed form (base + j*frames) mod L, so the 64 lanes of the
// voice's wave fill 64 blocks at a time.  Returns the playhead the reference holds after block K-1.
__device__ __forceinline__ uint64_t steady_tail(const FusedView& fv, int vi, int lane, int k_first, int K, const TailJob& job,
                                                const SampleDesc& sd, uint32_t gset, bool simple_ok, bool fx, bool fxp, bool rs_lean = true) {
    // rs_lean: this tail may use the voice's resampler template (VB_RS_LEAN) — at most ONE tail per voice and call does (a call with
    // a message in it has two steady stretches, and the second one's step or gains may differ from the first's)
    // fx: this voice has a biquad / delay (silence does not pass it); fxp: the PLAN is the chain plan — k_chain reads
    // either a VB_SIMPLE record (planar f32, or VB_SRC_ZERO) or a full descriptor for EVERY voice of the plan, dry ones too
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
    const bool has_src = !no_src && job.sample >= 0;
    const bool contiguous_f32 = has_src && sd.format == FMT_P_F32;
    const uint64_t n = (uint64_t)(K - k_first);
    // chain plan, nothing to fetch (cleared source, or a dry voice whose output is muted): a cleared-source block
    const uint32_t nosrc_flags = (fxp && !has_src) ? (VB_SRC_ZERO | (simple_ok ? VB_SIMPLE : 0u)) : 0u;
    // The record of a plain block — planar f32 source, constant gains, no wrap inside the block: what almost every block of a
    // steady bank is — written straight from the loop, ~25 instructions: put_blk's general route through a full VoiceBlk is
    // ~400, and a voice's wave (alone on its SIMD: 1 024 voices, 1 024 SIMDs) pays each of them in full, 12 times over for a
    // call of 768 blocks.  Same bits as put_blk would store (VB_SIMPLE record, class SF_P_F32, no full descriptor).
    int ramps_end = 0;  // blocks from here on carry no ramp
#pragma unroll
    for (int sl = 0; sl < 2 * FW_MAX_STAGES; ++sl) ramps_end = job.ramp_until[sl] > ramps_end ? job.ramp_until[sl] : ramps_end;
    const bool lean = !fxp && has_src && simple_ok && contiguous_f32 && sd.frames < 0xffffffffull;
    VoiceRef lean_ref;
    lean_ref.src_l = nullptr;
    lean_ref.r_delta = (job.flags & VB_MONO) ? 0u : (uint32_t)sd.frames;
    lean_ref.flags_gset = ((job.flags | VB_SIMPLE) & 0xffu) | (gset << 8) | ((uint32_t)SF_P_F32 << 16) | (job.flags & VB_SP_MASK);
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
            if (lean && k2 >= ramps_end && left >= fr) {
                lean_ref.src_l = (const float*)sd.data + (job.loop_start + r);
                fv.refs[ref_index(vi, k2, fv.ref_kgroups)] = lean_ref;
                r += step;
                if (r >= L) r -= L;
                continue;
            }
            const uint32_t rb = tail_ramp_bits(job, k2);
            t.flags = job.flags | nosrc_flags | (rb << VB_RAMP_SHIFT);
            t.off0 = job.loop_start + r;
            t.off1 = job.loop_start;
            t.src_l = t.src_r = nullptr;
            if (left < fr && !nosrc_flags) {  // wraps inside the block (nothing is fetched from a cleared source)
                t.n1 = (uint32_t)left;
                t.flags |= VB_WRAP;
            } else {
                t.n1 = frames;
                if (contiguous_f32) {
                    t.src_l = (const float*)sd.data + t.off0;
                    t.src_r = (t.flags & VB_MONO) ? t.src_l : t.src_l + sd.frames;
                }
                if (has_src && simple_ok && rb == 0 && simple_class(sd, t.off0, fxp) != SF_NONE) t.flags |= VB_SIMPLE;
            }
            put_blk(fv, vi, k2, t, gset, sd, fxp);
            r += step;
            if (r >= L) r -= L;
        }
        const uint64_t left = L - r_last;
        return left < fr ? job.loop_start + (fr - left) : job.loop_start + r_last + fr;
    }
    if (job.mode == 2) {
        for (int k2 = k_first + lane; k2 < K; k2 += WAVE) {
            if (lean && k2 >= ramps_end) {
                lean_ref.src_l = (const float*)sd.data + (job.playhead + (uint64_t)(k2 - k_first) * fr);
                fv.refs[ref_index(vi, k2, fv.ref_kgroups)] = lean_ref;
                continue;
            }
            const uint32_t rb = tail_ramp_bits(job, k2);
            t.flags = job.flags | nosrc_flags | (rb << VB_RAMP_SHIFT);
            t.off0 = job.playhead + (uint64_t)(k2 - k_first) * fr;
            t.src_l = t.src_r = nullptr;
            if (contiguous_f32) {
                t.src_l = (const float*)sd.data + t.off0;
                t.src_r = (t.flags & VB_MONO) ? t.src_l : t.src_l + sd.frames;
            }
            if (has_src && simple_ok && rb == 0 && simple_class(sd, t.off0, fxp) != SF_NONE) t.flags |= VB_SIMPLE;
            put_blk(fv, vi, k2, t, gset, sd, fxp);
        }
        return job.playhead + n * fr;
    }
    if (job.mode == 3) {  // resampling source: position of block j = (pos + j * frames * step) mod (len << 32) when it loops
        const uint64_t adv = fr * job.loop_start;
        const uint64_t M = job.loop_end;
        // the lane's first block, then 64 blocks further per round: ONE 64-bit remainder per lane instead of one per block
        uint64_t pos = job.playhead + (uint64_t)lane * adv;
        uint64_t inc = (uint64_t)WAVE * adv;
        if (M) {
            pos %= M;
            inc %= M;
        }
        t.off1 = job.loop_start;
        t.n1 = M ? 1u : 0u;
        t.src_l = (const float*)sd.data;
        t.src_r = nullptr;
        t.pad = (uint32_t)sd.frames;
        // Round 4: blocks without a ramp share everything but the position — ONE template per voice and call (rs_tmpl) and a
        // 16-byte record per block (VB_RS_LEAN: the position rides in src_l) instead of a 96-byte VoiceBlk row per block: those
        // rows were 75 MB per 768-block call of 1 024 voices, written 96 bytes at a time into lines two waves share.
        const uint32_t flags0 = job.flags | ((uint32_t)sd.format << VB_FMT_SHIFT);
        const bool lean_ok = rs_lean && !fxp && !(flags0 & (VB_SIMPLE | VB_SILENT)) && fv.rs_tmpl != nullptr;
        if (lean_ok && lane == 0) {
            t.flags = flags0;
            t.off0 = 0;
            fv.rs_tmpl[vi] = t;
        }
        VoiceRef lref;
        lref.r_delta = 0u;
        lref.flags_gset = (flags0 & 0xffu) | VB_RS_LEAN | (gset << 8) | ((uint32_t)SF_P_F32 << 16);
        for (int k2 = k_first + lane; k2 < K; k2 += WAVE) {
            const uint32_t rb = k2 >= ramps_end ? 0u : tail_ramp_bits(job, k2);
            if (lean_ok && rb == 0) {
                lref.src_l = (const float*)pos;
                fv.refs[ref_index(vi, k2, fv.ref_kgroups)] = lref;
            } else {
                t.flags = flags0 | (rb << VB_RAMP_SHIFT);
                t.off0 = pos;
                put_blk(fv, vi, k2, t, gset, sd, fxp);
            }
            pos += inc;
            if (M && pos >= M) pos -= M;
        }
        uint64_t end = job.playhead + n * adv;
        if (M) end %= M;
        return end;
    }
    // nothing moves (mode 0 <=> the sampler is frozen): with fx the block still runs (zeros in, constant gains)
    if (fxp && simple_ok) t.flags |= VB_SIMPLE;
    t.flags |= nosrc_flags;
    for (int k2 = k_first + lane; k2 < K; k2 += WAVE) put_blk(fv, vi, k2, t, fxp ? gset : 0u, sd, fxp);
    return job.playhead;
}

// Round 4, lazy records (fwgpu_types.h LazyRec): what a voice that ends the call steady leaves behind for the calls after it.
// `job` is the steady tail's job with the playhead as it stands AFTER this call.  The record is lazy-capable when every later
// block's compact record is a pure function of the block index: silent-flagged (a constant record), or a planar-f32 source read
// contiguously whose loop is a whole number of blocks long and entered on a block boundary (so that no block wraps inside itself),
// or a one-shot (until it runs out: that is the horizon).  Everything else — other formats, loops that wrap inside blocks,
// resampler sources — says "not capable" and the next call runs the control kernel as before.  Chain-plan voices (round 6) leave
// records of k_chain's compact classes the same way; what the ChainStart record holds k_chain then takes from node state.
// the playhead steady_tail(…, k_first, K, job, …) will return: the state behind the call's last block
__device__ __forceinline__ uint64_t tail_end_playhead(const TailJob& job, const int k_first, const int K, const uint64_t fr) {
    const uint64_t n = (uint64_t)(K - k_first);
    if (n == 0) return job.playhead;
    if (job.mode == 1) {
        const uint64_t L = job.loop_end - job.loop_start;
        const uint64_t base = job.playhead >= job.loop_end ? 0 : job.playhead - job.loop_start;
        const uint64_t r_last = (base + (n - 1) * fr) % L;
        const uint64_t left = L - r_last;
        return left < fr ? job.loop_start + (fr - left) : job.loop_start + r_last + fr;
    }
    if (job.mode == 2) return job.playhead + n * fr;
    if (job.mode == 3) {  // a resampling source's 32.32 position (step in loop_start, loop modulus in loop_end: steady_tail)
        const uint64_t end = job.playhead + n * (fr * job.loop_start);
        return job.loop_end ? end % job.loop_end : end;
    }
    return job.playhead;
}
// (called BEFORE the tail is written, with the playhead the tail will end on: the record is built and stored at once, and the tail's
//  store loops do not have to keep the job alive for it — the kernel sits at its register limit)
__device__ __forceinline__ void make_lazy(const FusedView& fv, const int vi, const int sampler_state, const TailJob& job, const uint64_t ph_end,
                                          const SampleDesc& sd, const bool fx, const bool fxp, const bool w0) {
    if (fv.lazy == nullptr) return;
    // (fields are stored as they are made, by lane 0: a LazyRec built in registers first cost the kernel its second wave per SIMD)
    LazyRec* const o = fv.lazy + vi;
    const uint64_t fr = (uint64_t)fv.frames;
    int mode = -1;
    unsigned long long horizon = 0ull;
    if (job.mode == 3 && !fxp && fv.lazy_tmpl != nullptr) {
        // Round 6 — a resampler voice (k_leaf_rs): block j's 32.32 position is (pos + j * frames * step) mod (len << 32), everything else
        // the template the record keeps beside it (lazy_tmpl, the rs_tmpl of lazy calls).  Capable when k_leaf_rs renders every block of
        // it (rs_pure_lane's position-independent conditions: a leaf it hands to the work-list kernel would be read from records nobody
        // wrote) and for as long as the position arithmetic stays inside 64 bits / the one-shot inside its sample.
        const uint64_t step = job.loop_start, M = job.loop_end, len = sd.frames;
        const int nfr = fv.frames < 256 ? fv.frames : 256;
        const uint64_t w_max = (((uint64_t)nfr * step + 0xffffffffull) >> 32) + 1 + RS_TAPS;
        const bool plain = !(job.flags & (VB_SIMPLE | VB_SILENT | VB_SRC_ZERO)) && job.sample >= 0 && sd.format == FMT_P_F32 && sd.data != nullptr;
        if (plain && w_max <= 512 && len >= 1 && len < (1ull << 30) - 8192 && (!M || len >= (uint64_t)(512 + RS_TAPS)) && step < (1ull << 33) && fr <= 4096) {
            mode = 3;
            const uint64_t adv = fr * step;
            if (M) {
                horizon = fv.abs_blk_end + (1ull << 17);  // (j * frames * step < 2^62)
            } else {
                // blocks the one-shot still has in it: rs_survives(pos, step, frames, j, len) for every j up to there
                const uint64_t lim = (len + RS_TAPS / 2) << 32;
                const uint64_t left = lim > ph_end ? (lim - 1 - ph_end) / (adv ? adv : 1) : 0;
                horizon = fv.abs_blk_end + (left > (1ull << 17) ? (1ull << 17) : left);
            }
            if (w0) {
                VoiceBlk t;
                t.flags = job.flags | ((uint32_t)sd.format << VB_FMT_SHIFT);
                t.n1 = M ? 1u : 0u;
                t.src_l = (const float*)sd.data;
                t.src_r = nullptr;
                t.off0 = 0;
                t.off1 = step;
                t.sample = job.sample;
                t.pad = (uint32_t)len;
#pragma unroll
                for (int j = 0; j < FW_MAX_STAGES; ++j) {
                    t.g[j][0] = job.g.g[j][0];
                    t.g[j][1] = job.g.g[j][1];
                }
                fv.lazy_tmpl[vi] = t;
                o->base = M;
                o->off0 = ph_end;
                o->loop_start = step;
                o->r_delta = 0u;
                o->flags_gset = (t.flags & 0xffu) | VB_RS_LEAN | ((uint32_t)SF_P_F32 << 16);
                o->q = 1u;
                o->r0b = 0u;
                o->bpf = 4u;
                o->g = job.g;
            }
        }
    } else if (job.mode >= 0 && job.mode <= 2) {
        // (round 6: chain plans too.  `silent` = the block fetches nothing.  A filter voice fetches its source whatever its output's flag
        //  says — the filters run on; what it does not fetch is a cleared source, VB_SRC_ZERO — steady_tail's no_src)
        const bool silent = fxp ? ((job.flags & VB_SRC_ZERO) != 0 || (!fx && (job.flags & VB_SILENT) != 0)) : (job.flags & VB_SILENT) != 0;
        const bool has_src = !(job.flags & VB_SRC_ZERO) && !silent && job.sample >= 0;
        // the compact record every later block gets (steady_tail's lean record for planar f32, put_blk's VB_SIMPLE record for the other
        // source classes): its class depends on the parity of the block's first source frame only, which is the origin's here — loop
        // start / sample start plus whole blocks of a multiple of 4 frames
        const uint64_t origin = job.mode == 1 ? job.loop_start : 0ull;
        const uint32_t cls = has_src ? simple_class(sd, job.mode == 1 ? origin : ph_end, fxp) : (uint32_t)SF_NONE;
        const bool lean = has_src && (fv.frames & 3) == 0 && cls != (uint32_t)SF_NONE && sd.frames < 0xffffffffull;
        // (address of source frame `origin` and bytes per frame, by class: put_blk's pointer arithmetic)
        uint64_t base = 0;
        uint32_t bpf = 4, rdelta = 0;
        if (lean) {
            const bool mono = (job.flags & VB_MONO) != 0;
            switch (cls) {
                case SF_P_F32: base = (uint64_t)((const float*)sd.data + origin); bpf = 4; rdelta = mono ? 0u : (uint32_t)sd.frames; break;
                case SF_P_I16:
                case SF_P_U16: base = (uint64_t)((const int16_t*)sd.data + origin); bpf = 2; rdelta = mono ? 0u : (uint32_t)sd.frames; break;
                case SF_I_I16:
                case SF_I_U16: base = (uint64_t)((const int16_t*)sd.data + 2 * origin); bpf = 4; rdelta = 1u; break;
                default: base = (uint64_t)((const float*)sd.data + 2 * origin); bpf = 8; rdelta = 1u; break;  // SF_I_F32
            }
        }
        bool moves_ok = true;
        horizon = ~0ull;
        if (job.mode == 1) {
            const uint64_t L = job.loop_end - job.loop_start;
            const uint64_t r0 = ph_end >= job.loop_end ? 0 : ph_end - job.loop_start;
            moves_ok = L >= fr && L % fr == 0 && r0 % fr == 0 && L / fr <= 0xffffffffull;
            if (moves_ok && w0) {
                o->q = (uint32_t)(L / fr);
                o->r0b = (uint32_t)(r0 / fr);
                o->loop_start = job.loop_start;
            }
        } else if (job.mode == 2) {
            // whole blocks left inside the sample (the block the one-shot ends in needs the state machines)
            horizon = fv.abs_blk_end + (sd.frames > ph_end ? (sd.frames - ph_end) / fr : 0);
            if (w0) o->off0 = ph_end;
        }
        // silent: put_blk's record of a block that fetches nothing (no source pointer, class P_F32); else steady_tail's lean record.
        // (nothing moves AND something sounds — a constant full descriptor — is not handled here)
        // (resampler plans: k_leaf_rs renders resampler voices and silence; a sounding sampler voice beside them goes through the
        //  work-list kernel, which reads real records — not capable)
        if (moves_ok && (silent || (job.mode != 0 && lean && !fv.has_rs))) {
            mode = job.mode;
            if (w0) {
                // (chain plan: every record is VB_SIMPLE — a source of one of k_chain's classes, or a cleared one — and keeps the chain output's flag)
                const uint32_t fl = fxp ? (job.flags | VB_SIMPLE | (has_src ? 0u : VB_SRC_ZERO)) : (silent ? job.flags : (job.flags | VB_SIMPLE));
                o->flags_gset = (fl & 0xffu) | ((has_src ? cls : (uint32_t)SF_P_F32) << 16) | (job.flags & VB_SP_MASK);
                o->r_delta = silent ? 0u : rdelta;
                o->base = base;
                o->bpf = bpf;
                o->g = job.g;
            }
        }
    }
    if (w0) {
        o->frames = (uint32_t)fv.frames;
        o->sampler_state = sampler_state;
        o->mode = mode;
        if (fv.horizon) atomicMin(fv.horizon, mode < 0 ? 0ull : horizon);
    }
}
// ... and node state brought up to date after `blocks` blocks rendered from the LazyRecs: the playhead is the only thing that moved
// (chain plans, round 6: and the position in the voice's delay line, which k_chain advanced from the state by itself in every lazy call)
__global__ __launch_bounds__(256) void k_lazy_flush(const LazyRec* __restrict__ lazy, NodeState* __restrict__ states, int n_voices, unsigned long long blocks,
                                                    const VoiceDesc* __restrict__ chain_voices) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_voices) return;
    const LazyRec r = lazy[v];
    if (blocks == 0) return;
    if (chain_voices != nullptr && chain_voices[v].dl_state >= 0) {
        NodeState* ds = &states[chain_voices[v].dl_state];
        const uint64_t D = ds->loop_end;
        ds->playhead = (ds->playhead + (blocks % D) * (uint64_t)r.frames) % D;
    }
    if (r.sampler_state < 0 || r.mode <= 0) return;
    if (r.mode == 3) {  // a resampler voice: its 32.32 position (tail_end_playhead's arithmetic)
        const uint64_t end = r.off0 + blocks * ((uint64_t)r.frames * r.loop_start);
        states[r.sampler_state].playhead = r.base ? end % r.base : end;
        return;
    }
    // (tail_end_playhead's value, not only an equivalent one: a last block that ends exactly on the loop end leaves playhead ==
    //  loop_end — rendered like loop_start, sampler.rs:441-452, but node state must not depend on whether a call was lazy: ADVICE r4)
    if (r.mode == 1) states[r.sampler_state].playhead = r.loop_start + ((uint64_t)((r.r0b + blocks - 1) % r.q) + 1) * r.frames;
    else states[r.sampler_state].playhead = r.off0 + blocks * r.frames;
}
// the horizon of the control kernel that has just run -> pinned host memory {horizon, seq}; the device word is re-armed
__global__ void k_lazy_publish(unsigned long long* d_horizon, unsigned long long* pub, unsigned long long seq) {
    const unsigned long long h = __hip_atomic_load(d_horizon, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d_horizon, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pub, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(pub + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One WAVE per voice: the state machines are run by all 64 lanes redundantly (wave-uniform; lane 0 stores),
// the steady tail is split across the lanes.  A voice that ended the previous call steady and has no message
// in this one skips the state machines altogether (VoiceCache): its whole call is a steady tail.
template <bool EARLY_VC = false>
__device__ inline void voice_control_wave(const FusedView& fv, const int vi, const int lane, const int K, const uint32_t cmd_block0) {
#ifdef FW_CTL_TRACE
    unsigned long long tr[12];
    for (int i = 0; i < 12; ++i) tr[i] = 0;
    bool tr_ramped = false;
    CTL_T(0);
#endif
    const bool w0 = lane == 0;
    const VoiceDesc vd = fv.voices[vi];
    // (the steady cache is indexed by the voice too: asked for HERE it shares the descriptor's round trip instead of following it —
    //  a message-free call's wave is three dependent round trips and a few stores, nothing else)
    //  — in the control KERNEL (EARLY_VC): the one-launch realtime kernels gain nothing from it, their steady voices take
    //  voice_control_lane_steady.  (Round 4 first blamed this load for 8 garbage frames per block out of k_rt_persist; the cause
    //  was the leaf bus stores' inline asm, k_leaf.hip.h bus_store_pair, which any change of register allocation could expose.)
    VoiceCache vc;
    if (EARLY_VC) vc = fv.cache[vi];
    const int frames = fv.frames;
    const bool simple_frames = (frames & 3) == 0;
    const bool fx = vd.bq_state >= 0 || vd.dl_state >= 0;  // the voice has a biquad / delay: silence does not pass it
    // (round 6) the stage indices in front of which a filter sits: the first filter, and the ones behind gain stages between filters.  A
    // filter never reports silence (SPEC nodes: out mask 0): the flag ends there.  Stages in front of the FIRST filter see the source's
    // flag (what reaches that filter cleared is VB_SRC_ZERO); a stage BETWEEN two filters that mutes hands the next filter a cleared
    // buffer — its gain goes out as the sentinel -1.0f, which k_chain turns into +0.0 (x * 1e-6 would not be, and x * 0.0f has x's sign)
    const int fb1 = fx ? vd.n_pre : 0x7fff, fb2 = fx && (vd.n_mid & 0xff) ? fb1 + (vd.n_mid & 0xff) : 0x7fff,
              fb3 = fx && ((vd.n_mid >> 8) & 0xff) ? (fb2 == 0x7fff ? fb1 : fb2) + ((vd.n_mid >> 8) & 0xff) : 0x7fff;
    const bool fxp = fv.fx_plan != 0;                        // chain plan: k_chain's descriptor conventions for EVERY voice
    if (vd.sampler_state < 0) {
        // a null voice = an unconnected port of a leaf SumNode: the cleared, silent-flagged buffer of schedule.rs:310-313
        // in every block (chain plan: as a VB_SIMPLE cleared-source record, which is what k_chain reads)
        VoiceRef r;
        r.src_l = nullptr;
        r.r_delta = 0;
        r.flags_gset = VB_SILENT | (fxp ? (VB_SRC_ZERO | VB_SIMPLE) : 0u);
        for (int k = lane; k < K; k += WAVE) fv.refs[ref_index(vi, k, fv.ref_kgroups)] = r;
        if (fxp && lane < FW_GSETS) {
            GainSet one;
#pragma unroll
            for (int j = 0; j < FW_MAX_STAGES; ++j) one.g[j][0] = one.g[j][1] = 1.0f;
            fv.gsets[(size_t)vi * FW_GSETS + lane] = one;
        }
        if (fv.lazy != nullptr && w0) {  // a constant silent record for as long as the plan lives (chain plan: as a cleared-source record)
            LazyRec lr;
            lr.base = lr.off0 = lr.loop_start = 0;
            lr.r_delta = 0;
            lr.flags_gset = VB_SILENT | (fxp ? (VB_SRC_ZERO | VB_SIMPLE) : 0u);
            lr.q = 1;
            lr.r0b = 0;
            lr.frames = (uint32_t)frames;
            lr.mode = 0;
            lr.sampler_state = -1;
            lr.bpf = 4;
            lr.pad[0] = lr.pad[1] = 0;
#pragma unroll
            for (int j = 0; j < FW_MAX_STAGES; ++j) lr.g.g[j][0] = lr.g.g[j][1] = 1.0f;
            lr.pad2[0] = lr.pad2[1] = lr.pad2[2] = lr.pad2[3] = 0;
            fv.lazy[vi] = lr;
        }
        return;
    }

    // a voice that ends in a spatialiser: the 64-frame mono history it enters this call with goes from the node's ext slice into
    // the call's scratch — the render waves of block 0 read the scratch while the waves of the last block write the pool
    const bool spv = vd.sp_ext_off >= 0;
    const int sp_j = vd.n_stages - 1;  // (the spatialiser is the last stage)
    // (its state slot by selects, not by vd.stage_state[sp_j]: a dynamic index sends the descriptor's arrays — and with them the
    //  stage registers loaded through them — to scratch and LDS; a kernel with a private segment pays for it at every dispatch)
    int sp_state = 0;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES - 1; ++j) sp_state = j == sp_j ? vd.stage_state[j] : sp_state;
    int sp_dl = 0, sp_dr = 0;
    if (spv) {
        // (control-ahead mode: this kernel runs beside the render kernel of the call BEFORE, which writes the pool at its end — the
        //  copy is then made on the render stream, right in front of this call's leaf kernel: k_sp_hist_copy)
        if (!fv.sp_hist_in_render) fv.hist[(size_t)vi * SP_HIST + lane] = fv.ext[(size_t)vd.sp_ext_off + lane];
        const NodeState* sn = &fv.states[sp_state];
        sp_dl = sn->playing;
        sp_dr = sn->has_loop;
    }
    auto sp_bits = [&]() -> uint32_t { return spv ? ((uint32_t)(sp_dl & 63) | ((uint32_t)(sp_dr & 63) << 6)) << VB_SP_SHIFT : 0u; };

    // ---- k_chain plan: what both channel workgroups of the voice's leaf share is owned HERE — the record holds the
    // values at the start of this call (k_chain replays the call's messages block by block from them), the node state
    // is advanced to the end of the call.  k_chain itself only reads the record.
    if (fx) {
        ChainStart cs;
        cs.pos = 0;
        cs.fb = 0.f;
        cs.mix = 0.f;
        cs.dry = 1.f;
        cs.co[0] = cs.co2[0] = 1.f;
        cs.co[1] = cs.co[2] = cs.co[3] = cs.co[4] = 0.f;
        cs.co2[1] = cs.co2[2] = cs.co2[3] = cs.co2[4] = 0.f;
        cs.pad[0] = cs.pad[1] = 0;
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
#pragma unroll
        for (int which = 0; which < 2; ++which) {  // the chain's biquad(s): the record holds the coefficients at the call's start, the ext
            const int bqs = which ? vd.bq2_state : vd.bq_state;  // pool the ones at its end (k_chain replays the messages in between)
            if (bqs < 0) continue;
            float* co = fv.ext + fv.states[bqs].ext_off;
#pragma unroll
            for (int j = 0; j < 5; ++j) (which ? cs.co2 : cs.co)[j] = co[j];
            if (fv.n_cmds) {
                bool found = false;
                float nc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
                for (int i = chain_cmd_lower_bound(fv.cmds, fv.n_cmds, bqs, cmd_block0); i < fv.n_cmds; ++i) {
                    const Cmd c = fv.cmds[i];
                    if (c.state != bqs || c.block >= cmd_block0 + (uint32_t)K) break;
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

    // first / last block of this call with a message for any node of the voice, and per node the index of its first
    // message (its cursor: the list is sorted by (node, block, seq), and the block loop below visits the blocks in order)
    int last_cmd = -1, first_cmd = 0x7fffffff;
    int cur_smp = 0, cur_st[FW_MAX_STAGES - 1];
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES - 1; ++j) cur_st[j] = 0;
    if (fv.n_cmds) {
        // up to 128 messages in the call (the usual case): every lane keeps two keys and all the nodes' spans come out of ONE
        // memory round trip; beyond that, a search per node
        const bool few = fv.n_cmds <= 2 * WAVE;
        long long key0 = 0, key1 = 0;
        if (few) {
            if (lane < fv.n_cmds) key0 = cmd_key_at(fv.cmds, lane);
            if (lane + WAVE < fv.n_cmds) key1 = cmd_key_at(fv.cmds, lane + WAVE);
        }
        const CmdSpan sp = few ? wave_cmd_span_held(key0, key1, fv.n_cmds, vd.sampler_state, cmd_block0, lane)
                               : wave_cmd_span(fv.cmds, fv.n_cmds, vd.sampler_state, cmd_block0, lane);
        last_cmd = sp.last;
        first_cmd = sp.first;
        cur_smp = sp.cursor;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
            if (j < vd.n_stages) {
                const CmdSpan sj = few ? wave_cmd_span_held(key0, key1, fv.n_cmds, vd.stage_state[j], cmd_block0, lane)
                                       : wave_cmd_span(fv.cmds, fv.n_cmds, vd.stage_state[j], cmd_block0, lane);
                last_cmd = sj.last > last_cmd ? sj.last : last_cmd;
                first_cmd = sj.first < first_cmd ? sj.first : first_cmd;
                cur_st[j] = sj.cursor;
            }
    }
    CTL_T(1);
    GainSet* my_gsets = fv.gsets + (size_t)vi * FW_GSETS;

    // ---- fast path: still steady from the previous call, up to the voice's first message of this call (a call may
    // span hundreds of blocks: walking them one by one to reach a message at block 400 would take milliseconds)
    int k0 = 0;              // the general path starts here
    uint64_t k0_playhead = 0;
    bool k0_moved = false;   // the sampler's playhead advanced in the steady blocks (a playing voice)
    bool k0_gset = false;    // gain set 0 of this call already holds the steady gains
    GainSet k0_gs;
    {
        if (!EARLY_VC) vc = fv.cache[vi];
        const int Kp = first_cmd < K ? first_cmd : K;  // blocks [0, Kp) are steady
        if (vc.epoch == fv.epoch && Kp > 0) {
            TailJob job;
            job.mode = vc.mode;
            job.flags = vc.flags;
            job.sample = vc.sample;
            job.g = vc.g;
            job.playhead = job.loop_start = job.loop_end = 0;
#pragma unroll
            for (int sl = 0; sl < 2 * FW_MAX_STAGES; ++sl) job.ramp_until[sl] = 0;
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
                if (vd.src_kind == 2) mono_adapt(sd);
                if (vc.mode == 2 && job.playhead + (uint64_t)Kp * (uint64_t)frames > sd.frames) ok = false;  // ends in these blocks
                if (vc.mode == 3) {
                    job.loop_end = sp->has_loop ? (sd.frames << 32) : 0;
                    if (!sp->has_loop && !rs_survives(job.playhead, job.loop_start, (uint64_t)frames, (uint64_t)Kp, sd.frames)) ok = false;
                }
            }
            if (ok) {
                const bool no_src = (job.flags & VB_SRC_ZERO) || (!fx && (job.flags & VB_SILENT));
                const bool simple_ok = vc.mode != 3 && (no_src ? (fxp && simple_frames)
                                                               : (job.sample >= 0 && simple_frames && simple_capable(sd, fxp)));
                if (simple_ok && w0) my_gsets[0] = job.g;
                if (Kp == K) make_lazy(fv, vi, vd.sampler_state, job, tail_end_playhead(job, 0, K, (uint64_t)frames), sd, fx, fxp, w0);
                uint64_t ph = steady_tail(fv, vi, lane, 0, Kp, job, sd, 0u, simple_ok, fx, fxp);
                if (Kp == K) {
                    if (w0 && vc.mode != 0) fv.states[vd.sampler_state].playhead = ph;
                    return;
                }
                k0 = Kp;  // nothing but the playhead moved in the steady blocks: the general path takes over at Kp
                k0_playhead = ph;
                k0_moved = vc.mode != 0;
                k0_gset = simple_ok;
                k0_gs = job.g;
            }
        }
    }

    CTL_T(2);
    // ---- general path
    // (asking for the state records before the fast path — behind its descriptor stores these loads wait for every one of
    // them, 1 to 5 us: loads and stores retire through the same in-order counter on gfx9 — made every voice slower: measured)
    NodeState ss = fv.states[vd.sampler_state];
    if (k0_moved) ss.playhead = k0_playhead;
    StageRegs st[FW_MAX_STAGES - 1];
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
        if (j < vd.n_stages) st[j] = *(const StageRegs*)&fv.states[vd.stage_state[j]];

    // gain sets used so far in this call (the current one is mirrored in registers)
    int n_gsets = 0;
    GainSet cur_gs;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES; ++j) cur_gs.g[j][0] = cur_gs.g[j][1] = 0.f;
    if (k0_gset) {
        n_gsets = 1;
        cur_gs = k0_gs;
    }
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
    if (ss.sample >= 0 && ss.playing) {  // the sample in use at the start of the call (messages may replace it below)
        sd = fv.samples[ss.sample];
        if (vd.src_kind == 2) mono_adapt(sd);
        cached_sample = ss.sample;
    }

    CTL_T(8);
    for (int k = k0; k < K; ++k) {
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

        // ---- this block's messages for every node of the voice, up front (the nodes' parameters are independent of one
        // another, so applying the stage nodes' messages before the sampler runs changes nothing).  Past the voice's last
        // message of the call there is nothing to look up (each lookup is a chain of dependent global loads: ~1 us
        // apiece, every block of a ramp).  ALL vector loads of the block loop live in this branch and it ends with an
        // explicit vmcnt(0): the rest of the body only stores, so the compiler has no pending load to guard and places no
        // vmcnt wait in the ramp code (with the lookups interleaved, every 64-frame ramp chunk store first drained the
        // previous one: loads and stores share the in-order vmcnt counter on gfx9).
        if (k <= last_cmd) {
            cur_smp = apply_cmds_from(ss, vd.sampler_state, cb, fv.cmds, fv.n_cmds, fv.samples, cur_smp);
#pragma unroll
            for (int j = 0; j < FW_MAX_STAGES - 1; ++j) {
                if (j < vd.n_stages) {
                    NodeState tmp;  // only p0/p1 apply to gain stages (+ a spatialiser's per-ear delays: CMD_SP_ITD)
                    tmp.p0 = st[j].p0;
                    tmp.p1 = st[j].p1;
                    tmp.playing = sp_dl;
                    tmp.has_loop = sp_dr;
                    cur_st[j] = apply_cmds_from(tmp, vd.stage_state[j], cb, fv.cmds, fv.n_cmds, fv.samples, cur_st[j]);
                    st[j].p0 = tmp.p0;
                    st[j].p1 = tmp.p1;
                    if (spv && j == sp_j) {
                        sp_dl = tmp.playing;
                        sp_dr = tmp.has_loop;
                    }
                }
            }
            if (ss.sample >= 0 && ss.playing && cached_sample != ss.sample) {
                sd = fv.samples[ss.sample];
                if (vd.src_kind == 2) mono_adapt(sd);
                cached_sample = ss.sample;
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
        }
        CTL_T(9);
        // ---- sampler (nodes/sampler.rs:323-561)
        // a sample destroyed under the sampler (fwgpu_sample_destroy: table entry with data == nullptr) counts as "no
        // sample": outputs cleared, nothing moves (sampler.rs:416-430) — never a fetch through the stale loop range
        bool silent = true;
        if (vd.src_kind == 1) {
            // SPEC resampling source (k_generic.hip.h K_RESAMPLER): position / step / loop flag, no gain of its own — stage
            // 0's gain stays 1.0, an exact multiply
            if (ss.playing && ss.sample >= 0 && sd.data != nullptr && sd.frames != 0) {
                silent = false;
                d.flags |= VB_RESAMPLE | ((uint32_t)sd.format << VB_FMT_SHIFT);
                d.sample = ss.sample;
                d.off0 = ss.playhead;
                d.off1 = ss.loop_start;
                d.n1 = ss.has_loop ? 1u : 0u;
                d.src_l = (const float*)sd.data;
                d.pad = (uint32_t)sd.frames;
                if (sd.channels == 1) d.flags |= VB_MONO;
                uint64_t np = ss.playhead + (uint64_t)frames * ss.loop_start;
                if (ss.has_loop) np %= (sd.frames << 32);
                else if ((np >> 32) >= sd.frames + RS_TAPS / 2) ss.playing = 0;
                ss.playhead = np;
            }
        } else if (ss.sample >= 0 && ss.playing && sd.data != nullptr) {
            GainRun run = smoother_begin(ss.s0, ss.p0, frames);
            if (!(!smoother_is_smoothing(ss.s0) && run.c < 0.00001f)) {
                Fetch ft;
                bool ok = sampler_advance(ss, sd.frames, (uint32_t)frames, ft);
                if (run.ramp) {
                    if (ramp_emit(run, frames, ramp_base, ramp_base + fv.stride, lane)) {
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
        CTL_T(10);
        // a biquad / delay between the sampler and the gain stages never reports silence (SPEC nodes: out mask 0)
        // (round 6, positional silence: the stages in FRONT of the filters — the first n_pre — see the source's flag and reset like any
        //  bank stage; `pre_silent` = what reaches the first filter: a cleared buffer it runs on all the same; from there on nothing is
        //  flagged until a stage mutes)
        const bool src_silent = silent;
        bool pre_silent = silent;
        bool sp_in_zero = false;  // the spatialiser's input buffers are cleared zeros this block
        // ---- chain stages in schedule order
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j) {
            if (fx && j == fb1) {
                pre_silent = silent;
                silent = false;
            }
            if (j == fb2 || j == fb3) silent = false;  // (another filter: whatever a stage in front of it muted, it runs on and flags nothing)
            if (j >= vd.n_stages) break;
            StageRegs& r = st[j];
            float* rb = ramp_base + (size_t)(j + 1) * 2 * fv.stride;
            const bool between = fx && j >= fb1 && j < (fb3 != 0x7fff ? fb3 : (fb2 != 0x7fff ? fb2 : fb1));  // a stage between two filters
            if (between && silent) d.g[j + 1][0] = d.g[j + 1][1] = -1.0f;  // (behind a muted stage of its segment: cleared in, cleared out)
            if (vd.stage_kind[j] == K_VOLUME) {  // nodes/volume.rs:84-145
                if (silent) {
                    smoother_reset(r.s0, r.p0);
                } else {
                    GainRun run = smoother_begin(r.s0, r.p0, frames);
                    if (!smoother_is_smoothing(r.s0) && run.c < 0.00001f) {
                        silent = true;
                        if (between) d.g[j + 1][0] = d.g[j + 1][1] = -1.0f;
                    } else {
                        if (run.ramp && ramp_emit(run, frames, rb, rb + fv.stride, lane)) {
                            d.flags |= 3u << (VB_RAMP_SHIFT + 2 * (j + 1));
                            r.s0.last = run.prev;
                        }
                        d.g[j + 1][0] = d.g[j + 1][1] = run.c;
                    }
                }
            } else if (vd.stage_kind[j] == K_PAN) {  // SPEC: the stereo path of volume.rs with one smoother per channel
                if (silent) {
                    smoother_reset(r.s0, r.p0);
                    smoother_reset(r.s1, r.p1);
                } else {
                    GainRun rl = smoother_begin(r.s0, r.p0, frames);
                    GainRun rr = smoother_begin(r.s1, r.p1, frames);
                    if (rl.ramp && ramp_emit(rl, frames, rb, nullptr, lane)) {
                        d.flags |= 1u << (VB_RAMP_SHIFT + 2 * (j + 1));
                        r.s0.last = rl.prev;
                    }
                    if (rr.ramp && ramp_emit(rr, frames, rb + fv.stride, nullptr, lane)) {
                        d.flags |= 2u << (VB_RAMP_SHIFT + 2 * (j + 1));
                        r.s1.last = rr.prev;
                    }
                    d.g[j + 1][0] = rl.c;
                    d.g[j + 1][1] = rr.c;
                }
            } else if (vd.stage_kind[j] == K_SPATIAL) {
                // SPEC spatialiser (k_generic.hip.h K_SPATIAL): no silence shortcut — both ear-gain smoothers advance every block,
                // the node always writes (delayed input x gain: the last 63 frames of a voice that stopped still sound), out mask 0
                sp_in_zero = silent;
                GainRun rl = smoother_begin(r.s0, r.p0, frames);
                GainRun rr = smoother_begin(r.s1, r.p1, frames);
                if (rl.ramp && ramp_emit(rl, frames, rb, nullptr, lane)) {
                    d.flags |= 1u << (VB_RAMP_SHIFT + 2 * (j + 1));
                    r.s0.last = rl.prev;
                }
                if (rr.ramp && ramp_emit(rr, frames, rb + fv.stride, nullptr, lane)) {
                    d.flags |= 2u << (VB_RAMP_SHIFT + 2 * (j + 1));
                    r.s1.last = rr.prev;
                }
                d.g[j + 1][0] = rl.c;
                d.g[j + 1][1] = rr.c;
                silent = false;
            } else if (vd.stage_kind[j] == K_WIDTH) {  // SPEC: silent input -> reset + clear; else mid/side with
                if (silent) {                           // the smoothed width, out mask 0 — silence passes through unchanged
                    smoother_reset(r.s0, r.p0);
                } else {
                    GainRun rw = smoother_begin(r.s0, r.p0, frames);
                    if (rw.ramp && ramp_emit(rw, frames, rb, nullptr, lane)) {
                        d.flags |= 1u << (VB_RAMP_SHIFT + 2 * (j + 1));
                        r.s0.last = rw.prev;
                    }
                    d.g[j + 1][0] = d.g[j + 1][1] = rw.c;
                }
            } else {  // K_HARD_CLIP (hard_clip.rs:51-95): no state; a silent (both-channel) input is zero-filled and stays flagged
                if (!(between && silent)) d.g[j + 1][0] = d.g[j + 1][1] = r.p0;
            }
        }
        CTL_T(11);
        if (fx && vd.n_pre >= FW_MAX_STAGES - 1) {  // (every slot a stage in front of the filters: the loop above never reached the boundary)
            pre_silent = silent;
            silent = false;
        }
        // a dry voice whose output is muted fetches nothing; a filter voice fetches unless what reaches its first filter is cleared
        const bool need_src = !src_silent && !sp_in_zero && (fx ? !pre_silent : !silent);
        d.flags |= sp_bits();
        if (need_src && !(d.flags & VB_RESAMPLE)) blk_set_source(d, sd, frames, fxp);
        else if (fxp && (d.flags & VB_RAMP_MASK) == 0 && simple_frames) d.flags |= VB_SIMPLE;  // chain plan: cleared-source block
        if (src_silent || sp_in_zero || (fxp && !need_src)) d.flags |= VB_SRC_ZERO;
        if (fxp && !need_src) d.flags &= ~(VB_WRAP | VB_TAIL_ZERO);  // nothing is fetched: where the source would wrap is moot
        if (silent) d.flags |= VB_SILENT;
        {
            uint32_t gs = pick_gset(d);
            if (w0) put_blk(fv, vi, k, d, gs, sd, fxp);
        }

        CTL_T(3);
        // ---- steady from the next block on?
        if (k < last_cmd) continue;
        bool steady = true;
        bool upstream_silent = false;
        bool ramping = false;  // some smoother is still moving: the rest of the call is a RAMP CONTINUATION, then steady
        int mode = 0;
        if (ss.sample < 0 || !ss.playing || sd.data == nullptr || (vd.src_kind == 1 && sd.frames == 0)) {
            upstream_silent = true;  // frozen sampler: nothing moves
        } else if (vd.src_kind == 1) {  // resampling source: the position has a closed form while it keeps playing
            if (cached_sample != ss.sample) steady = false;
            else if (ss.has_loop || rs_survives(ss.playhead, ss.loop_start, (uint64_t)frames, (uint64_t)(K - 1 - k), sd.frames)) mode = 3;
            else steady = false;  // the one-shot runs out inside this call: block by block
        } else {
            if (!smoother_is_constant(ss.s0, ss.p0)) ramping = true;
            if (!ramping && ss.s0.status == SM_INACTIVE && ss.s0.input < 0.00001f) upstream_silent = true;  // muted, frozen
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
        bool sil = upstream_silent;  // what the stage at hand is handed; a filter voice's flag ends at its first filter (pre_sil keeps it)
        bool pre_sil = false;
        bool mid_sil[FW_MAX_STAGES - 1];  // stage j sits between two filters and hands on a cleared buffer (muted itself, or behind a muted one)
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j) mid_sil[j] = false;
        bool sp_zero = false;  // steady with a spatialiser whose input is cleared zeros (source stopped / muted upstream)
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j) {
            if (fx && j == fb1) {
                pre_sil = sil;
                sil = false;
            }
            if (j == fb2 || j == fb3) sil = false;
            if (j >= vd.n_stages || !steady) break;
            mid_sil[j] = fx && j >= fb1 && j < (fb3 != 0x7fff ? fb3 : (fb2 != 0x7fff ? fb2 : fb1)) && sil;  // (cleared in: cleared out)
            const StageRegs& r = st[j];
            if (vd.stage_kind[j] == K_SPATIAL) {  // (its smoothers run whatever comes in; what goes out is never flagged silent)
                sp_zero = sil || upstream_silent;
                sil = false;
                if (!smoother_is_constant(r.s0, r.p0) || !smoother_is_constant(r.s1, r.p1)) ramping = true;
                continue;
            }
            if (sil) {  // reset() every block: idempotent once applied
                if (vd.stage_kind[j] != K_HARD_CLIP && !(r.s0.status == SM_INACTIVE && r.s0.input == r.p0)) steady = false;
                if (vd.stage_kind[j] == K_PAN && !(r.s1.status == SM_INACTIVE && r.s1.input == r.p1)) steady = false;
            } else if (vd.stage_kind[j] == K_VOLUME) {
                if (!smoother_is_constant(r.s0, r.p0)) ramping = true;
                else if (r.s0.status == SM_INACTIVE && r.s0.input < 0.00001f) {
                    sil = true;
                    mid_sil[j] = fx && j >= fb1 && j < (fb3 != 0x7fff ? fb3 : (fb2 != 0x7fff ? fb2 : fb1));
                }
            } else if (vd.stage_kind[j] == K_PAN) {
                if (!smoother_is_constant(r.s0, r.p0) || !smoother_is_constant(r.s1, r.p1)) ramping = true;
            } else if (vd.stage_kind[j] == K_WIDTH) {
                if (!smoother_is_constant(r.s0, r.p0)) ramping = true;
            }  // K_HARD_CLIP: nothing can move
        }
        // the continuation covers the common case only — a dry voice playing steadily while gains glide; silence anywhere
        // in the chain (resets instead of ramps) and chain-plan voices stay on the block-by-block path
        if (fx && vd.n_pre >= FW_MAX_STAGES - 1) {
            pre_sil = sil;
            sil = false;
        }
        if (ramping && (sil || upstream_silent || sp_zero || fx || k + 1 >= K)) steady = false;
        if (!steady) continue;
        // ---- ramp continuation.  From here to the end of the call nothing happens to this voice but (a) its playhead
        // advancing — closed form, as on a steady tail — and (b) smoothers gliding to their targets, a serial recurrence
        // (smoother.rs:169-175) that only needs ITSELF: run it now for as many blocks as it takes, 64 frames at a time,
        // straight into the blocks' ramp buffers, then emit all descriptors together.  (Walking those ~20 blocks one by
        // one through the whole state machine cost ~6 us each: a gain change per voice cost a third of config 2's step.)
        CTL_T(4);
#ifdef FW_CTL_TRACE
        tr_ramped = ramping;
#endif
        int ramp_until[2 * FW_MAX_STAGES];
#pragma unroll
        for (int sl = 0; sl < 2 * FW_MAX_STAGES; ++sl) ramp_until[sl] = 0;
        if (ramping) {
            // smoother by smoother (each one's recurrence is independent of the others'): see ramp_blocks
            const size_t blk_stride = (size_t)fv.n_voices * (size_t)fv.ramp_slots * (size_t)fv.stride;
            float* const rbase = fv.ramps + ((size_t)(k + 1) * fv.n_voices + vi) * (size_t)fv.ramp_slots * (size_t)fv.stride;
            int furthest = 0;
            if (!smoother_is_constant(ss.s0, ss.p0)) {  // the sampler's gain (both channels share the ramp: sampler.rs:530-533)
                const int u = ramp_glide(ss.s0, ss.p0, frames, k + 1, K, rbase, blk_stride, fv.stride, true, lane);
                ramp_until[0] = ramp_until[1] = u;
                furthest = u;
            }
#pragma unroll
            for (int j = 0; j < FW_MAX_STAGES - 1; ++j) {
                if (j < vd.n_stages) {
                    StageRegs& r = st[j];
                    float* rb = rbase + (size_t)(j + 1) * 2 * fv.stride;
                    const int kind = vd.stage_kind[j];
                    if (kind == K_VOLUME || kind == K_PAN || kind == K_WIDTH || kind == K_SPATIAL) {
                        if (!smoother_is_constant(r.s0, r.p0)) {
                            const int u = ramp_glide(r.s0, r.p0, frames, k + 1, K, rb, blk_stride, fv.stride, kind == K_VOLUME, lane);
                            ramp_until[2 * (j + 1)] = u;
                            if (kind == K_VOLUME) ramp_until[2 * (j + 1) + 1] = u;
                            furthest = u > furthest ? u : furthest;
                        }
                        if ((kind == K_PAN || kind == K_SPATIAL) && !smoother_is_constant(r.s1, r.p1)) {
                            const int u = ramp_glide(r.s1, r.p1, frames, k + 1, K, rb + fv.stride, blk_stride, fv.stride, false, lane);
                            ramp_until[2 * (j + 1) + 1] = u;
                            furthest = u > furthest ? u : furthest;
                        }
                    }
                }
            }
            ramping = furthest >= K;  // a glide that outlasts the call: the next call picks it up at block 0
        }
        CTL_T(5);
        // ---- steady: the descriptor every later block shares.  Constant gains are `input` for a settled
        // smoother and `last` for one stalled at its f32 fixed point (Q28).
        TailJob job;
        job.mode = mode;
        job.flags = (sil ? VB_SILENT : 0u) | ((upstream_silent || sp_zero || pre_sil) ? VB_SRC_ZERO : 0u) | sp_bits();
        job.sample = upstream_silent ? -1 : ss.sample;
        job.playhead = ss.playhead;
        job.loop_start = ss.loop_start;
        job.loop_end = ss.loop_end;
        if (mode == 3) job.loop_end = ss.has_loop ? (sd.frames << 32) : 0;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) job.g.g[j][0] = job.g.g[j][1] = 1.0f;
        if (!upstream_silent) {
            if (sd.channels == 1) job.flags |= VB_MONO;
            if (mode == 3) job.flags |= VB_RESAMPLE;
            else job.g.g[0][0] = job.g.g[0][1] = ss.s0.status == SM_ACTIVE ? ss.s0.last : ss.s0.input;
        }
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j) {
            if (j >= vd.n_stages) break;
            const StageRegs& r = st[j];
            job.g.g[j + 1][0] = vd.stage_kind[j] == K_HARD_CLIP ? r.p0 : (r.s0.status == SM_ACTIVE ? r.s0.last : r.s0.input);
            job.g.g[j + 1][1] = (vd.stage_kind[j] == K_PAN || vd.stage_kind[j] == K_SPATIAL) ? (r.s1.status == SM_ACTIVE ? r.s1.last : r.s1.input)
                                                          : job.g.g[j + 1][0];
            if (mid_sil[j]) job.g.g[j + 1][0] = job.g.g[j + 1][1] = -1.0f;  // (the sentinel: k_chain writes +0.0 there)
        }
#pragma unroll
        for (int sl = 0; sl < 2 * FW_MAX_STAGES; ++sl) job.ramp_until[sl] = ramp_until[sl];
        became_steady = !ramping;  // (a glide that outlasts the call: the next call picks it up block 0)
        if (w0 && became_steady) {
            VoiceCache vc;
            vc.epoch = fv.epoch;
            vc.mode = mode;
            vc.flags = job.flags;
            vc.sample = job.sample;
            vc.g = job.g;
            fv.cache[vi] = vc;
        }
        // (lazy records: what the calls after this one render from, fwgpu_types.h LazyRec)
        if (became_steady) make_lazy(fv, vi, vd.sampler_state, job, tail_end_playhead(job, k + 1, K, (uint64_t)frames), sd, fx, fxp, w0);
        if (k + 1 < K) {
            uint32_t tail_gs = 0;
            bool simple_ok = false;
            const bool tail_simple = mode != 3 && (fxp ? (simple_frames && (upstream_silent || pre_sil || (!fx && sil) || simple_capable(sd, true)))
                                                       : (!sil && !upstream_silent && simple_frames && simple_capable(sd, false)));
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
            CTL_T(6);
            uint64_t ph = steady_tail(fv, vi, lane, k + 1, K, job, sd, tail_gs, simple_ok, fx, fxp, k0 == 0);
            if (mode != 0) ss.playhead = ph;
        }
        break;
    }
#ifdef FW_CTL_TRACE
    CTL_T(7);
    if (w0 && tr_ramped)
        printf("ctl voice %d k0 %d: spans %llu fast %llu blocks %llu [state loads %llu, last block: msgs %llu sampler %llu stages %llu emit %llu] check %llu "
               "glide %llu job %llu tail %llu (x10 ns)\n", vi, k0, tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], tr[8] - tr[2], tr[9] - tr[8], tr[10] - tr[9],
               tr[11] - tr[10], tr[3] - tr[11], tr[4] - tr[3], tr[5] - tr[4], tr[6] - tr[5], tr[7] - tr[6]);
#endif
    if (!w0) return;
    if (!became_steady && fv.lazy != nullptr) {  // (lazy records: only a voice that ends the call steady leaves one)
        fv.lazy[vi].mode = -1;
        if (fv.horizon) atomicMin(fv.horizon, 0ull);
    }
    if (!became_steady) fv.cache[vi].epoch = 0;
    fv.states[vd.sampler_state] = ss;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
        if (j < vd.n_stages) *(StageRegs*)&fv.states[vd.stage_state[j]] = st[j];
    if (spv) {
        fv.states[sp_state].playing = sp_dl;
        fv.states[sp_state].has_loop = sp_dr;
    }
}

// Realtime edge (k_rt_block): ONE LANE per voice for the case that dominates a steady callback — a voice of the voice-bank
// plan that ended the previous call steady (VoiceCache) and has no message in this ONE-block call.  The same work as
// voice_control_wave's fast path (same steady_tail, so the same records), but 64 voices' dependent load chains
// (descriptor -> cache / sampler state -> sample table) run side by side in one wave instead of one after the other.
// Returns false — having written nothing — when the voice needs the state machines this block.
__device__ inline bool voice_control_lane_steady(const FusedView& fv, const int vi, const uint32_t cmd_block0) {
    const VoiceDesc vd = fv.voices[vi];
    const int frames = fv.frames;
    if (vd.sampler_state < 0) {  // a null voice: the cleared, silent-flagged buffer of schedule.rs:310-313
        VoiceRef r;
        r.src_l = nullptr;
        r.r_delta = 0;
        r.flags_gset = VB_SILENT;
        fv.refs[ref_index(vi, 0, fv.ref_kgroups)] = r;
        return true;
    }
    if (vd.bq_state >= 0 || vd.dl_state >= 0) return false;  // (chain-plan voices never come here)
    if (fv.n_cmds) {
        int first_cmd = first_cmd_block(fv.cmds, fv.n_cmds, vd.sampler_state, cmd_block0);
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
            if (j < vd.n_stages) {
                const int f = first_cmd_block(fv.cmds, fv.n_cmds, vd.stage_state[j], cmd_block0);
                first_cmd = f < first_cmd ? f : first_cmd;
            }
        if (first_cmd < 1) return false;
    }
    const VoiceCache vc = fv.cache[vi];
    if (vc.epoch != fv.epoch) return false;
    TailJob job;
    job.mode = vc.mode;
    job.flags = vc.flags;
    job.sample = vc.sample;
    job.g = vc.g;
    job.playhead = job.loop_start = job.loop_end = 0;
#pragma unroll
    for (int sl = 0; sl < 2 * FW_MAX_STAGES; ++sl) job.ramp_until[sl] = 0;
    SampleDesc sd;
    sd.data = nullptr;
    sd.frames = 0;
    sd.channels = 2;
    sd.format = FMT_P_F32;
    if (vc.mode != 0) {
        const NodeState* sp = &fv.states[vd.sampler_state];
        job.playhead = sp->playhead;
        job.loop_start = sp->loop_start;
        job.loop_end = sp->loop_end;
        sd = fv.samples[vc.sample];
        if (vd.src_kind == 2) mono_adapt(sd);
        if (vc.mode == 2 && job.playhead + (uint64_t)frames > sd.frames) return false;  // the one-shot ends in this block
        if (vc.mode == 3) {
            job.loop_end = sp->has_loop ? (sd.frames << 32) : 0;
            if (!sp->has_loop && !rs_survives(job.playhead, job.loop_start, (uint64_t)frames, 1, sd.frames)) return false;
        }
    }
    const bool no_src = (job.flags & VB_SRC_ZERO) || (job.flags & VB_SILENT);
    const bool simple_ok = vc.mode != 3 && !no_src && job.sample >= 0 && (frames & 3) == 0 && simple_capable(sd, false);
    if (simple_ok) fv.gsets[(size_t)vi * FW_GSETS] = job.g;
    const uint64_t ph = steady_tail(fv, vi, 0, 0, 1, job, sd, 0u, simple_ok, false, false);
    if (vc.mode != 0) fv.states[vd.sampler_state].playhead = ph;
    return true;
}

// Two register budgets: the full one (~240 VGPRs, nothing spilled) for a control kernel that has the GPU to itself, and 168 — three
// waves per SIMD, ~1 % of the dynamic instructions are scratch traffic — for control-ahead mode, where a control wave has to find its
// registers on a SIMD that holds five render waves of the call before: two of them retiring make room for the small one, three
// (without a refill in between) for the big one, which in practice only happens once the render grid has run dry.
template <int OCC>
__device__ __forceinline__ void voice_control_kernel(const FusedView& fv, const int K, const uint32_t cmd_block0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= fv.n_voices) return;
    // dispatch order: the voices with a message in this call (or the one before: their glides continue) go first — theirs are the
    // long waves (a glide is 22-25 us of serial latency)
    const int vi = fv.ctl_order ? __builtin_amdgcn_readfirstlane(fv.ctl_order[w]) : w;
    voice_control_wave<true>(fv, vi, threadIdx.x & (WAVE - 1), K, cmd_block0);
}
#ifndef CTL_SMALL_OCC
#define CTL_SMALL_OCC 3
#endif
__global__ __launch_bounds__(256) void k_voice_control(FusedView fv, int K, uint32_t cmd_block0) { voice_control_kernel<1>(fv, K, cmd_block0); }
__global__ __launch_bounds__(256, CTL_SMALL_OCC) void k_voice_control_small(FusedView fv, int K, uint32_t cmd_block0) { voice_control_kernel<3>(fv, K, cmd_block0); }
