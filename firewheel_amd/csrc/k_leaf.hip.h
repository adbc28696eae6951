// k_leaf.hip.h — part of the single device translation unit fwgpu_kernels.hip (included inside namespace fwgpu).
// Fused voice-bank plan: k_leaf_sum (HBM-streaming source fetch + gain stages + ordered leaf sums), upper sum tree, root + interleave.
#pragma once

// Leaf kernel: one wave per (leaf SumNode, block).  For each port in order: fetch the voice's source frames,
// run its gain stages in registers, and accumulate in the reference's summation order (nodes/sum.rs).
// HBM traffic = the source samples once (8 B per stereo voice-sample) + one partial-bus write per leaf.
// one chain stage on a quad of both channels (kinds: SK_* in fwgpu_types.h) — the arithmetic of the generic executor's
// K_VOLUME / K_PAN / K_WIDTH / K_HARD_CLIP cases (k_generic.hip.h), operation for operation
__device__ __forceinline__ void apply_stage(uint32_t kind, v4f g0, v4f g1, v4f& a, v4f& b) {
    if (kind == SK_GAIN) {
        a = a * g0;
        b = b * g1;
    } else if (kind == SK_WIDTH) {
        const v4f m = (a + b) * 0.5f;
        const v4f sd = ((a - b) * 0.5f) * g0;
        a = m + sd;
        b = m - sd;
    } else {  // SK_CLIP
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = fmaxf(fminf(a[j], g0[j]), -g0[j]);
            b[j] = fmaxf(fminf(b[j], g0[j]), -g0[j]);
        }
    }
}

// `prog`: the voice's stage program (4 bits per chain stage; 0 everywhere on a gains-only voice)
// LDS of the leaf kernel's program instantiation, for voices whose source is a resampler: the 2 KiB filter bank once per
// workgroup, and per wave the window of source frames its 256 output frames of ONE port read (both channels)
#define RS_WIN 1040  // window frames a wave can stage: 256 output frames x ratio <= 4, + RS_TAPS
struct RsLds {
    const float* tab;  // the filter bank in LDS, tap-major [RS_TAPS][RS_PHASES] (nullptr: no resampler voices in the plan)
    float* win;        // this wave's window [2][RS_WIN], then its result rows [2][256]
};

// source frame of a window slot: a loop wraps (by comparison: the slot is within a window of the loop), a one-shot reads 0 outside
__device__ __forceinline__ int64_t rs_slot(int64_t q, const int64_t len, const bool loop, bool& in) {
    in = true;
    if (loop) {
        while (q < 0) q += len;
        while (q >= len) q -= len;
    } else {
        in = q >= 0 && q < len;
    }
    return in ? q : 0;
}

// A leaf's two bus rows go out write-through (sc0 sc1: the root reads them from another XCD's L2 side).  ONE asm statement for the
// pair, both addresses computed before it and every operand live across it, and wait states behind it: written as two statements
// the compiler — which cannot see that the asm is a 128-bit store still reading its data registers — put the second store's
// address into v[2:3] of the first store's data three instructions after issuing it, and the last four lanes of every 16 took
// the pointer for audio (k_rt_persist, round 4: frames 48-63 of a leaf bus, whenever register allocation fell that way; round 3
// met the same garbage and blamed the kernel's size).  LLVM's hazard recognizer covers this for stores it knows
// (checkVALUHazardsHelper: VMEM store > 64 bits followed by a VALU write of its data); inline asm is opaque to it.
__device__ __forceinline__ void bus_store_pair(float* pl, float* pr, const v4f& a, const v4f& b) {
    asm volatile(
        "global_store_dwordx4 %0, %2, off sc0 sc1\n\t"
        "global_store_dwordx4 %1, %3, off sc0 sc1\n\t"
        "s_nop 7\n\t"
        "s_nop 7" ::"v"(pl),
        "v"(pr), "v"(a), "v"(b)
        : "memory");
}

// The full descriptor of (block k, voice): its VoiceBlk row — or, for a VB_RS_LEAN block (a steady resampler voice: fg = the
// record's flags, ref_src = its src_l field), the voice's template with the block's 32.32 position put in.
template <bool RS>
__device__ __forceinline__ VoiceBlk blk_load(const FusedView& fv, const size_t k, const int voice, const uint32_t fg, const float* ref_src) {
    if constexpr (RS) {
        if (fg & VB_RS_LEAN) {
            VoiceBlk d = fv.rs_tmpl[voice];
            d.off0 = (uint64_t)ref_src;
            return d;
        }
    }
    return fv.blks[k * fv.n_voices + voice];
}

// RS: the plan has voices whose source is a resampler (only the program instantiation of the leaf kernel carries that code)
// j_end: stages [0, j_end) are applied (a spatialiser voice stops in front of its last stage: leaf_sp_port does that one)
template <bool RS>
__device__ __forceinline__ void voice_eval(const FusedView& fv, const VoiceBlk& d, uint32_t k, int voice, int f0, int frames,
                                           v4f& xl, v4f& xr, uint32_t prog = 0u, RsLds rs = RsLds{nullptr, nullptr}, int j_end = FW_MAX_STAGES) {
    const bool mono = d.flags & VB_MONO;
    if (d.flags & VB_SRC_ZERO) {  // (spatialiser voices: the chain in front of the last stage is cleared this block)
        xl = xr = splat(0.f);
        return;
    }
    (void)rs;  // (resampler ports of a leaf are rendered by leaf_rs_piece below, staged through LDS and pipelined port over port; what
               //  reaches this function — a source format other than planar f32, a window beyond RS_WIN, a spatialiser's upstream —
               //  takes the frame-by-frame fetch)
    if (RS && (d.flags & VB_RESAMPLE)) {
        // SPEC resampling source — the arithmetic of the generic executor's K_RESAMPLER case (k_generic.hip.h), frame by frame:
        // 32.32 position, phase = top 5 fraction bits, 16-tap fmaf chain ascending from +0.0; outside a one-shot sample reads
        // 0, a loop wraps.  The taps of neighbouring frames overlap: the reuse is the L1's.
        SampleDesc sd;  // (field by field: a struct copy from global memory ends up in scratch here)
        sd.data = fv.samples[d.sample].data;
        sd.frames = fv.samples[d.sample].frames;
        sd.channels = fv.samples[d.sample].channels;
        sd.format = fv.samples[d.sample].format;
        const int64_t len = (int64_t)sd.frames;
        const bool loop = d.n1 != 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float al = 0.f, ar = 0.f;
            if (f0 + j < frames) {
                const uint64_t p = d.off0 + (uint64_t)(f0 + j) * d.off1;
                const float* hp = fv.rs_table + ((uint32_t)(p >> 27) & (RS_PHASES - 1)) * RS_TAPS;
                // index of tap 0.  A looping source keeps its position below len << 32 (len < 2^31) and a block adds less
                // than 2^20 frames: the frame index fits 32 bits — ONE 32-bit remainder per frame, the taps wrap by
                // comparison (the generic node's `j %= len` per tap is a 64-bit division each)
                const int64_t q0 = (loop ? (int64_t)((uint32_t)(p >> 32) % (uint32_t)len) : (int64_t)(p >> 32)) - (RS_TAPS / 2 - 1);
                for (int t = 0; t < RS_TAPS; ++t) {
                    int64_t q = q0 + t;
                    float x0 = 0.f, x1 = 0.f;
                    bool in = true;
                    if (loop) {
                        while (q < 0) q += len;
                        while (q >= len) q -= len;
                    } else {
                        in = q >= 0 && q < len;
                    }
                    if (in) {
                        x0 = sample_fetch(sd, 0, (uint64_t)q);
                        if (!mono) x1 = sample_fetch(sd, 1, (uint64_t)q);
                    }
                    al = __builtin_fmaf(hp[t], x0, al);
                    if (!mono) ar = __builtin_fmaf(hp[t], x1, ar);
                }
                if (mono) ar = al;
            }
            xl[j] = al;
            xr[j] = ar;
        }
    } else if (d.src_l && f0 + 4 <= frames) {  // planar f32, contiguous: one dwordx4 per channel per lane
        xl = *(const v4f_u*)(d.src_l + f0);
        xr = mono ? xl : *(const v4f_u*)(d.src_r + f0);
    } else {
        SampleDesc sd;  // (field by field: a struct copy from global memory ends up in scratch here)
        sd.data = fv.samples[d.sample].data;
        sd.frames = fv.samples[d.sample].frames;
        sd.channels = fv.samples[d.sample].channels;
        sd.format = fv.samples[d.sample].format;
        Fetch ft;
        ft.off0 = d.off0;
        ft.off1 = d.off1;
        ft.n1 = d.n1;
        ft.wrap = (d.flags & VB_WRAP) ? 1 : 0;
        ft.tail_zero = (d.flags & VB_TAIL_ZERO) ? 1 : 0;
        xl = sample_fetch4(sd, 0, ft, (uint32_t)f0, (uint32_t)frames);
        xr = mono ? xl : sample_fetch4(sd, 1, ft, (uint32_t)f0, (uint32_t)frames);
    }
    const uint32_t rbits = (d.flags & VB_RAMP_MASK) >> VB_RAMP_SHIFT;
    const uint32_t kinds = prog << 4;  // stage 0 is the sampler's own gain
    if (rbits == 0) {  // constant gains: sampler.rs:530-533 then volume.rs:123-126 / pan, one rounding each
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j)  // (no `break`: the loop must unroll, or d.g[] is indexed at run time and lands in scratch)
            if (j < fv.n_gain_stages && j < j_end) apply_stage((kinds >> (4 * j)) & 15u, splat(d.g[j][0]), splat(d.g[j][1]), xl, xr);
        // a mono sample is duplicated AFTER the sampler gain (sampler.rs:546-551); identical values either way
    } else {
        const float* rb = fv.ramps + ((size_t)k * fv.n_voices + voice) * (size_t)fv.ramp_slots * (size_t)fv.stride + f0;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) {
            if (j < fv.n_gain_stages && j < j_end) {
                v4f gl = (rbits >> (2 * j)) & 1u ? *(const v4f*)(rb + (size_t)(2 * j) * fv.stride) : splat(d.g[j][0]);
                v4f gr = (rbits >> (2 * j + 1)) & 1u ? *(const v4f*)(rb + (size_t)(2 * j + 1) * fv.stride) : splat(d.g[j][1]);
                apply_stage((kinds >> (4 * j)) & 15u, gl, gr, xl, xr);
            }
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
#ifndef LEAF_NT_STORE
// The leaf buses' stores: the next reader is another kernel.  1 = non-temporal (-1 % on the kernel against plain stores);
// 2 = write-through (sc0 sc1): as fast as 1 when the bus and the samples sit in different thirds of HBM, and 8 us (3 %) faster
// when they share one — reads and writes mixed on one rank of the stacks are what costs this kernel its 12 % in the "slow
// state" (scripts/ubench/bus_place.hip, DESIGN §7: nt 282 / plain 287 / sc0 sc1 274 us there; 252 / 256 / 251 us otherwise)
#define LEAF_NT_STORE 2
#endif
#ifndef LEAF_WPB
#define LEAF_WPB 4  // waves (leaf, block work items) per workgroup
static_assert(LEAF_WPB <= LEAF_WPB_MAX, "fwgpu_types.h: LEAF_WPB_MAX sizes the resampler work list");
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
#ifndef LEAF_SPEC_GSET
#define LEAF_SPEC_GSET 1  // gain set 0 of every port is requested together with the port's record (it is the one in use on a
                          // steady voice), not after it: one dependent memory round trip less at the head of every wave
#endif
// U: ports whose source loads are in flight together — LEAF_U in the throughput kernels (occupancy hides the round trips),
// 16 in the realtime kernel, where ONE wave adds a leaf's 32 ports and every round trip is on the callback's critical path
template <int NG, int U>
__device__ __forceinline__ void leaf_fast(const float* my_l, const float* my_r, const GainSet& my_g, int p_begin, int ports, int f0,
                                          v4f& accl, v4f& accr) {
    for (int p0 = p_begin; p0 < ports; p0 += U) {  // ports [p_begin, ports): a run of plain ports, accumulators carried in
        v4f xl[U], xr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p0 + u < ports) {
                xl[u] = gload4(readlane_ptr(my_l, p0 + u) + f0);
                xr[u] = gload4(readlane_ptr(my_r, p0 + u) + f0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
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

// the same with a stage program per voice (width / hard clip among the stages): lane p also holds port p's program; the
// stage kind of (port, stage) is wave-uniform, so the dispatch is a scalar branch
template <int U>
__device__ __forceinline__ void leaf_fast_prog(const float* my_l, const float* my_r, const GainSet& my_g, uint32_t my_prog, int ng,
                                               int p_begin, int ports, int f0, v4f& accl, v4f& accr) {
    for (int p0 = p_begin; p0 < ports; p0 += U) {
        v4f xl[U], xr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p0 + u < ports) {
                xl[u] = gload4(readlane_ptr(my_l, p0 + u) + f0);
                xr[u] = gload4(readlane_ptr(my_r, p0 + u) + f0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p0 + u < ports) {
                v4f a = xl[u], b = xr[u];
                const uint32_t kinds = (uint32_t)__builtin_amdgcn_readlane((int)my_prog, p0 + u) << 4;
#pragma unroll
                for (int j = 0; j < FW_MAX_STAGES; ++j) {
                    if (j >= ng) break;
                    apply_stage((kinds >> (4 * j)) & 15u, splat(readlane_f(my_g.g[j][0], p0 + u)), splat(readlane_f(my_g.g[j][1], p0 + u)), a, b);
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

// ---- VB_SIMPLE source classes other than planar f32 (SF_*, fwgpu_types.h): raw vector fetch of 4 consecutive frames
// of both channels, converted exactly as core/sample_resource.rs:338-345 does per element
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef const v4i __attribute__((address_space(1), aligned(4)))* gv4ip;
typedef const v2i __attribute__((address_space(1), aligned(4)))* gv2ip;
struct RawQuad {
    v4i a, b;
};
__device__ __forceinline__ float cvt_i16(int half) { return (float)(int)(short)half * (1.0f / 32767.0f); }             // :338-340
__device__ __forceinline__ float cvt_u16(int half) { return ((float)(unsigned)(half & 0xffff) * (2.0f / 65535.0f)) - 1.0f; }  // :343-345
template <uint32_t CLS>
__device__ __forceinline__ RawQuad raw_load(const float* base, uint32_t rdelta, int f0) {
    RawQuad q;
    q.a = q.b = (v4i){0, 0, 0, 0};
    const uint64_t p = (uint64_t)base;
    if (CLS == SF_P_I16 || CLS == SF_P_U16) {  // 4 x 16 bit = 8 bytes per channel
        const v2i l = __builtin_nontemporal_load((gv2ip)(p + 2ull * (uint64_t)f0));
        const v2i r = __builtin_nontemporal_load((gv2ip)(p + 2ull * ((uint64_t)rdelta + (uint64_t)f0)));
        q.a[0] = l[0];
        q.a[1] = l[1];
        q.b[0] = r[0];
        q.b[1] = r[1];
    } else if (CLS == SF_I_I16 || CLS == SF_I_U16) {  // 4 frames x (L,R) x 16 bit = one 16-byte load
        q.a = __builtin_nontemporal_load((gv4ip)(p + 4ull * (uint64_t)f0));
    } else {  // SF_I_F32: 4 frames x (L,R) x f32 = two 16-byte loads
        q.a = __builtin_nontemporal_load((gv4ip)(p + 8ull * (uint64_t)f0));
        q.b = __builtin_nontemporal_load((gv4ip)(p + 8ull * (uint64_t)f0 + 16ull));
    }
    return q;
}
template <uint32_t CLS>
__device__ __forceinline__ void raw_convert(const RawQuad& q, v4f& xl, v4f& xr) {
    if (CLS == SF_P_I16) {
        xl = (v4f){cvt_i16(q.a[0]), cvt_i16(q.a[0] >> 16), cvt_i16(q.a[1]), cvt_i16(q.a[1] >> 16)};
        xr = (v4f){cvt_i16(q.b[0]), cvt_i16(q.b[0] >> 16), cvt_i16(q.b[1]), cvt_i16(q.b[1] >> 16)};
    } else if (CLS == SF_P_U16) {
        xl = (v4f){cvt_u16(q.a[0]), cvt_u16(q.a[0] >> 16), cvt_u16(q.a[1]), cvt_u16(q.a[1] >> 16)};
        xr = (v4f){cvt_u16(q.b[0]), cvt_u16(q.b[0] >> 16), cvt_u16(q.b[1]), cvt_u16(q.b[1] >> 16)};
    } else if (CLS == SF_I_I16) {
        xl = (v4f){cvt_i16(q.a[0]), cvt_i16(q.a[1]), cvt_i16(q.a[2]), cvt_i16(q.a[3])};
        xr = (v4f){cvt_i16(q.a[0] >> 16), cvt_i16(q.a[1] >> 16), cvt_i16(q.a[2] >> 16), cvt_i16(q.a[3] >> 16)};
    } else if (CLS == SF_I_U16) {
        xl = (v4f){cvt_u16(q.a[0]), cvt_u16(q.a[1]), cvt_u16(q.a[2]), cvt_u16(q.a[3])};
        xr = (v4f){cvt_u16(q.a[0] >> 16), cvt_u16(q.a[1] >> 16), cvt_u16(q.a[2] >> 16), cvt_u16(q.a[3] >> 16)};
    } else {  // SF_I_F32
        xl = (v4f){__int_as_float(q.a[0]), __int_as_float(q.a[2]), __int_as_float(q.b[0]), __int_as_float(q.b[2])};
        xr = (v4f){__int_as_float(q.a[1]), __int_as_float(q.a[3]), __int_as_float(q.b[1]), __int_as_float(q.b[3])};
    }
}
// any class, one port (the mixed-leaf path)
__device__ __forceinline__ void simple_fetch(uint32_t cls, const float* base, uint32_t rdelta, int f0, v4f& xl, v4f& xr) {
    switch (cls) {
        case SF_P_F32:
            xl = gload4(base + f0);
            xr = gload4(base + rdelta + f0);
            break;
        case SF_P_I16: raw_convert<SF_P_I16>(raw_load<SF_P_I16>(base, rdelta, f0), xl, xr); break;
        case SF_P_U16: raw_convert<SF_P_U16>(raw_load<SF_P_U16>(base, rdelta, f0), xl, xr); break;
        case SF_I_I16: raw_convert<SF_I_I16>(raw_load<SF_I_I16>(base, rdelta, f0), xl, xr); break;
        case SF_I_U16: raw_convert<SF_I_U16>(raw_load<SF_I_U16>(base, rdelta, f0), xl, xr); break;
        default: raw_convert<SF_I_F32>(raw_load<SF_I_F32>(base, rdelta, f0), xl, xr); break;
    }
}
// every port VB_SIMPLE and of ONE class: LEAF_U ports' raw loads in flight together, converted afterwards
template <uint32_t CLS>
__device__ __forceinline__ void leaf_fast_cls(const float* my_l, uint32_t my_rd, const GainSet& my_g, int ng, int ports, int f0,
                                              v4f& accl, v4f& accr) {
    for (int p0 = 0; p0 < ports; p0 += LEAF_U) {
        RawQuad q[LEAF_U];
#pragma unroll
        for (int u = 0; u < LEAF_U; ++u)
            if (p0 + u < ports)
                q[u] = raw_load<CLS>(readlane_ptr(my_l, p0 + u), (uint32_t)__builtin_amdgcn_readlane((int)my_rd, p0 + u), f0);
#pragma unroll
        for (int u = 0; u < LEAF_U; ++u) {
            if (p0 + u < ports) {
                v4f a, b;
                raw_convert<CLS>(q[u], a, b);
#pragma unroll
                for (int j = 0; j < FW_MAX_STAGES; ++j) {  // sampler.rs:530-533, volume.rs:123-126, pan: one rounding each
                    if (j >= ng) break;
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

// ---- a voice whose LAST stage is a SPEC spatialiser (k_generic.hip.h K_SPATIAL): m = (L + R) * 0.5 of what the
// stages in front of it deliver; outL[i] = m[i - dL] * gL[i], outR[i] = m[i - dR] * gR[i]; the 64 frames in front of the wave's
// piece come from — the block before (k > 0: re-rendered from ITS record, 64 frames on 16 lanes), the call's history scratch
// (k == 0: what k_voice_control copied out of the node's ext slice), or the same block's earlier frames (a later piece of a long
// block).  The mono row goes through LDS (64 history + 256 current floats per wave) and comes back shifted by the ear delays.
// The wave that renders the LAST piece of the call's LAST block leaves the next call's history in the ext slice.
struct SpPort {
    uint32_t fg;       // the block's VoiceRef::flags_gset (VB_* flags, source class, ear delays)
    const float* src;  // VB_SIMPLE: source of frame 0 / channel-1 offset
    uint32_t rd;
    uint32_t fg_prev;  // the same of the block before (k > 0)
    const float* src_prev;
    uint32_t rd_prev;
};
template <bool RS>
__device__ __forceinline__ void sp_upstream(const FusedView& fv, uint32_t fg, const float* src, uint32_t rd, const GainSet* gs_lane, int p,
                                            uint32_t k, int voice, int f0, int frames, uint32_t prog, int js, bool active, v4f& a, v4f& b) {
    // what arrives at the spatialiser for frames f0 .. f0+3 of block k: source x stages [0, js]  (js = the spatialiser's index among
    // the chain stages: gain-set rows 0 .. js are the sampler's gain and the stages in front of it)
    a = b = splat(0.f);
    if (!active || (fg & VB_SRC_ZERO)) return;
    if (fg & VB_SIMPLE) {
        simple_fetch((fg >> 16) & 7u, src, rd, f0, a, b);
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j)
            if (j <= js) apply_stage(((prog << 4) >> (4 * j)) & 15u, splat(readlane_f(gs_lane->g[j][0], p)), splat(readlane_f(gs_lane->g[j][1], p)), a, b);
    } else {
        const VoiceBlk d = blk_load<RS>(fv, k, voice, fg, src);
        voice_eval<RS>(fv, d, k, voice, f0, frames, a, b, prog, RsLds{nullptr, nullptr}, js + 1);
    }
}

// ---- resampler voices (SPEC resampling source): a leaf with such ports renders its pieces here.
// Per resampler port the wave stages the source window its (up to) 256 output frames read — frames [i_first - 7, i_last + 8],
// both channels INTERLEAVED as {L, R} pairs — in LDS: the active lanes fetch it coalesced (consecutive lanes, consecutive frames;
// wrapped / zero-filled on the way in) and every lane then runs its four frames' 16-tap fmaf chains from LDS — ascending from
// +0.0, the arithmetic of k_generic's K_RESAMPLER.  What the round-2 version of this path paid for:
//   * LDS bytes: h, L and R as three ds_read_b32 per tap (128 B/clk/CU).  Now one ds_read_b64 {L, R} per tap and one ds_read_b64
//     {h[2t], h[2t+1]} per tap PAIR from a bank stored [tap pair][phase][2] — b64 reads run at 256 B/clk/CU on CDNA4 (3 LDS cycles
//     per tap instead of 6), and the two channels' fmaf are one packed instruction.
//   * latency: descriptor load -> window fetch -> LDS -> convolution, one port after the other, ~3 waves per SIMD to hide it.  Now
//     lane p holds port p's descriptor (one load for the leaf), and the NEXT resampler port's window is requested (8 rounds of
//     registers) before the current one is convolved.
// Frames are dealt to the lanes ROUND-ROBIN for the convolution (lane l: frames l, l + nact, ...): neighbouring lanes read
// neighbouring window slots (stride = the ratio) and coefficient addresses phase-apart — no 4-way bank conflicts as with four
// consecutive frames per lane; the results pass through LDS once more to come back as each lane's four consecutive frames.
typedef float v2f_rs __attribute__((ext_vector_type(2)));
#define RS_LOOP_BIT (1u << 31)  // lane-held copy of VoiceBlk::flags: n1 != 0 (the source loops) — bit 31 is free on a resampler block
#define RS_ROUNDS 8
typedef const float __attribute__((address_space(1)))* rs_gfp;  // (a generic pointer here makes the window fetch FLAT loads, which also
                                                                 //  count on lgkmcnt: every LDS wait would then wait for the prefetch)
typedef const volatile v2f_rs __attribute__((address_space(3)))* rs_lp;  // volatile: keeps ds_read_b64 (256 B/clk/CU) from being merged
                                                                         // into ds_read2_b64 (128 B/clk/CU) — MI355X_MICROARCH.md §LDS
struct RsPort {  // wave-uniform: one resampler port's piece
    rs_gfp s0;
    int len, qb, W;
    uint32_t i_first;
    uint64_t p_first, step;
    bool mono, loop;
    bool contig;  // the window, rounded up to whole quads, lies inside the sample: fetched as dwordx4 per lane and channel
};
typedef const v4f_u __attribute__((address_space(1)))* rs_g4p;
__device__ __forceinline__ int rs_slot32(const RsPort& P, const int r, bool& in) {
    int q = P.qb + r;
    in = r < P.W;
    if (P.loop) {  // (len >= the window: one step either way)
        if (q < 0) q += P.len;
        if (q >= P.len) q -= P.len;
    } else {
        in = in && q >= 0 && q < P.len;
    }
    return in ? q : 0;
}
// the window of port P is requested: RS_ROUNDS registers per channel — two rounds of quads when the window is contiguous in the
// sample (every steady block but the one a loop wraps in), else eight rounds of single frames
__device__ __forceinline__ void rs_issue(const RsPort& P, const int lane, const int nact, float (&a)[RS_ROUNDS], float (&b)[RS_ROUNDS]) {
    if (P.contig) {
#pragma unroll
        for (int u = 0; u < RS_ROUNDS / 4; ++u) {
            const int r = (lane + u * nact) * 4;
            if (u * nact * 4 < P.W) {  // (uniform)
                v4f x = splat(0.f), y = splat(0.f);
                if (r < P.W) {
                    x = *(rs_g4p)(P.s0 + P.qb + r);
                    if (!P.mono) y = *(rs_g4p)(P.s0 + P.len + P.qb + r);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[4 * u + e] = x[e];
                    b[4 * u + e] = y[e];
                }
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < RS_ROUNDS; ++u) {
            if (u * nact < P.W) {  // (uniform)
                bool in;
                const int q = rs_slot32(P, lane + u * nact, in);
                a[u] = in ? P.s0[q] : 0.f;
                b[u] = in && !P.mono ? P.s0[P.len + q] : 0.f;
            }
        }
    }
}
__device__ __forceinline__ void rs_stage(const RsPort& P, const int lane, const int nact, const float (&a)[RS_ROUNDS], const float (&b)[RS_ROUNDS],
                                         v2f_rs* win) {
    if (P.contig) {
#pragma unroll
        for (int u = 0; u < RS_ROUNDS / 4; ++u) {
            const int r = (lane + u * nact) * 4;
            if (u * nact * 4 < P.W && r < P.W) {
                if (P.mono) {
                    *(v4f*)(win + r) = (v4f){a[4 * u], a[4 * u], a[4 * u + 1], a[4 * u + 1]};
                    *(v4f*)(win + r + 2) = (v4f){a[4 * u + 2], a[4 * u + 2], a[4 * u + 3], a[4 * u + 3]};
                } else {
                    *(v4f*)(win + r) = (v4f){a[4 * u], b[4 * u], a[4 * u + 1], b[4 * u + 1]};
                    *(v4f*)(win + r + 2) = (v4f){a[4 * u + 2], b[4 * u + 2], a[4 * u + 3], b[4 * u + 3]};
                }
            }
        }
        for (int r = (RS_ROUNDS / 4 * nact + lane) * 4; r < P.W; r += nact * 4) {  // (a piece of few frames: the rest of the window)
            const v4f x = *(rs_g4p)(P.s0 + P.qb + r);
            const v4f y = P.mono ? x : *(rs_g4p)(P.s0 + P.len + P.qb + r);
            *(v4f*)(win + r) = (v4f){x[0], y[0], x[1], y[1]};
            *(v4f*)(win + r + 2) = (v4f){x[2], y[2], x[3], y[3]};
        }
    } else {
#pragma unroll
        for (int u = 0; u < RS_ROUNDS; ++u) {
            const int r = lane + u * nact;
            if (u * nact < P.W && r < P.W) win[r] = (v2f_rs){a[u], P.mono ? a[u] : b[u]};
        }
        for (int r = RS_ROUNDS * nact + lane; r < P.W; r += nact) {
            bool in;
            const int q = rs_slot32(P, r, in);
            const float x = in ? P.s0[q] : 0.f;
            const float y = in && !P.mono ? P.s0[P.len + q] : 0.f;
            win[r] = (v2f_rs){x, P.mono ? x : y};
        }
    }
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t x, int lane) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
}
__device__ __forceinline__ RsPort rs_port(const int p, const int fbase, const int nfr, const float* my_l, const uint32_t my_rd, const uint64_t my_off0,
                                          const uint64_t my_off1, const uint32_t my_df) {
    RsPort P;
    const uint32_t df = (uint32_t)__builtin_amdgcn_readlane((int)my_df, p);
    P.s0 = (rs_gfp)(uint64_t)readlane_ptr(my_l, p);
    P.len = __builtin_amdgcn_readlane((int)my_rd, p);
    P.step = readlane_u64(my_off1, p);
    P.p_first = readlane_u64(my_off0, p) + (uint64_t)fbase * P.step;
    const uint64_t p_last = P.p_first + (uint64_t)(nfr - 1) * P.step;
    P.i_first = (uint32_t)(P.p_first >> 32);
    P.W = (int)((uint32_t)(p_last >> 32) - P.i_first) + RS_TAPS;
    P.mono = df & VB_MONO;
    P.loop = df & RS_LOOP_BIT;
    uint32_t i0 = P.i_first;
    if (P.loop)
        while (i0 >= (uint32_t)P.len) i0 -= (uint32_t)P.len;  // (a looping source keeps its position below len << 32: at most a block's advance)
    P.qb = (int)i0 - (RS_TAPS / 2 - 1);
    P.contig = P.qb >= 0 && P.qb + ((P.W + 3) & ~3) <= P.len;
    return P;
}
template <bool PROG, bool RS>
__device__ __forceinline__ void leaf_rs_piece(const FusedView& fv, const LeafDesc& ld, const uint32_t k, const size_t row, const int f0, const int frames,
                                              const RsLds rs, const float* my_l, const uint32_t my_rd, const uint32_t my_cls, const GainSet& my_g,
                                              const uint32_t my_prog, const uint64_t my_off0, const uint64_t my_off1, const uint32_t my_df,
                                              const uint64_t silent_ports, const uint64_t simple_ports, const uint64_t rs_ports, const uint64_t lean_ports,
                                              const bool masked, v4f& accl, v4f& accr) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int fbase = __builtin_amdgcn_readfirstlane(f0 - lane * 4);  // the wave's piece of the block starts here
    const int nfr = frames - fbase < 256 ? frames - fbase : 256;
    const int nact = (nfr + 3) >> 2;  // active lanes: 0 .. nact-1
    v2f_rs* win = (v2f_rs*)rs.win;
    v2f_rs* orow = win + RS_WIN;
    const v2f_rs* tab = (const v2f_rs*)rs.tab;
    float wa[RS_ROUNDS], wb[RS_ROUNDS];
    RsPort N;  // the port whose window is in flight
    int pn = rs_ports ? __builtin_ctzll(rs_ports) : 64;
    if (pn < ld.ports) {
        N = rs_port(pn, fbase, nfr, my_l, my_rd, my_off0, my_off1, my_df);
        rs_issue(N, lane, nact, wa, wb);
    }
    for (int p = 0; p < ld.ports; ++p) {
        const bool psil = (silent_ports >> p) & 1ull;
        v4f xl = splat(0.f), xr = splat(0.f);  // a silent chain's buffers hold cleared zeros
        if (!psil) {
            const uint32_t prog = PROG ? (uint32_t)__builtin_amdgcn_readlane((int)my_prog, p) : 0u;
            if ((rs_ports >> p) & 1ull) {
                const RsPort P = N;  // (p == pn)
                rs_stage(P, lane, nact, wa, wb, win);
                const uint64_t later = p + 1 < 64 ? rs_ports >> (p + 1) : 0ull;
                pn = later ? p + 1 + __builtin_ctzll(later) : 64;
                if (pn < ld.ports) {
                    N = rs_port(pn, fbase, nfr, my_l, my_rd, my_off0, my_off1, my_df);
                    rs_issue(N, lane, nact, wa, wb);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint64_t dpos = (uint64_t)nact * P.step;
                const uint64_t pos_last = P.p_first + (uint64_t)(nfr - 1) * P.step;
                uint64_t pos = P.p_first + (uint64_t)lane * P.step;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // (frames past the piece's end convolve its last frame's window and drop the result: no branch around the
                    //  reads, so the four chains' LDS traffic can be scheduled together)
                    const int f = lane + i * nact;
                    const uint64_t ps = f < nfr ? pos : pos_last;
                    const rs_lp hp = (rs_lp)(tab + ((uint32_t)(ps >> 27) & (RS_PHASES - 1)));
                    const rs_lp wp = (rs_lp)(win + ((uint32_t)(ps >> 32) - P.i_first));
                    v2f_rs acc = (v2f_rs){0.f, 0.f};
#pragma unroll
                    for (int tp = 0; tp < RS_TAPS / 2; ++tp) {
                        const v2f_rs h = hp[tp * RS_PHASES];
                        const v2f_rs x0 = wp[2 * tp], x1 = wp[2 * tp + 1];
                        acc = __builtin_elementwise_fma((v2f_rs){h.x, h.x}, x0, acc);
                        acc = __builtin_elementwise_fma((v2f_rs){h.y, h.y}, x1, acc);
                    }
                    if (f < nfr) orow[f] = acc;
                    pos += dpos;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const v4f o01 = *(const v4f*)(orow + lane * 4), o23 = *(const v4f*)(orow + lane * 4 + 2);
                xl = (v4f){o01[0], o01[2], o23[0], o23[2]};
                xr = (v4f){o01[1], o01[3], o23[1], o23[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (f0 + j >= frames) xl[j] = xr[j] = 0.f;
                // the next port overwrites the window and the result row: everybody is done reading first
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // the voice's stages: constants from lane p's copy of the descriptor, or the block's ramps (voice_eval's tail)
                const uint32_t rbits = ((uint32_t)__builtin_amdgcn_readlane((int)my_df, p) & VB_RAMP_MASK) >> VB_RAMP_SHIFT;
                const uint32_t kinds = prog << 4;
                const float* rb = fv.ramps + ((size_t)k * fv.n_voices + (ld.first_voice + p)) * (size_t)fv.ramp_slots * (size_t)fv.stride + f0;
#pragma unroll
                for (int j = 0; j < FW_MAX_STAGES; ++j) {
                    if (j < fv.n_gain_stages) {
                        v4f gl = splat(readlane_f(my_g.g[j][0], p)), gr = splat(readlane_f(my_g.g[j][1], p));
                        if (rbits) {
                            if ((rbits >> (2 * j)) & 1u) gl = *(const v4f*)(rb + (size_t)(2 * j) * fv.stride);
                            if ((rbits >> (2 * j + 1)) & 1u) gr = *(const v4f*)(rb + (size_t)(2 * j + 1) * fv.stride);
                        }
                        apply_stage((kinds >> (4 * j)) & 15u, gl, gr, xl, xr);
                    }
                }
            } else if ((simple_ports >> p) & 1ull) {  // VB_SIMPLE implies frames % 4 == 0
                simple_fetch((uint32_t)__builtin_amdgcn_readlane((int)my_cls, p), readlane_ptr(my_l, p), (uint32_t)__builtin_amdgcn_readlane((int)my_rd, p), f0,
                             xl, xr);
#pragma unroll
                for (int j = 0; j < FW_MAX_STAGES; ++j)
                    if (j < fv.n_gain_stages)
                        apply_stage(((prog << 4) >> (4 * j)) & 15u, splat(readlane_f(my_g.g[j][0], p)), splat(readlane_f(my_g.g[j][1], p)), xl, xr);
            } else {
                const VoiceBlk d = blk_load<RS>(fv, k, ld.first_voice + p, (lean_ports >> p) & 1ull ? VB_RS_LEAN : 0u, readlane_ptr(my_l, p));
                voice_eval<RS>(fv, d, k, ld.first_voice + p, f0, frames, xl, xr, prog);
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

// A (leaf, block) is "resampler-pure" when every port is silent or a resampler voice in its plain state: planar f32 sample, constant
// gains, at most RS2_WIN window frames per 256-frame piece.  Such leaves are rendered by k_leaf_rs — leaf_rs_piece's arithmetic with
// nothing else in the kernel: 4 waves per SIMD instead of 3, a third of the instructions, no pass through LDS for the results.  The
// others (a voice that ramps, another sample format, sampler voices under the same mixer) are put on a work list by k_leaf_rs and
// rendered by k_leaf_sum_wl, the general kernel over that list.
#define RS2_WIN 512  // = the two rounds of quads a wave fetches: 256 output frames x ratio <= 1.93, + RS_TAPS
#define RS_CONTIG_BIT (1u << 30)  // lane-held flags: the window of the WHOLE block is contiguous inside the sample (no loop wrap, no one-shot edge)
struct RsPure {
    const float* s0;  // channel 0 of the sample (channel 1 = s0 + len)
    uint32_t len;     // sample frames
    int qb0;          // sample frame of window slot 0 of the block's first frame (loops: reduced into [−7, len))
    uint64_t off0, step;
    uint32_t df;      // VoiceBlk::flags | RS_LOOP_BIT | RS_CONTIG_BIT
    bool pure;
};
__device__ __forceinline__ RsPure rs_pure_lane(const FusedView& fv, const size_t row, const int lane, const int ports, const uint32_t my_flags,
                                               const int frames, const int voice, const uint64_t ref_pos) {
    RsPure r;
    r.s0 = nullptr;
    r.len = 0u;
    r.qb0 = 0;
    r.off0 = r.step = 0ull;
    r.df = 0u;
    r.pure = false;
    if (lane < ports && !(my_flags & (VB_SILENT | VB_SIMPLE))) {
        const bool lean = (my_flags & VB_RS_LEAN) != 0;  // a steady voice: the call's template + this block's position from the record
        const VoiceBlk* b = lean ? fv.rs_tmpl + voice : fv.blks + row + lane;
        const uint32_t df = b->flags;
        if ((df & VB_RESAMPLE) && !(df & VB_RAMP_MASK) && ((df >> VB_FMT_SHIFT) & 7u) == (uint32_t)FMT_P_F32) {
            const uint32_t len = b->pad;
            const uint64_t off0 = lean ? ref_pos : b->off0, step = b->off1;
            const bool loop = b->n1 != 0;
            const int nfr = frames < 256 ? frames : 256;
            const uint64_t w_max = (((uint64_t)nfr * step + 0xffffffffull) >> 32) + 1 + RS_TAPS;  // any piece of the block
            const uint64_t span = (((uint64_t)frames * step + 0xffffffffull) >> 32) + 1 + RS_TAPS + 4;  // the block's window, quad-rounded
            const uint64_t i_first = off0 >> 32;
            if (w_max <= RS2_WIN && i_first + span < (1ull << 30) && len < (1u << 30) && len >= 1u && (!loop || len >= (uint32_t)(RS2_WIN + RS_TAPS))) {
                const uint32_t i0 = loop ? (uint32_t)i_first % len : (uint32_t)i_first;
                const int qb0 = (int)i0 - (RS_TAPS / 2 - 1);
                const bool contig = qb0 >= 0 && (uint64_t)qb0 + span <= (uint64_t)len;
                r.s0 = b->src_l;
                r.len = len;
                r.qb0 = qb0;
                r.off0 = off0;
                r.step = step;
                r.df = (df & ~(RS_LOOP_BIT | RS_CONTIG_BIT)) | (loop ? RS_LOOP_BIT : 0u) | (contig ? RS_CONTIG_BIT : 0u);
                r.pure = true;
            }
        }
    }
    return r;
}

// ---- the spatialiser stage's steady path: every port of the leaf a VB_SIMPLE voice of ONE source class and ONE stage program that
// ends in a spatialiser.  SP_U ports' source quads (the piece + the 64 frames in front of it) are in flight together; each
// port has its own mono row in LDS, so one wave barrier serves the batch.  Arithmetic = the port-by-port path below.
#define SP_U 4
#define SP_ROW (SP_HIST + 256)
template <uint32_t CLS>
__device__ __forceinline__ RawQuad sp_raw(const float* base, uint32_t rdelta, int f) {
    if constexpr (CLS == SF_P_F32) {
        RawQuad q;
        q.a = (v4i)gload4(base + f);
        q.b = (v4i)gload4(base + rdelta + f);
        return q;
    } else {
        return raw_load<CLS>(base, rdelta, f);
    }
}
template <uint32_t CLS>
__device__ __forceinline__ void sp_cvt(const RawQuad& q, v4f& a, v4f& b) {
    if constexpr (CLS == SF_P_F32) {
        a = (v4f)q.a;
        b = (v4f)q.b;
    } else {
        raw_convert<CLS>(q, a, b);
    }
}
template <uint32_t CLS>
__device__ __forceinline__ void leaf_sp_fast(const FusedView& fv, const LeafDesc& ld, const uint32_t k, const int K, const int part, const int wpk,
                                             const uint32_t prog, const int jn, const float* my_l, const uint32_t my_rd, const GainSet& my_g,
                                             const uint32_t fg, const uint64_t plp, const uint32_t rdp, const GainSet& gp, float* sp_lds,
                                             float* outl, float* outr, float* tails, bool& tails_valid) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int frames = fv.frames;
    const int tq = (lane & 15) * 4;
    for (int f0 = lane * 4 + part * 256; f0 - lane * 4 < frames; f0 += 256 * wpk) {
        const int fb = f0 - lane * 4;
        const bool act = f0 < frames;
        // the 64 frames before the piece: same block / the block before / the call before — or (3) what this wave kept of the piece it
        // rendered just before this one (`tails`, LDS: [port][64] mono frames)
        const int hmode = tails_valid ? 3 : (fb > 0 ? 0 : (k > 0 ? 1 : 2));
        v4f accl = splat(0.f), accr = splat(0.f);
        for (int p0 = 0; p0 < ld.ports; p0 += SP_U) {
            RawQuad q[SP_U];
            v4f tm[SP_U];  // lanes < 16: the 64 mono frames in front of the piece
#pragma unroll
            for (int u = 0; u < SP_U; ++u) {
                q[u].a = q[u].b = (v4i){0, 0, 0, 0};
                tm[u] = splat(0.f);
            }
            if (act) {
#pragma unroll
                for (int u = 0; u < SP_U; ++u)
                    if (p0 + u < ld.ports)
                        q[u] = sp_raw<CLS>(readlane_ptr(my_l, p0 + u), (uint32_t)__builtin_amdgcn_readlane((int)my_rd, p0 + u), f0);
            }
            if (lane < 16) {
                if (hmode == 3) {
#pragma unroll
                    for (int u = 0; u < SP_U; ++u)
                        if (p0 + u < ld.ports) tm[u] = *(const v4f*)(tails + (p0 + u) * SP_HIST + tq);
                } else if (hmode == 2) {
#pragma unroll
                    for (int u = 0; u < SP_U; ++u)
                        if (p0 + u < ld.ports) tm[u] = gload4(fv.hist + (size_t)(ld.first_voice + p0 + u) * SP_HIST + tq);
                } else {
                    // re-rendered from the source (the wave's FIRST piece only, once it keeps tails): the frames in front of the piece in
                    // the same block, or the last 64 of the block before with that block's gains
#pragma unroll
                    for (int u = 0; u < SP_U; ++u)
                        if (p0 + u < ld.ports) {
                            const int p = p0 + u;
                            RawQuad t;
                            if (hmode == 0) {
                                t = sp_raw<CLS>(readlane_ptr(my_l, p), (uint32_t)__builtin_amdgcn_readlane((int)my_rd, p), fb - SP_HIST + tq);
                            } else {
                                const uint64_t qp = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(plp >> 32), p) << 32) |
                                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)plp, p);
                                t = sp_raw<CLS>((const float*)qp, (uint32_t)__builtin_amdgcn_readlane((int)rdp, p), frames - SP_HIST + tq);
                            }
                            v4f ta, tb;
                            sp_cvt<CLS>(t, ta, tb);
                            if (hmode == 0) {
#pragma unroll
                                for (int j = 0; j < FW_MAX_STAGES; ++j)
                                    if (j <= jn)
                                        apply_stage(((prog << 4) >> (4 * j)) & 15u, splat(readlane_f(my_g.g[j][0], p)), splat(readlane_f(my_g.g[j][1], p)), ta, tb);
                            } else {
#pragma unroll
                                for (int j = 0; j < FW_MAX_STAGES; ++j)
                                    if (j <= jn) apply_stage(((prog << 4) >> (4 * j)) & 15u, splat(readlane_f(gp.g[j][0], p)), splat(readlane_f(gp.g[j][1], p)), ta, tb);
                            }
                            tm[u] = (ta + tb) * 0.5f;
                        }
                }
            }
#pragma unroll
            for (int u = 0; u < SP_U; ++u) {
                if (p0 + u < ld.ports) {
                    const int p = p0 + u;
                    float* row = sp_lds + u * SP_ROW;
                    v4f a, b;
                    sp_cvt<CLS>(q[u], a, b);
#pragma unroll
                    for (int j = 0; j < FW_MAX_STAGES; ++j)
                        if (j <= jn) apply_stage(((prog << 4) >> (4 * j)) & 15u, splat(readlane_f(my_g.g[j][0], p)), splat(readlane_f(my_g.g[j][1], p)), a, b);
                    v4f m = (a + b) * 0.5f;
                    if (!act) m = splat(0.f);
                    if (lane < 16) *(v4f*)(row + tq) = tm[u];
                    *(v4f*)(row + SP_HIST + lane * 4) = m;
                    if (tails != nullptr && lane >= 48) *(v4f*)(tails + p * SP_HIST + (lane - 48) * 4) = m;  // (read above, by lanes < 16, before this)
                    // the last piece of the call's last block leaves the next call's history behind
                    if ((int)k == K - 1 && fb + 256 >= frames && act && f0 >= frames - SP_HIST)
                        *(v4f*)(fv.ext + (size_t)fv.voices[ld.first_voice + p].sp_ext_off + (f0 - (frames - SP_HIST))) = m;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int u = 0; u < SP_U; ++u) {
                if (p0 + u < ld.ports) {
                    const int p = p0 + u;
                    const float* row = sp_lds + u * SP_ROW + SP_HIST + lane * 4;
                    const uint32_t pfg = (uint32_t)__builtin_amdgcn_readlane((int)fg, p);
                    const int dl = (int)((pfg >> VB_SP_SHIFT) & 63u), dr = (int)((pfg >> (VB_SP_SHIFT + 6)) & 63u);
                    v4f ml, mr;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ml[e] = row[e - dl];
                        mr[e] = row[e - dr];
                    }
                    float gl = 1.0f, gr = 1.0f;
#pragma unroll
                    for (int j = 1; j < FW_MAX_STAGES; ++j)
                        if (j == jn + 1) {
                            gl = readlane_f(my_g.g[j][0], p);
                            gr = readlane_f(my_g.g[j][1], p);
                        }
                    const v4f xl = ml * gl, xr = mr * gr;
                    if (p == 0) {  // sum.rs:117 — and a spatialiser's outputs are never flagged silent: no port is skipped
                        accl = xl;
                        accr = xr;
                    } else {
                        accl = accl + xl;
                        accr = accr + xr;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the next batch overwrites the rows)
        }
        if (act) {
            bus_store_pair(outl + f0, outr + f0, accl, accr);
        }
        tails_valid = tails != nullptr && wpk == 1 && fb + 256 <= frames;  // a whole piece, and the next one this wave takes follows it in time
    }
}

// PROG: some voice of the plan has a stage that is not a plain gain (width / hard clip) — lane p then also carries port
// p's stage program.  The gains-only instantiation is the headline kernel and does not pay for the other ones' registers.
// LAZY (round 4, fwgpu_types.h LazyRec): no control kernel ran for this call — lane p computes port p's record from the voice's
// LazyRec (one 128-byte load: descriptor AND gains) and the block index.
template <bool PROG, bool RS = false, int U = LEAF_U, bool SP = false, bool LAZY = false>
__device__ __forceinline__ void leaf_sum_wave(const FusedView& fv, const int leaf, const uint32_t k, const int part, const int wpk,
                                              const RsLds rs = RsLds{nullptr, nullptr}, const int K = 1, float* sp_lds = nullptr, float* sp_tails = nullptr,
                                              bool* sp_tails_valid = nullptr) {
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
    GainSet my_g;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES; ++j) my_g.g[j][0] = my_g.g[j][1] = 1.0f;
#if LEAF_SPEC_GSET
    GainSet g0 = my_g;
    if constexpr (LAZY) {
        if (lane < ld.ports) {
            const LazyRec* lr = fv.lazy + (ld.first_voice + lane);
            const uint64_t base = lr->base, off0 = lr->off0;
            const uint32_t q = lr->q, r0b = lr->r0b, bpf = lr->bpf;
            const int mode = lr->mode;
            ref.r_delta = lr->r_delta;
            ref.flags_gset = lr->flags_gset;
            g0 = lr->g;
            // the block's source address: what steady_tail's lean record would hold — (r0 + block * frames) mod L from the loop
            // start (L a whole number of blocks: no block wraps inside itself), or straight on for a one-shot
            const uint64_t bi = fv.lazy_blk0 + (uint64_t)k;
            if (ref.flags_gset & VB_SIMPLE) {  // (frames into the record's source, then bytes: the classes differ in bytes per frame)
                const uint64_t fo = mode == 1 ? (uint64_t)((r0b + bi) % q) * (uint64_t)(uint32_t)frames : off0 + bi * (uint64_t)(uint32_t)frames;
                ref.src_l = (const float*)(base + fo * bpf);
            }
        }
    } else if (lane < ld.ports) {  // (slot 0 may hold anything on a voice that is not VB_SIMPLE this block: then it is not used)
        g0 = fv.gsets[(size_t)(ld.first_voice + lane) * FW_GSETS];
        ref = fv.refs[ref_index(ld.first_voice + lane, (int)k, fv.ref_kgroups)];
    }
    const uint32_t my_flags = ref.flags_gset & 0xffu;
    if (my_flags & VB_SIMPLE) {
        const uint32_t gi = LAZY ? 0u : (ref.flags_gset >> 8) & 0xffu;
        my_g = g0;
        if (gi != 0) my_g = fv.gsets[(size_t)(ld.first_voice + lane) * FW_GSETS + gi];
    }
#else
    // (the LAZY derivation is written for the LEAF_SPEC_GSET layout only: without it this branch would read refs / gsets an old
    //  control run left behind while the host skips the control kernel — ADVICE r4)
    static_assert(!LAZY, "k_leaf_sum_lazy needs LEAF_SPEC_GSET=1");
    if (lane < ld.ports) ref = fv.refs[ref_index(ld.first_voice + lane, (int)k, fv.ref_kgroups)];
    const uint32_t my_flags = ref.flags_gset & 0xffu;
    if (my_flags & VB_SIMPLE) my_g = fv.gsets[(size_t)(ld.first_voice + lane) * FW_GSETS + ((ref.flags_gset >> 8) & 0xffu)];
#endif
    uint32_t my_prog = 0u;
    if constexpr (PROG) {
        if (lane < ld.ports) my_prog = fv.progs[ld.first_voice + lane];
        asm volatile("" : "+v"(my_prog));  // (read with v_readlane inside the frame loop: see below)
    }
    const float* my_l = ref.src_l;
    uint32_t my_rd = ref.r_delta;
    uint32_t my_cls = (ref.flags_gset >> 16) & 7u;
    const float* my_r = ref.src_l + ref.r_delta;  // planar f32 class only
    // Everything lane p holds for port p is read below with v_readlane from inside the frame loop, where only the lanes
    // that own frames are active (block < 256 frames: fewer than 64).  Pin the values HERE, under the full exec mask:
    // left to itself the compiler may sink their computation into the loop, and lanes that are inactive there would
    // never compute them (found by the fuzz test: 19-port leaf, block of 64 frames -> garbage source addresses).
    {
        uint64_t pl = (uint64_t)my_l, pr = (uint64_t)my_r;
        uint32_t rd = my_rd, cl = my_cls;
        asm volatile("" : "+v"(pl), "+v"(pr), "+v"(rd), "+v"(cl));
        my_l = (const float*)pl;
        my_r = (const float*)pr;
        my_rd = rd;
        my_cls = cl;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) asm volatile("" : "+v"(my_g.g[j][0]), "+v"(my_g.g[j][1]));
    }
    if constexpr (SP) {
        // a leaf with a spatialiser voice among its ports goes port by port (leaf_sp below); the other leaves of the plan take
        // the batched paths as before
        bool my_sp = false;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES - 1; ++j) my_sp = my_sp || ((my_prog >> (4 * j)) & 15u) == SK_SPATIAL;
        if (__ballot(my_sp && lane < ld.ports)) {
            // lane p also keeps port p's record of the block BEFORE (the history of the wave's first piece is re-rendered from it)
            VoiceRef refp;
            refp.src_l = nullptr;
            refp.r_delta = 0;
            refp.flags_gset = VB_SRC_ZERO;
            GainSet gp = my_g;
            // (the wave rendered the block before this one itself, on the batched path, and kept its tails: no record of that block needed
            //  — IF this block takes the batched path too: a block with a wrap, a ramp or a silent port goes port by port below and
            //  re-renders its history from the record of the block before, tails or not)
            bool tv = sp_tails_valid != nullptr && *sp_tails_valid;
            if (sp_tails_valid != nullptr) *sp_tails_valid = false;
            if (tv) {
                const uint32_t cls_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_cls);
                const uint32_t prog_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_prog);
                const bool now_a = (ref.flags_gset & (VB_SIMPLE | VB_SRC_ZERO | VB_SILENT)) == VB_SIMPLE && my_cls == cls_a && my_prog == prog_a && my_sp;
                const uint64_t lin_a = mask_all_silent_bits(ld.ports);
                tv = (__ballot(now_a) & lin_a) == lin_a;
            }
            if (k > 0 && lane < ld.ports && my_sp && !tv) {
                refp = fv.refs[ref_index(ld.first_voice + lane, (int)k - 1, fv.ref_kgroups)];
                if (refp.flags_gset & VB_SIMPLE) gp = fv.gsets[(size_t)(ld.first_voice + lane) * FW_GSETS + ((refp.flags_gset >> 8) & 0xffu)];
            }
            uint32_t fg = ref.flags_gset, fgp = refp.flags_gset, rdp = refp.r_delta;
            uint64_t plp = (uint64_t)refp.src_l;
            asm volatile("" : "+v"(fg), "+v"(fgp), "+v"(rdp), "+v"(plp));
#pragma unroll
            for (int j = 0; j < FW_MAX_STAGES; ++j) asm volatile("" : "+v"(gp.g[j][0]), "+v"(gp.g[j][1]));
            {
                // steady leaf: every port a spatialiser voice, VB_SIMPLE with a live source in this block (and in the block before, whose
                // last 64 frames are the history), one source class, one stage program -> the batched path
                const uint64_t lin = mask_all_silent_bits(ld.ports);
                const uint32_t cls0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_cls);
                const uint32_t prog0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_prog);
                const bool ok_now = (fg & (VB_SIMPLE | VB_SRC_ZERO | VB_SILENT)) == VB_SIMPLE && my_cls == cls0 && my_prog == prog0 && my_sp;
                const bool ok_prev = k == 0 || tv || ((fgp & (VB_SIMPLE | VB_SRC_ZERO)) == VB_SIMPLE && ((fgp >> 16) & 7u) == cls0);
                if ((__ballot(ok_now && ok_prev) & lin) == lin) {
                    int jn0 = 0;
#pragma unroll
                    for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
                        if (((prog0 >> (4 * j)) & 15u) == SK_SPATIAL) jn0 = j;
#define SP_FAST(C) leaf_sp_fast<C>(fv, ld, k, K, part, wpk, prog0, jn0, my_l, my_rd, my_g, fg, plp, rdp, gp, sp_lds, outl, outr, sp_tails, tv)
                    switch (cls0) {
                        case SF_P_F32: SP_FAST(SF_P_F32); break;
                        case SF_P_I16: SP_FAST(SF_P_I16); break;
                        case SF_P_U16: SP_FAST(SF_P_U16); break;
                        case SF_I_I16: SP_FAST(SF_I_I16); break;
                        case SF_I_U16: SP_FAST(SF_I_U16); break;
                        default: SP_FAST(SF_I_F32); break;
                    }
#undef SP_FAST
                    if (sp_tails_valid != nullptr) *sp_tails_valid = tv;
                    if (lane < 2 && part == 0) bflags[ld.out_buf + lane] = 0;
                    return;
                }
            }
            const int path_ports = ld.pad ? ld.pad : ld.ports;
            const bool masked = !(path_ports == 2 || path_ports == 3 || path_ports == 4);  // sum.rs:67-133 (Q13)
            for (int f0 = lane * 4 + part * 256; f0 - lane * 4 < frames; f0 += 256 * wpk) {
                const int fb = f0 - lane * 4;  // first frame of the wave's piece
                const bool act = f0 < frames;
                v4f accl = splat(0.f), accr = splat(0.f);
                bool any_live = false;
                for (int p = 0; p < ld.ports; ++p) {
                    const uint32_t pfg = (uint32_t)__builtin_amdgcn_readlane((int)fg, p);
                    const uint32_t prog = (uint32_t)__builtin_amdgcn_readlane((int)my_prog, p);
                    const int voice = ld.first_voice + p;
                    const bool psil = (pfg & VB_SILENT) != 0;
                    int jn = -1;  // chain-stage index of the spatialiser
#pragma unroll
                    for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
                        if (((prog >> (4 * j)) & 15u) == SK_SPATIAL) jn = j;
                    v4f xl = splat(0.f), xr = splat(0.f);
                    if (jn < 0) {  // an ordinary voice under the same mixer
                        if (!psil && act) {
                            if (pfg & VB_SIMPLE) {
                                simple_fetch((pfg >> 16) & 7u, readlane_ptr(my_l, p), (uint32_t)__builtin_amdgcn_readlane((int)my_rd, p), f0, xl, xr);
#pragma unroll
                                for (int j = 0; j < FW_MAX_STAGES; ++j)
                                    if (j < fv.n_gain_stages)
                                        apply_stage(((prog << 4) >> (4 * j)) & 15u, splat(readlane_f(my_g.g[j][0], p)), splat(readlane_f(my_g.g[j][1], p)), xl, xr);
                            } else {
                                const VoiceBlk d = blk_load<RS>(fv, k, voice, pfg, readlane_ptr(my_l, p));
                                voice_eval<RS>(fv, d, k, voice, f0, frames, xl, xr, prog, rs);
                            }
                        }
                    } else {
                        // what reaches the spatialiser: this piece ...
                        v4f a, b;
                        sp_upstream<RS>(fv, pfg, readlane_ptr(my_l, p), (uint32_t)__builtin_amdgcn_readlane((int)my_rd, p), &my_g, p, k, voice, f0, frames,
                                        prog, jn, act, a, b);
                        const v4f m = (a + b) * 0.5f;
                        // ... and the 64 frames in front of it (every group of 16 lanes renders the same 16 quads)
                        v4f tm;
                        const int tq = (lane & 15) * 4;
                        if (fb > 0) {
                            v4f ta, tb;
                            sp_upstream<RS>(fv, pfg, readlane_ptr(my_l, p), (uint32_t)__builtin_amdgcn_readlane((int)my_rd, p), &my_g, p, k, voice,
                                            fb - SP_HIST + tq, frames, prog, jn, true, ta, tb);
                            tm = (ta + tb) * 0.5f;
                        } else if (k > 0) {
                            const uint32_t qfg = (uint32_t)__builtin_amdgcn_readlane((int)fgp, p);
                            const uint64_t qp = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(plp >> 32), p) << 32) |
                                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)plp, p);
                            v4f ta, tb;
                            sp_upstream<RS>(fv, qfg, (const float*)qp, (uint32_t)__builtin_amdgcn_readlane((int)rdp, p), &gp, p, k - 1, voice,
                                            frames - SP_HIST + tq, frames, prog, jn, true, ta, tb);
                            tm = (ta + tb) * 0.5f;
                        } else {
                            tm = *(const v4f*)(fv.hist + (size_t)voice * SP_HIST + tq);
                        }
                        if (lane < 16) *(v4f*)(sp_lds + tq) = tm;
                        *(v4f*)(sp_lds + SP_HIST + lane * 4) = m;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const int dl = (int)((pfg >> VB_SP_SHIFT) & 63u), dr = (int)((pfg >> (VB_SP_SHIFT + 6)) & 63u);
                        v4f ml, mr;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ml[e] = sp_lds[SP_HIST + lane * 4 + e - dl];
                            mr[e] = sp_lds[SP_HIST + lane * 4 + e - dr];
                        }
                        // the ear gains: constants of the gain set, or the block's ramps (full descriptor)
                        v4f gl, gr;
                        if (pfg & VB_SIMPLE) {
                            gl = gr = splat(1.0f);
#pragma unroll
                            for (int j = 1; j < FW_MAX_STAGES; ++j)
                                if (j == jn + 1) {
                                    gl = splat(readlane_f(my_g.g[j][0], p));
                                    gr = splat(readlane_f(my_g.g[j][1], p));
                                }
                        } else {
                            const VoiceBlk d = fv.blks[row + p];
                            const uint32_t rbits = (d.flags & VB_RAMP_MASK) >> VB_RAMP_SHIFT;
                            const float* rb = fv.ramps + ((size_t)k * fv.n_voices + voice) * (size_t)fv.ramp_slots * (size_t)fv.stride + (act ? f0 : 0);
                            gl = gr = splat(1.0f);
#pragma unroll
                            for (int j = 1; j < FW_MAX_STAGES; ++j)
                                if (j == jn + 1) {
                                    gl = (rbits >> (2 * j)) & 1u ? *(const v4f*)(rb + (size_t)(2 * j) * fv.stride) : splat(d.g[j][0]);
                                    gr = (rbits >> (2 * j + 1)) & 1u ? *(const v4f*)(rb + (size_t)(2 * j + 1) * fv.stride) : splat(d.g[j][1]);
                                }
                        }
                        xl = ml * gl;
                        xr = mr * gr;
                        // the last piece of the call's last block leaves the next call's history behind
                        if ((int)k == K - 1 && fb + 256 >= frames && act && f0 >= frames - SP_HIST)
                            *(v4f*)(fv.ext + (size_t)fv.voices[voice].sp_ext_off + (f0 - (frames - SP_HIST))) = m;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();  // (the next port overwrites the row)
                    }
                    if (p == 0) {  // sum.rs:117 copy_from_slice(port 0) — also when silent; 2/3/4-port: in1
                        accl = xl;
                        accr = xr;
                    } else if (!(masked && psil)) {  // :122-124 skip silent ports (n-port path only)
                        accl = accl + xl;
                        accr = accr + xr;
                    }
                    any_live = any_live || !psil;
                }
                if (act) {
                    bus_store_pair(outl + f0, outr + f0, accl, accr);
                }
            }
            if (lane < 2 && part == 0) bflags[ld.out_buf + lane] = 0;  // a spatialiser port is never silent: the mixer's out mask is 0
            return;
        }
    }
    const uint64_t lanes_in = mask_all_silent_bits(ld.ports);
    const uint64_t silent_ports = __ballot((my_flags & VB_SILENT) != 0) & lanes_in;
    const uint64_t simple_ports = __ballot((my_flags & VB_SIMPLE) != 0) & lanes_in;
    const bool all_silent = silent_ports == lanes_in;
    // resampler ports: lane p takes what the fetch of port p needs out of its full descriptor — sample data and length, 32.32
    // position and step, flags, the constant gains — and says whether the staged path can render it (leaf_rs_piece)
    uint64_t rs_ports = 0ull, my_off0 = 0ull, my_off1 = 0ull;
    uint32_t my_df = 0u;
    const uint64_t lean_ports = RS ? __ballot((my_flags & VB_RS_LEAN) != 0) & lanes_in : 0ull;
    if constexpr (RS) {
        bool el = false;
        if (rs.tab != nullptr && lane < ld.ports && !(my_flags & (VB_SILENT | VB_SIMPLE))) {
            const bool lean = (my_flags & VB_RS_LEAN) != 0;
            const VoiceBlk* b = lean ? fv.rs_tmpl + (ld.first_voice + lane) : fv.blks + row + lane;
            const uint32_t df = b->flags;
            if (df & VB_RESAMPLE) {
                const uint32_t len = b->pad;
                const uint64_t off0 = lean ? (uint64_t)ref.src_l : b->off0, step = b->off1;
                const bool loop = b->n1 != 0;
                const int nfr = frames < 256 ? frames : 256;
                const uint64_t w_max = (((uint64_t)nfr * step + 0xffffffffull) >> 32) + 1 + RS_TAPS;  // any piece of the block
                const uint64_t i_end = (off0 + (uint64_t)frames * step) >> 32;
                el = ((df >> VB_FMT_SHIFT) & 7u) == (uint32_t)FMT_P_F32 && w_max <= RS_WIN && (step >> 32) < 8 && i_end < (1ull << 30) && len < (1u << 30) &&
                     len >= 1u && (!loop || len >= (uint32_t)RS_WIN + RS_TAPS);
                if (el) {
                    my_l = b->src_l;
                    my_rd = len;
                    my_off0 = off0;
                    my_off1 = step;
                    my_df = (df & ~RS_LOOP_BIT) | (loop ? RS_LOOP_BIT : 0u);
#pragma unroll
                    for (int j = 0; j < FW_MAX_STAGES; ++j) {
                        my_g.g[j][0] = b->g[j][0];
                        my_g.g[j][1] = b->g[j][1];
                    }
                }
            }
        }
        rs_ports = __ballot(el) & lanes_in;
        {  // (read with v_readlane inside the frame loop: pinned under the full exec mask, as above)
            uint64_t pl = (uint64_t)my_l;
            uint32_t o0l = (uint32_t)my_off0, o0h = (uint32_t)(my_off0 >> 32), o1l = (uint32_t)my_off1, o1h = (uint32_t)(my_off1 >> 32);
            asm volatile("" : "+v"(pl), "+v"(my_rd), "+v"(o0l), "+v"(o0h), "+v"(o1l), "+v"(o1h), "+v"(my_df));
            my_l = (const float*)pl;
            my_off0 = ((uint64_t)o0h << 32) | o0l;
            my_off1 = ((uint64_t)o1h << 32) | o1l;
#pragma unroll
            for (int j = 0; j < FW_MAX_STAGES; ++j) asm volatile("" : "+v"(my_g.g[j][0]), "+v"(my_g.g[j][1]));
        }
    }
    const int path_ports = ld.pad ? ld.pad : ld.ports;  // (a leaf that is the leading voice ports of a wider SumNode takes ITS path)
    const bool masked = !(path_ports == 2 || path_ports == 3 || path_ports == 4);  // sum.rs:67-133 (Q13)
    const bool all_simple = simple_ports == lanes_in && (frames & 3) == 0;
    const uint32_t cls0 = (uint32_t)__builtin_amdgcn_readlane((int)my_cls, 0);  // ports >= 1
    const bool one_class = (__ballot(my_cls == cls0) & lanes_in) == lanes_in;
    const bool fast_cls = !PROG && all_simple && one_class && cls0 != SF_P_F32;  // (program voices on other formats: port by port)
    // ports of a mixed leaf whose source loads are batched: VB_SIMPLE (so frames % 4 == 0, not silent) planar f32
    const uint64_t batch_ports = (frames & 3) == 0 ? simple_ports & ~silent_ports & __ballot(my_cls == SF_P_F32) : 0ull;

    const int f_first = lane * 4 + part * 256;
    for (int f0 = f_first; f0 < frames; f0 += 256 * wpk) {
        v4f accl = splat(0.f), accr = splat(0.f);
        bool rs_done = false;
        if constexpr (RS) {
            if (rs_ports) {
                leaf_rs_piece<PROG, RS>(fv, ld, k, row, f0, frames, rs, my_l, my_rd, my_cls, my_g, my_prog, my_off0, my_off1, my_df, silent_ports,
                                        simple_ports, rs_ports, lean_ports, masked, accl, accr);
                rs_done = true;
            }
        }
        if (rs_done) {
        } else if (fast_cls) {
            const int ng = fv.n_gain_stages;
            switch (cls0) {
                case SF_P_I16: leaf_fast_cls<SF_P_I16>(my_l, my_rd, my_g, ng, ld.ports, f0, accl, accr); break;
                case SF_P_U16: leaf_fast_cls<SF_P_U16>(my_l, my_rd, my_g, ng, ld.ports, f0, accl, accr); break;
                case SF_I_I16: leaf_fast_cls<SF_I_I16>(my_l, my_rd, my_g, ng, ld.ports, f0, accl, accr); break;
                case SF_I_U16: leaf_fast_cls<SF_I_U16>(my_l, my_rd, my_g, ng, ld.ports, f0, accl, accr); break;
                default: leaf_fast_cls<SF_I_F32>(my_l, my_rd, my_g, ng, ld.ports, f0, accl, accr); break;
            }
        } else if (!all_silent) {
            // The ports are added in port order.  A RUN of plain ports (VB_SIMPLE planar f32 — on a steady bank the whole
            // leaf is one run) has its source loads issued LEAF_U ports at a time; a port that ramps, wraps, is silent or of
            // another source class is evaluated on its own, between two runs.  (A leaf with ONE gliding voice used to go
            // port by port altogether: 64 dependent memory round trips in a row made such a wave ~60 us long, and the 1 500
            // of them in a step of config 2's variant B — 68 gliding voices x 21 blocks — stretched the render kernel by 13 %.)
            int p = 0;
            while (p < ld.ports) {
                const uint64_t rest = ~(batch_ports >> p);
                int run = rest ? __builtin_ctzll(rest) : 64;
                run = run < ld.ports - p ? run : ld.ports - p;
                if (run > 0) {
                    if constexpr (PROG) {
                        leaf_fast_prog<U>(my_l, my_r, my_g, my_prog, fv.n_gain_stages, p, p + run, f0, accl, accr);
                    } else {
                        switch (fv.n_gain_stages) {
                            case 1: leaf_fast<1, U>(my_l, my_r, my_g, p, p + run, f0, accl, accr); break;
                            case 2: leaf_fast<2, U>(my_l, my_r, my_g, p, p + run, f0, accl, accr); break;
                            case 3: leaf_fast<3, U>(my_l, my_r, my_g, p, p + run, f0, accl, accr); break;
                            case 4: leaf_fast<4, U>(my_l, my_r, my_g, p, p + run, f0, accl, accr); break;
                            case 5: leaf_fast<5, U>(my_l, my_r, my_g, p, p + run, f0, accl, accr); break;
                            default: leaf_fast<6, U>(my_l, my_r, my_g, p, p + run, f0, accl, accr); break;
                        }
                    }
                    p += run;
                    continue;
                }
                const bool psil = (silent_ports >> p) & 1ull;
                v4f xl = splat(0.f), xr = splat(0.f);  // a silent chain's buffers hold cleared zeros
                if (!psil) {
                    const uint32_t prog = PROG ? (uint32_t)__builtin_amdgcn_readlane((int)my_prog, p) : 0u;
                    if ((simple_ports >> p) & 1ull) {  // VB_SIMPLE implies frames % 4 == 0
                        simple_fetch((uint32_t)__builtin_amdgcn_readlane((int)my_cls, p), readlane_ptr(my_l, p),
                                     (uint32_t)__builtin_amdgcn_readlane((int)my_rd, p), f0, xl, xr);
#pragma unroll
                        for (int j = 0; j < FW_MAX_STAGES; ++j)
                            if (j < fv.n_gain_stages)
                                apply_stage(((prog << 4) >> (4 * j)) & 15u, splat(readlane_f(my_g.g[j][0], p)), splat(readlane_f(my_g.g[j][1], p)),
                                            xl, xr);
                    } else {
                        const VoiceBlk d = blk_load<RS>(fv, k, ld.first_voice + p, (lean_ports >> p) & 1ull ? VB_RS_LEAN : 0u, readlane_ptr(my_l, p));
                        voice_eval<RS>(fv, d, k, ld.first_voice + p, f0, frames, xl, xr, prog, rs);
                    }
                }
                if (p == 0) {  // sum.rs:117 copy_from_slice(port 0) — also when silent; 2/3/4-port: in1
                    accl = xl;
                    accr = xr;
                } else if (!(masked && psil)) {  // :122-124 skip silent ports (n-port path only)
                    accl = accl + xl;
                    accr = accr + xr;
                }
                ++p;
            }
        }
#if LEAF_NT_STORE == 2
        bus_store_pair(outl + f0, outr + f0, accl, accr);
#elif LEAF_NT_STORE
        __builtin_nontemporal_store(accl, (v4f*)(outl + f0));
        __builtin_nontemporal_store(accr, (v4f*)(outr + f0));
#else
        *(v4f*)(outl + f0) = accl;  // all_silent: clear_all_outputs (sum.rs:52-56)
        *(v4f*)(outr + f0) = accr;
#endif
    }
    // out mask: all-silent -> both flagged; 1-port copy -> passthrough (sum.rs:58-65); else 0
    if (lane < 2 && part == 0) bflags[ld.out_buf + lane] = all_silent ? 1 : 0;
}

// dynamic LDS of the program instantiations (launched with 0 bytes when the plan has no resampler voice): filter bank, then
// one source window per wave
__device__ __forceinline__ RsLds rs_lds_setup(const FusedView& fv, float* dyn) {
    RsLds rs{nullptr, nullptr};
    if (fv.has_rs) {  // (uniform: every wave of the workgroup comes through here before anything can return)
        for (int i = threadIdx.x; i < RS_PHASES * RS_TAPS; i += blockDim.x) {  // [tap pair][phase][2]: one ds_read_b64 = two taps of a phase
            const int t = i % RS_TAPS, ph = i / RS_TAPS;
            dyn[((t >> 1) * RS_PHASES + ph) * 2 + (t & 1)] = fv.rs_table[i];
        }
        __syncthreads();
        rs.tab = dyn;
        rs.win = dyn + RS_PHASES * RS_TAPS + (threadIdx.x >> 6) * (2 * RS_WIN + 512);
    }
    return rs;
}
#define RS_LDS_BYTES(waves) ((RS_PHASES * RS_TAPS + (waves) * (2 * RS_WIN + 512)) * sizeof(float))

// Three instantiations, by what the plan's voices need (register appetite: 94 / ~105 / ~140 VGPRs):
//   <false, false>  every stage a plain gain — the headline kernel
//   <true,  false>  stage programs (width / hard clip)
//   <true,  true>   ... and voices whose source is a resampler (LDS-staged polyphase fetch)
//   <true,  false, true>  ... and voices that end in a spatialiser (a 320-float mono row per wave in LDS)
template <bool PROG, bool RS, bool SP, bool LAZY = false>
__device__ __forceinline__ void leaf_kernel_body(const FusedView& fv, const int K, const int wpk, const int sp_nb = 1) {
    extern __shared__ float s_leaf_dyn[];
    RsLds rs{nullptr, nullptr};
    if constexpr (RS) rs = rs_lds_setup(fv, s_leaf_dyn);
    __shared__ float s_sp[SP ? LEAF_WPB * SP_U * SP_ROW : 1];
    float* sp_lds = SP ? s_sp + (threadIdx.x >> 6) * (SP_U * SP_ROW) : nullptr;
    if constexpr (SP && !RS) {
        // Spatialiser plan, blocks of one piece per wave: a wave takes sp_nb CONSECUTIVE blocks of its leaf and keeps every port's last 64
        // mono frames in LDS from one to the next — the history of all but its first block costs no HBM read and no stage arithmetic
        // (round 3 re-fetched and re-rendered it per block: traffic 1.15x the algorithmic bytes).
        __shared__ float s_tails[LEAF_WPB * 32 * SP_HIST];
        if (wpk == 1) {
            const int leaf = blockIdx.x;
            const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.y * LEAF_WPB + wave) * sp_nb));
            bool tv = false;
            for (uint32_t kk = 0; kk < (uint32_t)sp_nb && k0 + kk < (uint32_t)K; ++kk) {
                int leaf_k = leaf;  // (opaque per iteration: what a block reads per lane — gain sets, programs — is loaded per block, not
                asm volatile("" : "+s"(leaf_k));  // hoisted out of the loop into a dozen registers that live across all of it)
                leaf_sum_wave<PROG, RS, LEAF_U, SP, LAZY>(fv, leaf_k, k0 + kk, 0, 1, rs, K, sp_lds, s_tails + wave * (32 * SP_HIST), &tv);
            }
            return;
        }
    }
#if LEAF_MAP_BLOCKS
    // the waves of a workgroup take CONSECUTIVE 256-frame pieces of one leaf's stream — wpk (1, 2 or 4) waves per
    // block, LEAF_WPB / wpk consecutive blocks: a steady voice's source is contiguous across blocks, so the workgroup
    // streams LEAF_WPB adjacent KiB per voice-channel instead of 1 KiB from LEAF_WPB x 32 places
    const int leaf = blockIdx.x;
    const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int part = wave & (wpk - 1);
    const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.y * (LEAF_WPB / wpk) + wave / wpk));
    if (k >= (uint32_t)K) return;
#else
    const int part = 0;
    const int leaf = blockIdx.x * LEAF_WPB + (threadIdx.x >> 6);
    if (leaf >= fv.n_leaves) return;
    const uint32_t k = blockIdx.y;
#endif
    leaf_sum_wave<PROG, RS, LEAF_U, SP, LAZY>(fv, leaf, k, part, wpk, rs, K, sp_lds);
}
template <bool PROG, bool RS, bool SP = false>
__global__ __launch_bounds__(WAVE* LEAF_WPB) void k_leaf_sum(FusedView fv, int K, int wpk) {
    leaf_kernel_body<PROG, RS, SP>(fv, K, wpk);
}
// the spatialiser instantiation under its own occupancy target: the default heuristic settles at 172 VGPRs, four over
// the three-waves-per-SIMD line; the kernel is latency-bound (one HBM round trip per SP_U ports, then LDS), so the third wave pays
#ifndef SP_OCC
#define SP_OCC 3
#endif
__global__ __launch_bounds__(WAVE* LEAF_WPB, SP_OCC) void k_leaf_sum_sp(FusedView fv, int K, int wpk, int sp_nb) {
    leaf_kernel_body<true, false, true>(fv, K, wpk, sp_nb);
}
// the same kernel for a call no control kernel ran for (plans without resampler sources / spatialiser stages)
template <bool PROG>
__global__ __launch_bounds__(WAVE* LEAF_WPB) void k_leaf_sum_lazy(FusedView fv, int K, int wpk) {
    leaf_kernel_body<PROG, false, false, true>(fv, K, wpk);
}
// The resampler-pure leaves of a plan (rs_pure_lane above).  One wave per (leaf, block, 256-frame piece); lane p holds port p.
// Accumulators, stages and the final store stay in the convolution's round-robin frame layout (lane l: frames l, l + nact, ...):
// the stages of a pure voice are per-frame constants, so nothing needs the four-consecutive-frames layout of the other kernels.
// LDS: the filter bank shifted by one tap [9 tap pairs][phase][2] = (h[2tp - 1], h[2tp]) (the pair convolution's second frame), the filter
// bank [8 tap pairs][phase][2] = (h[2tp], h[2tp + 1]), one window per wave (+ 2 slots the pair convolution may read past the last one)
#define RS2_TAB1 (RS_PHASES * (RS_TAPS + 2))
#define RS2_LDS_BYTES(waves) ((RS2_TAB1 + RS_PHASES * RS_TAPS + (waves) * (2 * RS2_WIN) + 4) * sizeof(float))
#ifndef RS_OCC
#define RS_OCC 4
#endif
__global__ __launch_bounds__(WAVE* LEAF_WPB, RS_OCC) void k_leaf_rs(FusedView fv, int K, int wpk) {
    extern __shared__ float s_leaf_dyn[];
    for (int i = threadIdx.x; i < RS_PHASES * RS_TAPS; i += blockDim.x) {  // [tap pair][phase][2]
        const int t = i % RS_TAPS, ph = i / RS_TAPS;
        s_leaf_dyn[RS2_TAB1 + ((t >> 1) * RS_PHASES + ph) * 2 + (t & 1)] = fv.rs_table[i];
    }
    for (int i = threadIdx.x; i < RS2_TAB1; i += blockDim.x) {  // the same, one tap later: [tap pair][phase][2] = taps 2tp - 1, 2tp (taps -1 and 16: never used)
        const int e = i & 1, ph = (i >> 1) % RS_PHASES, tp = (i >> 1) / RS_PHASES, t = 2 * tp - 1 + e;
        s_leaf_dyn[i] = (t >= 0 && t < RS_TAPS) ? fv.rs_table[ph * RS_TAPS + t] : 0.f;
    }
    __syncthreads();
    const int leaf = blockIdx.x;
    const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int part = wave & (wpk - 1);
    const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.y * (LEAF_WPB / wpk) + wave / wpk));
    if (k >= (uint32_t)K) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const LeafDesc ld = fv.leaves[leaf];
    const int frames = fv.frames;
    if (part * 256 >= frames) return;
    const v2f_rs* tab = (const v2f_rs*)(s_leaf_dyn + RS2_TAB1);
    const v2f_rs* tab1 = (const v2f_rs*)s_leaf_dyn;  // (in FRONT of the bank: the pair convolution reads one tap pair below `tab` and discards it)
    v2f_rs* win = (v2f_rs*)(s_leaf_dyn + RS2_TAB1 + RS_PHASES * RS_TAPS + wave * (2 * RS2_WIN));
    const size_t row = (size_t)k * fv.n_voices + ld.first_voice;
    uint32_t my_flags = VB_SILENT;
    uint64_t my_pos = 0ull;
    if (lane < ld.ports) {
        if (fv.lazy_rs) {
            // Round 6 — a lazy call (no control kernel ran): the block's record from the voice's LazyRec and the absolute block index —
            // a resampler voice's 32.32 position (pos + j * frames * step) mod (len << 32), its template in rs_tmpl (= the LazyRecs'
            // copy); anything else in such a plan is silent (k_control.hip.h make_lazy)
            const LazyRec* lr = fv.lazy + (ld.first_voice + lane);
            my_flags = lr->flags_gset & 0xffu;
            if (lr->mode == 3) {
                const uint64_t M = lr->base;
                const uint64_t pos = lr->off0 + (fv.lazy_blk0 + (uint64_t)k) * ((uint64_t)lr->frames * lr->loop_start);
                my_pos = M ? pos % M : pos;
            }
        } else {
            const VoiceRef ref = fv.refs[ref_index(ld.first_voice + lane, (int)k, fv.ref_kgroups)];
            my_flags = ref.flags_gset & 0xffu;
            my_pos = (uint64_t)ref.src_l;
        }
    }
    const uint64_t lanes_in = mask_all_silent_bits(ld.ports);
    const uint64_t silent_ports = __ballot((my_flags & VB_SILENT) != 0) & lanes_in;
    if (fv.lazy_rs && silent_ports == lanes_in) {
        // every port silent: clear_all_outputs (sum.rs:52-56) — the work-list kernel's job on a control call; its records do not exist here
        float* const bus0 = fv.bus + (size_t)k * fv.bus_blk_stride + (size_t)ld.out_buf * fv.stride;
        for (int f = part * 256 + lane * 4; f < fv.frames; f += 256 * wpk) {  // (this wave's 256-frame pieces; rows are padded to 64 frames)
            *(v4f*)(bus0 + f) = splat(0.f);
            *(v4f*)(bus0 + fv.stride + f) = splat(0.f);
        }
        if (lane < 2 && part == 0) (fv.bus_flags + (size_t)k * fv.bus_flags_blk_stride)[ld.out_buf + lane] = 1;
        return;
    }
    const RsPure me = rs_pure_lane(fv, row, lane, ld.ports, my_flags, frames, ld.first_voice + lane, my_pos);
    const uint64_t pure_ports = __ballot(me.pure) & lanes_in;
    if (!(pure_ports && ((pure_ports | silent_ports) & lanes_in) == lanes_in)) {  // not ours: onto the general kernel's work list
        if (lane == 0) {
            const unsigned int i = atomicAdd(fv.rs_wl, 1u);
            fv.rs_wl[2 + 2 * i] = (unsigned int)leaf;
            fv.rs_wl[3 + 2 * i] = k * 4u + (unsigned int)part;
        }
        return;
    }
    // lane p holds port p: sample, position, step, flags, stage program, constant gains (read with v_readlane below)
    const uint64_t my_s0 = (uint64_t)me.s0, my_off0 = me.off0, my_step = me.step;
    const uint32_t my_len = me.len, my_df = me.df;
    const int my_qb0 = me.qb0;
    uint32_t my_prog = 0u;
    float my_g[FW_MAX_STAGES][2];
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES; ++j) my_g[j][0] = my_g[j][1] = 1.0f;
    if (me.pure) {
        my_prog = fv.progs[ld.first_voice + lane];
        const VoiceBlk* b = (my_flags & VB_RS_LEAN) ? fv.rs_tmpl + (ld.first_voice + lane) : fv.blks + row + lane;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) {
            my_g[j][0] = b->g[j][0];
            my_g[j][1] = b->g[j][1];
        }
    }
    const int path_ports = ld.pad ? ld.pad : ld.ports;
    const bool masked = !(path_ports == 2 || path_ports == 3 || path_ports == 4);  // sum.rs:67-133 (Q13)
    const int ng = fv.n_gain_stages;
    float* bus = fv.bus + (size_t)k * fv.bus_blk_stride;
    float* outl = bus + (size_t)ld.out_buf * fv.stride;
    float* outr = outl + fv.stride;
    const int first = __builtin_ctzll(pure_ports);

    for (int fbase = part * 256; fbase < frames; fbase += 256 * wpk) {
        const int nfr = frames - fbase < 256 ? frames - fbase : 256;
        // (every lane works here — window fetch and convolution are dealt over the whole wave, whatever the piece's length)
        const int nact = nfr < WAVE ? nfr : WAVE;                       // lanes that own frames
        const int per = (nfr + nact - 1) / nact;                        // frames per lane: <= 4
        const bool pairs = nfr == 256 && !(fv.dbg & 512);               // (frames 2l, 2l+1, 128+2l, 128+2l+1 instead; FWGPU_CHAIN_SKIP=512: A/B)
        v4f accl = splat(0.f), accr = splat(0.f);                       // frames lane, lane + nact, lane + 2 nact, lane + 3 nact
        v4f wa[2], wb[2];
        wa[0] = wa[1] = wb[0] = wb[1] = splat(0.f);
        // one port's piece: source, window length, flags (wave-uniform; the NEXT port's set is computed once, when its window is
        // requested, and becomes the current one an iteration later)
        struct Rs2Port {
            uint64_t step, p_first;
            rs_gfp s0;
            int len, qb, W;
            uint32_t df;
        };
        auto port = [&](const int p) {
            Rs2Port P;
            P.step = readlane_u64(my_step, p);
            const uint64_t o0 = readlane_u64(my_off0, p);
            P.p_first = o0 + (uint64_t)fbase * P.step;
            P.len = __builtin_amdgcn_readlane((int)my_len, p);
            P.df = (uint32_t)__builtin_amdgcn_readlane((int)my_df, p);
            P.qb = __builtin_amdgcn_readlane(my_qb0, p) + (int)((uint32_t)(P.p_first >> 32) - (uint32_t)(o0 >> 32));
            if (P.df & RS_LOOP_BIT)
                while (P.qb >= P.len) P.qb -= P.len;
            P.s0 = (rs_gfp)readlane_u64(my_s0, p);
            P.W = (int)((uint32_t)((P.p_first + (uint64_t)(nfr - 1) * P.step) >> 32) - (uint32_t)(P.p_first >> 32)) + RS_TAPS;
            return P;
        };
        // a contiguous window is requested as quads: two rounds of registers per channel
#define RS2_ISSUE(P)                                                                   \
    if (P.df & RS_CONTIG_BIT) {                                                        \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                \
            const int r = (lane + u * WAVE) * 4;                                       \
            if (u * WAVE * 4 < P.W && r < P.W) {                                       \
                wa[u] = *(rs_g4p)(P.s0 + P.qb + r);                                    \
                if (!(P.df & VB_MONO)) wb[u] = *(rs_g4p)(P.s0 + P.len + P.qb + r);     \
            }                                                                          \
        }                                                                              \
    }
        Rs2Port N = port(first);
        RS2_ISSUE(N)
        for (int p = 0; p < ld.ports; ++p) {
            if ((silent_ports >> p) & 1ull) {  // cleared zeros: copied by port 0, added by a 2/3/4-port mixer, skipped by an n-port one
                if (p > 0 && !masked) {
                    accl = accl + splat(0.f);
                    accr = accr + splat(0.f);
                }
                continue;
            }
            const Rs2Port C = N;
            const uint64_t cstep = C.step, cp_first = C.p_first;
            const bool cmono = C.df & VB_MONO;
            // the window goes to LDS as {L, R} pairs ...
            if (C.df & RS_CONTIG_BIT) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = (lane + u * WAVE) * 4;
                    if (u * WAVE * 4 < C.W && r < C.W) {
                        const v4f y = cmono ? wa[u] : wb[u];
                        *(v4f*)(win + r) = (v4f){wa[u][0], y[0], wa[u][1], y[1]};
                        *(v4f*)(win + r + 2) = (v4f){wa[u][2], y[2], wa[u][3], y[3]};
                    }
                }
            } else {  // the block a loop wraps in, or a one-shot's edge: frame by frame, wrapped / zero-filled on the way in
                for (int r = lane; r < C.W; r += WAVE) {
                    int q = C.qb + r;
                    bool in = true;
                    if (C.df & RS_LOOP_BIT) {  // (len >= the window: one step either way)
                        if (q < 0) q += C.len;
                        if (q >= C.len) q -= C.len;
                    } else {
                        in = q >= 0 && q < C.len;
                    }
                    const float x = in ? C.s0[in ? q : 0] : 0.f;
                    const float y = cmono ? x : (in ? C.s0[C.len + q] : 0.f);
                    win[r] = (v2f_rs){x, y};
                }
            }
            // ... the next port's is requested ...
            {
                const uint64_t later = p + 1 < 64 ? pure_ports >> (p + 1) : 0ull;
                if (later) {
                    N = port(p + 1 + __builtin_ctzll(later));
                    RS2_ISSUE(N)
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ... and this one's frames are convolved: 16-tap fmaf chains ascending from +0.0, both channels in one packed instruction
            const uint64_t dpos = (uint64_t)nact * cstep;
            const uint64_t pos_last = cp_first + (uint64_t)(nfr - 1) * cstep;
            const uint32_t i_first = (uint32_t)(cp_first >> 32);
            uint64_t pos = cp_first + (uint64_t)lane * cstep;
            v4f xl = splat(0.f), xr = splat(0.f);
            if (pairs) {
                // Round 6 — a full piece (256 frames): lane l convolves frames 2l, 2l+1 and 128+2l, 128+2l+1.  The windows of two
                // NEIGHBOURING frames A, B overlap in all but s = floor(step) or floor(step) + 1 slots, so the pair reads 18 window slots
                // where two separate frames read 32 (the frame-per-lane loop below keeps the LDS pipe 91 % busy: SQ_LDS_IDX_ACTIVE,
                // profiles/r06_rs_pairs_sq.txt).  B's chain walks the slots A's chain walks — same registers — with ITS coefficients
                // moved instead: by s taps, which is a per-lane choice between the filter bank and its copy shifted by one tap (and one
                // tap pair up or down), made once per pair.  Where the shifted row has no tap (slot in front of B's first / behind its last)
                // the fmaf is computed and dropped by a select — not fed a zero: (-0.0) + 0 * x is +0.0, and 0 * inf is not 0.
                // Each frame's chain is the same 16 fmaf ascending from +0.0 as below — same bits.
                auto conv_pairs = [&](auto flc) {
                    constexpr int FL = decltype(flc)::value;  // floor(step): 0 or 1 (a pure port's step is < 2)
                    const v2f_rs zero2 = (v2f_rs){0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {  // (one pair at a time: two chains side by side, as below — four cost the kernel its registers)
                        const uint64_t pa = cp_first + (uint64_t)(2 * lane + 128 * h) * cstep, pb = pa + cstep;
                        const bool sh = ((uint32_t)(pb >> 32) - (uint32_t)(pa >> 32)) != (uint32_t)FL;  // B's window starts FL + 1 slots behind A's
                        // B's coefficients for slot pair tp: s = 0: the bank's pair tp; s = 1: the shifted bank's; s = 2: the bank's pair tp - 1
                        const v2f_rs* bt = FL == 0 ? (sh ? tab1 : tab) : (sh ? tab - RS_PHASES : tab1);
                        rs_lp ha = (rs_lp)(tab + ((uint32_t)(pa >> 27) & (RS_PHASES - 1)));
                        rs_lp hb = (rs_lp)(bt + ((uint32_t)(pb >> 27) & (RS_PHASES - 1)));
                        rs_lp wp = (rs_lp)(win + ((uint32_t)(pa >> 32) - i_first));
                        v2f_rs acca = zero2, accb = zero2;
                        // (one tap pair ahead, by hand, and the addresses AND the sums laundered through an empty asm per tap pair: left alone
                        //  the compiler hoists every load of the chain to its top — 100 registers — and spills: the kernel has 128 at four
                        //  waves per SIMD.  A sched_barrier alone held in one of the four copies of this loop and not in the others.)
                        v2f_rs lo = wp[0], hi = wp[1], c0 = ha[0], c1 = hb[0];
#pragma unroll
                        for (int tp = 0; tp < RS_TAPS / 2; ++tp) {
                            asm volatile("" : "+v"(wp), "+v"(ha), "+v"(hb), "+v"(acca), "+v"(accb));
                            const v2f_rs nlo = wp[2 * tp + 2];
                            v2f_rs nhi = zero2, n0 = zero2;
                            if (FL == 1 || tp + 1 < RS_TAPS / 2) nhi = wp[2 * tp + 3];  // (slot 17: only a step >= 1 gets there)
                            if (tp + 1 < RS_TAPS / 2) n0 = ha[(tp + 1) * RS_PHASES];
                            const v2f_rs n1 = hb[(tp + 1) * RS_PHASES];
                            acca = __builtin_elementwise_fma((v2f_rs){c0.x, c0.x}, lo, acca);
                            if (FL == 0) {  // slot 2 tp: B's first tap sits here (s = 0) or one slot on (s = 1)
                                const v2f_rs t = __builtin_elementwise_fma((v2f_rs){c1.x, c1.x}, lo, accb);
                                accb = tp == 0 ? (sh ? zero2 : t) : t;
                            } else if (tp > 0) {  // (slot 0 is never B's)
                                accb = __builtin_elementwise_fma((v2f_rs){c1.x, c1.x}, lo, accb);
                            }
                            acca = __builtin_elementwise_fma((v2f_rs){c0.y, c0.y}, hi, acca);
                            {
                                const v2f_rs t = __builtin_elementwise_fma((v2f_rs){c1.y, c1.y}, hi, accb);
                                accb = (FL == 1 && tp == 0) ? (sh ? zero2 : t) : t;  // slot 1: B's first tap when s = 1, none when s = 2
                            }
                            lo = nlo;
                            hi = nhi;
                            c0 = n0;
                            c1 = n1;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        // slots 16, 17: what is left of B's chain
                        if (FL == 0) {
                            const v2f_rs t = __builtin_elementwise_fma((v2f_rs){c1.x, c1.x}, lo, accb);
                            accb = sh ? t : accb;  // s = 1: tap 15; s = 0: done
                        } else {
                            accb = __builtin_elementwise_fma((v2f_rs){c1.x, c1.x}, lo, accb);  // s = 1: tap 15; s = 2: tap 14
                            const v2f_rs t = __builtin_elementwise_fma((v2f_rs){c1.y, c1.y}, hi, accb);
                            accb = sh ? t : accb;  // s = 2: tap 15
                        }
                        xl[2 * h] = acca.x;
                        xr[2 * h] = acca.y;
                        xl[2 * h + 1] = accb.x;
                        xr[2 * h + 1] = accb.y;
                    }
                };
                if ((uint32_t)(cstep >> 32) == 0u) conv_pairs(std::integral_constant<int, 0>());
                else conv_pairs(std::integral_constant<int, 1>());
            } else
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                if (i < per) {  // (uniform.  Two frames' chains side by side: each is 16 dependent instructions long)
                    // (frames past the piece's end convolve its last frame's window: no branch around the reads)
                    const uint64_t psa = lane + i * nact < nfr && lane < nact ? pos : pos_last;
                    const uint64_t psb = lane + (i + 1) * nact < nfr && lane < nact ? pos + dpos : pos_last;
                    const rs_lp ha = (rs_lp)(tab + ((uint32_t)(psa >> 27) & (RS_PHASES - 1)));
                    const rs_lp wpa = (rs_lp)(win + ((uint32_t)(psa >> 32) - i_first));
                    const rs_lp hb = (rs_lp)(tab + ((uint32_t)(psb >> 27) & (RS_PHASES - 1)));
                    const rs_lp wpb = (rs_lp)(win + ((uint32_t)(psb >> 32) - i_first));
                    v2f_rs acca = (v2f_rs){0.f, 0.f}, accb = (v2f_rs){0.f, 0.f};
#pragma unroll
                    for (int tp = 0; tp < RS_TAPS / 2; ++tp) {
                        const v2f_rs h0 = ha[tp * RS_PHASES], h1 = hb[tp * RS_PHASES];
                        const v2f_rs a0 = wpa[2 * tp], a1 = wpa[2 * tp + 1];
                        const v2f_rs b0 = wpb[2 * tp], b1 = wpb[2 * tp + 1];
                        acca = __builtin_elementwise_fma((v2f_rs){h0.x, h0.x}, a0, acca);
                        accb = __builtin_elementwise_fma((v2f_rs){h1.x, h1.x}, b0, accb);
                        acca = __builtin_elementwise_fma((v2f_rs){h0.y, h0.y}, a1, acca);
                        accb = __builtin_elementwise_fma((v2f_rs){h1.y, h1.y}, b1, accb);
                    }
                    xl[i] = acca.x;
                    xr[i] = acca.y;
                    xl[i + 1] = accb.x;
                    xr[i + 1] = accb.y;
                    pos += 2 * dpos;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the next port overwrites the window)
            const uint32_t kinds = (uint32_t)__builtin_amdgcn_readlane((int)my_prog, p) << 4;
#pragma unroll
            for (int j = 0; j < FW_MAX_STAGES; ++j)
                if (j < ng) apply_stage((kinds >> (4 * j)) & 15u, splat(readlane_f(my_g[j][0], p)), splat(readlane_f(my_g[j][1], p)), xl, xr);
            if (p == 0) {
                accl = xl;
                accr = xr;
            } else {
                accl = accl + xl;
                accr = accr + xr;
            }
        }
#undef RS2_ISSUE
        if (pairs) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int f = fbase + 128 * h + 2 * lane;
                __builtin_nontemporal_store(accl[2 * h], outl + f);
                __builtin_nontemporal_store(accl[2 * h + 1], outl + f + 1);
                __builtin_nontemporal_store(accr[2 * h], outr + f);
                __builtin_nontemporal_store(accr[2 * h + 1], outr + f + 1);
            }
        } else if (lane < nact) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = lane + i * nact;
                if (f < nfr) {
                    __builtin_nontemporal_store(accl[i], outl + fbase + f);
                    __builtin_nontemporal_store(accr[i], outr + fbase + f);
                }
            }
        }
    }
    if (lane < 2 && part == 0) (fv.bus_flags + (size_t)k * fv.bus_flags_blk_stride)[ld.out_buf + lane] = 0;  // a live port: the out mask is 0
}
// the general kernel over k_leaf_rs's work list: (leaf, block, piece) items that are not resampler-pure.  The last workgroup out
// empties the list for the next pair of launches.
__global__ __launch_bounds__(WAVE* LEAF_WPB, 3) void k_leaf_sum_wl(FusedView fv, int K, int wpk) {
    extern __shared__ float s_leaf_dyn[];
    // (an empty list — every leaf resampler-pure, the steady state of a bank of resampled voices — is the common case: nothing to set
    //  up, nothing to reset; the launch then costs its dispatch, not 768 filter-bank copies: 13.4 -> ~3 us per step, r04)
    const unsigned int count = __hip_atomic_load(fv.rs_wl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (count == 0u) return;
    const RsLds rs = rs_lds_setup(fv, s_leaf_dyn);
    const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (unsigned int it = blockIdx.x * LEAF_WPB + wave; it < count; it += gridDim.x * LEAF_WPB) {
        const int leaf = __builtin_amdgcn_readfirstlane((int)fv.rs_wl[2 + 2 * it]);
        const unsigned int kp = (unsigned int)__builtin_amdgcn_readfirstlane((int)fv.rs_wl[3 + 2 * it]);
        leaf_sum_wave<true, true, LEAF_U, false>(fv, leaf, kp >> 2, (int)(kp & 3u), wpk, rs, K, nullptr);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int prev = atomicAdd(fv.rs_wl + 1, 1u);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(fv.rs_wl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fv.rs_wl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// control-ahead mode, plans with spatialiser stages: the histories the voices enter this call with, from the ext pool into the call's
// scratch, on the RENDER stream (behind the render kernels of the call before, in front of this call's leaf kernel)
__global__ __launch_bounds__(256) void k_sp_hist_copy(FusedView fv) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = idx / SP_HIST, i = idx % SP_HIST;
    if (v >= fv.n_voices) return;
    const int off = fv.voices[v].sp_ext_off;
    if (off >= 0) fv.hist[(size_t)v * SP_HIST + i] = fv.ext[(size_t)off + i];
}

// Upper sum tree of the fused plan, K-batched: SumNode semantics (nodes/sum.rs:41-136) with one THREAD per
// frame (blockIdx = node, block, channel) so that a 1-node level still puts K * n_out * frames/64 waves in flight.
// one upper-tree SumNode, one channel of one block, by a workgroup of 256 threads (k_bus_sum: a workgroup per (node, block, channel);
// the one-launch realtime kernels: the workgroup that completed the node's children, k_rt.hip.h)
__device__ __forceinline__ void bus_sum_node_wg(const DevView& v, const NodeDesc& nd, const uint32_t blk, const int c) {
    const int lane = threadIdx.x & (WAVE - 1);
    float* pool = v.pool + (size_t)blk * v.pool_blk_stride;
    uint8_t* flags = v.flags + (size_t)blk * v.flags_blk_stride;
    const int* in_buf = v.in_buf + nd.in_off;
    const int* out_buf = v.out_buf + nd.out_off;
    const int n_in = nd.n_in, n_out = nd.n_out, ports = nd.aux0;
    int my_in = lane < n_in ? in_buf[lane] : 0;
    asm volatile("" : "+v"(my_in));  // read with v_readlane from the frame loops below, where lanes without frames are inactive
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
    } else if ((v.frames & 3) == 0) {
        // four frames per thread (round 5): a 1 024-frame block is ONE pass of the workgroup with 8 ports' 16-byte loads in flight per
        // thread — frame by frame the mixer above 32 leaf buses was 64 us on the one workgroup the realtime kernels give it.  Same adds
        // in the same order per frame.
        const bool masked = !(ports == 2 || ports == 3 || ports == 4);
        for (int f = threadIdx.x * 4; f < v.frames; f += blockDim.x * 4) {
            v4f acc = *(const v4f*)(pool + (size_t)__builtin_amdgcn_readlane(my_in, c) * v.stride + f);
            for (int p0 = 1; p0 < ports; p0 += 8) {
                v4f x[8];
                bool use[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    use[u] = false;
                    x[u] = splat(0.f);
                    if (p0 + u < ports) {
                        int ic = n_out * (p0 + u) + c;
                        use[u] = !(masked && mask_bit(in_mask, ic));  // :122-124
                        x[u] = *(const v4f*)(pool + (size_t)__builtin_amdgcn_readlane(my_in, ic) * v.stride + f);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (use[u]) acc = acc + x[u];
            }
            *(v4f*)(out + f) = acc;
        }
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
__global__ __launch_bounds__(256) void k_bus_sum(DevView v, const int* __restrict__ level_nodes) {
    const NodeDesc nd = v.nodes[level_nodes[blockIdx.x]];
    bus_sum_node_wg(v, nd, blockIdx.y, (int)blockIdx.z);
}

// The root SumNode of the fused plans (stereo, its ports are bus buffers) fused with read_graph_outputs +
// interleave_stereo (schedule.rs:255-287, util.rs:123-147): one launch fewer per call and the root's planar result
// never goes to memory.  Same arithmetic as k_bus_sum followed by k_graph_out: all inputs silent -> the sum clears and
// flags both channels -> interleave_stereo zero-fills; n_in == n_out -> copy with mask passthrough; otherwise ports
// added in order (silent ports skipped on the n-port path only) and both flags are clear.
// The kernel is a handful of dependent memory round trips long, so it is built to have ONE: the root's port table
// arrives in the kernel arguments (scalar loads), every port's frame is requested before anything is waited for, and
// the silence flags (which only decide whether a loaded value is added) are fetched alongside.
template <int NP>
__device__ __forceinline__ void root_out_body(const DevView& v, const RootArgs& ra, float* __restrict__ out, const uint32_t blk, const int f) {
    const int lane = threadIdx.x & (WAVE - 1);
    const float* pool = v.pool + (size_t)blk * v.pool_blk_stride;
    const uint8_t* flags = v.flags + (size_t)blk * v.flags_blk_stride;
    const int n_in = ra.n_in, ports = ra.ports;
    const int fc = f < v.frames ? f : 0;  // whole waves stay in: the ballot below needs lanes 0..n_in-1
    // lane i needs the flag of input channel i.  Indexing the argument struct by lane would spill it, and going through the device
    // copy of the table (ra.in_tab) put a second dependent cold miss in front of the flags — on the realtime edge, right behind an
    // L2 invalidate, ~2 us of the callback.  A chain of selects on the scalar arguments instead: 2 NP v_cndmask, no memory.
    int my_buf = 0;
#pragma unroll
    for (int i = 0; i < 2 * NP; ++i) my_buf = lane == i ? ra.in_buf[i] : my_buf;
    const uint8_t my_flag = lane < n_in ? flags[my_buf] : (uint8_t)0;
    float xl[NP], xr[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {  // ports past the last one re-read port 0 (never added)
        const int q = p < ports ? p : 0;
        xl[p] = pool[(size_t)ra.in_buf[2 * q] * v.stride + fc];
        xr[p] = pool[(size_t)ra.in_buf[2 * q + 1] * v.stride + fc];
    }
    const uint64_t in_mask = __ballot(my_flag != 0);
    if (f >= v.frames) return;
    float* o = out + (size_t)blk * v.frames * 2;
    float2 y = make_float2(0.f, 0.f);
    if (mask_all(in_mask, n_in)) {
        // sum.rs:52-56 then util.rs:129-134
    } else if (n_in == 2) {  // sum.rs:58-65: copy, flags pass through; both silent was handled above
        y = make_float2(xl[0], xr[0]);
    } else {
        const bool masked = !(ports == 2 || ports == 3 || ports == 4);
        float accl = xl[0], accr = xr[0];
#pragma unroll
        for (int p = 1; p < NP; ++p) {
            const bool in = p < ports;
            const bool ul = in && !(masked && mask_bit(in_mask, 2 * p));  // :122-124
            const bool ur = in && !(masked && mask_bit(in_mask, 2 * p + 1));
            const float sl = accl + xl[p], sr = accr + xr[p];
            accl = ul ? sl : accl;
            accr = ur ? sr : accr;
        }
        y = make_float2(accl, accr);
    }
    *(float2*)(o + (size_t)f * 2) = y;
}
__device__ __forceinline__ void root_out_any(const DevView& v, const RootArgs& ra, float* __restrict__ out, const uint32_t blk, const int f) {
    if (ra.ports <= 4) root_out_body<4>(v, ra, out, blk, f);
    else if (ra.ports <= 8) root_out_body<8>(v, ra, out, blk, f);
    else if (ra.ports <= 16) root_out_body<16>(v, ra, out, blk, f);
    else root_out_body<32>(v, ra, out, blk, f);
}
__global__ __launch_bounds__(256) void k_root_out(DevView v, RootArgs ra, float* __restrict__ out) {
    root_out_any(v, ra, out, blockIdx.y, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}
// (the top-level R-port SumNode of a voice-sharded graph, k_bus_sum_ordered, lives in k_exchange.hip.h)
