// k_generic.hip.h — part of the single device translation unit fwgpu_kernels.hip (included inside namespace fwgpu).
// Generic level-batched executor (every node kind, one device function each), state init and graph I/O edges.
#pragma once

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

// generic biquad: LDS staging rows (channels x 256 frames, padded off the 32-bank period)
#define BQ_LDS_CH 4
#define BQ_LDS_PITCH 264  // 2 history samples + 256 frames, rows 16-byte aligned
// (one set of rows per wave of the workgroup, shared by the per-block path and the batch walker)
__shared__ float g_bq_lds[WPB][3][BQ_LDS_CH][BQ_LDS_PITCH];
// generic biquad through LDS, 256 frames at a time.  Rows of one wave: X[c] = (x2, x1, x[0..n)) — the two samples of history in
// front —, FF[c], Y[c].  Phase A, all 64 lanes, a frame each: the feed-forward half ff[i] = ((b0*x[i]) + (b1*x[i-1])) + (b2*x[i-2]),
// unfused, exactly the three operations of the serial formulation — it does not depend on the output.  Phase B, lane c alone:
// y[i] = fma(-a1, y[i-1], fma(-a2, y[i-2], ff[i])), four frames per LDS access.  A wave issues one instruction per 4 clocks
// whatever its lanes do, so the serial part is what counts: 2 fma per frame instead of 5 operations (26 -> ~7 ns per frame).
// (the rows are named through the array itself, not through pointers: a pointer to LDS kept in a struct decays to a generic one
// and the accesses to flat instructions)
struct BqRows {
    int w;  // wave of the workgroup
};
__device__ __forceinline__ BqRows bq_rows() { return BqRows{(int)((threadIdx.x >> 6) % WPB)}; }
#define BQ_X(r, c) g_bq_lds[(r).w][0][c]
#define BQ_FF(r, c) g_bq_lds[(r).w][1][c]
#define BQ_Y(r, c) g_bq_lds[(r).w][2][c]
__device__ __forceinline__ void bq_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// before: X[c][2 + i] hold the chunk's input (any lanes wrote them); lane c < nch holds channel c's state in x1, x2, y1, y2
// after: Y[c][0..n) hold the output, the state is the chunk's last; ends on a wave barrier
__device__ __forceinline__ void bq_filter_chunk(const BqRows& r, int nch, int lane, int n, float b0, float b1, float b2, float a1, float a2,
                                                float& x1, float& x2, float& y1, float& y2) {
    if (lane < nch) {
        BQ_X(r, lane)[0] = x2;
        BQ_X(r, lane)[1] = x1;
    }
    bq_wave_sync();
    for (int c = 0; c < nch; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            if (i < n) {
                float acc = b0 * BQ_X(r, c)[2 + i];
                acc = acc + (b1 * BQ_X(r, c)[1 + i]);
                acc = acc + (b2 * BQ_X(r, c)[i]);
                BQ_FF(r, c)[i] = acc;
            }
        }
    bq_wave_sync();
    if (lane < nch) {
        const float* f = BQ_FF(r, lane);
        float* yo = BQ_Y(r, lane);
        int i = 0;
        // eight frames per step, the NEXT eight requested before these are filtered and stored (an LDS read behind the store of
        // the previous step would wait out its ~100 clocks of latency every four frames: the rows may alias for all the compiler
        // knows)
        v4f c0 = splat(0.f), c1 = splat(0.f);
        if (n >= 8) {
            c0 = *(const v4f*)(f);
            c1 = *(const v4f*)(f + 4);
        }
        for (; i + 8 <= n; i += 8) {
            const int nxt = i + 16 <= n ? i + 8 : i;  // (the last step re-reads itself)
            const v4f n0 = *(const v4f*)(f + nxt), n1 = *(const v4f*)(f + nxt + 4);
            v4f yv0, yv1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = __builtin_fmaf(-a2, y2, c0[e]);
                acc = __builtin_fmaf(-a1, y1, acc);
                y2 = y1;
                y1 = acc;
                yv0[e] = acc;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = __builtin_fmaf(-a2, y2, c1[e]);
                acc = __builtin_fmaf(-a1, y1, acc);
                y2 = y1;
                y1 = acc;
                yv1[e] = acc;
            }
            *(v4f*)(yo + i) = yv0;
            *(v4f*)(yo + i + 4) = yv1;
            c0 = n0;
            c1 = n1;
        }
        for (; i < n; ++i) {
            float acc = __builtin_fmaf(-a2, y2, f[i]);
            acc = __builtin_fmaf(-a1, y1, acc);
            y2 = y1;
            y1 = acc;
            yo[i] = acc;
        }
        x1 = BQ_X(r, lane)[1 + n];  // x[n-1]
        x2 = BQ_X(r, lane)[n];      // x[n-2]  (n == 1: the old x1, which sits at X[1])
    }
    bq_wave_sync();
}


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
// The node kernel exists in three instantiations, by register appetite: one kernel for every kind needed 248 VGPRs
// (2 waves per SIMD — nothing to hide HBM latency behind, 0.9 TB/s on a level of volume nodes).  Set 0: the streaming
// kinds (volume, pan, sum, hard clip, mono<->stereo, width: 127 VGPRs), set 1: serial recurrences / filter banks / libm
// (beep, biquad, delay, resampler, spatialiser: 147), set 2: the sampler (every sample format, wraps, tails, ramps).
__device__ __forceinline__ int kind_set(int kind) {
    if (kind == K_SAMPLER) return 2;
    return (kind == K_BEEP || kind == K_BIQUAD || kind == K_DELAY || kind == K_RESAMPLER || kind == K_SPATIAL) ? 1 : 0;
}
// SET: 0 / 1 / 2 = that set only, 3 = all kinds (single-node entry).
// adv_blocks: a frozen, playing sampler (k_level) — put its playhead where `adv_blocks` steady blocks leave it first
template <int SET>
__device__ void node_process_wave(const DevView& v, int node_idx, uint32_t blk, uint32_t cmd_block, bool store_state = true,
                                  uint32_t adv_blocks = 0, bool frozen_sampler = false) {
    const NodeDesc nd = v.nodes[node_idx];
    if (nd.is_graph_io || nd.kind == K_FIR) return;  // I/O edges (k_graph_in/out); FIR banks run as MFMA GEMMs
    // the other instantiation's kinds return here; their switch cases are compiled out below (`if constexpr`: a case
    // that is compiled out falls through, which nothing can reach)
    if constexpr (SET != 3) {
        if (kind_set(nd.kind) != SET) return;
    }
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

        case K_VOLUME: if constexpr (SET == 0 || SET == 3) {  // nodes/volume.rs:84-145
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

        case K_PAN: if constexpr (SET == 0 || SET == 3) {  // SPEC node: volume.rs stereo path with one smoother per channel
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

        case K_SUM: if constexpr (SET == 0 || SET == 3) {  // nodes/sum.rs:41-136
            const int n_in = nd.n_in, n_out = nd.n_out, ports = nd.aux0 & 0xffff;
            const int path_ports = (nd.aux0 >> 16) ? (nd.aux0 >> 16) : ports;
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
            const bool masked = !(path_ports == 2 || path_ports == 3 || path_ports == 4);  // :67-133 (Q13)
            // lane i keeps the buffer id of input channel i; ids are broadcast with v_readlane so the
            // per-port loads are independent and can be in flight together (8 at a time)
            int my_in = lane < n_in ? io.in_buf[lane] : 0;
            asm volatile("" : "+v"(my_in));  // materialised under the full exec mask: read with v_readlane inside the frame
                                             // loops, where lanes without frames are inactive (see k_leaf_sum)
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

        case K_SAMPLER: if constexpr (SET == 2 || SET == 3) {  // nodes/sampler.rs:323-561 (messages already applied above)
            if (s.sample < 0 || !s.playing) {  // :416-430
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            const SampleDesc sd = v.samples[s.sample];
            if (sd.data == nullptr) {  // destroyed under the sampler (fwgpu_sample_destroy): as if it held no sample
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            GainRun run = smoother_begin(s.s0, s.p0, frames);        // :432-433
            if (!smoother_is_smoothing(s.s0) && run.c < 0.00001f) {  // :437-443
                out_mask = clear_all_outputs(io, 0, nd.n_out);
                break;
            }
            // the batch-start playhead comes from k_frozen_scan's snapshot: the wave of the batch's last block stores the
            // advanced state while waves of earlier blocks may not have read theirs yet
            if (frozen_sampler) s.playhead = v.frozen_playhead[node_idx];
            if (adv_blocks) {  // closed form of `adv_blocks` steady blocks (sampler.rs:445-484: one wrap per block, L >= frames)
                const uint64_t adv = (uint64_t)adv_blocks * (uint64_t)frames;
                if (s.has_loop) {
                    const uint64_t L = s.loop_end - s.loop_start;
                    const uint64_t off = s.playhead >= s.loop_end ? 0 : s.playhead - s.loop_start;
                    if (L) s.playhead = s.loop_start + (off + adv) % L;  // (k_frozen_scan admits real loops only)
                } else {
                    s.playhead += adv;
                }
            }
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
            // planar f32, the whole block contiguous in the sample: one 16-byte load per lane and channel instead of
            // four format-switched element fetches (same values: no conversion on this format)
            const bool vec_src = sd.format == FMT_P_F32 && !ft.wrap && !ft.tail_zero && ft.n1 == (uint32_t)frames;
            const float* vsrc = (const float*)sd.data + ft.off0;
            for (int base = 0; base < frames; base += 256) {
                int n = frames - base < 256 ? frames - base : 256;
                v4f g = gain_chunk(run, n, lane);
                int f0 = base + lane * 4;
                if (f0 >= frames) continue;
                v4f first = splat(0.f);
                for (int c = 0; c < nfill; ++c) {
                    v4f x = (vec_src && f0 + 4 <= frames ? (v4f)(*(const v4f_u*)(vsrc + (size_t)c * sd.frames + f0))
                                                         : sample_fetch4(sd, c, ft, (uint32_t)f0, (uint32_t)frames)) * g;  // :521-543
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

        case K_BEEP: if constexpr (SET == 1 || SET == 3) {  // nodes/beep_test.rs:71-97
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

        case K_HARD_CLIP: if constexpr (SET == 0 || SET == 3) {  // nodes/hard_clip.rs:51-95
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

        case K_MONO_TO_STEREO: if constexpr (SET == 0 || SET == 3) {  // nodes/mono_to_stereo.rs:33-50
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

        case K_STEREO_TO_MONO: if constexpr (SET == 0 || SET == 3) {  // nodes/stereo_to_mono.rs:33-56
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
        case K_WIDTH: if constexpr (SET == 0 || SET == 3) {  // SPEC: mid/side width, one smoothed parameter
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

        case K_BIQUAD: if constexpr (SET == 1 || SET == 3) {  // SPEC: RBJ biquad, Direct Form I, f32 state: unfused feed-forward half, then
            // y = fma(-a1, y1, fma(-a2, y2, ff)) (one fma on the recurrence's critical path; SPEC: DESIGN.md §6).
            // Serial in time: lane c runs channel c (the generic executor's coverage path).
            float* ext = v.ext + s.ext_off;
            const int nch = nd.n_in < nd.n_out ? nd.n_in : nd.n_out;
            if (nch <= BQ_LDS_CH) {
                // Up to 4 channels (a bus effect, a master filter): the block goes through LDS 256 frames at a time — the wave
                // loads it coalesced, lane c runs channel c's recurrence on LDS operands, the wave stores the result.  With
                // the samples read from global memory inside the loop every frame paid a memory round trip behind the
                // previous frame's store (in and out may alias as far as the compiler knows): 19 us per 256-frame block, and a
                // bus filter is ONE such chain over all K blocks of a call.  Same operations, same order.
                const BqRows rows = bq_rows();
                // (the coefficients are the node's, every lane holds them for phase A; the state is per channel, in lane c)
                const float b0 = ext[0], b1 = ext[1], b2 = ext[2], a1 = ext[3], a2 = ext[4];
                float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f;
                float* st = ext + 5 + 4 * (lane < nch ? lane : 0);
                if (lane < nch) x1 = st[0], x2 = st[1], y1 = st[2], y2 = st[3];
                for (int base = 0; base < frames; base += 256) {
                    const int n = frames - base < 256 ? frames - base : 256;
                    for (int c = 0; c < nch; ++c) {
                        const float* in = io.in(c) + base;
                        for (int f = lane; f < n; f += WAVE) BQ_X(rows, c)[2 + f] = in[f];
                    }
                    bq_filter_chunk(rows, nch, lane, n, b0, b1, b2, a1, a2, x1, x2, y1, y2);
                    for (int c = 0; c < nch; ++c) {
                        float* out = io.out(c) + base;
                        for (int f = lane; f < n; f += WAVE) out[f] = BQ_Y(rows, c)[f];
                    }
                    bq_wave_sync();  // (the next 256 frames overwrite the rows)
                }
                if (lane < nch) {
                    st[0] = x1;
                    st[1] = x2;
                    st[2] = y1;
                    st[3] = y2;
                }
            } else if (lane < nch) {
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

        case K_DELAY: if constexpr (SET == 1 || SET == 3) {  // SPEC: integer-sample delay line with feedback, ring per channel in the ext pool
            const uint32_t D = (uint32_t)s.loop_end;
            const uint32_t pos = (uint32_t)s.playhead;
            const float fb = s.p0, mix = s.p1, dry = s.gain;
            const int nch = nd.n_in < nd.n_out ? nd.n_in : nd.n_out;
            // frames inside one chunk touch distinct ring slots: up to 256 of them (a lane takes 4, all loads before the first
            // store: one memory round trip per chunk — a delay on a bus is one chain of such chunks over all K blocks of a call)
            const uint32_t chunk = D < 64u ? D : (D < 256u ? 64u * (D / 64u) : 256u);
            for (int c = 0; c < nch; ++c) {
                float* ring = v.ext + s.ext_off + (size_t)c * D;
                const float* in = io.in(c);
                float* out = io.out(c);
                for (uint32_t base = 0; base < (uint32_t)frames; base += chunk) {
                    float x[4], d[4];
                    uint32_t slot[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t j = (uint32_t)lane + 64u * q, i = base + j;
                        x[q] = d[q] = 0.f;
                        slot[q] = 0u;
                        if (j < chunk && i < (uint32_t)frames) {
                            slot[q] = (pos + i) % D;
                            x[q] = in[i];
                            d[q] = ring[slot[q]];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t j = (uint32_t)lane + 64u * q, i = base + j;
                        if (j < chunk && i < (uint32_t)frames) {
                            ring[slot[q]] = x[q] + (d[q] * fb);
                            out[i] = (x[q] * dry) + (d[q] * mix);
                        }
                    }
                    if (D < (uint32_t)frames) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // next chunk re-reads these slots
                }
            }
            s.playhead = (uint64_t)((pos + (uint32_t)frames) % D);
            break;
        }

        case K_RESAMPLER: if constexpr (SET == 1 || SET == 3) {  // SPEC: resampling source, polyphase windowed sinc
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

        case K_SPATIAL: if constexpr (SET == 1 || SET == 3) {  // SPEC: distance gain + equal-power pan + per-ear integer delay
            float* hist = v.ext + s.ext_off;
            const int dl = s.playing, dr = s.has_loop;
            // lane l keeps hist[l] (SP_HIST == 64), hist[63] = newest.  A frozen spatialiser (k_frozen_scan: gains at rest, no
            // message in the batch, frames >= SP_HIST) runs its K blocks in parallel: the history of block b > 0 IS the tail of
            // block b-1's input, which the level above has already written for every block of the batch
            float hreg;
            if (frozen_sampler && blk > 0) {
                const int jp = frames - SP_HIST + lane;
                const float* q0 = io.in(0) - v.pool_blk_stride;
                hreg = nd.n_in >= 2 ? (q0[jp] + (io.in(1) - v.pool_blk_stride)[jp]) * 0.5f : q0[jp];
            } else {
                hreg = hist[lane];
            }
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
            // new history = the last SP_HIST samples of (hist ++ m[0..frames))  (frozen: spatial_finish, once per batch)
            if (!frozen_sampler) {
                const int j = frames - SP_HIST + lane;
                const float keep = __shfl(hreg, (SP_HIST + j) & 63);
                hist[lane] = j >= 0 ? mono(j) : keep;
            }
            break;
        }

        default: break;
    }

    if (stateful && store_state && lane == 0) v.states[nd.state] = s;
    // schedule.rs:338-341: every output buffer's flag is overwritten with the node's out mask bit
    if (lane < nd.n_out) io.flags[io.out_buf[lane]] = mask_bit(out_mask, lane) ? 1 : 0;
}

// ---- "frozen" nodes.  Volume / pan / width / hard clip carry state (smoothers, a threshold) but their audio is
// stateless: when no message for the node falls into the batch and every smoother rests on its target, nothing about
// the node can change from block to block, and its K blocks are as independent as a stateless node's.  Decided ONCE per
// batch, before the first level, from the state as the batch finds it (a decision taken inside k_level would race with
// the wave that walks a non-frozen node).  The one thing that still moves — an all-silent block resets a Deactivating
// smoother to Inactive, same values (smoother.rs:115-129, volume.rs:94-99) — is patched by the node's block-0 wave.
__device__ __forceinline__ bool smoother_at_rest(const Smoother& s, float target) {
    return s.status != SM_ACTIVE && s.input == target;
}
__global__ void k_frozen_scan(DevView v, int n_nodes, uint32_t cmd_block0, uint32_t K, uint8_t* __restrict__ frozen,
                              unsigned long long* __restrict__ playhead_snap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const NodeDesc nd = v.nodes[i];
    bool fz = false, adv = false;
    bool spat = false;
    if (!nd.is_graph_io && (nd.kind == K_VOLUME || nd.kind == K_PAN || nd.kind == K_WIDTH || nd.kind == K_HARD_CLIP ||
                            nd.kind == K_SAMPLER || nd.kind == K_SPATIAL)) {
        bool has_cmd = false;
        if (v.n_cmds) {
            const int c = chain_cmd_lower_bound(v.cmds, v.n_cmds, nd.state, cmd_block0);
            has_cmd = c < v.n_cmds && v.cmds[c].state == nd.state && v.cmds[c].block < cmd_block0 + K;
        }
        if (!has_cmd) {
            const NodeState& s = v.states[nd.state];
            switch (nd.kind) {
                case K_SAMPLER:  // steady playback: the playhead of block b has a closed form (see node_process_wave)
                    if (s.sample < 0 || !s.playing || v.samples[s.sample].data == nullptr) {
                        fz = true;  // outputs cleared, nothing moves (sampler.rs:416-430)
                    } else if (smoother_at_rest(s.s0, s.p0)) {
                        if (s.s0.status == SM_INACTIVE && s.s0.input < 0.00001f) {
                            fz = true;  // muted: cleared, the playhead does not move (:437-443)
                        } else if (s.has_loop) {
                            // (a range that is empty, inverted or reaches past the sample plays silence block by block:
                            // sampler_advance — keep it off the closed form, whose `% L` needs a real loop)
                            adv = fz = s.loop_end > s.loop_start && s.loop_end - s.loop_start >= (uint64_t)v.frames &&
                                       s.playhead >= s.loop_start && s.loop_end <= v.samples[s.sample].frames;
                        } else {  // one-shot that does not end inside the batch
                            adv = fz = s.playhead + (uint64_t)K * (uint64_t)v.frames <= v.samples[s.sample].frames;
                        }
                    }
                    break;
                case K_HARD_CLIP: fz = true; break;
                // SPEC spatialiser: with both gains at rest all that moves is the 64-frame mono history, and that is the
                // tail of the previous block's input (needs a block of at least SP_HIST frames)
                case K_SPATIAL: spat = fz = v.frames >= SP_HIST && smoother_at_rest(s.s0, s.p0) && smoother_at_rest(s.s1, s.p1); break;
                // volume.rs:104: a gain below 1e-5 is a mute only while the smoother is Inactive — keep such a node
                // on the serial path while it is Deactivating (the status matters there)
                case K_VOLUME: fz = smoother_at_rest(s.s0, s.p0) && (s.s0.status == SM_INACTIVE || !(s.s0.input < 0.00001f)); break;
                case K_WIDTH: fz = smoother_at_rest(s.s0, s.p0); break;
                default: fz = smoother_at_rest(s.s0, s.p0) && smoother_at_rest(s.s1, s.p1); break;  // K_PAN
            }
        }
    }
    if (v.chain_done)  // (this batch's "rendered upstream" bits: fz_links)
        for (int w = 0; w < v.chain_words; ++w) v.chain_done[(size_t)i * v.chain_words + w] = 0u;
    frozen[i] = fz ? (adv ? 2 : (spat ? 3 : 1)) : 0;  // 2: a playing sampler — the last block's wave stores the state; 3: a spatialiser
    if (adv) playhead_snap[i] = v.states[nd.state].playhead;
}
// block-0 wave of a frozen spatialiser, after its own blocks (it alone reads the stored history): the history the batch
// leaves behind is the tail of the LAST block's input
__device__ void spatial_finish(const DevView& v, int node_idx, uint32_t K) {
    const NodeDesc nd = v.nodes[node_idx];
    const int lane = threadIdx.x & (WAVE - 1);
    const float* pool = v.pool + (size_t)(K - 1) * v.pool_blk_stride;
    const int* in_buf = v.in_buf + nd.in_off;
    const int j = v.frames - SP_HIST + lane;
    const float m0 = pool[(size_t)in_buf[0] * v.stride + j];
    const float m = nd.n_in >= 2 ? (m0 + pool[(size_t)in_buf[1] * v.stride + j]) * 0.5f : m0;
    (v.ext + v.states[nd.state].ext_off)[lane] = m;
}
// block-0 wave of a frozen node: if some block of the batch had every input silent, its smoothers were reset
__device__ void frozen_finish(const DevView& v, int node_idx, uint32_t K) {
    const NodeDesc nd = v.nodes[node_idx];
    if (nd.kind == K_HARD_CLIP || nd.kind == K_SAMPLER) return;
    NodeState s = v.states[nd.state];
    if (s.s0.status != SM_DEACTIVATING && !(nd.kind == K_PAN && s.s1.status == SM_DEACTIVATING)) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const int* in_buf = v.in_buf + nd.in_off;
    bool any = false;
    for (uint32_t b = lane; b < K; b += WAVE) {
        const uint8_t* fl = v.flags + (size_t)b * v.flags_blk_stride;
        bool all = true;
        for (int c = 0; c < nd.n_in; ++c) all = all && fl[in_buf[c]] != 0;
        any = any || all;
    }
    if (__ballot(any) == 0ull) return;
    smoother_reset(s.s0, s.p0);
    if (nd.kind == K_PAN) smoother_reset(s.s1, s.p1);
    if (lane == 0) v.states[nd.state] = s;
}

// A bus biquad (<= 4 channels) with no message inside the batch: ONE wave walks the node's K blocks with everything that does
// not change from block to block — descriptor, port tables, coefficients, filter state — held in registers, the next 256
// frames of input requested before the current ones are filtered (every block's input is already there: the level above
// ran for the whole batch), the recurrence on LDS operands.  node_process_wave, called block by block, re-reads all of that
// behind the previous block's state store: ~8 us per block against ~1.5 us here — and a bus filter is one serial chain over
// all the blocks of a call.  Same operations in the same order as the K_BIQUAD case above.
__device__ __forceinline__ bool biquad_walk_ok(const DevView& v, const NodeDesc& nd, uint32_t cmd_block0, uint32_t K) {
    const int nch = nd.n_in < nd.n_out ? nd.n_in : nd.n_out;
    if (nd.kind != K_BIQUAD || nch < 1 || nch > 2 || K < 2 || (v.frames & 255)) return false;  // mono / stereo, whole 256-frame chunks
    if (v.n_cmds) {
        const int c = chain_cmd_lower_bound(v.cmds, v.n_cmds, nd.state, cmd_block0);
        if (c < v.n_cmds && v.cmds[c].state == nd.state && v.cmds[c].block < cmd_block0 + K) return false;
    }
    return true;
}
// (NCH = 1 or 2 and whole chunks: every load and store of the loop is unconditional, so that the compiler counts them and waits
// for the prefetched chunk with an exact vmcnt(N) — behind a predicate it falls back to vmcnt(0), which also waits for the
// chunk's own stores: a full memory round trip per chunk, 4.5 us instead of ~1.5)
template <int NCH>
__device__ __forceinline__ void biquad_walk(const DevView& v, const NodeDesc& nd, uint32_t K) {
    const BqRows rows = bq_rows();
    const int lane = threadIdx.x & (WAVE - 1);
    const int frames = v.frames;
    float* ext = v.ext + v.states[nd.state].ext_off;
    int ib[NCH], ob[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        ib[c] = (v.in_buf + nd.in_off)[c];
        ob[c] = (v.out_buf + nd.out_off)[c];
    }
    const int my_out = lane < nd.n_out ? (v.out_buf + nd.out_off)[lane] : 0;
    const float b0 = ext[0], b1 = ext[1], b2 = ext[2], a1 = ext[3], a2 = ext[4];
    float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f;
    float* st = ext + 5 + 4 * (lane < NCH ? lane : 0);
    if (lane < NCH) x1 = st[0], x2 = st[1], y1 = st[2], y2 = st[3];
    // chunk i = 256 frames: (block, base); lane l keeps frames l, l + 64, l + 128, l + 192 of each channel
    const int cpb = frames / 256;
    const uint32_t n_chunks = K * (uint32_t)cpb;
    float nx[NCH][4];
    auto fetch = [&](uint32_t ci) {
        const uint32_t b = ci / (uint32_t)cpb;
        const int base = (int)(ci % (uint32_t)cpb) * 256;
        const float* pool = v.pool + (size_t)b * v.pool_blk_stride + base + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) nx[c][q] = pool[(size_t)ib[c] * v.stride + 64 * q];
    };
    fetch(0);
    for (uint32_t ci = 0; ci < n_chunks; ++ci) {
        const uint32_t b = ci / (uint32_t)cpb;
        const int base = (int)(ci % (uint32_t)cpb) * 256;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) BQ_X(rows, c)[2 + lane + 64 * q] = nx[c][q];
        fetch(ci + 1 < n_chunks ? ci + 1 : ci);  // in flight while this chunk is filtered (the last one re-reads itself: one path)
        bq_filter_chunk(rows, NCH, lane, 256, b0, b1, b2, a1, a2, x1, x2, y1, y2);
        float* pool = v.pool + (size_t)b * v.pool_blk_stride + base + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) pool[(size_t)ob[c] * v.stride + 64 * q] = BQ_Y(rows, c)[lane + 64 * q];
        // schedule.rs:338-341: the node's out mask (0: a filter never reports silence) overwrites its output buffers' flags
        if (base == 0 && lane < nd.n_out) (v.flags + (size_t)b * v.flags_blk_stride)[my_out] = 0;
        bq_wave_sync();  // (the next chunk overwrites the rows)
    }
    if (lane < NCH) {
        st[0] = x1;
        st[1] = x2;
        st[2] = y1;
        st[3] = y2;
    }
}

// The same for a bus delay of at least 512 frames (<= 4 channels, no message inside the batch): 256 frames at a time, the next
// chunk's input AND ring slots requested before this chunk's are written (two chunks span less than the ring: they cannot
// meet), nothing re-read per block.  Same operations as the K_DELAY case above.
__device__ __forceinline__ bool delay_walk_ok(const DevView& v, const NodeDesc& nd, uint32_t cmd_block0, uint32_t K) {
    const int nch = nd.n_in < nd.n_out ? nd.n_in : nd.n_out;
    if (nd.kind != K_DELAY || nch < 1 || nch > 2 || K < 2 || (v.frames & 255)) return false;  // mono / stereo, whole 256-frame chunks
    if (v.states[nd.state].loop_end < 512) return false;
    if (v.n_cmds) {
        const int c = chain_cmd_lower_bound(v.cmds, v.n_cmds, nd.state, cmd_block0);
        if (c < v.n_cmds && v.cmds[c].state == nd.state && v.cmds[c].block < cmd_block0 + K) return false;
    }
    return true;
}
template <int NCH>
__device__ __forceinline__ void delay_walk(const DevView& v, const NodeDesc& nd, uint32_t K) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int frames = v.frames;
    NodeState* const sp = &v.states[nd.state];  // (field by field: a struct copy from global memory ends up in scratch)
    const uint32_t D = (uint32_t)sp->loop_end;
    const float fb = sp->p0, mix = sp->p1, dry = sp->gain;
    float* const ring0 = v.ext + sp->ext_off;
    int ib[NCH], ob[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        ib[c] = (v.in_buf + nd.in_off)[c];
        ob[c] = (v.out_buf + nd.out_off)[c];
    }
    const int my_out = lane < nd.n_out ? (v.out_buf + nd.out_off)[lane] : 0;
    const int cpb = frames / 256;
    const uint32_t n_chunks = K * (uint32_t)cpb;
    float nx[NCH][4], ndl[NCH][4];
    uint32_t nslot[4];
    uint32_t pos = (uint32_t)sp->playhead;  // ring position of the chunk being fetched
    uint32_t pos_after = pos;                // ... and behind the last chunk WRITTEN
    auto fetch = [&](uint32_t ci, uint32_t at) {
        const uint32_t b = ci / (uint32_t)cpb;
        const int base = (int)(ci % (uint32_t)cpb) * 256;
        const float* pool = v.pool + (size_t)b * v.pool_blk_stride + base + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t sl = at + (uint32_t)lane + 64u * q;  // at < D, offset < 256 <= D / 2: at most one wrap
            nslot[q] = sl >= D ? sl - D : sl;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                nx[c][q] = pool[(size_t)ib[c] * v.stride + 64 * q];
                ndl[c][q] = ring0[(size_t)c * D + nslot[q]];
            }
    };
    fetch(0, pos);
    for (uint32_t ci = 0; ci < n_chunks; ++ci) {
        const uint32_t b = ci / (uint32_t)cpb;
        const int base = (int)(ci % (uint32_t)cpb) * 256;
        float x[NCH][4], d[NCH][4];
        uint32_t slot[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[q] = nslot[q];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                x[c][q] = nx[c][q];
                d[c][q] = ndl[c][q];
            }
        pos_after = pos + 256u;
        if (pos_after >= D) pos_after -= D;
        // the next chunk's input and ring slots, in flight while this chunk is written (the last chunk re-reads itself: one path)
        const bool more = ci + 1 < n_chunks;
        fetch(more ? ci + 1 : ci, more ? pos_after : pos);
        pos = pos_after;
        float* pool = v.pool + (size_t)b * v.pool_blk_stride + base + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ring0[(size_t)c * D + slot[q]] = x[c][q] + (d[c][q] * fb);
                pool[(size_t)ob[c] * v.stride + 64 * q] = (x[c][q] * dry) + (d[c][q] * mix);
            }
        if (base == 0 && lane < nd.n_out) (v.flags + (size_t)b * v.flags_blk_stride)[my_out] = 0;  // out mask 0 (schedule.rs:338-341)
    }
    if (lane == 0) sp->playhead = (uint64_t)pos;  // (nothing else of the node's state moves)
}

// ---- frozen nodes of a wide level, several blocks at a time (round 4).  node_process_wave block by block is a chain of dependent
// loads (descriptor -> port table -> flags -> 128 bytes of state) in front of 4 KB of audio: config 2 on the levels alone ran its
// sampler / volume / pan levels at 1.9-2.4 TB/s.  For the kinds such a level is made of — a stereo VolumeNode, pan or width with
// resting smoothers, a stereo hard clip, a steadily playing stereo planar-f32 sampler — everything but the audio is the same in every block of the
// batch: the wave reads it once and streams FZ_U blocks at a time, their loads in flight together.  Same operations per sample
// as the cases of node_process_wave (volume.rs:94-142, sampler.rs:445-543); anything else returns false / goes block by block.
#ifndef FZ_U
#define FZ_U 4
#endif
// ---- vertical fusion (round 5).  The level executor moved 56 bytes per voice-sample on config 2's graph: every node of a
// sampler -> volume -> pan chain wrote its block to the pool and the next level read it back.  When the node a frozen wave renders feeds
// exactly ONE consumer with both its channels, in order (NodeDesc::aux0 = that node + 1: the host's chain table, fwgpu_plan_install.cpp),
// and the consumer is a frozen gain-like node too (volume / pan / width / hard clip, not muted), the wave applies the consumer's
// operation to the block IN REGISTERS, and the consumer's consumer's, up to FZ_LINKS of them, and stores only the LAST node's output.
// Only blocks whose head output is not flagged silent on either channel are fused: every link then sees in mask 0 and leaves out mask 0
// (volume.rs:110, hard_clip.rs:93) — the flags of every skipped buffer are still written (frozen_finish reads them), the audio is not:
// nobody else reads those buffers.  The links' own waves, a level later, find the blocks in `chain_done` and leave them alone.
// Same operations per sample as each node's own case: one rounding per product, in chain order.
#define FZ_LINKS 3
struct FzLinks {
    int n;
    int kind[FZ_LINKS], node[FZ_LINKS], o0[FZ_LINKS], o1[FZ_LINKS];
    float gl[FZ_LINKS], gr[FZ_LINKS];
};
__device__ __forceinline__ void fz_links(const DevView& v, const int next_plus_1, FzLinks& L) {
    L.n = 0;
#pragma unroll
    for (int j = 0; j < FZ_LINKS; ++j) {
        L.kind[j] = 0;
        L.node[j] = L.o0[j] = L.o1[j] = 0;
        L.gl[j] = L.gr[j] = 1.f;
    }
    if (!v.chain_done || !v.frozen) return;
    int nx = next_plus_1 - 1;
#pragma unroll
    for (int j = 0; j < FZ_LINKS; ++j) {
        if (nx < 0 || L.n != j) break;
        if (v.frozen[nx] != 1) break;
        const NodeDesc nn = v.nodes[nx];
        const NodeState& s = v.states[nn.state];
        const float gl = nn.kind == K_HARD_CLIP ? s.p0 : s.s0.input;
        if (nn.kind == K_VOLUME && s.s0.status == SM_INACTIVE && gl < 0.00001f) break;  // a mute clears and flags: not a fused shape
        L.kind[j] = nn.kind;
        L.node[j] = nx;
        L.gl[j] = gl;
        L.gr[j] = nn.kind == K_PAN ? s.s1.input : gl;
        L.o0[j] = (v.out_buf + nn.out_off)[0];
        L.o1[j] = (v.out_buf + nn.out_off)[1];
        L.n = j + 1;
        nx = nn.aux0 - 1;
    }
}
__device__ __forceinline__ void fz_apply(const FzLinks& L, v4f& yl, v4f& yr) {
#pragma unroll
    for (int j = 0; j < FZ_LINKS; ++j)
        if (j < L.n) {
            if (L.kind[j] == K_HARD_CLIP) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    yl[e] = clipf(yl[e], L.gl[j]);
                    yr[e] = clipf(yr[e], L.gl[j]);
                }
            } else if (L.kind[j] == K_WIDTH) {
                const v4f mid = (yl + yr) * 0.5f;
                const v4f sd = ((yl - yr) * 0.5f) * L.gl[j];
                yl = mid + sd;
                yr = mid - sd;
            } else {
                yl = yl * L.gl[j];
                yr = yr * L.gr[j];
            }
        }
}
// the flags of the buffers a fused block skipped (all clear), and the links' done bits for the blocks `chained` of [b0, ...)
__device__ __forceinline__ void fz_flags(const DevView& v, const FzLinks& L, const uint32_t blk, const int c) {
    uint8_t* fl = v.flags + (size_t)blk * v.flags_blk_stride;
#pragma unroll
    for (int j = 0; j < FZ_LINKS; ++j)
        if (j < L.n) fl[c ? L.o1[j] : L.o0[j]] = 0;
}
__device__ __forceinline__ void fz_done(const DevView& v, const FzLinks& L, const uint32_t b0, const uint32_t chained) {
    if (!chained || (threadIdx.x & (WAVE - 1)) != 0) return;
#pragma unroll
    for (int j = 0; j < FZ_LINKS; ++j)
        if (j < L.n) atomicOr(&v.chain_done[(size_t)L.node[j] * v.chain_words + (b0 >> 5)], chained << (b0 & 31u));
}
// Returns the blocks of [b0, b1) it did NOT render, bit (b - b0) each — all of them when the node is not one of the three shapes
// (at most 32 blocks per wave).
// `skip`: blocks an upstream wave has rendered already (chain_done).
template <int SET>
__device__ __forceinline__ uint32_t frozen_fast(const DevView& v, const int node, const uint8_t fz, const uint32_t b0, const uint32_t b1, const uint32_t K,
                                                const uint32_t skip) {
    const NodeDesc nd = v.nodes[node];
    const int lane = threadIdx.x & (WAVE - 1);
    const int frames = v.frames;
    if (nd.n_out != 2 || (frames & 3) || b1 - b0 > 32u) return ~0u;
    const int* obt = v.out_buf + nd.out_off;
    const int o0 = obt[0], o1 = obt[1];
    const uint32_t u_l = (uint32_t)lane >> 1;  // flags: lane 2u + c holds (block b + u, channel c)
    const int c_l = lane & 1;
    if constexpr (SET == 0) {
        const int kind = nd.kind;
        if (fz != 1 || !(kind == K_VOLUME || kind == K_PAN || kind == K_HARD_CLIP || kind == K_WIDTH) || nd.n_in != 2) return ~0u;
        const int* ibt = v.in_buf + nd.in_off;
        const int i0 = ibt[0], i1 = ibt[1];
        const NodeState& s = v.states[nd.state];
        const float gl = kind == K_HARD_CLIP ? s.p0 : s.s0.input;  // a resting smoother's block is its input (smoother.rs:162-167); clip: the threshold
        const float gr = kind == K_PAN ? s.s1.input : gl;
        const bool mute = kind == K_VOLUME && s.s0.status == SM_INACTIVE && gl < 0.00001f;  // volume.rs:104-108
        FzLinks L;
        fz_links(v, nd.aux0, L);
        const int lo0 = L.n ? L.o0[L.n - 1] : o0, lo1 = L.n ? L.o1[L.n - 1] : o1;
        uint32_t chained = 0u;
        for (uint32_t b = b0; b < b1; b += FZ_U) {
            uint8_t f = 0;
            const bool live_l = b + u_l < b1 && !((skip >> (b + u_l - b0)) & 1u);
            if (lane < 2 * FZ_U && live_l) f = (v.flags + (size_t)(b + u_l) * v.flags_blk_stride)[c_l ? i1 : i0];
            const uint32_t fm = (uint32_t)__ballot(f != 0);
            // blocks of this round that are rendered on into the links: the node's own out mask is 0 there
            uint32_t ch = 0u;
#pragma unroll
            for (int u = 0; u < FZ_U; ++u) {
                const uint32_t m = (fm >> (2 * u)) & 3u;
                if (L.n && b + u < b1 && !((skip >> (b + u - b0)) & 1u) && !mute && (kind == K_WIDTH ? m != 3u : m == 0u)) ch |= 1u << u;
            }
            chained |= ch << (b - b0);
            for (int f0 = lane * 4; f0 < frames; f0 += 256) {
                v4f x[FZ_U][2];
#pragma unroll
                for (int u = 0; u < FZ_U; ++u) {
                    x[u][0] = x[u][1] = splat(0.f);
                    const bool live = b + u < b1 && !((skip >> (b + u - b0)) & 1u);
                    if (live && ((fm >> (2 * u)) & 3u) != 3u && !mute) {  // (all inputs silent: cleared — volume.rs:94-100 and its like)
                        const float* pl = v.pool + (size_t)(b + u) * v.pool_blk_stride;
                        x[u][0] = *(const v4f*)(pl + (size_t)i0 * v.stride + f0);
                        x[u][1] = *(const v4f*)(pl + (size_t)i1 * v.stride + f0);
                    }
                }
#pragma unroll
                for (int u = 0; u < FZ_U; ++u)
                    if (b + u < b1 && !((skip >> (b + u - b0)) & 1u)) {
                        float* pl = v.pool + (size_t)(b + u) * v.pool_blk_stride;
                        const uint32_t m = (fm >> (2 * u)) & 3u;
                        v4f yl = splat(0.f), yr = splat(0.f);
                        if (m != 3u && !mute) {
                            if (kind == K_HARD_CLIP) {  // hard_clip.rs:60-93: a silent channel of a half-silent pair is written as zeros
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    yl[j] = (m & 1u) ? 0.f : clipf(x[u][0][j], gl);
                                    yr[j] = (m & 2u) ? 0.f : clipf(x[u][1][j], gl);
                                }
                            } else if (kind == K_WIDTH) {  // SPEC: m = (l + r) * 0.5; s = ((l - r) * 0.5) * w
                                const v4f mid = (x[u][0] + x[u][1]) * 0.5f;
                                const v4f sd = ((x[u][0] - x[u][1]) * 0.5f) * gl;
                                yl = mid + sd;
                                yr = mid - sd;
                            } else {  // volume.rs:123-126 (stereo path: both channels, flagged or not), pan
                                yl = x[u][0] * gl;
                                yr = x[u][1] * gr;
                            }
                        }
                        if ((ch >> u) & 1u) {  // on through the links, in registers: only the last one's buffers are written
                            fz_apply(L, yl, yr);
                            *(v4f*)(pl + (size_t)lo0 * v.stride + f0) = yl;
                            *(v4f*)(pl + (size_t)lo1 * v.stride + f0) = yr;
                        } else {
                            *(v4f*)(pl + (size_t)o0 * v.stride + f0) = yl;
                            *(v4f*)(pl + (size_t)o1 * v.stride + f0) = yr;
                        }
                    }
            }
            if (lane < 2 * FZ_U && live_l) {
                // out mask: all silent / muted -> both flagged; volume / pan / clip: the in mask (volume.rs:110, hard_clip.rs:93 — its fast
                // path leaves 0, which IS the in mask there); width: 0
                const uint32_t m = (fm >> (2 * u_l)) & 3u;
                (v.flags + (size_t)(b + u_l) * v.flags_blk_stride)[c_l ? o1 : o0] =
                    (m == 3u || mute) ? 1 : (kind == K_WIDTH ? 0 : (uint8_t)((m >> c_l) & 1u));
                if ((ch >> u_l) & 1u) fz_flags(v, L, b + u_l, c_l);
            }
        }
        fz_done(v, L, b0, chained);
        return 0u;
    } else if constexpr (SET == 2) {
        if (fz != 2 || nd.kind != K_SAMPLER) return ~0u;
        const NodeState& s0 = v.states[nd.state];
        const SampleDesc sd = v.samples[s0.sample];
        const uint64_t loop_start = s0.loop_start, loop_end = s0.loop_end;
        const int has_loop = s0.has_loop;
        uint32_t todo = 0u;
        if (sd.format != FMT_P_F32 || sd.channels != 2) return ~0u;
        const float g = s0.s0.input;  // (k_frozen_scan: the gain rests and is no mute)
        const uint64_t ph0 = v.frozen_playhead[node];
        FzLinks L;  // (a playing sampler's block is never flagged silent: every block of the fast path goes on through the links)
        fz_links(v, nd.aux0, L);
        const int lo0 = L.n ? L.o0[L.n - 1] : o0, lo1 = L.n ? L.o1[L.n - 1] : o1;
        uint32_t chained = 0u;
        for (uint32_t b = b0; b < b1; b += FZ_U) {
            const float* src[FZ_U];
            uint32_t slow = 0;
#pragma unroll
            for (int u = 0; u < FZ_U; ++u) {
                src[u] = nullptr;
                const uint32_t blk = b + u;
                if (blk >= b1) continue;
                NodeState t;  // the playhead of block blk in closed form (node_process_wave, K_SAMPLER), then the block's own advance
                t.loop_start = loop_start;
                t.loop_end = loop_end;
                t.has_loop = has_loop;
                t.playing = 1;
                t.playhead = ph0;
                const uint64_t adv = (uint64_t)blk * (uint64_t)frames;
                if (blk) {
                    if (t.has_loop) {
                        const uint64_t L = t.loop_end - t.loop_start;
                        const uint64_t off = t.playhead >= t.loop_end ? 0 : t.playhead - t.loop_start;
                        if (L) t.playhead = t.loop_start + (off + adv) % L;
                    } else {
                        t.playhead += adv;
                    }
                }
                Fetch ft;
                const bool ok = sampler_advance(t, sd.frames, (uint32_t)frames, ft);
                if (blk + 1 == K || !ok || ft.wrap || ft.tail_zero || ft.n1 != (uint32_t)frames) slow |= 1u << u;  // (the batch's last block stores the state)
                else src[u] = (const float*)sd.data + ft.off0;
            }
            for (int f0 = lane * 4; f0 < frames; f0 += 256) {
                v4f x[FZ_U][2];
#pragma unroll
                for (int u = 0; u < FZ_U; ++u)
                    if (src[u]) {
                        x[u][0] = (v4f)(*(const v4f_u*)(src[u] + f0));
                        x[u][1] = (v4f)(*(const v4f_u*)(src[u] + (size_t)sd.frames + f0));
                    }
#pragma unroll
                for (int u = 0; u < FZ_U; ++u)
                    if (src[u]) {
                        float* pl = v.pool + (size_t)(b + u) * v.pool_blk_stride;
                        v4f yl = x[u][0] * g, yr = x[u][1] * g;  // sampler.rs:521-543
                        if (L.n) fz_apply(L, yl, yr);
                        *(v4f*)(pl + (size_t)lo0 * v.stride + f0) = yl;
                        *(v4f*)(pl + (size_t)lo1 * v.stride + f0) = yr;
                    }
            }
            if (lane < 2 * FZ_U && b + u_l < b1 && !((slow >> u_l) & 1u)) {
                (v.flags + (size_t)(b + u_l) * v.flags_blk_stride)[c_l ? o1 : o0] = 0;
                if (L.n) fz_flags(v, L, b + u_l, c_l);
            }
            todo |= slow << (b - b0);
            if (L.n) chained |= ((b1 - b >= FZ_U ? (1u << FZ_U) - 1u : (1u << (b1 - b)) - 1u) & ~slow) << (b - b0);
        }
        fz_done(v, L, b0, chained);
        return todo;
    } else {
        return ~0u;
    }
}

// K blocks per launch (gridDim.y = K, one pool slice per block).  A node whose audio half carries state from block
// to block is run by ONE wave that walks its K blocks in order; stateless nodes — and frozen ones — take their K
// blocks in parallel.
#ifndef LEVEL_BPW
#define LEVEL_BPW 8  // consecutive blocks one wave takes for a stateless / frozen node of a WIDE level
#endif
#ifndef LEVEL_BPW_WIDE
#define LEVEL_BPW_WIDE 32  // ... of a level with >= 262 144 (node, block) pairs (fwgpu_kernels.hip launch_level; FWGPU_LEVEL_BPW_WIDE)
#endif
// gridDim.y = ceil(K / bpw): a wave takes bpw consecutive blocks of its node, so the node's descriptor, port tables and
// state come from HBM once and from the cache for the other blocks (the per-block work is ~4 KB behind a chain of dependent
// loads).  bpw = LEVEL_BPW on a level with thousands of (node, block) pairs; a level of one or two bus nodes — the root of a
// hybrid plan, a return chain — gets a wave per block instead: its blocks in sequence were 10-38 us per level of pure latency.
// Registers capped for THREE waves per SIMD (round 6): the fused head of a frozen chain (k_level<2>: sampler -> volume -> pan in
// registers) took 191 and ran two — config 2 on the levels alone 1.64-1.68e11 -> 1.76-1.83e11 voice-samples/s; four (128 registers)
// spills its way to 1.50e11.  (-DLEVEL_MINW=n: experiments.)
#ifndef LEVEL_MINW
#define LEVEL_MINW 3
#endif
template <int SET>
__global__ __launch_bounds__(WAVE* WPB, LEVEL_MINW) void k_level(DevView v, const int* __restrict__ level_nodes, int n_nodes,
                                                      uint32_t cmd_block0, uint32_t K, uint32_t bpw, uint32_t walkers) {
    int w = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (w >= n_nodes) return;
    const int node = level_nodes[w];
    const int kind = v.nodes[node].kind;
    if (kind_set(kind) != SET) return;  // another instantiation's node
    const uint32_t b0 = blockIdx.y * bpw;
    const uint32_t b1 = b0 + bpw < K ? b0 + bpw : K;
    if (kind_is_stateful(kind)) {
        const uint8_t fz = v.frozen ? v.frozen[node] : (uint8_t)0;
        if (fz) {
            const bool adv = fz == 2;  // playing sampler: per-block playhead in closed form, the last block stores the state
            const bool spat = fz == 3;
            // blocks an upstream wave rendered on into this node, in registers (fz_links): nothing left to do for them here
            // (bpw divides 32: the range never straddles a word)
            uint32_t skip = 0u;
            if (v.chain_done && fz == 1 && b1 > b0)
                skip = (v.chain_done[(size_t)node * v.chain_words + (b0 >> 5)] >> (b0 & 31u)) & (b1 - b0 >= 32u ? ~0u : (1u << (b1 - b0)) - 1u);
            // (a link whose every block of this range was rendered upstream has nothing to read here — its block-0 wave still patches the
            //  smoothers, frozen_finish)
            if (b1 > b0 && skip == (b1 - b0 >= 32u ? ~0u : (1u << (b1 - b0)) - 1u) && b0 != 0) return;
            uint32_t todo = bpw > 1 ? frozen_fast<SET>(v, node, fz, b0, b1, K, skip) : ~0u;  // (what is left goes block by block)
            todo &= ~skip;
            for (uint32_t b = b0; b < b1; ++b)
                if ((todo >> ((b - b0) & 31u)) & 1u) node_process_wave<SET>(v, node, b, cmd_block0 + b, adv && b + 1 == K, adv ? b : 0u, adv || spat);
            if (spat) {
                if (b0 == 0) spatial_finish(v, node, K);
            } else if (!adv && b0 == 0) {
                frozen_finish(v, node, K);
            }
            return;
        }
        if (blockIdx.y != 0) return;
        if constexpr (SET == 1) {  // (a bus filter / delay the batch walkers take: k_bus_iir, launched next to this kernel)
            if (walkers && (biquad_walk_ok(v, v.nodes[node], cmd_block0, K) || delay_walk_ok(v, v.nodes[node], cmd_block0, K))) return;
        }
        for (uint32_t b = 0; b < K; ++b) node_process_wave<SET>(v, node, b, cmd_block0 + b);
    } else {
        for (uint32_t b = b0; b < b1; ++b) node_process_wave<SET>(v, node, b, cmd_block0 + b);
    }
}

// The batch walkers' kernel: one wave per node of the level that is a bus biquad / delay they take (k_level<1>, launched with
// `walkers` set, leaves exactly those nodes alone).  A kernel of its own so that the walkers' registers (prefetched chunks)
// do not set k_level<1>'s occupancy.
__global__ __launch_bounds__(WAVE* WPB) void k_bus_iir(DevView v, const int* __restrict__ level_nodes, int n_nodes, uint32_t cmd_block0,
                                                        uint32_t K) {
    const int w = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (w >= n_nodes) return;
    const NodeDesc nd = v.nodes[level_nodes[w]];
    const int nch = nd.n_in < nd.n_out ? nd.n_in : nd.n_out;
    if (biquad_walk_ok(v, nd, cmd_block0, K)) {
        if (nch == 2) biquad_walk<2>(v, nd, K);
        else biquad_walk<1>(v, nd, K);
    } else if (delay_walk_ok(v, nd, cmd_block0, K)) {
        if (nch == 2) delay_walk<2>(v, nd, K);
        else delay_walk<1>(v, nd, K);
    }
}

// B1: one node on scratch buffers (single wave)
__global__ __launch_bounds__(WAVE) void k_single_node(DevView v, int node_idx) { node_process_wave<3>(v, node_idx, 0, 0); }

// ------------------------------------------------------------------ state init / graph I/O edges
struct StateInit {
    int index;
    int pad;
    NodeState st;
};
__global__ void k_scatter_ext(float* ext, const ExtInitHost* __restrict__ items, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ExtInitHost it = items[i];
    for (uint32_t j = 0; j < it.n; ++j) ext[it.off + j] = it.v[j];
}
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

