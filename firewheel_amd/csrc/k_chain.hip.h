// k_chain.hip.h — part of the single device translation unit fwgpu_kernels.hip (included inside namespace fwgpu).
// Fused chain plan (config 3): k_chain.
#pragma once

// ------------------------------------------------------------------ fused chain plan (config 3): k_chain
// Voices of the shape  sampler -> [volume|pan]* -> FX -> [volume|pan]* -> leaf SumNode, FX = B | BB | D | BD | BBD | DB | DBB (B = biquad,
// D = delay; fwgpu_plan_detect.cpp walk_voice_chain).  The biquad (SPEC: DF1 in
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
// Round 6: gain stages in FRONT of the filters are multiplied in by S1 (VoiceDesc::n_pre); a delay line in FRONT of the biquads
// (fx_order 1) is read-modify-written by S1 instead of S3a (both loops: in the steady-call loop such a lane's ring requests run two
// tiles ahead of S1 instead of behind it); a
// SECOND biquad (an EQ cascade) is two more pipeline stages in the <NQ, true> instantiation (six tile buffers, S3a / S3b two tiles
// later): S1b — the worker lanes form its feed-forward sums on the first biquad's output, in place — and S2b — its recurrence, a
// second serial wave.  Measured (clock64 per role, profiles/r06_chain_roles.txt): a serial wave issues one instruction per ~7 cycles
// whatever the instruction, the recurrence costs ~25 cycles per frame (two dependent fmas), and a wave that did the second filter's
// five feed-forward operations per frame as well took 45 — longer than a whole step of the other roles; four other arrangements of
// that work on ONE wave (frame-parallel feed-forward phase, lane pairs, hand-interleaved with the recurrence) were slower still.
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
// stage-skip switches (env FWGPU_CHAIN_SKIP) exist in profiling builds only; the product kernel has none of them
#ifdef FW_CHAIN_TRACE
#define CH_SKIP(bit) (fv.dbg & (bit))
#else
#define CH_SKIP(bit) false
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

// a quad of an interleaved stereo 16-bit source as it was loaded (4 frames x {L, R} x 16 bit = one dwordx4): this channel's halves
// as f32 (core/sample_resource.rs:338-345: the conversions of k_leaf.hip.h cvt_i16 / cvt_u16).  c16: 1 = i16, 2 = u16
__device__ __forceinline__ v4f chain_cvt16(const v4f raw, const int c16, const int ch) {
    v4f o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int w = __float_as_int(raw[e]);
        const int h = ch ? (w >> 16) : w;
        o[e] = c16 == 1 ? cvt_i16(h) : cvt_u16(h);
    }
    return o;
}
// ... when the whole wave holds ONE 16-bit format (wave-uniform `u16`): no per-lane select, three operations per sample instead of
// eight — a worker wave issues an instruction every ~12 cycles, and the generic form above cost config 3 on 16-bit sources 13 %
template <bool U16>
__device__ __forceinline__ v4f chain_cvt16_all(const v4f raw, const int ch) {
    v4f o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int w = __float_as_int(raw[e]);
        const int h = ch ? (w >> 16) : w;
        o[e] = U16 ? cvt_u16(h) : cvt_i16(h);
    }
    return o;
}

struct ChainInfo {  // what a worker lane carries from S1 of a tile to S3a of the same tile (two steps later)
    uint32_t flags;                // VB_* of the tile's block, ramp bits included
    float g[FW_CHAIN_STAGES - 1];    // this channel's constant post-gain stages (1..)
};

// S3b for a workgroup whose 32 rows are 32/P full leaves of P ports each with nothing to skip: straight-line adds in port
// order per leaf (static register indices, no per-row tests)
template <int P>
__device__ __forceinline__ void chain_mix_uniform(const v4f (&x)[32], uint32_t silent_rows, float* bus_blk, uint8_t* flag_blk,
                                                  const int* g_out, int ch, int stride, bool write_flag) {
#pragma unroll
    for (int l = 0; l < 32 / P; ++l) {
        v4f acc = x[l * P];  // sum.rs:117 copy port 0 (also when silent: a cleared buffer)
#pragma unroll
        for (int j = 1; j < P; ++j) acc = acc + x[l * P + j];
        const uint32_t rows = (P == 32 ? 0xffffffffu : ((1u << P) - 1u)) << (l * P);
        const bool all_silent = (silent_rows & rows) == rows;  // sum.rs:52-56
        const int ob = g_out[l];
        *(v4f*)(bus_blk + (size_t)(ob + ch) * stride) = all_silent ? splat(0.f) : acc;
        if (write_flag) flag_blk[ob + ch] = all_silent ? 1 : 0;
    }
}

// SITES: some voice of the plan has a gain stage BETWEEN two filters or a hard clip (the instantiation with the five-site stage logic; the
// other one knows "in front of the first filter" and "behind the last", multiplications only, and keeps its registers: the site tables cost
// the one-biquad kernel 28 bytes of scratch per lane when they were unconditional)
template <int NQ, bool BQ2, bool SITES>
__global__ __launch_bounds__(CH_THREADS) void k_chain(FusedView fv, int K, uint32_t cmd_block0) {
    constexpr int TT = 64 * NQ;        // frames per tile
    constexpr int PITCH = TT + 4;      // floats per voice row: + 4 -> the 32 S2 lanes' b128 reads are conflict-free
    constexpr int LF = 4 * NQ;         // frames per worker lane
    constexpr int NBUF = BQ2 ? 6 : CH_NBUF;  // tile buffers in flight
    constexpr int LAG3 = BQ2 ? 4 : 2;        // S3a runs this many tiles behind S1 (two-biquad instantiation: S1b at 2 and S2b at 3 sit in between)
    constexpr int LAG4 = LAG3 + 1;           // ... and S3b, the mixer, one more
#define CH_BUF(t) ((unsigned)((t) + NBUF) % (unsigned)NBUF)  // buffer of tile t (t >= -NBUF)
    __shared__ float tile[NBUF][32][PITCH];  // row = voice (this workgroup's channel)
    __shared__ uint32_t silf[NBUF][32];      // chain output cleared + flagged silent (VB_SILENT) per voice
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    // up to 8 consecutive leaves, <= 32 voices: this workgroup's rows.  Per-leaf fields are read through LDS (indexing a
    // register copy of the record by a run-time leaf index would push it to scratch)
    const ChainGroup* gp = &fv.groups[blockIdx.x];
    struct {
        int first_voice, n_voices, n_leaves, uniform_ports;
        uint32_t start_mask, masked_rows;
    } grp = {gp->first_voice, gp->n_voices, gp->n_leaves, gp->uniform_ports, gp->start_mask, gp->masked_rows};
    __shared__ int g_out[CH_GROUP_LEAVES];
    __shared__ uint32_t g_rows[CH_GROUP_LEAVES];  // row mask of each leaf
    if (threadIdx.x < CH_GROUP_LEAVES) {
        const int l = threadIdx.x < (unsigned)grp.n_leaves ? (int)threadIdx.x : 0;
        g_out[threadIdx.x] = gp->out_buf[l];
        g_rows[threadIdx.x] = (gp->ports[l] >= 32 ? 0xffffffffu : ((1u << gp->ports[l]) - 1u)) << gp->row0[l];
    }  // (made visible by the __syncthreads_or below)
    const int ch = blockIdx.y;  // L and R never meet before the mix bus: one workgroup per (leaf, channel)
    const int ports = grp.n_voices;  // voice rows in use
    const int frames = fv.frames;
    const int tpb = frames / TT;  // the plan guarantees frames % TT == 0
    const int n_tiles = K * tpb;
    // A workgroup's waves are dealt to the 4 SIMDs round-robin, so waves w, w+4, w+8 share a SIMD (measured: HW_ID).
    // The serial wave (2) gets a SIMD to itself — waves 6 and 10 only take part in the barriers — the mixer (11)
    // shares one with two workers, the other six workers fill the remaining two SIMDs.
    const bool is_serial = wave == 2;
#ifndef CH_S2B_WAVE
#define CH_S2B_WAVE 6
#endif
    // the second biquad's recurrence: wave 6 = a wave of its own on the first one's SIMD (no worker there).  Measured alternatives
    // (profiles/r06_chain_roles.txt): 11 = on the SIMD of workers 3 and 7, the mixer then on wave 6: ~1 % faster, not worth the second
    // role map; BOTH chains interleaved frame by frame in wave 2: 58 cycles per frame instead of 2 x 24 — one wave issues an instruction
    // every ~12 cycles however independent the next one is, so two chains in one wave simply take twice as long.
    const bool is_mixer = wave == ((BQ2 && CH_S2B_WAVE == 11) ? 6 : 11);
    const bool is_serial2 = BQ2 && wave == CH_S2B_WAVE;
    const bool is_idle = (wave == 6 || wave == 10) && !is_serial2 && !is_mixer;
    const bool is_worker = !is_serial && !is_mixer && !is_idle && !is_serial2;
    const int widx = wave - (wave > 2 ? 1 : 0) - (wave > 6 ? 1 : 0);  // 0..7 among the worker waves 0,1,3,4,5,7,8,9

    // ---- per-role persistent registers; worker lane = (voice v, frames [LF*q, LF*q + LF) of every tile),
    //      serial lane = voice v.  Everything both channels share (delay position / feedback / mix, biquad
    //      coefficients) is read from the ChainStart record k_voice_control wrote for this call and never written
    //      here: the two workgroups of a leaf are not ordered against each other.
    const int wl = widx * WAVE + lane;
    const int v = is_worker ? (wl >> 4) : lane;
    const int q = wl & 15;
    const bool active = v < ports && (is_worker || ((is_serial || is_serial2) && lane < 32));
    const int voice = grp.first_voice + (active ? v : 0);
    const VoiceDesc vd = fv.voices[voice];
    const bool has_bq = active && vd.bq_state >= 0, has_dl = active && vd.dl_state >= 0;
    ChainStart cs = fv.chain_start[voice];
    // Lazy call (round 6): no control kernel ran.  A voice's block records follow from its LazyRec and the block index (as in the leaf
    // kernel's lazy instantiation), its gains are the record's, and what the ChainStart holds is read from node state instead — which
    // nothing has written since the control kernel that made the records (no message since, by the host's lazy test): coefficients in
    // the ext pool, the delay's parameters, its position the blocks of the lazy calls so far further on (k_lazy_flush moves it for good).
    const bool lazy = fv.lazy_chain != 0;
    if (lazy) {
        if (vd.bq_state >= 0) {
            const float* co = fv.ext + fv.states[vd.bq_state].ext_off;
#pragma unroll
            for (int j = 0; j < 5; ++j) cs.co[j] = co[j];
        }
        if (BQ2 && vd.bq2_state >= 0) {
            const float* co = fv.ext + fv.states[vd.bq2_state].ext_off;
#pragma unroll
            for (int j = 0; j < 5; ++j) cs.co2[j] = co[j];
        }
        if (vd.dl_state >= 0) {
            const NodeState* ds = &fv.states[vd.dl_state];
            const uint64_t Dl = ds->loop_end;
            cs.pos = (uint32_t)((ds->playhead + (fv.lazy_blk0 % Dl) * (uint64_t)frames) % Dl);
            cs.fb = ds->p0;
            cs.mix = ds->p1;
            cs.dry = ds->gain;
        }
    }
    auto chain_ref = [&](const int pvoice, const int pk) -> VoiceRef {
        if (!lazy) return fv.refs[ref_index(pvoice, pk, fv.ref_kgroups)];
        const LazyRec* lr = fv.lazy + pvoice;
        VoiceRef r;
        const uint32_t mode = (uint32_t)lr->mode, lfr = lr->frames, bpf = lr->bpf;
        const uint64_t bi = fv.lazy_blk0 + (uint64_t)pk;
        uint64_t fo = 0;
        if (mode == 1u) fo = (uint64_t)(uint32_t)(((uint64_t)lr->r0b + bi) % lr->q) * lfr;
        else if (mode == 2u) fo = lr->off0 + bi * lfr;
        r.src_l = (const float*)(lr->base + fo * bpf);
        r.r_delta = lr->r_delta;
        r.flags_gset = lr->flags_gset & ~0xff00u;
        return r;
    };
    auto chain_gsets = [&](const int pvoice) -> const GainSet* { return lazy ? &fv.lazy[pvoice].g : &fv.gsets[(size_t)pvoice * FW_GSETS]; };
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
    // the second biquad of an EQ cascade: coefficients and state live on its own wave (S2b)
    const bool has_bq2 = BQ2 && active && vd.bq2_state >= 0;
    float c0 = cs.co2[0], c1 = cs.co2[1], c2 = cs.co2[2], d1 = cs.co2[3], d2 = cs.co2[4];
    float* bq2_st = nullptr;
    float z1 = 0.f, z2 = 0.f;  // its y[n-1], y[n-2]
    // its x[n-1], x[n-2] at the start of the next tile, per voice row: read by the row's first worker lane (S1b), written by its last
    __shared__ float bq2_u[BQ2 ? 32 : 1][2];
    if (has_bq2 && is_serial2) {
        bq2_st = fv.ext + fv.states[vd.bq2_state].ext_off + 5 + 4 * ch;
        bq2_u[lane][0] = bq2_st[0];
        bq2_u[lane][1] = bq2_st[1];
        z1 = bq2_st[2];
        z2 = bq2_st[3];
    }  // (made visible to the workers by the __syncthreads_or below)
    const bool dl_first = has_dl && vd.fx_order == 1;                // the delay line in front of the biquads: S1 does its read-modify-write
    // ---- where each gain-like stage of the voice is applied (round 6: stages in front of, between and behind the filters).  Five sites:
    //   A  S1, on the source               B  S1, behind a delay-first voice's delay line      C  S1b, in front of the second biquad
    //   D  S3a, in front of the delay line  E  S3a, last (behind every filter)
    // A stage's position among the filters (VoiceDesc::n_pre / n_mid) and the voice's filter order give its site; its kind (gain or hard
    // clip) comes from the voice's stage program.  Sites a wave's voices do not use cost it nothing (wave-uniform masks).
    enum { SITE_A = 0, SITE_B = 1, SITE_C = 2, SITE_D = 3, SITE_E = 4 };
    const int n_pre = active ? vd.n_pre : 0;
    uint32_t site_bits = 0u;  // SITES: 3 bits of site + 1 bit "hard clip" per stage (ONE register: per-stage arrays spilled)
    uint32_t my_sites = 0u;
    if constexpr (SITES) {
        const int n_mid1 = active ? (vd.n_mid & 0xff) : 0, n_mid2 = active ? ((vd.n_mid >> 8) & 0xff) : 0;
        const uint32_t prog = active ? fv.progs[voice] : 0u;
        const bool two_bq = active && vd.bq2_state >= 0;
#pragma unroll
        for (int j = 0; j < FW_CHAIN_STAGES - 1; ++j) {
            const int posn = j < n_pre ? 0 : (j < n_pre + n_mid1 ? 1 : (j < n_pre + n_mid1 + n_mid2 ? 2 : 3));
            int st = SITE_E;
            if (posn == 0) st = (has_bq || has_dl) ? SITE_A : SITE_E;  // (a dry voice of a chain plan: everything at the end, as ever)
            else if (posn == 1) st = dl_first ? SITE_B : (two_bq ? SITE_C : SITE_D);
            else if (posn == 2) st = dl_first ? SITE_C : SITE_D;
            const uint32_t clip = ((prog >> (4 * j)) & 15u) == SK_CLIP ? 8u : 0u;
            site_bits |= ((uint32_t)st | clip) << (4 * j);
            if (active && j < vd.n_stages) my_sites |= (1u << st) | (clip ? 32u : 0u);
        }
    } else {
        my_sites = n_pre > 0 ? 1u : 0u;
    }
    auto stage_site = [&](const int j) -> int {
        if constexpr (SITES) return (int)((site_bits >> (4 * j)) & 7u);
        else return j < n_pre ? SITE_A : SITE_E;
    };
    auto stage_clip = [&](const int j) -> bool {
        if constexpr (SITES) return ((site_bits >> (4 * j)) & 8u) != 0u;
        else return false;
    };
    const bool any_A = __syncthreads_or((int)(my_sites & 1u)) != 0;
    const bool any_B = SITES && __syncthreads_or((int)(my_sites & 2u)) != 0, any_C = SITES && __syncthreads_or((int)(my_sites & 4u)) != 0,
               any_D = SITES && __syncthreads_or((int)(my_sites & 8u)) != 0, any_clip = SITES && __syncthreads_or((int)(my_sites & 32u)) != 0;
    const bool any_dlf = __syncthreads_or(dl_first ? 1 : 0) != 0;  // (a workgroup without delay-first voices carries none of their code path)
    uint32_t D = 1, pos = cs.pos;
    float fb = cs.fb, mix = cs.mix, dry = cs.dry;
    float* ring = nullptr;
    if (has_dl && is_worker) {
        const NodeState* ds = &fv.states[vd.dl_state];
        D = (uint32_t)ds->loop_end;
        ring = fv.ext + ds->ext_off + (size_t)ch * D;
    }
    const bool ring_pref = has_dl && D >= 2u * TT && !CH_SKIP(16);
    const bool any_bq = __syncthreads_or(has_bq ? 1 : 0) != 0;

    // ---- steady call?  Every voice of the leaf keeps ONE descriptor shape for all K blocks — constant gains from one
    // gain set, planar f32 (or cleared) source that is contiguous in each block or wraps once at its loop end
    // (nodes/sampler.rs:445-484) — has both a biquad and a delay of >= 3 tiles, and no message is pending: the workers
    // then run the branch-free loop below (loads two tiles ahead, exact vmcnt waits) instead of the general one.
    // Per (voice, block) the scan parks in LDS: the address of frame 0, the frame the loop wraps at (or "never") and
    // the address frames past the wrap are relative to.
    __shared__ unsigned long long srcp[32][CH_FAST_KMAX];
    __shared__ unsigned long long srcp1[32][CH_FAST_KMAX];
    __shared__ uint32_t wrap_at[32][CH_FAST_KMAX];
    __shared__ uint32_t vcls[32];  // per voice: bit c set = some block of the call fetches source class c (SF_*): exactly one for a steady voice
    if (threadIdx.x < 32) vcls[threadIdx.x] = 0u;
    __syncthreads();
    bool fast_ok = K <= CH_FAST_KMAX && !(fv.dbg & 32);  // FWGPU_CHAIN_SKIP=32: A/B against the general loop
    if (fv.n_cmds && active) {
        // messages for THIS leaf's biquads / delays in this call (coefficients, feedback, mix) are replayed block by
        // block by the general loop; messages for other leaves, for master nodes or for later calls do not concern this
        // workgroup (gain / sampler messages show up in the descriptors scanned below)
        const uint32_t b1 = cmd_block0 + (uint32_t)K;
        if (has_bq) {
            const int i = chain_cmd_lower_bound(fv.cmds, fv.n_cmds, vd.bq_state, cmd_block0);
            fast_ok = fast_ok && !(i < fv.n_cmds && fv.cmds[i].state == vd.bq_state && fv.cmds[i].block < b1);
        }
        if (has_dl) {
            const int i = chain_cmd_lower_bound(fv.cmds, fv.n_cmds, vd.dl_state, cmd_block0);
            fast_ok = fast_ok && !(i < fv.n_cmds && fv.cmds[i].state == vd.dl_state && fv.cmds[i].block < b1);
        }
        if (has_bq2) {
            const int i = chain_cmd_lower_bound(fv.cmds, fv.n_cmds, vd.bq2_state, cmd_block0);
            fast_ok = fast_ok && !(i < fv.n_cmds && fv.cmds[i].state == vd.bq2_state && fv.cmds[i].block < b1);
        }
    }
    for (int i = threadIdx.x; i < ports * K; i += CH_THREADS) {
        const int pv = i / K, pk = i - pv * K, pvoice = grp.first_voice + pv;
        const VoiceRef rk = chain_ref(pvoice, pk);
        const uint32_t fk = rk.flags_gset & 0xffu, kind = fk & (VB_SIMPLE | VB_WRAP | VB_TAIL_ZERO);
        const uint32_t per_voice = VB_SILENT | VB_MONO | VB_SRC_ZERO;  // what must not change inside the call
        bool ok = lazy || ((fk ^ fv.refs[ref_index(pvoice, 0, fv.ref_kgroups)].flags_gset) & per_voice) == 0u;
        const float* a0 = nullptr;
        const float* a1 = nullptr;
        uint32_t wr = 0xffffffffu;
        if (kind == VB_SIMPLE) {
            const uint32_t cls = (rk.flags_gset >> 16) & 7u;
            const bool c16 = cls == SF_I_I16 || cls == SF_I_U16;  // interleaved stereo 16-bit: 4 bytes per frame, the channel is a half-word
            ok = ok && ((rk.flags_gset >> 8) & 0xffu) == 0u && (cls == SF_P_F32 || c16 || (fk & VB_SRC_ZERO));
            a0 = rk.src_l + ((ch && !c16) ? rk.r_delta : 0u);  // r_delta = 0 for a mono sample
            if (!(fk & VB_SRC_ZERO)) atomicOr(&vcls[pv], 1u << cls);
        } else if (kind == VB_WRAP && !(fk & VB_SRC_ZERO)) {
            const VoiceBlk* d = &fv.blks[(size_t)pk * fv.n_voices + pvoice];
            const GainSet* gs = &fv.gsets[(size_t)pvoice * FW_GSETS];
            ok = ok && (d->flags >> VB_RAMP_SHIFT) == 0u && d->sample >= 0;
#pragma unroll
            for (int j = 0; j < FW_CHAIN_STAGES; ++j) ok = ok && (j >= fv.n_gain_stages || d->g[j][ch] == gs->g[j][ch]);
            if (ok) {
                const SampleDesc sd = fv.samples[d->sample];
                const bool c16 = (sd.format == FMT_I_I16 || sd.format == FMT_I_U16) && sd.channels == 2 && !(fk & VB_MONO);
                ok = (sd.format == FMT_P_F32 || c16) && sd.frames < 0xffffffffull;
                atomicOr(&vcls[pv], 1u << (c16 ? (sd.format == FMT_I_I16 ? SF_I_I16 : SF_I_U16) : SF_P_F32));
                const float* base = (const float*)sd.data + ((fk & VB_MONO) || ch == 0 || c16 ? 0ull : sd.frames);
                a0 = base + d->off0;
                wr = d->n1;
                a1 = base + d->off1 - wr;  // frame f >= wr of the block is a1[f]
            }
        } else {
            ok = false;
        }
        fast_ok = fast_ok && ok;
        srcp[pv][pk] = (unsigned long long)a0;
        srcp1[pv][pk] = (unsigned long long)a1;
        wrap_at[pv][pk] = wr;
    }
    if (active && has_dl && is_worker) fast_ok = fast_ok && D >= 3u * TT;  // (voices without a biquad / delay pass through)
    __syncthreads();  // vcls complete
    const uint32_t my_cls = active ? vcls[v] : 0u;
    fast_ok = fast_ok && (my_cls & (my_cls - 1u)) == 0u;  // one source class per voice and call
    const int c16 = (my_cls >> SF_I_I16) & 1u ? 1 : ((my_cls >> SF_I_U16) & 1u ? 2 : 0);
    const bool any16 = __syncthreads_or(c16 ? 1 : 0) != 0;
    // (per wave: every lane with a source holds the same 16-bit format — the usual bank — or formats are mixed)
    const bool wave_all_i16 = __ballot(active && c16 != 1 && my_cls != 0u) == 0ull, wave_all_u16 = __ballot(active && c16 != 2 && my_cls != 0u) == 0ull;
    const bool wg_fast = __syncthreads_and(fast_ok ? 1 : 0) != 0;
    const int n_steps = wg_fast ? ((n_tiles + LAG4 + 1) & ~1) : n_tiles + LAG4;  // the fast loop is unrolled by two
    if (threadIdx.x == 0) atomicAdd(fv.chain_stats + (wg_fast ? 0 : 1), 1ull);  // fwgpu_plan_chain_stats (tests)

    // compute-side block registers (block of tile s) and issue-side ones (block of tile s+1, one step ahead)
    float g0 = 1.f;
    ChainInfo inf0, inf1, inf2, inf3, inf4;  // tiles s, s-1, s-2 (, s-3, s-4: S3a's in the two-biquad instantiation)
    inf0.flags = inf1.flags = inf2.flags = inf3.flags = inf4.flags = VB_SRC_ZERO | VB_SIMPLE;
#pragma unroll
    for (int j = 0; j < FW_CHAIN_STAGES - 1; ++j) inf0.g[j] = inf1.g[j] = inf2.g[j] = inf3.g[j] = inf4.g[j] = 1.f;
    // the general loop's form of a site: the stages of site X on quad jq of tile (block kk, tile tt) — per-block constants from the tile's
    // ChainInfo, per-frame ramps from the ramp rows, a clip's threshold, or the sentinel of a muted stage between two filters (-> +0.0)
    auto gsite1 = [&](const int X, v4f x, const ChainInfo& I, const int kk, const int tt, const int jq) -> v4f {
        const uint32_t rbits = I.flags >> VB_RAMP_SHIFT;
#pragma unroll
        for (int g = 1; g < FW_CHAIN_STAGES; ++g) {
            if (g >= fv.n_gain_stages) break;
            if (stage_site(g - 1) != X) continue;
            const float gcv = I.g[g - 1];
            if ((rbits >> (2 * g + ch)) & 1u) {
                const int f0 = tt * TT + LF * q + 4 * jq;
                const float* rb = fv.ramps + ((size_t)kk * fv.n_voices + voice) * (size_t)fv.ramp_slots * (size_t)fv.stride + f0;
                x = x * *(const v4f*)(rb + (size_t)(2 * g + ch) * fv.stride);
            } else if (SITES && gcv < 0.f) {
                x = splat(0.f);
            } else if (stage_clip(g - 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = clipf(x[e], gcv);
            } else {
                x = x * gcv;
            }
        }
        return x;
    };
    // S1b (two-biquad instantiation), worker lanes, tile s-2: the second biquad's feed-forward sums on the first one's output, in place:
    //   ff[n] = ((c0*x[n]) + (c1*x[n-1])) + (c2*x[n-2]), unfused (k_generic.hip.h K_BIQUAD).  x[n-1], x[n-2] of a lane's first frame are
    // the two floats in front of it in the row (lane q = 0: the tile before's last two, bq2_u).  A voice's 16 lanes sit in ONE wave and
    // every lane reads before any lane writes (DS operations of a wave retire in order): no barrier inside the stage.
    auto ff2_stage = [&](const int t_tile, const bool real, auto&& pre) {  // pre(x, j): the gain stages between the two biquads, on quad j
        float* const r2 = &tile[CH_BUF(t_tile)][v][LF * q];
        // (the neighbour's last QUAD, through the same stages between the biquads as this lane's own: the second filter's x[n-1], x[n-2]
        //  are ITS inputs, behind those stages — what bq2_u holds, too; lane 0: a valid address, the value is not used)
        v4f hq = *(const v4f*)(r2 - (q ? 4 : 0));
        pre(hq, q ? -1 : 0);  // (quad index -1: the frames in front of this lane's; lane 0's value is not used — and must not be fetched from in front of a ramp row)
        float h1 = q ? hq[3] : bq2_u[v][0], h2 = q ? hq[2] : bq2_u[v][1];
        // the neighbour's two floats are IN REGISTERS before any store of this stage is issued: lane q - 1's last quad overwrites them, and to
        // the compiler — which reasons per lane — that store and this load touch different addresses and may change places
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h1), "+v"(h2) : : "memory");
        const bool on = real && has_bq2;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {  // (quad by quad: the worker lanes have no registers to spare for the whole tile slice)
            v4f x = *(const v4f*)(r2 + 4 * j);
            pre(x, j);
            const v4f x1v = (v4f){h1, x[0], x[1], x[2]}, x2v = (v4f){h2, h1, x[0], x[1]};
            const v4f ff = ((x * c0) + (x1v * c1)) + (x2v * c2);
            h1 = x[3];
            h2 = x[2];
            if (on) *(v4f*)(r2 + 4 * j) = ff;
        }
        if (on && q == 15) {
            bq2_u[v][0] = h1;
            bq2_u[v][1] = h2;
        }
    };
    VoiceRef ref_n;        // descriptor of the block that starts two tiles ahead (in flight)
    ref_n.src_l = nullptr;
    ref_n.r_delta = 0;
    ref_n.flags_gset = VB_SRC_ZERO | VB_SIMPLE;
    float gs_n[FW_CHAIN_STAGES];  // this channel's gains of the issue-side block (in flight)
#pragma unroll
    for (int j = 0; j < FW_CHAIN_STAGES; ++j) gs_n[j] = 1.f;
    const float* nb_src = nullptr;  // issue-side block: this channel's source of frame 0, VB_* flags
    uint32_t nb_flags = VB_SRC_ZERO | VB_SIMPLE;
    int nb_c16 = 0, cur_c16 = 0;    // 16-bit interleaved source (1 = i16, 2 = u16) of the issue-side block / of the block S1 computes
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
    int k1 = 0, t1 = 0, k2 = 0, t2 = 0, k3 = 0, t3 = 0, k4 = 0, t4 = 0, kla = 0, tla = 0, kf = 0, tf = 0;
    const uint64_t port_mask = mask_all_silent_bits(ports);

    // issue the HBM loads of tile `la` (= the tile S1 computes in the next step); at a block start first adopt the
    // block's descriptor (in flight since the previous step) and request its gain set
    auto issue_source = [&]() {
        if (tla == 0) {
            nb_flags = ref_n.flags_gset & 0xffu;
            const uint32_t ncls = (ref_n.flags_gset >> 16) & 7u;
            nb_c16 = ncls == SF_I_I16 ? 1 : (ncls == SF_I_U16 ? 2 : 0);
            nb_src = ref_n.src_l + ((ch && !nb_c16) ? ref_n.r_delta : 0u);  // r_delta = 0 for a mono sample (sampler.rs:546-551)
            if (nb_flags & VB_SIMPLE) {
                const GainSet* gs = chain_gsets(voice) + ((ref_n.flags_gset >> 8) & 0xffu);
#pragma unroll
                for (int j = 0; j < FW_CHAIN_STAGES; ++j) gs_n[j] = gs->g[j][ch];
            }
        }
        if ((nb_flags & VB_SIMPLE) && !(nb_flags & VB_SRC_ZERO) && !CH_SKIP(4)) {
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
        if (tla == 0 && kla < K) ref_n = chain_ref(voice, kla);
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
    if (is_worker && active && !wg_fast) {  // prologue = the issue halves of steps -2 and -1
        ref_n = chain_ref(voice, 0);
        issue_source();
        if (dl_first && ring_pref) load_ring();  // (delay-first voices: S1 of tile 0 consumes them)
    }

#ifdef CH_PROF  // experiments: per-role busy cycles (clock64 = shader clock, wall_clock64 = 100 MHz) of workgroup (0, 0), printed at the end
    unsigned long long prof_busy = 0, prof_t0 = clock64(), prof_w0 = wall_clock64();
#define CH_PROF_BEGIN() const unsigned long long prof_a = clock64()
#define CH_PROF_END() prof_busy += clock64() - prof_a
#else
#define CH_PROF_BEGIN() do { } while (0)
#define CH_PROF_END() do { } while (0)
#endif
    // One loop per role (same number of barriers in each) so that the register allocation of a role does not
    // carry the other roles' loop state.
    if (is_worker && wg_fast) {
        // ================= steady call: branch-free worker steps.
        // Every step issues exactly NQ source loads, NQ ring stores and NQ ring loads, unconditionally (lanes / tiles
        // that have nothing to fetch or store use a dummy address), so the compiler's s_waitcnt insertion sees one
        // path and emits exact vmcnt(N) waits: the source of tile s+2 and the ring slots of tile s are requested in
        // step s and stay in flight for two whole steps (two static register sets, loop unrolled by two).  A quad
        // that straddles the end of its ring (once per lap) is fixed up on a rare path with plain in-step accesses.
        const uint32_t vflags = (lazy ? fv.lazy[voice].flags_gset : fv.refs[ref_index(voice, 0, fv.ref_kgroups)].flags_gset) & 0xffu;  // per-voice bits only are used
        const GainSet* gsp = chain_gsets(voice);
        const float g0f = gsp->g[0][ch];
        // stage j + 1's constant: a gain (the sentinel -1.0f from a muted stage between two filters: a cleared buffer) or a clip threshold
        float gc[FW_CHAIN_STAGES - 1];
        bool my_mute = false;
#pragma unroll
        for (int j = 0; j < FW_CHAIN_STAGES - 1; ++j) {
            gc[j] = gsp->g[j + 1][ch];
            my_mute = my_mute || (SITES && active && j < vd.n_stages && gc[j] < 0.f);
        }
        const bool wave_mute = SITES && __ballot(my_mute) != 0ull;
        // the stages of site X on a quad: every lane multiplies by its stage's gain there and by 1.0f (exact) elsewhere; clips and mutes
        // are per-lane selects the wave only pays for when one of its voices has one
        auto ssite1 = [&](const int X, v4f x) -> v4f {
#pragma unroll
            for (int g = 0; g < FW_CHAIN_STAGES - 1; ++g) {
                if (g + 1 >= fv.n_gain_stages) break;
                const bool here = stage_site(g) == X;
                const float gm = here ? gc[g] : 1.f;
                v4f y = x * gm;
                if (any_clip) {
                    v4f c;
#pragma unroll
                    for (int e = 0; e < 4; ++e) c[e] = clipf(x[e], gc[g]);
                    y = (here && stage_clip(g)) ? c : y;
                }
                if (wave_mute) y = (here && gc[g] < 0.f) ? splat(0.f) : y;
                x = y;
            }
            return x;
        };
        auto ssite = [&](const int X, v4f(&x)[NQ]) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) x[j] = ssite1(X, x[j]);
        };
        const bool src_zero = (vflags & VB_SRC_ZERO) != 0, silent = (vflags & VB_SILENT) != 0;
        float* const dummy = fv.chain_dummy + (size_t)threadIdx.x * (4 * NQ);
        const bool ringed = active && has_dl;  // this lane's voice has a delay line
        const bool o1 = ringed && dl_first;    // ... in FRONT of its biquads: S1 does the read-modify-write, on the tile it has just made
        const bool ring3 = ringed && !o1;      // ... behind them: S3a's
        const float* const rbase = ringed ? ring : dummy;
        const uint32_t Dv = ringed ? D : 0x7fffffffu;
        uint32_t pos_c = ringed ? pos : 0u, pos_i = pos_c;  // ring position of the tile S3a consumes / the tile requested
        int kli = 0, tli = 0;                               // (block, tile in block) of the next source tile to request
        int klc = 0, tlc = 0;                               // ... of the tile S1 computes
        v4f xsA[NQ], xsB[NQ], rgA[NQ], rgB[NQ];
        auto issue_src = [&](v4f(&xs)[NQ], int t) {
            const bool real = t < n_tiles && active && !src_zero;
            const uint32_t wr = wrap_at[v][kli];
            const float* a0 = (const float*)srcp[v][kli];
            const float* a1 = (const float*)srcp1[v][kli];
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const uint32_t f = (uint32_t)(tli * TT + LF * q + 4 * j);
                // a quad that straddles the wrap point is not fetched here: S1 patches it in when it consumes the tile
                const bool plain = real && !(f < wr && wr < f + 4u);
                const float* p = plain ? (f >= wr ? a1 : a0) + f : (const float*)dummy + 4 * j;
                asm volatile("" : "+v"(p));  // one opaque address: the select must not become conditional loads
                xs[j] = gload4(p);
            }
            const bool wrap = tli + 1 == tpb;
            tli = wrap ? 0 : tli + 1;
            kli = (wrap && kli + 1 < K) ? kli + 1 : kli;
        };
        auto slot_of = [&](uint32_t base, int j) -> uint32_t {
            uint32_t sl = base + (uint32_t)(LF * q + 4 * j);
            return sl >= Dv ? sl - Dv : sl;
        };
        auto issue_ring = [&](v4f(&rg)[NQ], int t) {
            const bool real = t >= 0 && t < n_tiles && ringed;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const uint32_t sl = slot_of(pos_i, j);
                const float* p = real ? rbase + (sl + 4u <= Dv ? sl : Dv - 4u) : (const float*)dummy;
                asm volatile("" : "+v"(p));
                rg[j] = *(gv4p)(uint64_t)p;  // global_load (the opaque address would otherwise be a flat access)
            }
            if (real) {
                pos_i += TT;
                if (pos_i >= Dv) pos_i -= Dv;
            }
        };
        auto wstep = [&](int s, v4f(&xs)[NQ], v4f(&rg)[NQ]) {
            const bool v1 = s < n_tiles;
            const bool v3 = ring3 && s >= LAG3 && s - LAG3 < n_tiles;  // S3a of a real tile of a voice with a delay line behind its filters
            CH_TRACE(0);
            CH_PROF_BEGIN();
            v4f yv[NQ];
            {
                const float* yrow = &tile[CH_BUF(s - LAG3)][v][LF * q];
#pragma unroll
                for (int j = 0; j < NQ; ++j) yv[j] = *(const v4f*)(yrow + 4 * j);
            }
            // ---- S1 on tile s (steps past the last tile compute on dummy data into a buffer nobody reads)
            {
                v4f x[NQ];
                {  // rare: this lane's quad contains the loop's wrap point (once per lap of the loop)
                    const uint32_t wr = wrap_at[v][klc];
                    const uint32_t fq = (uint32_t)(tlc * TT + LF * q);
                    const bool mine = v1 && active && !src_zero && fq < wr && wr < fq + 4u * NQ && (wr & 3u) != 0u;
                    if (__ballot(mine) != 0ull) {
                        if (mine) {
                            const float* a0 = (const float*)srcp[v][klc];
                            const float* a1 = (const float*)srcp1[v][klc];
#pragma unroll
                            for (int j = 0; j < NQ; ++j) {
                                const uint32_t f = fq + 4u * j;
                                if (f < wr && wr < f + 4u) {
                                    v4f t;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) t[e] = (f + e < wr ? a0 : a1)[f + e];
                                    asm volatile("" : "+v"(t));  // the wait for these loads stays inside the rare path
                                    xs[j] = t;
                                }
                            }
                        }
                    }
                    const bool wrapb = tlc + 1 == tpb;
                    tlc = wrapb ? 0 : tlc + 1;
                    klc = (wrapb && klc + 1 < K) ? klc + 1 : klc;
                }
                if (any16) {  // (wave-uniform: a plan of planar-f32 sources converts nothing)
                    if (wave_all_i16) {
#pragma unroll
                        for (int j = 0; j < NQ; ++j) xs[j] = chain_cvt16_all<false>(xs[j], ch);
                    } else if (wave_all_u16) {
#pragma unroll
                        for (int j = 0; j < NQ; ++j) xs[j] = chain_cvt16_all<true>(xs[j], ch);
                    } else {
#pragma unroll
                        for (int j = 0; j < NQ; ++j) xs[j] = c16 ? chain_cvt16(xs[j], c16, ch) : xs[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < NQ; ++j) x[j] = src_zero ? splat(0.f) : xs[j] * g0f;  // sampler.rs:530-533
                if (any_A) {  // gain stages in front of the filters: one rounding per stage, in schedule order (volume.rs:123-126)
                    ssite(SITE_A, x);
#pragma unroll
                    for (int j = 0; j < NQ; ++j) x[j] = src_zero ? splat(0.f) : x[j];
                }
                if (any_dlf) {
                    // delay-first voices (round 6): the ring slots of tile s were requested two steps ago into THIS register set (issue_ring
                    // below asks for tile s + 2 on their lanes); read-modify-write them here, branch-free like S3a's (dummy addresses for
                    // every other lane), and hand the wet signal to the feed-forward half
                    const bool v1d = o1 && v1;
                    uint32_t sl1[NQ];
                    bool str1 = false;
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        sl1[j] = slot_of(pos_c, j);
                        str1 = str1 || (v1d && sl1[j] + 4u > Dv);
                    }
                    if (__ballot(str1) != 0ull) {  // rare: this lane's quad wraps around the end of the ring
#pragma unroll
                        for (int j = 0; j < NQ; ++j) {
                            if (v1d && sl1[j] + 4u > Dv) {
                                v4f t;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    uint32_t se = sl1[j] + (uint32_t)e;
                                    if (se >= Dv) se -= Dv;
                                    t[e] = rbase[se];
                                }
                                asm volatile("" : "+v"(t));
                                rg[j] = t;
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const v4f nv = x[j] + (rg[j] * fb);  // ring[p] = x + (d*fb)
                        float* sp = (v1d && sl1[j] + 4u <= Dv) ? ring + sl1[j] : dummy + 4 * j;
                        asm volatile("" : "+v"(sp));
                        *(v4f_u __attribute__((address_space(1)))*)(uint64_t)sp = nv;
                        const v4f wet = (x[j] * dry) + (rg[j] * mix);  // out = (x*dry) + (d*mix)
                        if (__ballot(str1) != 0ull) {
                            if (v1d && sl1[j] + 4u > Dv) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    uint32_t se = sl1[j] + (uint32_t)e;
                                    if (se >= Dv) se -= Dv;
                                    ring[se] = nv[e];
                                }
                            }
                        }
                        x[j] = o1 ? wet : x[j];
                    }
                    if (v1d) {
                        pos_c += TT;
                        if (pos_c >= Dv) pos_c -= Dv;
                    }
                    if (any_B) ssite(SITE_B, x);
                }
                float* row = &tile[CH_BUF(s)][v][LF * q];
                const bool q15 = q == 15;
                float p1 = row_ror1(q15 ? prev_x[3] : x[NQ - 1][3]), p2 = row_ror1(q15 ? prev_x[2] : x[NQ - 1][2]);
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const v4f xc = x[j];
                    const v4f x1v = (v4f){p1, xc[0], xc[1], xc[2]}, x2v = (v4f){p2, p1, xc[0], xc[1]};
                    const v4f ff = ((xc * b0) + (x1v * b1)) + (x2v * b2);  // ((b0*x) + (b1*x1)) + (b2*x2)
                    const v4f a = has_bq ? ff : xc;                        // no biquad: the samples pass through untouched
                    p1 = xc[3];
                    p2 = xc[2];
                    *(v4f*)(row + 4 * j) = a;
                }
                prev_x = v1 ? x[NQ - 1] : prev_x;
            }
            issue_src(xs, s + 2);
            CH_TRACE(2);
            if constexpr (BQ2) ff2_stage(s - 2, s >= 2 && s - 2 < n_tiles, [&](v4f& xq, int) {
                if (any_C) xq = ssite1(SITE_C, xq);
            });
            // ---- S3a on tile s-LAG3
            {
                uint32_t sl[NQ];
                bool straddle = false;
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    sl[j] = slot_of(pos_c, j);
                    straddle = straddle || (v3 && sl[j] + 4u > Dv);
                }
                if (__ballot(straddle) != 0ull) {  // rare: this lane's quad wraps around the end of the ring
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        if (v3 && sl[j] + 4u > Dv) {
                            v4f t;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                uint32_t se = sl[j] + (uint32_t)e;
                                if (se >= Dv) se -= Dv;
                                t[e] = rbase[se];
                            }
                            asm volatile("" : "+v"(t));  // the wait for these loads stays inside the rare path
                            rg[j] = t;
                        }
                    }
                }
                float* row = &tile[CH_BUF(s - LAG3)][v][LF * q];
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    v4f y = yv[j];
                    if (any_D) y = ssite1(SITE_D, y);  // gain stages between the last biquad and the delay line
                    const v4f nv = y + (rg[j] * fb);  // ring[p] = x + (d*fb)
                    float* sp = (v3 && sl[j] + 4u <= Dv) ? ring + sl[j] : dummy + 4 * j;
                    asm volatile("" : "+v"(sp));
                    *(v4f_u __attribute__((address_space(1)))*)(uint64_t)sp = nv;
                    const v4f wet = (y * dry) + (rg[j] * mix);  // out = (x*dry) + (d*mix)
                    y = ring3 ? wet : y;                        // no delay line behind the filters: untouched
                    y = ssite1(SITE_E, y);
                    if (silent) y = splat(0.f);  // muted gain stage: cleared buffer
                    *(v4f*)(row + 4 * j) = y;
                    if (__ballot(straddle) != 0ull) {
                        if (v3 && sl[j] + 4u > Dv) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                uint32_t se = sl[j] + (uint32_t)e;
                                if (se >= Dv) se -= Dv;
                                ring[se] = nv[e];
                            }
                        }
                    }
                }
                if (v3) {
                    pos_c += TT;
                    if (pos_c >= Dv) pos_c -= Dv;
                }
                if (q == 0) silf[CH_BUF(s - LAG3)][v] = silent ? 1u : 0u;
            }
            issue_ring(rg, o1 ? s + 2 : s - (LAG3 - 2));  // (consumed two steps on: by S3a as tile s + 2 - LAG3, by a delay-first lane's S1 as tile s + 2)
            CH_TRACE(3);
            CH_PROF_END();
            __syncthreads();
            CH_TRACE(4);
        };
        issue_ring(rgA, o1 ? 0 : -1);  // (delay-first lanes: tiles 0 and 1; the others load their dummy line)
        issue_ring(rgB, o1 ? 1 : -1);
        issue_src(xsA, 0);
        issue_src(xsB, 1);
        for (int s = 0; s < n_steps; s += 2) {
            wstep(s, xsA, rgA);
            wstep(s + 1, xsB, rgB);
        }
    } else if (is_worker) {
        for (int s = 0; s < n_steps; ++s) {
            const bool do1 = active && s < n_tiles;
            const bool do3 = active && s >= LAG3 && s - LAG3 < n_tiles;
            const bool dl_on = has_dl && !CH_SKIP(8);
            CH_TRACE(0);
            if (do1 && t1 == 0) {  // new block: adopt the issue-side descriptor (its gain set has landed)
                cur_c16 = nb_c16;
                if (nb_flags & VB_SIMPLE) {
                    inf0.flags = nb_flags;
                    g0 = gs_n[0];
#pragma unroll
                    for (int j = 0; j < FW_CHAIN_STAGES - 1; ++j) inf0.g[j] = gs_n[j + 1];
                } else {
                    const VoiceBlk* d = &fv.blks[(size_t)k1 * fv.n_voices + voice];
                    inf0.flags = d->flags;
#pragma unroll
                    for (int j = 0; j < FW_CHAIN_STAGES - 1; ++j) inf0.g[j] = d->g[j + 1][ch];
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
                const float* yrow = &tile[CH_BUF(s - LAG3)][v][LF * q];
#pragma unroll
                for (int j = 0; j < NQ; ++j) yv[j] = *(const v4f*)(yrow + 4 * j);
            }
            // ================= S1 on tile s: sampler gain (+ the gain stages and the delay line in front of the filters) + the
            // feed-forward half of the biquad -> LDS
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
                    for (int j = 0; j < NQ; ++j) x[j] = (cur_c16 ? chain_cvt16(xs[j], cur_c16, ch) : xs[j]) * g0;
                }
                if (any_A && !(inf0.flags & VB_SRC_ZERO)) {  // the stages in front of the first filter (a cleared source stays cleared)
#pragma unroll
                    for (int j = 0; j < NQ; ++j) x[j] = gsite1(SITE_A, x[j], inf0, k1, t1, j);
                }
                if (dl_first && dl_on) {  // the delay line in front of the biquads: its read-modify-write on the tile S1 has just made
                    if (t1 == 0 && fv.n_cmds) {
                        const ChainDelay p = chain_delay_cmds(fv.cmds, fv.n_cmds, vd.dl_state, cmd_block0 + (uint32_t)k1, ChainDelay{fb, mix, dry});
                        fb = p.fb;
                        mix = p.mix;
                        dry = p.dry;
                    }
                    if (!ring_pref) load_ring();
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const v4f nv = x[j] + (rg[j] * fb);  // ring[p] = x + (d*fb)
                        const uint32_t sl = ring_slot(j);
                        if (sl + 4u <= D) {
                            *(v4f_u*)(ring + sl) = nv;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                uint32_t se = sl + (uint32_t)e;
                                if (se >= D) se -= D;
                                ring[se] = nv[e];
                            }
                        }
                        x[j] = (x[j] * dry) + (rg[j] * mix);  // out = (x*dry) + (d*mix)
                    }
                    pos += TT;
                    if (pos >= D) pos -= D;
                }
                if (any_B) {  // the stages between a delay-first voice's delay line and its first biquad
#pragma unroll
                    for (int j = 0; j < NQ; ++j) x[j] = gsite1(SITE_B, x[j], inf0, k1, t1, j);
                }
                float* row = &tile[CH_BUF(s)][v][LF * q];
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
            if constexpr (BQ2) {  // ================= S1b on tile s-2 (its coefficient messages at block starts, like S1's)
                const bool real = s >= 2 && s - 2 < n_tiles;
                if (real && has_bq2 && tf == 0 && fv.n_cmds) {
                    const ChainCoefs co = chain_find_coefs(fv.cmds, fv.n_cmds, vd.bq2_state, cmd_block0 + (uint32_t)kf);
                    if (co.found) {
                        c0 = co.b0;
                        c1 = co.b1;
                        c2 = co.b2;
                    }
                }
                ff2_stage(s - 2, real && active, [&](v4f& xq, int jq) {
                    // (inf2: the block of tile s - 2.  Real tiles only: behind the call's last tile kf is one past its last block, and a
                    //  stale ramp bit would fetch a ramp row behind the table — a memory fault at one block per call, found by the fuzz)
                    if (any_C && real && active) xq = gsite1(SITE_C, xq, inf2, kf, tf, jq);
                });
                if (real && ++tf == tpb) {
                    tf = 0;
                    ++kf;
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
            const bool dl3 = dl_on && !dl_first;  // S3a's delay line (filter-first voices)
            if (do3) {
                if (dl3) {
                    if (t3 == 0 && fv.n_cmds) {
                        const ChainDelay p = chain_delay_cmds(fv.cmds, fv.n_cmds, vd.dl_state, cmd_block0 + (uint32_t)k3,
                                                              ChainDelay{fb, mix, dry});
                        fb = p.fb;
                        mix = p.mix;
                        dry = p.dry;
                    }
                    if (!ring_pref) load_ring();
                }
                float* row = &tile[CH_BUF(s - LAG3)][v][LF * q];
                const ChainInfo& inf = BQ2 ? inf4 : inf2;  // the block of the tile S3a works on
#ifdef FW_CHAIN_TRACE
#pragma unroll
                for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(yv[j]));
                CH_TRACE(5);
#endif
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    v4f y = yv[j];
                    if (any_D) y = gsite1(SITE_D, y, inf, k3, t3, j);  // gain stages between the last biquad and the delay line
                    if (dl3) {
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
                    if (inf.flags & VB_SILENT) {  // muted gain stage / silent chain: cleared buffer
                        y = splat(0.f);
                    } else {
                        y = gsite1(SITE_E, y, inf, k3, t3, j);
                    }
                    *(v4f*)(row + 4 * j) = y;
                }
                CH_TRACE(6);
                if (dl3) {
                    pos += TT;
                    if (pos >= D) pos -= D;
                }
                if (q == 0) silf[CH_BUF(s - LAG3)][v] = (inf.flags & VB_SILENT) ? 1u : 0u;
                if (++t3 == tpb) {
                    t3 = 0;
                    ++k3;
                }
            }
            // ring slots of the tile the NEXT step consumes (S3a's tile s + 1 - LAG3; S1's tile s + 1 of a delay-first voice): issue
            // now, after this step's ring stores
            if (ring_pref && dl_on && (dl_first ? s + 1 < n_tiles : (s + 1 >= LAG3 && s + 1 - LAG3 < n_tiles))) load_ring();
            inf4 = inf3;
            inf3 = inf2;
            inf2 = inf1;
            inf1 = inf0;
            CH_TRACE(3);
            __syncthreads();
            CH_TRACE(4);
        }
    } else if (is_serial) {
        // the recurrence is the critical path of every step: its wave wins VALU arbitration on its SIMD
        __builtin_amdgcn_s_setprio(3);
        for (int s = 0; s < n_steps; ++s) {
            CH_TRACE(0);
            // ================= S2 on tile s-1: the recursive half of the biquad, lane = voice
            CH_PROF_BEGIN();
            if (any_bq && s >= 1 && s - 1 < n_tiles && !CH_SKIP(1) && !(fv.dbg & 256)) {  // (FWGPU_CHAIN_SKIP=256: timing experiments)
                if (has_bq && t2 == 0 && fv.n_cmds) {
                    const ChainCoefs co = chain_find_coefs(fv.cmds, fv.n_cmds, vd.bq_state, cmd_block0 + (uint32_t)k2);
                    if (co.found) {
                        a1 = co.a1;
                        a2 = co.a2;
                    }
                }
                if (has_bq) {
                    float* row = &tile[CH_BUF(s - 1)][v][0];
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
            CH_PROF_END();
            __syncthreads();
            CH_TRACE(4);
        }
    } else if (is_serial2) {
        // ================= S2b on tile s-3 (two-biquad instantiation): the second biquad's recurrence y[n] = fma(-d1, y[n-1], fma(-d2, y[n-2],
        // ff[n])) on the sums S1b left in the row, lane = voice — S2's loop with the other filter's coefficients and state
        __builtin_amdgcn_s_setprio(3);
        const bool any_bq2 = __ballot(has_bq2) != 0ull;
        int kb = 0, tb = 0;
        for (int s = 0; s < n_steps; ++s) {
            CH_PROF_BEGIN();
            if (any_bq2 && s >= 3 && s - 3 < n_tiles && !(fv.dbg & 128)) {  // (FWGPU_CHAIN_SKIP=128: timing experiments, wrong audio)
                if (has_bq2 && tb == 0 && fv.n_cmds) {
                    const ChainCoefs co = chain_find_coefs(fv.cmds, fv.n_cmds, vd.bq2_state, cmd_block0 + (uint32_t)kb);
                    if (co.found) {
                        d1 = co.a1;
                        d2 = co.a2;
                    }
                }
                if (has_bq2) {
                    float* row = &tile[CH_BUF(s - 3)][v][0];
                    v4f cur[4], nxt[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) cur[u] = *(const v4f*)(row + 4 * u);
#pragma unroll 2
                    for (int c = 0; c < TT / 16; ++c) {
                        if (c + 1 < TT / 16) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) nxt[u] = *(const v4f*)(row + 16 * (c + 1) + 4 * u);
                        }
                        v4f o[4];
                        float t = __builtin_fmaf(-d2, z2, cur[0][0]);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int nu = e == 3 ? u + 1 : u, ne = (e + 1) & 3;
                                const float tn = nu < 4 ? __builtin_fmaf(-d2, z1, cur[nu & 3][ne]) : 0.f;  // t of the next frame
                                const float y = __builtin_fmaf(-d1, z1, t);
                                __builtin_amdgcn_sched_barrier(0);
                                z2 = z1;
                                z1 = y;
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
            }
            if (s >= 3 && s - 3 < n_tiles && ++tb == tpb) {
                tb = 0;
                ++kb;
            }
            CH_PROF_END();
            __syncthreads();
        }
    } else if (is_idle) {
        for (int s = 0; s < n_steps; ++s) __syncthreads();
    } else {
        for (int s = 0; s < n_steps; ++s) {
            CH_TRACE(0);
            // ================= S3b on tile s-3: the leaf SumNode of this channel, lane = frame quad, ports in order
            CH_PROF_BEGIN();
            if (s >= LAG4 && s - LAG4 < n_tiles && lane < TT / 4 && !CH_SKIP(2)) {  // (the steady-call loop pads n_steps to even)
                const int buf = (int)CH_BUF(s - LAG4);
                // ONE LDS round trip: every port's row (row index clamped, so the reads are unconditional) and the
                // silence flags are requested together; the adds are masked
                const float* col = &tile[buf][0][4 * lane];
                v4f x[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) x[u] = *(const v4f*)(col + (size_t)(u < ports ? u : ports - 1) * PITCH);
                const uint32_t silent_rows = (uint32_t)(__ballot(lane < ports && silf[buf][lane & 31] != 0) & port_mask);
                float* bus_blk = fv.bus + (size_t)k4 * fv.bus_blk_stride + t4 * TT + 4 * lane;
                uint8_t* flag_blk = fv.bus_flags + (size_t)k4 * fv.bus_flags_blk_stride;
                // the usual workgroups: 32 rows = full leaves of one size with nothing to skip (a silent port matters only
                // on the n-port path, sum.rs:122-124) — straight-line adds; everything else: the row walk below
                const int up = (silent_rows & grp.masked_rows) || (fv.dbg & 64) ? 0 : grp.uniform_ports;  // (64: A/B)
                const bool wf = t4 == 0 && lane == 0;
                if (up == 32) {
                    chain_mix_uniform<32>(x, silent_rows, bus_blk, flag_blk, g_out, ch, fv.stride, wf);
                } else if (up == 16) {
                    chain_mix_uniform<16>(x, silent_rows, bus_blk, flag_blk, g_out, ch, fv.stride, wf);
                } else if (up == 8) {
                    chain_mix_uniform<8>(x, silent_rows, bus_blk, flag_blk, g_out, ch, fv.stride, wf);
                } else if (up == 4) {
                    chain_mix_uniform<4>(x, silent_rows, bus_blk, flag_blk, g_out, ch, fv.stride, wf);
                } else {
                    // several leaves share the rows: walk the rows once in order (static register indices; the leaf
                    // boundaries and the rows to add are bit masks in SGPRs), restart the sum at each leaf's port 0, store
                    // when a leaf is complete.  Selects instead of branches for the arithmetic (branches made the
                    // compiler shuffle the accumulator through copies: 7 100 cycles per tile); the only branch per row
                    // is the rarely taken "a leaf ends here".
                    const uint32_t row_mask = ports >= 32 ? 0xffffffffu : ((1u << ports) - 1u);
                    const uint32_t starts = grp.start_mask & row_mask;
                    const uint32_t adds = row_mask & ~starts & ~(silent_rows & grp.masked_rows);  // sum.rs:122-124: skipped
                    v4f acc = x[0];  // sum.rs:117 copy port 0 (also when silent: a cleared buffer)
                    int li = 0;
#define CH_FLUSH_LEAF()                                                                                       \
    do {                                                                                                      \
        const uint32_t rows__ = g_rows[li];                                                                   \
        const int ob__ = g_out[li];                                                                           \
        const bool as__ = (silent_rows & rows__) == rows__;                                                   \
        *(v4f*)(bus_blk + (size_t)(ob__ + ch) * fv.stride) = as__ ? splat(0.f) : acc; /* sum.rs:52-56 */       \
        if (t4 == 0 && lane == 0) flag_blk[ob__ + ch] = as__ ? 1 : 0;                                          \
        ++li;                                                                                                 \
    } while (0)
#pragma unroll
                    for (int u = 1; u < 32; ++u) {
                        const bool st = (starts >> u) & 1u, ad = (adds >> u) & 1u;  // wave-uniform
                        if (__builtin_amdgcn_readfirstlane((int)st)) CH_FLUSH_LEAF();
                        const v4f t = acc + x[u];
                        const v4f y = st ? x[u] : t;
                        acc = (st || ad) ? y : acc;
                    }
                    CH_FLUSH_LEAF();
#undef CH_FLUSH_LEAF
                }
                if (++t4 == tpb) {
                    t4 = 0;
                    ++k4;
                }
            }
            CH_TRACE(3);
            CH_PROF_END();
            __syncthreads();
            CH_TRACE(4);
        }
    }

#ifdef CH_PROF
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 2 || wave == 6 || wave == 11 || wave == 3)) {
        const unsigned long long tot = clock64() - prof_t0, wall = wall_clock64() - prof_w0;
        printf("k_chain prof: wave %d steps %d fast %d: busy %llu of %llu shader cycles (%.0f per step, total %.0f per step), wall %.2f us -> %.0f MHz\n", wave, n_steps,
               (int)wg_fast, prof_busy, tot, (double)prof_busy / n_steps, (double)tot / n_steps, wall / 100.0, (double)tot / (wall / 100.0));
    }
#endif
    // ---- write this channel's biquad state back (everything shared was advanced by k_voice_control)
    if (is_worker && has_bq && q == 15) {
        bq_st[0] = prev_x[3];
        bq_st[1] = prev_x[2];
    }
    if (is_serial && has_bq) {
        bq_st[2] = y1;
        bq_st[3] = y2;
    }
    if (is_serial2 && has_bq2) {  // (every S1b ran at least one barrier ago: bq2_u holds the inputs of the call's last two frames)
        bq2_st[0] = bq2_u[lane][0];
        bq2_st[1] = bq2_u[lane][1];
        bq2_st[2] = z1;
        bq2_st[3] = z2;
    }
#undef CH_BUF
}

