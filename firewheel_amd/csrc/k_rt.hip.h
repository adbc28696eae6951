// k_rt.hip.h — part of the single device translation unit fwgpu_kernels.hip (included inside namespace fwgpu).
// Realtime edge: ONE launch per callback for the voice-bank plan.
//
// A backend callback is one block (cpal/lib.rs:378-449 -> processor.rs:61-165 with frames <= max_block_frames); its
// compulsory HBM traffic is microseconds below one kernel boundary, so what a callback costs is the NUMBER of dependent
// launches: control kernel -> leaf sums -> root sum + interleave were 3 boundaries (~5 us each on this stack) + 3 more
// host-side launch calls.  Here one workgroup per leaf SumNode does all three in sequence:
//   1. its voices' control: one LANE per steady voice (voice_control_lane_steady), one WAVE per voice that needs its state
//      machines this block (voice_control_wave: messages, smoothers, playheads -> this block's records),
//   2. its leaf sum (leaf_sum_wave: the waves take 256-frame pieces of the block),
//   3. the LAST workgroup to finish — an agent-scope counter — adds the leaf buses in the root SumNode's port order and
//      interleaves into the (pinned, device-mapped) output block: root_out_any, the code of k_root_out — and raises the
//      completion flag in pinned host memory the audio thread polls.
// Same device functions as the throughput kernels, so the arithmetic is theirs bit for bit.  Used when the call is one
// block, the plan is the plain voice-bank plan (a mixer tree of any depth: round 5) and the stream is stereo; everything else takes the
// launch sequence.
// -DFW_RT_TRACE: the last workgroup of every 512th callback prints where its time went (10 ns ticks)
#ifdef FW_RT_TRACE
#define RT_T(i) rt_tr[i] = __builtin_amdgcn_s_memrealtime()
#else
#define RT_T(i)
#endif
// The leaf sum of a one-block callback, every lane ONE frame: leaf_sum_wave gives a wave 256 frames — four per lane, 16 ports' loads in
// flight, so a 32-port leaf is two cold HBM round trips on one wave while the workgroup's other three waves idle.  Here the four waves
// take 64 frames each and a lane requests its frame of ALL ports (64 loads) before it waits: one round trip.  For leaves whose every
// port is silent or a plain planar-f32 voice with gain stages only — the steady case; anything else returns false and takes
// leaf_sum_wave.  Same operations per sample in the same order (leaf_fast / the port-by-port loop of leaf_sum_wave: sum.rs:41-136).
typedef const float __attribute__((address_space(1)))* rt_gfp;
// (branch-free on purpose: 32 small uniform branches around 64 live values made the register allocator shuffle everything through
//  AGPRs and scratch — 2.4 KB per lane.  Absent and silent ports read the constant-zero bus and are kept out of the sum by selects)
template <int NG>
__device__ __forceinline__ void rt_leaf_quick_body(const float* my_l, const float* my_r, const GainSet& my_g, const int f, const uint64_t skip_mask, float& accl,
                                                   float& accr) {
    float xl[32], xr[32];
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        xl[p] = ((rt_gfp)readlane_ptr(my_l, p))[f];
        xr[p] = ((rt_gfp)readlane_ptr(my_r, p))[f];
    }
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        float a = xl[p], b = xr[p];
#pragma unroll
        for (int j = 0; j < NG; ++j) {  // sampler.rs:530-533, volume.rs:123-126, pan: one rounding each
            a = a * readlane_f(my_g.g[j][0], p);
            b = b * readlane_f(my_g.g[j][1], p);
        }
        if (p == 0) {  // sum.rs:117 copy_from_slice(port 0) — also when silent
            accl = a;
            accr = b;
        } else {  // :122-124: silent ports are skipped on the n-port path only; ports the leaf does not have, always
            const bool skip = (skip_mask >> p) & 1ull;
            const float sl = accl + a, sr = accr + b;
            accl = skip ? accl : sl;
            accr = skip ? accr : sr;
        }
    }
}
template <bool PROG>
__device__ __forceinline__ bool rt_leaf_quick(const FusedView& fv, const int leaf, const int wave) {
    const int lane = threadIdx.x & (WAVE - 1);
    const LeafDesc ld = fv.leaves[leaf];
    const int frames = fv.frames;
    if (frames > 256 || ld.ports < 1 || ld.ports > 32) return false;
    VoiceRef ref;
    ref.src_l = nullptr;
    ref.r_delta = 0;
    ref.flags_gset = VB_SILENT;
    GainSet my_g;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES; ++j) my_g.g[j][0] = my_g.g[j][1] = 1.0f;
    GainSet g0 = my_g;
    uint32_t my_prog = 0u;
    if (lane < ld.ports) {
        g0 = fv.gsets[(size_t)(ld.first_voice + lane) * FW_GSETS];  // (slot 0, requested beside the record; replaced below if the record names another)
        ref = fv.refs[ref_index(ld.first_voice + lane, 0, fv.ref_kgroups)];
        if constexpr (PROG) my_prog = fv.progs[ld.first_voice + lane];
    }
    const uint32_t fl = ref.flags_gset & 0xffu;
    const bool sil = (fl & VB_SILENT) != 0;
    const bool plain = !sil && (fl & VB_SIMPLE) != 0 && ((ref.flags_gset >> 16) & 7u) == SF_P_F32 && my_prog == 0u;
    if (__ballot(lane < ld.ports && !sil && !plain)) return false;
    if (plain) {  // (a silent port keeps gains of 1: its zeros stay +0)
        my_g = g0;
        const uint32_t gi = (ref.flags_gset >> 8) & 0xffu;
        if (gi != 0u) my_g = fv.gsets[(size_t)(ld.first_voice + lane) * FW_GSETS + gi];
    }
    const uint64_t lanes_in = mask_all_silent_bits(ld.ports);
    const uint64_t silent_ports = __ballot(sil) & lanes_in;
    const bool all_silent = silent_ports == lanes_in;
    const int path_ports = ld.pad ? ld.pad : ld.ports;
    const bool masked = !(path_ports == 2 || path_ports == 3 || path_ports == 4);  // sum.rs:67-133 (Q13)
    const uint64_t skip_mask = ~lanes_in | (masked ? silent_ports : 0ull);
    // silent and absent ports read bus 0 of block 0: the constant-zero bus
    const float* my_l = plain ? ref.src_l : fv.bus;
    const float* my_r = plain ? ref.src_l + ref.r_delta : fv.bus;
    {  // (read with v_readlane where only the lanes that own frames are active: pinned under the full exec mask, as in leaf_sum_wave)
        uint64_t pl = (uint64_t)my_l, pr = (uint64_t)my_r;
        asm volatile("" : "+v"(pl), "+v"(pr));
        my_l = (const float*)pl;
        my_r = (const float*)pr;
#pragma unroll
        for (int j = 0; j < FW_MAX_STAGES; ++j) asm volatile("" : "+v"(my_g.g[j][0]), "+v"(my_g.g[j][1]));
    }
    float* outl = fv.bus + (size_t)ld.out_buf * fv.stride;  // block 0
    float* outr = outl + fv.stride;
    const int f = wave * WAVE + lane;
    if (f < frames) {
        float accl = 0.f, accr = 0.f;
        switch (fv.n_gain_stages) {
            case 1: rt_leaf_quick_body<1>(my_l, my_r, my_g, f, skip_mask, accl, accr); break;
            case 2: rt_leaf_quick_body<2>(my_l, my_r, my_g, f, skip_mask, accl, accr); break;
            case 3: rt_leaf_quick_body<3>(my_l, my_r, my_g, f, skip_mask, accl, accr); break;
            case 4: rt_leaf_quick_body<4>(my_l, my_r, my_g, f, skip_mask, accl, accr); break;
            case 5: rt_leaf_quick_body<5>(my_l, my_r, my_g, f, skip_mask, accl, accr); break;
            default: rt_leaf_quick_body<6>(my_l, my_r, my_g, f, skip_mask, accl, accr); break;
        }
        if (all_silent) accl = accr = 0.f;  // clear_all_outputs (sum.rs:52-56)
        outl[f] = accl;
        outr[f] = accr;
    }
    if (lane < 2 && wave == 0) fv.bus_flags[ld.out_buf + lane] = all_silent ? 1 : 0;
    return true;
}

template <bool PROG, bool RS>
__device__ __forceinline__ void rt_block_body(const FusedView& fv, const DevView& upv, const RootArgs& ra, float* __restrict__ out, const uint32_t cmd_block0,
                                              unsigned* __restrict__ sync, unsigned long long* done_flag, const unsigned long long done_seq, const RsLds rs) {
#ifdef FW_RT_TRACE
    unsigned long long rt_tr[10];
    RT_T(0);
#endif
    const int leaf = blockIdx.x;
    const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & (WAVE - 1);
    const LeafDesc ld = fv.leaves[leaf];
    // 1. control.  Thread p takes port p's voice on the lane-per-voice steady path (a leaf has <= 32 ports: all of them sit
    // in wave 0); the voices that need their state machines this block — a message, a gliding gain, a one-shot ending —
    // are then run wave-wide, dealt out over the four waves
    RT_T(1);
    __shared__ unsigned long long s_need;
    bool need = false;
    if ((int)threadIdx.x < ld.ports) need = !voice_control_lane_steady(fv, ld.first_voice + (int)threadIdx.x, cmd_block0);
    if (wave == 0) {
        const unsigned long long m = __ballot(need);
        if (lane == 0) s_need = m;
    }
    __syncthreads();
    {
        unsigned long long m = s_need;
        int turn = 0;
        while (m) {
            const int p = __builtin_ctzll(m);
            m &= m - 1;
            if ((turn++ & 3) == wave) voice_control_wave(fv, ld.first_voice + p, lane, 1, cmd_block0);
        }
    }
    RT_T(2);
    // the records (refs / gain sets / descriptors / ramps) were written by all four waves and are read by all four:
    // same CU, same L1 — a workgroup-scope release / acquire around the barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    RT_T(3);
    if (!rt_leaf_quick<PROG>(fv, leaf, wave))
        leaf_sum_wave<PROG, RS, 16>(fv, leaf, 0u, wave, 4, rs);  // 16 ports in flight: two round trips per leaf, not eight
    RT_T(4);
    // grid-wide hand-over UP THE MIXER TREE (round 5: any depth; rounds 2-4: leaves + root only).  Every workgroup publishes what it
    // wrote (agent scope: the XCDs have separate L2s) and counts itself in at its bus's consumer (fv.rt_parent_*: the upper-tree node
    // that reads it); whoever completes a node's children renders that node — bus_sum_node_wg, the code of k_bus_sum — and carries on
    // at ITS consumer; whoever completes the root's children does the root + interleave below.  Nobody waits for anybody: a workgroup
    // that is not the last one at a node is done.  (The counters reset themselves: the completing arrival stores 0.)
    __shared__ int s_last;
    int node = fv.rt_parent_leaf[leaf];
    for (;;) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            // (relaxed: the release is the fence above, the acquire the fence the completing workgroup takes below — an acq_rel
            //  read-modify-write here was a second L2 write-back and an invalidate in EVERY workgroup, on the callback's critical path)
            unsigned* ctr = fv.rt_tree_sync + node;
            const unsigned prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = prev == (unsigned)fv.rt_kids[node] - 1u ? 1 : 0;
            if (s_last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next callback
        }
        __syncthreads();
        RT_T(5);
        if (!s_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (node == fv.rt_root) break;
        const NodeDesc nd = upv.nodes[node];
        bus_sum_node_wg(upv, nd, 0u, 0);
        bus_sum_node_wg(upv, nd, 0u, 1);
        node = fv.rt_parent_up[node];
    }
    RT_T(6);
    for (int f0 = 0; f0 < upv.frames; f0 += 256) root_out_any(upv, ra, out, 0u, f0 + (int)threadIdx.x);
    RT_T(7);
    // 4. completion: the output block sits in pinned host memory; publish it with a system-scope release and raise the
    // flag the audio thread is spinning on (a blocking stream sync costs a driver wake-up, ~15 us on this stack)
    if (done_flag) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef FW_RT_TRACE
    RT_T(8);
    if (threadIdx.x == 0 && (done_seq & 511) == 0)
        printf("rt wg %d: leafdesc %llu control %llu fence %llu leafsum %llu publish %llu acquire %llu root %llu flag %llu total %llu (x10 ns)\n",
               (int)blockIdx.x, rt_tr[1] - rt_tr[0], rt_tr[2] - rt_tr[1], rt_tr[3] - rt_tr[2], rt_tr[4] - rt_tr[3], rt_tr[5] - rt_tr[4],
               rt_tr[6] - rt_tr[5], rt_tr[7] - rt_tr[6], rt_tr[8] - rt_tr[7], rt_tr[8] - rt_tr[0]);
#endif
}

template <bool PROG, bool RS>
__global__ __launch_bounds__(256) void k_rt_block(FusedView fv, DevView upv, RootArgs ra, float* __restrict__ out, uint32_t cmd_block0,
                                                  unsigned* __restrict__ sync, unsigned long long* done_flag, unsigned long long done_seq) {
    extern __shared__ float s_rt_dyn[];
    RsLds rs{nullptr, nullptr};
    if constexpr (RS) rs = rs_lds_setup(fv, s_rt_dyn);
    rt_block_body<PROG, RS>(fv, upv, ra, out, cmd_block0, sync, done_flag, done_seq, rs);
}

// Between two callbacks the resident kernel has milliseconds to spare: every workgroup asks for the source frames its steady
// voices will read in the NEXT block (playheads have been advanced already), so that the two cold HBM round trips of the leaf sum
// (3 us each: 32 voices' 1 KiB rows, 64 KiB apart) find them in the Infinity Cache / L2.  Loads whose values are dropped; only
// voices whose next block lies inside their sample are touched (a wrap or a one-shot end takes the cold path once).
__device__ __forceinline__ void rt_prefetch_next(const FusedView& fv, float* sink) {
    const LeafDesc ld = fv.leaves[blockIdx.x];
    const int frames = fv.frames;
    __shared__ const float* s_pf[64];  // [port][channel], nullptr = nothing to fetch
    if ((int)threadIdx.x < 64) s_pf[threadIdx.x] = nullptr;
    __syncthreads();
    if ((int)threadIdx.x < ld.ports && ld.ports <= 32) {
        const int vi = ld.first_voice + (int)threadIdx.x;
        const VoiceDesc vd = fv.voices[vi];
        if (vd.sampler_state >= 0 && vd.src_kind == 0) {
            const VoiceCache vc = fv.cache[vi];
            if (vc.epoch == fv.epoch && (vc.mode == 1 || vc.mode == 2) && vc.sample >= 0) {
                const SampleDesc sd = fv.samples[vc.sample];
                const uint64_t ph = fv.states[vd.sampler_state].playhead;
                if (sd.format == FMT_P_F32 && sd.data && ph + (uint64_t)frames <= sd.frames) {
                    s_pf[2 * threadIdx.x] = (const float*)sd.data + ph;
                    if (sd.channels >= 2) s_pf[2 * threadIdx.x + 1] = (const float*)sd.data + sd.frames + ph;
                }
            }
        }
    }
    __syncthreads();
    float acc = 0.f;
    const int lines = (frames * 4 + 127) / 128;
    for (int i = threadIdx.x; i < 64 * lines; i += blockDim.x) {
        const float* p = s_pf[i / lines];
        if (p) acc += p[(i % lines) * 32 < frames ? (i % lines) * 32 : frames - 1];
    }
    if (acc == 1.2345678e-30f) *sink = acc;  // (keeps the loads; never true in practice, and harmless if it is)
}

// The same, resident: launched by the first steady callback of a run of them (no message pending, same plan, same output block)
// and fed through a doorbell in pinned host memory from then on — a callback then costs neither a launch call on the audio thread
// (6.5 us on this stack) nor the dispatch of a grid.  cpal/lib.rs:378-449 is the pattern: a backend thread that is woken per
// block and never torn down between blocks.
//   * workgroup 0 polls RtMailbox::doorbell (host memory over PCIe, one read per ~us) for the next sequence number and hands it to
//     the others through a word of device memory (`go`), so that every workgroup takes the same decision;
//   * the host asks for the end with `seq | RT_QUIT_BIT` (any call that is not a steady callback, a plan adoption, destroy);
//   * HOLD: a control call that may free device memory or synchronise with the device (fwgpu_update's table growth, sample_create,
//     exchange_open ...) raises RtMailbox::hold first and waits for alive == 0 (RtHold, fwgpu_ctx.h) — hipFree waits for EVERY stream,
//     and this kernel ends only when told to;
//   * WATCHDOG: no doorbell for `idle_ticks` (100 MHz ticks; default 20 ms) and workgroup 0 decides to quit on its own — a kernel that
//     never ends would hold its stream, and a shared pool, hostage.  The host finds RtMailbox::alive == 0 and launches a new one
//     with the next callback.  The other workgroups give up after 8 x that time without a word from workgroup 0.
template <bool PROG, bool RS>
__global__ __launch_bounds__(256) void k_rt_persist(FusedView fv, DevView upv, RootArgs ra, float* __restrict__ out, uint32_t cmd_block0,
                                                    unsigned* __restrict__ sync, unsigned long long* done_flag, RtMailbox* mb,
                                                    unsigned long long* go, unsigned long long first_seq, unsigned long long idle_ticks, int prefetch) {
    extern __shared__ float s_rt_dyn[];
    RsLds rs{nullptr, nullptr};
    if constexpr (RS) rs = rs_lds_setup(fv, s_rt_dyn);
    __shared__ unsigned long long s_cmd;
    unsigned long long seq = first_seq;
    unsigned long long t0 = 0;  // (thread 0) when the wait for `seq` began
    bool waiting = false, pf_done = false;
    for (;;) {
        if (threadIdx.x == 0) {
            if (!waiting) {
                t0 = __builtin_amdgcn_s_memrealtime();
                waiting = true;
                pf_done = !prefetch;
            }
            unsigned long long cmd = seq | RT_QUIT_BIT;
            if (blockIdx.x == 0) {
                for (;;) {
                    // (the control side's hold word travels with the doorbell: same line, requested first, no wait of its own)
                    const unsigned long long hold = __hip_atomic_load(&mb->hold, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const unsigned long long d = __hip_atomic_load(&mb->doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (d == seq || d == (seq | RT_QUIT_BIT)) {
                        cmd = d;
                        break;
                    }
                    if (hold) break;  // a control call is about to free / synchronise: end like the watchdog (cmd = quit); a doorbell
                                      // that arrives now finds alive == 0 and the block goes out as an ordinary launch
                    const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
                    if (dt > idle_ticks) break;  // watchdog: cmd = quit
                    if (!pf_done && dt > 1000) {  // 10 us without a doorbell: this is a paced stream, not a back-to-back one — use the
                        cmd = seq | RT_PREFETCH_BIT;  // gap to bring the next block's sources in (rt_prefetch_next)
                        pf_done = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                __hip_atomic_store(go, cmd, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (;;) {
                    const unsigned long long d = __hip_atomic_load(go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (d == seq || d == (seq | RT_QUIT_BIT)) {
                        cmd = d;
                        break;
                    }
                    if (!pf_done && d == (seq | RT_PREFETCH_BIT)) {
                        cmd = d;
                        pf_done = true;
                        break;
                    }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 8 * idle_ticks) break;
                    // (hundreds of workgroups polling ONE word every ~100 ns starve the workgroups that still work — the mixers on the
                    //  way up the tree: config 5's 256 leaves ran a callback in 113 us resident against 51 us launched, round 5)
                    if (gridDim.x > 64) __builtin_amdgcn_s_sleep(32);
                    else __builtin_amdgcn_s_sleep(2);
                }
            }
            s_cmd = cmd;
        }
        __syncthreads();
        const unsigned long long cmd = s_cmd;
        __syncthreads();  // (s_cmd is rewritten by the next round)
        if (cmd == (seq | RT_PREFETCH_BIT)) {
            rt_prefetch_next(fv, (float*)(go + 2));
            continue;
        }
        if (cmd != seq) break;
        // (no acquire here: between two doorbells nothing outside this kernel writes what it reads — anything that does ends it
        //  first — and an agent-scope acquire would empty the L2 of what rt_prefetch_next has just brought in)
        rt_block_body<PROG, RS>(fv, upv, ra, out, cmd_block0, sync, done_flag, seq, rs);
        ++seq;
        waiting = false;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&mb->alive, 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the same completion flag behind any launch sequence (realtime-sized calls that do not fit k_rt_block)
__global__ void k_signal_done(unsigned long long* done_flag, unsigned long long done_seq) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
