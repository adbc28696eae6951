// k_exchange.hip.h — part of the single device translation unit fwgpu_kernels.hip (included inside namespace fwgpu).
// Multi-GPU mix bus (path 2): the top-level R-port SumNode of a voice-sharded graph (nodes/sum.rs:41-136) as a
// ONE-SHOT exchange over peer-mapped slots instead of a ring collective.
//
//   k_bus_push    rank g writes its partial bus (K blocks, interleaved) and its per-(block, channel) silence flags into
//                 slot g of EVERY rank's exchange region — its own through a plain pointer, the peers' through pointers the
//                 host mapped from their IPC handles (xGMI point-to-point stores: every GPU of a node has a direct link to
//                 every other) — then publishes the step's sequence number into word g of every rank's arrival table with
//                 a system-scope release.  The last workgroup to finish does the publishing (agent-scope counter).
//   k_bus_wait    ONE wave waits until all R arrival words of ITS OWN region carry the step's sequence number (one lane per
//                 peer, system-scope acquire loads, s_sleep between polls, bounded by a wall-clock budget: a peer that never
//                 arrives turns into an error word + a zero-filled bus, never a hung GPU — and the error is STICKY on the device: the
//                 rank that timed out pushes and waits no more, so its peers time out at their next step instead of summing a
//                 slot it might be overwriting: after a timeout every rank ends on zero buses + its own error word);
//   k_bus_reduce  the launch behind it adds the R slots in rank order with the reference's SumNode semantics — all silent -> cleared; 1 port -> copy; 2/3/4 ports -> unmasked adds;
//                 otherwise out = in0, then += in_p skipping SILENT ports (sum.rs:111-133, Q13).
// Every rank ends with the bits of the single-process graph whose top node is that SumNode: the order is the port order,
// nothing is re-associated (an all-reduce ring re-associates for R > 2).  The regions are fine-grained device
// memory (hipDeviceMallocFinegrained): coherent between agents under the system-scope release / acquire used here.
//
// Two data parities (seq & 1): rank g may run push(s+1) while a peer still reads step s.  push(s+2) overwrites parity s —
// by then every peer has finished reduce(s): push and reduce of one rank are stream-ordered, my push(s+2) follows my
// reduce(s+1), which saw every peer's push(s+1), which that peer issued behind its reduce(s).
#pragma once

// layout of one rank's exchange region (bytes); the same on every rank of an exchange
#define EX_FLAGS_OFF 0      // unsigned long long arrival[FW_MAX_BUS_PARTS]
#define EX_ERR_OFF 1024     // unsigned long long: 0, or the sequence number of the first step whose wait ran out of time
#define EX_DATA_OFF 4096    // [2 parities][world][slot_bytes]; slot = [max_floats f32][max_sil u8, padded to 16]

__device__ __forceinline__ char* ex_slot(char* base, const ExchangeGeom& g, unsigned long long seq, int src_rank) {
    return base + EX_DATA_OFF + ((size_t)(seq & 1ull) * g.world + (size_t)src_rank) * g.slot_bytes;
}

__global__ __launch_bounds__(256) void k_bus_push(ExchangePeers peers, ExchangeGeom g, const float* __restrict__ part,
                                                  const uint8_t* __restrict__ sil, size_t n_floats, uint32_t n_sil,
                                                  unsigned long long seq, unsigned* __restrict__ counter) {
    // A rank whose wait once ran out of time is OUT of the exchange for good: it has skipped a step its peers may still be
    // reading, so the two-parity argument above no longer covers what it would overwrite next.  It pushes nothing more — the peers'
    // next wait runs out too, and every rank ends with zero buses and its own sticky error word, none with a silently wrong sum.
    // (The error word is written by k_bus_wait, stream-ordered before this launch: every workgroup reads the same value.)
    if (__hip_atomic_load((const unsigned long long*)(peers.base[g.rank] + EX_ERR_OFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull) return;
    const size_t n4 = n_floats / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        const v4f x = *(const v4f*)(part + 4 * i);
        for (int p = 0; p < g.world; ++p) *(v4f*)((float*)ex_slot(peers.base[p], g, seq, g.rank) + 4 * i) = x;
    }
    if (blockIdx.x == 0) {
        for (size_t j = n4 * 4 + threadIdx.x; j < n_floats; j += blockDim.x) {  // ragged tail
            const float x = part[j];
            for (int p = 0; p < g.world; ++p) ((float*)ex_slot(peers.base[p], g, seq, g.rank))[j] = x;
        }
        for (uint32_t j = threadIdx.x; j < n_sil; j += blockDim.x) {
            const uint8_t s = sil ? sil[j] : (uint8_t)0;
            for (int p = 0; p < g.world; ++p) ((uint8_t*)ex_slot(peers.base[p], g, seq, g.rank) + (size_t)g.max_floats * 4)[j] = s;
        }
    }
    // publish: every workgroup's stores are released at system scope before it counts itself in; the last one to arrive
    // raises this rank's arrival word on every peer
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == gridDim.x - 1 ? 1 : 0;
        if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
    if ((int)threadIdx.x < g.world)
        __hip_atomic_store((unsigned long long*)(peers.base[threadIdx.x] + EX_FLAGS_OFF) + g.rank, seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// silence of (part p, block b, channel c); `sil[p]` = that part's flags [n_blocks][n_ch], or null = nothing is silent
struct SilView {
    const uint8_t* sil[FW_MAX_BUS_PARTS];
};

// The R-port SumNode on 4 consecutive floats of the interleaved buses (sum.rs:41-136 + the graph_out edge, which adds
// nothing: interleave_stereo zero-fills iff both channels are silent, util.rs:129-134, and then the sum has cleared them)
template <class GetPart, class GetSil>
__device__ __forceinline__ v4f ordered_sum_quad(int world, size_t i4, uint32_t blk_floats, uint32_t n_ch, uint32_t n_blocks, GetPart part,
                                                GetSil silent, uint8_t out_sil[4]) {
    v4f acc = *(const v4f*)(part(0) + 4 * i4);
    if (blk_floats == 0) {  // no flags anywhere: nothing is silent
        for (int p0 = 1; p0 < world; p0 += 8) {
            v4f x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + u < world) x[u] = *(const v4f*)(part(p0 + u) + 4 * i4);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + u < world) acc = acc + x[u];
        }
        out_sil[0] = out_sil[1] = out_sil[2] = out_sil[3] = 0;
        return acc;
    }
    const bool masked = !(world == 2 || world == 3 || world == 4);  // sum.rs:67-110 vs :111-133
    uint32_t blk[4], ch[4];
    bool all_sil[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t f = 4 * i4 + e;
        blk[e] = (uint32_t)(f / blk_floats);
        ch[e] = (uint32_t)(f % n_ch);
        if (blk[e] >= n_blocks) blk[e] = n_blocks - 1;
        if (e > 0 && blk[e] == blk[e - 1]) {  // (a quad rarely straddles a block boundary: one scan of the flags serves all four)
            all_sil[e] = all_sil[e - 1];
            continue;
        }
        bool all = true;
        for (int p = 0; p < world; ++p)
            for (uint32_t c = 0; c < n_ch; ++c) all = all && silent(p, blk[e], c);
        all_sil[e] = all;  // sum.rs:52-56: every input channel silent -> clear_all_outputs
    }
    for (int p = 1; p < world; ++p) {
        const v4f x = *(const v4f*)(part(p) + 4 * i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool use = !(masked && silent(p, blk[e], ch[e]));  // sum.rs:122-124
            const float s = acc[e] + x[e];
            acc[e] = use ? s : acc[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (all_sil[e]) acc[e] = 0.f;
        out_sil[e] = all_sil[e] ? 1 : (world == 1 ? (silent(0, blk[e], ch[e]) ? 1 : 0) : 0);  // sum.rs:58-65 mask passthrough
    }
    return acc;
}

// Wait for the R arrivals of step `seq`: ONE wave, lane p polls peer p's arrival word (system-scope acquire loads of this
// rank's own fine-grained region, s_sleep between polls).  One wave, not the reduce grid: a grid of spinning workgroups holds
// wave slots the peers' render kernels may need — with several ranks time-sharing one device (tests, bench.py --share-device)
// the spinners of N-1 ranks starve the rank everybody is waiting for.  Bounded by a wall-clock budget (s_memrealtime, 100 MHz):
// a peer that never arrives turns into an error word + a zero bus, never a hung GPU.  `sync` (device-local):
// [1] = the step number when every peer arrived, else 0; [8 + p] = the longest this rank has waited for peer p (ticks).
__global__ __launch_bounds__(64) void k_bus_wait(char* __restrict__ base, int world, unsigned long long seq, unsigned long long budget_ticks,
                                                 unsigned long long* __restrict__ sync) {
    const int lane = threadIdx.x;
    // (sticky: after one timeout this rank neither pushes nor waits again — see k_bus_push)
    bool ok = __hip_atomic_load((const unsigned long long*)(base + EX_ERR_OFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0ull;
    if (ok && lane < world) {
        const unsigned long long* w = (const unsigned long long*)(base + EX_FLAGS_OFF) + lane;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long waited = 0;
        while (__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            waited = __builtin_amdgcn_s_memrealtime() - t0;
            if (waited > budget_ticks) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        if (waited > sync[8 + lane]) sync[8 + lane] = waited;
    }
    const bool all_ok = __ballot(ok) == __ballot(true);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (lane == 0) {
        sync[1] = all_ok ? seq : 0ull;
        if (!all_ok) {
            unsigned long long* err = (unsigned long long*)(base + EX_ERR_OFF);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0ull)
                __hip_atomic_store(err, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256) void k_bus_reduce(char* __restrict__ base, ExchangeGeom g, float* __restrict__ out,
                                                    uint8_t* __restrict__ out_sil, size_t n_floats, uint32_t n_sil, uint32_t frames,
                                                    uint32_t n_ch, unsigned long long seq, const unsigned long long* __restrict__ sync) {
    const bool ok = sync[1] == seq;  // written by k_bus_wait, the launch before this one on the same stream
    const size_t n4 = n_floats / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!ok) {  // a peer never arrived: leave a fully written (zero) bus behind (core/node.rs:41-42), every block flagged silent
        if (i < n4) *(v4f*)(out + 4 * i) = splat(0.f);
        if (i == 0)
            for (size_t j = n4 * 4; j < n_floats; ++j) out[j] = 0.f;
        if (out_sil)
            for (size_t j = i; j < n_sil; j += (size_t)gridDim.x * blockDim.x) out_sil[j] = 1;
        return;
    }
    const uint32_t n_blocks = n_ch ? n_sil / n_ch : 0u;
    const uint32_t blk_floats = n_blocks ? frames * n_ch : 0u;
    auto part = [&](int p) -> const float* { return (const float*)ex_slot(base, g, seq, p); };
    auto silent = [&](int p, uint32_t b, uint32_t c) -> bool {
        return ((const uint8_t*)ex_slot(base, g, seq, p) + (size_t)g.max_floats * 4)[(size_t)b * n_ch + c] != 0;
    };
    if (i < n4) {
        uint8_t os[4];
        const v4f y = ordered_sum_quad(g.world, i, blk_floats, n_ch, n_blocks, part, silent, os);
        *(v4f*)(out + 4 * i) = y;
        if (out_sil && blk_floats)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t f = 4 * i + e;
                if (f % blk_floats < n_ch) out_sil[(f / blk_floats) * n_ch + f % n_ch] = os[e];  // the block's first frame writes its flags
            }
    }
    if (i == 0)  // ragged tail (n % 4 floats): scalar, same semantics
        for (size_t j = n4 * 4; j < n_floats; ++j) {
            const uint32_t b = blk_floats ? (uint32_t)(j / blk_floats) : 0u, c = n_ch ? (uint32_t)(j % n_ch) : 0u;
            const bool masked = !(g.world == 2 || g.world == 3 || g.world == 4);
            float a = part(0)[j];
            bool all = blk_floats != 0;
            if (blk_floats)
                for (int p = 0; p < g.world; ++p)
                    for (uint32_t cc = 0; cc < n_ch; ++cc) all = all && silent(p, b, cc);
            for (int p = 1; p < g.world; ++p)
                if (!(blk_floats && masked && silent(p, b, c))) a = a + part(p)[j];
            out[j] = all ? 0.f : a;
        }
}

// The same node over parts the caller gathered itself (an all-gather's slots, peer-mapped buffers): fwgpu_bus_sum_ordered.
// sv.sil[p] = part p's silence flags [n_blocks][n_ch] or null.
__global__ __launch_bounds__(256) void k_bus_sum_ordered(BusParts bp, SilView sv, float* __restrict__ out, uint8_t* __restrict__ out_sil,
                                                         size_t n4, size_t n, uint32_t n_blocks, uint32_t frames, uint32_t n_ch) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t blk_floats = n_blocks ? frames * n_ch : 0u;
    auto part = [&](int p) -> const float* { return bp.part[p]; };
    auto silent = [&](int p, uint32_t b, uint32_t c) -> bool { return sv.sil[p] ? sv.sil[p][(size_t)b * n_ch + c] != 0 : false; };
    if (i < n4) {
        uint8_t os[4];
        const v4f y = ordered_sum_quad(bp.n, i, blk_floats, n_ch, n_blocks, part, silent, os);
        *(v4f*)(out + 4 * i) = y;
        if (out_sil && blk_floats)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t f = 4 * i + e;
                if (f % blk_floats < n_ch) out_sil[(f / blk_floats) * n_ch + f % n_ch] = os[e];
            }
    }
    if (i == 0)
        for (size_t j = n4 * 4; j < n; ++j) {
            const uint32_t b = blk_floats ? (uint32_t)(j / blk_floats) : 0u, c = n_ch ? (uint32_t)(j % n_ch) : 0u;
            const bool masked = !(bp.n == 2 || bp.n == 3 || bp.n == 4);
            float a = bp.part[0][j];
            bool all = blk_floats != 0;
            if (blk_floats)
                for (int p = 0; p < bp.n; ++p)
                    for (uint32_t cc = 0; cc < n_ch; ++cc) all = all && silent(p, b, cc);
            for (int p = 1; p < bp.n; ++p)
                if (!(blk_floats && masked && silent(p, b, c))) a = a + bp.part[p][j];
            out[j] = all ? 0.f : a;
        }
}

// Which graph-output channels of each block of a batch were flagged silent (what read_graph_outputs' mask says,
// schedule.rs:255-287) — the flags a shard's partial bus carries into the top-level SumNode.  One thread per (block, channel).
//   mode 0: the flags of `bufs[c]` (generic executor / master chain / non-stereo streams); channels past n_bufs read silent
//   mode 1: the fused plans' root SumNode (its out-mask is not stored anywhere: k_root_out keeps it in registers) —
//           all of its n_in inputs silent -> both silent; n_in == n_out -> passthrough; otherwise clear (sum.rs:52-65)
__global__ void k_out_flags(const uint8_t* __restrict__ flags, size_t flags_blk_stride, const int* __restrict__ bufs, int n_bufs, int mode,
                            int n_out_ch, int K, uint8_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= K * n_out_ch) return;
    const int b = t / n_out_ch, c = t % n_out_ch;
    const uint8_t* fl = flags + (size_t)b * flags_blk_stride;
    uint8_t s;
    if (mode == 0) {
        s = c < n_bufs ? (fl[bufs[c]] ? 1 : 0) : 1;
    } else {
        bool all = true;
        for (int j = 0; j < n_bufs; ++j) all = all && fl[bufs[j]] != 0;
        s = all ? 1 : (n_bufs == 2 && c < 2 ? (fl[bufs[c]] ? 1 : 0) : 0);
        if (c >= 2) s = 1;
    }
    out[t] = s;
}

// Host nodes (FWGPU_HOST_NODE: the caller's own AudioNodeProcessor::process, graph/processor.rs:243): the plan is cut at their
// level.  k_host_gather copies a node's INPUT buffers of the K blocks of a batch — and nothing else — into pinned,
// device-mapped host memory (stage[k][row0 + j][stride], one coalesced row per buffer) with their silence flags;
// k_host_scatter brings its OUTPUT buffers and the flags its callback reported back.  blockIdx = (frame chunk, buffer, block).
__global__ __launch_bounds__(256) void k_host_gather(const float* __restrict__ pool, const uint8_t* __restrict__ flags, int stride,
                                                     size_t pool_blk_stride, size_t flags_blk_stride, const int* __restrict__ bufs, int frames,
                                                     int row_pitch, float* __restrict__ stage, uint8_t* __restrict__ stage_flags) {
    const int j = blockIdx.y, k = blockIdx.z;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = bufs[j];
    if (f < frames) stage[((size_t)k * row_pitch + j) * stride + f] = pool[(size_t)k * pool_blk_stride + (size_t)b * stride + f];
    if (f == 0) stage_flags[(size_t)k * row_pitch + j] = flags[(size_t)k * flags_blk_stride + b];
}
__global__ __launch_bounds__(256) void k_host_scatter(float* __restrict__ pool, uint8_t* __restrict__ flags, int stride, size_t pool_blk_stride,
                                                      size_t flags_blk_stride, const int* __restrict__ bufs, int frames, int row_pitch,
                                                      const float* __restrict__ stage, const uint8_t* __restrict__ stage_flags) {
    const int j = blockIdx.y, k = blockIdx.z;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = bufs[j];
    if (f < frames) pool[(size_t)k * pool_blk_stride + (size_t)b * stride + f] = stage[((size_t)k * row_pitch + j) * stride + f];
    if (f == 0) flags[(size_t)k * flags_blk_stride + b] = stage_flags[(size_t)k * row_pitch + j];
}

// Plan adoption — everything the nodes an image activates need written into the state that outlives plans, in ONE launch
// (adoption runs at the start of a process call: every launch is host time of the audio thread):
//   blockIdx.y <  n_jobs : ext-pool job j — a slice handed to a new node: `zero_len` floats zeroed (a recycled slice starts from
//                          zeros like a fresh one; 0 for fresh ones, which are zero already) with the first n_head floats set
//                          (biquad coefficients) — one job owns its slice, so the two cannot race
//   blockIdx.y == n_jobs : the nodes' initial NodeState records (graph.rs:594-612 activate), one thread per record
struct AdoptExtJob {
    uint32_t off, zero_len, n_head, pad;
    float head[8];
};
static_assert(sizeof(AdoptExtJob) == 48, "AdoptExtJob layout");
// plan build: `rows` rows of `width` floats, `pitch` floats apart, cleared (the constant-zero buffer of every block's pool slice)
__global__ __launch_bounds__(256) void k_zero_rows(float* __restrict__ p, size_t pitch, int width, int rows) {
    for (int r = blockIdx.x; r < rows; r += gridDim.x)
        for (int i = threadIdx.x; i < width; i += blockDim.x) p[(size_t)r * pitch + i] = 0.f;
}
// plan build: byte 0 of each of `rows` rows, `pitch` bytes apart, set to v (the constant-zero buffer's silence flag of every block)
__global__ __launch_bounds__(256) void k_set_row_heads(uint8_t* __restrict__ p, size_t pitch, int rows, uint8_t v) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) p[(size_t)r * pitch] = v;
}
// plan build: every copy and fill of a build in one launch (fwgpu_plan_install.cpp, build_apply).  blockIdx.y = job; the job list
// and the copies' sources are pinned host memory.  Jobs never overlap, so they need no order among themselves.
__global__ __launch_bounds__(256) void k_build_apply(const BuildJob* __restrict__ jobs) {
    const BuildJob j = jobs[blockIdx.y];
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    uint8_t* const d = (uint8_t*)j.dst;
    if (j.src) {
        const uint8_t* const sp = (const uint8_t*)j.src;
        const bool al = (((uintptr_t)d | (uintptr_t)sp) & 15u) == 0;
        const size_t n16 = al ? j.row_bytes / 16 : 0;
        for (size_t i = tid; i < n16; i += nth) ((uint4*)d)[i] = ((const uint4*)sp)[i];
        for (size_t i = n16 * 16 + tid; i < j.row_bytes; i += nth) d[i] = sp[i];
        return;
    }
    const uint8_t v = (uint8_t)j.value;
    if ((((uintptr_t)d | j.pitch | j.row_bytes) & 3u) == 0) {
        const uint32_t v4 = v * 0x01010101u;
        const uint32_t h4 = j.head >= 0 ? ((v4 & 0xffffff00u) | (uint32_t)(j.head & 0xff)) : v4;
        const size_t wpr = j.row_bytes / 4, total = wpr * j.rows;
        for (size_t i = tid; i < total; i += nth) {
            const size_t r = i / wpr, w = i - r * wpr;
            ((uint32_t*)(d + r * j.pitch))[w] = w == 0 ? h4 : v4;
        }
    } else {
        const size_t total = j.row_bytes * j.rows;
        for (size_t i = tid; i < total; i += nth) {
            const size_t r = i / j.row_bytes, w = i - r * j.row_bytes;
            d[r * j.pitch + w] = (w == 0 && j.head >= 0) ? (uint8_t)j.head : v;
        }
    }
}
// Plan adoption: a voice whose chain is the same nodes in the new plan as in the old one keeps its steady cache (VoiceCache: "ended
// the last call steady" + the descriptor its blocks share), re-stamped with the new epoch.  Without this every voice of the graph
// runs its full state machines in the first callback after ANY edit — 50-100 us more for that callback on configs 2 and 3
// (scripts/adopt_cost.py).  The voices are matched by their sampler's state slot (stable while the node lives).
__device__ __forceinline__ void carry_cache_voice(const CarryArgs& a, const int v) {
    const VoiceDesc nv = a.new_voices[v];
    if (nv.sampler_state < 0 || nv.sampler_state >= a.n_old_slots) return;
    const int vo = a.old_slot_voice[nv.sampler_state];
    if (vo < 0) return;
    const VoiceDesc ov = a.old_voices[vo];
    bool same = nv.sampler_state == ov.sampler_state && nv.n_stages == ov.n_stages && nv.bq_state == ov.bq_state && nv.dl_state == ov.dl_state && nv.bq2_state == ov.bq2_state && nv.n_pre == ov.n_pre && nv.fx_order == ov.fx_order &&
                nv.src_kind == ov.src_kind && nv.sp_ext_off == ov.sp_ext_off;
#pragma unroll
    for (int j = 0; j < FW_MAX_STAGES - 1; ++j)
        same = same && (j >= nv.n_stages || (nv.stage_kind[j] == ov.stage_kind[j] && nv.stage_state[j] == ov.stage_state[j]));
    if (!same) return;
    VoiceCache c = a.old_cache[vo];
    if (c.epoch != a.old_epoch) return;
    c.epoch = a.new_epoch;
    a.new_cache[v] = c;
}
__global__ __launch_bounds__(256) void k_adopt_init(float* __restrict__ ext, const AdoptExtJob* __restrict__ jobs, int n_jobs,
                                                    NodeState* __restrict__ states, const uint8_t* __restrict__ inits, int n_inits, CarryArgs carry) {
    if ((int)blockIdx.y == n_jobs + 1) {  // (one launch for everything an adoption does on the device: each one is ~5 us of the audio thread)
        for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < carry.n_new; v += gridDim.x * blockDim.x) carry_cache_voice(carry, v);
        return;
    }
    if ((int)blockIdx.y < n_jobs) {
        const AdoptExtJob j = jobs[blockIdx.y];
        const uint32_t n = j.zero_len > j.n_head ? j.zero_len : j.n_head;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) ext[(size_t)j.off + i] = i < j.n_head ? j.head[i] : 0.f;
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_inits; i += gridDim.x * blockDim.x) {
        const StateInit* in = (const StateInit*)inits + i;
        states[in->index] = in->st;
    }
}
