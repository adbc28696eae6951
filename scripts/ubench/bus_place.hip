// micro-benchmark: does the PLACE of the bus in HBM decide the speed of a k_leaf_sum-shaped kernel?
// Every wave reads block k of 32 stereo streams (64 x 1 KiB, the sources: 1024 voices x 2 channels x 1 MiB, shared by
// every run) and writes the 2 x 1 KiB sum to bus[k][leaf] with non-temporal stores — 1.6 GB read, 50 MB written per launch,
// like config 2.  The bus is put at one offset after another of a large allocation, then into separately allocated
// buffers; the read-only and the write-only halves are timed at the same places.
// build: hipcc --offload-arch=gfx950 -O3 -o bus_place bus_place.hip ; run: ./bus_place [arena_GiB [step_MiB]]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(1)))* gp;
typedef v4f __attribute__((address_space(1)))* gwp;
constexpr int LEAVES = 32, K = 768, NBUS = 33, FRAMES = 256;
constexpr size_t STREAM = 262144;  // floats per channel
// store policies (gfx950 cache-policy bits of global_store): 0 nt, 1 none (L2 write-back), 2 sc1, 3 sc0 sc1, 4 nt sc0 sc1
template <int ST>
__device__ __forceinline__ void store4(float* p, v4f v) {
    if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
// MODE 0: read + write, 1: read only, 2: write only
// MAP 0: a workgroup = 4 leaves of one block; MAP 1: a workgroup = 4 consecutive blocks of one leaf (what k_leaf_sum does)
template <int MODE, int ST = 0, int LD = 0, int MAP = 0>
__global__ __launch_bounds__(256) void k_leaf(const float* __restrict__ src, float* __restrict__ bus, float* sink, size_t blk_stride, size_t leaf_stride) {
    // MAP 2: as 1 with a leaf's workgroups dispatched one after another (block group fastest); MAP 3: as 1, leaves fastest, but
    // the block groups dealt round-robin to the 8 XCDs' dispatch order (consecutive block groups 8 workgroups apart)
    const int leaf = MAP == 0 ? blockIdx.x * 4 + (threadIdx.x >> 6) : MAP == 2 ? blockIdx.y : blockIdx.x;
    const int kg = MAP == 2 ? blockIdx.x : blockIdx.y;
    const int k = MAP == 0 ? blockIdx.y : kg * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    v4f l = {0, 0, 0, 0}, r = {0, 0, 0, 0};
    if (MODE != 2) {
        for (int v0 = 0; v0 < 32; v0 += 4) {
            v4f x[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* p = src + ((size_t)(leaf * 32 + v0 + u) * 2) * STREAM + (size_t)k * FRAMES + lane * 4;
                x[2 * u] = LD ? *(gp)(uint64_t)p : __builtin_nontemporal_load((gp)(uint64_t)p);
                x[2 * u + 1] = LD ? *(gp)(uint64_t)(p + STREAM) : __builtin_nontemporal_load((gp)(uint64_t)(p + STREAM));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                l += x[2 * u];
                r += x[2 * u + 1];
            }
        }
    } else {
        l = (v4f){(float)leaf, (float)k, (float)lane, 1.f};
        r = l;
    }
    if (MODE != 1) {
        float* o = bus + (size_t)k * blk_stride + (size_t)leaf * leaf_stride + lane * 4;
        store4<ST>(o, l);
        store4<ST>(o + FRAMES, r);
    } else if (l[0] + r[1] == 123.456f) sink[0] = 1.f;
}
// "fat" waves: ONE wave renders all 32 leaves of a 128-frame half block (lanes 0-31: left, 32-63: right; 4 frames per lane),
// U voices' loads in flight at a time, the leaf sums added in a register — no leaf bus at all; 2 x 512 B written per wave.
// Each leaf's 32 voice pointers come from a table (lane p loads voice p's), the next leaf's requested a leaf ahead.
template <int U, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_fat(const float* const* __restrict__ vptr, float* __restrict__ out, int n_items) {
    const int item = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int lane = threadIdx.x & 63, k = item >> 1, half = item & 1;
    const size_t off = (size_t)k * FRAMES + half * 128 + (lane & 31) * 4 + (lane >> 5) * STREAM;
    v4f root = {0, 0, 0, 0};
    const float* nxt = vptr[lane & 31];
    for (int leaf = 0; leaf < LEAVES; ++leaf) {
        const float* mine = nxt;
        if (leaf + 1 < LEAVES) nxt = vptr[(leaf + 1) * 32 + (lane & 31)];
        v4f acc = {0, 0, 0, 0};
        for (int p0 = 0; p0 < 32; p0 += U) {
            v4f x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t b = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)mine >> 32), p0 + u) << 32) |
                                   (uint32_t)__builtin_amdgcn_readlane((int)(uint64_t)mine, p0 + u);
                x[u] = __builtin_nontemporal_load((gp)(b + off * 4));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += x[u] * 0.5f;
        }
        root += acc;
    }
    float* o = out + (size_t)k * 2 * FRAMES + (lane >> 5) * FRAMES + half * 128 + (lane & 31) * 4;
    *(v4f*)o = root;
}
template <int U, int WPB>
static float time_fat(const float* const* vptr, float* out, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int n_items = K * 2;
    for (int i = 0; i < 3; ++i) k_fat<U, WPB><<<(n_items + WPB - 1) / WPB, 64 * WPB>>>(vptr, out, n_items);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) k_fat<U, WPB><<<(n_items + WPB - 1) / WPB, 64 * WPB>>>(vptr, out, n_items);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int MODE, int ST = 0, int LD = 0, int MAP = 0>
static float time_us(const float* src, float* bus, float* sink, int reps, size_t blk_stride = (size_t)NBUS * 2 * FRAMES, size_t leaf_stride = 2 * FRAMES) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    dim3 grid(MAP == 0 ? LEAVES / 4 : MAP == 2 ? K / 4 : LEAVES, MAP == 0 ? K : MAP == 2 ? LEAVES : K / 4);
    for (int i = 0; i < 3; ++i) k_leaf<MODE, ST, LD, MAP><<<grid, 256>>>(src, bus, sink, blk_stride, leaf_stride);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) k_leaf<MODE, ST, LD, MAP><<<grid, 256>>>(src, bus, sink, blk_stride, leaf_stride);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms * 1e3f / reps;
}
int main(int argc, char** argv) {
    // ./bus_place fat: the fat-wave shape, sources and output in the same allocation order as everywhere else
    if (argc > 1 && !strcmp(argv[1], "fat")) {
        float *src, *out;
        const float** vptr;
        CK(hipMalloc(&src, (size_t)1024 * 2 * STREAM * 4));
        CK(hipMemset(src, 0, (size_t)1024 * 2 * STREAM * 4));
        CK(hipMalloc(&out, (size_t)K * 2 * FRAMES * 4));
        CK(hipMalloc(&vptr, 1024 * sizeof(float*)));
        std::vector<const float*> h(1024);
        for (int v = 0; v < 1024; ++v) h[v] = src + (size_t)v * 2 * STREAM;
        CK(hipMemcpy(vptr, h.data(), 1024 * sizeof(float*), hipMemcpyHostToDevice));
        float* sink;
        CK(hipMalloc(&sink, 256));
        printf("thin waves, read only (4 blocks of a leaf per workgroup): %.1f us\n", time_us<1, 3, 0, 1>(src, out, sink, 10));
        for (int rep = 0; rep < 2; ++rep) {
            printf("fat waves, 1536 items: U=8 %.1f  U=16 %.1f  U=32 %.1f us (1 wave / workgroup);  U=16 %.1f  U=32 %.1f us (2 waves / workgroup)\n",
                   time_fat<8, 1>(vptr, out, 10), time_fat<16, 1>(vptr, out, 10), time_fat<32, 1>(vptr, out, 10), time_fat<16, 2>(vptr, out, 10), time_fat<32, 2>(vptr, out, 10));
        }
        return 0;
    }
    // ./bus_place cands [spacer_GiB [n]]: what a context could do — n candidate buses with a spacer allocation between each two
    // (freed again at once), the sources allocated first; how long the allocations take and which candidates are fast
    if (argc > 1 && !strcmp(argv[1], "cands")) {
        const size_t gib = argc > 2 ? atoi(argv[2]) : 16;
        const int n = argc > 3 ? atoi(argv[3]) : 3;
        const size_t bus_bytes = (size_t)K * NBUS * 2 * FRAMES * 4;
        float *src, *sink;
        for (int trial = 0; trial < 3; ++trial) {  // (sources in three different places)
            char* pre = nullptr;
            if (trial) CK(hipMalloc(&pre, (size_t)trial * 20 << 30));
            CK(hipMalloc(&src, (size_t)1024 * 2 * STREAM * 4));
            CK(hipMemset(src, 0, (size_t)1024 * 2 * STREAM * 4));
            CK(hipMalloc(&sink, 256));
            std::vector<float*> cand(n);
            std::vector<char*> spacer(n, nullptr);
            hipEvent_t a, b;
            CK(hipDeviceSynchronize());
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (int i = 0; i < n; ++i) {
                CK(hipMalloc(&cand[i], bus_bytes));
                if (i + 1 < n) CK(hipMalloc(&spacer[i], gib << 30));
            }
            for (int i = 0; i + 1 < n; ++i) CK(hipFree(spacer[i]));
            clock_gettime(CLOCK_MONOTONIC, &t1);
            (void)a; (void)b;
            printf("trial %d: %d candidates %zu GiB apart allocated in %.1f ms:", trial, n, gib, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
            for (int i = 0; i < n; ++i) printf(" %.1f", time_us<0, 3>(src, cand[i], sink, 10));
            printf(" us (sc0 sc1)\n");
            for (int i = 0; i < n; ++i) CK(hipFree(cand[i]));
            CK(hipFree(src));
            CK(hipFree(sink));
            if (pre) CK(hipFree(pre));
        }
        return 0;
    }
    // ./bus_place vmm: can a PHYSICAL-ONLY spacer (hipMemCreate, never mapped, released at once) push the next hipMalloc into
    // another third of HBM, and how long does creating it take?  (round 3: the go / no-go for a fwgpu_tune_placement())
    if (argc > 1 && !strcmp(argv[1], "vmm")) {
        const size_t bus_bytes = (size_t)K * NBUS * 2 * FRAMES * 4;
        float *src, *sink;
        CK(hipMalloc(&src, (size_t)1024 * 2 * STREAM * 4));
        CK(hipMemset(src, 0, (size_t)1024 * 2 * STREAM * 4));
        CK(hipMalloc(&sink, 256));
        hipMemAllocationProp prop;
        memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("granularity %zu B\n", gran);
        for (int rep = 0; rep < 2; ++rep)
            for (size_t gib : {(size_t)0, (size_t)16, (size_t)48, (size_t)80, (size_t)112, (size_t)144, (size_t)176, (size_t)208}) {
                hipMemGenericAllocationHandle_t h{};
                struct timespec t0, t1, t2;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                bool have = false;
                if (gib) {
                    hipError_t e = hipMemCreate(&h, ((gib << 30) + gran - 1) / gran * gran, &prop, 0);
                    if (e != hipSuccess) {
                        printf("spacer %3zu GiB: hipMemCreate: %s\n", gib, hipGetErrorString(e));
                        (void)hipGetLastError();
                        continue;
                    }
                    have = true;
                }
                clock_gettime(CLOCK_MONOTONIC, &t1);
                float* bus;
                CK(hipMalloc(&bus, bus_bytes));
                if (have) CK(hipMemRelease(h));
                clock_gettime(CLOCK_MONOTONIC, &t2);
                const float us = time_us<0, 3>(src, bus, sink, 10);
                printf("spacer %3zu GiB: create %.1f ms, bus alloc + release %.1f ms, read + write %.1f us (sc0 sc1), bus %p\n", gib,
                       (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6, (t2.tv_sec - t1.tv_sec) * 1e3 + (t2.tv_nsec - t1.tv_nsec) * 1e-6, us,
                       (void*)bus);
                CK(hipFree(bus));
            }
        return 0;
    }
    // ./bus_place matrix [chunks [chunk_GiB]]: sources in chunk i, bus in chunk j of `chunks` separately allocated pieces
    if (argc > 1 && !strcmp(argv[1], "matrix")) {
        const int n = argc > 2 ? atoi(argv[2]) : 10;
        const size_t gib = argc > 3 ? atoi(argv[3]) : 8;
        std::vector<char*> c(n);
        float* sink;
        CK(hipMalloc(&sink, 256));
        for (int i = 0; i < n; ++i) {
            CK(hipMalloc(&c[i], gib << 30));
            CK(hipMemset(c[i], 0, gib << 30));
        }
        printf("%d chunks of %zu GiB in allocation order; rows: sources in chunk i (read-only us in the last column); columns: bus in chunk j\n      ", n, gib);
        for (int j = 0; j < n; ++j) printf(" %6d", j);
        printf("   rd only\n");
        for (int i = 0; i < n; ++i) {
            printf("src %2d", i);
            for (int j = 0; j < n; ++j) printf(" %6.1f", time_us<0>((const float*)c[i], (float*)(c[j] + ((size_t)3 << 30)), sink, 8));
            printf("   %6.1f\n", time_us<1>((const float*)c[i], (float*)c[i], sink, 8));
        }
        // store policies on one slow and one fast pairing
        int slow_j = -1, fast_j = -1;
        for (int j = 0; j < n; ++j) {
            const float t = time_us<0>((const float*)c[0], (float*)(c[j] + ((size_t)3 << 30)), sink, 8);
            if (t > 270.f && slow_j < 0) slow_j = j;
            if (t < 260.f && fast_j < 0) fast_j = j;
        }
        printf("sources in chunk 0; read + write us by store policy:\n%22s %9s %9s %9s %9s %9s\n", "", "nt", "none", "sc1", "sc0 sc1", "nt sc0sc1");
        for (int j : {slow_j, fast_j}) {
            if (j < 0) continue;
            float* bus = (float*)(c[j] + ((size_t)3 << 30));
            const float* src = (const float*)c[0];
            for (int rep = 0; rep < 2; ++rep)
                printf("bus in chunk %2d (%s) %9.1f %9.1f %9.1f %9.1f %9.1f\n", j, j == slow_j ? "slow" : "fast", time_us<0, 0>(src, bus, sink, 10),
                       time_us<0, 1>(src, bus, sink, 10), time_us<0, 2>(src, bus, sink, 10), time_us<0, 3>(src, bus, sink, 10), time_us<0, 4>(src, bus, sink, 10));
        }
        // workgroup shapes and bus layouts (sc0 sc1 stores): [k][leaf] = a block's 32 leaves adjacent (the library's), [leaf][k] =
        // a leaf's blocks adjacent
        printf("dispatch order, 4 blocks per workgroup, [k][l]: leaves fastest / a leaf's block groups fastest:\n");
        for (int j : {slow_j, fast_j}) {
            if (j < 0) continue;
            float* bus = (float*)(c[j] + ((size_t)3 << 30));
            const float* src = (const float*)c[0];
            for (int rep = 0; rep < 2; ++rep)
                printf("bus in chunk %2d (%s) %12.1f %12.1f   read only: %12.1f %12.1f\n", j, j == slow_j ? "slow" : "fast", time_us<0, 3, 0, 1>(src, bus, sink, 10),
                       time_us<0, 3, 0, 2>(src, bus, sink, 10), time_us<1, 3, 0, 1>(src, bus, sink, 10), time_us<1, 3, 0, 2>(src, bus, sink, 10));
        }
        printf("sources in chunk 0; read + write us by workgroup shape / bus layout:\n%22s %12s %12s %12s %12s\n", "", "4 leaves,[k][l]", "4 blocks,[k][l]", "4 blocks,[l][k]", "4 leaves,[l][k]");
        for (int j : {slow_j, fast_j}) {
            if (j < 0) continue;
            float* bus = (float*)(c[j] + ((size_t)3 << 30));
            const float* src = (const float*)c[0];
            const size_t kl_b = (size_t)NBUS * 2 * FRAMES, kl_l = 2 * FRAMES, lk_b = 2 * FRAMES, lk_l = (size_t)K * 2 * FRAMES;
            for (int rep = 0; rep < 2; ++rep)
                printf("bus in chunk %2d (%s) %12.1f %12.1f %12.1f %12.1f\n", j, j == slow_j ? "slow" : "fast", time_us<0, 3, 0, 0>(src, bus, sink, 10, kl_b, kl_l),
                       time_us<0, 3, 0, 1>(src, bus, sink, 10, kl_b, kl_l), time_us<0, 3, 0, 1>(src, bus, sink, 10, lk_b, lk_l), time_us<0, 3, 0, 0>(src, bus, sink, 10, lk_b, lk_l));
        }
        return 0;
    }
    const size_t arena_gib = argc > 1 ? atoi(argv[1]) : 6, step_mib = argc > 2 ? atoi(argv[2]) : 64;
    const size_t bus_bytes = (size_t)K * NBUS * 2 * FRAMES * 4;
    float *src, *sink;
    char* arena;
    CK(hipMalloc(&src, (size_t)1024 * 2 * STREAM * 4));
    CK(hipMemset(src, 0, (size_t)1024 * 2 * STREAM * 4));
    CK(hipMalloc(&sink, 256));
    CK(hipMalloc(&arena, arena_gib << 30));
    printf("bus %.1f MB; read 1.61 GB per launch; src %p arena %p\n", bus_bytes / 1e6, (void*)src, (void*)arena);
    printf("%10s %9s %9s %9s\n", "offset MiB", "rd+wr us", "rd us", "wr us");
    for (size_t off = 0; off + bus_bytes <= (arena_gib << 30); off += step_mib << 20) {
        float* bus = (float*)(arena + off);
        printf("%10zu %9.1f %9.1f %9.1f\n", off >> 20, time_us<0>(src, bus, sink, 10), time_us<1>(src, bus, sink, 5), time_us<2>(src, bus, sink, 10));
    }
    printf("store policies (read + write us), plain loads in the last column:\n%10s %9s %9s %9s %9s %9s %9s\n", "offset MiB", "nt", "none", "sc1", "sc0 sc1", "nt sc0sc1", "ld+nt st");
    for (size_t off = 0; off + bus_bytes <= (arena_gib << 30); off += (size_t)512 << 20) {
        float* bus = (float*)(arena + off);
        printf("%10zu %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f\n", off >> 20, time_us<0, 0>(src, bus, sink, 10), time_us<0, 1>(src, bus, sink, 10),
               time_us<0, 2>(src, bus, sink, 10), time_us<0, 3>(src, bus, sink, 10), time_us<0, 4>(src, bus, sink, 10), time_us<0, 0, 1>(src, bus, sink, 10));
    }
    printf("the bus base shifted by a little, at arena offsets 0 and %zu MiB (read + write us, nt):\n", (arena_gib << 9));
    for (size_t sh = 256; sh <= ((size_t)64 << 20); sh *= 2) {
        printf("%10zu B %9.1f %9.1f", sh, time_us<0>(src, (float*)(arena + sh), sink, 10), time_us<0>(src, (float*)(arena + (arena_gib << 29) + sh), sink, 10));
        printf("   x3: %9.1f %9.1f\n", time_us<0>(src, (float*)(arena + 3 * sh), sink, 10), time_us<0>(src, (float*)(arena + (arena_gib << 29) + 3 * sh), sink, 10));
    }
    printf("other bus pitches per block (floats), same two places:\n");
    for (size_t st : {(size_t)32 * 512, (size_t)33 * 512, (size_t)34 * 512, (size_t)36 * 512, (size_t)40 * 512, (size_t)48 * 512, (size_t)64 * 512, (size_t)65 * 512, (size_t)33 * 512 + 64, (size_t)33 * 512 + 1024})
        printf("%10zu %9.1f %9.1f\n", st, time_us<0>(src, (float*)arena, sink, 10, st), time_us<0>(src, (float*)(arena + (arena_gib << 29)), sink, 10, st));
    printf("separately allocated buses:\n");
    std::vector<float*> keep;
    for (int i = 0; i < 12; ++i) {
        float* bus;
        CK(hipMalloc(&bus, bus_bytes));
        keep.push_back(bus);
        printf("%10p %9.1f %9s %9.1f\n", (void*)bus, time_us<0>(src, bus, sink, 10), "", time_us<2>(src, bus, sink, 10));
    }
    return 0;
}
