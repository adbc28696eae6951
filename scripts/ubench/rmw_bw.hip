// micro-benchmark: the HBM ceiling for config 3's traffic mix — per element one streamed read (the source, non-temporal),
// one read and one write of a second array in place (the delay ring): 2 reads : 1 write, nothing else in the kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o rmw_bw rmw_bw.hip ; run: ./rmw_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(1)))* gp;
// ST: the ring store's cache policy — 0 plain, 1 nt, 2 sc0 sc1 (write-through), 3 sc1
template <int ST>
__device__ __forceinline__ void store4(float* p, v4f v) {
    if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int U, int ST = 0>
__global__ __launch_bounds__(256) void k_rmw(const float* __restrict__ src, float* ring, size_t n4_per_wave) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const float* p = src + wave * n4_per_wave * 256 + lane * 4;
    float* r = ring + wave * n4_per_wave * 256 + lane * 4;
    for (size_t i = 0; i < n4_per_wave; i += U) {
        v4f x[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = __builtin_nontemporal_load((gp)(uint64_t)(p + (i + u) * 256));
            d[u] = *(const v4f*)(r + (i + u) * 256);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) store4<ST>(r + (i + u) * 256, x[u] + d[u] * 0.5f);
    }
}
int main() {
    const size_t bytes = 1ull << 30;  // per array
    float *src, *ring;
    hipMalloc(&src, bytes);
    hipMalloc(&ring, bytes);
    hipMemset(src, 0, bytes);
    hipMemset(ring, 0, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int st = 0; st < 4; ++st)
    for (int waves_per_cu : {8, 16, 32, 64}) {
        const size_t n_waves = 256ull * waves_per_cu, n4 = bytes / 16 / 64 / n_waves;
        auto launch = [&] {
            if (st == 0) hipLaunchKernelGGL((k_rmw<4, 0>), dim3(n_waves / 4), dim3(256), 0, 0, src, ring, n4);
            if (st == 1) hipLaunchKernelGGL((k_rmw<4, 1>), dim3(n_waves / 4), dim3(256), 0, 0, src, ring, n4);
            if (st == 2) hipLaunchKernelGGL((k_rmw<4, 2>), dim3(n_waves / 4), dim3(256), 0, 0, src, ring, n4);
            if (st == 3) hipLaunchKernelGGL((k_rmw<4, 3>), dim3(n_waves / 4), dim3(256), 0, 0, src, ring, n4);
        };
        launch();
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        printf("2 reads : 1 write, U=4, store %-8s %2d waves/CU: %.1f us  %.2f TB/s total\n", st == 0 ? "plain" : st == 1 ? "nt" : st == 2 ? "sc0 sc1" : "sc1", waves_per_cu, best * 1e3,
               3.0 * n_waves * n4 * 1024 / (best * 1e-3) / 1e12);
    }
    return 0;
}
