// micro-benchmark: read-only streaming bandwidth of one MI355X with the access pattern of k_leaf_sum
// (each wave reads 1 KiB contiguous chunks of many independent streams) vs a plain linear read.
// build: hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip ; run: ./read_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(1)))* gp;
template <int NT, int U>
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ src, float* out, size_t n4_per_wave) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const float* p = src + wave * n4_per_wave * 256 + lane * 4;
    v4f acc = {0, 0, 0, 0};
    for (size_t i = 0; i < n4_per_wave; i += U) {
        v4f x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = NT ? __builtin_nontemporal_load((gp)(uint64_t)(p + (i + u) * 256)) : *(gp)(uint64_t)(p + (i + u) * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += x[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}
// streams: wave w reads block k of 32 streams x 2 channels (like a leaf of 32 voices), 1 KiB each
template <int NT>
__global__ __launch_bounds__(256) void k_streams(const float* __restrict__ src, float* out, int n_leaves, size_t stream_floats) {
    const int leaf = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int k = blockIdx.y, lane = threadIdx.x & 63;
    if (leaf >= n_leaves) return;
    v4f acc = {0, 0, 0, 0};
    for (int v0 = 0; v0 < 32; v0 += 4) {
        v4f x[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* p = src + ((size_t)(leaf * 32 + v0 + u) * 2) * stream_floats + (size_t)k * 256 + lane * 4;
            x[2 * u] = NT ? __builtin_nontemporal_load((gp)(uint64_t)p) : *(gp)(uint64_t)p;
            x[2 * u + 1] = NT ? __builtin_nontemporal_load((gp)(uint64_t)(p + stream_floats)) : *(gp)(uint64_t)(p + stream_floats);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}
// the same pattern with every stream starting m = (stream % 4) floats off a 16-byte boundary — what a looping sample whose
// length is not a multiple of 4 frames looks like after its first wrap.  MODE 1: the plain (4-byte aligned) dwordx4 load.
// MODE 2: an ALIGNED dwordx4 per lane + the neighbour lane's first m floats by DPP wave_shl:1, and for lane 63 alone one
// unaligned load of its own 16 bytes.
__device__ __forceinline__ float dpp_next(float x) {  // lane i <- lane i + 1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
template <int MODE>
__device__ __forceinline__ v4f load_mis(const float* p, int lane) {
    if (MODE == 1) return __builtin_nontemporal_load((gp)(uint64_t)p);
    const int m = (int)(((uint64_t)p >> 2) & 3);  // wave-uniform
    const v4f q = __builtin_nontemporal_load((gp)(uint64_t)(p - m));
    if (m == 0) return q;
    v4f own = q;
    if (lane == 63) own = __builtin_nontemporal_load((gp)(uint64_t)p);
    const float n0 = dpp_next(q[0]), n1 = dpp_next(q[1]), n2 = dpp_next(q[2]);
    v4f r;
    if (m == 1) r = (v4f){q[1], q[2], q[3], n0};
    else if (m == 2) r = (v4f){q[2], q[3], n0, n1};
    else r = (v4f){q[3], n0, n1, n2};
    return lane == 63 ? own : r;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_streams_mis(const float* __restrict__ src, float* out, int n_leaves, size_t stream_floats) {
    const int leaf = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int k = blockIdx.y, lane = threadIdx.x & 63;
    if (leaf >= n_leaves) return;
    v4f acc = {0, 0, 0, 0};
    for (int v0 = 0; v0 < 32; v0 += 4) {
        v4f x[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int stream = leaf * 32 + v0 + u;
            const float* p = src + ((size_t)stream * 2) * stream_floats + (size_t)k * 256 + lane * 4 + (stream & 3);
            x[2 * u] = load_mis<MODE>(p, lane);
            x[2 * u + 1] = load_mis<MODE>(p + stream_floats, lane);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}
int main() {
    const size_t bytes = 2ull << 30;
    float *src, *out;
    hipMalloc(&src, bytes);
    hipMalloc(&out, 256);
    hipMemset(src, 0, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto time = [&](const char* name, auto launch, double moved) {
        launch();
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(a);
            launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        printf("%-40s %.1f us  %.2f TB/s\n", name, best * 1e3, moved / (best * 1e-3) / 1e12);
    };
    // linear: 2 GiB over N waves
    for (int waves_per_cu : {8, 16, 32}) {
        const size_t n_waves = 256ull * waves_per_cu, n4 = bytes / 16 / 64 / n_waves;
        char nm[64];
        snprintf(nm, 64, "linear nt U=8 %d waves/CU", waves_per_cu);
        time(nm, [&] { hipLaunchKernelGGL((k_linear<1, 8>), dim3(n_waves / 4), dim3(256), 0, 0, src, out, n4); }, (double)n_waves * n4 * 1024);
        snprintf(nm, 64, "linear plain U=8 %d waves/CU", waves_per_cu);
        time(nm, [&] { hipLaunchKernelGGL((k_linear<0, 8>), dim3(n_waves / 4), dim3(256), 0, 0, src, out, n4); }, (double)n_waves * n4 * 1024);
    }
    // config-2 pattern: 32 leaves x 256 blocks, 1024 streams x 2 ch x 262144 floats = 2 GiB
    time("cfg2 pattern nt (512 MiB)", [&] { hipLaunchKernelGGL((k_streams<1>), dim3(8, 256), dim3(256), 0, 0, src, out, 32, (size_t)262144); }, 512.0 * 1048576);
    time("cfg2 pattern plain (512 MiB)", [&] { hipLaunchKernelGGL((k_streams<0>), dim3(8, 256), dim3(256), 0, 0, src, out, 32, (size_t)262144); }, 512.0 * 1048576);
    // 768 blocks (1.5 GiB), as one call of config 2
    time("cfg2 pattern nt, 768 blocks, aligned", [&] { hipLaunchKernelGGL((k_streams<1>), dim3(8, 768), dim3(256), 0, 0, src, out, 32, (size_t)262144); }, 1536.0 * 1048576);
    time("  streams off by 0..3 floats, dwordx4", [&] { hipLaunchKernelGGL((k_streams_mis<1>), dim3(8, 768), dim3(256), 0, 0, src, out, 32, (size_t)262144); }, 1536.0 * 1048576);
    time("  ... aligned load + DPP shift", [&] { hipLaunchKernelGGL((k_streams_mis<2>), dim3(8, 768), dim3(256), 0, 0, src, out, 32, (size_t)262144); }, 1536.0 * 1048576);
    return 0;
}
