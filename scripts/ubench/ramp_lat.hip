// ramp_lat.hip — how long is ONE smoother ramp?  The recurrence y[i] = in_a + (y[i-1] * b) (core/param/smoother.rs:171-175: two
// roundings per frame, serial by definition) bounds the control kernel's critical path whenever a gain glides.  Measures, on
// one wave: (a) the plain dependent mul + add chain, (b) the same with EXEC shrinking lane by lane (lane i keeps y[i]: what
// k_voice_control's ramp_emit does), (c) the old formulation with a compare + select per step.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ramp_lat ramp_lat.hip && ./ramp_lat
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_plain(float* out, float in_a, float b, int steps) {
    float y = out[0];
    for (int i = 0; i < steps; ++i) y = in_a + (y * b);
    out[threadIdx.x] = y;
}
__global__ void k_exec(float* out, float in_a, float b, int chunks) {
    float prev = out[0], acc = 0.f;
    for (int c = 0; c < chunks; ++c) {
        float y = prev, t;
        unsigned long long saved;
        asm volatile(
            "s_mov_b64 %2, exec\n"
            ".rept 64\n"
            "v_mul_f32 %1, %0, %4\n"
            "v_add_f32 %0, %3, %1\n"
            "s_lshl_b64 exec, exec, 1\n"
            ".endr\n"
            "s_mov_b64 exec, %2\n"
            : "+v"(y), "=&v"(t), "=&s"(saved)
            : "v"(in_a), "v"(b)
            : "scc");
        acc += y;
        prev = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));
    }
    out[threadIdx.x] = acc;
}
#define REP4(x) x x x x
#define REP64(x) REP4(REP4(REP4(x)))
#define STEP4                                                                                              \
    "v_mul_f32 %4, %3, %7\n" "v_add_f32 %0, %6, %4\n" "v_mul_f32 %4, %0, %7\n" "v_add_f32 %1, %6, %4\n" \
    "v_mul_f32 %4, %1, %7\n" "v_add_f32 %2, %6, %4\n" "v_mul_f32 %4, %2, %7\n" "v_add_f32 %3, %6, %4\n" \
    "s_lshl_b64 exec, exec, 1\n"
// (d) EXEC shrinking once per FOUR frames: lane q keeps frames 4q .. 4q+3 (what ramp_run does for whole 256-frame pieces)
__global__ void k_exec4(float* out, float in_a, float b, int chunks) {
    float prev = out[0], acc = 0.f;
    for (int c = 0; c < chunks; c += 4) {
        float d0, d1, d2, d3 = prev, t;
        unsigned long long saved;
        asm volatile("s_mov_b64 %5, exec\n" REP64(STEP4) "s_mov_b64 exec, %5\n"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "+v"(d3), "=&v"(t), "=&s"(saved)
                     : "v"(in_a), "v"(b)
                     : "scc");
        acc += d0 + d1 + d2 + d3;
        prev = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d3), 63));
    }
    out[threadIdx.x] = acc;
}
__global__ void k_select(float* out, float in_a, float b, int chunks) {
    float prev = out[0], acc = 0.f;
    const int lane = threadIdx.x;
    for (int c = 0; c < chunks; ++c) {
        float mine = 0.f;
#pragma unroll 16
        for (int i = 0; i < 64; ++i) {
            prev = in_a + (prev * b);
            mine = i == lane ? prev : mine;
        }
        acc += mine;
    }
    out[threadIdx.x] = acc;
}

int main() {
    float* d;
    hipMalloc(&d, 64 * sizeof(float));
    hipMemset(d, 0, 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int chunks = 4 * 21 * 8;  // 8 ramps of 21 blocks of 256 frames
    for (int which = 0; which < 4; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, 0, d, 0.002f, 0.998f, chunks * 64);
            if (which == 1) hipLaunchKernelGGL(k_exec, dim3(1), dim3(64), 0, 0, d, 0.002f, 0.998f, chunks);
            if (which == 2) hipLaunchKernelGGL(k_select, dim3(1), dim3(64), 0, 0, d, 0.002f, 0.998f, chunks);
            if (which == 3) hipLaunchKernelGGL(k_exec4, dim3(1), dim3(64), 0, 0, d, 0.002f, 0.998f, chunks);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double steps = (double)chunks * 64;
        printf("%s: %.1f us for %d frames = %.2f ns per frame = %.2f us per 256-frame block, %.1f us per 21-block ramp\n",
               which == 0 ? "plain chain " : which == 1 ? "exec-shrink " : which == 2 ? "cmp + select" : "exec-shrink/4", best * 1e3, (int)steps, best * 1e6 / steps,
               best * 1e3 / steps * 256, best * 1e3 / steps * 256 * 21);
    }
    return 0;
}
