/* fw_host.c — a plain C99 host of libfwgpu (no Python, no C++): what a Firewheel backend written in a compiled
 * language does through include/fwgpu.h.  It builds the config-2 graph shape with the graph-edit API
 * (graph/graph.rs:201-231, :396-477), activates it (graph/context.rs:93-137), drives it one callback at a time the
 * way the cpal backend drives FirewheelProcessor::process_interleaved (cpal/lib.rs:378-449, graph/processor.rs:61-165)
 * with control messages in between, and writes the interleaved stereo stream it got back as raw f32.
 *
 *   fw_host <voices> <block_frames> <callbacks> <out.f32>
 *
 * tests/test_gpu_parity.py runs it on the GPU box and compares the file bit for bit with the oracle fed the same
 * graph, samples and messages (the generators below are restated there).  */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "fwgpu.h"

#define SRC_FRAMES 1900u /* not a multiple of any block size: loops wrap inside blocks */
#define RADIX 8

static uint32_t xorshift32(uint32_t* s) {
    uint32_t x = *s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return *s = x;
}

static int die(fwgpu_ctx* cx, const char* what, long long rc) {
    fprintf(stderr, "fw_host: %s failed (%lld): %s\n", what, rc, cx ? fwgpu_last_error(cx) : fwgpu_create_error());
    return 1;
}
#define CK(call)                                   \
    do {                                           \
        long long rc__ = (long long)(call);        \
        if (rc__ < 0) return die(cx, #call, rc__); \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 5) {
        fprintf(stderr, "usage: fw_host <voices> <block_frames> <callbacks> <out.f32>\n");
        return 2;
    }
    const int voices = atoi(argv[1]);
    const uint32_t block = (uint32_t)atoi(argv[2]);
    const int callbacks = atoi(argv[3]);
    if (voices < 1 || voices > 4096 || block < 1 || callbacks < 1) return 2;

    fwgpu_ctx* cx = fwgpu_ctx_create(0, 48000, block, 0, 2, NULL);
    if (!cx) return die(NULL, "fwgpu_ctx_create", -1);

    int64_t* sampler = (int64_t*)malloc(sizeof(int64_t) * (size_t)voices);
    int64_t* volume = (int64_t*)malloc(sizeof(int64_t) * (size_t)voices);
    int64_t* level = (int64_t*)malloc(sizeof(int64_t) * (size_t)voices);
    float* data = (float*)malloc(sizeof(float) * 2u * SRC_FRAMES);
    float* out = (float*)malloc(sizeof(float) * 2u * block * 2u);
    if (!sampler || !volume || !level || !data || !out) return 2;

    /* V x (SamplerNode -> VolumeNode -> pan) */
    for (int v = 0; v < voices; ++v) {
        const float p_sampler = 100.0f;
        const float p_volume = (float)(10 + (v * 37) % 90);
        const float p_pan = (float)((v * 53) % 200 - 100) / 100.0f;
        int64_t s, g, p;
        CK(s = fwgpu_add_node(cx, FWGPU_SAMPLER, 0, 2, &p_sampler, 1));
        CK(g = fwgpu_add_node(cx, FWGPU_VOLUME, 2, 2, &p_volume, 1));
        CK(p = fwgpu_add_node(cx, FWGPU_STEREO_PAN, 2, 2, &p_pan, 1));
        for (uint32_t c = 0; c < 2; ++c) {
            CK(fwgpu_connect(cx, s, c, g, c, 0));
            CK(fwgpu_connect(cx, g, c, p, c, 0));
        }
        sampler[v] = s;
        volume[v] = g;
        level[v] = p;
    }
    /* radix-8 SumNode tree -> graph_out */
    int n = voices;
    for (;;) {
        int m = 0;
        for (int i = 0; i < n; i += RADIX) {
            const int ports = n - i < RADIX ? n - i : RADIX;
            int64_t sum;
            CK(sum = fwgpu_add_node(cx, FWGPU_SUM, (uint32_t)(2 * ports), 2, NULL, 0));
            for (int p = 0; p < ports; ++p)
                for (uint32_t c = 0; c < 2; ++c) CK(fwgpu_connect(cx, level[i + p], c, sum, (uint32_t)(2 * p) + c, 0));
            level[m++] = sum;
        }
        n = m;
        if (n == 1) break;
    }
    for (uint32_t c = 0; c < 2; ++c) CK(fwgpu_connect(cx, level[0], c, fwgpu_graph_out_node(cx), c, 0));
    CK(fwgpu_update(cx));

    /* samples: planar stereo f32, xorshift32 seeded per voice; every third voice one-shot, the rest looping */
    for (int v = 0; v < voices; ++v) {
        uint32_t st = 0x9E3779B9u ^ (uint32_t)(v * 2654435761u + 1u);
        for (uint32_t i = 0; i < 2u * SRC_FRAMES; ++i)
            data[i] = (float)(xorshift32(&st) >> 8) * (1.0f / 8388608.0f) - 1.0f; /* exact in f32 */
        int smp;
        CK(smp = fwgpu_sample_create(cx, FWGPU_PLANAR_F32, 2, SRC_FRAMES, data));
        CK(fwgpu_sampler_set_sample(cx, sampler[v], smp, 0, 0));
        if (v % 3 != 2) CK(fwgpu_sampler_set_loop_range(cx, sampler[v], 1, 0.0, 0.0, 0));
        CK(fwgpu_sampler_play(cx, sampler[v], 0));
    }

    FILE* f = fopen(argv[4], "wb");
    if (!f) return 2;
    for (int cb = 0; cb < callbacks; ++cb) {
        if (cb == callbacks / 3) /* a burst of parameter changes: smoother ramps */
            for (int v = 0; v < voices; v += 2) CK(fwgpu_node_set_param(cx, volume[v], 0, 20.0f + (float)(v % 7) * 10.0f, 0));
        if (cb == callbacks / 2) /* pause a few voices: silence masks through the tree */
            for (int v = 1; v < voices; v += 5) CK(fwgpu_sampler_pause(cx, sampler[v], 0));
        const uint64_t frames = (cb % 4 == 3) ? 2u * block : block; /* some callbacks ask for two blocks */
        CK(fwgpu_process_interleaved(cx, NULL, out, 0, 2, frames, 0.0, 0));
        if (fwrite(out, sizeof(float), (size_t)frames * 2u, f) != (size_t)frames * 2u) return 2;
    }
    fclose(f);
    char name[128];
    int cus = 0;
    uint64_t hbm = 0;
    CK(fwgpu_device_info(cx, name, (int)sizeof name, &cus, &hbm));
    printf("fw_host: %d voices, block %u, %d callbacks, plan kind %d on %s (%d CUs)\n", voices, block, callbacks,
           fwgpu_plan_kind(cx), name, cus);
    fwgpu_ctx_destroy(cx);
    free(sampler);
    free(volume);
    free(level);
    free(data);
    free(out);
    return 0;
}
