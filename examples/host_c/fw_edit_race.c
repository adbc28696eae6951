/* fw_edit_race.c — graph edits while the audio thread runs (graph/context.rs:93-137 + graph/processor.rs:167-206 through the
 * C ABI): what does an edit cost the callbacks?
 *
 * A config-3 shaped graph — V voices of sampler -> biquad LPF -> delay -> gain under a radix-32 SumNode tree — is driven one
 * block per callback by an audio thread (fwgpu_process_interleaved, the cpal callback's call, cpal/lib.rs:429-437) while a
 * control thread replaces one voice after another: remove its four nodes, add four new ones into the same mixer port,
 * fwgpu_update (a full recompile and re-upload of the launch plan: milliseconds — on THIS thread, off to the side), start the
 * new voice.  Every callback is timed on the monotonic clock.
 *
 *   fw_edit_race <voices> <block_frames> <steady_callbacks> <edits>
 *
 * prints one JSON line: callback time steady vs while edits are in flight (median / p99 / max, microseconds), how long the
 * updates took on the control thread, and the hand-over statistics of the library (plans adopted by a callback, the longest
 * adoption).  tests/test_gpu_benched_shapes.py runs it; profiles/r03_edit_race_cfg3.json keeps a run. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fwgpu.h"

#define SRC_FRAMES 24000u
#define RADIX 32

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
static int cmp_d(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static void die(fwgpu_ctx* cx, const char* what, long long rc) {
    fprintf(stderr, "fw_edit_race: %s failed (%lld): %s\n", what, rc, cx ? fwgpu_last_error(cx) : fwgpu_create_error());
    exit(1);
}
#define CK(call)                                \
    do {                                        \
        long long rc__ = (long long)(call);     \
        if (rc__ < 0) die(g_cx, #call, rc__);   \
    } while (0)

static fwgpu_ctx* g_cx;
static int g_sample;
static uint32_t g_block;
typedef struct {
    int64_t s, bq, dl, vol, leaf;
    uint32_t port;
} Voice;
static Voice* g_voice;
static int g_voices;

static void make_voice(Voice* v, int idx) {
    float p_s[1] = {100.f};
    float p_bq[3] = {0.f, 300.f + (float)((idx * 977) % 7000), 0.707f};
    float p_dl[3] = {0.010f + (float)((idx * 131) % 200) * 0.001f, 0.3f, 0.5f};
    float p_vol[1] = {20.f + (float)((idx * 37) % 80)};
    CK(v->s = fwgpu_add_node(g_cx, FWGPU_SAMPLER, 0, 2, p_s, 1));
    CK(v->bq = fwgpu_add_node(g_cx, FWGPU_BIQUAD, 2, 2, p_bq, 3));
    CK(v->dl = fwgpu_add_node(g_cx, FWGPU_DELAY, 2, 2, p_dl, 3));
    CK(v->vol = fwgpu_add_node(g_cx, FWGPU_VOLUME, 2, 2, p_vol, 1));
    for (uint32_t ch = 0; ch < 2; ++ch) {
        CK(fwgpu_connect(g_cx, v->s, ch, v->bq, ch, 0));
        CK(fwgpu_connect(g_cx, v->bq, ch, v->dl, ch, 0));
        CK(fwgpu_connect(g_cx, v->dl, ch, v->vol, ch, 0));
        CK(fwgpu_connect(g_cx, v->vol, ch, v->leaf, 2 * v->port + ch, 0));
    }
}
static void start_voice(const Voice* v) {
    CK(fwgpu_sampler_set_sample(g_cx, v->s, g_sample, 0, 0));
    CK(fwgpu_sampler_set_loop_range(g_cx, v->s, 1, 0.0, 0.0, 0));
    CK(fwgpu_sampler_play(g_cx, v->s, 0));
}

/* ---- the audio thread */
typedef struct {
    volatile int stop;
    volatile int editing; /* set by the editor while an edit (remove .. update .. start) is in flight */
    double* t_us;         /* per callback */
    unsigned char* tag;   /* 1 = an edit was in flight when the callback started */
    unsigned char* phase; /* the library's update phase when the callback started (diagnostics) */
    long n, cap;
} Audio;
static double g_period_us = 0.0;
static void* audio_main(void* arg) {
    Audio* a = (Audio*)arg;
    float* out = (float*)malloc(sizeof(float) * 2 * g_block);
    double stream_time = 0.0;
    double next = now_us();
    while (!a->stop && a->n < a->cap) {
        if (g_period_us > 0) { /* a paced stream: the callback of a device with this period (0: back to back, the stress case) */
            next += g_period_us;
            while (now_us() < next) {
            }
        }
        const int tag = a->editing;
        const int ph = fwgpu_update_phase(g_cx);
        const double t0 = now_us();
        long long rc = fwgpu_process_interleaved(g_cx, NULL, out, 0, 2, g_block, stream_time, 0);
        const double t1 = now_us();
        if (rc < 0) die(g_cx, "fwgpu_process_interleaved", rc);
        a->t_us[a->n] = t1 - t0;
        a->tag[a->n] = (unsigned char)tag;
        a->phase[a->n] = (unsigned char)ph;
        a->n++;
        stream_time += g_block / 48000.0;
    }
    free(out);
    return NULL;
}
static void stats(const double* t, const unsigned char* tag, long n, int want, double* med, double* p99, double* mx, long* cnt) {
    double* v = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    long m = 0;
    for (long i = 50; i < n; ++i) /* (the first callbacks warm the stack up) */
        if (tag[i] == want) v[m++] = t[i];
    qsort(v, (size_t)m, sizeof(double), cmp_d);
    *cnt = m;
    *med = m ? v[m / 2] : 0;
    *p99 = m ? v[(long)(m * 0.99)] : 0;
    *mx = m ? v[m - 1] : 0;
    free(v);
}

int main(int argc, char** argv) {
    if (argc != 5 && argc != 6) {
        fprintf(stderr, "usage: fw_edit_race <voices> <block_frames> <steady_callbacks> <edits> [callback_period_us]\n");
        return 2;
    }
    if (argc == 6) g_period_us = atof(argv[5]);
    g_voices = atoi(argv[1]);
    g_block = (uint32_t)atoi(argv[2]);
    const long steady = atol(argv[3]);
    const int edits = atoi(argv[4]);
    g_cx = fwgpu_ctx_create(0, 48000, g_block, 0, 2, NULL);
    if (!g_cx) die(NULL, "fwgpu_ctx_create", -1);
    CK(fwgpu_set_max_batch(g_cx, 8));
    float* src = (float*)malloc(sizeof(float) * 2 * SRC_FRAMES);
    uint32_t st = 12345u;
    for (uint32_t i = 0; i < 2 * SRC_FRAMES; ++i) {
        st ^= st << 13;
        st ^= st >> 17;
        st ^= st << 5;
        src[i] = (float)(st >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    CK(g_sample = fwgpu_sample_create(g_cx, FWGPU_PLANAR_F32, 2, SRC_FRAMES, src));
    /* the tree: leaves of RADIX voices, then radix-RADIX sums up to the root */
    g_voice = (Voice*)calloc((size_t)g_voices, sizeof(Voice));
    const int n_leaves = (g_voices + RADIX - 1) / RADIX;
    int64_t* level = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_leaves);
    for (int l = 0; l < n_leaves; ++l) {
        const int ports = g_voices - l * RADIX < RADIX ? g_voices - l * RADIX : RADIX;
        CK(level[l] = fwgpu_add_node(g_cx, FWGPU_SUM, 2 * (uint32_t)ports, 2, NULL, 0));
        for (int p = 0; p < ports; ++p) {
            Voice* v = &g_voice[l * RADIX + p];
            v->leaf = level[l];
            v->port = (uint32_t)p;
            make_voice(v, l * RADIX + p);
        }
    }
    int n = n_leaves;
    while (n > 1) {
        const int m = (n + RADIX - 1) / RADIX;
        for (int i = 0; i < m; ++i) {
            const int ports = n - i * RADIX < RADIX ? n - i * RADIX : RADIX;
            int64_t sum;
            CK(sum = fwgpu_add_node(g_cx, FWGPU_SUM, 2 * (uint32_t)ports, 2, NULL, 0));
            for (int p = 0; p < ports; ++p)
                for (uint32_t ch = 0; ch < 2; ++ch) CK(fwgpu_connect(g_cx, level[i * RADIX + p], ch, sum, 2 * (uint32_t)p + ch, 0));
            level[i] = sum;
        }
        n = m;
    }
    for (uint32_t ch = 0; ch < 2; ++ch) CK(fwgpu_connect(g_cx, level[0], ch, fwgpu_graph_out_node(g_cx), ch, 0));
    const double tb0 = now_us();
    CK(fwgpu_update(g_cx));
    const double first_update_us = now_us() - tb0;
    const int plan = fwgpu_plan_kind(g_cx);
    for (int v = 0; v < g_voices; ++v) start_voice(&g_voice[v]);

    Audio a;
    memset(&a, 0, sizeof(a));
    a.cap = steady + (long)edits * 400 + 4000;
    a.t_us = (double*)malloc(sizeof(double) * (size_t)a.cap);
    a.tag = (unsigned char*)malloc((size_t)a.cap);
    a.phase = (unsigned char*)calloc((size_t)a.cap, 1);
    pthread_t th;
    pthread_create(&th, NULL, audio_main, &a);
    while (a.n < steady) { /* phase 1: nobody edits */
        struct timespec ts = {0, 2000000};
        nanosleep(&ts, NULL);
    }
    /* phase 2: one voice after another is replaced while the callbacks go on */
    double upd_sum = 0, upd_max = 0;
    double* upd_all = (double*)malloc(sizeof(double) * (size_t)(edits > 0 ? edits : 1));
    for (int e = 0; e < edits; ++e) {
        Voice* v = &g_voice[(e * 977 + 13) % g_voices];
        a.editing = 1;
        CK(fwgpu_remove_node(g_cx, v->s));
        CK(fwgpu_remove_node(g_cx, v->bq));
        CK(fwgpu_remove_node(g_cx, v->dl));
        CK(fwgpu_remove_node(g_cx, v->vol));
        make_voice(v, g_voices + e);
        const double t0 = now_us();
        CK(fwgpu_update(g_cx));
        const double dt = now_us() - t0;
        upd_sum += dt;
        upd_all[e] = dt;
        if (dt > upd_max) upd_max = dt;
        a.editing = 2; /* built and published: the next callback adopts the plan, the one after it starts the new voice */
        start_voice(v);
        { /* the callbacks that pick the plan up are still "editing" ones: wait for two more before the flag drops */
            const long seen = a.n;
            while (a.n < seen + 2 && a.n < a.cap) {
                struct timespec ts = {0, 100000};
                nanosleep(&ts, NULL);
            }
        }
        a.editing = 0;
        struct timespec gap = {0, 3000000};
        nanosleep(&gap, NULL);
    }
    a.stop = 1;
    pthread_join(th, NULL);
    double m0, p0, x0, m1, p1, x1;
    long c0, c1;
    stats(a.t_us, a.tag, a.n, 0, &m0, &p0, &x0, &c0);
    stats(a.t_us, a.tag, a.n, 1, &m1, &p1, &x1, &c1);
    double m2, p2, x2;
    long c2;
    stats(a.t_us, a.tag, a.n, 2, &m2, &p2, &x2, &c2);
    { /* diagnostics on stderr: the slow callbacks among those that began while an edit was in flight — which edit, which update phase */
        int edit = -1;
        for (long i = 1; i < a.n; ++i) {
            if (a.tag[i] == 1 && a.tag[i - 1] != 1) ++edit;
            if (i >= 50 && a.tag[i] == 1 && a.t_us[i] > p0 + 15.0)
                fprintf(stderr, "slow callback %ld while edit %d was built: %.1f us, update phase %d (the one before it: %.1f us, tag %d)\n", i, edit, a.t_us[i], a.phase[i],
                        a.t_us[i - 1], a.tag[i - 1]);
        }
    }
    for (int ph = 0; ph <= 40; ++ph) { /* diagnostics on stderr: the callbacks by what the updating thread was doing when they began */
        double md, p9, mx;
        long cn;
        stats(a.t_us, a.phase, a.n, ph, &md, &p9, &mx, &cn);
        if (cn) fprintf(stderr, "update phase %d: %ld callbacks, median %.1f p99 %.1f max %.1f us\n", ph, cn, md, p9, mx);
    }
    /* (the first edit builds into the second plan image — its device allocations, tens of milliseconds on some boxes — and pulls the
     * mean: the median is what an edit costs) */
    double upd_median = 0;
    if (edits > 0) {
        qsort(upd_all, (size_t)edits, sizeof(double), cmp_d);
        upd_median = upd_all[edits / 2];
    }
    uint64_t adoptions = 0, by_audio = 0, worst_ns = 0;
    CK(fwgpu_plan_handover_stats(g_cx, &adoptions, &by_audio, &worst_ns));
    printf("{\"voices\": %d, \"block\": %u, \"launch_plan\": %d, \"callbacks\": %ld, \"edits\": %d, \"callback_period_us\": %.0f, \"first_update_ms\": %.2f, "
           "\"update_ms_mean\": %.3f, \"update_ms_median\": %.3f, \"update_ms_max\": %.3f, "
           "\"callback_us_steady\": {\"n\": %ld, \"median\": %.1f, \"p99\": %.1f, \"max\": %.1f}, "
           "\"callback_us_while_the_plan_is_built\": {\"n\": %ld, \"median\": %.1f, \"p99\": %.1f, \"max\": %.1f}, "
           "\"callback_us_adoption_and_the_two_after\": {\"n\": %ld, \"median\": %.1f, \"p99\": %.1f, \"max\": %.1f}, "
           "\"plans_adopted\": %llu, \"adopted_by_a_callback\": %llu, \"longest_adoption_us\": %.1f}\n",
           g_voices, g_block, plan, a.n, edits, g_period_us, first_update_us / 1e3, edits ? upd_sum / edits / 1e3 : 0.0, upd_median / 1e3, upd_max / 1e3, c0, m0, p0, x0, c1, m1, p1,
           x1, c2, m2, p2, x2, (unsigned long long)adoptions, (unsigned long long)by_audio, worst_ns / 1e3);
    fwgpu_ctx_destroy(g_cx);
    return 0;
}
