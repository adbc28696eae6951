// TEST DRIVER, CPU tier only (tests/test_realtime_contract.py builds and runs it against the host-only harness: fake HIP
// runtime + launch stubs, no audio computed).  It plays a host application with an audio thread and control threads:
//
//   rt_driver tsan   (built -fsanitize=thread)  two control threads send gain / pan / sampler messages for their own
//                    nodes while the audio thread runs one-block callbacks: the message ring, the drain epoch, the
//                    error buffers and the return ring must be free of data races, and no message may be lost.
//   rt_driver alloc  (built -DCOUNT_ALLOCS: malloc / calloc / realloc / free of the whole process forwarded to glibc
//                    and counted per thread)  10 000 steady callbacks, then 10 000 more with a control thread sending
//                    messages: the audio thread's host-heap allocation count must not move (SURVEY 8(b) realtime rules).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/fwgpu.h"

#ifdef COUNT_ALLOCS
extern "C" {
void* __libc_malloc(size_t);
void* __libc_calloc(size_t, size_t);
void* __libc_realloc(void*, size_t);
void __libc_free(void*);
}
static thread_local unsigned long long t_allocs = 0;
extern "C" void* malloc(size_t n) {
    t_allocs++;
    return __libc_malloc(n);
}
extern "C" void* calloc(size_t a, size_t b) {
    t_allocs++;
    return __libc_calloc(a, b);
}
extern "C" void* realloc(void* p, size_t n) {
    t_allocs++;
    return __libc_realloc(p, n);
}
extern "C" void free(void* p) { __libc_free(p); }
static unsigned long long thread_allocs() { return t_allocs; }
#else
static unsigned long long thread_allocs() { return 0; }
#endif

extern "C" unsigned long long fwh_launch_count(int which);
extern "C" unsigned long long fwh_alloc_count(void);
extern "C" unsigned long long fwh_cmds_seen(void);

#define CHECK(x)                                                               \
    do {                                                                       \
        if (!(x)) {                                                            \
            fprintf(stderr, "rt_driver: CHECK failed: %s (line %d)\n", #x, __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

struct Bank {
    fwgpu_ctx* c;
    std::vector<int64_t> samplers, volumes, pans;
};

static Bank build_bank(int voices, int block) {
    Bank b;
    b.c = fwgpu_ctx_create(0, 48000, (uint32_t)block, 0, 2, nullptr);
    CHECK(b.c);
    CHECK(fwgpu_set_max_batch(b.c, 16) == 0);
    std::vector<float> src(2 * 4096, 0.25f);
    const int smp = fwgpu_sample_create(b.c, FWGPU_PLANAR_F32, 2, 4096, src.data());
    CHECK(smp >= 0);
    const int64_t root = fwgpu_add_node(b.c, FWGPU_SUM, 2 * (uint32_t)voices, 2, nullptr, 0);
    CHECK(root >= 0);
    for (int v = 0; v < voices; ++v) {
        float p100 = 100.f, p50 = 50.f, p0 = 0.f;
        int64_t s = fwgpu_add_node(b.c, FWGPU_SAMPLER, 0, 2, &p100, 1);
        int64_t g = fwgpu_add_node(b.c, FWGPU_VOLUME, 2, 2, &p50, 1);
        int64_t p = fwgpu_add_node(b.c, FWGPU_STEREO_PAN, 2, 2, &p0, 1);
        CHECK(s >= 0 && g >= 0 && p >= 0);
        for (uint32_t ch = 0; ch < 2; ++ch) {
            CHECK(fwgpu_connect(b.c, s, ch, g, ch, 0) >= 0);
            CHECK(fwgpu_connect(b.c, g, ch, p, ch, 0) >= 0);
            CHECK(fwgpu_connect(b.c, p, ch, root, 2 * (uint32_t)v + ch, 0) >= 0);
        }
        b.samplers.push_back(s);
        b.volumes.push_back(g);
        b.pans.push_back(p);
    }
    for (uint32_t ch = 0; ch < 2; ++ch) CHECK(fwgpu_connect(b.c, root, ch, fwgpu_graph_out_node(b.c), ch, 0) >= 0);
    CHECK(fwgpu_update(b.c) == 0);
    CHECK(fwgpu_plan_kind(b.c) == 1);
    for (int64_t s : b.samplers) {
        CHECK(fwgpu_sampler_set_sample(b.c, s, smp, 0, 0) == 0);
        CHECK(fwgpu_sampler_set_loop_range(b.c, s, 1, 0.0, 0.0, 0) == 0);
        CHECK(fwgpu_sampler_play(b.c, s, 0) == 0);
    }
    return b;
}

int main(int argc, char** argv) {
    const bool tsan = argc > 1 && !strcmp(argv[1], "tsan");
    const int block = 64, voices = 24;
    Bank b = build_bank(voices, block);
    std::vector<float> out((size_t)block * 2);
    fwgpu_stream* st = fwgpu_stream_open(b.c, 0, 2);
    CHECK(st);
    double now = 0.0;
    auto callback = [&]() {
        now += block / 48000.0;
        CHECK(fwgpu_stream_callback(st, out.data(), (uint64_t)block, now) >= 0);
    };
    for (int i = 0; i < 50; ++i) callback();  // warm: every buffer has its size

    std::atomic<bool> go{false}, stop{false};
    std::atomic<unsigned long long> sent{0}, refused{0};
    auto control = [&](int first, int last, int rounds) {
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        for (int r = 0; r < rounds && !stop.load(std::memory_order_relaxed); ++r)
            for (int v = first; v < last; ++v) {
                int rc = fwgpu_node_set_param(b.c, b.volumes[v], 0, 20.f + (float)((r + v) % 70), (uint32_t)(r % 3));
                rc == 0 ? sent++ : refused++;
                rc = fwgpu_node_set_param(b.c, b.pans[v], 0, (float)((r * 7 + v) % 21) / 10.f - 1.f, 0);  // two messages
                rc == 0 ? (sent += 2) : refused++;
                if (r % 16 == 5) {
                    rc = (r & 16) ? fwgpu_sampler_pause(b.c, b.samplers[v], 1) : fwgpu_sampler_play(b.c, b.samplers[v], 0);
                    rc == 0 ? sent++ : refused++;
                }
                if ((r & 63) == 63) std::this_thread::yield();
            }
    };

    if (tsan) {
        const unsigned long long seen0 = fwh_cmds_seen();
        std::thread t1(control, 0, voices / 2, 400), t2(control, voices / 2, voices, 400);
        go.store(true, std::memory_order_release);
        for (int i = 0; i < 3000; ++i) {
            callback();
            if (i % 50 == 0) {  // the control side of the return path, from the audio thread's sibling: here the main thread
                int64_t nodes[8];
                int samples[8];
                (void)fwgpu_poll_returned_samples(b.c, nodes, samples, 8);
            }
        }
        t1.join();
        t2.join();
        for (int i = 0; i < 8; ++i) callback();  // everything sent has reached its block (at_block <= 2)
        CHECK(refused.load() == 0);
        // no message lost, none duplicated: the control-kernel stub counts the messages whose block falls into each launch
        const unsigned long long seen = fwh_cmds_seen() - seen0;
        CHECK(seen == sent.load());
        printf("tsan-run ok: %llu messages sent, %llu applied\n", sent.load(), seen);
    } else {
        const unsigned long long dev0 = fwh_alloc_count();
        std::thread t1(control, 0, voices, 250);  // (creating a thread allocates, on this thread: before the snapshots; it waits for `go`)
        const unsigned long long a0 = thread_allocs();
        for (int i = 0; i < 10000; ++i) callback();
        const unsigned long long a1 = thread_allocs();
        go.store(true, std::memory_order_release);
        for (int i = 0; i < 10000; ++i) callback();
        const unsigned long long a2 = thread_allocs();
        stop.store(true);
        t1.join();
        // a failing call must not allocate either: fixed error buffer (an invalid channel count is refused up front)
        CHECK(fwgpu_process_interleaved(b.c, nullptr, out.data(), 0, 65, 1, 0.0, 0) < 0);
        CHECK(strstr(fwgpu_last_error(b.c), "64 stream channels") != nullptr);
        const unsigned long long a3 = thread_allocs();
        printf("alloc-run: steady %llu, with messages %llu, failing call %llu, device/pinned %llu (sent %llu)\n", a1 - a0, a2 - a1,
               a3 - a2, fwh_alloc_count() - dev0, sent.load());
#ifdef COUNT_ALLOCS
        CHECK(a1 - a0 == 0);
        CHECK(a2 - a1 == 0);
        CHECK(a3 - a2 == 0);
#endif
        CHECK(fwh_alloc_count() - dev0 == 0);
        CHECK(sent.load() > 0);
        printf("alloc-run ok\n");
    }
    uint64_t cbs = 0, under = 0;
    CHECK(fwgpu_stream_stats(st, &cbs, &under, nullptr) == 0);
    CHECK(under == 0 && cbs > 0);
    fwgpu_stream_close(st);
    fwgpu_ctx_destroy(b.c);
    return 0;
}
