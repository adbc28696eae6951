// TEST DRIVER, CPU tier only (tests/test_realtime_contract.py builds and runs it against the host-only harness: fake HIP
// runtime + launch stubs, no audio computed).  It plays a host application with an audio thread and control threads:
//
//   rt_driver tsan   (built -fsanitize=thread)  two control threads send gain / pan / sampler messages for their own
//                    nodes while the audio thread runs one-block callbacks: the message ring, the drain epoch, the
//                    error buffers and the return ring must be free of data races, and no message may be lost.
//   rt_driver edits  (built -fsanitize=thread, or -DCOUNT_ALLOCS)  an EDITOR thread adds voice chains into spare mixer ports,
//                    updates, starts them, removes them again, updates — while the audio thread runs callbacks and another
//                    control thread sends messages to the standing voices: the plan hand-over (fwgpu_update builds off to
//                    the side, the next callback adopts: graph/processor.rs:167-206) must be free of data races, every
//                    callback must succeed, and adopting must not touch the host allocator on the audio thread.
//   rt_driver alloc  (built -DCOUNT_ALLOCS: malloc / calloc / realloc / free of the whole process forwarded to glibc
//                    and counted per thread)  10 000 steady callbacks, then 10 000 more with a control thread sending
//                    messages: the audio thread's host-heap allocation count must not move (SURVEY 8(b) realtime rules).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include <algorithm>
#include <chrono>

#include "../../include/fwgpu.h"
#include "../../firewheel_amd/csrc/fwgpu_ctx.h"  // the "gate" mode reads two counters of the context (a test may look inside)

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
    int64_t root = -1;
    int sample = -1;
    std::vector<int64_t> samplers, volumes, pans;
};

static Bank build_bank(int voices, int block, int spare_ports = 0) {
    Bank b;
    b.c = fwgpu_ctx_create(0, 48000, (uint32_t)block, 0, 2, nullptr);
    CHECK(b.c);
    CHECK(fwgpu_set_max_batch(b.c, 16) == 0);
    std::vector<float> src(2 * 4096, 0.25f);
    const int smp = fwgpu_sample_create(b.c, FWGPU_PLANAR_F32, 2, 4096, src.data());
    CHECK(smp >= 0);
    const int64_t root = fwgpu_add_node(b.c, FWGPU_SUM, 2 * (uint32_t)(voices + spare_ports), 2, nullptr, 0);
    CHECK(root >= 0);
    b.root = root;
    b.sample = smp;
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

// rt_driver gate  — ADVICE r4 (medium): a control thread that announced itself at ControlGate (gate_ctl_waiting raised) and was
//                  descheduled before it took the gate must not hold up the audio thread for more than the bounded deference.
static int gate_mode() {
    const int block = 64;
    Bank b = build_bank(8, block);
    std::vector<float> out((size_t)block * 2);
    CHECK(fwgpu_process_interleaved(b.c, nullptr, out.data(), 0, 2, (uint64_t)block, 0.0, 0) == 0);  // warm
    fwgpu_ctx* c = b.c;
    auto us_of_a_call = [&]() {
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(fwgpu_process_interleaved(b.c, nullptr, out.data(), 0, 2, (uint64_t)block, 0.0, 0) == 0);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    double base = 1e9;
    for (int i = 0; i < 200; ++i) base = std::min(base, us_of_a_call());
    // the waiter that never comes: the counter stays raised across 200 calls
    c->gate_ctl_waiting.fetch_add(1);
    double worst = 0.0, sum = 0.0;
    for (int i = 0; i < 200; ++i) {
        const double t = us_of_a_call();
        worst = std::max(worst, t);
        sum += t;
    }
    c->gate_ctl_waiting.fetch_sub(1);
    const unsigned long long expired = c->gate_defer_expired.load();
    printf("gate: bare call %.1f us; with a waiter that never comes: mean %.1f us, worst %.1f us, deference expired %llu times (bound %llu us)\n",
           base, sum / 200, worst, expired, (unsigned long long)(c->gate_defer_ns / 1000));
    CHECK(expired == 200);
    CHECK(sum / 200 < base + 2.0 * (double)(c->gate_defer_ns / 1000) + 50.0);  // bounded: not a scheduler quantum
    // and a waiter that DOES come still goes first: a control call beside back-to-back callbacks finishes promptly
    std::atomic<bool> stop{false};
    std::thread audio([&] {
        while (!stop.load(std::memory_order_relaxed)) CHECK(fwgpu_process_interleaved(b.c, nullptr, out.data(), 0, 2, (uint64_t)block, 0.0, 0) == 0);
    });
    std::vector<float> data(2 * 256, 0.25f);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 40; ++i) CHECK(fwgpu_sample_create(b.c, FWGPU_PLANAR_F32, 2, 256, data.data()) >= 0);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    stop = true;
    audio.join();
    printf("gate: 40 sample_create calls beside back-to-back callbacks: %.1f ms\n", ms);
    CHECK(ms < 2000.0);
    printf("gate-run ok\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "gate")) return gate_mode();
    const bool tsan = argc > 1 && !strcmp(argv[1], "tsan");
    const bool edits = argc > 1 && !strcmp(argv[1], "edits");
    const int block = 64, voices = 24, spare = edits ? 6 : 0;
    Bank b = build_bank(voices, block, spare);
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
    // The control side of a ctx is one thread at a time: message calls and graph calls may each overlap PROCESS calls, not one
    // another (include/fwgpu.h).  Two control threads of the "edits" run share this lock, as a host with two would.
    std::mutex ctl;
    auto control = [&](int first, int last, int rounds) {
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        for (int r = 0; r < rounds && !stop.load(std::memory_order_relaxed); ++r)
            for (int v = first; v < last; ++v) {
                std::unique_lock<std::mutex> lk(ctl, std::defer_lock);
                if (edits) lk.lock();
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

    if (edits) {
        // the editor: a voice chain into spare port p, update, start it; two rounds later remove it, update
        std::atomic<unsigned long long> updates{0};
        auto editor = [&](int rounds) {
            struct Live {
                int64_t s = -1, g = -1, p = -1;
                int sample = -1;
            } live[8];
            std::vector<int> to_destroy;  // samples of retired voices: destroyed once the device has let go of them
            std::vector<float> data(2 * 512, 0.125f);
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            for (int r = 0; r < rounds; ++r) {
                std::lock_guard<std::mutex> lk(ctl);
                const int port = r % spare;
                Live& l = live[port];
                if (l.s >= 0) {  // retire the chain that sits there
                    CHECK(fwgpu_remove_node(b.c, l.s) == 0);
                    CHECK(fwgpu_remove_node(b.c, l.g) == 0);
                    CHECK(fwgpu_remove_node(b.c, l.p) == 0);
                    to_destroy.push_back(l.sample);
                    l = Live();
                    CHECK(fwgpu_update(b.c) == 0);
                    updates++;
                }
                {  // ReturnSample (sampler.rs:339-343,563-571): a sample nobody holds any more may go — while callbacks run
                    int64_t nodes[8];
                    int samples[8];
                    (void)fwgpu_poll_returned_samples(b.c, nodes, samples, 8);
                    size_t w = 0;
                    for (int smp : to_destroy) {
                        if (fwgpu_sample_retired(b.c, smp) == 1) CHECK(fwgpu_sample_destroy(b.c, smp) == 0);
                        else to_destroy[w++] = smp;
                    }
                    to_destroy.resize(w);
                }
                l.sample = fwgpu_sample_create(b.c, FWGPU_PLANAR_F32, 2, 512, data.data());  // a new sample table entry, mid-stream
                CHECK(l.sample >= 0);
                float p100 = 100.f, pg = 30.f + (float)(r % 50), pp = (float)(r % 21) / 10.f - 1.f;
                l.s = fwgpu_add_node(b.c, FWGPU_SAMPLER, 0, 2, &p100, 1);
                l.g = fwgpu_add_node(b.c, FWGPU_VOLUME, 2, 2, &pg, 1);
                l.p = fwgpu_add_node(b.c, FWGPU_STEREO_PAN, 2, 2, &pp, 1);
                CHECK(l.s >= 0 && l.g >= 0 && l.p >= 0);
                for (uint32_t ch = 0; ch < 2; ++ch) {
                    CHECK(fwgpu_connect(b.c, l.s, ch, l.g, ch, 0) >= 0);
                    CHECK(fwgpu_connect(b.c, l.g, ch, l.p, ch, 0) >= 0);
                    CHECK(fwgpu_connect(b.c, l.p, ch, b.root, 2 * (uint32_t)(voices + port) + ch, 0) >= 0);
                }
                CHECK(fwgpu_sampler_set_sample(b.c, l.s, l.sample, 0, 0) == 0);  // before the update: waits for the plan that activates it
                CHECK(fwgpu_update(b.c) == 0);
                updates++;
                CHECK(fwgpu_plan_kind(b.c) == 1);
                CHECK(fwgpu_sampler_set_loop_range(b.c, l.s, 1, 0.0, 0.0, 0) == 0);
                CHECK(fwgpu_sampler_play(b.c, l.s, 1) == 0);
                CHECK(fwgpu_node_set_param(b.c, l.g, 0, 60.f, 2) == 0);
            }
        };
        std::thread te(editor, 300), tm(control, 0, voices, 300);
        const unsigned long long a0 = thread_allocs();
        go.store(true, std::memory_order_release);
        unsigned long long n_cb = 0;
        while (updates.load() < 590 || n_cb < 3000) {  // (300 rounds: 300 adds + 294 removals)
            callback();
            n_cb++;
            CHECK(n_cb < 4000000ull);
        }
        const unsigned long long a1 = thread_allocs();
        te.join();
        tm.join();
        for (int i = 0; i < 8; ++i) callback();
        uint64_t adoptions = 0, by_audio = 0, worst_ns = 0;
        CHECK(fwgpu_plan_handover_stats(b.c, &adoptions, &by_audio, &worst_ns) == 0);
        printf("edits-run: %llu callbacks, %llu updates, %llu adoptions (%llu by a callback, longest %.1f us); audio-thread allocations %llu\n", n_cb,
               updates.load(), (unsigned long long)adoptions, (unsigned long long)by_audio, worst_ns / 1e3, a1 - a0);
        CHECK(adoptions >= 2 && adoptions <= updates.load() + 1 && by_audio >= 1);
        CHECK(refused.load() == 0);
#ifdef COUNT_ALLOCS
        CHECK(a1 - a0 == 0);
#endif
        printf("edits-run ok\n");
    } else if (tsan) {
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
