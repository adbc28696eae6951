// fwgpu_plan_install.cpp — node activation (graph.rs:594-612), the launch plan's device tables, and the hand-over of a new
// plan to the audio side (graph/context.rs:93-137 + graph/processor.rs:167-206).
//
//   build_image   (control thread, may take milliseconds, may allocate, may fail) works out everything a plan needs in a
//                 PlanImage that no process call can see: tables, pool, voice descriptors, staging areas — and the list of
//                 things adopting it must do to the state that outlives plans (initial states of new nodes, larger arrays).
//   publish       hands the image over: adopted at once when no process call is in flight, else left for the next one.
//   adopt_image   (whoever holds the gate; on the audio thread: a member swap + a handful of asynchronous launches) makes
//                 the image the active one and sends the old one back.
// The control thread never writes into a buffer a running plan reads.
#include "fwgpu_ctx.h"

#include <chrono>
#include <thread>

namespace fwgpu {

// control-side upload of the image being built: its own stream (never the null stream: that would serialise with a caller's
// legacy default stream), complete on return (the sources are locals of the build)
// (round 3: staged through ONE pinned arena and copied asynchronously — a table is a memcpy into pinned memory plus a short, truly
//  asynchronous HIP call; the ~30 pageable copies with a blocking sync each that a build used to make held the runtime's locks long
//  enough to hold up the audio thread's launch calls: callbacks while editing p99 214 us -> see profiles/r03_edit_race_cfg3.json.
//  build_image waits for the stream once, at its end.)
// (round 3, late: what reaches the callbacks while a plan is built is the DEVICE side of these copies and fills — blit / fill kernels
//  on the build's stream; while one runs, the audio stream's kernels finish late (a 1-thread k_signal_done took 45-65 us next to a 50 us
//  copy in the rocprofv3 trace; fw_edit_race's phase tags: nothing during the graph compile, 2-3x callbacks from the first upload on).
//  Fewer CUs for the build's stream made it worse (p99 250 -> 325 us: the harm grows with how LONG the build's kernels run beside
//  the audio ones), cutting the copies into pieces that still ran back to back changed nothing.  What helps is TIME: the build's GPU
//  work goes out in pieces of a few microseconds, each issued when no process call is in flight (the gate word says so) and waited
//  for before the next — a callback meets at most the piece that was issued just before it began.  With the audio side saturated
//  (callbacks back to back, the stress of fw_edit_race) a piece goes out anyway after quiet_wait_us.  With no stream live — no process
//  call in the last 200 ms: single-threaded hosts, set-up — everything goes out whole as before.
//  A PACED stream (gaps between the calls of at least 2 x QUIET_MARGIN) is also looked at ahead: a group launched just before a
//  call begins still meets it, so when the next call is due within the margin — the last period and the last start say when — the
//  group waits for that call to come and go.)
// a group and its launch (FWGPU_QUIET_MARGIN_US: experiments)
static const uint64_t QUIET_MARGIN_NS = getenv("FWGPU_QUIET_MARGIN_US") ? (uint64_t)atoll(getenv("FWGPU_QUIET_MARGIN_US")) * 1000ull : 60000ull;
void quiet_window(fwgpu_ctx* c) {
    if (!c->quiet_wait_us) return;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto deadline = t0 + std::chrono::microseconds(3 * (uint64_t)c->quiet_wait_us);
    {  // (a paced stream: room for one whole call to come and go)
        const uint64_t period = c->cb_period_ns.load(std::memory_order_relaxed), dur = c->cb_dur_ns.load(std::memory_order_relaxed);
        if (period && period <= 200000000ull && period >= dur + 2 * QUIET_MARGIN_NS) deadline = std::max(deadline, t0 + std::chrono::nanoseconds(3 * dur));
    }
    auto wait_gate = [&](bool busy, clk::time_point until) {  // spin while (gate == 1) == busy; false: timed out
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                if ((c->gate.load(std::memory_order_acquire) == 1) != busy) return true;
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
            }
            if (clk::now() > until) return false;
        }
    };
    for (int round = 0; round < 2; ++round) {
        const uint64_t period = c->cb_period_ns.load(std::memory_order_relaxed), dur = c->cb_dur_ns.load(std::memory_order_relaxed);
        const uint64_t start = c->cb_start_ns.load(std::memory_order_relaxed);
        // A call in flight: wait for its end — quiet_wait_us for a stream without gaps (the group then goes out anyway: it queues
        // behind the call's kernels), but a PACED stream's call (gaps of two margins and more) is waited out: it ends within its usual
        // length, and a group launched from this thread while the audio thread is still enqueuing its kernels lands BETWEEN them
        // (round 6: one callback of +40-60 us per paced run once a build had shrunk to a single group; scripts/r06_edit_paced_ab.sh)
        const bool paced = period && period <= 200000000ull && period >= dur + 2 * QUIET_MARGIN_NS;
        const auto in_flight_until = paced ? std::max(t0 + std::chrono::microseconds(c->quiet_wait_us), t0 + std::chrono::nanoseconds(dur + dur / 2)) :
                                             t0 + std::chrono::microseconds(c->quiet_wait_us);
        if (c->gate.load(std::memory_order_acquire) == 1 && !wait_gate(true, round ? deadline : in_flight_until)) return;
        // no call in flight.  Is the next one about to begin?
        if (round || !quiet_next_call_is_due((uint64_t)clk::now().time_since_epoch().count(), start, period, dur, QUIET_MARGIN_NS)) return;
        if (!wait_gate(false, deadline)) return;  // ... wait for it to begin, then (second round) to end
    }
}
// ---- the build's device work as ONE kernel (round 3, late).  fw_edit_race's phase tags showed that what a callback pays is per
// OPERATION on the build's stream, not per byte: every copy / fill / small kernel that lands beside a callback costs it ~25-40 us
// whatever its size (its dispatch and its end-of-kernel cache write-back / invalidate reach the audio kernels' L2 lines), and a
// build made ~25 of them.  Now up() / zero() / the strided clears only record JOBS; the changed chunks wait in the pinned arena;
// build_apply() puts the job list into pinned memory too and launches k_build_apply ONCE — a grid of (blocks, jobs) that copies from pinned
// host memory over PCIe and fills — in a window with no process call in flight.  (FWGPU_BUILD_ONE_KERNEL=0: the calls go out one by
// one, in pieces while a stream is live, as before.)
static int build_apply(fwgpu_ctx* c) {
    if (c->build_jobs.empty()) return 0;
    // While a stream is live the jobs go out in GROUPS of a few microseconds each (about up_piece bytes of copy, 32x that of fill),
    // every group launched in a window with no process call in flight and waited for before the next: few operations AND short
    // ones (one 100 us kernel met a paced stream's callback in one build of six).  Jobs larger than a group are cut first.
    const bool live = audio_live(c);
    const size_t cp = c->up_piece, fp = (size_t)c->up_piece * 32;
    std::vector<BuildJob> jobs;
    jobs.reserve(c->build_jobs.size() + 16);
    for (const BuildJob& j : c->build_jobs) {
        if (!live) {
            jobs.push_back(j);
        } else if (j.src) {
            for (size_t off = 0; off < j.row_bytes; off += cp) {
                BuildJob t = j;
                t.dst = (char*)j.dst + off;
                t.src = (const char*)j.src + off;
                t.row_bytes = t.pitch = std::min(cp, (size_t)j.row_bytes - off);
                jobs.push_back(t);
            }
        } else if (j.rows == 1) {
            for (size_t off = 0; off < j.row_bytes; off += fp) {
                BuildJob t = j;
                t.dst = (char*)j.dst + off;
                t.row_bytes = t.pitch = std::min(fp, (size_t)j.row_bytes - off);
                if (off) t.head = -1;
                jobs.push_back(t);
            }
        } else {
            const uint32_t per = (uint32_t)std::max<size_t>(1, fp / (size_t)j.row_bytes);
            for (uint32_t r = 0; r < j.rows; r += per) {
                BuildJob t = j;
                t.dst = (char*)j.dst + (size_t)r * j.pitch;
                t.rows = std::min(per, j.rows - r);
                jobs.push_back(t);
            }
        }
    }
    c->build_jobs.clear();
    // the list travels in pinned memory of its own (the kernel reads it over PCIe), grown on this — the control — thread
    const size_t jb = jobs.size() * sizeof(BuildJob);
    if (jb > c->h_jobs_cap) {
        if (c->h_jobs) {
            HIPC(c, hipStreamSynchronize(c->up_stream));  // (a launch of an earlier flush may still read the old list)
            (void)hipHostFree(c->h_jobs);
        }
        c->h_jobs = nullptr;
        c->h_jobs_cap = 0;
        const size_t cap = std::max<size_t>(2 * jb, (size_t)64 << 10);
        HIPC(c, hipHostMalloc((void**)&c->h_jobs, cap, hipHostMallocDefault));
        c->h_jobs_cap = cap;
        c->h_jobs_used = 0;
    }
    if (c->h_jobs_used + jb > c->h_jobs_cap) {  // (earlier flushes of this build used the front: wait for them, start over)
        HIPC(c, hipStreamSynchronize(c->up_stream));
        c->h_jobs_used = 0;
    }
    BuildJob* at = (BuildJob*)(c->h_jobs + c->h_jobs_used);
    memcpy(at, jobs.data(), jb);
    c->h_jobs_used += jb;
    size_t i = 0;
    while (i < jobs.size()) {
        size_t k = i, cost = 0;
        while (k < jobs.size()) {
            const size_t jc = jobs[k].src ? (size_t)jobs[k].row_bytes : (size_t)jobs[k].row_bytes * jobs[k].rows / 32;
            if (live && k > i && cost + jc > cp) break;
            if (k - i >= 32768) break;  // (jobs are the grid's y dimension)
            cost += jc;
            ++k;
        }
        if (c->update_prof) {
            c->prof_groups++;
            c->prof_jobs += k - i;
            for (size_t q = i; q < k; ++q) (jobs[q].src ? c->prof_copy_bytes : c->prof_fill_bytes) += (uint64_t)jobs[q].row_bytes * jobs[q].rows;
        }
        if (live) quiet_window(c);
        // (a caller-supplied stream, or one under FWGPU_RT_GRAPH capture, is not ours to launch into from the control thread: a
        //  launch from another thread invalidates a thread-local capture — ADVICE r4: those contexts use the build's own stream)
        if (c->build_on_audio_stream && c->own_stream && !c->rt_use_graph) {
            // Round 4 (measured, scripts/r04_session1.sh, config 3's 4 096 voices, callbacks back to back): the same groups in the
            // AUDIO stream — the jobs only touch the image nobody reads, HIP streams take launches from two threads — cost a
            // callback their own few microseconds: p99 80-82 us while a plan is built against 59 steady (+21-23), maximum 87-88
            // against 70-75; on the build's own stream the same run had p99 99-126 and maxima of 130-148 — the fixed ~27 us a
            // callback paid per operation on the other stream was the price of a second hardware queue waking beside the audio
            // one (round 3's open question).  Waited for through an event (the stream itself never idles).
            if (!c->ev_build) HIPC(c, hipEventCreateWithFlags(&c->ev_build, hipEventDisableTiming));
            LCHK(c, launch_build_apply(c->stream, at + i, (int)(k - i)));
            HIPC(c, hipEventRecord(c->ev_build, c->stream));
            HIPC(c, hipEventSynchronize(c->ev_build));
        } else {
            LCHK(c, launch_build_apply(c->up_stream, at + i, (int)(k - i)));
            if (live) HIPC(c, hipStreamSynchronize(c->up_stream));
        }
        i = k;
    }
    return 0;
}
// room in the pinned arena for `need` more bytes; what is pending goes out first if there is none
static int arena_room(fwgpu_ctx* c, size_t need) {
    if (c->h_up && c->h_up_used + need <= c->h_up_cap) return 0;
    int rc = build_apply(c);
    if (rc) return rc;
    HIPC(c, hipStreamSynchronize(c->up_stream));  // what is in flight has left the arena
    c->h_up_used = 0;
    if (!c->h_up || need > c->h_up_cap) {
        if (c->h_up) (void)hipHostFree(c->h_up);
        c->h_up = nullptr;
        c->h_up_cap = 0;
        const size_t cap = std::max<size_t>(need, (size_t)8 << 20);
        HIPC(c, hipHostMalloc((void**)&c->h_up, cap, hipHostMallocDefault));
        c->h_up_cap = cap;
    }
    return 0;
}
// `device_writes`: the kernels write this table too (silence flags): always uploaded whole, no shadow.
// Every other table is read-only on the device (const in DevView / FusedView): the image keeps a host copy of what it last
// uploaded and a build copies only the 4 KiB chunks that differ — an edit of one voice of config 3's 4 096 leaves 89 % of the
// table bytes as the image's previous build left them (the two images alternate, so "previous" is two edits ago).
static int up_impl(fwgpu_ctx* c, const char* tag, DevBuf& b, const void* src, size_t bytes, bool device_writes = false) {
    HIPC(c, b.ensure_n(tag, bytes));
    if (!bytes) return 0;
    if (c->build_one_kernel && c->up_diff && !device_writes && b.shadow.size() == bytes) {
        // The usual case of an edit (round 6): compare FIRST, then copy only the chunks that differ — into the pinned arena and into
        // the shadow.  (Copying the whole table into the arena, comparing, and assigning the whole shadow were three passes over
        // 1.1 MB of tables per update of config 3's graph: most of the "node tables" phase was memcpy.)
        constexpr size_t CH = 4096;
        const char* s8 = (const char*)src;
        uint8_t* sh = b.shadow.data();
        size_t off = 0, sent = 0;
        auto same = [&](size_t o) { return !memcmp(sh + o, s8 + o, std::min(CH, bytes - o)); };
        while (off < bytes) {
            while (off < bytes && same(off)) off += CH;
            if (off >= bytes) break;
            size_t end = off;
            while (end < bytes && !same(end)) end += CH;
            end = std::min(end, bytes);
            const size_t len = end - off, room = (len + 255) & ~(size_t)255;
            int rc = arena_room(c, room);
            if (rc) return rc;
            char* at = c->h_up + c->h_up_used;
            memcpy(at, s8 + off, len);
            memcpy(sh + off, s8 + off, len);
            BuildJob j{};
            j.dst = (char*)b.p + off;
            j.src = at;
            j.row_bytes = len;
            j.pitch = len;
            j.rows = 1;
            c->build_jobs.push_back(j);
            c->h_up_used += room;
            sent += len;
            off = end;
        }
        if (c->update_prof_tables) fprintf(stderr, "fwgpu up %-22s %9zu bytes, %9zu sent\n", tag, bytes, sent);
        return 0;
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    int rc = arena_room(c, need);
    if (rc) return rc;
    memcpy(c->h_up + c->h_up_used, src, bytes);
    const bool one = c->build_one_kernel;
    const bool live = !one && audio_live(c);
    const size_t piece = live ? c->up_piece : bytes;
    const bool diff = c->up_diff && !device_writes;
    constexpr size_t CH = 4096;
    // runs of chunks that differ from the shadow
    size_t off = 0, sent = 0;
    while (off < bytes) {
        size_t end = bytes;
        if (diff) {
            auto same = [&](size_t o) {
                const size_t n = std::min(CH, bytes - o);
                return o + n <= b.shadow.size() && !memcmp(b.shadow.data() + o, (const char*)src + o, n);
            };
            while (off < bytes && same(off)) off += CH;
            if (off >= bytes) break;
            end = off;
            while (end < bytes && !same(end)) end += CH;
            end = std::min(end, bytes);
        }
        sent += end - off;
        if (one) {
            BuildJob j{};
            j.dst = (char*)b.p + off;
            j.src = c->h_up + c->h_up_used + off;
            j.row_bytes = end - off;
            j.pitch = j.row_bytes;
            j.rows = 1;
            c->build_jobs.push_back(j);
            off = end;
            continue;
        }
        for (; off < end; off += piece) {  // one by one, in pieces while a stream is live
            const size_t n = std::min(piece, end - off);
            if (live) quiet_window(c);
            HIPC(c, hipMemcpyAsync((char*)b.p + off, c->h_up + c->h_up_used + off, n, hipMemcpyHostToDevice, c->up_stream));
            if (live) HIPC(c, hipStreamSynchronize(c->up_stream));
        }
        off = end;
    }
    if (diff) b.shadow.assign((const uint8_t*)src, (const uint8_t*)src + bytes);
    else b.shadow.clear();
    c->h_up_used += need;
    if (c->update_prof_tables) fprintf(stderr, "fwgpu up %-22s %9zu bytes, %9zu sent\n", tag, bytes, sent);
    return 0;
}
#define up(c, b, ...) up_impl(c, #b, b, __VA_ARGS__)  // (the table's name: FWGPU_UPDATE_PROF=2 lists what every build sends, table by table)
// rows x row_bytes at a pitch, every byte `value`; head >= 0: byte 0 of every row is `head` instead
static int fill_rows(fwgpu_ctx* c, void* p, size_t row_bytes, size_t pitch, size_t rows, int value, int head = -1) {
    if (!row_bytes || !rows) return 0;
    if (c->build_one_kernel) {
        int rc = arena_room(c, 0);
        if (rc) return rc;
        BuildJob j{};
        j.dst = p;
        j.src = nullptr;
        j.row_bytes = row_bytes;
        j.pitch = pitch;
        j.rows = (uint32_t)rows;
        j.value = (uint32_t)(value & 0xff);
        j.head = head;
        c->build_jobs.push_back(j);
        return 0;
    }
    const bool live = audio_live(c);
    if (rows == 1 || pitch == row_bytes) {
        const size_t bytes = rows * row_bytes;
        const size_t piece = live ? (size_t)c->up_piece * 32 : bytes;  // (a fill runs ~30x faster than a copy over PCIe)
        for (size_t off = 0; off < bytes; off += piece) {
            if (live) quiet_window(c);
            HIPC(c, hipMemsetAsync((char*)p + off, value, std::min(piece, bytes - off), c->up_stream));
            if (live) HIPC(c, hipStreamSynchronize(c->up_stream));
        }
    } else {
        if (live) quiet_window(c);
        LCHK(c, launch_zero_rows(c->up_stream, (float*)p, pitch / sizeof(float), (int)(row_bytes / sizeof(float)), (int)rows));
    }
    if (head >= 0) {
        if (live) quiet_window(c);
        LCHK(c, launch_set_row_heads(c->up_stream, (uint8_t*)p, pitch, (int)rows, (uint8_t)head));
    }
    return 0;
}
static int zero(fwgpu_ctx* c, void* p, size_t bytes) { return fill_rows(c, p, bytes, bytes, 1, 0); }

void PlanImage::release_device() {
    DevBuf* bufs[] = {&d_nodes, &d_in_buf, &d_out_buf, &d_level_nodes, &d_pool, &d_flags, &d_gin_bufs, &d_gout_bufs, &d_groups, &d_blks2, &d_refs2,
                      &d_gsets2, &d_ramps2, &d_progs, &d_hist, &d_rs_wl, &d_rs_tmpl, &d_lazy, &d_ctl_order, &d_slot_voice, &d_voices, &d_leaves, &d_blks, &d_refs, &d_gsets, &d_cache, &d_ramps, &d_bus, &d_bus_flags,
                      &d_chain_start, &d_chain_dummy, &d_chain_stats, &d_up_nodes, &d_up_in, &d_up_out, &d_up_level_nodes, &d_root_bufs, &d_tail_nodes,
                      &d_tail_in, &d_tail_out, &d_tail_idx, &d_tail_frozen, &d_frozen_ph, &d_frozen, &d_chain_done, &d_rt_tree, &d_rt_tree_sync, &d_fir_rows, &d_fir_tiles, &d_fir_partials,
                      &d_hlevel_nodes, &grow_states, &grow_ext, &d_state_inits, &d_ext_jobs};
    for (DevBuf* b : bufs) b->release();
    if (h_ctl_order) (void)hipHostFree(h_ctl_order);
    h_ctl_order = nullptr;
    h_ctl_order_cap = 0;
    if (h_host_stage) (void)hipHostFree(h_host_stage);
    if (h_host_flags) (void)hipHostFree(h_host_flags);
    h_host_stage = d_host_stage = nullptr;
    h_host_flags = d_host_flags = nullptr;
    host_stage_floats = host_flag_bytes = 0;
    if (rt_graph.exec) (void)hipGraphExecDestroy(rt_graph.exec);
    rt_graph.exec = nullptr;
    if (retired_ev) (void)hipEventDestroy(retired_ev);
    retired_ev = nullptr;
}

// a recycled image keeps its buffers (capacity) and its pinned staging area; everything that DESCRIBES a plan starts over
static void reset_for_build(fwgpu_ctx* c, PlanImage& P) {
    c->spare_plan = std::move(P.plan);  // its node array goes to the next fwgpu_update (capacity, not contents)
    P.plan = Plan();
    P.have_plan = false;
    P.level_off.clear();
    P.level_cnt.clear();
    P.level_kinds.clear();
    P.n_gout_bufs = P.n_gin_bufs = 0;
    P.slot_index.clear();
    P.slot_voice.clear();
    P.ctl_order_live = false;
    P.fused = P.fused_fx = P.ctl_ahead_on = P.fused_rs = P.fused_prog = P.fused_sp = P.hybrid = P.hybrid_fx = P.lazy_capable = false;
    P.generic_k = 1;
    P.chain_nq = 1;
    P.n_voices = P.n_leaves = P.ramp_slots = P.n_groups = P.n_tail = P.n_fused_real = 0;
    P.rt_tree_leaves = P.rt_tree_up = 0;
    P.n_bus = 1;
    P.up_level_off.clear();
    P.up_level_cnt.clear();
    P.tail_kinds.clear();
    P.up_root_node = -1;
    memset(&P.root_args, 0, sizeof(P.root_args));
    P.fir_groups.clear();
    P.hlevel_off.clear();
    P.hlevel_cnt.clear();
    P.hlevel_kinds.clear();
    P.host_levels.clear();
    P.n_host_nodes = 0;
    P.host_callbacks = 0;
    if (P.rt_graph.exec) (void)hipGraphExecDestroy(P.rt_graph.exec);
    P.rt_graph = PlanImage::RtGraph();
    P.grow_states.release();
    P.grow_ext.release();
    P.grow_states_cap = P.grow_ext_cap = 0;
    P.n_state_inits = P.n_ext_jobs = 0;
    P.ir_convs.clear();
    P.activated.clear();
    P.dropped_samplers.clear();
    P.removed_slots.clear();
    P.slots_cap = 0;
    P.grow_cur_sample.clear();
    P.grow_slot_ids.clear();
}

// what the control kernel writes and the render kernels read, per voice and block of a batch (both fused plans and the hybrid one)
// node state slot -> voice of the voice-bank plan (upload_cmds: which voices a call's messages go to)
static void build_slot_voice(PlanImage& P, const std::vector<VoiceDesc>& voices) {
    int max_slot = -1;
    auto each = [&](const VoiceDesc& vd, auto&& f) {
        f(vd.sampler_state);
        for (int j = 0; j < vd.n_stages && j < FW_MAX_STAGES - 1; ++j) f(vd.stage_state[j]);
        f(vd.bq_state);
        f(vd.bq2_state);
        f(vd.dl_state);
    };
    for (const VoiceDesc& vd : voices) each(vd, [&](int s) { max_slot = std::max(max_slot, s); });
    P.slot_voice.assign((size_t)(max_slot + 1), -1);
    for (size_t v = 0; v < voices.size(); ++v)
        each(voices[v], [&](int s) {
            if (s >= 0) P.slot_voice[(size_t)s] = (int)v;
        });
}
static int alloc_voice_tables(fwgpu_ctx* c, PlanImage& P) {
    const size_t K = P.kmax;
    HIPC(c, P.d_blks.ensure_n("d_blks", K * P.n_voices * sizeof(VoiceBlk)));
    HIPC(c, P.d_refs.ensure_n("d_refs", ref_count(P.n_voices, K) * sizeof(VoiceRef)));
    HIPC(c, P.d_gsets.ensure_n("d_gsets", (size_t)P.n_voices * FW_GSETS * sizeof(GainSet)));
    HIPC(c, P.d_chain_start.ensure_n("d_chain_start", (size_t)P.n_voices * sizeof(ChainStart)));
    HIPC(c, P.d_chain_dummy.ensure_n("d_chain_dummy", 64 * 1024));
    HIPC(c, P.d_chain_stats.ensure_n("d_chain_stats", 2 * sizeof(unsigned long long)));
    int rc;
    if ((rc = zero(c, P.d_chain_stats.p, 2 * sizeof(unsigned long long)))) return rc;
    if ((rc = zero(c, P.d_chain_start.p, (size_t)P.n_voices * sizeof(ChainStart)))) return rc;
    {
        // a cache row counts only when its epoch stamp is the running one, and adoption bumps the epoch: the rows a RECYCLED image
        // still holds carry older stamps and are dead as they are.  Fresh memory can hold anything: cleared once.
        const void* before = P.d_cache.p;
        const size_t cap0 = P.d_cache.cap;
        HIPC(c, P.d_cache.ensure_n("d_cache", (size_t)P.n_voices * sizeof(VoiceCache)));
        if ((P.d_cache.p != before || P.d_cache.cap != cap0) && (rc = zero(c, P.d_cache.p, P.d_cache.cap))) return rc;
    }
    HIPC(c, P.d_ramps.ensure_n("d_ramps", K * P.n_voices * (size_t)P.ramp_slots * c->stride * sizeof(float)));
    if (!P.h_ctl_order || P.h_ctl_order_cap < (size_t)P.n_voices) {  // (a recycled image keeps its pinned block)
        if (P.h_ctl_order) (void)hipHostFree(P.h_ctl_order);
        P.h_ctl_order = nullptr;
        P.h_ctl_order_cap = 0;
        const size_t cap = std::max<size_t>(64, (size_t)P.n_voices + (size_t)P.n_voices / 4);
        HIPC(c, hipHostMalloc((void**)&P.h_ctl_order, cap * sizeof(int), hipHostMallocDefault));
        P.h_ctl_order_cap = cap;
    }
    HIPC(c, P.d_ctl_order.ensure_n("d_ctl_order", std::max<size_t>(1, (size_t)P.n_voices) * sizeof(int)));
    P.ctl_mark.assign((size_t)P.n_voices, 0);
    P.hot_prev.clear();
    P.hot_now.clear();
    P.hot_prev.reserve((size_t)P.n_voices);
    P.hot_now.reserve((size_t)P.n_voices);
    P.ctl_order_live = false;
    // lazy records: plain voice-bank plans only (the leaf kernel's lazy instantiation knows samplers and gain / program stages)
    // (P.lazy_capable is set by the caller: the voice-bank branch of install_plan, never the hybrid one)
    if (P.lazy_capable && P.n_voices > 0) {
        HIPC(c, P.d_lazy.ensure_n("d_lazy", (size_t)P.n_voices * sizeof(LazyRec)));
        if ((rc = fill_rows(c, P.d_lazy.p, (size_t)P.n_voices * sizeof(LazyRec), (size_t)P.n_voices * sizeof(LazyRec), 1, 0xff))) return rc;  // mode = -1 everywhere
    }
    if (P.fused_rs) {  // one item per (leaf, block, 256-frame piece) at most
        HIPC(c, P.d_rs_wl.ensure_n("d_rs_wl", (2 + 2 * (size_t)P.n_leaves * K * LEAF_WPB_MAX) * sizeof(unsigned int)));
        if ((rc = zero(c, P.d_rs_wl.p, 2 * sizeof(unsigned int)))) return rc;
        // (two copies for the control kernel a call ahead, a third for the lazy records' templates)
        HIPC(c, P.d_rs_tmpl.ensure_n("d_rs_tmpl", 3 * std::max<size_t>(1, (size_t)P.n_voices) * sizeof(VoiceBlk)));
    }
    return 0;
}

// k_chain workgroups: consecutive leaves packed greedily into groups of <= 32 voices / <= 8 leaves (the voices of
// consecutive leaves are consecutive), so that a tree of small leaves fills the 32 voice rows
static int upload_chain_groups(fwgpu_ctx* c, PlanImage& P, const std::vector<LeafDesc>& leaves) {
    int rc;
    std::vector<ChainGroup> groups;
    for (size_t l = 0; l < leaves.size(); ++l) {
        const LeafDesc& ld = leaves[l];
        if (groups.empty() || groups.back().n_voices + ld.ports > 32 || groups.back().n_leaves >= CH_GROUP_LEAVES) {
            ChainGroup g;
            memset(&g, 0, sizeof(g));
            g.first_voice = ld.first_voice;
            groups.push_back(g);
        }
        ChainGroup& g = groups.back();
        const int li = g.n_leaves++;
        g.out_buf[li] = ld.out_buf;
        g.row0[li] = g.n_voices;
        g.ports[li] = ld.ports;
        g.start_mask |= 1u << g.n_voices;
        const int path_ports = ld.pad ? ld.pad : ld.ports;  // (a leaf that leads a wider SumNode takes that node's path)
        if (!(path_ports == 2 || path_ports == 3 || path_ports == 4))  // sum.rs:67-133 (Q13): the n-port path skips silent ports
            g.masked_rows |= (ld.ports >= 32 ? 0xffffffffu : ((1u << ld.ports) - 1u)) << g.n_voices;
        g.n_voices += ld.ports;
    }
    for (ChainGroup& g : groups) {
        const int P = g.ports[0];
        bool uni = g.n_voices == 32 && (P == 32 || P == 16 || P == 8 || P == 4);
        for (int i = 0; i < g.n_leaves && uni; ++i) uni = g.ports[i] == P;
        g.uniform_ports = uni ? P : 0;
    }
    P.n_groups = (int)groups.size();
    if ((rc = up(c, P.d_groups, groups.data(), groups.size() * sizeof(ChainGroup)))) return rc;
    return 0;
}

// Everything a plan needs, worked out on the control thread in an image no process call can see.  Nothing the audio side reads
// is written; the control side's own bookkeeping (activated flags, init records, ext_used, ir_cache) is committed only at the
// very end, when nothing can fail any more — a failure anywhere leaves the ctx exactly as it was and the next fwgpu_update
// tries the same nodes again (the reference keeps its schedule when a compile fails, context.rs:115-131).
static int build_image(fwgpu_ctx* c, Plan& plan, PlanImage& P) {
    reset_for_build(c, P);
    c->build_jobs.clear();
    c->h_up_used = 0;
    c->h_jobs_used = 0;
    P.kmax = c->kmax_req;
    P.gen = ++c->build_gen;
    phase_mark(c, 21);
    // 1. node state capacity (persists across recompiles: processor.rs:19,195-197): a larger array is allocated here and
    //    swapped in — old contents copied over on the ctx stream — when the image is adopted
    {
        const size_t need = c->graph.nodes.size();
        if (need > c->ctl_states_cap) {
            const size_t cap = std::max<size_t>(need * 2, 4096);
            HIPC(c, P.grow_states.ensure_n("grow_states", cap * sizeof(NodeState)));
            int rc0 = zero(c, P.grow_states.p, cap * sizeof(NodeState));
            if (rc0) return rc0;
            P.grow_states_cap = cap;
        }
        P.slots_cap = std::max<size_t>(need, 1);
        if (c->cur_sample.size() < need) {  // (control side reads the size only; the audio side swaps the vectors in at adoption)
            const size_t cap = std::max<size_t>(need * 2, 4096);
            P.grow_cur_sample.assign(cap, -1);
            P.grow_slot_ids.assign(cap, -1);
        }
    }
    phase_mark(c, 22);
    // 2. activate new nodes (graph.rs:594-612): their initial states and ext-pool slices are worked out here and applied at
    //    adoption (scatter kernels on the ctx stream), never written into live buffers from this thread
    struct Act {
        uint32_t slot;
        NodeState st;  // the node's init record with its ext slice / FIR ring geometry filled in
    };
    std::vector<Act> acts;
    std::vector<StateInitHost> inits;
    std::map<std::pair<int, int>, uint32_t> new_ir;                // impulse responses to convert to f32 -> ext offset
    size_t ext_need = c->ext_used;
    std::map<size_t, std::vector<uint32_t>> free_backup = c->ext_free;  // restored on failure
    struct Rollback {  // any return before `armed = false` puts the free lists back
        fwgpu_ctx* c;
        std::map<size_t, std::vector<uint32_t>>* backup;
        uint64_t* gen;
        bool armed = true;
        ~Rollback() {
            if (armed) {
                c->ext_free = *backup;
                --*gen;
            }
        }
    } rollback{c, &free_backup, &c->build_gen};
    auto take_ext = [&](size_t len, uint32_t* off) -> bool {  // recycled slice of exactly this (64-rounded) size, else bump
        const size_t rounded = (len + 63) / 64 * 64;
        auto it = c->ext_free.find(rounded);
        if (it != c->ext_free.end() && !it->second.empty()) {
            *off = it->second.back();
            it->second.pop_back();
            return true;
        }
        if (ext_need + rounded > 0xffffffffull) return false;
        *off = (uint32_t)ext_need;
        ext_need += rounded;
        return true;
    };
    std::vector<AdoptExtJobHost> ext_jobs;  // per new node with an ext slice: recycled slices zeroed, head floats (coefficients) set
    for (uint32_t slot : c->graph.nodes_to_activate) {
        const HostNode& n = c->graph.nodes[slot];
        if (!n.alive || n.activated) continue;
        if (n.kind == K_HOST && (slot >= c->host_procs.size() || !c->host_procs[slot].fn))
            return fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, "host node without a process function (fwgpu_host_node_set_process)");
        Act a;
        a.slot = slot;
        a.st = n.init;
        uint32_t nch = n.n_in < n.n_out ? n.n_in : n.n_out;
        size_t len = 0;
        std::vector<float> head;
        if (n.kind == K_BIQUAD) {
            len = 5 + 4 * (size_t)nch;
            head.resize(5);
            biquad_coefs(n.init.enabled, n.init.p0, n.init.p1, c->sample_rate, head.data());
        } else if (n.kind == K_DELAY) {
            len = (size_t)nch * (size_t)n.init.loop_end;
        } else if (n.kind == K_SPATIAL) {
            len = SP_HIST;
        } else if (n.kind == K_FIR) {
            int ir = n.init.sample;
            if (ir < 0 || ir >= (int)c->samples.size() || !c->samples[ir].alive)
                return fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, "FIR node: impulse-response sample was destroyed");
            uint64_t T = c->samples[ir].desc.frames;
            if (T == 0 || T > (1u << 24)) return fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, "FIR node: 1 <= taps <= 2^24");
            uint64_t R = T - 1 + (uint64_t)P.kmax * c->mbf;  // every block of a K-batch finds its whole window in the ring
            a.st.loop_start = T;
            a.st.loop_end = R;
            a.st.playhead = 0;
            len = (size_t)nch * 2 * (size_t)R;
            for (uint32_t ch = 0; ch < nch; ++ch) {
                auto key = std::make_pair(ir, (int)std::min<uint32_t>(ch, (uint32_t)c->samples[ir].desc.channels - 1));
                if (!c->ir_cache.count(key)) new_ir.emplace(key, 0u);  // offset assigned below, once the pool layout is final
            }
        }
        if (len) {
            const size_t before = ext_need;
            uint32_t off = 0;
            if (!take_ext(len, &off)) return fail(c, FWGPU_ERR_INVALID, "ext state pool exceeds 2^32 floats");
            a.st.ext_off = off;
            a.st.ext_len = (uint32_t)len;
            AdoptExtJobHost j;
            memset(&j, 0, sizeof(j));
            j.off = off;
            j.zero_len = ext_need == before ? (uint32_t)((len + 63) / 64 * 64) : 0u;  // (a bumped slice is zero already)
            j.n_head = (uint32_t)std::min<size_t>(head.size(), 8);
            for (uint32_t k = 0; k < j.n_head; ++k) j.head[k] = head[k];
            if (j.zero_len || j.n_head) ext_jobs.push_back(j);
        }
        StateInitHost si;
        si.index = (int)slot;
        si.pad = 0;
        si.st = a.st;
        inits.push_back(si);
        acts.push_back(a);
    }
    for (auto& kv : new_ir) {  // one f32 copy of each impulse-response channel
        uint64_t T = c->samples[kv.first.first].desc.frames;
        if (ext_need + (T + 63) / 64 * 64 > 0xffffffffull) return fail(c, FWGPU_ERR_INVALID, "ext state pool exceeds 2^32 floats");
        kv.second = (uint32_t)ext_need;
        ext_need += (T + 63) / 64 * 64;
        PlanImage::IrConv ic;
        ic.sample = kv.first.first;
        ic.ch = kv.first.second;
        ic.off = kv.second;
        ic.T = (uint32_t)T;
        P.ir_convs.push_back(ic);
    }
    int rc;
    if (ext_need > c->ctl_ext_cap) {
        const size_t cap = std::max<size_t>(ext_need * 2, 4096);
        HIPC(c, P.grow_ext.ensure_n("grow_ext", (cap + 256) * sizeof(float)));  // slack: vector loads may overhang the last slice
        if ((rc = zero(c, P.grow_ext.p, (cap + 256) * sizeof(float)))) return rc;
        P.grow_ext_cap = cap;
    }
    if (!ext_jobs.empty()) {
        if ((rc = up(c, P.d_ext_jobs, ext_jobs.data(), ext_jobs.size() * sizeof(AdoptExtJobHost)))) return rc;
        P.n_ext_jobs = (int)ext_jobs.size();
    }
    if (!inits.empty()) {
        if ((rc = up(c, P.d_state_inits, inits.data(), inits.size() * sizeof(StateInitHost)))) return rc;
        P.n_state_inits = (int)inits.size();
    }
    // what the tables below need from nodes this image activates (their init records are committed only at the end)
    auto node_init = [&](uint32_t slot) -> NodeState {
        for (const Act& a : acts)
            if (a.slot == slot) return a.st;
        return c->graph.nodes[slot].init;
    };
    auto ir_off = [&](const std::pair<int, int>& key) -> uint32_t {
        auto it = new_ir.find(key);
        return it != new_ir.end() ? it->second : c->ir_cache[key];
    };
    // (the launch plans are chosen before the tables are written: a hybrid plan with split SumNodes adds partial buses to the
    // pool and continuation nodes to the node table)
    FusedBuild fb, hb;
    const bool is_fused = !c->force_generic && detect_fused(plan, c->graph, c->mbf, fb);
    const bool is_hybrid = !is_fused && !c->force_generic && detect_hybrid(plan, c->graph, c->mbf, hb);
    if (is_hybrid)
        for (const FusedBuild::Split& sp : hb.splits) {  // a partial bus (two pool buffers) per split SumNode
            hb.leaves[sp.leaf].out_buf = plan.num_buffers;
            plan.num_buffers += 2;
        }
    phase_mark(c, 23);
    // 3. node tables
    const int N = (int)plan.nodes.size();
    std::vector<NodeDesc> nd(N);
    std::vector<int> in_tab, out_tab;
    std::vector<std::vector<int>> levels(plan.num_levels);
    std::vector<int> gin_bufs, gout_bufs, host_nodes;
    {  // (sized up front: growing these by doubling was a third of this step on config 3)
        size_t n_in = 0;
        std::vector<int> per_level(plan.num_levels, 0);
        for (const PlanNode& p : plan.nodes) {
            n_in += (size_t)p.n_in;
            if (!p.is_graph_io && p.level >= 0 && p.level < plan.num_levels) per_level[p.level]++;
        }
        in_tab.reserve(n_in + 64);
        out_tab.reserve((size_t)plan.num_buffers + 64);
        for (int l = 0; l < plan.num_levels; ++l) levels[l].reserve(per_level[l]);
    }
    for (int i = 0; i < N; ++i) {
        const PlanNode& p = plan.nodes[i];
        NodeDesc& d = nd[i];
        memset(&d, 0, sizeof(d));
        d.kind = p.kind;
        d.n_in = p.n_in;
        d.n_out = p.n_out;
        d.in_off = (int)in_tab.size();
        d.out_off = (int)out_tab.size();
        d.state = (int)p.slot;
        d.aux0 = (p.kind == K_SUM && p.n_out > 0) ? p.n_in / p.n_out : 0;
        d.is_graph_io = p.is_graph_io;
        in_tab.insert(in_tab.end(), p.in_buf.begin(), p.in_buf.end());
        out_tab.insert(out_tab.end(), p.out_buf.begin(), p.out_buf.end());
        if (p.is_graph_io == 1) gin_bufs = p.out_buf;
        else if (p.is_graph_io == 2) gout_bufs = p.in_buf;
        else if (p.kind == K_HOST) host_nodes.push_back(i);  // no kernel runs it: the plan is cut at its level (step 3c)
        else levels[p.level].push_back(i);
    }
    // vertical fusion of the level executor (k_generic.hip.h fz_links): a stereo sampler or gain-like node whose two output buffers
    // are read by exactly ONE node — a 2 -> 2 volume / pan / width / hard clip, channel for channel — names it in aux0 (+ 1; aux0 is
    // the port count of SumNodes only).  Whether a batch uses the link is decided on the device (both nodes frozen, block not silent).
    {
        std::vector<int> cnt((size_t)plan.num_buffers, 0), who((size_t)plan.num_buffers, -1), port((size_t)plan.num_buffers, -1);
        for (int i = 0; i < N; ++i) {
            const PlanNode& p = plan.nodes[i];
            for (int q = 0; q < p.n_in; ++q) {
                const int b = p.in_buf[q];
                if (b <= 0 || b >= plan.num_buffers) continue;
                cnt[b]++;
                who[b] = i;
                port[b] = q;
            }
        }
        auto gainlike = [](int k) { return k == K_VOLUME || k == K_PAN || k == K_WIDTH || k == K_HARD_CLIP; };
        for (int i = 0; i < N; ++i) {
            const PlanNode& p = plan.nodes[i];
            if (p.is_graph_io || p.n_out != 2) continue;
            if (!((p.kind == K_SAMPLER && p.n_in == 0) || (gainlike(p.kind) && p.n_in == 2))) continue;
            const int b0 = p.out_buf[0], b1 = p.out_buf[1];
            if (b0 <= 0 || b1 <= 0 || cnt[b0] != 1 || cnt[b1] != 1 || who[b0] != who[b1] || port[b0] != 0 || port[b1] != 1) continue;
            const PlanNode& q = plan.nodes[who[b0]];
            if (q.is_graph_io || !gainlike(q.kind) || q.n_in != 2 || q.n_out != 2) continue;
            nd[i].aux0 = who[b0] + 1;
        }
    }
    // hybrid plan: the continuation of a split SumNode — (partial bus, the ports behind the leading voices) on the path of the
    // node's full port count — as an extra entry behind the plan's nodes; the hybrid level lists name it instead of the node
    std::vector<int> split_entry(N, -1);
    if (is_hybrid)
        for (const FusedBuild::Split& sp : hb.splits) {
            const PlanNode& p = plan.nodes[sp.sum];
            NodeDesc d = nd[sp.sum];
            const int total = p.n_in / 2, rest = total - sp.lead;
            d.in_off = (int)in_tab.size();
            d.n_in = 2 * (1 + rest);
            d.aux0 = (1 + rest) | (total << 16);
            const int pb = hb.leaves[sp.leaf].out_buf;
            in_tab.push_back(pb);
            in_tab.push_back(pb + 1);
            in_tab.insert(in_tab.end(), p.in_buf.begin() + 2 * sp.lead, p.in_buf.end());
            split_entry[sp.sum] = (int)nd.size();
            nd.push_back(d);
        }
    if (in_tab.empty()) in_tab.push_back(0);
    if (out_tab.empty()) out_tab.push_back(0);
    if ((rc = up(c, P.d_nodes, nd.data(), nd.size() * sizeof(NodeDesc)))) return rc;
    if ((rc = up(c, P.d_in_buf, in_tab.data(), in_tab.size() * sizeof(int)))) return rc;
    if ((rc = up(c, P.d_out_buf, out_tab.data(), out_tab.size() * sizeof(int)))) return rc;
    std::vector<int> flat;
    P.level_off.clear();
    P.level_cnt.clear();
    P.level_kinds.clear();
    for (auto& l : levels) {
        P.level_off.push_back((int)flat.size());
        P.level_cnt.push_back((int)l.size());
        flat.insert(flat.end(), l.begin(), l.end());
        int kinds = 0;
        for (int i : l) kinds |= host_kind_bits(nd[i].kind);
        P.level_kinds.push_back(kinds);
    }
    if (flat.empty()) flat.push_back(0);
    if ((rc = up(c, P.d_level_nodes, flat.data(), flat.size() * sizeof(int)))) return rc;
    P.n_gin_bufs = (int)gin_bufs.size();
    P.n_gout_bufs = (int)gout_bufs.size();
    if (gin_bufs.empty()) gin_bufs.push_back(0);
    if (gout_bufs.empty()) gout_bufs.push_back(0);
    if ((rc = up(c, P.d_gin_bufs, gin_bufs.data(), gin_bufs.size() * sizeof(int)))) return rc;
    if ((rc = up(c, P.d_gout_bufs, gout_bufs.data(), gout_bufs.size() * sizeof(int)))) return rc;
    phase_mark(c, 24);
    // 3c. host nodes (K_HOST): per level, what the audio side needs to call them — and one pinned, device-mapped staging area
    //     for their inputs and outputs of a whole K-batch, allocated here (a process call never allocates)
    {
        P.host_levels.assign(plan.num_levels, {});
        P.n_host_nodes = (int)host_nodes.size();
        P.host_callbacks = 0;
        size_t floats = 0, flag_bytes = 0, max_in = 1, max_out = 1;
        for (int i : host_nodes) {
            const PlanNode& p = plan.nodes[i];
            PlanImage::HostCall hc;
            hc.node_idx = i;
            hc.n_in = p.n_in;
            hc.n_out = p.n_out;
            hc.in_off = nd[i].in_off;
            hc.out_off = nd[i].out_off;
            hc.fn = p.slot < c->host_procs.size() ? c->host_procs[p.slot].fn : nullptr;
            hc.user = p.slot < c->host_procs.size() ? c->host_procs[p.slot].user : nullptr;
            if (!hc.fn) return fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, "host node without a process function (fwgpu_host_node_set_process)");
            hc.stage_off = floats;
            hc.flag_off = flag_bytes;
            floats += (size_t)P.kmax * (size_t)(p.n_in + p.n_out) * c->stride;
            flag_bytes += (size_t)P.kmax * (size_t)(p.n_in + p.n_out);
            max_in = std::max<size_t>(max_in, (size_t)p.n_in);
            max_out = std::max<size_t>(max_out, (size_t)p.n_out);
            P.host_levels[p.level].push_back(hc);
        }
        if (floats > P.host_stage_floats || flag_bytes > P.host_flag_bytes) {  // (a recycled image keeps its staging area when it is large enough)
            if (P.h_host_stage) (void)hipHostFree(P.h_host_stage);
            if (P.h_host_flags) (void)hipHostFree(P.h_host_flags);
            P.h_host_stage = nullptr;
            P.h_host_flags = nullptr;
            P.host_stage_floats = P.host_flag_bytes = 0;
            void *hs = nullptr, *hf = nullptr, *ds = nullptr, *df = nullptr;
            HIPC(c, hipHostMalloc(&hs, floats * sizeof(float), hipHostMallocMapped));
            P.h_host_stage = (float*)hs;
            HIPC(c, hipHostMalloc(&hf, flag_bytes + 64, hipHostMallocMapped));
            P.h_host_flags = (uint8_t*)hf;
            HIPC(c, hipHostGetDevicePointer(&ds, hs, 0));
            HIPC(c, hipHostGetDevicePointer(&df, hf, 0));
            P.d_host_stage = (float*)ds;
            P.d_host_flags = (uint8_t*)df;
            P.host_stage_floats = floats;
            P.host_flag_bytes = flag_bytes;
        }
        P.host_in_ptrs.assign(max_in, nullptr);
        P.host_out_ptrs.assign(max_out, nullptr);
    }
    phase_mark(c, 25);
    // 3b. FIR banks: one GEMM per (level, impulse-response channel)
    {
        std::map<std::tuple<int, uint32_t, uint32_t>, std::vector<FirRow>> groups;
        for (int i = 0; i < N; ++i) {
            const PlanNode& p = plan.nodes[i];
            if (p.kind != K_FIR) continue;
            const NodeState hst = node_init(p.slot);
            int ir = hst.sample;
            uint32_t T = (uint32_t)hst.loop_start;
            int nch = std::min(p.n_in, p.n_out);
            for (int ch = 0; ch < nch; ++ch) {
                auto key = std::make_pair(ir, std::min(ch, c->samples[ir].desc.channels - 1));
                FirRow r;
                r.state = (int)p.slot;
                r.ch = ch;
                r.in_buf = p.in_buf[ch];
                r.out_buf = p.out_buf[ch];
                groups[std::make_tuple(p.level, ir_off(key), T)].push_back(r);
            }
        }
        // one launch per (level, T); inside it rows are sorted by impulse-response channel and padded so that
        // every 32-row tile convolves with a single h (tile_h_off)
        std::vector<FirRow> flat_rows;
        std::vector<uint32_t> flat_tiles;
        P.fir_groups.clear();
        size_t partial_need = 0;
        std::map<std::pair<int, uint32_t>, std::vector<std::pair<uint32_t, std::vector<FirRow>*>>> launches;
        for (auto& g : groups)
            launches[std::make_pair(std::get<0>(g.first), std::get<2>(g.first))].emplace_back(std::get<1>(g.first), &g.second);
        for (auto& l : launches) {
            PlanImage::FirGroup fg;
            fg.level = l.first.first;
            fg.T = l.first.second;
            fg.row_off = (int)flat_rows.size();
            fg.tile_off = (int)flat_tiles.size();
            for (auto& part : l.second) {
                for (const FirRow& r : *part.second) flat_rows.push_back(r);
                while ((flat_rows.size() - fg.row_off) % 32) {
                    FirRow pad;
                    pad.state = -1;
                    pad.ch = pad.in_buf = pad.out_buf = 0;
                    flat_rows.push_back(pad);
                }
                while (flat_tiles.size() - fg.tile_off < (flat_rows.size() - fg.row_off) / 32) flat_tiles.push_back(part.first);
            }
            fg.n_rows = (int)flat_rows.size() - fg.row_off;
            P.fir_groups.push_back(fg);
            size_t W = (size_t)fg.T - 1 + c->mbf;
            size_t segs = (W + FIR_SEG - 1) / FIR_SEG;
            partial_need = std::max(partial_need, segs * (size_t)fg.n_rows * (size_t)((c->mbf + 255) / 256 * 256) * P.kmax);
        }
        if (!flat_rows.empty()) {
            if ((rc = up(c, P.d_fir_rows, flat_rows.data(), flat_rows.size() * sizeof(FirRow)))) return rc;
            if ((rc = up(c, P.d_fir_tiles, flat_tiles.data(), flat_tiles.size() * sizeof(uint32_t)))) return rc;
            HIPC(c, P.d_fir_partials.ensure_n("d_fir_partials", partial_need * sizeof(float)));
        }
    }
    phase_mark(c, 26);
    // 4. buffer pool: a new schedule starts from zeroed buffers (schedule.rs:202-203); one slice per block of a
    //    generic K-batch.  generic_k: the FIR history rings were sized for the batch size in force when their node was
    //    activated — a later, larger kmax must not outrun them.
    P.generic_k = P.kmax;
    for (int i = 0; i < N; ++i) {
        if (plan.nodes[i].kind != K_FIR) continue;
        const NodeState hst = node_init(plan.nodes[i].slot);
        uint64_t room = (hst.loop_end - (hst.loop_start - 1)) / c->mbf;  // (R - (T-1)) / block
        P.generic_k = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(P.generic_k, room));
    }
    {
        const size_t Kg = P.generic_k;
        size_t pool_bytes = Kg * (size_t)plan.num_buffers * c->stride * sizeof(float);
        HIPC(c, P.d_pool.ensure_n("d_pool", pool_bytes));
        // buffer 0 of every block is the constant-zero buffer (flagged silent below) and has to BE zero; every other buffer is written
        // by the node that owns it before anybody reads it (the level order) — the GPU suite passes with FWGPU_POISON=1 filling
        // fresh pools with 0xCB.  So only those rows are cleared: the whole pool was a 0.5 GB fill per build on config 3, ~100 us of
        // fill kernel during which the callbacks that ran beside the build took 2-3x as long (fw_edit_race's phase tags).
        if (plan.num_buffers > 0 && (rc = fill_rows(c, P.d_pool.p, c->stride * sizeof(float), (size_t)plan.num_buffers * c->stride * sizeof(float), Kg, 0))) return rc;
        // silence flags (the kernels write them): all clear, buffer 0 — constant zero — always flagged silent.  Made on the device:
        // K x num_buffers bytes were the largest upload of an edit (0.4 of 1.5 MB on config 3)
        const size_t fl_bytes = Kg * (size_t)plan.num_buffers;
        HIPC(c, P.d_flags.ensure_n("d_flags", fl_bytes));
        if (plan.num_buffers > 0 && (rc = fill_rows(c, P.d_flags.p, (size_t)plan.num_buffers, (size_t)plan.num_buffers, Kg, 0, 1))) return rc;
    }

    phase_mark(c, 27);
    // 5. fused voice-bank plan
    P.fused = false;
    P.hybrid = false;
    P.fused_fx = false;
    if (is_fused) {
        P.fused_fx = fb.has_fx;
        // k_chain tile = 64*nq frames: the larger tile needs whole tiles per block and every delay >= one tile
        P.chain_nq = (c->mbf % 128 == 0 && fb.min_delay >= 128) ? 2 : 1;
        if (const char* e = getenv("FWGPU_CHAIN_NQ")) {  // experiments: force the smaller tile
            if (atoi(e) == 1) P.chain_nq = 1;
        }
        for (const VoiceDesc& vd : fb.voices) {  // an EQ cascade somewhere: the instantiation with the second recurrence stage (bit 2); a gain
            if (vd.bq2_state >= 0) P.chain_nq |= 4;  // stage between two filters or a hard clip: the one with the five-site stage logic (bit 3)
            if (vd.sampler_state >= 0 && vd.n_mid) P.chain_nq |= 8;
            for (int j = 0; j < vd.n_stages && vd.sampler_state >= 0; ++j)
                if (vd.stage_kind[j] == K_HARD_CLIP) P.chain_nq |= 8;
        }
        P.n_voices = (int)fb.voices.size();
        P.n_leaves = (int)fb.leaves.size();
        P.n_bus = fb.n_bus;
        P.ramp_slots = 2 * (1 + fb.max_stages);
        for (VoiceDesc& vd : fb.voices)  // the spatialiser's history slice in the ext pool (its node may be activated by this very plan)
            if (vd.sp_ext_off >= 0) vd.sp_ext_off = (int)node_init((uint32_t)vd.stage_state[vd.n_stages - 1]).ext_off;
        P.fused_sp = fb.has_sp;
        if (fb.has_sp) HIPC(c, P.d_hist.ensure_n("d_hist", fb.voices.size() * SP_HIST * sizeof(float)));
        build_slot_voice(P, fb.voices);
        if ((rc = up(c, P.d_slot_voice, P.slot_voice.data(), P.slot_voice.size() * sizeof(int)))) return rc;
        if ((rc = up(c, P.d_voices, fb.voices.data(), fb.voices.size() * sizeof(VoiceDesc)))) return rc;
        if ((rc = up(c, P.d_leaves, fb.leaves.data(), fb.leaves.size() * sizeof(LeafDesc)))) return rc;
        if ((rc = up(c, P.d_progs, fb.progs.data(), fb.progs.size() * sizeof(uint32_t)))) return rc;
        P.fused_prog = fb.has_prog;
        P.fused_rs = fb.has_rs;
        P.n_groups = 0;
        if (P.fused_fx) {
            if ((rc = upload_chain_groups(c, P, fb.leaves))) return rc;
        }
        const size_t K = P.kmax;
        // (round 6: chain plans too — k_chain derives its records from the LazyRecs — and resampler banks (k_leaf_rs); spatialiser banks
        //  keep their control kernel)
        P.lazy_capable = c->lazy_on && !fb.has_sp && !(fb.has_rs && fb.has_fx);
        if ((rc = alloc_voice_tables(c, P))) return rc;
        // (spatialiser stages: their 64-frame history goes from the LAST block of a call to the first block of the next through the ext
        //  pool; in this mode the copy into the call's scratch is made on the render stream — k_sp_hist_copy — because the control
        //  kernel runs beside the render kernel that writes it.  A race until the graph fuzz found it at seeds 91, 196, 384.)
        if (c->ctl_ahead && c->ctl_stream && !P.fused_fx && fb.tail_nodes.empty() && K > 1) {
            // the second copy of what the control kernel writes and the render kernels read
            bool ok = P.d_blks2.ensure_n("d_blks2", K * P.n_voices * sizeof(VoiceBlk)) == hipSuccess &&
                      P.d_refs2.ensure_n("d_refs2", ref_count(P.n_voices, K) * sizeof(VoiceRef)) == hipSuccess &&
                      P.d_gsets2.ensure_n("d_gsets2", (size_t)P.n_voices * FW_GSETS * sizeof(GainSet)) == hipSuccess &&
                      P.d_ramps2.ensure_n("d_ramps2", K * P.n_voices * (size_t)P.ramp_slots * c->stride * sizeof(float)) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            P.ctl_ahead_on = ok;
        }
        size_t bus_bytes = K * (size_t)P.n_bus * c->stride * sizeof(float);
        HIPC(c, P.d_bus.ensure_n("d_bus", bus_bytes));
        // bus 0 of every block is the constant-zero bus (flagged silent below) and has to BE zero; every other bus is written by
        // its leaf / sum / master-chain kernel before anything reads it, as in the pool (step 4): clearing all of it was a 4.4 MB
        // fill per edit of config 3 — most of what an edit put on the stream
        if (P.n_bus > 0 && (rc = fill_rows(c, P.d_bus.p, c->stride * sizeof(float), (size_t)P.n_bus * c->stride * sizeof(float), K, 0))) return rc;
        std::vector<uint8_t> bf(K * P.n_bus, 0);
        for (size_t k = 0; k < K; ++k) bf[k * P.n_bus] = 1;
        if ((rc = up(c, P.d_bus_flags, bf.data(), bf.size(), true))) return rc;  // (kernels write flags)
        if (fb.up_nodes.empty()) {
            NodeDesc z;
            memset(&z, 0, sizeof(z));
            fb.up_nodes.push_back(z);
        }
        if (fb.up_in.empty()) fb.up_in.push_back(0);
        if (fb.up_out.empty()) fb.up_out.push_back(0);
        if ((rc = up(c, P.d_up_nodes, fb.up_nodes.data(), fb.up_nodes.size() * sizeof(NodeDesc)))) return rc;
        if ((rc = up(c, P.d_up_in, fb.up_in.data(), fb.up_in.size() * sizeof(int)))) return rc;
        if ((rc = up(c, P.d_up_out, fb.up_out.data(), fb.up_out.size() * sizeof(int)))) return rc;
        std::vector<int> uflat;
        P.up_level_off.clear();
        P.up_level_cnt.clear();
        for (auto& l : fb.up_levels) {
            P.up_level_off.push_back((int)uflat.size());
            P.up_level_cnt.push_back((int)l.size());
            uflat.insert(uflat.end(), l.begin(), l.end());
        }
        P.up_root_node = (!fb.up_levels.empty() && fb.up_levels.back().size() == 1) ? fb.up_levels.back()[0] : -1;
        P.n_tail = (int)fb.tail_nodes.size();
        P.tail_kinds.clear();
        for (const NodeDesc& t : fb.tail_nodes) P.tail_kinds.push_back(host_kind_bits(t.kind));
        if (P.n_tail) {
            P.up_root_node = -1;  // the root's planar result feeds the master chain: no fused root + interleave
            std::vector<int> idx(P.n_tail);
            for (int i = 0; i < P.n_tail; ++i) idx[i] = i;
            if ((rc = up(c, P.d_tail_nodes, fb.tail_nodes.data(), fb.tail_nodes.size() * sizeof(NodeDesc)))) return rc;
            if ((rc = up(c, P.d_tail_in, fb.tail_in.data(), fb.tail_in.size() * sizeof(int)))) return rc;
            if ((rc = up(c, P.d_tail_out, fb.tail_out.data(), fb.tail_out.size() * sizeof(int)))) return rc;
            if ((rc = up(c, P.d_tail_idx, idx.data(), idx.size() * sizeof(int)))) return rc;
            HIPC(c, P.d_tail_frozen.ensure_n("d_tail_frozen", (size_t)P.n_tail * 16));  // (also a dummy playhead-snapshot area)
        }
        if (P.up_root_node >= 0) {
            const NodeDesc& rn = fb.up_nodes[P.up_root_node];
            if (rn.n_out == 2 && rn.n_in >= 2 && rn.n_in <= 64 && rn.n_in % 2 == 0) {
                memset(&P.root_args, 0, sizeof(P.root_args));
                P.root_args.n_in = rn.n_in;
                P.root_args.ports = rn.n_in / 2;
                for (int i = 0; i < rn.n_in; ++i) P.root_args.in_buf[i] = fb.up_in[rn.in_off + i];
                P.root_args.in_tab = P.d_up_in.as<int>() + rn.in_off;
            } else {
                P.up_root_node = -1;
            }
        }
        // the one-launch realtime kernels' way up the mixer tree (k_rt.hip.h): who reads each leaf's / upper node's bus, and how many
        // connected children each upper node waits for.  Only with a fused root (stereo, no master chain) and a tree in which every
        // bus has exactly one reader — which detect_fused guarantees; anything unexpected leaves the extents 0: launch sequence.
        P.rt_tree_leaves = P.rt_tree_up = 0;
        if (P.up_root_node >= 0 && !fb.leaves.empty() && !is_hybrid) {
            const int nl = (int)fb.leaves.size(), nu = (int)fb.up_nodes.size();
            std::vector<int> tree((size_t)nl + 2 * (size_t)nu, -1);
            int* parent_leaf = tree.data();
            int* parent_up = tree.data() + nl;
            int* kids = tree.data() + nl + nu;
            for (int u = 0; u < nu; ++u) kids[u] = 0;
            std::vector<int> who((size_t)P.n_bus, -1);  // bus -> leaf i (i) or upper node u (nl + u)
            for (int i = 0; i < nl; ++i)
                if (fb.leaves[i].out_buf > 0 && fb.leaves[i].out_buf < P.n_bus) who[fb.leaves[i].out_buf] = i;
            for (int u = 0; u < nu; ++u) {
                const int ob = fb.up_out[fb.up_nodes[u].out_off];
                if (ob > 0 && ob < P.n_bus) who[ob] = nl + u;
            }
            bool ok = true;
            for (int u = 0; u < nu && ok; ++u) {
                const NodeDesc& nd = fb.up_nodes[u];
                for (int p = 0; p < nd.n_in; p += 2) {
                    const int b = fb.up_in[nd.in_off + p];
                    if (b == 0) continue;  // an unconnected port: the constant-zero bus, nobody arrives for it
                    const int w = (b > 0 && b < P.n_bus) ? who[b] : -1;
                    if (w < 0) {
                        ok = false;
                        break;
                    }
                    int& par = w < nl ? parent_leaf[w] : parent_up[w - nl];
                    if (par != -1) ok = false;  // a bus with two readers
                    par = u;
                    kids[u]++;
                }
            }
            for (int i = 0; i < nl && ok; ++i) ok = parent_leaf[i] >= 0;
            for (int u = 0; u < nu && ok; ++u) ok = (u == P.up_root_node) ? parent_up[u] == -1 && kids[u] > 0 : parent_up[u] >= 0 && kids[u] > 0;
            if (ok) {
                if ((rc = up(c, P.d_rt_tree, tree.data(), tree.size() * sizeof(int)))) return rc;
                HIPC(c, P.d_rt_tree_sync.ensure_n("d_rt_tree_sync", (size_t)nu * sizeof(unsigned)));
                if ((rc = zero(c, P.d_rt_tree_sync.p, (size_t)nu * sizeof(unsigned)))) return rc;
                P.rt_tree_leaves = nl;
                P.rt_tree_up = nu;
            }
        }
        if (uflat.empty()) uflat.push_back(0);
        if ((rc = up(c, P.d_up_level_nodes, uflat.data(), uflat.size() * sizeof(int)))) return rc;
        if ((rc = up(c, P.d_root_bufs, fb.root_buf, sizeof(fb.root_buf)))) return rc;
        P.fused = true;
        P.n_fused_real = 0;
        for (const VoiceDesc& vd : fb.voices) P.n_fused_real += vd.sampler_state >= 0 ? 1 : 0;
    }
    phase_mark(c, 28);
    // 5b. hybrid plan: not a fused shape as a whole, but with voice banks inside that the fused kernels render
    // straight into their mixers' pool buffers; the level executor then runs the rest.
    P.hybrid_fx = false;
    if (is_hybrid) {
        P.n_voices = (int)hb.voices.size();
        P.n_leaves = (int)hb.leaves.size();
        P.ramp_slots = 2 * (1 + hb.max_stages);
        P.fused_prog = hb.has_prog;
        P.fused_rs = hb.has_rs;
        P.n_groups = 0;
        P.hybrid_fx = hb.has_fx;
        if (P.hybrid_fx) {  // the banks go through k_chain: its workgroups, its tile size, at most 64 blocks per launch
            if ((rc = upload_chain_groups(c, P, hb.leaves))) return rc;
            P.chain_nq = (c->mbf % 128 == 0 && hb.min_delay >= 128) ? 2 : 1;
            if (const char* e = getenv("FWGPU_CHAIN_NQ")) {
                if (atoi(e) == 1) P.chain_nq = 1;
            }
            for (const VoiceDesc& vd : hb.voices) {
                if (vd.bq2_state >= 0) P.chain_nq |= 4;
                if (vd.sampler_state >= 0 && vd.n_mid) P.chain_nq |= 8;
                for (int j = 0; j < vd.n_stages && vd.sampler_state >= 0; ++j)
                    if (vd.stage_kind[j] == K_HARD_CLIP) P.chain_nq |= 8;
            }
            P.generic_k = std::min<uint32_t>(P.generic_k, CH_FAST_KMAX);
        }
        P.n_tail = 0;
        P.up_root_node = -1;
        P.up_level_off.clear();
        P.up_level_cnt.clear();
        for (VoiceDesc& vd : hb.voices)
            if (vd.sp_ext_off >= 0) vd.sp_ext_off = (int)node_init((uint32_t)vd.stage_state[vd.n_stages - 1]).ext_off;
        P.fused_sp = hb.has_sp;
        if (hb.has_sp) HIPC(c, P.d_hist.ensure_n("d_hist", hb.voices.size() * SP_HIST * sizeof(float)));
        build_slot_voice(P, hb.voices);
        if ((rc = up(c, P.d_slot_voice, P.slot_voice.data(), P.slot_voice.size() * sizeof(int)))) return rc;
        if ((rc = up(c, P.d_voices, hb.voices.data(), hb.voices.size() * sizeof(VoiceDesc)))) return rc;
        if ((rc = up(c, P.d_leaves, hb.leaves.data(), hb.leaves.size() * sizeof(LeafDesc)))) return rc;
        if ((rc = up(c, P.d_progs, hb.progs.data(), hb.progs.size() * sizeof(uint32_t)))) return rc;
        P.lazy_capable = false;
        if ((rc = alloc_voice_tables(c, P))) return rc;
        // the level lists without the nodes the voice-bank kernels render
        std::vector<char> cov(N, 0);
        for (int i : hb.covered) cov[i] = 1;
        std::vector<int> hflat;
        P.hlevel_off.clear();
        P.hlevel_cnt.clear();
        P.hlevel_kinds.clear();
        for (auto& l : levels) {
            P.hlevel_off.push_back((int)hflat.size());
            int kinds = 0, cnt = 0;
            for (int i : l)
                if (!cov[i]) {
                    hflat.push_back(split_entry[i] >= 0 ? split_entry[i] : i);
                    kinds |= host_kind_bits(nd[i].kind);
                    cnt++;
                }
            P.hlevel_cnt.push_back(cnt);
            P.hlevel_kinds.push_back(kinds);
        }
        if (hflat.empty()) hflat.push_back(0);
        if ((rc = up(c, P.d_hlevel_nodes, hflat.data(), hflat.size() * sizeof(int)))) return rc;
        P.hybrid = true;
        P.n_fused_real = 0;
        for (const VoiceDesc& vd : hb.voices) P.n_fused_real += vd.sampler_state >= 0 ? 1 : 0;
    }
    // k_frozen_scan's verdict tables (generic executor, K > 1): sized here, on the control thread — a process call never
    // allocates
    HIPC(c, P.d_frozen.ensure_n("d_frozen", nd.size()));
    HIPC(c, P.d_frozen_ph.ensure_n("d_frozen_ph", nd.size() * sizeof(unsigned long long)));
    P.chain_words = (int)((P.generic_k + 31) / 32);
    HIPC(c, P.d_chain_done.ensure_n("d_chain_done", nd.size() * (size_t)P.chain_words * sizeof(uint32_t)));
    P.slot_index.assign(c->graph.nodes.size(), -1);
    for (int i = 0; i < N; ++i)
        if (plan.nodes[i].slot < P.slot_index.size()) P.slot_index[plan.nodes[i].slot] = i;
    P.plan = std::move(plan);  // (nothing below, and no caller, looks at `plan` again: a copy was 33 000 small vectors on config 3)
    P.have_plan = true;
    phase_mark(c, 3);
    if ((rc = build_apply(c))) return rc;
    HIPC(c, hipStreamSynchronize(c->up_stream));  // every table and every zeroed pool of the image is in place
    phase_mark(c, 4);
    c->h_up_used = 0;
    // ---- commit the control side's own bookkeeping: from here on the image WILL be adopted
    for (const Act& a : acts) {
        HostNode& n = c->graph.nodes[a.slot];
        n.init = a.st;
        n.activated = true;
        P.activated.emplace_back(a.slot, c->graph.id_of(a.slot));
    }
    for (auto& kv : new_ir) {
        c->ir_cache[kv.first] = kv.second;
        c->ir_len[kv.first] = (uint32_t)c->samples[kv.first.first].desc.frames;
    }
    c->ext_used = ext_need;
    if (P.grow_states.p) c->ctl_states_cap = P.grow_states_cap;
    if (P.grow_ext.p) c->ctl_ext_cap = P.grow_ext_cap;
    c->graph.nodes_to_activate.clear();
    P.dropped_samplers.swap(c->dropped_samplers_ctl);
    c->dropped_samplers_ctl.clear();
    P.removed_slots.swap(c->pending_removed);
    c->pending_removed.clear();
    c->graph.needs_compile = false;
    rollback.armed = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- hand-over
// Make `n` the active image.  Whoever calls holds the gate: the audio thread at the start of a process call, or the control
// thread when it found the audio side idle.  Nothing here allocates, waits for the device or can fail in a way that leaves
// the ctx half-swapped: the persistent state's changes are asynchronous launches on the ctx stream (ordered before whatever
// the next process call enqueues), the rest is member swaps and small host loops.
void adopt_image(fwgpu_ctx* c, PlanImage* n, bool on_audio_thread) {
    const auto t0 = std::chrono::steady_clock::now();
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) (void)hipSetDevice(c->device);
    (void)rt_persist_stop(c);  // the resident realtime kernel was launched with the old plan's tables
    (void)lazy_flush(c);       // blocks rendered from the OLD plan's LazyRecs reach node state before the voices are renumbered
    (void)join_streams(c);  // control-ahead mode: the control stream's work so far is ordered before the swap
    c->ahead_seq = 0;
    // 1. larger persistent arrays: old contents copied over on the stream, then the pointers change hands
    if (n->grow_states.p) {
        if (c->d_states.p && c->states_cap)
            (void)hipMemcpyAsync(n->grow_states.p, c->d_states.p, c->states_cap * sizeof(NodeState), hipMemcpyDeviceToDevice, c->stream);
        std::swap(c->d_states, n->grow_states);  // (the old array rides back in the retired image and is freed there)
        c->states_cap = n->grow_states_cap;
    }
    if (n->grow_ext.p) {
        if (c->d_ext.p && c->ext_cap)
            (void)hipMemcpyAsync(n->grow_ext.p, c->d_ext.p, c->ext_cap * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
        std::swap(c->d_ext, n->grow_ext);
        c->ext_cap = n->grow_ext_cap;
    }
    // 2. the nodes this image activates (graph.rs:594-612): ext slices zeroed / initialised, impulse responses converted,
    //    initial states scattered — into slots and slices no running plan uses
    //    ... and, in the same launch, the steady caches of the voices the edit did not touch travel to the new plan (same nodes in the
    //    same order: carry_cache_voice), stamped with the epoch the swap below sets; `c` still is the old image here
    CarryArgs carry;
    memset(&carry, 0, sizeof(carry));
    if (n->n_voices > 0 && c->n_voices > 0 && !c->slot_voice.empty() && n->d_cache.p && c->d_cache.p && c->d_slot_voice.p && n->d_voices.p && c->d_voices.p) {
        carry.new_cache = n->d_cache.as<VoiceCache>();
        carry.new_voices = n->d_voices.as<VoiceDesc>();
        carry.n_new = n->n_voices;
        carry.old_cache = c->d_cache.as<VoiceCache>();
        carry.old_voices = c->d_voices.as<VoiceDesc>();
        carry.old_slot_voice = c->d_slot_voice.as<int>();
        carry.n_old_slots = (int)c->slot_voice.size();
        carry.old_epoch = c->epoch;
        carry.new_epoch = c->epoch + 1;
    }
    if (n->n_ext_jobs || n->n_state_inits || carry.n_new)
        (void)launch_adopt_init(c->stream, c->d_ext.as<float>(), n->d_ext_jobs.p, n->n_ext_jobs, c->d_states.as<NodeState>(), n->d_state_inits.p,
                                n->n_state_inits, carry);
    if (!n->ir_convs.empty()) {
        (void)upload_sample_table(c);
        for (const PlanImage::IrConv& ic : n->ir_convs)
            (void)launch_ir_convert(c->stream, c->d_samples.as<SampleDesc>(), ic.sample, ic.ch, c->d_ext.as<float>() + ic.off, ic.T);
    }

    // 3. sampler bookkeeping (audio-side tables).  A removed sampler's processor is dropped with the old schedule and hands
    //    its sample back (sampler.rs:563-571); a newly activated node starts without one; messages still queued for a removed
    //    node go with it (its slot — the message key — is reused only after this swap)
    if (n->grow_cur_sample.size() > c->cur_sample.size()) {
        std::copy(c->cur_sample.begin(), c->cur_sample.end(), n->grow_cur_sample.begin());
        std::copy(c->slot_ids.begin(), c->slot_ids.end(), n->grow_slot_ids.begin());
        c->cur_sample.swap(n->grow_cur_sample);
        c->slot_ids.swap(n->grow_slot_ids);
    }
    auto hand_back = [&](int sample) {  // "silent" return: the control side's reference count drops, nothing is reported
        RetItem it;
        it.node = RET_SILENT;
        it.sample = sample;
        it.ticket = c->ret_ticket;
        if (c->returns.stage(it)) c->ret_this_call = true;
    };
    for (uint32_t slot : n->dropped_samplers)
        if (slot < c->cur_sample.size()) {
            if (c->cur_sample[slot] >= 0) hand_back(c->cur_sample[slot]);
            c->cur_sample[slot] = -1;
        }
    if (!n->removed_slots.empty()) {
        drain_ring(c);  // (the gate makes this thread the ring's consumer: messages not drained yet are filtered too)
        size_t w = 0;
        for (const Cmd& m : c->cmds) {
            bool gone = false;
            for (uint32_t slot : n->removed_slots) gone = gone || m.state == (int)slot;
            if (gone) {
                if (m.type == CMD_SMP_SET_SAMPLE && m.i0 >= 0) hand_back(m.i0);  // never reached its sampler: handed straight back
                continue;
            }
            c->cmds[w++] = m;
        }
        c->cmds.resize(w);
    }
    for (const auto& a : n->activated)
        if (a.first < c->cur_sample.size()) {
            c->cur_sample[a.first] = -1;
            c->slot_ids[a.first] = a.second;
        }
    if (c->ret_this_call) finish_returns(c);  // (completion event on the ctx stream: the old plan's kernels are in front of it)
    // 4. the swap
    std::swap(static_cast<PlanImage&>(*c), *n);
    std::swap(c->retired_ev, n->retired_ev);  // (the event belongs to the heap object that travels through the ring)
    c->epoch++;  // cached steady descriptors belong to the old plan (those that travel were stamped with this value above)
    c->adopted_gen.store(c->gen, std::memory_order_release);
    // 5. the old image goes back to the control side, which waits for `retired_ev` before it touches the buffers
    if (n->retired_ev) (void)hipEventRecord(n->retired_ev, c->stream);  // (created with the object, on the control thread)
    const uint32_t t = c->retired_tail.load(std::memory_order_relaxed);
    if (t - c->retired_head.load(std::memory_order_acquire) < fwgpu_ctx::RETIRE_CAP) {
        c->retired[t % fwgpu_ctx::RETIRE_CAP] = n;
        c->retired_tail.store(t + 1, std::memory_order_release);
    }  // (a full ring cannot happen: the control side collects before every build and at most one image is pending)
    const uint64_t ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    if (on_audio_thread) {
        if (ns > c->adopt_ns_max) c->adopt_ns_max = ns;
        c->audio_adoptions++;
    }
    c->adoptions++;
}

// control side: what became reusable now that image `gen` is the active one — slots and ext slices of removed nodes
static void release_limbo(fwgpu_ctx* c, uint64_t gen) {
    size_t w = 0;
    for (const fwgpu_ctx::Limbo& l : c->limbo) {
        if (l.gen > gen) {
            c->limbo[w++] = l;
            continue;
        }
        if (l.what == 0) c->graph.free_nodes.push_back(l.a);
        else c->ext_free[(size_t)l.b].push_back(l.a);
    }
    c->limbo.resize(w);
}

// control side: images the audio side is done with.  Waits for their kernels (the control thread may block), keeps one as
// the next build target — its buffers are reused, steady edits allocate nothing — and frees the rest.
void collect_retired(fwgpu_ctx* c) {
    for (;;) {
        const uint32_t h = c->retired_head.load(std::memory_order_relaxed);
        if (h == c->retired_tail.load(std::memory_order_acquire)) break;
        PlanImage* img = c->retired[h % fwgpu_ctx::RETIRE_CAP];
        c->retired_head.store(h + 1, std::memory_order_release);
        if (img->retired_ev) (void)hipEventSynchronize(img->retired_ev);
        else (void)hipStreamSynchronize(c->stream);
        img->grow_states.release();  // (after a swap these hold the OLD arrays)
        img->grow_ext.release();
        if (!c->spare) {
            c->spare = img;
        } else {
            img->release_device();
            delete img;
        }
    }
    release_limbo(c, c->adopted_gen.load(std::memory_order_acquire));
}

static bool try_adopt_from_control(fwgpu_ctx* c, PlanImage* img) {
    // FWGPU_LAZY_ADOPT=1 (tests): never adopt on the control thread — every plan is picked up by the next process call, the
    // path a host with a running audio thread takes; the whole GPU suite is run in this mode too
    static const bool lazy = getenv("FWGPU_LAZY_ADOPT") && atoi(getenv("FWGPU_LAZY_ADOPT")) != 0;
    if (lazy) return false;
    int expected = 0;
    if (!c->gate.compare_exchange_strong(expected, 2, std::memory_order_acquire)) return false;
    adopt_image(c, img, false);
    c->gate.store(0, std::memory_order_release);
    return true;
}

// Hand the image to the audio side.  At most one image waits at a time: an earlier one that no process call has picked up yet
// is adopted first (in order — each image carries what ITS adoption must do to the persistent state).
static void publish(fwgpu_ctx* c, PlanImage* img) {
    for (;;) {
        PlanImage* prev = c->pending.load(std::memory_order_acquire);
        if (!prev) break;
        int expected = 0;
        if (c->gate.compare_exchange_strong(expected, 2, std::memory_order_acquire)) {
            prev = c->pending.exchange(nullptr, std::memory_order_acq_rel);
            if (prev) adopt_image(c, prev, false);
            c->gate.store(0, std::memory_order_release);
            break;
        }
        std::this_thread::yield();  // a process call is running: it (or the next one) takes the pending image at its start
    }
    for (const Cmd& m : c->early_msgs) (void)c->ring.push(m);  // messages sent to nodes before the update that activates them
    c->early_msgs.clear();
    if (!try_adopt_from_control(c, img)) c->pending.store(img, std::memory_order_release);
    collect_retired(c);
}

// fwgpu_update / fwgpu_schedule_upload: build off to the side, publish.  A failure leaves the active plan (and a pending one)
// exactly as they were, like the reference keeps its schedule when a compile fails (context.rs:115-131).
int install_plan(fwgpu_ctx* c, Plan& plan) {
    if (!c->up_stream) {  // the build's uploads and memsets: lowest priority.  (A stream restricted to 16 CUs — hipExtStreamCreateWithCUMask —
                          // made the callbacks beside a build SLOWER, p99 250 -> 325 us: what costs them is how LONG the build's
                          // copy / fill kernels run next to them, not how many CUs those take.)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->up_stream, hipStreamNonBlocking, lo) != hipSuccess) {
            (void)hipGetLastError();
            HIPC(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        }
    }
    collect_retired(c);
    PlanImage* P = c->spare;
    c->spare = nullptr;
    if (!P) {
        P = new (std::nothrow) PlanImage();
        if (!P) return fail(c, FWGPU_ERR_DEVICE, "out of host memory");
        if (hipEventCreateWithFlags(&P->retired_ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            P->retired_ev = nullptr;  // (collect_retired then waits for the whole stream instead)
        }
    }
    const int rc = build_image(c, plan, *P);
    if (rc != 0) {
        (void)build_apply(c);  // (the shadows of the tables say these chunks are on the device: they go there, into an image nobody reads)
        (void)hipStreamSynchronize(c->up_stream);
        (void)hipGetLastError();
        c->h_up_used = 0;
        c->spare = P;  // (its buffers are reusable whatever state the build left them in)
        return rc;
    }
    // the control side's mirror of what the introspection calls report
    c->info.have_plan = true;
    c->info.kind = c->force_generic ? 0 : (P->fused ? (P->fused_fx ? 2 : 1) : (P->hybrid ? 3 : 0));
    c->info.fused_voices = (c->force_generic || !(P->fused || P->hybrid)) ? 0 : P->n_fused_real;
    c->info.n_host_nodes = P->n_host_nodes;
    c->info.plan = P->plan;
    publish(c, P);
    return 0;
}

}  // namespace fwgpu
