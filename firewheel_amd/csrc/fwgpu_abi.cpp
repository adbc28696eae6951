// fwgpu_abi.cpp — the C ABI of include/fwgpu.h: device counterpart of FirewheelGraphCtx + FirewheelProcessor
// (graph/context.rs, graph/processor.rs).  The host half keeps the editable graph, the device half keeps node state,
// the buffer pool and the launch plan in HBM.  No CPU compute path exists: every process call is kernels.
#include "fwgpu_ctx.h"

#include <chrono>
#include <memory>
#include <stdio.h>

namespace {

// hipSetDevice costs ~8 us per call on this stack (measured: it was most of the host time of a realtime callback); the
// thread's current device is asked first, which is a thread-local read
inline void use_device(fwgpu_ctx* c) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) (void)hipSetDevice(c->device);
}

thread_local std::string g_create_error;  // fwgpu_create_error(): of the calling thread's last failed fwgpu_ctx_create

// Control side of a message: validate against the graph (owned by the control side), account for the sampler's ring
// capacity, then hand the message to the lock-free ring.  May run while the audio thread is inside a process call.
int push_cmd(fwgpu_ctx* c, int64_t node, int want_kind, Cmd m, bool counts_as_msg) {
    HostNode* n = c->graph.get(node);
    if (!n) return fail(c, FWGPU_ERR_INVALID, "unknown node id");
    if (want_kind >= 0 && n->kind != want_kind) return fail(c, FWGPU_ERR_INVALID, "node kind does not accept this message");
    if (counts_as_msg) {  // sampler.rs:14 CHANNEL_CAPACITY messages between two drains of the ring
        const uint64_t ep = c->drain_epoch.load(std::memory_order_acquire);
        if (n->pending_epoch != ep) {
            n->pending_epoch = ep;
            n->pending_msgs = 0;
        }
        if (n->pending_msgs >= 128) return fail(c, FWGPU_ERR_QUEUE_FULL, "sampler message ring full");
    }
    m.state = (int)(node & 0xffffffff);
    if (!n->activated) {
        // no built plan holds this node yet (add_node, then a message, then fwgpu_update — the reference's ring simply waits
        // for the processor's first block): the message is released when the plan that activates it is published, so that a
        // process call running on the OLD plan neither consumes it nor applies it to a stale slot.  Control side, like update.
        if (c->early_msgs.size() >= fwgpu_ctx::RING_CAP) return fail(c, FWGPU_ERR_QUEUE_FULL, "too many messages for nodes that are not activated yet");
        c->early_msgs.push_back(m);
        if (counts_as_msg) n->pending_msgs++;
        return 0;
    }
    if (!c->ring.push(m)) return fail(c, FWGPU_ERR_QUEUE_FULL, "context message ring full (no process call is draining it)");
    if (counts_as_msg) n->pending_msgs++;
    return 0;
}

void rebuild_sample_table(fwgpu_ctx* c) {  // control side: host copy of the device sample table
    c->h_sample_tab.resize(std::max<size_t>(c->samples.size(), 1));
    if (c->samples.empty()) memset(&c->h_sample_tab[0], 0, sizeof(SampleDesc));
    for (size_t i = 0; i < c->samples.size(); ++i) c->h_sample_tab[i] = c->samples[i].desc;
    c->samples_dirty = true;  // copied to the device by the next process / update call
}

// swapped-out samples whose process call has completed on the device: RetRing -> ret_ready, reference counts updated
void collect_returns(fwgpu_ctx* c) {
    RetItem it;
    while (c->returns.peek(it)) {
        // tickets are issued in call order: an unfinished call ends the scan.  An event slot that has been re-recorded
        // since (64 calls with returns later) answers for the later call — later, never earlier, than the truth.
        if (it.ticket >= c->ret_done_ticket.load(std::memory_order_acquire)) {
            const uint32_t slot = it.ticket % fwgpu_ctx::RET_EVENTS;
            if (c->ret_event_ticket[slot].load(std::memory_order_acquire) <= it.ticket) break;  // not recorded for this call yet
            if (hipEventQuery(c->ret_events[slot]) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
        }
        c->returns.pop();
        if (it.sample >= 0 && (size_t)it.sample < c->sample_refs.size() && c->sample_refs[it.sample] > 0) c->sample_refs[it.sample]--;
        if (it.node != RET_SILENT) c->ret_ready.push_back(it);
    }
}

}  // namespace

// ================================================================= C ABI
extern "C" {

// every entry point that takes a context: a null handle is an error return, never a crash
#define NEED_CTX(c, ret) \
    do {                 \
        if (!(c)) return (ret); \
    } while (0)

const char* fwgpu_create_error(void) { return g_create_error.c_str(); }

fwgpu_ctx* fwgpu_ctx_create(int device, uint32_t sample_rate, uint32_t max_block_frames, uint32_t num_graph_inputs,
                            uint32_t num_graph_outputs, void* hip_stream) {
    g_create_error.clear();
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e) +
                         " (libfwgpu has no CPU fallback)";
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "device index out of range";
        return nullptr;
    }
    if (max_block_frames == 0 || num_graph_inputs > 64 || num_graph_outputs > 64) {
        g_create_error = "invalid arguments (max_block_frames > 0, <= 64 graph channels)";
        return nullptr;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_create_error = std::string("device is ") + prop.gcnArchName + "; libfwgpu ships gfx950 (MI355X) code only";
        return nullptr;
    }
    fwgpu_ctx* c = new fwgpu_ctx(num_graph_inputs, num_graph_outputs);
    c->graph.limbo = &c->limbo_slots;  // a removed node's slot is reused only once no running plan holds the node
    c->device = device;
    c->sample_rate = sample_rate;
    c->mbf = max_block_frames;
    c->stride = (int)((max_block_frames + 63) / 64 * 64);
    c->n_gin = num_graph_inputs;
    c->n_gout = num_graph_outputs;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
            g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
            delete c;
            return nullptr;
        }
        c->own_stream = true;
    }
    // message path: every buffer gets its final size here (a process call never allocates for messages)
    {
        bool ok = c->ring.init(fwgpu_ctx::RING_CAP) && c->returns.init(4096);
        c->cmds.reserve(fwgpu_ctx::CMD_CAP);
        c->cmds_scratch.reserve(fwgpu_ctx::CMD_CAP);
        c->ret_ready.reserve(4096);
        ok = ok && c->d_cmds.ensure_n("d_cmds", fwgpu_ctx::CMD_CAP * sizeof(Cmd)) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&c->h_cmds, fwgpu_ctx::CMD_CAP * sizeof(Cmd), hipHostMallocDefault) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->cmds_copied, hipEventDisableTiming) == hipSuccess;
        for (uint32_t i = 0; ok && i < fwgpu_ctx::RET_EVENTS; ++i)
            ok = hipEventCreateWithFlags(&c->ret_events[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            g_create_error = "message ring / staging allocation failed";
            fwgpu_ctx_destroy(c);
            return nullptr;
        }
    }
    rebuild_sample_table(c);
    {
        std::vector<float> tab(RS_PHASES * RS_TAPS);
        resampler_table(tab.data());
        if (upload(c, c->d_rs_table, tab.data(), tab.size() * sizeof(float)) != 0) {
            g_create_error = c->err_ctl;
            fwgpu_ctx_destroy(c);
            return nullptr;
        }
    }
    if (upload_sample_table(c) != 0 || c->d_mask.ensure_n("d_mask", 64) != hipSuccess) {
        g_create_error = c->err_ctl;
        fwgpu_ctx_destroy(c);
        return nullptr;
    }
    // realtime I/O blocks: allocated here, on the control thread (the process calls never allocate).  A failure is not
    // fatal: process_interleaved then always takes the staged-copy path.
    if (hipHostMalloc((void**)&c->h_rt_in, RT_IO_BYTES, hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc((void**)&c->h_rt_out, RT_IO_BYTES, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_rt_in, c->h_rt_in, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_rt_out, c->h_rt_out, 0) != hipSuccess) {
        if (c->h_rt_in) (void)hipHostFree(c->h_rt_in);
        if (c->h_rt_out) (void)hipHostFree(c->h_rt_out);
        c->h_rt_in = c->h_rt_out = nullptr;
        (void)hipGetLastError();
    }
    if (hipHostMalloc((void**)&c->h_rt_flag, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_rt_flag, c->h_rt_flag, 0) != hipSuccess) {
        if (c->h_rt_flag) (void)hipHostFree(c->h_rt_flag);
        c->h_rt_flag = c->d_rt_flag = nullptr;
        (void)hipGetLastError();
    } else {
        *c->h_rt_flag = 0;
    }
    if (const char* e = getenv("FWGPU_RT_SPIN")) {
        if (atoi(e) == 0 && c->h_rt_flag) {  // experiments: blocking stream sync instead of the polled completion flag
            (void)hipHostFree(c->h_rt_flag);
            c->h_rt_flag = c->d_rt_flag = nullptr;
        }
    }
    if (const char* e = getenv("FWGPU_CTL_AHEAD")) {
        c->ctl_ahead = atoi(e) != 0;
        if (atoi(e) == 1) c->ctl_ahead_mode = 1;
    }
    if (c->ctl_ahead) {  // its own high-priority stream + the events that order it against the render stream
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        bool ok = hipStreamCreateWithPriority(&c->ctl_stream, hipStreamNonBlocking, hi) == hipSuccess;
        for (int i = 0; ok && i < 2; ++i)
            ok = hipEventCreateWithFlags(&c->ev_ctl[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&c->ev_render[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            c->ctl_ahead = false;
        }
    }
    if (const char* e = getenv("FWGPU_LAZY")) c->lazy_on = atoi(e) != 0;
    if (c->lazy_on) {  // the control kernels' horizon word + where it is published for the host (fwgpu_types.h LazyRec)
        bool ok = c->d_lazy_horizon.ensure_n("d_lazy_horizon", 256) == hipSuccess && hipMemset(c->d_lazy_horizon.p, 0xff, 256) == hipSuccess &&
                  hipHostMalloc((void**)&c->h_lazy_pub, 64, hipHostMallocMapped) == hipSuccess &&
                  hipHostGetDevicePointer((void**)&c->d_lazy_pub, c->h_lazy_pub, 0) == hipSuccess;
        if (ok) {
            memset(c->h_lazy_pub, 0, 64);
        } else {
            (void)hipGetLastError();
            c->lazy_on = false;
        }
    }
    if (const char* e = getenv("FWGPU_RT_PERSIST")) c->rt_persist = atoi(e) != 0;
    if (const char* e = getenv("FWGPU_LEVEL_FUSE")) c->level_fuse = atoi(e) != 0;
    if (const char* e = getenv("FWGPU_GATE_DEFER_US")) c->gate_defer_ns = (uint64_t)std::max(0, atoi(e)) * 1000ull;
    if (const char* e = getenv("FWGPU_RT_IDLE_MS")) c->rt_idle_ms = (uint32_t)std::max(1, atoi(e));
    const bool quiet_given = getenv("FWGPU_QUIET_WAIT_US") != nullptr;
    if (const char* e = getenv("FWGPU_QUIET_WAIT_US")) c->quiet_wait_us = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("FWGPU_UP_DIFF")) c->up_diff = atoi(e) != 0;
    if (const char* e = getenv("FWGPU_BUILD_ONE_KERNEL")) c->build_one_kernel = atoi(e) != 0;
    if (const char* e = getenv("FWGPU_BUILD_STREAM")) c->build_on_audio_stream = strcmp(e, "own") != 0;
    // a group in the AUDIO stream queues behind whatever callback is in flight and costs the next one its own few microseconds wherever it
    // lands: beside back-to-back callbacks the 100 us a group used to wait for a window that never comes were 0.2 ms of every edit
    // (fw_edit_race, same box: typical edit 1.13 -> 0.96 ms, same p99).  On its own stream a group still waits the longer time.
    if (!quiet_given && !c->build_on_audio_stream) c->quiet_wait_us = 100;
    if (const char* e = getenv("FWGPU_UP_PIECE")) c->up_piece = (uint32_t)std::max(4096, atoi(e));
    if (c->rt_persist && c->h_rt_flag) {  // mailbox in pinned, device-mapped host memory + the kernel's own (non-blocking) stream
        bool ok = hipHostMalloc((void**)&c->h_rt_mb, sizeof(RtMailbox), hipHostMallocMapped) == hipSuccess &&
                  hipHostGetDevicePointer((void**)&c->d_rt_mb, c->h_rt_mb, 0) == hipSuccess;
        if (ok) memset(c->h_rt_mb, 0, sizeof(RtMailbox));
        // HIGH priority, not for the priority: ROCm multiplexes a process's streams onto a few hardware queues PER PRIORITY LEVEL, and
        // whatever shares a queue with a kernel that never ends waits behind it — a plain hipMemcpy of a control thread (the null
        // stream) sat out the whole 4 s of a fed resident kernel (r04, tests/test_rt_resident.py).  The high-priority pool holds
        // only this stream and ctl_stream, which is never used while a resident kernel runs (throughput calls end it first).
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        ok = ok && hipStreamCreateWithPriority(&c->rt_stream, hipStreamNonBlocking, hi) == hipSuccess &&
             hipEventCreateWithFlags(&c->rt_ev, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            c->rt_persist = false;
        }
    } else {
        c->rt_persist = false;
    }
    if (const char* e = getenv("FWGPU_HOST_PROF")) c->host_prof = atoi(e) != 0;
    if (const char* e = getenv("FWGPU_PLAN_ORDER")) c->graph.canonical_order = strcmp(e, "reference") != 0;
    if (const char* e = getenv("FWGPU_UPDATE_PROF")) {
        c->update_prof = atoi(e) != 0;
        c->update_prof_tables = atoi(e) >= 2;
    }
    if (const char* e = getenv("FWGPU_RT_GRAPH")) c->rt_use_graph = atoi(e) != 0;
    if (const char* e = getenv("FWGPU_RT_ONE_LAUNCH")) c->rt_one_launch = atoi(e) != 0;
    if (const char* e = getenv("FWGPU_RT_PERSIST_MAX_LEAVES")) c->rt_persist_max_leaves = std::max(0, atoi(e));
    if (c->d_rt_sync.ensure_n("d_rt_sync", 256) != hipSuccess || hipMemset(c->d_rt_sync.p, 0, 256) != hipSuccess) c->d_rt_sync.release();
    return c;
}

void fwgpu_ctx_destroy(fwgpu_ctx* c) {
    if (!c) return;
    if (c->host_prof && c->hp_calls)
        fprintf(stderr, "fwgpu host profile: %llu process calls, %.2f us each inside run_blocks; %llu k_rt_block launches, %.2f us each inside the HIP launch call\n",
                (unsigned long long)c->hp_calls, c->hp_call_ns / 1e3 / (double)c->hp_calls, (unsigned long long)c->hp_launches,
                c->hp_launches ? c->hp_launch_ns / 1e3 / (double)c->hp_launches : 0.0);
    if (c->host_prof && c->hp_calls) {
        fprintf(stderr, "fwgpu host profile: host time inside run_blocks, 25 us bins:");
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", (unsigned long long)c->hp_hist[i]);
        fprintf(stderr, "\n");
    }
    if (c->update_prof && c->phase_updates > 1) {
        const double n = (double)(c->phase_updates - 1);
        fprintf(stderr, "fwgpu update profile: %.0f updates after the first; mean us per phase (fwgpu_update_phase):", n);
        for (int i = 1; i < 32; ++i)
            if (c->phase_ns[i]) fprintf(stderr, " [%d] %.1f", i, c->phase_ns[i] / 1e3 / n);
        fprintf(stderr, "; per update: %.1f launches of k_build_apply, %.1f jobs, %.1f KiB copied, %.1f KiB filled\n", c->prof_groups / n, c->prof_jobs / n,
                c->prof_copy_bytes / 1024.0 / n, c->prof_fill_bytes / 1024.0 / n);
    }
    use_device(c);
    (void)rt_persist_stop(c);
    // (in the default output mode the graph-output kernel writes the mapped host staging below FROM c->stream; a ctx
    //  destroyed with a begun, un-ended ticket must not unregister / free it under a kernel that is still storing into it)
    (void)hipStreamSynchronize(c->stream);
    if (c->rt_stream) (void)hipStreamDestroy(c->rt_stream);
    if (c->rt_ev) (void)hipEventDestroy(c->rt_ev);
    if (c->ao.copy_stream) {
        (void)hipStreamSynchronize(c->ao.copy_stream);
        (void)hipStreamDestroy(c->ao.copy_stream);
    }
    if (c->host_prof && c->ao.prof_calls)
        fprintf(stderr, "fwgpu host profile: %llu process_interleaved_end calls: %.1f us waiting for the copy back, %.1f us in memcpy each\n",
                (unsigned long long)c->ao.prof_calls, c->ao.prof_wait_ns / 1e3 / (double)c->ao.prof_calls, c->ao.prof_copy_ns / 1e3 / (double)c->ao.prof_calls);
    for (int i = 0; i < 2; ++i) {
        if (c->ao.ev_render[i]) (void)hipEventDestroy(c->ao.ev_render[i]);
        if (c->ao.ev_copy[i]) (void)hipEventDestroy(c->ao.ev_copy[i]);
        if (c->ao.h[i]) {
            (void)hipHostUnregister(c->ao.h[i]);
            free(c->ao.h[i]);
        }
        c->ao.d[i].release();
    }
    if (c->h_rt_mb) (void)hipHostFree(c->h_rt_mb);
    if (c->h_lazy_pub) (void)hipHostFree(c->h_lazy_pub);
    (void)hipStreamSynchronize(c->stream);
    if (c->ctl_stream) {
        (void)hipStreamSynchronize(c->ctl_stream);
        (void)hipStreamDestroy(c->ctl_stream);
    }
    for (hipEvent_t e : {c->ev_ctl[0], c->ev_ctl[1], c->ev_render[0], c->ev_render[1], c->ev_join})
        if (e) (void)hipEventDestroy(e);
    for (SampleRec& s : c->samples)
        if (s.alive && s.owned && s.d_data) (void)hipFree(s.d_data);
    // plan images: the active one (this ctx's base), a published one nobody adopted, the spare, whatever the ring still holds
    if (PlanImage* p = c->pending.exchange(nullptr)) {
        p->release_device();
        delete p;
    }
    if (c->spare) {
        c->spare->release_device();
        delete c->spare;
    }
    for (uint32_t h = c->retired_head.load(); h != c->retired_tail.load(); ++h) {
        PlanImage* p = c->retired[h % fwgpu_ctx::RETIRE_CAP];
        p->release_device();
        delete p;
    }
    c->release_device();
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->h_up) (void)hipHostFree(c->h_up);
    if (c->h_jobs) (void)hipHostFree(c->h_jobs);
    if (c->ev_build) (void)hipEventDestroy(c->ev_build);
    DevBuf* bufs[] = {&c->d_states, &c->d_ext, &c->d_samples, &c->d_rt_sync, &c->d_lazy_horizon, &c->d_cmds, &c->d_in_stage, &c->d_out_stage, &c->d_scratch_pool,
                      &c->d_scratch_flags, &c->d_scratch_tab, &c->d_mask, &c->d_trace, &c->d_rs_table};
    for (DevBuf* b : bufs) b->release();
    for (TimerCat& t : c->timers)
        for (auto& p : t.ev) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    if (c->h_rt_in) (void)hipHostFree(c->h_rt_in);
    if (c->h_rt_out) (void)hipHostFree(c->h_rt_out);
    if (c->h_rt_flag) (void)hipHostFree(c->h_rt_flag);
    if (c->h_cmds) (void)hipHostFree(c->h_cmds);
    if (c->cmds_copied) (void)hipEventDestroy(c->cmds_copied);
    for (hipEvent_t e : c->ret_events)
        if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* fwgpu_last_error(fwgpu_ctx* c) {
    if (!c) return "null ctx";
    return c->err_last.load(std::memory_order_acquire) ? c->err_audio : c->err_ctl;
}

int64_t fwgpu_graph_in_node(fwgpu_ctx* c) { return c ? c->graph.id_of(c->graph.graph_in_slot) : FWGPU_ERR_INVALID; }
int64_t fwgpu_graph_out_node(fwgpu_ctx* c) { return c ? c->graph.id_of(c->graph.graph_out_slot) : FWGPU_ERR_INVALID; }

int64_t fwgpu_add_node(fwgpu_ctx* c, int kind, uint32_t n_in, uint32_t n_out, const float* params, int n_params) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (kind < 0 || kind > K_LAST) return fail(c, FWGPU_ERR_INVALID, "unsupported node kind");
    if (n_params < 0 || (n_params > 0 && !params)) return fail(c, FWGPU_ERR_INVALID, "params is null but n_params > 0");
    if (kind == K_FIR || kind == K_RESAMPLER) {
        int ir = n_params > 0 ? (int)params[0] : -1;
        if (ir < 0 || ir >= (int)c->samples.size() || !c->samples[ir].alive || c->samples[ir].desc.frames == 0)
            return fail(c, FWGPU_ERR_INVALID, kind == K_FIR ? "FIR node: params[0] must be the id of a non-empty impulse-response sample"
                                                            : "Resampler node: params[0] must be the id of a non-empty source sample");
        if (kind == K_FIR && c->samples[ir].desc.frames > (1u << 24))
            return fail(c, FWGPU_ERR_INVALID, "FIR node: at most 2^24 taps");
        if (kind == K_RESAMPLER && c->samples[ir].desc.frames >= (1ull << 31))
            return fail(c, FWGPU_ERR_INVALID, "Resampler node: source longer than 2^31 frames");
    }
    if (n_in > 64 || n_out > 64) return fail(c, FWGPU_ERR_INVALID, "a node has at most 64 ports per side (core/node.rs:62,69)");
    NodeState st = make_state(kind, params, n_params, c->sample_rate);
    return c->graph.add_node(kind, n_in, n_out, st);
}
int fwgpu_remove_node(fwgpu_ctx* c, int64_t node) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    // the node's slice of the ext state pool (delay rings, FIR history, biquad / spatialiser state) goes back to a free
    // list keyed by size; the next activation of a node that needs the same amount reuses it, zeroed (install_plan) —
    // a host that spawns and retires effect voices does not grow HBM without bound
    struct Freed {
        bool any = false, activated = false;
        uint32_t off = 0, len = 0;
        int kind = 0, ir = -1;
    } fr;
    if (HostNode* hn = c->graph.get(node))
        if (hn->activated) {
            fr.activated = true;
            fr.any = hn->init.ext_len != 0;
            fr.off = hn->init.ext_off;
            fr.len = hn->init.ext_len;
            fr.ir = hn->kind == K_FIR ? hn->init.sample : -1;
        }
    if (HostNode* hn = c->graph.get(node)) fr.kind = hn->kind;
    int rc = c->graph.remove_node(node);
    if (rc) return fail(c, rc, "remove_node: unknown node or graph in/out node");
    if (fr.any) c->limbo.push_back({c->build_gen + 1, 1, fr.off, (uint32_t)(((size_t)fr.len + 63) / 64 * 64)});
    if (fr.ir >= 0) {  // the f32 copy of an impulse response goes with its last FIR user
        bool used = false;
        for (const HostNode& n : c->graph.nodes)
            if (n.alive && n.kind == K_FIR && n.init.sample == fr.ir) used = true;
        if (!used)
            for (auto it = c->ir_cache.begin(); it != c->ir_cache.end();) {
                if (it->first.first == fr.ir) {
                    c->limbo.push_back({c->build_gen + 1, 1, it->second, (uint32_t)(((size_t)c->ir_len[it->first] + 63) / 64 * 64)});
                    c->ir_len.erase(it->first);
                    it = c->ir_cache.erase(it);
                } else {
                    ++it;
                }
            }
    }
    // The node is gone from the GRAPH; the running plan may hold it until the next plan is adopted.  What it still owns —
    // its slot (the message key), its ext slice, messages queued for it, a sampler's sample — is released AT that adoption
    // (adopt_image) or after it (release_limbo), never from this thread while a process call may be using it.
    const uint32_t slot = (uint32_t)(node & 0xffffffff);
    const uint64_t tag = c->build_gen + 1;  // the first image built without the node
    for (uint32_t sl : c->limbo_slots) c->limbo.push_back({tag, 0, sl, 0});
    c->limbo_slots.clear();
    if (fr.activated) c->pending_removed.push_back(slot);
    {  // (messages that never left the control side)
        size_t w = 0;
        for (const Cmd& m : c->early_msgs) {
            if (m.state == (int)slot) {
                if (m.type == CMD_SMP_SET_SAMPLE && m.i0 >= 0 && (size_t)m.i0 < c->sample_refs.size() && c->sample_refs[m.i0] > 0) c->sample_refs[m.i0]--;
                continue;
            }
            c->early_msgs[w++] = m;
        }
        c->early_msgs.resize(w);
    }
    if (fr.kind == K_SAMPLER && fr.activated) c->dropped_samplers_ctl.push_back(slot);  // Drop: sampler.rs:563-571
    return 0;
}
int64_t fwgpu_connect(fwgpu_ctx* c, int64_t src, uint32_t sp, int64_t dst, uint32_t dp, int check) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return c->graph.connect(src, sp, dst, dp, check != 0);
}
int fwgpu_disconnect(fwgpu_ctx* c, int64_t src, uint32_t sp, int64_t dst, uint32_t dp) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return c->graph.disconnect(src, sp, dst, dp);
}
int fwgpu_disconnect_edge(fwgpu_ctx* c, int64_t e) { return c ? c->graph.disconnect_edge(e) : FWGPU_ERR_INVALID; }
int fwgpu_cycle_detected(fwgpu_ctx* c) { return c ? (c->graph.cycle_detected() ? 1 : 0) : FWGPU_ERR_INVALID; }

int fwgpu_host_node_set_process(fwgpu_ctx* c, int64_t node, fwgpu_host_process_fn fn, void* user) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    HostNode* hn = c->graph.get(node);
    if (!hn || hn->kind != K_HOST) return fail(c, FWGPU_ERR_INVALID, "not a host node (fwgpu_add_node with FWGPU_HOST_NODE)");
    if (!fn) return fail(c, FWGPU_ERR_INVALID, "host node: null process function");
    const size_t slot = (size_t)(node & 0xffffffff);
    if (c->host_procs.size() <= slot) c->host_procs.resize(std::max<size_t>(slot + 1, c->host_procs.size() * 2));
    c->host_procs[slot].fn = fn;
    c->host_procs[slot].user = user;
    c->graph.needs_compile = true;  // the plan holds a copy of the function pointer: the next update installs the new one
    return 0;
}
int fwgpu_plan_host_nodes(fwgpu_ctx* c, uint64_t* callbacks_run) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (callbacks_run) *callbacks_run = c->host_callbacks;  // (of the active image: meaningful once the audio side is quiet)
    return c->info.have_plan ? c->info.n_host_nodes : -1;
}

// which part of fwgpu_update the control thread is in (include/fwgpu.h): read by any thread, written by the updating one
int fwgpu_update_phase(fwgpu_ctx* c) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return c->update_phase.load(std::memory_order_relaxed);
}
int fwgpu_update(fwgpu_ctx* c) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    use_device(c);
    if (!c->graph.needs_compile && c->info.have_plan) return 0;
    Plan plan = std::move(c->spare_plan);  // (the node array of the image recycled by the last build: control side only)
    c->spare_plan = Plan();
    std::string err;
    phase_mark(c, 1);
    int rc = c->graph.build_plan(plan, err);
    if (rc) {
        phase_mark(c, 0);
        return fail(c, rc, err);
    }
    phase_mark(c, 2);
    {
        RtHold hold(c);  // table growth frees device memory (hipFree waits for every stream): no resident realtime kernel meanwhile
        rc = install_plan(c, plan);
    }
    phase_mark(c, 0);
    return rc;
}

int fwgpu_schedule_upload(fwgpu_ctx* c, const fwgpu_sched_node* sn, uint32_t n_nodes, uint32_t num_buffers) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    use_device(c);
    if (n_nodes < 2 || !sn) return fail(c, FWGPU_ERR_INVALID, "a schedule holds at least graph_in and graph_out");
    for (uint32_t i = 0; i < n_nodes; ++i)
        if ((sn[i].num_inputs && (!sn[i].in_buffer_index || !sn[i].in_should_clear)) || (sn[i].num_outputs && !sn[i].out_buffer_index))
            return fail(c, FWGPU_ERR_INVALID, "schedule node with ports but null buffer tables");
    Plan plan;
    std::vector<std::pair<int, int>> last_writer(num_buffers, std::make_pair(-1, 0));
    std::vector<char> seen(c->graph.nodes.size(), 0);  // a node runs once per block (schedule.rs:289-344): two entries for one
                                                       // node would be two waves read-modify-writing one NodeState
    for (uint32_t i = 0; i < n_nodes; ++i) {
        HostNode* hn = c->graph.get(sn[i].node);
        if (!hn) return fail(c, FWGPU_ERR_INVALID, "schedule names an unknown node");
        {
            const uint32_t slot = (uint32_t)(sn[i].node & 0xffffffff);
            if (seen[slot]) return fail(c, FWGPU_ERR_INVALID, "schedule names a node twice");
            seen[slot] = 1;
            const bool io = slot == c->graph.graph_in_slot || slot == c->graph.graph_out_slot;
            if (io && i != 0 && i != n_nodes - 1)
                return fail(c, FWGPU_ERR_INVALID, "graph_in / graph_out may only be the first / last schedule entry");
        }
        if (hn->n_in != sn[i].num_inputs || hn->n_out != sn[i].num_outputs)
            return fail(c, FWGPU_ERR_INVALID, "schedule port counts differ from add_node");
        std::string err;
        if (!check_activation(hn->kind, hn->n_in, hn->n_out, err)) return fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, err);
        PlanNode pn;
        pn.slot = (uint32_t)(sn[i].node & 0xffffffff);
        pn.kind = hn->kind;
        pn.n_in = (int)hn->n_in;
        pn.n_out = (int)hn->n_out;
        pn.level = 0;
        pn.is_graph_io = pn.slot == c->graph.graph_in_slot ? 1 : (pn.slot == c->graph.graph_out_slot ? 2 : 0);
        pn.in_src_node.assign(pn.n_in, -1);
        pn.in_src_port.assign(pn.n_in, 0);
        for (int p = 0; p < pn.n_in; ++p) {
            if (sn[i].in_should_clear[p]) continue;  // unconnected (InBufferAssignment.should_clear)
            uint32_t b = sn[i].in_buffer_index[p];
            if (b >= num_buffers || last_writer[b].first < 0)
                return fail(c, FWGPU_ERR_INVALID, "schedule input reads a buffer no earlier node wrote");
            pn.in_src_node[p] = last_writer[b].first;
            pn.in_src_port[p] = last_writer[b].second;
        }
        for (int p = 0; p < pn.n_out; ++p) {
            uint32_t b = sn[i].out_buffer_index[p];
            if (b >= num_buffers) return fail(c, FWGPU_ERR_INVALID, "schedule buffer index out of range");
            last_writer[b] = std::make_pair((int)i, p);
        }
        plan.nodes.push_back(pn);
    }
    if (plan.nodes.front().is_graph_io != 1 || plan.nodes.back().is_graph_io != 2)
        return fail(c, FWGPU_ERR_INVALID, "schedule must start with graph_in and end with graph_out");
    finalize_plan(plan);
    RtHold hold(c);  // (as fwgpu_update)
    return install_plan(c, plan);
}

// The introspection calls describe the LATEST BUILT plan (adopted already or waiting for the next process call): the control
// side's own mirror, so they never look at tables the audio side may be swapping.
int fwgpu_plan_kind(fwgpu_ctx* c) {
    if (!c || !c->info.have_plan) return -1;
    if (c->force_generic) return 0;
    return c->info.kind;
}
int fwgpu_plan_fused_voices(fwgpu_ctx* c) {
    if (!c || !c->info.have_plan) return -1;
    return c->force_generic ? 0 : c->info.fused_voices;
}
int fwgpu_plan_num_levels(fwgpu_ctx* c) { return c && c->info.have_plan ? c->info.plan.num_levels : -1; }
int fwgpu_plan_node_level(fwgpu_ctx* c, int64_t node) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (!c->info.have_plan || !c->graph.get(node)) return -1;
    uint32_t slot = (uint32_t)(node & 0xffffffff);
    for (const PlanNode& p : c->info.plan.nodes)
        if (p.slot == slot) return p.level;
    return -1;
}
int fwgpu_plan_node_inputs_clear(fwgpu_ctx* c, int64_t node, int* should_clear, int cap) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (!c->info.have_plan || !c->graph.get(node)) return -1;
    uint32_t slot = (uint32_t)(node & 0xffffffff);
    for (const PlanNode& p : c->info.plan.nodes)
        if (p.slot == slot) {
            for (int i = 0; i < p.n_in && i < cap; ++i) should_clear[i] = p.in_buf[i] == 0 ? 1 : 0;
            return p.n_in;
        }
    return -1;
}
int fwgpu_set_force_generic(fwgpu_ctx* c, int on) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    ControlGate gate(c);
    c->force_generic = on != 0;
    return 0;
}
int fwgpu_plan_handover_stats(fwgpu_ctx* c, uint64_t* adoptions, uint64_t* audio_adoptions, uint64_t* max_adopt_ns) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (adoptions) *adoptions = c->adoptions;
    if (audio_adoptions) *audio_adoptions = c->audio_adoptions;
    if (max_adopt_ns) *max_adopt_ns = c->adopt_ns_max;
    return 0;
}
int fwgpu_plan_pending(fwgpu_ctx* c) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return c->pending.load(std::memory_order_acquire) != nullptr ? 1 : 0;
}
int fwgpu_lazy_stats(fwgpu_ctx* c, uint64_t* lazy_batches, uint64_t* control_batches) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (lazy_batches) *lazy_batches = c->lazy_calls;
    if (control_batches) *control_batches = c->ctl_calls;
    return 0;
}
void* fwgpu_hip_stream(fwgpu_ctx* c) { return c ? (void*)c->stream : nullptr; }
int fwgpu_rt_resident_stats(fwgpu_ctx* c, uint64_t* launches, uint64_t* doorbells) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (launches) *launches = c->rtp.launches;
    if (doorbells) *doorbells = c->rtp.doorbells;
    return 0;
}
int fwgpu_rt_path_stats(fwgpu_ctx* c, uint64_t* paths) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (!paths) return fail(c, FWGPU_ERR_INVALID, "null paths");
    for (int i = 0; i < 4; ++i) paths[i] = c->rt_path[i];
    return 0;
}
int fwgpu_plan_chain_stats(fwgpu_ctx* c, uint64_t* steady_workgroups, uint64_t* general_workgroups) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    use_device(c);
    unsigned long long h[2] = {0, 0};
    if (c->d_chain_stats.p) {
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, hipMemcpy(h, c->d_chain_stats.p, sizeof(h), hipMemcpyDeviceToHost));
    }
    if (steady_workgroups) *steady_workgroups = h[0];
    if (general_workgroups) *general_workgroups = h[1];
    return 0;
}
int fwgpu_set_max_batch(fwgpu_ctx* c, uint32_t k) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (k == 0) return fail(c, FWGPU_ERR_INVALID, "max batch must be >= 1");
    c->kmax_req = k;
    c->graph.needs_compile = true;  // K-sized buffers are (re)allocated by the next fwgpu_update
    return 0;
}

static size_t fmt_elem_size(int fmt) { return (fmt == FMT_I_F32 || fmt == FMT_P_F32) ? 4 : 2; }

static int sample_add(fwgpu_ctx* c, int format, uint32_t channels, uint64_t frames, const void* data, bool on_device) {
    use_device(c);
    if (format < 0 || format > FMT_P_F32 || channels == 0) return fail(c, FWGPU_ERR_INVALID, "bad sample format/channels");
    if (frames > (1ull << 40) / channels) return fail(c, FWGPU_ERR_INVALID, "sample too large (frames x channels > 2^40)");
    if (frames && !data) return fail(c, FWGPU_ERR_INVALID, "sample data is null");
    SampleRec r;
    r.alive = true;
    size_t bytes = (size_t)frames * channels * fmt_elem_size(format);
    if (on_device) {
        r.owned = false;
        r.d_data = (void*)data;
    } else {
        r.owned = true;
        HIPC(c, hipMalloc(&r.d_data, bytes + 256));  // slack: a wave's last dwordx4 may overhang the data
        hipError_t e = hipMemset((char*)r.d_data + bytes, 0, 256);
        // (like a plan build's uploads — fwgpu_plan_install.cpp, quiet_window —: in pieces, each when no process call is in flight)
        const bool live = audio_live(c);
        const size_t piece = live ? c->up_piece : (bytes ? bytes : 1);
        for (size_t off = 0; e == hipSuccess && off < bytes; off += piece) {
            if (live) quiet_window(c);
            e = hipMemcpy((char*)r.d_data + off, (const char*)data + off, std::min(piece, bytes - off), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            (void)hipFree(r.d_data);
            return hipfail(c, e, "sample upload");
        }
    }
    r.desc.data = r.d_data;
    r.desc.frames = frames;
    r.desc.channels = (int)channels;
    r.desc.format = format;
    // the data is in HBM; now the table entry — what a process call reads — under the gate: no process call is running, the
    // next one finds the whole entry (and the room for it on the device) in place
    // (a table that has to grow frees its old device copy: hipFree waits for every stream, so the resident realtime kernel is told
    //  to end BEFORE the gate is taken — behind the gate the audio thread could not ring it out, and the free would sit out its
    //  20 ms watchdog with the audio thread parked at its entry: ADVICE r3)
    const bool grows = (c->samples.size() + 1) * sizeof(SampleDesc) > c->d_samples.cap;
    std::unique_ptr<RtHold> hold(grows ? new RtHold(c) : nullptr);
    ControlGate gate(c);
    c->samples.push_back(r);
    c->sample_refs.push_back(0);
    rebuild_sample_table(c);  // copied to the device by the next process / update call; room for it is made here
    if (c->h_sample_tab.size() * sizeof(SampleDesc) > c->d_samples.cap) {
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, c->d_samples.ensure_n("d_samples", c->h_sample_tab.size() * 2 * sizeof(SampleDesc)));
    }
    return (int)c->samples.size() - 1;
}
int fwgpu_sample_create(fwgpu_ctx* c, int format, uint32_t channels, uint64_t frames, const void* data) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return sample_add(c, format, channels, frames, data, false);
}
int fwgpu_sample_create_device(fwgpu_ctx* c, int format, uint32_t channels, uint64_t frames, const void* device_data) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    return sample_add(c, format, channels, frames, device_data, true);
}
int fwgpu_sample_destroy(fwgpu_ctx* c, int sample) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (sample < 0 || sample >= (int)c->samples.size() || !c->samples[sample].alive)
        return fail(c, FWGPU_ERR_INVALID, "unknown sample id");
    // FIR / resampler nodes name their sample at construction and keep it for life: refuse while one exists.  Samplers
    // pick samples by message, which the host does not track: see the contract in fwgpu.h.
    for (const HostNode& n : c->graph.nodes)
        if (n.alive && (n.kind == K_FIR || n.kind == K_RESAMPLER) && n.init.sample == sample)
            return fail(c, FWGPU_ERR_INVALID, "sample is in use by a FIR / resampler node");
    use_device(c);
    RtHold hold(c);       // the resident realtime kernel ends now, and none is launched until the data is gone
    ControlGate gate(c);  // no process call runs while the entry is emptied and the data freed
    (void)rt_persist_stop(c);  // ... and no resident realtime kernel: between two callbacks it may be fetching the frames a voice reads next
    (void)hipStreamSynchronize(c->stream);
    if (c->ctl_stream) (void)hipStreamSynchronize(c->ctl_stream);
    SampleRec& r = c->samples[sample];
    if (r.owned && r.d_data) (void)hipFree(r.d_data);
    r.alive = false;
    r.d_data = nullptr;
    r.desc.data = nullptr;
    r.desc.frames = 0;
    rebuild_sample_table(c);
    return 0;
}

int fwgpu_poll_returned_samples(fwgpu_ctx* c, int64_t* nodes, int* samples, int cap) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (cap < 0 || (cap > 0 && (!nodes || !samples))) return fail(c, FWGPU_ERR_INVALID, "null output arrays");
    use_device(c);
    collect_returns(c);
    int n = 0;
    for (; n < cap && (size_t)n < c->ret_ready.size(); ++n) {
        nodes[n] = c->ret_ready[n].node;
        samples[n] = c->ret_ready[n].sample;
    }
    c->ret_ready.erase(c->ret_ready.begin(), c->ret_ready.begin() + n);
    return n;
}
int fwgpu_sample_retired(fwgpu_ctx* c, int sample) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (sample < 0 || sample >= (int)c->samples.size()) return fail(c, FWGPU_ERR_INVALID, "unknown sample id");
    use_device(c);
    collect_returns(c);
    for (const HostNode& n : c->graph.nodes)
        if (n.alive && (n.kind == K_FIR || n.kind == K_RESAMPLER) && n.init.sample == sample) return 0;
    return c->sample_refs[sample] == 0 ? 1 : 0;
}
int fwgpu_ext_pool_floats(fwgpu_ctx* c, uint64_t* in_use, uint64_t* capacity) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (in_use) *in_use = (uint64_t)c->ext_used;
    if (capacity) *capacity = (uint64_t)c->ext_cap;
    return 0;
}

int fwgpu_node_set_param(fwgpu_ctx* c, int64_t node, int param, float value, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    HostNode* n = c->graph.get(node);
    if (!n) return fail(c, FWGPU_ERR_INVALID, "unknown node id");
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    switch (n->kind) {
        case K_VOLUME:
        case K_SAMPLER:  // volume.rs:28-34, sampler.rs:171-177
            if (param != 0) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            m.type = CMD_SET_P0;
            m.f0 = percent_volume_to_raw_gain(value);
            return push_cmd(c, node, -1, m, false);
        case K_BEEP:  // beep_test.rs:30-32
            if (param != 0) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            m.type = CMD_SET_ENABLED;
            m.i0 = value != 0.0f;
            return push_cmd(c, node, -1, m, false);
        case K_PAN: {
            if (param != 0) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            float gl, gr;
            pan_to_gains(value, &gl, &gr);
            m.type = CMD_SET_P0;
            m.f0 = gl;
            int rc = push_cmd(c, node, -1, m, false);
            if (rc) return rc;
            m.type = CMD_SET_P1;
            m.f0 = gr;
            return push_cmd(c, node, -1, m, false);
        }
        case K_WIDTH:
            if (param != 0) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            m.type = CMD_SET_P0;
            m.f0 = fmaxf(value, 0.0f);
            return push_cmd(c, node, -1, m, false);
        case K_BIQUAD: {  // param 1 = cutoff_hz, 2 = Q: recompute the coefficients on the control side
            if (param != 1 && param != 2) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            if (param == 1) n->init.p0 = value;
            else n->init.p1 = value;
            float co[5];
            biquad_coefs(n->init.enabled, n->init.p0, n->init.p1, c->sample_rate, co);
            m.type = CMD_SET_COEFS;
            m.f0 = co[0];
            memcpy(&m.i0, &co[1], 4);
            memcpy(&m.i1, &co[2], 4);
            uint32_t lo, hi;
            memcpy(&lo, &co[3], 4);
            memcpy(&hi, &co[4], 4);
            uint64_t u = ((uint64_t)hi << 32) | lo;
            memcpy(&m.d0, &u, 8);
            return push_cmd(c, node, -1, m, false);
        }
        case K_DELAY: {  // param 1 = feedback, 2 = mix (the delay time is fixed at construction)
            if (param == 1) {
                m.type = CMD_SET_P0;
                m.f0 = fminf(fmaxf(value, 0.0f), 0.999f);
                return push_cmd(c, node, -1, m, false);
            }
            if (param != 2) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            float mix = fminf(fmaxf(value, 0.0f), 1.0f);
            m.type = CMD_SET_P1;
            m.f0 = mix;
            int rc = push_cmd(c, node, -1, m, false);
            if (rc) return rc;
            m.type = CMD_SET_GAIN;
            m.f0 = 1.0f - mix;
            return push_cmd(c, node, -1, m, false);
        }
        case K_RESAMPLER: {  // 1 = ratio (source frames per output frame), 3 = playing, 4 = seek to a source frame
            uint64_t u;
            if (param == 1) {
                m.type = CMD_RS_STEP;
                u = resampler_step(value);
                memcpy(&m.d0, &u, 8);
                return push_cmd(c, node, -1, m, false);
            }
            if (param == 3) {
                m.type = value != 0.0f ? CMD_SMP_PLAY : CMD_SMP_PAUSE;
                return push_cmd(c, node, -1, m, false);
            }
            if (param != 4) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            m.type = CMD_RS_SEEK;
            u = (uint64_t)fmaxf(value, 0.0f);
            memcpy(&m.d0, &u, 8);
            return push_cmd(c, node, -1, m, false);
        }
        case K_SPATIAL: {  // 0 / 1 / 2 = x / y / z of the source relative to the listener
            if (param < 0 || param > 2) return fail(c, FWGPU_ERR_INVALID, "unknown param");
            if (param == 0) n->init.phasor = value;
            else if (param == 1) n->init.phasor_inc = value;
            else n->init.gain = value;
            float gl, gr;
            int dl, dr;
            spatial_params(n->init.phasor, n->init.phasor_inc, n->init.gain, c->sample_rate, &gl, &gr, &dl, &dr);
            m.type = CMD_SET_P0;
            m.f0 = gl;
            int rc = push_cmd(c, node, -1, m, false);
            if (rc) return rc;
            m.type = CMD_SET_P1;
            m.f0 = gr;
            if ((rc = push_cmd(c, node, -1, m, false))) return rc;
            m.type = CMD_SP_ITD;
            m.i0 = dl;
            m.i1 = dr;
            return push_cmd(c, node, -1, m, false);
        }
        default:
            return fail(c, FWGPU_ERR_INVALID, "node kind has no runtime params");
    }
}
int fwgpu_node_set_params(fwgpu_ctx* c, uint32_t n, const int64_t* nodes, const int* params, const float* values, const uint32_t* at_blocks) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (n && (!nodes || !params || !values || !at_blocks)) return fail(c, FWGPU_ERR_INVALID, "null message list");
    for (uint32_t i = 0; i < n; ++i) {
        const int rc = fwgpu_node_set_param(c, nodes[i], params[i], values[i], at_blocks[i]);
        if (rc < 0) return rc;
    }
    return 0;
}
int fwgpu_sampler_set_sample(fwgpu_ctx* c, int64_t node, int sample, int stop_playback, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (sample < 0 || sample >= (int)c->samples.size() || !c->samples[sample].alive)
        return fail(c, FWGPU_ERR_INVALID, "unknown sample id");
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    m.type = CMD_SMP_SET_SAMPLE;
    m.i0 = sample;
    m.i1 = stop_playback != 0;
    int rc = push_cmd(c, node, K_SAMPLER, m, true);
    if (rc == 0) c->sample_refs[sample]++;  // the message carries the reference (an Arc clone in the reference: sampler.rs:67-79)
    return rc;
}
static int simple_msg(fwgpu_ctx* c, int64_t node, int type, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    m.type = type;
    return push_cmd(c, node, K_SAMPLER, m, true);
}
int fwgpu_sampler_play(fwgpu_ctx* c, int64_t node, uint32_t b) { return simple_msg(c, node, CMD_SMP_PLAY, b); }
int fwgpu_sampler_pause(fwgpu_ctx* c, int64_t node, uint32_t b) { return simple_msg(c, node, CMD_SMP_PAUSE, b); }
int fwgpu_sampler_stop(fwgpu_ctx* c, int64_t node, uint32_t b) { return simple_msg(c, node, CMD_SMP_STOP, b); }
int fwgpu_sampler_set_playhead_secs(fwgpu_ctx* c, int64_t node, double secs, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    m.type = CMD_SMP_SET_PLAYHEAD;
    m.d0 = secs;
    return push_cmd(c, node, K_SAMPLER, m, true);
}
int fwgpu_sampler_set_loop_range(fwgpu_ctx* c, int64_t node, int mode, double start, double end, uint32_t at_block) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (mode < 0 || mode > 2) return fail(c, FWGPU_ERR_INVALID, "loop mode must be 0, 1 or 2");
    Cmd m;
    memset(&m, 0, sizeof(m));
    m.block = at_block;
    m.type = CMD_SMP_SET_LOOP;
    m.i0 = mode;
    m.d0 = start;
    m.d1 = end;
    return push_cmd(c, node, K_SAMPLER, m, true);
}

static int process_interleaved_impl(fwgpu_ctx* c, const float* input, float* output, uint32_t n_in_ch, uint32_t n_out_ch,
                                    uint64_t frames, size_t out_bytes) {
    const float* d_in = nullptr;
    const size_t in_bytes_rt = (n_in_ch > 0 && input) ? (size_t)frames * n_in_ch * sizeof(float) : 0;
    if (c->h_rt_out && out_bytes <= RT_IO_BYTES && in_bytes_rt <= RT_IO_BYTES) {
        // realtime-sized call: graph inputs are read from, and the interleaved output written to, pinned host blocks
        // mapped into the device — the only wait is the stream sync (SURVEY 8(b) "realtime rules")
        if (in_bytes_rt) {
            memcpy(c->h_rt_in, input, in_bytes_rt);
            d_in = c->d_rt_in;
        }
        const bool spin = c->h_rt_flag != nullptr && !c->rt_use_graph;
        c->rt_signal_seq = spin ? ++c->rt_seq : 0;
        c->rt_signalled = false;
        int rc = run_blocks(c, frames, d_in, (int)n_in_ch, c->d_rt_out, (int)n_out_ch, true);
        c->rt_signal_seq = 0;
        if (rc) return rc;
        bool done = false;
        if (spin && c->rt_signalled) {
            // poll the completion flag (plain loads of pinned memory: no driver call, no sleep); a device that has not
            // answered after ~20 ms of spinning has a problem the stream sync below will name
            const unsigned long long want = c->rt_seq;
            volatile unsigned long long* flag = c->h_rt_flag;
            for (int attempt = 0; attempt < 2 && !done; ++attempt) {
                bool retry = false;
                // (bounded by the CLOCK, looked at every 1024 polls: an iteration count of `pause`s was ~1 s, not 20 ms — ADVICE r2)
                const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
                for (unsigned spins = 1;; ++spins) {
                    if (*flag == want) {
                        done = true;
                        break;
                    }
                    // the resident kernel's watchdog fired between two callbacks and the doorbell came too late for it: it says so
                    // (alive = 0, stored after its last completion flag) — this block has not been rendered and will not be
                    if (c->rtp.launched && (spins & 63u) == 0 && !*(volatile unsigned long long*)&c->h_rt_mb->alive) {
                        std::atomic_thread_fence(std::memory_order_acquire);
                        if (*flag != want) {
                            retry = true;
                            break;
                        }
                    }
#if defined(__x86_64__) || defined(__i386__)
                    __builtin_ia32_pause();
#endif
                    if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() > give_up) break;
                }
                if (!retry) break;
                rc = rt_block_relaunch(c, c->d_rt_out, want);  // (rare: a pause of the watchdog's length) the block as an ordinary launch
                if (rc) return rc;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!done) {
            if (c->rtp.launched) HIPC(c, hipStreamSynchronize(c->rt_stream));
            HIPC(c, hipStreamSynchronize(c->stream));
            // the streams are empty — which says the block was rendered only if its completion flag says so too: a resident kernel that
            // waited for another number has ended on its watchdog WITHOUT rendering this one.  One ordinary launch of the
            // block, then the flag must be there; stale audio is never copied out as if it were this block's.
            if (spin && c->rt_signalled && *(volatile unsigned long long*)c->h_rt_flag != c->rt_seq) {
                rc = rt_block_relaunch(c, c->d_rt_out, c->rt_seq);
                if (rc) return rc;
                HIPC(c, hipStreamSynchronize(c->stream));
                if (*(volatile unsigned long long*)c->h_rt_flag != c->rt_seq)
                    return fail(c, FWGPU_ERR_DEVICE, "realtime block was not rendered (completion flag missing after a stream sync)");
            }
        }
        c->ret_done_ticket.store(c->ret_ticket, std::memory_order_release);  // everything this call handed back is final
        if (out_bytes) memcpy(output, c->h_rt_out, out_bytes);
        return 0;
    }
    if (n_in_ch > 0 && input) {
        size_t in_bytes = (size_t)frames * n_in_ch * sizeof(float);
        if (in_bytes > c->d_in_stage.cap) {  // (first call of this size only)
            HIPC(c, hipStreamSynchronize(c->stream));
            HIPC(c, c->d_in_stage.ensure_n("d_in_stage", in_bytes));
        }
        HIPC(c, hipMemcpyAsync(c->d_in_stage.p, input, in_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = c->d_in_stage.as<float>();
    }
    if (out_bytes > c->d_out_stage.cap) {
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, c->d_out_stage.ensure_n("d_out_stage", out_bytes));
    }
    int rc = run_blocks(c, frames, d_in, (int)n_in_ch, c->d_out_stage.as<float>(), (int)n_out_ch);
    if (rc) return rc;
    if (out_bytes) HIPC(c, hipMemcpyAsync(output, c->d_out_stage.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    c->ret_done_ticket.store(c->ret_ticket, std::memory_order_release);
    return 0;
}

int fwgpu_process_interleaved(fwgpu_ctx* c, const float* input, float* output, uint32_t n_in_ch, uint32_t n_out_ch,
                              uint64_t frames, double stream_time_secs, uint32_t stream_status) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    AudioCallScope audio;
    use_device(c);
    if (n_in_ch > 64 || n_out_ch > 64) return fail(c, FWGPU_ERR_INVALID, "at most 64 stream channels per side (processor.rs:43-44)");
    if (frames > (1ull << 32)) return fail(c, FWGPU_ERR_INVALID, "more than 2^32 frames in one call");
    size_t out_bytes = (size_t)frames * n_out_ch * sizeof(float);
    if (out_bytes && !output) return fail(c, FWGPU_ERR_INVALID, "output is null");
    // ProcInfo::stream_time_secs / stream_status of this call (core/node.rs:111-118): no built-in node reads them; a
    // custom node mixed in through fwgpu_node_process gets them from fwgpu_proc_info
    AudioGate gate(c);  // (picks up a plan the control thread published since the last call: processor.rs:167-206)
    c->proc_stream_time = stream_time_secs;
    c->proc_stream_status = stream_status;
    if (stream_status & 2u) c->n_underflows++;
    if (stream_status & 1u) c->n_overflows++;
    if (!c->have_plan || frames == 0) {  // processor.rs:86-89 (Q19)
        if (out_bytes) memset(output, 0, out_bytes);
        return 0;
    }
    const int rc = process_interleaved_impl(c, input, output, n_in_ch, n_out_ch, frames, out_bytes);
    // "all output buffers MUST be filled" (core/node.rs:41-42): whatever failed — a launch, the wait for the stream, the
    // copy back — the caller's buffer never keeps what it held before the call
    if (rc != 0 && out_bytes) memset(output, 0, out_bytes);
    return rc;
}
// ---- the call split in two (include/fwgpu.h): render n + 1 while call n's output crosses PCIe into the caller's buffer
// (Round 5, measured on config 2, 1.5 MiB per call: staging in hipHostMalloc memory + memcpy in `end`: 0.67 ms per step against 0.345 for
//  the synchronous call — the host's memcpy out of that memory; the copy issued by `end` straight into the caller's pageable buffer:
//  0.318 against 0.312 — a blit kernel that queues behind the next ticket's kernels, no overlap.  Hence registered ordinary memory.)
int64_t fwgpu_process_interleaved_begin(fwgpu_ctx* c, const float* input, uint32_t n_in_ch, uint32_t n_out_ch, uint64_t frames,
                                        double stream_time_secs, uint32_t stream_status) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    AudioCallScope audio;
    use_device(c);
    if (n_in_ch > 64 || n_out_ch > 64) return fail(c, FWGPU_ERR_INVALID, "at most 64 stream channels per side (processor.rs:43-44)");
    if (frames > (1ull << 32)) return fail(c, FWGPU_ERR_INVALID, "more than 2^32 frames in one call");
    fwgpu_ctx::AsyncOut& a = c->ao;
    const int slot = (int)(a.next & 1);
    if (a.busy[slot]) return fail(c, FWGPU_ERR_INVALID, "two calls in flight already: end the older ticket first");
    const size_t out_bytes = (size_t)frames * n_out_ch * sizeof(float);
    AudioGate gate(c);
    c->proc_stream_time = stream_time_secs;
    c->proc_stream_status = stream_status;
    if (stream_status & 2u) c->n_underflows++;
    if (stream_status & 1u) c->n_overflows++;
    a.bytes[slot] = out_bytes;
    a.zeros[slot] = !c->have_plan || frames == 0;  // processor.rs:86-89 (Q19): no schedule yet -> silence
    if (!a.zeros[slot]) {
        if (!a.copy_stream) HIPC(c, hipStreamCreateWithFlags(&a.copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            if (!a.ev_render[i]) HIPC(c, hipEventCreateWithFlags(&a.ev_render[i], hipEventDisableTiming));
            if (!a.ev_copy[i]) HIPC(c, hipEventCreateWithFlags(&a.ev_copy[i], hipEventDisableTiming));
        }
        // (the device-side output block exists only in the copy modes: in the default mode 3 nothing ever wrote it — ADVICE r5)
        static const int mode = getenv("FWGPU_ASYNC_MODE") ? atoi(getenv("FWGPU_ASYNC_MODE")) : 3;
        if ((mode != 3 && out_bytes > a.d[slot].cap) || out_bytes > a.hcap[slot]) {  // (first call of this size only; the slot is idle: nothing reads it)
            RtHold hold(c);
            HIPC(c, hipStreamSynchronize(c->stream));
            if (mode != 3) HIPC(c, a.d[slot].ensure_n("d_async_out", out_bytes));
            if (out_bytes > a.hcap[slot]) {
                // ORDINARY (cacheable) host memory, page-locked for the DMA engine: the copy back is an SDMA transfer that runs beside the
                // next ticket's kernels (a copy into pageable memory is a blit KERNEL: it queued behind them — measured, no overlap), and
                // the host's memcpy out of it runs at memory speed (out of hipHostMalloc memory it cost 0.3 ms per 1.5 MiB — measured)
                if (a.h[slot]) {
                    (void)hipHostUnregister(a.h[slot]);
                    free(a.h[slot]);
                }
                a.h[slot] = nullptr;
                a.hcap[slot] = 0;
                const size_t cap = (out_bytes + 4095) & ~(size_t)4095;
                void* m = nullptr;
                if (posix_memalign(&m, 4096, cap) != 0 || !m) return fail(c, FWGPU_ERR_DEVICE, "out of host memory for the output staging");
                memset(m, 0, cap);
                hipError_t e = hipHostRegister(m, cap, hipHostRegisterMapped);
                void* dp = nullptr;
                if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, m, 0);
                if (e != hipSuccess) {
                    free(m);
                    return hipfail(c, e, "hipHostRegister(output staging)");
                }
                a.h[slot] = m;
                a.h_dev[slot] = (float*)dp;
                a.hcap[slot] = cap;
            }
        }
        const float* d_in = nullptr;
        if (n_in_ch > 0 && input) {
            const size_t in_bytes = (size_t)frames * n_in_ch * sizeof(float);
            if (in_bytes > c->d_in_stage.cap) {
                HIPC(c, hipStreamSynchronize(c->stream));
                HIPC(c, c->d_in_stage.ensure_n("d_in_stage", in_bytes));
            }
            HIPC(c, hipMemcpyAsync(c->d_in_stage.p, input, in_bytes, hipMemcpyHostToDevice, c->stream));
            d_in = c->d_in_stage.as<float>();
        }
        // Where the frames cross PCIe (round 5, config 2, 1.5 MiB per call, same box; the synchronous call: 0.313 ms per step):
        //   3 (default)  the graph-output kernel writes the interleaved frames STRAIGHT into the mapped host staging: no copy at all
        //   2            a copy in the ctx stream behind the render: 0.310 (DMA engine) / 0.325 (blit kernel) — serial with the next render
        //   0            a copy on a stream of its own behind the render's event: 0.579 (DMA engine) / 0.388 (blit kernel, HSA_ENABLE_SDMA=0):
        //                the cross-stream hand-over costs more than the overlap gives
        float* const d_out = mode == 3 ? a.h_dev[slot] : a.d[slot].as<float>();
        const int rc = run_blocks(c, frames, d_in, (int)n_in_ch, d_out, (int)n_out_ch);
        if (rc) return rc;
        if (out_bytes) {
            if (mode == 3) {
                HIPC(c, hipEventRecord(a.ev_copy[slot], c->stream));
            } else if (mode == 2) {
                HIPC(c, hipMemcpyAsync(a.h[slot], a.d[slot].p, out_bytes, hipMemcpyDeviceToHost, c->stream));
                HIPC(c, hipEventRecord(a.ev_copy[slot], c->stream));
            } else {
                HIPC(c, hipEventRecord(a.ev_render[slot], c->stream));
                HIPC(c, hipStreamWaitEvent(a.copy_stream, a.ev_render[slot], 0));
                HIPC(c, hipMemcpyAsync(a.h[slot], a.d[slot].p, out_bytes, hipMemcpyDeviceToHost, a.copy_stream));
                HIPC(c, hipEventRecord(a.ev_copy[slot], a.copy_stream));
            }
        }
    }
    a.ret_ticket[slot] = c->ret_ticket;
    a.busy[slot] = true;
    return a.next++;
}
int fwgpu_process_interleaved_end(fwgpu_ctx* c, int64_t ticket, float* output) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    AudioCallScope audio;
    use_device(c);
    fwgpu_ctx::AsyncOut& a = c->ao;
    const int slot = (int)(ticket & 1);
    if (ticket < 0 || ticket != a.done || ticket >= a.next || !a.busy[slot])
        return fail(c, FWGPU_ERR_INVALID, "not the oldest ticket in flight (tickets are ended in the order they were begun)");
    const size_t n = a.bytes[slot];
    if (n && !output) return fail(c, FWGPU_ERR_INVALID, "output is null (the ticket stays in flight: end it with a buffer, or fwgpu_process_interleaved_cancel)");
    a.busy[slot] = false;
    a.done++;
    if (a.zeros[slot] || n == 0) {
        if (n) memset(output, 0, n);
        return 0;
    }
    // the copy back went out in `begin`, on its own stream behind the ticket's last launch: the ctx stream — which may be rendering
    // the NEXT ticket — is not waited for
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipEventSynchronize(a.ev_copy[slot]);
    if (e != hipSuccess) {
        memset(output, 0, n);  // "all output buffers MUST be filled" (core/node.rs:41-42)
        return hipfail(c, e, "copy back of a process_interleaved_begin ticket");
    }
    const auto t1 = std::chrono::steady_clock::now();
    memcpy(output, a.h[slot], n);
    a.prof_wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
    a.prof_copy_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count();
    a.prof_calls++;
    c->ret_done_ticket.store(a.ret_ticket[slot], std::memory_order_release);  // what that call handed back is final
    return 0;
}
// Abandon tickets: every ticket in flight up to and including `ticket` is waited for and its slot released, no frames copied — a host
// that unwinds (a Rust `Ticket` dropped on an error path) must not leave `begin` refusing for ever after two.
int fwgpu_process_interleaved_cancel(fwgpu_ctx* c, int64_t ticket) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    AudioCallScope audio;
    use_device(c);
    fwgpu_ctx::AsyncOut& a = c->ao;
    if (ticket < a.done || ticket >= a.next) return fail(c, FWGPU_ERR_INVALID, "no such ticket in flight");
    int rc = 0;
    while (a.done <= ticket) {
        const int slot = (int)(a.done & 1);
        if (a.busy[slot] && !a.zeros[slot] && a.bytes[slot]) {
            const hipError_t e = hipEventSynchronize(a.ev_copy[slot]);  // the staging block is idle again only once its writer has ended
            if (e != hipSuccess) rc = hipfail(c, e, "waiting for a cancelled process_interleaved_begin ticket");
            c->ret_done_ticket.store(a.ret_ticket[slot], std::memory_order_release);
        }
        a.busy[slot] = false;
        a.done++;
    }
    return rc;
}
int fwgpu_proc_info(fwgpu_ctx* c, double* stream_time_secs, uint32_t* stream_status, uint64_t* output_underflows,
                    uint64_t* input_overflows) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (stream_time_secs) *stream_time_secs = c->proc_stream_time;
    if (stream_status) *stream_status = c->proc_stream_status;
    if (output_underflows) *output_underflows = c->n_underflows;
    if (input_overflows) *input_overflows = c->n_overflows;
    return 0;
}

// ---- headless stream: DataCallback::callback (cpal/lib.rs:378-449) without a device.  The caller plays the part of
// cpal: it names the instant of each callback on its own clock; the stream-time / underflow bookkeeping is the reference's.
struct fwgpu_stream {
    fwgpu_ctx* ctx;
    uint32_t n_in, n_out;
    double sample_rate_recip;      // :371
    bool have_first_instant;       // first_stream_instant.is_some()
    double first_stream_instant;
    double predicted_stream_secs;  // :373
    bool is_first_callback;        // :374
    uint64_t callbacks, underflows;
    double last_stream_time;
};
fwgpu_stream* fwgpu_stream_open(fwgpu_ctx* c, uint32_t num_in_channels, uint32_t num_out_channels) {
    if (!c || num_in_channels > 64 || num_out_channels > 64) return nullptr;
    fwgpu_stream* s = new (std::nothrow) fwgpu_stream;
    if (!s) return nullptr;
    s->ctx = c;
    s->n_in = num_in_channels;
    s->n_out = num_out_channels;
    s->sample_rate_recip = 1.0 / (double)c->sample_rate;
    s->have_first_instant = false;
    s->first_stream_instant = 0.0;
    s->predicted_stream_secs = 1.0;
    s->is_first_callback = true;
    s->callbacks = s->underflows = 0;
    s->last_stream_time = 0.0;
    return s;
}
void fwgpu_stream_close(fwgpu_stream* s) { delete s; }
int fwgpu_stream_callback(fwgpu_stream* s, float* output, uint64_t frames, double callback_instant_secs) {
    if (!s) return FWGPU_ERR_INVALID;
    double stream_time_secs;
    bool underflow = false;
    const double block_secs = (double)frames * s->sample_rate_recip;
    if (s->is_first_callback) {  // :393-400 (the first callback's instant is ignored)
        s->is_first_callback = false;
        s->predicted_stream_secs = block_secs;
        stream_time_secs = 0.0;
    } else if (s->have_first_instant) {  // :401-420
        stream_time_secs = callback_instant_secs - s->first_stream_instant;
        underflow = stream_time_secs > s->predicted_stream_secs;
        s->predicted_stream_secs = stream_time_secs + (block_secs * 1.2);
    } else {  // :421-426
        s->have_first_instant = true;
        s->first_stream_instant = callback_instant_secs;
        stream_time_secs = s->predicted_stream_secs;
        s->predicted_stream_secs += block_secs * 1.2;
    }
    s->callbacks++;
    if (underflow) s->underflows++;
    s->last_stream_time = stream_time_secs;
    const uint32_t status = underflow ? 2u : 0u;  // StreamStatus::OUTPUT_UNDERFLOW (core/node.rs:130)
    // :434-442 process_interleaved(&[], output, ...); no processor yet (:446-449) = no schedule: output.fill(0.0) — which
    // fwgpu_process_interleaved does itself (Q19)
    const int rc = fwgpu_process_interleaved(s->ctx, nullptr, output, s->n_in, s->n_out, frames, stream_time_secs, status);
    return rc < 0 ? rc : (int)status;
}
int fwgpu_stream_run(fwgpu_stream* s, float* output, uint64_t frames, uint32_t n_callbacks, double first_instant_secs,
                     double* elapsed_secs) {
    if (!s || !output) return FWGPU_ERR_INVALID;
    const double period = (double)frames * s->sample_rate_recip;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = 0;
    for (uint32_t i = 0; i < n_callbacks; ++i) {
        rc = fwgpu_stream_callback(s, output, frames, first_instant_secs + (double)i * period);
        if (rc < 0) break;
    }
    if (elapsed_secs) *elapsed_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc < 0 ? rc : 0;
}
int fwgpu_stream_stats(fwgpu_stream* s, uint64_t* callbacks, uint64_t* underflows, double* last_stream_time_secs) {
    if (!s) return FWGPU_ERR_INVALID;
    if (callbacks) *callbacks = s->callbacks;
    if (underflows) *underflows = s->underflows;
    if (last_stream_time_secs) *last_stream_time_secs = s->last_stream_time;
    return 0;
}

int fwgpu_process_blocks_device_flags(fwgpu_ctx* c, uint32_t num_blocks, float* d_output, uint32_t n_out_ch, uint8_t* d_silence) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    AudioCallScope audio;
    use_device(c);
    AudioGate gate(c);
    if (!c->have_plan) return fail(c, FWGPU_ERR_INVALID, "no schedule: call fwgpu_update first");
    if (num_blocks == 0) return 0;
    if (n_out_ch > 64 || (n_out_ch && !d_output)) return fail(c, FWGPU_ERR_INVALID, "bad output (null, or more than 64 channels)");
    c->out_sil = d_silence;
    const int rc = run_blocks(c, (uint64_t)num_blocks * c->mbf, nullptr, 0, d_output, (int)n_out_ch);
    c->out_sil = nullptr;
    return rc;
}
int fwgpu_process_blocks_device(fwgpu_ctx* c, uint32_t num_blocks, float* d_output, uint32_t n_out_ch) {
    return fwgpu_process_blocks_device_flags(c, num_blocks, d_output, n_out_ch, nullptr);
}
int fwgpu_process_blocks_device_io(fwgpu_ctx* c, uint32_t num_blocks, const float* d_input, uint32_t n_in_ch, float* d_output, uint32_t n_out_ch,
                                   uint8_t* d_silence) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    AudioCallScope audio;
    use_device(c);
    AudioGate gate(c);
    if (!c->have_plan) return fail(c, FWGPU_ERR_INVALID, "no schedule: call fwgpu_update first");
    if (num_blocks == 0) return 0;
    if (n_in_ch > 64 || n_out_ch > 64 || (n_out_ch && !d_output)) return fail(c, FWGPU_ERR_INVALID, "bad stream buffers (null output, or more than 64 channels)");
    c->out_sil = d_silence;
    const int rc = run_blocks(c, (uint64_t)num_blocks * c->mbf, (d_input && n_in_ch) ? d_input : nullptr, (d_input && n_in_ch) ? (int)n_in_ch : 0, d_output,
                              (int)n_out_ch);
    c->out_sil = nullptr;
    return rc;
}

int fwgpu_bus_sum_ordered_flags(fwgpu_ctx* c, const float* const* d_parts, const uint8_t* const* d_silence, uint32_t n_parts, float* d_out,
                                uint8_t* d_out_silence, uint64_t n_floats, uint32_t frames_per_block, uint32_t n_channels) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    AudioCallScope audio;
    use_device(c);
    if (n_parts == 0 || n_parts > FW_MAX_BUS_PARTS || !d_parts || !d_out) return fail(c, FWGPU_ERR_INVALID, "1..64 partial buses");
    BusParts bp;
    bp.n = (int)n_parts;
    for (uint32_t r = 0; r < n_parts; ++r) {
        if (!d_parts[r] || ((uintptr_t)d_parts[r] & 15u)) return fail(c, FWGPU_ERR_INVALID, "partial bus null or not 16-byte aligned");
        bp.part[r] = d_parts[r];
    }
    for (uint32_t r = n_parts; r < FW_MAX_BUS_PARTS; ++r) bp.part[r] = nullptr;
    if ((uintptr_t)d_out & 15u) return fail(c, FWGPU_ERR_INVALID, "output not 16-byte aligned");
    uint32_t n_blocks = 0;
    if (d_silence) {
        if (frames_per_block == 0 || n_channels == 0) return fail(c, FWGPU_ERR_INVALID, "silence flags need the block geometry (frames, channels)");
        const uint64_t per = (uint64_t)frames_per_block * n_channels;
        n_blocks = (uint32_t)((n_floats + per - 1) / per);
    }
    LCHK(c, launch_bus_sum_ordered(c->stream, bp, d_silence, d_out, d_out_silence, (size_t)n_floats, n_blocks, frames_per_block, n_channels));
    return 0;
}
int fwgpu_bus_sum_ordered(fwgpu_ctx* c, const float* const* d_parts, uint32_t n_parts, float* d_out, uint64_t n_floats) {
    return fwgpu_bus_sum_ordered_flags(c, d_parts, nullptr, n_parts, d_out, nullptr, n_floats, 0, 0);
}

int fwgpu_synchronize(fwgpu_ctx* c) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int fwgpu_node_process(fwgpu_ctx* c, int64_t node, uint64_t frames, const float* const* inputs, uint32_t n_in,
                       float* const* outputs, uint32_t n_out, uint64_t in_mask, uint64_t* out_mask, double, uint32_t) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    AudioCallScope audio;
    use_device(c);
    AudioGate gate(c);
    { const int prc = rt_persist_stop(c); if (prc) return prc; }
    { const int prc = lazy_flush(c); if (prc) return prc; }  // (this node's state may lag behind lazily rendered blocks)
    // the node as the ACTIVE plan knows it (the graph belongs to the control thread, which may be editing it right now:
    // Firewheel activates and deactivates nodes while the audio thread is inside process(), graph.rs:586-612)
    const uint32_t nslot = (uint32_t)(node & 0xffffffff);
    const int pidx = (c->have_plan && nslot < c->slot_index.size()) ? c->slot_index[nslot] : -1;
    if (pidx < 0 || nslot >= c->slot_ids.size() || c->slot_ids[nslot] != node)
        return fail(c, FWGPU_ERR_INVALID, "node is not activated (call fwgpu_update)");
    const PlanNode* hn = &c->plan.nodes[pidx];
    if ((uint32_t)hn->n_in != n_in || (uint32_t)hn->n_out != n_out) return fail(c, FWGPU_ERR_INVALID, "port counts differ from add_node");
    if (frames > c->mbf) return fail(c, FWGPU_ERR_INVALID, "frames > max_block_frames");
    if (hn->kind == K_FIR) return fail(c, FWGPU_ERR_INVALID, "FIR banks run at graph level (fwgpu_process_interleaved), not per node");
    if (hn->kind == K_HOST) return fail(c, FWGPU_ERR_INVALID, "a host node's processor lives in the caller: call it there");
    if (n_in + n_out == 0) return fail(c, FWGPU_ERR_INVALID, "node has no ports");
    if (frames == 0) {  // the reference never calls a node with an empty block (processor.rs:86-89 returns first): nothing to do
        if (out_mask) *out_mask = 0;
        return 0;
    }
    if ((n_in && !inputs) || (n_out && !outputs)) return fail(c, FWGPU_ERR_INVALID, "null channel table");
    for (uint32_t i = 0; frames && i < n_in; ++i)
        if (!inputs[i]) return fail(c, FWGPU_ERR_INVALID, "null input channel");
    for (uint32_t i = 0; frames && i < n_out; ++i)
        if (!outputs[i]) return fail(c, FWGPU_ERR_INVALID, "null output channel");
    const size_t stride = (size_t)c->stride;
    const int nb = 1 + (int)n_in + (int)n_out;
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, c->d_scratch_pool.ensure_n("d_scratch_pool", (size_t)nb * stride * sizeof(float)));
    HIPC(c, c->d_scratch_flags.ensure_n("d_scratch_flags", (size_t)nb));
    HIPC(c, hipMemsetAsync(c->d_scratch_pool.p, 0, stride * sizeof(float), c->stream));
    uint8_t fl[1 + 64 + 64] = {0};  // (at most 64 ports per side)
    fl[0] = 1;
    for (uint32_t i = 0; i < n_in; ++i) {
        fl[1 + i] = (in_mask >> i) & 1ull;
        HIPC(c, hipMemcpyAsync(c->d_scratch_pool.as<float>() + (1 + i) * stride, inputs[i], frames * sizeof(float),
                               hipMemcpyHostToDevice, c->stream));
    }
    for (uint32_t i = 0; i < n_out; ++i)  // nodes that leave outputs untouched (dummy.rs, beep_test.rs:83-86)
        HIPC(c, hipMemcpyAsync(c->d_scratch_pool.as<float>() + (1 + n_in + i) * stride, outputs[i], frames * sizeof(float),
                               hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(c->d_scratch_flags.p, fl, nb, hipMemcpyHostToDevice, c->stream));
    // temp tables: [NodeDesc][in ids][out ids]
    int tab[sizeof(NodeDesc) / sizeof(int) + 64 + 64 + 2] = {0};
    const size_t tab_ints = sizeof(NodeDesc) / sizeof(int) + n_in + n_out + 2;
    NodeDesc nd;
    memset(&nd, 0, sizeof(nd));
    nd.kind = hn->kind;
    nd.n_in = (int)n_in;
    nd.n_out = (int)n_out;
    nd.in_off = 0;
    nd.out_off = 0;
    nd.state = (int)(node & 0xffffffff);
    nd.aux0 = (hn->kind == K_SUM && n_out) ? (int)(n_in / n_out) : 0;
    memcpy(tab, &nd, sizeof(nd));
    int* ins = tab + sizeof(NodeDesc) / sizeof(int);
    int* outs = ins + n_in + 1;
    for (uint32_t i = 0; i < n_in; ++i) ins[i] = 1 + (int)i;
    for (uint32_t i = 0; i < n_out; ++i) outs[i] = 1 + (int)n_in + (int)i;
    HIPC(c, c->d_scratch_tab.ensure_n("d_scratch_tab", tab_ints * sizeof(int)));
    HIPC(c, hipMemcpyAsync(c->d_scratch_tab.p, tab, tab_ints * sizeof(int), hipMemcpyHostToDevice, c->stream));
    int rc = upload_sample_table(c);
    if (rc) return rc;
    if ((rc = join_streams(c))) return rc;  // (a control-ahead call may still be in flight: one node, the main stream)
    c->cmds_on_ctl = false;
    rc = upload_cmds(c);
    if (rc) return rc;
    DevView v = generic_view(c, (int)frames);
    v.nodes = (const NodeDesc*)c->d_scratch_tab.p;
    v.in_buf = c->d_scratch_tab.as<int>() + sizeof(NodeDesc) / sizeof(int);
    v.out_buf = v.in_buf + n_in + 1;
    v.pool = c->d_scratch_pool.as<float>();
    v.flags = c->d_scratch_flags.as<uint8_t>();
    c->epoch++;
    LCHK(c, launch_single_node(c->stream, v, 0));
    for (uint32_t i = 0; i < n_out; ++i)
        HIPC(c, hipMemcpyAsync(outputs[i], c->d_scratch_pool.as<float>() + (1 + n_in + i) * stride, frames * sizeof(float),
                               hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(fl, c->d_scratch_flags.p, nb, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    uint64_t om = 0;
    for (uint32_t i = 0; i < n_out; ++i)
        if (fl[1 + n_in + i]) om |= 1ull << i;
    if (out_mask) *out_mask = om;
    retire_cmds_node(c, (int)(node & 0xffffffff));
    return 0;
}

int fwgpu_timing_enable(fwgpu_ctx* c, int on) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    c->timing = on != 0;
    return 0;
}
int fwgpu_timing_read(fwgpu_ctx* c, int which, double* total_ms, uint64_t* launches) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    if (which < 0 || which > 4) return fail(c, FWGPU_ERR_INVALID, "timer index");
    timer_drain(c);
    *total_ms = c->timers[which].acc_ms;
    *launches = c->timers[which].launches;
    return 0;
}
int fwgpu_timing_reset(fwgpu_ctx* c) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    timer_drain(c);
    for (TimerCat& t : c->timers) {
        t.acc_ms = 0.0;
        t.launches = 0;
    }
    return 0;
}
#ifdef FW_DEBUG_BUS
// debugging builds only (make EXTRA=-DFW_DEBUG_BUS; scripts/dbg_rt.py — how round 4 found the leaf-bus store hazard): the fused plan's
// mix buses of block 0 of the last call, [n_bus][stride] floats
int fwgpu_debug_read_bus(fwgpu_ctx* c, float* out, uint64_t max_floats) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    (void)rt_persist_stop(c);
    HIPC(c, hipStreamSynchronize(c->stream));
    if (!c->d_bus.p) return fail(c, FWGPU_ERR_INVALID, "no fused plan");
    const uint64_t n = std::min<uint64_t>(max_floats, (uint64_t)c->n_bus * c->stride);
    HIPC(c, hipMemcpy(out, c->d_bus.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return (int)c->n_bus;
}
#endif
#ifdef FW_CHAIN_TRACE
// profiling builds only (scripts/chain_trace.py): timestamps [step 0..63][wave 0..15][slot 0..7] of workgroup 0
int fwgpu_debug_read_trace(fwgpu_ctx* c, unsigned long long* out) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    HIPC(c, hipStreamSynchronize(c->stream));
    if (!c->d_trace.p) return fail(c, FWGPU_ERR_INVALID, "no trace");
    HIPC(c, hipMemcpy(out, c->d_trace.p, 64 * 16 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif

int fwgpu_device_info(fwgpu_ctx* c, char* name, int name_cap, int* cus, uint64_t* hbm) {
    NEED_CTX(c, FWGPU_ERR_INVALID);
    hipDeviceProp_t prop;
    HIPC(c, hipGetDeviceProperties(&prop, c->device));
    if (name && name_cap > 0) {
        strncpy(name, prop.name[0] ? prop.name : prop.gcnArchName, (size_t)name_cap - 1);  // no marketing name: the ISA
        name[name_cap - 1] = 0;
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = (uint64_t)prop.totalGlobalMem;
    return 0;
}

}  // extern "C"

#ifdef FW_PROBE
// scripts/placement_probe.py builds only (make EXTRA=-DFW_PROBE): exchange one device table between two contexts that hold
// the same graph — which table's place in HBM carries k_leaf_sum's slow / fast state?  Never part of a shipped library.
extern "C" int fwgpu_probe_swap(fwgpu_ctx* a, fwgpu_ctx* b, int which) {
    auto sw = [](DevBuf& x, DevBuf& y) {
        std::swap(x.p, y.p);
        std::swap(x.cap, y.cap);
    };
    switch (which) {
        case 0: sw(a->d_bus, b->d_bus); sw(a->d_bus_flags, b->d_bus_flags); break;
        case 1: sw(a->d_refs, b->d_refs); break;
        case 2: sw(a->d_gsets, b->d_gsets); break;
        case 3: sw(a->d_leaves, b->d_leaves); break;
        case 4: sw(a->d_samples, b->d_samples); break;
        case 5: sw(a->d_voices, b->d_voices); sw(a->d_blks, b->d_blks); sw(a->d_cache, b->d_cache); break;
        case 6: sw(a->d_ramps, b->d_ramps); break;
        default: return -1;
    }
    return 0;
}
#endif
