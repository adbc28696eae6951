// fwgpu_run.cpp — per-call work of the device-resident FirewheelProcessor (graph/processor.rs:61-165): message upload and
// retirement, the kernel sequence of each launch plan, HIP-event timing.
#include "fwgpu_ctx.h"

#include <chrono>

namespace fwgpu {

namespace {
thread_local int t_audio_depth = 0;  // > 0 while this thread is inside a process entry point
void set_error(fwgpu_ctx* c, const char* a, const char* b) {
    const bool audio = t_audio_depth > 0;
    char* buf = audio ? c->err_audio : c->err_ctl;
    const size_t cap = sizeof(c->err_ctl);
    size_t n = 0;
    for (; a && *a && n + 1 < cap; ++a) buf[n++] = *a;
    if (b) {
        if (n + 2 < cap) {
            buf[n++] = ':';
            buf[n++] = ' ';
        }
        for (; *b && n + 1 < cap; ++b) buf[n++] = *b;
    }
    buf[n] = 0;
    c->err_last.store(audio ? 1 : 0, std::memory_order_release);
}
}  // namespace
AudioCallScope::AudioCallScope() { ++t_audio_depth; }
AudioCallScope::~AudioCallScope() { --t_audio_depth; }

int fail(fwgpu_ctx* c, int code, const char* msg) {
    set_error(c, msg, nullptr);
    return code;
}
int hipfail(fwgpu_ctx* c, hipError_t e, const char* what) {
    set_error(c, what, hipGetErrorString(e));
    return FWGPU_ERR_DEVICE;
}

int upload(fwgpu_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    HIPC(c, b.ensure_n("b", bytes));
    if (bytes) HIPC(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

// The host copy of the table (h_sample_tab) and the room for it on the device are kept up to date by the calls that
// change it (fwgpu_sample_create / _destroy: control calls); the process call that follows only copies.
int join_streams(fwgpu_ctx* c) {
    if (!c->streams_split) return 0;
    HIPC(c, hipEventRecord(c->ev_join, c->ctl_stream));
    HIPC(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    c->streams_split = false;
    c->ahead_seq = 0;  // (the render events of earlier batches are behind everything the main stream will do next)
    return 0;
}

int upload_sample_table(fwgpu_ctx* c) {
    if (!c->samples_dirty) return 0;
    c->samples_dirty = false;
    c->epoch++;  // cached steady descriptors hold sample indices / sizes
    HIPC(c, hipStreamSynchronize(c->stream));
    if (c->ctl_stream) HIPC(c, hipStreamSynchronize(c->ctl_stream));
    const size_t bytes = c->h_sample_tab.size() * sizeof(SampleDesc);
    HIPC(c, c->d_samples.ensure_n("d_samples", bytes));  // (already large enough: sized where the table changed)
    HIPC(c, hipMemcpy(c->d_samples.p, c->h_sample_tab.data(), bytes, hipMemcpyHostToDevice));
    return 0;
}

// ---------------------------------------------------------------- messages
namespace {
inline bool cmd_less(const Cmd& a, const Cmd& b) { return a.state != b.state ? a.state < b.state : a.block < b.block; }
// stable bottom-up merge sort of a[0, n) through `tmp` (same capacity): no allocation, arrival order kept inside (node, block)
void merge_sort_cmds(Cmd* a, size_t n, Cmd* tmp) {
    if (n < 2) return;
    for (size_t lo = 0; lo < n; lo += 8) {  // insertion-sorted runs of 8
        const size_t hi = std::min(lo + 8, n);
        for (size_t i = lo + 1; i < hi; ++i) {
            Cmd k = a[i];
            size_t j = i;
            for (; j > lo && cmd_less(k, a[j - 1]); --j) a[j] = a[j - 1];
            a[j] = k;
        }
    }
    Cmd *src = a, *dst = tmp;
    for (size_t w = 8; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            const size_t mid = std::min(lo + w, n), hi = std::min(lo + 2 * w, n);
            std::merge(src + lo, src + mid, src + mid, src + hi, dst + lo, cmd_less);  // stable: ties from the left run
        }
        std::swap(src, dst);
    }
    if (src != a) memcpy(a, src, n * sizeof(Cmd));
}
}  // namespace

// ring -> cmds.  `cmds` stays sorted by (node, block) with arrival order inside: what was kept from earlier calls is
// sorted already (retire_cmds shifts every block by the same amount), the new messages are sorted among themselves and
// merged in behind the older ones.
void drain_ring(fwgpu_ctx* c) {
    const size_t n0 = c->cmds.size();
    Cmd m;
    while (c->cmds.size() < fwgpu_ctx::CMD_CAP && c->ring.pop(m)) c->cmds.push_back(m);  // (capacity reserved: no growth)
    c->drain_epoch.fetch_add(1, std::memory_order_release);
    const size_t n = c->cmds.size();
    if (n == n0) return;
    c->cmds_scratch.resize(n);  // within the reserved capacity
    merge_sort_cmds(c->cmds.data() + n0, n - n0, c->cmds_scratch.data());
    if (n0) {
        std::merge(c->cmds.begin(), c->cmds.begin() + n0, c->cmds.begin() + n0, c->cmds.end(), c->cmds_scratch.begin(), cmd_less);
        memcpy(c->cmds.data(), c->cmds_scratch.data(), n * sizeof(Cmd));
    }
}

int upload_cmds(fwgpu_ctx* c, bool drained) {
    if (!drained) drain_ring(c);
    c->n_cmds_dev = (int)c->cmds.size();
    // the voices these messages go to, then the ones of the call before (a glide started there may still be running)
    c->hot_now.clear();
    size_t n_now = 0;
    if (!c->slot_voice.empty() && c->h_ctl_order) {
        for (const Cmd& m : c->cmds) {
            if (m.state < 0 || (size_t)m.state >= c->slot_voice.size()) continue;
            const int v = c->slot_voice[m.state];
            if (v >= 0 && !c->ctl_mark[v]) {
                c->ctl_mark[v] = 1;
                c->hot_now.push_back(v);
            }
        }
        n_now = c->hot_now.size();
        for (const int v : c->hot_prev)
            if (!c->ctl_mark[v]) {
                c->ctl_mark[v] = 1;
                c->hot_now.push_back(v);
            }
    }
    const bool order = !c->hot_now.empty();
    c->ctl_order_live = order;
    if (c->n_cmds_dev == 0 && !order) {
        c->hot_prev.clear();
        return 0;
    }
    size_t bytes = c->cmds.size() * sizeof(Cmd);
    HIPC(c, hipEventSynchronize(c->cmds_copied));  // the previous upload has left the pinned buffers
    // control-ahead mode: the message list belongs to the control stream (k_voice_control is its only reader there, and the
    // stream's order keeps this copy behind the control kernels of the previous call)
    hipStream_t s = c->cmds_on_ctl ? c->ctl_stream : c->stream;
    if (bytes) {
        memcpy(c->h_cmds, c->cmds.data(), bytes);
        HIPC(c, hipMemcpyAsync(c->d_cmds.p, c->h_cmds, bytes, hipMemcpyHostToDevice, s));
    }
    if (order) {
        const int nv = (int)c->ctl_mark.size();
        int w = 0;
        for (const int v : c->hot_now) c->h_ctl_order[w++] = v;
        for (int v = 0; v < nv; ++v)
            if (!c->ctl_mark[v]) c->h_ctl_order[w++] = v;
        for (const int v : c->hot_now) c->ctl_mark[v] = 0;
        HIPC(c, hipMemcpyAsync(c->d_ctl_order.p, c->h_ctl_order, (size_t)nv * sizeof(int), hipMemcpyHostToDevice, s));
    }
    c->hot_prev.assign(c->hot_now.begin(), c->hot_now.begin() + n_now);  // (within the reserved capacity)
    HIPC(c, hipEventRecord(c->cmds_copied, s));
    return 0;
}
// A SetSample message has been applied by the work enqueued so far: the sampler let go of the sample it held
// (sampler.rs:339-343 ReturnSample).  The host knows which one without asking the device — only SetSample changes it.
static void note_set_sample(fwgpu_ctx* c, const Cmd& m) {
    if (m.type != CMD_SMP_SET_SAMPLE || m.state < 0 || (size_t)m.state >= c->cur_sample.size()) return;
    const int old = c->cur_sample[m.state];
    c->cur_sample[m.state] = m.i0;
    if (old < 0) return;
    RetItem it;
    it.node = c->slot_ids[m.state];
    it.sample = old;
    it.ticket = c->ret_ticket;
    if (c->returns.stage(it)) c->ret_this_call = true;  // (a full ring drops the notice like the reference's `let _ = push`)
}
// The notices of a call become visible to the control side only AFTER the call's completion event has been recorded: a
// poller that saw the item first would query an event that was never recorded (hipSuccess) or was recorded 64 tickets ago,
// and report a sample returned while this call's kernels still read it.  ret_event_ticket says which ticket
// a slot's event currently answers for: a slot that has not been recorded for this ticket yet reads "not ready".
void finish_returns(fwgpu_ctx* c) {
    if (!c->ret_this_call) return;
    c->ret_this_call = false;
    const uint32_t slot = c->ret_ticket % fwgpu_ctx::RET_EVENTS;
    (void)hipEventRecord(c->ret_events[slot], c->stream);
    c->ret_event_ticket[slot].store(c->ret_ticket + 1, std::memory_order_release);
    c->returns.publish();
    c->ret_ticket++;
}
void retire_cmds(fwgpu_ctx* c, uint32_t nblocks) {
    size_t w = 0;  // in place: nothing is allocated on the process path
    for (const Cmd& m : c->cmds)
        if (m.block >= nblocks) {
            Cmd k = m;
            k.block -= nblocks;
            c->cmds[w++] = k;
        } else {
            note_set_sample(c, m);
        }
    c->cmds.resize(w);
    finish_returns(c);
}
// B1 (fwgpu_node_process): ONE node consumed one block.  Only its own queue moves — the reference keeps a ring / an
// atomic per node (sampler.rs:205-208, volume.rs:10), so a message for node B must survive node A's process().
void retire_cmds_node(fwgpu_ctx* c, int slot) {
    size_t w = 0;
    for (const Cmd& m : c->cmds) {
        if (m.state == slot) {
            if (m.block == 0) {  // applied by this call
                note_set_sample(c, m);
                continue;
            }
            Cmd k = m;
            k.block -= 1;
            c->cmds[w++] = k;
        } else {
            c->cmds[w++] = m;
        }
    }
    c->cmds.resize(w);
    finish_returns(c);
}

// ---------------------------------------------------------------- timing helpers
void timer_begin(fwgpu_ctx* c, int cat, hipEvent_t* e0, hipEvent_t* e1) {
    *e0 = *e1 = nullptr;
    if (!c->timing) return;
    TimerCat& t = c->timers[cat];
    if (t.used == t.ev.size()) {
        if (t.ev.size() >= 8192) return;  // drained by timing_read
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        t.ev.emplace_back(a, b);
    }
    *e0 = t.ev[t.used].first;
    *e1 = t.ev[t.used].second;
    t.used++;
    t.launches++;
    (void)hipEventRecord(*e0, c->stream);
}
void timer_end(fwgpu_ctx* c, hipEvent_t e1) {
    if (e1) (void)hipEventRecord(e1, c->stream);
}
void timer_drain(fwgpu_ctx* c) {
    (void)hipStreamSynchronize(c->stream);
    for (TimerCat& t : c->timers) {
        for (size_t i = 0; i < t.used; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, t.ev[i].first, t.ev[i].second) == hipSuccess) t.acc_ms += ms;
        }
        t.used = 0;
    }
}

// ---------------------------------------------------------------- executors
DevView generic_view(fwgpu_ctx* c, int frames) {
    DevView v;
    v.nodes = c->d_nodes.as<NodeDesc>();
    v.in_buf = c->d_in_buf.as<int>();
    v.out_buf = c->d_out_buf.as<int>();
    v.states = c->d_states.as<NodeState>();
    v.samples = c->d_samples.as<SampleDesc>();
    v.ext = c->d_ext.as<float>();
    v.rs_table = c->d_rs_table.as<float>();
    v.pool = c->d_pool.as<float>();
    v.flags = c->d_flags.as<uint8_t>();
    v.pool_blk_stride = (size_t)c->plan.num_buffers * c->stride;  // one pool slice per block of a K-batch
    v.flags_blk_stride = (size_t)c->plan.num_buffers;
    v.stride = c->stride;
    v.frames = frames;
    v.cmds = c->d_cmds.as<Cmd>();
    v.n_cmds = c->n_cmds_dev;
    v.frozen = nullptr;
    v.frozen_playhead = nullptr;
    v.chain_done = nullptr;
    v.chain_words = 0;
    return v;
}

// what the voice-bank kernels see (the buses are the fused plan's; the hybrid plan points them at the pool)
static void fill_fused_view(fwgpu_ctx* c, FusedView& fv) {
    fv.voices = c->d_voices.as<VoiceDesc>();
    fv.leaves = c->d_leaves.as<LeafDesc>();
    fv.states = c->d_states.as<NodeState>();
    fv.samples = c->d_samples.as<SampleDesc>();
    fv.blks = c->d_blks.as<VoiceBlk>();
    fv.refs = c->d_refs.as<VoiceRef>();
    fv.ref_kgroups = (int)((c->kmax + FW_REF_TILE_BLOCKS - 1) / FW_REF_TILE_BLOCKS);
    fv.gsets = c->d_gsets.as<GainSet>();
    fv.cache = c->d_cache.as<VoiceCache>();
    fv.epoch = c->epoch;
    fv.rs_table = c->d_rs_table.as<float>();
    fv.progs = c->d_progs.as<uint32_t>();
    fv.has_prog = c->fused_prog ? 1 : 0;
    fv.has_rs = c->fused_rs ? 1 : 0;
    fv.has_sp = c->fused_sp ? 1 : 0;
    fv.rt_parent_leaf = fv.rt_parent_up = fv.rt_kids = nullptr;
    fv.rt_tree_sync = nullptr;
    fv.rt_root = -1;
    if (c->rt_tree_leaves > 0 && c->d_rt_tree.p && c->d_rt_tree_sync.p) {
        fv.rt_parent_leaf = c->d_rt_tree.as<int>();
        fv.rt_parent_up = fv.rt_parent_leaf + c->rt_tree_leaves;
        fv.rt_kids = fv.rt_parent_up + c->rt_tree_up;
        fv.rt_tree_sync = c->d_rt_tree_sync.as<unsigned>();
        fv.rt_root = c->up_root_node;
    }
    fv.rs_wl = c->d_rs_wl.as<unsigned int>();
    fv.rs_tmpl = c->fused_rs ? c->d_rs_tmpl.as<VoiceBlk>() : nullptr;
    fv.lazy = (c->lazy_capable && c->d_lazy.p) ? c->d_lazy.as<LazyRec>() : nullptr;
    fv.lazy_tmpl = (fv.lazy && c->fused_rs && c->d_rs_tmpl.p) ? c->d_rs_tmpl.as<VoiceBlk>() + 2 * (size_t)c->n_voices : nullptr;
    fv.horizon = fv.lazy ? c->d_lazy_horizon.as<unsigned long long>() : nullptr;
    fv.abs_blk_end = 0;
    fv.lazy_blk0 = 0;
    fv.ctl_order = c->ctl_order_live ? c->d_ctl_order.as<int>() : nullptr;
    fv.sp_hist_in_render = (c->ahead_this_call && c->fused_sp) ? 1 : 0;
    fv.hist = c->d_hist.as<float>();
    fv.n_gain_stages = c->ramp_slots / 2;
    fv.ramps = c->d_ramps.as<float>();
    fv.ramp_slots = c->ramp_slots;
    fv.bus = c->d_bus.as<float>();
    fv.bus_flags = c->d_bus_flags.as<uint8_t>();
    fv.bus_blk_stride = (size_t)c->n_bus * c->stride;
    fv.bus_flags_blk_stride = (size_t)c->n_bus;
    fv.cmds = c->d_cmds.as<Cmd>();
    fv.n_cmds = c->n_cmds_dev;
    fv.n_voices = c->n_voices;
    fv.n_leaves = c->n_leaves;
    fv.stride = c->stride;
    fv.frames = (int)c->mbf;
    fv.fx_plan = c->fused_fx ? 1 : 0;
    fv.groups = c->d_groups.as<ChainGroup>();
    fv.n_groups = c->n_groups;
    fv.ext = c->d_ext.as<float>();
    fv.chain_start = c->d_chain_start.as<ChainStart>();
    fv.chain_dummy = c->d_chain_dummy.as<float>();
    fv.chain_stats = c->d_chain_stats.as<unsigned long long>();
    fv.trace = nullptr;
#ifdef FW_CHAIN_TRACE
    if (c->d_trace.ensure_n("d_trace", 64 * 16 * 8 * sizeof(unsigned long long)) == hipSuccess) fv.trace = c->d_trace.as<unsigned long long>();
#endif
    {
        static const int dbg = getenv("FWGPU_CHAIN_SKIP") ? atoi(getenv("FWGPU_CHAIN_SKIP")) : 0;
        fv.dbg = dbg;
    }
}

// The plan is CUT at a level that holds host nodes (K_HOST: the caller's own AudioNodeProcessor::process, graph/processor.rs:243).
// The device nodes of the level have been launched; for each host node: its input buffers of the K blocks -> pinned host memory
// (one small kernel; nothing else crosses), wait for the stream, the callback once per block in block order on THIS (the audio)
// thread with the ProcInfo of the call in progress, its outputs + the silence mask it reported -> back into the pool.
static int run_host_level(fwgpu_ctx* c, const DevView& v, const std::vector<fwgpu_ctx::HostCall>& calls, int K, int frames) {
    for (const fwgpu_ctx::HostCall& hc : calls) {
        const int rows = hc.n_in + hc.n_out;
        if (hc.n_in > 0)
            LCHK(c, launch_host_gather(c->stream, v.pool, v.flags, c->stride, v.pool_blk_stride, v.flags_blk_stride, c->d_in_buf.as<int>() + hc.in_off,
                                   hc.n_in, frames, K, rows, c->d_host_stage + hc.stage_off, c->d_host_flags + hc.flag_off));
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    for (const fwgpu_ctx::HostCall& hc : calls) {
        const int rows = hc.n_in + hc.n_out;
        float* st = c->h_host_stage + hc.stage_off;
        uint8_t* fl = c->h_host_flags + hc.flag_off;
        for (int k = 0; k < K; ++k) {
            uint64_t in_mask = 0, out_mask = 0;  // (processor.rs:233: the out mask arrives cleared)
            for (int j = 0; j < hc.n_in; ++j) {
                c->host_in_ptrs[j] = st + ((size_t)k * rows + j) * c->stride;
                if (fl[(size_t)k * rows + j]) in_mask |= 1ull << j;
            }
            for (int j = 0; j < hc.n_out; ++j) {
                float* o = st + ((size_t)k * rows + hc.n_in + j) * c->stride;
                memset(o, 0, (size_t)frames * sizeof(float));  // a node that breaks "fill every output" leaves zeros, not another block's data
                c->host_out_ptrs[j] = o;
            }
            hc.fn(hc.user, (uint64_t)frames, c->host_in_ptrs.data(), (uint32_t)hc.n_in, c->host_out_ptrs.data(), (uint32_t)hc.n_out, in_mask, &out_mask,
                  c->proc_stream_time, c->proc_stream_status);
            for (int j = 0; j < hc.n_out; ++j) fl[(size_t)k * rows + hc.n_in + j] = (out_mask >> j) & 1ull ? 1 : 0;
            c->host_callbacks++;
        }
        if (hc.n_out > 0)
            LCHK(c, launch_host_scatter(c->stream, v.pool, v.flags, c->stride, v.pool_blk_stride, v.flags_blk_stride, c->d_out_buf.as<int>() + hc.out_off,
                                    hc.n_out, frames, K, rows, c->d_host_stage + hc.stage_off + (size_t)hc.n_in * c->stride,
                                    c->d_host_flags + hc.flag_off + hc.n_in));
    }
    return 0;
}

// K blocks of `frames` frames through the level-batched executor (schedule.rs:289-344 as one launch per level for
// all K blocks: each block has its own pool slice, a stateful node walks its K blocks in order inside one wave)
int run_generic_batch(fwgpu_ctx* c, int K, int frames, uint32_t cmd_block, const float* d_in, int n_in_ch, float* d_out,
                      int n_out_ch) {
    c->lazy_valid = false;  // (the level executor and the hybrid plan move node state their own way)
    if (K == 1) c->rt_path[3]++;
    DevView v = generic_view(c, frames);
    // which gain-like stateful nodes cannot change during this batch (their blocks then run in parallel): decided once,
    // before the first level
    if (K > 1 && c->d_frozen.ensure_n("d_frozen", (size_t)c->plan.nodes.size()) == hipSuccess &&
        c->d_frozen_ph.ensure_n("d_frozen_ph", (size_t)c->plan.nodes.size() * sizeof(unsigned long long)) == hipSuccess) {
        // vertical fusion: the scan clears this batch's "rendered upstream" bits (sized by the plan build for generic_k blocks)
        if (c->level_fuse && c->d_chain_done.p && c->chain_words > 0 && K <= c->chain_words * 32 &&
            c->d_chain_done.cap >= c->plan.nodes.size() * (size_t)c->chain_words * sizeof(uint32_t)) {
            v.chain_done = c->d_chain_done.as<uint32_t>();
            v.chain_words = c->chain_words;
        }
        LCHK(c, launch_frozen_scan(c->stream, v, (int)c->plan.nodes.size(), cmd_block, K, c->d_frozen.as<uint8_t>(),
                                   c->d_frozen_ph.as<unsigned long long>()));
        v.frozen = c->d_frozen.as<uint8_t>();
        v.frozen_playhead = c->d_frozen_ph.as<unsigned long long>();
    }
    if (c->n_gin_bufs > 0)
        LCHK(c, launch_graph_in(c->stream, v.pool, v.flags, c->stride, v.pool_blk_stride, v.flags_blk_stride,
                                c->d_gin_bufs.as<int>(), c->n_gin_bufs, d_in, d_in ? n_in_ch : 0, frames, K));
    // hybrid plan (whole blocks): the voice banks of the graph — SumNodes whose every port is a dry voice chain — are
    // rendered by the voice-bank kernels straight into those SumNodes' pool buffers; the levels below then run without them
    const bool hy = c->hybrid && !c->force_generic && frames == (int)c->mbf;
    hipEvent_t e0, e1;
    if (hy) {
        FusedView fv;
        fill_fused_view(c, fv);
        fv.bus = v.pool;
        fv.bus_flags = v.flags;
        fv.bus_blk_stride = v.pool_blk_stride;
        fv.bus_flags_blk_stride = v.flags_blk_stride;
        fv.fx_plan = c->hybrid_fx ? 1 : 0;
        timer_begin(c, 1, &e0, &e1);
        LCHK(c, launch_voice_control(c->stream, fv, K, cmd_block));
        timer_end(c, e1);
        timer_begin(c, 0, &e0, &e1);
        if (c->hybrid_fx) LCHK(c, launch_chain(c->stream, fv, K, cmd_block, c->chain_nq));
        else LCHK(c, launch_leaf_sum(c->stream, fv, K));
        timer_end(c, e1);
    } else {
        c->epoch++;  // node state moves outside the fused control kernel: cached steady descriptors are stale
    }
    const std::vector<int>& lv_off = hy ? c->hlevel_off : c->level_off;
    const std::vector<int>& lv_cnt = hy ? c->hlevel_cnt : c->level_cnt;
    const std::vector<int>& lv_kinds = hy ? c->hlevel_kinds : c->level_kinds;
    const int* lv_nodes = hy ? c->d_hlevel_nodes.as<int>() : c->d_level_nodes.as<int>();
    timer_begin(c, 3, &e0, &e1);
    for (size_t l = 0; l < lv_cnt.size(); ++l) {
        if (lv_cnt[l] > 0) LCHK(c, launch_level(c->stream, v, lv_nodes + lv_off[l], lv_cnt[l], K, cmd_block, lv_kinds[l]));
        for (const fwgpu_ctx::FirGroup& g : c->fir_groups)
            if (g.level == (int)l) {
                hipEvent_t g0 = nullptr, g1 = nullptr;
                if (c->timing) {  // the GEMM alone, on its own event pair (no record of its own: launch_fir does it)
                    TimerCat& t = c->timers[4];
                    if (t.used == t.ev.size() && t.ev.size() < 8192) {
                        hipEvent_t a, b;
                        if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) t.ev.emplace_back(a, b);
                    }
                    if (t.used < t.ev.size()) {
                        g0 = t.ev[t.used].first;
                        g1 = t.ev[t.used].second;
                        t.used++;
                        t.launches++;
                    }
                }
                LCHK(c, launch_fir(c->stream, v, c->d_fir_rows.as<FirRow>() + g.row_off, g.n_rows,
                                   c->d_fir_tiles.as<uint32_t>() + g.tile_off, g.T, c->d_fir_partials.as<float>(),
                                   c->d_fir_partials.cap / sizeof(float), K, g0, g1));
            }
        if (l < c->host_levels.size() && !c->host_levels[l].empty()) {
            const int rc = run_host_level(c, v, c->host_levels[l], K, frames);
            if (rc) return rc;
        }
    }
    timer_end(c, e1);
    LCHK(c, launch_graph_out(c->stream, v.pool, v.flags, c->stride, v.pool_blk_stride, v.flags_blk_stride,
                             c->d_gout_bufs.as<int>(), c->n_gout_bufs, d_out, n_out_ch, frames, K));
    if (c->out_sil)  // read_graph_outputs' silence mask per block (schedule.rs:255-287), for the top-level SumNode of a sharded graph
        LCHK(c, launch_out_flags(c->stream, v.flags, v.flags_blk_stride, c->d_gout_bufs.as<int>(), c->n_gout_bufs, 0, n_out_ch, K,
                                 c->out_sil + (size_t)cmd_block * n_out_ch));
    return 0;
}

// ---------------------------------------------------------------- lazy records (fwgpu_types.h LazyRec)
// Node state lags behind the audio by the blocks that were rendered straight from the LazyRecs; whatever reads or moves it next
// — a control kernel, the realtime kernels, the level executor, fwgpu_node_process, a plan adoption — calls this first.  One small
// launch on the ctx stream (nothing of a lazy call ever runs on the control stream); the LazyRecs are spent afterwards.
int lazy_flush(fwgpu_ctx* c) {
    c->lazy_valid = false;
    if (!c->lazy_pending) return 0;
    const uint64_t n = c->lazy_pending;
    c->lazy_pending = 0;
    if (!c->d_lazy.p || c->n_voices <= 0) return 0;
    LCHK(c, launch_lazy_flush(c->stream, c->d_lazy.as<LazyRec>(), c->d_states.as<NodeState>(), c->n_voices, n,
                              c->fused_fx ? c->d_voices.as<VoiceDesc>() : nullptr));
    return 0;
}

// K full blocks through the fused voice-bank plan
// ---------------------------------------------------------------- the resident realtime kernel (k_rt.hip.h)
int rt_persist_stop(fwgpu_ctx* c) {
    fwgpu_ctx::RtResident& r = c->rtp;
    if (!r.launched) return 0;
    r.launched = false;
    volatile unsigned long long* alive = &c->h_rt_mb->alive;
    if (*alive) {
        __atomic_store_n(&c->h_rt_mb->doorbell, r.next_seq | RT_QUIT_BIT, __ATOMIC_RELEASE);
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(10 * (long)c->rt_idle_ms + 50);
        for (unsigned spins = 1; *alive; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() > give_up) break;  // (the stream sync below names the problem)
        }
    }
    // the kernel's own end (its stores are released at agent scope block by block; this orders the stream behind it)
    HIPC(c, hipStreamSynchronize(c->rt_stream));
    return 0;
}
// first steady callback of a run (or the one after the watchdog ended the kernel): launch it with this call's sequence number
// already in the doorbell
static int rt_persist_launch(fwgpu_ctx* c, const FusedView& fv, const DevView& v, float* d_out, uint32_t cmd_block0, unsigned long long seq) {
    fwgpu_ctx::RtResident& r = c->rtp;
    HIPC(c, hipEventRecord(c->rt_ev, c->stream));  // behind everything the ctx stream holds (adoption launches, uploads)
    HIPC(c, hipStreamWaitEvent(c->rt_stream, c->rt_ev, 0));
    __atomic_store_n(&c->h_rt_mb->alive, 1ull, __ATOMIC_SEQ_CST);
    if (__atomic_load_n(&c->h_rt_mb->hold, __ATOMIC_SEQ_CST)) {  // a control call holds the device (RtHold): no resident kernel now
        __atomic_store_n(&c->h_rt_mb->alive, 0ull, __ATOMIC_SEQ_CST);
        r.held++;
        return 1;  // (the caller renders this block with an ordinary launch)
    }
    __atomic_store_n(&c->h_rt_mb->doorbell, seq, __ATOMIC_RELEASE);
    unsigned long long* go = (unsigned long long*)((char*)c->d_rt_sync.p + 128);
    // (the kernel before this one may have left `seq | quit` there — its watchdog fired while it waited for this very number)
    HIPC(c, hipMemsetAsync(go, 0, sizeof(unsigned long long), c->rt_stream));
    LCHK(c, launch_rt_persist(c->rt_stream, fv, v, c->root_args, d_out, cmd_block0, c->d_rt_sync.as<unsigned>(), c->d_rt_flag, c->d_rt_mb, go, seq,
                              (unsigned long long)c->rt_idle_ms * 100000ull));
    r.launched = true;
    r.epoch = c->epoch;
    r.d_out = d_out;
    r.blks = fv.blks;
    r.next_seq = seq + 1;
    r.launches++;
    return 0;
}

static void rt_root_view(const fwgpu_ctx* c, const FusedView& fv, DevView& v) {
    v = DevView{};
    // (the root's port table travels in RootArgs; the mixers between the leaves and the root — bus_sum_node_wg on the way up the tree,
    //  k_rt.hip.h — read theirs from the upper tree's tables)
    v.nodes = c->d_up_nodes.as<NodeDesc>();
    v.in_buf = c->d_up_in.as<int>();
    v.out_buf = c->d_up_out.as<int>();
    v.pool = fv.bus;
    v.flags = fv.bus_flags;
    v.pool_blk_stride = fv.bus_blk_stride;
    v.flags_blk_stride = fv.bus_flags_blk_stride;
    v.stride = c->stride;
    v.frames = (int)c->mbf;
}
int rt_block_relaunch(fwgpu_ctx* c, float* d_out, unsigned long long seq) {
    int rc = rt_persist_stop(c);
    if (rc) return rc;
    FusedView fv;
    fill_fused_view(c, fv);
    fv.lazy = nullptr;  // (the one-launch kernels run their own control and leave no LazyRecs)
    fv.horizon = nullptr;
    c->lazy_valid = false;
    DevView v;
    rt_root_view(c, fv, v);
    LCHK(c, launch_rt_block(c->stream, fv, v, c->root_args, d_out, 0, c->d_rt_sync.as<unsigned>(), c->d_rt_flag, seq));
    return 0;
}

int run_fused_batch(fwgpu_ctx* c, int K, uint32_t cmd_block0, float* d_out, int n_out_ch) {
    FusedView fv;
    fill_fused_view(c, fv);
    // realtime edge: one block, tree = leaves + root, stereo stream -> the whole callback is ONE launch (k_rt_block)
    if (K == 1 && c->rt_one_launch && !c->lazy_this_call && !c->fused_sp && !c->out_sil && !c->ahead_this_call && !c->fused_fx && !c->timing && c->n_tail == 0 && c->up_root_node >= 0 && n_out_ch == 2 &&
        c->rt_tree_leaves > 0 && c->rt_tree_leaves == c->n_leaves && c->d_rt_sync.p) {
        DevView v;
        rt_root_view(c, fv, v);
        c->lazy_valid = false;  // (the one-launch kernels run their own control: the LazyRecs no longer describe the voices)
        fv.lazy = nullptr;
        fv.horizon = nullptr;
        unsigned long long* flag = nullptr;
        if (c->rt_signal_seq && c->rt_last_batch) {  // the realtime edge asked for the completion flag and this is the call's
                                                       // last launch: the kernel raises it itself
            flag = c->d_rt_flag;
            c->rt_signalled = true;
        }
        // a steady callback (no message on the device, the completion flag asked for): the resident kernel takes it
        // (on a big tree the resident kernel LOSES to a launch per callback — config 5's 256 leaf workgroups: 94-113 us resident, 51-53
        //  launched, round 5: hundreds of idle workgroups polling beside the few that carry the block up the tree — so it is the small
        //  trees' edge: rt_persist_max_leaves, FWGPU_RT_PERSIST_MAX_LEAVES)
        if (c->rt_persist && c->n_leaves <= c->rt_persist_max_leaves && c->rt_stream && flag && c->n_cmds_dev == 0 && !c->host_prof && cmd_block0 == 0) {
            fwgpu_ctx::RtResident& r = c->rtp;
            const unsigned long long seq = c->rt_signal_seq;
            if (r.launched && c->h_rt_mb->alive && r.epoch == c->epoch && r.d_out == d_out && r.blks == fv.blks && r.next_seq == seq &&
                !__atomic_load_n(&c->h_rt_mb->hold, __ATOMIC_RELAXED)) {
                __atomic_store_n(&c->h_rt_mb->doorbell, seq, __ATOMIC_RELEASE);
                r.next_seq = seq + 1;
                r.doorbells++;
                c->rt_path[0]++;
                return 0;
            }
            int rc = rt_persist_stop(c);  // (the watchdog ended it, or it was launched for another plan / epoch / output block)
            if (rc) return rc;
            if (!__atomic_load_n(&c->h_rt_mb->hold, __ATOMIC_SEQ_CST)) {
                rc = rt_persist_launch(c, fv, v, d_out, cmd_block0, seq);
                if (rc == 0) c->rt_path[0]++;  // (the block the resident kernel was launched with)
                if (rc <= 0) return rc;
            }
            // a control call holds the device: this callback is an ordinary launch (below)
        }
        {
            int rc = rt_persist_stop(c);
            if (rc) return rc;
        }
        c->rt_path[1]++;
        if (c->host_prof) {
            const auto t0 = std::chrono::steady_clock::now();
            LCHK(c, launch_rt_block(c->stream, fv, v, c->root_args, d_out, cmd_block0, c->d_rt_sync.as<unsigned>(), flag, c->rt_signal_seq));
            c->hp_launch_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            c->hp_launches++;
            return 0;
        }
        LCHK(c, launch_rt_block(c->stream, fv, v, c->root_args, d_out, cmd_block0, c->d_rt_sync.as<unsigned>(), flag, c->rt_signal_seq));
        return 0;
    }
    if (K == 1) c->rt_path[2]++;
    hipEvent_t e0, e1;
    const bool lazy = c->lazy_this_call && fv.lazy != nullptr;
    fv.abs_blk_end = c->abs_blk + (uint64_t)K;
    if (lazy) {
        // no control kernel: every voice's records of these K blocks follow from its LazyRec and the block index
        fv.lazy_blk0 = c->abs_blk - c->lazy_base_blk;
        fv.lazy_chain = c->fused_fx ? 1 : 0;
        if (c->fused_rs && !c->fused_fx) {  // k_leaf_rs: records from the LazyRecs, templates from the copy the last control kernel left beside them
            fv.lazy_rs = 1;
            fv.rs_tmpl = fv.lazy_tmpl;
        }
        c->lazy_pending += (uint64_t)K;
        c->lazy_calls++;
    } else if (c->ahead_this_call) {
        const int p = (int)(c->ahead_seq & 1);
        if (p) {
            fv.blks = c->d_blks2.as<VoiceBlk>();
            fv.refs = c->d_refs2.as<VoiceRef>();
            fv.gsets = c->d_gsets2.as<GainSet>();
            fv.ramps = c->d_ramps2.as<float>();
            if (fv.rs_tmpl) fv.rs_tmpl += c->n_voices;
        }
        if (c->ahead_seq >= 2) HIPC(c, hipStreamWaitEvent(c->ctl_stream, c->ev_render[p], 0));  // batch b-2 has read this copy
        LCHK(c, launch_voice_control(c->ctl_stream, fv, K, cmd_block0, true));
        if (fv.lazy) LCHK(c, launch_lazy_publish(c->ctl_stream, fv.horizon, c->d_lazy_pub, ++c->ctl_launch_seq));
        HIPC(c, hipEventRecord(c->ev_ctl[p], c->ctl_stream));
        HIPC(c, hipStreamWaitEvent(c->stream, c->ev_ctl[p], 0));
        c->streams_split = true;
    } else {
        timer_begin(c, 1, &e0, &e1);
        LCHK(c, launch_voice_control(c->stream, fv, K, cmd_block0));
        timer_end(c, e1);
        if (fv.lazy) LCHK(c, launch_lazy_publish(c->stream, fv.horizon, c->d_lazy_pub, ++c->ctl_launch_seq));
    }
    if (!lazy) {
        c->ctl_calls++;
        if (fv.lazy) {  // the LazyRecs this control kernel leaves: block 0 = right behind this batch
            c->lazy_base_blk = c->abs_blk + (uint64_t)K;
            c->lazy_epoch = c->epoch;
            c->lazy_valid = true;
        }
    }
    c->abs_blk += (uint64_t)K;
    timer_begin(c, 0, &e0, &e1);
    if (fv.sp_hist_in_render) LCHK(c, launch_sp_hist_copy(c->stream, fv));
    if (c->fused_fx) LCHK(c, launch_chain(c->stream, fv, K, cmd_block0, c->chain_nq));
    else if (lazy && !c->fused_rs) LCHK(c, launch_leaf_sum_lazy(c->stream, fv, K));
    else LCHK(c, launch_leaf_sum(c->stream, fv, K));
    timer_end(c, e1);
    timer_begin(c, 2, &e0, &e1);
    if (!c->up_level_cnt.empty()) {
        DevView v;
        v.nodes = c->d_up_nodes.as<NodeDesc>();
        v.in_buf = c->d_up_in.as<int>();
        v.out_buf = c->d_up_out.as<int>();
        v.states = c->d_states.as<NodeState>();
        v.samples = c->d_samples.as<SampleDesc>();
        v.ext = c->d_ext.as<float>();
        v.rs_table = c->d_rs_table.as<float>();
        v.pool = fv.bus;
        v.flags = fv.bus_flags;
        v.pool_blk_stride = fv.bus_blk_stride;
        v.flags_blk_stride = fv.bus_flags_blk_stride;
        v.stride = c->stride;
        v.frames = (int)c->mbf;
        v.cmds = nullptr;
        v.n_cmds = 0;
        v.frozen = nullptr;
        v.frozen_playhead = nullptr;
        // the root SumNode is fused with read_graph_outputs + interleave_stereo when the stream is stereo
        const bool fuse_root = c->up_root_node >= 0 && n_out_ch == 2;
        const size_t n_levels = c->up_level_cnt.size() - (fuse_root ? 1 : 0);
        for (size_t l = 0; l < n_levels; ++l)
            LCHK(c, launch_bus_sum(c->stream, v, c->d_up_level_nodes.as<int>() + c->up_level_off[l], c->up_level_cnt[l], K, 2));
        if (fuse_root) {
            LCHK(c, launch_root_out(c->stream, v, c->root_args, d_out, K));
            if (c->out_sil)  // the root's out-mask (k_root_out keeps it in registers): recomputed from its inputs' flags
                LCHK(c, launch_out_flags(c->stream, v.flags, v.flags_blk_stride, c->root_args.in_tab, c->root_args.n_in, 1, n_out_ch, K,
                                         c->out_sil + (size_t)cmd_block0 * n_out_ch));
            timer_end(c, e1);
            if (c->ahead_this_call) {
                HIPC(c, hipEventRecord(c->ev_render[c->ahead_seq & 1], c->stream));
                c->ahead_seq++;
            }
            return 0;
        }
    }
    if (c->n_tail) {  // master chain on the mix bus: the generic node kernel, K-batched, one launch per node
        DevView v;
        v.nodes = c->d_tail_nodes.as<NodeDesc>();
        v.in_buf = c->d_tail_in.as<int>();
        v.out_buf = c->d_tail_out.as<int>();
        v.states = c->d_states.as<NodeState>();
        v.samples = c->d_samples.as<SampleDesc>();
        v.ext = c->d_ext.as<float>();
        v.rs_table = c->d_rs_table.as<float>();
        v.pool = fv.bus;
        v.flags = fv.bus_flags;
        v.pool_blk_stride = fv.bus_blk_stride;
        v.flags_blk_stride = fv.bus_flags_blk_stride;
        v.stride = c->stride;
        v.frames = (int)c->mbf;
        v.cmds = fv.cmds;
        v.n_cmds = fv.n_cmds;
        v.frozen = nullptr;
        v.frozen_playhead = nullptr;  // (no sampler can sit in a master chain)
        if (K > 1) {
            LCHK(c, launch_frozen_scan(c->stream, v, c->n_tail, cmd_block0, K, c->d_tail_frozen.as<uint8_t>(),
                                       c->d_tail_frozen.as<unsigned long long>()));
            v.frozen = c->d_tail_frozen.as<uint8_t>();
        }
        for (int j = 0; j < c->n_tail; ++j)
            LCHK(c, launch_level(c->stream, v, c->d_tail_idx.as<int>() + j, 1, K, cmd_block0, c->tail_kinds[j]));
    }
    LCHK(c, launch_graph_out(c->stream, fv.bus, fv.bus_flags, c->stride, fv.bus_blk_stride, fv.bus_flags_blk_stride,
                             c->d_root_bufs.as<int>(), 2, d_out, n_out_ch, (int)c->mbf, K));
    if (c->out_sil)
        LCHK(c, launch_out_flags(c->stream, fv.bus_flags, fv.bus_flags_blk_stride, c->d_root_bufs.as<int>(), 2, 0, n_out_ch, K,
                                 c->out_sil + (size_t)cmd_block0 * n_out_ch));
    timer_end(c, e1);
    if (c->ahead_this_call) {
        HIPC(c, hipEventRecord(c->ev_render[c->ahead_seq & 1], c->stream));
        c->ahead_seq++;
    }
    return 0;
}

// all blocks of one call; d_in may be null.  frames may end in a partial block.
static int run_blocks_impl(fwgpu_ctx* c, uint64_t frames, const float* d_in, int n_in_ch, float* d_out, int n_out_ch, bool stable_out);
int run_blocks(fwgpu_ctx* c, uint64_t frames, const float* d_in, int n_in_ch, float* d_out, int n_out_ch,
               bool stable_out) {
    if (!c->host_prof) return run_blocks_impl(c, frames, d_in, n_in_ch, d_out, n_out_ch, stable_out);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = run_blocks_impl(c, frames, d_in, n_in_ch, d_out, n_out_ch, stable_out);
    const uint64_t ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    c->hp_call_ns += ns;
    if (c->hp_calls > 50) {  // (histogram of the host time inside run_blocks, 25 us bins: where the launch calls themselves are held up)
        const uint64_t bin = std::min<uint64_t>(ns / 25000, 15);
        c->hp_hist[bin]++;
    }
    c->hp_calls++;
    return rc;
}
static int run_blocks_impl(fwgpu_ctx* c, uint64_t frames, const float* d_in, int n_in_ch, float* d_out, int n_out_ch,
                           bool stable_out) {
    const uint32_t mbf = c->mbf;
    const uint32_t nblocks = (uint32_t)((frames + mbf - 1) / mbf);
    int rc = upload_sample_table(c);
    if (rc) return rc;
    const bool can_fuse = c->fused && !c->force_generic;
    // the resident realtime kernel owns the voice state between two steady callbacks: anything else ends it first
    if (c->rtp.launched && !(stable_out && frames == mbf && can_fuse && !c->timing)) {
        rc = rt_persist_stop(c);
        if (rc) return rc;
    }
    // control-ahead mode for this call?  Whole blocks only, more than one, no event timers, not the realtime edge — and (round 4,
    // FWGPU_CTL_AHEAD=2, the default) only a call that has something to hide: a message on the list, or voices the call before
    // sent messages to (their glides continue).  A message-free call's control kernel is ~8 us of steady tails; beside the render
    // kernel of the call before it cost that kernel 7-15 us of co-residency and the step a ~10 us cross-stream event.
    drain_ring(c);
    const bool hot = c->ctl_ahead_mode == 1 || !c->cmds.empty() || !c->hot_prev.empty();
    const bool ahead = c->ctl_ahead_on && hot && can_fuse && !c->timing && !stable_out && frames % mbf == 0 && frames / mbf > 1;
    if (!ahead && c->streams_split) {
        rc = join_streams(c);
        if (rc) return rc;
    }
    // Lazy records: no message anywhere on the list, none in the call before (glides), whole blocks, the plan's LazyRecs made under
    // this epoch and not overtaken by anything else — and the host has SEEN what the control kernel that made them reported: its
    // sequence number and, beside it, the absolute block up to which every voice of the plan holds (0: some voice is not plain).
    c->lazy_this_call = false;
    // (one-block callbacks belong to the realtime kernels, which run their own control)
    if (c->lazy_on && can_fuse && c->lazy_capable && !ahead && !(stable_out && frames == mbf) && !c->rt_use_graph && frames % mbf == 0 && c->cmds.empty() &&
        c->hot_prev.empty() &&
        c->lazy_valid && c->lazy_epoch == c->epoch && c->h_lazy_pub) {
        const unsigned long long seq = __atomic_load_n(&c->h_lazy_pub[1], __ATOMIC_ACQUIRE);
        const unsigned long long horizon = __atomic_load_n(&c->h_lazy_pub[0], __ATOMIC_RELAXED);
        c->lazy_this_call = seq == c->ctl_launch_seq && c->abs_blk + nblocks <= horizon;
    }
    if (!c->lazy_this_call) {
        rc = lazy_flush(c);  // (on the ctx stream, in front of everything this call launches — the control stream joins behind it)
        if (rc) return rc;
    }
    if (ahead && !c->streams_split) {  // the control stream picks up behind everything the main stream holds so far
        HIPC(c, hipEventRecord(c->ev_join, c->stream));
        HIPC(c, hipStreamWaitEvent(c->ctl_stream, c->ev_join, 0));
        c->ahead_seq = 0;
    }
    c->ahead_this_call = ahead;
    c->cmds_on_ctl = ahead;
    rc = upload_cmds(c, true);
    if (rc) {
        c->ahead_this_call = false;
        return rc;
    }
    uint64_t done = 0;
    uint32_t blk = 0;
    // steady realtime call: no message on the device, one fused batch, the same output block as last time — every
    // kernel argument repeats (block counters and playheads live in device state), so the launch sequence is replayed
    // from a hipGraph instead of being re-issued kernel by kernel
    if (stable_out && c->rt_use_graph && can_fuse && !c->timing && c->n_cmds_dev == 0 && frames % mbf == 0 &&
        frames / mbf <= (c->fused_fx ? std::min<uint32_t>(c->kmax, CH_FAST_KMAX) : c->kmax)) {
        const uint32_t K = (uint32_t)(frames / mbf);
        fwgpu_ctx::RtGraph& g = c->rt_graph;
        if (!g.exec || g.epoch != c->epoch || g.K != K || g.d_out != d_out || g.n_out_ch != n_out_ch) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            g.exec = nullptr;
            HIPC(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            rc = run_fused_batch(c, (int)K, 0, d_out, n_out_ch);
            hipGraph_t graph = nullptr;
            hipError_t ce = hipStreamEndCapture(c->stream, &graph);
            if (rc || ce != hipSuccess) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc ? rc : hipfail(c, ce, "hipStreamEndCapture");
            }
            ce = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ce != hipSuccess) {
                g.exec = nullptr;
                return hipfail(c, ce, "hipGraphInstantiate");
            }
            g.epoch = c->epoch;
            g.K = K;
            g.d_out = d_out;
            g.n_out_ch = n_out_ch;
        }
        HIPC(c, hipGraphLaunch(g.exec, c->stream));
        retire_cmds(c, nblocks);
        return 0;
    }
    while (done < frames) {
        uint64_t left = frames - done;
        if (can_fuse && left >= mbf) {
            const uint32_t kcap = c->fused_fx ? std::min<uint32_t>(c->kmax, CH_FAST_KMAX) : c->kmax;
            uint32_t K = (uint32_t)std::min<uint64_t>(left / mbf, kcap);
            c->rt_last_batch = (uint64_t)K * mbf == left;
            rc = run_fused_batch(c, (int)K, blk, d_out + done * n_out_ch, n_out_ch);
            c->rt_last_batch = false;
            if (rc) return rc;
            done += (uint64_t)K * mbf;
            blk += K;
            continue;
        }
        // generic executor: whole blocks in batches of generic_k, a trailing partial block on its own
        int bf = (int)std::min<uint64_t>(left, mbf);
        int K = bf == (int)mbf ? (int)std::min<uint64_t>(left / mbf, c->generic_k) : 1;
        rc = run_generic_batch(c, K, bf, blk, d_in ? d_in + done * n_in_ch : nullptr, n_in_ch, d_out + done * n_out_ch, n_out_ch);
        if (rc) return rc;
        done += (uint64_t)K * bf;
        blk += K;
    }
    if (c->rt_signal_seq && !c->rt_signalled) {
        LCHK(c, launch_signal_done(c->stream, c->d_rt_flag, c->rt_signal_seq));
        c->rt_signalled = true;
    }
    c->ahead_this_call = false;
    retire_cmds(c, nblocks);
    return 0;
}

}  // namespace fwgpu
