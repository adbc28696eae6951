// fwgpu_plan_install.cpp — node activation (graph.rs:594-612) and upload of the launch plan's device tables.
#include "fwgpu_ctx.h"

namespace fwgpu {

// what the control kernel writes and the render kernels read, per voice and block of a batch (both fused plans and the hybrid one)
static int alloc_voice_tables(fwgpu_ctx* c) {
    const size_t K = c->kmax;
    HIPC(c, c->d_blks.ensure(K * c->n_voices * sizeof(VoiceBlk)));
    HIPC(c, c->d_refs.ensure(ref_count(c->n_voices, K) * sizeof(VoiceRef)));
    HIPC(c, c->d_gsets.ensure((size_t)c->n_voices * FW_GSETS * sizeof(GainSet)));
    HIPC(c, c->d_chain_start.ensure((size_t)c->n_voices * sizeof(ChainStart)));
    HIPC(c, c->d_chain_dummy.ensure(64 * 1024));
    HIPC(c, c->d_chain_stats.ensure(2 * sizeof(unsigned long long)));
    HIPC(c, hipMemset(c->d_chain_stats.p, 0, 2 * sizeof(unsigned long long)));
    HIPC(c, hipMemset(c->d_chain_start.p, 0, (size_t)c->n_voices * sizeof(ChainStart)));
    HIPC(c, c->d_cache.ensure((size_t)c->n_voices * sizeof(VoiceCache)));
    HIPC(c, hipMemset(c->d_cache.p, 0, (size_t)c->n_voices * sizeof(VoiceCache)));
    c->epoch++;
    HIPC(c, c->d_ramps.ensure(K * c->n_voices * (size_t)c->ramp_slots * c->stride * sizeof(float)));
    return 0;
}

// k_chain workgroups: consecutive leaves packed greedily into groups of <= 32 voices / <= 8 leaves (the voices of
// consecutive leaves are consecutive), so that a tree of small leaves fills the 32 voice rows
static int upload_chain_groups(fwgpu_ctx* c, const std::vector<LeafDesc>& leaves) {
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
    c->n_groups = (int)groups.size();
    if ((rc = upload(c, c->d_groups, groups.data(), groups.size() * sizeof(ChainGroup)))) return rc;
    return 0;
}

static int install_plan_impl(fwgpu_ctx* c, Plan& plan, bool* tables_touched) {
    HIPC(c, hipStreamSynchronize(c->stream));
    if (c->ctl_stream) HIPC(c, hipStreamSynchronize(c->ctl_stream));
    c->streams_split = false;
    c->ahead_seq = 0;
    c->ctl_ahead_on = false;
    c->kmax = c->kmax_req;
    // 1. node state capacity (persists across recompiles: processor.rs:19,195-197)
    size_t need = c->graph.nodes.size();
    if (need > c->states_cap) {
        size_t cap = std::max<size_t>(need * 2, 64);
        DevBuf nb;
        HIPC(c, nb.ensure(cap * sizeof(NodeState)));
        HIPC(c, hipMemset(nb.p, 0, cap * sizeof(NodeState)));
        if (c->d_states.p && c->states_cap)
            HIPC(c, hipMemcpy(nb.p, c->d_states.p, c->states_cap * sizeof(NodeState), hipMemcpyDeviceToDevice));
        c->d_states = std::move(nb);
        c->states_cap = cap;
    }
    // 2. activate new nodes (graph.rs:594-612): scatter their initial states, carve their ext-pool slices.
    //    Two-phase: every offset / initial state / impulse-response slot is worked out in locals, the device copies are
    //    made, and only then is the host bookkeeping (activated flags, init records, ext_used, ir_cache) committed — a
    //    failure anywhere leaves the ctx exactly as it was and the next fwgpu_update tries the same nodes again.
    {
        struct Act {
            uint32_t slot;
            NodeState st;  // the node's init record with its ext slice / FIR ring geometry filled in
        };
        std::vector<Act> acts;
        std::vector<StateInitHost> inits;
        std::vector<std::pair<size_t, std::vector<float>>> ext_inits;  // (offset, initial floats)
        std::map<std::pair<int, int>, uint32_t> new_ir;                // impulse responses to convert to f32 -> ext offset
        size_t ext_need = c->ext_used;
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
        std::map<size_t, std::vector<uint32_t>> free_backup = c->ext_free;  // restored on failure
        auto rollback = [&](int rc) {
            c->ext_free = free_backup;
            return rc;
        };
        std::vector<std::pair<uint32_t, uint32_t>> zero_slices;  // recycled slices start from zeros like fresh ones
        for (uint32_t slot : c->graph.nodes_to_activate) {
            const HostNode& n = c->graph.nodes[slot];
            if (!n.alive || n.activated) continue;
            if (n.kind == K_HOST && (slot >= c->host_procs.size() || !c->host_procs[slot].fn))
                return rollback(fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, "host node without a process function (fwgpu_host_node_set_process)"));
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
                    return rollback(fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, "FIR node: impulse-response sample was destroyed"));
                uint64_t T = c->samples[ir].desc.frames;
                if (T == 0 || T > (1u << 24))
                    return rollback(fail(c, FWGPU_ERR_NODE_ACTIVATION_FAILED, "FIR node: 1 <= taps <= 2^24"));
                uint64_t R = T - 1 + (uint64_t)c->kmax * c->mbf;  // every block of a K-batch finds its whole window in the ring
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
                if (!take_ext(len, &off)) return rollback(fail(c, FWGPU_ERR_INVALID, "ext state pool exceeds 2^32 floats"));
                a.st.ext_off = off;
                a.st.ext_len = (uint32_t)len;
                if (ext_need == before) zero_slices.emplace_back(off, (uint32_t)((len + 63) / 64 * 64));
                if (!head.empty()) ext_inits.emplace_back(off, head);
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
            if (ext_need + (T + 63) / 64 * 64 > 0xffffffffull)
                return rollback(fail(c, FWGPU_ERR_INVALID, "ext state pool exceeds 2^32 floats"));
            kv.second = (uint32_t)ext_need;
            ext_need += (T + 63) / 64 * 64;
        }
        int arc = 0;
        auto device_side = [&]() -> int {
            if (ext_need > c->ext_cap) {
                size_t cap = std::max<size_t>(ext_need * 2, 4096);
                DevBuf nb;
                HIPC(c, nb.ensure((cap + 256) * sizeof(float)));  // slack: vector loads may overhang the last slice
                HIPC(c, hipMemset(nb.p, 0, (cap + 256) * sizeof(float)));
                if (c->d_ext.p && c->ext_used)
                    HIPC(c, hipMemcpy(nb.p, c->d_ext.p, c->ext_used * sizeof(float), hipMemcpyDeviceToDevice));
                c->d_ext = std::move(nb);  // (same contents up to ext_used, more room: consistent whether or not the rest succeeds)
                c->ext_cap = cap;
            }
            for (auto& z : zero_slices)
                HIPC(c, hipMemsetAsync(c->d_ext.as<float>() + z.first, 0, (size_t)z.second * sizeof(float), c->stream));
            if (!ext_inits.empty()) {  // one upload + one scatter launch, however many nodes were activated
                std::vector<ExtInitHost> items(ext_inits.size());
                for (size_t i = 0; i < ext_inits.size(); ++i) {
                    items[i].off = (uint32_t)ext_inits[i].first;
                    items[i].n = (uint32_t)std::min<size_t>(ext_inits[i].second.size(), 6);
                    for (uint32_t j = 0; j < 6; ++j) items[i].v[j] = j < items[i].n ? ext_inits[i].second[j] : 0.f;
                }
                DevBuf tmp;
                int rc2 = upload(c, tmp, items.data(), items.size() * sizeof(ExtInitHost));
                if (rc2) return rc2;
                LCHK(c, launch_scatter_ext(c->stream, c->d_ext.as<float>(), tmp.p, (int)items.size()));
                HIPC(c, hipStreamSynchronize(c->stream));
                tmp.release();
            }
            if (!new_ir.empty()) {
                int rc = upload_sample_table(c);
                if (rc) return rc;
                for (auto& kv : new_ir)
                    LCHK(c, launch_ir_convert(c->stream, c->d_samples.as<SampleDesc>(), kv.first.first, kv.first.second,
                                              c->d_ext.as<float>() + kv.second, (uint32_t)c->samples[kv.first.first].desc.frames));
                HIPC(c, hipStreamSynchronize(c->stream));
            }
            if (!inits.empty()) {
                DevBuf tmp;
                int rc = upload(c, tmp, inits.data(), inits.size() * sizeof(StateInitHost));
                if (rc) return rc;
                LCHK(c, launch_scatter_states(c->stream, c->d_states.as<NodeState>(), tmp.p, (int)inits.size()));
                HIPC(c, hipStreamSynchronize(c->stream));
                tmp.release();
            }
            return 0;
        };
        if ((arc = device_side()) != 0) return rollback(arc);
        // commit
        // (audio-side tables: no process call overlaps an update.)  A removed sampler's processor is dropped with the old
        // schedule and hands its sample back (sampler.rs:563-571); a newly activated node starts without one.
        if (c->cur_sample.size() < c->graph.nodes.size()) {
            const size_t cap = std::max<size_t>(c->graph.nodes.size() * 2, 64);
            c->cur_sample.resize(cap, -1);
            c->slot_ids.resize(cap, -1);
        }
        for (uint32_t slot : c->dropped_samplers) {
            const int smp = c->cur_sample[slot];
            if (smp >= 0 && (size_t)smp < c->sample_refs.size() && c->sample_refs[smp] > 0) c->sample_refs[smp]--;
            c->cur_sample[slot] = -1;
        }
        c->dropped_samplers.clear();
        for (const Act& a : acts) {
            HostNode& n = c->graph.nodes[a.slot];
            n.init = a.st;
            n.activated = true;
            c->cur_sample[a.slot] = -1;
            c->slot_ids[a.slot] = c->graph.id_of(a.slot);
        }
        for (auto& kv : new_ir) {
            c->ir_cache[kv.first] = kv.second;
            c->ir_len[kv.first] = (uint32_t)c->samples[kv.first.first].desc.frames;
        }
        c->ext_used = ext_need;
        c->graph.nodes_to_activate.clear();
    }
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
    // 3. node tables (from here on the device tables of the OLD plan are being overwritten)
    *tables_touched = true;
    const int N = (int)plan.nodes.size();
    std::vector<NodeDesc> nd(N);
    std::vector<int> in_tab, out_tab;
    std::vector<std::vector<int>> levels(plan.num_levels);
    std::vector<int> gin_bufs, gout_bufs, host_nodes;
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
    int rc;
    if ((rc = upload(c, c->d_nodes, nd.data(), nd.size() * sizeof(NodeDesc)))) return rc;
    if ((rc = upload(c, c->d_in_buf, in_tab.data(), in_tab.size() * sizeof(int)))) return rc;
    if ((rc = upload(c, c->d_out_buf, out_tab.data(), out_tab.size() * sizeof(int)))) return rc;
    std::vector<int> flat;
    c->level_off.clear();
    c->level_cnt.clear();
    c->level_kinds.clear();
    for (auto& l : levels) {
        c->level_off.push_back((int)flat.size());
        c->level_cnt.push_back((int)l.size());
        flat.insert(flat.end(), l.begin(), l.end());
        int kinds = 0;
        for (int i : l) kinds |= host_kind_bits(nd[i].kind);
        c->level_kinds.push_back(kinds);
    }
    if (flat.empty()) flat.push_back(0);
    if ((rc = upload(c, c->d_level_nodes, flat.data(), flat.size() * sizeof(int)))) return rc;
    c->n_gin_bufs = (int)gin_bufs.size();
    c->n_gout_bufs = (int)gout_bufs.size();
    if (gin_bufs.empty()) gin_bufs.push_back(0);
    if (gout_bufs.empty()) gout_bufs.push_back(0);
    if ((rc = upload(c, c->d_gin_bufs, gin_bufs.data(), gin_bufs.size() * sizeof(int)))) return rc;
    if ((rc = upload(c, c->d_gout_bufs, gout_bufs.data(), gout_bufs.size() * sizeof(int)))) return rc;
    // 3c. host nodes (K_HOST): per level, what the audio side needs to call them — and one pinned, device-mapped staging area
    //     for their inputs and outputs of a whole K-batch, allocated here (a process call never allocates)
    {
        c->host_levels.assign(plan.num_levels, {});
        c->n_host_nodes = (int)host_nodes.size();
        c->host_callbacks = 0;
        size_t floats = 0, flag_bytes = 0, max_in = 1, max_out = 1;
        for (int i : host_nodes) {
            const PlanNode& p = plan.nodes[i];
            fwgpu_ctx::HostCall hc;
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
            floats += (size_t)c->kmax * (size_t)(p.n_in + p.n_out) * c->stride;
            flag_bytes += (size_t)c->kmax * (size_t)(p.n_in + p.n_out);
            max_in = std::max<size_t>(max_in, (size_t)p.n_in);
            max_out = std::max<size_t>(max_out, (size_t)p.n_out);
            c->host_levels[p.level].push_back(hc);
        }
        if (floats > c->host_stage_floats || flag_bytes > c->host_flag_bytes) {
            if (c->h_host_stage) (void)hipHostFree(c->h_host_stage);
            if (c->h_host_flags) (void)hipHostFree(c->h_host_flags);
            c->h_host_stage = nullptr;
            c->h_host_flags = nullptr;
            c->host_stage_floats = c->host_flag_bytes = 0;
            void *hs = nullptr, *hf = nullptr, *ds = nullptr, *df = nullptr;
            HIPC(c, hipHostMalloc(&hs, floats * sizeof(float), hipHostMallocMapped));
            c->h_host_stage = (float*)hs;
            HIPC(c, hipHostMalloc(&hf, flag_bytes + 64, hipHostMallocMapped));
            c->h_host_flags = (uint8_t*)hf;
            HIPC(c, hipHostGetDevicePointer(&ds, hs, 0));
            HIPC(c, hipHostGetDevicePointer(&df, hf, 0));
            c->d_host_stage = (float*)ds;
            c->d_host_flags = (uint8_t*)df;
            c->host_stage_floats = floats;
            c->host_flag_bytes = flag_bytes;
        }
        c->host_in_ptrs.assign(max_in, nullptr);
        c->host_out_ptrs.assign(max_out, nullptr);
    }
    // 3b. FIR banks: one GEMM per (level, impulse-response channel)
    {
        std::map<std::tuple<int, uint32_t, uint32_t>, std::vector<FirRow>> groups;
        for (int i = 0; i < N; ++i) {
            const PlanNode& p = plan.nodes[i];
            if (p.kind != K_FIR) continue;
            const HostNode& hn = c->graph.nodes[p.slot];
            int ir = hn.init.sample;
            uint32_t T = (uint32_t)hn.init.loop_start;
            int nch = std::min(p.n_in, p.n_out);
            for (int ch = 0; ch < nch; ++ch) {
                auto key = std::make_pair(ir, std::min(ch, c->samples[ir].desc.channels - 1));
                FirRow r;
                r.state = (int)p.slot;
                r.ch = ch;
                r.in_buf = p.in_buf[ch];
                r.out_buf = p.out_buf[ch];
                groups[std::make_tuple(p.level, c->ir_cache[key], T)].push_back(r);
            }
        }
        // one launch per (level, T); inside it rows are sorted by impulse-response channel and padded so that
        // every 32-row tile convolves with a single h (tile_h_off)
        std::vector<FirRow> flat_rows;
        std::vector<uint32_t> flat_tiles;
        c->fir_groups.clear();
        size_t partial_need = 0;
        std::map<std::pair<int, uint32_t>, std::vector<std::pair<uint32_t, std::vector<FirRow>*>>> launches;
        for (auto& g : groups)
            launches[std::make_pair(std::get<0>(g.first), std::get<2>(g.first))].emplace_back(std::get<1>(g.first), &g.second);
        for (auto& l : launches) {
            fwgpu_ctx::FirGroup fg;
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
            c->fir_groups.push_back(fg);
            size_t W = (size_t)fg.T - 1 + c->mbf;
            size_t segs = (W + FIR_SEG - 1) / FIR_SEG;
            partial_need = std::max(partial_need, segs * (size_t)fg.n_rows * (size_t)((c->mbf + 255) / 256 * 256) * c->kmax);
        }
        if (!flat_rows.empty()) {
            if ((rc = upload(c, c->d_fir_rows, flat_rows.data(), flat_rows.size() * sizeof(FirRow)))) return rc;
            if ((rc = upload(c, c->d_fir_tiles, flat_tiles.data(), flat_tiles.size() * sizeof(uint32_t)))) return rc;
            HIPC(c, c->d_fir_partials.ensure(partial_need * sizeof(float)));
        }
    }
    // 4. buffer pool: a new schedule starts from zeroed buffers (schedule.rs:202-203); one slice per block of a
    //    generic K-batch.  generic_k: the FIR history rings were sized for the batch size in force when their node was
    //    activated — a later, larger kmax must not outrun them.
    c->generic_k = c->kmax;
    for (int i = 0; i < N; ++i) {
        if (plan.nodes[i].kind != K_FIR) continue;
        const HostNode& hn = c->graph.nodes[plan.nodes[i].slot];
        uint64_t room = (hn.init.loop_end - (hn.init.loop_start - 1)) / c->mbf;  // (R - (T-1)) / block
        c->generic_k = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(c->generic_k, room));
    }
    {
        const size_t Kg = c->generic_k;
        size_t pool_bytes = Kg * (size_t)plan.num_buffers * c->stride * sizeof(float);
        HIPC(c, c->d_pool.ensure(pool_bytes));
        HIPC(c, hipMemset(c->d_pool.p, 0, pool_bytes));
        std::vector<uint8_t> fl(Kg * (size_t)plan.num_buffers, 0);
        for (size_t k = 0; k < Kg; ++k) fl[k * (size_t)plan.num_buffers] = 1;  // buffer 0: constant zero, always flagged silent
        if ((rc = upload(c, c->d_flags, fl.data(), fl.size()))) return rc;
    }

    // 5. fused voice-bank plan
    c->fused = false;
    c->hybrid = false;
    c->fused_fx = false;
    if (is_fused) {
        c->fused_fx = fb.has_fx;
        // k_chain tile = 64*nq frames: the larger tile needs whole tiles per block and every delay >= one tile
        c->chain_nq = (c->mbf % 128 == 0 && fb.min_delay >= 128) ? 2 : 1;
        if (const char* e = getenv("FWGPU_CHAIN_NQ")) {  // experiments: force the smaller tile
            if (atoi(e) == 1) c->chain_nq = 1;
        }
        c->n_voices = (int)fb.voices.size();
        c->n_leaves = (int)fb.leaves.size();
        c->n_bus = fb.n_bus;
        c->ramp_slots = 2 * (1 + fb.max_stages);
        if ((rc = upload(c, c->d_voices, fb.voices.data(), fb.voices.size() * sizeof(VoiceDesc)))) return rc;
        if ((rc = upload(c, c->d_leaves, fb.leaves.data(), fb.leaves.size() * sizeof(LeafDesc)))) return rc;
        if ((rc = upload(c, c->d_progs, fb.progs.data(), fb.progs.size() * sizeof(uint32_t)))) return rc;
        c->fused_prog = fb.has_prog;
        c->fused_rs = fb.has_rs;
        c->n_groups = 0;
        if (c->fused_fx) {
            if ((rc = upload_chain_groups(c, fb.leaves))) return rc;
        }
        const size_t K = c->kmax;
        if ((rc = alloc_voice_tables(c))) return rc;
        if (c->ctl_ahead && c->ctl_stream && !c->fused_fx && fb.tail_nodes.empty() && K > 1) {
            // the second copy of what the control kernel writes and the render kernels read
            bool ok = c->d_blks2.ensure(K * c->n_voices * sizeof(VoiceBlk)) == hipSuccess &&
                      c->d_refs2.ensure(ref_count(c->n_voices, K) * sizeof(VoiceRef)) == hipSuccess &&
                      c->d_gsets2.ensure((size_t)c->n_voices * FW_GSETS * sizeof(GainSet)) == hipSuccess &&
                      c->d_ramps2.ensure(K * c->n_voices * (size_t)c->ramp_slots * c->stride * sizeof(float)) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            c->ctl_ahead_on = ok;
        }
        size_t bus_bytes = K * (size_t)c->n_bus * c->stride * sizeof(float);
        HIPC(c, c->d_bus.ensure(bus_bytes));
        HIPC(c, hipMemset(c->d_bus.p, 0, bus_bytes));
        std::vector<uint8_t> bf(K * c->n_bus, 0);
        for (size_t k = 0; k < K; ++k) bf[k * c->n_bus] = 1;
        if ((rc = upload(c, c->d_bus_flags, bf.data(), bf.size()))) return rc;
        if (fb.up_nodes.empty()) {
            NodeDesc z;
            memset(&z, 0, sizeof(z));
            fb.up_nodes.push_back(z);
        }
        if (fb.up_in.empty()) fb.up_in.push_back(0);
        if (fb.up_out.empty()) fb.up_out.push_back(0);
        if ((rc = upload(c, c->d_up_nodes, fb.up_nodes.data(), fb.up_nodes.size() * sizeof(NodeDesc)))) return rc;
        if ((rc = upload(c, c->d_up_in, fb.up_in.data(), fb.up_in.size() * sizeof(int)))) return rc;
        if ((rc = upload(c, c->d_up_out, fb.up_out.data(), fb.up_out.size() * sizeof(int)))) return rc;
        std::vector<int> uflat;
        c->up_level_off.clear();
        c->up_level_cnt.clear();
        for (auto& l : fb.up_levels) {
            c->up_level_off.push_back((int)uflat.size());
            c->up_level_cnt.push_back((int)l.size());
            uflat.insert(uflat.end(), l.begin(), l.end());
        }
        c->up_root_node = (!fb.up_levels.empty() && fb.up_levels.back().size() == 1) ? fb.up_levels.back()[0] : -1;
        c->n_tail = (int)fb.tail_nodes.size();
        c->tail_kinds.clear();
        for (const NodeDesc& t : fb.tail_nodes) c->tail_kinds.push_back(host_kind_bits(t.kind));
        if (c->n_tail) {
            c->up_root_node = -1;  // the root's planar result feeds the master chain: no fused root + interleave
            std::vector<int> idx(c->n_tail);
            for (int i = 0; i < c->n_tail; ++i) idx[i] = i;
            if ((rc = upload(c, c->d_tail_nodes, fb.tail_nodes.data(), fb.tail_nodes.size() * sizeof(NodeDesc)))) return rc;
            if ((rc = upload(c, c->d_tail_in, fb.tail_in.data(), fb.tail_in.size() * sizeof(int)))) return rc;
            if ((rc = upload(c, c->d_tail_out, fb.tail_out.data(), fb.tail_out.size() * sizeof(int)))) return rc;
            if ((rc = upload(c, c->d_tail_idx, idx.data(), idx.size() * sizeof(int)))) return rc;
            HIPC(c, c->d_tail_frozen.ensure((size_t)c->n_tail * 16));  // (also a dummy playhead-snapshot area)
        }
        if (c->up_root_node >= 0) {
            const NodeDesc& rn = fb.up_nodes[c->up_root_node];
            if (rn.n_out == 2 && rn.n_in >= 2 && rn.n_in <= 64 && rn.n_in % 2 == 0) {
                memset(&c->root_args, 0, sizeof(c->root_args));
                c->root_args.n_in = rn.n_in;
                c->root_args.ports = rn.n_in / 2;
                for (int i = 0; i < rn.n_in; ++i) c->root_args.in_buf[i] = fb.up_in[rn.in_off + i];
                c->root_args.in_tab = c->d_up_in.as<int>() + rn.in_off;
            } else {
                c->up_root_node = -1;
            }
        }
        if (uflat.empty()) uflat.push_back(0);
        if ((rc = upload(c, c->d_up_level_nodes, uflat.data(), uflat.size() * sizeof(int)))) return rc;
        if ((rc = upload(c, c->d_root_bufs, fb.root_buf, sizeof(fb.root_buf)))) return rc;
        c->fused = true;
        c->n_fused_real = 0;
        for (const VoiceDesc& vd : fb.voices) c->n_fused_real += vd.sampler_state >= 0 ? 1 : 0;
    }
    // 5b. hybrid plan: not a fused shape as a whole, but with voice banks inside that the fused kernels render
    // straight into their mixers' pool buffers; the level executor then runs the rest.
    c->hybrid_fx = false;
    if (is_hybrid) {
        c->n_voices = (int)hb.voices.size();
        c->n_leaves = (int)hb.leaves.size();
        c->ramp_slots = 2 * (1 + hb.max_stages);
        c->fused_prog = hb.has_prog;
        c->fused_rs = hb.has_rs;
        c->n_groups = 0;
        c->hybrid_fx = hb.has_fx;
        if (c->hybrid_fx) {  // the banks go through k_chain: its workgroups, its tile size, at most 64 blocks per launch
            if ((rc = upload_chain_groups(c, hb.leaves))) return rc;
            c->chain_nq = (c->mbf % 128 == 0 && hb.min_delay >= 128) ? 2 : 1;
            if (const char* e = getenv("FWGPU_CHAIN_NQ")) {
                if (atoi(e) == 1) c->chain_nq = 1;
            }
            c->generic_k = std::min<uint32_t>(c->generic_k, CH_FAST_KMAX);
        }
        c->n_tail = 0;
        c->up_root_node = -1;
        c->up_level_off.clear();
        c->up_level_cnt.clear();
        if ((rc = upload(c, c->d_voices, hb.voices.data(), hb.voices.size() * sizeof(VoiceDesc)))) return rc;
        if ((rc = upload(c, c->d_leaves, hb.leaves.data(), hb.leaves.size() * sizeof(LeafDesc)))) return rc;
        if ((rc = upload(c, c->d_progs, hb.progs.data(), hb.progs.size() * sizeof(uint32_t)))) return rc;
        if ((rc = alloc_voice_tables(c))) return rc;
        // the level lists without the nodes the voice-bank kernels render
        std::vector<char> cov(N, 0);
        for (int i : hb.covered) cov[i] = 1;
        std::vector<int> hflat;
        c->hlevel_off.clear();
        c->hlevel_cnt.clear();
        c->hlevel_kinds.clear();
        for (auto& l : levels) {
            c->hlevel_off.push_back((int)hflat.size());
            int kinds = 0, cnt = 0;
            for (int i : l)
                if (!cov[i]) {
                    hflat.push_back(split_entry[i] >= 0 ? split_entry[i] : i);
                    kinds |= host_kind_bits(nd[i].kind);
                    cnt++;
                }
            c->hlevel_cnt.push_back(cnt);
            c->hlevel_kinds.push_back(kinds);
        }
        if (hflat.empty()) hflat.push_back(0);
        if ((rc = upload(c, c->d_hlevel_nodes, hflat.data(), hflat.size() * sizeof(int)))) return rc;
        c->hybrid = true;
        c->n_fused_real = 0;
        for (const VoiceDesc& vd : hb.voices) c->n_fused_real += vd.sampler_state >= 0 ? 1 : 0;
    }
    // k_frozen_scan's verdict tables (generic executor, K > 1): sized here, on the control thread — a process call never
    // allocates
    HIPC(c, c->d_frozen.ensure(nd.size()));
    HIPC(c, c->d_frozen_ph.ensure(nd.size() * sizeof(unsigned long long)));
    c->plan = plan;
    c->have_plan = true;
    c->graph.needs_compile = false;
    return 0;
}

// A failure before the device tables are touched (node activation: a destroyed impulse response, an exhausted ext pool,
// a HIP error while scattering initial states) leaves the previous plan installed and valid, like the reference keeps its
// schedule when a compile fails (context.rs:115-131).  A failure after that point (only HIP errors: out of memory) has
// overwritten part of the old plan's tables: the ctx then has NO plan — process calls output silence (processor.rs:86-89)
// — and stays dirty, so the next fwgpu_update builds everything again.
int install_plan(fwgpu_ctx* c, Plan& plan) {
    bool touched = false;
    const int rc = install_plan_impl(c, plan, &touched);
    if (rc != 0 && touched) {
        c->have_plan = false;
        c->fused = false;
        c->hybrid = false;
        c->hybrid_fx = false;
        c->graph.needs_compile = true;
        c->epoch++;
    }
    return rc;
}

}  // namespace fwgpu
