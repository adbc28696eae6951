// fwgpu_plan_detect.cpp — launch-plan selection: recognise the graph shapes the fused plans render.
#include "fwgpu_ctx.h"

namespace fwgpu {

// ---------------------------------------------------------------- fused voice-bank plan detection
namespace {
// consumer counts per (node, output port), flat (a vector per node was 16 500 allocations on config 3)
struct ConsCounts {
    std::vector<int> off, cnt;
    explicit ConsCounts(const Plan& plan) {
        const int N = (int)plan.nodes.size();
        off.resize(N + 1);
        int t = 0;
        for (int i = 0; i < N; ++i) {
            off[i] = t;
            t += plan.nodes[i].n_out;
        }
        off[N] = t;
        cnt.assign(t, 0);
        for (const PlanNode& n : plan.nodes)
            for (int p = 0; p < n.n_in; ++p)
                if (n.in_src_node[p] >= 0) cnt[off[n.in_src_node[p]] + n.in_src_port[p]]++;
    }
    const int* operator[](int node) const { return cnt.data() + off[node]; }
    int n_out(int node) const { return off[node + 1] - off[node]; }
};

// both channels of input port pair `port0` come from ONE stereo node, each consumed exactly once
bool stereo_src(const Plan& plan, const ConsCounts& cons, const PlanNode& n, int port0, int& src) {
    int a = n.in_src_node[port0], b = n.in_src_node[port0 + 1];
    if (a < 0 || a != b) return false;
    if (n.in_src_port[port0] != 0 || n.in_src_port[port0 + 1] != 1) return false;
    if (plan.nodes[a].n_out != 2 || cons[a][0] != 1 || cons[a][1] != 1) return false;
    src = a;
    return true;
}

VoiceDesc null_voice() {  // an unconnected mixer port: k_voice_control emits a constant silent, cleared-source record for it
    VoiceDesc vd;
    memset(&vd, 0, sizeof(vd));
    vd.sampler_state = vd.bq_state = vd.dl_state = vd.sp_ext_off = vd.bq2_state = -1;
    return vd;
}

// One voice chain, walked UPSTREAM from its last node to its source — the grammar of both fused plans:
//   voice bank:  source -> {volume, pan, width, clip}* [-> spatialiser]            source = sampler | resampler | sampler(0->1) -> MonoToStereo
//   chain plan:  sampler -> G* -> F1 [-> G* -> F2 [-> G* -> F3]] -> G*     G = volume | pan | hard clip (<= 3 in all), F1 F2 F3 = B | BB | D | BD | BBD | DB | DBB
// (B biquad, D delay >= 64 frames).  Round 6: gain stages anywhere around and between the filters (the ones in front of the first see
// the source's silence flag, a muted one between two filters hands the next filter a cleared buffer: positional silence in
// k_voice_control), two biquads (an EQ cascade: the second one's recurrence runs on a wave of its own), the delay line in front of
// the biquads, hard clips at every position.  Still refused: B D B and more than three filters, a stereo width in a voice with a
// filter (it needs both channels; k_chain's workgroups own one), a resampler or spatialiser with a filter.  A refused voice is not lost:
// the hybrid plan renders its longest acceptable prefix as a solo voice, the level executor the rest.
struct VoiceWalk {
    bool ok = false;
    VoiceDesc vd;
    uint32_t prog_bits = 0;
    int nodes[FW_MAX_STAGES + 6], n_nodes = 0;  // plan indices, the source first
    bool prog = false, rs = false, fx = false, sp = false, width = false;
    uint64_t delay = ~0ull;
};
VoiceWalk walk_voice_chain(const Plan& plan, const HostGraph& graph, const ConsCounts& cons, uint32_t mbf, const std::vector<char>& taken, int cur) {
    VoiceWalk w;
    w.vd = null_voice();
    int chain[FW_MAX_STAGES], n_chain = 0;  // gain-like stages in walk order (nearest the mixer first)
    int seg_cnt[4] = {0, 0, 0, 0};  // gain-like stages met with 0 / 1 / 2 / 3 filters already behind us on the walk (= downstream of them)
    int fxn[3], n_fx = 0;  // biquads / the delay in walk order
    char fxk[4] = {0, 0, 0, 0};
    bool sp_voice = false;
    int mono_adapter = -1;
    for (;;) {
        const PlanNode& n = plan.nodes[cur];
        if (taken[cur] || n.is_graph_io) return w;
        if (n.kind == K_SAMPLER || n.kind == K_RESAMPLER) {
            if (!(n.n_in == 0 && n.n_out == 2)) return w;
            break;
        }
        if (n.kind == K_MONO_TO_STEREO) {
            // sampler(0 -> 1) -> MonoToStereoNode (mono_to_stereo.rs:33-50): the reference's own adapter behind a ONE-output sampler —
            // channel 0 of its sample on both outputs, silence passed on: a voice whose every block is VB_MONO (src_kind 2;
            // k_control.hip.h mono_adapt).  Round 6: also in front of filters (k_chain fetches a mono source like any other: r_delta 0).
            if (n.n_in != 1 || n.n_out != 2 || sp_voice) return w;
            const int sidx = n.in_src_node[0];
            if (sidx < 0 || n.in_src_port[0] != 0 || taken[sidx]) return w;
            const PlanNode& sn = plan.nodes[sidx];
            if (sn.kind != K_SAMPLER || sn.n_in != 0 || sn.n_out != 1 || sn.is_graph_io || cons[sidx][0] != 1) return w;
            mono_adapter = cur;
            cur = sidx;
            break;
        }
        if (n.n_in != 2 || n.n_out != 2) return w;
        if (n.kind == K_VOLUME || n.kind == K_PAN || n.kind == K_WIDTH || n.kind == K_HARD_CLIP) {
            if (n_chain >= FW_MAX_STAGES - 1) return w;
            w.width = w.width || n.kind == K_WIDTH;
            seg_cnt[n_fx]++;
            chain[n_chain++] = cur;
        } else if (n.kind == K_SPATIAL) {
            // a spatialiser as the LAST node of a dry voice (the first one met walking up from the mixer); its 64-frame history needs
            // whole 64-frame blocks
            if (n_chain || n_fx || mbf % 64 != 0) return w;
            sp_voice = true;
            chain[n_chain++] = cur;
        } else if (n.kind == K_DELAY || n.kind == K_BIQUAD) {
            if (n_fx >= 3 || sp_voice) return w;
            if (n.kind == K_DELAY) {
                if (graph.nodes[n.slot].init.loop_end < 64) return w;  // shorter than one k_chain tile
                w.delay = graph.nodes[n.slot].init.loop_end;
            }
            fxk[n_fx] = n.kind == K_DELAY ? 'D' : 'B';
            fxn[n_fx++] = cur;
        } else {
            return w;
        }
        int src;
        if (!stereo_src(plan, cons, n, 0, src)) return w;
        cur = src;
    }
    // the filters in SCHEDULE order (the walk met them last one first)
    int bq = -1, bq2 = -1, dl = -1, order = 0;
    if (n_fx) {
        char sched[4] = {0, 0, 0, 0};
        int sn[3] = {-1, -1, -1};
        for (int i = 0; i < n_fx; ++i) {
            sched[i] = fxk[n_fx - 1 - i];
            sn[i] = fxn[n_fx - 1 - i];
        }
        const std::string q(sched);
        if (q == "B") bq = sn[0];
        else if (q == "BB") bq = sn[0], bq2 = sn[1];
        else if (q == "D") dl = sn[0];
        else if (q == "BD") bq = sn[0], dl = sn[1];
        else if (q == "BBD") bq = sn[0], bq2 = sn[1], dl = sn[2];
        else if (q == "DB") dl = sn[0], bq = sn[1], order = 1;
        else if (q == "DBB") dl = sn[0], bq = sn[1], bq2 = sn[2], order = 1;
        else return w;
    }
    if (n_fx && w.width) return w;  // (a stereo width needs both channels of the voice: a chain-plan workgroup owns one)
    const bool rs = plan.nodes[cur].kind == K_RESAMPLER;
    if (rs && (n_fx || sp_voice)) return w;  // (the chain plan's source fetch is the sampler's; a spatialiser voice is a dry sampler voice)
    w.vd.sp_ext_off = sp_voice ? 0 : -1;  // the node's ext slice: filled in by the plan build (the node may be activated by this very plan)
    w.sp = sp_voice;
    w.prog = sp_voice;
    w.vd.sampler_state = (int)plan.nodes[cur].slot;
    w.vd.src_kind = mono_adapter >= 0 ? 2 : (rs ? 1 : 0);
    w.vd.bq_state = bq >= 0 ? (int)plan.nodes[bq].slot : -1;
    w.vd.bq2_state = bq2 >= 0 ? (int)plan.nodes[bq2].slot : -1;
    w.vd.dl_state = dl >= 0 ? (int)plan.nodes[dl].slot : -1;
    w.vd.fx_order = order;
    // walk segment s (s filters downstream of the stage) is schedule position n_fx - s: 0 = in front of the first filter ... n_fx = behind the last
    w.vd.n_pre = n_fx ? seg_cnt[n_fx] : 0;
    w.vd.n_mid = (n_fx >= 2 ? seg_cnt[n_fx - 1] : 0) | ((n_fx >= 3 ? seg_cnt[n_fx - 2] : 0) << 8);
    w.fx = n_fx != 0;
    w.rs = rs;
    w.vd.n_stages = n_chain;
    for (int j = 0; j < n_chain; ++j) {  // schedule order: nearest the source first
        const PlanNode& n = plan.nodes[chain[n_chain - 1 - j]];
        w.vd.stage_kind[j] = n.kind;
        w.vd.stage_state[j] = (int)n.slot;
        w.prog_bits |= (n.kind == K_WIDTH ? SK_WIDTH : n.kind == K_HARD_CLIP ? SK_CLIP : n.kind == K_SPATIAL ? SK_SPATIAL : SK_GAIN) << (4 * j);
        w.prog = w.prog || n.kind == K_WIDTH || n.kind == K_HARD_CLIP;
    }
    w.nodes[w.n_nodes++] = cur;
    if (mono_adapter >= 0) w.nodes[w.n_nodes++] = mono_adapter;
    for (int i = 0; i < n_fx; ++i) w.nodes[w.n_nodes++] = fxn[i];
    for (int j = 0; j < n_chain; ++j) w.nodes[w.n_nodes++] = chain[j];
    w.ok = true;
    return w;
}
}  // namespace


// `graph`: for the delay lengths (k_chain needs D >= one tile); `mbf` must then be a multiple of the tile
bool detect_fused(const Plan& plan, const HostGraph& graph, uint32_t mbf, FusedBuild& fb) {
    const int N = (int)plan.nodes.size();
    if (N < 3) return false;
    const PlanNode& gout = plan.nodes.back();
    if (gout.is_graph_io != 2 || gout.n_in != 2) return false;
    // consumer counts per (node, port)
    const ConsCounts cons(plan);
    auto stereo_src = [&](const PlanNode& n, int port0, int& src) -> bool { return fwgpu::stereo_src(plan, cons, n, port0, src); };
    int root;
    if (!stereo_src(gout, 0, root)) return false;
    std::vector<char> covered(N, 0);
    covered[N - 1] = 1;
    std::vector<int> tail;  // plan indices, graph_out side first
    while (plan.nodes[root].kind != K_SUM) {
        const PlanNode& n = plan.nodes[root];
        const bool master_kind = n.kind == K_VOLUME || n.kind == K_HARD_CLIP || n.kind == K_PAN || n.kind == K_WIDTH ||
                                 n.kind == K_BIQUAD || n.kind == K_DELAY;
        if (!master_kind || n.n_in != 2 || n.n_out != 2 || covered[root] || tail.size() >= 16) return false;
        covered[root] = 1;
        tail.push_back(root);
        int src;
        if (!stereo_src(n, 0, src)) return false;
        root = src;
    }
    for (int i = 0; i < N; ++i)
        if (plan.nodes[i].is_graph_io == 1) {
            covered[i] = 1;
            for (int p = 0; p < cons.n_out(i); ++p)
                if (cons[i][p]) return false;  // graph inputs feed the graph: generic executor
        }
    // walk the sum tree breadth-first
    struct SumRec {
        int node;
        bool leaf;
        std::vector<int> kids;  // plan indices (sum nodes) or chain ends
        int out_buf;
    };
    std::vector<SumRec> sums;
    std::map<int, int> sum_index;
    std::vector<int> work{root};
    while (!work.empty()) {
        int si = work.back();
        work.pop_back();
        const PlanNode& s = plan.nodes[si];
        if (s.kind != K_SUM || s.n_out != 2 || s.n_in < 2 || s.n_in % 2) return false;
        if (covered[si]) return false;
        covered[si] = 1;
        SumRec r;
        r.node = si;
        r.out_buf = 0;
        int n_sum = 0, n_chain = 0;
        for (int p = 0; p < s.n_in / 2; ++p) {
            int src;
            if (s.in_src_node[2 * p] < 0 && s.in_src_node[2 * p + 1] < 0) {
                // an unconnected stereo port (a voice slot nothing is plugged into): the reference feeds it the cleared,
                // silent-flagged buffer (schedule.rs:310-313) — a null kid: a null voice under a leaf, bus 0 above
                r.kids.push_back(-1);
                continue;
            }
            if (!stereo_src(s, 2 * p, src)) return false;
            r.kids.push_back(src);
            if (plan.nodes[src].kind == K_SUM) n_sum++;
            else n_chain++;
        }
        if (n_sum && n_chain) return false;
        r.leaf = n_sum == 0;  // (a SumNode with nothing plugged in at all is a leaf of null voices)
        if (!r.leaf)
            for (int k : r.kids)
                if (k >= 0) work.push_back(k);
        sum_index[si] = (int)sums.size();
        sums.push_back(r);
    }
    // leaves in plan order (deterministic), chains in port order
    std::vector<int> leaf_order;
    for (int i = 0; i < (int)sums.size(); ++i)
        if (sums[i].leaf) leaf_order.push_back(i);
    std::sort(leaf_order.begin(), leaf_order.end(), [&](int a, int b) { return sums[a].node < sums[b].node; });
    int next_bus = 1;
    {
        size_t nv = 0;
        for (int li : leaf_order) nv += sums[li].kids.size();
        fb.voices.reserve(nv);
        fb.progs.reserve(nv);
        fb.leaves.reserve(leaf_order.size());
    }
    for (int li : leaf_order) {
        SumRec& r = sums[li];
        LeafDesc ld;
        ld.first_voice = (int)fb.voices.size();
        ld.ports = (int)r.kids.size();
        ld.out_buf = next_bus;
        ld.pad = 0;
        r.out_buf = next_bus;
        next_bus += 2;
        for (int end : r.kids) {
            if (end < 0) {  // null voice: k_voice_control emits a constant silent, cleared-source record for it
                fb.voices.push_back(null_voice());
                fb.progs.push_back(0u);
                continue;
            }
            const VoiceWalk w = walk_voice_chain(plan, graph, cons, mbf, covered, end);  // (a node met twice: covered -> refused)
            if (!w.ok) return false;
            for (int i = 0; i < w.n_nodes; ++i) covered[w.nodes[i]] = 1;
            fb.has_fx = fb.has_fx || w.fx;
            fb.has_prog = fb.has_prog || w.prog || w.rs;  // (the polyphase fetch lives in the leaf kernel's program instantiation)
            fb.has_rs = fb.has_rs || w.rs;
            fb.has_sp = fb.has_sp || w.sp;
            fb.has_width = fb.has_width || w.width;
            fb.min_delay = std::min(fb.min_delay, w.delay);
            fb.progs.push_back(w.prog_bits);
            fb.max_stages = std::max(fb.max_stages, w.vd.n_stages);
            fb.voices.push_back(w.vd);
        }
        fb.leaves.push_back(ld);
    }
    for (int i = 0; i < N; ++i)
        if (!covered[i]) return false;  // anything else in the graph: generic executor
    // upper sums: heights above the leaves, children's buses resolved bottom-up
    std::vector<int> height(sums.size(), -1);
    std::function<int(int)> h = [&](int i) -> int {
        if (height[i] >= 0) return height[i];
        if (sums[i].leaf) return height[i] = 0;
        int m = 0;
        for (int k : sums[i].kids)
            if (k >= 0) m = std::max(m, h(sum_index[k]) + 1);
        return height[i] = m;
    };
    int maxh = 0;
    for (int i = 0; i < (int)sums.size(); ++i) maxh = std::max(maxh, h(i));
    fb.up_levels.assign(maxh, std::vector<int>());
    for (int lv = 1; lv <= maxh; ++lv) {
        std::vector<int> at;
        for (int i = 0; i < (int)sums.size(); ++i)
            if (height[i] == lv) at.push_back(i);
        std::sort(at.begin(), at.end(), [&](int a, int b) { return sums[a].node < sums[b].node; });
        for (int i : at) {
            SumRec& r = sums[i];
            r.out_buf = next_bus;
            next_bus += 2;
            NodeDesc nd;
            memset(&nd, 0, sizeof(nd));
            nd.kind = K_SUM;
            nd.n_in = (int)r.kids.size() * 2;
            nd.n_out = 2;
            nd.in_off = (int)fb.up_in.size();
            nd.out_off = (int)fb.up_out.size();
            nd.state = 0;
            nd.aux0 = (int)r.kids.size();
            for (int k : r.kids) {
                const int cb = k >= 0 ? sums[sum_index[k]].out_buf : 0;  // unconnected: bus 0, the cleared + silent-flagged buffer
                fb.up_in.push_back(cb);
                fb.up_in.push_back(k >= 0 ? cb + 1 : 0);
            }
            fb.up_out.push_back(r.out_buf);
            fb.up_out.push_back(r.out_buf + 1);
            fb.up_levels[lv - 1].push_back((int)fb.up_nodes.size());
            fb.up_nodes.push_back(nd);
        }
    }
    int rb = sums[sum_index[root]].out_buf;
    for (int j = (int)tail.size() - 1; j >= 0; --j) {  // root side first
        const PlanNode& n = plan.nodes[tail[j]];
        NodeDesc nd;
        memset(&nd, 0, sizeof(nd));
        nd.kind = n.kind;
        nd.n_in = nd.n_out = 2;
        nd.in_off = (int)fb.tail_in.size();
        nd.out_off = (int)fb.tail_out.size();
        nd.state = (int)n.slot;
        fb.tail_in.push_back(rb);
        fb.tail_in.push_back(rb + 1);
        rb = next_bus;
        next_bus += 2;
        fb.tail_out.push_back(rb);
        fb.tail_out.push_back(rb + 1);
        fb.tail_nodes.push_back(nd);
    }
    fb.root_buf[0] = rb;
    fb.root_buf[1] = rb + 1;
    fb.n_bus = next_bus;
    if (fb.has_fx) {  // k_chain: whole tiles, one workgroup per leaf of <= 32 voices, gain stages only behind the filter / delay
        // (k_chain's stages: gains and hard clips per channel; no width, no resampler fetch, no spatialiser)
        if (mbf % 64 != 0 || fb.has_width || fb.has_rs || fb.has_sp || fb.max_stages > FW_CHAIN_STAGES - 1) return false;
        for (const LeafDesc& l : fb.leaves)
            if (l.ports > 32) return false;
    }
    return !fb.voices.empty();
}

// ---------------------------------------------------------------- hybrid plan detection
// Banks: SumNodes whose every stereo port is a voice chain of a fused plan's shape (or nothing: a null voice).  If any bank
// holds a biquad / delay voice the banks go through the chain plan's kernels (k_chain, which also renders dry voices) and
// only banks that plan can take are kept — gains only, <= 3 of them, sampler sources, <= 32 ports, whole 64-frame tiles;
// otherwise through the voice-bank plan's (k_leaf_sum: stage programs, resampler sources).
bool detect_hybrid(const Plan& plan, const HostGraph& graph, uint32_t mbf, FusedBuild& fb) {
    const int N = (int)plan.nodes.size();
    const ConsCounts cons(plan);
    auto stereo_src = [&](const PlanNode& n, int port0, int& src) -> bool { return fwgpu::stereo_src(plan, cons, n, port0, src); };
    struct Bank {
        int sum;  // the SumNode — or, for a solo voice, the last node of its chain: the node whose output buffers the leaf writes
        std::vector<VoiceDesc> voices;
        std::vector<uint32_t> progs;
        std::vector<int> nodes;
        bool prog = false, rs = false, fx = false, sp = false, width = false;
        int stages = 0, real = 0;
        uint64_t min_delay = ~0ull;
        bool split = false;  // only the leading ports are voices: the SumNode stays on the levels as a continuation
        bool solo = false;   // one voice chain on its own (below)
    };
    typedef VoiceWalk Walk;  // one voice chain, walked upstream from its last node (walk_voice_chain above: the grammar of both fused plans)
    std::vector<char> taken(N, 0);  // nodes of a candidate bank (a chain node feeds one consumer, so banks cannot overlap)
    auto walk_voice = [&](int cur) -> Walk { return walk_voice_chain(plan, graph, cons, mbf, taken, cur); };
    auto take = [&](Bank& bk, const Walk& w) {
        bk.sp = bk.sp || w.sp;
        bk.width = bk.width || w.width;
        bk.prog = bk.prog || w.prog;
        bk.fx = bk.fx || w.fx;
        bk.rs = bk.rs || w.rs;
        bk.min_delay = std::min(bk.min_delay, w.delay);
        bk.stages = std::max(bk.stages, w.vd.n_stages);
        bk.nodes.insert(bk.nodes.end(), w.nodes, w.nodes + w.n_nodes);
        bk.voices.push_back(w.vd);
        bk.progs.push_back(w.prog_bits);
        bk.real++;
    };
    std::vector<Bank> banks;
    for (int si = 0; si < N; ++si) {
        const PlanNode& s = plan.nodes[si];
        if (s.kind != K_SUM || s.is_graph_io || s.n_out != 2 || s.n_in < 2 || s.n_in % 2 || s.n_in > 64) continue;
        if (s.out_buf[1] != s.out_buf[0] + 1) continue;  // the kernels write channel 1 in the row behind channel 0
        Bank bk;
        bk.sum = si;
        bool ok = true;
        for (int p = 0; p < s.n_in / 2; ++p) {  // (a port that turns out not to be a voice chain leaves the bank as it was)
            if (s.in_src_node[2 * p] < 0 && s.in_src_node[2 * p + 1] < 0) {  // an empty voice slot: a null voice
                bk.voices.push_back(null_voice());
                bk.progs.push_back(0u);
                continue;
            }
            int cur;
            if (!stereo_src(s, 2 * p, cur)) {
                ok = false;
                break;
            }
            const Walk w = walk_voice(cur);
            if (!w.ok) {
                ok = false;
                break;
            }
            take(bk, w);
        }
        if (bk.real == 0) continue;
        if (!ok) {
            // the leading ports are voices, a later one is something else: split the node (the partial sum of the leading
            // ports is the reference's accumulator at that point) — not for 2- / 3- / 4-port sums, whose paths are spelled out
            // port by port (sum.rs:67-110), and not when nothing but null slots leads
            const int P = s.n_in / 2;
            if (P == 2 || P == 3 || P == 4) continue;
            bk.split = true;
        } else {
            bk.nodes.push_back(si);
        }
        for (int i : bk.nodes) taken[i] = 1;
        banks.push_back(std::move(bk));
    }
    bool fx_mode = false;
    for (const Bank& bk : banks) fx_mode = fx_mode || bk.fx;
    if (fx_mode && mbf % 64 != 0) {  // k_chain renders whole tiles: fall back to the dry banks
        fx_mode = false;
    }
    auto keeps = [&](const Bank& bk, bool fxm) {
        return fxm ? (!bk.width && !bk.rs && !bk.sp && bk.stages <= FW_CHAIN_STAGES - 1 && (int)bk.voices.size() <= 32) : !bk.fx;
    };
    if (fx_mode) {  // no bank with a filter survives the chain plan's rules: the dry banks are voice-bank banks, all of them
        bool any_fx = false;
        for (const Bank& bk : banks) any_fx = any_fx || (bk.fx && keeps(bk, true));
        fx_mode = any_fx;
    }
    // Solo voices (round 4): a voice chain that is NOT a port of a bank — behind a mixer's first non-voice input, on a 2- / 3- /
    // 4-port mixer beside a bus, feeding an effect or two consumers — is a leaf of ONE port whose output is the chain's last node's
    // own pool buffers: a one-port sum is a copy (sum.rs:58-65, out mask passed through), so whatever reads those buffers on the
    // levels sees the node's output and silence flag bit for bit.  The walk goes DOWN from each free source as far as the mode's
    // kernels render (dry mode: gain / program stages, a spatialiser; chain mode: biquad, delay, plain gains), then the bank
    // ports' own validator walks back up; a chain it refuses is tried again one node shorter (a bare source always passes).
    static const bool solo_on = !(getenv("FWGPU_SOLO") && atoi(getenv("FWGPU_SOLO")) == 0);  // FWGPU_SOLO=0: banks only (A/B runs)
    if (solo_on) {
        // a bank the mode's kernels refuse as a whole (a width / clip stage behind a delay, a resampler source beside filters, more than
        // 32 ports on the chain kernels) gives its nodes back: its voices are tried one by one below, as far as the kernels render them
        for (const Bank& bk : banks)
            if (!keeps(bk, fx_mode))
                for (int i : bk.nodes) taken[i] = 0;
        std::vector<int> cons_node(cons.cnt.size(), -1);  // who reads (node, port): meaningful where the count is 1
        for (int i = 0; i < N; ++i) {
            const PlanNode& n = plan.nodes[i];
            for (int p = 0; p < n.n_in; ++p)
                if (n.in_src_node[p] >= 0) cons_node[cons.off[n.in_src_node[p]] + n.in_src_port[p]] = i;
        }
        bool any_kept = false;
        for (const Bank& bk : banks) any_kept = any_kept || keeps(bk, fx_mode);
        if (!any_kept && mbf % 64 == 0)  // no bank chose the mode: solo voices with a filter / delay behind the source choose the chain plan's kernels
            for (int i = 0; i < N && !fx_mode; ++i) {
                const PlanNode& src = plan.nodes[i];
                if (src.kind != K_SAMPLER || src.n_in != 0 || src.n_out != 2 || cons[i][0] != 1 || cons[i][1] != 1) continue;
                const int nx = cons_node[cons.off[i]];
                if (nx < 0 || nx != cons_node[cons.off[i] + 1]) continue;
                const PlanNode& n = plan.nodes[nx];
                fx_mode = (n.kind == K_BIQUAD || (n.kind == K_DELAY && graph.nodes[n.slot].init.loop_end >= 64)) && n.n_in == 2 && n.n_out == 2 &&
                          n.in_src_port[0] == 0 && n.in_src_port[1] == 1;
            }
        for (int i = 0; i < N; ++i) {
            const PlanNode& src = plan.nodes[i];
            if (!(src.kind == K_SAMPLER || src.kind == K_RESAMPLER) || taken[i] || src.n_in != 0 || src.n_out != 2) continue;
            if (fx_mode && src.kind == K_RESAMPLER) continue;  // (the chain plan's source fetch is the sampler's)
            int end = i, stages = 0;
            for (;;) {
                if (cons[end][0] != 1 || cons[end][1] != 1) break;
                const int nx = cons_node[cons.off[end]];
                if (nx < 0 || nx != cons_node[cons.off[end] + 1] || taken[nx]) break;
                const PlanNode& n = plan.nodes[nx];
                if (n.is_graph_io || n.n_in != 2 || n.n_out != 2) break;
                if (n.in_src_node[0] != end || n.in_src_port[0] != 0 || n.in_src_node[1] != end || n.in_src_port[1] != 1) break;
                const bool gain = n.kind == K_VOLUME || n.kind == K_PAN || (fx_mode && n.kind == K_HARD_CLIP);  // (k_chain clips per channel)
                const bool progk = n.kind == K_WIDTH || n.kind == K_HARD_CLIP || n.kind == K_SPATIAL;
                const bool fxk = n.kind == K_BIQUAD || n.kind == K_DELAY;
                if (!(gain || (progk && !fx_mode) || (fxk && fx_mode))) break;
                if (gain || progk) {
                    if (++stages > (fx_mode ? FW_CHAIN_STAGES - 1 : FW_MAX_STAGES - 1)) break;
                }
                end = nx;
            }
            for (int cand = end;; cand = plan.nodes[cand].in_src_node[0]) {
                const PlanNode& e = plan.nodes[cand];
                const Walk w = e.out_buf[1] == e.out_buf[0] + 1 ? walk_voice(cand) : Walk();
                if (w.ok && w.nodes[0] == i) {
                    Bank bk;
                    bk.sum = cand;
                    bk.solo = true;
                    take(bk, w);
                    if (keeps(bk, fx_mode)) {
                        for (int k : bk.nodes) taken[k] = 1;
                        banks.push_back(std::move(bk));
                        break;
                    }
                }
                if (cand == i) break;
            }
        }
    }
    int real_voices = 0;
    for (const Bank& bk : banks) {
        if (!keeps(bk, fx_mode)) continue;
        LeafDesc ld;
        ld.first_voice = (int)fb.voices.size();
        ld.ports = (int)bk.voices.size();
        ld.out_buf = plan.nodes[bk.sum].out_buf[0];
        ld.pad = 0;
        if (bk.split) {  // (out_buf: a partial bus of its own, numbered by install_plan; pad: the node's full port count)
            FusedBuild::Split sp;
            sp.sum = bk.sum;
            sp.leaf = (int)fb.leaves.size();
            sp.lead = ld.ports;
            fb.splits.push_back(sp);
            ld.out_buf = -1;
            ld.pad = plan.nodes[bk.sum].n_in / 2;
        }
        fb.leaves.push_back(ld);
        fb.voices.insert(fb.voices.end(), bk.voices.begin(), bk.voices.end());
        fb.progs.insert(fb.progs.end(), bk.progs.begin(), bk.progs.end());
        fb.has_prog = fb.has_prog || bk.prog || bk.rs;
        fb.has_rs = fb.has_rs || bk.rs;
        fb.has_sp = fb.has_sp || bk.sp;
        fb.has_fx = fb.has_fx || bk.fx;
        fb.min_delay = std::min(fb.min_delay, bk.min_delay);
        fb.max_stages = std::max(fb.max_stages, bk.stages);
        real_voices += bk.real;
        fb.covered.insert(fb.covered.end(), bk.nodes.begin(), bk.nodes.end());
    }
    // a handful of voices is not worth two more launches per batch
    return real_voices >= 8;
}

}  // namespace fwgpu
