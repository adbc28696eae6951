// fwgpu_graph.cpp — AudioGraph mirror + launch planner (see fwgpu_graph.h).
#include "fwgpu_graph.h"

#include <algorithm>
#include <deque>

#include "../../include/fwgpu.h"

namespace fwgpu {

HostGraph::HostGraph(uint32_t n_graph_in, uint32_t n_graph_out) {
    // graph/graph.rs:125-168: graph_in = Dummy(0 -> n_graph_in), graph_out = Dummy(n_graph_out -> 0)
    NodeState z{};
    graph_in_slot = (uint32_t)(add_node(K_DUMMY, 0, n_graph_in, z) & 0xffffffff);
    graph_out_slot = (uint32_t)(add_node(K_DUMMY, n_graph_out, 0, z) & 0xffffffff);
}

HostNode* HostGraph::get(int64_t id) {
    if (id < 0) return nullptr;
    uint32_t slot = (uint32_t)(id & 0xffffffff), gen = (uint32_t)((uint64_t)id >> 32);
    if (slot >= nodes.size()) return nullptr;
    HostNode& n = nodes[slot];
    if (!n.alive || n.gen != gen) return nullptr;
    return &n;
}

int64_t HostGraph::add_node(int kind, uint32_t n_in, uint32_t n_out, const NodeState& init) {
    // graph.rs:201-231 (Q25: no validation against AudioNodeInfo; the <=64 assert fires at compile time)
    uint32_t slot;
    if (!free_nodes.empty()) {
        slot = free_nodes.back();
        free_nodes.pop_back();
    } else {
        slot = (uint32_t)nodes.size();
        nodes.emplace_back();
    }
    HostNode& n = nodes[slot];
    uint32_t gen = n.gen + 1;
    n = HostNode();
    n.alive = true;
    n.gen = gen;
    n.kind = kind;
    n.n_in = n_in;
    n.n_out = n_out;
    n.init = init;
    n.in_edge.assign(n_in, -1);
    n.out_edges.assign(n_out, std::vector<int>());
    if (meta.size() < nodes.size()) meta.resize(nodes.size());
    meta[slot] = NodeMeta{kind, n_in, n_out, true};
    nodes_to_activate.push_back(slot);
    needs_compile = true;
    return make_id(slot, gen);
}

void HostGraph::remove_edge_slot(uint32_t e) {
    HostEdge& ed = edges[e];
    if (!ed.alive) return;
    nodes[ed.dst].in_edge[ed.dport] = -1;
    auto& oe = nodes[ed.src].out_edges[ed.sport];
    oe.erase(std::remove(oe.begin(), oe.end(), (int)e), oe.end());
    ed.alive = false;
    free_edges.push_back(e);
    needs_compile = true;
}

int HostGraph::remove_node(int64_t id) {
    // graph.rs:268-299
    HostNode* n = get(id);
    if (!n) return FWGPU_ERR_INVALID;
    uint32_t slot = (uint32_t)(id & 0xffffffff);
    if (slot == graph_in_slot || slot == graph_out_slot) return FWGPU_ERR_INVALID;
    for (uint32_t p = 0; p < n->n_in; ++p)
        if (n->in_edge[p] >= 0) remove_edge_slot((uint32_t)n->in_edge[p]);
    for (uint32_t p = 0; p < n->n_out; ++p) {
        std::vector<int> es = n->out_edges[p];
        std::sort(es.begin(), es.end());  // arena (slot) order, as graph.rs:529-546 walks it: keeps later EdgeIDs the reference's
        for (int e : es) remove_edge_slot((uint32_t)e);
    }
    n->alive = false;
    n->activated = false;
    meta[slot].alive = false;
    if (limbo) limbo->push_back(slot);  // reusable only once no running plan holds the node (fwgpu_plan_install.cpp)
    else free_nodes.push_back(slot);
    nodes_to_activate.erase(std::remove(nodes_to_activate.begin(), nodes_to_activate.end(), slot), nodes_to_activate.end());
    needs_compile = true;
    return 0;
}

int64_t HostGraph::connect(int64_t src, uint32_t sport, int64_t dst, uint32_t dport, bool check_cycles) {
    // graph.rs:396-477 — same checks in the same order
    HostNode* s = get(src);
    if (!s) return FWGPU_ERR_SRC_NODE_NOT_FOUND;
    HostNode* d = get(dst);
    if (!d) return FWGPU_ERR_DST_NODE_NOT_FOUND;
    if (sport >= s->n_out) return FWGPU_ERR_OUT_PORT_OUT_OF_RANGE;
    if (dport >= d->n_in) return FWGPU_ERR_IN_PORT_OUT_OF_RANGE;
    uint32_t sslot = (uint32_t)(src & 0xffffffff), dslot = (uint32_t)(dst & 0xffffffff);
    if (sslot == dslot) return FWGPU_ERR_CYCLE_DETECTED;
    if (d->in_edge[dport] >= 0) {
        const HostEdge& e = edges[d->in_edge[dport]];
        if (e.src == sslot && e.sport == sport) return FWGPU_ERR_EDGE_ALREADY_EXISTS;
        return FWGPU_ERR_INPUT_PORT_ALREADY_CONNECTED;
    }
    uint32_t eslot;
    if (!free_edges.empty()) {
        eslot = free_edges.back();
        free_edges.pop_back();
    } else {
        eslot = (uint32_t)edges.size();
        edges.emplace_back();
    }
    HostEdge& e = edges[eslot];
    e.alive = true;
    e.gen += 1;
    e.src = sslot;
    e.dst = dslot;
    e.sport = sport;
    e.dport = dport;
    d->in_edge[dport] = (int)eslot;
    s->out_edges[sport].push_back((int)eslot);
    if (check_cycles && cycle_detected()) {
        // the reference removes the edge but leaves its bookkeeping maps populated (graph.rs:466-471);
        // that leak is a control-plane bug we do not reproduce: the edit is rolled back completely.
        remove_edge_slot(eslot);
        return FWGPU_ERR_CYCLE_DETECTED;
    }
    needs_compile = true;
    return make_id(eslot, e.gen);
}

int HostGraph::disconnect(int64_t src, uint32_t sport, int64_t dst, uint32_t dport) {
    HostNode* d = get(dst);
    HostNode* s = get(src);
    if (!d || !s || dport >= d->n_in || d->in_edge[dport] < 0) return 0;
    const HostEdge& e = edges[d->in_edge[dport]];
    if (e.src != (uint32_t)(src & 0xffffffff) || e.sport != sport) return 0;
    remove_edge_slot((uint32_t)d->in_edge[dport]);
    return 1;
}

int HostGraph::disconnect_edge(int64_t edge) {
    if (edge < 0) return 0;
    uint32_t slot = (uint32_t)(edge & 0xffffffff), gen = (uint32_t)((uint64_t)edge >> 32);
    if (slot >= edges.size() || !edges[slot].alive || edges[slot].gen != gen) return 0;
    remove_edge_slot(slot);
    return 1;
}

bool HostGraph::topo_order(std::vector<uint32_t>& order) {
    // graph/graph/compiler.rs:232-300 (Kahn BFS: graph_in first, other roots in slot order, out-edges in
    // edge-slot order, graph_out forced last).
    // The walk never touches a HostNode (200 bytes apiece, vectors inside): one sequential pass over the edge arena builds a
    // CSR adjacency by source — an edge arena walked in slot order leaves every node's list in edge-slot order, the order the
    // reference collects them in — and the in-degrees; per node only `meta` (8 bytes) is read.
    const size_t NS = nodes.size();
    order.clear();
    order.reserve(NS);
    std::vector<uint32_t>& off = cs_off;
    std::vector<uint32_t>& adj = cs_adj;
    std::vector<int>& in_degree = cs_indeg;
    off.assign(NS + 1, 0);
    in_degree.assign(NS, 0);
    for (const HostEdge& e : edges)
        if (e.alive) {
            off[e.src + 1]++;
            in_degree[e.dst]++;
        }
    for (size_t s = 0; s < NS; ++s) off[s + 1] += off[s];
    adj.resize(off[NS]);
    cs_cur.assign(off.begin(), off.end() - 1);
    for (const HostEdge& e : edges)
        if (e.alive) adj[cs_cur[e.src]++] = e.dst;
    std::vector<uint32_t>& queue = cs_queue;  // (a FIFO that only grows: read through `head`)
    queue.clear();
    queue.reserve(NS);
    size_t head = 0, alive = 0;
    queue.push_back(graph_in_slot);
    for (uint32_t s = 0; s < NS; ++s) {
        if (!meta[s].alive) continue;
        alive++;
        if (s != graph_in_slot && in_degree[s] == 0) queue.push_back(s);
    }
    size_t visited = 0;
    while (head < queue.size()) {
        const uint32_t s = queue[head++];
        visited++;
        for (uint32_t k = off[s]; k < off[s + 1]; ++k) {
            const uint32_t d = adj[k];
            if (--in_degree[d] == 0) queue.push_back(d);
        }
        if (s != graph_out_slot) order.push_back(s);
    }
    order.push_back(graph_out_slot);
    return visited == alive;
}

bool HostGraph::cycle_detected() {
    std::vector<uint32_t> order;
    return !topo_order(order);
}

bool check_activation(int kind, uint32_t n_in, uint32_t n_out, std::string& err) {
    switch (kind) {
        case K_VOLUME:  // volume.rs:63-65
            if (n_in != n_out) {
                err = "The number of inputs on a VolumeNode node must equal the number of outputs. Got num_inputs: " +
                      std::to_string(n_in) + ", num_outputs: " + std::to_string(n_out);
                return false;
            }
            return true;
        case K_SUM:  // sum.rs:27-29
            if (n_out == 0 || n_in % n_out != 0) {
                err = "The number of inputs on a SumNode must be a multiple of the number of outputs. Got num_inputs: " +
                      std::to_string(n_in) + ", num_outputs: " + std::to_string(n_out);
                return false;
            }
            return true;
        case K_HARD_CLIP:  // hard_clip.rs:37-39
            if (n_in != n_out) {
                err = "The number of inputs on a HardClip node must equal the number of outputs. Got num_inputs: " +
                      std::to_string(n_in) + ", num_outputs: " + std::to_string(n_out);
                return false;
            }
            return true;
        case K_PAN:
            if (n_in != 2 || n_out != 2) {
                err = "StereoPanNode needs exactly 2 inputs and 2 outputs.";
                return false;
            }
            return true;
        case K_WIDTH:
            if (n_in != 2 || n_out != 2) {
                err = "StereoWidthNode needs exactly 2 inputs and 2 outputs.";
                return false;
            }
            return true;
        case K_BIQUAD:
        case K_DELAY:
        case K_FIR:
            if (n_in != n_out || n_in == 0) {
                err = "Biquad/Delay nodes need as many outputs as inputs (>= 1).";
                return false;
            }
            return true;
        case K_RESAMPLER:
            if (n_in != 0 || n_out == 0) {
                err = "Resampler node is a source: 0 inputs, >= 1 outputs and a source sample.";
                return false;
            }
            return true;
        case K_SPATIAL:
            if (!(n_in == 1 || n_in == 2) || n_out != 2) {
                err = "Spatial node needs 1 or 2 inputs and exactly 2 outputs.";
                return false;
            }
            return true;
        case K_MONO_TO_STEREO:
            if (n_in < 1 || n_out < 2) {
                err = "MonoToStereoNode needs 1 input and 2 outputs.";
                return false;
            }
            return true;
        default:
            return true;
    }
}

void finalize_plan(Plan& plan) {
    int next_buf = 1;  // 0 = constant zero buffer
    int max_level = 0;
    for (PlanNode& n : plan.nodes) {
        n.out_buf.resize(n.n_out);
        for (int p = 0; p < n.n_out; ++p) n.out_buf[p] = next_buf++;
    }
    for (PlanNode& n : plan.nodes) {
        n.in_buf.assign(n.n_in, 0);
        int lvl = 0;
        for (int p = 0; p < n.n_in; ++p) {
            int src = n.in_src_node[p];
            if (src < 0) continue;
            n.in_buf[p] = plan.nodes[src].out_buf[n.in_src_port[p]];
            lvl = std::max(lvl, plan.nodes[src].level + 1);
        }
        n.level = lvl;
        if (n.is_graph_io != 2) max_level = std::max(max_level, lvl);
    }
    // graph_out closes the schedule (compiler.rs:286-292)
    if (!plan.nodes.empty() && plan.nodes.back().is_graph_io == 2) {
        plan.nodes.back().level = std::max(plan.nodes.back().level, max_level + 1);
        max_level = plan.nodes.back().level;
    }
    plan.num_buffers = next_buf;
    plan.num_levels = max_level + 1;
}

int HostGraph::build_plan(Plan& plan, std::string& err) {
    const size_t NS = nodes.size();
    for (size_t s = 0; s < NS; ++s)
        if (meta[s].alive && (meta[s].n_in > 64 || meta[s].n_out > 64)) {  // compiler.rs:202-203 (assert in the reference)
            err = "a node has more than 64 ports";
            return FWGPU_ERR_INVALID;
        }
    std::vector<uint32_t>& order = cs_order;
    if (!canonical_order) {  // the reference's walk (cycle check included); the canonical order below makes its own
        if (!topo_order(order)) {
            err = "cycle detected";
            return FWGPU_ERR_COMPILE_CYCLE;
        }
    } else {
        order.clear();
        for (uint32_t s = 0; s < NS; ++s)
            if (meta[s].alive) order.push_back(s);  // (which nodes: the walk below puts them in order)
    }
    // producers per (node, input port), flat, from one more sequential pass over the edge arena
    std::vector<uint32_t>& in_off = cs_off;  // (the walk is over: its arrays are free)
    in_off.assign(NS + 1, 0);
    for (size_t s = 0; s < NS; ++s) in_off[s + 1] = in_off[s] + (meta[s].alive ? meta[s].n_in : 0u);
    std::vector<int>& in_e = cs_indeg;
    in_e.assign(in_off[NS], -1);
    std::vector<uint32_t>& in_s = cs_insrc;  // ... and the producer's slot beside it (the passes below then leave the edge arena alone)
    if (canonical_order) in_s.resize(in_off[NS]);
    for (size_t e = 0; e < edges.size(); ++e)
        if (edges[e].alive) {
            const uint32_t at = in_off[edges[e].dst] + edges[e].dport;
            in_e[at] = (int)e;
            if (canonical_order) in_s[at] = edges[e].src;
        }
    std::vector<uint32_t>& index_of = cs_cur;
    index_of.assign(NS, 0);
    const size_t N = order.size();
    bool cyclic = false;
    // Round 6 — a CANONICAL order for the plan's tables: the post-order of a depth-first walk from graph_out over the INPUT ports,
    // ascending (a node behind all of its producers: a schedule; nodes graph_out does not reach: the same walk from each, slots
    // ascending, behind the rest).  The reference's order (the Kahn walk above: compiler.rs:232-300) follows the edge arena, so
    // replacing one voice — its new edges go to the arena's end — moved that voice to the end of its level and shifted every node,
    // buffer id and voice behind it by one voice: 0.4-0.6 MB of the 1.1 MB of tables differed after every edit of config 3's 4 096
    // voices (FWGPU_UPDATE_PROF=2); ordering by slot would do the same (a removed node's slot is reusable only when no plan holds it
    // any more: the replacement gets new slots).  Every table names its producers by index and levels follow from the producers, so
    // any schedule renders the same audio; this one depends on WHERE a node is connected only: a voice put into the mixer port of
    // the voice it replaces takes that voice's entries, and an edit differs in O(1) chunks.  One pass, O(nodes + edges), no sort.
    if (canonical_order && N >= 2) {
        std::vector<uint32_t>& out = cs_byl;
        std::vector<uint8_t>& seen = cs_seen;
        std::vector<uint32_t>& st_node = cs_adj;  // (the walk's adjacency is spent) the DFS stack: node, next input port
        std::vector<uint32_t>& st_port = cs_fill;
        out.resize(N);
        seen.assign(NS, 0);
        st_node.resize(N + 1);
        st_port.resize(N + 1);
        // (raw pointers: the walk is a third of the compile as it is)
        uint32_t* const o = out.data();
        uint8_t* const sn = seen.data();
        uint32_t* const sk_n = st_node.data();
        uint32_t* const sk_p = st_port.data();
        const int* const ine = in_e.data();
        const uint32_t* const ins = in_s.data();
        const uint32_t* const ioff = in_off.data();
        const NodeMeta* const mt = meta.data();
        size_t pos = 0;
        o[pos++] = graph_in_slot;  // (first, as in the reference's order)
        // (the walk is the cycle check too — the reference's Kahn walk is not run in this mode: a node met again while it is still ON the
        //  stack (1; done nodes are 2) closes a cycle, and every alive node is walked)
        sn[graph_in_slot] = 2;
        auto walk = [&](uint32_t root) {
            size_t sp = 0;
            sk_n[0] = root;
            sk_p[0] = 0;
            sn[root] = 1;
            for (;;) {
                const uint32_t n = sk_n[sp];
                uint32_t p = sk_p[sp];
                const uint32_t n_in = mt[n].n_in;
                const int* ie = ine + ioff[n];
                const uint32_t* is = ins + ioff[n];
                while (p < n_in && (ie[p] < 0 || sn[is[p]] == 2)) ++p;
                if (p < n_in) {
                    if (sn[is[p]] == 1) {
                        cyclic = true;
                        return;
                    }
                    sk_p[sp] = p + 1;
                    sn[is[p]] = 1;
                    ++sp;
                    sk_n[sp] = is[p];
                    sk_p[sp] = 0;
                } else {
                    sn[n] = 2;
                    if (n != graph_out_slot) o[pos++] = n;
                    if (sp == 0) break;
                    --sp;
                }
            }
        };
        walk(graph_out_slot);
        if (!cyclic && pos + 1 < N)
            for (uint32_t slot = 0; slot < NS && !cyclic; ++slot)
                if (mt[slot].alive && !sn[slot]) walk(slot);
        if (cyclic) {
            err = "cycle detected";
            return FWGPU_ERR_COMPILE_CYCLE;
        }
        out[pos++] = graph_out_slot;
        order.swap(out);
    }
    for (uint32_t slot : order) {
        const NodeMeta& m = meta[slot];
        if (!check_activation(m.kind, m.n_in, m.n_out, err)) return FWGPU_ERR_NODE_ACTIVATION_FAILED;
    }
    for (size_t i = 0; i < N; ++i) index_of[order[i]] = (uint32_t)i;
    // a recycled Plan keeps its node array AND the nodes in it (every field is assigned below; wide nodes keep their heap lists):
    // a fresh array was 3 MB of first-touch page faults and 20 000 constructor / destructor pairs on config 3
    plan.nodes.resize(N);
    std::vector<uint32_t>& out_base = cs_queue;  // renamed buffer id of output port 0, per plan node (the order array is in cs_order)
    out_base.assign(N, 0);
    // levelise + rename in the same pass (finalize_plan's arithmetic: a node's producers come before it in the order)
    int next_buf = 1, max_level = 0;  // 0 = constant zero buffer
    for (size_t i = 0; i < N; ++i) {
        const uint32_t slot = order[i];
        const NodeMeta& m = meta[slot];
        PlanNode& pn = plan.nodes[i];
        pn.slot = slot;
        pn.kind = m.kind;
        pn.n_in = (int)m.n_in;
        pn.n_out = (int)m.n_out;
        pn.is_graph_io = slot == graph_in_slot ? 1 : (slot == graph_out_slot ? 2 : 0);
        out_base[i] = (uint32_t)next_buf;
        int* ob = pn.out_buf.reset(m.n_out);
        for (uint32_t p = 0; p < m.n_out; ++p) ob[p] = next_buf++;
        int* sn = pn.in_src_node.reset(m.n_in);
        int* sp = pn.in_src_port.reset(m.n_in);
        int* ib = pn.in_buf.reset(m.n_in);
        int lvl = 0;
        const int* ie = in_e.data() + in_off[slot];
        for (uint32_t p = 0; p < m.n_in; ++p) {
            if (ie[p] < 0) {
                sn[p] = -1;
                sp[p] = 0;
                ib[p] = 0;
                continue;
            }
            const HostEdge& e = edges[ie[p]];
            const uint32_t src = index_of[e.src];
            sn[p] = (int)src;
            sp[p] = (int)e.sport;
            ib[p] = (int)(out_base[src] + e.sport);
            lvl = std::max(lvl, plan.nodes[src].level + 1);
        }
        pn.level = lvl;
        if (pn.is_graph_io != 2) max_level = std::max(max_level, lvl);
    }
    // graph_out closes the schedule (compiler.rs:286-292)
    if (N && plan.nodes.back().is_graph_io == 2) {
        plan.nodes.back().level = std::max(plan.nodes.back().level, max_level + 1);
        max_level = plan.nodes.back().level;
    }
    plan.num_buffers = next_buf;
    plan.num_levels = max_level + 1;
    return 0;
}

}  // namespace fwgpu
