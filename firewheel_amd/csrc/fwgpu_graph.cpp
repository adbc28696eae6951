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
    // edge-slot order, graph_out forced last)
    order.clear();
    order.reserve(nodes.size());
    std::vector<int> in_degree(nodes.size(), 0);
    size_t alive = 0;
    for (const HostEdge& e : edges)
        if (e.alive) in_degree[e.dst] += 1;
    std::vector<uint32_t> queue;  // (a FIFO that only grows: read through `head`)
    queue.reserve(nodes.size());
    size_t head = 0;
    queue.push_back(graph_in_slot);
    for (uint32_t s = 0; s < nodes.size(); ++s) {
        if (!nodes[s].alive) continue;
        alive++;
        if (s == graph_in_slot) continue;
        bool has_in = false;
        for (int e : nodes[s].in_edge)
            if (e >= 0) has_in = true;
        if (!has_in) queue.push_back(s);
    }
    size_t visited = 0;
    std::vector<int> outs;
    while (head < queue.size()) {
        uint32_t s = queue[head++];
        visited++;
        // outgoing edges in edge-slot order (the reference collects them by iterating the edge arena)
        outs.clear();
        for (const auto& pe : nodes[s].out_edges)
            for (int e : pe) outs.push_back(e);
        std::sort(outs.begin(), outs.end());
        for (int e : outs) {
            uint32_t d = edges[e].dst;
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
    for (const HostNode& n : nodes)
        if (n.alive && (n.n_in > 64 || n.n_out > 64)) {  // compiler.rs:202-203 (assert in the reference)
            err = "a node has more than 64 ports";
            return FWGPU_ERR_INVALID;
        }
    std::vector<uint32_t> order;
    if (!topo_order(order)) {
        err = "cycle detected";
        return FWGPU_ERR_COMPILE_CYCLE;
    }
    for (uint32_t slot : order) {
        const HostNode& n = nodes[slot];
        if (!check_activation(n.kind, n.n_in, n.n_out, err)) return FWGPU_ERR_NODE_ACTIVATION_FAILED;
    }
    std::vector<int> index_of(nodes.size(), -1);
    plan = Plan();
    plan.nodes.reserve(order.size());
    for (uint32_t slot : order) {
        index_of[slot] = (int)plan.nodes.size();
        PlanNode pn;
        const HostNode& n = nodes[slot];
        pn.slot = slot;
        pn.kind = n.kind;
        pn.n_in = (int)n.n_in;
        pn.n_out = (int)n.n_out;
        pn.level = 0;
        pn.is_graph_io = slot == graph_in_slot ? 1 : (slot == graph_out_slot ? 2 : 0);
        plan.nodes.push_back(pn);
    }
    for (PlanNode& pn : plan.nodes) {
        const HostNode& n = nodes[pn.slot];
        pn.in_src_node.assign(pn.n_in, -1);
        pn.in_src_port.assign(pn.n_in, 0);
        for (int p = 0; p < pn.n_in; ++p) {
            int e = n.in_edge[p];
            if (e < 0) continue;
            pn.in_src_node[p] = index_of[edges[e].src];
            pn.in_src_port[p] = (int)edges[e].sport;
        }
    }
    finalize_plan(plan);
    return 0;
}

}  // namespace fwgpu
