// fwgpu_graph.h — host-side mirror of Firewheel's AudioGraph edit API (graph/graph.rs) and the launch
// planner that turns a graph (or an imported CompiledSchedule) into a device plan.
//
// Not a translation of graph/graph/compiler.rs: the reference allocates buffers for a SEQUENTIAL node loop
// (LIFO reuse right after a node is assigned, compiler.rs:110-130,402-404), which would serialise a whole
// level on WAR hazards on the GPU.  Here every output port gets its own buffer id ("renamed"), unconnected
// inputs all read the constant zero buffer 0, and nodes are grouped into topological levels that run as
// one launch each.  Order inside the schedule follows the reference's Kahn BFS (compiler.rs:232-300) so
// introspection matches graph/graph/compiler/schedule.rs's tests.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "fwgpu_types.h"

namespace fwgpu {

struct HostEdge {
    bool alive = false;
    uint32_t gen = 0;
    uint32_t src = 0, dst = 0;  // node slots
    uint32_t sport = 0, dport = 0;
};

struct HostNode {
    bool alive = false;
    uint32_t gen = 0;
    int kind = K_DUMMY;
    uint32_t n_in = 0, n_out = 0;
    bool activated = false;  // has a NodeState on the device
    NodeState init;          // initial audio-half state built from the constructor params
    std::vector<int> in_edge;              // per input port: edge slot or -1
    std::vector<std::vector<int>> out_edges;  // per output port: edge slots (one-to-many)
    // sampler ring occupancy since the last drain (sampler.rs:14 CHANNEL_CAPACITY).  Control-side only: the audio
    // thread publishes a drain epoch, the count restarts when the producer sees a new one.
    int pending_msgs = 0;
    uint64_t pending_epoch = 0;
};

inline int64_t make_id(uint32_t slot, uint32_t gen) { return (int64_t(gen) << 32) | int64_t(slot); }

// IR handed to the executor
// A node's port lists (buffer ids, producers): up to PORTS_INLINE ints sit inside the object, longer lists (wide SumNodes) on the
// heap.  With std::vector a plan of config 3's 16 519 nodes was 66 000 small allocations to build and as many to drop — a third of
// fwgpu_update's time.  Only what the planner uses of a vector.
class PortInts {
public:
    static constexpr uint32_t PORTS_INLINE = 4;
    PortInts() = default;
    PortInts(const PortInts& o) { copy_from(o.data(), o.n_); }
    PortInts(PortInts&& o) noexcept { steal(o); }
    PortInts& operator=(const PortInts& o) {
        if (this != &o) copy_from(o.data(), o.n_);
        return *this;
    }
    PortInts& operator=(PortInts&& o) noexcept {
        if (this != &o) {
            drop();
            steal(o);
        }
        return *this;
    }
    ~PortInts() { drop(); }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    int* data() { return heap_ ? heap_ : inl_; }
    const int* data() const { return heap_ ? heap_ : inl_; }
    int& operator[](size_t i) { return data()[i]; }
    const int& operator[](size_t i) const { return data()[i]; }
    int* begin() { return data(); }
    int* end() { return data() + n_; }
    const int* begin() const { return data(); }
    const int* end() const { return data() + n_; }
    void assign(size_t n, int v) {
        reserve_exact(n, false);
        n_ = (uint32_t)n;
        int* d = data();
        for (size_t i = 0; i < n; ++i) d[i] = v;
    }
    int* reset(size_t n) {  // n elements of unspecified value; returns data()
        reserve_exact(n, false);
        n_ = (uint32_t)n;
        return data();
    }
    void resize(size_t n) {  // new elements are 0
        const uint32_t old = n_;
        reserve_exact(n, true);
        int* d = data();
        for (size_t i = old; i < n; ++i) d[i] = 0;
        n_ = (uint32_t)n;
    }
    operator std::vector<int>() const { return std::vector<int>(begin(), end()); }

private:
    void reserve_exact(size_t n, bool keep) {
        const uint32_t cap = heap_ ? cap_ : PORTS_INLINE;
        if (n <= cap) return;
        int* h = new int[n];
        if (keep) {
            const int* d = data();
            for (uint32_t i = 0; i < n_; ++i) h[i] = d[i];
        }
        delete[] heap_;
        heap_ = h;
        cap_ = (uint32_t)n;
    }
    void copy_from(const int* src, uint32_t n) {
        reserve_exact(n, false);
        int* d = data();
        for (uint32_t i = 0; i < n; ++i) d[i] = src[i];
        n_ = n;
    }
    void steal(PortInts& o) {
        heap_ = o.heap_;
        cap_ = o.cap_;
        n_ = o.n_;
        for (uint32_t i = 0; i < PORTS_INLINE; ++i) inl_[i] = o.inl_[i];
        o.heap_ = nullptr;
        o.cap_ = 0;
        o.n_ = 0;
    }
    void drop() {
        delete[] heap_;
        heap_ = nullptr;
        cap_ = 0;
        n_ = 0;
    }
    int* heap_ = nullptr;
    uint32_t n_ = 0, cap_ = 0;
    int inl_[PORTS_INLINE] = {0, 0, 0, 0};
};

struct PlanNode {
    uint32_t slot;
    int kind;
    int n_in, n_out;
    int level;
    int is_graph_io;            // 0, 1 = graph_in, 2 = graph_out
    PortInts in_buf;    // renamed buffer id, 0 = unconnected (zero buffer)
    PortInts out_buf;
    PortInts in_src_node;  // index into Plan::nodes of the producer, -1 = unconnected
    PortInts in_src_port;
};

struct Plan {
    std::vector<PlanNode> nodes;  // schedule order: graph_in first, graph_out last, Kahn BFS between
    int num_buffers = 1;          // including the zero buffer
    int num_levels = 0;
};

class HostGraph {
  public:
    HostGraph(uint32_t n_graph_in, uint32_t n_graph_out);
    uint32_t graph_in_slot, graph_out_slot;
    std::vector<HostNode> nodes;
    // what the compiler reads of a node, 16 bytes apiece beside the 200-byte HostNodes (kept by add_node / remove_node)
    struct NodeMeta {
        int kind;
        uint32_t n_in, n_out;
        bool alive;
    };
    std::vector<NodeMeta> meta;
    std::vector<uint32_t> free_nodes;
    std::vector<uint32_t>* limbo = nullptr;  // when set, remove_node parks the freed slot there instead of free_nodes
    std::vector<HostEdge> edges;
    std::vector<uint32_t> free_edges;
    bool needs_compile = true;
    bool canonical_order = true;  // build_plan: level by level, slots ascending (FWGPU_PLAN_ORDER=reference: the reference's Kahn order)
    std::vector<uint32_t> nodes_to_activate;

    HostNode* get(int64_t id);
    int64_t id_of(uint32_t slot) const { return make_id(slot, nodes[slot].gen); }

    int64_t add_node(int kind, uint32_t n_in, uint32_t n_out, const NodeState& init);
    int remove_node(int64_t id);
    int64_t connect(int64_t src, uint32_t sport, int64_t dst, uint32_t dport, bool check_cycles);
    int disconnect(int64_t src, uint32_t sport, int64_t dst, uint32_t dport);
    int disconnect_edge(int64_t edge);
    bool cycle_detected();

    // Kahn BFS in the reference's order.  Returns false on a cycle.
    bool topo_order(std::vector<uint32_t>& order);
    // graph -> Plan.  0 or a CompileGraphError code; err gets the message.
    int build_plan(Plan& plan, std::string& err);

  private:
    void remove_edge_slot(uint32_t e);
    // the compiler's scratch arrays, kept between compiles (topo_order / build_plan say what each holds when)
    std::vector<uint32_t> cs_off, cs_adj, cs_cur, cs_queue, cs_order, cs_byl, cs_fill, cs_insrc;
    std::vector<uint8_t> cs_seen;
    std::vector<int> cs_indeg;
};

// node activation checks (AudioNode::activate of each kind: volume.rs:56-66, sum.rs:20-30,
// hard_clip.rs:30-40).  Returns false + message on failure.
bool check_activation(int kind, uint32_t n_in, uint32_t n_out, std::string& err);

// levelise + rename: fills level, in_buf/out_buf from in_src_*; nodes must be topologically ordered.
void finalize_plan(Plan& plan);

}  // namespace fwgpu
