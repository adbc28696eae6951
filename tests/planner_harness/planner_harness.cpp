// Test harness (CPU tier): the PRODUCT's host-side graph mirror + launch planner (firewheel_amd/csrc/fwgpu_graph.cpp —
// plain C++, no HIP) behind a tiny C surface, so that `pytest -m "not gpu"` can run the reference's routing KATs
// (graph/graph/compiler/schedule.rs:407-710) and random-DAG cross-checks against the oracle's compiler on it.
// Built by tests/fwapi.py with g++ into tests/planner_harness/_planner.so; nothing here ships in libfwgpu.so.
#include <string.h>

#include <string>

#include "../../firewheel_amd/csrc/fwgpu_graph.h"

using namespace fwgpu;

struct Harness {
    HostGraph g;
    Plan plan;
    std::string err;
    Harness(uint32_t gin, uint32_t gout) : g(gin, gout) {}
};

extern "C" {
void* fwp_new(uint32_t gin, uint32_t gout) { return new Harness(gin, gout); }
void fwp_free(void* h) { delete (Harness*)h; }
const char* fwp_last_error(void* h) { return ((Harness*)h)->err.c_str(); }
int64_t fwp_graph_in_node(void* h) { return ((Harness*)h)->g.id_of(((Harness*)h)->g.graph_in_slot); }
int64_t fwp_graph_out_node(void* h) { return ((Harness*)h)->g.id_of(((Harness*)h)->g.graph_out_slot); }
int64_t fwp_add_node(void* h, int kind, uint32_t n_in, uint32_t n_out) {
    NodeState st;
    memset(&st, 0, sizeof(st));
    return ((Harness*)h)->g.add_node(kind, n_in, n_out, st);
}
int fwp_remove_node(void* h, int64_t id) { return ((Harness*)h)->g.remove_node(id); }
int64_t fwp_connect(void* h, int64_t s, uint32_t sp, int64_t d, uint32_t dp, int check) {
    return ((Harness*)h)->g.connect(s, sp, d, dp, check != 0);
}
int fwp_disconnect(void* h, int64_t s, uint32_t sp, int64_t d, uint32_t dp) { return ((Harness*)h)->g.disconnect(s, sp, d, dp); }
int fwp_disconnect_edge(void* h, int64_t e) { return ((Harness*)h)->g.disconnect_edge(e); }
void fwp_set_canonical_order(void* h, int on) { ((Harness*)h)->g.canonical_order = on != 0; }  // build_plan's table order (round 6)
int fwp_cycle_detected(void* h) { return ((Harness*)h)->g.cycle_detected() ? 1 : 0; }
int fwp_update(void* h) {
    Harness* x = (Harness*)h;
    x->err.clear();
    return x->g.build_plan(x->plan, x->err);
}
int fwp_sched_len(void* h) { return (int)((Harness*)h)->plan.nodes.size(); }
int fwp_sched_num_buffers(void* h) { return ((Harness*)h)->plan.num_buffers; }
int fwp_sched_num_levels(void* h) { return ((Harness*)h)->plan.num_levels; }
int64_t fwp_sched_node(void* h, int i) { return ((Harness*)h)->g.id_of(((Harness*)h)->plan.nodes[i].slot); }
int fwp_sched_level(void* h, int i) { return ((Harness*)h)->plan.nodes[i].level; }
int fwp_sched_in(void* h, int i, int* buf, int* clear, int cap) {
    const PlanNode& n = ((Harness*)h)->plan.nodes[i];
    for (int p = 0; p < n.n_in && p < cap; ++p) {
        buf[p] = n.in_buf[p];
        clear[p] = n.in_src_node[p] < 0 ? 1 : 0;  // InBufferAssignment.should_clear: the constant zero buffer (id 0)
    }
    return n.n_in;
}
int fwp_sched_out(void* h, int i, int* buf, int cap) {
    const PlanNode& n = ((Harness*)h)->plan.nodes[i];
    for (int p = 0; p < n.n_out && p < cap; ++p) buf[p] = n.out_buf[p];
    return n.n_out;
}
}
