// TEST DOUBLE (see fakehip/hip/hip_runtime_api.h): the kernel launch wrappers of fwgpu_kernels.hip as counted no-ops.
// Nothing is computed; the harness only lets the host half of libfwgpu run on the CPU tier.
#include "../../firewheel_amd/csrc/fwgpu_launch.h"

namespace {
unsigned long long g_launches[8];  // 0 level, 1 voice_control, 2 leaf_sum, 3 chain, 4 bus_sum, 5 root_out, 6 fir, 7 other
}
extern "C" unsigned long long fwh_launch_count(int which) { return which >= 0 && which < 8 ? g_launches[which] : 0; }
extern "C" void fwh_launch_reset(void) {
    for (auto& x : g_launches) x = 0;
}

namespace fwgpu {
int launch_level(hipStream_t, const DevView&, const int*, int, int, uint32_t, int) { g_launches[0]++; return 0; }
int launch_frozen_scan(hipStream_t, const DevView&, int, uint32_t, int, uint8_t*, unsigned long long*) { g_launches[7]++; return 0; }
int launch_bus_sum(hipStream_t, const DevView&, const int*, int, int, int) { g_launches[4]++; return 0; }
int launch_root_out(hipStream_t, const DevView&, const RootArgs&, float*, int) { g_launches[5]++; return 0; }
int launch_ir_convert(hipStream_t, const SampleDesc*, int, int, float*, uint32_t) { g_launches[7]++; return 0; }
int launch_fir(hipStream_t, const DevView&, const FirRow*, int, const uint32_t*, uint32_t, float*, size_t, int, hipEvent_t, hipEvent_t) { g_launches[6]++; return 0; }
int launch_single_node(hipStream_t, const DevView&, int) { g_launches[7]++; return 0; }
int launch_scatter_states(hipStream_t, NodeState*, const void*, int) { g_launches[7]++; return 0; }
int launch_graph_in(hipStream_t, float*, uint8_t*, int, size_t, size_t, const int*, int, const float*, int, int, int) { g_launches[7]++; return 0; }
int launch_graph_out(hipStream_t, const float*, const uint8_t*, int, size_t, size_t, const int*, int, float*, int, int, int) { g_launches[7]++; return 0; }
int launch_set_flags(hipStream_t, uint8_t*, const int*, int, uint64_t) { g_launches[7]++; return 0; }
int launch_get_flags(hipStream_t, const uint8_t*, const int*, int, uint64_t*) { g_launches[7]++; return 0; }
int launch_voice_control(hipStream_t, const FusedView&, int, uint32_t) { g_launches[1]++; return 0; }
int launch_leaf_sum(hipStream_t, const FusedView&, int) { g_launches[2]++; return 0; }
int launch_chain(hipStream_t, const FusedView&, int, uint32_t, int) { g_launches[3]++; return 0; }
int launch_scatter_ext(hipStream_t, float*, const void*, int) { g_launches[7]++; return 0; }
}  // namespace fwgpu
