// fakerccl.cpp — CPU tier only: the six RCCL entry points libfwgpu dlopen's (fwgpu_rccl.cpp), for ranks that are THREADS of one
// process on the host-only harness (fake HIP runtime: "device" pointers are host pointers, streams are inert, so every collective is
// a rendezvous of `world` threads and a loop).  Never linked into the product; FWGPU_RCCL_LIB points the harness library at it.
// The all-reduce adds in rank order 0, 1, ... — one of the orders a real ring may produce; tests that compare it with a reference sum
// use that order.
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<const void*> send;
    std::vector<void*> recv;
};
struct Comm {
    std::shared_ptr<Group> g;
    int rank;
};
std::mutex g_mu;
std::map<std::string, std::shared_ptr<Group>> g_groups;
std::atomic<uint64_t> g_next{1};
std::atomic<uint64_t> g_calls{0};

template <class F>
void collective(Comm* c, const void* send, void* recv, F&& op) {
    Group& g = *c->g;
    std::unique_lock<std::mutex> lk(g.mu);
    g.send[c->rank] = send;
    g.recv[c->rank] = recv;
    if (++g.arrived == g.world) {
        op(g);
        g.arrived = 0;
        g.gen++;
        g.cv.notify_all();
    } else {
        const uint64_t gen = g.gen;
        g.cv.wait(lk, [&] { return g.gen != gen; });
    }
}
}  // namespace

struct ncclUniqueId {
    char internal[128];
};

extern "C" {
int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    const uint64_t n = g_next++;
    memcpy(id->internal, "FAKERCCL", 8);
    memcpy(id->internal + 8, &n, sizeof(n));
    return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || rank < 0 || rank >= nranks || memcmp(id.internal, "FAKERCCL", 8) != 0) return 4;  // ncclInvalidArgument
    std::lock_guard<std::mutex> lk(g_mu);
    std::shared_ptr<Group>& g = g_groups[std::string(id.internal, sizeof(id.internal))];
    if (!g) {
        g = std::make_shared<Group>();
        g->world = nranks;
        g->send.assign((size_t)nranks, nullptr);
        g->recv.assign((size_t)nranks, nullptr);
    }
    if (g->world != nranks) return 4;
    *comm = new Comm{g, rank};
    return 0;
}
int ncclCommDestroy(void* comm) {
    delete (Comm*)comm;
    return 0;
}
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void*) {
    if (dtype != 7 || op != 0) return 4;  // ncclFloat32, ncclSum
    g_calls++;
    collective((Comm*)comm, send, recv, [&](Group& g) {
        std::vector<float> acc((const float*)g.send[0], (const float*)g.send[0] + count);
        for (int r = 1; r < g.world; ++r)
            for (size_t i = 0; i < count; ++i) acc[i] = acc[i] + ((const float*)g.send[r])[i];
        for (int r = 0; r < g.world; ++r) memcpy(g.recv[r], acc.data(), count * sizeof(float));
    });
    return 0;
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void*) {
    const size_t bytes = count * (dtype == 7 ? 4u : 1u);
    if (dtype != 7 && dtype != 1) return 4;
    g_calls++;
    collective((Comm*)comm, send, recv, [&](Group& g) {
        std::vector<char> all((size_t)g.world * bytes);  // (staged: rank r's send buffer may BE its slot of its receive buffer)
        for (int r = 0; r < g.world; ++r) memcpy(all.data() + (size_t)r * bytes, g.send[r], bytes);
        for (int r = 0; r < g.world; ++r) memcpy(g.recv[r], all.data(), all.size());
    });
    return 0;
}
const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : "fake rccl: invalid argument"; }
unsigned long long fakerccl_calls(void) { return g_calls.load(); }
}
