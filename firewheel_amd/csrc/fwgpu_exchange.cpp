// fwgpu_exchange.cpp — multi-GPU mix bus behind the C ABI (SURVEY §8e path 2; include/fwgpu.h "bus exchange").
//
// One process per GPU, voices sharded, nothing exchanged until the mix bus: the top-level R-port SumNode
// (nodes/sum.rs:111-133) is the only step where the shards meet.  This file gives a host WITHOUT torch / RCCL that step:
// every rank opens an exchange (one region of fine-grained HBM with R slots and R arrival words), publishes its IPC handle
// through whatever side channel the host has (a file, a pipe, MPI, the Rust shim's own rendezvous), maps the peers'
// handles (hipIpcOpenMemHandle: dmabuf over xGMI), and from then on a step is two kernels on the ctx stream:
// k_bus_push (store my partial bus + silence flags into my slot on every rank, release, raise my arrival word) and
// k_bus_reduce (wait for the R arrivals, add the R slots in rank order) — no host round trip, no ring, bit-identical to the
// single-process graph.  Ranks that live in one process (virtual shards on one device, tests) connect by pointer.
#include "fwgpu_ctx.h"

#include <unistd.h>

namespace {

constexpr uint64_t EX_MAGIC = 0x4657455843483031ull;  // "FWEXCH01"
constexpr size_t EX_ERR_OFF = 1024, EX_DATA_OFF = 4096;  // k_exchange.hip.h

struct ExHandle {  // what fwgpu_bus_exchange_export writes (<= FWGPU_EXCHANGE_HANDLE_BYTES)
    uint64_t magic;
    uint64_t pid;
    uint64_t ptr;    // the region's address in the exporting process
    uint64_t bytes;
    uint64_t max_floats, slot_bytes;
    uint32_t world, rank;
    int32_t device;
    uint32_t pad;
    hipIpcMemHandle_t ipc;
};
static_assert(sizeof(ExHandle) <= FWGPU_EXCHANGE_HANDLE_BYTES, "exchange handle size");

}  // namespace

struct fwgpu_bus_exchange {
    fwgpu_ctx* ctx = nullptr;
    ExchangeGeom geom{};
    ExchangePeers peers{};
    char* region = nullptr;  // this rank's region (fine-grained device memory)
    size_t bytes = 0;
    uint32_t max_sil = 0;
    bool connected[FW_MAX_BUS_PARTS] = {false};
    bool ipc_mapped[FW_MAX_BUS_PARTS] = {false};
    unsigned* d_counter = nullptr;      // [0] k_bus_push's workgroup counter; as u64 words: [1] k_bus_wait's verdict, [8 + p] longest wait for peer p
    unsigned long long seq = 0;        // steps pushed so far
    unsigned long long reduced = 0;    // steps reduced so far
    unsigned long long budget_ticks = 300000000ull;  // 3 s at 100 MHz
    bool have_ipc = false;
    bool failed = false;               // status() has seen the device's sticky error word: push / reduce refuse from then on
    hipIpcMemHandle_t ipc{};
};

extern "C" {

fwgpu_bus_exchange* fwgpu_bus_exchange_open(fwgpu_ctx* c, uint32_t rank, uint32_t world, uint64_t max_floats, uint32_t max_silence_bytes) {
    if (!c) return nullptr;
    if (world == 0 || world > FW_MAX_BUS_PARTS || rank >= world || max_floats == 0 || max_floats > (1ull << 32)) {
        fail(c, FWGPU_ERR_INVALID, "bus exchange: 1..64 ranks, rank < world, 1..2^32 floats per bus");
        return nullptr;
    }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) (void)hipSetDevice(c->device);
    RtHold hold(c);  // hipDeviceSynchronize / hipFree below wait for every stream: no resident realtime kernel meanwhile
    fwgpu_bus_exchange* ex = new (std::nothrow) fwgpu_bus_exchange();
    if (!ex) {
        fail(c, FWGPU_ERR_DEVICE, "bus exchange: out of host memory");
        return nullptr;
    }
    ex->ctx = c;
    ex->geom.world = (int)world;
    ex->geom.rank = (int)rank;
    ex->geom.max_floats = (max_floats + 3) & ~3ull;
    ex->max_sil = max_silence_bytes;
    ex->geom.slot_bytes = (ex->geom.max_floats * 4 + max_silence_bytes + 255) & ~255ull;
    ex->bytes = EX_DATA_OFF + 2 * (size_t)world * ex->geom.slot_bytes;
    void* p = nullptr;
    // fine-grained device memory: the type whose contract is coherence between agents under system-scope fences / atomics —
    // what a peer stored and released is what this device's acquiring load sees.  (Round 3 first used hipDeviceMallocUncached:
    // correct while the region lives, but once FREED such memory was handed back out by hipMalloc and the ordinary buffers that
    // landed on it read stale data — a flaky parity failure of an unrelated context later in the same process, found by the
    // lazy-adoption test mode.  Fine-grained regions, and never-freed uncached ones, do not do that: 8 / 8 clean runs each.)
    hipError_t e = hipExtMallocWithFlags(&p, ex->bytes, hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(p, 0, ex->bytes);
    void* ctr = nullptr;
    if (e == hipSuccess) e = hipMalloc(&ctr, 1024);
    if (e == hipSuccess) e = hipMemset(ctr, 0, 1024);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        hipfail(c, e, "bus exchange: allocating the slot region");
        if (p) (void)hipFree(p);
        if (ctr) (void)hipFree(ctr);
        delete ex;
        return nullptr;
    }
    ex->region = (char*)p;
    ex->d_counter = (unsigned*)ctr;
    ex->peers.base[rank] = ex->region;
    ex->connected[rank] = true;
    // the IPC handle is asked for once, here; a stack that cannot export (no dmabuf) still serves same-process peers
    ex->have_ipc = hipIpcGetMemHandle(&ex->ipc, ex->region) == hipSuccess;
    if (!ex->have_ipc) (void)hipGetLastError();
    return ex;
}

void fwgpu_bus_exchange_close(fwgpu_bus_exchange* ex) {
    if (!ex) return;
    RtHold hold(ex->ctx);
    if (ex->ctx) (void)hipStreamSynchronize(ex->ctx->stream);
    for (int p = 0; p < ex->geom.world; ++p)
        if (ex->ipc_mapped[p] && ex->peers.base[p]) (void)hipIpcCloseMemHandle(ex->peers.base[p]);
    if (ex->region) (void)hipFree(ex->region);
    if (ex->d_counter) (void)hipFree(ex->d_counter);
    delete ex;
}

int fwgpu_bus_exchange_export(fwgpu_bus_exchange* ex, void* handle) {
    if (!ex || !handle) return FWGPU_ERR_INVALID;
    ExHandle h;
    memset(&h, 0, sizeof(h));
    h.magic = EX_MAGIC;
    h.pid = (uint64_t)getpid();
    h.ptr = (uint64_t)(uintptr_t)ex->region;
    h.bytes = ex->bytes;
    h.max_floats = ex->geom.max_floats;
    h.slot_bytes = ex->geom.slot_bytes;
    h.world = (uint32_t)ex->geom.world;
    h.rank = (uint32_t)ex->geom.rank;
    h.device = ex->ctx->device;
    if (ex->have_ipc) h.ipc = ex->ipc;
    else h.pad = 1;  // "this region cannot leave its process"
    memset(handle, 0, FWGPU_EXCHANGE_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    return 0;
}

int fwgpu_bus_exchange_connect(fwgpu_bus_exchange* ex, uint32_t peer_rank, const void* handle) {
    if (!ex || !handle) return FWGPU_ERR_INVALID;
    fwgpu_ctx* c = ex->ctx;
    ExHandle h;
    memcpy(&h, handle, sizeof(h));
    if (h.magic != EX_MAGIC) return fail(c, FWGPU_ERR_INVALID, "bus exchange: not an exchange handle");
    if (peer_rank >= (uint32_t)ex->geom.world || h.rank != peer_rank) return fail(c, FWGPU_ERR_INVALID, "bus exchange: handle is not that rank's");
    if (h.world != (uint32_t)ex->geom.world || h.max_floats != ex->geom.max_floats || h.slot_bytes != ex->geom.slot_bytes || h.bytes != ex->bytes)
        return fail(c, FWGPU_ERR_INVALID, "bus exchange: the peer was opened with another world size / bus size");
    if ((int)peer_rank == ex->geom.rank) return 0;  // its own region is connected from the start
    if (ex->connected[peer_rank]) return fail(c, FWGPU_ERR_INVALID, "bus exchange: peer already connected");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) (void)hipSetDevice(c->device);
    if (h.pid == (uint64_t)getpid()) {  // a rank of this process (virtual shards, several devices under one host): by pointer
        if (h.device != c->device) {
            hipError_t e = hipDeviceEnablePeerAccess(h.device, 0);
            if (e != hipSuccess) (void)hipGetLastError();  // (already enabled is fine; a real refusal shows up as a fault-free error below)
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, c->device, h.device) != hipSuccess || !can)
                return fail(c, FWGPU_ERR_DEVICE, "bus exchange: no peer access between the two devices");
        }
        ex->peers.base[peer_rank] = (char*)(uintptr_t)h.ptr;
    } else {
        if (h.pad == 1) return fail(c, FWGPU_ERR_DEVICE, "bus exchange: the peer could not export an IPC handle (hipIpcGetMemHandle failed there)");
        void* p = nullptr;
        HIPC(c, hipIpcOpenMemHandle(&p, h.ipc, hipIpcMemLazyEnablePeerAccess));
        ex->peers.base[peer_rank] = (char*)p;
        ex->ipc_mapped[peer_rank] = true;
    }
    ex->connected[peer_rank] = true;
    return 0;
}

int fwgpu_bus_exchange_set_timeout_ms(fwgpu_bus_exchange* ex, uint32_t ms) {
    if (!ex) return FWGPU_ERR_INVALID;
    ex->budget_ticks = (unsigned long long)(ms ? ms : 1) * 100000ull;  // s_memrealtime: 100 MHz
    return 0;
}

int fwgpu_bus_exchange_push(fwgpu_bus_exchange* ex, const float* d_partial, const uint8_t* d_silence, uint64_t n_floats, uint32_t n_blocks,
                            uint32_t n_channels) {
    if (!ex) return FWGPU_ERR_INVALID;
    fwgpu_ctx* c = ex->ctx;
    AudioCallScope audio;
    if (!d_partial || n_floats == 0 || n_floats > ex->geom.max_floats || ((uintptr_t)d_partial & 15u))
        return fail(c, FWGPU_ERR_INVALID, "bus exchange: partial bus null, unaligned or longer than the slots");
    // (a rank that reports no flags still CLEARS its slot's flag bytes — "nothing is silent" — for the blocks of the step: a peer
    //  that reduces with flags reads them, and what an earlier step left there is not this step's: ADVICE r3)
    uint64_t n_sil = (uint64_t)n_blocks * n_channels;
    if (n_sil > ex->max_sil) {
        if (d_silence) return fail(c, FWGPU_ERR_INVALID, "bus exchange: more silence flags than the slots hold");
        n_sil = ex->max_sil;
    }
    if (ex->failed) return fail(c, FWGPU_ERR_DEVICE, "bus exchange: a peer did not arrive in an earlier step; the exchange is closed (reopen it on every rank)");
    for (int p = 0; p < ex->geom.world; ++p)
        if (!ex->connected[p]) return fail(c, FWGPU_ERR_INVALID, "bus exchange: not every peer is connected");
    if (ex->seq != ex->reduced) return fail(c, FWGPU_ERR_INVALID, "bus exchange: push without the previous step's reduce");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) (void)hipSetDevice(c->device);
    ex->seq++;
    LCHK(c, launch_bus_push(c->stream, ex->peers, ex->geom, d_partial, d_silence, (size_t)n_floats, (uint32_t)n_sil, ex->seq, ex->d_counter));
    return 0;
}

int fwgpu_bus_exchange_reduce(fwgpu_bus_exchange* ex, float* d_out, uint8_t* d_out_silence, uint64_t n_floats, uint32_t n_blocks,
                              uint32_t frames_per_block, uint32_t n_channels, int have_silence) {
    if (!ex) return FWGPU_ERR_INVALID;
    fwgpu_ctx* c = ex->ctx;
    AudioCallScope audio;
    if (!d_out || n_floats == 0 || n_floats > ex->geom.max_floats || ((uintptr_t)d_out & 15u))
        return fail(c, FWGPU_ERR_INVALID, "bus exchange: output null, unaligned or longer than the slots");
    if (ex->failed) return fail(c, FWGPU_ERR_DEVICE, "bus exchange: a peer did not arrive in an earlier step; the exchange is closed (reopen it on every rank)");
    if (ex->reduced + 1 != ex->seq) return fail(c, FWGPU_ERR_INVALID, "bus exchange: reduce without a push");
    uint64_t n_sil = 0;
    if (have_silence) {
        n_sil = (uint64_t)n_blocks * n_channels;
        if (n_sil == 0 || n_sil > ex->max_sil || frames_per_block == 0 || (uint64_t)n_blocks * frames_per_block * n_channels < n_floats)
            return fail(c, FWGPU_ERR_INVALID, "bus exchange: the silence flags do not cover the bus (blocks x frames x channels)");
    }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) (void)hipSetDevice(c->device);
    ex->reduced++;
    LCHK(c, launch_bus_reduce(c->stream, ex->region, ex->geom, d_out, have_silence ? d_out_silence : nullptr, (size_t)n_floats, (uint32_t)n_sil,
                              frames_per_block, n_channels, ex->seq, ex->budget_ticks, (unsigned long long*)ex->d_counter));
    return 0;
}

int fwgpu_bus_exchange_step(fwgpu_bus_exchange* ex, const float* d_partial, const uint8_t* d_silence, float* d_out, uint8_t* d_out_silence,
                            uint64_t n_floats, uint32_t n_blocks, uint32_t frames_per_block, uint32_t n_channels) {
    int rc = fwgpu_bus_exchange_push(ex, d_partial, d_silence, n_floats, n_blocks, n_channels);
    if (rc) return rc;
    return fwgpu_bus_exchange_reduce(ex, d_out, d_out_silence, n_floats, n_blocks, frames_per_block, n_channels, d_silence ? 1 : 0);
}

int fwgpu_bus_exchange_status(fwgpu_bus_exchange* ex, uint64_t* steps, uint64_t* failed_step) {
    if (!ex) return FWGPU_ERR_INVALID;
    fwgpu_ctx* c = ex->ctx;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) (void)hipSetDevice(c->device);
    HIPC(c, hipStreamSynchronize(c->stream));
    unsigned long long err = 0;
    HIPC(c, hipMemcpy(&err, ex->region + EX_ERR_OFF, sizeof(err), hipMemcpyDeviceToHost));
    if (steps) *steps = ex->reduced;
    if (failed_step) *failed_step = err;
    if (err) {
        ex->failed = true;  // (the device side is out of the exchange already — k_bus_push; from here on the host says so too)
        return fail(c, FWGPU_ERR_DEVICE, "bus exchange: a peer did not arrive within the time budget (its bus was replaced by zeros)");
    }
    return 0;
}

int fwgpu_bus_exchange_wait_stats(fwgpu_bus_exchange* ex, uint64_t* max_wait_us, uint32_t cap, int reset) {
    if (!ex) return FWGPU_ERR_INVALID;
    fwgpu_ctx* c = ex->ctx;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) (void)hipSetDevice(c->device);
    HIPC(c, hipStreamSynchronize(c->stream));
    unsigned long long w[FW_MAX_BUS_PARTS];
    HIPC(c, hipMemcpy(w, (unsigned long long*)ex->d_counter + 8, sizeof(w), hipMemcpyDeviceToHost));
    for (uint32_t p = 0; p < cap && p < (uint32_t)ex->geom.world; ++p)
        if (max_wait_us) max_wait_us[p] = w[p] / 100ull;  // s_memrealtime ticks of 10 ns
    if (reset) HIPC(c, hipMemset((unsigned long long*)ex->d_counter + 8, 0, sizeof(w)));
    return ex->geom.world;
}

}  // extern "C"
