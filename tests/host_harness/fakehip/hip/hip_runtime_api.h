/* TEST DOUBLE, CPU tier only (tests/host_harness): just enough of the HIP runtime API for the HOST half of libfwgpu
 * (fwgpu_abi.cpp, fwgpu_run.cpp, fwgpu_plan_*.cpp) to run without a device — "device" memory is calloc'd host memory, streams / events / graphs are inert
 * tokens.  The kernels are NOT here: every launch_* is a no-op stub (launch_stubs.cpp), so no audio is ever computed on
 * this path; it exists to test graph editing, planning, plan selection, message bookkeeping and the error conventions
 * of the C ABI with `pytest -m "not gpu"`.  Never linked into the product. */
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <malloc.h>
#include <unistd.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct fake_stream* hipStream_t;
typedef struct fake_event* hipEvent_t;
typedef struct fake_graph* hipGraph_t;
typedef struct fake_graph_exec* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2 };
enum hipStreamCaptureMode { hipStreamCaptureModeThreadLocal = 1 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
};

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake hip error"; }
static inline hipError_t hipGetLastError(void) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "host harness (no device)");
    strcpy(p->gcnArchName, "gfx950:host-harness");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30;
    return hipSuccess;
}
extern "C" unsigned long long fwh_alloc_calls;  /* defined in launch_stubs.cpp: device / pinned allocations so far */
extern "C" long long fwh_fail_alloc_in; /* > 0: the n-th device / pinned allocation from now fails (then disarms) */
static inline hipError_t hipMalloc(void** p, size_t n) {
    fwh_alloc_calls++;
    if (fwh_fail_alloc_in > 0 && --fwh_fail_alloc_in == 0) return hipErrorOutOfMemory;
    if (n > ((size_t)1 << 32)) return hipErrorOutOfMemory;  /* the harness never needs more; keeps a bad size from eating the host */
    *p = calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T>
static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
extern "C" void fwh_freed(const void* p);  // launch_stubs.cpp: books the stubs keep per device buffer end with the buffer
static inline hipError_t hipFree(void* p) { if (p) fwh_freed(p); free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
enum { hipHostRegisterDefault = 0, hipHostRegisterMapped = 2 };
static inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
extern "C" unsigned long long fwh_h2d_copies, fwh_h2d_max_bytes, fwh_h2d_bytes; /* launch_stubs.cpp: asynchronous host-to-device copies so far, the largest */
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) {
    if (k == hipMemcpyHostToDevice) {
        __atomic_fetch_add(&fwh_h2d_copies, 1ull, __ATOMIC_RELAXED); /* (control and audio threads both copy) */
        __atomic_fetch_add(&fwh_h2d_bytes, (unsigned long long)n, __ATOMIC_RELAXED);
        unsigned long long m = __atomic_load_n(&fwh_h2d_max_bytes, __ATOMIC_RELAXED);
        while (n > m && !__atomic_compare_exchange_n(&fwh_h2d_max_bytes, &m, (unsigned long long)n, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
        }
    }
    if (n) memmove(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset2DAsync(void* d, size_t pitch, int v, size_t w, size_t h, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memset((char*)d + r * pitch, v, w);
    return hipSuccess;
}
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }  /* every "kernel" has finished when its launch returns */
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = (hipGraph_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t, void*, void*, size_t) { *x = (hipGraphExec_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t x) { free(x); return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }

/* ---- peer / IPC memory (fwgpu_exchange.cpp).  A handle carries (pid, pointer, usable size): inside one process it opens to
 * the very same memory, from another process to a zeroed stand-in of the same size (the harness computes nothing). */
typedef struct { char reserved[64]; } hipIpcMemHandle_t;
enum { hipDeviceMallocFinegrained = 1, hipDeviceMallocUncached = 3, hipIpcMemLazyEnablePeerAccess = 1 };
static inline hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
struct fake_ipc { unsigned long long pid, ptr, bytes; };
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
    fake_ipc f = {(unsigned long long)getpid(), (unsigned long long)(uintptr_t)p, (unsigned long long)malloc_usable_size(p)};
    memset(h, 0, sizeof(*h));
    memcpy(h->reserved, &f, sizeof(f));
    return hipSuccess;
}
extern "C" void* fwh_foreign_maps[64]; /* launch_stubs.cpp: stand-ins handed out for other processes' handles */
static inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
    fake_ipc f;
    memcpy(&f, h.reserved, sizeof(f));
    if (f.pid == (unsigned long long)getpid()) { *p = (void*)(uintptr_t)f.ptr; return hipSuccess; }
    *p = calloc(f.bytes ? f.bytes : 1, 1);
    for (int i = 0; i < 64; ++i)
        if (!fwh_foreign_maps[i]) { fwh_foreign_maps[i] = *p; break; }
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipIpcCloseMemHandle(void* p) {
    for (int i = 0; i < 64; ++i)
        if (fwh_foreign_maps[i] == p) { fwh_foreign_maps[i] = NULL; free(p); break; }
    return hipSuccess;
}
