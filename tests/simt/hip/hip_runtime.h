// stand-in for <hip/hip_runtime.h>: liborbx's host AND device code compiled for the CPU SIMT emulator (tests/simt/simt.h).
// Streams and events are dummies -- every launch and copy completes before the call returns.  Test infrastructure only.
#pragma once
#include <memory>
#include <vector>
#include "../simt.h"
#include <functional>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotSupported = 801 };
typedef struct simt_stream *hipStream_t;
typedef struct simt_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocCoherent = 0x40000000, hipHostRegisterDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnregistered = 0 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void *devicePointer, *hostPointer; int isManaged; unsigned allocationFlags; };

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "simt: emulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
namespace simt { void sync_all(); }
inline hipError_t hipDeviceSynchronize() { simt::sync_all(); return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
// ---- streams and events.  Default: every operation completes before the call returns.  SIMT_STREAM_FUZZ=<seed>: operations are
// QUEUED per stream and executed, at synchronisation points, in a RANDOM order among all orders the recorded dependencies allow
// (in-order per stream, hipStreamWaitEvent edges) -- a result that changes with the seed is a missing dependency between streams.
namespace simt {
struct Rt;
Rt &rt();
bool fuzz();
void enqueue(hipStream_t s, std::function<void()> op);
void record(hipEvent_t e, hipStream_t s);
void wait_event(hipStream_t s, hipEvent_t e);
void sync_stream(hipStream_t s);
void sync_event(hipEvent_t e);
void sync_all();
hipStream_t new_stream();
hipStream_t null_stream();
hipEvent_t new_event();
}
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = simt::new_stream(); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = simt::new_stream(); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = simt::new_stream(); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { simt::sync_stream(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t s) { simt::sync_stream(s ? s : simt::null_stream()); return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { simt::wait_event(s, e); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = simt::new_event(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = simt::new_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { simt::sync_event(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { simt::record(e, s); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t e) { simt::sync_event(e); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
// Device memory is not zero on allocation: SIMT_MALLOC_FILL=<byte value> fills it with that byte, SIMT_MALLOC_FILL=r<seed> with different
// garbage per allocation -- a result that changes with it depends on device memory nothing wrote.
inline hipError_t hipMalloc(void **p, size_t n) {
    const size_t bytes = (n + 255) & ~(size_t)255;
    *p = aligned_alloc(256, bytes);
    if (!*p) return hipErrorInvalidValue;
    static const char *fill = getenv("SIMT_MALLOC_FILL");
    if (fill && fill[0] == 'r') {
        static unsigned long long st = strtoull(fill + 1, nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
        uint64_t *q = (uint64_t *)*p;
        for (size_t i = 0; i < bytes / 8; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; q[i] = st; }
    } else if (fill) memset(*p, (int)strtol(fill, nullptr, 0), bytes);
    return hipSuccess;
}
inline hipError_t hipFree(void *p) { simt::sync_all(); free(p); return hipSuccess; }   // hipFree synchronises the device
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }   // "device" memory is host memory here
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
inline hipError_t hipMemset(void *p, int v, size_t n) { simt::enqueue(simt::null_stream(), [=] { memset(p, v, n); }); return hipSuccess; }   // see null_stream()
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s) { simt::enqueue(s, [=] { memset(p, v, n); }); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { simt::sync_stream(simt::null_stream()); memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st) { simt::enqueue(st, [=] { memmove(d, s, n); }); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st) {
    simt::enqueue(st, [=] { for (size_t y = 0; y < h; y++) memmove((char *)d + y * dp, (const char *)s + y * sp, w); });
    return hipSuccess;
}
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {   // every pointer counts as pinned host memory
    memset(a, 0, sizeof(*a)); a->type = hipMemoryTypeHost; a->hostPointer = (void *)p; a->devicePointer = (void *)p; return hipSuccess;
}
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

// the virtual-memory API of the guard allocator (ORBX_GUARD): not emulated
typedef void *hipMemGenericAllocationHandle_t;
enum { hipMemAllocationTypePinned = 1, hipMemLocationTypeDevice = 1, hipMemAccessFlagsProtReadWrite = 3, hipMemAllocationGranularityMinimum = 0 };
struct hipMemLocation { int type, id; };
struct hipMemAllocationProp { int type; hipMemLocation location; int requestedHandleType; void *win32HandleMetaData; };
struct hipMemAccessDesc { hipMemLocation location; int flags; };
inline hipError_t hipMemGetAllocationGranularity(size_t *g, const hipMemAllocationProp *, int) { *g = 4096; return hipErrorNotSupported; }
inline hipError_t hipMemAddressReserve(void **, size_t, size_t, void *, unsigned long long) { return hipErrorNotSupported; }
inline hipError_t hipMemAddressFree(void *, size_t) { return hipErrorNotSupported; }
inline hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *, size_t, const hipMemAllocationProp *, unsigned long long) { return hipErrorNotSupported; }
inline hipError_t hipMemMap(void *, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long) { return hipErrorNotSupported; }
inline hipError_t hipMemUnmap(void *, size_t) { return hipErrorNotSupported; }
inline hipError_t hipMemRelease(hipMemGenericAllocationHandle_t) { return hipErrorNotSupported; }
inline hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipErrorNotSupported; }

// kernel launch: the blocks of the grid run one after the other, each under the fiber scheduler
namespace simt {
void launch(const char *name, dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body);
std::vector<uint32_t> block_order(dim3 grid, const char *name = nullptr);
int kernel_split();
bool fuzz();
void launch_blocks(const char *name, dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body, const uint32_t *ids, size_t n,
                   bool announce);
}
#include <tuple>
namespace simt {
template <typename F, typename... A>
inline void launch_on(hipStream_t st, const char *name, dim3 grid, dim3 block, size_t lds, F kernel, A... args) {
    auto tup = std::make_tuple(args...);   // kernel arguments are evaluated and copied at the launch call, as on the real runtime
    const int split = fuzz() && st ? kernel_split() : 0;
    if (split > 1) {   // SIMT_KERNEL_SPLIT: the launch as several queue entries (launch.cc)
        auto ids = std::make_shared<std::vector<uint32_t>>(block_order(grid, name));
        const size_t nb = ids->size(), pieces = std::min<size_t>(nb, (size_t)split), per = (nb + pieces - 1) / std::max<size_t>(pieces, 1);
        for (size_t c = 0; c * per < nb; c++)
            enqueue(st, [=] { launch_blocks(name, grid, block, lds, [&] { std::apply(kernel, tup); }, ids->data() + c * per, std::min(per, nb - c * per), c == 0); });
        return;
    }
    enqueue(st, [=] { launch(name, grid, block, lds, [&] { std::apply(kernel, tup); }); });
}
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    simt::launch_on(stream, #kernel, dim3(grid), dim3(block), (size_t)(lds), kernel, __VA_ARGS__)
