// Shared host-side helpers for libsctc_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sctc.h"

namespace sctc {

// thread-local message returned by sctc_last_error()
char* err_buf();
int set_error(int code, const char* fmt, ...);

#define SCTC_HIP_TRY(expr)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return ::sctc::set_error(SCTC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                                     hipGetErrorString(e__), __FILE__, __LINE__);            \
    } while (0)

#define SCTC_CHECK_ARG(cond, ...)                                                            \
    do {                                                                                     \
        if (!(cond)) return ::sctc::set_error(SCTC_ERR_ARG, __VA_ARGS__);                    \
    } while (0)

#define SCTC_TRY(expr)                                                                       \
    do {                                                                                     \
        int rc__ = (expr);                                                                   \
        if (rc__ != SCTC_OK) return rc__;                                                    \
    } while (0)

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Pinned host staging for small per-call descriptor uploads.  A hipMemcpyAsync from PAGEABLE memory beyond a few
// tens of KB makes the runtime pin and unpin the source around the copy while the stream (and the GPU behind it)
// waits: 2 ms for 52 KB of CTC descriptors in the middle of a cfg-5 step.  One buffer, one event: `acquire` waits
// until the previous upload from the buffer has executed (it has, long ago, in every steady state) before the
// caller overwrites it; `uploaded` marks the new one.  acquire() returns nullptr when pinned memory cannot be
// had: the caller then uploads from its pageable source as before.
struct PinnedStage {
    void* pin = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    int ev_dev = -1;            // device the event belongs to: an event cannot be recorded on another device's stream
    bool pending = false;
    PinnedStage() = default;
    PinnedStage(const PinnedStage&) = delete;
    PinnedStage& operator=(const PinnedStage&) = delete;
    ~PinnedStage()
    {
        if (ev) { if (pending) (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
        if (pin) (void)hipHostFree(pin);
    }
    // the staging buffer, free to be overwritten (waits for the previous upload); nullptr: stage from pageable memory
    void* acquire(size_t bytes)
    {
        if (pending) { (void)hipEventSynchronize(ev); pending = false; }
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (ev && ev_dev != dev) {      // a per-thread stage used with another GPU's stream: new event there
            (void)hipEventDestroy(ev);
            ev = nullptr;
        }
        if (!ev) {
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                ev = nullptr;
                return nullptr;
            }
            ev_dev = dev;
        }
        if (cap < bytes) {
            if (pin) (void)hipHostFree(pin);
            pin = nullptr;
            cap = 0;
            const size_t want = bytes + bytes / 2 + 4096;
            if (hipHostMalloc(&pin, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                pin = nullptr;
                return nullptr;
            }
            cap = want;
        }
        return pin;
    }
    // after the asynchronous copies out of the buffer have been queued on `s`.  If the event cannot be recorded (a
    // stream of a device other than the current one), the copies are waited for instead: the buffer is never handed
    // out again while a copy may still read it.
    hipError_t uploaded(hipStream_t s)
    {
        hipError_t e = hipEventRecord(ev, s);
        pending = e == hipSuccess;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipStreamSynchronize(s);
        }
        return e;
    }
};
// Between PinnedStage::acquire and ::uploaded an error return must not leave a queued copy reading a buffer that the
// next call overwrites: the guard waits for the stream unless uploaded() was reached.
struct PinnedUploadGuard {
    hipStream_t s;
    bool armed;
    PinnedUploadGuard(hipStream_t stream, bool on) : s(stream), armed(on) {}
    ~PinnedUploadGuard() { if (armed) (void)hipStreamSynchronize(s); }
    void done() { armed = false; }
};

// bump allocator over a caller-provided workspace
struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    bool overflow = false;
    void init(void* p, size_t n) { base = (char*)p; cap = n; used = 0; overflow = false; }
    template <typename T>
    T* take(size_t count)
    {
        size_t bytes = align256(count * sizeof(T));
        if (used + bytes > cap) { overflow = true; used += bytes; return nullptr; }
        T* r = (T*)(base + used);
        used += bytes;
        return r;
    }
};

}  // namespace sctc
