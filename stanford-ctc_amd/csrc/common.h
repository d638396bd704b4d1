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
