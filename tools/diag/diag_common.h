// Error plumbing of libsctc_diag.so (diagnostics only; the product library is libsctc_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "sctc_diag.h"

#define SCTC_OK 0
#define SCTC_ERR_ARG (-1)
#define SCTC_ERR_HIP (-2)

namespace sctc {

char* diag_err_buf();
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

}  // namespace sctc
