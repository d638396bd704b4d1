#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sctc {

int launch_gather_rows(float* dst, int64_t ldd, const float* src, int64_t lds, const int32_t* idx,
                       int64_t rows, int cols, hipStream_t s);
int launch_scatter_rows(float* dst, int64_t ldd, const float* src, int64_t lds,
                        const int32_t* idx, int64_t rows, int cols, hipStream_t s);
int launch_add(float* out, const float* a, const float* b, int64_t n, hipStream_t s);
size_t sumsq_ws_bytes();
int launch_sumsq(const float* x, int64_t n, double* out, double* ws, hipStream_t s);

}  // namespace sctc
