#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sctc {

int launch_gather_rows(float* dst, int64_t ldd, const float* src, int64_t lds, const int32_t* idx,
                       int64_t rows, int cols, hipStream_t s);
int launch_scatter_rows(float* dst, int64_t ldd, const float* src, int64_t lds,
                        const int32_t* idx, int64_t rows, int cols, hipStream_t s);
int launch_add(float* out, const float* a, const float* b, int64_t n, hipStream_t s);
// 16-bit shadow copies (operand_dtype = SCTC_F16): float16 and / or bfloat16, nullable outputs
int launch_cvt16(const float* src, uint16_t* dst_f16, uint16_t* dst_bf16, int64_t n, hipStream_t s);
int launch_add16(float* out, const float* a, const float* b, uint16_t* o_f16, uint16_t* o_bf16, int64_t n,
                 hipStream_t s, uint16_t* a_bf16 = nullptr, uint16_t* b_bf16 = nullptr);
int launch_transpose_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int rows, int cols,
                          hipStream_t s);
int launch_gather_rows16(float* dst, uint16_t* d_f16, uint16_t* d_bf16, int64_t ldd, const float* src,
                         int64_t lds, const int32_t* idx, int64_t rows, int cols, hipStream_t s);
size_t sumsq_ws_bytes();
int launch_sumsq(const float* x, int64_t n, double* out, double* ws, hipStream_t s);

}  // namespace sctc
