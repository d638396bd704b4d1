// CU-masked streams (hipExtStreamCreateWithCUMask) -- diagnostics for the question "can other work
// run on the compute units a persistent recurrent grid leaves idle?" (DESIGN.md 4.2).
//   sctc_diag_stream_cu_mask: a stream whose kernels only run on the CUs whose mask bit is set;
//   sctc_diag_where:          one workgroup per entry reports where it ran: XCC id, shader engine,
//                             shader array, CU id (HW_REG_XCC_ID / HW_REG_HW_ID);
//   sctc_diag_stream_destroy.
#include "diag_common.h"

namespace sctc {

__global__ __launch_bounds__(256) void where_kernel(int* out, unsigned long long hold_ticks)
{
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 0xf;   // XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((32 - 1) << 11));           // HW_ID
        // gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
        out[blockIdx.x * 4 + 0] = (int)xcc;
        out[blockIdx.x * 4 + 1] = (int)((hw >> 13) & 7);
        out[blockIdx.x * 4 + 2] = (int)((hw >> 12) & 1);
        out[blockIdx.x * 4 + 3] = (int)((hw >> 8) & 15);
    }
    // hold the CU for a while so that the blocks of one launch spread over every CU they may use
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
}

}  // namespace sctc

extern "C" int sctc_diag_stream_cu_mask(const uint32_t* mask_words, int32_t n_words, void** stream_out)
{
    using namespace sctc;
    SCTC_CHECK_ARG(mask_words && n_words >= 1 && n_words <= 32 && stream_out, "stream_cu_mask: bad argument");
    hipStream_t s = nullptr;
    SCTC_HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words));
    *stream_out = (void*)s;
    return SCTC_OK;
}

extern "C" int sctc_diag_stream_destroy(void* stream)
{
    using namespace sctc;
    SCTC_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return SCTC_OK;
}

extern "C" int sctc_diag_where(int32_t* out_host, int32_t n_wgs, int32_t hold_us, void* stream)
{
    using namespace sctc;
    SCTC_CHECK_ARG(out_host && n_wgs >= 1 && n_wgs <= 65536 && hold_us >= 0 && hold_us <= 100000,
                   "diag_where: bad argument");
    int* dev = nullptr;
    SCTC_HIP_TRY(hipMalloc(&dev, sizeof(int) * 4 * n_wgs));
    hipLaunchKernelGGL(where_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream, dev,
                       (unsigned long long)hold_us * 100ull);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpy(out_host, dev, sizeof(int) * 4 * n_wgs, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    SCTC_HIP_TRY(e);
    return SCTC_OK;
}
