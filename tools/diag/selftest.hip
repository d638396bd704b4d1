// sctc_selftest(): probes the hardware idioms the kernels rely on, on the device
// itself: DPP wave shift / DPP reduction against __shfl references, ds_bpermute
// gather, readlane broadcast, and the operand/accumulator lane maps of the two
// f32 MFMA shapes used by the BRNN kernels (asymmetric operands, so a transposed
// map cannot pass).  Returns a bitmask of failed probes (0 = all good).
#include "diag_common.h"
#include "xlane.h"

namespace sctc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void selftest_kernel(int* fail)
{
    const int lane = threadIdx.x;
    int bad = 0;
    // 1: wave shift right by one
    {
        float v = 3.0f * lane + 1.0f;
        float got = lane_shr1(v);
        float want = lane == 0 ? 0.f : 3.0f * (lane - 1) + 1.0f;
        if (got != want) bad |= 1;
        double vd = 1e-3 * lane + 7.0;
        double gd = lane_shr1(vd);
        double wd = lane == 0 ? 0.0 : 1e-3 * (lane - 1) + 7.0;
        if (gd != wd) bad |= 1;
    }
    // 2: wave sum (exact in integers-as-floats)
    {
        float v = (float)(lane * lane + 1);
        float got = wave_sum(v);
        float want = 0.f;
        for (int i = 0; i < 64; ++i) want += (float)(i * i + 1);
        if (got != want) bad |= 2;
        double vd = (double)(lane * 1000003 + 5);
        double gd = wave_sum(vd);
        double wd = 0.0;
        for (int i = 0; i < 64; ++i) wd += (double)(i * 1000003 + 5);
        if (gd != wd) bad |= 2;
    }
    // 4: gather / broadcast
    {
        float v = 10.0f + lane;
        int src = (lane * 7 + 3) & 63;
        if (lane_gather(v, src) != 10.0f + src) bad |= 4;
        if (lane_bcast(v, 37) != 47.0f) bad |= 4;
        double vd = 0.5 + lane;
        if (lane_gather(vd, src) != 0.5 + src) bad |= 4;
        if (lane_bcast(vd, 5) != 5.5) bad |= 4;
    }
    // 8: mfma_f32_16x16x4f32: A[i][k] lane (i = l&15, k = l>>4), B[k][j] lane (j = l&15, k = l>>4)
    //    D[row][col]: col = l&15, row = 4*(l>>4) + reg
    {
        const int i = lane & 15, k = lane >> 4;
        float a = (float)(i + 1) + 0.25f * k;        // A[i][k]
        float bv = (float)(2 * i + 1) - 0.5f * k;     // B[k][j=i]
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r, col = lane & 15;
            float want = 0.f;
            for (int kk = 0; kk < 4; ++kk)
                want += ((float)(row + 1) + 0.25f * kk) * ((float)(2 * col + 1) - 0.5f * kk);
            if (fabsf(acc[r] - want) > 1e-3f) bad |= 8;
        }
    }
    // 16: mfma_f32_32x32x2f32: A[i][k] lane (i = l&31, k = l>>5), B[k][j] lane (j = l&31, k = l>>5)
    //     D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
    {
        const int i = lane & 31, k = lane >> 5;
        float a = (float)(i + 1) + 0.25f * k;
        float bv = (float)(2 * i + 1) - 0.5f * k;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            float want = 0.f;
            for (int kk = 0; kk < 2; ++kk)
                want += ((float)(row + 1) + 0.25f * kk) * ((float)(2 * col + 1) - 0.5f * kk);
            if (fabsf(acc[r] - want) > 1e-3f) bad |= 16;
        }
    }
    if (bad) atomicOr(fail, bad);
}

}  // namespace sctc

extern "C" int sctc_selftest(void* stream)
{
    using namespace sctc;
    int* d = nullptr;
    SCTC_HIP_TRY(hipMalloc(&d, sizeof(int)));
    SCTC_HIP_TRY(hipMemsetAsync(d, 0, sizeof(int), (hipStream_t)stream));
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d);
    int h = -1;
    hipError_t e = hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(SCTC_ERR_HIP, "selftest: %s", hipGetErrorString(e));
    return h;
}
