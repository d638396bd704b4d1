// HBM-bound helpers around the GEMM / recurrent / CTC kernels: row gather/scatter
// between the caller's per-utterance layout and the packed time-major minibatch,
// elementwise sums, bias gradients, and the flat-buffer optimizer primitives that
// replace the cudamat calls of sgd.py / NNet.updateParams.
#include "common.h"
#include "elementwise.h"

namespace sctc {

// dst[r][0..cols) = src[idx[r]][0..cols), zero for cols..ldd  (one wave per row)
__global__ __launch_bounds__(256) void gather_rows_kernel(float* __restrict__ dst, int64_t ldd,
                                                          const float* __restrict__ src,
                                                          int64_t lds, const int32_t* idx,
                                                          int64_t rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* s = src + (int64_t)idx[r] * lds;
    float* d = dst + r * ldd;
    for (int c = lane; c < ldd; c += 64) d[c] = c < cols ? s[c] : 0.f;
}

// dst[idx[r]][0..cols) = src[r][0..cols)
__global__ __launch_bounds__(256) void scatter_rows_kernel(float* __restrict__ dst, int64_t ldd,
                                                           const float* __restrict__ src,
                                                           int64_t lds, const int32_t* idx,
                                                           int64_t rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* s = src + r * lds;
    float* d = dst + (int64_t)idx[r] * ldd;
    for (int c = lane; c < cols; c += 64) d[c] = s[c];
}

__global__ __launch_bounds__(256) void add_kernel(float4* __restrict__ out,
                                                  const float4* __restrict__ a,
                                                  const float4* __restrict__ b, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

// ---- 16-bit shadow copies for the mixed-precision GEMMs (operand_dtype = SCTC_F16): every
// producer of an activation / delta matrix also writes it rounded to float16 (forward operands)
// and / or bfloat16 (backward operands), so that the consuming GEMMs load half the bytes and
// convert nothing.  Round-to-nearest-even, the same rounding the GEMM applies on the fly.
__device__ __forceinline__ unsigned short to_f16(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
__device__ __forceinline__ unsigned short to_bf16(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

// dst16a (float16, nullable) / dst16b (bfloat16, nullable) <- src, n4 float4 groups, same layout
__global__ __launch_bounds__(256) void cvt16_kernel(const float4* __restrict__ src, ushort4* __restrict__ dst16a,
                                                    ushort4* __restrict__ dst16b, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = src[i];
        if (dst16a) dst16a[i] = make_ushort4(to_f16(v.x), to_f16(v.y), to_f16(v.z), to_f16(v.w));
        if (dst16b) dst16b[i] = make_ushort4(to_bf16(v.x), to_bf16(v.y), to_bf16(v.z), to_bf16(v.w));
    }
}

// out = a + b (fp32; nullable: the fp16 configuration reads the sum through its shadows only) plus its
// 16-bit shadows, plus (a16b / b16b, nullable) the bfloat16 copies of the two inputs themselves -- the
// operands of the recurrent weight gradient, which used to be two more passes over the same matrices
__global__ __launch_bounds__(256) void add16_kernel(float4* __restrict__ out, const float4* __restrict__ a,
                                                    const float4* __restrict__ b, ushort4* __restrict__ o16a,
                                                    ushort4* __restrict__ o16b, ushort4* __restrict__ a16b,
                                                    ushort4* __restrict__ b16b, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        const float4 v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        if (out) out[i] = v;
        if (o16a) o16a[i] = make_ushort4(to_f16(v.x), to_f16(v.y), to_f16(v.z), to_f16(v.w));
        if (o16b) o16b[i] = make_ushort4(to_bf16(v.x), to_bf16(v.y), to_bf16(v.z), to_bf16(v.w));
        if (a16b) a16b[i] = make_ushort4(to_bf16(x.x), to_bf16(x.y), to_bf16(x.z), to_bf16(x.w));
        if (b16b) b16b[i] = make_ushort4(to_bf16(y.x), to_bf16(y.y), to_bf16(y.z), to_bf16(y.w));
    }
}

// bfloat16 transpose of a row-major fp32 matrix: dst[c][r] = bf16(src[r][c]), rows x cols (padded
// sizes, multiples of 32), through a 32 x 33 LDS tile (W^T as the K-contiguous B operand of the
// delta-propagation GEMM)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const float* __restrict__ src, int64_t lds_,
                                                             unsigned short* __restrict__ dst, int64_t ldd,
                                                             int rows, int cols)
{
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < cols && r < rows) dst[(int64_t)c * ldd + r] = to_bf16(tile[tx][k]);
    }
}

// gather with shadows: dst[r][0..ldd) = src[idx[r]][0..cols) (zero padded), fp32 + float16 + bfloat16
__global__ __launch_bounds__(256) void gather_rows16_kernel(float* __restrict__ dst, unsigned short* __restrict__ d16a,
                                                            unsigned short* __restrict__ d16b, int64_t ldd,
                                                            const float* __restrict__ src, int64_t lds_,
                                                            const int32_t* idx, int64_t rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* s = src + (int64_t)idx[r] * lds_;
    for (int c = lane; c < ldd; c += 64) {
        const float v = c < cols ? s[c] : 0.f;
        dst[r * ldd + c] = v;
        if (d16a) d16a[r * ldd + c] = to_f16(v);      // either shadow may be absent (forward-only
        if (d16b) d16b[r * ldd + c] = to_bf16(v);     // models have no bfloat16 shadows)
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y,
                                                   const float* __restrict__ x, float alpha,
                                                   int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] += alpha * x[i];
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, float alpha, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        x[i] *= alpha;
}

// sum of squares, float64 accumulation, two deterministic stages
static constexpr int SS_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, int64_t n,
                                                            double* __restrict__ partial)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = x[i];
        s += v * v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ partial,
                                                          int nblocks, double* __restrict__ out)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

// sgd.py:129-141,161 fused over the flat buffers (one pass instead of ~13)
__global__ __launch_bounds__(256) void nesterov_kernel(float* __restrict__ w, float* __restrict__ v,
                                                       const float* __restrict__ g, int64_t n,
                                                       float mom, float alpha, float max_gnorm,
                                                       float grad_scale,
                                                       const double* __restrict__ sumsq)
{
    float alph = alpha;
    if (sumsq) {
        const double gnorm = sqrt(*sumsq) * (double)grad_scale;
        if (gnorm > (double)max_gnorm) alph = (float)((double)alpha * ((double)max_gnorm / gnorm));
    }
    const float ga = alph * grad_scale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float nv = mom * v[i] - ga * g[i];
        v[i] = nv;
        w[i] += nv;
    }
}

// minibatch / data-parallel form: effective gradient e = grad_scale*g + reg*(w + mom*v) -- the L2
// term once, evaluated where the reference evaluates it: at the Nesterov look-ahead point
// (sgd.py:91-93 moves the weights to w + mom*v before costAndGrad, brnnet.py:197-198 adds reg*W
// there; `w` here is the weight after the look-ahead has been undone, `v` the velocity before this
// step's update).  The bias ranges of the flat buffer carry no L2 term.
static constexpr int MAX_NOREG_RANGES = 128;
struct NoRegRanges {
    int n;                                  // ranges; edge[2k], edge[2k+1] = [beg, end) ascending, disjoint
    int64_t edge[2 * MAX_NOREG_RANGES];
};
__device__ __forceinline__ float reg_at(const NoRegRanges& r, int64_t i, float reg)
{
    // upper bound over the sorted edges: an odd count of edges <= i means i lies inside a range
    int lo = 0, hi = 2 * r.n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (r.edge[mid] <= i) lo = mid + 1; else hi = mid;
    }
    return (lo & 1) ? 0.f : reg;
}

__global__ __launch_bounds__(256) void sumsq_reg_partial_kernel(const float* __restrict__ g,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ vel, float mom,
                                                                float grad_scale, float reg,
                                                                NoRegRanges nr, int64_t n,
                                                                double* __restrict__ partial)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float wl = vel ? w[i] + mom * vel[i] : w[i];
        const double v = (double)(grad_scale * g[i] + reg_at(nr, i, reg) * wl);
        s += v * v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void nesterov_reg_kernel(float* __restrict__ w, float* __restrict__ v,
                                                           const float* __restrict__ g, int64_t n,
                                                           float mom, float alpha, float max_gnorm,
                                                           float grad_scale, float reg,
                                                           NoRegRanges nr,
                                                           const double* __restrict__ sumsq)
{
    float alph = alpha;
    if (sumsq) {
        const double gnorm = sqrt(*sumsq);
        if (gnorm > (double)max_gnorm) alph = (float)((double)alpha * ((double)max_gnorm / gnorm));
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float vo = v[i];
        const float e = grad_scale * g[i] + reg_at(nr, i, reg) * (w[i] + mom * vo);
        const float nv = mom * vo - alph * e;
        v[i] = nv;
        w[i] += nv;
    }
}

static inline int grid_for(int64_t n)
{
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int launch_gather_rows(float* dst, int64_t ldd, const float* src, int64_t lds, const int32_t* idx,
                       int64_t rows, int cols, hipStream_t s)
{
    if (rows <= 0) return SCTC_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dst,
                       ldd, src, lds, idx, rows, cols);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int launch_scatter_rows(float* dst, int64_t ldd, const float* src, int64_t lds,
                        const int32_t* idx, int64_t rows, int cols, hipStream_t s)
{
    if (rows <= 0) return SCTC_OK;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dst,
                       ldd, src, lds, idx, rows, cols);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int launch_add(float* out, const float* a, const float* b, int64_t n, hipStream_t s)
{
    if (n <= 0) return SCTC_OK;
    SCTC_CHECK_ARG(n % 4 == 0, "add: element count must be a multiple of 4");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, (float4*)out,
                       (const float4*)a, (const float4*)b, n / 4);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int launch_cvt16(const float* src, uint16_t* dst_f16, uint16_t* dst_bf16, int64_t n, hipStream_t s)
{
    if (n <= 0 || (!dst_f16 && !dst_bf16)) return SCTC_OK;
    SCTC_CHECK_ARG(n % 4 == 0, "cvt16: element count must be a multiple of 4");
    hipLaunchKernelGGL(cvt16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, (const float4*)src,
                       (ushort4*)dst_f16, (ushort4*)dst_bf16, n / 4);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int launch_add16(float* out, const float* a, const float* b, uint16_t* o_f16, uint16_t* o_bf16, int64_t n,
                 hipStream_t s, uint16_t* a_bf16, uint16_t* b_bf16)
{
    if (n <= 0) return SCTC_OK;
    SCTC_CHECK_ARG(n % 4 == 0, "add: element count must be a multiple of 4");
    hipLaunchKernelGGL(add16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, (float4*)out, (const float4*)a,
                       (const float4*)b, (ushort4*)o_f16, (ushort4*)o_bf16, (ushort4*)a_bf16, (ushort4*)b_bf16, n / 4);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int launch_transpose_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int rows, int cols,
                          hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return SCTC_OK;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, src,
                       ld_src, dst, ld_dst, rows, cols);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int launch_gather_rows16(float* dst, uint16_t* d_f16, uint16_t* d_bf16, int64_t ldd, const float* src,
                         int64_t lds, const int32_t* idx, int64_t rows, int cols, hipStream_t s)
{
    if (rows <= 0) return SCTC_OK;
    hipLaunchKernelGGL(gather_rows16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dst, d_f16,
                       d_bf16, ldd, src, lds, idx, rows, cols);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

size_t sumsq_ws_bytes() { return sizeof(double) * SS_BLOCKS; }

int launch_sumsq(const float* x, int64_t n, double* out, double* ws, hipStream_t s)
{
    const int blocks = (int)std::min<int64_t>(SS_BLOCKS, std::max<int64_t>(1, (n + 255) / 256));
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, s, x, n, ws);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, ws, blocks, out);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

}  // namespace sctc

using namespace sctc;

extern "C" {

int sctc_axpy(float* y_dev, const float* x_dev, float alpha, int64_t n, void* stream)
{
    SCTC_CHECK_ARG(y_dev && x_dev && n >= 0, "axpy: bad argument");
    if (n == 0) return SCTC_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y_dev,
                       x_dev, alpha, n);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int sctc_scale(float* x_dev, float alpha, int64_t n, void* stream)
{
    SCTC_CHECK_ARG(x_dev && n >= 0, "scale: bad argument");
    if (n == 0) return SCTC_OK;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x_dev,
                       alpha, n);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int sctc_sumsq(const float* x_dev, int64_t n, double* out_dev, void* workspace_dev,
               size_t workspace_bytes, void* stream)
{
    SCTC_CHECK_ARG(x_dev && out_dev && workspace_dev && n >= 0, "sumsq: bad argument");
    if (workspace_bytes < sumsq_ws_bytes())
        return set_error(SCTC_ERR_WORKSPACE, "sumsq: workspace %zu < %zu bytes", workspace_bytes,
                         sumsq_ws_bytes());
    return launch_sumsq(x_dev, n, out_dev, (double*)workspace_dev, (hipStream_t)stream);
}

int sctc_nesterov_step(float* w_dev, float* v_dev, const float* g_dev, int64_t n, float mom,
                       float alpha, float max_gnorm, float grad_scale, const double* sumsq_dev,
                       void* stream)
{
    SCTC_CHECK_ARG(w_dev && v_dev && g_dev && n >= 0, "nesterov_step: bad argument");
    if (n == 0) return SCTC_OK;
    hipLaunchKernelGGL(nesterov_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                       w_dev, v_dev, g_dev, n, mom, alpha, max_gnorm, grad_scale, sumsq_dev);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

static int make_ranges(const int64_t* host, int32_t n, NoRegRanges* out)
{
    SCTC_CHECK_ARG(n >= 0 && n <= MAX_NOREG_RANGES && (n == 0 || host),
                   "noreg ranges: 0..%d [beg,end) pairs (got %d)", MAX_NOREG_RANGES, n);
    out->n = n;
    for (int k = 0; k < 2 * n; ++k) {
        out->edge[k] = host[k];
        SCTC_CHECK_ARG(k == 0 || host[k] >= host[k - 1], "noreg ranges must be ascending and disjoint");
    }
    return SCTC_OK;
}

int sctc_sumsq_reg(const float* g_dev, const float* w_dev, const float* v_dev, float mom,
                   float grad_scale, float reg, int64_t n,
                   const int64_t* noreg_ranges_host, int32_t n_ranges, double* out_dev,
                   void* workspace_dev, size_t workspace_bytes, void* stream)
{
    SCTC_CHECK_ARG(g_dev && w_dev && out_dev && workspace_dev && n >= 0, "sumsq_reg: bad argument");
    NoRegRanges nr;
    SCTC_TRY(make_ranges(noreg_ranges_host, n_ranges, &nr));
    if (workspace_bytes < sumsq_ws_bytes())
        return set_error(SCTC_ERR_WORKSPACE, "sumsq_reg: workspace %zu < %zu bytes", workspace_bytes,
                         sumsq_ws_bytes());
    const int blocks = (int)std::min<int64_t>(SS_BLOCKS, std::max<int64_t>(1, (n + 255) / 256));
    hipLaunchKernelGGL(sumsq_reg_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       g_dev, w_dev, v_dev, mom, grad_scale, reg, nr, n, (double*)workspace_dev);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       (const double*)workspace_dev, blocks, out_dev);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int sctc_nesterov_step_reg(float* w_dev, float* v_dev, const float* g_dev, int64_t n, float mom,
                           float alpha, float max_gnorm, float grad_scale, float reg,
                           const int64_t* noreg_ranges_host, int32_t n_ranges,
                           const double* sumsq_dev, void* stream)
{
    SCTC_CHECK_ARG(w_dev && v_dev && g_dev && n >= 0, "nesterov_step_reg: bad argument");
    if (n == 0) return SCTC_OK;
    NoRegRanges nr;
    SCTC_TRY(make_ranges(noreg_ranges_host, n_ranges, &nr));
    hipLaunchKernelGGL(nesterov_reg_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                       w_dev, v_dev, g_dev, n, mom, alpha, max_gnorm, grad_scale, reg, nr, sumsq_dev);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

}  // extern "C"
