// fp32 MFMA GEMM used for every time-batched contraction of the BRNN
// (brnnet.py:140 fwd, :196 wgrad, :204 dgrad, :227-230 recurrent wgrad;
// the reference calls cm.dot == cuBLAS sgemm, fp32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sctc {

// C[M][N] (row-major, ldc) = sum_k A(m,k) * B(k,n)   then the epilogue.
//   a_kcontig : A(m,k) = A[m*lda + k]        else A(m,k) = A[ka(k)*lda + m]
//   b_kcontig : B(k,n) = B[n*ldb + k]        else B(k,n) = B[kb(k)*ldb + n]
//   ka(k) = idx_a ? idx_a[k] : k  (row gather along K, used by the recurrent wgrad)
// Feature dimensions (the contiguous ones) must be multiples of 4 and 16-byte aligned.
struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int64_t lda, ldb, ldc;
    int32_t M, N, K;
    int32_t a_kcontig, b_kcontig;
    const int32_t* idx_a;   // nullable, only for !a_kcontig
    const int32_t* idx_b;   // nullable, only for !b_kcontig
    // epilogue: v = acc (+ bias[n]) ; relu ; * (mask[m][n] > 0) ; + add_scale*addend[m][n] ; (C += v)
    const float* bias;      // nullable [N]
    const float* mask;      // nullable [M][ldmask]
    int64_t ldmask;
    const float* addend;    // nullable [M][ldadd]
    int64_t ldadd;
    float add_scale;
    int32_t relu;
    int32_t accumulate;
    // Fused bias gradient (brnnet.py:200 deltas.sum(axis=1)): when A is row-contiguous
    // ([K][M], the weight-gradient layout) and `colsum_a` is set, colsum_a[m] (+)= sum_k A(m,k),
    // accumulated by the blocks of the first N tile from the A tiles they stage anyway, summed in
    // a fixed order (`accumulate` selects += like for C).
    float* colsum_a;        // nullable [M]
    // split-K (deterministic two-pass): partials in `splitk_ws` ([splits][M][N] floats, then
    // [splits][M] column-sum partials)
    float* splitk_ws;
    int32_t splits;
    // operand precision: 0 = fp32 (v_mfma_f32_32x32x2_f32, exact fp32 fma chain);
    // 1 = float16, 2 = bfloat16: both operands rounded to 16 bit on the way into LDS, fp32
    // accumulate (gemm_h16.hip; the "fp16 activations" configuration);
    // 3 = fp32 operands split exactly into three bfloat16 terms, six cross products on the
    // bfloat16 matrix cores, fp32 accumulate: fp32-accurate (gemm_h16.hip, SCTC_BF16X3)
    int32_t prec;
    // prec != 0 only.  in16: both operands are ALREADY 16-bit in memory (float16 for prec 1,
    // bfloat16 for prec 2; `A` / `B` then point at 2-byte elements, lda / ldb count them): the
    // shadow copies the producers below write.  K-contiguous 16-bit operands need K % 8 == 0.
    int32_t in16;
    // optional 16-bit shadow copies of the result, written by the epilogue next to C (same
    // row stride ldc16 for both): C16a = float16, C16b = bfloat16 -- the operands of the GEMMs that
    // consume C next, rounded once by their producer instead of by every consumer
    uint16_t* C16a;
    uint16_t* C16b;
    int64_t ldc16;
    // "fp16 activations" proper (round 4): skip_c32 != 0 -- the fp32 result is NOT stored, only the
    // 16-bit shadow(s) are (a hidden layer's activations / a delta whose only readers are 16-bit
    // GEMM operands); needs C16a or C16b and no accumulate.  mask16: the ReLU mask operand as a
    // 16-bit matrix (float16 or bfloat16 bits, row stride ldmask16): keep where the value is > 0,
    // i.e. sign bit clear and not zero -- rounding to 16 bit keeps sign and zero-ness (float16
    // flushes |x| < 2^-25 to zero).  Exclusive with `mask`.
    int32_t skip_c32;
    const uint16_t* mask16;
    int64_t ldmask16;
    // prec 0 only: A is the SUM of two matrices of the same layout and leading dimension, A(m,k) =
    // A[..] + A2[..], added while the tile is staged (brnnet.py:153 hActs = hActsFor + hActsBack is the A
    // operand of the next layer's forward GEMM, :233 deltasOut = deltasFor + deltasBack the A operand of
    // the next weight gradient; the fp32 sum is the one add_kernel computed: bit-identical).  `a_sum`
    // (nullable, same layout as A): the blocks of the first N tile also store the sum there, once per
    // element, for the sum's later readers.  Layouts NT and TN only, no row gather.
    const float* A2;
    float* a_sum;
};
// the 16-bit value v (float16 or bfloat16 bits) is > 0
__host__ __device__ inline bool gemm_pos16(unsigned v) { return (v & 0x8000u) == 0u && (v & 0x7fffu) != 0u; }

// XCD-aware, bijective block -> tile remap, correct for split-K grids (round 4).  Workgroups are dealt to
// the 8 XCDs round-robin by their LINEAR id (blockIdx.y * gridDim.x + blockIdx.x for a 2-D grid), and an
// XCD should walk CONSECUTIVE tiles of ONE K slice, so that the blocks it runs at a time share operand
// panels in its 4 MiB L2.  Rounds 1-3 took blockIdx.x % 8 for the XCD: right for gridDim.y == 1, but with
// split-K (the weight gradients: 285 tiles x 7 K slices at cfg-3) slice y starts at XCD (y * 285) % 8, the
// "XCD c" group of tiles was in fact spread over all eight L2s, and the launch fetched 3.5 GB from the
// memory side where the forward GEMM of the same size fetches 2.0 GB (PMC, profiles/r04_*).
// Within slice y: block x runs on XCD (x + sh) % 8, sh = (y * nblk) % 8; the tiles are handed out class by
// class (XCD 0's blocks first), position inside a class in dispatch order.
#ifdef __HIPCC__
__device__ __forceinline__ int gemm_xcd_tile(int nblk)
{
    const int sh = (int)(((long long)blockIdx.y * nblk) & 7);
    const int lin = (int)blockIdx.x + sh, c = lin & 7;
    auto count = [](int a, int cls) { return (a + 7 - cls) >> 3; };     // integers in [0, a) congruent to cls mod 8
    int before = 0;
    for (int k = 0; k < c; ++k) before += count(nblk + sh, k) - count(sh, k);
    return before + count(lin, c) - count(sh, c);
}
#endif

// picks the split-K factor; returns the workspace floats needed (0 when splits == 1)
int64_t gemm_plan_splits(int M, int N, int K, int* splits, int prec = 0, int in16 = 0);   // in16: operands 16-bit in memory, same layout
int launch_gemm_f32(GemmArgs a, hipStream_t stream);     // dispatches on a.prec
// gemm_h16.hip
int64_t gemm_h16_plan_splits(int M, int N, int K, int* splits, int prec, int in16);
int launch_gemm_h16_tiles(const GemmArgs& a, hipStream_t stream);
// gemm_g16.hip: 16-bit operands in memory, staged by LDS-DMA (256 x 256 tiles)
bool gemm_g16_applies(const GemmArgs& a);
bool gemm_g16_enabled();
int launch_gemm_g16(const GemmArgs& a, hipStream_t stream);
// prec 3 (gemm_s3.hip)
int64_t gemm_s3_plan_splits(int M, int N, int K, int* splits);
int launch_gemm_s3(const GemmArgs& a, hipStream_t stream);

}  // namespace sctc
