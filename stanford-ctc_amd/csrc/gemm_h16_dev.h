// Device-side pieces shared by the 16-bit-operand GEMM kernels (gemm_h16.hip) and the three-term
// bfloat16 split GEMM (gemm_s3.hip): vector types, compile-time loops, the MFMA wrappers, the XCD-aware
// block remap, the fused epilogue and the bias-gradient reduction.
#pragma once
#include <utility>

#include "common.h"
#include "gemm_f32.h"

namespace sctc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// compile-time loop: the accumulator tiles must be indexed by constants everywhere, or the whole
// acc[][] array is demoted to scratch memory (with 8 tiles per wave `#pragma unroll` alone left the
// epilogue's tile loop rolled, and every MFMA of the main loop then went through scratch)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int BIG> struct HTile;
template <> struct HTile<0> { static constexpr int BM = 128, BN = 128, NT = 256, WGN = 2, TM = 2, TN = 2; };
template <> struct HTile<1> { static constexpr int BM = 256, BN = 256, NT = 512, WGN = 4, TM = 4, TN = 2; };
#ifndef SCTC_H16_OCC
#define SCTC_H16_OCC 3
#endif
static constexpr int H_OCC = SCTC_H16_OCC;     // 128x128 tile, fp32 operands: blocks per CU
static constexpr int X_OCC = 2;                // 128x128 tile, 16-bit operands (K tile 64: 72 KiB of LDS)

template <bool BF> struct H16;
template <> struct H16<false> {
    using V8 = h16x8;
    using V4 = h16x4;
    static __device__ __forceinline__ V4 cvt(float a, float b, float c, float d)
    {
        V4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
        return v;
    }
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float tofloat(unsigned h)
    {
        return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
    }
};
template <> struct H16<true> {
    using V8 = b16x8;
    using V4 = b16x4;
    static __device__ __forceinline__ V4 cvt(float a, float b, float c, float d)
    {
        V4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
        return v;
    }
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float tofloat(unsigned h) { return __uint_as_float(h << 16); }
};

// XCD-aware, bijective block remap (block b runs on XCD b % 8): an XCD walks consecutive tiles
__device__ __forceinline__ int h16_swizzle(int nblk) { return gemm_xcd_tile(nblk); }

// One k-tile of MFMAs from the LDS image (LDH = row stride in halves, KB = k per tile)
template <bool BF, int TM, int TN, int KB, int LDH>
__device__ __forceinline__ void h16_compute(const unsigned short* a, const unsigned short* b,
                                            f32x16 (&acc)[TM][TN])
{
    using HT = H16<BF>;
    using V8 = typename HT::V8;
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks) {
        V8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const V8*>(a + i * 32 * LDH + 16 * ks);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const V8*>(b + j * 32 * LDH + 16 * ks);
        // operands SWAPPED (B fragment first): the accumulator tile is then the transpose of the
        // 32x32 output block -- lane l owns output ROW l & 31 and, per group of 4 accumulator
        // registers, 4 CONSECUTIVE COLUMNS -- so the epilogue stores 16 B (fp32) / 8 B (16-bit
        // shadows) per lane instead of 4 B / 2 B
        static_for<TM * TN>([&](auto IJ) {
            constexpr int i = decltype(IJ)::value / TN, j = decltype(IJ)::value % TN;
            acc[i][j] = HT::mfma(bf[j], af[i], acc[i][j]);
        });
    }
}

// Epilogue shared by both kernels.  With the swapped MFMA operands accumulator register r of lane l
// is D[row = l & 31][col = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)] of the 32x32 block: 4 consecutive
// columns per register quad.  bias / relu / mask / addend / accumulate like gemm_f32.hip, plus the
// optional 16-bit shadow copies of the result (C16a float16, C16b bfloat16); split-K partials go to the
// workspace raw.  Vector path (16-byte fp32, 8-byte 16-bit accesses) when every pointer / stride
// allows it, element-wise otherwise.
template <int TM, int TN>
__device__ __forceinline__ void h16_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0,
                                             int wm, int wn, int lane)
{
    const int M = p.M, N = p.N;
    const bool partial = p.splits > 1;
    float* out = partial ? p.splitk_ws + (int64_t)blockIdx.y * M * N : p.C;
    const int64_t ldo = partial ? N : p.ldc;
    const bool has_mask = !partial && p.mask, has_add = !partial && p.addend;
    const bool has_mask16 = !partial && p.mask16, skip32 = !partial && p.skip_c32;
    const bool has_acc = !partial && p.accumulate;
    const bool has_bias = !partial && p.bias;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool vec = (N % 4 == 0) && (ldo % 4 == 0) && al16(out) &&
                     (!has_mask || (p.ldmask % 4 == 0 && al16(p.mask))) &&
                     (!has_mask16 || (p.ldmask16 % 4 == 0 && ((uintptr_t)p.mask16 & 7) == 0)) &&
                     (!has_add || (p.ldadd % 4 == 0 && al16(p.addend))) &&
                     (!has_acc || (p.ldc % 4 == 0 && al16(p.C))) && (!has_bias || al16(p.bias)) &&
                     (partial || ((!p.C16a || (p.ldc16 % 4 == 0 && ((uintptr_t)p.C16a & 7) == 0)) &&
                                  (!p.C16b || (p.ldc16 % 4 == 0 && ((uintptr_t)p.C16b & 7) == 0))));
    static_for<TM * TN>([&](auto IJ) {
        constexpr int i = decltype(IJ)::value / TN, j = decltype(IJ)::value % TN;
        const int row = m0 + wm * (TM * 32) + i * 32 + (lane & 31);
        const int cbase = n0 + wn * (TN * 32) + j * 32 + 4 * (lane >> 5);
        const int64_t rc = min(row, M - 1);       // loads from clamped addresses; never stored
        static_for<4>([&](auto G) {
            constexpr int g = decltype(G)::value;
            const int col = cbase + 8 * g;
            float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            if (vec) {
                const int colc = min(col, N - 4);
                if (!partial) {
                    if (has_bias) {
                        const float4 b = *reinterpret_cast<const float4*>(p.bias + colc);
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (has_mask) {
                        const float4 m = *reinterpret_cast<const float4*>(p.mask + rc * p.ldmask + colc);
                        v[0] = m.x > 0.f ? v[0] : 0.f; v[1] = m.y > 0.f ? v[1] : 0.f;
                        v[2] = m.z > 0.f ? v[2] : 0.f; v[3] = m.w > 0.f ? v[3] : 0.f;
                    }
                    if (has_mask16) {
                        const u32x2 m = *reinterpret_cast<const u32x2*>(p.mask16 + rc * p.ldmask16 + colc);
                        v[0] = gemm_pos16(m[0] & 0xffffu) ? v[0] : 0.f; v[1] = gemm_pos16(m[0] >> 16) ? v[1] : 0.f;
                        v[2] = gemm_pos16(m[1] & 0xffffu) ? v[2] : 0.f; v[3] = gemm_pos16(m[1] >> 16) ? v[3] : 0.f;
                    }
                    if (has_add) {
                        const float4 a = *reinterpret_cast<const float4*>(p.addend + rc * p.ldadd + colc);
                        v[0] += p.add_scale * a.x; v[1] += p.add_scale * a.y;
                        v[2] += p.add_scale * a.z; v[3] += p.add_scale * a.w;
                    }
                    if (has_acc) {
                        const float4 c = *reinterpret_cast<const float4*>(p.C + rc * p.ldc + colc);
                        v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w;
                    }
                }
                if (row < M && col < N) {
                    if (!skip32) *reinterpret_cast<float4*>(out + (int64_t)row * ldo + col) = make_float4(v[0], v[1], v[2], v[3]);
                    if (!partial) {
                        if (p.C16a) {
                            const h16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                            *reinterpret_cast<h16x4*>(p.C16a + (int64_t)row * p.ldc16 + col) = h;
                        }
                        if (p.C16b) {
                            const b16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                            *reinterpret_cast<b16x4*>(p.C16b + (int64_t)row * p.ldc16 + col) = h;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = col + e;
                    if (row < M && c < N) {
                        float x = v[e];
                        if (!partial) {
                            if (has_bias) x += p.bias[c];
                            if (p.relu) x = fmaxf(x, 0.f);
                            if (has_mask) x = p.mask[(int64_t)row * p.ldmask + c] > 0.f ? x : 0.f;
                            if (has_mask16) x = gemm_pos16(p.mask16[(int64_t)row * p.ldmask16 + c]) ? x : 0.f;
                            if (has_add) x += p.add_scale * p.addend[(int64_t)row * p.ldadd + c];
                            if (has_acc) x += p.C[(int64_t)row * p.ldc + c];
                            if (p.C16a) p.C16a[(int64_t)row * p.ldc16 + c] = __builtin_bit_cast(unsigned short, (_Float16)x);
                            if (p.C16b) p.C16b[(int64_t)row * p.ldc16 + c] = __builtin_bit_cast(unsigned short, (__bf16)x);
                        }
                        if (!skip32) out[(int64_t)row * ldo + c] = x;
                    }
                }
            }
        });
    });
}

// column sums of a row-contiguous A operand (bias gradient): asum[e] holds this thread's partial
// for row 4*rq + e; the KQ lanes that share a row quad are adjacent (lane bits 0..log2(KQ)-1)
template <int KQ>
__device__ __forceinline__ void h16_colsum_out(const GemmArgs& p, float (&asum)[4], int m0, int rq, int kq)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = asum[e];
#pragma unroll
        for (int off = 1; off < KQ; off <<= 1) v += __shfl_xor(v, off, 64);
        asum[e] = v;
    }
    if (kq == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + 4 * rq + e;
            if (m < p.M) {
                if (p.splits > 1)
                    p.splitk_ws[(int64_t)p.splits * p.M * p.N + (int64_t)blockIdx.y * p.M + m] = asum[e];
                else
                    p.colsum_a[m] = p.accumulate ? p.colsum_a[m] + asum[e] : asum[e];
            }
        }
    }
}
}  // namespace sctc
