// Mixed-precision GEMM for the "fp16 activations" configuration (BASELINE configs[4], cfg-5):
// the same contractions as gemm_f32.hip (brnnet.py:140 fwd, :196 wgrad, :204 dgrad,
// :227-230 recurrent wgrad) with BOTH operands rounded to a 16-bit type and fp32 accumulation on
// the matrix cores:
//   prec 1: float16  (v_mfma_f32_32x32x16_f16)   -- forward pass (activations in [0, 20])
//   prec 2: bfloat16 (v_mfma_f32_32x32x16_bf16)  -- backward pass (deltas need fp32's exponent range)
// Master weights, activations, deltas and gradients stay fp32 in HBM; the rounding is
// round-to-nearest-even (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32), so the numerics are exactly
// "operands rounded to 16 bit, exact products, fp32 sums".
//
// Two kernels:
//   gemm_h16_kernel  fp32 operands in memory, rounded in registers on their way into LDS (the public
//                    sctc_gemm_h16 entry).  K tile 32.
//   gemm_x16_kernel  operands ALREADY 16-bit in memory: the shadow copies the engine's producers write
//                    (GEMM epilogues, gather, adds, converts).  K tile 64: one K-contiguous row of a
//                    tile is then a full 128-byte line.  Ablation on the MI355X (8192^3, 256x256 tile):
//                    MFMAs + fragment reads alone 0.82 ms, + LDS stores and barrier 1.06 ms, + global
//                    loads 2.10 ms, and MFMAs removed 2.02 ms -- the loop is bound by the number of
//                    cache-line requests of the operand loads (not by their bytes: halving the bytes at
//                    the same line count changed nothing), so the 16-bit kernel halves the lines per flop.
// Tile shapes (both kernels): 128x128, 256 threads = 2x2 waves of 64x64 (2x2 MFMA tiles of
// 32x32x16); 256x256, 512 threads = 2x4 waves of 128x64 (4x2 MFMA tiles), one block per CU, chosen
// for large outputs.  LDS image of both operands: [row][k] halves with a row stride of k + 8
// halves: a fragment (lane l: row l&31, 8 consecutive k at 8*(l>>5)) is ONE conflict-free ds_read_b128.
//   K-contiguous operand  : 16-byte pieces of a row go straight to LDS (fp32: converted first);
//   row-contiguous operand: every thread loads a 4(k) x 4(row) micro-tile, transposes it in registers
//     (fp32: while converting; 16-bit: v_perm_b32) and writes 4 x ds_write_b64.
#include <algorithm>
#include <mutex>
#include <set>
#include <utility>

#include "common.h"
#include "gemm_f32.h"
#include "gemm_h16_dev.h"

namespace sctc {


// ------------------------------------------------------------------ fp32 operands in memory

template <bool AK, bool BKC, bool BF, int BIG>
__global__ __launch_bounds__(HTile<BIG>::NT, BIG ? 2 : H_OCC) void gemm_h16_kernel(GemmArgs p)
{
    constexpr int HBM = HTile<BIG>::BM, HBN = HTile<BIG>::BN, NT = HTile<BIG>::NT;
    constexpr int WGN = HTile<BIG>::WGN, TM = HTile<BIG>::TM, TN = HTile<BIG>::TN;
    constexpr int HBK = 32, HLD = HBK + 8;
    using HT = H16<BF>;
    using V4 = typename HT::V4;
    extern __shared__ __attribute__((aligned(16))) unsigned short hsmem[];
    constexpr int OP = HBM * HLD;                 // halves per operand per buffer (HBM == HBN)
    unsigned short* As = hsmem;                   // [2][HBM][HLD]
    unsigned short* Bs = hsmem + 2 * OP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int M = p.M, N = p.N, K = p.K;
    const int mt = (M + HBM - 1) / HBM, nt = (N + HBN - 1) / HBN;
    const int swz = h16_swizzle(mt * nt);
    const int tile_n = swz % nt, tile_m = swz / nt;
    const int m0 = tile_m * HBM, n0 = tile_n * HBN;
    const int ktiles = (K + HBK - 1) / HBK;
    const int per = (ktiles + p.splits - 1) / p.splits;
    const int kt_beg = blockIdx.y * per;
    const int kt_end = min(ktiles, kt_beg + per);

    f32x16 acc[TM][TN];
    static_for<TM * TN>([&](auto IJ) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[decltype(IJ)::value / TN][decltype(IJ)::value % TN][r] = 0.f;
    });

    const bool do_colsum = !AK && p.colsum_a != nullptr && tile_n == 0;
    float asum[4] = {0.f, 0.f, 0.f, 0.f};    // row-contiguous A: sums over k of this thread's 4 rows
    const int kq = tid & 7;                  // K-contiguous: item q -> row (tid + NT q) / 8, k quad kq
                                             // row-contiguous: k quad kq, row quad tid >> 3
    if (kt_beg < kt_end) {
        float4 ra[4], rb[4];
        const int Kc4 = (K - 1) & ~3, Kc1 = K - 1;
        const float* pa[4];
        const float* pb[4];
        int ia[4], ib[4];
        if constexpr (AK) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pa[q] = p.A + (int64_t)min(m0 + ((tid + NT * q) >> 3), M - 1) * p.lda;
        } else {
            pa[0] = p.A + min(m0 + 4 * (tid >> 3), (M - 1) & ~3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = min(kt_beg * HBK + 4 * kq + j, Kc1);
                ia[j] = p.idx_a ? p.idx_a[k] : k;
            }
        }
        if constexpr (BKC) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pb[q] = p.B + (int64_t)min(n0 + ((tid + NT * q) >> 3), N - 1) * p.ldb;
        } else {
            pb[0] = p.B + min(n0 + 4 * (tid >> 3), (N - 1) & ~3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = min(kt_beg * HBK + 4 * kq + j, Kc1);
                ib[j] = p.idx_b ? p.idx_b[k] : k;
            }
        }
        auto gload = [&](int kt) {
            const int k0 = kt * HBK;
            if constexpr (AK) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ra[q] = *reinterpret_cast<const float4*>(pa[q] + min(k0 + 4 * kq, Kc4));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ra[j] = *reinterpret_cast<const float4*>(pa[0] + (uint32_t)ia[j] * (uint32_t)p.lda);
                    const int kn = min(k0 + HBK + 4 * kq + j, Kc1);
                    ia[j] = p.idx_a ? p.idx_a[kn] : kn;     // row of the NEXT tile, one tile ahead
                }
            }
            if constexpr (BKC) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    rb[q] = *reinterpret_cast<const float4*>(pb[q] + min(k0 + 4 * kq, Kc4));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    rb[j] = *reinterpret_cast<const float4*>(pb[0] + (uint32_t)ib[j] * (uint32_t)p.ldb);
                    const int kn = min(k0 + HBK + 4 * kq + j, Kc1);
                    ib[j] = p.idx_b ? p.idx_b[kn] : kn;
                }
            }
        };
        auto zero4 = [](float4& v) { v.x = 0.f; v.y = 0.f; v.z = 0.f; v.w = 0.f; };
        // registers -> LDS image [row][k] of K tile kt (rounded to 16 bit here)
        auto lstore = [&](int buf, int kt) {
            unsigned short* a = As + buf * OP;
            unsigned short* b = Bs + buf * OP;
            const bool tail = (kt + 1) * HBK > K;     // uniform
            if constexpr (AK) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (tail && kt * HBK + 4 * kq >= K) zero4(ra[q]);    // K % 4 == 0 here
                    const int r = (tid + NT * q) >> 3;
                    *reinterpret_cast<V4*>(a + r * HLD + 4 * kq) = HT::cvt(ra[q].x, ra[q].y, ra[q].z, ra[q].w);
                }
            } else {
                if (tail) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (kt * HBK + 4 * kq + j >= K) zero4(ra[j]);
                }
                const int r = 4 * (tid >> 3);
                *reinterpret_cast<V4*>(a + (r + 0) * HLD + 4 * kq) = HT::cvt(ra[0].x, ra[1].x, ra[2].x, ra[3].x);
                *reinterpret_cast<V4*>(a + (r + 1) * HLD + 4 * kq) = HT::cvt(ra[0].y, ra[1].y, ra[2].y, ra[3].y);
                *reinterpret_cast<V4*>(a + (r + 2) * HLD + 4 * kq) = HT::cvt(ra[0].z, ra[1].z, ra[2].z, ra[3].z);
                *reinterpret_cast<V4*>(a + (r + 3) * HLD + 4 * kq) = HT::cvt(ra[0].w, ra[1].w, ra[2].w, ra[3].w);
                if (do_colsum) {     // bias gradient: exact fp32 sums of the unrounded deltas
                    asum[0] += (ra[0].x + ra[1].x) + (ra[2].x + ra[3].x);
                    asum[1] += (ra[0].y + ra[1].y) + (ra[2].y + ra[3].y);
                    asum[2] += (ra[0].z + ra[1].z) + (ra[2].z + ra[3].z);
                    asum[3] += (ra[0].w + ra[1].w) + (ra[2].w + ra[3].w);
                }
            }
            if constexpr (BKC) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (tail && kt * HBK + 4 * kq >= K) zero4(rb[q]);
                    const int r = (tid + NT * q) >> 3;
                    *reinterpret_cast<V4*>(b + r * HLD + 4 * kq) = HT::cvt(rb[q].x, rb[q].y, rb[q].z, rb[q].w);
                }
            } else {
                if (tail) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (kt * HBK + 4 * kq + j >= K) zero4(rb[j]);
                }
                const int r = 4 * (tid >> 3);
                *reinterpret_cast<V4*>(b + (r + 0) * HLD + 4 * kq) = HT::cvt(rb[0].x, rb[1].x, rb[2].x, rb[3].x);
                *reinterpret_cast<V4*>(b + (r + 1) * HLD + 4 * kq) = HT::cvt(rb[0].y, rb[1].y, rb[2].y, rb[3].y);
                *reinterpret_cast<V4*>(b + (r + 2) * HLD + 4 * kq) = HT::cvt(rb[0].z, rb[1].z, rb[2].z, rb[3].z);
                *reinterpret_cast<V4*>(b + (r + 3) * HLD + 4 * kq) = HT::cvt(rb[0].w, rb[1].w, rb[2].w, rb[3].w);
            }
        };

        gload(kt_beg);
        lstore(0, kt_beg);
        __syncthreads();
        int buf = 0;
        const int li = lane & 31, kg = lane >> 5;
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            const bool more = kt + 1 < kt_end;
            if (more) gload(kt + 1);          // in flight behind this tile's MFMAs
            h16_compute<BF, TM, TN, HBK, HLD>(As + buf * OP + (wm * (TM * 32) + li) * HLD + 8 * kg,
                                              Bs + buf * OP + (wn * (TN * 32) + li) * HLD + 8 * kg, acc);
            if (more) lstore(buf ^ 1, kt + 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    if constexpr (!AK) {
        if (do_colsum) h16_colsum_out<8>(p, asum, m0, tid >> 3, kq);   // block-uniform
    }
    h16_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------ 16-bit operands in memory

// KC: both operands K-contiguous (forward, delta propagation with W^T) or both row-contiguous
// (weight gradients).  K tile 64; lda / ldb count 16-bit elements.
template <bool KC, bool BF, int BIG>
__global__ __launch_bounds__(HTile<BIG>::NT, BIG ? 2 : X_OCC) void gemm_x16_kernel(GemmArgs p)
{
    constexpr int HBM = HTile<BIG>::BM, HBN = HTile<BIG>::BN, NT = HTile<BIG>::NT;
    constexpr int WGN = HTile<BIG>::WGN, TM = HTile<BIG>::TM, TN = HTile<BIG>::TN;
    constexpr int KB = 64, LDH = KB + 8;          // 144-byte LDS rows
    constexpr int NLK = HBM * (KB / 8) / NT;      // K-contiguous: 16-byte pieces per thread and operand (4)
    constexpr int KQ = KB / 4;                    // row-contiguous: k quads per tile (16)
    constexpr int NMT = KQ * (HBM / 4) / NT;      // ... micro-tiles per thread and operand (2)
    using HT = H16<BF>;
    extern __shared__ __attribute__((aligned(16))) unsigned short hsmem[];
    constexpr int OP = HBM * LDH;
    unsigned short* As = hsmem;                   // [2][HBM][LDH]
    unsigned short* Bs = hsmem + 2 * OP;
    const unsigned short* A16 = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* B16 = reinterpret_cast<const unsigned short*>(p.B);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int M = p.M, N = p.N, K = p.K;
    const int mt = (M + HBM - 1) / HBM, nt = (N + HBN - 1) / HBN;
    const int swz = h16_swizzle(mt * nt);
    const int tile_n = swz % nt, tile_m = swz / nt;
    const int m0 = tile_m * HBM, n0 = tile_n * HBN;
    const int ktiles = (K + KB - 1) / KB;
    const int per = (ktiles + p.splits - 1) / p.splits;
    const int kt_beg = blockIdx.y * per;
    const int kt_end = min(ktiles, kt_beg + per);

    f32x16 acc[TM][TN];
    static_for<TM * TN>([&](auto IJ) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[decltype(IJ)::value / TN][decltype(IJ)::value % TN][r] = 0.f;
    });
    const bool do_colsum = !KC && p.colsum_a != nullptr && tile_n == 0;
    float asum[NMT][4];
#pragma unroll
    for (int u = 0; u < NMT; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) asum[u][e] = 0.f;

    if (kt_beg < kt_end) {
        u32x4 ha[NLK], hb[NLK];                   // K-contiguous staging
        u32x2 ga[NMT][4], gb[NMT][4];             // row-contiguous staging: 4 k-rows x 4 rows
        const unsigned short* pa16[KC ? NLK : NMT];
        const unsigned short* pb16[KC ? NLK : NMT];
        int ia[NMT][4], ib[NMT][4];
        const int Kc8 = (K - 1) & ~7, Kc1 = K - 1;
        if constexpr (KC) {
#pragma unroll
            for (int q = 0; q < NLK; ++q) {
                const int f = tid + NT * q, row = f / (KB / 8), k8 = 8 * (f % (KB / 8));
                pa16[q] = A16 + (int64_t)min(m0 + row, M - 1) * p.lda + k8;
                pb16[q] = B16 + (int64_t)min(n0 + row, N - 1) * p.ldb + k8;
            }
        } else {
#pragma unroll
            for (int u = 0; u < NMT; ++u) {
                const int mtile = tid + NT * u, kq = mtile % KQ, rq = mtile / KQ;
                pa16[u] = A16 + min(m0 + 4 * rq, (M - 1) & ~3);
                pb16[u] = B16 + min(n0 + 4 * rq, (N - 1) & ~3);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = min(kt_beg * KB + 4 * kq + j, Kc1);
                    ia[u][j] = p.idx_a ? p.idx_a[k] : k;
                    ib[u][j] = p.idx_b ? p.idx_b[k] : k;
                }
            }
        }
        auto gload = [&](int kt) {
            const int k0 = kt * KB;
            if constexpr (KC) {
#pragma unroll
                for (int q = 0; q < NLK; ++q) {
                    const int k8 = 8 * ((tid + NT * q) % (KB / 8));
                    const int off = min(k0, Kc8 - k8);         // the pointers already include k8
                    ha[q] = *reinterpret_cast<const u32x4*>(pa16[q] + off);
                    hb[q] = *reinterpret_cast<const u32x4*>(pb16[q] + off);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NMT; ++u) {
                    const int kq = (tid + NT * u) % KQ;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ga[u][j] = *reinterpret_cast<const u32x2*>(pa16[u] + (uint32_t)ia[u][j] * (uint32_t)p.lda);
                        gb[u][j] = *reinterpret_cast<const u32x2*>(pb16[u] + (uint32_t)ib[u][j] * (uint32_t)p.ldb);
                        const int kn = min(k0 + KB + 4 * kq + j, Kc1);
                        ia[u][j] = p.idx_a ? p.idx_a[kn] : kn;       // rows of the NEXT tile, one tile ahead
                        ib[u][j] = p.idx_b ? p.idx_b[kn] : kn;
                    }
                }
            }
        };
        auto lstore = [&](int buf, int kt) {
            unsigned short* a = As + buf * OP;
            unsigned short* b = Bs + buf * OP;
            const bool tail = (kt + 1) * KB > K;     // uniform
            if constexpr (KC) {
#pragma unroll
                for (int q = 0; q < NLK; ++q) {
                    const int f = tid + NT * q, r = f / (KB / 8), k8 = 8 * (f % (KB / 8));
                    const bool z = tail && kt * KB + k8 >= K;        // K % 8 == 0 here
                    const u32x4 zero = {0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4*>(a + r * LDH + k8) = z ? zero : ha[q];
                    *reinterpret_cast<u32x4*>(b + r * LDH + k8) = z ? zero : hb[q];
                }
            } else {
                // 4 x 4 transpose of 16-bit values: row r + e gets {k0, k1, k2, k3}
                auto tr = [](const u32x2 (&g)[4], unsigned short* dst) {
                    const u32x2 o0 = {__builtin_amdgcn_perm(g[1][0], g[0][0], 0x05040100u), __builtin_amdgcn_perm(g[3][0], g[2][0], 0x05040100u)};
                    const u32x2 o1 = {__builtin_amdgcn_perm(g[1][0], g[0][0], 0x07060302u), __builtin_amdgcn_perm(g[3][0], g[2][0], 0x07060302u)};
                    const u32x2 o2 = {__builtin_amdgcn_perm(g[1][1], g[0][1], 0x05040100u), __builtin_amdgcn_perm(g[3][1], g[2][1], 0x05040100u)};
                    const u32x2 o3 = {__builtin_amdgcn_perm(g[1][1], g[0][1], 0x07060302u), __builtin_amdgcn_perm(g[3][1], g[2][1], 0x07060302u)};
                    *reinterpret_cast<u32x2*>(dst + 0 * LDH) = o0;
                    *reinterpret_cast<u32x2*>(dst + 1 * LDH) = o1;
                    *reinterpret_cast<u32x2*>(dst + 2 * LDH) = o2;
                    *reinterpret_cast<u32x2*>(dst + 3 * LDH) = o3;
                };
#pragma unroll
                for (int u = 0; u < NMT; ++u) {
                    const int mtile = tid + NT * u, kq = mtile % KQ, rq = mtile / KQ;
                    if (tail) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (kt * KB + 4 * kq + j >= K) { ga[u][j] = {0u, 0u}; gb[u][j] = {0u, 0u}; }
                    }
                    tr(ga[u], a + 4 * rq * LDH + 4 * kq);
                    tr(gb[u], b + 4 * rq * LDH + 4 * kq);
                    if (do_colsum) {     // bias gradient from the (rounded) 16-bit deltas, fp32 sums
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            asum[u][0] += HT::tofloat(ga[u][j][0] & 0xffffu);
                            asum[u][1] += HT::tofloat(ga[u][j][0] >> 16);
                            asum[u][2] += HT::tofloat(ga[u][j][1] & 0xffffu);
                            asum[u][3] += HT::tofloat(ga[u][j][1] >> 16);
                        }
                    }
                }
            }
        };

        gload(kt_beg);
        lstore(0, kt_beg);
        __syncthreads();
        int buf = 0;
        const int li = lane & 31, kg = lane >> 5;
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            const bool more = kt + 1 < kt_end;
            if (more) gload(kt + 1);          // in flight behind this tile's MFMAs
            h16_compute<BF, TM, TN, KB, LDH>(As + buf * OP + (wm * (TM * 32) + li) * LDH + 8 * kg,
                                             Bs + buf * OP + (wn * (TN * 32) + li) * LDH + 8 * kg, acc);
            if (more) lstore(buf ^ 1, kt + 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    if constexpr (!KC) {
        if (do_colsum) {   // block-uniform
#pragma unroll
            for (int u = 0; u < NMT; ++u) {
                const int mtile = tid + NT * u;
                h16_colsum_out<KQ>(p, asum[u], m0, mtile / KQ, mtile % KQ);
            }
        }
    }
    h16_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------ host side

// the 256x256 tile when the output is large enough to give every CU whole tiles of it and its
// padding does not cost more than it saves (H = 1824 = 7.1 x 256 pads 12 %)
static int h16_pick_big(int M, int N, int K, bool g16)
{
    const char* force = getenv("SCTC_H16_TILE");     // diagnostics: 0 / 1
    if (force) return atoi(force) ? 1 : 0;
    auto padded = [](int v, int t) { return (double)((v + t - 1) / t * t) / v; };
    const double small = padded(M, 128) * padded(N, 128), big = padded(M, 256) * padded(N, 256);
    const int64_t tiles_big = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    // the LDS-DMA kernel runs the long-K weight gradients at twice the rate of the 128 x 128 register-staged
    // one (cfg-5 input layer, 2048 x 640 x 64000: 0.63 ms there): it may waste up to 25 % on padding
    // and fills the machine through split-K
    if (g16 && K >= 4096 && tiles_big >= 8 && big <= 1.25 * small) return 1;
    return (tiles_big >= 32 && big <= 1.06 * small) ? 1 : 0;
}

int64_t gemm_h16_plan_splits(int M, int N, int K, int* splits, int prec, int in16)
{
    if (prec == 3) return gemm_s3_plan_splits(M, N, K, splits);
    const int big = h16_pick_big(M, N, K, in16 && gemm_g16_enabled());
    const int bm = big ? 256 : 128, occ = big ? 1 : 2;
    const int mt = (M + bm - 1) / bm, nt = (N + bm - 1) / bm;
    const int ktiles = (K + 63) / 64;
    int s = 1;
    // fill whole rounds of 256 CUs x `occ` resident blocks, >= 4 K tiles of 64 (256 k) per split
    const int tiles = mt * nt, slots = 256 * occ;
    if (tiles < 2 * slots) {
        double best = 0.0;
        const int smax = std::min(64, std::max(1, ktiles / 4));
        for (int c = 1; c <= smax; ++c) {
            const int blocks = tiles * c;
            const int rounds = (blocks + slots - 1) / slots;
            const double eff = (double)blocks / ((double)rounds * slots) - 0.002 * c;
            if (eff > best + 1e-9) { best = eff; s = c; }
        }
    }
    *splits = s;
    return s > 1 ? (int64_t)s * M * (N + 1) : 0;
}

static int set_lds_once(void (*kern)(GemmArgs), size_t smem)
{
    if (smem <= 64 * 1024) return SCTC_OK;
    static std::mutex mu;
    static std::set<const void*> done;
    std::lock_guard<std::mutex> lock(mu);
    if (!done.count(reinterpret_cast<const void*>(kern))) {
        SCTC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        done.insert(reinterpret_cast<const void*>(kern));
    }
    return SCTC_OK;
}

template <int BIG>
static int launch_h16(const GemmArgs& a, hipStream_t stream)
{
    constexpr int BM = HTile<BIG>::BM, BN = HTile<BIG>::BN, NT = HTile<BIG>::NT;
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    dim3 grid(mt * nt, a.splits), block(NT);
    void (*kern)(GemmArgs) = nullptr;
    const bool bf = a.prec == 2;
    size_t smem;
    if (a.in16) {     // both operands 16-bit in memory, same layout (checked by launch_gemm_f32)
        smem = sizeof(unsigned short) * 2 * (BM + BN) * (64 + 8);
        if (a.a_kcontig) kern = bf ? gemm_x16_kernel<true, true, BIG> : gemm_x16_kernel<true, false, BIG>;
        else kern = bf ? gemm_x16_kernel<false, true, BIG> : gemm_x16_kernel<false, false, BIG>;
    } else {
        smem = sizeof(unsigned short) * 2 * (BM + BN) * (32 + 8);
#define SCTC_H16_PICK(AKV, BKV)                                                              \
    kern = bf ? gemm_h16_kernel<AKV, BKV, true, BIG> : gemm_h16_kernel<AKV, BKV, false, BIG>
        if (a.a_kcontig && a.b_kcontig) SCTC_H16_PICK(true, true);
        else if (a.a_kcontig && !a.b_kcontig) SCTC_H16_PICK(true, false);
        else if (!a.a_kcontig && a.b_kcontig) SCTC_H16_PICK(false, true);
        else SCTC_H16_PICK(false, false);
#undef SCTC_H16_PICK
    }
    SCTC_TRY(set_lds_once(kern, smem));
    hipLaunchKernelGGL(kern, grid, block, smem, stream, a);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int launch_gemm_h16_tiles(const GemmArgs& a, hipStream_t stream)
{
    if (a.prec == 3) return launch_gemm_s3(a, stream);
    if (h16_pick_big(a.M, a.N, a.K, gemm_g16_applies(a))) return gemm_g16_applies(a) ? launch_gemm_g16(a, stream) : launch_h16<1>(a, stream);
    return launch_h16<0>(a, stream);
}

}  // namespace sctc
