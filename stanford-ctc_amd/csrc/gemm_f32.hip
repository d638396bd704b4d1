// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fma
// chain, 157 TFLOP/s dense peak -- MI355X has no TF32/xf32 path, so this IS the
// matrix-core instruction for the reference's fp32 cm.dot calls).
//
// Tiling: 128x128 block tile, BK = 16, 256 threads = 4 waves in a 2x2 grid, each
// wave owning a 64x64 sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).
// Operands are staged through LDS in [k][m] / [k][n] order so that an MFMA
// fragment read (lane l -> element (k = l>>5, m = l&31)) is a conflict-free
// ds_read_b32 over 32 consecutive floats.  K-contiguous operands are transposed on
// the way into LDS (row stride 129: conflict-free scalar writes), row-contiguous
// operands are written with ds_write_b128 (row stride 132).  Global loads of tile
// t+1 are issued before the MFMAs of tile t and written to the other LDS buffer
// afterwards: one workgroup barrier per K tile; the LDS fragments of MFMA step kk+2
// are requested before the MFMAs of step kk issue.  Three blocks per CU (3 waves per
// SIMD, 34 KiB LDS each) overlap one block's barrier/epilogue with the others' MFMAs:
// measured 100-105 TFLOP/s on the cfg-3 shapes vs 88-100 with BK = 32 / 2 blocks.
// Workgroup ids are remapped so that the blocks of one XCD walk the N tiles of one
// A row-panel consecutively (panel stays in that XCD's 4 MiB L2).
#include "common.h"
#include "gemm_f32.h"
#include "gemm_h16_dev.h"     // h16_epilogue: the vectorised epilogue shared with the 16-bit kernels

namespace sctc {


#ifndef SCTC_GEMM_BK
#define SCTC_GEMM_BK 16
#endif
#ifndef SCTC_GEMM_OCC
#define SCTC_GEMM_OCC 3
#endif
#ifndef SCTC_GEMM_OCC1
#define SCTC_GEMM_OCC1 4
#endif
static constexpr int BK = SCTC_GEMM_BK;
static constexpr int KQ = BK / 4;   // float4 per row of a K-contiguous operand tile
// Block tile shapes (every wave owns TM x TN MFMA tiles of 32x32):
//   0: 128x128, 2x2 waves of 64x64, 256 threads -- the default;
//   1: 128x96,  4x1 waves of 32x96, 256 threads -- column counts that 128 tiles badly
//      (H = 1824 = 19 x 96 = 14.25 x 128);
// (96x96 with three waves was measured too: no quantisation loss on the H x H weight gradients,
// but 105.9 instead of 110.6 TFLOP/s -- smaller tiles, three-wave blocks.)
template <int SHAPE> struct TileCfg;
template <> struct TileCfg<0> { static constexpr int BM = 128, BN = 128, WGM = 2, WGN = 2, NT = 256, OCC = SCTC_GEMM_OCC; };
template <> struct TileCfg<1> { static constexpr int BM = 128, BN = 96, WGM = 4, WGN = 1, NT = 256, OCC = SCTC_GEMM_OCC1; };
//   2: 64x128 and 3: 128x64, 2x2 waves of 32x64 / 64x32 -- the output layer's contractions, whose
//      M (weight gradient) or N (forward) is the alphabet (33 -> 64 padded): a 128-wide tile there
//      is 3/4 padding (the weight gradient of the output layer ran 0.56 ms for 3.9 GFLOP)
template <> struct TileCfg<2> { static constexpr int BM = 64, BN = 128, WGM = 2, WGN = 2, NT = 256, OCC = 4; };
template <> struct TileCfg<3> { static constexpr int BM = 128, BN = 64, WGM = 2, WGN = 2, NT = 256, OCC = 4; };
static constexpr int N_SHAPES = 4;
// LDS row stride (floats): rows + 4 for both staging patterns.  Row-contiguous operands are
// written with ds_write_b128; K-contiguous ones are transposed on the way in, lane (r = l/4,
// c = l%4) writing element (k = 4c + j, row r): with a stride of 4 mod 16 the 64 lanes of one
// k-row write hit 64 distinct banks.  (SQ_LDS_BANK_CONFLICT still reports 15-23 % of the
// LDS-active cycles for these variants with this stride and with rows + 1 alike -- the compiler
// pairs two k-rows into one ds_write2_b32, 128 dwords for 64 banks -- and the GEMM rate is the
// same either way.)  Fragment reads are 32 consecutive floats per half-wave: conflict-free.
__host__ __device__ constexpr int lds_stride(bool kcontig, int rows) { (void)kcontig; return rows + 4; }
__host__ __device__ constexpr int lds_floats(int bm, int bn) { return 2 * BK * (bm + 4) + 2 * BK * (bn + 4); }

__device__ __forceinline__ float gemm_epilogue(const GemmArgs& p, float v, int row, int col)
{
    if (p.bias) v += p.bias[col];
    if (p.relu) v = fmaxf(v, 0.f);
    if (p.mask) v = p.mask[(int64_t)row * p.ldmask + col] > 0.f ? v : 0.f;
    if (p.mask16) v = gemm_pos16(p.mask16[(int64_t)row * p.ldmask16 + col]) ? v : 0.f;
    if (p.addend) v += p.add_scale * p.addend[(int64_t)row * p.ldadd + col];
    if (p.accumulate) v += p.C[(int64_t)row * p.ldc + col];
    return v;
}

template <bool AK, bool BKC, int SHAPE, bool S2 = false>
__global__ __launch_bounds__(TileCfg<SHAPE>::NT, TileCfg<SHAPE>::OCC) void gemm_f32_kernel(GemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = TileCfg<SHAPE>::BM, BN_ = TileCfg<SHAPE>::BN, NTHREADS = TileCfg<SHAPE>::NT;
    constexpr int WGM = TileCfg<SHAPE>::WGM, WGN = TileCfg<SHAPE>::WGN;
    constexpr int TM = BM / (WGM * 32), TN = BN_ / (WGN * 32);   // 32x32 MFMA tiles per wave
    constexpr int LDA = lds_stride(AK, BM), LDB = lds_stride(BKC, BN_);
    constexpr int OPA = BK * (BM + 4), OPB = BK * (BN_ + 4);     // floats per operand per buffer
    constexpr int NA = BM * BK / 4, NB = BN_ * BK / 4;           // float4 per operand tile
    constexpr int NLDA = (NA + NTHREADS - 1) / NTHREADS, NLDB = (NB + NTHREADS - 1) / NTHREADS;
    constexpr int MQ = BM / 4, NQ = BN_ / 4;                     // float4 per row-contiguous k-row
    float* As = smem;             // [2][BK][LDA]
    float* Bs = smem + 2 * OPA;   // [2][BK][LDB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int M = p.M, N = p.N, K = p.K;
    const int mt = (M + BM - 1) / BM, nt = (N + BN_ - 1) / BN_;
    const int nblk = mt * nt;
    const int swz = gemm_xcd_tile(nblk);      // XCD-aware, bijective, split-K aware (gemm_f32.h)
    const int tile_n = swz % nt, tile_m = swz / nt;
    const int m0 = tile_m * BM, n0 = tile_n * BN_;

    // split-K range of this block
    const int ktiles = (K + BK - 1) / BK;
    const int per = (ktiles + p.splits - 1) / p.splits;
    const int kt_beg = blockIdx.y * per;
    const int kt_end = min(ktiles, kt_beg + per);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fused column sums of A (bias gradient): only the blocks of the first N tile keep them
    constexpr int NLDA_ = (BM * BK / 4 + NTHREADS - 1) / NTHREADS;
    const bool do_colsum = !AK && p.colsum_a != nullptr && tile_n == 0;
    float4 asum[NLDA_];
#pragma unroll
    for (int q = 0; q < NLDA_; ++q) asum[q] = make_float4(0.f, 0.f, 0.f, 0.f);

    if (kt_beg < kt_end) {
        // ---- global -> register staging.  Everything that does not depend on the K tile is
        // hoisted out of the K loop: per-thread base pointers and, for gathered rows, the row
        // index of the NEXT tile (fetched one tile ahead, so no load waits on a load of the same
        // iteration).  Loads are unconditional from clamped addresses: rows/columns beyond M/N
        // read some valid row and only feed accumulators the epilogue never stores; the K tail
        // is zeroed when the tile goes to LDS.  (A load behind a divergent branch costs an
        // exec-mask branch plus an early s_waitcnt, a select right behind the load a vmcnt(0).)
        // Item f of a tile: K-contiguous -> row f / KQ, k = 4 (f % KQ);
        //                   row-contiguous -> k-row f / (rows/4), column 4 (f % (rows/4)).
        float4 ra[NLDA], rb[NLDB];
        float4 ra2[S2 ? NLDA : 1];                 // S2: the second addend of A (GemmArgs::A2)
        const int64_t d2 = S2 ? p.A2 - p.A : 0;    // same layout: one uniform element offset
        const bool sum_out = S2 && p.a_sum != nullptr && tile_n == 0;
        const float* pa[NLDA];
        const float* pb[NLDB];
        int ia[NLDA], ib[NLDB];   // row-contiguous operands: (gathered) row of the next tile
        const int Kc4 = (K - 1) & ~3, Kc1 = K - 1;
#pragma unroll
        for (int q = 0; q < NLDA; ++q) {
            const int f = min(tid + NTHREADS * q, NA - 1);
            if constexpr (AK) {
                pa[q] = p.A + (int64_t)min(m0 + f / KQ, M - 1) * p.lda;
            } else {
                pa[q] = p.A + min(m0 + 4 * (f % MQ), (M - 1) & ~3);
                const int k = min(kt_beg * BK + f / MQ, Kc1);
                ia[q] = p.idx_a ? p.idx_a[k] : k;
            }
        }
#pragma unroll
        for (int q = 0; q < NLDB; ++q) {
            const int f = min(tid + NTHREADS * q, NB - 1);
            if constexpr (BKC) {
                pb[q] = p.B + (int64_t)min(n0 + f / KQ, N - 1) * p.ldb;
            } else {
                pb[q] = p.B + min(n0 + 4 * (f % NQ), (N - 1) & ~3);
                const int k = min(kt_beg * BK + f / NQ, Kc1);
                ib[q] = p.idx_b ? p.idx_b[k] : k;
            }
        }
        auto gload = [&](int kt) {
            const int k0 = kt * BK;
#pragma unroll
            for (int q = 0; q < NLDA; ++q) {
                const int f = min(tid + NTHREADS * q, NA - 1);
                if constexpr (AK) {
                    ra[q] = *reinterpret_cast<const float4*>(pa[q] + min(k0 + 4 * (f % KQ), Kc4));
                    if constexpr (S2) ra2[q] = *reinterpret_cast<const float4*>(pa[q] + d2 + min(k0 + 4 * (f % KQ), Kc4));
                } else {
                    ra[q] = *reinterpret_cast<const float4*>(pa[q] + (uint32_t)ia[q] * (uint32_t)p.lda);
                    if constexpr (S2) ra2[q] = *reinterpret_cast<const float4*>(pa[q] + d2 + (uint32_t)ia[q] * (uint32_t)p.lda);
                    const int kn = min(k0 + BK + f / MQ, Kc1);
                    ia[q] = p.idx_a ? p.idx_a[kn] : kn;
                }
            }
#pragma unroll
            for (int q = 0; q < NLDB; ++q) {
                const int f = min(tid + NTHREADS * q, NB - 1);
                if constexpr (BKC) {
                    rb[q] = *reinterpret_cast<const float4*>(pb[q] + min(k0 + 4 * (f % KQ), Kc4));
                } else {
                    rb[q] = *reinterpret_cast<const float4*>(pb[q] + (uint32_t)ib[q] * (uint32_t)p.ldb);
                    const int kn = min(k0 + BK + f / NQ, Kc1);
                    ib[q] = p.idx_b ? p.idx_b[kn] : kn;
                }
            }
        };
        auto zero_tail = [&](float4& v, int k) {   // component-wise: a struct select goes via scratch
            const bool ok = k < K;
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        };
        // registers -> LDS tile `buf` of K tile kt; the A and B halves are separate so that they
        // can be slotted between different MFMA groups
        auto lstore_a = [&](int buf, int kt) {
            float* a = As + buf * OPA;
            const bool tail = (kt + 1) * BK > K;   // uniform
#pragma unroll
            for (int q = 0; q < NLDA; ++q) {
                const int f = tid + NTHREADS * q;
                if (NA % NTHREADS != 0 && f >= NA) continue;
                if constexpr (S2) {   // the sum, and its one stored copy (in-range elements only: clamped loads repeat)
                    ra[q].x += ra2[q].x; ra[q].y += ra2[q].y; ra[q].z += ra2[q].z; ra[q].w += ra2[q].w;
                    if (sum_out) {
                        const int row = AK ? m0 + f / KQ : kt * BK + f / MQ;
                        const int col = AK ? kt * BK + 4 * (f % KQ) : m0 + 4 * (f % MQ);
                        if (row < (AK ? M : K) && col < (AK ? K : M))
                            *reinterpret_cast<float4*>(p.a_sum + (int64_t)row * p.lda + col) = ra[q];
                    }
                }
                if constexpr (AK) {
                    const int r = f / KQ, kq = 4 * (f % KQ);
                    if (tail) zero_tail(ra[q], kt * BK + kq);
                    a[(kq + 0) * LDA + r] = ra[q].x;
                    a[(kq + 1) * LDA + r] = ra[q].y;
                    a[(kq + 2) * LDA + r] = ra[q].z;
                    a[(kq + 3) * LDA + r] = ra[q].w;
                } else {
                    const int kr = f / MQ, m = 4 * (f % MQ);
                    if (tail) zero_tail(ra[q], kt * BK + kr);
                    *reinterpret_cast<float4*>(a + kr * LDA + m) = ra[q];
                    if (do_colsum) {
                        asum[q].x += ra[q].x; asum[q].y += ra[q].y;
                        asum[q].z += ra[q].z; asum[q].w += ra[q].w;
                    }
                }
            }
        };
        auto lstore_b = [&](int buf, int kt) {
            float* b = Bs + buf * OPB;
            const bool tail = (kt + 1) * BK > K;
#pragma unroll
            for (int q = 0; q < NLDB; ++q) {
                const int f = tid + NTHREADS * q;
                if (NB % NTHREADS != 0 && f >= NB) continue;   // 128x96: waves 2,3 hold no 2nd item
                if constexpr (BKC) {
                    const int r = f / KQ, kq = 4 * (f % KQ);
                    if (tail) zero_tail(rb[q], kt * BK + kq);
                    b[(kq + 0) * LDB + r] = rb[q].x;
                    b[(kq + 1) * LDB + r] = rb[q].y;
                    b[(kq + 2) * LDB + r] = rb[q].z;
                    b[(kq + 3) * LDB + r] = rb[q].w;
                } else {
                    const int kr = f / NQ, n = 4 * (f % NQ);
                    if (tail) zero_tail(rb[q], kt * BK + kr);
                    *reinterpret_cast<float4*>(b + kr * LDB + n) = rb[q];
                }
            }
        };

        gload(kt_beg);
        lstore_a(0, kt_beg);
        lstore_b(0, kt_beg);
        __syncthreads();
        int buf = 0;
        const int kh = lane >> 5, li = lane & 31;
        constexpr int NSTEP = BK / 2;
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            const bool more = kt + 1 < kt_end;
#ifdef SCTC_GEMM_STAMP
            const bool st_on = p.splitk_ws && p.splits == 1 && blockIdx.x == 8 && tid == 0 &&
                               kt >= kt_beg + 16 && kt < kt_beg + 32;
            unsigned* st_out = reinterpret_cast<unsigned*>(p.splitk_ws) + (kt - kt_beg - 16) * 8;
            if (st_on) { st_out[0] = (unsigned)clock64(); st_out[5] = (unsigned)wall_clock64(); }
#endif
            const float* a = As + buf * OPA + kh * LDA + wm * (TM * 32) + li;
            const float* b = Bs + buf * OPB + kh * LDB + wn * (TN * 32) + li;
            // Software pipeline inside one K tile (NSTEP groups of TM x TN MFMAs, 64 matrix-pipe
            // cycles each).  Everything that is not an MFMA is slotted BEHIND a group so that it
            // issues in that group's shadow: the LDS fragments of group s+1, the global loads of
            // tile kt+1 behind group 0, its LDS stores behind the last groups (the data has had
            // >= NSTEP-3 groups to arrive).
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = a[32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = b[32 * j];
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int kk = 2 * s;
                float an[TM], bn[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) an[i] = (s + 1 < NSTEP) ? a[(kk + 2) * LDA + 32 * i] : 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) bn[j] = (s + 1 < NSTEP) ? b[(kk + 2) * LDB + 32 * j] : 0.f;
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above the MFMAs
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        // operands SWAPPED: the accumulator tile is the transpose of the output block
                        // (lane = output row, 4 consecutive columns per register quad), see h16_epilogue
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j], af[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = an[i];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = bn[j];
                if (more) {
                    if (s == 0) { __builtin_amdgcn_sched_barrier(0); gload(kt + 1); }
                    if (s == NSTEP - 3) { __builtin_amdgcn_sched_barrier(0); lstore_a(buf ^ 1, kt + 1); }
                    if (s == NSTEP - 2) { __builtin_amdgcn_sched_barrier(0); lstore_b(buf ^ 1, kt + 1); }
                }
            }
#ifdef SCTC_GEMM_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (st_on) st_out[3] = (unsigned)clock64();
#endif
            __syncthreads();
#ifdef SCTC_GEMM_STAMP
            if (st_on) st_out[4] = (unsigned)clock64();
#endif
            buf ^= 1;
        }
    }
    if constexpr (!AK) {
        if (do_colsum) {   // block-uniform; the K loop's last barrier has released the LDS tiles
            constexpr int LDA_ = lds_stride(AK, BM), MQ_ = BM / 4;
#pragma unroll
            for (int q = 0; q < NLDA_; ++q) {
                const int f = tid + NTHREADS * q;
                *reinterpret_cast<float4*>(smem + (f / MQ_) * LDA_ + 4 * (f % MQ_)) = asum[q];
            }
            __syncthreads();
            if (tid < BM && m0 + tid < M) {
                float v = 0.f;
#pragma unroll
                for (int kr = 0; kr < BK; ++kr) v += smem[kr * LDA_ + tid];   // fixed order
                if (p.splits > 1)
                    p.splitk_ws[(int64_t)p.splits * M * N + (int64_t)blockIdx.y * M + m0 + tid] = v;
                else
                    p.colsum_a[m0 + tid] = p.accumulate ? p.colsum_a[m0 + tid] + v : v;
            }
        }
    }

    // epilogue: bias / relu / mask / addend / accumulate, 16-byte accesses (each lane owns 4 consecutive
    // columns of its row).  The scalar version this replaces kept ~50 dwords per lane in scratch.
    h16_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p)
{
    const int64_t total = (int64_t)p.M * p.N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int z = 0; z < p.splits; ++z) v += p.splitk_ws[z * total + i];  // fixed order
        const int row = (int)(i / p.N), col = (int)(i % p.N);
        const float o = gemm_epilogue(p, v, row, col);
        if (!p.skip_c32) p.C[(int64_t)row * p.ldc + col] = o;
        if (p.C16a) p.C16a[(int64_t)row * p.ldc16 + col] = __builtin_bit_cast(unsigned short, (_Float16)o);
        if (p.C16b) p.C16b[(int64_t)row * p.ldc16 + col] = __builtin_bit_cast(unsigned short, (__bf16)o);
    }
    if (p.colsum_a && !p.a_kcontig) {
        const float* part = p.splitk_ws + (int64_t)p.splits * total;
        for (int m = blockIdx.x * 256 + threadIdx.x; m < p.M; m += gridDim.x * 256) {
            float v = 0.f;
            for (int z = 0; z < p.splits; ++z) v += part[(int64_t)z * p.M + m];   // fixed order
            p.colsum_a[m] = p.accumulate ? p.colsum_a[m] + v : v;
        }
    }
}

struct ShapeInfo { int bm, bn, occ; double penalty; };
static const ShapeInfo kShapes[N_SHAPES] = {
    {TileCfg<0>::BM, TileCfg<0>::BN, TileCfg<0>::OCC, 0.00},
    {TileCfg<1>::BM, TileCfg<1>::BN, TileCfg<1>::OCC, 0.02},   // smaller tiles must save at least this much padded work
    {TileCfg<2>::BM, TileCfg<2>::BN, TileCfg<2>::OCC, 0.10},
    {TileCfg<3>::BM, TileCfg<3>::BN, TileCfg<3>::OCC, 0.10},
};

// tile shape with the least padded (wasted) matrix-core work for an M x N output
int gemm_pick_shape(int M, int N)
{
    const char* force = getenv("SCTC_GEMM_SHAPE");   // diagnostics: 0 / 1
    if (force) return std::max(0, std::min(N_SHAPES - 1, atoi(force)));
    int best = 0;
    double best_cost = 1e30;
    for (int i = 0; i < N_SHAPES; ++i) {
        const double pm = (double)((M + kShapes[i].bm - 1) / kShapes[i].bm * kShapes[i].bm) / M;
        const double pn = (double)((N + kShapes[i].bn - 1) / kShapes[i].bn * kShapes[i].bn) / N;
        const double cost = pm * pn + kShapes[i].penalty;
        if (cost < best_cost - 1e-12) { best_cost = cost; best = i; }
    }
    return best;
}

int64_t gemm_plan_splits(int M, int N, int K, int* splits, int prec, int in16)
{
    if (prec) return gemm_h16_plan_splits(M, N, K, splits, prec, in16);
    const ShapeInfo& sh = kShapes[gemm_pick_shape(M, N)];
    const int mt = (M + sh.bm - 1) / sh.bm, nt = (N + sh.bn - 1) / sh.bn;
    const int ktiles = (K + BK - 1) / BK;
    int s = 1;
    // 256 CUs x `occ` resident blocks.  A grid that is not a multiple of that leaves a partial
    // last round (225 tiles x 3 splits = 675 blocks ran at 66 %); pick the split that fills
    // whole rounds best, keeping >= 8 K tiles per split.
    const int tiles = mt * nt, slots = 256 * sh.occ;
    if (tiles < 2 * slots) {
        double best = 0.0;
        const int smax = std::min(64, std::max(1, ktiles / 8));
        for (int c = 1; c <= smax; ++c) {
            const int blocks = tiles * c;
            const int rounds = (blocks + slots - 1) / slots;
            // efficiency of the last round, mildly penalising the extra partial traffic
            const double eff = (double)blocks / ((double)rounds * slots) - 0.002 * c;
            if (eff > best + 1e-9) { best = eff; s = c; }
        }
    }
    *splits = s;
    return s > 1 ? (int64_t)s * M * (N + 1) : 0;   // + [splits][M] column-sum partials
}

template <int SHAPE>
static int launch_tiles(GemmArgs a, hipStream_t stream)
{
    constexpr int BM = TileCfg<SHAPE>::BM, BN_ = TileCfg<SHAPE>::BN;
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN_ - 1) / BN_;
    dim3 grid(mt * nt, a.splits), block(TileCfg<SHAPE>::NT);
    const size_t smem = sizeof(float) * lds_floats(BM, BN_);  // 2 operands x 2 buffers
    void (*kern)(GemmArgs) = nullptr;
    if (a.A2) kern = a.a_kcontig ? gemm_f32_kernel<true, true, SHAPE, true> : gemm_f32_kernel<false, false, SHAPE, true>;
    else if (a.a_kcontig && a.b_kcontig) kern = gemm_f32_kernel<true, true, SHAPE>;
    else if (a.a_kcontig && !a.b_kcontig) kern = gemm_f32_kernel<true, false, SHAPE>;
    else if (!a.a_kcontig && a.b_kcontig) kern = gemm_f32_kernel<false, true, SHAPE>;
    else kern = gemm_f32_kernel<false, false, SHAPE>;
    static bool attr_set[6] = {false, false, false, false, false, false};
    const int vi = a.A2 ? 4 + (a.a_kcontig ? 1 : 0) : (a.a_kcontig ? 2 : 0) + (a.b_kcontig ? 1 : 0);
    if (!attr_set[vi]) {
        SCTC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set[vi] = true;
    }
    hipLaunchKernelGGL(kern, grid, block, smem, stream, a);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

int launch_gemm_f32(GemmArgs a, hipStream_t stream)
{
    if (a.M <= 0 || a.N <= 0) return SCTC_OK;
    SCTC_CHECK_ARG(a.K >= 0, "gemm: negative K");
    SCTC_CHECK_ARG(a.lda % 4 == 0 && a.ldb % 4 == 0, "gemm: lda/ldb must be multiples of 4");
    SCTC_CHECK_ARG(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0,
                   "gemm: operands must be 16-byte aligned");
    if (a.in16) {
        SCTC_CHECK_ARG(a.prec != 0 && a.a_kcontig == a.b_kcontig && a.lda % 8 == 0 && a.ldb % 8 == 0 &&
                           (!a.a_kcontig || a.K % 8 == 0),
                       "gemm: 16-bit operands need prec != 0, equal operand layouts, lda/ldb % 8 == 0 "
                       "and (K-contiguous) K % 8 == 0");
    } else {
        SCTC_CHECK_ARG((!a.C16a && !a.C16b) || a.prec != 0, "gemm: 16-bit shadow outputs need prec != 0");
    }
    SCTC_CHECK_ARG(!a.skip_c32 || ((a.C16a || a.C16b) && !a.accumulate && a.prec != 0),
                   "gemm: skip_c32 needs a 16-bit shadow output, no accumulate and prec != 0");
    SCTC_CHECK_ARG(!a.mask16 || (!a.mask && a.prec != 0 && a.ldmask16 % 4 == 0), "gemm: mask16 excludes mask, needs prec != 0");
    if (a.a_kcontig) SCTC_CHECK_ARG(a.K % 4 == 0, "gemm: K must be a multiple of 4 (A K-contig)");
    else SCTC_CHECK_ARG(a.M % 4 == 0, "gemm: M must be a multiple of 4 (A row-contig)");
    if (a.b_kcontig) SCTC_CHECK_ARG(a.K % 4 == 0, "gemm: K must be a multiple of 4 (B K-contig)");
    else SCTC_CHECK_ARG(a.N % 4 == 0, "gemm: N must be a multiple of 4 (B row-contig)");
    SCTC_CHECK_ARG(!a.A2 || (a.prec == 0 && a.a_kcontig == a.b_kcontig && !a.idx_a && ((uintptr_t)a.A2 & 15) == 0 &&
                              ((uintptr_t)a.a_sum & 15) == 0),
                   "gemm: a two-addend A operand needs prec 0, layout NT or TN, no row gather, 16-byte alignment");
    SCTC_CHECK_ARG(!a.a_sum || a.A2, "gemm: a_sum without A2");
    if (a.splits < 1) a.splits = 1;
    if (a.splits > 1) SCTC_CHECK_ARG(a.splitk_ws != nullptr, "gemm: split-K without workspace");
    if (a.prec) {
        SCTC_CHECK_ARG(a.prec >= 1 && a.prec <= 3, "gemm: unknown operand precision %d", a.prec);
        SCTC_CHECK_ARG(a.prec != 3 || (!a.in16 && !a.C16a && !a.C16b),
                       "gemm: the three-term split takes fp32 operands and writes no 16-bit shadows");
        SCTC_TRY(launch_gemm_h16_tiles(a, stream));
    } else {
        switch (gemm_pick_shape(a.M, a.N)) {
            case 1: SCTC_TRY(launch_tiles<1>(a, stream)); break;
            case 2: SCTC_TRY(launch_tiles<2>(a, stream)); break;
            case 3: SCTC_TRY(launch_tiles<3>(a, stream)); break;
            default: SCTC_TRY(launch_tiles<0>(a, stream)); break;
        }
    }
    if (a.splits > 1) {
        const int64_t total = (int64_t)a.M * a.N;
        int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
        SCTC_HIP_TRY(hipGetLastError());
    }
    return SCTC_OK;
}

}  // namespace sctc

// ---- C ABI: cm.dot(A, B, target=C) of cudamat as used by brnnet.py:140,196,204,227-230
static int gemm_entry(const float* A_dev, int64_t lda, int32_t a_kcontig, const float* B_dev,
                      int64_t ldb, int32_t b_kcontig, float* C_dev, int64_t ldc, int32_t M, int32_t N,
                      int32_t K, const float* bias_dev, int32_t relu, void* workspace_dev,
                      size_t workspace_bytes, void* stream, int prec);

extern "C" int sctc_gemm_h16(const float* A_dev, int64_t lda, int32_t a_kcontig, const float* B_dev,
                             int64_t ldb, int32_t b_kcontig, float* C_dev, int64_t ldc, int32_t M,
                             int32_t N, int32_t K, const float* bias_dev, int32_t relu,
                             int32_t operand_dtype, void* workspace_dev, size_t workspace_bytes,
                             void* stream)
{
    using namespace sctc;
    const int in16 = (operand_dtype & SCTC_OPERANDS_16BIT) ? 1 : 0;
    operand_dtype &= ~SCTC_OPERANDS_16BIT;
    SCTC_CHECK_ARG(operand_dtype == SCTC_F16 || operand_dtype == SCTC_BF16 || (operand_dtype == SCTC_BF16X3 && !in16),
                   "gemm_h16: operand_dtype must be SCTC_F16, SCTC_BF16 (optionally | SCTC_OPERANDS_16BIT) or SCTC_BF16X3");
    return gemm_entry(A_dev, lda, a_kcontig, B_dev, ldb, b_kcontig, C_dev, ldc, M, N, K, bias_dev,
                      relu, workspace_dev, workspace_bytes, stream,
                      (operand_dtype == SCTC_F16 ? 1 : (operand_dtype == SCTC_BF16 ? 2 : 3)) | (in16 ? 0x100 : 0));
}

extern "C" int sctc_gemm_f32(const float* A_dev, int64_t lda, int32_t a_kcontig,
                             const float* B_dev, int64_t ldb, int32_t b_kcontig, float* C_dev,
                             int64_t ldc, int32_t M, int32_t N, int32_t K, const float* bias_dev,
                             int32_t relu, void* workspace_dev, size_t workspace_bytes,
                             void* stream)
{
    return gemm_entry(A_dev, lda, a_kcontig, B_dev, ldb, b_kcontig, C_dev, ldc, M, N, K, bias_dev,
                      relu, workspace_dev, workspace_bytes, stream, 0);
}

static int gemm_entry(const float* A_dev, int64_t lda, int32_t a_kcontig, const float* B_dev,
                      int64_t ldb, int32_t b_kcontig, float* C_dev, int64_t ldc, int32_t M, int32_t N,
                      int32_t K, const float* bias_dev, int32_t relu, void* workspace_dev,
                      size_t workspace_bytes, void* stream, int prec)
{
    using namespace sctc;
    SCTC_CHECK_ARG(A_dev && B_dev && C_dev, "gemm: null pointer");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A_dev; g.lda = lda; g.a_kcontig = a_kcontig;
    g.B = B_dev; g.ldb = ldb; g.b_kcontig = b_kcontig;
    g.C = C_dev; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias_dev; g.relu = relu;
    g.prec = prec & 0xff;
    g.in16 = (prec & 0x100) ? 1 : 0;
    prec &= 0xff;
    int splits = 1;
    const int64_t need = gemm_plan_splits(M, N, K, &splits, prec, g.in16 && a_kcontig == b_kcontig);
    if (splits > 1 && workspace_dev && workspace_bytes >= (size_t)need * sizeof(float)) {
        g.splits = splits;
        g.splitk_ws = (float*)workspace_dev;
    } else {
        g.splits = 1;
#ifdef SCTC_GEMM_STAMP
        g.splitk_ws = (float*)workspace_dev;   // diagnostics build: timeline stamps land here
#endif
    }
    return launch_gemm_f32(g, (hipStream_t)stream);
}
