// 16-bit-operand GEMM staged by LDS-DMA (round 4): the contractions of the "fp16 activations"
// configuration (BASELINE configs[4]; brnnet.py:140 fwd, :196 wgrad, :204 dgrad, :227-230 recurrent
// wgrad) on operands that are ALREADY 16-bit in memory (the shadow copies their producers write),
// 256 x 256 output tiles, K tile 64, one block of 8 waves (2 x 4, 128 x 64 per wave) per CU.
//
// What round 2's gemm_x16_kernel (gemm_h16.hip) was bound by -- its ablation: MFMAs + fragment reads
// 0.82 ms, + LDS stores and barrier 1.06 ms, + global loads 2.10 ms at 8192^3 -- is the operand path
// global -> VGPR -> ds_write -> LDS.  Here the operands never touch a VGPR on their way in:
//   * global_load_lds_dwordx4: every lane names 16 bytes of global memory, the wave's 1 KiB lands at
//     one wave-uniform LDS address, lane-linear.  No staging registers, no ds_write pass, no
//     s_waitcnt on load data in front of the MFMAs; the loads of K tile t+1 are issued before the
//     MFMAs of tile t and waited for once, in front of the tile's one barrier.
//   * the LDS image is unpadded and XOR-swizzled; because the LDS side of a glds is lane-linear the
//     swizzle is applied to the lane's SOURCE address and undone by the fragment reads:
//       K-contiguous operands (forward, delta propagation with W^T): image [row][8 x 16 B], the 16-byte
//         k-slot s of row r sits at slot s ^ ((r >> 1) & 7).  Eight consecutive lanes fetch one full
//         128-byte line; a fragment (lane l: row l & 31, 8 k at slot 2 kk + (l >> 5)) is ONE
//         ds_read_b128 whose 16-lane groups cover the 64 banks exactly once.
//       row-contiguous operands (weight gradients, [k][m] in memory): image [k][32 x 16 B] (512-byte
//         k-rows, straight copy), the 16-byte m-piece q of k-row k at piece q ^ ((k & 3) << 2).  32
//         consecutive lanes fetch 512 contiguous bytes.  MFMA fragments need 8 consecutive k per
//         lane: two ds_read_b64_tr_b16 (the hardware transposing read: 16 lanes name a [4 k][16 m]
//         block, lane i receives column i's 4 k); the swizzle puts the four k-rows of a read on the
//         four 64-byte quarters of the bank row.
//   * operands SWAPPED in the MFMA (B fragment first) like gemm_x16_kernel: a lane owns 4 consecutive
//     output columns per accumulator quad, the shared epilogue (gemm_h16_dev.h) stores 16 B / 8 B.
//   * rows / k beyond the matrix: the lane's source address is redirected to a 16-byte block of zeros
//     (a glds cannot be predicated per lane without leaving stale LDS bytes).
//   * tile order: bands of 4 tile rows walked column by column, so that the 32 tiles an XCD runs at a
//     time are a 4 x 8 block (4 A panels + 8 B panels through its 4 MiB L2) -- at N = K = 2048 that
//     equals the row-major order, for square problems it cuts the memory-side traffic.
// The fused bias gradient (column sums of a row-contiguous A, brnnet.py:200) is taken from the LDS
// image by the blocks of the first N tile (fp32 sums of the 16-bit deltas, fixed order).
#include <mutex>
#include <set>
#include <type_traits>

#include "common.h"
#include "gemm_f32.h"
#include "gemm_h16_dev.h"

namespace sctc {

typedef __attribute__((address_space(3))) void g16_lds_void;
typedef __attribute__((address_space(1))) const void g16_gbl_void;
typedef __attribute__((address_space(3))) b16x4 g16_lds_b16x4;

static __device__ __attribute__((aligned(16))) unsigned g16_zero[4];   // the source of out-of-range pieces

static constexpr int G_BM = 256, G_BN = 256, G_BK = 64, G_NT = 512;
static constexpr int G_OPB = G_BM * G_BK * 2;    // bytes per operand and buffer (32 KiB)
static constexpr int G_BUFB = 2 * G_OPB;         // A + B
static constexpr int G_GR = 4;                   // tile rows per band of the tile walk
// diagnostics builds (tools/build_variant.sh ... "-DSCTC_G16_ABLATE=n", tests/gpu_g16.py speed): bit 0 no MFMAs,
// bit 1 no fragment reads, bit 2 no LDS-DMA after the first K tile, bit 3 no vmcnt wait / barrier per K tile
#ifndef SCTC_G16_ABLATE
#define SCTC_G16_ABLATE 0
#endif
static constexpr int G_ABL = SCTC_G16_ABLATE;

template <bool KC, bool BF>
__global__ __launch_bounds__(G_NT, 2) void gemm_g16_kernel(GemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    using HT = H16<BF>;
    using V8 = typename HT::V8;
    constexpr int TM = 4, TN = 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int M = p.M, N = p.N, K = p.K;
    const int mt = (M + G_BM - 1) / G_BM, nt = (N + G_BN - 1) / G_BN;
    // XCD-aware bijective remap (block b runs on XCD b % 8: an XCD walks consecutive positions), then
    // the banded walk: position -> (band of G_GR tile rows, column, row in band)
    int tile_m, tile_n;
    {
        const int pos = h16_swizzle(mt * nt);
        const int band = pos / (G_GR * nt), in_band = pos - band * (G_GR * nt);
        const int rows = min(G_GR, mt - band * G_GR);
        tile_n = in_band / rows;
        tile_m = band * G_GR + (in_band - tile_n * rows);
    }
    const int m0 = tile_m * G_BM, n0 = tile_n * G_BN;
    const int ktiles = (K + G_BK - 1) / G_BK;
    const int per = (ktiles + p.splits - 1) / p.splits;
    const int kt_beg = blockIdx.y * per;
    const int kt_end = min(ktiles, kt_beg + per);
    const unsigned short* A16 = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* B16 = reinterpret_cast<const unsigned short*>(p.B);

    f32x16 acc[TM][TN];
    static_for<TM * TN>([&](auto IJ) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[decltype(IJ)::value / TN][decltype(IJ)::value % TN][r] = 0.f;
    });
    const bool do_colsum = !KC && p.colsum_a != nullptr && tile_n == 0;   // block-uniform
    float csum[4] = {0.f, 0.f, 0.f, 0.f};

    if (kt_beg < kt_end) {
        const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(g16_zero);
        // ---- staging: 4 pieces of 16 B per thread, operand and K tile; LDS position of piece
        // (round q, thread t) = (q * 512 + t) * 16, i.e. wave-uniform base + lane * 16
        const unsigned short* srcA[4];
        const unsigned short* srcB[4];
        bool okA[4], okB[4];
        int kpiece;      // KC: the thread's k offset inside a tile (elements); !KC: unused
        int krow[4];     // !KC: the k-row of round q
        if constexpr (KC) {
            const int s = (tid & 7) ^ ((tid >> 4) & 7);            // source k-slot of LDS slot tid & 7
            kpiece = 8 * s;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = q * 64 + (tid >> 3);
                srcA[q] = A16 + (int64_t)min(m0 + r, M - 1) * p.lda + kpiece;
                srcB[q] = B16 + (int64_t)min(n0 + r, N - 1) * p.ldb + kpiece;
                okA[q] = okB[q] = true;
                krow[q] = 0;
            }
        } else {
            kpiece = 0;
            const int m8 = (tid & 31) ^ (((tid >> 5) & 3) << 2);   // source m-piece of LDS piece tid & 31
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                krow[q] = q * 16 + (tid >> 5);
                okA[q] = m0 + 8 * m8 < M;
                okB[q] = n0 + 8 * m8 < N;
                srcA[q] = A16 + m0 + 8 * m8;
                srcB[q] = B16 + n0 + 8 * m8;
            }
        }
        // one round (q) of the next K tile's pieces: 1 KiB of A and 1 KiB of B per wave
        auto stage_q = [&](int buf, int kt, int q) {
            const int k0 = kt * G_BK;
            unsigned char* dst = gsm + buf * G_BUFB + wave * 1024 + q * 8192;
            const unsigned short* a;
            const unsigned short* b;
            if constexpr (KC) {
                const bool kin = k0 + kpiece < K;               // K % 8 == 0
                a = kin ? srcA[q] + k0 : zsrc;
                b = kin ? srcB[q] + k0 : zsrc;
            } else {
                const int k = k0 + krow[q];
                const bool kin = k < K;
                a = (kin && okA[q]) ? srcA[q] + (int64_t)k * p.lda : zsrc;
                b = (kin && okB[q]) ? srcB[q] + (int64_t)k * p.ldb : zsrc;
            }
            if ((G_ABL & 4) && kt != kt_beg) return;
            __builtin_amdgcn_global_load_lds((g16_gbl_void*)a, (g16_lds_void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((g16_gbl_void*)b, (g16_lds_void*)(dst + G_OPB), 16, 0, 0);
        };
        auto stage = [&](int buf, int kt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) stage_q(buf, kt, q);
        };

        // ---- fragment addresses (bytes from the start of a buffer's A / B image)
        const int li = lane & 31, kg = lane >> 5;
        int fa, fb;
        if constexpr (KC) {
            const int bx = (kg ^ ((li >> 1) & 7)) * 16;
            fa = (wm * 128 + li) * 128 + bx;          // fragment (kk, i): (fa ^ (32 kk)) + i * 4096
            fb = (wn * 64 + li) * 128 + bx;
        } else {
            const int j4 = (lane & 15) >> 2, grp = (lane >> 4) & 1, mq = lane & 3;
            const int low = grp * 32 + (mq >> 1) * 16 + (mq & 1) * 8;
            fa = (8 * kg + j4) * 512 + wm * 256 + j4 * 64 + low;                         // (kk, i, h): (fa ^ (64 i)) + kk * 8192 + h * 2048
            fb = (8 * kg + j4) * 512 + (wn >> 1) * 256 + ((((wn & 1) << 1) ^ j4) * 64) + low;
        }
        auto frag = [&](const unsigned char* img, int base, int kk, int t) -> V8 {
            if constexpr (KC) {
                return *reinterpret_cast<const V8*>(img + (base ^ (32 * kk)) + t * 4096);
            } else {
                const unsigned char* q = img + (base ^ (64 * t)) + kk * 8192;
                const b16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((g16_lds_b16x4*)q);
                const b16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((g16_lds_b16x4*)(q + 2048));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo4), h2 = __builtin_bit_cast(u32x2, hi4);
                const u32x4 v = {l2[0], l2[1], h2[0], h2[1]};
                return __builtin_bit_cast(V8, v);
            }
        };
        // One K tile: 4 k-steps of 16.  The fragments of step kk + 1 are requested BEFORE the MFMAs of step
        // kk issue (two register sets), and one round of the next tile's LDS-DMA goes out per step: with
        // both waves of a SIMD leaving the tile's barrier together, a wave that waits for its own
        // ds_reads in front of every MFMA group leaves the matrix pipe idle half of the time (first
        // version of this kernel: 1.94 us per K tile against 1.0 of MFMA time).
        V8 af[2][TM], bf[2][TN];
        auto load_frags = [&](int buf, int kk, int set) {
            if ((G_ABL & 2) && !(buf == 0 && kk == 0 && set == 0)) return;
            const unsigned char* ia = gsm + buf * G_BUFB;
            const unsigned char* ib = ia + G_OPB;
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[set][j] = frag(ib, fb, kk, j);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[set][i] = frag(ia, fa, kk, i);
        };
        auto compute = [&](int buf, int kt_next, auto MORE) {      // MORE: compile-time (a branch around the
#pragma unroll                                                       // LDS-DMA makes every lgkmcnt wait a full drain)
            for (int kk = 0; kk < 4; ++kk) {
                // first half of the step's MFMAs, THEN the next step's fragment reads and a round of
                // LDS-DMA, then the second half: hipcc waits lgkmcnt(0) in front of the first MFMA that
                // reads a fragment (never a counted wait once an LDS-DMA sits in the block), so the
                // reads of step kk + 1 must not be in flight yet when step kk's first MFMA issues
                __builtin_amdgcn_s_setprio(1);
                static_for<TM * TN / 2>([&](auto IJ) {
                    constexpr int i = decltype(IJ)::value / TN, j = decltype(IJ)::value % TN;
                    const int st = (G_ABL & 2) ? 0 : (kk & 1);
                    const V8 fb_ = bf[st][j], fa_ = af[st][i];
                    if constexpr (G_ABL & 1) asm volatile("" :: "v"(fb_), "v"(fa_));
                    else acc[i][j] = HT::mfma(fb_, fa_, acc[i][j]);
                });
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < 3) load_frags(buf, kk + 1, (kk + 1) & 1);
                if constexpr (decltype(MORE)::value) stage_q(buf ^ 1, kt_next, kk);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                static_for<TM * TN / 2>([&](auto IJ) {
                    constexpr int i = TM / 2 + decltype(IJ)::value / TN, j = decltype(IJ)::value % TN;
                    const int st = (G_ABL & 2) ? 0 : (kk & 1);
                    const V8 fb_ = bf[st][j], fa_ = af[st][i];
                    if constexpr (G_ABL & 1) asm volatile("" :: "v"(fb_), "v"(fa_));
                    else acc[i][j] = HT::mfma(fb_, fa_, acc[i][j]);
                });
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // bias gradient: thread t sums columns 4 (t & 63) .. + 3 over the 8 k-rows 8 (t >> 6) .. + 7 of the tile
        auto colsum = [&](int buf) {
            const unsigned char* ia = gsm + buf * G_BUFB;
            const int m8 = (tid & 63) >> 1, half = (tid & 1) * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 8 * (tid >> 6) + u;
                const u32x2 v = *reinterpret_cast<const u32x2*>(ia + k * 512 + ((m8 ^ ((k & 3) << 2)) * 16) + half);
                csum[0] += HT::tofloat(v[0] & 0xffffu);
                csum[1] += HT::tofloat(v[0] >> 16);
                csum[2] += HT::tofloat(v[1] & 0xffffu);
                csum[3] += HT::tofloat(v[1] >> 16);
            }
        };

        stage(0, kt_beg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int kt = kt_beg; kt + 1 < kt_end; ++kt) {
            load_frags(buf, 0, 0);
            compute(buf, kt + 1, std::true_type{});           // the next tile's pieces land in the other buffer
            if (do_colsum) colsum(buf);
            if constexpr (!(G_ABL & 8)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt + 1 are in LDS
                __syncthreads();                                   // everybody's are, and buffer `buf` is free
            }
            buf ^= 1;
        }
        load_frags(buf, 0, 0);
        compute(buf, 0, std::false_type{});
        if (do_colsum) colsum(buf);
        __syncthreads();                                       // (the bias-gradient reduction reuses the LDS)
    }
    if constexpr (!KC) {
        if (do_colsum) {     // block-uniform: [8 k-parts][256 columns] partial sums -> one sum per column, fixed order
            float* part = reinterpret_cast<float*>(gsm);
            *reinterpret_cast<float4*>(part + (tid >> 6) * 256 + 4 * (tid & 63)) = make_float4(csum[0], csum[1], csum[2], csum[3]);
            __syncthreads();
            if (tid < 256 && m0 + tid < M) {
                float s = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) s += part[u * 256 + tid];
                const int m = m0 + tid;
                if (p.splits > 1)
                    p.splitk_ws[(int64_t)p.splits * M * N + (int64_t)blockIdx.y * M + m] = s;
                else
                    p.colsum_a[m] = p.accumulate ? p.colsum_a[m] + s : s;
            }
        }
    }
    h16_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------ host side

bool gemm_g16_applies(const GemmArgs& a)
{
    static const int off = [] { const char* e = getenv("SCTC_G16"); return (e && atoi(e) == 0) ? 1 : 0; }();
    if (off) return false;       // diagnostics: SCTC_G16=0 keeps round 2's register-staged kernel
    return a.in16 && (a.prec == 1 || a.prec == 2) && a.a_kcontig == a.b_kcontig && !a.idx_a && !a.idx_b &&
           a.lda % 8 == 0 && a.ldb % 8 == 0 && (a.a_kcontig ? a.K % 8 == 0 : (a.M % 8 == 0 && a.N % 8 == 0));
}

int launch_gemm_g16(const GemmArgs& a, hipStream_t stream)
{
    const int mt = (a.M + G_BM - 1) / G_BM, nt = (a.N + G_BN - 1) / G_BN;
    dim3 grid(mt * nt, a.splits), block(G_NT);
    const bool bf = a.prec == 2;
    void (*kern)(GemmArgs) = a.a_kcontig ? (bf ? gemm_g16_kernel<true, true> : gemm_g16_kernel<true, false>)
                                         : (bf ? gemm_g16_kernel<false, true> : gemm_g16_kernel<false, false>);
    const size_t smem = 2 * G_BUFB;
    {
        static std::mutex mu;
        static std::set<const void*> done;
        std::lock_guard<std::mutex> lock(mu);
        if (!done.count(reinterpret_cast<const void*>(kern))) {
            SCTC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            done.insert(reinterpret_cast<const void*>(kern));
        }
    }
    hipLaunchKernelGGL(kern, grid, block, smem, stream, a);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

}  // namespace sctc
