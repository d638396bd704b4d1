// CTC forward-backward on gfx950: the device counterpart of
// ctc_fast/ctc-loss/ctc_fast.pyx:13-152 (reference: Cython, CPU, float64).
//
// Three kernels per batch of utterances:
//   softmax_rows   (brnnet.py:161-168)  logits -> probs, one wave per frame
//   ctc_lattice    (ctc_fast.pyx:42-114) one wave64 per (utterance, direction):
//                  the scaled alpha (dir 0) or beta (dir 1) recursion, serial in t,
//                  parallel over the 2U+1 lattice states held K-per-lane in VGPRs;
//                  s-1/s-2 neighbours arrive by one DPP wave shift, the per-frame
//                  normaliser by a DPP reduction; the frame's probabilities are read
//                  coalesced (lane k <- y[t][k]), prefetched PF frames ahead, and
//                  gathered per state with ds_bpermute.  The scaled lattice row is
//                  written with one vector store per lane per frame.
//   ctc_grad       (ctc_fast.pyx:117-145) one wave per frame: alpha*beta products,
//                  per-label sums in ascending-state order (bit-reproducible), the
//                  y - g/(y*Z) formula.
//
// beta is computed as an alpha pass on the reversed problem: with s' = L-1-s and
// tau = T-1-t the beta recursion of ctc_fast.pyx:85-114 is the alpha recursion of
// :48-76 on the reversed label sequence (L = 2U+1 is odd, so blank/label parity is
// preserved), and the band [start,end) maps onto itself.  Its lattice is stored in
// s' order and read back reversed by ctc_grad.
//
// Scaling: the reference renormalises every frame by c_t = sum over the band
// (ctc_fast.pyx:70-76).  We multiply by r ~ 1/c at the rescale points (see the lattice
// kernel) and account llForward -= log(r) in float64, exact for whatever r was applied:
// r is v_rcp_f64 + two Newton steps (half the dependent operations of the IEEE division
// sequence, which sat on every frame's critical path), and the applied factors are not
// tracked in registers but stored into the one lattice column no state ever occupies
// (2U+1 is odd, the row stride even) and summed up after the last frame.
#include <type_traits>

#include "common.h"
#include "ctc_kernels.h"
#include "xlane.h"

namespace sctc {


template <typename R>
struct Vec;
template <>
struct Vec<float> {
    using v4 = float4;
};
template <>
struct Vec<double> {
    using v4 = double4;
};

// ---------------------------------------------------------------- softmax_rows

template <int NA>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, int64_t rows,
                                                           int A, int64_t ld)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld;
    float v[NA];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        int k = lane + 64 * q;
        v[q] = k < A ? xr[k] : -INFINITY;
        m = fmaxf(m, v[q]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        int k = lane + 64 * q;
        v[q] = k < A ? expf(v[q] - m) : 0.f;
        s += v[q];
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;  // cm.pow(rowVec,-1) then mult_by_row, brnnet.py:167-168
    float* yr = y + row * ld;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        int k = lane + 64 * q;
        if (k < A) yr[k] = v[q] * inv;
    }
}

int launch_softmax_rows(const float* x, float* y, int64_t rows, int A, int64_t ld,
                        hipStream_t stream)
{
    if (rows <= 0) return SCTC_OK;
    SCTC_CHECK_ARG(A >= 1, "softmax_rows: alphabet %d < 1", A);
    if (A > 256) return launch_softmax_rows_generic(x, y, rows, A, ld, stream);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (A <= 64)
        hipLaunchKernelGGL(softmax_rows_kernel<1>, grid, block, 0, stream, x, y, rows, A, ld);
    else if (A <= 128)
        hipLaunchKernelGGL(softmax_rows_kernel<2>, grid, block, 0, stream, x, y, rows, A, ld);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<4>, grid, block, 0, stream, x, y, rows, A, ld);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

// ---------------------------------------------------------------- argmax_rows

template <typename R>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const R* __restrict__ y,
                                                          int32_t* __restrict__ best,
                                                          int64_t rows, int A, int64_t ld)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const R* yr = y + row * ld;
    R bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < A; k += 64) {
        R v = yr[k];
        if (v > bv) { bv = v; bi = k; }  // first maximum wins within the lane (np.argmax)
    }
    for (int off = 32; off > 0; off >>= 1) {
        R ov = __shfl_xor(bv, off, 64);
        int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) best[row] = bi == 0x7fffffff ? 0 : bi;
}

int launch_argmax_rows(const void* y, int dtype, int32_t* best, int64_t rows, int A, int64_t ld,
                       hipStream_t stream)
{
    if (rows <= 0) return SCTC_OK;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == SCTC_F32)
        hipLaunchKernelGGL(argmax_rows_kernel<float>, grid, block, 0, stream, (const float*)y,
                           best, rows, A, ld);
    else
        hipLaunchKernelGGL(argmax_rows_kernel<double>, grid, block, 0, stream, (const double*)y,
                           best, rows, A, ld);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

// ---------------------------------------------------------------- ctc_lattice

__device__ __forceinline__ int64_t frame_row(const CtcUtt& u, const int32_t* rowbase, int t)
{
    return rowbase ? (int64_t)rowbase[t] + u.row0 : u.row0 + t;
}

// K states per lane (even), NA = ceil(A/64) probability registers per frame.
// RI: storage type of probs; the recursion itself runs in float64 (see ctc_kernels.h).
//
// Rescaling.  The reference divides every frame by its band sum c_t and accumulates
// llForward = sum_t log c_t (ctc_fast.pyx:70-76); that schedule (RS = 1) is the default for both
// probability types, so the lattices -- and with them the reference's underflow behaviour in
// the gradient -- are reproduced exactly.  LAZY (SCTC_CTC_LAZY=1, float32 probabilities only)
// rescales every 4th frame instead: the per-frame scale cancels in the gradient (:138-145),
// float32's smallest value 1e-45 keeps four unscaled float64 steps above 1e-180, llForward is
// the log of the last frame's mass minus the logs of the applied factors -- the same number --
// and the 64-lane reduction + division leave most steps of the dependency chain (0.44 instead
// of 0.7 ms at cfg-3).  It is numerically MORE robust than the reference (no underflow of the
// forward-backward overlap), which is exactly why it is not the default: parity first.
template <typename RI, int K, int NA, int W, bool LAZY>
__global__ __launch_bounds__(64 * W) void ctc_lattice_kernel(CtcLatticeArgs<RI> p)
{
    // W waves share one (utterance, pass): state s lives in global lane gl = s / K.  The two
    // cross-wave couplings of a frame go through LDS: the neighbour state of a wave's lane 0
    // (last state of the previous wave) and the band sum of a rescaling frame.
    // per frame parity and wave: {band-sum partial, last state's unnormalised value} -- one 16-byte slot,
    // written by lane 63 with one ds_write_b128
    __shared__ __attribute__((aligned(16))) double xw[2][W > 1 ? W : 1][2];
    double bprev = 0.0;   // W > 1: the previous wave's last state as of the last exchange (read right behind its barrier)
    using R = double;
    static_assert(K % 2 == 0 && K >= 2, "K must be even");
    constexpr int KH = K / 2;
    constexpr int RS = LAZY ? 4 : 1;
    // Frames of probabilities prefetched per block.  gfx950 counts loads and stores in ONE
    // in-order counter (vmcnt), so waiting for a block's prefetch also waits for the lattice
    // rows stored before it: one HBM write latency per block.  Long blocks amortise it.
    // (8 waves = two per SIMD = 256 registers per lane: 32 prefetched frames with their gathered
    // float64 probabilities no longer fit -- 38..170 spilled registers in the frame loop -- 16 do)
    // Wide alphabets in float64 (32 bytes per prefetched frame and lane) get shorter blocks for the same reason.
    constexpr int ROWB = NA * (int)sizeof(RI);
    constexpr int PF_K = K <= 4 ? 32 : (K <= 8 ? 16 : 8);
    constexpr int PF_A = ROWB >= 32 ? 8 : (ROWB >= 16 ? 16 : 32);
    constexpr int PF = W >= 8 ? (ROWB <= 4 ? 16 : 8) : (PF_K < PF_A ? PF_K : PF_A);
    const int b = blockIdx.x;
    const int dir = blockIdx.y;  // 0: alpha, 1: beta (== alpha of the reversed problem)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x;
    (void)wave;
    const CtcUtt u = p.utts[b];
    const int T = u.T, U = u.U, L = 2 * U + 1;
    const int LP = p.lp;  // lattice row stride (64*K)
    R* lat = (dir == 0 ? p.alpha : p.beta) + u.lat_off;
    const int32_t* seq = p.labels + u.lab_off;
    const int blank = p.blank;

    // ---- per-lane constants: labels of my odd states, skip-transition permission
    int lab[KH];
    bool allow[KH];
    bool valid_lab[KH];
#pragma unroll
    for (int jj = 0; jj < KH; ++jj) {
        const int idx = KH * gl + jj;  // label index of state K*gl + 2*jj + 1
        const bool ok = idx < U;
        const int i0 = ok ? (dir ? U - 1 - idx : idx) : 0;
        lab[jj] = seq[i0];
        int prev = blank;
        if (ok && idx >= 1) prev = seq[dir ? U - idx : idx - 1];
        // three-term transition only between different labels and not for s == 1
        // (ctc_fast.pyx:64-68 / :103-107)
        allow[jj] = ok && idx >= 1 && lab[jj] != prev;
        valid_lab[jj] = ok;
    }
    R allowf[KH];
#pragma unroll
    for (int jj = 0; jj < KH; ++jj) allowf[jj] = allow[jj] ? (R)1 : (R)0;
    // blank states K*gl + 2*jj exist while 2*(KH*gl+jj) <= 2U
    bool valid_blk[KH];
#pragma unroll
    for (int jj = 0; jj < KH; ++jj) valid_blk[jj] = (KH * gl + jj) <= U;

    const RI* probs = p.probs;
    const int64_t ld = p.ld;
    const int A = p.A;

    // a frame's probabilities stay in their storage type until they are gathered
    auto load_row = [&](int64_t row, RI (&dst)[NA]) {
        const RI* yr = probs + row * ld;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int k = lane + 64 * q;
            dst[q] = k < A ? yr[k] : (RI)0;
        }
    };
    auto load_frame = [&](int tau, RI (&dst)[NA]) {
        const int t = dir ? T - 1 - tau : tau;
        load_row(frame_row(u, p.rowbase, t), dst);
    };
    // Row indices of a whole block of frames with ONE vector load (lane i <- frame tau0 + i, clamped),
    // fetched a block before they are needed.  Looking rowbase[t] up frame by frame put a dependent
    // load and an s_waitcnt vmcnt(0) -- which also waits for every lattice-row store in flight -- in
    // front of each prefetched frame: 1000 of the 1600 shader cycles a frame cost (s_memtime stamps).
    auto block_rows = [&](int tau0) -> int {
        const int tau = min(tau0 + lane, T - 1);
        const int t = dir ? T - 1 - tau : tau;
        return p.rowbase ? p.rowbase[t] : t;
    };
    auto gather = [&](const RI (&y)[NA], int k) -> R {
        RI out = lane_gather(y[0], k & 63);
#pragma unroll
        for (int q = 1; q < NA; ++q) {
            RI o = lane_gather(y[q], k & 63);
            if ((k >> 6) == q) out = o;
        }
        return (R)out;   // probs.astype(np.float64), brnnet.py:175
    };
    auto bcast = [&](const RI (&y)[NA], int k) -> R {
        RI out = lane_bcast(y[0], k & 63);
#pragma unroll
        for (int q = 1; q < NA; ++q) {
            RI o = lane_bcast(y[q], k & 63);
            if ((k >> 6) == q) out = o;
        }
        return (R)out;
    };
    // A row is stored together with the factor it was scaled with: L = 2U+1 is odd and the row
    // stride LP even, so column LP-1 (last register of the last lane, always a zero state) is free.
    // Keeping the factors in registers instead -- one per lane, folded 64 at a time -- cost a lane
    // compare, a 64-bit select and a branch per frame: 166 of a frame's ~600 cycles (s_memtime).
    const bool last_lane = gl == 64 * W - 1;
    auto store_row = [&](int tau, const R (&v)[K], R factor) {
#ifdef SCTC_CTC_NOSTORE
        if (tau != 0x7fffffff) return;
#endif
        R* row = lat + (int64_t)tau * LP + (int64_t)K * gl;
        const R vlast = last_lane ? factor : v[K - 1];
        if constexpr (K % 4 == 0) {
            typename Vec<R>::v4* dst = reinterpret_cast<typename Vec<R>::v4*>(row);
#pragma unroll
            for (int j = 0; j < K; j += 4)
                dst[j / 4] = {v[j], v[j + 1], v[j + 2], j + 3 == K - 1 ? vlast : v[j + 3]};
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) row[j] = j == K - 1 ? vlast : v[j];
        }
    };
    // 1/c for the row scaling: hardware estimate + two Newton-Raphson steps (the reciprocal part of
    // the IEEE division sequence without its scaling / fix-up tail; <= 1 ulp).  Which double is
    // applied does not matter for parity: llForward takes the log of exactly this value, and the
    // per-frame factors cancel in the gradient (ctc_fast.pyx:138-145).
    auto recip = [&](R c) -> R {
        R x = __builtin_amdgcn_rcp(c);
        R e = fma(-c, x, (R)1);
        x = fma(x, e, x);
        e = fma(-c, x, (R)1);
        return fma(x, e, x);
    };

    // workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the
    // lattice-row stores to HBM (vmcnt) in every frame
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // W > 1: ONE workgroup barrier per frame.  Before it every wave publishes its band-sum partial and
    // its last state's UNNORMALISED value n[K-1]; after it every wave knows the frame's c (and r = 1/c),
    // and the next frame's neighbour state of lane 0 is bnd * r -- the very multiplication the owning
    // wave performs, so the lattice is bit-identical to the two-barrier schedule this replaces (publish
    // the normalised boundary after the sum, meet again at the start of the next frame: 1.08 us per
    // frame at T = 8000 / U = 800).  Slots alternate by frame parity: a wave that races ahead writes the
    // other slot, and cannot reach this one again before everybody has passed the next barrier.
    // sum over all states of the block, identical in every lane (fixed order)
    auto block_sum = [&](R loc, R last_unnorm, int tau) -> R {
        const R ws = wave_sum(loc);
        if constexpr (W == 1) {
            return ws;
        } else {
            if (lane == 63) {
                *reinterpret_cast<double2*>(&xw[tau & 1][wave][0]) = make_double2(ws, last_unnorm);
            }
            lds_barrier();
            // every read behind the barrier is issued at once: the W partials and the boundary state the
            // NEXT frame needs (it used to be a dependent LDS read at the head of that frame); the partials
            // are summed as a tree in a fixed order
            R pw[W];
#pragma unroll
            for (int w = 0; w < W; ++w) pw[w] = xw[tau & 1][w][0];
            bprev = xw[tau & 1][wave > 0 ? wave - 1 : 0][1];
#pragma unroll
            for (int st = 1; st < W; st *= 2)
#pragma unroll
                for (int w = 0; w + st < W; w += 2 * st) pw[w] += pw[w + st];
            return pw[0];
        }
    };
    // frames without a rescale (lazy schedule): the boundary state alone
    auto publish = [&](R last, int tau) {
        if constexpr (W > 1) {
            if (lane == 63) xw[tau & 1][wave][1] = last;
            lds_barrier();
            bprev = xw[tau & 1][wave > 0 ? wave - 1 : 0][1];
        }
    };
    // last state of global lane gl - 1 as of the previous frame (all threads call, start of a frame);
    // rprev = the factor the previous frame's row was scaled with (1 if it was not)
    auto shift_in = [&](const R (&v)[K], int tau, R rprev) -> R {
        R prev = lane_shr1(v[K - 1]);
        if constexpr (W > 1) {
            (void)tau;
            const R across = bprev * rprev;          // the very product the owning wave stored
            prev = (lane == 0 && wave > 0) ? across : prev;
        }
        return prev;
    };

    // sum of a lane's K new states as a balanced tree (log2 K dependent additions instead of K-1)
    auto local_sum = [&](const R (&n)[K]) -> R {
        R t[K];
#pragma unroll
        for (int j = 0; j < K; ++j) t[j] = n[j];
#pragma unroll
        for (int w = 1; w < K; w *= 2)
#pragma unroll
            for (int j = 0; j + w < K; j += 2 * w) t[j] += t[j + w];
        return t[0];
    };

    R a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = (R)0;

    // A zero band sum (the reference's ZeroDivisionError, ctc_fast.pyx:75) is not branched on frame by
    // frame: the frame index is latched in a scalar, the recursion runs on (through NaNs, whose rows
    // nobody reads) and the block of frames ends the pass.  A compare-and-branch per frame put a
    // VALU -> SALU -> branch round trip on the recursion's dependency chain and cut the unrolled block
    // into one basic block per frame.
    constexpr int NO_BAD = 0x7fffffff;
    int first_bad = NO_BAD;
    int skip = 0;
    int n_rows = T;    // rows stored with their factor: all of them, or those before the frame that skipped
    R rprev = (R)1;    // factor applied to the previous frame's row (uniform)
    R cfac = (R)1;     // the band sum that factor is the reciprocal of (1 where the row was not rescaled): stored beside
                       // the row; llForward takes the logarithm of the c_t themselves like the reference (a cost of 1e-8
                       // -- one frame, c = 1 - 1e-8 -- would otherwise carry the reciprocal's rounding, 2e-8 relative)
    // T < U: the band [start,end) is empty at every frame t >= 1 (L >= 2T+2 does not depend
    // on t): the reference divides nothing, takes log(0) = -inf and returns cost +inf with
    // skip False (ctc_fast.pyx:70-76 on an empty range)
    const bool empty_band = (L >= 2 * T + 2) && T > 1;

    // ---- tau = 0: ctc_fast.pyx:42-47 / :79-84
    RI ycur[PF][NA];
    {
        RI y0[NA];
        load_frame(0, y0);
        const R yb = bcast(y0, blank);
        const R yl = gather(y0, lab[0]);
        if (gl == 0) { a[0] = yb; a[1] = yl; }
        const R c = block_sum(a[0] + a[1], a[K - 1], 0);
        if (c == (R)0) {
            skip = 1;  // ZeroDivisionError at :45
            n_rows = 0;
        } else {
            const R r = recip(c);
            a[0] *= r;
            a[1] *= r;
            rprev = r;
            cfac = c;
        }
        store_row(0, a, cfac);
    }

    if (!skip && empty_band) {
        R z[K];
#pragma unroll
        for (int j = 0; j < K; ++j) z[j] = (R)0;
        for (int tau = 1; tau < T; ++tau) store_row(tau, z, (R)1);
    } else if (!skip && T > 1) {
        static_assert(PF <= 64, "one lane per frame of a block");
        int rbv = block_rows(1);
#pragma unroll
        for (int i = 0; i < PF; ++i)                  // unconditional (clamped): a load behind a
            load_row((int64_t)__builtin_amdgcn_readlane(rbv, i) + u.row0, ycur[i]);   // branch is waited for at once
        rbv = block_rows(1 + PF);
        for (int tb = 1; tb < T && !skip; tb += PF) {
            RI ynxt[PF][NA];
#pragma unroll
            for (int i = 0; i < PF; ++i) load_row((int64_t)__builtin_amdgcn_readlane(rbv, i) + u.row0, ynxt[i]);
            rbv = block_rows(tb + 2 * PF);            // for the next iteration, a block ahead
            // For short label rows the per-state probabilities of the whole block are gathered
            // (ds_bpermute / readlane) before the serial part starts, so that the LDS-crossbar
            // latency is off the recursion's dependency chain.
            constexpr bool PREG = K <= 8;
            R ybv[PREG ? PF : 1], ylv[PREG ? PF : 1][KH];
            if constexpr (PREG) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    ybv[i] = bcast(ycur[i], blank);
#pragma unroll
                    for (int jj = 0; jj < KH; ++jj) {
                        // every lane takes part in the gather (cross-lane op); states beyond the
                        // label row then multiply by 0
                        const R gl = gather(ycur[i], lab[jj]);
                        ylv[i][jj] = valid_lab[jj] ? gl : (R)0;
                    }
                }
            }
            // Frames whose band starts at state 0 (L <= 2(T - tau): all but the last ~U frames)
            // need no band test, and the states beyond the label row stay exactly zero because
            // their label probability was zeroed at gather time (a blank state beyond the row
            // only ever adds zeros).  For a whole block of such frames the step is 2 DPP moves,
            // 5 float64 operations per state pair and the row store -- no compares or selects on
            // the recursion's dependency chain.  `in + allow*below` as one fma rounds like the add.
            const bool fast = PREG && (tb + PF - 1 < T) && (L <= 2 * (T - (tb + PF - 1)));
            if (fast) {
                if constexpr (PREG) {
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const int tau = tb + i;
                        const R yb = ybv[i];
                        const R prev_last = shift_in(a, tau, rprev);
                        R n[K];
#pragma unroll
                        for (int jj = 0; jj < KH; ++jj) {
                            const R below = jj == 0 ? prev_last : a[2 * jj - 1];
                            n[2 * jj] = (a[2 * jj] + below) * yb;
                            n[2 * jj + 1] = fma(below, allowf[jj], a[2 * jj + 1] + a[2 * jj]) * ylv[i][jj];
                        }
                        if ((tau % RS) == 0) {
                            const R c = block_sum(local_sum(n), n[K - 1], tau);
                            first_bad = (c == (R)0 && first_bad == NO_BAD) ? tau : first_bad;
                            const R r = recip(c);
#pragma unroll
                            for (int j = 0; j < K; ++j) a[j] = n[j] * r;
                            rprev = r;
                            cfac = c;
                        } else {
                            publish(n[K - 1], tau);
                            rprev = (R)1;
                            cfac = (R)1;
#pragma unroll
                            for (int j = 0; j < K; ++j) a[j] = n[j];
                        }
                        store_row(tau, a, cfac);
                    }
                }
            } else {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int tau = tb + i;
                if (tau < T) {
                    // lower band limit, ctc_fast.pyx:49-53 (identical for the reversed problem);
                    // states >= end are exactly zero by construction
                    const int rem = 2 * (T - tau);
                    const int start = L <= rem ? 0 : L - rem;
                    R yb;
                    if constexpr (PREG) yb = ybv[i]; else yb = bcast(ycur[i], blank);
                    const R prev_last = shift_in(a, tau, rprev);  // state K*gl - 1
                    R n[K];
#pragma unroll
                    for (int jj = 0; jj < KH; ++jj) {
                        // blank state s = K*gl + 2*jj  (:58-62)
                        const R below = jj == 0 ? prev_last : a[2 * jj - 1];
                        const int sb = K * gl + 2 * jj;
                        const R vb = (a[2 * jj] + below) * yb;
                        n[2 * jj] = (valid_blk[jj] && sb >= start) ? vb : (R)0;
                        // label state s = K*gl + 2*jj + 1  (:63-68)
                        R yl;
                        if constexpr (PREG) yl = ylv[i][jj]; else yl = gather(ycur[i], lab[jj]);
                        R in = a[2 * jj + 1] + a[2 * jj];
                        if (allow[jj]) in += below;
                        const R vl = in * yl;
                        n[2 * jj + 1] = (valid_lab[jj] && sb + 1 >= start) ? vl : (R)0;
                    }
                    if ((tau % RS) == 0 || tau == T - 1) {
                        // rescale (every frame when RS == 1, like :70-76); the last frame always,
                        // so that llForward is complete
                        const R c = block_sum(local_sum(n), n[K - 1], tau);
                        // c == 0: ZeroDivisionError at :75 (the band is non-empty here)
                        first_bad = (c == (R)0 && first_bad == NO_BAD) ? tau : first_bad;
                        const R r = recip(c);
#pragma unroll
                        for (int j = 0; j < K; ++j) a[j] = n[j] * r;
                        rprev = r;
                        cfac = c;
                    } else {
                        publish(n[K - 1], tau);
                        rprev = (R)1;
                        cfac = (R)1;
#pragma unroll
                        for (int j = 0; j < K; ++j) a[j] = n[j];
                    }
                    store_row(tau, a, cfac);
                }
            }
            }
            if (first_bad != NO_BAD) {
                skip = 1;
                n_rows = first_bad;
            }
#pragma unroll
            for (int i = 0; i < PF; ++i)
#pragma unroll
                for (int q = 0; q < NA; ++q) ycur[i][q] = ynxt[i][q];
        }
    }
    // llForward = sum_t log c_t (ctc_fast.pyx:47,76) = sum over the logarithms of the stored band sums.  The wave that
    // stored them reads them back (its own stores, drained first; the lines were never cached):
    // lane l takes frames l, l+64, ... in ascending order -- the order the in-register scheme this
    // replaces summed them in, so llForward is bit-identical to it.
    if (wave == W - 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        double ll = 0.0;   // per-lane partial of llForward (float64 like the reference)
        const R* fac = lat + (LP - 1);
        for (int tau = lane; tau < n_rows; tau += 64) ll += log((double)fac[(int64_t)tau * LP]);
        double total = wave_sum(ll);
        if (lane == 0) {
            if (empty_band && !skip) total = -INFINITY;  // math.log(0.0)
            p.ll[2 * b + dir] = total;                   // llForward (dir 0) / llBackward (dir 1)
            p.skip2[2 * b + dir] = skip;
        }
    }
}

// ---------------------------------------------------------------- ctc_grad

// NB: 64-state slices of the two lattice rows fetched per trip (launcher: by the row stride; its own instantiation per
// value -- sixteen slices of two float64 lattices are 128 registers, which the short-row shapes, eight blocks to a CU, do
// not have)
template <typename RI, int NB>
__global__ __launch_bounds__(256) void ctc_grad_kernel(CtcGradArgs<RI> p)
{
    using R = double;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    const CtcUtt u = p.utts[b];
    const int T = u.T, U = u.U, L = 2 * U + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int LP = p.lp;
    const int t0 = blockIdx.x * 4;
    if (t0 >= T) return;
    int32_t* lab_s = reinterpret_cast<int32_t*>(smem);                    // [LP]
    R* ab_s = reinterpret_cast<R*>(smem + (size_t)LP * sizeof(int32_t));   // [4][LP]

    const int32_t* seq = p.labels + u.lab_off;
    const int L4 = (L + 3) & ~3;  // vector reads below run to L4; pad label = -1 never matches
    for (int s = threadIdx.x; s < L4; s += 256)
        lab_s[s] = s >= L ? -1 : ((s & 1) ? seq[(s - 1) >> 1] : p.blank);
    // [LP/2] odd states grouped by label, [A+1] group offsets (behind the four waves' copies of the frame's probabilities)
    int32_t* ord_s = reinterpret_cast<int32_t*>(smem + (size_t)LP * sizeof(int32_t) + 4 * (size_t)LP * sizeof(R) +
                                                4 * (size_t)((p.A + 3) & ~3) * sizeof(RI));
    int32_t* start_s = ord_s + LP / 2;
    for (int j = threadIdx.x; j < U; j += 256) ord_s[j] = p.by_label[u.lab_off + j];
    for (int k = threadIdx.x; k <= p.A; k += 256) start_s[k] = p.label_start[(int64_t)b * (p.A + 1) + k];
    __syncthreads();

    const int skip = p.skip2[2 * b] | p.skip2[2 * b + 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.cost[b] = -p.ll[2 * b];  // -llForward, ctc_fast.pyx:149,152
        p.skip[b] = skip;
    }
    const int t = t0 + wave;
    if (t >= T) return;
    const int64_t row = frame_row(u, p.rowbase, t);
    const RI* yr = p.probs + row * p.ld;
    RI* gr = p.grad + row * p.ld;
    if (skip) {
        // the reference returns its zero-initialised grad (ctc_fast.pyx:31-32,149)
        for (int k = lane; k < p.A; k += 64) gr[k] = (RI)0;
        return;
    }
    const R* al = p.alpha + u.lat_off + (int64_t)t * LP;
    const R* be = p.beta + u.lat_off + (int64_t)(T - 1 - t) * LP;  // stored reversed in t and s
    R* ab = ab_s + (size_t)wave * LP;
    // Lazy schedule only (SCTC_CTC_LAZY=1): alpha and beta rows carry the lattice kernel's
    // every-4th-frame scaling (a row can sit at 1e-180 for extremely peaked inputs), so both are
    // brought to unit magnitude by an exact power of two before they are multiplied; the common
    // factor cancels in g / (y * Z) (ctc_fast.pyx:141-145).  The default schedule normalises every
    // frame like the reference and keeps its exact operation order -- including its underflow:
    // where forward and backward normalisers have drifted apart by more than 1e308 the reference
    // gets Z == 0 and returns grad = y for that frame, and so does this kernel.
    int ea = 0, eb = 0;
    if (p.lazy) {
        R ma = (R)0, mb = (R)0;
        for (int s = lane; s < L; s += 64) {
            ma = fmax(ma, al[s]);
            mb = fmax(mb, be[s]);
        }
        ma = wave_max(ma);
        mb = wave_max(mb);
        ea = ma > (R)0 ? -ilogb(ma) : 0;
        eb = mb > (R)0 ? -ilogb(mb) : 0;
    }
    // The frame's probabilities go to LDS once (one copy per wave) and the lattice rows are fetched
    // eight 64-state slices at a time (round 4).  The loop used to handle one slice per trip -- two
    // lattice loads, then a dependent global gather of y[label] behind them: two memory round trips per
    // slice, 26 slices per frame at cfg-5 (2U + 1 = 1601): 65 us per frame, 2.17 of the 6.9 ms of
    // the CTC phase at minibatch 8.  Same operations in the same order per state and per lane.
    RI* y_s = reinterpret_cast<RI*>(smem + (size_t)LP * sizeof(int32_t) + 4 * (size_t)LP * sizeof(R)) +
              (size_t)wave * ((p.A + 3) & ~3);
    for (int k = lane; k < p.A; k += 64) y_s[k] = yr[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // Round 6: written for latency.  A frame used to cost its wave ~20 us at L = 1601 -- four dependent trips of
    // eight lattice loads each (a memory round trip per trip) and then the blank lane's chain of (L + 1) / 2 dependent
    // additions while 63 lanes watched (0.63 ms per cfg-5 step of 8 utterances, 1.3 TB/s).  Now sixteen slices per trip
    // (two trips at most), loads unconditional (clamped, masked by a select: a load under a lane mask is waited for at
    // once), and the blank sum in parallel: lane l holds the states l, l + 64, ... -- an even lane only blank states, an
    // odd lane only label states -- so the even lanes sum their products as they form them and ONE wave reduction makes the
    // blank sum; labels equal to the blank id (legal) come from the blank's list like any label's.
    R zpart = (R)0, epart = (R)0;
    const int L32 = (L + 31) & ~31;   // <= LP; the pads ab[L .. L32) are written as +0 (read by the sums below)
    const bool blank_lane = (lane & 1) == 0;
    for (int s0 = lane; s0 < L32; s0 += 64 * NB) {
        R av[NB], bv[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int s = min(s0 + 64 * i, L - 1);
            av[i] = al[s];
            bv[i] = be[L - 1 - s];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int s = s0 + 64 * i;
            if (s < L32) {
                R v = (R)0;
                if (s < L) {
                    if (p.lazy) v = scalbn(av[i], ea) * scalbn(bv[i], eb);
                    else v = av[i] * bv[i];                    // :119
                    ab[s] = v;
                    epart += blank_lane ? v : (R)0;            // :122-124
                    if (v != (R)0) v = v / (R)y_s[lab_s[s]];   // :125-126 / :130-131
                } else {
                    ab[s] = (R)0;
                }
                zpart += v;
            }
        }
    }
    const R Z = wave_sum(zpart);                        // absum[t], :133-136
    const R gb = wave_sum(epart);
    // LDS writes above are read by other lanes of this wave only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // grad[k, t] = sum of ab over the states of label k (:120-131): every lane walks its label's host-built list of
    // states (by_label), eight entries per trip (entries past the end read the zero pad ab[L]); the blank adds the blank sum
    for (int k = lane; k < p.A; k += 64) {
        R g = (R)0;
        const int j0 = start_s[k], j1 = start_s[k + 1];
        for (int j = j0; j < j1; j += 8) {
            int idx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) idx[i] = ord_s[min(j + i, j1 - 1)];
            R a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = ab[j + i < j1 ? idx[i] : L];
#pragma unroll
            for (int i = 0; i < 8; ++i) g += a[i];
        }
        if (k == p.blank) g += gb;
        const R y = (R)y_s[k];
        const R tmp = y * Z;                        // :141
        gr[k] = (RI)(tmp > (R)0 ? y - g / tmp : y); // :142-145 (cast: CUDAMatrix(deltas), brnnet.py:188)
    }
}

// ---------------------------------------------------------------- launchers

template <typename RI, int K, int W>
static int launch_lattice_k(const CtcLatticeArgs<RI>& a, int B, int NA, int lazy, hipStream_t stream)
{
    dim3 grid(B, 2), block(64 * W);
    if constexpr (sizeof(RI) == 4) {
        if (lazy) {
            if (NA == 1) hipLaunchKernelGGL((ctc_lattice_kernel<RI, K, 1, W, true>), grid, block, 0, stream, a);
            else if (NA == 2) hipLaunchKernelGGL((ctc_lattice_kernel<RI, K, 2, W, true>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((ctc_lattice_kernel<RI, K, 4, W, true>), grid, block, 0, stream, a);
            SCTC_HIP_TRY(hipGetLastError());
            return SCTC_OK;
        }
    }
    if (NA == 1) hipLaunchKernelGGL((ctc_lattice_kernel<RI, K, 1, W, false>), grid, block, 0, stream, a);
    else if (NA == 2) hipLaunchKernelGGL((ctc_lattice_kernel<RI, K, 2, W, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((ctc_lattice_kernel<RI, K, 4, W, false>), grid, block, 0, stream, a);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

// Lattice shape for label rows of up to max_L = 2U+1 states: K states per lane, W waves per
// (utterance, pass).  Up to 256 states one wave holds everything (no cross-wave traffic); longer
// rows are spread over 4 waves so that a frame costs K/4 as many dependent float64 operations, and
// rows of more than 1024 states over 8 waves (two per SIMD): at T = 8000 / U = 800 a frame costs 0.65
// instead of 0.73 us with 4 states per lane instead of 8 (round 3; 16 waves x 2 states: 1.2 us --
// four waves per SIMD meet at the frame's barrier; 8 waves x 2 states for 513..1024 states: 2-4 %
// slower than 4 waves x 4).  Costs and gradients agree to all printed digits
// between the shapes (tests/gpu_ctc_shape.py); the band sum is taken in another order, so the last
// bits of a lattice may differ from one shape to the other.
int ctc_lattice_shape(int max_L, int* waves)
{
    const char* force = getenv("SCTC_CTC_WAVES");   // diagnostics: 1 / 4 / 8
    const int fw = force ? atoi(force) : 0;
    if ((fw == 0 && max_L <= 256) || fw == 1) {
        *waves = 1;
        for (int k = 2; k <= 32; k *= 2)
            if (max_L <= 64 * k) return k;
        return 0;
    }
    if (fw == 8 || (fw == 0 && max_L > 1024)) {
        *waves = 8;
        return max_L <= 1024 ? 2 : (max_L <= 2048 ? 4 : 0);
    }
    *waves = 4;
    for (int k = 2; k <= 8; k *= 2)
        if (max_L <= 256 * k) return k;
    return 0;
}

template <typename RI>
int launch_ctc_lattice(const CtcLatticeArgs<RI>& a, int B, int K, int W, int lazy, hipStream_t stream)
{
    using R = RI;
    const int NA = a.A <= 64 ? 1 : (a.A <= 128 ? 2 : 4);
    if (W == 8) {
        switch (K) {
            case 2: return launch_lattice_k<R, 2, 8>(a, B, NA, lazy, stream);
            case 4: return launch_lattice_k<R, 4, 8>(a, B, NA, lazy, stream);
        }
    }
    if (W == 4) {
        switch (K) {
            case 2: return launch_lattice_k<R, 2, 4>(a, B, NA, lazy, stream);
            case 4: return launch_lattice_k<R, 4, 4>(a, B, NA, lazy, stream);
            case 8: return launch_lattice_k<R, 8, 4>(a, B, NA, lazy, stream);
        }
    } else {
        switch (K) {
            case 2: return launch_lattice_k<R, 2, 1>(a, B, NA, lazy, stream);
            case 4: return launch_lattice_k<R, 4, 1>(a, B, NA, lazy, stream);
            case 8: return launch_lattice_k<R, 8, 1>(a, B, NA, lazy, stream);
            case 16: return launch_lattice_k<R, 16, 1>(a, B, NA, lazy, stream);
            case 32: return launch_lattice_k<R, 32, 1>(a, B, NA, lazy, stream);
        }
    }
    return set_error(SCTC_ERR_ARG, "ctc: label sequence too long (K=%d, W=%d)", K, W);
}

template <typename R, int NB>
static int launch_ctc_grad_nb(const CtcGradArgs<R>& a, int B, int max_T, hipStream_t stream)
{
    dim3 grid((max_T + 3) / 4, B), block(256);
    size_t smem = (size_t)a.lp * sizeof(int32_t) + 4 * (size_t)a.lp * sizeof(double) + 4 * (size_t)((a.A + 3) & ~3) * sizeof(R) +
                  ((size_t)a.lp / 2 + a.A + 1) * sizeof(int32_t);
    if (smem > 48 * 1024)
        SCTC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_grad_kernel<R, NB>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((ctc_grad_kernel<R, NB>), grid, block, smem, stream, a);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

template <typename R>
int launch_ctc_grad(const CtcGradArgs<R>& a, int B, int max_T, hipStream_t stream)
{
    // (short rows: four slices per trip -- the clamped loads of slices beyond the row are not free: 4096 utterances of
    // 201 states took 7.3 instead of 5.0 ms with sixteen)
    if (a.lp <= 256) return launch_ctc_grad_nb<R, 4>(a, B, max_T, stream);
    if (a.lp <= 512) return launch_ctc_grad_nb<R, 8>(a, B, max_T, stream);
    return launch_ctc_grad_nb<R, 16>(a, B, max_T, stream);
}

template int launch_ctc_lattice<float>(const CtcLatticeArgs<float>&, int, int, int, int, hipStream_t);
template int launch_ctc_lattice<double>(const CtcLatticeArgs<double>&, int, int, int, int, hipStream_t);
template int launch_ctc_grad<float>(const CtcGradArgs<float>&, int, int, hipStream_t);
template int launch_ctc_grad<double>(const CtcGradArgs<double>&, int, int, hipStream_t);

}  // namespace sctc
