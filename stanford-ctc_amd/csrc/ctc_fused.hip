// CTC forward-backward with the gradient formed INSIDE the two recursions ("meet in the middle"):
// the device counterpart of ctc_fast/ctc-loss/ctc_fast.pyx:13-152 for label rows of up to 512
// lattice states (2U+1 <= 512: every shape of BASELINE configs[0..3]; 2, 4 or 8 states per lane).
//
// One workgroup = one utterance: wave 0 runs the scaled alpha recursion (ctc_fast.pyx:42-76), wave 1
// the beta recursion (:79-114) as an alpha pass on the reversed problem (ctc_kernels.hip, header);
// with helper waves (HELP, batches of up to 256 utterances) waves 2..5 form the gradient of phase 1
// from the rows the two hand over through an LDS ring.  With Ta = T/2:
//
//   phase 0   alpha walks t = 0 .. Ta-1, beta walks t = T-1 .. Ta; each STORES its normalised rows
//   barrier   (the workgroup's own stores, same CU / same L2)
//   phase 1   alpha walks t = Ta .. T-1 and at every frame multiplies its row -- still in
//             registers -- with the beta row of the same frame stored in phase 0; beta walks
//             t = Ta-1 .. 0 against the stored alpha rows.  Each wave forms ab = alpha*beta
//             (:117-119), the per-label sums (:120-131), absum (:133-136) and the frame's
//             gradient row (:138-145) and writes it: every frame's gradient is produced exactly
//             once, from the recursion that reaches it second (by that wave itself, or by its two
//             helper waves: see HELP below).
//
// Against ctc_lattice_kernel + ctc_grad_kernel (both lattices stored in float64 over a 64K-wide
// row, both read back by a third kernel: 4 lattice passes over HBM) a lattice element is stored
// at most once and read at most once, rows are packed to round_up(2U+1, K) states, and the read
// happens in the workgroup that wrote it (the row is in that XCD's L2 unless the batch is so
// large that it has been evicted).  Rounds 1-4 stored 2 x 8 x 256 bytes per frame and direction
// at cfg-3; this kernel stores 4 x 204 per frame and ONE direction.
//
// Storage type ST of the kept rows:
//   double    float64 probabilities (the ctc_fast.ctc_loss host signature): nothing is rounded;
//   uint32_t  float32 probabilities (the BRNN path): the row is normalised, every value lies in
//             [0, 1] (+ an ulp), so the sign bit and the top exponent bit of the float64 pattern
//             are always zero: bits 61..30 are kept, rounded to nearest -- 10 exponent bits (the
//             FULL float64 range below 2.0) and 22 mantissa bits (float32 has 23).  A plain
//             float32 copy would NOT do: at T >> 2U the states where alpha and beta overlap carry
//             1e-40 .. 1e-300 of a frame's mass (random inputs, T = 1000, U = 100: gradient error
//             1.0 with float32 rows, 4e-8 with this format -- below the quantum of the float32
//             gradient it is written to; tests/test_ctc_store_model.py restates the experiment).
//             SCTC_CTC_STORE=64 keeps float64 rows on the float32 path too (A/B, tests).
// The recursion itself, the normalisers and llForward stay float64 like the reference.
//
// Summation order.  absum[t] divides every state's product by ITS label's probability before it
// sums (:125-131): that order is kept (the per-label form sum_k g_k / y_k is the same number until
// the products are denormal -- then it is off by 1e-3, and the reference's value is what
// counts).  The division is a multiplication with the reciprocal (float32 probabilities: the
// hardware estimate + two Newton steps, <= 1 ulp; float64 probabilities, which may be denormal: the
// IEEE division), formed once per symbol and frame, off the recursion's dependency chain.  The sums themselves
// are taken as fixed trees (bit-reproducible run to run; the reference adds in ascending state
// order: differences are in the last bits, the float64 golden vectors hold at 1e-11 / 1e-9).
#include <mutex>
#include <type_traits>

// frames per block of the two-wave form (4: 190 registers, two waves per SIMD; 8: 296, one)
#ifndef SCTC_FUSED_PF2
#define SCTC_FUSED_PF2 4
#endif

#include "common.h"
#include "ctc_kernels.h"
#include "ctc_store.h"
#include "xlane.h"

namespace sctc {

// The two-wave form on float32 probabilities with 32-bit rows, up to 4 states per lane, up to 64 symbols (the
// saturating-batch form of BASELINE configs[0..2]) goes on a REGISTER DIET (round 6): SQ counters showed
// the SIMDs 0.64 VALU-busy with two waves each (194 registers), a third wave needs <= 168.  Three things: list entries
// 8..15 of a label are not kept in registers (NI2 = NI: the slow path beyond 8 reads them from LDS), the label products
// of a finished block are not fetched ahead of the transposed sums (EARLY: 64 registers at the peak), and the
// compiler is told the occupancy to aim for (amdgpu_waves_per_eu; 4 spill slots outside the frame loop at 4 states
// per lane, none at 2 -- 116 registers: four waves).
// Measured (tools/ab_ctc_diet.sh, both libraries in alternation on one box, events around the C entry): 4096 utterances of
// the cfg-3 shape 4.43-4.52 ms against 4.72-4.86 (-6 %), 1024 utterances 1.27 against 1.23 (+3 %: two waves per SIMD
// either way, and the dieted code was a little slower) -- so it is its own instantiation.  From 1025 utterances on: two waves
// per utterance and two / three waves per SIMD mean 1024 / 1536 utterances resident at a time -- 1536 utterances are one round
// instead of two (1.54 against 1.92 ms), 4096 three instead of four (4.12 / 4.53); up to 1024 the two forms are level since the
// later trims (1.100 / 1.107 ms; tools/ctc_form_sweep.py).
template <typename RI, typename ST, int K, int NA, bool HELP>
struct FusedDiet {
#ifdef SCTC_FUSED_NO_DIET       // A/B build (tools/build_variant.sh nodiet ctc_fused.hip -DSCTC_FUSED_NO_DIET)
    static constexpr bool possible = false;
#else
    static constexpr bool possible = !HELP && NA == 1 && K <= 4 && sizeof(RI) == 4 && sizeof(ST) == 4;
#endif
};
static constexpr int FUSED_DIET_MIN_B = 1025;

template <typename RI, typename ST, int K, int NA, bool HELP, bool DIET = false>
__global__ __launch_bounds__(HELP ? 384 : 128) __attribute__((amdgpu_waves_per_eu(DIET ? 3 : 1, 8)))
void ctc_fused_kernel(CtcFusedArgs<RI> p)
{
    static_assert(!DIET || FusedDiet<RI, ST, K, NA, HELP>::possible, "the dieted form exists for the two-wave float32 / 32-bit-row kernel of up to 4 states per lane only");
    using R = double;
    static_assert(K == 2 || K == 4 || K == 8, "one wave per direction: 2, 4 or 8 states per lane (rows of up to 512 states)");
    constexpr int KH = K / 2;
    // Frames per block: the probabilities (and, phase 1, the other direction's rows) of a block are
    // prefetched a block ahead, a block's rows are stored / its gradient rows finished behind its last
    // frame, so that the frames themselves are ONE basic block without a store or a branch: whatever
    // does not feed the recursion is scheduled into the gaps of its dependency chain.
    // (two-wave form: 4 -- 190 instead of 296 registers, two waves per SIMD instead of one: 4.1 -> 3.5 ms at 4096
    // utterances; it is the form of the batches that fill the device)
    constexpr int PF = HELP ? (K == 8 ? 4 : 8) : SCTC_FUSED_PF2;
    constexpr int NI = NA == 1 ? 8 : 4;     // list entries of a label summed unconditionally
    constexpr int NI2 = (NA == 1 && !DIET) ? 16 : NI;  // list entries kept in registers (those beyond NI: summed when some list is that long)
    constexpr int NPOS = 64 * KH;           // label positions of one direction; slot NPOS holds 0.0
    constexpr int LSTR = NPOS + 2;
    constexpr int RSTR = (HELP && K == 2) ? 64 : 66;   // row stride of the reduction scratch (bank spread where it fits)
    using BlkU = RowBlockU<ST, K>;
    using BlkA = RowBlockA<ST, K>;

    // HELP: four more waves per utterance, two per direction, form the gradient (phase 1) from the rows the
    // recursion waves hand over through an LDS ring of 2 x HB frames -- helper `par` of a direction takes the
    // hand-overs of parity par, from ring half par.  A phase-1 frame then costs the recursion wave LESS than a
    // phase-0 frame (the row goes to LDS instead of HBM: 0.21 against 0.23-0.29 us, round 5), and the products,
    // sums and gradient rows issue on other SIMDs.  Without helpers they issue in the recursion wave itself: a
    // phase-1 frame is then issue-bound at 2-3x a phase-0 frame (0.45 ms against the three-kernel path's 0.31 at
    // cfg-3 minibatch 32); one helper per direction is not enough either (0.33-0.47 us per frame: its hand-over is a
    // chain of LDS round trips).  The saturating batch runs without helpers (more utterances resident per CU).
    constexpr int HB = 4;                   // frames per hand-over
    constexpr int FS = HELP ? HB : PF;      // frames per finish (LDS slots of label products, transposed sums)
    constexpr int NH = HELP ? 2 : 1;        // helpers per direction
    __shared__ __attribute__((aligned(16))) double lab_s[2][NH][FS][LSTR];   // alpha*beta of the label states, per frame
    // the transposed sums of a finish: their own scratch without helpers; with helpers the ring half the helper has
    // just read into registers (released once the sums are read back)
    // (8 states per lane: the ring is dynamic LDS and the helpers have room for sums of their own: their rows are
    // then multiplied one at a time instead of all HB being held in registers first)
    __shared__ __attribute__((aligned(16))) double red_s[(HELP && K < 8) ? 1 : (HELP ? 4 : 2)][(HELP && K < 8) ? 1 : 2 * FS][(HELP && K < 8) ? 2 : RSTR];
    // the ring: [direction][half][HB frames][64 K states]; 8 states per lane (rows of up to 512 states, round 5:
    // cfg-4's 401) make it 64 KiB, which only fits as dynamic LDS beside the rest
    constexpr bool DYN_RING = HELP && K == 8;
    extern __shared__ __attribute__((aligned(16))) double ring_dyn[];
    __shared__ __attribute__((aligned(16))) double ring_static[(HELP && !DYN_RING) ? 2 * 2 * HB * 64 * K : 2];
    double* const ring_base = DYN_RING ? ring_dyn : ring_static;
    auto ring_half = [&](int d, int half) -> double* { return ring_base + (size_t)((d * 2 + half) * HB) * 64 * K; };
    static_assert(!HELP || K == 8 || 2 * HB * RSTR <= HB * 64 * K, "the sums of a hand-over fit its ring half");
    __shared__ int32_t prod_s[2], cons_s[2][2];   // hand-overs published per direction / consumed per helper
    __shared__ int32_t ord_pos[2][NPOS];
    __shared__ int32_t sh_skip[2];
    __shared__ double sh_cost;     // written by wave 0 (dir 0), read after a barrier

    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const int dir = wave & 1;           // 0: alpha, 1: beta (== alpha of the reversed problem)
    const bool helper = wave >= 2;      // HELP only: waves 2..5
    const int par = helper ? (wave - 2) >> 1 : 0;
    const int lane = threadIdx.x & 63, gl = lane;
    const CtcUtt u = p.utts[b];
    const int T = u.T, U = u.U, L = 2 * U + 1;
    const int stride = (L + K - 1) / K * K;
    const int Ta = T / 2, Tb = T - Ta;
    const int Tst = dir ? Tb : Ta;      // my stored rows: tau in [0, Tst)
    ST* mine = reinterpret_cast<ST*>(p.store) + u.lat_off + (dir ? (int64_t)Ta * stride : 0);
    const ST* other = reinterpret_cast<const ST*>(p.store) + u.lat_off + (dir ? 0 : (int64_t)Ta * stride);
    const int32_t* seq = p.labels + u.lab_off;
    const int blank = p.blank;
    const int A = p.A;
    const RI* probs = p.probs;
    RI* grad = p.grad;
    const int64_t ld = p.ld;

    // ---- per-lane constants: labels of my odd states, skip-transition permission (ctc_lattice_kernel)
    int lab[KH];
    bool allow[KH], valid_lab[KH], valid_blk[KH];
    R allowf[KH];
#pragma unroll
    for (int jj = 0; jj < KH; ++jj) {
        const int idx = KH * gl + jj;
        const bool ok = idx < U;
        const int i0 = ok ? (dir ? U - 1 - idx : idx) : 0;
        lab[jj] = seq[i0];
        int prev = blank;
        if (ok && idx >= 1) prev = seq[dir ? U - idx : idx - 1];
        allow[jj] = ok && idx >= 1 && lab[jj] != prev;   // ctc_fast.pyx:64-68 / :103-107
        allowf[jj] = allow[jj] ? (R)1 : (R)0;
        valid_lab[jj] = ok;
        valid_blk[jj] = idx <= U;
    }
    // my states that exist (the other direction's block is read mirrored and runs past the row's
    // first element in the lane that holds state L-1)
    bool inrow[K];
#pragma unroll
    for (int j = 0; j < K; ++j) inrow[j] = K * gl + j < L;
    const bool store_lane = K * gl < stride;
    const int other_off = max(L - K * (gl + 1), -K);     // my block of the other direction's (mirrored) row
    ST* const pad = reinterpret_cast<ST*>(p.store) + u.lat_off - K;   // K elements in front of the utterance's rows: the dump

    // ---- gradient side: lane k (+64q) owns label k's list of label positions, in MY direction's order
    int lidx[NA][NI2];
    int llen[NA], lj0[NA];
    int maxlen = 0;
    {
        const int32_t* start = p.label_start + (int64_t)b * (A + 1);
        const int32_t* byl = p.by_label + u.lab_off;
        for (int j = lane; j < U; j += 64) {
            const int i = byl[j] >> 1;                    // by_label holds states 2i+1
            ord_pos[dir][j] = dir ? U - 1 - i : i;
        }
        if (lane < FS) {
            lab_s[dir][par][lane][NPOS] = 0.0;
            lab_s[dir][par][lane][NPOS + 1] = 0.0;
        }
        if (lane == 0 && !helper) {
            prod_s[dir] = 0;
            cons_s[dir][0] = 0;
            cons_s[dir][1] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int k = lane + 64 * q;
            const int j0 = k < A ? start[k] : 0, j1 = k < A ? start[k + 1] : 0;
            lj0[q] = j0;
            llen[q] = j1 - j0;
            maxlen = max(maxlen, llen[q]);
#pragma unroll
            for (int n = 0; n < NI2; ++n) lidx[q][n] = n < llen[q] ? ord_pos[dir][j0 + n] : NPOS;
        }
        maxlen = wave_max(maxlen);
    }

    // Accesses of ONE wave to LDS execute in program order: between a wave's own LDS writes and its reads of what
    // other lanes wrote (and between a hand-over's rows and its flag) only the COMPILER has to keep the order.  (A
    // workgroup-scope fence would also wait for the wave's global loads and stores: the helper would stall on its
    // gradient rows once per hand-over.)
    auto lds_order = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    };
    // NO load or store of the frame loops sits under a branch, not even a lane mask (round 6): behind control flow the
    // compiler no longer knows how many memory operations are in flight and every later wait becomes s_waitcnt vmcnt(0)
    // -- a block's prefetch then waited for the rows (phase 0) or gradient rows (phase 1) stored behind the block before,
    // an HBM write round trip per block of frames (found in ctc_fusedw.hip, where it cost 1.3 us per block).  Columns and
    // rows are clamped and masked by a select; lanes / frames with nothing to store write to the pad in front of the
    // utterance's rows, which nobody reads unmasked.
    auto load_row = [&](int64_t row, RI (&dst)[NA]) {
        const RI* yr = probs + row * ld;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int k = lane + 64 * q;
            const RI v = yr[min(k, A - 1)];
            dst[q] = k < A ? v : (RI)0;
        }
    };
    const int32_t* const rb_tab = p.rowbase ? p.rowbase : reinterpret_cast<const int32_t*>(p.utts);
    auto block_rows = [&](int tau0) -> int {
        const int tau = min(tau0 + lane, T - 1);
        const int t = dir ? T - 1 - tau : tau;
        const int v = rb_tab[p.rowbase ? t : 0];
        return p.rowbase ? v : t;
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
    auto gather_d = [&](const R (&y)[NA], int k) -> R {
        R out = lane_gather(y[0], k & 63);
#pragma unroll
        for (int q = 1; q < NA; ++q) {
            R o = lane_gather(y[q], k & 63);
            if ((k >> 6) == q) out = o;
        }
        return out;
    };
    auto bcast_d = [&](const R (&y)[NA], int k) -> R {
        R out = lane_bcast(y[0], k & 63);
#pragma unroll
        for (int q = 1; q < NA; ++q) {
            R o = lane_bcast(y[q], k & 63);
            if ((k >> 6) == q) out = o;
        }
        return out;
    };
    // the other direction's row of MY frame tau: its own time index is T-1-tau, its state order mine mirrored
    auto load_other = [&](int tau, BlkU& dst) {
        const int taup = T - 1 - min(tau, T - 1);
        // (a lane beyond the row reads the K elements in front of it; products masks them)
        dst = *reinterpret_cast<const BlkU*>(other + (int64_t)taup * stride + other_off);
    };
    auto recip = [&](R c) -> R {   // ctc_lattice_kernel: hardware estimate + two Newton-Raphson steps (<= 1 ulp)
        R x = __builtin_amdgcn_rcp(c);
        R e = fma(-c, x, (R)1);
        x = fma(x, e, x);
        e = fma(-c, x, (R)1);
        return fma(x, e, x);
    };
    // 1/y for absum's per-state division (:125-131); y == 0 only ever meets a product that is exactly zero
    // (the reference's `ab != 0` guard).  float32 probabilities are normal float64 numbers: the Newton
    // reciprocal; float64 probabilities may be denormal: the IEEE division.
    auto recip_or_zero = [&](R y) -> R {
        if constexpr (sizeof(RI) == 4) {
            const R r = recip(y);
            return y > (R)0 ? r : (R)0;
        } else {
            return y > (R)0 ? (R)1 / y : (R)0;
        }
    };
    auto local_sum = [&](const R (&n)[K]) -> R {
        if constexpr (K == 8) return ((n[0] + n[1]) + (n[2] + n[3])) + ((n[4] + n[5]) + (n[6] + n[7]));
        if constexpr (K == 2) return n[0] + n[1];
        else return (n[0] + n[1]) + (n[2] + n[3]);
    };
    auto quad_xor = [&](R v, auto ctrl_tag) -> R {
        constexpr int CTRL = decltype(ctrl_tag)::value;
        long long bits = __double_as_longlong(v);
        int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), CTRL, 0xF, 0xF, true);
        int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xF, 0xF, true);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };

    R a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = (R)0;
    constexpr int NO_BAD = 0x7fffffff;
    int first_bad = NO_BAD;
    int skip = 0;
    // llForward = sum_t log c_t (ctc_fast.pyx:47,76) = log of the product of the band sums c_t: the product is carried
    // as mantissa x 2^exponent (three operations per frame, beside the recursion's chain) and its logarithm taken
    // once; frames from the first zero band sum on do not count (the reference's exception leaves llForward as it
    // was, :147-149).  The product of the c_t themselves, not of the applied factors 1/c_t: for T = 1 the cost is then
    // the logarithm of the very double the reference takes it of (a cost of 1e-8 -- one frame, c = 1 - 1e-8 -- would
    // otherwise carry the reciprocal's rounding: 2e-8 relative, found by the fuzz test).
    R ll_m = (R)1;
    int ll_e = 0;
    auto ll_account = [&](R c) {
        const R f = first_bad == NO_BAD ? c : (R)1;
        ll_m *= f;
        ll_e += __builtin_amdgcn_frexp_exp(ll_m);
        ll_m = __builtin_amdgcn_frexp_mant(ll_m);
    };
    // T < U: empty band at every frame t >= 1 -- the reference divides nothing, adds log(0) and
    // returns cost +inf, grad = params, skip False (ctc_fast.pyx:70-76 on an empty range)
    const bool empty_band = (L >= 2 * T + 2) && T > 1;

    // one frame of the recursion; FAST: the band starts at state 0 (no band tests, ctc_lattice_kernel)
    auto step = [&](auto fast_tag, int tau, R yb, const R (&yl)[KH]) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const R prev_last = lane_shr1(a[K - 1]);
        R n[K];
        if constexpr (FAST) {
#pragma unroll
            for (int jj = 0; jj < KH; ++jj) {
                const R below = jj == 0 ? prev_last : a[2 * jj - 1];
                n[2 * jj] = (a[2 * jj] + below) * yb;
                n[2 * jj + 1] = fma(below, allowf[jj], a[2 * jj + 1] + a[2 * jj]) * yl[jj];
            }
        } else {
            // lower band limit, ctc_fast.pyx:49-53; states >= end are exactly zero by construction
            const int rem = 2 * (T - tau);
            const int start = L <= rem ? 0 : L - rem;
#pragma unroll
            for (int jj = 0; jj < KH; ++jj) {
                const R below = jj == 0 ? prev_last : a[2 * jj - 1];
                const int sb = K * gl + 2 * jj;
                const R vb = (a[2 * jj] + below) * yb;                 // :58-62
                n[2 * jj] = (valid_blk[jj] && sb >= start) ? vb : (R)0;
                R in = a[2 * jj + 1] + a[2 * jj];                      // :63-68
                if (allow[jj]) in += below;
                const R vl = in * yl[jj];
                n[2 * jj + 1] = (valid_lab[jj] && sb + 1 >= start) ? vl : (R)0;
            }
        }
        const R c = wave_sum(local_sum(n));
        R r;
        if constexpr (FAST) {
            first_bad = (c == (R)0 && first_bad == NO_BAD) ? tau : first_bad;   // ZeroDivisionError at :75
            r = recip(c);
        } else {
            first_bad = (c == (R)0 && !empty_band && first_bad == NO_BAD) ? tau : first_bad;
            r = empty_band ? (R)1 : recip(c);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) a[j] = n[j] * r;
        if constexpr (FAST) ll_account(c);
        else ll_account(empty_band ? (R)1 : c);
    };
    // phase 1, inside the frame: alpha*beta of my states against the other direction's stored row;
    // label products to the frame's LDS slot, the lane's share of absum[t] (zl) and of the blank sum (eb)
    auto products = [&](int slot, const R (&av)[K], const BlkU& ob, R rb, const R (&rl)[KH], R& zl, R& eb) {
        R ab[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const R o = Store<ST>::dec(ob.v[K - 1 - j]);
            // a state beyond the row is exactly 0.0 in av; what the mirrored block holds there (the row before, the pad) is
            // garbage -- finite whatever its bits in the 32-bit format (the top exponent bit is never set), so no mask there
            if constexpr (sizeof(ST) == 4) ab[j] = av[j] * o;           // :119
            else ab[j] = av[j] * (inrow[j] ? o : (R)0);
        }
        R z = (R)0, e = (R)0;
#pragma unroll
        for (int jj = 0; jj < KH; ++jj) {
            e += ab[2 * jj];                                             // blank states, :122-124
            z = fma(ab[2 * jj], rb, z);                                  // :125-126
            z = fma(ab[2 * jj + 1], rl[jj], z);                          // :130-131
        }
        zl = z;
        eb = e;
        if constexpr (KH == 4) {
            *reinterpret_cast<double2*>(&lab_s[dir][par][slot][4 * gl]) = make_double2(ab[1], ab[3]);
            *reinterpret_cast<double2*>(&lab_s[dir][par][slot][4 * gl + 2]) = make_double2(ab[5], ab[7]);
        } else if constexpr (KH == 2) {
            *reinterpret_cast<double2*>(&lab_s[dir][par][slot][2 * gl]) = make_double2(ab[1], ab[3]);
        } else {
            lab_s[dir][par][slot][gl] = ab[1];
        }
    };
    // phase 1, behind a block's frames: absum and the blank sum of its PF frames (2 PF sums over the wave,
    // transposed through LDS: lane -> (sum, quarter) instead of 2 PF wave reductions), the per-label sums
    // (:120-131) from the frames' LDS slots, the gradient rows (:138-145)
    auto finish_block = [&](auto nf_tag, double* red, auto&& release, int tb, int t_end,
                            const RI (&yc)[decltype(nf_tag)::value][NA], int rows,
                            const R (&zl)[decltype(nf_tag)::value], const R (&eb)[decltype(nf_tag)::value]) {
        constexpr int NF = decltype(nf_tag)::value;     // frames finished together: PF, or HB with helper waves
        constexpr int PARTS = 64 / (2 * NF);            // lanes per sum
        constexpr int PER = 64 / PARTS;                 // values per lane
        static_assert(NF <= FS && (NF == 8 || NF == 4), "one lane per (sum, part)");
        // the label products of the frames' slots: where they fit into registers they are fetched first and return
        // while the sums cross LDS
        constexpr bool EARLY = !DIET && NF * NA * NI <= 32;
        R lv[EARLY ? NF : 1][EARLY ? NA : 1][NI];
        if constexpr (EARLY) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int q = 0; q < NA; ++q)
#pragma unroll
                    for (int n = 0; n < NI; ++n) lv[i][q][n] = lab_s[dir][par][i][lidx[q][n]];
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            red[i * RSTR + lane] = zl[i];
            red[(NF + i) * RSTR + lane] = eb[i];
        }
        lds_order();
        R tot;
        {
            const double2* src = reinterpret_cast<const double2*>(red + (lane / PARTS) * RSTR + (lane % PARTS) * PER);
            double2 v[PER / 2];
#pragma unroll
            for (int n = 0; n < PER / 2; ++n) v[n] = src[n];
            lds_order();
            release();                                  // (helpers: the ring half may be written again)
            R s[PER / 2];
#pragma unroll
            for (int n = 0; n < PER / 2; ++n) s[n] = v[n].x + v[n].y;
#pragma unroll
            for (int w = 1; w < PER / 2; w *= 2)
#pragma unroll
                for (int n = 0; n + w < PER / 2; n += 2 * w) s[n] += s[n + w];
            tot = s[0];
            tot += quad_xor(tot, std::integral_constant<int, 0xB1>());   // quad_perm [1,0,3,2]
            tot += quad_xor(tot, std::integral_constant<int, 0x4E>());   // quad_perm [2,3,0,1]
            if constexpr (PARTS == 8) tot += __shfl_xor(tot, 4, 64);
        }
        R g[NF][NA];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                R v[NI];
#pragma unroll
                for (int n = 0; n < NI; ++n) {
                    if constexpr (EARLY) v[n] = lv[i][q][n];
                    else v[n] = lab_s[dir][par][i][lidx[q][n]];
                }
#pragma unroll
                for (int w = 1; w < NI; w *= 2)
#pragma unroll
                    for (int n = 0; n + w < NI; n += 2 * w) v[n] += v[n + w];
                g[i][q] = v[0];
            }
        if (maxlen > NI) {   // a label with more than NI positions (uniform over the wave)
            if constexpr (NI2 > NI) {
                // positions NI .. NI2-1 from registers, four at a time
#pragma unroll
                for (int n0 = NI; n0 < NI2; n0 += 4) {
                    if (maxlen > n0) {
#pragma unroll
                        for (int q = 0; q < NA; ++q) {
                            R w4[NF][4];
#pragma unroll
                            for (int i = 0; i < NF; ++i)
#pragma unroll
                                for (int n = 0; n < 4; ++n) w4[i][n] = lab_s[dir][par][i][lidx[q][n0 + n]];
#pragma unroll
                            for (int i = 0; i < NF; ++i) g[i][q] += (w4[i][0] + w4[i][1]) + (w4[i][2] + w4[i][3]);
                        }
                    }
                }
            }
            if (maxlen > NI2) {
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    for (int n = NI2; n < maxlen; ++n) {
                        const int pos = n < llen[q] ? ord_pos[dir][lj0[q] + n] : NPOS;
#pragma unroll
                        for (int i = 0; i < NF; ++i) g[i][q] += lab_s[dir][par][i][pos];
                    }
            }
        }
        RI out[NF][NA];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const R Z = lane_bcast(tot, PARTS * i);                      // absum[t], :133-136
            const R gb = lane_bcast(tot, PARTS * (NF + i));
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int k = lane + 64 * q;
                const R gk = k == blank ? g[i][q] + gb : g[i][q];
                const R y = (R)yc[i][q];
                const R tmp = y * Z;                                     // :141
                out[i][q] = (RI)(tmp > (R)0 ? y - gk / tmp : y);         // :142-145 (cast: brnnet.py:188)
            }
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int k = lane + 64 * q;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                RI* dst = (k < A && tb + i < t_end) ? grad + ((int64_t)__builtin_amdgcn_readlane(rows, i) + u.row0) * ld + k
                                                    : reinterpret_cast<RI*>(pad);
                *dst = out[i][q];
            }
        }
        // the next block's products overwrite the slots: LDS executes a wave's accesses in order, the
        // compiler must keep them in order too
        lds_order();
    };

    // ---- frames [t_begin, t_end) of my direction; PH 0: store the rows, PH 1: form the gradient
    auto run = [&](auto ph_tag, int t_begin, int t_end) {
        constexpr int PH = decltype(ph_tag)::value;
        if (skip || t_begin >= t_end) return;
        RI ycur[PF][NA];
        BlkU ocur[PH ? PF : 1];
        int rb_cur = block_rows(t_begin);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            load_row((int64_t)__builtin_amdgcn_readlane(rb_cur, i) + u.row0, ycur[i]);
            if constexpr (PH) load_other(t_begin + i, ocur[i]);
        }
        int rb_nxt = block_rows(t_begin + PF);
        for (int tb = t_begin; tb < t_end && !skip; tb += PF) {
            RI ynxt[PF][NA];
            BlkU onxt[PH ? PF : 1];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                load_row((int64_t)__builtin_amdgcn_readlane(rb_nxt, i) + u.row0, ynxt[i]);
                if constexpr (PH) load_other(tb + PF + i, onxt[i]);
            }
            const int rb_next2 = block_rows(tb + 2 * PF);
            // the block's probabilities per state, gathered before the serial part starts
            R ybv[PF], ylv[PF][KH];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                ybv[i] = bcast(ycur[i], blank);
#pragma unroll
                for (int jj = 0; jj < KH; ++jj) {
                    const R g = gather(ycur[i], lab[jj]);
                    ylv[i][jj] = valid_lab[jj] ? g : (R)0;
                }
            }
            BlkA enc[PH ? 1 : PF];
            R zl[PH ? PF : 1], eb[PH ? PF : 1];
            auto post = [&](int i) {
                if constexpr (PH) {
                    R rl[KH], rb;
                    if constexpr (NA < KH + 1) {
                        // 1/y once per SYMBOL (lane k <- 1/y[k]) and gathered per state like y itself, instead of once per
                        // state: NA reciprocals (8 VALU instructions each) instead of KH + 1 -- the same values bit for bit
                        R ry[NA];
#pragma unroll
                        for (int q = 0; q < NA; ++q) ry[q] = recip_or_zero((R)ycur[i][q]);
                        rb = bcast_d(ry, blank);
#pragma unroll
                        for (int jj = 0; jj < KH; ++jj) {
                            const R g = gather_d(ry, lab[jj]);
                            rl[jj] = valid_lab[jj] ? g : (R)0;
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < KH; ++jj) rl[jj] = recip_or_zero(ylv[i][jj]);
                        rb = recip_or_zero(ybv[i]);
                    }
                    products(i, a, ocur[i], rb, rl, zl[i], eb[i]);
                } else {
#pragma unroll
                    for (int j = 0; j < K; ++j) enc[i].v[j] = Store<ST>::enc(a[j]);
                }
            };
            const bool fast = (tb + PF - 1 < t_end) && (L <= 2 * (T - (tb + PF - 1)));
            if (fast) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    step(std::true_type(), tb + i, ybv[i], ylv[i]);
                    post(i);
                }
            } else {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    if (tb + i < t_end) {
                        step(std::false_type(), tb + i, ybv[i], ylv[i]);
                        post(i);
                    } else if constexpr (PH) {
                        zl[i] = (R)0;
                        eb[i] = (R)0;
                    }
                }
            }
            if constexpr (PH) {
                if constexpr (!HELP)
                    finish_block(std::integral_constant<int, PF>(), &red_s[dir][0][0], [] {}, tb, t_end, ycur, rb_cur, zl, eb);
            } else {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    ST* dst = (store_lane && tb + i < t_end) ? mine + (int64_t)(tb + i) * stride + K * gl : pad;
                    *reinterpret_cast<BlkA*>(dst) = enc[i];
                }
            }
            if (first_bad != NO_BAD) skip = 1;
#pragma unroll
            for (int i = 0; i < PF; ++i) {
#pragma unroll
                for (int q = 0; q < NA; ++q) ycur[i][q] = ynxt[i][q];
                if constexpr (PH) ocur[i] = onxt[i];
            }
            rb_cur = rb_nxt;
            rb_nxt = rb_next2;
        }
    };

    // ---- HELP, phase 1, recursion wave: the frames [t_begin, t_end) like phase 0, the rows handed to the helper
    // through ring_s -- HB frames per hand-over, two hand-overs in flight
    constexpr int ABORT = 0x7fffffff;
    auto publish = [&](int v) {
        asm volatile("" ::: "memory");
        __hip_atomic_store(&prod_s[dir], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
    };
    // every spin is bounded (about half a second): both waves of a hand-over belong to one workgroup and are
    // resident together, so a wait is a few microseconds -- but a wait that never ends would take the device with
    // it.  A wait that ran out makes the utterance's cost NaN (the host sees it).
    int timed_out = 0;
    auto wait_for = [&](int32_t* word, int need) {
        if (need <= 0) return;
        int spins = 0;
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 23)) {
                timed_out = 1;
                break;
            }
        }
        asm volatile("" ::: "memory");
    };
    auto produce = [&](int t_begin, int t_end) {
        if (t_begin >= t_end) return;
        if (skip) {
            publish(ABORT);
            return;
        }
        const int nblk = (t_end - t_begin + PF - 1) / PF;
        RI ycur[PF][NA];
        int rb_cur = block_rows(t_begin);
#pragma unroll
        for (int i = 0; i < PF; ++i) load_row((int64_t)__builtin_amdgcn_readlane(rb_cur, i) + u.row0, ycur[i]);
        int rb_nxt = block_rows(t_begin + PF);
        for (int blk = 0; blk < nblk && !skip; ++blk) {
            const int tb = t_begin + blk * PF;
            RI ynxt[PF][NA];
#pragma unroll
            for (int i = 0; i < PF; ++i) load_row((int64_t)__builtin_amdgcn_readlane(rb_nxt, i) + u.row0, ynxt[i]);
            const int rb_next2 = block_rows(tb + 2 * PF);
            R ybv[PF], ylv[PF][KH];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                ybv[i] = bcast(ycur[i], blank);
#pragma unroll
                for (int jj = 0; jj < KH; ++jj) {
                    const R g = gather(ycur[i], lab[jj]);
                    ylv[i][jj] = valid_lab[jj] ? g : (R)0;
                }
            }
            const bool fast = (tb + PF - 1 < t_end) && (L <= 2 * (T - (tb + PF - 1)));
#pragma unroll
            for (int hb = 0; hb < PF / HB; ++hb) {
                const int h = blk * (PF / HB) + hb;
                if (!(p.diag & 8)) wait_for(&cons_s[dir][h & 1], h >> 1);   // hand-over h - 2 (same ring half) has been consumed
                double2* slot = reinterpret_cast<double2*>(ring_half(dir, h & 1));
                if (fast) {
#pragma unroll
                    for (int i = 0; i < HB; ++i) {
                        step(std::true_type(), tb + hb * HB + i, ybv[hb * HB + i], ylv[hb * HB + i]);
#pragma unroll
                        for (int j = 0; j < K; j += 2) slot[(i * 64 * K + K * lane + j) / 2] = make_double2(a[j], a[j + 1]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < HB; ++i) {
                        if (tb + hb * HB + i < t_end)
                            step(std::false_type(), tb + hb * HB + i, ybv[hb * HB + i], ylv[hb * HB + i]);
#pragma unroll
                        for (int j = 0; j < K; j += 2) slot[(i * 64 * K + K * lane + j) / 2] = make_double2(a[j], a[j + 1]);
                    }
                }
                publish(h + 1);
            }
            if (first_bad != NO_BAD) skip = 1;
#pragma unroll
            for (int i = 0; i < PF; ++i)
#pragma unroll
                for (int q = 0; q < NA; ++q) ycur[i][q] = ynxt[i][q];
            rb_cur = rb_nxt;
            rb_nxt = rb_next2;
        }
        if (skip) publish(ABORT);   // the helper runs through what is left (its rows are taken back below)
    };
    // ---- HELP, phase 1, helper wave `par`: the hand-overs h = par, par + 2, ... -- per hand-over the products against
    // the other direction's stored rows, then the HB gradient rows
    auto consume = [&](int t_begin, int t_end) {
        if (t_begin >= t_end) return;
        const int nh = (t_end - t_begin + PF - 1) / PF * (PF / HB);
        RI ycur[HB][NA];
        BlkU ocur[HB];
        int rb_cur = block_rows(t_begin + par * HB);
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            load_row((int64_t)__builtin_amdgcn_readlane(rb_cur, i) + u.row0, ycur[i]);
            load_other(t_begin + par * HB + i, ocur[i]);
        }
        int rb_nxt = block_rows(t_begin + (par + 2) * HB);
        double* half = ring_half(dir, par);
        for (int h = par; h < nh; h += 2) {
            const int tb = t_begin + h * HB;
            RI ynxt[HB][NA];
            BlkU onxt[HB];
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                load_row((int64_t)__builtin_amdgcn_readlane(rb_nxt, i) + u.row0, ynxt[i]);
                load_other(tb + 2 * HB + i, onxt[i]);
            }
            const int rb_next2 = block_rows(tb + 4 * HB);
            // 1/y for every symbol of a frame at once (lane k <- 1/y[k]), then gathered per state like y itself
            R rbv[HB], rlv[HB][KH];
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                R ry[NA];
#pragma unroll
                for (int q = 0; q < NA; ++q) ry[q] = recip_or_zero((R)ycur[i][q]);
                rbv[i] = bcast_d(ry, blank);
#pragma unroll
                for (int jj = 0; jj < KH; ++jj) {
                    const R g = gather_d(ry, lab[jj]);
                    rlv[i][jj] = valid_lab[jj] ? g : (R)0;
                }
            }
            wait_for(&prod_s[dir], h + 1);
            R zl[HB], eb[HB];
            const double2* slot = reinterpret_cast<const double2*>(half);
            auto release = [&]() {
                __hip_atomic_store(&cons_s[dir][par], (h >> 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
            };
            if constexpr (DYN_RING) {
#pragma unroll
                for (int i = 0; i < HB; ++i) {
                    R av[K];
#pragma unroll
                    for (int j = 0; j < K; j += 2) {
                        const double2 v = slot[(i * 64 * K + K * lane + j) / 2];
                        av[j] = v.x;
                        av[j + 1] = v.y;
                    }
                    products(i, av, ocur[i], rbv[i], rlv[i], zl[i], eb[i]);
                }
                lds_order();
                release();       // the rows have been read; the sums go to this helper's own scratch
                finish_block(std::integral_constant<int, HB>(), &red_s[2 * dir + par][0][0], [] {}, tb, t_end, ycur, rb_cur, zl, eb);
            } else {
            R av[HB][K];
#pragma unroll
            for (int i = 0; i < HB; ++i)
#pragma unroll
                for (int j = 0; j < K; j += 2) {
                    const double2 v = slot[(i * 64 * K + K * lane + j) / 2];
                    av[i][j] = v.x;
                    av[i][j + 1] = v.y;
                }
            lds_order();     // the rows are in registers before the half is reused for the sums
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                if (!(p.diag & 4)) products(i, av[i], ocur[i], rbv[i], rlv[i], zl[i], eb[i]);
                else { zl[i] = av[i][0]; eb[i] = av[i][1]; }
            }
            if (!(p.diag & 2)) finish_block(std::integral_constant<int, HB>(), half, release, tb, t_end, ycur, rb_cur, zl, eb);
            else release();
            }
#pragma unroll
            for (int i = 0; i < HB; ++i) {
#pragma unroll
                for (int q = 0; q < NA; ++q) ycur[i][q] = ynxt[i][q];
                ocur[i] = onxt[i];
            }
            rb_cur = rb_nxt;
            rb_nxt = rb_next2;
        }
    };

    // ---- tau = 0 (ctc_fast.pyx:42-47 / :79-84); it belongs to phase 1 only when T == 1 (alpha stores nothing)
    if (!helper) {
        RI y0[NA];
        const int t = dir ? T - 1 : 0;
        const int64_t row0 = (int64_t)(p.rowbase ? p.rowbase[t] : t) + u.row0;
        load_row(row0, y0);
        const R yb = bcast(y0, blank);
        const R yl = gather(y0, lab[0]);
        if (gl == 0) { a[0] = yb; a[1] = yl; }
        const R c = wave_sum(a[0] + a[1]);
        if (c == (R)0) {
            skip = 1;   // ZeroDivisionError at :45
            first_bad = 0;
        } else {
            const R r = recip(c);
            a[0] *= r;
            a[1] *= r;
            ll_account(c);
        }
        if (Tst > 0 && store_lane) {
            BlkA blk;
#pragma unroll
            for (int j = 0; j < K; ++j) blk.v[j] = Store<ST>::enc(a[j]);
            *reinterpret_cast<BlkA*>(mine + K * gl) = blk;
        }
        run(std::integral_constant<int, 0>(), 1, Tst);
    }
    __syncthreads();   // phase 0 rows of both directions are in L2 (vmcnt(0) + barrier; same CU)
    if (!helper && Tst == 0 && !skip) {
        // T == 1, alpha: the one frame's gradient against beta's stored row 0 (the recursion wave itself)
        RI yc[FS][NA];
        const int rows = p.rowbase ? p.rowbase[0] : 0;
#pragma unroll
        for (int i = 0; i < FS; ++i) load_row((int64_t)rows + u.row0, yc[i]);
        BlkU ob;
        load_other(0, ob);
        R rl[KH], zl[FS], eb[FS];
#pragma unroll
        for (int i = 0; i < FS; ++i) { zl[i] = (R)0; eb[i] = (R)0; }
#pragma unroll
        for (int jj = 0; jj < KH; ++jj) {
            const R g = gather(yc[0], lab[jj]);
            rl[jj] = recip_or_zero(valid_lab[jj] ? g : (R)0);
        }
        products(0, a, ob, recip_or_zero(bcast(yc[0], blank)), rl, zl[0], eb[0]);
        finish_block(std::integral_constant<int, FS>(), (HELP && K < 8) ? ring_half(0, 0) : &red_s[0][0][0], [] {}, 0, 1, yc, rows, zl, eb);
    }
    const int t1_end = (p.diag & 1) ? 0 : T;
    if constexpr (HELP) {
        if (helper) consume(Tst > 0 ? Tst : 1, t1_end);
        else produce(Tst > 0 ? Tst : 1, t1_end);
    } else {
        run(std::integral_constant<int, 1>(), Tst > 0 ? Tst : 1, t1_end);
    }

    if (wave == 0 && lane == 0) {
        // -llForward (ctc_fast.pyx:149,152); math.log(0.0) for the empty band.  Mantissa in [sqrt(1/2), sqrt(2)): a
        // product near 1 has exponent 0 and its logarithm no cancellation against exponent x ln 2
        if (ll_m < 0.70710678118654752440) {
            ll_m *= 2.0;
            ll_e -= 1;
        }
        double cost = -(log(ll_m) + (double)ll_e * 0.693147180559945309417232121458);
        if (empty_band && !skip) cost = INFINITY;
        sh_cost = cost;
    }
    if (lane == 0 && !helper) sh_skip[dir] = skip;
    __syncthreads();
    if (timed_out && lane == 0) sh_cost = NAN;     // (benign race: every writer writes NaN)
    __syncthreads();
    const int any_skip = sh_skip[0] | sh_skip[1];
    if (threadIdx.x == 0) {
        p.cost[b] = sh_cost;
        p.skip[b] = any_skip;
    }
    if (any_skip) {
        // the reference returns its zero-initialised grad (ctc_fast.pyx:31-32,149): rows written before
        // the failing frame was reached are taken back
        for (int t = wave; t < T; t += (HELP ? 6 : 2)) {
            RI* gr = grad + ((int64_t)(p.rowbase ? p.rowbase[t] : t) + u.row0) * ld;
            for (int k = lane; k < A; k += 64) gr[k] = (RI)0;
        }
    }
}

// ---------------------------------------------------------------- launcher

template <typename RI, typename ST, int K, int NA, bool HELP>
static int launch_fused_one(const CtcFusedArgs<RI>& a, int B, hipStream_t stream)
{
    dim3 grid(B), block(HELP ? 384 : 128);
    // 8 states per lane with helper waves: the 64 KiB ring is dynamic LDS (static + dynamic > 64 KiB needs the attribute)
    const size_t dyn = (HELP && K == 8) ? sizeof(double) * 2 * 2 * 4 * 64 * K : 0;
    if (dyn) {
        static std::once_flag once;     // one per instantiation
        static hipError_t attr = hipSuccess;
        std::call_once(once, [&] {
            attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_fused_kernel<RI, ST, K, NA, HELP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        });
        SCTC_HIP_TRY(attr);
    }
    if constexpr (FusedDiet<RI, ST, K, NA, HELP>::possible) {
        const char* dz = getenv("SCTC_CTC_DIET_MIN_B");      // tests / A/B: the dieted form from this many utterances on
        if (B >= (dz ? atoi(dz) : FUSED_DIET_MIN_B)) {
            hipLaunchKernelGGL((ctc_fused_kernel<RI, ST, K, NA, HELP, true>), grid, block, dyn, stream, a);
            SCTC_HIP_TRY(hipGetLastError());
            return SCTC_OK;
        }
    }
    hipLaunchKernelGGL((ctc_fused_kernel<RI, ST, K, NA, HELP>), grid, block, dyn, stream, a);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

template <typename RI, typename ST, int K, bool HELP>
static int launch_fused_kh(const CtcFusedArgs<RI>& a, int B, int NA, hipStream_t stream)
{
    if constexpr (!(HELP && K == 8 && sizeof(ST) == 8)) {
        if (NA == 1) return launch_fused_one<RI, ST, K, 1, HELP>(a, B, stream);
    }
    if constexpr (!(HELP && K == 8)) {
        if (NA == 2) return launch_fused_one<RI, ST, K, 2, HELP>(a, B, stream);
    }
    if constexpr (!HELP) return launch_fused_one<RI, ST, K, 4, false>(a, B, stream);
    return set_error(SCTC_ERR_ARG, "ctc: no helper form for %d probability registers per frame", NA);
}

template <typename RI, typename ST, int K>
static int launch_fused_k(const CtcFusedArgs<RI>& a, int B, int NA, hipStream_t stream)
{
    // Helper waves (six waves per utterance) while the batch leaves SIMDs idle; beyond SCTC_CTC_HELPER_MAX_B
    // utterances (default 256: one six-wave workgroup per CU) the two-wave form keeps more utterances resident
    // (four workgroups per CU; 4096 utterances of the cfg-3 shape: 3.5 ms against 4.5).  Alphabets
    // of more than 128 symbols (four probability registers per frame) always take the two-wave form: the helper's
    // register budget is 256.  SCTC_CTC_HELPER=0 / 1 forces one form (A/B, tests).
    const char* hz = getenv("SCTC_CTC_HELPER");
    const char* mz = getenv("SCTC_CTC_HELPER_MAX_B");
    const int max_b = mz ? atoi(mz) : 256;
    // (8 states per lane: helpers only for alphabets of up to 64 symbols -- the register budget again)
    // and for the 32-bit row store -- the register budget again)
    const bool help = (K == 8 ? (NA == 1 && sizeof(ST) == 4) : NA <= 2) && (hz ? atoi(hz) != 0 : B <= max_b);
    if (help) return launch_fused_kh<RI, ST, K, true>(a, B, NA, stream);
    return launch_fused_kh<RI, ST, K, false>(a, B, NA, stream);
}

template <typename RI>
int launch_ctc_fused(const CtcFusedArgs<RI>& a, int B, int K, int store_bytes, hipStream_t stream)
{
    const int NA = a.A <= 64 ? 1 : (a.A <= 128 ? 2 : 4);
    if constexpr (sizeof(RI) == 4) {
        if (store_bytes == 4) {
            if (K == 2) return launch_fused_k<RI, uint32_t, 2>(a, B, NA, stream);
            if (K == 4) return launch_fused_k<RI, uint32_t, 4>(a, B, NA, stream);
            if (K == 8) return launch_fused_k<RI, uint32_t, 8>(a, B, NA, stream);
        }
    }
    if (store_bytes == 8) {
        if (K == 2) return launch_fused_k<RI, double, 2>(a, B, NA, stream);
        if (K == 4) return launch_fused_k<RI, double, 4>(a, B, NA, stream);
        if (K == 8) return launch_fused_k<RI, double, 8>(a, B, NA, stream);
    }
    return set_error(SCTC_ERR_ARG, "ctc: no fused kernel for K=%d, %d-byte rows", K, store_bytes);
}

template int launch_ctc_fused<float>(const CtcFusedArgs<float>&, int, int, int, hipStream_t);
template int launch_ctc_fused<double>(const CtcFusedArgs<double>&, int, int, int, hipStream_t);

}  // namespace sctc
