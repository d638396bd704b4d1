/*
 * oracle/ctc_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float64) of the reference's CTC loss
 *   /root/reference/ctc_fast/ctc-loss/ctc_fast.pyx:13-152  (ctc_loss)
 *   /root/reference/ctc_fast/ctc-loss/ctc_fast.pyx:154-187 (decode_best_path)
 * (ctc/ctc_fast.pyx:14-153 is the same body).  It is the parity checker for the
 * HIP kernels and the "port" CPU baseline of bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the
 * product path (stanford-ctc_amd/) never links or imports anything from here.
 *
 * Pinning: checked against golden vectors produced by the reference's own
 * Cython module (tests/golden/make_golden.py, tests/test_oracle_golden.py):
 * ctc/time_trials.py input -> 1710.233966660, brute-force cases, skip cases.
 *
 * Memory convention: the reference takes `params` as float64 (A,T) Fortran
 * order, i.e. T consecutive frames of A probabilities.  Here that is the
 * row-major array y[t*A + k].  grad uses the same layout.  The alpha/beta
 * lattices are (L',T) Fortran order in the reference == lat[t*L + s] here.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* label id of lattice state s: blanks on even s, seq[(s-1)/2] on odd s
 * (ctc_fast.pyx:56-68: `l = (s-1)/2`, blank branch on s%2==0). */
static inline int state_label(const int32_t *seq, int blank, int s)
{
    return (s & 1) ? seq[(s - 1) >> 1] : blank;
}

/* band limits at frame t (ctc_fast.pyx:49-54 and :87-92, same formula) */
static inline void band(int L, int T, int t, int *start, int *end)
{
    int st = 2 * (T - t);
    *start = (L <= st) ? 0 : L - st;
    int en = 2 * t + 2;
    *end = en < L ? en : L;
}

/*
 * Returns 0 on success, 1 when the reference would have taken its
 * `except (FloatingPointError, ZeroDivisionError)` exit (ctc_fast.pyx:147-149):
 * a frame normaliser c == 0.  In that case *cost = -llForward accumulated so
 * far and grad holds whatever had been written (all zeros: the gradient block
 * runs after both passes), exactly like the reference.
 * Negative return: argument error.
 */
int sctc_oracle_ctc_loss(const double *y, int A, int T, const int32_t *seq, int U,
                         int blank, double *grad, double *cost, double *ll_backward)
{
    if (!y || !seq || !grad || !cost || A <= 0 || T <= 0 || U <= 0) return -1;
    const int L = 2 * U + 1;
    double *alpha = (double *)calloc((size_t)L * T, sizeof(double));
    double *beta = (double *)calloc((size_t)L * T, sizeof(double));
    double *ab = (double *)malloc((size_t)L * T * sizeof(double));
    if (!alpha || !beta || !ab) { free(alpha); free(beta); free(ab); return -2; }
    memset(grad, 0, (size_t)A * T * sizeof(double));

    int skip = 0;
    double llf = 0.0, llb = 0.0, c;

    /* --- alpha init, ctc_fast.pyx:42-47 ---------------------------------- */
    alpha[0] = y[blank];
    alpha[1] = y[seq[0]];
    c = alpha[0] + alpha[1];
    if (c == 0.0) { skip = 1; goto done; }
    alpha[0] /= c;
    alpha[1] /= c;
    llf = log(c);

    /* --- alpha recursion, ctc_fast.pyx:48-76 ----------------------------- */
    for (int t = 1; t < T; ++t) {
        int start, end;
        band(L, T, t, &start, &end);
        const double *yt = y + (size_t)t * A;
        const double *prev = alpha + (size_t)(t - 1) * L;
        double *cur = alpha + (size_t)t * L;
        for (int s = start; s < L; ++s) {          /* :55 runs to L, not end */
            double in = prev[s];
            if (s >= 1) in += prev[s - 1];
            /* skip transition only between distinct non-blank labels (:64-68) */
            if ((s & 1) && s >= 3 && seq[(s - 1) / 2] != seq[(s - 1) / 2 - 1]) in += prev[s - 2];
            cur[s] = in * yt[state_label(seq, blank, s)];
        }
        c = 0.0;
        for (int s = start; s < end; ++s) c += cur[s];     /* :71-73 */
        /* ZeroDivisionError at :75 only if the band is non-empty; an empty band
         * (start >= end, happens when T is too short for U) divides nothing and
         * the reference goes on to math.log(0.0) = -inf  =>  cost = +inf, skip False */
        if (c == 0.0 && start < end) { skip = 1; goto done; }
        for (int s = start; s < end; ++s) cur[s] /= c;
        llf += log(c);
    }

    /* --- beta init, ctc_fast.pyx:79-84 ------------------------------------ */
    {
        double *last = beta + (size_t)(T - 1) * L;
        const double *yt = y + (size_t)(T - 1) * A;
        last[L - 1] = yt[blank];
        last[L - 2] = yt[seq[U - 1]];
        c = last[L - 1] + last[L - 2];
        if (c == 0.0) { skip = 1; goto done; }
        last[L - 1] /= c;
        last[L - 2] /= c;
        llb = log(c);
    }
    /* --- beta recursion, ctc_fast.pyx:85-114 ------------------------------ */
    for (int t = T - 2; t >= 0; --t) {
        int start, end;
        band(L, T, t, &start, &end);
        const double *yt = y + (size_t)t * A;
        const double *nxt = beta + (size_t)(t + 1) * L;
        double *cur = beta + (size_t)t * L;
        for (int s = end - 1; s >= 0; --s) {       /* :93 runs down to 0, not start */
            double in = nxt[s];
            if (s <= L - 2) in += nxt[s + 1];
            if ((s & 1) && s <= L - 4 && seq[(s - 1) / 2] != seq[(s - 1) / 2 + 1]) in += nxt[s + 2];
            cur[s] = in * yt[state_label(seq, blank, s)];
        }
        c = 0.0;
        for (int s = start; s < end; ++s) c += cur[s];
        if (c == 0.0 && start < end) { skip = 1; goto done; }
        for (int s = start; s < end; ++s) cur[s] /= c;
        llb += log(c);
    }

    /* --- gradient wrt pre-softmax activations, ctc_fast.pyx:117-145 ------- */
    for (int t = 0; t < T; ++t) {
        const double *yt = y + (size_t)t * A;
        double *gt = grad + (size_t)t * A;
        double *abt = ab + (size_t)t * L;
        const double *at = alpha + (size_t)t * L, *bt = beta + (size_t)t * L;
        double z = 0.0;
        for (int s = 0; s < L; ++s) {
            double v = at[s] * bt[s];                          /* :119 */
            int k = state_label(seq, blank, s);
            gt[k] += v;                                        /* :124, :129 */
            if (v != 0.0) v = v / yt[k];                       /* :125-126, :130-131 */
            abt[s] = v;
            z += v;                                            /* :133-136 absum */
        }
        for (int k = 0; k < A; ++k) {                          /* :139-145 */
            double tmp = yt[k] * z;
            gt[k] = (tmp > 0.0) ? yt[k] - gt[k] / tmp : yt[k];
        }
    }

done:
    *cost = -llf;
    if (ll_backward) *ll_backward = -llb;
    free(alpha); free(beta); free(ab);
    return skip;
}

/*
 * Convenience for the BRNN restatement and the fp32 device path: takes
 * pre-softmax activations (logits, row-major [T][A]), applies the reference's
 * softmax (brnnet.py:161-168 / rnnetcpu.py:99-103: subtract the frame max, exp,
 * divide by the frame sum) in float64 and then runs ctc_loss.  `probs_out`
 * (optional) receives the softmax.
 */
int sctc_oracle_ctc_loss_logits(const double *logits, int A, int T, const int32_t *seq,
                                int U, int blank, double *grad, double *cost,
                                double *probs_out)
{
    double *y = (double *)malloc((size_t)A * T * sizeof(double));
    if (!y) return -2;
    for (int t = 0; t < T; ++t) {
        const double *x = logits + (size_t)t * A;
        double *yt = y + (size_t)t * A;
        double m = x[0], z = 0.0;
        for (int k = 1; k < A; ++k) if (x[k] > m) m = x[k];
        for (int k = 0; k < A; ++k) { yt[k] = exp(x[k] - m); z += yt[k]; }
        for (int k = 0; k < A; ++k) yt[k] /= z;
    }
    int rc = sctc_oracle_ctc_loss(y, A, T, seq, U, blank, grad, cost, NULL);
    if (probs_out) memcpy(probs_out, y, (size_t)A * T * sizeof(double));
    free(y);
    return rc;
}

/*
 * Batched driver used for the multi-core CPU baseline: B independent
 * utterances, OpenMP over utterances (the reference itself is single-threaded
 * and holds the GIL; SURVEY 8(d) asks for both figures).
 * y/grad: concatenated [sum_b T_b][A]; frame_off[b] = first frame of utterance b;
 * seq concatenated, seq_off[b] = first label.  skip[b] receives the return code.
 */
void sctc_oracle_ctc_loss_batch(const double *y, int A, int B, const int32_t *T_b,
                                const int64_t *frame_off, const int32_t *seq,
                                const int32_t *U_b, const int64_t *seq_off, int blank,
                                double *grad, double *cost, int32_t *skip, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b) {
        skip[b] = sctc_oracle_ctc_loss(y + frame_off[b] * A, A, T_b[b], seq + seq_off[b], U_b[b],
                                       blank, grad + frame_off[b] * A, cost + b, NULL);
    }
    (void)nthreads;
}

/*
 * decode_best_path, ctc_fast.pyx:154-187: per-frame argmax (np.argmax: first
 * maximum wins), drop blanks, drop the hard-coded ids 1, 2, 8 (:176-179),
 * collapse repeats while recording the last frame of each emitted label.
 * Returns the hypothesis length; hyp/align must hold T entries.
 */
int sctc_oracle_decode_best_path(const double *y, int A, int T, int blank, int32_t *hyp,
                                 int32_t *align)
{
    int n = 0;
    int prev = -1;
    for (int t = 0; t < T; ++t) {
        const double *yt = y + (size_t)t * A;
        int b = 0;
        for (int k = 1; k < A; ++k) if (yt[k] > yt[b]) b = k;
        int prev_b = prev;
        prev = b;
        if (b == blank) continue;
        if (b == 1 || b == 2 || b == 8) continue;
        if (t != 0 && b == prev_b) {
            /* :181-183 `align[-1] = i` -- the reference indexes the last list
             * element; on an empty list that raises IndexError there.  We
             * mirror the effect only when something was emitted. */
            if (n > 0) align[n - 1] = t; else return -3;
            continue;
        }
        hyp[n] = b;
        align[n] = t;
        ++n;
    }
    return n;
}
