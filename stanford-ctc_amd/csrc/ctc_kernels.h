// Host-visible declarations of the CTC kernels (ctc_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sctc {

// per-utterance descriptor, uploaded by the host once per batch
struct CtcUtt {
    int32_t T, U;
    int64_t row0;     // first row (contiguous layout) or rank in the packed minibatch
    int64_t lat_off;  // element offset of this utterance's lattice
    int64_t lab_off;  // first label in the concatenated label array
};

// RI = type of probs/grad in HBM (float or double).  The lattices and all lattice
// arithmetic are float64 like the reference (ctc_fast.pyx:8,28-30): a float32 lattice
// loses utterances with T >> 2U, where the states that survive the end-of-utterance band
// carry < 1e-38 of a frame's mass (measured: 0.5-6 % cost error).
template <typename RI>
struct CtcLatticeArgs {
    const CtcUtt* utts;
    const RI* probs;
    int64_t ld;
    int32_t A, blank, lp;
    const int32_t* rowbase;  // nullable
    const int32_t* labels;
    double* alpha;
    double* beta;
    double* ll;      // [2B] llForward / llBackward
    int32_t* skip2;  // [2B]
    double* scratch; // ctc_generic.hip only: [B][2 directions][2][lp] unnormalised rows
};

template <typename RI>
struct CtcGradArgs {
    const CtcUtt* utts;
    const RI* probs;
    RI* grad;
    int64_t ld;
    int32_t A, blank, lp;
    const int32_t* rowbase;
    const int32_t* labels;
    // the odd (label) states of every utterance grouped by label, ascending inside a group: by_label[lab_off + j]
    // = state index 2i+1, group k = [label_start[b*(A+1) + k], label_start[b*(A+1) + k + 1]) (host-built, capi_ctc.hip)
    const int32_t* by_label;
    const int32_t* label_start;
    const double* alpha;
    const double* beta;
    const double* ll;
    const int32_t* skip2;
    double* cost;   // [B]
    int32_t* skip;  // [B]
    int32_t lazy;   // lattices were produced with the every-4th-frame rescaling
};

// ctc_fused.hip: both recursions and the gradient in one kernel (label rows of up to 512 states).
// `store` holds, per utterance, T rows of round_up(2U+1, K) states (alpha's first T/2 frames, then
// beta's first T - T/2) behind a K-element pad, as float64 or in the 32-bit format of ctc_fused.hip;
// CtcUtt::lat_off counts elements of that type.
template <typename RI>
struct CtcFusedArgs {
    const CtcUtt* utts;
    const RI* probs;
    RI* grad;
    int64_t ld;
    int32_t A, blank;
    const int32_t* rowbase;  // nullable
    const int32_t* labels;
    const int32_t* by_label;     // as in CtcGradArgs
    const int32_t* label_start;
    void* store;
    double* cost;    // [B]
    int32_t* skip;   // [B]
    int32_t diag;    // SCTC_CTC_DIAG, timing experiments only (results are wrong): 1 no phase 1, 2 helper without
                     // finish, 4 helper without products, 8 recursion waves do not wait for the helper
    uint32_t* sync;  // ctc_fusedw.hip only: ctc_fusedw_sync_bytes(B) of flag words (the launcher zeroes them)
};
template <typename RI>
int launch_ctc_fused(const CtcFusedArgs<RI>& a, int B, int K, int store_bytes, hipStream_t stream);
// ctc_fusedw.hip: the same schedule for rows of 513..2048 states -- one workgroup of W = 4 / 8 waves per (utterance,
// direction), K = 2 / 4 states per lane; the store is laid out like the narrow kernel's
template <typename RI>
int launch_ctc_fusedw(const CtcFusedArgs<RI>& a, int B, int K, int W, int store_bytes, hipStream_t stream);
size_t ctc_fusedw_sync_bytes(int B);

// K states per lane and W waves per (utterance, pass) for rows of up to max_L states;
// returns K (0 if 2U+1 > 2048: the generic kernels take over), *waves = W; the lattice row stride is 64*W*K
int ctc_lattice_shape(int max_L, int* waves);
template <typename RI>
int launch_ctc_lattice(const CtcLatticeArgs<RI>& a, int B, int K, int W, int lazy, hipStream_t stream);
template <typename RI>
int launch_ctc_grad(const CtcGradArgs<RI>& a, int B, int max_T, hipStream_t stream);
int launch_softmax_rows(const float* x, float* y, int64_t rows, int A, int64_t ld,
                        hipStream_t stream);
// ctc_generic.hip: any label length, any alphabet (rows of more than 2048 states, more than 256 symbols)
int launch_softmax_rows_generic(const float* x, float* y, int64_t rows, int A, int64_t ld, hipStream_t stream);
template <typename RI>
int launch_ctc_generic(const CtcLatticeArgs<RI>& la, const CtcGradArgs<RI>& ga, int B, int max_T, hipStream_t stream);
int launch_argmax_rows(const void* y, int dtype, int32_t* best, int64_t rows, int A, int64_t ld,
                       hipStream_t stream);

// Host-side driver shared by the C ABI and the BRNN engine: lays the descriptors
// out in `ws`, uploads them and launches lattice + grad.
struct CtcPlan {
    int B = 0, A = 0, blank = 0, K = 0, W = 1, lp = 0, max_T = 0;   // lp = 64*W*K
    int lazy = 0;           // SCTC_CTC_LAZY=1: float32 probabilities rescale every 4th frame only
    int generic = 0;        // ctc_generic.hip: 2U+1 > 2048 or A > 256 (or SCTC_CTC_GENERIC=1); lp = round_up(2U+2, 64)
    int fused = 0;          // 1: ctc_fused_kernel (one wave per direction, rows of <= 512 states), 2: ctc_fusedw_kernel (W waves
                            // per direction, rows of <= 2048 states); lat_elems then counts the elements of the ONE packed
                            // row store, store_bytes (4 / 8) each
    int store_bytes = 8;
    int64_t frames = 0;
    int64_t lat_elems = 0;  // elements per lattice (alpha or beta)
    int64_t n_labels = 0;
    size_t bytes = 0;       // workspace bytes for this plan (float64 lattices)
};
int ctc_make_plan(int B, int A, int blank, int dtype, const int32_t* T_b, const int32_t* U_b,
                  CtcPlan* plan);

}  // namespace sctc
