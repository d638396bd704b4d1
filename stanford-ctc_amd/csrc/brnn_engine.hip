// BRNN engine: NNet.costAndGrad (ctc_fast/nnets/brnnet.py:117-249) for a minibatch
// of utterances, orchestrating the fp32 MFMA GEMM, the persistent recurrent kernel
// and the CTC kernels on one HIP stream.  No device allocation happens here: the
// caller provides the flat parameter / gradient buffers and one workspace.
//
// Data layout in HBM (all fp32, row-major, feature dimensions padded to 32):
//   packed time-major minibatch: utterances sorted by length (longest first);
//   frame t of the utterance with rank b is row rowbase[t] + b of every activation /
//   delta matrix ([rows][ld]).  A time step of the recurrence therefore touches one
//   contiguous block of rows, and the time-batched GEMMs see N = sum_b T_b frames.
//   Parameters: tensor i of `stack` at params + offset_i, [rows_p][cols_p], zero padded.
#include <algorithm>
#include <numeric>
#include <vector>

#include "common.h"
#include <dlfcn.h>

#include <mutex>

#include "ctc_kernels.h"
#include "elementwise.h"
#include "gemm_f32.h"
#include "recurrent.h"

namespace sctc {
int ctc_run_batch(const sctc_ctc_batch* bt, const void* probs, void* grad, double* cost,
                  int32_t* skip, void* ws, size_t ws_bytes, hipStream_t stream, void* keep_stage);
void* ctc_new_stage();
void ctc_free_stage(void* p);
}

using namespace sctc;

static constexpr int PAD = 32;
// Row strides are multiples of 64 floats (256 B): with the natural stride of H = 1824
// (7296 B = 57 x 128 B) a GEMM whose A and C matrices share that stride ran 13 % slower
// (measured: 102 vs 115 TFLOP/s), any multiple of 256 B is fine.
static inline int LD(int padded_dim) { return (int)((padded_dim + 63) / 64 * 64); }
static constexpr int CTC_LP_MAX = 2048;  // worst-case lattice row (2U+1 <= 2048)

// GemmArgs::prec of the time-batched contractions for a configured operand_dtype (SCTC_BF16X3: fp32
// operands in memory, split into three bfloat16 terms inside the GEMM -- no shadow copies)
static int fwd_prec(int operand_dtype)
{
    return operand_dtype == SCTC_F16 ? 1 : (operand_dtype == SCTC_BF16X3 ? 3 : 0);
}
static int bwd_prec(int operand_dtype)
{
    return operand_dtype == SCTC_F16 ? 2 : (operand_dtype == SCTC_BF16X3 ? 3 : 0);
}

struct sctc_brnn {
    sctc_brnn_config cfg;
    int D, Dp, H, Hp, A, Ap, NL, TL;
    int64_t maxF;
    int maxB;
    float* params;
    float* grads;
    std::vector<sctc_tensor_info> tinfo;  // stack order: W1,b1,...,W_{NL+1},b_{NL+1},(Wf,Wb)
    int64_t param_elems, param_count;

    // workspace carve-up
    float* X0;
    std::vector<float*> act;  // act[0] = X0, act[i] = output of layer i (i = 1..NL)
    float *Z, *hF, *hB;       // temporal layer: pre-activation, forward / backward states
    float *logits, *probs, *dlogits;
    float *dA, *dBuf, *dF, *dBk;  // deltas: ping-pong pair + recurrent pair
    // 16-bit shadow copies (operand_dtype = SCTC_F16 only; nullptr otherwise): f = float16 (forward
    // operands), b = bfloat16 (backward operands).  Written by the producer of the fp32 matrix.
    std::vector<uint16_t*> act16f, act16b;      // [NL + 1], shadows of act[i]
    uint16_t *hF16b = nullptr, *hB16b = nullptr, *dF16b = nullptr, *dBk16b = nullptr;
    uint16_t *dlogits16 = nullptr, *dA16 = nullptr, *dBuf16 = nullptr;
    uint16_t* W16f = nullptr;                   // float16 copy of the flat parameter buffer (same offsets)
    std::vector<uint16_t*> WT16b;               // [NL + 1]: bfloat16 W^T of layer l, [inp_p][LD(outp)]
    int32_t *d_rowbase, *d_nact, *d_Ts, *d_src_row, *d_idx_lo, *d_idx_hi, *d_xbase;
    void* ctc_ws;
    size_t ctc_ws_bytes;
    void* ctc_ws_ext = nullptr;     // caller-owned replacement (sctc_brnn_set_ctc_workspace): label rows beyond the reserved share
    size_t ctc_ws_ext_bytes = 0;
    float* splitk_ws;
    int64_t splitk_floats;
    float* xbuf;
    unsigned* counters;
    unsigned* rec_debug = nullptr;  // [2 passes][REC_DEBUG_WORDS] step timestamps (SCTC_REC_DEBUG=1)
    int rec_debug_on = 0;
    double* d_cost;   // [maxB] sorted order
    int32_t* d_skip;  // [maxB]
    double* d_cost_out;  // [maxB] caller order
    int32_t* d_skip_out;
    double* d_sumsq;  // [n weight tensors]
    double* sumsq_ws;
    int32_t* d_perm;  // [maxB] rank -> caller index

    // per-call plan (host)
    std::vector<int32_t> order, Ts, rowbase, nact, src_row, idx_lo, idx_hi, xbase;
    int64_t n_xrows = 0;
    int64_t N = 0;
    int B = 0, Tmax = 0;
    int64_t npairs = 0;
    bool pairs_contig = false;

    // recorded on the step's stream when the gradient of parameter tensor i (weights: together
    // with their bias) is final -- data-parallel callers start that tensor's all-reduce then
    std::vector<hipEvent_t> grad_ev;

    // profiling: 0 off; 1 phase timers that synchronise at every phase change (exact, perturbing);
    // 2 asynchronous: one hipEvent per phase change is recorded on the step's stream and resolved by
    // sctc_brnn_phase_ms after the step -- no host sync is added, so it can stay on while a benchmark
    // times its steps
    int profiling = 0;
    std::vector<hipEvent_t> tev;      // [TEV_MAX] timing events (mode 2)
    int tev_phase[128];               // phase that STARTS at event i, -1 = end of the call
    int tev_n = 0;
    hipEvent_t ev[SCTC_N_PHASES + 1][2];
    bool ev_ready = false;
    float phase_ms[SCTC_N_PHASES];
    int rec_sync_mode = 0;
    bool fuse_add = true;      // env SCTC_FUSE_ADD=0: the two sums around the temporal layer by add_kernel (A/B, bit-identity tests)
    int rec_poll_delay = -1;   // env SCTC_REC_POLL_DELAY (s_sleep units before a step's first poll; default by layer size)
    int rec_variant = 0;   // env SCTC_REC_VARIANT: 1 forces the one-workgroup-per-CU recurrent kernel, 3 the per-step fallback
    int rec_force_fallback = 0;   // set while a timed-out step is retried on the per-step fallback
    int rec_path[2] = {0, 0};     // REC_PATH_* of the last forward / BPTT recurrence
    int rec_retries = 0;          // steps of this handle that were re-run after SCTC_ERR_TIMEOUT
    const float* last_delta1 = nullptr;   // delta entering layer 1 (A operand of the dW1 GEMM) of the last backward pass
    // host staging of the CTC descriptors (must outlive the async uploads)
    void* ctc_stage = nullptr;
    PinnedStage plan_stage;    // make_plan's uploads
    std::vector<int32_t> ctc_U, ctc_labels;
    std::vector<int64_t> ctc_frame_off, ctc_label_off;
};

static int weight_index(const sctc_brnn* h, int layer) { return 2 * layer; }      // W_{layer+1}
static int bias_index(const sctc_brnn* h, int layer) { return 2 * layer + 1; }
static int wf_index(const sctc_brnn* h) { return 2 * (h->NL + 1); }
static int wb_index(const sctc_brnn* h) { return 2 * (h->NL + 1) + 1; }

struct Dims {
    int D, Dp, H, Hp, A, Ap, NL, TL;
};

static int derive_dims(const sctc_brnn_config* c, Dims* d)
{
    SCTC_CHECK_ARG(c, "brnn: null config");
    SCTC_CHECK_ARG(c->input_dim >= 1 && c->output_dim >= 2 && c->layer_size >= 1 &&
                       c->num_layers >= 1,
                   "brnn: bad dimensions (inputDim %d outputDim %d layerSize %d numLayers %d)",
                   c->input_dim, c->output_dim, c->layer_size, c->num_layers);
    SCTC_CHECK_ARG(c->max_frames >= 1 && c->max_utts >= 1, "brnn: bad capacity");
    SCTC_CHECK_ARG(c->operand_dtype == SCTC_F32 || c->operand_dtype == SCTC_F16 || c->operand_dtype == SCTC_BF16X3,
                   "brnn: operand_dtype must be SCTC_F32, SCTC_F16 or SCTC_BF16X3 (got %d)", c->operand_dtype);
    d->D = c->input_dim;
    d->H = c->layer_size;
    d->A = c->output_dim;
    d->NL = c->num_layers;
    // brnnet.py:27-30
    d->TL = (c->temporal_layer <= 0 || c->temporal_layer >= c->num_layers) ? -1 : c->temporal_layer;
    d->Dp = (int)round_up(d->D, PAD);
    d->Hp = (int)round_up(d->H, PAD);
    d->Ap = (int)round_up(d->A, PAD);
    if (d->TL > 0) {
        char why[128];
        if (!recurrent_supported(d->Hp, c->max_utts, why, sizeof(why)))
            return set_error(SCTC_ERR_ARG, "brnn: %s", why);
    }
    return SCTC_OK;
}

static void build_tensor_table(const Dims& d, std::vector<sctc_tensor_info>* t, int64_t* elems,
                               int64_t* count)
{
    t->clear();
    int64_t off = 0, cnt = 0;
    for (int l = 0; l <= d.NL; ++l) {
        const int in = l == 0 ? d.D : d.H, inp = l == 0 ? d.Dp : d.Hp;
        const int out = l == d.NL ? d.A : d.H, outp = l == d.NL ? d.Ap : d.Hp;
        sctc_tensor_info w = {off, out, in, LD(inp), 0};
        t->push_back(w);
        off += (int64_t)outp * LD(inp);
        sctc_tensor_info b = {off, out, 1, 1, 1};
        t->push_back(b);
        off += outp;
        cnt += (int64_t)out * in + out;
    }
    if (d.TL > 0) {
        for (int k = 0; k < 2; ++k) {
            sctc_tensor_info w = {off, d.H, d.H, LD(d.Hp), 2};
            t->push_back(w);
            off += (int64_t)d.Hp * LD(d.Hp);
            cnt += (int64_t)d.H * d.H;
        }
    }
    *elems = off;
    *count = cnt;
}

// lays the workspace out; with h == nullptr only measures
static size_t carve(const sctc_brnn_config* c, const Dims& d, sctc_brnn* h, void* ws, size_t bytes)
{
    Arena ar;
    ar.init(ws ? ws : (void*)256, ws ? bytes : (size_t)-1 / 2);
    const int64_t F = c->max_frames;
    const int Bm = c->max_utts;
    auto f = [&](int64_t n) { return ar.take<float>((size_t)n); };
    float* X0 = f(F * LD(d.Dp));
    std::vector<float*> act(d.NL + 1);
    act[0] = X0;
    for (int i = 1; i <= d.NL; ++i) act[i] = f(F * LD(d.Hp));
    float *Z = nullptr, *hF = nullptr, *hB = nullptr;
    if (d.TL > 0) { Z = f(F * LD(d.Hp)); hF = f(F * LD(d.Hp)); hB = f(F * LD(d.Hp)); }
    float* logits = f(F * LD(d.Ap));
    float* probs = f(F * LD(d.Ap));
    float *dlogits = nullptr, *dA = nullptr, *dBuf = nullptr, *dF = nullptr, *dBk = nullptr;
    if (c->train) {
        dlogits = f(F * LD(d.Ap));
        dA = f(F * LD(d.Hp));
        dBuf = f(F * LD(d.Hp));
        if (d.TL > 0) { dF = f(F * LD(d.Hp)); dBk = f(F * LD(d.Hp)); }
    }
    // 16-bit shadows
    const bool h16 = c->operand_dtype == SCTC_F16;
    std::vector<uint16_t*> act16f(d.NL + 1, nullptr), act16b(d.NL + 1, nullptr), WT16b(d.NL + 1, nullptr);
    uint16_t *hF16b = nullptr, *hB16b = nullptr, *dF16b = nullptr, *dBk16b = nullptr, *dlogits16 = nullptr,
             *dA16 = nullptr, *dBuf16 = nullptr, *W16f = nullptr;
    if (h16) {
        auto hh = [&](int64_t n) { return ar.take<uint16_t>((size_t)n); };
        int64_t pe = 0, pc = 0;
        {
            std::vector<sctc_tensor_info> t;
            build_tensor_table(d, &t, &pe, &pc);
        }
        W16f = hh(pe);
        for (int i = 0; i <= d.NL; ++i) {
            const int dim = i == 0 ? d.Dp : d.Hp;
            act16f[i] = hh(F * LD(dim));
            if (c->train) act16b[i] = hh(F * LD(dim));
        }
        if (c->train) {
            for (int l = 1; l <= d.NL; ++l) {
                const int outp = l == d.NL ? d.Ap : d.Hp;
                WT16b[l] = hh((int64_t)d.Hp * LD(outp));
            }
            dlogits16 = hh(F * LD(d.Ap));
            dA16 = hh(F * LD(d.Hp));
            dBuf16 = hh(F * LD(d.Hp));
            if (d.TL > 0) {
                hF16b = hh(F * LD(d.Hp)); hB16b = hh(F * LD(d.Hp));
                dF16b = hh(F * LD(d.Hp)); dBk16b = hh(F * LD(d.Hp));
            }
        }
    }
    int32_t* d_rowbase = ar.take<int32_t>(F);
    int32_t* d_nact = ar.take<int32_t>(F);
    int32_t* d_Ts = ar.take<int32_t>(Bm);
    int32_t* d_src_row = ar.take<int32_t>(F);
    int32_t* d_idx_lo = ar.take<int32_t>(F);
    int32_t* d_idx_hi = ar.take<int32_t>(F);
    int32_t* d_xbase = ar.take<int32_t>(F);
    int32_t* d_perm = ar.take<int32_t>(Bm);
    // CTC workspace, worst case: every frame carries a CTC_LP_MAX-wide lattice row (x2)
    size_t ctc_bytes = 0;
    void* ctc_ws = nullptr;
    if (c->train) {
        ctc_bytes = align256(sizeof(CtcUtt) * Bm) + align256(sizeof(int32_t) * (2 * F + (int64_t)Bm * (d.A + 1))) +
                    align256(sizeof(double) * 2 * Bm) + align256(sizeof(int32_t) * 2 * Bm) +
                    2 * align256(sizeof(double) * F * CTC_LP_MAX) +
                    align256(sizeof(double) * 4 * (size_t)Bm * CTC_LP_MAX);   // ctc_generic.hip's row scratch
        ctc_ws = ar.take<char>(ctc_bytes);
    }
    // split-K partials: worst case over the weight-gradient GEMMs
    int64_t sk = 0;
    if (c->train) {
        for (int l = 0; l <= d.NL; ++l) {
            const int inp = l == 0 ? d.Dp : d.Hp, outp = l == d.NL ? d.Ap : d.Hp;
            int sp = 1;
            sk = std::max<int64_t>(sk, gemm_plan_splits(outp, inp, (int)F, &sp, bwd_prec(c->operand_dtype), c->operand_dtype == SCTC_F16));
        }
        if (d.TL > 0) {
            int sp = 1;
            sk = std::max<int64_t>(sk, gemm_plan_splits(d.Hp, d.Hp, (int)F, &sp, bwd_prec(c->operand_dtype), c->operand_dtype == SCTC_F16));
        }
    }
    // small minibatches (few row tiles) split K in the forward / delta-propagation GEMMs as well:
    // blocks x splits stays below ~2 rounds of resident blocks, i.e. <= 2*1024 tiles of 128x128
    sk = std::max<int64_t>(sk, std::min<int64_t>((int64_t)2 * 1024 * 128 * 128,
                                                 (int64_t)64 * F * (std::max(d.Hp, d.Ap) + 1)));
    float* splitk_ws = sk ? f(sk) : nullptr;
    float* xbuf = d.TL > 0 ? f((int64_t)recurrent_xbuf_floats(d.Hp, recurrent_xrows_bound(F, F))) : nullptr;
    unsigned* counters = ar.take<unsigned>(REC_COUNTER_WORDS);
    unsigned* rec_debug = ar.take<unsigned>(2 * REC_DEBUG_WORDS);
    double* d_cost = ar.take<double>(Bm);
    int32_t* d_skip = ar.take<int32_t>(Bm);
    double* d_cost_out = ar.take<double>(Bm);
    int32_t* d_skip_out = ar.take<int32_t>(Bm);
    double* d_sumsq = ar.take<double>(d.NL + 3);
    double* sumsq_ws = ar.take<double>(sumsq_ws_bytes() / sizeof(double));
    if (h && !ar.overflow) {
        h->X0 = X0; h->act = act; h->Z = Z; h->hF = hF; h->hB = hB;
        h->logits = logits; h->probs = probs; h->dlogits = dlogits;
        h->dA = dA; h->dBuf = dBuf; h->dF = dF; h->dBk = dBk;
        h->d_rowbase = d_rowbase; h->d_nact = d_nact; h->d_Ts = d_Ts; h->d_src_row = d_src_row;
        h->d_idx_lo = d_idx_lo; h->d_idx_hi = d_idx_hi; h->d_perm = d_perm; h->d_xbase = d_xbase;
        h->ctc_ws = ctc_ws; h->ctc_ws_bytes = ctc_bytes;
        h->splitk_ws = splitk_ws; h->splitk_floats = sk;
        h->xbuf = xbuf; h->counters = counters; h->rec_debug = rec_debug;
        h->act16f = act16f; h->act16b = act16b; h->WT16b = WT16b; h->W16f = W16f;
        h->hF16b = hF16b; h->hB16b = hB16b; h->dF16b = dF16b; h->dBk16b = dBk16b;
        h->dlogits16 = dlogits16; h->dA16 = dA16; h->dBuf16 = dBuf16;
        h->d_cost = d_cost; h->d_skip = d_skip; h->d_cost_out = d_cost_out;
        h->d_skip_out = d_skip_out; h->d_sumsq = d_sumsq; h->sumsq_ws = sumsq_ws;
    }
    return ar.overflow && ws ? 0 : ar.used;
}

// ------------------------------------------------------------------ per-call plan

static int make_plan(sctc_brnn* h, const sctc_minibatch* mb, bool need_labels, hipStream_t stream)
{
    SCTC_CHECK_ARG(mb && mb->T_b && mb->feats_dev, "brnn: null minibatch field");
    SCTC_CHECK_ARG(mb->B >= 1 && mb->B <= h->maxB, "brnn: %d utterances, capacity %d", mb->B,
                   h->maxB);
    if (need_labels) SCTC_CHECK_ARG(mb->labels && mb->U_b, "brnn: labels required for training");
    const int B = mb->B;
    int64_t N = 0;
    for (int b = 0; b < B; ++b) {
        SCTC_CHECK_ARG(mb->T_b[b] >= 1, "brnn: utterance %d has no frames", b);
        N += mb->T_b[b];
    }
    // setViews(): "Batch size exceeds max batch", brnnet.py:100
    SCTC_CHECK_ARG(N <= h->maxF, "Batch size exceeds max batch (%lld frames > %lld)", (long long)N,
                   (long long)h->maxF);
    h->B = B;
    h->N = N;
    h->order.resize(B);
    std::iota(h->order.begin(), h->order.end(), 0);
    std::stable_sort(h->order.begin(), h->order.end(),
                     [&](int a, int b) { return mb->T_b[a] > mb->T_b[b]; });
    h->Ts.resize(B);
    std::vector<int64_t> foff(B);
    {
        int64_t o = 0;
        for (int b = 0; b < B; ++b) { foff[b] = o; o += mb->T_b[b]; }
    }
    for (int r = 0; r < B; ++r) h->Ts[r] = mb->T_b[h->order[r]];
    const int Tmax = h->Ts[0];
    h->Tmax = Tmax;
    h->rowbase.assign(Tmax, 0);
    h->nact.assign(Tmax, 0);
    h->xbase.assign(Tmax, 0);
    {
        int na = B;
        int64_t base = 0, xb = 0;
        for (int t = 0; t < Tmax; ++t) {
            while (na > 0 && h->Ts[na - 1] <= t) --na;
            h->nact[t] = na;
            h->rowbase[t] = (int32_t)base;
            h->xbase[t] = (int32_t)xb;     // exchange rows: every step's block starts on 256 B (whole tiles from 17 utterances on)
            base += na;
            xb += recurrent_step_xrows(na);
        }
        h->n_xrows = xb;
    }
    h->src_row.resize(N);
    h->idx_lo.clear();
    h->idx_hi.clear();
    for (int t = 0; t < Tmax; ++t)
        for (int r = 0; r < h->nact[t]; ++r) {
            const int row = h->rowbase[t] + r;
            h->src_row[row] = (int32_t)(foff[h->order[r]] + t);
            if (t >= 1) {
                h->idx_hi.push_back(row);
                h->idx_lo.push_back(h->rowbase[t - 1] + r);
            }
        }
    h->npairs = (int64_t)h->idx_hi.size();
    // equal-length minibatches pair two CONTIGUOUS row ranges: then the recurrent weight gradient
    // needs no row gather (plain pointers, 10 % faster GEMM)
    h->pairs_contig = h->npairs > 0;
    for (int64_t k = 1; k < h->npairs && h->pairs_contig; ++k)
        h->pairs_contig = h->idx_hi[k] == h->idx_hi[0] + k && h->idx_lo[k] == h->idx_lo[0] + k;
    // the plan goes up through pinned staging (common.h, PinnedStage): src_row / idx_lo / idx_hi are N ints each
    const size_t I = sizeof(int32_t);
    const size_t total = I * (3 * (size_t)Tmax + 2 * (size_t)B + (size_t)N + 2 * (size_t)h->npairs) + 8 * 64;
    char* pin = static_cast<char*>(h->plan_stage.acquire(total));
    size_t off = 0;
    auto up = [&](void* dev, const void* src, size_t bytes) -> int {
        if (bytes == 0) return SCTC_OK;
        const void* from = src;
        if (pin) {
            memcpy(pin + off, src, bytes);
            from = pin + off;
            off += (bytes + 63) & ~(size_t)63;
        }
        SCTC_HIP_TRY(hipMemcpyAsync(dev, from, bytes, hipMemcpyHostToDevice, stream));
        return SCTC_OK;
    };
    SCTC_TRY(up(h->d_rowbase, h->rowbase.data(), I * Tmax));
    SCTC_TRY(up(h->d_nact, h->nact.data(), I * Tmax));
    SCTC_TRY(up(h->d_xbase, h->xbase.data(), I * Tmax));
    SCTC_TRY(up(h->d_Ts, h->Ts.data(), I * B));
    SCTC_TRY(up(h->d_src_row, h->src_row.data(), I * N));
    SCTC_TRY(up(h->d_perm, h->order.data(), I * B));
    SCTC_TRY(up(h->d_idx_lo, h->idx_lo.data(), I * h->npairs));
    SCTC_TRY(up(h->d_idx_hi, h->idx_hi.data(), I * h->npairs));
    if (pin) SCTC_HIP_TRY(h->plan_stage.uploaded(stream));
    return SCTC_OK;
}

// ------------------------------------------------------------------ phases / profiling

struct PhaseTimer {
    sctc_brnn* h;
    hipStream_t s;
    int cur = -1;
    void mark(int phase)
    {
        if (h->tev_n >= (int)h->tev.size()) return;
        (void)hipEventRecord(h->tev[h->tev_n], s);
        h->tev_phase[h->tev_n++] = phase;
    }
    void begin(int phase)
    {
        if (!h->profiling) return;
        if (h->profiling == 2) { mark(phase); return; }
        end();
        cur = phase;
        (void)hipEventRecord(h->ev[SCTC_N_PHASES][0], s);
    }
    void end()
    {
        if (h->profiling == 2) { mark(-1); return; }
        if (!h->profiling || cur < 0) return;
        (void)hipEventRecord(h->ev[SCTC_N_PHASES][1], s);
        (void)hipEventSynchronize(h->ev[SCTC_N_PHASES][1]);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, h->ev[SCTC_N_PHASES][0], h->ev[SCTC_N_PHASES][1]);
        h->phase_ms[cur] += ms;
        cur = -1;
    }
};

static GemmArgs gemm_defaults()
{
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.splits = 1;
    g.add_scale = 0.f;
    return g;
}

// deterministic split-K when the output has too few tiles to fill the machine (small minibatch)
static void maybe_split(const sctc_brnn* h, GemmArgs& g)
{
    int sp = 1;
    const int64_t need = gemm_plan_splits(g.M, g.N, g.K, &sp, g.prec, g.in16);
    if (sp > 1 && h->splitk_ws && need <= h->splitk_floats) {
        g.splits = sp;
        g.splitk_ws = h->splitk_ws;
    }
}

static const float* tensor_ptr(const sctc_brnn* h, const float* base, int idx)
{
    return base + h->tinfo[idx].offset;
}

// ------------------------------------------------------------------ forward

static int run_forward(sctc_brnn* h, const sctc_minibatch* mb, hipStream_t s, PhaseTimer& pt)
{
    const int64_t N = h->N;
    pt.begin(SCTC_PHASE_OTHER);
    // brnnet.py:136 hActs[0] <- data, here also the re-ordering into the packed layout
    const bool h16 = h->cfg.operand_dtype == SCTC_F16;
    if (h16) {
        SCTC_TRY(launch_gather_rows16(h->X0, h->act16f[0], h->cfg.train ? h->act16b[0] : nullptr, LD(h->Dp),
                                      mb->feats_dev, h->D, h->d_src_row, N, h->D, s));
        // parameters change between calls: their 16-bit copies are refreshed per call (35 M elements)
        SCTC_TRY(launch_cvt16(h->params, h->W16f, nullptr, round_up(h->param_elems, 4), s));
    } else {
        SCTC_TRY(launch_gather_rows(h->X0, LD(h->Dp), mb->feats_dev, h->D, h->d_src_row, N, h->D, s));
    }
    bool fused_sum = false;   // the sum of the two recurrent outputs is left to the next GEMM (native fp32 only)
    for (int i = 1; i <= h->NL + 1; ++i) {
        pt.begin(SCTC_PHASE_FWD_GEMM);
        const int l = i - 1;
        const sctc_tensor_info& wi = h->tinfo[weight_index(h, l)];
        const int inp = l == 0 ? h->Dp : h->Hp;
        const int outp = i == h->NL + 1 ? h->Ap : h->Hp;
        GemmArgs g = gemm_defaults();
        g.A = h->act[i - 1];           // [N][inp]
        g.lda = LD(inp);
        g.a_kcontig = 1;
        g.B = h->params + wi.offset;   // W [outp][inp]: B(k,n) = W[n][k]
        g.ldb = LD(inp);
        g.b_kcontig = 1;
        g.M = (int)N;
        g.N = outp;
        g.K = inp;
        g.bias = tensor_ptr(h, h->params, bias_index(h, l));   // add_col_vec, brnnet.py:141
        float* dst = i == h->NL + 1 ? h->logits : (i == h->TL ? h->Z : h->act[i]);
        g.C = dst;
        g.ldc = LD(outp);
        g.relu = (i <= h->NL && i != h->TL) ? 1 : 0;             // brnnet.py:155-157
        g.prec = fwd_prec(h->cfg.operand_dtype);                 // SCTC_F16 -> forward: float16 operands
        if (fused_sum) {
            // hActs[TL] = hActsFor + hActsBack (brnnet.py:153) is formed while this GEMM stages its A operand;
            // the blocks of the first N tile store it to act[TL] for the weight gradient of this layer
            g.A = h->hF;
            g.A2 = h->hB;
            g.a_sum = h->act[i - 1];
            fused_sum = false;
        }
        if (h16) {
            g.in16 = 1;
            g.A = reinterpret_cast<const float*>(h->act16f[i - 1]);
            g.B = reinterpret_cast<const float*>(h->W16f + wi.offset);
            if (dst != h->Z && dst != h->logits) {               // a hidden layer's output: shadows
                g.C16a = h->act16f[i];
                g.C16b = h->cfg.train ? h->act16b[i] : nullptr;
                g.ldc16 = LD(outp);
                // "fp16 activations": a ReLU layer's output is read by the next layer's GEMM (float16),
                // by this layer's weight gradient (bfloat16) and by the ReLU mask of the delta GEMM (its
                // sign) -- nobody needs the fp32 copy, and not writing it halves the GEMM's stores
                g.skip_c32 = 1;
            }
        }
        maybe_split(h, g);
        SCTC_TRY(launch_gemm_f32(g, s));
        if (i == h->TL) {
            pt.begin(SCTC_PHASE_FWD_REC);
            RecArgs r;
            memset(&r, 0, sizeof(r));
            r.W[0] = tensor_ptr(h, h->params, wf_index(h));
            r.W[1] = tensor_ptr(h, h->params, wb_index(h));
            r.ldw = LD(h->Hp);
            r.transpose = 0;
            r.descending[0] = 0;
            r.descending[1] = 1;
            r.pre[0] = r.pre[1] = h->Z;
            r.act[0] = r.act[1] = nullptr;
            r.out[0] = h->hF;
            r.out[1] = h->hB;
            r.ld = LD(h->Hp);
            r.Hp = h->Hp;
            r.B = h->B;
            r.Tmax = h->Tmax;
            r.rowbase = h->d_rowbase;
            r.T_b = h->d_Ts;
            r.max_act = h->cfg.max_act;
            r.xbuf = h->xbuf;
            r.xbase = h->d_xbase;
            r.n_xrows = (int)h->n_xrows;
            r.counters = h->counters;
            r.sync_mode = h->rec_sync_mode;
            r.variant = h->rec_force_fallback ? 3 : h->rec_variant;
            r.poll_delay = h->rec_poll_delay;
            r.debug = h->rec_debug_on ? h->rec_debug : nullptr;
            r.prec16 = h->cfg.operand_dtype == SCTC_F16;
            r.T_host = h->Ts.data();
            SCTC_TRY(launch_recurrent(r, s, &h->rec_path[0]));
            pt.begin(SCTC_PHASE_OTHER);
            // hActs[i] = hActsFor + hActsBack, brnnet.py:153
            if (h16) {
                // one pass: the 16-bit shadows of the sum (nobody reads its fp32 copy in this configuration)
                // and, when training, the bfloat16 copies of hF / hB (B operands of the recurrent weight gradient)
                SCTC_TRY(launch_add16(nullptr, h->hF, h->hB, h->act16f[i], h->cfg.train ? h->act16b[i] : nullptr,
                                      N * LD(h->Hp), s, h->cfg.train ? h->hF16b : nullptr,
                                      h->cfg.train ? h->hB16b : nullptr));
            } else if (g.prec == 0 && h->fuse_add) {
                fused_sum = true;
            } else {
                SCTC_TRY(launch_add(h->act[i], h->hF, h->hB, N * LD(h->Hp), s));
            }
        }
    }
    pt.begin(SCTC_PHASE_CTC);
    SCTC_TRY(launch_softmax_rows(h->logits, h->probs, N, h->A, LD(h->Ap), s));  // brnnet.py:161-168
    return SCTC_OK;
}

static int check_recurrent_error(sctc_brnn* h, hipStream_t s)
{
    if (h->TL <= 0) return SCTC_OK;
    unsigned e = 0;
    SCTC_HIP_TRY(hipMemcpyAsync(&e, h->counters + 2, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    SCTC_HIP_TRY(hipStreamSynchronize(s));
    if (e != 0) {
        // whoever else holds compute units of this device will still be there on the next launch:
        // from now on every persistent launch of this process takes the inter-process lease
        if (!recurrent_shared_device_mode())
            fprintf(stderr, "sctc: a persistent recurrent launch timed out (its workgroups were not co-resident): "
                            "shared-device mode is on for the rest of this process\n");
        recurrent_set_shared_device_mode(1);
        return set_error(SCTC_ERR_TIMEOUT, "recurrent kernel: a persistent launch gave up waiting for "
                         "its peers (the %d workgroups of a pass were not co-resident: device shared "
                         "with another process or CU-masked); shared-device mode is now on",
                         2 * (h->Hp / 16));
    }
    return SCTC_OK;
}

// A timed-out step is re-run: first under the inter-process lease (check_recurrent_error has just
// switched shared-device mode on), then on the per-step fallback, which cannot time out.  Not
// possible when the step accumulates into the gradients (the failed attempt has already added
// garbage): that error goes to the caller.
template <typename Step>
static int with_retry(sctc_brnn* h, bool retry_ok, Step step)
{
    int rc = step();
    for (int attempt = 0; rc == SCTC_ERR_TIMEOUT && retry_ok && attempt < 2; ++attempt) {
        if (getenv("SCTC_VERBOSE"))
            fprintf(stderr, "sctc: %s -- retrying the step %s\n", err_buf(),
                    attempt == 0 ? "under the device lease" : "on the per-step fallback");
        ++h->rec_retries;
        h->rec_force_fallback = attempt == 1;
        rc = step();
        h->rec_force_fallback = 0;
    }
    return rc;
}

// ------------------------------------------------------------------ CTC + backward

static int run_ctc(sctc_brnn* h, const sctc_minibatch* mb, hipStream_t s)
{
    const int B = h->B;
    std::vector<int32_t>& U = h->ctc_U;
    std::vector<int32_t>& labels = h->ctc_labels;
    std::vector<int64_t>& frame_off = h->ctc_frame_off;
    std::vector<int64_t>& label_off = h->ctc_label_off;
    U.assign(B, 0);
    labels.clear();
    frame_off.assign(B, 0);
    label_off.assign(B, 0);
    std::vector<int64_t> src_off(B);
    {
        int64_t o = 0;
        for (int b = 0; b < B; ++b) { src_off[b] = o; o += mb->U_b[b]; }
    }
    for (int r = 0; r < B; ++r) {
        const int b = h->order[r];
        U[r] = mb->U_b[b];
        frame_off[r] = r;  // rank in the packed layout
        label_off[r] = (int64_t)labels.size();
        labels.insert(labels.end(), mb->labels + src_off[b], mb->labels + src_off[b] + mb->U_b[b]);
    }
    sctc_ctc_batch bt;
    bt.B = B;
    bt.A = h->A;
    bt.blank = 0;  // brnnet.py:175-176 blank=0
    bt.dtype = SCTC_F32;
    bt.ld = LD(h->Ap);
    bt.T_b = h->Ts.data();
    bt.U_b = U.data();
    bt.frame_off = frame_off.data();
    bt.labels = labels.data();
    bt.label_off = label_off.data();
    bt.rowbase_dev = h->d_rowbase;
    if (!h->ctc_stage) h->ctc_stage = ctc_new_stage();
    return ctc_run_batch(&bt, h->probs, h->dlogits, h->d_cost, h->d_skip, h->ctc_ws_ext ? h->ctc_ws_ext : h->ctc_ws,
                         h->ctc_ws_ext ? h->ctc_ws_ext_bytes : h->ctc_ws_bytes, s, h->ctc_stage);
}

__global__ void unpermute_results_kernel(const double* cost, const int32_t* skip,
                                         const int32_t* perm, int B, double* cost_out,
                                         int32_t* skip_out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < B) {
        cost_out[perm[r]] = cost[r];
        skip_out[perm[r]] = skip[r];
    }
}

static int run_backward(sctc_brnn* h, int flags, hipStream_t s, PhaseTimer& pt)
{
    const int64_t N = h->N;
    const int acc = (flags & SCTC_FLAG_ACCUMULATE) ? 1 : 0;
    const float reg = (flags & SCTC_FLAG_NO_REG_GRAD) ? 0.f : h->cfg.reg;
    const int bprec = bwd_prec(h->cfg.operand_dtype);             // SCTC_F16 -> backward: bfloat16 operands
    const bool h16 = h->cfg.operand_dtype == SCTC_F16;            // 16-bit shadow operands
    const float* d_in = h->dlogits;
    const uint16_t* d_in16 = h->dlogits16;
    int d_in_ld = LD(h->Ap);
    float* bufs[2] = {h->dA, h->dBuf};
    uint16_t* bufs16[2] = {h->dA16, h->dBuf16};
    if (h16) {
        SCTC_TRY(launch_cvt16(h->dlogits, nullptr, h->dlogits16, N * LD(h->Ap), s));
        for (int l = 1; l <= h->NL; ++l) {        // W^T as the K-contiguous B operand of the delta GEMMs
            const sctc_tensor_info& wl = h->tinfo[weight_index(h, l)];
            const int outp = l == h->NL ? h->Ap : h->Hp;
            SCTC_TRY(launch_transpose_bf16(h->params + wl.offset, LD(h->Hp), h->WT16b[l], LD(outp), outp, h->Hp, s));
        }
    }
    int which = 0;
    bool fused_sum = false;   // the sum of the two BPTT outputs is left to the next weight gradient (native fp32 only)
    for (int i = h->NL; i >= 0; --i) {          // brnnet.py:191-243
        pt.begin(SCTC_PHASE_BWD_GEMM);
        const sctc_tensor_info& wi = h->tinfo[weight_index(h, i)];
        const int inp = i == 0 ? h->Dp : h->Hp;
        const int outp = i == h->NL ? h->Ap : h->Hp;
        const float* W = h->params + wi.offset;
        // dW = deltasIn . hActs[i]^T (+ reg*W), brnnet.py:196-198
        {
            GemmArgs g = gemm_defaults();
            g.A = d_in;             // A(m=out, k=frame) = d_in[frame][out]
            g.lda = d_in_ld;
            g.a_kcontig = 0;
            if (fused_sum) {
                // deltasOut = deltasFor + deltasBack (brnnet.py:233) is formed while this GEMM stages its A
                // operand (its column sums, the bias gradient, are taken of the sum); the blocks of the first
                // N tile store it to d_in for the delta GEMM below
                g.A = h->dF;
                g.A2 = h->dBk;
                g.a_sum = const_cast<float*>(d_in);
                fused_sum = false;
            }
            g.B = h->act[i];        // B(k=frame, n=in) = act[frame][in]
            g.ldb = LD(inp);
            g.b_kcontig = 0;
            g.M = outp;
            g.N = inp;
            g.K = (int)N;
            g.C = h->grads + wi.offset;
            g.ldc = LD(inp);
            g.accumulate = acc;
            if (reg > 0.f) { g.addend = W; g.ldadd = LD(inp); g.add_scale = reg; }
            // db = deltasIn.sum(axis=1), brnnet.py:200: column sums of the A operand, fused
            g.colsum_a = h->grads + h->tinfo[bias_index(h, i)].offset;
            g.splitk_ws = h->splitk_ws;
            g.prec = bprec;
            if (h16) {
                g.in16 = 1;
                g.A = reinterpret_cast<const float*>(d_in16);
                g.B = reinterpret_cast<const float*>(h->act16b[i]);
            }
            int splits = 1;
            gemm_plan_splits(g.M, g.N, g.K, &splits, g.prec, g.in16);
            g.splits = splits;
            SCTC_TRY(launch_gemm_f32(g, s));
            // The weight gradients of the layers above the temporal layer (loop indices i >= TL: W_{TL+1}
            // .. W_{NL+1}) finish BEFORE the BPTT recurrence starts.  A collective
            // started on their event would hold compute units while the persistent BPTT grid is being
            // placed (all of its workgroups must be resident at once; the two-chain kernel takes every
            // register of 200 CUs and half of the other 56): their events are recorded after BPTT has retired instead (below), so a
            // data-parallel caller's all-reduces only ever overlap the time-batched GEMMs.
            if (!(h->TL > 0 && i >= h->TL))
                SCTC_HIP_TRY(hipEventRecord(h->grad_ev[weight_index(h, i)], s));
        }
        if (i == 0) break;
        pt.begin(SCTC_PHASE_BWD_GEMM);
        // deltasOut = W^T deltasIn, brnnet.py:204  (+ sign(hActs[i]) mask, :235-237)
        float* d_out = bufs[which];
        {
            GemmArgs g = gemm_defaults();
            g.A = d_in;             // [N][outp], K = outp
            g.lda = d_in_ld;
            g.a_kcontig = 1;
            g.B = W;                // B(k=out, n=in) = W[out][in]
            g.ldb = LD(inp);
            g.b_kcontig = 0;
            g.M = (int)N;
            g.N = inp;
            g.K = outp;
            g.C = d_out;
            g.ldc = LD(inp);
            if (i != h->TL) { g.mask = h->act[i]; g.ldmask = LD(h->Hp); }
            g.prec = bprec;
            if (h16) {
                g.in16 = 1;
                g.A = reinterpret_cast<const float*>(d_in16);
                g.B = reinterpret_cast<const float*>(h->WT16b[i]);   // B(k = out, n = in) = W^T[in][out]
                g.ldb = LD(outp);
                g.b_kcontig = 1;
                g.C16b = bufs16[which];
                g.ldc16 = LD(inp);
                if (i != h->TL) {
                    // the layer's activations exist as 16-bit shadows only (run_forward): the mask is their
                    // sign; and this delta is read by 16-bit GEMM operands only -- the BPTT recurrence, which
                    // takes the fp32 delta as its additive term, follows the dgrad of layer TL alone
                    g.mask = nullptr;
                    g.mask16 = h->act16b[i];
                    g.ldmask16 = LD(h->Hp);
                    g.skip_c32 = 1;
                }
            }
            maybe_split(h, g);
            SCTC_TRY(launch_gemm_f32(g, s));
        }
        if (i == h->TL) {
            pt.begin(SCTC_PHASE_BWD_REC);
            RecArgs r;
            memset(&r, 0, sizeof(r));
            r.W[0] = tensor_ptr(h, h->params, wf_index(h));
            r.W[1] = tensor_ptr(h, h->params, wb_index(h));
            r.ldw = LD(h->Hp);
            r.transpose = 1;
            r.descending[0] = 1;   // deltasFor runs from T-1 down, brnnet.py:217-218
            r.descending[1] = 0;   // deltasBack runs from 0 up,    brnnet.py:219-220
            r.pre[0] = r.pre[1] = d_out;
            r.act[0] = h->hF;
            r.act[1] = h->hB;
            r.out[0] = h->dF;
            r.out[1] = h->dBk;
            r.ld = LD(h->Hp);
            r.Hp = h->Hp;
            r.B = h->B;
            r.Tmax = h->Tmax;
            r.rowbase = h->d_rowbase;
            r.T_b = h->d_Ts;
            r.max_act = h->cfg.max_act;
            r.xbuf = h->xbuf;
            r.xbase = h->d_xbase;
            r.n_xrows = (int)h->n_xrows;
            r.counters = h->counters;
            r.sync_mode = h->rec_sync_mode;
            r.variant = h->rec_force_fallback ? 3 : h->rec_variant;
            r.poll_delay = h->rec_poll_delay;
            r.debug = h->rec_debug_on ? h->rec_debug + REC_DEBUG_WORDS : nullptr;
            r.prec16 = h->cfg.operand_dtype == SCTC_F16;
            r.T_host = h->Ts.data();
            SCTC_TRY(launch_recurrent(r, s, &h->rec_path[1]));
            for (int l = h->NL; l >= h->TL; --l)     // the held-back events of the layers above (see there)
                SCTC_HIP_TRY(hipEventRecord(h->grad_ev[weight_index(h, l)], s));
            pt.begin(SCTC_PHASE_BWD_GEMM);
            // deltasOut = deltasFor + deltasBack, brnnet.py:233 -- in the fp16 configuration first, in the
            // pass that also writes the bfloat16 copies of dF / dBk (A operands of the recurrent weight
            // gradient); its fp32 sum has no reader (the delta GEMM below takes the bfloat16 shadow)
            if (h16) SCTC_TRY(launch_add16(nullptr, h->dF, h->dBk, nullptr, bufs16[which], N * LD(h->Hp), s,
                                           h->dF16b, h->dBk16b));
            // dwtf = deltasFor[:,1:T] . hActsFor[:,0:T-1]^T ; dwtb = deltasBack[:,0:T-1] . hActsBack[:,1:T]^T
            // (brnnet.py:227-230) over the (lo = frame t, hi = frame t+1) row pairs of every utterance
            for (int k = 0; k < 2; ++k) {
                const sctc_tensor_info& ri = h->tinfo[k == 0 ? wf_index(h) : wb_index(h)];
                GemmArgs g = gemm_defaults();
                g.A = k == 0 ? h->dF : h->dBk;
                g.lda = LD(h->Hp);
                g.a_kcontig = 0;
                g.B = k == 0 ? h->hF : h->hB;
                g.ldb = LD(h->Hp);
                g.b_kcontig = 0;
                const uint16_t* a16 = k == 0 ? h->dF16b : h->dBk16b;
                const uint16_t* b16 = k == 0 ? h->hF16b : h->hB16b;
                if (h->pairs_contig) {
                    const int64_t hi0 = h->idx_hi[0], lo0 = h->idx_lo[0];
                    g.A += (k == 0 ? hi0 : lo0) * LD(h->Hp);
                    g.B += (k == 0 ? lo0 : hi0) * LD(h->Hp);
                    a16 += (k == 0 ? hi0 : lo0) * LD(h->Hp);
                    b16 += (k == 0 ? lo0 : hi0) * LD(h->Hp);
                } else {
                    g.idx_a = k == 0 ? h->d_idx_hi : h->d_idx_lo;
                    g.idx_b = k == 0 ? h->d_idx_lo : h->d_idx_hi;
                }
                g.M = h->Hp;
                g.N = h->Hp;
                g.K = (int)h->npairs;
                g.C = h->grads + ri.offset;
                g.ldc = LD(h->Hp);
                g.accumulate = acc;
                if (reg > 0.f) {           // brnnet.py:244-247
                    g.addend = h->params + ri.offset;
                    g.ldadd = LD(h->Hp);
                    g.add_scale = reg;
                }
                g.splitk_ws = h->splitk_ws;
                g.prec = bprec;
                if (h16) {
                    g.in16 = 1;
                    g.A = reinterpret_cast<const float*>(a16);
                    g.B = reinterpret_cast<const float*>(b16);
                }
                int splits = 1;
                gemm_plan_splits(g.M, g.N, std::max(g.K, 1), &splits, g.prec, g.in16 && !g.idx_a);
                g.splits = splits;
                SCTC_TRY(launch_gemm_f32(g, s));
                SCTC_HIP_TRY(hipEventRecord(h->grad_ev[k == 0 ? wf_index(h) : wb_index(h)], s));
            }
            // deltasOut = deltasFor + deltasBack, brnnet.py:233
            pt.begin(SCTC_PHASE_OTHER);
            if (!h16) {
                if (bprec == 0 && h->fuse_add) fused_sum = true;
                else SCTC_TRY(launch_add(d_out, h->dF, h->dBk, N * LD(h->Hp), s));
            }
        }
        d_in = d_out;
        d_in16 = bufs16[which];
        d_in_ld = LD(h->Hp);
        which ^= 1;
    }
    h->last_delta1 = d_in;
    return SCTC_OK;
}

static int run_cost_and_grad(sctc_brnn* h, const sctc_minibatch* mb, int flags, hipStream_t s,
                             bool* all_skipped)
{
    SCTC_CHECK_ARG(h && h->cfg.train, "brnn: model was created with train=0");
    PhaseTimer pt{h, s};
    if (h->profiling) memset(h->phase_ms, 0, sizeof(h->phase_ms));
    h->tev_n = 0;
    SCTC_TRY(make_plan(h, mb, true, s));
    if (h->TL > 0) SCTC_TRY(recurrent_clear_error(h->counters, s));   // sticky for the whole step
    SCTC_TRY(run_forward(h, mb, s, pt));
    SCTC_TRY(run_ctc(h, mb, s));
    pt.begin(SCTC_PHASE_OTHER);
    hipLaunchKernelGGL(unpermute_results_kernel, dim3((h->B + 63) / 64), dim3(64), 0, s, h->d_cost,
                       h->d_skip, h->d_perm, h->B, h->d_cost_out, h->d_skip_out);
    *all_skipped = false;
    if (flags & SCTC_FLAG_SYNC_SKIP) {
        std::vector<int32_t> sk(h->B);
        SCTC_HIP_TRY(hipMemcpyAsync(sk.data(), h->d_skip, sizeof(int32_t) * h->B,
                                    hipMemcpyDeviceToHost, s));
        SCTC_HIP_TRY(hipStreamSynchronize(s));
        bool all = true;
        for (int v : sk) all = all && v != 0;
        *all_skipped = all;
    }
    if (!*all_skipped) SCTC_TRY(run_backward(h, flags, s, pt));
    pt.end();
    return SCTC_OK;
}

extern "C" {

int sctc_brnn_query(const sctc_brnn_config* cfg, sctc_brnn_sizes* out)
{
    SCTC_CHECK_ARG(out, "brnn_query: null output");
    Dims d;
    SCTC_TRY(derive_dims(cfg, &d));
    std::vector<sctc_tensor_info> t;
    build_tensor_table(d, &t, &out->param_elems, &out->param_count);
    out->n_tensors = (int32_t)t.size();
    out->workspace_bytes = carve(cfg, d, nullptr, nullptr, 0);
    return SCTC_OK;
}

int sctc_brnn_create(const sctc_brnn_config* cfg, float* params_dev, float* grads_dev,
                     void* workspace_dev, size_t workspace_bytes, sctc_brnn_t* out)
{
    SCTC_CHECK_ARG(out && params_dev && workspace_dev, "brnn_create: null argument");
    Dims d;
    SCTC_TRY(derive_dims(cfg, &d));
    SCTC_CHECK_ARG(!cfg->train || grads_dev, "brnn_create: train model needs a gradient buffer");
    SCTC_CHECK_ARG(((uintptr_t)params_dev & 15) == 0 && ((uintptr_t)workspace_dev & 255) == 0,
                   "brnn_create: params must be 16-byte and workspace 256-byte aligned");
    sctc_brnn* h = new sctc_brnn();
    h->cfg = *cfg;
    h->D = d.D; h->Dp = d.Dp; h->H = d.H; h->Hp = d.Hp; h->A = d.A; h->Ap = d.Ap;
    h->NL = d.NL; h->TL = d.TL;
    h->maxF = cfg->max_frames;
    h->maxB = cfg->max_utts;
    h->params = params_dev;
    h->grads = grads_dev;
    build_tensor_table(d, &h->tinfo, &h->param_elems, &h->param_count);
    const size_t need = carve(cfg, d, nullptr, nullptr, 0);
    if (workspace_bytes < need) {
        delete h;
        return set_error(SCTC_ERR_WORKSPACE, "brnn_create: workspace %zu bytes < %zu needed",
                         workspace_bytes, need);
    }
    carve(cfg, d, h, workspace_dev, workspace_bytes);
    // padding columns of the CTC gradient are never written by the kernels: zero once
    hipError_t e = hipSuccess;
    if (cfg->train) e = hipMemset(h->dlogits, 0, sizeof(float) * h->maxF * LD(h->Ap));
    // the error word and the step flags: an engine that never runs a recurrent launch (a stream lane
    // without utterances, a net without a temporal layer) must not report what the allocator left here
    if (e == hipSuccess) e = hipMemset(h->counters, 0, sizeof(unsigned) * REC_COUNTER_WORDS);
    if (e != hipSuccess) {
        delete h;
        return set_error(SCTC_ERR_HIP, "brnn_create: %s", hipGetErrorString(e));
    }
    h->grad_ev.resize(h->tinfo.size(), nullptr);
    for (size_t i = 0; i < h->tinfo.size(); ++i)
        if (cfg->train && h->tinfo[i].kind != 1 &&
            hipEventCreateWithFlags(&h->grad_ev[i], hipEventDisableTiming) != hipSuccess) {
            delete h;
            return set_error(SCTC_ERR_HIP, "brnn_create: hipEventCreate failed");
        }
    const char* sm = getenv("SCTC_REC_SYNC");
    h->rec_sync_mode = sm ? atoi(sm) : 1;
    const char* rv = getenv("SCTC_REC_VARIANT");
    h->rec_variant = rv ? atoi(rv) : 0;
    const char* pd = getenv("SCTC_REC_POLL_DELAY");
    h->rec_poll_delay = pd ? atoi(pd) : -1;
    if (const char* fa = getenv("SCTC_FUSE_ADD")) h->fuse_add = atoi(fa) != 0;
    const char* dbg = getenv("SCTC_REC_DEBUG");
    h->rec_debug_on = dbg ? atoi(dbg) : 0;
    *out = h;
    return SCTC_OK;
}

int sctc_brnn_destroy(sctc_brnn_t h)
{
    if (!h) return SCTC_OK;
    if (h->ev_ready)
        for (int i = 0; i <= SCTC_N_PHASES; ++i) {
            (void)hipEventDestroy(h->ev[i][0]);
            (void)hipEventDestroy(h->ev[i][1]);
        }
    if (h->ctc_stage) ctc_free_stage(h->ctc_stage);
    for (hipEvent_t e : h->grad_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->tev)
        if (e) (void)hipEventDestroy(e);
    delete h;
    return SCTC_OK;
}

int sctc_brnn_tensor_info(sctc_brnn_t h, int32_t index, sctc_tensor_info* out)
{
    SCTC_CHECK_ARG(h && out && index >= 0 && index < (int)h->tinfo.size(),
                   "brnn_tensor_info: bad index %d", index);
    *out = h->tinfo[index];
    return SCTC_OK;
}

int sctc_brnn_cost_and_grad_async(sctc_brnn_t h, const sctc_minibatch* mb, int32_t flags,
                                  double* cost_dev, int32_t* skip_dev, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    bool all_skipped = false;
    SCTC_TRY(run_cost_and_grad(h, mb, flags, s, &all_skipped));
    if (cost_dev)
        SCTC_HIP_TRY(hipMemcpyAsync(cost_dev, h->d_cost_out, sizeof(double) * h->B,
                                    hipMemcpyDeviceToDevice, s));
    if (skip_dev)
        SCTC_HIP_TRY(hipMemcpyAsync(skip_dev, h->d_skip_out, sizeof(int32_t) * h->B,
                                    hipMemcpyDeviceToDevice, s));
    return SCTC_OK;
}

void* sctc_brnn_grad_event(sctc_brnn_t h, int32_t index)
{
    if (!h || index < 0 || index >= (int)h->grad_ev.size()) {
        set_error(SCTC_ERR_ARG, "brnn_grad_event: bad index %d", index);
        return nullptr;
    }
    return (void*)h->grad_ev[index];
}

int sctc_stream_wait_event(void* stream, void* event)
{
    SCTC_CHECK_ARG(event, "stream_wait_event: null event");
    SCTC_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return SCTC_OK;
}

// ---- the gradient exchange as a C entry (round 5; the Python host side does the same through
// torch.distributed in dist_sgd.allreduce_overlapped).  RCCL is resolved with dlopen at first use: libsctc_hip.so
// itself has no link dependency on it.
namespace {
struct Rccl {
    typedef int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef const char* (*ErrStr)(int);
    void* lib = nullptr;
    AllReduce all_reduce = nullptr;
    ErrStr err_str = nullptr;
    bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
}  // namespace

int sctc_brnn_allreduce_grads(sctc_brnn_t h, void* rccl_comm, void* compute_stream, void* side_stream,
                              double* side_dev, int32_t side_count, int32_t backward_queued)
{
    SCTC_CHECK_ARG(h && rccl_comm, "brnn_allreduce_grads: null handle / communicator");
    SCTC_CHECK_ARG(h->grads, "brnn_allreduce_grads: the model has no gradient buffer (train = 0)");
    SCTC_CHECK_ARG(side_stream && side_stream != compute_stream, "brnn_allreduce_grads: needs a side stream of its own");
    SCTC_CHECK_ARG(side_count >= 0 && (side_count == 0 || side_dev), "brnn_allreduce_grads: bad side message");
    Rccl::AllReduce rccl_all_reduce = nullptr;      // the function pointers are read under the lock that guards their loading
    Rccl::ErrStr rccl_err_str = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_rccl_mu);
        if (!g_rccl.tried) {
            g_rccl.tried = true;
            const char* names[3] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
            for (int k = 0; k < 3 && !g_rccl.lib; ++k) g_rccl.lib = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.lib) {
                g_rccl.all_reduce = (Rccl::AllReduce)dlsym(g_rccl.lib, "ncclAllReduce");
                g_rccl.err_str = (Rccl::ErrStr)dlsym(g_rccl.lib, "ncclGetErrorString");
            }
        }
        rccl_all_reduce = g_rccl.all_reduce;
        rccl_err_str = g_rccl.err_str;
    }
    if (!rccl_all_reduce)
        return set_error(SCTC_ERR_STATE, "brnn_allreduce_grads: librccl.so (ncclAllReduce) could not be loaded");
    hipStream_t cs = (hipStream_t)compute_stream, ss = (hipStream_t)side_stream;
    constexpr int NCCL_SUM = 0, NCCL_FLOAT = 7, NCCL_DOUBLE = 8;      // rccl.h: ncclSum, ncclFloat32, ncclFloat64
    auto reduce = [&](void* p, size_t n, int dt) -> int {
        const int rc = rccl_all_reduce(p, p, n, dt, NCCL_SUM, rccl_comm, ss);
        if (rc != 0)
            return set_error(SCTC_ERR_HIP, "ncclAllReduce failed: %s", rccl_err_str ? rccl_err_str(rc) : "?");
        return SCTC_OK;
    };
    // an event of our own orders "everything queued on the compute stream so far" in front of the side stream
    hipEvent_t ev = nullptr;
    SCTC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int rc = SCTC_OK;
    auto after_compute = [&]() -> int {
        SCTC_HIP_TRY(hipEventRecord(ev, cs));
        SCTC_HIP_TRY(hipStreamWaitEvent(ss, ev, 0));
        return SCTC_OK;
    };
    if (!backward_queued) rc = after_compute();   // no backward pass this step: the buffer was zeroed on the compute stream
    // buckets in the order the backward pass finishes them: output layer first, the recurrent pair right behind
    // the temporal layer's BPTT (nnets/brnnet.py gradBuckets)
    const int n = (int)h->tinfo.size();
    auto slice_end = [&](int i) -> int64_t { return i + 1 < n ? h->tinfo[i + 1].offset : h->param_elems; };
    for (int i = h->NL; i >= 0 && rc == SCTC_OK; --i) {
        const int w = weight_index(h, i);
        if (backward_queued) {
            const hipError_t e = hipStreamWaitEvent(ss, h->grad_ev[w], 0);
            if (e != hipSuccess) { rc = set_error(SCTC_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(e)); break; }
        }
        rc = reduce(h->grads + h->tinfo[w].offset, (size_t)(slice_end(w + 1) - h->tinfo[w].offset), NCCL_FLOAT);   // W and b
        if (rc == SCTC_OK && i == h->TL) {
            for (int k : {wf_index(h), wb_index(h)}) {
                if (backward_queued) {      // the recurrent pair's sum must not start before BPTT has written it
                    const hipError_t e = hipStreamWaitEvent(ss, h->grad_ev[k], 0);
                    if (e != hipSuccess) { rc = set_error(SCTC_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(e)); break; }
                }
                rc = reduce(h->grads + h->tinfo[k].offset, (size_t)(slice_end(k) - h->tinfo[k].offset), NCCL_FLOAT);
                if (rc != SCTC_OK) break;
            }
        }
    }
    if (rc == SCTC_OK && side_count > 0) {
        rc = after_compute();                      // the side message is produced on the compute stream
        if (rc == SCTC_OK) rc = reduce(side_dev, (size_t)side_count, NCCL_DOUBLE);
    }
    if (rc == SCTC_OK) {                           // the compute stream continues behind the collectives
        hipError_t e = hipEventRecord(ev, ss);
        if (e == hipSuccess) e = hipStreamWaitEvent(cs, ev, 0);
        if (e != hipSuccess) rc = set_error(SCTC_ERR_HIP, "brnn_allreduce_grads: %s", hipGetErrorString(e));
    }
    (void)hipEventDestroy(ev);                     // (released by the runtime once the recorded work has run)
    return rc;
}

int sctc_brnn_check(sctc_brnn_t h, void* stream)
{
    SCTC_CHECK_ARG(h, "brnn_check: null handle");
    return check_recurrent_error(h, (hipStream_t)stream);
}

static int cost_and_grad_once(sctc_brnn_t h, const sctc_minibatch* mb, int32_t flags,
                              double* cost_host, int32_t* skip_host, double* regcost_host,
                              void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    bool all_skipped = false;
    SCTC_TRY(run_cost_and_grad(h, mb, flags, s, &all_skipped));
    int nw = 0;
    if (regcost_host && h->cfg.reg > 0.f) {
        // (reg/2) * sum ||w||^2 over every weight tensor incl. Wf, Wb (brnnet.py:178-183)
        for (size_t i = 0; i < h->tinfo.size(); ++i) {
            const sctc_tensor_info& t = h->tinfo[i];
            if (t.kind == 1) continue;
            const int64_t rows_p = round_up(t.rows, PAD);
            SCTC_TRY(launch_sumsq(h->params + t.offset, rows_p * t.ld, h->d_sumsq + nw,
                                  h->sumsq_ws, s));
            ++nw;
        }
    }
    if (cost_host)
        SCTC_HIP_TRY(hipMemcpyAsync(cost_host, h->d_cost_out, sizeof(double) * h->B,
                                    hipMemcpyDeviceToHost, s));
    if (skip_host)
        SCTC_HIP_TRY(hipMemcpyAsync(skip_host, h->d_skip_out, sizeof(int32_t) * h->B,
                                    hipMemcpyDeviceToHost, s));
    std::vector<double> ss(nw);
    if (nw)
        SCTC_HIP_TRY(hipMemcpyAsync(ss.data(), h->d_sumsq, sizeof(double) * nw,
                                    hipMemcpyDeviceToHost, s));
    SCTC_TRY(check_recurrent_error(h, s));  // synchronises the stream
    if (regcost_host) {
        double rc = 0.0;
        for (double v : ss) rc += 0.5 * (double)h->cfg.reg * v;
        *regcost_host = rc;
    }
    return SCTC_OK;
}

int sctc_brnn_cost_and_grad(sctc_brnn_t h, const sctc_minibatch* mb, int32_t flags,
                            double* cost_host, int32_t* skip_host, double* regcost_host,
                            void* stream)
{
    SCTC_CHECK_ARG(h, "brnn_cost_and_grad: null handle");
    return with_retry(h, !(flags & SCTC_FLAG_ACCUMULATE), [&]() {
        return cost_and_grad_once(h, mb, flags, cost_host, skip_host, regcost_host, stream);
    });
}

static int forward_once(sctc_brnn_t h, const sctc_minibatch* mb, float* probs_dev, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    PhaseTimer pt{h, s};
    if (h->profiling) memset(h->phase_ms, 0, sizeof(h->phase_ms));
    h->tev_n = 0;
    SCTC_TRY(make_plan(h, mb, false, s));
    if (h->TL > 0) SCTC_TRY(recurrent_clear_error(h->counters, s));
    SCTC_TRY(run_forward(h, mb, s, pt));
    // probs back in the caller's per-utterance order, brnnet.py:170-173
    SCTC_TRY(launch_scatter_rows(probs_dev, h->A, h->probs, LD(h->Ap), h->d_src_row, h->N, h->A, s));
    pt.end();
    return check_recurrent_error(h, s);
}

int sctc_brnn_forward(sctc_brnn_t h, const sctc_minibatch* mb, float* probs_dev, void* stream)
{
    SCTC_CHECK_ARG(h && probs_dev, "brnn_forward: null argument");
    return with_retry(h, true, [&]() { return forward_once(h, mb, probs_dev, stream); });
}

int sctc_set_shared_device(int32_t on)
{
    recurrent_set_shared_device_mode(on);
    return SCTC_OK;
}

int sctc_shared_device(void) { return recurrent_shared_device_mode(); }

int sctc_device_pci_bus_id(int32_t device, char* out, int32_t out_len)
{
    SCTC_CHECK_ARG(out && out_len >= 16, "device_pci_bus_id: buffer of >= 16 bytes needed");
    int dev = device;
    if (dev < 0) SCTC_HIP_TRY(hipGetDevice(&dev));
    SCTC_HIP_TRY(hipDeviceGetPCIBusId(out, out_len, dev));
    return SCTC_OK;
}

int sctc_brnn_recurrent_path(sctc_brnn_t h, int32_t* forward_path, int32_t* bptt_path, int32_t* retries)
{
    SCTC_CHECK_ARG(h, "brnn_recurrent_path: null handle");
    if (forward_path) *forward_path = h->rec_path[0];
    if (bptt_path) *bptt_path = h->rec_path[1];
    if (retries) *retries = h->rec_retries;
    return SCTC_OK;
}

/* diagnostics: s_memtime stamps of the recurrent kernel (SCTC_REC_DEBUG=1), [2 passes][2 wgs][16 steps][8] */
int sctc_brnn_debug_read(sctc_brnn_t h, uint32_t* out, int32_t n_words)
{
    SCTC_CHECK_ARG(h && out && n_words >= 0 && n_words <= 2 * REC_DEBUG_WORDS, "debug_read: bad argument");
    SCTC_HIP_TRY(hipMemcpy(out, h->rec_debug, sizeof(uint32_t) * n_words, hipMemcpyDeviceToHost));
    return SCTC_OK;
}

/* diagnostics: device pointers to the engine's internal matrices of the LAST call (packed
 * time-major rows: frame t of the utterance with length rank b is row rowbase[t] + b; one
 * utterance: row t) */
int sctc_brnn_debug_buffer(sctc_brnn_t h, int32_t which, void** dev_ptr, int64_t* rows, int64_t* cols,
                           int64_t* ld)
{
    SCTC_CHECK_ARG(h && dev_ptr && rows && cols && ld, "debug_buffer: null argument");
    SCTC_CHECK_ARG(!(h->cfg.operand_dtype == SCTC_F16 && ((which >= 1 && which <= h->NL) || which == 200)),
                   "debug_buffer: in the fp16-operand configuration the hidden layers' activations and the deltas "
                   "between them exist as 16-bit shadows only (buffer %d has no fp32 copy; 0, 100, 101 do)", which);
    const float* p = nullptr;
    int64_t c = h->Hp, l = LD(h->Hp);
    if (which >= 0 && which <= h->NL) {
        p = h->act[which];
        if (which == 0) { c = h->Dp; l = LD(h->Dp); }
    } else if (which == 100) p = h->hF;
    else if (which == 101) p = h->hB;
    else if (which == 102) p = h->Z;
    else if (which == 200) p = h->last_delta1;
    SCTC_CHECK_ARG(p, "debug_buffer: no buffer %d (layers 0..%d, 100 hF, 101 hB, 102 z of the temporal layer, 200 delta_1)", which, h->NL);
    *dev_ptr = (void*)p;
    *rows = h->N;
    *cols = c;
    *ld = l;
    return SCTC_OK;
}

int sctc_brnn_ctc_workspace_bytes(sctc_brnn_t h, const sctc_minibatch* mb, size_t* needed, size_t* reserved)
{
    SCTC_CHECK_ARG(h && mb && needed, "brnn_ctc_workspace_bytes: null argument");
    SCTC_CHECK_ARG(h->cfg.train, "brnn_ctc_workspace_bytes: the model was created with train = 0 (no CTC)");
    SCTC_CHECK_ARG(mb->B >= 1 && mb->B <= h->cfg.max_utts && mb->T_b && mb->U_b, "brnn_ctc_workspace_bytes: bad minibatch");
    CtcPlan plan;
    SCTC_TRY(ctc_make_plan(mb->B, h->A, 0, SCTC_F32, mb->T_b, mb->U_b, &plan));
    *needed = plan.bytes;
    if (reserved) *reserved = h->ctc_ws_ext ? h->ctc_ws_ext_bytes : h->ctc_ws_bytes;
    return SCTC_OK;
}

int sctc_brnn_set_ctc_workspace(sctc_brnn_t h, void* workspace_dev, size_t workspace_bytes)
{
    SCTC_CHECK_ARG(h, "null handle");
    SCTC_CHECK_ARG(h->cfg.train, "brnn_set_ctc_workspace: the model was created with train = 0 (no CTC)");
    SCTC_CHECK_ARG((workspace_dev == nullptr) == (workspace_bytes == 0), "brnn_set_ctc_workspace: a buffer and its size, or NULL and 0");
    SCTC_CHECK_ARG(((uintptr_t)workspace_dev & 255) == 0, "brnn_set_ctc_workspace: the buffer must be 256-byte aligned");
    h->ctc_ws_ext = workspace_dev;
    h->ctc_ws_ext_bytes = workspace_bytes;
    return SCTC_OK;
}

int sctc_brnn_set_profiling(sctc_brnn_t h, int32_t enable)
{
    SCTC_CHECK_ARG(h, "null handle");
    if (enable && !h->ev_ready) {
        for (int i = 0; i <= SCTC_N_PHASES; ++i) {
            SCTC_HIP_TRY(hipEventCreate(&h->ev[i][0]));
            SCTC_HIP_TRY(hipEventCreate(&h->ev[i][1]));
        }
        h->ev_ready = true;
    }
    if (enable == 2 && h->tev.empty()) {
        h->tev.resize(128);
        for (hipEvent_t& e : h->tev) SCTC_HIP_TRY(hipEventCreate(&e));
    }
    h->profiling = enable == 2 ? 2 : (enable ? 1 : 0);
    h->tev_n = 0;
    return SCTC_OK;
}

int sctc_brnn_phase_ms(sctc_brnn_t h, float* ms_out)
{
    SCTC_CHECK_ARG(h && ms_out, "null argument");
    if (h->profiling == 2 && h->tev_n >= 2) {
        // resolve the events of the last call: the interval [event i, event i+1) belongs to the phase
        // that started at event i
        SCTC_HIP_TRY(hipEventSynchronize(h->tev[h->tev_n - 1]));
        memset(h->phase_ms, 0, sizeof(h->phase_ms));
        for (int i = 0; i + 1 < h->tev_n; ++i) {
            if (h->tev_phase[i] < 0) continue;
            float ms = 0.f;
            SCTC_HIP_TRY(hipEventElapsedTime(&ms, h->tev[i], h->tev[i + 1]));
            h->phase_ms[h->tev_phase[i]] += ms;
        }
    }
    memcpy(ms_out, h->phase_ms, sizeof(h->phase_ms));
    return SCTC_OK;
}

int sctc_brnn_flops(sctc_brnn_t h, const sctc_minibatch* mb, double* total, double* gemm,
                    double* recurrent)
{
    SCTC_CHECK_ARG(h && mb && mb->T_b, "null argument");
    // per utterance, multiply-add = 2, unpadded dimensions
    double fg = 0.0, fr = 0.0;
    for (int b = 0; b < mb->B; ++b) {
        const double T = mb->T_b[b];
        for (int l = 0; l <= h->NL; ++l) {
            const double in = l == 0 ? h->D : h->H, out = l == h->NL ? h->A : h->H;
            fg += 2.0 * out * in * T;              // forward
            if (h->cfg.train) {
                fg += 2.0 * out * in * T;          // weight gradient
                if (l > 0) fg += 2.0 * out * in * T;  // delta propagation
            }
        }
        if (h->TL > 0) {
            const double HH = (double)h->H * h->H;
            fr += 2.0 * 2.0 * HH * (T - 1);        // forward recurrence, two directions
            if (h->cfg.train) {
                fr += 2.0 * 2.0 * HH * (T - 1);    // BPTT
                fg += 2.0 * 2.0 * HH * (T - 1);    // dWf, dWb (time-batched GEMMs)
            }
        }
    }
    if (total) *total = fg + fr;
    if (gemm) *gemm = fg;
    if (recurrent) *recurrent = fr;
    return SCTC_OK;
}

}  // extern "C"
