// C ABI: library housekeeping + the CTC entry points of include/sctc.h.
#include <math.h>

#include <atomic>
#include <vector>

#include "common.h"
#include "ctc_kernels.h"

namespace sctc {

char* err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int ctc_make_plan(int B, int A, int blank, int dtype, const int32_t* T_b, const int32_t* U_b,
                  CtcPlan* plan)
{
    SCTC_CHECK_ARG(B >= 1, "ctc: empty batch");
    SCTC_CHECK_ARG(A >= 1, "ctc: alphabet size %d < 1", A);
    SCTC_CHECK_ARG(blank >= 0 && blank < A, "ctc: blank id %d outside the alphabet", blank);
    int max_L = 0, max_T = 0;
    int64_t n_labels = 0, frames = 0;
    for (int b = 0; b < B; ++b) {
        SCTC_CHECK_ARG(T_b[b] >= 1, "ctc: utterance %d has no frames", b);
        // U == 0 is undefined behaviour in the reference (out-of-bounds read)
        SCTC_CHECK_ARG(U_b[b] >= 1, "ctc: utterance %d has an empty label sequence", b);
        max_L = std::max(max_L, 2 * U_b[b] + 1);
        max_T = std::max(max_T, T_b[b]);
        n_labels += U_b[b];
        frames += T_b[b];
    }
    int W = 1;
    int K = ctc_lattice_shape(max_L, &W);
    // label rows of more than 2048 states and alphabets of more than 256 symbols (the reference bounds neither,
    // ctc_fast.pyx:22-32): the generic kernels of ctc_generic.hip
    {
        const char* gz = getenv("SCTC_CTC_GENERIC");
        plan->generic = K == 0 || A > 256 || (gz && atoi(gz) != 0);
    }
    if (plan->generic) { K = 0; W = 0; }
    plan->B = B;
    plan->A = A;
    plan->blank = blank;
    plan->K = K;
    plan->W = W;
    {
        const char* lz = getenv("SCTC_CTC_LAZY");
        plan->lazy = (lz && !plan->generic) ? atoi(lz) : 0;
    }
    plan->lp = plan->generic ? (int)round_up(max_L + 1, 64) : 64 * W * K;
    plan->max_T = max_T;
    plan->frames = frames;
    plan->n_labels = n_labels;
    // Rows of up to 512 states: both recursions and the gradient in ONE kernel that keeps one packed half lattice
    // per direction (ctc_fused.hip).  SCTC_CTC_FUSED=0 runs the three-kernel path (A/B, tests); the lazy rescaling
    // schedule exists there only.  Float32 probabilities keep their rows in the 32-bit format of ctc_fused.hip
    // unless SCTC_CTC_STORE=64.
    {
        const char* fz = getenv("SCTC_CTC_FUSED");
        const char* sb = getenv("SCTC_CTC_STORE");
        // the fused kernel's own shape: one wave per direction with 2 / 4 / 8 states per lane (rows of up to 128 / 256 /
        // 512 states: cfg-4's 401 included)
        const int fk = max_L <= 128 ? 2 : (max_L <= 256 ? 4 : (max_L <= 512 ? 8 : 0));
        plan->fused = !plan->generic && fk > 0 && A <= 256 && !plan->lazy && (fz ? atoi(fz) != 0 : true);
        if (plan->fused) {
            K = fk;
            W = 1;
            plan->K = K;
            plan->W = 1;
            plan->lp = 64 * K;
        }
        plan->store_bytes = (dtype == SCTC_F32 && !(sb && atoi(sb) == 64)) ? 4 : 8;
        // Rows of 513..2048 states (round 6): the same schedule on W = 4 / 8 waves per direction (ctc_fusedw.hip), from
        // SCTC_CTC_WIDE_MIN_B utterances on (default 18; rows of up to 1024 states, four waves: 24).  Below that the lattice
        // + grad kernels are faster -- their gradient kernel spreads over the CUs the few recursion workgroups leave idle
        // (ms per call, wide / lattice + grad; cfg-5 shape, T = 8000 / U = 800: 8 utterances 5.40 / 4.95, 16: 5.45 / 5.39, 24:
        // 5.48 / 5.82, 128: 6.0 / 15; T = 4000 / U = 511: 16: 2.17 / 2.05, 24: 2.19 / 2.19, 32: 2.22 / 2.28) -- at five times
        // the traffic and ten times the workspace.
        const char* wz = getenv("SCTC_CTC_WIDE");
        const char* wb = getenv("SCTC_CTC_WIDE_MIN_B");
        const int wide = wz ? atoi(wz) : -1;
        const bool fused_on = fz ? atoi(fz) != 0 : true;
        const bool wide_auto = !plan->fused && B >= (wb ? atoi(wb) : (max_L > 1024 ? 18 : 24));
        if (!plan->generic && !plan->lazy && A <= 256 && max_L <= 2048 && fused_on && wide != 0 && (wide == 1 || wide_auto)) {
            const char* force = getenv("SCTC_CTC_WAVES");
            W = (max_L > 1024 || (force && atoi(force) == 8)) ? 8 : 4;
            K = max_L <= 64 * W * 2 ? 2 : 4;
            plan->fused = 2;
            plan->K = K;
            plan->W = W;
            plan->lp = 64 * W * K;
        }
    }
    const size_t head = align256(sizeof(CtcUtt) * B) + align256(sizeof(int32_t) * (2 * n_labels + (int64_t)B * (A + 1)));
    if (plan->fused) {
        int64_t elems = 0;
        for (int b = 0; b < B; ++b) elems += K + (int64_t)T_b[b] * round_up(2 * U_b[b] + 1, K);
        plan->lat_elems = elems;
        plan->bytes = head + align256((size_t)plan->store_bytes * elems) + (plan->fused == 2 ? align256(ctc_fusedw_sync_bytes(B)) : 0);
        return SCTC_OK;
    }
    plan->lat_elems = frames * plan->lp;
    if (plan->generic) {    // per-utterance row strides (ctc_generic.hip)
        plan->lat_elems = 0;
        for (int b = 0; b < B; ++b) plan->lat_elems += (int64_t)T_b[b] * round_up(2 * U_b[b] + 2, 64);
    }
    // labels, the same labels grouped by value (ctc_grad's per-label sums), group offsets
    plan->bytes = head + align256(sizeof(double) * 2 * B) + align256(sizeof(int32_t) * 2 * B) +
                  2 * align256(sizeof(double) * plan->lat_elems) +
                  (plan->generic ? align256(sizeof(double) * 4 * (size_t)B * plan->lp) : 0);
    return SCTC_OK;
}

// `keep` (optional): caller-owned host staging that outlives the async uploads; without
// it the function synchronises the stream before its local staging goes away.
struct CtcHostStage {
    std::vector<CtcUtt> utts;
    std::vector<int32_t> labels;    // [n_labels] labels | [n_labels] odd states grouped by label | [B][A+1] group offsets
    std::vector<int32_t> cursor;
    // a kept stage (the BRNN engine's) uploads from pinned memory: the descriptors go up in the middle of a
    // step, between the forward pass and the lattice kernel (common.h, PinnedStage)
    PinnedStage pinned;
};

template <typename R>
static int run_ctc(const sctc_ctc_batch* bt, const CtcPlan& plan, const R* probs, R* grad,
                   double* cost, int32_t* skip, void* ws, size_t ws_bytes, hipStream_t stream,
                   CtcHostStage* keep)
{
    Arena ar;
    ar.init(ws, ws_bytes);
    CtcUtt* d_utts = ar.take<CtcUtt>(plan.B);
    const int64_t n_int = 2 * plan.n_labels + (int64_t)plan.B * (plan.A + 1);
    int32_t* d_labels = ar.take<int32_t>(n_int);
    double *d_ll = nullptr, *d_alpha = nullptr, *d_beta = nullptr, *d_scratch = nullptr;
    int32_t* d_skip2 = nullptr;
    char* d_store = nullptr;
    uint32_t* d_sync = nullptr;
    if (plan.fused) {
        d_store = ar.take<char>((size_t)plan.store_bytes * plan.lat_elems);
        if (plan.fused == 2) d_sync = ar.take<uint32_t>(ctc_fusedw_sync_bytes(plan.B) / sizeof(uint32_t));
    } else {
        d_ll = ar.take<double>(2 * plan.B);
        d_skip2 = ar.take<int32_t>(2 * plan.B);
        d_alpha = ar.take<double>(plan.lat_elems);
        d_beta = ar.take<double>(plan.lat_elems);
        if (plan.generic) d_scratch = ar.take<double>(4 * (size_t)plan.B * plan.lp);
    }
    if (ar.overflow)
        return set_error(SCTC_ERR_WORKSPACE, "ctc: workspace %zu bytes < %zu needed", ws_bytes,
                         ar.used);

    // no caller-owned stage (the standalone sctc_ctc_loss_batch): a per-thread stage that lives as long as the
    // process (never destroyed: its pinned buffer and event must not be freed behind the HIP runtime's back at exit),
    // so these uploads come from pinned memory too -- from pageable vectors the 4 MB of descriptors of a
    // 4096-utterance batch cost 1 ms with the stream waiting
    static thread_local CtcHostStage* tls_stage = new CtcHostStage();
    CtcHostStage& st = keep ? *keep : *tls_stage;
    st.utts.resize(plan.B);
    st.labels.resize(n_int);
    std::vector<CtcUtt>& utts = st.utts;
    std::vector<int32_t>& labels = st.labels;
    int64_t lat_off = 0, lab_off = 0;
    for (int b = 0; b < plan.B; ++b) {
        CtcUtt& u = utts[b];
        u.T = bt->T_b[b];
        u.U = bt->U_b[b];
        u.row0 = bt->frame_off[b];
        u.lat_off = plan.fused ? lat_off + plan.K : lat_off;   // fused: K elements of pad in front (ctc_fused.hip)
        u.lab_off = lab_off;
        const int32_t* src = bt->labels + bt->label_off[b];
        for (int i = 0; i < u.U; ++i) {
            SCTC_CHECK_ARG(src[i] >= 0 && src[i] < plan.A, "ctc: label %d of utterance %d is %d, "
                           "outside the alphabet [0,%d)", i, b, src[i], plan.A);
            labels[lab_off + i] = src[i];
        }
        // stable counting sort of the label positions by label value: the gradient kernel sums a label's states
        // in ascending order (ctc_fast.pyx:120-131) without scanning the whole row for every label
        int32_t* start = labels.data() + 2 * plan.n_labels + (int64_t)b * (plan.A + 1);
        int32_t* grouped = labels.data() + plan.n_labels + lab_off;
        std::fill(start, start + plan.A + 1, 0);
        for (int i = 0; i < u.U; ++i) ++start[src[i] + 1];
        for (int k = 0; k < plan.A; ++k) start[k + 1] += start[k];
        st.cursor.assign(start, start + plan.A);
        for (int i = 0; i < u.U; ++i) grouped[st.cursor[src[i]]++] = 2 * i + 1;
        lat_off += plan.fused ? plan.K + (int64_t)u.T * round_up(2 * u.U + 1, plan.K)
                              : (plan.generic ? (int64_t)u.T * round_up(2 * u.U + 2, 64) : (int64_t)u.T * plan.lp);
        lab_off += u.U;
    }
    const size_t utt_bytes = sizeof(CtcUtt) * plan.B, utt_span = align256(utt_bytes), lab_bytes = sizeof(int32_t) * n_int;
    char* pin = static_cast<char*>(st.pinned.acquire(utt_span + lab_bytes));
    const bool staged_pinned = pin != nullptr;
    if (pin) {
        memcpy(pin, utts.data(), utt_bytes);
        memcpy(pin + utt_span, labels.data(), lab_bytes);
        PinnedUploadGuard guard(stream, true);
        SCTC_HIP_TRY(hipMemcpyAsync(d_utts, pin, utt_bytes, hipMemcpyHostToDevice, stream));
        SCTC_HIP_TRY(hipMemcpyAsync(d_labels, pin + utt_span, lab_bytes, hipMemcpyHostToDevice, stream));
        SCTC_HIP_TRY(st.pinned.uploaded(stream));
        guard.done();
    } else {
        SCTC_HIP_TRY(hipMemcpyAsync(d_utts, utts.data(), utt_bytes, hipMemcpyHostToDevice, stream));
        SCTC_HIP_TRY(hipMemcpyAsync(d_labels, labels.data(), lab_bytes, hipMemcpyHostToDevice, stream));
    }

    if (plan.fused) {
        CtcFusedArgs<R> fa;
        fa.utts = d_utts;
        fa.probs = probs;
        fa.grad = grad;
        fa.ld = bt->ld;
        fa.A = plan.A;
        fa.blank = plan.blank;
        fa.rowbase = bt->rowbase_dev;
        fa.labels = d_labels;
        fa.by_label = d_labels + plan.n_labels;
        fa.label_start = d_labels + 2 * plan.n_labels;
        fa.store = d_store;
        fa.cost = cost;
        fa.skip = skip;
        {
            // timing decomposition of the fused kernel (tools/ctc_diag_timing.py): parts of the schedule are switched
            // off and the results are WRONG -- never silently
            const char* dz = getenv("SCTC_CTC_DIAG");
            fa.diag = dz ? atoi(dz) : 0;
            static std::atomic<bool> warned{false};
            if (fa.diag != 0 && !warned.exchange(true))
                fprintf(stderr, "sctc: SCTC_CTC_DIAG=%d is set: the fused CTC kernel skips parts of its schedule, costs and "
                                "gradients are WRONG (diagnostic timing runs only; unset it)\n", fa.diag);
        }
        fa.sync = d_sync;
        if (plan.fused == 2) SCTC_TRY(launch_ctc_fusedw<R>(fa, plan.B, plan.K, plan.W, plan.store_bytes, stream));
        else SCTC_TRY(launch_ctc_fused<R>(fa, plan.B, plan.K, plan.store_bytes, stream));
        if (!staged_pinned) SCTC_HIP_TRY(hipStreamSynchronize(stream));   // pageable staging must outlive the copies
        return SCTC_OK;
    }
    CtcLatticeArgs<R> la;
    la.utts = d_utts;
    la.probs = probs;
    la.ld = bt->ld;
    la.A = plan.A;
    la.blank = plan.blank;
    la.lp = plan.lp;
    la.rowbase = bt->rowbase_dev;
    la.labels = d_labels;
    la.alpha = d_alpha;
    la.beta = d_beta;
    la.ll = d_ll;
    la.skip2 = d_skip2;
    la.scratch = d_scratch;
    if (!plan.generic)
        SCTC_TRY(launch_ctc_lattice<R>(la, plan.B, plan.K, plan.W, sizeof(R) == 4 ? plan.lazy : 0, stream));

    CtcGradArgs<R> ga;
    ga.utts = d_utts;
    ga.probs = probs;
    ga.grad = grad;
    ga.ld = bt->ld;
    ga.A = plan.A;
    ga.blank = plan.blank;
    ga.lp = plan.lp;
    ga.rowbase = bt->rowbase_dev;
    ga.labels = d_labels;
    ga.by_label = d_labels + plan.n_labels;
    ga.label_start = d_labels + 2 * plan.n_labels;
    ga.alpha = d_alpha;
    ga.beta = d_beta;
    ga.ll = d_ll;
    ga.skip2 = d_skip2;
    ga.cost = cost;
    ga.skip = skip;
    ga.lazy = sizeof(R) == 4 ? plan.lazy : 0;
    if (plan.generic)
        SCTC_TRY(launch_ctc_generic<R>(la, ga, plan.B, plan.max_T, stream));
    else
        SCTC_TRY(launch_ctc_grad<R>(ga, plan.B, plan.max_T, stream));
    // pageable host staging must outlive the async copies
    if (!staged_pinned) SCTC_HIP_TRY(hipStreamSynchronize(stream));
    return SCTC_OK;
}

int ctc_run_batch(const sctc_ctc_batch* bt, const void* probs, void* grad, double* cost,
                  int32_t* skip, void* ws, size_t ws_bytes, hipStream_t stream, void* keep_stage)
{
    CtcHostStage* keep = static_cast<CtcHostStage*>(keep_stage);
    SCTC_CHECK_ARG(bt && probs && grad && cost && skip, "ctc: null argument");
    SCTC_CHECK_ARG(bt->dtype == SCTC_F32 || bt->dtype == SCTC_F64, "ctc: bad dtype %d", bt->dtype);
    SCTC_CHECK_ARG(bt->ld >= bt->A, "ctc: ld %lld < A %d", (long long)bt->ld, bt->A);
    CtcPlan plan;
    SCTC_TRY(ctc_make_plan(bt->B, bt->A, bt->blank, bt->dtype, bt->T_b, bt->U_b, &plan));
    if (bt->dtype == SCTC_F32)
        return run_ctc<float>(bt, plan, (const float*)probs, (float*)grad, cost, skip, ws,
                              ws_bytes, stream, keep);
    return run_ctc<double>(bt, plan, (const double*)probs, (double*)grad, cost, skip, ws,
                           ws_bytes, stream, keep);
}

void* ctc_new_stage() { return new CtcHostStage(); }
void ctc_free_stage(void* p) { delete static_cast<CtcHostStage*>(p); }

}  // namespace sctc

using namespace sctc;

extern "C" {

int sctc_abi_version(void) { return SCTC_ABI_VERSION; }

const char* sctc_last_error(void) { return err_buf(); }

int sctc_set_device(int device)
{
    SCTC_HIP_TRY(hipSetDevice(device));
    return SCTC_OK;
}

int sctc_device_info(int* compute_units, int* lds_bytes_per_cu, int64_t* total_mem_bytes,
                     char* name, int name_len)
{
    int dev = 0;
    SCTC_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SCTC_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (total_mem_bytes) *total_mem_bytes = (int64_t)prop.totalGlobalMem;
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return SCTC_OK;
}

size_t sctc_ctc_workspace_bytes(const sctc_ctc_batch* bt)
{
    if (!bt) return 0;
    CtcPlan plan;
    if (ctc_make_plan(bt->B, bt->A, bt->blank, bt->dtype, bt->T_b, bt->U_b, &plan) != SCTC_OK) return 0;
    return plan.bytes;
}

int sctc_ctc_loss_batch(const sctc_ctc_batch* batch, const void* probs_dev, void* grad_dev,
                        double* cost_dev, int32_t* skip_dev, void* workspace_dev,
                        size_t workspace_bytes, void* stream)
{
    return ctc_run_batch(batch, probs_dev, grad_dev, cost_dev, skip_dev, workspace_dev,
                         workspace_bytes, (hipStream_t)stream, nullptr);
}

int sctc_softmax_rows(const float* logits_dev, float* probs_dev, int64_t rows, int32_t A,
                      int64_t ld, void* stream)
{
    SCTC_CHECK_ARG(logits_dev && probs_dev, "softmax_rows: null pointer");
    return launch_softmax_rows(logits_dev, probs_dev, rows, A, ld, (hipStream_t)stream);
}

int sctc_argmax_rows(const void* probs_dev, int32_t dtype, int32_t* best_dev, int64_t rows,
                     int32_t A, int64_t ld, void* stream)
{
    SCTC_CHECK_ARG(probs_dev && best_dev, "argmax_rows: null pointer");
    return launch_argmax_rows(probs_dev, dtype, best_dev, rows, A, ld, (hipStream_t)stream);
}

}  // extern "C"
