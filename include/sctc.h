/*
 * sctc.h -- C ABI of libsctc_hip.so: the MI355X (gfx950) drop-in for the
 * per-utterance training hot path of amaas/stanford-ctc.
 *
 * Every entry point replaces one interface of the reference (cited per function,
 * paths relative to the reference root).  The reference is Python over (a) a
 * Cython extension `ctc_fast` and (b) the cudamat CUDA library; a maintainer binds
 * this header with ctypes (INTEGRATION.md shows the stubs; the host-side mirror in
 * stanford-ctc_amd/ is exactly that binding).
 *
 * Conventions
 *   - plain C types only; `void* stream` is a hipStream_t (NULL = default stream).
 *   - "dev" pointers are HIP device pointers, "host" pointers ordinary memory.
 *   - the library never allocates device memory: the caller hands in parameter,
 *     gradient and workspace buffers (sizes come from the *_query functions).
 *   - return 0 (SCTC_OK) on success, < 0 on error (sctc_last_error() has the text).
 *     Numerical failure of an utterance (the reference's `skip`, ctc_fast.pyx:147-149)
 *     is reported through skip flags, never through the return code.
 *   - matrices: the reference stores (features x frames) column-major, i.e. one
 *     frame = `features` consecutive floats.  Here that is row-major [frames][ld].
 */
#ifndef SCTC_H_
#define SCTC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCTC_ABI_VERSION 6

#define SCTC_OK 0
#define SCTC_ERR_ARG (-1)       /* bad argument (the reference raises ValueError/AssertionError) */
#define SCTC_ERR_HIP (-2)       /* HIP runtime error */
#define SCTC_ERR_WORKSPACE (-3) /* workspace too small */
#define SCTC_ERR_TIMEOUT (-4)   /* a persistent recurrent launch gave up waiting for its peers (device not
                                   exclusively ours).  The synchronous entry points re-run such a step by
                                   themselves (device lease, then per-step launches) and only report this
                                   when they may not (SCTC_FLAG_ACCUMULATE); sctc_brnn_check reports it for
                                   asynchronous steps */
#define SCTC_ERR_STATE (-5)

#define SCTC_F32 0
#define SCTC_F64 1
#define SCTC_F16 2   /* operand type of the mixed-precision GEMMs / recurrent step (fp32 accumulate) */
#define SCTC_BF16 3
#define SCTC_BF16X3 4 /* fp32 operands split exactly into three bfloat16 terms (x = x1 + x2 + x3), six cross
                         products on the bfloat16 matrix cores, fp32 accumulate: an fp32-accurate GEMM */
#define SCTC_OPERANDS_16BIT 0x100 /* sctc_gemm_h16 only, OR-ed into SCTC_F16 / SCTC_BF16: A_dev and B_dev already
                                     point at 16-bit elements of that type (lda / ldb count them) -- the shadow
                                     copies a producer wrote -- instead of fp32 values to be rounded */

/* ---- library ---------------------------------------------------------- */

int sctc_abi_version(void);
const char* sctc_last_error(void);
/* cm.cuda_set_device(n), runNNet.py:117-120 */
int sctc_set_device(int device);
/* compute units / LDS per CU / total memory of the current device (any may be NULL) */
int sctc_device_info(int* compute_units, int* lds_bytes_per_cu, int64_t* total_mem_bytes,
                     char* name, int name_len);
/* Shared-device mode.  The reference's per-step launches (brnnet.py:148-152, :215-224) work on a
 * GPU that other processes use as well; the persistent recurrent kernels that replace them need
 * every workgroup of a pass resident at once.  With shared-device mode on, each persistent launch
 * runs under an inter-process lease (flock on a per-device file in /dev/shm), synchronously: two
 * ranks on one GPU take turns instead of dead-locking each other.  Initial value: environment
 * variable SCTC_SHARED_DEVICE (the Python mirror sets it when two ranks of a job report the same
 * physical device, sctc_device_pci_bus_id below); switched on automatically, for the rest of the
 * process, by the first SCTC_ERR_TIMEOUT (one line on stderr).  One rank per GPU pays nothing. */
int sctc_set_shared_device(int32_t on);
int sctc_shared_device(void);
/* PCI bus id ("0000:c1:00.0") of HIP device `device` (-1: the current one): the PHYSICAL identity of
 * the GPU, the same string in every process whatever HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
 * renumbering the launcher applied.  The reference selects its GPU by ordinal (runNNet.py:117-120,
 * cm.cuda_set_device) and never needs to know; the data-parallel ranks exchange this id to find out
 * whether two of them sit on one GPU (stanford-ctc_amd/dist_sgd.py).  Writes a NUL-terminated string. */
int sctc_device_pci_bus_id(int32_t device, char* out, int32_t out_len);

/* ---- CTC: ctc_fast/ctc-loss/ctc_fast.pyx ------------------------------- */

/* Layout of a batch of utterances over the rows of `probs`/`grad`:
 *   rowbase_dev == NULL : utterance b owns rows frame_off[b] .. frame_off[b]+T_b[b]-1
 *                         (frame t = row frame_off[b]+t) -- the reference's (A,T) F-order
 *                         array is exactly one such block with ld == A;
 *   rowbase_dev != NULL : packed time-major minibatch, frame t of utterance b is row
 *                         rowbase_dev[t] + frame_off[b]  (utterances sorted by length,
 *                         frame_off[b] = rank of b; used by the BRNN engine).            */
typedef struct sctc_ctc_batch {
    int32_t B;                 /* utterances */
    int32_t A;                 /* alphabet size incl. blank (params.shape[0], ctc_fast.pyx:24) */
    int32_t blank;             /* blank id (ctc_fast.pyx:14) */
    int32_t dtype;             /* SCTC_F32 | SCTC_F64: type of probs, grad and the lattices */
    int64_t ld;                /* row stride of probs/grad in elements (>= A) */
    const int32_t* T_b;        /* host [B]  frames per utterance  (params.shape[1]) */
    const int32_t* U_b;        /* host [B]  labels per utterance  (seq.shape[0]), >= 1 */
    const int64_t* frame_off;  /* host [B]  see above */
    const int32_t* labels;     /* host, concatenated label ids (seq) */
    const int64_t* label_off;  /* host [B]  first label of utterance b in `labels` */
    const int32_t* rowbase_dev;/* device [max T] or NULL */
} sctc_ctc_batch;

/* bytes of device workspace sctc_ctc_loss_batch needs for this batch (0: the batch is rejected, sctc_last_error()).
 * No bound on the label length or the alphabet (round 5; ctc_fast.pyx:22-32 has none): label rows of up to 512
 * lattice states (2U+1) keep ONE packed half lattice per direction (the alpha / beta / gradient kernel of
 * csrc/ctc_fused.hip), rows of up to 2048 states two float64 lattices of 64 K states per frame
 * (csrc/ctc_kernels.hip), longer rows and alphabets beyond 256 symbols two float64 lattices of round_up(2U+2, 64)
 * states per frame (csrc/ctc_generic.hip). */
size_t sctc_ctc_workspace_bytes(const sctc_ctc_batch* batch);

/* ctc_loss(params, seq, blank) of ctc_fast.pyx:13-152 for B utterances at once.
 *   probs_dev : softmax outputs (the reference's `params`), rows as described above
 *   grad_dev  : d cost / d (pre-softmax activations), same layout as probs (ctc_fast.pyx:138-145);
 *               zeros for utterances with skip != 0
 *   cost_dev  : device double[B], -llForward (ctc_fast.pyx:152); +inf when T is too short
 *               for the label sequence (empty band, log(0))
 *   skip_dev  : device int32[B], 1 where the reference takes its except-branch
 *               (a frame normaliser == 0: infeasible alignment / zero-probability label) */
int sctc_ctc_loss_batch(const sctc_ctc_batch* batch, const void* probs_dev, void* grad_dev,
                        double* cost_dev, int32_t* skip_dev, void* workspace_dev,
                        size_t workspace_bytes, void* stream);

/* brnnet.py:161-168 softmax over the alphabet for every frame (row):
 * subtract the row max, exp, scale by 1/sum.  In place allowed. f32 only. */
int sctc_softmax_rows(const float* logits_dev, float* probs_dev, int64_t rows, int32_t A,
                      int64_t ld, void* stream);

/* decode_best_path argmax stage (ctc_fast.pyx:165): per-row argmax over A, first maximum wins */
int sctc_argmax_rows(const void* probs_dev, int32_t dtype, int32_t* best_dev, int64_t rows,
                     int32_t A, int64_t ld, void* stream);

/* ---- BRNN: ctc_fast/nnets/brnnet.py NNet ------------------------------- */

typedef struct sctc_brnn_config {
    int32_t input_dim;       /* inputDim   brnnet.py:10 */
    int32_t output_dim;      /* outputDim  (alphabet incl. blank) */
    int32_t layer_size;      /* layerSize  */
    int32_t num_layers;      /* numLayers  */
    int32_t temporal_layer;  /* temporalLayer; <=0 or >=numLayers means none (brnnet.py:27-30) */
    int32_t max_frames;      /* capacity in frames summed over the minibatch (maxBatch for B=1) */
    int32_t max_utts;        /* capacity in utterances per call (1 = reference semantics) */
    float max_act;           /* maxAct = 20.0 (brnnet.py:32); <= 0 disables the ceiling (rnnetcpu.py) */
    float reg;               /* L2 coefficient (brnnet.py:22) */
    int32_t train;           /* 0: forward-only model (train=False) */
    int32_t operand_dtype;   /* SCTC_F32 (0, default): the reference's fp32 arithmetic everywhere.
                                SCTC_F16: the "fp16 activations" configuration (BASELINE configs[4]):
                                the operands of every time-batched contraction and of the
                                6..16-utterance recurrent step are rounded to 16 bit (float16 in
                                the forward pass, bfloat16 in the backward pass: deltas need
                                fp32's exponent range), products exact, accumulation fp32;
                                parameters, gradients, activations in memory, softmax and the
                                CTC lattices (float64) are unchanged.
                                SCTC_BF16X3: fp32 semantics everywhere (same tolerances as SCTC_F32);
                                only the time-batched contractions change instruction: each fp32
                                operand is split exactly into three bfloat16 terms and six cross
                                products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
                                (dropped terms <= 2^-23 of a product); recurrent step, softmax, CTC
                                and the update stay on the fp32 / fp64 paths */
} sctc_brnn_config;

/* One parameter tensor of `stack` (brnnet.py:58-59,71-72): order
 * [W1,b1] ... [W_{NL+1},b_{NL+1}] [Wf] [Wb]; stored row-major [rows][ld] inside the
 * flat parameter buffer (ld = rows/cols rounded up, zero padded). */
typedef struct sctc_tensor_info {
    int64_t offset;  /* element offset in the flat params / grads buffers */
    int32_t rows, cols, ld, kind; /* kind: 0 weight, 1 bias (cols == 1, ld == 1), 2 recurrent */
} sctc_tensor_info;

typedef struct sctc_brnn_sizes {
    int64_t param_elems;      /* floats in the flat (padded) parameter buffer */
    int64_t param_count;      /* unpadded count == NNet.paramCount() minus the dummy biases */
    size_t workspace_bytes;   /* device workspace for activations, deltas, CTC lattices */
    int32_t n_tensors;        /* 2*(NL+1) (+2 when there is a temporal layer) */
} sctc_brnn_sizes;

typedef struct sctc_brnn* sctc_brnn_t;

int sctc_brnn_query(const sctc_brnn_config* cfg, sctc_brnn_sizes* out);
/* NNet.__init__/initParams buffers (brnnet.py:10-86).  params/grads: device float[param_elems]
 * owned by the caller for the lifetime of the handle (grads may be NULL when !train). */
int sctc_brnn_create(const sctc_brnn_config* cfg, float* params_dev, float* grads_dev,
                     void* workspace_dev, size_t workspace_bytes, sctc_brnn_t* out);
int sctc_brnn_destroy(sctc_brnn_t h);
int sctc_brnn_tensor_info(sctc_brnn_t h, int32_t index, sctc_tensor_info* out);

/* Minibatch descriptor.  feats_dev: float [sum T][input_dim], utterances
 * concatenated in caller order, each one the reference's (inputDim,T) F-order array. */
typedef struct sctc_minibatch {
    int32_t B;
    const int32_t* T_b;       /* host [B] */
    const float* feats_dev;
    const int32_t* labels;    /* host concatenated (NULL for forward only) */
    const int32_t* U_b;       /* host [B] */
} sctc_minibatch;

#define SCTC_FLAG_SYNC_SKIP 1   /* reference B=1 semantics: look at `skip` on the host before
                                   the backward pass; if every utterance is skipped, return
                                   with the gradients untouched (brnnet.py:185-186) */
#define SCTC_FLAG_ACCUMULATE 2  /* add into grads instead of overwriting */
#define SCTC_FLAG_NO_REG_GRAD 4 /* leave the L2 term reg*W (brnnet.py:197-198,244-247) out of the
                                   gradients: minibatch / data-parallel callers add it ONCE, after
                                   the all-reduce and the 1/n_valid scaling (sctc_sumsq_reg,
                                   sctc_nesterov_step_reg) */

/* NNet.costAndGrad(data, labels) (brnnet.py:117-249) for a minibatch: forward,
 * softmax + CTC, backward.  Gradients are SUMMED over utterances (and frames) into
 * grads_dev; skipped utterances contribute zero.
 *   cost_host/skip_host [B] (optional, forces a stream sync), caller order;
 *   costs include nothing of the L2 term: *regcost_host (optional) receives
 *   sum (reg/2)*||w||^2 over all weight tensors (brnnet.py:178-183).                     */
int sctc_brnn_cost_and_grad(sctc_brnn_t h, const sctc_minibatch* mb, int32_t flags,
                            double* cost_host, int32_t* skip_host, double* regcost_host,
                            void* stream);
/* same, results stay on the device (no host sync): cost_dev double[B], skip_dev int32[B].
 * The caller must follow up with sctc_brnn_check before trusting the results. */
int sctc_brnn_cost_and_grad_async(sctc_brnn_t h, const sctc_minibatch* mb, int32_t flags,
                                  double* cost_dev, int32_t* skip_dev, void* stream);
/* Data-parallel overlap ("bucket by layer and launch as each wgrad GEMM finishes",
 * output layer first, brnnet.py:191-193): the hipEvent_t the engine records on the step's stream
 * right after the gradient of parameter tensor `index` is final (a weight tensor's event covers
 * its bias; NULL for bias indices), and hipStreamWaitEvent for a caller that only holds raw
 * stream pointers -- a side stream waits for tensor i's event and starts its all-reduce while
 * the backward pass continues.  The weight gradients of the layers above the temporal layer
 * are final before the BPTT recurrence starts; their events are recorded only after it has retired, so that no
 * collective kernel occupies compute units while the persistent BPTT grid (which needs all of
 * its workgroups resident at once) is being placed. */
void* sctc_brnn_grad_event(sctc_brnn_t h, int32_t index);
int sctc_stream_wait_event(void* stream, void* event);
/* SURVEY 8(e) (the reference has no multi-GPU path): the ONE exchange step of the data-parallel path as a C entry --
 * the sum of the weight gradients over the ranks of an RCCL communicator, overlapped with the backward pass that
 * sctc_brnn_cost_and_grad_async queued on `compute_stream`.  `side_stream` (a stream of its own) waits for the
 * event of each layer's gradient in the order the backward pass finishes them (output layer first; the layers above
 * the temporal layer and the recurrent pair after BPTT has retired) and starts ncclAllReduce(sum, float, in place)
 * on that layer's slice of the flat gradient buffer; `side_dev` (nullable: double[side_count] on the device, e.g.
 * [n_valid, cost_sum, regcost, has_regcost], produced on the compute stream) is reduced last; on return
 * `compute_stream` has been made to wait for all of it (nothing is synchronised with the host).
 *   rccl_comm        an ncclComm_t the caller created for this rank's device (ncclCommInitRank), one per rank
 *   backward_queued  0: this rank queued no backward pass this step (empty shard, gradient buffer zeroed on the
 *                    compute stream): the side stream is ordered behind the compute stream as a whole
 * librccl.so is resolved with dlopen at first use (libsctc_hip.so does not link against it): SCTC_ERR_STATE if it is
 * not there.  The Python host side (dist_sgd.allreduce_overlapped) issues the same sequence through
 * torch.distributed; this entry is for hosts that own their communicator. */
int sctc_brnn_allreduce_grads(sctc_brnn_t h, void* rccl_comm, void* compute_stream, void* side_stream,
                              double* side_dev, int32_t side_count, int32_t backward_queued);
/* synchronises `stream` and reports a failure of the asynchronous work queued on it:
 * SCTC_ERR_TIMEOUT when a persistent recurrent kernel gave up waiting for its peers (results of
 * that call are then meaningless), SCTC_OK otherwise */
int sctc_brnn_check(sctc_brnn_t h, void* stream);
/* which recurrence the last step ran (diagnostics / tests): 1 = one persistent launch per pass,
 * 2 = the same under the device lease, 3 = the non-persistent fallback (one launch per time step;
 * taken when a persistent grid cannot be co-resident: layer sizes beyond 2048 units, CU masks, a
 * step re-run after SCTC_ERR_TIMEOUT, SCTC_REC_VARIANT=3), 0 = none yet; *retries = steps of this
 * handle that were re-run after a timeout.  Any pointer may be NULL. */
int sctc_brnn_recurrent_path(sctc_brnn_t h, int32_t* forward_path, int32_t* bptt_path, int32_t* retries);

/* NNet(train=False).costAndGrad(data) (brnnet.py:171-173): probs_dev float [sum T][output_dim],
 * caller order, each utterance the reference's (outputDim,T) F-order probs */
int sctc_brnn_forward(sctc_brnn_t h, const sctc_minibatch* mb, float* probs_dev, void* stream);

/* The CTC scratch of a step lives in the model's workspace, which reserves 2048 lattice states (2U+1) per frame and
 * utterance slot.  The reference bounds no label row (ctc_fast.pyx:22-32 allocates (2U+1) x T per call): a minibatch
 * with a longer row may need more.  sctc_brnn_ctc_workspace_bytes says how much THIS minibatch needs (*needed) and
 * how much the handle currently offers (*reserved: its own share, or what was last set); a caller that finds
 * needed > reserved allocates a device buffer of at least `needed` bytes and hands it over with
 * sctc_brnn_set_ctc_workspace -- it replaces the built-in share for all later steps and stays OWNED BY THE CALLER, who
 * keeps it alive until it sets another one, resets (dev = NULL) or destroys the handle.  Without this a step whose
 * CTC scratch does not fit returns SCTC_ERR_WORKSPACE.  (nnets/brnnet.py does this by itself: round 6.) */
int sctc_brnn_ctc_workspace_bytes(sctc_brnn_t h, const sctc_minibatch* mb, size_t* needed, size_t* reserved);
int sctc_brnn_set_ctc_workspace(sctc_brnn_t h, void* workspace_dev, size_t workspace_bytes);

/* rocprof-free timing hooks: milliseconds spent in the phases of the last call, measured with hipEvents
 * on the step's stream.  sctc_brnn_set_profiling(h, 1): exact phase timers that synchronise the
 * stream at every phase change (perturbing); (h, 2): asynchronous -- one event per phase change is
 * recorded and only resolved by sctc_brnn_phase_ms after the step, no host sync is added, so it can
 * stay on while a benchmark times its steps (bench.py's roofline leg); (h, 0): off */
#define SCTC_PHASE_FWD_GEMM 0
#define SCTC_PHASE_FWD_REC 1
#define SCTC_PHASE_CTC 2
#define SCTC_PHASE_BWD_GEMM 3
#define SCTC_PHASE_BWD_REC 4
#define SCTC_PHASE_OTHER 5
#define SCTC_N_PHASES 6
int sctc_brnn_set_profiling(sctc_brnn_t h, int32_t enable);
/* diagnostics only: device pointer / shape of an internal fp32 matrix of the last call (which:
 * 0..numLayers = hActs[i] (0 = the packed input), 100 / 101 = hActsFor / hActsBack, 200 = the delta
 * entering layer 1); rows follow the packed time-major layout (one utterance: row t = frame t).
 * Lets a test separate the error of one contraction from the error of its operands. */
int sctc_brnn_debug_buffer(sctc_brnn_t h, int32_t which, void** dev_ptr, int64_t* rows, int64_t* cols,
                           int64_t* ld);
/* diagnostics only: per-step s_memtime stamps of the recurrent kernel when SCTC_REC_DEBUG=1 */
int sctc_brnn_debug_read(sctc_brnn_t h, uint32_t* out, int32_t n_words);
int sctc_brnn_phase_ms(sctc_brnn_t h, float* ms_out /* [SCTC_N_PHASES] */);
/* algorithmic FLOPs of one cost_and_grad over this minibatch, SURVEY 8(d) formula
 * (total, and the part in the time-batched GEMMs / in the recurrent steps) */
int sctc_brnn_flops(sctc_brnn_t h, const sctc_minibatch* mb, double* total, double* gemm,
                    double* recurrent);

/* cm.dot(A, B, target=C) of cudamat on row-major fp32 device matrices (brnnet.py:140 forward,
 * :196 weight gradient, :204 delta propagation, :227-230 recurrent weight gradient):
 *   C[M][N] = sum_k A(m,k) B(k,n) (+ bias[n]) (relu)
 *   a_kcontig: A(m,k) = A[m*lda+k], else A[k*lda+m];  b_kcontig: B(k,n) = B[n*ldb+k], else B[k*ldb+n]
 * fp32 operands, fp32 accumulate on the matrix cores (v_mfma_f32_32x32x2_f32).
 * workspace (optional) enables deterministic split-K for small M*N. */
int sctc_gemm_f32(const float* A_dev, int64_t lda, int32_t a_kcontig, const float* B_dev,
                  int64_t ldb, int32_t b_kcontig, float* C_dev, int64_t ldc, int32_t M, int32_t N,
                  int32_t K, const float* bias_dev, int32_t relu, void* workspace_dev,
                  size_t workspace_bytes, void* stream);

/* The same contraction with both operands ROUNDED to a 16-bit type (operand_dtype = SCTC_F16 or
 * SCTC_BF16, round-to-nearest-even) on their way to the matrix cores and fp32 accumulation
 * (v_mfma_f32_32x32x16_f16 / _bf16): the GEMM of the "fp16 activations" configuration
 * (BASELINE configs[4]).  Operands and result stay fp32 in memory.
 * operand_dtype = SCTC_BF16X3: no rounding -- the three-term split described at sctc_brnn_config.
 * operand_dtype | SCTC_OPERANDS_16BIT (SCTC_F16 / SCTC_BF16 only): the operands are 16-bit in memory already
 * (both in the SAME layout: a_kcontig == b_kcontig, lda / ldb multiples of 8 elements); large problems
 * then run on the LDS-DMA staged kernel (csrc/gemm_g16.hip), the path the engine's GEMMs take. */
int sctc_gemm_h16(const float* A_dev, int64_t lda, int32_t a_kcontig, const float* B_dev,
                  int64_t ldb, int32_t b_kcontig, float* C_dev, int64_t ldc, int32_t M, int32_t N,
                  int32_t K, const float* bias_dev, int32_t relu, int32_t operand_dtype,
                  void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- what sgd.py needs from cudamat objects (sgd.py:21-23,93-106,129-141,161) ---- */

/* y += alpha * x   (CUDAMatrix.add_mult, NNet.updateParams brnnet.py:251-256) */
int sctc_axpy(float* y_dev, const float* x_dev, float alpha, int64_t n, void* stream);
/* x *= alpha       (CUDAMatrix.mult) */
int sctc_scale(float* x_dev, float alpha, int64_t n, void* stream);
/* sum of squares of n floats into *out_dev (double); euclid_norm()**2 over the whole
 * gradient stack in one launch instead of 14 host syncs (sgd.py:103-107) */
int sctc_sumsq(const float* x_dev, int64_t n, double* out_dev, void* workspace_dev,
               size_t workspace_bytes, void* stream);
/* Fused Nesterov step of sgd.py:129-141,161 on the flat buffers:
 *   alph = alpha * min(1, maxGNorm/gnorm)  with gnorm = sqrt(*sumsq_dev) * grad_scale
 *   v = mom*v - alph*grad_scale*g ;  w += v                                             */
int sctc_nesterov_step(float* w_dev, float* v_dev, const float* g_dev, int64_t n, float mom,
                       float alpha, float max_gnorm, float grad_scale, const double* sumsq_dev,
                       void* stream);

/* Minibatch / data-parallel form of the two calls above with the L2 term applied exactly once:
 * the effective gradient is  e = grad_scale * g + reg * (w + mom * v)  (g = SUM of the data
 * gradients over the minibatch, all-reduced over the ranks; grad_scale = 1/n_valid;
 * brnnet.py:197-198 adds reg*W to each utterance's gradient AT THE LOOK-AHEAD POINT w + mom*v
 * where sgd.py:91-95 evaluates it; w here is the weight after the look-ahead was undone
 * (sgd.py:97-100), v the velocity before this step's update; v_dev NULL = plain reg*w).
 *   sctc_sumsq_reg:          *out_dev = sum e^2 (float64 accumulation, two deterministic stages)
 *   sctc_nesterov_step_reg:  alph = alpha * min(1, maxGNorm / sqrt(*sumsq_dev));
 *                            v = mom*v - alph*e ;  w += v                                   */
/*   noreg_ranges_host: n_ranges (<= 128) ascending, disjoint element ranges [beg, end) of the flat
 *                      buffers that carry NO L2 term -- the biases (brnnet.py:197-200 regularises w only) */
int sctc_sumsq_reg(const float* g_dev, const float* w_dev, const float* v_dev, float mom,
                   float grad_scale, float reg, int64_t n,
                   const int64_t* noreg_ranges_host, int32_t n_ranges, double* out_dev,
                   void* workspace_dev, size_t workspace_bytes, void* stream);
int sctc_nesterov_step_reg(float* w_dev, float* v_dev, const float* g_dev, int64_t n, float mom,
                           float alpha, float max_gnorm, float grad_scale, float reg,
                           const int64_t* noreg_ranges_host, int32_t n_ranges,
                           const double* sumsq_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCTC_H_ */
