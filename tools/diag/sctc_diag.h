/*
 * sctc_diag.h -- C ABI of libsctc_diag.so: hardware probes used while developing and measuring
 * the kernels of libsctc_hip.so.  NOT part of the drop-in boundary (include/sctc.h): nothing in
 * stanford-ctc_amd/ loads this library; tests/gpu_diag.py, bench.py's "sustained peak" note and
 * __graft_entry__.smoke() do.  These entry points allocate (and free) their own scratch memory.
 */
#ifndef SCTC_DIAG_H_
#define SCTC_DIAG_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* sctc_diag_last_error(void);
/* runs the cross-lane / MFMA fragment-layout probes the kernels rely on; 0 = all as expected,
 * else a bitmask of the failed probes */
int sctc_selftest(void* stream);
/* hand-off latencies between workgroups on the same / on different XCDs (flag ping-pong per
 * polling-load scope, 1 KiB tagged payload).  results_host[10]: partner block ids (same, cross
 * XCD), then microseconds per round trip: same-XCD {sc0, sc1, sc0+sc1}, cross-XCD {sc0, sc1,
 * sc0+sc1}, tagged payload {same, cross}; -1 = timed out (a scope that never observes the store) */
int sctc_probe_fabric(float* results_host, int32_t n_results, void* stream);
/* sustained fp32 matrix-pipe rate of register-only MFMA loops on every SIMD: results_host[8] =
 * {TFLOP/s, ms} for v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 with constant operands,
 * then the same two with fresh random operands per MFMA group */
int sctc_probe_mfma(float* results_host, int32_t n_results, void* stream);

/* flag-after-drain hand-off with IMMEDIATE plain loads by 15 consumer workgroups on other XCDs,
 * 20 000 fresh 1 KiB blocks.  results_host[9]: per mode {stale payload dwords, iterations (-1: timed
 * out), microseconds per iteration}; mode 0 = the recurrent kernels' protocol (drain, plain loads),
 * 1 = no drain (must show stale data), 2 = drain + L2-bypassing loads */
int sctc_probe_handoff(float* results_host, int32_t n_results, void* stream);
/* n_wgs workgroups of 256 threads that hold their compute units for `microseconds`: the stand-in
 * for a collective kernel on a side stream (one workgroup per channel, like RCCL) */
int sctc_diag_spin(void* stream, int32_t n_wgs, int32_t microseconds);

/* a stream whose kernels run only on the compute units whose bit is set in mask_words (n_words x 32
 * bits; hipExtStreamCreateWithCUMask) */
int sctc_diag_stream_cu_mask(const uint32_t* mask_words, int32_t n_words, void** stream_out);
int sctc_diag_stream_destroy(void* stream);
/* n_wgs workgroups of 256 threads, each holding its CU for hold_us: out_host[4 * i + {0,1,2,3}] =
 * XCC id, shader engine, shader array, CU id of workgroup i (synchronous) */
int sctc_diag_where(int32_t* out_host, int32_t n_wgs, int32_t hold_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif
