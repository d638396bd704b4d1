// Persistent, weight-stationary kernel for the bi-directional recurrent layer.
//
// Reference (brnnet.py:148-152, :215-224): 2(T-1) SGEMV + 2(T-1) clip launches
// per pass, each SGEMV re-reading the 13.3 MB weight matrix from HBM.
// Here: ONE launch per pass.  Workgroup `wg` of group g (g = direction) owns 16
// output units; its 16 x H slab of W (or W^T for BPTT) is loaded ONCE into LDS
// (114 KiB at H = 1824) in MFMA-fragment order and stays there for all T steps.
// The minibatch (B utterances) is the N dimension of a 16x16x4 f32 MFMA, so one
// time step is a (16 x H) . (H x B) product per workgroup, split over the 4 waves
// as 2 K-halves x 2 utterance groups and reduced through LDS.  A step's result
// is published twice: into the [rows][H] activation matrix used by the
// time-batched GEMMs, and into the exchange buffer, a chunk-major copy
// [16-unit chunk][exchange row][16]: chunk c is written only by workgroup c, a
// step's rows form one contiguous 256-byte-aligned block per chunk, and readers
// fetch them as 16-byte-per-lane loads that land directly in MFMA B-fragment
// order.  All loads of a step are issued before its first MFMA (one wave per
// SIMD owns the whole register file).
//
// Step synchronisation (per direction): an all-gather of per-producer step flags.
// Producer: exchange stores -> s_waitcnt vmcnt(0) per wave -> workgroup barrier ->
// one lane publishes flag[wg] = step+1.  Consumer: one wave polls all H/16 flags
// (relaxed agent-scope loads, two per lane) until every producer has arrived, then
// reads with plain cacheable loads.  Every exchange address is written exactly once
// per launch and its cache line holds data of one step only, so neither a CU's L1 nor
// an XCD's L2 can hold a stale copy: no acquire invalidate is needed and the 14
// same-XCD readers of a line share one fabric fetch.
//   sync_mode 0: plain payload stores + agent-scope release fence before the flag;
//   sync_mode 1: write-through (sc1) payload stores, no fence.
// Every spin is bounded (0.5 s) and raises an error word the host turns into
// SCTC_ERR_TIMEOUT; flags are zeroed by a memset node before each launch.
//
// The K index inside a 16-chunk is permuted (k = 16c + 4*(lane>>4) + q for MFMA q)
// identically for W and x, so both operands are 16-byte vector loads.
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <string>
#include <type_traits>
#include <sys/stat.h>
#include <sys/file.h>
#include <time.h>
#include <unistd.h>

#include "common.h"
#include "recurrent.h"
#include "xlane.h"

namespace sctc {

typedef void (*RecKernel)(RecArgs);
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// A wait only ever expires when some workgroup of the grid is not running (device shared with
// another persistent launch, CU masking): 0.5 s of the 100 MHz clock is 5 orders of magnitude above
// a step and still fails a caller quickly.  The host then retries (device lease, per-step fallback).
static constexpr unsigned long long SPIN_TIMEOUT_TICKS = 50000000ull;

// s_sleep units (64 cycles) a consumer waits before its FIRST poll of a step (RecArgs.poll_delay).
// A poll is a fabric round trip (~1 us); one issued the moment the own results are published is
// almost always too early and delays the next one by that round trip.  Measured (sentinel kernel,
// one utterance): 5 units = -16 % per step at H = 1824 and -3 % at H = 2048, but +8 % at H = 1024
// (fewer producers, less skew): the launcher picks 5 above 1024 units, 0 below; the flag kernels
// gain 3 % at 32 utterances with 5.
__device__ __forceinline__ void first_poll_delay(int units)
{
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(1);
}

// Bounded spinning.  Called by a whole wave every few hundred polls: gives up when some
// workgroup has already raised the error word or when this wait has lasted SPIN_TIMEOUT_TICKS,
// in which case it raises the error word itself (the host turns it into SCTC_ERR_TIMEOUT).
__device__ __forceinline__ bool spin_expired(unsigned* err, unsigned long long t0, int lane)
{
    bool give_up = false;
    if (lane == 0) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
            give_up = true;
        else if (wall_clock64() - t0 > SPIN_TIMEOUT_TICKS) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            give_up = true;
        }
    }
    return __any(give_up) != 0;
}

// One step's result for four consecutive units of one utterance (brnnet.py:146-152 forward:
// clip(z + W h, 0, maxAct); :208-224 BPTT: (delta + W^T delta') masked by 0 < h < maxAct).
__device__ __forceinline__ float4 step_result(const float4& pre, const float4& s, const float4& act,
                                              bool bptt, float hi)
{
    float4 o;
    if (!bptt) {
        o.x = fminf(fmaxf(pre.x + s.x, 0.f), hi);
        o.y = fminf(fmaxf(pre.y + s.y, 0.f), hi);
        o.z = fminf(fmaxf(pre.z + s.z, 0.f), hi);
        o.w = fminf(fmaxf(pre.w + s.w, 0.f), hi);
    } else {
        o.x = (act.x > 0.f && act.x < hi) ? pre.x + s.x : 0.f;
        o.y = (act.y > 0.f && act.y < hi) ? pre.y + s.y : 0.f;
        o.z = (act.z > 0.f && act.z < hi) ? pre.z + s.z : 0.f;
        o.w = (act.w > 0.f && act.w < hi) ? pre.w + s.w : 0.f;
    }
    return o;
}

// For how many steps a tile of 16 utterances keeps its KB per chunk in LANE order ([k quarter][utterance][4 units]: a wave
// access is one contiguous KB; row-major, [utterance][16 units], makes every 16-lane group of a load touch eight lines):
// as long as the tile's 16 exchange rows all lie inside the step's block.  The engine (brnn_engine.hip, plan_minibatch)
// rounds a step's block up to 16 rows from 17 alive utterances on and to 4 below -- so (round 6) every tile but the first
// keeps lane order for its WHOLE life (while one of its utterances is alive more than 16 of the minibatch are), dead
// slots simply stay unwritten; the first tile of the first launch while at least 13 utterances are alive (13..16 round up
// to 16 rows).  Rounds 5: only while all 16 utterances of the tile were alive -- a ragged minibatch ran most of its steps
// row-major (7.7 instead of 6.4 us per step at 32 utterances of T/2..T frames).
// uT: the lengths of the tile's utterances, lane uj = utterance uj (0 beyond the minibatch); tile: launch-local index.
__device__ __forceinline__ int lane_order_steps(int uT, int tile, int b_off, int variant)
{
    if (variant == 46) return 0;        // row-major throughout (rounds 1-5a; bit-identical A/B)
    return (tile == 0 && b_off == 0) ? __builtin_amdgcn_readlane(uT, 12) : __builtin_amdgcn_readlane(uT, 0);
}

__device__ __forceinline__ float4 ld_x(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off,
                                       unsigned chunk_off)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, chunk_off, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                       __uint_as_float(v[3]));
}

__device__ __forceinline__ void st_x(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off,
                                     unsigned chunk_off, float4 v, int sync_mode)
{
    u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z),
               __float_as_uint(v.w)};
    if (sync_mode == 1)
        __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, lane_off, chunk_off, 16 /* sc1 */);
    else
        __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, lane_off, chunk_off, 0);
}

// all threads call; publishes this workgroup's stores of the step as flag[wg] = value
__device__ __forceinline__ void publish_step(unsigned* flag, unsigned value, int sync_mode)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sync_mode == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // keep the wait behind buffer_wbl2
        }
        __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// all threads call; returns once every one of the `nwg` producers has published >= target
__device__ __forceinline__ void wait_all(unsigned* flags, int nwg, unsigned target, unsigned* err,
                                         int poll_delay)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        first_poll_delay(poll_delay);
        for (;;) {
            unsigned f0 = target, f1 = target;
            if (lane < nwg)
                f0 = __hip_atomic_load(flags + lane * REC_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane + 64 < nwg)
                f1 = __hip_atomic_load(flags + (lane + 64) * REC_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(f0 >= target && f1 >= target)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
                if (spin_expired(err, t0, lane)) break;
            }
        }
    }
    __syncthreads();
}

#define SCTC_MFMA4(ACC, A, X)                                                                 \
    ACC[0] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (X).x, ACC[0], 0, 0, 0);             \
    ACC[1] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (X).y, ACC[1], 0, 0, 0);             \
    ACC[2] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (X).z, ACC[2], 0, 0, 0);             \
    ACC[3] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (X).w, ACC[3], 0, 0, 0);

// NTW: utterance tiles (16 each) per wave.  NCHH: 16-unit K chunks per wave when known at
// compile time (Hp == 32*NCHH: fully unrolled, branch-free), 0 = run-time loop.
// PIPE (round 5, NCHH > 0 only): the exchange loads of batch k+1 are issued before the MFMAs of batch k (two
// half-size batches in flight, the compiler's counted s_waitcnt vmcnt in front of each batch's first MFMA) instead
// of all loads of a batch, a fence, then its MFMAs: at 64 / 128 utterances a step is 7 / 14 us of matrix work and
// 3.5 / 7 us of exchange traffic through the CU's one vector-memory pipe, which used to run one after the other.
// Same MFMAs on the same accumulators in the same order: bit-identical results.
template <int NTW, int NCHH, bool PIPE = false>
__global__ __launch_bounds__(256, 1) void brnn_recurrent_kernel(RecArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    const int g = blockIdx.x & 1, wg = blockIdx.x >> 1;
    const int Hp = p.Hp, nch = Hp >> 4, nwg = nch;
    const int nch_half = NCHH > 0 ? NCHH : (nch >> 1);
    const int row0 = wg * 16;
    const int uj = lane & 15, kq = lane >> 4;
    const int kh = wave >> 1, ng = wave & 1;
    const int sync_mode = p.sync_mode;
    float4* Wl = lds4;                      // [nch][64] fragment-ordered weight slab
    float4* red = lds4 + (size_t)nch * 64;  // [2 ng][NTW][64] partial sums of the upper K half

    // ---- stationary weights: Wl[c][lane] = { Wop[row0 + (lane&15)][16c + 4*(lane>>4) + q] }_q
    {
        const float* W = p.W[g];
        for (int c = wave; c < nch; c += 4) {
            float4 v;
            if (!p.transpose) {
                v = *reinterpret_cast<const float4*>(W + (int64_t)(row0 + uj) * p.ldw + 16 * c + 4 * kq);
            } else {
                const float* col = W + (int64_t)(16 * c + 4 * kq) * p.ldw + row0 + uj;
                v.x = col[0];
                v.y = col[p.ldw];
                v.z = col[2 * p.ldw];
                v.w = col[3 * p.ldw];
            }
            Wl[c * 64 + lane] = v;
        }
    }
    __syncthreads();

    const bool desc = p.descending[g] != 0;
    const float* pre = p.pre[g];
    const float* act = p.act[g];
    float* out = p.out[g];
    const int64_t ld = p.ld;
    const float hi = p.max_act > 0.f ? p.max_act : INFINITY;
    unsigned* flags = p.counters + 32 + g * 128 * REC_FLAG_STRIDE;  // [2][128] step flags
    unsigned* err = p.counters + 2;
    const unsigned chunk_stride = (unsigned)p.n_xrows * 64u;  // bytes per 16-unit chunk
    float* xg = p.xbuf + (size_t)g * p.n_xrows * Hp;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        xg, 0, (int)((size_t)p.n_xrows * Hp * sizeof(float)), 0x00020000);

    int ub[NTW], uT[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        ub[i] = (ng + 2 * i) * 16 + uj;
        uT[i] = ub[i] < p.B ? p.T_b[ub[i]] : 0;
    }
    // Steps at which a tile's 16 exchange rows lie inside the step's block (lane_order_steps above) keep the tile's 1 KB of a
    // chunk as [k quarter][utterance][4 units]: lane l = uj + 16 kq then reads / writes byte 16 l, a wave one contiguous KB.
    int tile_T[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) tile_T[i] = lane_order_steps(uT[i], ng + 2 * i, p.b_off, p.variant);
    const int c_beg = kh * nch_half, c_end = c_beg + nch_half;
    // Which K half adds the other's partial sums to its own and stores the step's result of tile i.  Rounds 1-4: the
    // lower half, every tile (the upper half's waves sat out the epilogue: 1.8 us of a 128-utterance step).  Round 5:
    // the halves take alternate tiles -- own + other is the same float addition either way round, bit-identical.
    const bool split_epilogue = NTW >= 2 && p.variant != 40;
    auto finishes = [&](int i) -> bool { return split_epilogue ? ((i & 1) == kh) : (kh == 0); };

    const int dbg_sel = (p.debug && g == 0 && tid == 0) ? (wg == 0 ? 0 : (wg == nwg - 1 ? 1 : -1)) : -1;
    auto stamp = [&](int j, int k) {
        if (dbg_sel >= 0 && j >= 64 && j < 80)
            p.debug[(dbg_sel * 16 + (j - 64)) * 8 + k] = (unsigned)clock64();
        if (p.debug && tid == 0 && j == 70) {   // every workgroup: global 100 MHz clock, chain tag
            p.debug[REC_DEBUG_ALL_OFF + blockIdx.x * 8 + k] = (unsigned)wall_clock64();
            if (k == 0) p.debug[REC_DEBUG_ALL_OFF + blockIdx.x * 8 + 7] = 1u + (unsigned)g;
        }
    };

    for (int j = 0; j < p.Tmax; ++j) {
        stamp(j, 0);
        // four independent accumulators per tile: the 16x16x4 f32 MFMA has a 40-cycle
        // dependent latency against a 32-cycle issue interval
        f32x4 acc[NTW][4];
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] = {0.f, 0.f, 0.f, 0.f};
        bool active[NTW];
        int64_t orow[NTW];
        unsigned xin[NTW], xout[NTW];  // byte offsets (within a chunk) of the previous / this row
        const unsigned xb_cur = (unsigned)__builtin_amdgcn_readfirstlane(p.xbase[j]);
        const unsigned xb_prev = (unsigned)__builtin_amdgcn_readfirstlane(p.xbase[j > 0 ? j - 1 : 0]);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            active[i] = j < uT[i];
            const int t = desc ? uT[i] - 1 - j : j;
            orow[i] = active[i] ? (int64_t)p.rowbase[t] + p.b_off + ub[i] : 0;
            // exchange rows are indexed by STEP (not by frame): all rows of one 256-byte-aligned
            // block are written in the same step in both time orders.  Finished / empty slots
            // read exchange row 0 (valid memory, result discarded).
            const unsigned xrow = active[i] ? xb_cur + (unsigned)ub[i] : 0u;
            const unsigned prow = (active[i] && j > 0) ? xb_prev + (unsigned)ub[i] : 0u;
            xin[i] = prow * 64u + (unsigned)kq * 16u;
            xout[i] = xrow * 64u + (unsigned)kq * 16u;
            const unsigned tile_row = (unsigned)(ng + 2 * i) * 16u;
            if (j > 0 && j <= tile_T[i] && active[i])       // the tile was complete at step j - 1
                xin[i] = (xb_prev + tile_row) * 64u + (unsigned)lane * 16u;
            if (j < tile_T[i])
                xout[i] = (xb_cur + tile_row) * 64u + (unsigned)lane * 16u;
        }
        // prefetch the per-frame additive term (independent of the recurrence)
        float4 pre4[NTW], act4[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            pre4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            act4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (finishes(i)) {
                if (active[i]) {
                    pre4[i] = *reinterpret_cast<const float4*>(pre + orow[i] * ld + row0 + 4 * kq);
                    if (act) act4[i] = *reinterpret_cast<const float4*>(act + orow[i] * ld + row0 + 4 * kq);
                }
            }
        }

        if (j > 0) {
            wait_all(flags, nwg, (unsigned)j, err, p.poll_delay);
            stamp(j, 1);
            // All x loads of a batch are issued, unconditionally, before its first MFMA
            // (up to 64 float4 = 256 VGPRs per lane): one latency + streaming per step.
            if constexpr (NCHH > 0 && PIPE) {
                constexpr int XB = NCHH < 32 / NTW ? NCHH : 32 / NTW;     // chunks per batch; two batches of registers
                constexpr int NBAT = (NCHH + XB - 1) / XB;
                float4 x[2][XB][NTW];
#pragma unroll
                for (int u = 0; u < XB; ++u)
#pragma unroll
                    for (int i = 0; i < NTW; ++i) x[0][u][i] = ld_x(xrsrc, xin[i], (unsigned)(c_beg + u) * chunk_stride);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int bi = 0; bi < NBAT; ++bi) {
                    if (bi + 1 < NBAT) {
#pragma unroll
                        for (int u = 0; u < XB; ++u)
                            if ((bi + 1) * XB + u < NCHH) {
#pragma unroll
                                for (int i = 0; i < NTW; ++i)
                                    x[(bi + 1) & 1][u][i] = ld_x(xrsrc, xin[i], (unsigned)(c_beg + (bi + 1) * XB + u) * chunk_stride);
                            }
                    }
                    // the next batch's loads stay in front of this batch's MFMAs
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < XB; ++u)
                        if (bi * XB + u < NCHH) {
                            const float4 a = Wl[(c_beg + bi * XB + u) * 64 + lane];
#pragma unroll
                            for (int i = 0; i < NTW; ++i) { SCTC_MFMA4(acc[i], a, x[bi & 1][u][i]) }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (NCHH > 0) {
                constexpr int XB = NCHH < 64 / NTW ? NCHH : 64 / NTW;
#pragma unroll
                for (int cb = 0; cb < NCHH; cb += XB) {
                    float4 x[XB][NTW];
#pragma unroll
                    for (int u = 0; u < XB; ++u)
                        if (cb + u < NCHH) {
#pragma unroll
                            for (int i = 0; i < NTW; ++i)
                                x[u][i] = ld_x(xrsrc, xin[i], (unsigned)(c_beg + cb + u) * chunk_stride);
                        }
                    // keep every load ahead of the first MFMA (the scheduler would otherwise sink
                    // them next to their uses to save registers and re-expose the latency)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < XB; ++u)
                        if (cb + u < NCHH) {
                            const float4 a = Wl[(c_beg + cb + u) * 64 + lane];
#pragma unroll
                            for (int i = 0; i < NTW; ++i) { SCTC_MFMA4(acc[i], a, x[u][i]) }
                        }
                }
            } else {
                constexpr int XB = 64 / NTW;
                for (int cb = c_beg; cb < c_end; cb += XB) {
                    float4 x[XB][NTW];
#pragma unroll
                    for (int u = 0; u < XB; ++u) {
                        const int c = min(cb + u, c_end - 1);
#pragma unroll
                        for (int i = 0; i < NTW; ++i)
                            x[u][i] = ld_x(xrsrc, xin[i], (unsigned)c * chunk_stride);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < XB; ++u) {
                        const int c = cb + u;
                        if (c < c_end) {
                            const float4 a = Wl[c * 64 + lane];
#pragma unroll
                            for (int i = 0; i < NTW; ++i) { SCTC_MFMA4(acc[i], a, x[u][i]) }
                        }
                    }
                }
            }
            if (p.debug) {
                asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]));
                stamp(j, 2);
            }
            // fold the two K halves: a wave parks the partials of the tiles the OTHER half finishes in LDS
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                if (!finishes(i)) {
                    f32x4 s = (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
                    red[(ng * NTW + i) * 64 + lane] = make_float4(s[0], s[1], s[2], s[3]);
                }
            }
            __syncthreads();
            stamp(j, 3);
        }

        {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                if (!finishes(i) || !active[i]) continue;
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j > 0) {
                    const float4 r = red[(ng * NTW + i) * 64 + lane];
                    const f32x4 q = (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
                    s = make_float4(q[0] + r.x, q[1] + r.y, q[2] + r.z, q[3] + r.w);
                }
                const float4 o = step_result(pre4[i], s, act4[i], act != nullptr, hi);
                *reinterpret_cast<float4*>(out + orow[i] * ld + row0 + 4 * kq) = o;
                st_x(xrsrc, xout[i], (unsigned)wg * chunk_stride, o, sync_mode);
            }
        }
        stamp(j, 4);
        if (j + 1 < p.Tmax) publish_step(flags + wg * REC_FLAG_STRIDE, (unsigned)(j + 1), sync_mode);
        stamp(j, 5);
    }
}

// ---------------------------------------------------------------------------------------
// Two-chains-per-CU variant (17..32 utterances).  One step of the kernel above is a serial
// chain: flags -> exchange loads (a fabric round trip) -> MFMAs -> reduce -> stores -> flag;
// the matrix pipes idle during everything but the MFMAs.  Different utterances never interact,
// so the minibatch splits into independent chains of 16 utterances.  Here a workgroup owns
// 16 output units x ONE utterance tile, and two workgroups (of different chains) share a CU:
// while one waits for its exchange the other one computes.  (The dispatcher does not pack them
// two by two: at H = 1824 it puts the first 32 workgroups of an XCC on 32 different CUs and the
// other 25 next to them -- 200 CUs hold two workgroups, 56 hold one, none is idle; a workgroup
// alone on its CU needs 3.8 us from "flags seen" to "published", one that shares needs 5.1 us.)  Per workgroup: 4 waves = 4 K
// quarters (the 114 chunks of H = 1824 split 29/29/28/28); the 16 x H weight slab no longer
// fits twice into 160 KiB of LDS, so each wave keeps NREG of its chunks in registers and the
// rest in a wave-private LDS region.  Exchange traffic per CU and step is unchanged (2 x 16
// utterances), flags are per chain: counters[32 + (chain*128 + wg) * REC_FLAG_STRIDE] (own 32-byte
// slots: 114 writers and 456 pollers on four shared cache lines cost 8 % of the step).
template <int NCQ, int NREG>
__global__ __launch_bounds__(256, 2) void brnn_recurrent_q_kernel(RecArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    constexpr int NLDS = NCQ - NREG;        // chunks per wave held in LDS
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hp = p.Hp, nch = Hp >> 4, nwg = nch;
    // Block -> (chain, producer) map.  Workgroups are dealt to XCDs round-robin (block b runs on
    // XCD b % 8, probed by sctc_probe_fabric); the XCD-grouped map confines utterance tile 0 to
    // XCDs 0-3 and tile 1 to XCDs 4-7 (both directions each), so an XCD's L2 fetches the state
    // of two chains instead of four and a quarter of a chain's producers are L2-local; the two
    // workgroups that land on one CU (k and k+32 of an XCD) still belong to different chains.
    // Placement only affects speed: any block -> XCD assignment gives the same results.
    int chain, wg;
    if (p.variant != 2 && !p.linear_map) {
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3, P = nwg >> 1;   // P blocks per XCD
        const int gx = x >> 2, xl = x & 3;
        const int n0_even = (P + 1) >> 1, n0_odd = P >> 1;                  // chain-0 share per XCD
        const int n0 = (xl & 1) ? n0_odd : n0_even;
        const int cl = k < n0 ? 0 : 1;
        const int i = cl == 0 ? k : k - n0;
        int before = 0;   // producers of this chain on the XCDs before this one
        for (int q = 0; q < xl; ++q) {
            const int m0 = (q & 1) ? n0_odd : n0_even;
            before += cl == 0 ? m0 : P - m0;
        }
        chain = 2 * gx + cl;
        wg = before + i;
    } else {
        chain = blockIdx.x / nwg;
        wg = blockIdx.x - chain * nwg;
    }
    const int g = chain & 1, tile = chain >> 1;
    const int row0 = wg * 16;
    const int uj = lane & 15, kq = lane >> 4;
    const int sync_mode = p.sync_mode;
    // K split: `base` chunks per wave, the first `rem` waves take one more
    const int base = nch >> 2, rem = nch & 3;
    const int cnt = base + (wave < rem ? 1 : 0);
    const int c_beg = wave * base + min(wave, rem);
    float4* Wl = lds4 + (size_t)wave * NLDS * 64;     // [NLDS][64] wave-private, fragment order
    float4* red = lds4 + (size_t)4 * NLDS * 64;       // [3][64] partial sums of waves 1..3

    // ---- stationary weights: fragment of chunk c = { Wop[row0 + (lane&15)][16c + 4*(lane>>4) + q] }_q
    auto load_w = [&](int c) {
        const float* W = p.W[g];
        float4 v;
        if (!p.transpose) {
            v = *reinterpret_cast<const float4*>(W + (int64_t)(row0 + uj) * p.ldw + 16 * c + 4 * kq);
        } else {
            const float* col = W + (int64_t)(16 * c + 4 * kq) * p.ldw + row0 + uj;
            v.x = col[0];
            v.y = col[p.ldw];
            v.z = col[2 * p.ldw];
            v.w = col[3 * p.ldw];
        }
        return v;
    };
    float4 wreg[NREG > 0 ? NREG : 1];
#pragma unroll
    for (int u = 0; u < NREG; ++u) wreg[u] = load_w(c_beg + min(u, cnt - 1));
#pragma unroll 4
    for (int u = NREG; u < NCQ; ++u) Wl[(u - NREG) * 64 + lane] = load_w(c_beg + min(u, cnt - 1));
    __syncthreads();

    const bool desc = p.descending[g] != 0;
    const float* pre = p.pre[g];
    const float* act = p.act[g];
    float* out = p.out[g];
    const int64_t ld = p.ld;
    const float hi = p.max_act > 0.f ? p.max_act : INFINITY;
    unsigned* flags = p.counters + 32 + chain * 128 * REC_FLAG_STRIDE;
    unsigned* err = p.counters + 2;
    const unsigned chunk_stride = (unsigned)p.n_xrows * 64u;
    float* xg = p.xbuf + (size_t)g * p.n_xrows * Hp;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        xg, 0, (int)((size_t)p.n_xrows * Hp * sizeof(float)), 0x00020000);

    const int ub = tile * 16 + uj;
    const int uT = ub < p.B ? p.T_b[ub] : 0;
    // this chain is finished once its longest utterance (the tile's first) is
    const int Tchain = tile * 16 < p.B ? p.T_b[tile * 16] : 0;
    // exchange layout of the tile's KB per chunk: in lane order while its 16 rows lie inside the step's block (j < tile_T),
    // row-major after that -- lane_order_steps above (SCTC_REC_VARIANT=46: row-major throughout)
    const int tile_T = lane_order_steps(uT, tile, p.b_off, p.variant);

    const int dbg_sel = (p.debug && chain == 0 && tid == 0) ? (wg == 0 ? 0 : (wg == nwg - 1 ? 1 : -1)) : -1;
    auto stamp = [&](int j, int k) {
        if (dbg_sel >= 0 && j >= 64 && j < 80)
            p.debug[(dbg_sel * 16 + (j - 64)) * 8 + k] = (unsigned)clock64();
        if (p.debug && tid == 0 && j == 70) {   // every workgroup: global 100 MHz clock, chain tag
            p.debug[REC_DEBUG_ALL_OFF + blockIdx.x * 8 + k] = (unsigned)wall_clock64();
            if (k == 0) p.debug[REC_DEBUG_ALL_OFF + blockIdx.x * 8 + 7] = 1u + (unsigned)chain;
#ifdef SCTC_REC_WHERE
            // placement (diagnostics build, tests/gpu_diag.py recdbg1): XCC id and HW_ID bits 15:8 (cu 11:8,
            // sh 12, se 15:13) of the CU this workgroup runs on.  Measured (round 3): the dispatcher spreads
            // the 57 workgroups of an XCC over all 32 CUs -- 25 CUs hold two (always of different chains),
            // 7 hold one (blocks 200..255 of the grid), none is idle.
            if (k == 0)
                p.debug[REC_DEBUG_ALL_OFF + blockIdx.x * 8 + 6] =
                    ((__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 0xf) << 16) |
                    ((__builtin_amdgcn_s_getreg(4 | (0 << 6) | ((32 - 1) << 11)) >> 8) & 0xff);
#endif
        }
    };

    // variants 5 / 6 / 7: static issue priority by direction / tile / chain parity (which two chains
    // share a CU is the dispatcher's choice: with the XCD-grouped map the two directions of one tile)
    if ((p.variant == 5 && g == 0) || (p.variant == 6 && tile == 0) || (p.variant == 7 && ((g ^ tile) & 1) == 0))
        __builtin_amdgcn_s_setprio(3);
    // rowbase / xbase of a step are fetched one step ahead: three dependent L2 round trips at
    // the head of every step would otherwise sit on the recurrence's critical path
    int rb_next = uT > 0 ? p.rowbase[desc ? uT - 1 : 0] : 0;
    int xb_next = p.xbase[0], xb_cur = 0;
    for (int j = 0; j < Tchain; ++j) {
        stamp(j, 0);
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = {0.f, 0.f, 0.f, 0.f};
        const bool active = j < uT;
        const int xb_prev = xb_cur;
        const int rb = rb_next;
        xb_cur = xb_next;
        {
            const int jn = min(j + 1, Tchain - 1);
            const int tn = desc ? uT - 1 - jn : jn;
            rb_next = p.rowbase[min(max(tn, 0), p.Tmax - 1)];
            xb_next = p.xbase[jn];
        }
        const int64_t orow = active ? (int64_t)rb + p.b_off + ub : 0;
        const unsigned xrow = active ? (unsigned)xb_cur + (unsigned)ub : 0u;
        const unsigned prow = (active && j > 0) ? (unsigned)xb_prev + (unsigned)ub : 0u;
        unsigned xin = prow * 64u + (unsigned)kq * 16u;
        unsigned xout = xrow * 64u + (unsigned)kq * 16u;
        if (j > 0 && j <= tile_T && active) xin = ((unsigned)xb_prev + (unsigned)tile * 16u) * 64u + (unsigned)lane * 16u;
        if (j < tile_T) xout = ((unsigned)xb_cur + (unsigned)tile * 16u) * 64u + (unsigned)lane * 16u;
        float4 pre4 = make_float4(0.f, 0.f, 0.f, 0.f), act4 = pre4;
        if (wave == 0 && active) {
            pre4 = *reinterpret_cast<const float4*>(pre + orow * ld + row0 + 4 * kq);
            if (act) act4 = *reinterpret_cast<const float4*>(act + orow * ld + row0 + 4 * kq);
        }

        if (j > 0) {
            wait_all(flags, nwg, (unsigned)j, err, p.poll_delay);
            stamp(j, 1);
            float4 x[NCQ];
#pragma unroll
            for (int u = 0; u < NCQ; ++u) {
                x[u] = ld_x(xrsrc, xin, (unsigned)(c_beg + min(u, cnt - 1)) * chunk_stride);
                // One s_sleep (64 cycles) between two loads.  The CU's address path takes a 1 KiB wave
                // load every 16 cycles, so four waves pacing themselves at 64 cycles still hand it
                // work at its full rate -- the last request leaves no later -- but a wave that sleeps
                // between its loads leaves the SIMD's issue slots to the wave of the OTHER chain that
                // shares the SIMD and is feeding the matrix pipe at that moment.  Measured (round 3,
                // tests/gpu_ab_rec.py, five A/B pairs): 6.72-6.92 instead of 6.99-7.13 us per step;
                // 128 cycles: 7.25; busy-waiting the same 32..80 cycles with s_nop instead: 7.6-9.2 (the
                // wave keeps its issue slot); s_setprio 1 / 3 around the MFMAs: no effect.
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);   // every load ahead of the first MFMA
#pragma unroll
            for (int u = 0; u < NCQ; ++u) {
                if (u < NCQ - 1 || cnt == NCQ) {   // the last chunk exists only in the longer waves
                    const float4 a = u < NREG ? wreg[u < NREG ? u : 0] : Wl[(u - NREG) * 64 + lane];
                    SCTC_MFMA4(acc, a, x[u])
                }
            }
            if (p.debug) {
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
                stamp(j, 2);
            }
            if (wave != 0) {
                const f32x4 s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                red[(wave - 1) * 64 + lane] = make_float4(s[0], s[1], s[2], s[3]);
            }
            __syncthreads();
            stamp(j, 3);
        }

        if (wave == 0 && active) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j > 0) {
                const float4 r1 = red[lane], r2 = red[64 + lane], r3 = red[128 + lane];
                const f32x4 q = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                s = make_float4((q[0] + r1.x) + (r2.x + r3.x), (q[1] + r1.y) + (r2.y + r3.y),
                                (q[2] + r1.z) + (r2.z + r3.z), (q[3] + r1.w) + (r2.w + r3.w));
            }
            const float4 o = step_result(pre4, s, act4, act != nullptr, hi);
            *reinterpret_cast<float4*>(out + orow * ld + row0 + 4 * kq) = o;
            st_x(xrsrc, xout, (unsigned)wg * chunk_stride, o, sync_mode);
        }
        stamp(j, 4);
        if (j + 1 < Tchain) publish_step(flags + wg * REC_FLAG_STRIDE, (unsigned)(j + 1), sync_mode);
        stamp(j, 5);
    }
}

// ---------------------------------------------------------------------------------------
// More than 32 utterances, tiled over UNITS x UTTERANCES (round 6).  brnn_recurrent_kernel above gives every
// compute unit a 16 x H slab and the WHOLE minibatch: at 128 utterances every CU re-reads all 934 KB of the
// direction's state from L2 each step (213 MB over the grid, 12.7 TB/s: that, not the matrix pipes, bounded
// it at 0.54 of peak), and a step is one serial chain flags -> loads -> MFMAs -> stores -> flag.
// Here a workgroup (one per CU) owns 32 units x HALF of the utterance tiles of one direction:
//   * the same number of MFMAs per CU, half the state bytes per CU and step, no partial sums leaving the CU;
//   * grid = 4 combos (direction, utterance half) x H/32 unit blocks (228 at H = 1824, 256 at H = 2048).  Blocks
//     are dealt to XCDs round-robin (block b on XCD b % 8), combo = b & 3: a combo lives on two XCDs, whose L2s
//     fetch one combo's state (a quarter of the exchange traffic of the grid) instead of everything;
//   * the 32 x H slab (233 KB at H = 1824) is held partly in LDS, partly in registers (fragment order, as in
//     the two-chain kernel), 4 waves = 4 K quarters, every x fragment is fetched ONCE per CU and multiplies both
//     16-unit groups, every weight fragment multiplies all utterance tiles of the phase;
//   * the CU's utterance tiles form TWO independent sub-chains (A, B) whose steps ("phases") alternate: while
//     the results of A's step travel (store acknowledge, flag, fetch: three fabric trips, ~3.5 us), the CU
//     multiplies B's step.  Nothing of a phase but its MFMAs is on the matrix pipes' critical path:
//       - exchange loads run R - 1 batches ahead of the MFMAs through a ring of R register batches; the flags
//         of the NEXT phase are polled, and its first R - 1 batches issued, from inside the current phase's MFMA
//         stream (every wave polls for itself: no barrier in the stream);
//       - the epilogue of a phase (K-quarter partial sums through LDS, clip / mask, the two stores, the flag)
//         is executed in pieces BETWEEN the MFMAs of the next phase's first batch; the stores' acknowledge is
//         awaited with a counted s_waitcnt behind that batch, each finishing wave publishes its own flag word
//         (no publish barrier; a consumer lane reads the NC words of one producer).
//     A sub-chain that runs alone (the other one finished, or fewer than 3 tiles in this half) has nothing to
//     hide its latency behind: its epilogue is flushed at once and the next step starts cold.
// The K split (base / rem quarters), the four accumulators per tile and the order of every addition are
// those of the two-chain kernel brnn_recurrent_q_kernel: an utterance's rows are bit-identical to what it gets
// in a minibatch of 17..32 (tools/rec_tiled_check.py).  Tile t of the (sorted) minibatch belongs to half t & 1,
// sub-chain (t >> 1) & 1, slot t >> 2: all four chains of a direction get utterances of every length class.
// NCQ: chunks per wave; NREGF: weight fragments (chunk, unit group) per wave kept in registers; NT: utterance
// tiles per sub-chain (1: up to 64 utterances, 2: up to 128); NBAT batches per phase, ring of R (NBAT % R == 0).
template <int... I, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, I...>, F&& f)
{
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_seq(std::make_integer_sequence<int, N>(), static_cast<F&&>(f));
}

template <int NCQ, int NREGF, int NT, int NBAT, int R, int PUB, int PB>
__global__ __launch_bounds__(256, 1) void brnn_recurrent_t_kernel(RecArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    constexpr int NFR = 2 * NCQ;             // weight fragments per wave: f = 2 * chunk + unit group
    constexpr int NLDSF = NFR - NREGF;       // of them in LDS
    constexpr int XB = (NCQ + NBAT - 1) / NBAT;   // chunks per batch
    constexpr int NC = 2 * NT;               // results per phase: c = 2 * tile slot + unit group; wave c finishes c
    static_assert(NC <= 4, "one result per wave");
    static_assert(NBAT % R == 0 && NBAT >= 2 * R && R >= 2, "ring");
    static_assert((NBAT - 1) * XB < NCQ, "no empty batch");
    static_assert(PUB >= 0 && PUB + 2 <= PB && PB <= NBAT - R + 1 && PB >= 1,
                  "the flag is published (behind batch PUB) before the next phase's flags are polled (in front of batch PB - 1) and "
                  "seen (in front of batch PB), and that before the next phase's first exchange loads (batch NBAT - R + 1)");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hp = p.Hp, nch = Hp >> 4, nprod = Hp >> 5;
    const int combo = blockIdx.x & 3, ublk = blockIdx.x >> 2;
    const int g = combo & 1, half = combo >> 1;
    const int row0 = ublk * 32;
    const int uj = lane & 15, kq = lane >> 4;
    const int sync_mode = p.sync_mode;
    const int base = nch >> 2, rem = nch & 3;
    const int cnt = base + (wave < rem ? 1 : 0);
    const int c_beg = wave * base + min(wave, rem);
    float4* Wl = lds4 + (size_t)wave * NLDSF * 64;     // [NLDSF][64] wave-private, fragment order
    float4* red = lds4 + (size_t)4 * NLDSF * 64;       // [2 phase parities][NC][3 other waves][64] partial sums

    auto load_w = [&](int c, int ug) {
        const float* W = p.W[g];
        const int r = row0 + 16 * ug + uj;
        float4 v;
        if (!p.transpose) {
            v = *reinterpret_cast<const float4*>(W + (int64_t)r * p.ldw + 16 * c + 4 * kq);
        } else {
            const float* col = W + (int64_t)(16 * c + 4 * kq) * p.ldw + r;
            v.x = col[0];
            v.y = col[p.ldw];
            v.z = col[2 * p.ldw];
            v.w = col[3 * p.ldw];
        }
        return v;
    };
    // The register-resident fragments live in ACCUMULATION registers (an MFMA takes its A operand from either
    // file): together with the exchange ring they would not fit the 256 architectural registers, and what the
    // compiler then parks in the other file it copies back through a waited-for load -- the prefetch is gone.
    float4 wreg[NREGF > 0 ? NREGF : 1];
    // (a wave whose K quarter has one chunk less than NCQ multiplies that chunk too, with ZERO weights: + 0 exactly, and
    // the MFMA stream has no branch -- a branch around the last chunk's MFMAs cost 0.24 us per phase in waits at its join)
    auto load_w0 = [&](int f) {
        const float4 w = load_w(c_beg + min(f >> 1, cnt - 1), f & 1);
        return (f >> 1) < cnt ? w : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int f = 0; f < NREGF; ++f) {
        const float4 w = load_w0(f);
        asm("v_accvgpr_write_b32 %0, %1" : "=a"(wreg[f].x) : "v"(w.x));
        asm("v_accvgpr_write_b32 %0, %1" : "=a"(wreg[f].y) : "v"(w.y));
        asm("v_accvgpr_write_b32 %0, %1" : "=a"(wreg[f].z) : "v"(w.z));
        asm("v_accvgpr_write_b32 %0, %1" : "=a"(wreg[f].w) : "v"(w.w));
    }
#pragma unroll 4
    for (int f = NREGF; f < NFR; ++f) Wl[(f - NREGF) * 64 + lane] = load_w0(f);
    __syncthreads();

    const bool desc = p.descending[g] != 0;
    const float* pre = p.pre[g];
    const float* act = p.act[g];
    const float* act_or_pre = act ? act : pre;
    float* out = p.out[g];
    const int64_t ld = p.ld;
    const float hi = p.max_act > 0.f ? p.max_act : INFINITY;
    unsigned* err = p.counters + 2;
    const unsigned chunk_stride = (unsigned)p.n_xrows * 64u;
    float* xg = p.xbuf + (size_t)g * p.n_xrows * Hp;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        xg, 0, (int)((size_t)p.n_xrows * Hp * sizeof(float)), 0x00020000);

    struct Sub {
        int ub[NT], uT[NT], tile_T[NT];
        unsigned trow[NT];
        int T;                  // steps of this sub-chain (its first tile's first utterance is its longest)
        unsigned* flags;        // [nprod][REC_FLAG_STRIDE]: words 0..NC-1 of a producer = its finishing waves' step flags
        int rb_next[NT];        // rowbase of the next step's frame (fetched one step ahead)
        unsigned xb_next, xb_cur;
    };
    auto init_sub = [&](Sub& S, int sub) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int tile = (i * 2 + sub) * 2 + half;
            S.trow[i] = (unsigned)tile * 16u;
            S.ub[i] = tile * 16 + uj;
            S.uT[i] = S.ub[i] < p.B ? p.T_b[S.ub[i]] : 0;
            S.tile_T[i] = lane_order_steps(S.uT[i], tile, p.b_off, p.variant);
            S.rb_next[i] = S.uT[i] > 0 ? p.rowbase[desc ? S.uT[i] - 1 : 0] : 0;
        }
        S.T = __builtin_amdgcn_readfirstlane((int)S.trow[0] < p.B ? p.T_b[S.trow[0]] : 0);
        S.flags = p.counters + 32 + (size_t)((combo * 2 + sub) * 64) * REC_FLAG_STRIDE;
        S.xb_next = (unsigned)p.xbase[0];
        S.xb_cur = 0;
    };
    // xbase[j] is the same for every lane, which the compiler answers with a load and a v_readfirstlane waited for on
    // the spot (vmcnt(0) at the head of every phase: the exchange ring drained).  An opaque zero in the index keeps
    // the value in a vector register, first needed a phase later.
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    Sub A, Bc;
    init_sub(A, 0);
    init_sub(Bc, 1);

    // diagnostics (SCTC_REC_DEBUG=1): wall-clock (100 MHz) stamps of steps 64..71 of both sub-chains, first and
    // last unit block of combo 0: [wg][(step - 64) * 2 + sub][8]
    // (kept in LDS and written out at the end: a store inside the time loop would change the counted waits)
    const int dbg_sel = (p.debug && combo == 0 && tid == 0) ? (ublk == 0 ? 0 : (ublk == nprod - 1 ? 1 : -1)) : -1;
    unsigned* dbg_lds = reinterpret_cast<unsigned*>(red + (size_t)2 * NC * 3 * 64);      // [16][8]
    // SCTC_T_STAMPS builds only (tools/build_variant.sh recurrent.hip -DSCTC_T_STAMPS; tools/rec_tiled_timeline.py): eight
    // branches per phase in the production kernel are 0.1-0.2 us of its 8.5
#if defined(SCTC_T_STAMPS) || defined(SCTC_T_STAMP_CHUNKS)
    auto stamp = [&](int sub, int j, int k) {
        if (dbg_sel >= 0 && j >= 64 && j < 72)
            dbg_lds[((j - 64) * 2 + sub) * 8 + k] = (unsigned)wall_clock64();
    };
#else
    auto stamp = [&](int, int, int) {};
    (void)dbg_lds;
#endif
    // (a stamp behind every chunk, SCTC_T_STAMP_CHUNKS builds only: 29 more branches per phase; [(step - 64) * 2 + sub][32],
    // written out behind the [512][8] region's head; tools/rec_tiled_timeline.py prints the profile when it finds one)
#ifdef SCTC_T_STAMP_CHUNKS
    unsigned* dbg_fine = dbg_lds + 16 * 8;
    auto stamp_chunk = [&](int sub, int j, int c) {
        if (dbg_sel >= 0 && j >= 64 && j < 68 && c < 32)
            dbg_fine[((j - 64) * 2 + sub) * 32 + c] = (unsigned)wall_clock64();
    };
#else
    auto stamp_chunk = [&](int, int, int) {};
#endif

    float4 x[R][XB][NT];

    // byte offsets (within a chunk) of the rows of step j - 1 that step j of sub-chain S multiplies
    auto xin_of = [&](const Sub& S, int j, unsigned xb_prev, unsigned (&xin)[NT]) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const bool active = j < S.uT[i];
            const unsigned prow = (active && j > 0) ? xb_prev + (unsigned)S.ub[i] : 0u;
            xin[i] = prow * 64u + (unsigned)kq * 16u;
            if (j > 0 && j <= S.tile_T[i] && active)       // the tile was complete at step j - 1: lane order
                xin[i] = (xb_prev + S.trow[i]) * 64u + (unsigned)lane * 16u;
            if (p.variant == 49) xin[i] = (unsigned)lane * 16u;     // DIAGNOSTIC (wrong results): every load re-reads step 0's rows -- lines that stay cached -- to time the kernel without the arrival of fresh exchange data
        }
    };
    // batch bi of a phase into ring slot bi % R
    auto issue_part = [&](const unsigned (&xin)[NT], auto bi_c, auto lo_c, auto hi_c) {
        constexpr int bi = decltype(bi_c)::value, lo = decltype(lo_c)::value, hi_ = decltype(hi_c)::value;
#pragma unroll
        for (int u = lo; u < hi_; ++u) {
            const int cu = bi * XB + u;
            if (cu < NCQ) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    x[bi % R][u][i] = ld_x(xrsrc, xin[i], (unsigned)(c_beg + min(cu, cnt - 1)) * chunk_stride);
            }
        }
    };
    auto issue_batch = [&](const unsigned (&xin)[NT], auto bi_c) {
        issue_part(xin, bi_c, std::integral_constant<int, 0>(), std::integral_constant<int, XB>());
    };
    // A lane reads the NC flag words of producer `lane` (lanes beyond the producers re-read the last one's): ONE ordinary
    // 16-byte load, agent-coherent (sc1), unconditional -- an atomic load per word is waited for with vmcnt(0) on the
    // spot, which would drain the exchange ring twice per phase.  The chain's step is the smallest word.
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.counters, 0, (int)(REC_COUNTER_WORDS * sizeof(unsigned)), 0x00020000);
    const unsigned flane = (unsigned)min(lane, nprod - 1) * (REC_FLAG_STRIDE * 4u);
    auto flags_off = [&](const unsigned* fl) -> unsigned { return (unsigned)((fl - p.counters) * 4); };
    auto poll_issue = [&](unsigned fl_off) -> u32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(frsrc, flane, fl_off, 16 /* sc1 */);
    };
    auto poll_min = [&](const u32x4& v) -> unsigned {
        unsigned f = v[0];
#pragma unroll
        for (int c = 1; c < NC; ++c) f = min(f, v[c]);
        return f;
    };
    // every wave for itself: returns once all producers of the chain have published >= target
    auto wait_flags = [&](unsigned fl_off, unsigned f, unsigned target) {
        if (__all(f >= target)) return;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");      // the flags change under us: load again
            f = poll_min(poll_issue(fl_off));
            if (__all(f >= target)) break;
            if ((++spins & 255u) == 0) {
                if (spin_expired(err, t0, lane)) break;
            }
        }
    };
    // cold start of a sub-chain's step j: flags of step j - 1, then the first R - 1 batches of its exchange loads
    auto prologue = [&](const Sub& S, int j) {
        first_poll_delay(p.poll_delay);
        wait_flags(flags_off(S.flags), poll_min(poll_issue(flags_off(S.flags))), (unsigned)j);
        unsigned xin[NT];
        xin_of(S, j, S.xb_cur, xin);
        static_for<R - 1>([&](auto b) { issue_batch(xin, b); });
    };

    // ---- the pending epilogue of a phase
    struct Epi {
        bool pending, zero;     // uniform: something to finish; step 0 (no partial sums)
        bool lane_active;       // this lane's utterance is alive at that step (and this wave finishes a result)
        int parity;             // which half of `red` holds the partial sums
        f32x4 mine;
        float4 pre4, act4, r0, r1, r2, s, o;
        int64_t out_off;
        unsigned xo, xchunk;
        unsigned* flag;
        unsigned val;
    };
    Epi E;
    E.pending = false;
    auto epi_barrier = [&](const Epi& e) {
        if (e.pending && !e.zero) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
    auto epi_read = [&](Epi& e) {
        if (e.pending && !e.zero && wave < NC) {
            const float4* r = red + (size_t)((e.parity * NC + wave) * 3) * 64 + lane;
            e.r0 = r[0];
            e.r1 = r[64];
            e.r2 = r[128];
        }
    };
    auto epi_sum = [&](Epi& e) {
        e.s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e.pending && wave < NC && !e.zero) {
            const float4 m = make_float4(e.mine[0], e.mine[1], e.mine[2], e.mine[3]);
            // partial sums of waves 0..3 in that order, (p0 + p1) + (p2 + p3): the two-chain kernel's sum
            float4 p0 = m, p1 = e.r0, p2 = e.r1, p3 = e.r2;
            if (wave == 1) { p0 = e.r0; p1 = m; }
            if (wave == 2) { p0 = e.r0; p1 = e.r1; p2 = m; }
            if (wave == 3) { p0 = e.r0; p1 = e.r1; p2 = e.r2; p3 = m; }
            e.s = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                              (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
        }
    };
    auto epi_result = [&](Epi& e) {
        if (e.pending && wave < NC) e.o = step_result(e.pre4, e.s, e.act4, act != nullptr, hi);
    };
    auto epi_stores = [&](const Epi& e) {
        if (e.pending && wave < NC && e.lane_active) {
            *reinterpret_cast<float4*>(out + e.out_off) = e.o;
            st_x(xrsrc, e.xo, e.xchunk, e.o, sync_mode);
        }
    };
    auto epi_store = [&](Epi& e) {
        epi_sum(e);
        epi_result(e);
        epi_stores(e);
    };
    // n_after: exchange loads this wave has issued since epi_store (they stay in flight)
    auto epi_publish = [&](Epi& e, auto n_after_c) {
        constexpr int n_after = decltype(n_after_c)::value;
        if (e.pending) {
            if (wave < NC) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_after) : "memory");
                if (lane == 0) {
                    if (sync_mode == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __hip_atomic_store(e.flag, e.val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            e.pending = false;
        }
    };
    auto flush = [&](Epi& e) {
        epi_barrier(e);
        epi_read(e);
        epi_store(e);
        epi_publish(e, std::integral_constant<int, 0>());
    };

    int phase_no = 0;
    // One step of sub-chain S.  j >= 1: on entry batches 0..R-2 of its exchange loads are in flight and `E` may
    // hold the previous phase's epilogue, finished inside this phase's first batch.  If pre_next, the flags of N's
    // step jn are polled and its first R - 1 batches issued from inside this phase's MFMA stream.  j == 0: no
    // recurrent term.  On return E holds this phase's epilogue.
    auto phase = [&](Sub& S, int sub, int j, bool pre_next, const Sub& N, int jn, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;     // step 0, known at compile time: the time loop's copy has one path
        stamp(sub, j, 0);
        stamp_chunk(sub, j, 0);
        // The loads whose results are needed LATER -- the next step's xbase / rowbase, this step's additive term (its
        // epilogue runs inside the next phase) -- are issued behind the first batches, not here: anything loaded at the
        // head of the phase ends up being waited for inside batch 0 together with the exchange ring.
        auto late_loads = [&](Epi& en) {
            const int jn2 = min(j + 1, S.T - 1);
            S.xb_next = (unsigned)p.xbase[jn2 + vzero];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int tn = desc ? S.uT[i] - 1 - jn2 : jn2;
                S.rb_next[i] = p.rowbase[min(max(tn, 0), p.Tmax - 1)];
            }
            // unconditional (an idle lane's out_off is 0; without a mask matrix the additive term is read twice): a load
            // under a branch meets its zero default in a copy, and the copy waits with vmcnt(0)
            en.pre4 = *reinterpret_cast<const float4*>(pre + en.out_off);
            en.act4 = *reinterpret_cast<const float4*>(act_or_pre + en.out_off);
        };
        // The phase's addresses: exchange offsets of its own later batches, what this wave's epilogue needs (result c = wave:
        // tile slot c >> 1, unit group c & 1).  Nothing the first MFMAs take (their batch is in flight, the weights are there):
        // computed BEHIND the first group of four, under the matrix pipe, not in front of it.
        unsigned xin[NT];
        Epi En;
        auto setup = [&]() {
        const unsigned xb_prev = S.xb_cur, xb_cur = S.xb_next;
        S.xb_cur = xb_cur;
        int rb[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) rb[i] = S.rb_next[i];
        xin_of(S, j, xb_prev, xin);
        En.pending = true;
        En.zero = FIRST;
        En.parity = phase_no & 1;
        En.lane_active = false;
        En.out_off = 0;
        En.xo = 0;
        En.xchunk = (unsigned)(2 * ublk + (wave & 1)) * chunk_stride;
        En.flag = S.flags + ublk * REC_FLAG_STRIDE + wave;
        En.val = (unsigned)(j + 1);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if ((wave >> 1) == i && wave < NC) {
                const bool active = j < S.uT[i];
                const int64_t orow = active ? (int64_t)rb[i] + p.b_off + S.ub[i] : 0;
                const unsigned xrow = active ? xb_cur + (unsigned)S.ub[i] : 0u;
                En.xo = xrow * 64u + (unsigned)kq * 16u;
                if (j < S.tile_T[i]) En.xo = (xb_cur + S.trow[i]) * 64u + (unsigned)lane * 16u;
                En.lane_active = active;
                En.out_off = orow * ld + row0 + 16 * (wave & 1) + 4 * kq;
            }
        }
        };
        if constexpr (FIRST) { setup(); late_loads(En); }
        f32x4 acc[2][NT][4];
#pragma unroll
        for (int ug = 0; ug < 2; ++ug)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[ug][i][q] = {0.f, 0.f, 0.f, 0.f};

        // The LDS-resident weight fragments are read WA groups ahead of the MFMAs that take them (a rotating set of
        // WA + 1 registers quadruples): read in front of its use, a fragment costs its group the LDS round trip -- measured
        // 0.28-0.32 us for a chunk with LDS weights against 0.24 with register weights, 1 us per phase, with every exchange
        // load served from cache (SCTC_REC_VARIANT=49: same times -- it was never the exchange data).
        constexpr int WA = 2;
        float4 aq[WA + 1];
        auto prefetch_a = [&](auto G_c) {      // G: group index within the phase = chunk * NC + result
            constexpr int G = decltype(G_c)::value, cu = G / NC, f = 2 * cu + ((G % NC) & 1);
            if constexpr (cu < NCQ && f >= NREGF) aq[G % (WA + 1)] = Wl[(f - NREGF) * 64 + lane];
        };
        // four MFMAs: chunk u of batch bi, result g = 2 * tile slot + unit group
        auto mfma_group = [&](auto bi_c, auto u_c, auto g_c) {
            constexpr int bi = decltype(bi_c)::value, u = decltype(u_c)::value, cu = bi * XB + u;
            constexpr int gi = decltype(g_c)::value >> 1, ug = decltype(g_c)::value & 1;
            constexpr int G = cu * NC + decltype(g_c)::value;
            prefetch_a(std::integral_constant<int, G + WA>());
            if constexpr (cu < NCQ) {
                constexpr int f = 2 * cu + ug;
                float4 a;
                if constexpr (f < NREGF) a = wreg[f]; else a = aq[G % (WA + 1)];
                SCTC_MFMA4(acc[ug][gi], a, x[bi % R][u][gi])
            }
        };
        auto mfma_chunk = [&](auto bi_c, auto u_c) {
            static_for<NC>([&](auto g_c) { mfma_group(bi_c, u_c, g_c); });
        };
        if constexpr (!FIRST) {
            // the next phase's poll and first batches are issued unconditionally (without a next phase: this chain's
            // own flags, exchange row 0, results unused) -- a load under a branch makes every later wait a vmcnt(0)
            const unsigned nfl_off = flags_off(N.flags);
            u32x4 fl = {0u, 0u, 0u, 0u};
            unsigned xin_n[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) xin_n[i] = (unsigned)kq * 16u;
            // ONE exchange load behind a group of four MFMAs (128 cycles of the matrix pipe), never a batch at once: the CU's
            // one vector-memory path takes a KB (a wave's 16-byte-per-lane load) every 16 cycles, so ten loads from each of
            // the four waves keep it busy for 640 cycles, and a wave whose load does not issue does not issue MFMAs either
            // (measured: +0.3 us per batch issued in one go).
            auto issue_one = [&](const unsigned (&xs)[NT], auto bi_c, auto l_c) {
                constexpr int bi = decltype(bi_c)::value, l = decltype(l_c)::value, u = l / NT, i = l % NT, cu = bi * XB + u;
                if constexpr (cu < NCQ)
                    x[bi % R][u][i] = ld_x(xrsrc, xs[i], (unsigned)(c_beg + min(cu, cnt - 1)) * chunk_stride);
            };
            // K quarters: wave c keeps its own partial sum of result c, the others park theirs in LDS.  Result c is final
            // behind group c of the LAST chunk: parked behind the next group's MFMAs, under the matrix pipe -- only the last
            // result's sum is taken with the pipe idle.
            auto park = [&](auto c_c) {
                constexpr int c = decltype(c_c)::value;
                float4* rp = red + (size_t)(En.parity * NC * 3) * 64 + lane;
                const f32x4 s = (acc[c & 1][c >> 1][0] + acc[c & 1][c >> 1][1]) + (acc[c & 1][c >> 1][2] + acc[c & 1][c >> 1][3]);
                if (wave == c) En.mine = s;
                else rp[(c * 3 + (wave < c ? wave : wave - 1)) * 64] = make_float4(s[0], s[1], s[2], s[3]);
            };
            static_for<WA>([&](auto G_c) { prefetch_a(G_c); });
            static_for<NBAT>([&](auto b_c) {
                constexpr int b = decltype(b_c)::value;
                constexpr int nb = b + R - 1;       // the batch whose loads are issued between batch b's MFMAs
                constexpr int XBb = (b + 1) * XB <= NCQ ? XB : NCQ - b * XB, NG = XBb * NC;      // chunks, groups of this batch
                constexpr int NL = XB * NT;                                                    // load slots of a batch
                // batch 0: the previous phase's epilogue in pieces behind groups NC-1 .., then the loads, one per group;
                // other batches: a load behind every second group
#define SCTC_T_SLOT(piece) (NC - 1 + (piece) < NG - 1 ? NC - 1 + (piece) : NG - 1)
                if constexpr (b == PB - 1) {
                    fl = poll_issue(nfl_off);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (b == PB) {
                    if (pre_next) {
                        wait_flags(nfl_off, poll_min(fl), (unsigned)jn);
                        xin_of(N, jn, N.xb_cur, xin_n);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                static_for<NG>([&](auto s_c) {
                    constexpr int s = decltype(s_c)::value;
                    mfma_group(b_c, std::integral_constant<int, s / NC>(), std::integral_constant<int, s % NC>());
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (b == 0 && s == 0) { setup(); __builtin_amdgcn_sched_barrier(0); }
                    if constexpr (b * XB + s / NC == NCQ - 1 && s % NC >= 1) {
                        park(std::integral_constant<int, s % NC - 1>());
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (s % NC == NC - 1) stamp_chunk(sub, j, 1 + b * XB + s / NC);
                    if constexpr (b == 0) {
                        if constexpr (s % NC == NC - 1 && s / NC < 5) stamp(sub, j, 1 + s / NC);
                        if constexpr (s == SCTC_T_SLOT(0)) { epi_barrier(E); __builtin_amdgcn_sched_barrier(0); }
                        if constexpr (s == SCTC_T_SLOT(1)) { epi_read(E); __builtin_amdgcn_sched_barrier(0); }
                        if constexpr (s == SCTC_T_SLOT(2)) { epi_sum(E); __builtin_amdgcn_sched_barrier(0); }
                        if constexpr (s == SCTC_T_SLOT(3)) { epi_result(E); __builtin_amdgcn_sched_barrier(0); }
                        if constexpr (s == SCTC_T_SLOT(4)) { epi_stores(E); __builtin_amdgcn_sched_barrier(0); }
                    }
                    static_for<NL>([&](auto l_c) {
                        constexpr int l = decltype(l_c)::value, raw = b == 0 ? SCTC_T_SLOT(5) + l : l * NG / NL;
                        if constexpr ((raw < NG - 1 ? raw : NG - 1) == s) {
                            if constexpr (nb < NBAT) issue_one(xin, std::integral_constant<int, nb>(), l_c);
                            else issue_one(xin_n, std::integral_constant<int, nb - NBAT>(), l_c);
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (b == NBAT - 1) {
                    // The late loads get a use here: left alone the compiler sinks them to the loop latch, next to the
                    // copies that swap the sub-chains, and waits for them there with vmcnt(0) -- the ring drained.
                    asm volatile("" : "+v"(S.xb_next));
#pragma unroll
                    for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(S.rb_next[i]));
                    asm volatile("" : "+v"(En.pre4.x), "+v"(En.pre4.y), "+v"(En.pre4.z), "+v"(En.pre4.w));
                    asm volatile("" : "+v"(En.act4.x), "+v"(En.act4.y), "+v"(En.act4.z), "+v"(En.act4.w));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (b == PUB) {
                    // exchange loads issued since the epilogue's stores: batches R-1 .. R-1+PUB of this phase
                    constexpr int first = (R - 1) * XB, end = (R + PUB) * XB < NCQ ? (R + PUB) * XB : NCQ;
                    epi_publish(E, std::integral_constant<int, NT * (end - first)>());
                    __builtin_amdgcn_sched_barrier(0);
                    late_loads(En);
                    __builtin_amdgcn_sched_barrier(0);
                    stamp(sub, j, 6);
                }
            });
            stamp(sub, j, 7);
            park(std::integral_constant<int, NC - 1>());       // (the other results were parked between the last chunk's groups)
        } else {
            En.mine = {0.f, 0.f, 0.f, 0.f};
        }
        E = En;
        ++phase_no;
    };

    // step 0 of both sub-chains (no recurrent term), then A and B alternate while B lasts (T_A >= T_B: the
    // minibatch is sorted), then A alone with its exchange latency exposed.  ONE copy of the phase body: `cur` and
    // `nxt` swap places after a phase that has started the other chain's loads (two inlined copies get two register
    // assignments for the exchange ring, and the copies between them wait for loads that are still in flight).
    if (A.T > 0) { phase(A, 0, 0, false, A, 0, std::true_type()); flush(E); }
    if (Bc.T > 0) { phase(Bc, 1, 0, false, Bc, 0, std::true_type()); flush(E); }
    Sub cur = A, nxt = Bc;
    int jc = 1, jn = 1, subc = 0;
    if (jc < cur.T) prologue(cur, jc);
    while (jc < cur.T) {
        const bool pre_next = jn < nxt.T;
        phase(cur, subc, jc, pre_next, nxt, jn, std::false_type());
        ++jc;
        if (pre_next) {
            const Sub t = cur;
            cur = nxt;
            nxt = t;
            const int tj = jc;
            jc = jn;
            jn = tj;
            subc ^= 1;
        } else {
            flush(E);
            if (jc < cur.T) prologue(cur, jc);
        }
    }
#if defined(SCTC_T_STAMPS) || defined(SCTC_T_STAMP_CHUNKS)
    if (dbg_sel >= 0) {
        for (int i = 0; i < 16 * 8; ++i) p.debug[dbg_sel * 16 * 8 + i] = dbg_lds[i];
#ifdef SCTC_T_STAMP_CHUNKS
        for (int i = 0; i < 8 * 32; ++i) p.debug[REC_DEBUG_ALL_OFF + dbg_sel * 8 * 32 + i] = dbg_fine[i];
#endif
    }
#endif
}

#ifdef SCTC_REC_EXPERIMENTS    // measured slower / superseded kernels, kept for the A/Bs of profiles/r04_recurrence_q8.md and r05 (build with -DSCTC_REC_EXPERIMENTS)
#include "recurrent_experiments_q8.inc"
#endif  // SCTC_REC_EXPERIMENTS

// ---------------------------------------------------------------------------------------
// Small-batch variant (1..4 utterances: the reference's minibatch-1 mode).  With one utterance
// a step is a 16 x H matrix-vector product per workgroup (0.1 us of VALU work) and the whole
// step is hand-off latency, so the exchange protocol changes:
//   * no flags, no store drain: the exchange buffer ([step*4 + utterance][H], row-major) is
//     filled with a sentinel bit pattern (0xFFFFFFFF, a NaN no arithmetic here produces) before
//     the launch, every address is written once, and consumers re-read a row with L2-bypassing
//     (sc1) loads until no dword is the sentinel: ONE fabric hop per step instead of three
//     (data ack -> flag -> data fetch).  The bypass traffic is 7 KiB per workgroup, utterance and
//     poll -- affordable only at this batch size, which is why the larger kernels use flags;
//   * weights live in registers (2 rows x H/32 values per thread), the previous state is staged
//     once per step into LDS by the polling waves (one wave per utterance) and read as
//     broadcasts; fp32 FMAs + a DPP reduction over the 32 K slices replace the MFMA tile that
//     would be 15/16 padding.
// NK = H/32 values per thread and row.
static constexpr unsigned XSENT = 0xffffffffu;

template <int NK, int SB>
__global__ __launch_bounds__(256, 1) void brnn_recurrent_s_kernel(RecArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    float* xs = reinterpret_cast<float*>(lds4);   // [2 parities][SB][Hp]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x & 1, wg = blockIdx.x >> 1;
    const int Hp = p.Hp;
    const int rg = tid >> 5, ks = tid & 31;          // row pair, K slice
    const int row0 = wg * 16 + 2 * rg;               // this thread's two output units
    const bool owner = ks == 31;                     // holds the reduced sums after the DPP tree

    // ---- stationary weights in registers: w[r][i] = Wop[row0 + r][32 i + ks]
    float w0[NK], w1[NK];
    {
        const float* W = p.W[g];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int k = 32 * i + ks;
            if (!p.transpose) {
                w0[i] = W[(int64_t)row0 * p.ldw + k];
                w1[i] = W[(int64_t)(row0 + 1) * p.ldw + k];
            } else {
                w0[i] = W[(int64_t)k * p.ldw + row0];
                w1[i] = W[(int64_t)k * p.ldw + row0 + 1];
            }
        }
    }
    const bool desc = p.descending[g] != 0;
    const float* pre = p.pre[g];
    const float* act = p.act[g];
    float* out = p.out[g];
    const int64_t ld = p.ld;
    const float hi = p.max_act > 0.f ? p.max_act : INFINITY;
    unsigned* err = p.counters + 2;
    float* xg = p.xbuf + (size_t)g * p.n_xrows * Hp;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        xg, 0, (int)((size_t)p.n_xrows * Hp * sizeof(float)), 0x00020000);
    int Tb[SB];
#pragma unroll
    for (int b = 0; b < SB; ++b) Tb[b] = b < p.B ? p.T_b[b] : 0;   // sorted, longest first
    constexpr int NQ = (NK * 32 / 4 + 63) / 64;      // float4 loads per lane for one state row
    const int n4 = Hp >> 2;

    const int dbg_sel = (p.debug && g == 0 && tid == 0) ? (wg == 0 ? 0 : (wg == (Hp >> 4) - 1 ? 1 : -1)) : -1;
    auto stamp = [&](int j, int k) {
        if (dbg_sel >= 0 && j >= 64 && j < 80)
            p.debug[(dbg_sel * 16 + (j - 64)) * 8 + k] = (unsigned)clock64();
    };

    // exchange row of utterance b at step j = xbase[j] + b (fetched one step ahead)
    int xb_next = p.xbase[0], xb_cur = 0;
    for (int j = 0; j < p.Tmax; ++j) {
        stamp(j, 0);
        const int xb_prev = xb_cur;
        xb_cur = xb_next;
        xb_next = p.xbase[min(j + 1, p.Tmax - 1)];
        int nb = 0;   // active utterances at this step (a prefix, utterances are sorted by length)
#pragma unroll
        for (int b = 0; b < SB; ++b) nb += j < Tb[b] ? 1 : 0;
        if (nb == 0) break;
        // per-frame additive term and mask source of the owners (independent of the recurrence)
        float2 pre2[SB], act2[SB];
        int64_t orow[SB];
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            pre2[b] = make_float2(0.f, 0.f);
            act2[b] = make_float2(0.f, 0.f);
            orow[b] = 0;
            if (b < nb) {
                const int t = desc ? Tb[b] - 1 - j : j;
                orow[b] = (int64_t)p.rowbase[t] + p.b_off + b;
                if (owner) {
                    pre2[b] = *reinterpret_cast<const float2*>(pre + orow[b] * ld + row0);
                    if (act) act2[b] = *reinterpret_cast<const float2*>(act + orow[b] * ld + row0);
                }
            }
        }
        float s0[SB], s1[SB];
#pragma unroll
        for (int b = 0; b < SB; ++b) s0[b] = s1[b] = 0.f;
        if (j > 0) {
            float* xcur = xs + (size_t)(j & 1) * SB * Hp;
            // ---- wave b fetches utterance b's previous state row: re-read until complete
            first_poll_delay(p.poll_delay);
            for (int bb = wave; bb < nb; bb += 4) {   // 4 waves, up to SB rows
                const unsigned rowoff = (unsigned)(xb_prev + bb) * (unsigned)Hp * 4u;
                u32x4 v[NQ];
                const unsigned long long t0 = wall_clock64();
                unsigned spins = 0;
                for (;;) {
#pragma unroll
                    for (int c = 0; c < NQ; ++c) {
                        const int item = min(c * 64 + lane, n4 - 1);
                        v[c] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (unsigned)item * 16u, rowoff, 16 /* sc1 */);
                    }
                    bool ok = true;
#pragma unroll
                    for (int c = 0; c < NQ; ++c)
                        ok = ok && v[c][0] != XSENT && v[c][1] != XSENT && v[c][2] != XSENT && v[c][3] != XSENT;
                    if (__all(ok)) break;
                    if ((++spins & 255u) == 0) {
                        if (spin_expired(err, t0, lane)) break;
                    }
                }
#pragma unroll
                for (int c = 0; c < NQ; ++c) {
                    const int item = c * 64 + lane;
                    if (item < n4)
                        *reinterpret_cast<u32x4*>(xcur + (size_t)bb * Hp + 4 * item) = v[c];
                }
            }
            __syncthreads();
            stamp(j, 1);
            // ---- 2 x (H/32) FMAs per utterance; lanes of a K-slice group read consecutive floats,
            // the two row pairs of a wave read the same addresses (LDS broadcast)
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                if (b < nb) {
                    const float* xb = xcur + (size_t)b * Hp + ks;
                    float a0 = 0.f, a1 = 0.f, c0 = 0.f, c1 = 0.f;   // two chains per row: shorter FMA dependency
#pragma unroll
                    for (int i = 0; i + 1 < NK; i += 2) {
                        const float x0 = xb[32 * i], x1 = xb[32 * (i + 1)];
                        a0 = fmaf(w0[i], x0, a0);
                        a1 = fmaf(w1[i], x0, a1);
                        c0 = fmaf(w0[i + 1], x1, c0);
                        c1 = fmaf(w1[i + 1], x1, c1);
                    }
                    if (NK & 1) {
                        const float x0 = xb[32 * (NK - 1)];
                        a0 = fmaf(w0[NK - 1], x0, a0);
                        a1 = fmaf(w1[NK - 1], x0, a1);
                    }
                    float r0 = a0 + c0, r1 = a1 + c1;
                    // sum over the 32 K slices = two DPP rows: lane 31 / 63 end up with the total
                    r0 = dpp_add<DPP_ROW_SHR1, 0xF>(r0); r1 = dpp_add<DPP_ROW_SHR1, 0xF>(r1);
                    r0 = dpp_add<DPP_ROW_SHR2, 0xF>(r0); r1 = dpp_add<DPP_ROW_SHR2, 0xF>(r1);
                    r0 = dpp_add<DPP_ROW_SHR4, 0xF>(r0); r1 = dpp_add<DPP_ROW_SHR4, 0xF>(r1);
                    r0 = dpp_add<DPP_ROW_SHR8, 0xF>(r0); r1 = dpp_add<DPP_ROW_SHR8, 0xF>(r1);
                    r0 = dpp_add<DPP_ROW_BCAST15, 0xA>(r0); r1 = dpp_add<DPP_ROW_BCAST15, 0xA>(r1);
                    s0[b] = r0;
                    s1[b] = r1;
                }
            }
            stamp(j, 2);
        }
        if (owner) {
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                if (b < nb) {
                    float2 o;
                    if (!act) {
                        o.x = fminf(fmaxf(pre2[b].x + s0[b], 0.f), hi);
                        o.y = fminf(fmaxf(pre2[b].y + s1[b], 0.f), hi);
                    } else {
                        o.x = (act2[b].x > 0.f && act2[b].x < hi) ? pre2[b].x + s0[b] : 0.f;
                        o.y = (act2[b].y > 0.f && act2[b].y < hi) ? pre2[b].y + s1[b] : 0.f;
                    }
                    *reinterpret_cast<float2*>(out + orow[b] * ld + row0) = o;
                    // publish: write-through, self-validating (no flag, no drain)
                    __builtin_amdgcn_raw_buffer_store_b64(
                        (u32x2){__float_as_uint(o.x), __float_as_uint(o.y)}, xrsrc, (unsigned)row0 * 4u,
                        (unsigned)(xb_cur + b) * (unsigned)Hp * 4u, 16 /* sc1 */);
                }
            }
        }
        stamp(j, 3);
    }
}

#ifdef SCTC_REC_EXPERIMENTS    // superseded at 6..16 utterances by the single-chain flag kernel (round 5); SCTC_REC_VARIANT=42
#include "recurrent_experiments_m.inc"
#endif  // SCTC_REC_EXPERIMENTS

// ---------------------------------------------------------------------------------------
// 16-bit-operand variant of the mid-batch kernel for the "fp16 activations" configuration
// (BASELINE configs[4]; 6..16 utterances, e.g. cfg-5 with minibatch 8): the recurrent weights
// and the exchanged state are rounded to 16 bit (float16 in the forward pass, bfloat16 in BPTT:
// deltas need fp32's exponent range), products are exact and accumulate in fp32 on
// v_mfma_f32_16x16x32_{f16,bf16}; the per-frame term `pre`, the clip / mask and the stored
// results stay fp32.  Same one-hop sentinel exchange as above, but a state row is Hp x 2 bytes:
// half the fabric traffic per poll (the limit of the fp32 kernel at 16 utterances) and half the
// LDS staging; a chunk is 32 K values, the A fragments (8 halves per lane) live in registers.
// NCH = ceil(H/32/4) chunks per wave.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

template <bool BF> struct Rec16;
template <> struct Rec16<false> {
    using V8 = h16x8;
    static __device__ __forceinline__ unsigned short bits(float v)
    {
        const _Float16 h = (_Float16)v;
        return __builtin_bit_cast(unsigned short, h);
    }
    static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Rec16<true> {
    using V8 = b16x8;
    static __device__ __forceinline__ unsigned short bits(float v)
    {
        const __bf16 h = (__bf16)v;
        return __builtin_bit_cast(unsigned short, h);
    }
    static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <int NCH, bool BF>
__global__ __launch_bounds__(256, 1) void brnn_recurrent_mh_kernel(RecArgs p)
{
    using R = Rec16<BF>;
    using V8 = typename R::V8;
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x & 1, wg = blockIdx.x >> 1;
    const int Hp = p.Hp, nch = Hp >> 5;                      // 32-wide K chunks
    const int row0 = wg * 16;
    const int uj = lane & 15, kq = lane >> 4;
    const int base = nch >> 2, rem = nch & 3;
    const int cnt = base + (wave < rem ? 1 : 0);
    const int c_beg = wave * base + min(wave, rem);
    const int xld = Hp + 8;                                  // LDS row stride (halves): conflict-free b128 reads
    unsigned short* xs = reinterpret_cast<unsigned short*>(lds4);          // [16][xld]
    float4* red = lds4 + ((size_t)16 * xld * 2 + 15) / 16;                 // [3][64]

    // ---- stationary weights as A fragments: wf[u] = { Wop[row0 + uj][32c + 8kq + e] }_e, 16 bit
    V8 wf[NCH];
    {
        const float* W = p.W[g];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int c = c_beg + min(u, cnt - 1);
            unsigned short hb[8];
            if (!p.transpose) {
                const float4 v0 = *reinterpret_cast<const float4*>(W + (int64_t)(row0 + uj) * p.ldw + 32 * c + 8 * kq);
                const float4 v1 = *reinterpret_cast<const float4*>(W + (int64_t)(row0 + uj) * p.ldw + 32 * c + 8 * kq + 4);
                hb[0] = R::bits(v0.x); hb[1] = R::bits(v0.y); hb[2] = R::bits(v0.z); hb[3] = R::bits(v0.w);
                hb[4] = R::bits(v1.x); hb[5] = R::bits(v1.y); hb[6] = R::bits(v1.z); hb[7] = R::bits(v1.w);
            } else {
                const float* col = W + (int64_t)(32 * c + 8 * kq) * p.ldw + row0 + uj;
#pragma unroll
                for (int e = 0; e < 8; ++e) hb[e] = R::bits(col[(int64_t)e * p.ldw]);
            }
            u32x4 packed = {(unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[2] | ((unsigned)hb[3] << 16),
                            (unsigned)hb[4] | ((unsigned)hb[5] << 16), (unsigned)hb[6] | ((unsigned)hb[7] << 16)};
            wf[u] = __builtin_bit_cast(V8, packed);
        }
    }
    const bool desc = p.descending[g] != 0;
    const float* pre = p.pre[g];
    const float* act = p.act[g];
    float* out = p.out[g];
    const int64_t ld = p.ld;
    const float hi = p.max_act > 0.f ? p.max_act : INFINITY;
    unsigned* err = p.counters + 2;
    // exchange rows of 16-bit values: [direction][xrow][Hp] halves inside the fp32-sized buffer
    unsigned short* xg = reinterpret_cast<unsigned short*>(p.xbuf) + (size_t)g * p.n_xrows * Hp;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        xg, 0, (int)((size_t)p.n_xrows * Hp * 2), 0x00020000);
    const int uT = uj < p.B ? p.T_b[uj] : 0;                 // lane's utterance (sorted, longest first)
    const int n16 = Hp >> 3;                                 // 16-byte pieces per state row
    constexpr int NQ = (NCH * 4 * 4 + 63) / 64;              // pieces per lane: ceil(Hp/8/64)

    int xb_next = p.xbase[0], xb_cur = 0;
    for (int j = 0; j < p.Tmax; ++j) {
        const int xb_prev = xb_cur;
        xb_cur = xb_next;
        xb_next = p.xbase[min(j + 1, p.Tmax - 1)];
        const int nb = __builtin_amdgcn_readfirstlane(__popcll(__ballot(j < uT && kq == 0)));   // active prefix
        const bool active = j < uT;
        const int t = desc ? uT - 1 - j : j;
        const int64_t orow = active ? (int64_t)p.rowbase[t] + p.b_off + uj : 0;
        float4 pre4 = make_float4(0.f, 0.f, 0.f, 0.f), act4 = pre4;
        if (wave == 0 && active) {
            pre4 = *reinterpret_cast<const float4*>(pre + orow * ld + row0 + 4 * kq);
            if (act) act4 = *reinterpret_cast<const float4*>(act + orow * ld + row0 + 4 * kq);
        }
        f32x4 acc[2];
        acc[0] = {0.f, 0.f, 0.f, 0.f};
        acc[1] = {0.f, 0.f, 0.f, 0.f};
        if (j > 0) {
            // ---- stage the previous state: rows round-robin over the waves (wave w: rows w, w+4,
            // w+8, w+12), all rows of a wave in flight at once, incomplete ones re-read until
            // complete (half-size rows: -3 % per step at 8 utterances; the fp32 kernel keeps one row
            // in flight, where two were measured slower)
            first_poll_delay(p.poll_delay);
            {
                constexpr int MR = 4;
                u32x4 v[MR][NQ];
                bool need[MR];
#pragma unroll
                for (int r = 0; r < MR; ++r) need[r] = wave + 4 * r < nb;     // wave-uniform
                const unsigned long long t0 = wall_clock64();
                unsigned spins = 0;
                for (;;) {
#pragma unroll
                    for (int r = 0; r < MR; ++r) {
                        if (need[r]) {
                            const unsigned rowoff = (unsigned)(xb_prev + wave + 4 * r) * (unsigned)Hp * 2u;
#pragma unroll
                            for (int c = 0; c < NQ; ++c) {
                                const int item = min(c * 64 + lane, n16 - 1);
                                v[r][c] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (unsigned)item * 16u, rowoff, 16 /* sc1 */);
                            }
                        }
                    }
                    bool pending = false;
#pragma unroll
                    for (int r = 0; r < MR; ++r) {
                        if (need[r]) {
                            bool ok = true;
#pragma unroll
                            for (int c = 0; c < NQ; ++c)
                                ok = ok && v[r][c][0] != XSENT && v[r][c][1] != XSENT && v[r][c][2] != XSENT && v[r][c][3] != XSENT;
                            if (__all(ok)) {
#pragma unroll
                                for (int c = 0; c < NQ; ++c) {
                                    const int item = c * 64 + lane;
                                    if (item < n16)
                                        *reinterpret_cast<u32x4*>(xs + (size_t)(wave + 4 * r) * xld + 8 * item) = v[r][c];
                                }
                                need[r] = false;
                            } else {
                                pending = true;
                            }
                        }
                    }
                    if (!pending) break;
                    if ((++spins & 255u) == 0) {
                        if (spin_expired(err, t0, lane)) break;
                    }
                }
            }
            __syncthreads();
            // ---- products: B fragment of chunk c = { x[uj][32c + 8kq + e] }_e from LDS
            const unsigned short* xrow = xs + (size_t)uj * xld + 8 * kq;
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                if (u < NCH - 1 || cnt == NCH) {
                    const V8 x = *reinterpret_cast<const V8*>(xrow + 32 * (c_beg + u));
                    acc[u & 1] = R::mfma(wf[u], x, acc[u & 1]);
                }
            }
            if (wave != 0) {
                const f32x4 sres = acc[0] + acc[1];
                red[(wave - 1) * 64 + lane] = make_float4(sres[0], sres[1], sres[2], sres[3]);
            }
            __syncthreads();
        }
        if (wave == 0 && active) {
            float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j > 0) {
                const float4 r1 = red[lane], r2 = red[64 + lane], r3 = red[128 + lane];
                const f32x4 q = acc[0] + acc[1];
                sv = make_float4((q[0] + r1.x) + (r2.x + r3.x), (q[1] + r1.y) + (r2.y + r3.y),
                                 (q[2] + r1.z) + (r2.z + r3.z), (q[3] + r1.w) + (r2.w + r3.w));
            }
            const float4 o = step_result(pre4, sv, act4, act != nullptr, hi);
            *reinterpret_cast<float4*>(out + orow * ld + row0 + 4 * kq) = o;
            // publish the 16-bit state: write-through, self-validating (no flag, no drain)
            const u32x2 ou = {(unsigned)R::bits(o.x) | ((unsigned)R::bits(o.y) << 16),
                              (unsigned)R::bits(o.z) | ((unsigned)R::bits(o.w) << 16)};
            __builtin_amdgcn_raw_buffer_store_b64(ou, xrsrc, (unsigned)(row0 + 4 * kq) * 2u,
                                                  (unsigned)(xb_cur + uj) * (unsigned)Hp * 2u, 16 /* sc1 */);
        }
        // the next step's staging overwrites the state rows in LDS: every wave must be done reading
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// Non-persistent fallback: ONE launch per time step, both directions -- the structure of the
// reference itself (brnnet.py:148-152: mvdot_col_slice + minmax per step; :215-224 BPTT), used
// whenever a whole-device persistent grid cannot be guaranteed co-resident: the device is shared
// with another process and the lease cannot be had, the grid is larger than the device (layers
// beyond 2048 units, CU masking), or a persistent launch has timed out and the step is retried.
// Workgroup (chunk, direction, utterance tile) computes 16 units x 16 utterances of step j:
// 4 waves = 4 K quarters on v_mfma_f32_16x16x4_f32, weights streamed from L2 / the Infinity
// Cache (13.3 MB per direction at H = 1824), the previous state read straight from the rows of
// the output matrix that the launch of step j-1 wrote (stream order is the step barrier: no
// exchange buffer, no flags, no spinning).  ~4x slower than the persistent kernels, always safe.
__global__ __launch_bounds__(256) void brnn_recurrent_step_kernel(RecArgs p, int j)
{
    __shared__ float4 red[3 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x, g = blockIdx.y, tile = blockIdx.z;
    // utterances are sorted by length: a tile is finished once its first utterance is
    const int Ttile = tile * 16 < p.B ? p.T_b[tile * 16] : 0;
    if (j >= Ttile) return;
    const int nch = p.Hp >> 4;
    const int row0 = wg * 16;
    const int uj = lane & 15, kq = lane >> 4;
    const int base = nch >> 2, rem = nch & 3;
    const int cnt = base + (wave < rem ? 1 : 0);
    const int c_beg = wave * base + min(wave, rem);
    const bool desc = p.descending[g] != 0;
    const float* W = p.W[g];
    const float* pre = p.pre[g];
    const float* act = p.act[g];
    float* out = p.out[g];
    const int64_t ld = p.ld;
    const float hi = p.max_act > 0.f ? p.max_act : INFINITY;
    const int ub = tile * 16 + uj;
    const int uT = ub < p.B ? p.T_b[ub] : 0;
    const bool active = j < uT;
    const int t = desc ? uT - 1 - j : j;
    const int tp = desc ? t + 1 : t - 1;
    const int64_t orow = active ? (int64_t)p.rowbase[t] + p.b_off + ub : 0;
    // finished / empty slots read row 0 (valid memory, result discarded)
    const int64_t prow = (active && j > 0) ? (int64_t)p.rowbase[tp] + p.b_off + ub : 0;
    float4 pre4 = make_float4(0.f, 0.f, 0.f, 0.f), act4 = pre4;
    if (wave == 0 && active) {
        pre4 = *reinterpret_cast<const float4*>(pre + orow * ld + row0 + 4 * kq);
        if (act) act4 = *reinterpret_cast<const float4*>(act + orow * ld + row0 + 4 * kq);
    }
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = {0.f, 0.f, 0.f, 0.f};
    if (j > 0) {
        const float* xr = out + prow * ld + 4 * kq;
        for (int u = 0; u < cnt; ++u) {
            const int c = c_beg + u;
            float4 a;
            if (!p.transpose) {
                a = *reinterpret_cast<const float4*>(W + (int64_t)(row0 + uj) * p.ldw + 16 * c + 4 * kq);
            } else {
                const float* col = W + (int64_t)(16 * c + 4 * kq) * p.ldw + row0 + uj;
                a.x = col[0];
                a.y = col[p.ldw];
                a.z = col[2 * p.ldw];
                a.w = col[3 * p.ldw];
            }
            const float4 x = *reinterpret_cast<const float4*>(xr + 16 * c);
            SCTC_MFMA4(acc, a, x)
        }
        if (wave != 0) {
            const f32x4 s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            red[(wave - 1) * 64 + lane] = make_float4(s[0], s[1], s[2], s[3]);
        }
        __syncthreads();
    }
    if (wave == 0 && active) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j > 0) {
            const float4 r1 = red[lane], r2 = red[64 + lane], r3 = red[128 + lane];
            const f32x4 q = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            s = make_float4((q[0] + r1.x) + (r2.x + r3.x), (q[1] + r1.y) + (r2.y + r3.y),
                            (q[2] + r1.z) + (r2.z + r3.z), (q[3] + r1.w) + (r2.w + r3.w));
        }
        const float4 o = step_result(pre4, s, act4, act != nullptr, hi);
        *reinterpret_cast<float4*>(out + orow * ld + row0 + 4 * kq) = o;
    }
}

size_t recurrent_xbuf_floats(int Hp, int64_t max_xrows)
{
    return (size_t)2 * (size_t)max_xrows * Hp;  // one chunk-major state copy per direction
}

int recurrent_supported(int Hp, int B, char* why, int why_len)
{
    if (Hp % 32 != 0) { snprintf(why, why_len, "layer size %d not padded to 32", Hp); return 0; }
    (void)B;   // any minibatch size: more than 128 utterances run as several launches
    return 1;  // any layer size: grids beyond the device run on the per-step fallback
}

template <int NTW>
static RecKernel pick_kernel(int nch_half, bool pipe)
{
    // two utterance tiles per wave (33..64 utterances) keep the round 1-4 order: there the loads of the second
    // (last) batch are the only ones not under MFMAs, and the pipelined order measured SLOWER (13.1 -> 14.7 us per step
    // at 64 utterances; 24.6 -> 21.6 at 128: profiles/r05_recurrence_large.md)
    if constexpr (NTW >= 4) {
        if (pipe) {
            switch (nch_half) {
                case 16: return brnn_recurrent_kernel<NTW, 16, true>;   // H = 512
                case 32: return brnn_recurrent_kernel<NTW, 32, true>;   // H = 1024
                case 57: return brnn_recurrent_kernel<NTW, 57, true>;   // H = 1824
                case 64: return brnn_recurrent_kernel<NTW, 64, true>;   // H = 2048
                default: break;
            }
        }
    }
    switch (nch_half) {
        case 16: return brnn_recurrent_kernel<NTW, 16>;   // H = 512
        case 32: return brnn_recurrent_kernel<NTW, 32>;   // H = 1024
        case 57: return brnn_recurrent_kernel<NTW, 57>;   // H = 1824
        case 64: return brnn_recurrent_kernel<NTW, 64>;   // H = 2048
        default: return brnn_recurrent_kernel<NTW, 0>;
    }
}

// ------------------------------------------------------------------ co-residency guards
//
// Persistent kernels spin on other workgroups: every workgroup of the grid must be resident at
// once.  Three things can break that, each with its own guard:
//  (1) the grid does not fit the device at all (occupancy x CUs < grid): checked per (device,
//      kernel) before the launch -> the per-step fallback runs instead;
//  (2) ANOTHER PROCESS launches a persistent grid on the same device at the same time (two ranks
//      sharing one GPU): both get part of the CUs and spin forever.  "Shared-device mode" takes an
//      inter-process lease -- flock() on a per-device file in /dev/shm -- around every persistent
//      launch: stream drained, lock, launch, wait, unlock.  It is on when SCTC_SHARED_DEVICE=1
//      (the Python mirror sets it when the local ranks outnumber the visible devices) and turns
//      itself on for the rest of the process the first time a launch times out; the normal
//      one-rank-per-GPU path pays nothing;
//  (3) another STREAM of this process does the same (one-utterance-per-stream): the in-process
//      gate below admits a persistent launch only while the CUs of all persistent launches in
//      flight on other streams, plus its own, fit the device; otherwise it first waits (on the
//      host) for the oldest of them.
// What still gets through (CU masks the runtime does not report, a foreign persistent kernel of
// some other library) ends in the bounded spin -> SCTC_ERR_TIMEOUT -> the engine retries the step
// on the fallback.

static std::atomic<int> g_shared_mode{-1};   // -1: not yet read from the environment
static std::atomic<int> g_shared_hard{0};    // SCTC_SHARED_DEVICE was in the environment: no automatic switch

int recurrent_shared_device_mode()
{
    int m = g_shared_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("SCTC_SHARED_DEVICE");
        m = (e && atoi(e) != 0) ? 1 : 0;
        if (e) g_shared_hard.store(1, std::memory_order_relaxed);
        g_shared_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}

void recurrent_set_shared_device_mode(int on) { g_shared_mode.store(on ? 1 : 0, std::memory_order_relaxed); }

namespace {

static constexpr double LEASE_WAIT_S = 30.0;
static constexpr int Q1_MIN_B = 4;     // fewest utterances the single-chain flag kernel takes by default (below: the sentinel / VALU kernel; measured round 5, tools/rec_small_bench.py: 3 utterances 3.9 vs 4.6 us per step, 4: 4.9 vs 4.6, 5: 7.0 vs 4.5 at H = 1824)
static double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct DeviceLease {      // inter-process: one lock file per physical device
    std::mutex mu;                       // serialises the threads of this process (flock is per open file)
    std::map<int, int> fds;
    int fd_for(int dev)
    {
        auto it = fds.find(dev);
        if (it != fds.end()) return it->second;
        char bus[64] = "unknown";
        (void)hipDeviceGetPCIBusId(bus, sizeof(bus), dev);
        for (char* c = bus; *c; ++c)
            if (*c == ':' || *c == '.' || *c == '/') *c = '_';
        char path[160];
        int fd = -1;
        const char* dirs[2] = {"/dev/shm", "/tmp"};
        for (int k = 0; k < 2 && fd < 0; ++k) {
            snprintf(path, sizeof(path), "%s/sctc_gpu_%s.lock", dirs[k], bus);
            // the name is predictable and the directory world-writable: never follow a planted symlink,
            // never touch the mode of anything but a regular file of our own
            // the process umask is left alone (another thread creating a file at that moment would inherit a
            // cleared one: ADVICE r04): the mode is widened on the DESCRIPTOR, and only of a regular file this
            // user owns (O_NOFOLLOW: never through a planted symlink)
            fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); fd = -1; }
                else if (st.st_uid == geteuid() && (st.st_mode & 0666) != 0666) (void)fchmod(fd, 0666);
            }
        }
        fds[dev] = fd;
        return fd;
    }
};
DeviceLease g_lease;

struct LeaseGuard {
    int fd = -1;
    bool locked = false;
    explicit LeaseGuard(int dev)
    {
        g_lease.mu.lock();
        fd = g_lease.fd_for(dev);
        if (fd >= 0) {
            // bounded: another user holding the lock for good must not hang this process with its
            // mutex held -- after LEASE_WAIT_S the caller runs the step on the per-step fallback
            // Polling with a growing back-off (200 us .. 5 ms) instead of a blocking flock: the wait must end, and a
            // waiter that polls every 200 us re-acquires ahead of one that slept longer -- so every rank backs off the
            // same way and a rank that has just held the lease yields once (below) before it asks again.
            const double t0 = now_s();
            useconds_t nap = 200;
            for (;;) {
                const int rc = flock(fd, LOCK_EX | LOCK_NB);
                if (rc == 0) { locked = true; break; }
                if (errno != EWOULDBLOCK && errno != EINTR) break;
                if (now_s() - t0 > LEASE_WAIT_S) {
                    // not silent: the step falls back to one launch per time step
                    fprintf(stderr, "sctc: device lease not obtained within %.0f s (another process holds %s): this step "
                                    "runs on the per-step recurrent kernels\n", LEASE_WAIT_S, "the GPU's persistent-launch lock");
                    break;
                }
                usleep(nap);
                if (nap < 5000) nap += nap / 2;
            }
        }
    }
    ~LeaseGuard()
    {
        if (locked) (void)flock(fd, LOCK_UN);
        g_lease.mu.unlock();
    }
};

// Who else uses this PHYSICAL device?  Every process that launches a persistent grid keeps an
// exclusively flock'ed marker file /dev/shm/sctc_gpu_<pci-bus-id>.user.<pid> for its lifetime and touches it
// at every persistent launch; counting the markers that are locked AND fresh tells how many processes are
// launching persistent grids on the GPU right now,
// whatever HIP_VISIBLE_DEVICES / LOCAL_WORLD_SIZE / device ordinals look like (round 3 guessed from
// LOCAL_WORLD_SIZE > visible devices: wrong on an 8-GPU node whose launcher shows every rank one
// device, and wrong the other way on a box that exports a global visibility mask).  More than one ->
// shared-device mode switches itself on (one line on stderr).  Checked at persistent launches number
// 1, 2, 4, ... 32 and every 32nd after that (a directory scan of /dev/shm, ~20 us) while the mode is
// off; ranks that exchange their bus ids over a process group (dist_sgd.DataParallel) decide at once.
static constexpr double USER_ACTIVE_S = 0.3;     // a marker untouched for this long belongs to an idle process (a trainer launches every few ms)
static constexpr double MARKER_STALE_S = 2.0;   // an UNLOCKED marker older than this is a dead process's
struct DeviceUsers {
    std::mutex mu;
    std::map<int, std::pair<int, std::string>> mine;    // device -> (held marker fd, marker prefix)
    std::map<int, unsigned> launches;
    static bool held_by_somebody(const char* path)
    {
        const int fd = open(path, O_RDWR | O_CLOEXEC | O_NOFOLLOW);
        if (fd < 0) return false;
        struct stat st;
        bool held = false;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            if (flock(fd, LOCK_SH | LOCK_NB) != 0) {
                // alive -- and ACTIVE: its owner touches the marker at every persistent launch; a process
                // that merely still exists (a test driver waiting for its child, an idle notebook) is no
                // reason to serialise launches
                struct timespec now;
                clock_gettime(CLOCK_REALTIME, &now);
                held = errno == EWOULDBLOCK && (double)(now.tv_sec - st.st_mtim.tv_sec) + 1e-9 * (double)(now.tv_nsec - st.st_mtim.tv_nsec) < USER_ACTIVE_S;
            } else {
                // unlocked: its owner is gone -- or has created it a moment ago and not locked it yet (create and
                // lock are two calls): only a marker that has been lying there for a while is removed
                struct timespec now;
                clock_gettime(CLOCK_REALTIME, &now);
                if ((double)(now.tv_sec - st.st_mtim.tv_sec) > MARKER_STALE_S) (void)unlink(path);
                (void)flock(fd, LOCK_UN);
            }
        }
        close(fd);
        return held;
    }
    // number of live processes (this one included) that registered for device `dev`; 0: no markers to be had
    int count(int dev)
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = mine.find(dev);
        if (it == mine.end()) {
            char bus[64] = "unknown";
            (void)hipDeviceGetPCIBusId(bus, sizeof(bus), dev);
            for (char* c = bus; *c; ++c)
                if (*c == ':' || *c == '.' || *c == '/') *c = '_';
            std::string prefix = std::string("sctc_gpu_") + bus + ".user.";
            it = mine.emplace(dev, std::make_pair(-1, prefix)).first;
        }
        if (it->second.first < 0) {
            // (re)create the marker: a failed attempt -- or a marker a scanner removed between our open() and
            // flock() -- is tried again at the next check instead of leaving this process invisible for good
            const std::string path = "/dev/shm/" + it->second.second + std::to_string((long)getpid());
            int fd = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
            struct stat st, on_disk;
            if (fd >= 0 && (flock(fd, LOCK_EX | LOCK_NB) != 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) ||
                            stat(path.c_str(), &on_disk) != 0 || on_disk.st_ino != st.st_ino)) {
                close(fd);      // not lockable, or the name no longer leads to the file we hold
                fd = -1;
            }
            if (fd >= 0 && st.st_uid == geteuid() && (st.st_mode & 0666) != 0666) (void)fchmod(fd, 0666);
            it->second.first = fd;
        }
        if (it->second.first < 0) return 0;
        int n = 1;
        const std::string own = it->second.second + std::to_string((long)getpid());
        if (DIR* d = opendir("/dev/shm")) {
            while (struct dirent* e = readdir(d)) {
                if (strncmp(e->d_name, it->second.second.c_str(), it->second.second.size()) != 0) continue;
                if (own == e->d_name) continue;
                if (held_by_somebody((std::string("/dev/shm/") + e->d_name).c_str())) ++n;
            }
            closedir(d);
        }
        return n;
    }
    bool due(int dev)
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = mine.find(dev);
        if (it != mine.end() && it->second.first >= 0) (void)futimens(it->second.first, nullptr);   // "active now"
        const unsigned k = ++launches[dev];
        return k <= 32 ? (k & (k - 1)) == 0 : (k & 31) == 0;
    }
};
DeviceUsers g_users;

struct PersistentGate {   // in-process: persistent launches in flight, per device
    struct Entry { hipEvent_t ev; hipStream_t stream; int cus; };
    std::mutex mu;
    std::map<int, std::vector<Entry>> inflight;
    std::vector<hipEvent_t> pool;

    // Waits (on the host) until `cus` more compute units' worth of persistent workgroups fit next to
    // what other streams have in flight, then launches and registers the launch's event -- all under
    // ONE lock: two host threads on different streams cannot both see an empty device and both
    // launch a whole-device grid (check-then-act race of the round-3 admit() / launched() pair).
    template <typename Launch>
    int admit_and_launch(int dev, hipStream_t stream, int cus, int device_cus, Launch launch)
    {
        std::lock_guard<std::mutex> lock(mu);
        std::vector<Entry>& v = inflight[dev];
        for (;;) {
            int other = 0;
            for (size_t i = 0; i < v.size();) {
                const hipError_t q = hipEventQuery(v[i].ev);
                if (q == hipSuccess) {
                    pool.push_back(v[i].ev);
                    v.erase(v.begin() + i);
                    continue;
                }
                if (q != hipErrorNotReady) (void)hipGetLastError();
                if (v[i].stream != stream) other += v[i].cus;   // same stream: serialised anyway
                ++i;
            }
            if (other == 0 || other + cus <= device_cus) break;
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i].stream != stream) {
                    SCTC_HIP_TRY(hipEventSynchronize(v[i].ev));
                    break;
                }
        }
        SCTC_HIP_TRY(launch());
        hipEvent_t ev;
        if (!pool.empty()) { ev = pool.back(); pool.pop_back(); }
        else SCTC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        SCTC_HIP_TRY(hipEventRecord(ev, stream));
        v.push_back(Entry{ev, stream, cus});
        return SCTC_OK;
    }
};
PersistentGate g_gate;

}  // namespace

// The dynamic-LDS attribute is set once per (device, kernel) and the grid is checked against the
// blocks per CU the kernel can have: its LDS share (these kernels are LDS-limited by construction,
// their __launch_bounds__ guarantee the registers for 1 or 2 blocks per CU; the runtime's
// occupancy query under-reports kernels that use most of the 160 KiB -- it says 1 for the
// two-chain kernel at 2 x 79 KiB, which does run two per CU -- so it is only the second opinion).
static int prepare_kernel(RecKernel k, int threads, size_t smem, int max_per_cu, int dev, int* blocks_per_cu)
{
    struct Info { size_t smem_set = 0; size_t occ_smem = (size_t)-1; int occ = 0; };
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, Info> table;
    std::lock_guard<std::mutex> lock(mu);
    Info& in = table[std::make_pair(dev, reinterpret_cast<const void*>(k))];
    if (in.smem_set < smem) {
        SCTC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        in.smem_set = smem;
    }
    if (in.occ_smem != smem) {
        int n = 0, lds = 0;
        SCTC_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(k), threads, smem));
        SCTC_HIP_TRY(hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev));
        const int by_lds = smem ? (int)std::min<size_t>((size_t)max_per_cu, (size_t)lds / smem) : max_per_cu;
        in.occ = std::max(std::min(n, max_per_cu), by_lds);
        in.occ_smem = smem;
        if (getenv("SCTC_VERBOSE"))
            fprintf(stderr, "sctc: recurrent kernel %p on device %d: %zu B LDS, occupancy query %d, LDS model %d blocks/CU\n",
                    reinterpret_cast<const void*>(k), dev, smem, n, by_lds);
    }
    *blocks_per_cu = in.occ;
    return SCTC_OK;
}

static int launch_fallback(const RecArgs& a, hipStream_t stream)
{
    const int nch = a.Hp / 16, ntiles = (a.B + 15) / 16;
    for (int j = 0; j < a.Tmax; ++j)
        hipLaunchKernelGGL(brnn_recurrent_step_kernel, dim3(nch, 2, ntiles), dim3(256), 0, stream, a, j);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

struct LaunchCtx {
    int dev, cus;
    hipStream_t stream;
    int* path;      // out: REC_PATH_* of the last launch
};

// launches a persistent grid if it can be co-resident (returns 1), else leaves it to the caller (0)
static int read_error_word(const RecArgs& a, hipStream_t stream, unsigned* e)
{
    *e = 0;
    SCTC_HIP_TRY(hipMemcpyAsync(e, a.counters + 2, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    SCTC_HIP_TRY(hipStreamSynchronize(stream));
    return SCTC_OK;
}

static int launch_persistent(RecKernel k, int grid, size_t smem, int max_per_cu, size_t sentinel_bytes,
                             const RecArgs& a, const LaunchCtx& cx, bool* done, int threads = 256)
{
    *done = false;
    int per_cu = 0;
    SCTC_TRY(prepare_kernel(k, threads, smem, max_per_cu, cx.dev, &per_cu));
    if (per_cu < 1 || (int64_t)per_cu * cx.cus < grid) return SCTC_OK;
    const int need_cus = (grid + per_cu - 1) / per_cu;
    if (sentinel_bytes)   // sentinel-fill the exchange rows (both directions)
        SCTC_HIP_TRY(hipMemsetAsync(a.xbuf, 0xFF, sentinel_bytes, cx.stream));
    if (!recurrent_shared_device_mode() && !g_shared_hard.load(std::memory_order_relaxed) && g_users.due(cx.dev)) {
        const int n = g_users.count(cx.dev);
        if (n > 1) {
            recurrent_set_shared_device_mode(1);
            fprintf(stderr, "sctc: %d processes run persistent kernels on this GPU: shared-device mode on "
                            "(their launches take turns under /dev/shm/sctc_gpu_*.lock)\n", n);
        }
    }
    if (recurrent_shared_device_mode()) {
        // the lease is held exactly while the grid runs: nothing of ours queued in front of it,
        // nobody else's persistent grid next to it
        SCTC_HIP_TRY(hipStreamSynchronize(cx.stream));
        LeaseGuard lease(cx.dev);
        if (!lease.locked) return SCTC_OK;            // no lock file to be had: per-step fallback
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), smem, cx.stream, a);
        SCTC_HIP_TRY(hipGetLastError());
        // synchronous anyway: a time-out is reported by THIS launch, not found at the end of the step
        // behind later launches that ran on its garbage
        unsigned e = 0;
        SCTC_TRY(read_error_word(a, cx.stream, &e));
        *done = true;
        if (cx.path) *cx.path = REC_PATH_PERSISTENT_LEASED;
        if (e != 0)
            return set_error(SCTC_ERR_TIMEOUT, "recurrent kernel: a persistent launch under the device lease gave "
                             "up waiting for its peers (%d workgroups not co-resident)", grid);
        return SCTC_OK;
    }
    SCTC_TRY(g_gate.admit_and_launch(cx.dev, cx.stream, need_cus, cx.cus, [&]() {
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), smem, cx.stream, a);
        return hipGetLastError();
    }));
    *done = true;
    if (cx.path) *cx.path = REC_PATH_PERSISTENT;
    return SCTC_OK;
}

static int launch_recurrent_one(const RecArgs& a, const LaunchCtx& cx)
{
    const int nwg = a.Hp / 16;
    const int ntiles = (a.B + 15) / 16;
    hipStream_t stream = cx.stream;
    bool done = false;
    // Step flags only: the error word (counters[2]) is STICKY across the launches of a step -- the
    // engine clears it once per step (recurrent_clear_error) and reads it once at the end, so a
    // forward-pass time-out is not wiped by the BPTT launch's memset, nor an earlier chunk's (more
    // than 128 utterances) by a later one's.
    SCTC_HIP_TRY(hipMemsetAsync(a.counters + 32, 0, (REC_COUNTER_WORDS - 32) * sizeof(unsigned), stream));
    if (a.variant == 3 || 2 * nwg > cx.cus) {
        if (cx.path) *cx.path = REC_PATH_FALLBACK;
        return launch_fallback(a, stream);
    }
    // measured at H=1824: 2.5 us per step for one utterance + ~1.2 us per further one (VALU FMAs),
    // against 7.4 us for the flag/MFMA kernel: worth it up to 5 utterances
    // 4..16 utterances are ONE 16-utterance chain: the flag kernel of the 17..32 case with a single chain per direction
    // (one workgroup per CU instead of two).  Round 5 ("7.16 us at B = 16 is suspicious" -- it was): at
    // H = 1824 it needs 4.66 / 4.57 / 4.80 / 4.97 us per step at 6 / 8 / 12 / 16 utterances where the sentinel / MFMA
    // kernel below needs 4.92 / 5.00 / 6.09 / 7.16 (H = 2048: 5.45 against 7.31 at 16), with bit-identical costs and
    // gradients (tools/rec_mid_bench.py): it is the default for fp32 operands; variant 42 keeps the sentinel kernel.
    // At 4 and 5 utterances it also beats the sentinel / VALU kernel (whose step grows by 1 us per utterance and jumps at
    // the fifth: 2.2 / 3.0 / 3.9 / 4.9 / 7.0 us for 1..5): variant 43 keeps that one up to 8, 44 forces this one from 1.
    if ((!a.prec16 || a.B <= 5) && (a.B > Q1_MIN_B - 1 || a.variant == 44) && a.B <= 16 && a.variant != 1 && a.variant != 42 && a.variant != 43 &&
        2 * nwg <= cx.cus) {
        RecKernel qk = nullptr;
        int ncq = 0, nreg = 0;
        switch (nwg) {
            case 32:  qk = brnn_recurrent_q_kernel<8, 0>;   ncq = 8;  nreg = 0;  break;  // H = 512
            case 64:  qk = brnn_recurrent_q_kernel<16, 0>;  ncq = 16; nreg = 0;  break;  // H = 1024
            case 114: qk = brnn_recurrent_q_kernel<29, 10>; ncq = 29; nreg = 10; break;  // H = 1824
            case 128: qk = brnn_recurrent_q_kernel<32, 13>; ncq = 32; nreg = 13; break;  // H = 2048
            default: break;
        }
        if (qk) {
            RecArgs b = a;
            b.linear_map = 1;   // linear block -> (chain, producer) map: chain = direction, tile 0 only (variant keeps its A/B meaning)
            if (b.variant == 44) b.variant = 0;
            const size_t smem = sizeof(float4) * ((size_t)4 * (ncq - nreg) * 64 + 3 * 64);
            SCTC_TRY(launch_persistent(qk, 2 * nwg, smem, 2, 0, b, cx, &done));
            if (done) return SCTC_OK;
        }
    }
    if ((a.B <= 5 || (a.B <= 8 && a.variant == 43)) && a.variant != 1) {     // 43: the 8-utterance instantiation up to 8 (experiment)
        RecKernel sk = nullptr;
        const bool s4 = a.B <= 4;
        switch (a.Hp / 32) {
            case 16: sk = s4 ? brnn_recurrent_s_kernel<16, 4> : brnn_recurrent_s_kernel<16, 8>; break;   // H = 512
            case 32: sk = s4 ? brnn_recurrent_s_kernel<32, 4> : brnn_recurrent_s_kernel<32, 8>; break;   // H = 1024
            case 57: sk = s4 ? brnn_recurrent_s_kernel<57, 4> : brnn_recurrent_s_kernel<57, 8>; break;   // H = 1824
            case 64: sk = s4 ? brnn_recurrent_s_kernel<64, 4> : brnn_recurrent_s_kernel<64, 8>; break;   // H = 2048
            default: break;
        }
        if (sk) {
            const size_t smem = sizeof(float) * 2 * (s4 ? 4 : 8) * a.Hp;
            // <= 239 VGPRs and 58 KiB of LDS (1..4 utterances at H = 1824): TWO of these workgroups fit
            // a CU, so the grids of two streams (one utterance per stream) can be co-resident -- the
            // in-process gate counts this launch as half the device
            SCTC_TRY(launch_persistent(sk, 2 * nwg, smem, 2, (size_t)2 * a.n_xrows * a.Hp * sizeof(float), a, cx, &done));
            if (done) return SCTC_OK;
        }
    }
    if (a.prec16 && a.B > 5 && a.B <= 16 && a.variant != 1) {
        // "fp16 activations": 16-bit state exchange and weights (float16 forward, bfloat16 BPTT)
        RecKernel hk = nullptr;
        const bool bf = a.transpose != 0;
        const int nch = a.Hp / 32;
        switch ((nch + 3) / 4) {
            case 4:  if (nch == 16) hk = bf ? brnn_recurrent_mh_kernel<4, true>  : brnn_recurrent_mh_kernel<4, false>;  break;  // H = 512
            case 8:  if (nch == 32) hk = bf ? brnn_recurrent_mh_kernel<8, true>  : brnn_recurrent_mh_kernel<8, false>;  break;  // H = 1024
            case 15: if (nch == 57) hk = bf ? brnn_recurrent_mh_kernel<15, true> : brnn_recurrent_mh_kernel<15, false>; break;  // H = 1824
            case 16: if (nch == 64) hk = bf ? brnn_recurrent_mh_kernel<16, true> : brnn_recurrent_mh_kernel<16, false>; break;  // H = 2048
            default: break;
        }
        const size_t smem = (((size_t)16 * (a.Hp + 8) * 2 + 15) / 16) * 16 + 3 * 64 * sizeof(float4);
        if (hk && smem <= 160 * 1024) {
            SCTC_TRY(launch_persistent(hk, 2 * nwg, smem, 1, (size_t)2 * a.n_xrows * a.Hp * 2, a, cx, &done));
            if (done) return SCTC_OK;
        }
    }
#ifdef SCTC_REC_EXPERIMENTS
    if (a.B > 5 && a.B <= 16 && a.variant != 1) {
        RecKernel mk = nullptr;
        switch ((nwg + 3) / 4) {
            case 8:  if (nwg == 32)  mk = brnn_recurrent_m_kernel<8>;  break;   // H = 512
            case 16: if (nwg == 64)  mk = brnn_recurrent_m_kernel<16>; break;   // H = 1024
            case 29: if (nwg == 114) mk = brnn_recurrent_m_kernel<29>; break;   // H = 1824
            case 32: if (nwg == 128) mk = brnn_recurrent_m_kernel<32>; break;   // H = 2048
            default: break;
        }
        const size_t smem = sizeof(float) * ((size_t)16 * (a.Hp + 4) + 3 * 256);
        if (mk && smem <= 160 * 1024) {
            SCTC_TRY(launch_persistent(mk, 2 * nwg, smem, 1, (size_t)2 * a.n_xrows * a.Hp * sizeof(float), a, cx, &done));
            if (done) return SCTC_OK;
        }
    }
#endif
#ifdef SCTC_REC_EXPERIMENTS
    // 17..32 utterances, both chains of a direction in one 8-wave workgroup (round 4); variant 4 keeps
    // the two-workgroups-per-CU kernel of rounds 1-3
    const int q8_variant = (a.variant >= 8 && a.variant < 40) ? a.variant : (a.variant == 0 ? REC_Q8_DEFAULT : 0);   // 40..: A/B switches of other kernels
    if (ntiles == 2 && q8_variant >= 8) {
        RecKernel k8 = nullptr;
        switch (nwg) {
            case 32:  k8 = brnn_recurrent_q8_kernel<8>;  break;   // H = 512
            case 64:  k8 = brnn_recurrent_q8_kernel<16>; break;   // H = 1024
            case 114: k8 = brnn_recurrent_q8_kernel<29>; break;   // H = 1824
            case 128: k8 = brnn_recurrent_q8_kernel<32>; break;   // H = 2048
            default: break;
        }
        const int grid = 8 * ((nwg + 3) / 4);
        const size_t smem = sizeof(float4) * ((size_t)nwg * 64 + 6 * 64) + sizeof(unsigned) * Q8_SYNC_WORDS;
        if (k8 && grid <= cx.cus && smem <= 160 * 1024) {
            RecArgs b = a;
            b.variant = q8_variant;
            SCTC_TRY(launch_persistent(k8, grid, smem, 1, 0, b, cx, &done, 512));
            if (done) return SCTC_OK;
        }
    }
#endif
    if (ntiles == 2 && a.variant != 1 && 4 * nwg <= 2 * cx.cus && (4 * nwg) % 8 == 0) {
        // two chains per CU; NREG keeps the LDS share of the slab at <= 76 KiB per workgroup
        RecKernel qk = nullptr;
        int ncq = 0, nreg = 0;
        switch (nwg) {
            case 32:  qk = brnn_recurrent_q_kernel<8, 0>;   ncq = 8;  nreg = 0;  break;  // H = 512
            case 64:  qk = brnn_recurrent_q_kernel<16, 0>;  ncq = 16; nreg = 0;  break;  // H = 1024
            case 114: qk = brnn_recurrent_q_kernel<29, 10>; ncq = 29; nreg = 10; break;  // H = 1824
            case 128: qk = brnn_recurrent_q_kernel<32, 13>; ncq = 32; nreg = 13; break;  // H = 2048
            default: break;
        }
        if (qk) {
            const size_t smem = sizeof(float4) * ((size_t)4 * (ncq - nreg) * 64 + 3 * 64);
            SCTC_TRY(launch_persistent(qk, 4 * nwg, smem, 2, 0, a, cx, &done));
            if (done) return SCTC_OK;     // otherwise: the one-workgroup-per-CU kernel below
        }
    }
    // 33..128 utterances: 32 units x half the utterance tiles per CU, two alternating sub-chains (round 6);
    // variants 1 / 40 / 47 keep the one-slab-per-CU kernel below
    // (H = 512 / 1024: 64 / 128 workgroups, no faster than the one-slab-per-CU kernel on as many CUs -- 9.3 against 8.4 us per
    // step at H = 1024, 64 utterances: there the tiled form runs only when asked for, SCTC_REC_VARIANT=50: tests)
    if (ntiles > 2 && a.variant != 1 && a.variant != 40 && a.variant != 47 && 4 * (a.Hp / 32) <= cx.cus &&
        (a.Hp >= 1824 || a.variant == 50)) {
        RecKernel tk = nullptr;
        const bool one = ntiles <= 4;       // one utterance tile per sub-chain
        int nldsf = 0;
        // <chunks per wave, weight fragments in registers, tiles per sub-chain, batches per phase, ring depth, publish batch,
        // batch in front of which the next phase's flags are checked>; SCTC_REC_TCFG=1: the other of (eight batches, ring of
        // four) / (six, three) (A/B of the schedule, tools/rec_tiled_sweep.sh: equal within the noise); 2: ring of two (one batch of lookahead: 128 utterances
        // 19.5 instead of 17.2 us per step -- the ring's depth is what hides the exchange loads' latency)
        static const int tcfg = getenv("SCTC_REC_TCFG") ? atoi(getenv("SCTC_REC_TCFG")) : 0;
        switch (nwg) {
            case 32:  tk = one ? brnn_recurrent_t_kernel<8, 0, 1, 4, 2, 0, 3>  : brnn_recurrent_t_kernel<8, 0, 2, 4, 2, 0, 3>;  nldsf = 16; break;  // H = 512
            case 64:  tk = one ? brnn_recurrent_t_kernel<16, 0, 1, 4, 2, 1, 3> : brnn_recurrent_t_kernel<16, 0, 2, 4, 2, 1, 3>; nldsf = 32; break;  // H = 1024
            case 114:   // H = 1824
                nldsf = 33;
                if (one) tk = tcfg == 1 ? brnn_recurrent_t_kernel<29, 25, 1, 6, 3, 1, 4> : tcfg == 2 ? brnn_recurrent_t_kernel<29, 25, 1, 8, 2, 1, 7>
                              : brnn_recurrent_t_kernel<29, 25, 1, 8, 4, 1, 5>;
                else tk = tcfg == 1 ? brnn_recurrent_t_kernel<29, 25, 2, 8, 4, 1, 5> : tcfg == 2 ? brnn_recurrent_t_kernel<29, 25, 2, 8, 2, 1, 7>
                          : brnn_recurrent_t_kernel<29, 25, 2, 6, 3, 1, 4>;     // (8, 4: 128 ring registers, 28 of them parked in the other file)
                break;
            case 128:   // H = 2048
                nldsf = 33;
                if (one) tk = tcfg == 1 ? brnn_recurrent_t_kernel<32, 31, 1, 6, 3, 1, 4> : brnn_recurrent_t_kernel<32, 31, 1, 8, 4, 1, 5>;
                else tk = tcfg == 1 ? brnn_recurrent_t_kernel<32, 31, 2, 6, 3, 1, 4> : brnn_recurrent_t_kernel<32, 31, 2, 8, 4, 1, 5>;
                break;
            default: break;
        }
        if (tk) {
            const size_t smem = sizeof(float4) * 64 * ((size_t)4 * nldsf + 2 * 3 * (one ? 2 : 4)) + (16 * 8 + 8 * 32) * sizeof(unsigned);
            SCTC_TRY(launch_persistent(tk, 4 * (a.Hp / 32), smem, 1, 0, a, cx, &done));
            if (done) return SCTC_OK;
        }
    }
    const int ntw = ntiles <= 2 ? 1 : (ntiles <= 4 ? 2 : 4);
    const size_t smem = sizeof(float4) * ((size_t)nwg * 64 + 2 * ntw * 64);
    if (smem <= 160 * 1024) {
        const bool pipe = a.variant != 40;     // 40: exchange loads and MFMAs of a batch one after the other (rounds 1-4; A/B)
        RecKernel kern = ntw == 1 ? pick_kernel<1>(nwg / 2, false)
                         : (ntw == 2 ? pick_kernel<2>(nwg / 2, pipe) : pick_kernel<4>(nwg / 2, pipe));
        SCTC_TRY(launch_persistent(kern, 2 * nwg, smem, 1, 0, a, cx, &done));
        if (done) return SCTC_OK;
    }
    if (cx.path) *cx.path = REC_PATH_FALLBACK;
    return launch_fallback(a, stream);
}

int recurrent_clear_error(unsigned* counters, hipStream_t stream)
{
    SCTC_HIP_TRY(hipMemsetAsync(counters, 0, 32 * sizeof(unsigned), stream));
    return SCTC_OK;
}

int launch_recurrent(const RecArgs& a_in, hipStream_t stream, int* path)
{
    RecArgs a = a_in;
    if (a.poll_delay < 0) a.poll_delay = a.Hp > 1024 ? 5 : 0;
    char why[128];
    if (!recurrent_supported(a.Hp, a.B, why, sizeof(why)))
        return set_error(SCTC_ERR_ARG, "recurrent kernel: %s", why);
    if (a.Tmax <= 0 || a.B <= 0) return SCTC_OK;
    if ((size_t)a.n_xrows * a.Hp * sizeof(float) >= ((size_t)1 << 31))
        return set_error(SCTC_ERR_ARG, "recurrent kernel: %d exchange rows x %d units exceed the "
                         "2 GiB buffer addressing", a.n_xrows, a.Hp);
    LaunchCtx cx;
    cx.stream = stream;
    cx.path = path;
    cx.dev = 0;
    cx.cus = 0;
    SCTC_HIP_TRY(hipGetDevice(&cx.dev));
    SCTC_HIP_TRY(hipDeviceGetAttribute(&cx.cus, hipDeviceAttributeMultiprocessorCount, cx.dev));
    // utterances are independent: more than 128 run as consecutive launches of <= 128 (sorted by
    // length, so a later launch covers no more steps than its first utterance has frames)
    constexpr int MAXB = 128;
    for (int b0 = 0; b0 < a_in.B;) {
        int nb = std::min(MAXB, a_in.B - b0);
        // How a minibatch is cut into launches (utterances are independent, sorted by length: a later launch covers only as
        // many steps as ITS longest utterance has).  Round 6, with the units x utterances kernel (tools/rec_tiled_sweep.sh,
        // H = 1824, us per time step, equal lengths): 33..64 utterances in ONE launch (40 / 48: 10.5 against 11.2 / 11.7 for
        // 32 + the rest, round 5's cut); 65..96 as 64 + the rest (72 / 80: 15.0 against 18.7 in one launch of the two-tile
        // form, which multiplies its empty tile slots; 96 = 64 + 32: 9.5 + 6.4); 97..128 in one launch (18.7).
        // Variant 45: never cut.  Kernels without the tiled form (variants 1 / 40 / 47, other layer sizes) keep round 5's cuts.
        const bool tiled = ((a.variant == 0 && (a.Hp == 1824 || a.Hp == 2048)) || (a.variant == 50 && (a.Hp == 512 || a.Hp == 1024 || a.Hp == 1824 || a.Hp == 2048))) &&
                           !a.prec16 && 4 * (a.Hp / 32) <= cx.cus;
        if (tiled) {
            if (nb > 64 && nb <= 96) nb = 64;
        } else {
            if (nb > 32 && nb <= 48 && a.variant == 0 && !a.prec16) nb = 32;
            else if (nb > 64 && nb <= 80 && a.variant == 0 && !a.prec16) nb = 64;
        }
        RecArgs c = a;
        c.b_off = a_in.b_off + b0;
        c.T_b = a_in.T_b + b0;
        c.B = nb;
        if (a_in.T_host) c.Tmax = std::min(a_in.Tmax, (int)a_in.T_host[b0]);
        if (c.variant == 45) c.variant = 0;
        SCTC_TRY(launch_recurrent_one(c, cx));
        b0 += nb;
    }
    return SCTC_OK;
}

}  // namespace sctc
