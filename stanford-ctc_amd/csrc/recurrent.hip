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
// time-batched GEMMs, and into a small exchange buffer laid out
// [16-unit chunk][utterance][16] -- chunk c is written by workgroup c as one
// contiguous block and read by every workgroup as 1 KiB-per-wave coalesced
// loads that land directly in MFMA B-fragment order.
// Steps are separated by a grid barrier per direction (monotonic arrival counter,
// agent-scope release/acquire as MI355X_MICROARCH prescribes; every spin bounded).
//
// The K index inside a 16-chunk is permuted (k = 16c + 4*(lane>>4) + q for MFMA q)
// identically for W and x, so both operands are 16-byte vector loads.
#include "common.h"
#include "recurrent.h"

namespace sctc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr unsigned long long SPIN_TIMEOUT_TICKS = 300000000ull;  // 3 s of the 100 MHz clock

__device__ __forceinline__ float4 ld_x(const float* p, int sync_mode)
{
    if (sync_mode == 1) {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
        unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float4 r;
        r.x = __uint_as_float((unsigned)lo);
        r.y = __uint_as_float((unsigned)(lo >> 32));
        r.z = __uint_as_float((unsigned)hi);
        r.w = __uint_as_float((unsigned)(hi >> 32));
        return r;
    }
    return *reinterpret_cast<const float4*>(p);
}

__device__ __forceinline__ void st_x(float* p, float4 v, int sync_mode)
{
    if (sync_mode == 1) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
        unsigned long long lo = (unsigned long long)__float_as_uint(v.x) |
                                ((unsigned long long)__float_as_uint(v.y) << 32);
        unsigned long long hi = (unsigned long long)__float_as_uint(v.z) |
                                ((unsigned long long)__float_as_uint(v.w) << 32);
        __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    *reinterpret_cast<float4*>(p) = v;
}

// all threads call; publishes this workgroup's stores of the step
__device__ __forceinline__ void grid_arrive(unsigned* ctr, int sync_mode)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sync_mode == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // keep the wait behind buffer_wbl2
        }
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// all threads call; returns after every workgroup of the group has arrived `target` times
__device__ __forceinline__ void grid_wait(unsigned* ctr, unsigned target, unsigned* err,
                                          int sync_mode)
{
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (wall_clock64() - t0 > SPIN_TIMEOUT_TICKS) {
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        if (sync_mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int NTW>
__global__ __launch_bounds__(256, 1) void brnn_recurrent_kernel(RecArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x & 1, wg = blockIdx.x >> 1;
    const int Hp = p.Hp, nch = Hp >> 4, nch_half = nch >> 1, nwg = nch;
    const int row0 = wg * 16;
    const int uj = lane & 15, kq = lane >> 4;
    const int kh = wave >> 1, ng = wave & 1;
    const int Bp = p.Bp;
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
    unsigned* ctr = p.counters + g;
    unsigned* err = p.counters + 2;
    float* xg = p.xbuf + (size_t)g * 2 * nch * Bp * 16;

    // my utterances (fixed over time) and their lengths
    int ub[NTW], uT[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        ub[i] = (ng + 2 * i) * 16 + uj;
        uT[i] = ub[i] < p.B ? p.T_b[ub[i]] : 0;
    }

    for (int j = 0; j < p.Tmax; ++j) {
        const int nact = p.nact[j];
        // four independent accumulators per tile: the 16x16x4 f32 MFMA has a 40-cycle
        // dependent latency against a 32-cycle issue interval
        f32x4 acc[NTW][4];
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] = {0.f, 0.f, 0.f, 0.f};
        bool active[NTW];
        int64_t orow[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            active[i] = j < uT[i];
            const int t = desc ? uT[i] - 1 - j : j;
            orow[i] = active[i] ? (int64_t)p.rowbase[t] + ub[i] : 0;
        }
        // prefetch the per-frame additive term (independent of the recurrence)
        float4 pre4[NTW], act4[NTW];
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                pre4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                act4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (active[i]) {
                    pre4[i] = *reinterpret_cast<const float4*>(pre + orow[i] * ld + row0 + 4 * kq);
                    if (act) act4[i] = *reinterpret_cast<const float4*>(act + orow[i] * ld + row0 + 4 * kq);
                }
            }
        }

        if (j > 0) {
            grid_wait(ctr, (unsigned)j * nwg, err, sync_mode);
            const float* xr = xg + (size_t)((j - 1) & 1) * nch * Bp * 16;
            const int c_beg = kh * nch_half, c_end = c_beg + nch_half;
            // every tile (ng + 2i) with a live utterance at step j
            bool tile_on[NTW];
#pragma unroll
            for (int i = 0; i < NTW; ++i) tile_on[i] = (ng + 2 * i) * 16 < nact;
            // All x loads of the step are issued before the first MFMA (up to 64 float4 =
            // 256 VGPRs per lane; one wave per SIMD owns the whole register file): the
            // exchange buffer comes from a remote L2 / the Infinity Cache with microsecond
            // latency, so the step time is latency + streaming, not latency per batch.
            constexpr int XB = 64 / NTW;
            for (int cb = c_beg; cb < c_end; cb += XB) {
                float4 x[XB][NTW];
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int c = cb + u;
#pragma unroll
                    for (int i = 0; i < NTW; ++i) {
                        if (c < c_end && tile_on[i])
                            x[u][i] = ld_x(xr + ((size_t)c * Bp + ub[i]) * 16 + 4 * kq, sync_mode);
                        else
                            x[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int c = cb + u;
                    if (c < c_end) {
                        const float4 a = Wl[c * 64 + lane];
#pragma unroll
                        for (int i = 0; i < NTW; ++i) {
                            if (tile_on[i]) {
                                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x[u][i].x, acc[i][0], 0, 0, 0);
                                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x[u][i].y, acc[i][1], 0, 0, 0);
                                acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x[u][i].z, acc[i][2], 0, 0, 0);
                                acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x[u][i].w, acc[i][3], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            // fold the two K halves: upper half parks its partials in LDS
            if (kh == 1) {
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    f32x4 s = (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
                    red[(ng * NTW + i) * 64 + lane] = make_float4(s[0], s[1], s[2], s[3]);
                }
            }
            __syncthreads();
        }

        if (kh == 0) {
            float* xw = xg + (size_t)(j & 1) * nch * Bp * 16;
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                if (!active[i]) continue;
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j > 0) {
                    const float4 r = red[(ng * NTW + i) * 64 + lane];
                    const f32x4 q = (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
                    s = make_float4(q[0] + r.x, q[1] + r.y, q[2] + r.z, q[3] + r.w);
                }
                float4 o;
                if (!act) {
                    // hFor[:,t] = clip(z[:,t] + Wf hFor[:,t-1], 0, maxAct)   brnnet.py:146-152
                    o.x = fminf(fmaxf(pre4[i].x + s.x, 0.f), hi);
                    o.y = fminf(fmaxf(pre4[i].y + s.y, 0.f), hi);
                    o.z = fminf(fmaxf(pre4[i].z + s.z, 0.f), hi);
                    o.w = fminf(fmaxf(pre4[i].w + s.w, 0.f), hi);
                } else {
                    // deltas[:,t] = (deltas[:,t] + W^T deltas[:,t+-1]) * [0 < h < maxAct]  brnnet.py:208-224
                    o.x = (act4[i].x > 0.f && act4[i].x < hi) ? pre4[i].x + s.x : 0.f;
                    o.y = (act4[i].y > 0.f && act4[i].y < hi) ? pre4[i].y + s.y : 0.f;
                    o.z = (act4[i].z > 0.f && act4[i].z < hi) ? pre4[i].z + s.z : 0.f;
                    o.w = (act4[i].w > 0.f && act4[i].w < hi) ? pre4[i].w + s.w : 0.f;
                }
                *reinterpret_cast<float4*>(out + orow[i] * ld + row0 + 4 * kq) = o;
                st_x(xw + ((size_t)wg * Bp + ub[i]) * 16 + 4 * kq, o, sync_mode);
            }
        }
        if (j + 1 < p.Tmax) grid_arrive(ctr, sync_mode);
    }
}

size_t recurrent_xbuf_floats(int Hp, int B)
{
    const int Bp = (int)round_up(B, 16);
    return (size_t)2 * 2 * (Hp / 16) * Bp * 16;
}

int recurrent_supported(int Hp, int B, char* why, int why_len)
{
    if (Hp % 32 != 0) { snprintf(why, why_len, "layer size %d not padded to 32", Hp); return 0; }
    if (2 * (Hp / 16) > 256) {
        snprintf(why, why_len, "layer size %d needs %d co-resident workgroups (> 256 CUs)", Hp,
                 2 * (Hp / 16));
        return 0;
    }
    if (B > 128) { snprintf(why, why_len, "minibatch %d > 128 utterances per launch", B); return 0; }
    return 1;
}

int launch_recurrent(const RecArgs& a, hipStream_t stream)
{
    char why[128];
    if (!recurrent_supported(a.Hp, a.B, why, sizeof(why)))
        return set_error(SCTC_ERR_ARG, "recurrent kernel: %s", why);
    if (a.Tmax <= 0 || a.B <= 0) return SCTC_OK;
    int dev = 0, cus = 0;
    SCTC_HIP_TRY(hipGetDevice(&dev));
    SCTC_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nwg = a.Hp / 16;
    if (2 * nwg > cus)
        return set_error(SCTC_ERR_ARG, "recurrent kernel: needs %d co-resident workgroups, device "
                         "has %d CUs", 2 * nwg, cus);
    const int ntiles = (a.B + 15) / 16;
    const int ntw = ntiles <= 2 ? 1 : (ntiles <= 4 ? 2 : 4);
    const size_t smem = sizeof(float4) * ((size_t)nwg * 64 + 2 * ntw * 64);
    SCTC_HIP_TRY(hipMemsetAsync(a.counters, 0, 4 * sizeof(unsigned), stream));
    void (*kern)(RecArgs) = ntw == 1 ? brnn_recurrent_kernel<1>
                            : (ntw == 2 ? brnn_recurrent_kernel<2> : brnn_recurrent_kernel<4>);
    SCTC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(2 * nwg), dim3(256), smem, stream, a);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}

}  // namespace sctc
