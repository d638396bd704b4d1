// sctc_probe_fabric(): measures, on the device, the hand-off latencies the persistent
// recurrent kernel's exchange protocol is built from: a flag ping-pong between two
// workgroups on the SAME XCD and on DIFFERENT XCDs (HW_REG_XCC_ID tells which), for each
// cache-scope encoding of the polling load (sc0 / sc1 / sc0+sc1), and a 1 KiB "tagged
// payload" hand-off (every 16-byte granule carries its own sequence number, no separate flag).
// Diagnostics only (tests/gpu_diag.py fabric); results feed DESIGN.md 4.2.
#include "diag_common.h"

namespace sctc {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int PROBE_WGS = 256;
static constexpr int PROBE_ITERS = 2000;
static constexpr unsigned long long PROBE_TIMEOUT = 50000000ull;  // 0.5 s of the 100 MHz clock

__device__ __forceinline__ unsigned xcc_id()
{
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 0xf;
}

template <int MODE>
__device__ __forceinline__ unsigned poll_load(const unsigned* p)
{
    unsigned v;
    if (MODE == 0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__device__ bool wait_flag(const unsigned* p, unsigned want)
{
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while (poll_load<MODE>(p) < want) {
        if ((++spins & 1023u) == 0 && wall_clock64() - t0 > PROBE_TIMEOUT) return false;
    }
    return true;
}

__device__ __forceinline__ void store_flag(unsigned* p, unsigned v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one ping-pong series; returns microseconds per round trip, or -1 on timeout
template <int MODE>
__device__ float pingpong(bool is_a, unsigned* fa, unsigned* fb)
{
    const unsigned long long t0 = wall_clock64();
    bool ok = true;
    for (unsigned i = 1; i <= PROBE_ITERS && ok; ++i) {
        if (is_a) {
            store_flag(fa, i);
            ok = wait_flag<MODE>(fb, i);
        } else {
            ok = wait_flag<MODE>(fa, i);
            store_flag(fb, i);
        }
    }
    if (!ok) {   // release the partner
        store_flag(fa, 0xffffffffu);
        store_flag(fb, 0xffffffffu);
        return -1.f;
    }
    return (float)(wall_clock64() - t0) * 0.01f / PROBE_ITERS;
}

// 1 KiB tagged payload: lane l of wave 0 stores {x, y, z, seq} as one 16-byte store;
// the receiver reloads until every lane sees seq.  One-way latency = round trip / 2.
__device__ float tagged_pingpong(bool is_a, u32x4* da, u32x4* db, int lane)
{
    const unsigned long long t0 = wall_clock64();
    bool ok = true;
    for (unsigned i = 1; i <= PROBE_ITERS && ok; ++i) {
        u32x4 v = {i * 3u + lane, i, (unsigned)lane, i};
        u32x4* mine = is_a ? da : db;
        u32x4* theirs = is_a ? db : da;
        if (is_a) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(mine + lane), "v"(v) : "memory");
        const unsigned long long t1 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            u32x4 r;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(theirs + lane) : "memory");
            if (__all(r[3] >= i)) break;
            if ((++spins & 1023u) == 0 && wall_clock64() - t1 > PROBE_TIMEOUT) { ok = false; break; }
        }
        if (!is_a && ok) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(mine + lane), "v"(v) : "memory");
    }
    return ok ? (float)(wall_clock64() - t0) * 0.01f / PROBE_ITERS : -1.f;
}

// buf: [0..255] xcc table (+1, 0 = not yet), [256] arrival counter, [512 + 64*k] flags, payload after
__global__ __launch_bounds__(64) void probe_kernel(unsigned* buf, float* out)
{
    extern __shared__ float pad[];   // forces one workgroup per CU
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ int partner_same, partner_cross;
    if (lane == 0) {
        pad[0] = 0.f;
        store_flag(buf + b, xcc_id() + 1);
        atomicAdd(buf + 256, 1u);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(buf + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < PROBE_WGS &&
               wall_clock64() - t0 < PROBE_TIMEOUT) {}
        const unsigned x0 = __hip_atomic_load(buf + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ps = -1, pc = -1;
        for (int i = 1; i < PROBE_WGS; ++i) {
            const unsigned xi = __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (xi == 0) continue;
            if (xi == x0 && ps < 0) ps = i;
            if (xi != x0 && pc < 0) pc = i;
        }
        partner_same = ps;
        partner_cross = pc;
    }
    __syncthreads();
    const int ps = partner_same, pc = partner_cross;
    if (b != 0 && b != ps && b != pc) return;
    if (b == 0 && lane == 0) {
        out[0] = (float)(ps >= 0 ? ps : -1);
        out[1] = (float)(pc >= 0 ? pc : -1);
    }
    unsigned* flags = buf + 512;
    u32x4* payload = reinterpret_cast<u32x4*>(buf + 4096);
    // series k: pair (same / cross) x polling mode 0..3; then the tagged payload per pair
    for (int pair = 0; pair < 2; ++pair) {
        const int partner = pair == 0 ? ps : pc;
        if (partner < 0) continue;
        const bool in = (b == 0 || b == partner);
        if (!in) continue;
        const bool is_a = (b == 0);
        for (int mode = 0; mode < 3; ++mode) {
            const int k = pair * 3 + mode;
            float us = 0.f;
            if (lane == 0) {
                unsigned* fa = flags + 64 * (2 * k), * fb = flags + 64 * (2 * k + 1);
                us = mode == 0 ? pingpong<0>(is_a, fa, fb) : mode == 1 ? pingpong<1>(is_a, fa, fb)
                                                                          : pingpong<2>(is_a, fa, fb);
                if (is_a) out[2 + k] = us;
            }
            __syncthreads();
        }
        const float us = tagged_pingpong(is_a, payload + (2 * pair) * 64, payload + (2 * pair + 1) * 64, lane);
        if (is_a && lane == 0) out[8 + pair] = us;
        __syncthreads();
    }
}

}  // namespace sctc

extern "C" int sctc_probe_fabric(float* results_host, int32_t n_results, void* stream)
{
    using namespace sctc;
    SCTC_CHECK_ARG(results_host && n_results >= 10, "probe_fabric: need room for 10 floats");
    hipStream_t s = (hipStream_t)stream;
    unsigned* buf = nullptr;
    float* out = nullptr;
    const size_t buf_bytes = (4096 + 4 * 64 * 4) * sizeof(unsigned);
    SCTC_HIP_TRY(hipMalloc(&buf, buf_bytes));
    SCTC_HIP_TRY(hipMalloc(&out, 16 * sizeof(float)));
    SCTC_HIP_TRY(hipMemsetAsync(buf, 0, buf_bytes, s));
    SCTC_HIP_TRY(hipMemsetAsync(out, 0, 16 * sizeof(float), s));
    const size_t smem = 84 * 1024;
    SCTC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(probe_kernel, dim3(PROBE_WGS), dim3(64), smem, s, buf, out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(results_host, out, 10 * sizeof(float), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(buf);
    (void)hipFree(out);
    if (e != hipSuccess) return set_error(SCTC_ERR_HIP, "probe_fabric: %s", hipGetErrorString(e));
    return SCTC_OK;
}

// ---- sustained matrix-pipe rate: register-only MFMA loops (no memory traffic), every SIMD of
// every CU busy.  The fp32 MFMA peak of 157.3 TFLOP/s assumes 2.4 GHz; this measures what the
// part sustains under that load (power management lowers the shader clock).
namespace sctc {

typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, bool RANDOM>
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* sink, int iters)
{
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    unsigned rng = 0x9e3779b9u * (threadIdx.x + 1 + blockIdx.x * 256);
    if (SHAPE == 32) {
        pf32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (RANDOM) {   // fresh random mantissas: realistic toggling in the multiplier arrays
                    rng = rng * 1664525u + 1013904223u;
                    a = __uint_as_float(0x3f000000u | (rng >> 9));
                    b = __uint_as_float(0x3f000000u | ((rng * 2654435761u) >> 9));
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        if (s == 123.456f) sink[0] = s;
    } else {
        pf32x4 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (RANDOM) {
                    rng = rng * 1664525u + 1013904223u;
                    a = __uint_as_float(0x3f000000u | (rng >> 9));
                    b = __uint_as_float(0x3f000000u | ((rng * 2654435761u) >> 9));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 123.456f) sink[0] = s;
    }
}

}  // namespace sctc

extern "C" int sctc_probe_mfma(float* results_host, int32_t n_results, void* stream)
{
    using namespace sctc;
    SCTC_CHECK_ARG(results_host && n_results >= 8, "probe_mfma: need room for 8 floats");
    hipStream_t s = (hipStream_t)stream;
    float* sink = nullptr;
    SCTC_HIP_TRY(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    SCTC_HIP_TRY(hipEventCreate(&e0));
    SCTC_HIP_TRY(hipEventCreate(&e1));
    const int blocks = 256 * 3, iters = 4000;   // 3 blocks x 4 waves per CU = 3 waves per SIMD
    for (int shape = 0; shape < 4; ++shape) {   // 0,1: constant operands; 2,3: random operands
        for (int rep = 0; rep < 2; ++rep) {     // rep 0 warms up / lets the clock settle
            (void)hipEventRecord(e0, s);
            if (shape == 0) hipLaunchKernelGGL((mfma_rate_kernel<32, false>), dim3(blocks), dim3(256), 0, s, sink, iters);
            else if (shape == 1) hipLaunchKernelGGL((mfma_rate_kernel<16, false>), dim3(blocks), dim3(256), 0, s, sink, iters);
            else if (shape == 2) hipLaunchKernelGGL((mfma_rate_kernel<32, true>), dim3(blocks), dim3(256), 0, s, sink, iters);
            else hipLaunchKernelGGL((mfma_rate_kernel<16, true>), dim3(blocks), dim3(256), 0, s, sink, iters);
            (void)hipEventRecord(e1, s);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            // flops: blocks * 4 waves * iters * 32 MFMAs * (2*32*32*2) [32x32x2]  or 64 MFMAs * (2*16*16*4)
            const double fl = (double)blocks * 4 * iters * ((shape & 1) == 0 ? 32.0 * 4096.0 : 64.0 * 2048.0);
            results_host[shape * 2 + 0] = (float)(fl / (ms * 1e-3) / 1e12);
            results_host[shape * 2 + 1] = ms;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(SCTC_ERR_HIP, "probe_mfma: %s", hipGetErrorString(e));
    return SCTC_OK;
}

// ---------------------------------------------------------------------------------------
// sctc_diag_spin: n_wgs workgroups of 256 threads that do nothing but hold their compute units
// for `microseconds` -- the stand-in for a collective kernel (RCCL: one workgroup per channel)
// running on a side stream next to the backward pass (tests/test_gpu_shared.py).
namespace sctc {
__global__ __launch_bounds__(256) void spin_kernel(unsigned long long ticks, unsigned* sink)
{
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); ++n; }
    if (sink && n == 0xffffffffu) *sink = n;
}
}  // namespace sctc

extern "C" int sctc_diag_spin(void* stream, int32_t n_wgs, int32_t microseconds)
{
    using namespace sctc;
    SCTC_CHECK_ARG(n_wgs >= 1 && n_wgs <= 4096 && microseconds >= 0 && microseconds <= 1000000,
                   "diag_spin: bad argument");
    hipLaunchKernelGGL(spin_kernel, dim3(n_wgs), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long)microseconds * 100ull, (unsigned*)nullptr);
    SCTC_HIP_TRY(hipGetLastError());
    return SCTC_OK;
}
