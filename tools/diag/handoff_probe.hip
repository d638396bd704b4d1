// sctc_probe_handoff(): does "write-through payload stores -> s_waitcnt vmcnt(0) -> flag store" make the
// payload visible to PLAIN (cacheable) loads that another workgroup issues the moment it has seen
// the flag?  This is the hand-off of the flag-based recurrent kernels (csrc/recurrent.hip); they
// normally fetch the payload >= 1 us after all but the last producer's flag, and a round-3
// experiment that consumed every producer's block within nanoseconds of its flag produced wrong
// results.  One producer workgroup, 15 consumers (block b runs on XCD b % 8); every iteration uses
// a FRESH 1 KiB block (written once, never cached before), the producer waits for all consumers'
// acknowledgements so that they are always polling when the flag is stored.
//   mode 0: producer drains (vmcnt(0)) before the flag, consumers use plain loads   <- the shipped protocol
//   mode 1: producer does NOT drain                                                  <- must show errors
//   mode 2: producer drains, consumers use L2-bypassing (sc1) loads
// results_host[3 * 3]: per mode {payload dwords that were stale, iterations, microseconds per iteration}
#include "diag_common.h"

namespace sctc {

typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
static constexpr int HO_WGS = 16;
static constexpr int HO_ITERS = 20000;
static constexpr unsigned long long HO_TIMEOUT = 100000000ull;   // 1 s

__device__ __forceinline__ unsigned ho_poll(const unsigned* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(64) void handoff_kernel(unsigned* data, unsigned* flags, unsigned* out)
{
    // flags: [0] = producer's iteration counter, [32 * (1 + c)] = consumer c's acknowledgement
    const int lane = threadIdx.x, wg = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        data, 0, (int)((size_t)HO_ITERS * 256 * sizeof(unsigned)), 0x00020000);
    unsigned stale = 0;
    const unsigned long long t_begin = wall_clock64();
    bool ok = true;
    for (unsigned it = 1; it <= HO_ITERS && ok; ++it) {
        const unsigned base = (it - 1) * 1024u;             // byte offset of this iteration's block
        if (wg == 0) {
            const u32x4h v = {it * 4u + 0u + lane * 65536u, it * 4u + 1u + lane * 65536u,
                              it * 4u + 2u + lane * 65536u, it * 4u + 3u + lane * 65536u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (unsigned)lane * 16u, base, 16 /* sc1 */);
            if (MODE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(flags, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // lock step: every consumer has checked this block
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                unsigned a = it;
                if (lane >= 1 && lane < HO_WGS) a = ho_poll(flags + 32 * lane);
                if (__all(a >= it)) break;
                if (wall_clock64() - t0 > HO_TIMEOUT) { ok = false; break; }
            }
        } else {
            const unsigned long long t0 = wall_clock64();
            while (ho_poll(flags) < it) {
                if (wall_clock64() - t0 > HO_TIMEOUT) { ok = false; break; }
            }
            // the payload, the moment the flag has been seen
            u32x4h v;
            if (MODE == 2) v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)lane * 16u, base, 16 /* sc1 */);
            else v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)lane * 16u, base, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) stale += v[e] != it * 4u + (unsigned)e + lane * 65536u ? 1u : 0u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(flags + 32 * wg, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long t_end = wall_clock64();
    for (int off = 32; off > 0; off >>= 1) stale += __shfl_xor(stale, off, 64);
    if (lane == 0) {
        if (wg != 0) atomicAdd(out + 0, stale);
        if (!ok) atomicAdd(out + 1, 1u);
        if (wg == 0) out[2] = (unsigned)(t_end - t_begin);
    }
}

}  // namespace sctc

extern "C" int sctc_probe_handoff(float* results_host, int32_t n_results, void* stream)
{
    using namespace sctc;
    SCTC_CHECK_ARG(results_host && n_results >= 9, "probe_handoff: need room for 9 floats");
    hipStream_t s = (hipStream_t)stream;
    unsigned *data = nullptr, *flags = nullptr, *out = nullptr;
    const size_t dbytes = (size_t)HO_ITERS * 256 * sizeof(unsigned);
    SCTC_HIP_TRY(hipMalloc(&data, dbytes));
    SCTC_HIP_TRY(hipMalloc(&flags, 32 * (HO_WGS + 1) * sizeof(unsigned)));
    SCTC_HIP_TRY(hipMalloc(&out, 16 * sizeof(unsigned)));
    hipError_t e = hipSuccess;
    for (int mode = 0; mode < 3 && e == hipSuccess; ++mode) {
        (void)hipMemsetAsync(data, 0xFF, dbytes, s);
        (void)hipMemsetAsync(flags, 0, 32 * (HO_WGS + 1) * sizeof(unsigned), s);
        (void)hipMemsetAsync(out, 0, 16 * sizeof(unsigned), s);
        if (mode == 0) hipLaunchKernelGGL(handoff_kernel<0>, dim3(HO_WGS), dim3(64), 0, s, data, flags, out);
        else if (mode == 1) hipLaunchKernelGGL(handoff_kernel<1>, dim3(HO_WGS), dim3(64), 0, s, data, flags, out);
        else hipLaunchKernelGGL(handoff_kernel<2>, dim3(HO_WGS), dim3(64), 0, s, data, flags, out);
        unsigned h[3] = {0, 0, 0};
        e = hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        results_host[3 * mode + 0] = (float)h[0];
        results_host[3 * mode + 1] = h[1] ? -1.f : (float)HO_ITERS;
        results_host[3 * mode + 2] = (float)h[2] * 0.01f / HO_ITERS;
    }
    (void)hipFree(data);
    (void)hipFree(flags);
    (void)hipFree(out);
    if (e != hipSuccess) return set_error(SCTC_ERR_HIP, "probe_handoff: %s", hipGetErrorString(e));
    return SCTC_OK;
}
