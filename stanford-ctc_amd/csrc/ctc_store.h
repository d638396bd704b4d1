// Row store of the meet-in-the-middle CTC kernels (ctc_fused.hip, ctc_fusedw.hip): how a normalised lattice row is
// kept between the recursion that writes it and the one that multiplies it (ctc_fused.hip's header has the why).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sctc {

template <typename ST>
struct Store;
template <>
struct Store<double> {
    static __device__ __forceinline__ double enc(double x) { return x; }
    static __device__ __forceinline__ double dec(double s) { return s; }
};
// bits 61..30 of the float64 pattern, rounded to nearest: a normalised row lies in [0, 1], sign and top exponent bit
// are always zero -- 10 exponent bits (the full float64 range below 2.0) and 22 mantissa bits in 32 bits
template <>
struct Store<uint32_t> {
    static __device__ __forceinline__ uint32_t enc(double x)
    {
        return (uint32_t)(((uint64_t)__double_as_longlong(x) + (1ull << 29)) >> 30);
    }
    static __device__ __forceinline__ double dec(uint32_t s)
    {
        return __longlong_as_double((long long)((uint64_t)s << 30));
    }
};

// K stored states as one memory block; the reader's block is K-element contiguous but only
// element-aligned (it mirrors the writer's lane order), so the type promises no more than that
template <typename ST, int K>
struct __attribute__((packed, aligned(sizeof(ST)))) RowBlockU {
    ST v[K];
};
template <typename ST, int K>
struct __attribute__((aligned(sizeof(ST) * K))) RowBlockA {
    ST v[K];
};

}  // namespace sctc
