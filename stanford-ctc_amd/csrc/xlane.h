// Cross-lane primitives for wave64 on gfx950 (CDNA4).
//
// The CTC recursion (ctc_fast.pyx:48-76) is a 3-point stencil along the 2|l|+1
// label axis followed by a sum over that axis, once per frame.  Both are done in
// registers with DPP (data-parallel-primitive) VALU modifiers: no LDS traffic,
// no ds_bpermute round trip.  SCTC_SAFE_XLANE swaps in __shfl-based versions
// (LDS crossbar, slower) -- the GPU selftest compares the two.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sctc {

// DPP control words (GFX9 ISA, DPP_CTRL field)
enum : int {
    DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR3 = 0x113, DPP_ROW_SHR4 = 0x114,
    DPP_ROW_SHR8 = 0x118, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142,
    DPP_ROW_BCAST31 = 0x143,
};

template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND_CTRL = true>
__device__ __forceinline__ int dpp_i32(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL);
}

// value of lane (i-1) delivered to lane i; lane 0 receives `fill` (0 here)
__device__ __forceinline__ float lane_shr1(float v)
{
#ifdef SCTC_SAFE_XLANE
    float r = __shfl_up(v, 1, 64);
    return (threadIdx.x & 63) == 0 ? 0.f : r;
#else
    return __int_as_float(dpp_i32<DPP_WAVE_SHR1>(0, __float_as_int(v)));
#endif
}

__device__ __forceinline__ double lane_shr1(double v)
{
#ifdef SCTC_SAFE_XLANE
    double r = __shfl_up(v, 1, 64);
    return (threadIdx.x & 63) == 0 ? 0.0 : r;
#else
    long long b = __double_as_longlong(v);
    int lo = dpp_i32<DPP_WAVE_SHR1>(0, (int)(b & 0xffffffffll));
    int hi = dpp_i32<DPP_WAVE_SHR1>(0, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
#endif
}

// Lanes that receive no source add 0.  With every row enabled (ROW_MASK 0xF) that is what bound_ctrl
// delivers (a shifted-in lane reads 0) and the `old` operand is dead -- no register pair has to be zeroed
// in front of the move (two v_mov + a hazard nop per stage; it matters where the wave is issue-bound: the
// 8-wave CTC lattice).  With rows masked off (the row_bcast stages) the disabled rows keep `old` = 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    constexpr bool BC = ROW_MASK == 0xF;
    return v + __int_as_float(dpp_i32<CTRL, ROW_MASK, 0xF, BC>(0, __float_as_int(v)));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    constexpr bool BC = ROW_MASK == 0xF;
    long long b = __double_as_longlong(v);
    int lo = dpp_i32<CTRL, ROW_MASK, 0xF, BC>(0, (int)(b & 0xffffffffll));
    int hi = dpp_i32<CTRL, ROW_MASK, 0xF, BC>(0, (int)(b >> 32));
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes of a wave, returned in every lane (wave-uniform).
// Fixed reduction tree => bit-reproducible run to run.
template <typename R>
__device__ __forceinline__ R wave_sum(R v)
{
#ifdef SCTC_SAFE_XLANE
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
#else
    // inclusive scan inside each row of 16 lanes ...
    v = dpp_add<DPP_ROW_SHR1, 0xF>(v);
    v = dpp_add<DPP_ROW_SHR2, 0xF>(v);
    v = dpp_add<DPP_ROW_SHR4, 0xF>(v);
    v = dpp_add<DPP_ROW_SHR8, 0xF>(v);
    // ... lane 15 of row r into row r+1 (rows 1 and 3), then lane 31 into rows 2,3
    v = dpp_add<DPP_ROW_BCAST15, 0xA>(v);
    v = dpp_add<DPP_ROW_BCAST31, 0xC>(v);
    // lane 63 now holds the wave total
    if constexpr (sizeof(R) == 4) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int((float)v), 63));
    } else {
        long long b = __double_as_longlong((double)v);
        int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
        int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
        return (R)__longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
#endif
}

template <typename R>
__device__ __forceinline__ R wave_max(R v)
{
    for (int off = 32; off > 0; off >>= 1) {
        R o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// broadcast the value held by lane `src` (wave-uniform index) to all lanes
__device__ __forceinline__ float lane_bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// arbitrary per-lane gather from another lane's register (ds_bpermute crossbar)
__device__ __forceinline__ float lane_gather(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ double lane_gather(double v, int src_lane)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b & 0xffffffffll));
    int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

}  // namespace sctc
