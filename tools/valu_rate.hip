// Diagnostics: the sustained rate of v_mfma_f32_32x32x16_bf16 on constant and on random operands
// (register-only loops, 1..3 waves per SIMD: the clock follows the power draw), and the issue cost of the
// VALU instructions the three-term bfloat16 split (gemm_s3.hip) is made of, and the dependent-chain
// latency of the float64 operations of the CTC lattice recursion (add / mul / fma, IEEE division, one
// DPP reduction stage) with the shader clock they ran at.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, uint64_t* cyc, int iters)
{
    float a = threadIdx.x * 1.0001f + 1.f, b = a * 1.3f, c = a * 0.7f, d = b * 1.1f;
    f32x2 x = {a, b}, y = {c, d}, z = {a + 3.f, b + 5.f}, w = {c + 7.f, d + 11.f};
    uint32_t u0 = __float_as_uint(a), u1 = __float_as_uint(b), u2 = __float_as_uint(c), u3 = __float_as_uint(d);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (MODE == 0) {          // 4 independent v_cvt_pk_bf16_f32
                b16x2 h0 = __builtin_convertvector(x, b16x2), h1 = __builtin_convertvector(y, b16x2);
                b16x2 h2 = __builtin_convertvector(z, b16x2), h3 = __builtin_convertvector(w, b16x2);
                x[0] += __uint_as_float(__builtin_bit_cast(uint32_t, h0) << 16) * 0.f;
                y[0] += __uint_as_float(__builtin_bit_cast(uint32_t, h1) << 16) * 0.f;
                z[0] += __uint_as_float(__builtin_bit_cast(uint32_t, h2) << 16) * 0.f;
                w[0] += __uint_as_float(__builtin_bit_cast(uint32_t, h3) << 16) * 0.f;
            } else if constexpr (MODE == 1) {   // 4 independent v_pk_add_f32
                x = x - y; y = y - z; z = z - w; w = w - x;
            } else if constexpr (MODE == 2) {   // 4 independent v_and_b32 + 4 v_add_u32
                u0 = (u0 + 0x8000u) & 0xffff0000u; u1 = (u1 + 0x8000u) & 0xffff0000u;
                u2 = (u2 + 0x8000u) & 0xffff0000u; u3 = (u3 + 0x8000u) & 0xffff0000u;
                u0 ^= u1; u2 ^= u3;
            } else if constexpr (MODE == 3) {   // 4 v_perm_b32
                u0 = __builtin_amdgcn_perm(u0, u1, 0x07060302u); u1 = __builtin_amdgcn_perm(u1, u2, 0x07060302u);
                u2 = __builtin_amdgcn_perm(u2, u3, 0x07060302u); u3 = __builtin_amdgcn_perm(u3, u0, 0x07060302u);
            } else if constexpr (MODE == 4) {   // 4 independent v_sub_f32
                a = a - b; b = b - c; c = c - d; d = d - a;
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + x[0] + x[1] + y[0] + y[1] + z[0] + w[0] +
                                         __uint_as_float(u0 ^ u1 ^ u2 ^ u3);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int ops_per_rep)
{
    float* out; uint64_t* cyc;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
    s /= 1024;
    // 1024 waves on 1024 SIMDs, one each: kernel time / instructions of one wave
    printf("%-34s %7.2f ns per wave-instruction (kernel time) | %8.2f counter ticks (%d instr per rep)\n", name,
           ms * 1e6 / ((double)iters * 16 * ops_per_rep), s / ((double)iters * 16 * ops_per_rep), ops_per_rep);
    hipFree(out); hipFree(cyc);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

// sustained rate of v_mfma_f32_32x32x16_bf16: register-only loops, operands with random bit patterns
// (data toggling sets the power draw and with it the clock), W waves per SIMD
__global__ __launch_bounds__(256) void mfma_bf16_kernel(float* out, int iters, int random)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    b16x8 a[2], b[2];
    for (int t = 0; t < 2; ++t)
        for (int e = 0; e < 8; ++e) {
            seed = seed * 1664525u + 1013904223u;
            const float fa = random ? (float)((int)(seed >> 9) - (1 << 22)) * 1e-6f : 1.0f;
            seed = seed * 1664525u + 1013904223u;
            const float fb = random ? (float)((int)(seed >> 9) - (1 << 22)) * 1e-6f : 1.0f;
            a[t][e] = (__bf16)fa; b[t][e] = (__bf16)fb;
        }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[3], 0, 0, 0);
        }
    }
    float v = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) v += acc[t][r];
    if (v == 123.456f) out[0] = v;
}

static void run_mfma(int waves_per_simd, int random)
{
    float* out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd, iters = 20000;
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {       // the last repetition counts: the clock has settled
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_bf16_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, random);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double fl = (double)blocks * 4 * iters * 32.0 * (2.0 * 32 * 32 * 16);
    printf("v_mfma_f32_32x32x16_bf16, %d wave(s)/SIMD, %s operands: %.0f TFLOP/s (%.1f ms)\n", waves_per_simd,
           random ? "random" : "constant", fl / (ms * 1e-3) / 1e12, ms);
    hipFree(out);
}

// dependent-chain latency (shader cycles per instruction, one wave) of the float64 operations the CTC
// lattice recursion is made of
template <int MODE>
__global__ __launch_bounds__(64) void lat_kernel(double* out, uint64_t* cyc, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0000001, c = 0.5;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (MODE == 0) a = a + b;
            else if constexpr (MODE == 1) a = a * b;
            else if constexpr (MODE == 2) a = __builtin_fma(a, b, c);
            else if constexpr (MODE == 3) a = 1.0 / a + b;
            else if constexpr (MODE == 4) {        // one DPP reduction stage: row_shr:1 of both halves + add
                const int lo = __builtin_amdgcn_update_dpp(0, (int)__double2loint(a), 0x111, 0xf, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(0, (int)__double2hiint(a), 0x111, 0xf, 0xf, false);
                a = a + __hiloint2double(hi, lo) * 1e-9;
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int MODE>
static void run_lat(const char* name, int ops)
{
    double* out; uint64_t* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16);
    const int iters = 200000;
    hipLaunchKernelGGL(lat_kernel<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(lat_kernel<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    uint64_t h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-44s %7.1f s_memtime ticks = %6.1f ns per dependent step (%d instr per step); ticks run at %.0f MHz\n", name,
           (double)h[0] / ((double)iters * 16), (double)h[1] * 10.0 / ((double)iters * 16), ops, 100.0 * (double)h[0] / (double)h[1]);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run_lat<0>("v_add_f64 chain", 1); run_lat<1>("v_mul_f64 chain", 1); run_lat<2>("v_fma_f64 chain", 1);
    run_lat<3>("1.0 / x + b (IEEE division) chain", 12); run_lat<4>("DPP row_shr + mul + add chain (f64)", 4);

    run_mfma(1, 0); run_mfma(1, 1); run_mfma(2, 1); run_mfma(3, 1);
    run<0>("v_cvt_pk_bf16_f32 (+shift,+fma)", 12);
    run<1>("v_pk_add_f32", 4);
    run<2>("v_add_u32 + v_and_b32 (+xor)", 10);
    run<3>("v_perm_b32", 4);
    run<4>("v_sub_f32", 4);
    return 0;
}
