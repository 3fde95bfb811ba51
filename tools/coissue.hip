// Do the matrix pipe and the VALU of a gfx950 SIMD run concurrently when DIFFERENT waves feed them?  Two waves per SIMD: one issues
// a stream of independent v_mfma_f32_16x16x32_f16, the other a stream of independent v_fma_f32; timed alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
template <int MODE>  // 1: MFMA waves only, 2: VALU waves only, 3: both, 4: both streams interleaved inside EVERY wave
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float4_t c[8];
    for (int i = 0; i < 8; i++) c[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    half8_t a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i;
    const float k = out[0];
    const bool mf = MODE == 4 || wave < 4, va = MODE == 4 || wave >= 4;
    if constexpr (MODE == 4) {
        for (int it = 0; it < iters / 2; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(4 * i + q) & 15]) : "v"(k));
            }
        }
    } else {
        if (mf && (MODE & 1)) {
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
            }
        }
        if (va && (MODE & 2)) {
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int i = 0; i < 32; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 15]) : "v"(k));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += c[i].x + c[i].y;
    for (int i = 0; i < 16; i++) s += v[i];
    out[1 + blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
float run(float* d, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<256, 512>>>(d, 100);
    (void)hipEventRecord(e0); probe<MODE><<<256, 512>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 24); (void)hipMemset(d, 0, 1 << 24);
    const int iters = 200000;
    const float m = run<1>(d, iters), v = run<2>(d, iters), both = run<3>(d, iters), inter = run<4>(d, iters);
    printf("one wave per SIMD issuing 8 MFMA per iteration: %.2f ms (%.1f clk per MFMA @2.4GHz)\n", m, m * 1e-3 * 2.4e9 / (8.0 * iters));
    printf("one wave per SIMD issuing 32 v_fma per iteration: %.2f ms (%.1f clk per VALU @2.4GHz)\n", v, v * 1e-3 * 2.4e9 / (32.0 * iters));
    printf("both waves together: %.2f ms  (sum %.2f, max %.2f)\n", both, m + v, m > v ? m : v);
    printf("both streams interleaved in each of two waves per SIMD (same total work): %.2f ms\n", inter);
    return 0;
}
