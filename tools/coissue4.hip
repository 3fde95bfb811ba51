// Do MFMA issue and VALU issue of one gfx950 SIMD add up or overlap when FOUR waves share it (the fused PNA / DGN layers' regime)?
// Per SIMD the same total work in every mode -- 4 x M MFMAs (v_mfma_f32_16x16x32_f16) and 4 x V VALU instructions:
//   mode 1: four waves, MFMAs only          mode 2: four waves, VALU only
//   mode 3: two waves with 2 M MFMAs each + two waves with 2 V VALU each (different waves feed the two pipes)
//   mode 4: every wave alternates bursts of 8 MFMAs and 8 V / M VALU (phases inside a wave)
// VALU kinds: 0 = v_fma_f32 (independent chains), 1 = v_pk_fma_f32, 2 = v_min3_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void valu32(float2_t (&v)[16], float k) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 15].x) : "v"(k));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 15]) : "v"(v[(i + 7) & 15]));
        if (KIND == 2) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(v[i & 15].x) : "v"(k));
    }
}
template <int MODE, int KIND>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
    const int wave = threadIdx.x >> 6;  // 16 waves: waves w, w + 4, w + 8, w + 12 share a SIMD
    float4_t c[8];
    for (int i = 0; i < 8; i++) c[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    half8_t a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    float2_t v[16];
    for (int i = 0; i < 16; i++) v[i] = (float2_t){threadIdx.x * 0.001f + i, 1.0f};
    const float k = out[0];
    // per wave and iteration: 8 MFMAs (128 pipe clocks) and 32 VALU
    if (MODE == 1) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
        }
    } else if (MODE == 2) {
        for (int it = 0; it < iters; it++) valu32<KIND>(v, k);
    } else if (MODE == 3) {
        if ((wave >> 2) & 1) {
            for (int it = 0; it < 2 * iters; it++) valu32<KIND>(v, k);
        } else {
            for (int it = 0; it < 2 * iters; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
            valu32<KIND>(v, k);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += c[i].x + c[i].y;
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y;
    out[1 + blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int MODE, int KIND>
float run(float* d, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE, KIND><<<256, 1024>>>(d, 100);
    (void)hipEventRecord(e0); probe<MODE, KIND><<<256, 1024>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int KIND>
void report(float* d, const char* name) {
    const int iters = 100000;
    const float m = run<1, KIND>(d, iters), v = run<2, KIND>(d, iters), split = run<3, KIND>(d, iters), phased = run<4, KIND>(d, iters);
    const double clk = 2.4e6;  // clocks per ms at 2.4 GHz
    printf("%s: MFMA only %.2f ms (%.1f clk per MFMA and SIMD) | VALU only %.2f ms (%.2f clk per VALU and SIMD) | two waves each: %.2f ms | "
           "phases inside every wave: %.2f ms | sum %.2f, max %.2f\n", name, m, m * clk / (32.0 * iters), v, v * clk / (128.0 * iters), split, phased,
           m + v, m > v ? m : v);
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 24); (void)hipMemset(d, 0, 1 << 24);
    report<0>(d, "v_fma_f32   ");
    report<1>(d, "v_pk_fma_f32");
    report<2>(d, "v_min3_f32  ");
    return 0;
}
