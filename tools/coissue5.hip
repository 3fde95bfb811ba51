// Follow-up of coissue4.hip: does FINE-GRAINED interleaving inside ONE wave hide VALU issue behind the matrix pipe?
// Per wave and iteration 8 MFMAs (v_mfma_f32_16x16x32_f16, eight independent accumulators) and 8 x R VALU instructions (sixteen
// independent chains), in program order "MFMA, R x VALU, MFMA, R x VALU, ..." (asm volatile: hipcc keeps the order).
//   W = waves per SIMD (1, 2, 4): workgroups of 256 W threads, one per CU
// Reported: MFMA only, VALU only, interleaved -- against the sum and the max of the two.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void valu1(float2_t (&v)[16], int i, float k) {
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 15].x) : "v"(k));
    if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 15]) : "v"(v[(i + 7) & 15]));
    if (KIND == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i & 15].x) : "v"(k));
    if (KIND == 3) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[i & 15].x) : "v"(k));
}
// MODE 1: MFMAs only, 2: VALU only, 3: interleaved
template <int MODE, int KIND, int R, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(float* out, int iters) {
    float4_t c[8];
    for (int i = 0; i < 8; i++) c[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    half8_t a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    float2_t v[16];
    for (int i = 0; i < 16; i++) v[i] = (float2_t){threadIdx.x * 0.001f + i, 1.0f};
    const float k = out[0];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE != 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
            if (MODE != 1) {
#pragma unroll
                for (int r = 0; r < R; r++) valu1<KIND>(v, i * R + r, k);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += c[i].x + c[i].y;
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y;
    out[1 + blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int MODE, int KIND, int R, int THREADS>
float run(float* d, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE, KIND, R, THREADS><<<256, THREADS>>>(d, 100);
    (void)hipEventRecord(e0); probe<MODE, KIND, R, THREADS><<<256, THREADS>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int KIND, int R, int W>
void report(float* d, const char* name) {
    const int iters = 200000 / W;
    const float m = run<1, KIND, R, 256 * W>(d, iters), v = run<2, KIND, R, 256 * W>(d, iters), x = run<3, KIND, R, 256 * W>(d, iters);
    const double clk = 2.4e6;
    printf("%s R=%d W=%d: MFMA only %.2f ms (%.1f clk per MFMA and SIMD) | VALU only %.2f ms (%.2f clk each) | interleaved %.2f ms = %.2f of the sum, %.2f of the max\n",
           name, R, W, m, m * clk / (8.0 * W * iters), v, v * clk / (8.0 * R * W * iters), x, x / (m + v), x / (m > v ? m : v));
}
template <int KIND>
void kinds(float* d, const char* name) {
    report<KIND, 2, 1>(d, name); report<KIND, 4, 1>(d, name); report<KIND, 2, 2>(d, name); report<KIND, 4, 2>(d, name);
    report<KIND, 2, 4>(d, name); report<KIND, 4, 4>(d, name);
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 24); (void)hipMemset(d, 0, 1 << 24);
    kinds<0>(d, "v_fma_f32      ");
    kinds<1>(d, "v_pk_fma_f32   ");
    kinds<2>(d, "v_cvt_pkrtz    ");
    kinds<3>(d, "v_max_i32      ");
    return 0;
}
