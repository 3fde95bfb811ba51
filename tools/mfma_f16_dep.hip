#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
template <int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rate(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f / (1 + i)); }
    float4_t c[NACC];
    for (int i = 0; i < NACC; i++) c[i] = (float4_t){0, 0, 0, 0};
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int r = 0; r < 8 / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; i++) s += c[i].x + c[i].y + c[i].z + c[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int WAVES>
void run(float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    rate<NACC, WAVES><<<blocks, WAVES * 64>>>(d, 100);
    hipEventRecord(e0); rate<NACC, WAVES><<<blocks, WAVES * 64>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * WAVES * iters * 8 * 16 * 16 * 32 * 2;
    printf("acc=%d waves/CU=%d (per SIMD %d): %.1f TFLOP/s, %.1f clk/MFMA/SIMD @2.4GHz\n", NACC, WAVES, WAVES / 4, flops / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / ((double)iters * 8 * (WAVES / 4)));
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    run<1, 4>(d); run<2, 4>(d); run<4, 4>(d); run<8, 4>(d);
    run<1, 8>(d); run<2, 8>(d); run<4, 8>(d);
    run<1, 12>(d); run<2, 12>(d);
    return 0;
}
