// MFMA issue-rate probe for gfx950: v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16, constant-like vs random operands,
// 1 / 2 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rate(const float* in, float* out, int iters) {
    half8 a[4], b[4];
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 8; i++) {
            a[k][i] = (_Float16)in[(threadIdx.x * 8 + i + 64 * k) & 4095];
            b[k][i] = (_Float16)in[(threadIdx.x * 8 + i + 1000 + 64 * k) & 4095];
        }
    float s = 0;
    if constexpr (SHAPE == 16) {
        float4_t c[NACC];
        for (int i = 0; i < NACC; i++) c[i] = (float4_t){0, 0, 0, 0};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int r = 0; r < 16 / NACC; r++)
#pragma unroll
                for (int i = 0; i < NACC; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(r + i) & 3], b[i & 3], c[i], 0, 0, 0);
        for (int i = 0; i < NACC; i++) s += c[i].x + c[i].y + c[i].z + c[i].w;
    } else {
        float16_t c[NACC];
        for (int i = 0; i < NACC; i++)
            for (int k = 0; k < 16; k++) c[i][k] = 0;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int r = 0; r < 16 / NACC; r++)
#pragma unroll
                for (int i = 0; i < NACC; i++) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(r + i) & 3], b[i & 3], c[i], 0, 0, 0);
        for (int i = 0; i < NACC; i++)
            for (int k = 0; k < 16; k++) s += c[i][k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE, int NACC, int WAVES>
void run(const float* din, float* d, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    rate<SHAPE, NACC, WAVES><<<blocks, WAVES * 64>>>(din, d, 100);
    hipEventRecord(e0); rate<SHAPE, NACC, WAVES><<<blocks, WAVES * 64>>>(din, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = SHAPE == 16 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2;
    const double flops = (double)blocks * WAVES * iters * 16 * per;
    printf("%s shape %dx%d acc=%d waves/SIMD=%d: %.0f TFLOP/s, %.1f clk/MFMA/SIMD @2.4GHz\n", what, SHAPE, SHAPE, NACC, WAVES / 4, flops / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (WAVES / 4)));
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    float *din, h[4096];
    hipMalloc(&din, sizeof(h));
    for (int mode = 0; mode < 2; mode++) {
        srand(1);
        for (int i = 0; i < 4096; i++) h[i] = mode ? (float)rand() / RAND_MAX * 4.0f - 2.0f : 1.0f;
        hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
        const char* w = mode ? "random" : "ones  ";
        run<16, 4, 4>(din, d, w); run<16, 8, 4>(din, d, w); run<16, 4, 8>(din, d, w); run<16, 8, 8>(din, d, w);
        run<32, 2, 4>(din, d, w); run<32, 4, 4>(din, d, w); run<32, 2, 8>(din, d, w); run<32, 4, 8>(din, d, w);
    }
    return 0;
}
