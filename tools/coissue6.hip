// How many VALU instructions hide in the shadow of an MFMA, in shader CYCLES (s_memtime; wall time moves with the DVFS state)?
//   SHAPE 0: v_mfma_f32_16x16x32_f16 (16 pipe cycles)     SHAPE 1: v_mfma_f32_32x32x16_f16 (32 pipe cycles, the same flops per cycle)
// Program order per wave (asm volatile): MFMA, R x VALU, MFMA, R x VALU, ... over independent accumulators / sixteen VALU chains;
// W waves per SIMD (workgroups of 256 W threads, one per CU).  Printed: cycles per MFMA for R = 0 .. RMAX.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void valu1(float2_t (&v)[16], int i, float k) {
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 15].x) : "v"(k));
    if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 15]) : "v"(v[(i + 7) & 15]));
    if (KIND == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i & 15].x) : "v"(k));
    if (KIND == 3) { float4_t t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((int)(threadIdx.x & 63) * 16)); asm volatile("" :: "v"(t)); }  // (never waited for: issue only)
    if (KIND == 4) asm volatile("s_mov_b32 s20, 1" ::: "s20");
    if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(v[(i + 7) & 15]));
    if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(v[(i + 7) & 15]));
}
template <int SHAPE, int KIND, int R, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(float* out, long long* cyc, int iters) {
    half8_t a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    float2_t v[16];
    for (int i = 0; i < 16; i++) v[i] = (float2_t){threadIdx.x * 0.001f + i, 1.0f};
    const float k = out[0];
    float s = 0;
    long long t0 = 0, t1 = 0;
    __shared__ unsigned long long s_t0, s_t1;
    __shared__ float s_pad[1024]; if (threadIdx.x < 1024) s_pad[threadIdx.x] = 0.f;  // the workgroup's span: first wave in, last wave out (the arbiter favours the oldest wave)
    if (threadIdx.x == 0) { s_t0 = ~0ull; s_t1 = 0ull; }
    __syncthreads();
    if (SHAPE == 0) {
        float4_t c[8];
        for (int i = 0; i < 8; i++) c[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
        t0 = clock64();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int r = 0; r < R; r++) valu1<KIND>(v, i * R + r, k);
            }
        }
        t1 = clock64();
        for (int i = 0; i < 8; i++) s += c[i].x + c[i].y;
    } else {
        float16_t c[4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) c[i][j] = 0.f;
        t0 = clock64();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[i & 3]) : "v"(a), "v"(b));
#pragma unroll
                for (int r = 0; r < R; r++) valu1<KIND>(v, i * R + r, k);
            }
        }
        t1 = clock64();
        for (int i = 0; i < 4; i++) s += c[i][0] + c[i][5];
    }
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y;
    out[1 + blockIdx.x * THREADS + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { atomicMin(&s_t0, (unsigned long long)t0); atomicMax(&s_t1, (unsigned long long)t1); }
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = (long long)(s_t1 - s_t0);
}
template <int SHAPE, int KIND, int R, int W>
double run(float* d, long long* c) {
    const int iters = 20000;
    probe<SHAPE, KIND, R, 256 * W><<<256, 256 * W>>>(d, c, 100);
    probe<SHAPE, KIND, R, 256 * W><<<256, 256 * W>>>(d, c, iters);
    (void)hipDeviceSynchronize();
    long long h[256]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 256; i++) sum += (double)h[i];
    return sum / 256.0 / (8.0 * iters * W);  // cycles of the SIMD per MFMA (W waves share it)
}
template <int SHAPE, int KIND, int W>
void row(float* d, long long* c, const char* name) {
    printf("%s %s W=%d: cycles of a SIMD per MFMA with R VALU behind each, R = 0..6:  %.1f  %.1f  %.1f  %.1f  %.1f  %.1f  %.1f\n", SHAPE ? "32x32x16" : "16x16x32", name, W,
           run<SHAPE, KIND, 0, W>(d, c), run<SHAPE, KIND, 1, W>(d, c), run<SHAPE, KIND, 2, W>(d, c), run<SHAPE, KIND, 3, W>(d, c),
           run<SHAPE, KIND, 4, W>(d, c), run<SHAPE, KIND, 5, W>(d, c), run<SHAPE, KIND, 6, W>(d, c));
}
template <int KIND>
void kind(float* d, long long* c, const char* name) {
    row<0, KIND, 1>(d, c, name); row<1, KIND, 1>(d, c, name);
    row<0, KIND, 2>(d, c, name); row<1, KIND, 2>(d, c, name);
    row<0, KIND, 4>(d, c, name); row<1, KIND, 4>(d, c, name);
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 24); (void)hipMemset(d, 0, 1 << 24);
    long long* c; (void)hipMalloc(&c, 256 * sizeof(long long));
    kind<0>(d, c, "v_fma_f32     ");
    kind<1>(d, c, "v_pk_fma_f32  ");
    kind<2>(d, c, "v_cvt_pkrtz   ");
    kind<3>(d, c, "ds_read_b128  ");
    kind<4>(d, c, "s_mov_b32     ");
    kind<5>(d, c, "v_pk_add_f32  ");
    kind<6>(d, c, "v_pk_mul_f32  ");
    return 0;
}
