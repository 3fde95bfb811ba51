// VALU issue-rate probe for gfx950: plain v_add_f32 / v_max_f32 vs v_pk_add_f32, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rate(float* out, int iters) {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    const float b = out[0];
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float2_t v = {a[i], a[i + 1]};
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"((float2_t){b, b}));
                a[i] = v.x; a[i + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i];
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int WAVES>
void run(float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 100000, blocks = 256;
    rate<MODE, WAVES><<<blocks, WAVES * 64>>>(d, 100);
    (void)hipEventRecord(e0); rate<MODE, WAVES><<<blocks, WAVES * 64>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const int instr = MODE == 1 ? 8 : 16;
    printf("%s waves/SIMD=%d: %.2f clk per wave-instruction per SIMD @2.4GHz (%.1f T lane-ops/s)\n", MODE == 0 ? "v_add_f32   " : MODE == 1 ? "v_pk_add_f32" : "v_max_f32   ",
           WAVES / 4, ms * 1e-3 * 2.4e9 / ((double)iters * instr * (WAVES / 4)), (double)blocks * WAVES * 64 * iters * 16 / ms * 1e-9);
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 24); (void)hipMemset(d, 0, 1 << 24);
    run<0, 4>(d); run<0, 8>(d); run<0, 16>(d);
    run<1, 4>(d); run<1, 8>(d); run<1, 16>(d);
    run<2, 4>(d); run<2, 8>(d);
    return 0;
}
