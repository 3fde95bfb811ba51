// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 issue rate on gfx950 for the accumulator patterns
// the GIN layer kernel uses, with constant vs random operands (DVFS: operand toggling costs clock).
// Build on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/mfma_f32_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NACC>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* out, int iters) {
    f4 acc[NACC];
    float a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        a[r] = in[(r * 256 + threadIdx.x)];
        b[r] = in[((r + 8) * 256 + threadIdx.x)];
    }
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = MFMA(a[r], b[(r + i) & 7], acc[i]);
    }
    f4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; i++) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int NACC>
void run(int blocks_per_cu, const float* din, float* d, const char* tag) {
    const int iters = 4000;
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(din, d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(din, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)blocks * 4 * iters * 8 * NACC;
    double tf = mfmas * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-8s nacc=%d waves/SIMD=%d  %.3f ms  %.1f TF\n", tag, NACC, blocks_per_cu, ms, tf);
}

int main() {
    float *d, *dz, *dr;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    std::vector<float> z(16 * 256, 1.0f), r(16 * 256);
    srand(1);
    for (auto& x : r) x = ((rand() % 20001) - 10000) * 1e-4f;
    (void)hipMalloc(&dz, z.size() * 4); (void)hipMalloc(&dr, r.size() * 4);
    (void)hipMemcpy(dz, z.data(), z.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dr, r.data(), r.size() * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 3; w++) {
        run<2>(w, dz, d, "const"); run<2>(w, dr, d, "random");
        run<7>(w, dz, d, "const"); run<7>(w, dr, d, "random");
    }
    return 0;
}
