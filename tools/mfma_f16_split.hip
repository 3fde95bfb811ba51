// Microbenchmark behind DESIGN.md's "split-f16" discussion: (1) does v_mfma_f32_16x16x32_f16 keep f16 subnormal
// inputs, (2) how accurate is the 3-product hi/lo split of an fp32 GEMM, (3) what rate does the instruction reach.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_f16_split.hip -o /tmp/split && /tmp/split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

__device__ inline void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
    hi = __builtin_bit_cast(uint32_t, h);
    const float ra = a - (float)h.x, rb = b - (float)h.y;
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

// D[16x16] = A[16xK] * B[Kx16], one wave; A row-major [16][K], B stored as Bt [16][K]; K multiple of 32
__global__ void gemm_split(const float* A, const float* Bt, float* D, int K, int mode) {
    const int l = threadIdx.x, j = l & 15, g = l >> 4;
    float4_t acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        uint32_t ah[4], al[4], bh[4], bl[4];
        for (int p = 0; p < 4; p++) {
            split2(A[j * K + k0 + 8 * g + 2 * p], A[j * K + k0 + 8 * g + 2 * p + 1], ah[p], al[p]);
            split2(Bt[j * K + k0 + 8 * g + 2 * p], Bt[j * K + k0 + 8 * g + 2 * p + 1], bh[p], bl[p]);
        }
        half8 Ah = __builtin_bit_cast(half8, *(uint4*)ah), Al = __builtin_bit_cast(half8, *(uint4*)al);
        half8 Bh = __builtin_bit_cast(half8, *(uint4*)bh), Bl = __builtin_bit_cast(half8, *(uint4*)bl);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, acc, 0, 0, 0);
        if (mode >= 1) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, acc, 0, 0, 0);
        }
        if (mode >= 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bl, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) D[(4 * g + r) * 16 + j] = acc[r];  // row = 4g+r (A row), col = j (B column)
}

__global__ void rate(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f / (1 + i)); }
    float4_t c[8];
    for (int i = 0; i < 8; i++) c[i] = (float4_t){0, 0, 0, 0};
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; i++) s += c[i].x + c[i].y + c[i].z + c[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int K = 224;
    std::vector<float> A(16 * K), Bt(16 * K), D(256);
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dD, 1 << 22);
    // (1) subnormals: A = 2^-20 everywhere (f16 subnormal), B = 1  ->  K * 2^-20 if kept, 0 if flushed
    for (auto& v : A) v = ldexpf(1.f, -20);
    for (auto& v : Bt) v = 1.f;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
    gemm_split<<<1, 64>>>(dA, dB, dD, K, 0);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    printf("subnormal input: got %.6e expect %.6e (0 => flushed)\n", D[0], K * ldexp(1.0, -20));
    // (2) accuracy on N(0,1)-ish data of several magnitudes
    for (float scale : {1.f, 1e-3f, 30.f}) {
        srand(1);
        for (auto& v : A) v = scale * ((rand() / (float)RAND_MAX) * 2 - 1);
        for (auto& v : Bt) v = ((rand() / (float)RAND_MAX) * 2 - 1);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; mode++) {
            gemm_split<<<1, 64>>>(dA, dB, dD, K, mode);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            double emax = 0, ref_max = 0, e32 = 0;
            for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
                double r = 0; float f = 0;
                for (int k = 0; k < K; k++) { r += (double)A[i * K + k] * Bt[j * K + k]; f += A[i * K + k] * Bt[j * K + k]; }
                emax = fmax(emax, fabs(D[i * 16 + j] - r)); ref_max = fmax(ref_max, fabs(r)); e32 = fmax(e32, fabs(f - r));
            }
            printf("scale %g mode %d (products %d): max abs err %.3e  (|ref|max %.3e, rel %.3e; plain fp32 loop err %.3e)\n",
                   scale, mode, mode == 0 ? 1 : mode == 1 ? 3 : 4, emax, ref_max, emax / ref_max, e32);
        }
    }
    // (3) rate
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    rate<<<blocks, 256>>>(dD, 100);
    hipEventRecord(e0); rate<<<blocks, 256>>>(dD, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * 16 * 16 * 32 * 2;
    printf("v_mfma_f32_16x16x32_f16: %.1f TFLOP/s\n", flops / ms * 1e-9);
    return 0;
}
