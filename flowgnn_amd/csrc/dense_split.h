// Dense layer out[v][:] = act(W in[v][:] + b) for 100 input features on the f16 matrix pipe, fp32-accurate by operand
// splitting (the scheme of gin_split.hip: x = hi + lo in f16, w x ~= w_hi x_hi + w_hi x_lo + w_lo x_hi, fp32 accumulate,
// weights pre-scaled by a power of two, *range_flag raised beyond the f16 range so that the engine repeats the pass on
// the fp32 kernels).  Against dense100_kernel (fp32 MFMA, MFMA-bound at 2.1 ms per 7 M rows) this one is bound by its
// 800 B of HBM traffic per row.
//   * one persistent 8-wave workgroup per ~1/3 CU; all fragments of the layer (OT x 6.25 KiB + biases) live in LDS for the
//     whole kernel: no weight streaming, no barriers after the first;
//   * each wave walks 16-row tiles; the rows of its next tile are loaded (registers) before the current one is computed,
//     so the loads have a whole tile of MFMAs and stores to land in.
// Fragment layout (host: pack_dense100_split): per output tile t: 3 K-steps x {hi, lo} x 1 KiB (lane l = (i = l & 15,
// gk = l >> 4), slot e: W[16 t + i][16 (2 ks + (e >> 2)) + 4 gk + (e & 3)]), then the fp32 K-tail fragments OT x 64
// floats (W[16 t + i][96 + gk]), then the bias padded to 16 OT floats (pre-scaled), then 1 / scale.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "device_common.h"

namespace fg {

typedef _Float16 ds_half8_t __attribute__((ext_vector_type(8)));
typedef uint32_t ds_uint4_t __attribute__((ext_vector_type(4)));

constexpr size_t dense100_split_bytes(int OT) { return (size_t)OT * 6 * 1024 + (size_t)OT * 256 + (size_t)OT * 64 + 16; }

#define DS_MFMA16(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ds_half8_t, (a)), __builtin_bit_cast(ds_half8_t, (b)), (c), 0, 0, 0)
#define DS_SPLIT2(a, b, HI, LO)                                                                                   \
    do {                                                                                                          \
        const float a_ = (a), b_ = (b);                                                                           \
        const uint32_t hp_ = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a_, b_));                    \
        float la_, lb_; /* a - (float)hi: one v_fma_mix each (f16 source read in place) instead of cvt + sub */   \
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la_) : "v"(hp_), "v"(a_));                  \
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb_) : "v"(hp_), "v"(b_));   \
        (HI) = hp_;                                                                                               \
        (LO) = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(la_, lb_));                                \
    } while (0)

template <int OT, bool RELU_OUT>
__global__ __launch_bounds__(512) void dense100_split_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              const uint8_t* __restrict__ wpk, int n_tot, int out_dim,
                                                              int* __restrict__ range_flag) {
    constexpr int WBYTES = (int)dense100_split_bytes(OT);
    constexpr int TAIL_OFF = OT * 6 * 1024, BIAS_OFF = TAIL_OFF + OT * 256, SCALE_OFF = BIAS_OFF + OT * 64;
    __shared__ __attribute__((aligned(16))) char s_w[WBYTES];
    for (int i = threadIdx.x; i < WBYTES / 16; i += 512)
        reinterpret_cast<float4*>(s_w)[i] = reinterpret_cast<const float4*>(wpk)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const float oscale = *reinterpret_cast<const float*>(s_w + SCALE_OFF);
    const long long n_tiles = ((long long)n_tot + 15) / 16;
    long long tile = (long long)blockIdx.x * 8 + wave;
    const long long stride = (long long)gridDim.x * 8;
    float vmax = 0.0f;
    float4 x[6];
    float xt = 0.0f;
    if (tile < n_tiles) {
        long long node = tile * 16 + j;
        if (node >= n_tot) node = n_tot - 1;
        const float* row = in + (size_t)node * 100;
#pragma unroll
        for (int q = 0; q < 6; q++) x[q] = *reinterpret_cast<const float4*>(row + 16 * q + 4 * g);
        xt = row[96 + g];
    }
    for (; tile < n_tiles; tile += stride) {
        // B operands of this tile from the rows loaded one iteration ago
        ds_uint4_t b_hi[3], b_lo[3];
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            DS_SPLIT2(x[2 * ks].x, x[2 * ks].y, b_hi[ks].x, b_lo[ks].x);
            DS_SPLIT2(x[2 * ks].z, x[2 * ks].w, b_hi[ks].y, b_lo[ks].y);
            DS_SPLIT2(x[2 * ks + 1].x, x[2 * ks + 1].y, b_hi[ks].z, b_lo[ks].z);
            DS_SPLIT2(x[2 * ks + 1].z, x[2 * ks + 1].w, b_hi[ks].w, b_lo[ks].w);
        }
#pragma unroll
        for (int q = 0; q < 6; q++) {
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(x[q].x)), __builtin_fabsf(x[q].y));
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(x[q].z)), __builtin_fabsf(x[q].w));
        }
        asm volatile("" : "+v"(vmax));
        const float b_t = xt;
        {   // rows of this wave's next tile
            const long long nt = tile + stride;
            if (nt < n_tiles) {
                long long node = nt * 16 + j;
                if (node >= n_tot) node = n_tot - 1;
                const float* row = in + (size_t)node * 100;
#pragma unroll
                for (int q = 0; q < 6; q++) x[q] = *reinterpret_cast<const float4*>(row + 16 * q + 4 * g);
                xt = row[96 + g];
            }
        }
        const long long node = tile * 16 + j;
#pragma unroll
        for (int t = 0; t < OT; t++) {
            const float4 bv = *reinterpret_cast<const float4*>(s_w + BIAS_OFF + (16 * t + 4 * g) * 4);
            float4_t acc = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                const ds_uint4_t a_hi = *reinterpret_cast<const ds_uint4_t*>(s_w + ((t * 3 + ks) * 2 + 0) * 1024 + lane * 16);
                const ds_uint4_t a_lo = *reinterpret_cast<const ds_uint4_t*>(s_w + ((t * 3 + ks) * 2 + 1) * 1024 + lane * 16);
                acc = DS_MFMA16(a_hi, b_hi[ks], acc);
                acc = DS_MFMA16(a_hi, b_lo[ks], acc);
                acc = DS_MFMA16(a_lo, b_hi[ks], acc);
            }
            const float at = *reinterpret_cast<const float*>(s_w + TAIL_OFF + t * 256 + lane * 4);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at, b_t, acc, 0, 0, 0);
            const int col = 16 * t + 4 * g;
            if (col < out_dim && node < n_tot) {
                float4_t r = acc * oscale;
                if (RELU_OUT) { r.x = relu1(r.x); r.y = relu1(r.y); r.z = relu1(r.z); r.w = relu1(r.w); }
                *reinterpret_cast<float4*>(out + (size_t)node * out_dim + col) = make_float4(r.x, r.y, r.z, r.w);
            }
        }
    }
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

// out[v][:] = res[v][:] + relu(b + W_0 in[v][0][:] + W_1 in[v][1][:]): two blocks of 100 input features per row (DGN: mean
// aggregate | directional derivative).  Same scheme; the fragments of both blocks (2 x OT x 6.25 KiB = 90 KB) stay in LDS
// for the whole kernel, so one persistent 16-wave workgroup per CU (4 waves per SIMD).  Layout: block 0 fragments, block 1
// fragments (each as in dense100_split_kernel: [t][ks][hi/lo] x 1 KiB, then OT x 64 fp32 tail floats), bias, 1 / scale.
constexpr size_t dense200_split_bytes(int OT) { return 2 * ((size_t)OT * 6 * 1024 + (size_t)OT * 256) + (size_t)OT * 64 + 16; }

template <int OT>
__global__ __launch_bounds__(1024) void dense200_res_relu_split_kernel(const float* __restrict__ in, const float* __restrict__ res,
                                                                        float* __restrict__ out, const uint8_t* __restrict__ wpk,
                                                                        int n_tot, int out_dim, int* __restrict__ range_flag) {
    constexpr int BLK = OT * 6 * 1024 + OT * 256;
    constexpr int WBYTES = (int)dense200_split_bytes(OT);
    constexpr int BIAS_OFF = 2 * BLK, SCALE_OFF = BIAS_OFF + OT * 64;
    __shared__ __attribute__((aligned(16))) char s_w[WBYTES];
    for (int i = threadIdx.x; i < WBYTES / 16; i += 1024)
        reinterpret_cast<float4*>(s_w)[i] = reinterpret_cast<const float4*>(wpk)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const float oscale = *reinterpret_cast<const float*>(s_w + SCALE_OFF);
    const long long n_tiles = ((long long)n_tot + 15) / 16;
    const long long stride = (long long)gridDim.x * 16;
    float vmax = 0.0f;
    for (long long tile = (long long)blockIdx.x * 16 + wave; tile < n_tiles; tile += stride) {
        long long node = tile * 16 + j;
        const bool valid = node < n_tot;
        if (!valid) node = n_tot - 1;
        float4 x[2][6];
        float xt[2];
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const float* row = in + ((size_t)node * 2 + b) * 100;
#pragma unroll
            for (int q = 0; q < 6; q++) x[b][q] = *reinterpret_cast<const float4*>(row + 16 * q + 4 * g);
            xt[b] = row[96 + g];
        }
        float4_t acc[OT];
#pragma unroll
        for (int t = 0; t < OT; t++) {
            const float4 bv = *reinterpret_cast<const float4*>(s_w + BIAS_OFF + (16 * t + 4 * g) * 4);
            acc[t] = (float4_t){bv.x, bv.y, bv.z, bv.w};
        }
#pragma unroll
        for (int b = 0; b < 2; b++) {
            ds_uint4_t b_hi[3], b_lo[3];
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                DS_SPLIT2(x[b][2 * ks].x, x[b][2 * ks].y, b_hi[ks].x, b_lo[ks].x);
                DS_SPLIT2(x[b][2 * ks].z, x[b][2 * ks].w, b_hi[ks].y, b_lo[ks].y);
                DS_SPLIT2(x[b][2 * ks + 1].x, x[b][2 * ks + 1].y, b_hi[ks].z, b_lo[ks].z);
                DS_SPLIT2(x[b][2 * ks + 1].z, x[b][2 * ks + 1].w, b_hi[ks].w, b_lo[ks].w);
            }
#pragma unroll
            for (int q = 0; q < 6; q++) {
                vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(x[b][q].x)), __builtin_fabsf(x[b][q].y));
                vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(x[b][q].z)), __builtin_fabsf(x[b][q].w));
            }
            asm volatile("" : "+v"(vmax));
            const char* wb = s_w + b * BLK;
#pragma unroll
            for (int t = 0; t < OT; t++) {
#pragma unroll
                for (int ks = 0; ks < 3; ks++) {
                    const ds_uint4_t a_hi = *reinterpret_cast<const ds_uint4_t*>(wb + ((t * 3 + ks) * 2 + 0) * 1024 + lane * 16);
                    const ds_uint4_t a_lo = *reinterpret_cast<const ds_uint4_t*>(wb + ((t * 3 + ks) * 2 + 1) * 1024 + lane * 16);
                    acc[t] = DS_MFMA16(a_hi, b_hi[ks], acc[t]);
                    acc[t] = DS_MFMA16(a_hi, b_lo[ks], acc[t]);
                    acc[t] = DS_MFMA16(a_lo, b_hi[ks], acc[t]);
                }
                const float at = *reinterpret_cast<const float*>(wb + OT * 6 * 1024 + t * 256 + lane * 4);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, xt[b], acc[t], 0, 0, 0);
            }
        }
        if (valid) {
#pragma unroll
            for (int t = 0; t < OT; t++) {
                const int col = 16 * t + 4 * g;
                if (col < out_dim) {
                    const size_t off = (size_t)node * out_dim + col;
                    const float4 hv = *reinterpret_cast<const float4*>(res + off);
                    const float4_t r = acc[t] * oscale;
                    *reinterpret_cast<float4*>(out + off) =
                        make_float4(hv.x + relu1(r.x), hv.y + relu1(r.y), hv.z + relu1(r.z), hv.w + relu1(r.w));
                }
            }
        }
    }
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

// host: W [out_dim][2][100] (out, block, in), b [out_dim]  ->  dense200_split_bytes(OT)
static inline void pack_dense200_split(const float* W, const float* b, int out_dim, int OT, uint8_t* out) {
    const size_t total = dense200_split_bytes(OT);
    std::memset(out, 0, total);
    float m = 0.0f;
    for (size_t i = 0; i < (size_t)out_dim * 200; i++) m = std::fmax(m, std::fabs(W[i]));
    const float sc = (m > 0.0f && std::isfinite(m)) ? std::ldexp(1.0f, -std::ilogb(m)) : 1.0f;
    const size_t blk = (size_t)OT * 6 * 1024 + (size_t)OT * 256;
    for (int blkid = 0; blkid < 2; blkid++)
        for (int t = 0; t < OT; t++)
            for (int lane = 0; lane < 64; lane++) {
                const int i = lane & 15, gk = lane >> 4, o = 16 * t + i;
                uint8_t* base = out + blkid * blk;
                for (int ks = 0; ks < 3; ks++)
                    for (int e = 0; e < 8; e++) {
                        const int f = 16 * (2 * ks + (e >> 2)) + 4 * gk + (e & 3);
                        const float v = o < out_dim ? W[((size_t)o * 2 + blkid) * 100 + f] * sc : 0.0f;
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        std::memcpy(base + (size_t)((t * 3 + ks) * 2 + 0) * 1024 + lane * 16 + e * 2, &hi, 2);
                        std::memcpy(base + (size_t)((t * 3 + ks) * 2 + 1) * 1024 + lane * 16 + e * 2, &lo, 2);
                    }
                const float tail = o < out_dim ? W[((size_t)o * 2 + blkid) * 100 + 96 + gk] * sc : 0.0f;
                std::memcpy(base + (size_t)OT * 6 * 1024 + (size_t)t * 256 + lane * 4, &tail, 4);
            }
    for (int xx = 0; xx < 16 * OT; xx++) {
        const float bb = xx < out_dim ? b[xx] * sc : 0.0f;
        std::memcpy(out + 2 * blk + (size_t)xx * 4, &bb, 4);
    }
    const float os = 1.0f / sc;
    std::memcpy(out + 2 * blk + (size_t)OT * 64, &os, 4);
}

// host: W [out_dim][100] row-major, b [out_dim]  ->  dense100_split_bytes(OT) bytes for dense100_split_kernel<OT>
static inline void pack_dense100_split(const float* W, const float* b, int out_dim, int OT, uint8_t* out) {
    const size_t total = dense100_split_bytes(OT);
    std::memset(out, 0, total);
    float m = 0.0f;
    for (size_t i = 0; i < (size_t)out_dim * 100; i++) m = std::fmax(m, std::fabs(W[i]));
    const float sc = (m > 0.0f && std::isfinite(m)) ? std::ldexp(1.0f, -std::ilogb(m)) : 1.0f;  // max |W| * sc in [1, 2)
    const size_t tail_off = (size_t)OT * 6 * 1024, bias_off = tail_off + (size_t)OT * 256, scale_off = bias_off + (size_t)OT * 64;
    for (int t = 0; t < OT; t++) {
        for (int lane = 0; lane < 64; lane++) {
            const int i = lane & 15, gk = lane >> 4, o = 16 * t + i;
            for (int ks = 0; ks < 3; ks++)
                for (int e = 0; e < 8; e++) {
                    const int f = 16 * (2 * ks + (e >> 2)) + 4 * gk + (e & 3);
                    const float v = o < out_dim ? W[(size_t)o * 100 + f] * sc : 0.0f;
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    std::memcpy(out + (size_t)((t * 3 + ks) * 2 + 0) * 1024 + lane * 16 + e * 2, &hi, 2);
                    std::memcpy(out + (size_t)((t * 3 + ks) * 2 + 1) * 1024 + lane * 16 + e * 2, &lo, 2);
                }
            const float tail = o < out_dim ? W[(size_t)o * 100 + 96 + gk] * sc : 0.0f;
            std::memcpy(out + tail_off + (size_t)t * 256 + lane * 4, &tail, 4);
        }
        for (int xx = 0; xx < 16; xx++) {
            const float bb = 16 * t + xx < out_dim ? b[16 * t + xx] * sc : 0.0f;
            std::memcpy(out + bias_off + (size_t)(16 * t + xx) * 4, &bb, 4);
        }
    }
    const float os = 1.0f / sc;
    std::memcpy(out + scale_off, &os, 4);
}

}  // namespace fg
