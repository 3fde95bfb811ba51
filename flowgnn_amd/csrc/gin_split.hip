// GIN layer, "split-f16" variant.  Test/bench infrastructure excluded, this is the default GIN hot path.
//
// Why: the node MLP is 4e4 MAC per node.  On the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, 157 TFLOP/s measured
// peak) that is 3.4 ms per layer at 2^18 molhiv graphs before anything else happens, five times the HBM time of the
// layer.  The f16 pipe is 12.5x faster (v_mfma_f32_16x16x32_f16: 1.95 PFLOP/s measured, tools/mfma_f16_split.hip).
//
// How: every fp32 operand is split exactly into two f16 values, x = hi + lo + r with |r| <= 2^-20 |x|
// (hi = rtz_f16(x), lo = rtz_f16(x - hi); the subtraction is exact), and a product is evaluated as
//      w x  ~=  w_hi x_hi + w_hi x_lo + w_lo x_hi            (three MFMAs, fp32 accumulate; the dropped
//                                                              w_lo x_lo term is <= 2^-22 |w x|)
// so each product carries a relative error of about 2^-20 -- the same size as the fp32 rounding error the plain
// fp32 dot product accumulates over K = 100..200 terms (measured side by side in tools/mfma_f16_split.hip:
// 6.5e-6 vs 5.9e-6 absolute on |ref| = 15).  f16 subnormals are honoured by the MFMA (same tool), so small
// operands degrade to an ABSOLUTE error of 6e-8, not to zero.  Weights are pre-scaled by a power of two per matrix
// (exact; undone in the epilogue) so that their largest entry is in [1,2).  Operands beyond the f16 range
// (|x| > 6e4; the reference's own Q6.10 activations are confined to [-32,32)) set *range_flag, and the engine then
// repeats the forward pass on the fp32 MFMA kernel (gin.hip), so the result is fp32-accurate for every input.
//
// Shape: as gin_layer_fused_kernel (gin.hip) -- transposed product (nodes are MFMA columns), the accumulators of
// the first linear layer become the B operands of the second without leaving the wave, weights pre-packed in
// fragment order and streamed L2 -> LDS by LDS-DMA, double buffered in two distinct LDS objects.
//   lane (j = lane & 15, g = lane >> 4); B operand of K-step ks, slot e (0..7): feature 16 (2 ks + (e >> 2)) + 4 g +
//   (e & 3) -- exactly what the lane gathered as float4 pieces q = 2 ks, 2 ks + 1, and exactly what the lane holds of
//   hidden tiles 2 ks, 2 ks + 1 after the first layer.  K = 100 = 3 x 32 + 4: the 4-feature tail is one fp32 MFMA.
//   8 steps: step s runs hidden tiles 2s, 2s+1 of MLP1 (13 tiles, 20 MFMAs per node tile) and K-step s-1 of MLP2
//   (7 output tiles x 3 = 21 MFMAs per node tile).
#include "gin_split.h"

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "device_common.h"

namespace fg {

namespace {

constexpr int GS_D = 100;
constexpr int GS_H = 200;
constexpr int GS_T2 = 7;

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

#define GS_MFMA16(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, (a)), __builtin_bit_cast(half8_t, (b)), (c), 0, 0, 0)
#define GS_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// (a, b) -> packed f16 pairs HI, LO with a = hi.x + lo.x (+ 2^-20 |a|), same for b  (a macro: the targets are
// elements of ext vectors, which cannot bind to references)
#define GS_SPLIT2(a, b, HI, LO)                                                                                   \
    do {                                                                                                          \
        const float a_ = (a), b_ = (b);                                                                           \
        const uint32_t hp_ = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a_, b_));                    \
        float la_, lb_; /* a - (float)hi: one v_fma_mix each (f16 source read in place) instead of cvt + sub */   \
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la_) : "v"(hp_), "v"(a_));                  \
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb_) : "v"(hp_), "v"(b_));   \
        (HI) = hp_;                                                                                               \
        (LO) = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(la_, lb_));                                \
    } while (0)

__device__ __forceinline__ float gs_relu(float x) {  // max(x, 0) as v_max_i32: one instruction where fmaxf on an MFMA
    const int b = __builtin_bit_cast(int, x);         // result costs two (the compiler must quiet a possible sNaN first)
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

constexpr int GS_HUB = 8;     // rows with more in-edges than this are summed by the whole wave
constexpr int GS_MAXHUB = 4;  // ... if the wave has at most this many of them
template <int CTRL>
__device__ __forceinline__ float gs_dpp(float v) {  // v of another lane of the same row of 16, selected by the DPP control
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int WAVES>
__device__ __forceinline__ void gs_issue_chunk(const uint8_t* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < (27 + WAVES - 1) / WAVES; p++) {
        const int piece = wave + WAVES * p;  // 26 full pieces of 1 KiB + 640 B
        if (piece < 26 || (piece == 26 && lane < 40)) {
            const uint8_t* g = gchunk + piece * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds_buf + piece * 1024), 16, 0, 0);
        }
    }
}

template <int NT>
__device__ __forceinline__ void gs_step(const char* wb, int s, int lane, int g, const uint4_t (&in_hi)[NT][3],
                                        const uint4_t (&in_lo)[NT][3], const float (&in_t)[NT], uint4_t (&h_hi)[NT],
                                        uint4_t (&h_lo)[NT], float4_t (&acc2)[NT][GS_T2], float& vmax) {
    uint4_t n_hi[NT], n_lo[NT];
    if (s < GS_STEPS - 1) {  // MLP1: hidden tiles 2s, 2s+1
        float4_t acc1[2][NT];
#pragma unroll
        for (int tl = 0; tl < 2; tl++) {
            const float4 b = *reinterpret_cast<const float4*>(wb + GS_B1_OFF + tl * 64 + g * 16);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc1[tl][nt] = (float4_t){b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            uint4_t a[2][2];
#pragma unroll
            for (int tl = 0; tl < 2; tl++)
#pragma unroll
                for (int p = 0; p < 2; p++)
                    a[tl][p] = *reinterpret_cast<const uint4_t*>(wb + ((ks * 2 + tl) * 2 + p) * 1024 + lane * 16);
#pragma unroll
            for (int tl = 0; tl < 2; tl++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc1[tl][nt] = GS_MFMA16(a[tl][0], in_hi[nt][ks], acc1[tl][nt]);
#pragma unroll
            for (int tl = 0; tl < 2; tl++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc1[tl][nt] = GS_MFMA16(a[tl][0], in_lo[nt][ks], acc1[tl][nt]);
#pragma unroll
            for (int tl = 0; tl < 2; tl++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc1[tl][nt] = GS_MFMA16(a[tl][1], in_hi[nt][ks], acc1[tl][nt]);
        }
#pragma unroll
        for (int tl = 0; tl < 2; tl++) {
            const float at = *reinterpret_cast<const float*>(wb + GS_TAIL_OFF + tl * 256 + lane * 4);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc1[tl][nt] = GS_MFMA32(at, in_t[nt], acc1[tl][nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            float4_t r0 = acc1[0][nt], r1 = acc1[1][nt];
            r0.x = gs_relu(r0.x); r0.y = gs_relu(r0.y); r0.z = gs_relu(r0.z); r0.w = gs_relu(r0.w);
            r1.x = gs_relu(r1.x); r1.y = gs_relu(r1.y); r1.z = gs_relu(r1.z); r1.w = gs_relu(r1.w);
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, r0.x), r0.y);
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, r0.z), r0.w);
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, r1.x), r1.y);
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, r1.z), r1.w);
            asm volatile("" : "+v"(vmax));
            GS_SPLIT2(r0.x, r0.y, n_hi[nt].x, n_lo[nt].x);
            GS_SPLIT2(r0.z, r0.w, n_hi[nt].y, n_lo[nt].y);
            GS_SPLIT2(r1.x, r1.y, n_hi[nt].z, n_lo[nt].z);
            GS_SPLIT2(r1.z, r1.w, n_hi[nt].w, n_lo[nt].w);
        }
    }
    if (s > 0) {  // MLP2: K-step s-1 = hidden tiles 2(s-1), 2(s-1)+1 of the previous step
#pragma unroll
        for (int t0 = 0; t0 < GS_T2; t0 += 2) {
            uint4_t a[2][2];
#pragma unroll
            for (int tl = 0; tl < 2; tl++)
#pragma unroll
                for (int p = 0; p < 2; p++)
                    if (t0 + tl < GS_T2)
                        a[tl][p] = *reinterpret_cast<const uint4_t*>(wb + GS_W2_OFF + ((t0 + tl) * 2 + p) * 1024 + lane * 16);
#pragma unroll
            for (int tl = 0; tl < 2; tl++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    if (t0 + tl < GS_T2) acc2[nt][t0 + tl] = GS_MFMA16(a[tl][0], h_hi[nt], acc2[nt][t0 + tl]);
#pragma unroll
            for (int tl = 0; tl < 2; tl++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    if (t0 + tl < GS_T2) acc2[nt][t0 + tl] = GS_MFMA16(a[tl][0], h_lo[nt], acc2[nt][t0 + tl]);
#pragma unroll
            for (int tl = 0; tl < 2; tl++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    if (t0 + tl < GS_T2) acc2[nt][t0 + tl] = GS_MFMA16(a[tl][1], h_hi[nt], acc2[nt][t0 + tl]);
        }
    }
    if (s < GS_STEPS - 1) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) { h_hi[nt] = n_hi[nt]; h_lo[nt] = n_lo[nt]; }
    }
}

template <int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gin_layer_split_kernel(const float* __restrict__ h, float* __restrict__ hout,
                                                               const int* __restrict__ row_ptr,
                                                               const int* __restrict__ src,
                                                               const uint8_t* __restrict__ ecode,
                                                               const float* __restrict__ ecomb,
                                                               const uint8_t* __restrict__ wchunks, int n_tot, int relu_out,
                                                               int* __restrict__ range_flag, const float* __restrict__ pool_w) {
    // two DISTINCT LDS objects: the compiler can then prove that the LDS-DMA into one does not alias the ds_reads
    // of the other and leaves the DMA in flight under the MFMAs (see gin_layer_fused_kernel)
    __shared__ __attribute__((aligned(16))) char s_a[GS_CHUNK_BYTES];  // edge-embedding combos, then odd chunks
    __shared__ __attribute__((aligned(16))) char s_b[GS_CHUNK_BYTES];  // even chunks
    __shared__ float s_hub[NT == 1 ? WAVES * GS_MAXHUB * GS_D : 1];     // parked sums of hub rows
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // in an SGPR: DMA addresses = scalar base + lane * 16
    const int j = lane & 15, g = lane >> 4;
    const long long node_base = (long long)blockIdx.x * (WAVES * 16 * NT) + wave * (16 * NT);

    // ---- gather (MP unit): a = h[v] + sum_e relu(h[src_e] + ecomb[code_e]), CSR order
    // A workgroup spends about half its life before its first MFMA (measured with s_memtime: staging the combos 3.5 us,
    // gather 4.5 us, 8 MLP steps 10 us), and all of that is a chain of dependent round trips.  So: the CSR row bounds are
    // requested first; chunk 0 and the edge-embedding combos go to LDS by DMA (no register round trip) while the row
    // bounds come back; the node's own row and the first two CSR entries are requested next; only then does the wave wait.
    float bq[NT][25];
    int e_cur[NT], e_end[NT];
    float4 self_x[NT][6];
    float self_t[NT];
    long long self_row[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        long long node = node_base + nt * 16 + j;
        const bool valid = node < n_tot;
        if (!valid) node = n_tot - 1;
        self_row[nt] = node;
        e_cur[nt] = row_ptr[node];
        e_end[nt] = row_ptr[node + 1];
        if (!valid) e_end[nt] = e_cur[nt];
    }
    gs_issue_chunk<WAVES>(wchunks, s_b, wave, lane);  // chunk 0
#pragma unroll
    for (int p = 0; p < (24 + WAVES - 1) / WAVES; p++) {  // 60 x 400 B of combos = 23 pieces of 1 KiB + 448 B
        const int piece = wave + WAVES * p;
        if (piece < 23 || (piece == 23 && lane < 28)) {
            const char* gp = reinterpret_cast<const char*>(ecomb) + piece * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(s_a + piece * 1024), 16, 0, 0);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const float* hr = h + (size_t)self_row[nt] * GS_D + 4 * g;
#pragma unroll
        for (int q = 0; q < 6; q++) self_x[nt][q] = *reinterpret_cast<const float4*>(hr + 16 * q);
        self_t[nt] = h[(size_t)self_row[nt] * GS_D + 96 + g];
#pragma unroll
        for (int k = 0; k < 25; k++) bq[nt][k] = 0.0f;
    }
    // CSR entries two ahead of the row gathers (first use of the row bounds)
    int ua[NT], ca[NT], ub[NT], cb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int ea = e_cur[nt] < e_end[nt] ? e_cur[nt] : 0, eb = e_cur[nt] + 1 < e_end[nt] ? e_cur[nt] + 1 : 0;
        ua[nt] = src[ea]; ca[nt] = ecode[ea];
        ub[nt] = src[eb]; cb[nt] = ecode[eb];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the combos (and of chunk 0)
    __syncthreads();
    const float* s_ecomb = reinterpret_cast<const float*>(s_a);
    // Hub rows (in-degree > GS_HUB; GIN-VN's virtual node has one in-edge per node of its graph): left to the loop below,
    // one such row keeps its whole wave iterating with 4 of 64 lanes busy.  Instead the row's in-edges are dealt to the 16
    // node lanes of the wave (edge i of the row to lane i mod 16), summed per lane in CSR order and combined with a
    // butterfly over the node lanes (DPP: xor 1, xor 2, half-row mirror, row mirror).  The association differs from the
    // oracle's strictly sequential sum (covered by the stated tolerance) but depends only on the row, not on where it
    // sits in the batch, so results stay bit-identical under any batch split or order.  The owner lanes park the row
    // sum in LDS (400 B per hub row) until the regular rows are done.
    bool is_hub = false;
    int hub_slot = 0;
    if constexpr (NT == 1) {
        const int deg = e_end[0] - e_cur[0];
        unsigned long long hubs = __ballot(deg > GS_HUB) & 0xFFFFull;  // lanes 0..15: one bit per node of the wave
        if (__popcll(hubs) > GS_MAXHUB) hubs = 0;  // uniformly dense rows (kNN graphs): the ordinary loop keeps all lanes busy
        int nh = 0;
        while (hubs != 0) {  // wave-uniform, at most GS_MAXHUB trips
            const int hj = __ffsll((long long)hubs) - 1;
            hubs &= hubs - 1;
            const int hb = __builtin_amdgcn_readlane(e_cur[0], hj), he = __builtin_amdgcn_readlane(e_end[0], hj);
            for (int e = hb + j; __any(e < he); e += 16) {
                if (e < he) {
                    const int u = src[e];
                    const int code = ecode[e];
                    const float* hr = h + (size_t)u * GS_D + 4 * g;
                    const float* er = s_ecomb + code * GS_D + 4 * g;
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        const float4 x = *reinterpret_cast<const float4*>(hr + 16 * q);
                        const float4 w = *reinterpret_cast<const float4*>(er + 16 * q);
                        bq[0][4 * q + 0] += relu1(w.x + x.x);
                        bq[0][4 * q + 1] += relu1(w.y + x.y);
                        bq[0][4 * q + 2] += relu1(w.z + x.z);
                        bq[0][4 * q + 3] += relu1(w.w + x.w);
                    }
                    bq[0][24] += relu1(s_ecomb[code * GS_D + 96 + g] + h[(size_t)u * GS_D + 96 + g]);
                }
            }
#pragma unroll
            for (int k = 0; k < 25; k++) {
                float v = bq[0][k];
                v += gs_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
                v += gs_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
                v += gs_dpp<0x141>(v);  // row_half_mirror
                v += gs_dpp<0x140>(v);  // row_mirror
                if (j == hj) s_hub[(wave * GS_MAXHUB + nh) * GS_D + g * 25 + k] = v;
                bq[0][k] = 0.0f;
            }
            if (j == hj) { is_hub = true; hub_slot = nh; }
            nh++;
        }
        if (is_hub) e_cur[0] = e_end[0];  // this row's edges are done
    }
    // one in-edge per trip, CSR entries one trip ahead
    while (true) {
        bool any = false;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) any |= (e_cur[nt] < e_end[nt]);
        if (!__any(any)) break;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            if (e_cur[nt] < e_end[nt]) {
                const int u = ua[nt];
                const int code = ca[nt];
                e_cur[nt]++;
                ua[nt] = ub[nt]; ca[nt] = cb[nt];
                if (e_cur[nt] + 1 < e_end[nt]) {
                    ub[nt] = src[e_cur[nt] + 1];
                    cb[nt] = ecode[e_cur[nt] + 1];
                }
                const float* hr = h + (size_t)u * GS_D + 4 * g;
                const float* er = s_ecomb + code * GS_D + 4 * g;
                float4 x[6];
#pragma unroll
                for (int q = 0; q < 6; q++) x[q] = *reinterpret_cast<const float4*>(hr + 16 * q);
                const float xt = h[(size_t)u * GS_D + 96 + g];
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const float4 w = *reinterpret_cast<const float4*>(er + 16 * q);
                    bq[nt][4 * q + 0] += relu1(w.x + x[q].x);
                    bq[nt][4 * q + 1] += relu1(w.y + x[q].y);
                    bq[nt][4 * q + 2] += relu1(w.z + x[q].z);
                    bq[nt][4 * q + 3] += relu1(w.w + x[q].w);
                }
                bq[nt][24] += relu1(s_ecomb[code * GS_D + 96 + g] + xt);
            }
        }
    }
    if constexpr (NT == 1) {
        if (is_hub) {
#pragma unroll
            for (int k = 0; k < 25; k++) bq[0][k] = s_hub[(wave * GS_MAXHUB + hub_slot) * GS_D + g * 25 + k];
        }
    }
    float vmax = 0.0f;
    uint4_t in_hi[NT][3], in_lo[NT][3];
    float in_t[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {  // + (1 + eps) h[v], eps == 0; then split into the MLP1 B operands
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 x = self_x[nt][q];
            bq[nt][4 * q + 0] += x.x; bq[nt][4 * q + 1] += x.y; bq[nt][4 * q + 2] += x.z; bq[nt][4 * q + 3] += x.w;
        }
        bq[nt][24] += self_t[nt];
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            GS_SPLIT2(bq[nt][8 * ks + 0], bq[nt][8 * ks + 1], in_hi[nt][ks].x, in_lo[nt][ks].x);
            GS_SPLIT2(bq[nt][8 * ks + 2], bq[nt][8 * ks + 3], in_hi[nt][ks].y, in_lo[nt][ks].y);
            GS_SPLIT2(bq[nt][8 * ks + 4], bq[nt][8 * ks + 5], in_hi[nt][ks].z, in_lo[nt][ks].z);
            GS_SPLIT2(bq[nt][8 * ks + 6], bq[nt][8 * ks + 7], in_hi[nt][ks].w, in_lo[nt][ks].w);
        }
#pragma unroll
        for (int k = 0; k < 24; k += 2)
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(bq[nt][k])), __builtin_fabsf(bq[nt][k + 1]));
        in_t[nt] = bq[nt][24];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk 0
    __syncthreads();  // chunk 0 resident; every wave is done with the edge-embedding combos: s_a may be overwritten

    // ---- node MLP (NT unit), weights streamed through LDS
    float4_t acc2[NT][GS_T2];
#pragma unroll
    for (int t2 = 0; t2 < GS_T2; t2++) {
        const float4 b = *reinterpret_cast<const float4*>(s_b + GS_W2_OFF + (16 * t2 + 4 * g) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc2[nt][t2] = (float4_t){b.x, b.y, b.z, b.w};
    }
    const float oscale = *reinterpret_cast<const float*>(s_b + GS_W2_OFF + 112 * 4);
    uint4_t h_hi[NT], h_lo[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) { h_hi[nt] = (uint4_t){0, 0, 0, 0}; h_lo[nt] = (uint4_t){0, 0, 0, 0}; }
#pragma unroll 1
    for (int c = 0; c < GS_STEPS; c += 2) {
        // even step: compute from s_b while chunk c+1 streams into s_a
        gs_issue_chunk<WAVES>(wchunks + (size_t)(c + 1) * GS_CHUNK_STRIDE, s_a, wave, lane);
        gs_step<NT>(s_b, c, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk c+1 have landed
        __syncthreads();                                  // everyone's landed; everyone is done with s_b
        // odd step: compute from s_a while chunk c+2 streams into s_b
        if (c + 2 < GS_STEPS) gs_issue_chunk<WAVES>(wchunks + (size_t)(c + 2) * GS_CHUNK_STRIDE, s_b, wave, lane);
        gs_step<NT>(s_a, c + 1, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (pool_w != nullptr) {
        // Last layer with the readout folded in: the graph logit is mean_v(h'[v]) . w + b = mean_v(h'[v] . w) + b, so only the
        // per-node dot product leaves the kernel (4 B per node instead of a 400 B row that the readout would read back);
        // hout is then a float[n_tot].  Fixed summation order per node (7 tiles x 4 in the lane, then the 4 lanes of the node).
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            float part = 0.0f;
#pragma unroll
            for (int t2 = 0; t2 < GS_T2; t2++) {
                const int col = 16 * t2 + 4 * g;
                if (col < GS_D) {
                    float4_t r = acc2[nt][t2] * oscale;
                    if (relu_out) { r.x = gs_relu(r.x); r.y = gs_relu(r.y); r.z = gs_relu(r.z); r.w = gs_relu(r.w); }
                    const float4 pw = *reinterpret_cast<const float4*>(pool_w + col);
                    part += r.x * pw.x; part += r.y * pw.y; part += r.z * pw.z; part += r.w * pw.w;
                }
            }
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            const long long node = node_base + nt * 16 + j;
            if (g == 0 && node < n_tot) hout[node] = part;
        }
    } else {
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const long long node = node_base + nt * 16 + j;
        if (node >= n_tot) continue;
        float* row = hout + (size_t)node * GS_D;
#pragma unroll
        for (int t2 = 0; t2 < GS_T2; t2++) {
            const int col = 16 * t2 + 4 * g;
            if (col < GS_D) {
                float4_t r = acc2[nt][t2] * oscale;
                if (relu_out) { r.x = gs_relu(r.x); r.y = gs_relu(r.y); r.z = gs_relu(r.z); r.w = gs_relu(r.w); }
                *reinterpret_cast<float4*>(row + col) = make_float4(r.x, r.y, r.z, r.w);
            }
        }
    }
    }
    // operands beyond the f16 range (inf after pkrtz is impossible, saturation is silent): tell the engine
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

inline float pow2_scale(const float* w, size_t n) {
    float m = 0.0f;
    for (size_t i = 0; i < n; i++) m = std::fmax(m, std::fabs(w[i]));
    if (!(m > 0.0f) || !std::isfinite(m)) return 1.0f;
    return std::ldexp(1.0f, -std::ilogb(m));  // m * scale in [1, 2)
}

inline void put_split(uint8_t* frag, int lane, int e, float v) {
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    std::memcpy(frag + lane * 16 + e * 2, &hi, 2);
    std::memcpy(frag + 1024 + lane * 16 + e * 2, &lo, 2);
}

}  // namespace

void gin_split_pack_layer(const float* w1, const float* b1, const float* w2, const float* b2, uint8_t* out) {
    std::memset(out, 0, GS_LAYER_BYTES);
    const float s1 = pow2_scale(w1, (size_t)GS_H * GS_D);
    const float s2 = pow2_scale(w2, (size_t)GS_D * GS_H);
    for (int s = 0; s < GS_STEPS; s++) {
        uint8_t* ck = out + (size_t)s * GS_CHUNK_STRIDE;
        if (s < GS_STEPS - 1) {
            for (int tl = 0; tl < 2; tl++) {
                const int t = 2 * s + tl;
                for (int lane = 0; lane < 64; lane++) {
                    const int i = lane & 15, gk = lane >> 4;
                    const int o = 16 * t + i;
                    for (int ks = 0; ks < 3; ks++)
                        for (int e = 0; e < 8; e++) {
                            const int f = 16 * (2 * ks + (e >> 2)) + 4 * gk + (e & 3);
                            put_split(ck + (size_t)((ks * 2 + tl) * 2) * 1024, lane, e, o < GS_H ? w1[o * GS_D + f] * s1 : 0.0f);
                        }
                    const float tail = o < GS_H ? w1[o * GS_D + 96 + gk] * s1 : 0.0f;
                    std::memcpy(ck + GS_TAIL_OFF + tl * 256 + lane * 4, &tail, 4);
                }
                for (int x = 0; x < 16; x++) {
                    const int o = 16 * t + x;
                    const float b = o < GS_H ? b1[o] * s1 : 0.0f;
                    std::memcpy(ck + GS_B1_OFF + tl * 64 + x * 4, &b, 4);
                }
            }
        }
        if (s > 0) {
            const int ks = s - 1;
            for (int t2 = 0; t2 < GS_T2; t2++)
                for (int lane = 0; lane < 64; lane++) {
                    const int i = lane & 15, gk = lane >> 4;
                    const int d = 16 * t2 + i;
                    for (int e = 0; e < 8; e++) {
                        const int k = 16 * (2 * ks + (e >> 2)) + 4 * gk + (e & 3);
                        put_split(ck + GS_W2_OFF + (size_t)(t2 * 2) * 1024, lane, e, (d < GS_D && k < GS_H) ? w2[d * GS_H + k] * s2 : 0.0f);
                    }
                }
        } else {
            for (int x = 0; x < 16 * GS_T2; x++) {
                const float b = x < GS_D ? b2[x] * s1 * s2 : 0.0f;
                std::memcpy(ck + GS_W2_OFF + x * 4, &b, 4);
            }
            const float os = 1.0f / (s1 * s2);
            std::memcpy(ck + GS_W2_OFF + 112 * 4, &os, 4);
        }
    }
}

void launch_gin_layer_split(const float* h, float* hout, const int* row_ptr, const int* src, const uint8_t* ecode,
                            const float* ecomb, const uint8_t* chunks, int n_tot, int e_tot, int relu_out, int* range_flag,
                            int nt, hipStream_t s, const float* pool_w) {
    if (nt == 4) {  // 8 waves, 128 nodes per workgroup, 2 workgroups per CU
        const int blocks = (int)ceil_div_ll(n_tot, 128);
        gin_layer_split_kernel<1, 8><<<blocks, 512, 0, s>>>(h, hout, row_ptr, src, ecode, ecomb, chunks, n_tot, relu_out, range_flag, pool_w);
        return;
    }
    if (nt == 2) {
        const int blocks = (int)ceil_div_ll(n_tot, 128);
        gin_layer_split_kernel<2, 4><<<blocks, 256, 0, s>>>(h, hout, row_ptr, src, ecode, ecomb, chunks, n_tot, relu_out, range_flag, pool_w);
    } else {
        const int blocks = (int)ceil_div_ll(n_tot, 64);
        gin_layer_split_kernel<1, 4><<<blocks, 256, 0, s>>>(h, hout, row_ptr, src, ecode, ecomb, chunks, n_tot, relu_out, range_flag, pool_w);
    }
}

}  // namespace fg
