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
// repeats the forward pass on the fp32 MFMA kernel (gin.hip).  The guarantee is therefore: per product a relative error of
// 2^-20 for operands in the f16 normal range, and an ABSOLUTE error of 6e-8 per operand below it (|x| < 6e-5: nothing flags a
// model whose activations are uniformly tiny -- FLOWGNN_<M>_MFMA=f32 is the switch for such a model); NaN operands do not raise
// the flag either (fmax drops them) but propagate to the output.
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

// ---------------------------------------------------------------- graph-resident GIN: all five layers of a tile in ONE workgroup
// The FPGA keeps one graph in BRAM across its layer loop (GIN/src/GIN_compute.cc:72-94).  The counterpart here: a
// persistent 8-wave workgroup (one per CU) owns a tile of WHOLE graphs (<= GR_ROWS rows, <= GR_EDGES in-edges; packed on
// the host, GraphTiles) and keeps the tile's node embeddings in LDS across all five layers:
//   * h never goes to HBM between layers: the gather reads neighbour rows from LDS, the epilogue writes h' in place;
//   * the tile's CSR slice is staged once per tile as 16-bit words (row inside the tile << 6 | edge code) + 16-bit row offsets;
//   * every wave owns TWO MFMA column tiles (32 nodes): each weight fragment read from LDS feeds two MFMAs, which is what
//     takes the LDS array out of the critical path (one column tile per wave needs 650 B of fragments per 16-cycle MFMA);
//   * the weight stream (8 chunks + the 60-row edge-embedding table per layer, the same for every tile) runs through the two
//     chunk buffers without ever draining: chunk s+1 lands while step s computes, the NEXT layer's table lands during step 7
//     (so the buffers swap roles from layer to layer), chunk 0 lands during the gather;
//   * during the last layer's MLP the NEXT tile's rows (encoder output, 400 B per node: the only per-node HBM read of the whole
//     model) and CSR slice are brought in, an eighth per step; the readout (mean of the per-node dot products with the
//     prediction head) happens in the kernel: 4 B per GRAPH leave it.
// LDS: 2 x 27 264 (chunks) + 102 400 (rows) + 2 560 (edges) + 520 (row offsets) + 1 024 (per-node readout terms) = 161 032 B.
constexpr int GR_ROWS = 256;
constexpr int GR_EDGES = 1280;
constexpr float GR_MSG_SCALE = 1.0f / 65536.0f, GR_MSG_UNSCALE = 65536.0f;  // the walk's scaled domain (gr_layer: GR_MSG2)
constexpr int GR_WAVES = 8;
constexpr int GR_STEPS = 7;  // MLP steps of a layer of the resident kernel (the eight-step schedule of the per-layer kernel with its last two merged)

// Weight stream of the resident kernel ("GR chunks"; gin_resident_pack_layer builds it).  Same 8 steps as the per-layer kernel's
// stream, with the padding taken out of the matrix work:
//   * the K tail of the first linear layer (features 96..99) is ONE f16 MFMA per tile whose 32 K-slots carry all three split
//     products -- slots 0-3 w_hi x_hi, 4-7 w_hi x_lo, 8-11 w_lo x_hi -- instead of an fp32 MFMA of twice the cost;
//   * step 6 computes hidden tile 12 only (tile 13 is pure padding: 200 = 12.5 tiles);
//   * step 7 (hidden units 192..199, the last 8 real ones) is ONE packed MFMA per output tile in the same way
//     (slots 0-7 w_hi h_hi | w_hi h_lo of units 192..195, 8-15 the same of 196..199, 16-19 / 24-27 w_lo h_hi).
// 526 MFMAs of 16 cycles per layer and 32-node wave instead of 574 + 28 fp32 ones (= 630 of 16 cycles).
// chunk s:  [0, 12288)       W1 fragments: tile tl (2s + tl) at tl * 6144 + (ks * 2 + p) * 1024   (p = 0 hi, 1 lo)
//           [12288, 13312)   K-tail fragments, 512 B per tile (lanes 0..31: g = 0 [w_hi, w_hi], g = 1 [w_lo, 0])
//           [13312, 13440)   b1 slices, 2 x 16 floats (pre-scaled)
//           [13440, 27776)   W2 fragments of K-step s-1: (t2 * 2 + p) * 1024; step 7: packed, t2 * 1024; chunk 0: b2[112] + 1/(s1 s2)
constexpr int GRC_PK_OFF = 6144;             // chunk 6: packed K-step fragments of output tiles 0..5 (the slot of hidden tile 13, which does not exist)
constexpr int GRC_PK6_OFF = 12288 + 512;      // chunk 6: ... of output tile 6, rows 96..99 only (16 lanes x 16 B, in the K-tail slot of hidden tile 13)
constexpr int GRC_TAIL_OFF = 12288;
constexpr int GRC_B1_OFF = 13312;
constexpr int GRC_W2_OFF = 13440;
constexpr int GRC_CHUNK_BYTES = 27776;
constexpr int GRC_CHUNK_STRIDE = 28672;  // 28 pieces of 1 KiB in global memory (the last one: 128 B = 8 lanes)

// GR_DEALERS: how many of the workgroup's eight waves issue the LDS-DMA of the weight / table / row streams.  An LDS-DMA instruction
// costs its wave 100-150 cycles of issue; with all eight dealing, both waves of every SIMD stall on their pieces at the top of a
// step and the matrix pipe idles; with waves 0-3 dealing (one per SIMD, the half WITHOUT the static priority), each SIMD's other
// wave multiplies meanwhile.
#ifndef GR_DEALERS
#define GR_DEALERS 4
#endif
// which dealer (0 .. DEAL - 1) a wave is, or -1: waves GR_DEAL_BASE .. GR_DEAL_BASE + DEAL - 1 deal (all eight when DEAL == 8)
#ifndef GR_DEAL_BASE
#define GR_DEAL_BASE 0
#endif
template <int DEAL>
__device__ __forceinline__ int gr_dealer(int wave) {
    if (DEAL >= GR_WAVES) return wave;
    const int d = wave - GR_DEAL_BASE;
    return (d >= 0 && d < DEAL) ? d : -1;
}
template <int DEAL = GR_DEALERS>
__device__ __forceinline__ void grc_issue_chunk(const uint8_t* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
    wave = gr_dealer<DEAL>(wave);
    if (wave < 0) return;
    const uint32_t lb = lds_addr_of(lds_buf);
#pragma unroll
    for (int p = 0; p < (28 + DEAL - 1) / DEAL; p++) {
        const int piece = wave + DEAL * p;  // 27 full pieces + 128 B
        if (piece < 27 || (piece == 27 && lane < 8)) lds_dma16(gchunk + piece * 1024, (uint32_t)lane * 16u, lb + piece * 1024);
    }
}

// first-layer part of a chunk only (W1 fragments, K-tail, b1: 13 440 B = 13 pieces + 128 B): all the folded last layer reads
template <int DEAL = GR_DEALERS>
__device__ __forceinline__ void grc_issue_chunk_w1(const uint8_t* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
    wave = gr_dealer<DEAL>(wave);
    if (wave < 0) return;
    const uint32_t lb = lds_addr_of(lds_buf);
#pragma unroll
    for (int p = 0; p < (14 + DEAL - 1) / DEAL; p++) {
        const int piece = wave + DEAL * p;
        if (piece < 13 || (piece == 13 && lane < 8)) lds_dma16(gchunk + piece * 1024, (uint32_t)lane * 16u, lb + piece * 1024);
    }
}

#define GR_LD(off) (*reinterpret_cast<const uint4_t*>(wb + (off) + lane * 16))
#define GR_SB() __builtin_amdgcn_sched_barrier(0)
// GR_DEFER (compile-time, -DGR_DEFER=true): finish a step's last hidden tile at the top of the NEXT step (gr_step, PEND_IN / PEND_OUT).
// Measured on MI355X: 8.95 ms per launch against 8.70 -- eight more live registers across the barrier cost 16 B of scratch per lane,
// and VALU work at the head of a segment is exactly what MI355X_MICROARCH.md says not to put there.  Off.
#ifndef GR_DEFER
#define GR_DEFER false
#endif
// At equal priority the SIMD's older wave (0-3) wins the arbitration and the younger one (4-7) is the gather's straggler by 20 %
// (2 330 us against 1 950 per wave by phase stamps); with the priority on 4-7 for the whole gather it is the other way round.  So
// waves 4-7 hold it for the first half of their trips: 2 280 against 2 050, wait behind the gather 483 -> 418 us, launch -0.4 %.
#ifndef GR_PRIO_HALF
#define GR_PRIO_HALF true
#endif
#ifndef GR_FLIP
#define GR_FLIP 0
#endif
#ifndef GR_PRIO_QUARTERS
#define GR_PRIO_QUARTERS 3  // waves 4-7 hold the priority for this many quarters of the trips that walk both column tiles (1: +0.7 %, 2: level, 3: -0.3 %, 4: +0.8 % of the launch)
#endif
#ifdef GR_PROF_WALK
#define GR_TACC_DMA 3  /* folded into "epilogue" (unused in this development build's report) */
#define GR_TACC_BAR 3
#else
#define GR_TACC_DMA 5
#define GR_TACC_BAR 4
#endif
#ifndef GR_PRIO_BY_PHASE
#define GR_PRIO_BY_PHASE true
#endif

// One step of the node MLP for TWO column tiles per wave.  The work is a chain of "units" -- two fragments (hi, lo of one
// 16-row weight tile and one K-step) feeding six MFMAs (w_hi x_hi, w_hi x_lo, w_lo x_hi for both column tiles) -- and the
// fragments of unit k+1 are requested before the MFMAs of unit k issue (scheduling barriers pin that order): one LDS round trip
// is always covered by ~120 cycles of matrix pipe, with 16 fragment registers in all.
// KIND 0: step 0 (first layer's hidden tiles 0,1 only) | 1: steps 1..5 (K-step s - 1 of the second layer, then hidden tiles 2s, 2s + 1).
// The layer's last step (hidden tile 12, K-step 5 and the packed K-step of hidden units 192..199) is gr_step_final.
// KIND 4 / 5: the LAST layer with the readout folded through its second linear layer (head_u, below): hidden tiles only (two / one),
// each ReLU'd tile is dotted with its slice of u instead of being split for a second layer that is never computed.
// PEND_OUT: the step's LAST hidden tile is left un-finished in `pend` (its accumulators); PEND_IN: the previous step did that, and this
// step finishes it AFTER requesting its own first fragments -- the ReLU + split VALU work then covers the LDS round trip every step
// otherwise starts with (both waves of a SIMD sit behind the step barrier waiting ~200 clocks for their first fragment).
struct GrNoHook { __device__ __forceinline__ void operator()() const {} };
// HOOK: work of the caller's run in the MIDDLE of the step -- between its two hidden tiles (KIND 4), behind the first unit of its only one
// (KIND 5) -- where the SIMD's two waves arrive at different times: one wave's VALU / memory instructions then issue beside its partner's
// MFMAs.  At the step's ends, where both waves stand at the barrier together, the same instructions stop the matrix pipe.
template <int KIND, bool PEND_IN = false, bool PEND_OUT = false, class HOOK = GrNoHook>
__device__ __forceinline__ void gr_step(const char* wb, int lane, int g, const uint4_t (&in_hi)[2][3], const uint4_t (&in_lo)[2][3],
                                        const uint4_t (&in_tb)[2], uint4_t (&hb_hi)[2], uint4_t (&hb_lo)[2], float4_t (&acc2)[2][GS_T2],
                                        float& vmax, float4_t (&pend)[2], const float* u_step = nullptr, float* dot = nullptr, int wave = 0,
                                        HOOK hook = HOOK()) {
    static_assert(KIND == 0 || KIND == 1 || KIND == 4 || KIND == 5, "step kinds");
    constexpr bool DO2 = KIND == 1, DO1 = true, DOT = KIND >= 4;
    // GR_FLIP: the static priority of waves 4-7 changes hands in the middle of the step's unit chain (and back at its end)
#define GR_PRIO_FLIP(TO_OLD) if (GR_FLIP) { if ((wave >= 4) != (TO_OLD)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    constexpr int NTL = KIND == 5 ? 1 : 2;
    uint4_t f0[2], f1[2];  // fragment double buffer: f0 = even units, f1 = odd units
#define GR_U2_LOAD(F, T) F[0] = GR_LD(GRC_W2_OFF + (2 * (T)) * 1024); F[1] = GR_LD(GRC_W2_OFF + (2 * (T) + 1) * 1024);
#define GR_U2_MFMA(F, T)                                          \
    acc2[0][T] = GS_MFMA16(F[0], hb_hi[0], acc2[0][T]);           \
    acc2[1][T] = GS_MFMA16(F[0], hb_hi[1], acc2[1][T]);           \
    acc2[0][T] = GS_MFMA16(F[0], hb_lo[0], acc2[0][T]);           \
    acc2[1][T] = GS_MFMA16(F[0], hb_lo[1], acc2[1][T]);           \
    acc2[0][T] = GS_MFMA16(F[1], hb_hi[0], acc2[0][T]);           \
    acc2[1][T] = GS_MFMA16(F[1], hb_hi[1], acc2[1][T]);
#define GR_U1_LOAD(F, TL, KS) F[0] = GR_LD((TL) * 6144 + (2 * (KS)) * 1024); F[1] = GR_LD((TL) * 6144 + (2 * (KS) + 1) * 1024);
#define GR_U1_MFMA(F, TL, KS)                                             \
    acc1[TL][0] = GS_MFMA16(F[0], in_hi[0][KS], acc1[TL][0]);             \
    acc1[TL][1] = GS_MFMA16(F[0], in_hi[1][KS], acc1[TL][1]);             \
    acc1[TL][0] = GS_MFMA16(F[0], in_lo[0][KS], acc1[TL][0]);             \
    acc1[TL][1] = GS_MFMA16(F[0], in_lo[1][KS], acc1[TL][1]);             \
    acc1[TL][0] = GS_MFMA16(F[1], in_hi[0][KS], acc1[TL][0]);             \
    acc1[TL][1] = GS_MFMA16(F[1], in_hi[1][KS], acc1[TL][1]);
    float4_t acc1[2];  // one hidden tile at a time: its ReLU + split then overlap the next tile's MFMAs
#define GR_U1_MFMA1(F, KS)                                        \
    acc1[0] = GS_MFMA16(F[0], in_hi[0][KS], acc1[0]);             \
    acc1[1] = GS_MFMA16(F[0], in_hi[1][KS], acc1[1]);             \
    acc1[0] = GS_MFMA16(F[0], in_lo[0][KS], acc1[0]);             \
    acc1[1] = GS_MFMA16(F[0], in_lo[1][KS], acc1[1]);             \
    acc1[0] = GS_MFMA16(F[1], in_hi[0][KS], acc1[0]);             \
    acc1[1] = GS_MFMA16(F[1], in_hi[1][KS], acc1[1]);
#define GR_TAIL_LOAD(F, TL) F[0] = *reinterpret_cast<const uint4_t*>(wb + GRC_TAIL_OFF + (TL) * 512 + (lane & 31) * 16);
#define GR_BIAS(TL) acc1[0] = *reinterpret_cast<const float4_t*>(wb + GRC_B1_OFF + (TL) * 64 + g * 16);
    // ReLU, range watch, split of hidden tile TL into its half (.xy / .zw) of the next step's B operands
#define GR_FINISH(TL) GR_FINISH_FROM(acc1, TL)
#define GR_FINISH_FROM(SRC, TL)                                                                           \
    _Pragma("unroll") for (int nt = 0; nt < 2; nt++) {                                                    \
        float4_t r = SRC[nt];                                                                             \
        r.x = gs_relu(r.x); r.y = gs_relu(r.y); r.z = gs_relu(r.z); r.w = gs_relu(r.w);                   \
        if constexpr (DOT) {                                                                              \
            const float4 uu = *reinterpret_cast<const float4*>(u_step + 16 * (TL) + 4 * g);              \
            dot[nt] += r.x * uu.x; dot[nt] += r.y * uu.y; dot[nt] += r.z * uu.z; dot[nt] += r.w * uu.w;   \
            continue;                                                                                     \
        }                                                                                                 \
        vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, r.x), r.y);                                          \
        vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, r.z), r.w);                                          \
        if ((TL) == 0) { GS_SPLIT2(r.x, r.y, hb_hi[nt].x, hb_lo[nt].x); GS_SPLIT2(r.z, r.w, hb_hi[nt].y, hb_lo[nt].y); } \
        else { GS_SPLIT2(r.x, r.y, hb_hi[nt].z, hb_lo[nt].z); GS_SPLIT2(r.z, r.w, hb_hi[nt].w, hb_lo[nt].w); }           \
    }
    if constexpr (DO2) {  // second linear layer, K-step s-1: acc2[nt][t] += W2 frag(t) x relu(hidden) of the previous step
        GR_U2_LOAD(f0, 0)
        GR_SB();
        GR_U2_LOAD(f1, 1) GR_SB();
        if constexpr (PEND_IN) { GR_FINISH_FROM(pend, 1) GR_SB(); }  // the previous step's second hidden tile -> hb_*.zw, under the two requests
        GR_U2_MFMA(f0, 0) GR_SB();
        GR_U2_LOAD(f0, 2) GR_SB(); GR_U2_MFMA(f1, 1) GR_SB();
        GR_U2_LOAD(f1, 3) GR_SB(); GR_U2_MFMA(f0, 2) GR_SB();
        GR_U2_LOAD(f0, 4) GR_SB(); GR_U2_MFMA(f1, 3) GR_SB();
        GR_U2_LOAD(f1, 5) GR_SB(); GR_U2_MFMA(f0, 4) GR_SB();
        GR_U2_LOAD(f0, 6) GR_SB(); GR_U2_MFMA(f1, 5) GR_SB();
        // first unit of the first layer (and its bias) requested under the last unit of the second
        GR_U1_LOAD(f1, 0, 0)
        GR_BIAS(0)
        GR_SB();
        GR_U2_MFMA(f0, 6)
        GR_SB();
        GR_PRIO_FLIP(true)
    }
    if constexpr (DO1) {  // first linear layer: hidden tiles 2s (, 2s+1) = b1 + W1 a, K = 96 as three K-steps + the packed tail
        if constexpr (!DO2) {
            GR_U1_LOAD(f1, 0, 0)
            GR_BIAS(0)
            GR_SB();
        }
        acc1[1] = acc1[0];
        GR_U1_LOAD(f0, 0, 1) GR_SB(); GR_U1_MFMA1(f1, 0) GR_SB();
        GR_U1_LOAD(f1, 0, 2) GR_SB(); GR_U1_MFMA1(f0, 1) GR_SB();
        GR_TAIL_LOAD(f0, 0) GR_SB(); GR_U1_MFMA1(f1, 2) GR_SB();
        if constexpr (NTL == 2) { GR_U1_LOAD(f1, 1, 0) GR_SB(); }
        acc1[0] = GS_MFMA16(f0[0], in_tb[0], acc1[0]);
        acc1[1] = GS_MFMA16(f0[0], in_tb[1], acc1[1]);
        if constexpr (NTL == 2) {
            if constexpr (!DO2) { GR_PRIO_FLIP(true) }
            GR_FINISH(0)
            GR_BIAS(1)
            if constexpr (DOT) { GR_SB(); hook(); GR_SB(); }
            acc1[1] = acc1[0];
            GR_U1_LOAD(f0, 1, 1) GR_SB(); GR_U1_MFMA1(f1, 0) GR_SB();
            GR_U1_LOAD(f1, 1, 2) GR_SB(); GR_U1_MFMA1(f0, 1) GR_SB();
            GR_TAIL_LOAD(f0, 1) GR_SB(); GR_U1_MFMA1(f1, 2) GR_SB();
            acc1[0] = GS_MFMA16(f0[0], in_tb[0], acc1[0]);
            acc1[1] = GS_MFMA16(f0[0], in_tb[1], acc1[1]);
            if constexpr (PEND_OUT) { pend[0] = acc1[0]; pend[1] = acc1[1]; }
            else { GR_FINISH(1) }
        } else {
            static_assert(DOT, "a one-tile step is the folded layer's last (KIND 5)");
            GR_SB(); hook(); GR_SB();  // (KIND 5: one hidden tile; behind its MFMAs, in front of its ReLU + dot)
            GR_FINISH(0)
        }
        asm volatile("" : "+v"(vmax));
    }
    if constexpr (DO2 || NTL == 2) { GR_PRIO_FLIP(false) }
#undef GR_PRIO_FLIP
#undef GR_U1_MFMA1
#undef GR_TAIL_LOAD
#undef GR_BIAS
#undef GR_FINISH
#undef GR_FINISH_FROM
#undef GR_U2_LOAD
#undef GR_U2_MFMA
#undef GR_U1_LOAD
#undef GR_U1_MFMA
}
// The LAST step of a layer that computes its second linear layer (steps 6 and 7 of the eight-step schedule as ONE step; chunk 6 of the
// stream carries both, GR chunks above).  Hidden tile 12 first (the half tile: hidden units 192..199), then per OUTPUT tile t the unit
// of K-step 5 and the packed MFMA of K-step 6 back to back -- the same order of accumulation as in two steps, so the same bits -- which
// makes acc2[.][t] FINAL after unit t: with STORE its ReLU'd rows go back into the tile (in place: nobody reads the old rows during the
// MLP) one unit later, under the MFMAs of the following output tiles.  Stored after the last step the tile's 102 KB of rows cost
// 0.28 ms per launch at the ~80 B per clock of ds_write_b128 with the matrix pipe idle (measured by leaving them out), and the
// eighth step a barrier and a fragment round trip of its own for 14 MFMAs.
template <bool STORE>
__device__ __forceinline__ void gr_step_final(const char* wb, int lane, int g, const uint4_t (&in_hi)[2][3], const uint4_t (&in_lo)[2][3],
                                              const uint4_t (&in_tb)[2], const uint4_t (&hb_hi)[2], const uint4_t (&hb_lo)[2],
                                              float4_t (&acc2)[2][GS_T2], float& vmax, float oscale, float* rw0, float* rw1) {
    uint4_t f0[2], f1[2], pa, hp[2];
    float4_t acc1[2];
#define GRF_U2_LOAD(F, T) F[0] = GR_LD(GRC_W2_OFF + (2 * (T)) * 1024); F[1] = GR_LD(GRC_W2_OFF + (2 * (T) + 1) * 1024);
#define GRF_U2_MFMA(F, T)                                         \
    acc2[0][T] = GS_MFMA16(F[0], hb_hi[0], acc2[0][T]);           \
    acc2[1][T] = GS_MFMA16(F[0], hb_hi[1], acc2[1][T]);           \
    acc2[0][T] = GS_MFMA16(F[0], hb_lo[0], acc2[0][T]);           \
    acc2[1][T] = GS_MFMA16(F[0], hb_lo[1], acc2[1][T]);           \
    acc2[0][T] = GS_MFMA16(F[1], hb_hi[0], acc2[0][T]);           \
    acc2[1][T] = GS_MFMA16(F[1], hb_hi[1], acc2[1][T]);
#define GRF_U1_LOAD(F, KS) F[0] = GR_LD((2 * (KS)) * 1024); F[1] = GR_LD((2 * (KS) + 1) * 1024);
#define GRF_U1_MFMA(F, KS)                                        \
    acc1[0] = GS_MFMA16(F[0], in_hi[0][KS], acc1[0]);             \
    acc1[1] = GS_MFMA16(F[0], in_hi[1][KS], acc1[1]);             \
    acc1[0] = GS_MFMA16(F[0], in_lo[0][KS], acc1[0]);             \
    acc1[1] = GS_MFMA16(F[0], in_lo[1][KS], acc1[1]);             \
    acc1[0] = GS_MFMA16(F[1], in_hi[0][KS], acc1[0]);             \
    acc1[1] = GS_MFMA16(F[1], in_hi[1][KS], acc1[1]);
    // packed fragment of output tile T: tiles 0..5 in the slot of the (non-existent) hidden tile 13, tile 6 -- outputs 96..99: four real
    // rows -- as 16 lanes x 16 B behind the K-tail fragments
#define GRF_PK_LOAD(P, T)                                                                                                          \
    if constexpr ((T) < 6) P = GR_LD(GRC_PK_OFF + (T) * 1024);                                                                     \
    else P = (lane & 15) < 4 ? *reinterpret_cast<const uint4_t*>(wb + GRC_PK6_OFF + ((((lane >> 4) << 2) | (lane & 3)) << 4)) : (uint4_t){0u, 0u, 0u, 0u};
#define GRF_PK_MFMA(P, T)                                  \
    acc2[0][T] = GS_MFMA16(P, hp[0], acc2[0][T]);          \
    acc2[1][T] = GS_MFMA16(P, hp[1], acc2[1][T]);
#define GRF_STORE(T)                                                                                                  \
    if constexpr (STORE) {                                                                                            \
        constexpr int col0_ = 16 * (T);                                                                               \
        if (col0_ + 4 * g < GS_D) {                                                                                   \
            float4_t r0_ = acc2[0][T] * oscale, r1_ = acc2[1][T] * oscale;                                            \
            r0_.x = gs_relu(r0_.x); r0_.y = gs_relu(r0_.y); r0_.z = gs_relu(r0_.z); r0_.w = gs_relu(r0_.w);           \
            r1_.x = gs_relu(r1_.x); r1_.y = gs_relu(r1_.y); r1_.z = gs_relu(r1_.z); r1_.w = gs_relu(r1_.w);           \
            *reinterpret_cast<float4_t*>(rw0 + col0_ + 4 * g) = r0_;                                                  \
            *reinterpret_cast<float4_t*>(rw1 + col0_ + 4 * g) = r1_;                                                  \
        }                                                                                                             \
    }
    // ---- hidden tile 12 = b1 + W1 a (three K-steps + the packed K tail)
    GRF_U1_LOAD(f1, 0)
    acc1[0] = *reinterpret_cast<const float4_t*>(wb + GRC_B1_OFF + g * 16);
    GR_SB();
    acc1[1] = acc1[0];
    GRF_U1_LOAD(f0, 1) GR_SB(); GRF_U1_MFMA(f1, 0) GR_SB();
    GRF_U1_LOAD(f1, 2) GR_SB(); GRF_U1_MFMA(f0, 1) GR_SB();
    f0[0] = *reinterpret_cast<const uint4_t*>(wb + GRC_TAIL_OFF + (lane & 31) * 16); GR_SB();
    GRF_U1_MFMA(f1, 2) GR_SB();
    GRF_U2_LOAD(f1, 0) GR_SB();  // the first unit of the second linear layer, requested under the tail
    acc1[0] = GS_MFMA16(f0[0], in_tb[0], acc1[0]);
    acc1[1] = GS_MFMA16(f0[0], in_tb[1], acc1[1]);
    GR_SB();
    // ---- K-step 5 unit 0 needs nothing of tile 12: its MFMAs cover the ReLU + split + pack of the tile
    GRF_U2_LOAD(f0, 1) GRF_PK_LOAD(pa, 0) GR_SB();
    GRF_U2_MFMA(f1, 0) GR_SB();
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        float4_t r = acc1[nt];
        r.x = gs_relu(r.x); r.y = gs_relu(r.y); r.z = gs_relu(r.z); r.w = gs_relu(r.w);
        vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, r.x), r.y);
        vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, r.z), r.w);
        uint32_t hx, hy, lx, ly;  // lanes g = 0 / 1 hold hidden units 192..195 / 196..199
        GS_SPLIT2(r.x, r.y, hx, lx);
        GS_SPLIT2(r.z, r.w, hy, ly);
        // the packed operand:  g = 0, 1: [hi, lo] (own)   g = 2, 3: [hi of lane - 32, 0]
        const uint32_t ox = __shfl(hx, lane & 31, 64), oy = __shfl(hy, lane & 31, 64);
        hp[nt] = g < 2 ? (uint4_t){hx, hy, lx, ly} : (uint4_t){ox, oy, 0u, 0u};
    }
    asm volatile("" : "+v"(vmax));
    GR_SB();
    GRF_PK_MFMA(pa, 0) GR_SB();
    // (one packed-fragment register quad: the next tile's is requested behind the MFMAs that read this one's)
    GRF_U2_LOAD(f1, 2) GRF_PK_LOAD(pa, 1) GR_SB(); GRF_U2_MFMA(f0, 1) GRF_PK_MFMA(pa, 1) GR_SB(); GRF_STORE(0) GR_SB();
    GRF_U2_LOAD(f0, 3) GRF_PK_LOAD(pa, 2) GR_SB(); GRF_U2_MFMA(f1, 2) GRF_PK_MFMA(pa, 2) GR_SB(); GRF_STORE(1) GR_SB();
    GRF_U2_LOAD(f1, 4) GRF_PK_LOAD(pa, 3) GR_SB(); GRF_U2_MFMA(f0, 3) GRF_PK_MFMA(pa, 3) GR_SB(); GRF_STORE(2) GR_SB();
    GRF_U2_LOAD(f0, 5) GRF_PK_LOAD(pa, 4) GR_SB(); GRF_U2_MFMA(f1, 4) GRF_PK_MFMA(pa, 4) GR_SB(); GRF_STORE(3) GR_SB();
    GRF_U2_LOAD(f1, 6) GRF_PK_LOAD(pa, 5) GR_SB(); GRF_U2_MFMA(f0, 5) GRF_PK_MFMA(pa, 5) GR_SB(); GRF_STORE(4) GR_SB();
    GRF_PK_LOAD(pa, 6) GR_SB();                    GRF_U2_MFMA(f1, 6) GRF_PK_MFMA(pa, 6) GR_SB(); GRF_STORE(5) GR_SB();
    GRF_STORE(6)
#undef GRF_U2_LOAD
#undef GRF_U2_MFMA
#undef GRF_U1_LOAD
#undef GRF_U1_MFMA
#undef GRF_PK_LOAD
#undef GRF_PK_MFMA
#undef GRF_STORE
}
#undef GR_LD

struct GrTile { int t0, rows, g0, g1; };

__device__ __forceinline__ void gr_issue_ecomb(const float* __restrict__ ecomb, char* lds_buf, int wave, int lane) {
    wave = gr_dealer<GR_DEALERS>(wave);
    if (wave < 0) return;
#pragma unroll
    for (int r = 0; r < (24 + GR_DEALERS - 1) / GR_DEALERS; r++) {
        const int piece = wave + GR_DEALERS * r;
        if (piece < 23 || (piece == 23 && lane < 28)) lds_dma16(reinterpret_cast<const char*>(ecomb) + piece * 1024, (uint32_t)lane * 16u, lds_addr_of(lds_buf) + piece * 1024);
    }
}
// pieces [13 part, 13 part + 13) of the tile's rows (<= 100 pieces of 1 KiB; a piece may run past the tile's last row: the
// rows array has 4 KiB of slack and the surplus lands in unused rows of the buffer)
__device__ __forceinline__ void gr_issue_rows(const float* __restrict__ h0, char* s_rows, const GrTile& t, int part, int wave, int lane) {
    wave = gr_dealer<GR_DEALERS>(wave);
    if (wave < 0) return;
    const int np = (t.rows * (GS_D * 4) + 1023) >> 10;
#pragma unroll
    for (int r = 0; r < (13 + GR_DEALERS - 1) / GR_DEALERS; r++) {
        const int piece = 13 * part + wave + GR_DEALERS * r;
        if (piece < 13 * part + 13 && piece < np)
            lds_dma16(reinterpret_cast<const char*>(h0) + (size_t)t.t0 * (GS_D * 4) + (size_t)piece * 1024, (uint32_t)lane * 16u, lds_addr_of(s_rows) + piece * 1024);
    }
}

// tstride 1: consecutive tiles (entry t+1 = end of tile t); 2: (start, end) pairs -- GraphTiles::big_row / big_graph
__device__ __forceinline__ GrTile gr_load_tile(const int* __restrict__ tile_row, const int* __restrict__ tile_graph, int tile, int n_tiles, int tstride = 1) {
    GrTile t{0, 0, 0, 0};
    if (tile < n_tiles) {
        t.t0 = tile_row[tile * tstride];
        t.rows = tile_row[tile * tstride + 1] - t.t0;
        t.g0 = tile_graph[tile * tstride];
        t.g1 = tile_graph[tile * tstride + 1];
        if (t.rows > GR_ROWS) t.rows = GR_ROWS;
    }
    return t;
}

// Per-tile descriptor, built by gin_tile_prep_kernel after the CSR and brought into LDS by DMA (no registers, no VALU in the
// resident kernel):   [0, 2560)  u16 edge words of the tile's CSR slice: (row inside the tile << 6) | edge code
//                     [2560, 3088) u16 row offsets into that slice (rows + 1 used)      [3088, 3344) u8 column owner table (below)
constexpr int GR_DESC_RP = GR_EDGES * 2;
constexpr int GR_DESC_PERM = GR_DESC_RP + 528;
constexpr int GR_DESC_BYTES = 3584;  // 3.5 pieces of 1 KiB
constexpr int GR_PLANES_FLOATS = 6 * 4 * 64 * 4, GR_PLANES_OFF = 5 * EDGE_COMBOS * GS_D;  // behind the five layers' row-major tables
constexpr unsigned GR_NO_EDGE = (unsigned)GR_ROWS << 6;  // edge word of "no in-edge left": source row GR_ROWS (all -1e30), code 0
constexpr int GR_HUB_DEG = 8;        // HUBS kernels: rows with more in-edges than this are walked by their whole 16-lane group

// Column owner table: which row of the tile each MFMA column (wave, column tile, lane) owns.  The gather walks the in-edges of
// the 16 rows of a column tile in lockstep, so a column tile costs as many trips as its LONGEST row: rows are dealt to column
// tiles in order of decreasing in-degree (stable: ties in row order), and the 16 column tiles to the 8 waves as (k, 15 - k), so
// that every wave gets a long and a short one.  Placement never changes a row's arithmetic (MFMA columns are independent, a
// row's in-edges are summed in CSR order whoever owns it): results are identical for any permutation.  Rows beyond the tile's
// last sort last.   slot = wave * 32 + nt * 16 + j  ->  row inside the tile (0..255)
// Column owner table of a tile (desc + GR_DESC_PERM) from every row's in-degree; thread r = row r of a 256-thread workgroup.
// column tile kt (0 = the longest rows) -> the wave that owns it and which of the wave's two it is.  Waves w and w + 4 share a SIMD
// (and with it VALU / LDS issue): GR_SIMD_DEAL deals the sixteen tiles so that every SIMD gets the same share of long and short ones --
// SIMD w: {w, 7 - w, 8 + w, 15 - w}, wave w the outer two, wave w + 4 the inner two; 0 = (kt, 15 - kt) per wave, which gives the
// SIMD of waves 0 and 4 the tiles {0, 4, 11, 15} and a fifth more trips than the SIMD of waves 3 and 7.
#ifndef GR_SIMD_DEAL
#define GR_SIMD_DEAL 1
#endif
__device__ __forceinline__ void gr_tile_owner(int kt, int& wave, int& nt) {
    if (GR_SIMD_DEAL) {
        nt = kt >> 3;
        if (kt < 4) wave = kt;
        else if (kt < 8) wave = 11 - kt;   // 4..7 -> waves 7..4
        else if (kt < 12) wave = kt - 4;   // 8..11 -> waves 4..7
        else wave = 15 - kt;               // 12..15 -> waves 3..0
    } else {
        wave = kt < 8 ? kt : 15 - kt;
        nt = kt < 8 ? 0 : 1;
    }
}
__device__ __forceinline__ void gr_write_column_order(uint8_t* d, int r, int rows, int deg, int order) {
    constexpr int NKEY = 18;
    __shared__ int s_cnt[4][NKEY];
    const int lane = r & 63, wv = r >> 6;
    int key = NKEY - 1;  // absent row
    if (r < rows) key = 16 - (deg < 16 ? deg : 16);  // ascending key = descending in-degree
    if (order == 1) {  // development switch: natural order (column tile k = rows 16k..16k+15)
        d[GR_DESC_PERM + r] = (uint8_t)r;
        return;
    }
    if (order == 2) {
        // Bank-aware order: a column tile takes ONE row of every residue class r mod 16 -- the k-th longest of each class -- so the
        // 16 rows a gather instruction reads start in 16 different bank groups (row stride 100 dwords = 9 bank groups mod 16), and so do
        // their self rows and the epilogue's stores; neighbours of rows with distinct residues (atoms are numbered along chains) mostly
        // have distinct residues too.  Degrees within a column tile stay close (k-th of 16 per class).
        __shared__ int s_key[GR_ROWS];
        s_key[r] = key;
        __syncthreads();
        const int rho = r & 15;
        int rank = 0;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int r2 = rho + 16 * m, k2 = s_key[r2];
            rank += (k2 < key) | ((k2 == key) & (r2 < r));
        }
        const int wave2 = rank < 8 ? rank : 15 - rank, nt2 = rank < 8 ? 0 : 1;
        d[GR_DESC_PERM + wave2 * 32 + nt2 * 16 + rho] = (uint8_t)r;
        return;
    }
    if (order == 3) {
        // Hub order (GIN-VN: one virtual node of in-degree n per graph): rows of more than GR_HUB_DEG in-edges are walked by all 16
        // lanes of their column tile together (gr_layer), so they are dealt round-robin over the 16 column tiles -- hub i to column
        // tile i mod 16, lane i / 16 -- and the other rows fill the remaining lanes in order of decreasing in-degree as above.
        __shared__ int s_hub[4], s_cnt2[4][NKEY];
        const bool hub = r < rows && deg > GR_HUB_DEG;
        const unsigned long long hm = __ballot(hub);
        if (lane == 0) s_hub[wv] = __popcll(hm);
        const int k2 = hub ? NKEY : key;  // hubs leave the degree classes
        int mine2 = 0;
        for (int k = 0; k < NKEY; k++) {
            const unsigned long long m = __ballot(k2 == k);
            if (lane == 0) s_cnt2[wv][k] = __popcll(m);
            if (k2 == k) mine2 = __popcll(m & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        const int H = s_hub[0] + s_hub[1] + s_hub[2] + s_hub[3];
        int kt, jj;
        if (hub) {
            int hr = __popcll(hm & ((1ull << lane) - 1ull));
            for (int w = 0; w < wv; w++) hr += s_hub[w];
            kt = hr & 15; jj = hr >> 4;
        } else {
            int p = mine2;
            for (int k = 0; k < NKEY; k++)
                for (int w = 0; w < 4; w++) {
                    const int c = s_cnt2[w][k];
                    if (k < key || (k == key && w < wv)) p += c;
                }
            kt = 0; jj = 0;
            for (int t = 0; t < 16; t++) {  // column tile t has 16 - hubs_in(t) free lanes
                const int hin = (H >> 4) + (t < (H & 15) ? 1 : 0), cap = 16 - hin;
                if (p < cap) { kt = t; jj = hin + p; break; }
                p -= cap;
            }
        }
        const int wave3 = kt < 8 ? kt : 15 - kt, nt3 = kt < 8 ? 0 : 1;  // (the hubs are dealt round-robin: every tile as long as every other; gr_tile_owner's deal measured +0.3 % here)
        d[GR_DESC_PERM + wave3 * 32 + nt3 * 16 + jj] = (uint8_t)r;
        return;
    }
    int below = 0, mine = 0;
    for (int k = 0; k < NKEY; k++) {
        const unsigned long long m = __ballot(key == k);
        if (lane == 0) s_cnt[wv][k] = __popcll(m);
        if (key == k) mine = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    for (int k = 0; k < NKEY; k++)
        for (int w = 0; w < 4; w++) {
            const int c = s_cnt[w][k];
            if (k < key || (k == key && w < wv)) below += c;
        }
    const int pos = below + mine;          // rank in (key, row) order, 0..255
    const int kt = pos >> 4, j = pos & 15;  // column tile kt (0 = longest rows), lane j
    int wave, nt;
    gr_tile_owner(kt, wave, nt);
    d[GR_DESC_PERM + wave * 32 + nt * 16 + j] = (uint8_t)r;
}

__global__ __launch_bounds__(256) void gin_tile_prep_kernel(const int* __restrict__ row_ptr, const int* __restrict__ src,
                                                            const uint8_t* __restrict__ ecode, const int* __restrict__ tile_row,
                                                            uint8_t* __restrict__ desc, int n_tiles, int order, int tstride) {
    const int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const int t0 = tile_row[tile * tstride];
    int rows = tile_row[tile * tstride + 1] - t0;
    if (rows > GR_ROWS) rows = GR_ROWS;
    const int e0 = row_ptr[t0];
    int ne = row_ptr[t0 + rows] - e0;
    if (ne > GR_EDGES) ne = GR_EDGES;  // cannot happen for a validated batch (the host packed by edge count); never overrun LDS
    uint8_t* d = desc + (size_t)tile * GR_DESC_BYTES;
    uint16_t* d_edge = reinterpret_cast<uint16_t*>(d);
    uint16_t* d_rp = reinterpret_cast<uint16_t*>(d + GR_DESC_RP);
    const int r = threadIdx.x;
    for (int i = r; i < ne; i += 256)
        d_edge[i] = (uint16_t)((((unsigned)(src[e0 + i] - t0) & 0xFFu) << 6) | ((unsigned)ecode[e0 + i] & 63u));
    int deg = 0;
    {
        const int lo = r <= rows ? row_ptr[t0 + r] - e0 : ne;
        const int lo_c = lo < 0 ? 0 : (lo > ne ? ne : lo);
        d_rp[r] = (uint16_t)lo_c;
        if (r == 255) d_rp[256] = (uint16_t)ne;
        if (r < rows) {
            const int hi = row_ptr[t0 + r + 1] - e0;
            deg = (hi > ne ? ne : hi) - lo_c;
            if (deg < 0) deg = 0;
        }
    }
    gr_write_column_order(d, r, rows, deg, order);
}

// ---------------------------------------------------------------- the tile's descriptor straight from the caller's arrays
// gin_tile_build_kernel = load_graph (GIN/src/load_inputs.cc:87-172) + the index part of the atom encoder (:174-220) for ONE tile of
// whole graphs, with no global CSR in between: what build_csr_graph_kernel + gin_tile_prep_kernel + the encoder's feature validation
// did in three passes over HBM.  One 256-thread workgroup per tile:
//   * the tile's raw edges (a contiguous slice of edge_list / edge_attr: graphs are stored in batch order) are validated, turned
//     into (source row, destination row, edge code) and counting-sorted by destination in LDS; inside a row they are rank-sorted by
//     (source row, input index) -- unique keys, so the order is the CSR's (graph_build.hip) whatever order the LDS atomics landed in;
//   * the descriptor is written in gin_tile_prep_kernel's format (edge words, row offsets, column owner table);
//   * every node's nine features are validated and turned into three row numbers of the pre-combined encoder table (below), 4 B per
//     node: what the resident kernel's tile loader reads instead of a 400 B row of h_0.
// Pre-combined encoder table (GinModel::set_weights; rows of 100 floats; GRB_* = first row of each part; 2 060 rows = 824 KB, L2-resident):
//   T01[f0][f1] = E0[f0] + E1[f1]   T234[f2][f3][f4] = E2[f2] + (E3[f3] + E4[f4])   T5678[f5][f6][f7][f8] = ((E5 + E6) + E7) + E8
//   h_0 = (T01 + T234) + T5678: the reference's nine-term sum (load_inputs.cc:207-214) re-associated -- it differs from the
//   sequential order by fp32 rounding only (<= 4 ulp of a sum of magnitude ~1; the stated parity tolerance is 1e-4).
//   Three rows per node, not four or nine: what limits the loader is the bytes it pulls through the CU's vector-memory return path
//   (64 B per clock: four 400-B rows per node were 59 KB per MLP step of the folded layer, beside the step's own weight DMA).
// A node's row numbers, local to their parts, in one word: T01 row | T234 row << 9 | T5678 row << 20.
constexpr int GRB_T01 = 0, GRB_T234 = 476, GRB_T5678 = 1916, GRB_ROWS = 2060;

__device__ __forceinline__ int gr_wave_inclusive_scan(int x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

__global__ __launch_bounds__(256) void gin_tile_build_kernel(BatchView b, const int* __restrict__ tile_row, const int* __restrict__ tile_graph,
                                                             uint8_t* __restrict__ desc, uint32_t* __restrict__ enc_idx, int n_tiles, int order,
                                                             int* __restrict__ err, const int* __restrict__ list) {
    constexpr int EPT = GR_EDGES / 256;  // edges per thread
    __shared__ int s_eoff[GR_ROWS + 1], s_noff[GR_ROWS + 1];  // edge / row offsets of the tile's graphs, relative to the tile
    __shared__ int s_ebase[GR_ROWS + 1], s_nbase[GR_ROWS + 1];  // + these = the batch's edge / node behind a tile-local edge / row of that graph
    __shared__ int s_cnt[GR_ROWS + 1], s_cur[GR_ROWS];
    __shared__ unsigned s_bucket[GR_EDGES];
    __shared__ int s_feat[GR_ROWS * ND_FEATURE];
    __shared__ int s_wtot[4];
    const int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const int r = threadIdx.x, lane = r & 63, wv = r >> 6;
    const int t0 = tile_row[tile];
    int rows = tile_row[tile + 1] - t0;
    if (rows > GR_ROWS) rows = GR_ROWS;
    const int g0 = tile_graph[tile];
    int ng = tile_graph[tile + 1] - g0;
    if (ng > GR_ROWS) ng = GR_ROWS;  // every graph has at least one node, so a validated tile never has more graphs than rows
    int ne;
    if (list == nullptr) {
        const int e0 = b.edge_off[g0];
        ne = b.edge_off[g0 + ng] - e0;
        for (int i = r; i <= ng; i += 256) {
            s_eoff[i] = b.edge_off[g0 + i] - e0;
            s_noff[i] = b.node_off[g0 + i] - t0;
            s_ebase[i] = e0;
            s_nbase[i] = t0;
        }
    } else {  // a LIST of graphs (bin-packed tiles): running sums of their counts (thread = graph, ng <= 256)
        const int gph = r < ng ? list[g0 + r] : 0;
        const int cn = r < ng ? b.nums_of_nodes[gph] : 0, ce = r < ng ? b.nums_of_edges[gph] : 0;
        const int in_n = gr_wave_inclusive_scan(cn, lane), in_e = gr_wave_inclusive_scan(ce, lane);
        __shared__ int s_wn[4], s_we[4];
        if (lane == 63) { s_wn[wv] = in_n; s_we[wv] = in_e; }
        __syncthreads();
        int sn = in_n - cn, se = in_e - ce;
        for (int w = 0; w < wv; w++) { sn += s_wn[w]; se += s_we[w]; }
        if (r < ng) {
            s_noff[r] = sn; s_eoff[r] = se;
            s_nbase[r] = b.node_off[gph] - sn;
            s_ebase[r] = b.edge_off[gph] - se;
        }
        ne = s_we[0] + s_we[1] + s_we[2] + s_we[3];
        if (r == 0) { s_noff[ng] = s_wn[0] + s_wn[1] + s_wn[2] + s_wn[3]; s_eoff[ng] = ne; }
    }
    if (ne > GR_EDGES) ne = GR_EDGES;  // cannot happen for a batch packed by flowgnn_set_batch; never overrun LDS
    s_cnt[r] = 0;
    s_cur[r] = 0;
    if (r == 0) s_cnt[GR_ROWS] = 0;
    if (list == nullptr) {
#pragma unroll
        for (int p = 0; p < ND_FEATURE; p++) {  // the tile's node features, coalesced
            const int i = r + 256 * p;
            s_feat[i] = i < rows * ND_FEATURE ? b.node_feature[(size_t)t0 * ND_FEATURE + i] : 0;
        }
    }
    __syncthreads();
    if (list != nullptr && r < rows) {  // (a list's rows are scattered over the batch: every row finds its graph, then its node)
        int lo = 0, hi = ng - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_noff[mid] <= r) lo = mid; else hi = mid - 1;
        }
        const int* nf = b.node_feature + (size_t)(s_nbase[lo] + r) * ND_FEATURE;
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) s_feat[r * ND_FEATURE + k] = nf[k];
    }
    unsigned ekey[EPT];  // (source row << 17) | (edge index inside the tile << 6) | edge code
    int edst[EPT];       // destination row, -1 = no edge
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int i = r + 256 * k;
        edst[k] = -1;
        ekey[k] = 0;
        if (i < ne) {
            int lo = 0, hi = ng - 1;  // the graph of edge i: the last one whose first edge is <= i
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_eoff[mid] <= i) lo = mid; else hi = mid - 1;
            }
            const size_t ge = (size_t)(s_ebase[lo] + i);  // the edge's place in the caller's arrays
            const int2 uv = reinterpret_cast<const int2*>(b.edge_list)[ge];
            const int a0 = b.edge_attr[3 * ge], a1 = b.edge_attr[3 * ge + 1], a2 = b.edge_attr[3 * ge + 2];
            const int base = s_noff[lo], n = s_noff[lo + 1] - base;
            int u = uv.x, v = uv.y;
            if (!((u >= 0) & (u < n) & (v >= 0) & (v < n))) {  // flag it, then treat as a self-loop on node 0 (as build_csr does)
                atomicMax(err, ERR_EDGE_RANGE);
                u = 0;
                v = 0;
            }
            const bool aok = (a0 >= 0) & (a0 < 5) & (a1 >= 0) & (a1 < 6) & (a2 >= 0) & (a2 < 2);  // cardinalities {5,6,2}: GIN/src/host_load.cc:6
            if (!aok) atomicMax(err, ERR_EDGE_ATTR);
            const unsigned code = aok ? (unsigned)((a0 * 6 + a1) * 2 + a2) : 0u;
            edst[k] = base + v;
            ekey[k] = ((unsigned)(base + u) << 17) | ((unsigned)i << 6) | code;
            atomicAdd(&s_cnt[base + v], 1);
        }
    }
    __syncthreads();
    const int deg = s_cnt[r];  // rows beyond the tile's last have none
    {
        const int incl = gr_wave_inclusive_scan(deg, lane);
        if (lane == 63) s_wtot[wv] = incl;
        __syncthreads();
        int start = incl - deg;
        for (int w = 0; w < wv; w++) start += s_wtot[w];
        s_cnt[r] = start;
        if (r == 255) s_cnt[GR_ROWS] = start + deg;  // = ne
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; k++)
        if (edst[k] >= 0) s_bucket[s_cnt[edst[k]] + atomicAdd(&s_cur[edst[k]], 1)] = ekey[k];
    __syncthreads();
    uint8_t* d = desc + (size_t)tile * GR_DESC_BYTES;
    uint16_t* d_edge = reinterpret_cast<uint16_t*>(d);
    uint16_t* d_rp = reinterpret_cast<uint16_t*>(d + GR_DESC_RP);
#pragma unroll
    for (int k = 0; k < EPT; k++)
        if (edst[k] >= 0) {
            const int beg = s_cnt[edst[k]], end = s_cnt[edst[k] + 1];
            const unsigned mine = ekey[k] >> 6;
            int rank = 0;
            for (int t = beg; t < end; t++) rank += (s_bucket[t] >> 6) < mine;
            d_edge[beg + rank] = (uint16_t)(((ekey[k] >> 17) << 6) | (ekey[k] & 63u));
        }
    d_rp[r] = (uint16_t)s_cnt[r];
    if (r == 255) d_rp[256] = (uint16_t)s_cnt[GR_ROWS];
    if (r < rows) {  // the node's rows of the pre-combined encoder table (validated: table cardinalities, GIN/src/host_load.cc:5)
        int f[ND_FEATURE];
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) {
            f[k] = s_feat[r * ND_FEATURE + k];
            if (f[k] < 0 || f[k] >= c_nd_card[k]) {
                atomicMax(err, ERR_NODE_FEAT);
                f[k] = 0;
            }
        }
        const unsigned i0 = f[0] * 4 + f[1], i1 = (f[2] * 12 + f[3]) * 10 + f[4], i2 = ((f[5] * 6 + f[6]) * 2 + f[7]) * 2 + f[8];
        enc_idx[(size_t)t0 + r] = i0 | (i1 << 9) | (i2 << 20);
    }
    gr_write_column_order(d, r, rows, r < rows ? (deg < ne ? deg : ne) : 0, order);
}

// ---- the tile loader's atom encoder (resident kernel, ENC form): rows of h_0 computed straight into the tile's LDS rows
// A tile's rows are encoded in GRE_PARTS parts of GRE_PART rows, one part per MLP step of the folded last layer (the rows of the
// CURRENT tile are dead by then).  A part is 37 rows = 925 float4 -- contiguous in the tile's LDS rows -- dealt to the eight waves
// 128 at a time: lane L of wave w takes float4 n = 128 w + 64 k + L (k = 0, 1) of the part, i.e. chunk n mod 25 of row n / 25: every
// lane of every load carries data (one 400-B row per 25 lanes), and a wave's stores are 1 KiB of consecutive LDS.
constexpr int GRE_PART = 37, GRE_PARTS = 7;  // 7 x 37 = 259 >= GR_ROWS
struct GrEncIdx { uint32_t k0, k1; };
struct GrEncVal { float4 a0, a1, a2, b0, b1, b2; };

// float4 index of (wave, lane, k) inside its part, or -1 beyond the part / the tile
__device__ __forceinline__ int gre_slot(int part, int wave, int lane, int k, int rows) {
    const int n = 128 * wave + 64 * k + lane;
    const int r = (n * 1311) >> 15;  // n / 25 for n < 1 024
    return (n < GRE_PART * 25 && GRE_PART * part + r < rows) ? n : -1;
}
__device__ __forceinline__ GrEncIdx gre_issue_idx(const uint32_t* __restrict__ enc_idx, const GrTile& t, int part, int wave, int lane) {
    GrEncIdx x;
    const int n0 = gre_slot(part, wave, lane, 0, t.rows), n1 = gre_slot(part, wave, lane, 1, t.rows);
    // (unconditional, from a clamped row: ALWAYS two loads -- the s_waitcnt vmcnt(..) of gr_layer count on that -- and any row's
    // numbers are valid table rows; gre_finish writes only the real rows)
    x.k0 = enc_idx[(size_t)t.t0 + (n0 >= 0 ? GRE_PART * part + ((n0 * 1311) >> 15) : 0)];
    x.k1 = enc_idx[(size_t)t.t0 + (n1 >= 0 ? GRE_PART * part + ((n1 * 1311) >> 15) : 0)];
    return x;
}
#define GRE_LD3(V0, V1, V2, K, C)                                                  \
    V0 = tab[((unsigned)GRB_T01 + ((K) & 511u)) * 25u + (C)];                      \
    V1 = tab[((unsigned)GRB_T234 + (((K) >> 9) & 2047u)) * 25u + (C)];             \
    V2 = tab[((unsigned)GRB_T5678 + ((K) >> 20)) * 25u + (C)];
__device__ __forceinline__ GrEncVal gre_issue_tab(const float4* __restrict__ tab, const GrEncIdx& x, int wave, int lane) {
    GrEncVal v;
    const unsigned n0 = 128u * (unsigned)wave + (unsigned)lane, n1 = n0 + 64u;  // (< 1 024: chunk n - 25 (n / 25) is in range for every lane)
    const unsigned c0 = n0 - 25u * ((n0 * 1311u) >> 15), c1 = n1 - 25u * ((n1 * 1311u) >> 15);
    GRE_LD3(v.a0, v.a1, v.a2, x.k0, c0)
    GRE_LD3(v.b0, v.b1, v.b2, x.k1, c1)
    return v;
}
#undef GRE_LD3
#define GRE_SUM(V0, V1, V2) make_float4((V0.x + V1.x) + V2.x, (V0.y + V1.y) + V2.y, (V0.z + V1.z) + V2.z, (V0.w + V1.w) + V2.w)
__device__ __forceinline__ void gre_finish(float* s_h, const GrEncVal& v, const GrTile& t, int part, int wave, int lane) {
    const int n0 = gre_slot(part, wave, lane, 0, t.rows), n1 = gre_slot(part, wave, lane, 1, t.rows);
    float4* dst = reinterpret_cast<float4*>(s_h) + GRE_PART * 25 * part;
    if (n0 >= 0) dst[n0] = GRE_SUM(v.a0, v.a1, v.a2);
    if (n1 >= 0) dst[n1] = GRE_SUM(v.b0, v.b1, v.b2);
}
#undef GRE_SUM

__device__ __forceinline__ void gr_issue_desc(const uint8_t* __restrict__ desc, int tile, char* s_desc, int wave, int lane) {
    if (wave < 4 && (wave < 3 || lane < 32))
        lds_dma16(reinterpret_cast<const char*>(desc) + (size_t)tile * GR_DESC_BYTES + wave * 1024, (uint32_t)lane * 16u, lds_addr_of(s_desc) + wave * 1024);
}

template <bool PROF, bool HUBS, bool LAST, bool FOLD, bool ENC>
__device__ __forceinline__ void gr_layer(unsigned long long (&tacc)[6], char* bx, char* by, float* s_h, char* s_desc, float* s_dot,
                                         const GrTile& cur, const GrTile& nxt, bool has_next, int next_tile, int l,
                                         const float* __restrict__ h0, const uint8_t* __restrict__ desc, const float* __restrict__ ecomb_all,
                                         const uint8_t* __restrict__ wchunks_all, const float* __restrict__ pool_w,
                                         float* __restrict__ hout, float& vmax, int wave, int lane, const float* s_u,
                                         const uint32_t* __restrict__ enc_idx, const float4* __restrict__ enc_tab, int (&tile_trips)[2]) {
    constexpr int NT = 2;
    // (opaque per layer: what is computed from the lane id -- LDS and global addresses of the fragment reads and DMA pieces, shuffle
    // indices -- is otherwise hoisted out of the tile loop, forty values that live across the whole kernel, spill, and are reloaded inside
    // the MLP steps behind an s_waitcnt vmcnt(0) that also waits for the chunk DMA in flight)
    asm volatile("" : "+v"(lane));
    const int j = lane & 15, g = lane >> 4;
    constexpr bool last = LAST;  // compile-time: the first four layers carry none of the last layer's code (and registers)
    // last layer with the readout folded through its second linear layer: h_5 . w = hid . (W2^T w) + b2 . w, so only the hidden tiles
    // are computed and dotted with u = W2^T w_pred (s_u, pre-divided by the first layer's power-of-two scale)
    constexpr bool fold = LAST && FOLD;  // compile-time as well: the folded kernel carries no second linear layer for its last layer
    const uint16_t* s_edge = reinterpret_cast<const uint16_t*>(s_desc);
    const uint16_t* s_rp = reinterpret_cast<const uint16_t*>(s_desc + GR_DESC_RP);
    const uint8_t* s_perm = reinterpret_cast<const uint8_t*>(s_desc + GR_DESC_PERM);
    const uint8_t* wchunks = wchunks_all + (size_t)l * GS_STEPS * GRC_CHUNK_STRIDE;
    const int ln = last ? 0 : l + 1;  // the layer whose table is prefetched during step 7 (the next tile starts at layer 0)
    unsigned long long tp = 0;
    if constexpr (PROF) tp = wall_clock64();
    // Static priority for the second-dispatched half (waves 4-7) pays in the MLP (they lose every arbitration otherwise), but in the
    // gather it makes their SIMD partners (waves 0-3) the phase's stragglers -- 2 331 us against 1 799 per wave by phase stamps, and
    // the phase ends with its slowest wave: equal priority while gathering, raised again behind the gather's barrier (then age
    // decides and waves 4-7 trail by less: 2 305 against 1 959; launch 8.70 -> 8.60 ms.  Priorities alternating trip by trip even the
    // halves out but cost more than they return: 8.69 ms).
    if (GR_PRIO_BY_PHASE) { if (GR_PRIO_HALF && wave >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    // ENC (the tile loader computes h_0 itself): the next tile's rows are encoded during this layer's MLP steps, a part per step;
    // the table-row numbers of part 0 are requested here, a whole gather ahead of their use
    GrEncIdx enc_ix{};
    GrEncVal enc_v{};
    if constexpr (ENC && LAST) {
        if (has_next) enc_ix = gre_issue_idx(enc_idx, nxt, 0, wave, lane);
    }

    // ---- gather (MP unit) out of LDS: a = h[v] + sum_e relu(h[src_e] + ecomb[code_e]), CSR order
    float bq[NT][25];
    int e_cur[NT], e_end[NT];
    int hub_beg[NT] = {0, 0}, hub_end[NT] = {0, 0};
    unsigned wd[NT];
    int row[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        row[nt] = s_perm[wave * (16 * NT) + nt * 16 + j];
        const bool valid = row[nt] < cur.rows;
        const int rr = valid ? row[nt] : 0;
        e_cur[nt] = s_rp[rr];
        e_end[nt] = s_rp[rr + 1];
        if (!valid) e_end[nt] = e_cur[nt];
        if constexpr (HUBS) {  // hub rows sit out the per-lane walk (below: the whole 16-lane group walks them)
            hub_beg[nt] = e_cur[nt];
            hub_end[nt] = e_end[nt];
            if (e_end[nt] - e_cur[nt] > GR_HUB_DEG) e_end[nt] = e_cur[nt];
            else hub_end[nt] = hub_beg[nt];
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) wd[nt] = e_cur[nt] < e_end[nt] ? (unsigned)s_edge[e_cur[nt]] : GR_NO_EDGE;
    const float* s_ecomb = reinterpret_cast<const float*>(bx);
    // Two memories feed the walk: three of a table row's six quads come through the vector-memory path, from the table's plane-ordered
    // copy in global memory (gin.hip set_weights; [quad][quarter][code] of 16 B: a quarter wave's sixteen codes fall into 8 cache
    // lines) -- the CU's L1 path is idle between weight chunks, and the walk waits for the LDS array (HUBS: `lds_busy` is the kernel's
    // limit, limits.json) or at least for its latency under bank conflicts.  As in gcn_resident_kernel's walk, where the split returns
    // 6 %; same values, same arithmetic: bit-identical.  Measured (2^18 graphs, A/B on one box, gin_resident):
    //   GIN-VN  quads 0..2: 9.02 -> 8.87 ms (2 / 3 / 4 quads: -1.0 / -1.7 / -1.4 %; quads 3..5: +0.3 % on that; requested a trip ahead:
    //           +0.4 %, 24 more live registers in a loop at 250; in the hub walk as well: +-0)
    //   GIN     quads 3..5 -- the ones the fold consumes LAST, so the loads' longer latency hides behind the LDS-fed quads: 7.70 -> 7.61
    //           (3 / 4 / 5 quads: -1.2 / -0.9 / +0.2 %); quads 0..2, consumed first: +-0.3 %
    constexpr int EQ_VMEM = 3, EQ_LO = HUBS ? 0 : 3;
    const float* epl = ecomb_all + GR_PLANES_OFF + (size_t)l * GR_PLANES_FLOATS + 256 * g;
    (void)epl;
    // A lane whose row has no in-edge left walks the NO-EDGE word: source row GR_ROWS of the tile (-1e30 in every feature, written
    // once per workgroup), so its message is relu(-1e30 + e) = +0 and the accumulation needs no per-lane guard.  (With the guard,
    // `if (lane active) bq += ...`, hipcc keeps two copies of the 25 accumulators and moves them back and forth: 54 v_mov per trip
    // beside the 48 packed adds and 50 max that are the work.)  Adding +0 leaves every sum's bits as they were.
    // Trip counts are wave-uniform and known up front (the column tile's largest in-degree), so the walk is three branch-free
    // loops -- both column tiles, then whichever one has rows left -- and in the first both tiles' LDS reads of a trip are requested
    // BEFORE either is folded: the second tile's 14 reads are in flight while the first one's VALU instructions issue.  (A single
    // `while (any lane active)` loop with per-tile guards costs the same moves again: the accumulators become loop phis that hipcc
    // copies on every path.)  Same sums, same order.
    // (once per tile, in layer 0, and carried in scalar registers: the rows and their in-degrees are the same in all five layers, and
    // the two six-step butterflies are twelve dependent cross-lane round trips in front of the walk's first trip)
    int trips[NT];
    if (l == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            int d = e_end[nt] - e_cur[nt];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) d = max(d, __shfl_xor(d, m, 64));
            tile_trips[nt] = __builtin_amdgcn_readfirstlane(d);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) trips[nt] = tile_trips[nt];
#ifdef FLOWGNN_DEV
#include "dev/walk_timing_variants.h"  // development: timing variants of the walk (wrong sums on purpose), selected by extra -D flags
#endif
#ifndef GR_WALK_ROW
#define GR_WALK_ROW(U) (U)
#endif
#ifndef GR_WALK_CODE
#define GR_WALK_CODE(C) (C)
#define GR_WALK_TIMING_SETUP()
#endif
#define GR_READ(NTI, X, W, XT, WT)                                                                                                \
    {                                                                                                                             \
        const unsigned u = GR_WALK_ROW(wd[NTI] >> 6), code = GR_WALK_CODE(wd[NTI] & 63u);                                                    \
        const float* hr = s_h + u * GS_D + 4 * g;                                                                                 \
        const float* er = s_ecomb + code * GS_D + 4 * g;                                                                          \
        _Pragma("unroll") for (int q = EQ_LO; q < EQ_LO + EQ_VMEM; q++) W[q] = *reinterpret_cast<const float4_t*>(epl + code * 4 + 1024 * q); \
        _Pragma("unroll") for (int q = 0; q < 6; q++) {                                                                           \
            X[q] = *reinterpret_cast<const float4_t*>(hr + 16 * q);                                                               \
            if (q < EQ_LO || q >= EQ_LO + EQ_VMEM) W[q] = *reinterpret_cast<const float4_t*>(er + 16 * q);                        \
        }                                                                                                                         \
        XT = s_h[u * GS_D + 96 + g];                                                                                              \
        WT = s_ecomb[code * GS_D + 96 + g];                                                                                       \
    }
#define GR_NEXT(NTI)                                                                                                              \
    {                                                                                                                             \
        e_cur[NTI]++;                                                                                                             \
        const bool more = e_cur[NTI] < e_end[NTI];                                                                                \
        const unsigned nw = s_edge[more ? e_cur[NTI] : 0];                                                                        \
        wd[NTI] = more ? nw : GR_NO_EDGE;                                                                                         \
    }
    // The message relu(x + e) as ONE packed instruction (round 5).  The SIMD issues MFMAs and VALU instructions through one port
    // (tools/coissue4.hip), so the walk's instruction count is kernel time: add + two v_max + add per two values were 51 VALU
    // instructions per in-edge and lane.  The VOP3P clamp bit clamps a float result to [0, 1]; with the table stored as e * 2^-16
    // (gin.hip: the resident kernel's own copy) v_pk_fma_f32(x, 2^-16, e * 2^-16) clamp = relu(x + e) * 2^-16 EXACTLY -- scaling
    // by a power of two commutes with every rounding -- as long as x + e < 2^16.  The sums run in the scaled domain and come back with
    // the self term (a = fma(m', 2^16, h[v]) below): the same bits as before.  x + e >= 2^16 needs h[u] > 6e4 (|e| < 4 096 is checked
    // when the weights are set), and then a[u] = h[u] + m[u] >= h[u] raises the range flag in this very layer: the pass is repeated on
    // the exact kernels, as for any operand beyond the f16 range.  26 VALU instructions per in-edge and lane.
    const float2_t gr_c2 = {GR_MSG_SCALE, GR_MSG_SCALE};
#define GR_MSG2(D, XX, WW) asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(D) : "v"(XX), "s"(gr_c2), "v"(WW))  /* (the constant in a scalar pair: two more live vector registers spilled) */
#define GR_MSG1(D, XX, WW) asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(D) : "v"(XX), "s"(GR_MSG_SCALE), "v"(WW))
#define GR_FOLD(NTI, X, W, XT, WT)                                                                                                \
    {                                                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 6; q++) {                                                                           \
            float2_t a, b;                                                                                                        \
            { const float2_t xl = X[q].lo, wl = W[q].lo, xh = X[q].hi, wh = W[q].hi; GR_MSG2(a, xl, wl); GR_MSG2(b, xh, wh); }      \
            aq[NTI][2 * q + 0] += a;                                                                                              \
            aq[NTI][2 * q + 1] += b;                                                                                              \
        }                                                                                                                         \
        { float m1; GR_MSG1(m1, XT, WT); at[NTI] += m1; }                                                                         \
    }
    // the 24 + 1 sums of a row as 12 register PAIRS: one v_pk_fma_f32 (clamp) for the message, one v_pk_add_f32 for the accumulation
    float2_t aq[NT][12];
    float at[NT] = {0.0f, 0.0f};
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
#pragma unroll
        for (int k = 0; k < 12; k++) aq[nt][k] = (float2_t){0.0f, 0.0f};
    }
    const int tboth = min(trips[0], trips[1]);
    GR_WALK_TIMING_SETUP()
#ifdef GR_PROF_WALK  // development: tacc[5] = the three walk loops alone, tacc[4] = self term + operand split (instead of DMA wait / step barriers)
    unsigned long long tw0 = 0;
    if constexpr (PROF) tw0 = wall_clock64();
#endif
#pragma unroll 1
    for (int t = 0; t < tboth; t++) {
        float4_t x0[6], w0[6], x1[6], w1[6];
        float xt0, wt0, xt1, wt1;
        if (GR_PRIO_HALF && wave >= 4 && t == ((tboth * (HUBS ? 4 : GR_PRIO_QUARTERS)) >> 2)) __builtin_amdgcn_s_setprio(0);  // (GIN-VN: through all of them, -0.2 %)
        GR_READ(0, x0, w0, xt0, wt0)
        GR_READ(1, x1, w1, xt1, wt1)
        GR_NEXT(0)
        GR_NEXT(1)
        GR_FOLD(0, x0, w0, xt0, wt0)
        GR_FOLD(1, x1, w1, xt1, wt1)
    }
    // Chunk 0 of the weight stream (dealt by all eight waves: nobody multiplies now) is requested HERE, behind the trips that walk both
    // column tiles, and lands under the tails and the operand split.  At the top of the walk -- where it used to be -- every wave paid
    // its LDS-DMA issue (100-150 cycles an instruction) in front of its first trip and the pieces landed through the LDS the walk is
    // bound by: launch 8.33 ms; behind the whole walk 8.31 (the wait in front of the barrier then sees some of the latency); here
    // 8.27 (same box, scripts/dev/ab.py).  GCN's kernel moved its W_{l+1} the same way for -2.8 %.
    if (fold) grc_issue_chunk_w1<8>(wchunks, by, wave, lane);
    else grc_issue_chunk<8>(wchunks, by, wave, lane);
#pragma unroll 1
    for (int t = tboth; t < trips[0]; t++) {
        float4_t x0[6], w0[6];
        float xt0, wt0;
        GR_READ(0, x0, w0, xt0, wt0)
        GR_NEXT(0)
        GR_FOLD(0, x0, w0, xt0, wt0)
    }
#pragma unroll 1
    for (int t = tboth; t < trips[1]; t++) {
        float4_t x1[6], w1[6];
        float xt1, wt1;
        GR_READ(1, x1, w1, xt1, wt1)
        GR_NEXT(1)
        GR_FOLD(1, x1, w1, xt1, wt1)
    }
#undef GR_READ
#undef GR_NEXT
#undef GR_FOLD  // (GR_MSG2 / GR_MSG1 serve the hub walk below as well)
#ifdef GR_PROF_WALK
    if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[5] += t - tw0; tw0 = t; }
#endif
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
#pragma unroll
        for (int k = 0; k < 12; k++) { bq[nt][2 * k] = aq[nt][k].x; bq[nt][2 * k + 1] = aq[nt][k].y; }
        bq[nt][24] = at[nt];
    }
    if constexpr (HUBS) {
        // Hub rows (GIN-VN's virtual nodes: in-degree = graph size): one lane walking 26 in-edges would hold its column tile for 26
        // trips.  Instead every lane j of the column tile takes the hub's in-edges j, j + 16, ... (in CSR order), and the 16 partial
        // sums are combined with a fixed butterfly -- an association that depends on the row only, so results stay bit-identical
        // under any batch split or order (as in the per-layer kernel's hub path, gin_split.hip above).
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            unsigned long long hm = __ballot(hub_end[nt] > hub_beg[nt] && g == 0);
            while (hm) {
                const int jh = __ffsll((long long)hm) - 1;
                hm &= hm - 1;
                const int hb = __builtin_amdgcn_readlane(hub_beg[nt], jh), he = __builtin_amdgcn_readlane(hub_end[nt], jh);
                float part[25];
                float2_t pq[12];  // as in the row walk: v_pk_add_f32 for the message and for the accumulation
#pragma unroll
                for (int k = 0; k < 12; k++) pq[k] = (float2_t){0.0f, 0.0f};
                float pt = 0.0f;
                const int hub_trips = (he - hb + 15) >> 4;  // wave-uniform; lanes past the hub's last in-edge walk the no-edge word
#pragma unroll 1
                for (int t = 0; t < hub_trips; t++) {
                    const int e = hb + j + 16 * t;
                    const unsigned w2 = e < he ? (unsigned)s_edge[e < he ? e : 0] : GR_NO_EDGE;
                    const unsigned u = w2 >> 6, code = w2 & 63u;
                    const float* hr = s_h + u * GS_D + 4 * g;
                    const float* er = s_ecomb + code * GS_D + 4 * g;
                    float4_t x[6], w[6];
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        x[q] = *reinterpret_cast<const float4_t*>(hr + 16 * q);
                        w[q] = *reinterpret_cast<const float4_t*>(er + 16 * q);
                    }
                    const float xt = s_h[u * GS_D + 96 + g];
                    const float wt = s_ecomb[code * GS_D + 96 + g];
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        float2_t a, b;
                        { const float2_t xl = x[q].lo, wl = w[q].lo, xh = x[q].hi, wh = w[q].hi; GR_MSG2(a, xl, wl); GR_MSG2(b, xh, wh); }
                        pq[2 * q + 0] += a;
                        pq[2 * q + 1] += b;
                    }
                    { float m1; GR_MSG1(m1, xt, wt); pt += m1; }
                }
#pragma unroll
                for (int k = 0; k < 12; k++) { part[2 * k] = pq[k].x; part[2 * k + 1] = pq[k].y; }
                part[24] = pt;
                // all-reduce over the 16 lanes of the column tile with DPP row rotations (8, 4, 2, 1): every lane adds the same pairs
                // at every level, so all 16 hold the same bits whichever lane owns the hub
                // (one v_add_f32_dpp per register and level, spelled out: hipcc leaves `x += update_dpp(x)` as v_mov_b32_dpp + v_add_f32 -- 582
                // unfused moves in this kernel.  A level's 25 adds are independent and a register is read again 25 instructions after it was
                // written, beyond the two wait states a DPP read needs; the s_nop covers the first read of each level.)
#define GR_ROR_ADD(ROR)                                                                                                           \
    asm volatile("s_nop 1");                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < 25; k++)                                                                                \
        asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:" #ROR " row_mask:0xf bank_mask:0xf" : "=v"(part[k]) : "v"(part[k]));  /* volatile: in source order */
                GR_ROR_ADD(8) GR_ROR_ADD(4) GR_ROR_ADD(2) GR_ROR_ADD(1)
#undef GR_ROR_ADD
                if (j == jh) {
#pragma unroll
                    for (int k = 0; k < 25; k++) bq[nt][k] += part[k];
                }
            }
        }
    }
#undef GR_MSG2
#undef GR_MSG1
    uint4_t in_hi[NT][3], in_lo[NT][3], in_tb[NT];
    {   // + (1 + eps) h[v], eps == 0; rows beyond the tile contribute zeros.  All fourteen reads of the wave's own rows are requested
        // before the first add: left to itself hipcc re-uses ONE register quad and serialises them -- ds_read_b128, s_waitcnt
        // lgkmcnt(0), two adds, fourteen times per layer (the walk's read registers are dead here: there is room for all of them)
        float4_t sx[NT][6];
        float sxt[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int rr = row[nt] < cur.rows ? row[nt] : 0;
            const float* hr = s_h + rr * GS_D + 4 * g;
#pragma unroll
            for (int q = 0; q < 6; q++) sx[nt][q] = *reinterpret_cast<const float4_t*>(hr + 16 * q);
            sxt[nt] = s_h[rr * GS_D + 96 + g];
        }
        GR_SB();
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            // a = h[v] + m = fma(m * 2^-16, 2^16, h[v]): the scaled sums come back exactly.  Rows beyond the tile's last are left to run
            // (unmasked: no exec-mask region between the walk and the first MFMAs): their sums are +0, their self term is row 0's, what the
            // layers make of it is stored in LDS rows no graph owns and is read by nobody
            {
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    bq[nt][4 * q + 0] = __builtin_fmaf(bq[nt][4 * q + 0], GR_MSG_UNSCALE, sx[nt][q].x); bq[nt][4 * q + 1] = __builtin_fmaf(bq[nt][4 * q + 1], GR_MSG_UNSCALE, sx[nt][q].y);
                    bq[nt][4 * q + 2] = __builtin_fmaf(bq[nt][4 * q + 2], GR_MSG_UNSCALE, sx[nt][q].z); bq[nt][4 * q + 3] = __builtin_fmaf(bq[nt][4 * q + 3], GR_MSG_UNSCALE, sx[nt][q].w);
                }
                bq[nt][24] = __builtin_fmaf(bq[nt][24], GR_MSG_UNSCALE, sxt[nt]);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
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
        vmax = __builtin_fmaxf(vmax, __builtin_fabsf(bq[nt][24]));
        {   // K tail (features 96..99, one per lane group): all four to every lane, then the packed operand
            //   g = 0: [hi(96..99), lo(96..99)]   g = 1: [hi(96..99), 0]   g = 2, 3: 0      (GR chunks, K-tail fragments)
            const float t0 = __shfl(bq[nt][24], j, 64), t1 = __shfl(bq[nt][24], j + 16, 64);
            const float t2 = __shfl(bq[nt][24], j + 32, 64), t3 = __shfl(bq[nt][24], j + 48, 64);
            uint32_t h01, h23, l01, l23;
            GS_SPLIT2(t0, t1, h01, l01);
            GS_SPLIT2(t2, t3, h23, l23);
            in_tb[nt] = g == 0 ? (uint4_t){h01, h23, l01, l23} : (g == 1 ? (uint4_t){h01, h23, 0u, 0u} : (uint4_t){0u, 0u, 0u, 0u});
        }
    }
#ifdef GR_PROF_WALK
    if constexpr (PROF) { asm volatile("" :: "v"(in_hi[0][0].x), "v"(in_lo[1][2].w), "v"(in_tb[1].x)); const unsigned long long t = wall_clock64(); tacc[4] += t - tw0; }
#endif
    if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[0] += t - tp; tp = t; }
    if constexpr (ENC && LAST) {
        // part 0's table rows (its row numbers were requested at the top of the layer) and part 1's row numbers: eight loads that stay
        // in flight across the barrier -- step 0's hook sums part 0 (the chunk-0 DMA pieces are older: vmcnt retires in order)
        if (has_next) {
            enc_v = gre_issue_tab(enc_tab, enc_ix, wave, lane);
            enc_ix = gre_issue_idx(enc_idx, nxt, 1, wave, lane);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk 0
    }
    __syncthreads();  // chunk 0 resident; every wave is done with the table (bx), with the tile's rows and with its CSR slice
    if (GR_PRIO_BY_PHASE && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[1] += t - tp; tp = t; }

    // last layer: the rows and the descriptor of this tile are dead -- bring in the next tile's (rows: an eighth per MLP step)
    if (last && has_next) gr_issue_desc(desc, next_tile, s_desc, wave, lane);

    // ---- node MLP (NT unit), weights streamed through LDS
    float4_t acc2[NT][GS_T2];
    uint4_t h_hi[NT], h_lo[NT];
    float oscale = 1.0f;
    float dot[NT] = {0.0f, 0.0f};
    float4_t pend[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // a step's last hidden tile, finished at the top of the next step (gr_step)
    if (fold) {
        // seven steps of hidden tiles only; the next layer's table (layer 0 of the next tile) lands in bx during step 6, when bx is free
        // -- the buffers do NOT swap roles after this layer (gin_resident_kernel)
#pragma unroll 1
        for (int c = 0; c < GS_STEPS - 1; c++) {
            char* cb = (c & 1) ? bx : by;
            char* nb = (c & 1) ? by : bx;
            if (c + 1 < GS_STEPS - 1) grc_issue_chunk_w1(wchunks + (size_t)(c + 1) * GRC_CHUNK_STRIDE, nb, wave, lane);
            else gr_issue_ecomb(ecomb_all + (size_t)ln * EDGE_COMBOS * GS_D, nb, wave, lane);
            if constexpr (!ENC) {
                if (has_next) {
                    gr_issue_rows(h0, reinterpret_cast<char*>(s_h), nxt, c, wave, lane);
                    if (c == GS_STEPS - 2) gr_issue_rows(h0, reinterpret_cast<char*>(s_h), nxt, c + 1, wave, lane);
                }
            }
            // ENC (the tile loader computes h_0 itself), in the MIDDLE of the step (gr_step, HOOK): part c of the next tile -- its table rows
            // were requested a step ago -- is summed into the tile's rows, then part c + 1's table rows and part c + 2's row numbers are
            // requested.  At the step's ends, where this used to sit, the ~120 VALU / memory instructions per wave issued with both waves
            // of every SIMD off the matrix pipe: 0.3 ms per launch.
            auto enc_hook = [&]() {
                if constexpr (ENC) {
                    if (has_next) {
                        gre_finish(s_h, enc_v, nxt, c, wave, lane);
                        if (c + 1 < GRE_PARTS) enc_v = gre_issue_tab(enc_tab, enc_ix, wave, lane);
                        if (c + 2 < GRE_PARTS) enc_ix = gre_issue_idx(enc_idx, nxt, c + 2, wave, lane);
                    }
                }
            };
            if (c < GS_STEPS - 2) gr_step<4>(cb, lane, g, in_hi, in_lo, in_tb, h_hi, h_lo, acc2, vmax, pend, s_u + 32 * c, dot, wave, enc_hook);
            else gr_step<5>(cb, lane, g, in_hi, in_lo, in_tb, h_hi, h_lo, acc2, vmax, pend, s_u + 32 * c, dot, wave, enc_hook);
            if (c + 1 < GS_STEPS - 1) {
                unsigned long long tw = 0;
                if constexpr (PROF) tw = wall_clock64();
                // this wave's DMA pieces have landed.  (ENC) the eight loads the hook requested BEHIND them -- six of table rows, two of row
                // numbers; always exactly that many (gre_issue_*) -- stay in flight: vmcnt retires in order, so "all but the newest
                // eight" covers every DMA piece.  Waiting for them here would put an L2 / HBM round trip at the end of every step.
                if (ENC && has_next && c + 2 < GRE_PARTS) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (ENC && has_next && c + 1 < GRE_PARTS) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[GR_TACC_DMA] += t - tw; tw = t; }
                if (c + 1 < GS_STEPS - 1) {
                    __syncthreads();
                    if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[GR_TACC_BAR] += t - tw; }
                }
            }
        }
    } else {
#pragma unroll
    for (int t2 = 0; t2 < GS_T2; t2++) {
        const float4 b = *reinterpret_cast<const float4*>(by + GRC_W2_OFF + (16 * t2 + 4 * g) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc2[nt][t2] = (float4_t){b.x, b.y, b.z, b.w};
    }
    oscale = *reinterpret_cast<const float*>(by + GRC_W2_OFF + 112 * 4);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) { h_hi[nt] = (uint4_t){0, 0, 0, 0}; h_lo[nt] = (uint4_t){0, 0, 0, 0}; }
#pragma unroll 1
    for (int c = 0; c < GR_STEPS - 1; c += 2) {
        // even step: compute from by while chunk c+1 streams into bx
        grc_issue_chunk(wchunks + (size_t)(c + 1) * GRC_CHUNK_STRIDE, bx, wave, lane);
        if (last && has_next) gr_issue_rows(h0, reinterpret_cast<char*>(s_h), nxt, c, wave, lane);
        if (c == 0) gr_step<0, false, GR_DEFER>(by, lane, g, in_hi, in_lo, in_tb, h_hi, h_lo, acc2, vmax, pend, nullptr, nullptr, wave);
        else gr_step<1, GR_DEFER, GR_DEFER>(by, lane, g, in_hi, in_lo, in_tb, h_hi, h_lo, acc2, vmax, pend, nullptr, nullptr, wave);
        unsigned long long tw = 0;
        if constexpr (PROF) tw = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk c+1 (and of the next tile) have landed
        if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[GR_TACC_DMA] += t - tw; tw = t; }
        __syncthreads();                                  // everyone's landed; everyone is done with by
        if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[GR_TACC_BAR] += t - tw; }
        // odd step: compute from bx while chunk c+2 streams into by
        grc_issue_chunk(wchunks + (size_t)(c + 2) * GRC_CHUNK_STRIDE, by, wave, lane);
        if (last && has_next) gr_issue_rows(h0, reinterpret_cast<char*>(s_h), nxt, c + 1, wave, lane);
        if (c + 1 == GR_STEPS - 2) gr_step<1, GR_DEFER, false>(bx, lane, g, in_hi, in_lo, in_tb, h_hi, h_lo, acc2, vmax, pend, nullptr, nullptr, wave);
        else gr_step<1, GR_DEFER, GR_DEFER>(bx, lane, g, in_hi, in_lo, in_tb, h_hi, h_lo, acc2, vmax, pend, nullptr, nullptr, wave);
        if constexpr (PROF) tw = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[GR_TACC_DMA] += t - tw; tw = t; }
        __syncthreads();
        if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[GR_TACC_BAR] += t - tw; }
    }
    // the last step (chunk 6, in by): hidden tile 12, K-step 5 and the packed K-step 6; the next layer's table streams into bx, which
    // nobody reads any more -- an odd number of steps, so the two buffers keep their roles from layer to layer.  In the layers that
    // feed another one the finished rows go back into the tile from inside the step (gr_step_final).
    gr_issue_ecomb(ecomb_all + (size_t)ln * EDGE_COMBOS * GS_D, bx, wave, lane);
    if (last && has_next) {
        gr_issue_rows(h0, reinterpret_cast<char*>(s_h), nxt, GR_STEPS - 1, wave, lane);
        gr_issue_rows(h0, reinterpret_cast<char*>(s_h), nxt, GR_STEPS, wave, lane);
    }
    gr_step_final<!last>(by, lane, g, in_hi, in_lo, in_tb, h_hi, h_lo, acc2, vmax, oscale, s_h + row[0] * GS_D, s_h + row[1] * GS_D);

    }
    if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[2] += t - tp; tp = t; }
    // ---- epilogue: h' back into the tile (in place: nobody reads the old rows any more), or the readout terms
    if (!last) {
        // (the rows went back into the tile from inside the last step)
    } else if (fold) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {  // 13 hidden tiles x 4 units per lane, then the node's 4 lanes: a fixed order per node
            float part = dot[nt];
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (g == 0) s_dot[row[nt]] = part;
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            float part = 0.0f;
            const bool valid = row[nt] < cur.rows;
#pragma unroll
            for (int t2 = 0; t2 < GS_T2; t2++) {
                const int col = 16 * t2 + 4 * g;
                if (col < GS_D) {
                    const float4_t r = acc2[nt][t2] * oscale;  // no ReLU after the last layer (GIN/src/node_embedding.cc:185-191)
                    const float4 pw = *reinterpret_cast<const float4*>(pool_w + col);
                    part += r.x * pw.x; part += r.y * pw.y; part += r.z * pw.z; part += r.w * pw.w;
                    if (hout != nullptr && valid)  // debug tap (flowgnn_get_h): the rows themselves
                        *reinterpret_cast<float4*>(hout + (size_t)(cur.t0 + row[nt]) * GS_D + col) = make_float4(r.x, r.y, r.z, r.w);
                }
            }
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (g == 0) s_dot[row[nt]] = part;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next layer's table (and the last pieces of the next tile)
    __syncthreads();
    if constexpr (PROF) { const unsigned long long t = wall_clock64(); tacc[3] += t - tp; }
}

template <bool PROF, bool HUBS, bool FOLD, bool ENC>
__global__ __launch_bounds__(GR_WAVES * 64, 2) void gin_resident_kernel(const float* __restrict__ h0, float* __restrict__ hout,
                                                                       const float* __restrict__ ecomb_all,
                                                                       const uint8_t* __restrict__ wchunks_all,
                                                                       const float* __restrict__ pool_w, const float* __restrict__ pool_b,
                                                                       const int* __restrict__ tile_row, const int* __restrict__ tile_graph,
                                                                       const uint8_t* __restrict__ desc,
                                                                       const int* __restrict__ node_off, float* __restrict__ out, int n_tiles,
                                                                       int* __restrict__ range_flag, unsigned long long* __restrict__ prof_out,
                                                                       const float* __restrict__ head_u, const uint32_t* __restrict__ enc_idx,
                                                                       const float4* __restrict__ enc_tab, int tstride,
                                                                       const int* __restrict__ list, const int* __restrict__ lrow) {
    static_assert(!ENC || FOLD, "the in-kernel encoder rides on the folded last layer's steps");
    __shared__ __attribute__((aligned(16))) char s_a[GRC_CHUNK_BYTES];
    __shared__ __attribute__((aligned(16))) char s_b[GRC_CHUNK_BYTES];
    __shared__ __attribute__((aligned(16))) float s_h[(GR_ROWS + 1) * GS_D];  // + the no-edge row (GR_NO_EDGE)
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tk0 = 0;
    if constexpr (PROF) tk0 = wall_clock64();
    __shared__ __attribute__((aligned(16))) char s_desc[GR_DESC_BYTES];
    __shared__ float s_dot[GR_ROWS];
    __shared__ __attribute__((aligned(16))) float s_u[208];  // 13 hidden tiles x 16 units  // head_u: the readout folded through the last layer's W2 (gr_layer)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    constexpr bool fold_head = FOLD;  // the launcher instantiates FOLD only with head_u, a single-task readout and no per-node tap
    if (fold_head && (int)threadIdx.x < 208) s_u[threadIdx.x] = head_u[threadIdx.x];
    if ((int)threadIdx.x < GS_D) s_h[GR_ROWS * GS_D + threadIdx.x] = -1.0e30f;
    const float head_c = fold_head ? head_u[208] : 0.0f;
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // the second-dispatched half loses every arbitration otherwise (MI355X_MICROARCH: static priority)
    GrTile cur = gr_load_tile(tile_row, tile_graph, tile, n_tiles, tstride);
    // prologue: this workgroup's first tile (rows, descriptor) and the first table
    gr_issue_ecomb(ecomb_all, s_a, wave, lane);
    gr_issue_desc(desc, tile, s_desc, wave, lane);
    if constexpr (ENC) {  // this workgroup's first tile is encoded on the spot (later ones under the previous tile's last layer)
#pragma unroll 1
        for (int part = 0; part < GRE_PARTS; part++) {
            const GrEncIdx ix = gre_issue_idx(enc_idx, cur, part, wave, lane);
            const GrEncVal v = gre_issue_tab(enc_tab, ix, wave, lane);
            gre_finish(s_h, v, cur, part, wave, lane);
        }
    } else {
#pragma unroll 1
        for (int part = 0; part < 8; part++) gr_issue_rows(h0, reinterpret_cast<char*>(s_h), cur, part, wave, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float vmax = 0.0f;
    int tile_trips[2] = {0, 0};
    while (true) {
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < n_tiles;
        const GrTile nxt = gr_load_tile(tile_row, tile_graph, ntile, n_tiles, tstride);  // used five layers from now
#pragma unroll 1
        for (int l = 0; l < 4; l++)
            gr_layer<PROF, HUBS, false, FOLD, ENC>(tacc, s_a, s_b, s_h, s_desc, s_dot, cur, nxt, has_next, ntile, l, h0, desc, ecomb_all, wchunks_all, pool_w, hout, vmax, wave, lane, nullptr, enc_idx, enc_tab, tile_trips);
        // (every layer runs an odd number of MLP steps -- seven -- so the table buffer s_a and the first chunk's buffer s_b keep their roles)
        gr_layer<PROF, HUBS, true, FOLD, ENC>(tacc, s_a, s_b, s_h, s_desc, s_dot, cur, nxt, has_next, ntile, 4, h0, desc, ecomb_all, wchunks_all, pool_w, hout, vmax, wave, lane, fold_head ? s_u : nullptr, enc_idx, enc_tab, tile_trips);
        // readout (GIN/src/finalize.cc:36-113): out[g] = mean_v(h5[v] . w) + b, node order; the terms stay valid until the next
        // tile's last layer rewrites them, so no barrier is needed before the next tile starts
        {
            // (the readout's lanes are wave 7's: the wave that does it enters the next tile's walk late, and wave 0 -- the natural choice --
            // owns the tile's longest rows and deals LDS-DMA in the MLP steps: launch -0.4 %)
            if (out != nullptr && wave == GR_WAVES - 1) {  // out == null: multi-task readout, done by the caller from the hout rows
                for (int gi = cur.g0 + lane; gi < cur.g1; gi += 64) {  // (a tile of one-node graphs has up to GR_ROWS of them)
                    // a range of graphs, or (bin-packed tiles, ENC form only) list positions: the graph's id and its first row inside the tile
                    const int gph = list ? list[gi] : gi;
                    const int n0 = node_off[gph], n1 = node_off[gph + 1];
                    const float sum = lds_sum_in_order(s_dot + (list ? lrow[gi] : n0 - cur.t0), n1 - n0);
                    out[gph] = sum / (float)(n1 - n0) + pool_b[0] + head_c;
                }
            }
        }
        if (!has_next) break;
        tile = ntile;
        cur = nxt;
    }
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
    if constexpr (PROF) {  // per-wave phase totals in 10 ns ticks (s_memrealtime)
        if (lane == 0) {
            for (int i = 0; i < 6; i++) prof_out[((size_t)blockIdx.x * GR_WAVES + wave) * 7 + i] = tacc[i];
            prof_out[((size_t)blockIdx.x * GR_WAVES + wave) * 7 + 6] = wall_clock64() - tk0;
        }
    }
}


#ifdef FLOWGNN_DEV
#include "dev/gin_pp_device.inc"  // gin_pp_kernel: the ping-pong form, measured slower -- development builds only
#endif

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

size_t gin_resident_layer_bytes() { return (size_t)GS_STEPS * GRC_CHUNK_STRIDE; }

void gin_resident_head_fold(const float* w1_last, const float* w2_last, const float* b2_last, const float* pool_w, float* out) {
    // the hidden tiles the kernel holds are scaled by s1 (the first layer's weights and bias are pre-scaled by that power of two)
    const double s1 = (double)pow2_scale(w1_last, (size_t)GS_H * GS_D);
    for (int k = 0; k < 208; k++) {
        double a = 0.0;
        if (k < GS_H)
            for (int d = 0; d < GS_D; d++) a += (double)w2_last[(size_t)d * GS_H + k] * (double)pool_w[d];
        out[k] = (float)(a / s1);
    }
    double c = 0.0;
    for (int d = 0; d < GS_D; d++) c += (double)b2_last[d] * (double)pool_w[d];
    out[208] = (float)c;
}

// merged = true (the resident kernel's stream): chunk 6 also carries the packed K-step of hidden units 192..199 -- output tiles 0..5 in
// the W1 slot of hidden tile 13 (pure padding: 200 hidden units are 12.5 tiles), output tile 6 (rows 96..99) as 16 lanes x 16 B behind
// the K tails -- and chunk 7 is empty: seven MLP steps (gr_step_final).  merged = false: the eight-chunk form gin_pp_pack_layer re-cuts.
void gin_resident_pack_layer(const float* w1, const float* b1, const float* w2, const float* b2, uint8_t* out, bool merged) {
    std::memset(out, 0, gin_resident_layer_bytes());
    const float s1 = pow2_scale(w1, (size_t)GS_H * GS_D);
    const float s2 = pow2_scale(w2, (size_t)GS_D * GS_H);
    auto w1s = [&](int o, int f) { return (o < GS_H && f < GS_D) ? w1[o * GS_D + f] * s1 : 0.0f; };
    auto w2s = [&](int d, int k) { return (d < GS_D && k < GS_H) ? w2[d * GS_H + k] * s2 : 0.0f; };
    auto hi16 = [](float v) { return (_Float16)v; };
    auto lo16 = [](float v) { const _Float16 h = (_Float16)v; return (_Float16)(v - (float)h); };
    auto put16 = [](uint8_t* p, _Float16 v) { std::memcpy(p, &v, 2); };
    for (int s = 0; s < GS_STEPS; s++) {
        uint8_t* ck = out + (size_t)s * GRC_CHUNK_STRIDE;
        if (s < GS_STEPS - 1) {
            for (int tl = 0; tl < 2; tl++) {
                const int t = 2 * s + tl;
                if (merged && t == 13) continue;  // hidden tile 13 does not exist (zeros): its slots in chunk 6 carry the packed K-step
                for (int lane = 0; lane < 64; lane++) {
                    const int i = lane & 15, gk = lane >> 4;
                    const int o = 16 * t + i;
                    for (int ks = 0; ks < 3; ks++)
                        for (int e = 0; e < 8; e++) {
                            const int f = 16 * (2 * ks + (e >> 2)) + 4 * gk + (e & 3);
                            put_split(ck + (size_t)tl * 6144 + (size_t)(ks * 2) * 1024, lane, e, w1s(o, f));
                        }
                    if (lane < 32) {  // K tail: g = 0 -> [w_hi(96..99), w_hi(96..99)], g = 1 -> [w_lo(96..99), 0]
                        uint8_t* tp = ck + GRC_TAIL_OFF + tl * 512 + lane * 16;
                        for (int e = 0; e < 4; e++) {
                            const float v = w1s(o, 96 + e);
                            if (gk == 0) { put16(tp + e * 2, hi16(v)); put16(tp + 8 + e * 2, hi16(v)); }
                            else { put16(tp + e * 2, lo16(v)); put16(tp + 8 + e * 2, (_Float16)0.0f); }
                        }
                    }
                }
                for (int x = 0; x < 16; x++) {
                    const int o = 16 * t + x;
                    const float b = o < GS_H ? b1[o] * s1 : 0.0f;
                    std::memcpy(ck + GRC_B1_OFF + tl * 64 + x * 4, &b, 4);
                }
            }
        }
        if (s >= 1 && s <= 6) {
            const int ks = s - 1;
            for (int t2 = 0; t2 < GS_T2; t2++)
                for (int lane = 0; lane < 64; lane++) {
                    const int i = lane & 15, gk = lane >> 4;
                    const int d = 16 * t2 + i;
                    for (int e = 0; e < 8; e++) {
                        const int k = 16 * (2 * ks + (e >> 2)) + 4 * gk + (e & 3);
                        put_split(ck + GRC_W2_OFF + (size_t)(t2 * 2) * 1024, lane, e, w2s(d, k));
                    }
                }
        } else if (s == 7) {  // packed K-step: hidden units 192..199
            for (int t2 = 0; t2 < GS_T2; t2++)
                for (int lane = 0; lane < 64; lane++) {
                    const int i = lane & 15, gk = lane >> 4;
                    const int d = 16 * t2 + i;
                    uint8_t* fp = ck + GRC_W2_OFF + (size_t)t2 * 1024 + lane * 16;
                    if (merged) {  // into chunk 6 (one chunk back)
                        uint8_t* c6 = out + (size_t)6 * GRC_CHUNK_STRIDE;
                        if (t2 < 6) fp = c6 + GRC_PK_OFF + (size_t)t2 * 1024 + lane * 16;
                        else if (i < 4) fp = c6 + GRC_PK6_OFF + ((gk << 2) | i) * 16;
                        else continue;  // rows 100..111 of the last output tile: zeros, not stored
                    }
                    const int k0 = 192 + 4 * (gk & 1);
                    for (int e = 0; e < 4; e++) {
                        const float v = w2s(d, k0 + e);
                        if (gk < 2) { put16(fp + e * 2, hi16(v)); put16(fp + 8 + e * 2, hi16(v)); }
                        else { put16(fp + e * 2, lo16(v)); put16(fp + 8 + e * 2, (_Float16)0.0f); }
                    }
                }
        } else {
            for (int x = 0; x < 16 * GS_T2; x++) {
                const float b = x < GS_D ? b2[x] * s1 * s2 : 0.0f;
                std::memcpy(ck + GRC_W2_OFF + x * 4, &b, 4);
            }
            const float os = 1.0f / (s1 * s2);
            std::memcpy(ck + GRC_W2_OFF + 112 * 4, &os, 4);
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

void launch_gin_resident(const float* h0, float* hout, const int* row_ptr, const int* src, const uint8_t* ecode, const float* ecomb_all,
                         const uint8_t* chunks_all, const float* pool_w, const float* pool_b, const int* tile_row, const int* tile_graph,
                         uint8_t* tile_desc, const int* node_off, float* out, int n_tiles, int* range_flag, hipStream_t s, bool hubs,
                         const float* head_u, int col_order, bool prof, const GinTileBuild* tb, int tstride) {
    if (n_tiles <= 0) return;
    const int order = hubs ? 3 : col_order;
    const bool fold = head_u != nullptr && out != nullptr && hout == nullptr;  // single-task readout, no per-node tap
    const bool enc = tb != nullptr && fold;  // descriptor + encoder indices straight from the caller's arrays, h_0 computed by the tile loader
    if (!enc) gin_tile_prep_kernel<<<n_tiles, 256, 0, s>>>(row_ptr, src, ecode, tile_row, tile_desc, n_tiles, order, tstride);
    const int grid = n_tiles < 256 ? n_tiles : 256;  // persistent: one 8-wave workgroup per CU (157 KB of LDS)
    unsigned long long* d = nullptr;
    const size_t cnt = (size_t)grid * GR_WAVES * 7;
    if (prof) {  // development aid: phase breakdown from s_memrealtime stamps, printed per launch (synchronises!)
        if (hipMalloc((void**)&d, cnt * 8) != hipSuccess) return;
        (void)hipMemsetAsync(d, 0, cnt * 8, s);
    }
    const uint32_t* eidx = enc ? reinterpret_cast<const uint32_t*>(tb->enc_idx) : nullptr;
    const float4* etab = enc ? reinterpret_cast<const float4*>(tb->enc_tab) : nullptr;
#define GR_LAUNCH(P, H, F, E)                                                                                                        \
    gin_resident_kernel<P, H, F, E><<<grid, GR_WAVES * 64, 0, s>>>(h0, hout, ecomb_all, chunks_all, pool_w, pool_b, tile_row, tile_graph, \
                                                                   tile_desc, node_off, out, n_tiles, range_flag, d, head_u, eidx, etab, tstride, \
                                                                   enc ? tb->list : nullptr, enc ? tb->lrow : nullptr)
#define GR_LAUNCH_FE(P, H)                                    \
    do {                                                      \
        if (enc) GR_LAUNCH(P, H, true, true);                 \
        else if (fold) GR_LAUNCH(P, H, true, false);          \
        else GR_LAUNCH(P, H, false, false);                   \
    } while (0)
    if (prof) {
        if (hubs) GR_LAUNCH_FE(true, true); else GR_LAUNCH_FE(true, false);
    } else {
        if (hubs) GR_LAUNCH_FE(false, true); else GR_LAUNCH_FE(false, false);
    }
#undef GR_LAUNCH_FE
#undef GR_LAUNCH
    if (prof) {
        std::vector<unsigned long long> hbuf(cnt);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hbuf.data(), d, cnt * 8, hipMemcpyDeviceToHost);
        (void)hipFree(d);
        double tot[7] = {0, 0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < cnt; i++) tot[i % 7] += (double)hbuf[i];
        const double nw = (double)grid * GR_WAVES;
        fprintf(stderr, "[gin_resident prof] tiles %d grid %d | per wave, us: gather %.1f  wait+barrier %.1f  mlp %.1f (of which step-end DMA wait %.1f, barrier %.1f)  epilogue+barrier %.1f  kernel %.1f\n",
                n_tiles, grid, tot[0] / nw / 100.0, tot[1] / nw / 100.0, tot[2] / nw / 100.0, tot[5] / nw / 100.0, tot[4] / nw / 100.0, tot[3] / nw / 100.0, tot[6] / nw / 100.0);
        {   // the same by wave index (0..7: which of the eight waves of a workgroup), gather and the wait behind it
            double gw[GR_WAVES] = {0}, ww[GR_WAVES] = {0};
            for (size_t i = 0; i < cnt; i += 7) { const int wi = (int)((i / 7) % GR_WAVES); gw[wi] += (double)hbuf[i]; ww[wi] += (double)hbuf[i + 1]; }
            fprintf(stderr, "[gin_resident prof] by wave index, us: gather");
            for (int wi = 0; wi < GR_WAVES; wi++) fprintf(stderr, " %.0f", gw[wi] / grid / 100.0);
            fprintf(stderr, " | wait behind it");
            for (int wi = 0; wi < GR_WAVES; wi++) fprintf(stderr, " %.0f", ww[wi] / grid / 100.0);
            double kmin = 1e300, kmax = 0.0;  // workgroup lifetimes (wave 0's): how long the launch waits for its last workgroup
            for (size_t i = 0; i < cnt; i += 7 * GR_WAVES) { const double k = (double)hbuf[i + 6]; kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; }
            fprintf(stderr, " | workgroup lifetime min %.0f max %.0f\n", kmin / 100.0, kmax / 100.0);
            double mw[GR_WAVES] = {0}, bw[GR_WAVES] = {0}, ew[GR_WAVES] = {0};  // MLP without its barriers, the MLP's step barriers, epilogue + barrier
            for (size_t i = 0; i < cnt; i += 7) {
                const int wi = (int)((i / 7) % GR_WAVES);
                mw[wi] += (double)hbuf[i + 2] - (double)hbuf[i + 4]; bw[wi] += (double)hbuf[i + 4]; ew[wi] += (double)hbuf[i + 3];
            }
            fprintf(stderr, "[gin_resident prof] by wave index, us: mlp compute");
            for (int wi = 0; wi < GR_WAVES; wi++) fprintf(stderr, " %.0f", mw[wi] / grid / 100.0);
            fprintf(stderr, " | mlp step barriers");
            for (int wi = 0; wi < GR_WAVES; wi++) fprintf(stderr, " %.0f", bw[wi] / grid / 100.0);
            fprintf(stderr, " | epilogue+barrier");
            for (int wi = 0; wi < GR_WAVES; wi++) fprintf(stderr, " %.0f", ew[wi] / grid / 100.0);
            fprintf(stderr, "\n");
        }
    }
}

#ifdef FLOWGNN_DEV
#include "dev/gin_pp_host.inc"
#endif

void launch_gin_tile_build(const GinTileBuild& tb, const int* tile_row, const int* tile_graph, uint8_t* tile_desc, int n_tiles, bool hubs,
                           int col_order, hipStream_t s) {
    if (n_tiles <= 0) return;
    gin_tile_build_kernel<<<n_tiles, 256, 0, s>>>(tb.batch, tile_row, tile_graph, tile_desc, reinterpret_cast<uint32_t*>(tb.enc_idx), n_tiles,
                                                  hubs ? 3 : col_order, tb.err, tb.list);
}

// the pre-combined encoder table of gin_tile_build_kernel / the resident kernel's tile loader (GRB_* layout above)
size_t gin_resident_enc_table_floats() { return (size_t)GRB_ROWS * GS_D; }
void gin_resident_pack_enc_table(const float* nemb /* [173][100] */, float* out) {
    static const int off[ND_FEATURE] = {0, 119, 123, 135, 147, 157, 163, 169, 171};  // load_inputs.cc:5
    static const int card[ND_FEATURE] = {119, 4, 12, 12, 10, 6, 6, 2, 2};            // host_load.cc:5
    auto E = [&](int k, int f, int d) { return nemb[(size_t)(off[k] + f) * GS_D + d]; };
    for (int f0 = 0; f0 < card[0]; f0++)
        for (int f1 = 0; f1 < card[1]; f1++)
            for (int d = 0; d < GS_D; d++) { float s = 0.0f; s += E(0, f0, d); s += E(1, f1, d); out[(size_t)(GRB_T01 + f0 * 4 + f1) * GS_D + d] = s; }
    for (int f2 = 0; f2 < card[2]; f2++)
        for (int f3 = 0; f3 < card[3]; f3++)
            for (int f4 = 0; f4 < card[4]; f4++)
                for (int d = 0; d < GS_D; d++) {
                    const float t34 = E(3, f3, d) + E(4, f4, d);
                    out[(size_t)(GRB_T234 + (f2 * 12 + f3) * 10 + f4) * GS_D + d] = E(2, f2, d) + t34;
                }
    for (int f5 = 0; f5 < card[5]; f5++)
        for (int f6 = 0; f6 < card[6]; f6++)
            for (int f7 = 0; f7 < card[7]; f7++)
                for (int f8 = 0; f8 < card[8]; f8++)
                    for (int d = 0; d < GS_D; d++)
                        out[(size_t)(GRB_T5678 + ((f5 * 6 + f6) * 2 + f7) * 2 + f8) * GS_D + d] = ((E(5, f5, d) + E(6, f6, d)) + E(7, f7, d)) + E(8, f8, d);
}

}  // namespace fg
