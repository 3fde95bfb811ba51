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
constexpr int GS_ECOMB_BYTES = EDGE_COMBOS * GS_D * 4;  // 24000 <= GS_CHUNK_BYTES: shares the odd-chunk buffer

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
        const auto hp_ = __builtin_amdgcn_cvt_pkrtz(a_, b_);                                                      \
        (HI) = __builtin_bit_cast(uint32_t, hp_);                                                                 \
        (LO) = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a_ - (float)hp_.x, b_ - (float)hp_.y));   \
    } while (0)

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
            r0.x = relu1(r0.x); r0.y = relu1(r0.y); r0.z = relu1(r0.z); r0.w = relu1(r0.w);
            r1.x = relu1(r1.x); r1.y = relu1(r1.y); r1.z = relu1(r1.z); r1.w = relu1(r1.w);
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
                    if (relu_out) { r.x = relu1(r.x); r.y = relu1(r.y); r.z = relu1(r.z); r.w = relu1(r.w); }
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
                if (relu_out) { r.x = relu1(r.x); r.y = relu1(r.y); r.z = relu1(r.z); r.w = relu1(r.w); }
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

// workgroup barrier without the fence of __syncthreads(): the fence makes hipcc wait for vmcnt(0) first, and the
// kernels below keep transfers in flight across barriers on purpose (LDS visibility of a landed DMA needs no fence)
#define GSP_BAR()                                          \
    do {                                                   \
        __builtin_amdgcn_sched_barrier(0);                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                      \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)

// gs_step for one node tile per wave, written as an explicit software pipeline: the fragments of group k+1 are
// read from LDS while the MFMAs of group k issue, and scheduling fences keep the compiler from hoisting all 26
// fragment reads of a step to its top (104 registers; with the 168 available at 3 waves per SIMD that spilled).
// Groups: MLP1 K-steps 0,1,2 (4 fragments, 6 MFMAs each), the fp32 K-tail (2 MFMAs), MLP2 output-tile pairs
// (0,1) (2,3) (4,5) (4 fragments, 6 MFMAs) and tile 6 (2 fragments, 3 MFMAs).  The relu + split of the new hidden
// tiles is placed under the MLP2 MFMAs, which do not depend on it.
#define GS_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void gs_ld4(uint4_t (&f)[4], const char* p, int lane) {
#pragma unroll
    for (int i = 0; i < 4; i++) f[i] = *reinterpret_cast<const uint4_t*>(p + i * 1024 + lane * 16);
}
// two accumulators (two hidden tiles / two output tiles) x {hi hi, hi lo, lo hi}; f = {tile0 hi, tile0 lo, tile1 hi, tile1 lo}
__device__ __forceinline__ void gs_mm2(const uint4_t (&f)[4], const uint4_t& bh, const uint4_t& bl, float4_t& c0, float4_t& c1) {
    c0 = GS_MFMA16(f[0], bh, c0); c1 = GS_MFMA16(f[2], bh, c1);
    c0 = GS_MFMA16(f[0], bl, c0); c1 = GS_MFMA16(f[2], bl, c1);
    c0 = GS_MFMA16(f[1], bh, c0); c1 = GS_MFMA16(f[3], bh, c1);
}

// The step also issues the LDS-DMA of a later chunk, one 1 KiB piece after each of its first three groups rather than
// three in a row at the top of the step: the vector-memory path takes 64 B per clock and CU, so twelve waves issuing 36
// pieces together sat in front of a full queue for about 900 cycles per step -- with their MFMAs behind it.
// LDS-DMA with a scalar base and a 32-bit per-lane offset ("saddr" addressing): no 64-bit per-lane addresses, which
// hipcc otherwise precomputes for every (step, piece) pair at the top of the tile loop and spills.
__device__ __forceinline__ void gs_dma16s(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

struct GsDma {
    const uint8_t* g;  // global address of piece 0 of the chunk (wave-uniform)
    uint32_t lane_off; // lane * 16
    uint32_t l;        // LDS address of piece 0 of the destination buffer (wave-uniform)
    int p0, p1, p2;    // this wave's three pieces (wave-uniform)
    bool on;
    // step 1 of the tile-staged kernel: 7 more pieces (rows of the next tile), issued after the chunk's
    bool rows;
    const char* rbase;   // first row of the tile in h (wave-uniform)
    uint32_t rlim;       // last valid 16-byte offset from rbase (bytes past the end of h are read from its last 16 bytes)
    int rwave, rwaves, rpieces;
    uint32_t rl;         // LDS address of row piece 0
};
__device__ __forceinline__ void gs_dma_row(const GsDma& d, int k) {
    if (d.rows) {
        int piece = d.rwave + d.rwaves * k;
        if (piece >= d.rpieces) piece = d.rwave;  // past the end: this wave's first piece again (same bytes)
        uint32_t off = piece * 1024 + d.lane_off;
        off = off < d.rlim ? off : d.rlim;
        gs_dma16s(d.rbase, off, d.rl + piece * 1024);
    }
}
__device__ __forceinline__ void gs_dma_piece(const GsDma& d, int piece) {
    if (d.on) gs_dma16s(d.g + piece * 1024, d.lane_off, d.l + piece * 1024);
}

template <int S>
__device__ __forceinline__ void gs_step_p(const char* wb, int lane, int g, const uint4_t (&in_hi)[1][3],
                                          const uint4_t (&in_lo)[1][3], const float (&in_t)[1], uint4_t (&h_hi)[1],
                                          uint4_t (&h_lo)[1], float4_t (&acc2)[1][GS_T2], float& vmax, const GsDma& dma) {
    constexpr bool M1 = S < GS_STEPS - 1, M2 = S > 0;
    uint4_t fa[4], fb[4];
    float4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    float t0 = 0.f, t1 = 0.f;
    const char* w2 = wb + GS_W2_OFF;
    if (M1) {
        const float4 b0 = *reinterpret_cast<const float4*>(wb + GS_B1_OFF + g * 16);
        const float4 b1 = *reinterpret_cast<const float4*>(wb + GS_B1_OFF + 64 + g * 16);
        a0 = (float4_t){b0.x, b0.y, b0.z, b0.w};
        a1 = (float4_t){b1.x, b1.y, b1.z, b1.w};
        gs_ld4(fa, wb, lane);
        GS_FENCE();
        gs_ld4(fb, wb + 4096, lane);
        gs_mm2(fa, in_hi[0][0], in_lo[0][0], a0, a1);
        gs_dma_piece(dma, dma.p0);
        GS_FENCE();
        gs_ld4(fa, wb + 8192, lane);
        gs_mm2(fb, in_hi[0][1], in_lo[0][1], a0, a1);
        gs_dma_piece(dma, dma.p1);
        GS_FENCE();
        t0 = *reinterpret_cast<const float*>(wb + GS_TAIL_OFF + lane * 4);
        t1 = *reinterpret_cast<const float*>(wb + GS_TAIL_OFF + 256 + lane * 4);
        if (M2) gs_ld4(fb, w2, lane);
        gs_mm2(fa, in_hi[0][2], in_lo[0][2], a0, a1);
        gs_dma_piece(dma, dma.p2);
        GS_FENCE();
        a0 = GS_MFMA32(t0, in_t[0], a0);
        a1 = GS_MFMA32(t1, in_t[0], a1);
    } else {
        gs_ld4(fb, w2, lane);
        gs_dma_piece(dma, dma.p0);
        gs_dma_piece(dma, dma.p1);
        gs_dma_piece(dma, dma.p2);
        GS_FENCE();
    }
    uint4_t n_hi = {0, 0, 0, 0}, n_lo = {0, 0, 0, 0};
    if (M2) {
        gs_ld4(fa, w2 + 4096, lane);
        gs_mm2(fb, h_hi[0], h_lo[0], acc2[0][0], acc2[0][1]);
        if (S == 2) { gs_dma_row(dma, 0); gs_dma_row(dma, 1); }
        GS_FENCE();
        gs_ld4(fb, w2 + 8192, lane);
        gs_mm2(fa, h_hi[0], h_lo[0], acc2[0][2], acc2[0][3]);
        if (S == 2) { gs_dma_row(dma, 2); gs_dma_row(dma, 3); }
    }
    if (M1) {  // under the MLP2 MFMAs just issued
        a0.x = relu1(a0.x); a0.y = relu1(a0.y); a0.z = relu1(a0.z); a0.w = relu1(a0.w);
        a1.x = relu1(a1.x); a1.y = relu1(a1.y); a1.z = relu1(a1.z); a1.w = relu1(a1.w);
        vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, a0.x), a0.y);
        vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, a0.z), a0.w);
        vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, a1.x), a1.y);
        vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, a1.z), a1.w);
        asm volatile("" : "+v"(vmax));  // computed here (LLVM otherwise sinks the chain to the kernel's end and spills a0, a1)
        GS_SPLIT2(a0.x, a0.y, n_hi.x, n_lo.x);
        GS_SPLIT2(a0.z, a0.w, n_hi.y, n_lo.y);
        GS_SPLIT2(a1.x, a1.y, n_hi.z, n_lo.z);
        GS_SPLIT2(a1.z, a1.w, n_hi.w, n_lo.w);
    }
    if (M2) {
        GS_FENCE();
        fa[0] = *reinterpret_cast<const uint4_t*>(w2 + 12288 + lane * 16);
        fa[1] = *reinterpret_cast<const uint4_t*>(w2 + 13312 + lane * 16);
        gs_mm2(fb, h_hi[0], h_lo[0], acc2[0][4], acc2[0][5]);
        if (S == 2) { gs_dma_row(dma, 4); gs_dma_row(dma, 5); }
        GS_FENCE();
        acc2[0][6] = GS_MFMA16(fa[0], h_hi[0], acc2[0][6]);
        acc2[0][6] = GS_MFMA16(fa[0], h_lo[0], acc2[0][6]);
        acc2[0][6] = GS_MFMA16(fa[1], h_hi[0], acc2[0][6]);
        if (S == 2) gs_dma_row(dma, 6);
    }
    if (M1) { h_hi[0] = n_hi; h_lo[0] = n_lo; }
}

// ---------------------------------------------------------------- persistent, tile-staged variant (FLOWGNN_GIN_SPLIT_NT=3)
// The two kernels above leave the matrix pipe idle most of the time for the same reason: every wave waits for
// global memory inside its own critical path -- gin_layer_split_kernel in a per-tile gather prologue (three dependent
// round trips row_ptr -> src -> h[u] with 12 waves per CU to hide them), a register-pipelined variant (removed) at the end of every step
// (a step lasts about 1 us, a gather round trip under load about 2 us, so the step becomes the round trip).  Here
// global memory is touched only by LDS-DMA transfers that are issued a whole step or more before anything depends
// on them, and all of them are contiguous:
//   * one persistent 12-wave workgroup per CU walks tiles of 192 consecutive nodes;
//   * while the MLP of tile t runs, the 192 rows of tile t+1 (76.8 KB), its row_ptr slice and the CSR entries of its
//     rows (src ids and edge codes: contiguous ranges of the CSR arrays) are copied into LDS.  Molecule batches are
//     block diagonal with consecutive node ids, so a node's neighbours are almost always rows of its own tile; the
//     rare exception (a graph straddling a tile boundary) is fetched from global memory.  (Per-lane loads of the
//     CSR entries were tried first: 96 divergent load instructions per tile kept the CU's vector-memory pipeline,
//     and the waves queued behind it, busy for 3400 cycles.)
//   * the gather of tile t+1 is LDS-only and is folded into steps 3..7 of tile t, one in-edge per step, after the
//     wave's MFMAs of that step (the VALU work of one wave runs under the MFMAs of the other two on its SIMD);
//     in-degrees above 5 and the node's own row finish in a short loop at the end of the tile;
//   * the 8 weight-stream steps are those of gin_layer_split_kernel (two LDS buffers, one step ahead); the DMA pieces
//     of a step are issued between its MFMA groups, not in a burst at its top (the vector-memory path moves 64 B
//     per clock and CU; twelve waves issuing 36 KiB together queued for ~900 cycles with their MFMAs behind them).
// Waits: every step ends with s_waitcnt vmcnt(0) + one workgroup barrier.  Partial waits ("vmcnt(7): everything but the
// seven row pieces issued last") were tried and are NOT safe: now and then a wave passed one with a weight piece still
// in flight (17 of 4113 molhiv graphs wrong), i.e. LDS-DMA transfers of a wave do not retire strictly in issue order.
// The DMA is issued from inline asm (see gs_dma16s), so hipcc neither counts it nor waits for it, and the end-of-step
// waits are the s_waitcnt BUILTIN so that hipcc's own bookkeeping of its loads and stores is reset at the same points.
constexpr int GT_WAVES = 12;
constexpr int GT_TILE = GT_WAVES * 16;            // 192 rows
constexpr int GT_ROW_BYTES = GT_TILE * GS_D * 4;  // 76800 = 75 pieces of 1 KiB
constexpr int GT_ECAP = GT_WAVES * 64;            // CSR entries of a tile staged in LDS (a molhiv tile has ~420)
#define GT_R 7  // row pieces per wave and tile: ceil(75 / 12)
static_assert(3 * GT_WAVES >= GS_CHUNK_STRIDE / 1024 && GT_R * GT_WAVES >= GT_ROW_BYTES / 1024, "piece counts");
#define GT_STR2(x) #x
#define GT_STR(x) GT_STR2(x)
// LDS map (one object, carved by hand, weight buffers first: their fragment reads are then "lane * 16 + immediate";
// ds_read offsets are 16 bit, and above 64 KB every fragment needed its own address register)
constexpr int GT_OFF_WA = 0;                               // even chunks
constexpr int GT_OFF_WB = GS_CHUNK_STRIDE;                 // odd chunks
constexpr int GT_OFF_ECOMB = 2 * GS_CHUNK_STRIDE;          // 60 edge-embedding combos
constexpr int GT_OFF_ROWS = GT_OFF_ECOMB + GS_ECOMB_BYTES;  // h rows of the next / current tile
constexpr int GT_OFF_RP = GT_OFF_ROWS + GT_ROW_BYTES;      // row_ptr[tile_base .. +255]
constexpr int GT_OFF_SRC = GT_OFF_RP + 1024;               // src[e0 .. e0 + GT_ECAP)
constexpr int GT_OFF_CODE = GT_OFF_SRC + GT_ECAP * 4;      // ecode[e0 & ~3 .. +1024)
constexpr int GT_LDS_BYTES = GT_OFF_CODE + 1024;
static_assert(GT_LDS_BYTES <= 160 * 1024, "LDS budget");

// One LDS-DMA instruction (64 lanes x 16 B, lane-linear from M0) issued from inline asm.  hipcc then neither counts it
// nor orders LDS reads against it: with the builtin, its waitcnt pass put "s_waitcnt vmcnt(0)" in front of the first
// ds_read of every odd step (reads of the higher-addressed buffer while the DMA into the lower one was in flight),
// which is the very stall this kernel is built to avoid.
__device__ __forceinline__ void gt_dma16(const void* gaddr, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gaddr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint32_t gt_lds_addr(const void* p) {
    return (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ int load_u8_rare(const uint8_t* p) {
    int v;
    asm volatile("global_load_ubyte %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ void gt_issue_chunk(const uint8_t* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
    uint32_t lane_off = lane * 16;
    asm volatile("" : "+v"(lane_off));  // keep the per-lane addresses out of the tile loop's invariants (they spill)
#pragma unroll
    for (int p = 0; p < 3; p++) {
        int piece = wave + GT_WAVES * p;
        if (piece >= GS_CHUNK_STRIDE / 1024) piece = wave;  // past the end: this wave's first piece again (same bytes)
        gt_dma16(gchunk + piece * 1024 + lane_off, __builtin_amdgcn_readfirstlane(gt_lds_addr(lds_buf) + piece * 1024));
    }
}

// this lane's global address of row piece `piece` of tile `tile`; bytes past the end of h are read from its last 16
// bytes instead (those LDS rows belong to no node)
__device__ __forceinline__ const char* gt_row_addr(const float* __restrict__ h, long long tile, int n_tot, int piece, uint32_t lane_off) {
    const long long lim = (long long)n_tot * (GS_D * 4) - 16;
    long long off = tile * GT_ROW_BYTES + piece * 1024 + lane_off;
    off = off < lim ? off : lim;
    return reinterpret_cast<const char*>(h) + off;
}
__device__ __forceinline__ int gt_row_piece(int wave, int k) {
    const int piece = wave + GT_WAVES * k;
    return piece < GT_ROW_BYTES / 1024 ? piece : wave;
}

// row_ptr slice, src ids and edge codes of tile `tile` (first CSR entry e0) -> LDS; 1 or 2 instructions per wave.
// Scalar base + 32-bit lane offset; lanes past the end of an array re-read its last element (entries of no row).
__device__ __forceinline__ void gt_dma4s(const void* sbase, uint32_t voff, uint32_t lds_addr) {  // 64 lanes x 4 B
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void gt_issue_csr(const int* __restrict__ row_ptr, const int* __restrict__ src,
                                             const uint8_t* __restrict__ ecode, char* smem, long long tile, int e0, int n_tot,
                                             int e_tot, int wave, int lane) {
    {
        const int eb = e0 < e_tot ? e0 : e_tot - 1;  // wave-uniform
        const uint32_t last = (uint32_t)(e_tot - 1 - eb) * 4u;
        uint32_t off = (uint32_t)(wave * 64 + lane) * 4u;
        off = off < last ? off : last;
        gt_dma4s(src + eb, off, __builtin_amdgcn_readfirstlane(gt_lds_addr(smem + GT_OFF_SRC) + wave * 256));
    }
    if (wave < 4) {
        const int ab = (e0 < e_tot ? e0 : e_tot - 1) & ~3;
        const uint32_t last = (uint32_t)(((e_tot - 1) & ~3) - ab);
        uint32_t off = (uint32_t)(wave * 64 + lane) * 4u;
        off = off < last ? off : last;
        gt_dma4s(ecode + ab, off, __builtin_amdgcn_readfirstlane(gt_lds_addr(smem + GT_OFF_CODE) + wave * 256));
    } else if (wave < 8) {
        const long long nb = tile * GT_TILE;
        const uint32_t last = (uint32_t)(n_tot - nb) * 4u;  // row_ptr has n_tot + 1 entries
        uint32_t off = (uint32_t)((wave - 4) * 64 + lane) * 4u;
        off = off < last ? off : last;
        gt_dma4s(row_ptr + nb, off, __builtin_amdgcn_readfirstlane(gt_lds_addr(smem + GT_OFF_RP) + (wave - 4) * 256));
    }
}

// One in-edge of the node this lane gathers for: bqn += relu(h[src] + ecomb[code]); everything from LDS except
// neighbours outside the tile and CSR entries beyond the staged GT_ECAP (dense tiles: kNN graphs).
// LDS reads are unconditional (clamped) and pinned with an empty asm, global memory is touched only in branches through
// asm loads: a `cond ? lds : global` select makes hipcc emit flat loads with a full wait after each.
#define GT_ROUND()                                                                                                     \
    do {                                                                                                               \
        if (ecur < eend) {                                                                                             \
            const int ei_ = ecur < GT_ECAP ? ecur : GT_ECAP - 1;                                                       \
            int u_ = s_src[ei_];                                                                                       \
            int code_ = s_code[coff + ei_];                                                                            \
            asm volatile("" : "+v"(u_), "+v"(code_));                                                                  \
            if (ecur >= GT_ECAP) {                                                                                     \
                u_ = load_i32_rare(src + (size_t)e0n + ecur);                                                          \
                code_ = load_u8_rare(ecode + (size_t)e0n + ecur);                                                      \
            }                                                                                                          \
            ecur++;                                                                                                    \
            const unsigned ul_ = (unsigned)(u_ - nbase);                                                               \
            const bool in_ = ul_ < (unsigned)GT_TILE;                                                                  \
            const float* ur_ = s_rows + (in_ ? ul_ : 0u) * GS_D + 4 * g;                                               \
            const float* er_ = s_ecomb + code_ * GS_D + 4 * g;                                                         \
            float4 x_[6];                                                                                              \
            _Pragma("unroll") for (int q = 0; q < 6; q++) {                                                            \
                x_[q] = *reinterpret_cast<const float4*>(ur_ + 16 * q);                                                \
                asm volatile("" : "+v"(x_[q].x), "+v"(x_[q].y), "+v"(x_[q].z), "+v"(x_[q].w));                         \
            }                                                                                                          \
            float xt_ = ur_[96 - 3 * g];                                                                               \
            asm volatile("" : "+v"(xt_));                                                                              \
            if (!in_) {                                                                                                \
                const float* gr_ = h + (size_t)u_ * GS_D + 4 * g;                                                      \
                _Pragma("unroll") for (int q = 0; q < 6; q++)                                                          \
                    x_[q] = load_f4_rare(reinterpret_cast<const float4*>(gr_ + 16 * q));                               \
                xt_ = load_f32_rare(h + (size_t)u_ * GS_D + 96 + g);                                                   \
            }                                                                                                          \
            _Pragma("unroll") for (int q = 0; q < 6; q++) {                                                            \
                const float4 w_ = *reinterpret_cast<const float4*>(er_ + 16 * q);                                      \
                bqn[4 * q + 0] += relu1(w_.x + x_[q].x);                                                               \
                bqn[4 * q + 1] += relu1(w_.y + x_[q].y);                                                               \
                bqn[4 * q + 2] += relu1(w_.z + x_[q].z);                                                               \
                bqn[4 * q + 3] += relu1(w_.w + x_[q].w);                                                               \
            }                                                                                                          \
            bqn[24] += relu1(er_[96 - 3 * g] + xt_);                                                                   \
        }                                                                                                              \
    } while (0)

// start of a gather: this lane's CSR row bounds (relative to the tile's first entry) from the staged row_ptr slice
#define GT_GATHER_INIT(valid_)                                                           \
    do {                                                                                 \
        const int nl_ = wave * 16 + j;                                                   \
        const int r0_ = s_rp[nl_], r1_ = s_rp[nl_ + 1];                                  \
        ecur = (valid_) ? r0_ - e0n : 0;                                                 \
        eend = (valid_) ? r1_ - e0n : 0;                                                 \
        _Pragma("unroll") for (int k = 0; k < 25; k++) bqn[k] = 0.0f;                    \
    } while (0)

// end of a gather: remaining in-edges, then + (1 + eps) h[v], eps == 0 (the oracle's order: edges, then the own row)
#define GT_GATHER_FINISH()                                                               \
    do {                                                                                 \
        while (__any(ecur < eend)) GT_ROUND();                                           \
        const float* sr_ = s_rows + (wave * 16 + j) * GS_D + 4 * g;                      \
        _Pragma("unroll") for (int q = 0; q < 6; q++) {                                  \
            const float4 x_ = *reinterpret_cast<const float4*>(sr_ + 16 * q);            \
            bqn[4 * q + 0] += x_.x; bqn[4 * q + 1] += x_.y; bqn[4 * q + 2] += x_.z; bqn[4 * q + 3] += x_.w; \
        }                                                                                \
        bqn[24] += sr_[96 - 3 * g];                                                      \
    } while (0)

// The end-of-step wait is the BUILTIN s_waitcnt vmcnt(0) (0x0F70: expcnt and lgkmcnt untouched), not inline asm: hipcc's
// waitcnt pass then knows that nothing of its own is pending any more.  With an asm wait it kept the tile's output
// stores on its books and, when it reused their data registers early in the next step 0, inserted "s_waitcnt vmcnt(1)"
// -- which, because it cannot see the asm-issued DMA, was a wait for the chunk that had just been requested.
#define GT_WAIT_ALL()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_waitcnt(0x0F70);    \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

__global__ __launch_bounds__(GT_WAVES * 64) void gin_layer_split_tiled_kernel(
    const float* __restrict__ h, float* __restrict__ hout, const int* __restrict__ row_ptr, const int* __restrict__ src,
    const uint8_t* __restrict__ ecode, const float* __restrict__ ecomb, const uint8_t* __restrict__ wchunks, int n_tot,
    int e_tot, int n_tiles, int relu_out, int* __restrict__ range_flag) {
    __shared__ __attribute__((aligned(16))) char smem[GT_LDS_BYTES];
    char* const s_wa = smem + GT_OFF_WA;
    char* const s_wb = smem + GT_OFF_WB;
    float* const s_ecomb = reinterpret_cast<float*>(smem + GT_OFF_ECOMB);
    float* const s_rows = reinterpret_cast<float*>(smem + GT_OFF_ROWS);
    const int* const s_rp = reinterpret_cast<const int*>(smem + GT_OFF_RP);
    const int* const s_src = reinterpret_cast<const int*>(smem + GT_OFF_SRC);
    const uint8_t* const s_code = reinterpret_cast<const uint8_t*>(smem + GT_OFF_CODE);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    int tile = blockIdx.x;  // wave-uniform, as is everything derived from it
    if (tile >= n_tiles) return;
    uint32_t lane_off = lane * 16;
    asm volatile("" : "+v"(lane_off));  // keeps 64-bit per-lane DMA addresses from becoming (spilled) loop invariants

    // ---- prologue: chunk 0, rows + CSR of the first tile, edge-embedding combos; then its gather, not overlapped
    float bqn[25];
    int ecur, eend;
    {
        const int e0n = row_ptr[(long long)tile * GT_TILE];
        const int nbase = tile * GT_TILE;
        const int coff = e0n & 3;
        gt_issue_chunk(wchunks, s_wa, wave, lane);
        gt_issue_csr(row_ptr, src, ecode, smem, tile, e0n, n_tot, e_tot, wave, lane);
#pragma unroll
        for (int k = 0; k < GT_R; k++) {
            const int piece = gt_row_piece(wave, k);
            gt_dma16(gt_row_addr(h, tile, n_tot, piece, lane_off), __builtin_amdgcn_readfirstlane(gt_lds_addr(s_rows) + piece * 1024));
        }
        for (int i = threadIdx.x; i < GS_ECOMB_BYTES / 16; i += GT_WAVES * 64)
            reinterpret_cast<float4*>(s_ecomb)[i] = reinterpret_cast<const float4*>(ecomb)[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GT_GATHER_INIT((long long)nbase + wave * 16 + j < n_tot);
        GT_GATHER_FINISH();
    }

    float vmax = 0.0f;
    while (true) {
        // B operands of this tile's first linear layer from the finished gather
        uint4_t in_hi[1][3], in_lo[1][3];
        float in_t[1];
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            GS_SPLIT2(bqn[8 * ks + 0], bqn[8 * ks + 1], in_hi[0][ks].x, in_lo[0][ks].x);
            GS_SPLIT2(bqn[8 * ks + 2], bqn[8 * ks + 3], in_hi[0][ks].y, in_lo[0][ks].y);
            GS_SPLIT2(bqn[8 * ks + 4], bqn[8 * ks + 5], in_hi[0][ks].z, in_lo[0][ks].z);
            GS_SPLIT2(bqn[8 * ks + 6], bqn[8 * ks + 7], in_hi[0][ks].w, in_lo[0][ks].w);
        }
#pragma unroll
        for (int k = 0; k < 24; k += 2)
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(bqn[k])), __builtin_fabsf(bqn[k + 1]));
        asm volatile("" : "+v"(vmax));  // here, not sunk to its next use: that kept all of bqn live across step 0 (spills)
        in_t[0] = bqn[24];
        float4_t acc2[1][GS_T2];
#pragma unroll
        for (int t2 = 0; t2 < GS_T2; t2++) {
            const float4 b = *reinterpret_cast<const float4*>(s_wa + GS_W2_OFF + (16 * t2 + 4 * g) * 4);  // chunk 0 is resident
            acc2[0][t2] = (float4_t){b.x, b.y, b.z, b.w};
        }
        const float oscale = *reinterpret_cast<const float*>(s_wa + GS_W2_OFF + 112 * 4);
        uint4_t h_hi[1], h_lo[1];
        h_hi[0] = (uint4_t){0, 0, 0, 0};
        h_lo[0] = (uint4_t){0, 0, 0, 0};

        const int next = tile + gridDim.x;
        const bool has_next = next < n_tiles;  // workgroup-uniform
        const int nbase = next * GT_TILE;
        const bool nvalid = has_next && (long long)nbase + wave * 16 + j < n_tot;
        const int e0n = has_next ? row_ptr[(long long)nbase] : 0;  // scalar load: first CSR entry of the next tile
        const int coff = e0n & 3;

        GsDma dma;
        dma.on = true;
        dma.p0 = wave; dma.p1 = wave + GT_WAVES;
        dma.p2 = wave + 2 * GT_WAVES < GS_CHUNK_STRIDE / 1024 ? wave + 2 * GT_WAVES : wave;
        dma.rows = false;
        dma.lane_off = lane_off;
        const uint8_t* const wl = wchunks;
        const uint32_t la = gt_lds_addr(s_wa), lb = gt_lds_addr(s_wb);

        // ---- step 0: chunk 1 -> B
        GT_WAIT_ALL();  // the previous tile's stores (long done): nothing of hipcc's own may be pending when DMA is in flight
        dma.g = wl + 1 * GS_CHUNK_STRIDE; dma.l = lb;
        gs_step_p<0>(s_wa, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax, dma);
        GT_WAIT_ALL();
        GSP_BAR();  // every wave has finished the gather of this tile: rows and CSR staging may be overwritten
        // ---- step 1: CSR of the next tile; chunk 2 -> A
        if (has_next) gt_issue_csr(row_ptr, src, ecode, smem, next, e0n, n_tot, e_tot, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        dma.g = wl + 2 * GS_CHUNK_STRIDE; dma.l = la;
        gs_step_p<1>(s_wb, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax, dma);
        GT_WAIT_ALL();
        GSP_BAR();
        // ---- step 2: chunk 3 -> B; then the rows of the next tile, between the MLP2 groups
        dma.g = wl + 3 * GS_CHUNK_STRIDE; dma.l = lb;
        dma.rows = true;
        dma.rl = gt_lds_addr(s_rows);
        {
            const long long roff = (long long)(has_next ? next : tile) * GT_ROW_BYTES;
            const long long rest = (long long)n_tot * (GS_D * 4) - 16 - roff;  // >= 0: the tile has at least one row
            dma.rbase = reinterpret_cast<const char*>(h) + roff;
            dma.rlim = rest < GT_ROW_BYTES ? (uint32_t)rest : (uint32_t)GT_ROW_BYTES;
        }
        dma.rwave = wave; dma.rwaves = GT_WAVES; dma.rpieces = GT_ROW_BYTES / 1024;
        gs_step_p<2>(s_wa, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax, dma);
        dma.rows = false;
        GT_WAIT_ALL();
        GSP_BAR();
        // ---- steps 3..7: one in-edge of the next tile's gather after the MFMAs of each
        GT_GATHER_INIT(nvalid);
#define GT_STEP(S, CUR)                                                                               \
    dma.g = wl + (size_t)(((S) + 1) & 7) * GS_CHUNK_STRIDE; dma.l = ((S) & 1) ? la : lb;              \
    gs_step_p<S>(CUR, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax, dma);                      \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    GT_ROUND()
        GT_STEP(3, s_wb); GT_WAIT_ALL(); GSP_BAR();
        GT_STEP(4, s_wa); GT_WAIT_ALL(); GSP_BAR();
        GT_STEP(5, s_wb); GT_WAIT_ALL(); GSP_BAR();
        GT_STEP(6, s_wa); GT_WAIT_ALL(); GSP_BAR();
        GT_STEP(7, s_wb);  // chunk 0 again, for the next tile
#undef GT_STEP
        {   // the tile's outputs, stored inside step 7 so that its wait overlaps them with the other waves' work
            const long long node = (long long)tile * GT_TILE + wave * 16 + j;
            if (node < n_tot) {
                float* row = hout + (size_t)node * GS_D;
#pragma unroll
                for (int t2 = 0; t2 < GS_T2; t2++) {
                    const int col = 16 * t2 + 4 * g;
                    if (col < GS_D) {
                        float4_t r = acc2[0][t2] * oscale;
                        if (relu_out) { r.x = relu1(r.x); r.y = relu1(r.y); r.z = relu1(r.z); r.w = relu1(r.w); }
                        *reinterpret_cast<float4*>(row + col) = make_float4(r.x, r.y, r.z, r.w);
                    }
                }
            }
        }
        GT_WAIT_ALL();
        GSP_BAR();
        if (!has_next) break;
        GT_GATHER_FINISH();
        tile = next;
    }
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
    if (pool_w && nt == 3) nt = 4;  // the tile-staged variant has no folded readout
    if (e_tot == 0 && nt == 3) nt = 1;  // the tile-staged kernel stages CSR entries unconditionally
    if (nt == 3) {
        const int n_tiles = (int)ceil_div_ll(n_tot, GT_TILE);
        const int grid = n_tiles < 256 ? n_tiles : 256;  // persistent: one 12-wave workgroup per CU
        gin_layer_split_tiled_kernel<<<grid, GT_WAVES * 64, 0, s>>>(h, hout, row_ptr, src, ecode, ecomb, chunks, n_tot, e_tot, n_tiles,
                                                                    relu_out, range_flag);
        return;
    }
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
