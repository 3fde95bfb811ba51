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
#include <vector>

#include "device_common.h"
#include "gin_pipe.h"

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

__device__ __forceinline__ void gs_issue_chunk(const uint8_t* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < 7; p++) {
        const int piece = wave + 4 * p;  // 26 full pieces of 1 KiB + 640 B
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

template <int NT>
__global__ __launch_bounds__(256) void gin_layer_split_kernel(const float* __restrict__ h, float* __restrict__ hout,
                                                               const int* __restrict__ row_ptr,
                                                               const int* __restrict__ src,
                                                               const uint8_t* __restrict__ ecode,
                                                               const float* __restrict__ ecomb,
                                                               const uint8_t* __restrict__ wchunks, int n_tot, int relu_out,
                                                               int* __restrict__ range_flag, int abl) {
    // two DISTINCT LDS objects: the compiler can then prove that the LDS-DMA into one does not alias the ds_reads
    // of the other and leaves the DMA in flight under the MFMAs (see gin_layer_fused_kernel)
    __shared__ __attribute__((aligned(16))) char s_a[GS_CHUNK_BYTES];  // edge-embedding combos, then odd chunks
    __shared__ __attribute__((aligned(16))) char s_b[GS_CHUNK_BYTES];  // even chunks
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // in an SGPR: DMA addresses = scalar base + lane * 16
    const int j = lane & 15, g = lane >> 4;
    const long long node_base = (long long)blockIdx.x * (64 * NT) + wave * (16 * NT);

    gs_issue_chunk(wchunks, s_b, wave, lane);  // chunk 0 in flight while we gather
    for (int i = threadIdx.x; i < GS_ECOMB_BYTES / 16; i += 256)
        reinterpret_cast<float4*>(s_a)[i] = reinterpret_cast<const float4*>(ecomb)[i];
    __syncthreads();
    const float* s_ecomb = reinterpret_cast<const float*>(s_a);

    // ---- gather (MP unit): a = h[v] + sum_e relu(h[src_e] + ecomb[code_e]), CSR order
    float bq[NT][25];
    int e_cur[NT], e_end[NT], u_nx[NT], c_nx[NT];
    long long self_row[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        long long node = node_base + nt * 16 + j;
        const bool valid = node < n_tot;
        if (!valid) node = n_tot - 1;
        self_row[nt] = node;
        e_cur[nt] = valid ? row_ptr[node] : 0;
        e_end[nt] = valid ? row_ptr[node + 1] : 0;
        if (abl & 1) e_end[nt] = e_cur[nt];
#pragma unroll
        for (int k = 0; k < 25; k++) bq[nt][k] = 0.0f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {  // indices one edge ahead of the feature gathers
        const bool on = e_cur[nt] < e_end[nt];
        u_nx[nt] = on ? src[e_cur[nt]] : 0;
        c_nx[nt] = on ? ecode[e_cur[nt]] : 0;
    }
    while (true) {
        bool any = false;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) any |= (e_cur[nt] < e_end[nt]);
        if (!__any(any)) break;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            if (e_cur[nt] < e_end[nt]) {
                const int u = u_nx[nt];
                const int code = c_nx[nt];
                e_cur[nt]++;
                if (e_cur[nt] < e_end[nt]) {
                    u_nx[nt] = src[e_cur[nt]];
                    c_nx[nt] = ecode[e_cur[nt]];
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
    float vmax = 0.0f;
    uint4_t in_hi[NT][3], in_lo[NT][3];
    float in_t[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {  // + (1 + eps) h[v], eps == 0; then split into the MLP1 B operands
        const float* hr = h + (size_t)self_row[nt] * GS_D + 4 * g;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 x = *reinterpret_cast<const float4*>(hr + 16 * q);
            bq[nt][4 * q + 0] += x.x; bq[nt][4 * q + 1] += x.y; bq[nt][4 * q + 2] += x.z; bq[nt][4 * q + 3] += x.w;
        }
        bq[nt][24] += h[(size_t)self_row[nt] * GS_D + 96 + g];
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
    for (int c = 0; c < ((abl & 2) ? 0 : GS_STEPS); c += 2) {
        // even step: compute from s_b while chunk c+1 streams into s_a
        gs_issue_chunk(wchunks + (size_t)(c + 1) * GS_CHUNK_STRIDE, s_a, wave, lane);
        gs_step<NT>(s_b, c, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk c+1 have landed
        __syncthreads();                                  // everyone's landed; everyone is done with s_b
        // odd step: compute from s_a while chunk c+2 streams into s_b
        if (c + 2 < GS_STEPS) gs_issue_chunk(wchunks + (size_t)(c + 2) * GS_CHUNK_STRIDE, s_b, wave, lane);
        gs_step<NT>(s_a, c + 1, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

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
    // operands beyond the f16 range (inf after pkrtz is impossible, saturation is silent): tell the engine
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

// ---------------------------------------------------------------- persistent + software-pipelined variant
// Same math, but one persistent 12-wave workgroup per CU walks tiles of 192 nodes, and
//  * every wave gathers its NEXT tile while the MFMA steps of the current one run (gin_pipe.h: step 0 CSR row bounds,
//    step 1 own row, steps 2..7 one in-edge each; in-degrees above 6 finish in a residual loop), so HBM/L2 latency
//    hides behind the matrix pipe instead of being a per-tile prologue;
//  * the weight stream is read once per 192 nodes instead of once per 64 (7.6 GB instead of 23 GB of L2 -> LDS traffic
//    per layer at 2^18 molhiv graphs) and runs TWO steps ahead through four LDS buffers: with 16x faster MFMAs a
//    step lasts well under a microsecond, less than the latency of the LDS-DMA that feeds the next one.
// A step issues, in this order, its gather slice and then exactly GSP_PIECES LDS-DMA instructions per wave; its single wait
// is s_waitcnt vmcnt(GSP_PIECES): loads return in order, so everything older than this step's DMA -- the gather slice and the
// previous step's DMA, i.e. the chunk the next step reads -- has landed, while this step's DMA stays in flight.
#define GSP_PIECES 4  // LDS-DMA instructions per wave and chunk: ceil(27 / GSP_WAVES)
#define GSP_STR2(x) #x
#define GSP_STR(x) GSP_STR2(x)
constexpr int GSP_WAVES = 8;
static_assert(GSP_PIECES * GSP_WAVES >= GS_CHUNK_STRIDE / 1024 && (GSP_PIECES - 1) * GSP_WAVES < GS_CHUNK_STRIDE / 1024, "piece count");
constexpr int GSP_TILE = GSP_WAVES * 16;

__device__ __forceinline__ void gsp_issue_chunk(const uint8_t* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
    // opaque per call: otherwise the 64-bit per-lane addresses of all 8 x GSP_PIECES (chunk, piece) pairs become
    // loop invariants of the tile loop, spill, and their scratch reloads break the step's vmcnt accounting
    uint32_t lane_off = lane * 16;
    asm volatile("" : "+v"(lane_off));
#pragma unroll
    for (int p = 0; p < GSP_PIECES; p++) {
        int piece = wave + GSP_WAVES * p;              // 27 pieces of 1 KiB over the waves
        if (piece >= GS_CHUNK_STRIDE / 1024) piece = wave;  // past the end: this wave's first piece again (same bytes)
        const uint8_t* g = gchunk + piece * 1024 + lane_off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds_buf + piece * 1024), 16, 0, 0);
    }
}

#define GSP_BAR()                                          \
    do {                                                   \
        __builtin_amdgcn_sched_barrier(0);                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                      \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)

// step S of a tile: reads chunk S from BUF_CUR, streams chunk S+2 (of this or the next tile) into BUF_PF
#define GSP_STEP(S, BUF_CUR, BUF_PF)                                                                      \
    do {                                                                                                  \
        GIN_PIPE_ISSUE(S);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        gsp_issue_chunk(wchunks + (size_t)(((S) + 2) & 7) * GS_CHUNK_STRIDE, BUF_PF, wave, lane);         \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        gs_step<1>(BUF_CUR, S, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax);                      \
        GIN_PIPE_WAIT_N(GSP_STR(GSP_PIECES));                                                                             \
        GIN_PIPE_CONSUME();                                                                               \
        GSP_BAR();                                                                                        \
    } while (0)

__global__ __launch_bounds__(GSP_WAVES * 64) void gin_layer_split_persistent_kernel(
    const float* __restrict__ h, float* __restrict__ hout, const int* __restrict__ row_ptr, const int* __restrict__ src,
    const uint8_t* __restrict__ ecode, const float* __restrict__ ecomb, const uint8_t* __restrict__ wchunks, int n_tot,
    int n_tiles, int relu_out, int* __restrict__ range_flag, int getenv_abl) {
    // five DISTINCT LDS objects: the DMA into one weight buffer provably does not alias the ds_reads of another
    __shared__ __attribute__((aligned(16))) float s_ecomb[GS_ECOMB_BYTES / 4];
    __shared__ __attribute__((aligned(16))) char s_w0[GS_CHUNK_STRIDE];
    __shared__ __attribute__((aligned(16))) char s_w1[GS_CHUNK_STRIDE];
    __shared__ __attribute__((aligned(16))) char s_w2[GS_CHUNK_STRIDE];
    __shared__ __attribute__((aligned(16))) char s_w3[GS_CHUNK_STRIDE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // in an SGPR: DMA addresses = scalar base + lane * 16
    const int j = lane & 15, g = lane >> 4;
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;

    gsp_issue_chunk(wchunks, s_w0, wave, lane);
    gsp_issue_chunk(wchunks + GS_CHUNK_STRIDE, s_w1, wave, lane);
    for (int i = threadIdx.x; i < GS_ECOMB_BYTES / 16; i += GSP_WAVES * 64)
        reinterpret_cast<float4*>(s_ecomb)[i] = reinterpret_cast<const float4*>(ecomb)[i];
    __syncthreads();

    // ---- prologue: un-pipelined gather of the first tile
    float bq[25];
    {
        long long node = (long long)tile * GSP_TILE + wave * 16 + j;
        const bool valid = node < n_tot;
        if (!valid) node = n_tot - 1;
        int e = valid ? row_ptr[node] : 0;
        const int e_end = valid ? row_ptr[node + 1] : 0;
        const float* hr = h + (size_t)node * GS_D + 4 * g;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 x = *reinterpret_cast<const float4*>(hr + 16 * q);
            bq[4 * q + 0] = x.x; bq[4 * q + 1] = x.y; bq[4 * q + 2] = x.z; bq[4 * q + 3] = x.w;
        }
        bq[24] = h[(size_t)node * GS_D + 96 + g];
        while (__any(e < e_end)) {
            if (e < e_end) {
                const int u = src[e];
                const int code = ecode[e];
                e++;
                const float* ur = h + (size_t)u * GS_D + 4 * g;
                const float* er = s_ecomb + code * GS_D + 4 * g;
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const float4 x = *reinterpret_cast<const float4*>(ur + 16 * q);
                    const float4 w = *reinterpret_cast<const float4*>(er + 16 * q);
                    bq[4 * q + 0] += relu1(w.x + x.x); bq[4 * q + 1] += relu1(w.y + x.y);
                    bq[4 * q + 2] += relu1(w.z + x.z); bq[4 * q + 3] += relu1(w.w + x.w);
                }
                bq[24] += relu1(s_ecomb[code * GS_D + 96 + g] + h[(size_t)u * GS_D + 96 + g]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunks 0 and 1 resident

    float vmax = 0.0f;
    while (true) {
        const int next = tile + gridDim.x;
        const bool has_next = next < n_tiles;  // workgroup-uniform
        long long nnode = (long long)next * GSP_TILE + wave * 16 + j;
        const bool nvalid = has_next && nnode < n_tot && !getenv_abl;
        if (!nvalid) nnode = n_tot - 1;
        float bqn[25];
        int p_ecur = 0, p_eend = 0, p_unx = 0, p_cnx = 0, p_unew = 0, p_cnew = 0, p_code = 0, p_mode = 0, p_rp0 = 0, p_rp1 = 0;
        float4_t px0 = (float4_t){0.f, 0.f, 0.f, 0.f}, px1 = px0, px2 = px0, px3 = px0, px4 = px0, px5 = px0;
        float pxt = 0.f;
#pragma unroll
        for (int k = 0; k < 25; k++) bqn[k] = 0.0f;

        // B operands of this tile's first linear layer
        uint4_t in_hi[1][3], in_lo[1][3];
        float in_t[1];
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            GS_SPLIT2(bq[8 * ks + 0], bq[8 * ks + 1], in_hi[0][ks].x, in_lo[0][ks].x);
            GS_SPLIT2(bq[8 * ks + 2], bq[8 * ks + 3], in_hi[0][ks].y, in_lo[0][ks].y);
            GS_SPLIT2(bq[8 * ks + 4], bq[8 * ks + 5], in_hi[0][ks].z, in_lo[0][ks].z);
            GS_SPLIT2(bq[8 * ks + 6], bq[8 * ks + 7], in_hi[0][ks].w, in_lo[0][ks].w);
        }
#pragma unroll
        for (int k = 0; k < 24; k += 2)
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(bq[k])), __builtin_fabsf(bq[k + 1]));
        asm volatile("" : "+v"(vmax));  // here, not sunk to its next use: that kept all of bq live across step 0 (spills)
        in_t[0] = bq[24];

        float4_t acc2[1][GS_T2];
#pragma unroll
        for (int t2 = 0; t2 < GS_T2; t2++) {
            const float4 b = *reinterpret_cast<const float4*>(s_w0 + GS_W2_OFF + (16 * t2 + 4 * g) * 4);  // chunk 0 is resident
            acc2[0][t2] = (float4_t){b.x, b.y, b.z, b.w};
        }
        const float oscale = *reinterpret_cast<const float*>(s_w0 + GS_W2_OFF + 112 * 4);
        uint4_t h_hi[1], h_lo[1];
        h_hi[0] = (uint4_t){0, 0, 0, 0};
        h_lo[0] = (uint4_t){0, 0, 0, 0};

        GSP_STEP(0, s_w0, s_w2);
        GSP_STEP(1, s_w1, s_w3);
        GSP_STEP(2, s_w2, s_w0);
        GSP_STEP(3, s_w3, s_w1);
        GSP_STEP(4, s_w0, s_w2);
        GSP_STEP(5, s_w1, s_w3);
        GSP_STEP(6, s_w2, s_w0);  // chunk 0 of the next tile
        GSP_STEP(7, s_w3, s_w1);  // chunk 1 of the next tile

        {
            const long long node = (long long)tile * GSP_TILE + wave * 16 + j;
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
        if (!has_next) break;

        // residual: in-degree > 6 (hub nodes, kNN graphs) -- not overlapped, same order
        while (__any(p_ecur < p_eend)) {
            if (p_ecur < p_eend) {
                const int u = p_unx;
                const int code = p_cnx;
                p_ecur++;
                if (p_ecur < p_eend) { p_unx = src[p_ecur]; p_cnx = ecode[p_ecur]; }
                const float* ur = h + (size_t)u * GS_D + 4 * g;
                const float* er = s_ecomb + code * GS_D + 4 * g;
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const float4 x = *reinterpret_cast<const float4*>(ur + 16 * q);
                    const float4 w = *reinterpret_cast<const float4*>(er + 16 * q);
                    bqn[4 * q + 0] += relu1(w.x + x.x); bqn[4 * q + 1] += relu1(w.y + x.y);
                    bqn[4 * q + 2] += relu1(w.z + x.z); bqn[4 * q + 3] += relu1(w.w + x.w);
                }
                bqn[24] += relu1(s_ecomb[code * GS_D + 96 + g] + h[(size_t)u * GS_D + 96 + g]);
            }
        }
#pragma unroll
        for (int k = 0; k < 25; k++) bq[k] = bqn[k];
        tile = next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last steps' prefetches
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

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

template <int S>
__device__ __forceinline__ void gs_step_p(const char* wb, int lane, int g, const uint4_t (&in_hi)[1][3],
                                          const uint4_t (&in_lo)[1][3], const float (&in_t)[1], uint4_t (&h_hi)[1],
                                          uint4_t (&h_lo)[1], float4_t (&acc2)[1][GS_T2], float& vmax) {
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
        GS_FENCE();
        gs_ld4(fa, wb + 8192, lane);
        gs_mm2(fb, in_hi[0][1], in_lo[0][1], a0, a1);
        GS_FENCE();
        t0 = *reinterpret_cast<const float*>(wb + GS_TAIL_OFF + lane * 4);
        t1 = *reinterpret_cast<const float*>(wb + GS_TAIL_OFF + 256 + lane * 4);
        if (M2) gs_ld4(fb, w2, lane);
        gs_mm2(fa, in_hi[0][2], in_lo[0][2], a0, a1);
        GS_FENCE();
        a0 = GS_MFMA32(t0, in_t[0], a0);
        a1 = GS_MFMA32(t1, in_t[0], a1);
    } else {
        gs_ld4(fb, w2, lane);
        GS_FENCE();
    }
    uint4_t n_hi = {0, 0, 0, 0}, n_lo = {0, 0, 0, 0};
    if (M2) {
        gs_ld4(fa, w2 + 4096, lane);
        gs_mm2(fb, h_hi[0], h_lo[0], acc2[0][0], acc2[0][1]);
        GS_FENCE();
        gs_ld4(fb, w2 + 8192, lane);
        gs_mm2(fa, h_hi[0], h_lo[0], acc2[0][2], acc2[0][3]);
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
        GS_FENCE();
        acc2[0][6] = GS_MFMA16(fa[0], h_hi[0], acc2[0][6]);
        acc2[0][6] = GS_MFMA16(fa[0], h_lo[0], acc2[0][6]);
        acc2[0][6] = GS_MFMA16(fa[1], h_hi[0], acc2[0][6]);
    }
    if (M1) { h_hi[0] = n_hi; h_lo[0] = n_lo; }
}

// ---------------------------------------------------------------- persistent, tile-staged variant (default)
// The two kernels above leave the matrix pipe idle most of the time for the same reason: every wave waits for
// global memory inside its own critical path -- gin_layer_split_kernel in a per-tile gather prologue (three dependent
// round trips row_ptr -> src -> h[u] with 12 waves per CU to hide them), the pipelined one at the end of every step
// (a step lasts 0.6 us, a gather round trip under load about 2 us, so the step becomes the round trip).  Here global
// memory is touched only by transfers that are issued a whole step or more before anything depends on them:
//   * one persistent 12-wave workgroup per CU walks tiles of 192 consecutive nodes;
//   * the 192 rows of the NEXT tile are copied into LDS by LDS-DMA while the MLP of the current tile runs (molecule
//     batches are block diagonal with consecutive node ids, so a node's neighbours are almost always rows of its
//     own tile; the rare exception is fetched from global memory), and each lane prefetches the CSR row bounds and
//     the first four CSR entries of the node it handles next into registers (steps 0 and 1 of the 8 MLP steps);
//   * the gather of a tile is then a short LDS-only burst at the top of the tile (rows, edge-embedding combos),
//     in CSR order, followed by the 8 weight-stream steps of gin_layer_split_kernel (two LDS buffers, one step
//     ahead: with 3 waves per SIMD a step is about 1 us).
// vmcnt bookkeeping: loads return in order, so "s_waitcnt vmcnt(N)" after issuing N LDS-DMA instructions last means
// "everything issued before them has landed".  Every wave issues exactly GT_P weight pieces per step and exactly
// GT_R row pieces in step 1 (out-of-range pieces are clamped to valid addresses, not skipped).
constexpr int GT_WAVES = 12;
constexpr int GT_TILE = GT_WAVES * 16;            // 192 rows
constexpr int GT_ROW_BYTES = GT_TILE * GS_D * 4;  // 76800 = 75 pieces of 1 KiB
#define GT_P 3  // weight pieces per wave and step: ceil(27 / 12)
#define GT_R 7  // row pieces per wave and tile: ceil(75 / 12)
static_assert(GT_P * GT_WAVES >= GS_CHUNK_STRIDE / 1024 && GT_R * GT_WAVES >= GT_ROW_BYTES / 1024, "piece counts");
#define GT_STR2(x) #x
#define GT_STR(x) GT_STR2(x)

// One LDS-DMA instruction (64 lanes x 16 B, lane-linear from M0) issued from inline asm.  hipcc then neither counts it
// nor orders LDS reads against it: with the builtin, its waitcnt pass put "s_waitcnt vmcnt(0)" in front of the first
// ds_read of every odd step (reads of the higher-addressed buffer while the DMA into the lower one was in flight),
// which is the very stall this kernel is built to avoid.  All waits for these transfers are explicit below.
__device__ __forceinline__ void gt_dma16(const void* gaddr, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gaddr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint32_t gt_lds_addr(const void* p) {
    return (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ void gt_issue_chunk(const uint8_t* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
    uint32_t lane_off = lane * 16;
    asm volatile("" : "+v"(lane_off));  // keep the per-lane addresses out of the tile loop's invariants (they spill)
#pragma unroll
    for (int p = 0; p < GT_P; p++) {
        int piece = wave + GT_WAVES * p;
        if (piece >= GS_CHUNK_STRIDE / 1024) piece = wave;  // past the end: this wave's first piece again (same bytes)
        gt_dma16(gchunk + piece * 1024 + lane_off, __builtin_amdgcn_readfirstlane(gt_lds_addr(lds_buf) + piece * 1024));
    }
}

// rows [tile * 192, +192) of h -> s_rows; bytes past the end of h are read from its last 16 bytes instead (those LDS
// rows belong to no node)
__device__ __forceinline__ void gt_issue_rows(const float* __restrict__ h, float* s_rows, long long tile, int n_tot, int wave,
                                              int lane) {
    uint32_t lane_off = lane * 16;
    asm volatile("" : "+v"(lane_off));
    const long long lim = (long long)n_tot * (GS_D * 4) - 16;
#pragma unroll
    for (int p = 0; p < GT_R; p++) {
        int piece = wave + GT_WAVES * p;
        if (piece >= GT_ROW_BYTES / 1024) piece = wave;
        long long off = tile * GT_ROW_BYTES + piece * 1024 + lane_off;
        off = off < lim ? off : lim;
        gt_dma16(reinterpret_cast<const char*>(h) + off, __builtin_amdgcn_readfirstlane(gt_lds_addr(s_rows) + piece * 1024));
    }
}

__global__ __launch_bounds__(GT_WAVES * 64) void gin_layer_split_tiled_kernel(
    const float* __restrict__ h, float* __restrict__ hout, const int* __restrict__ row_ptr, const int* __restrict__ src,
    const uint8_t* __restrict__ ecode, const float* __restrict__ ecomb, const uint8_t* __restrict__ wchunks, int n_tot,
    int n_tiles, int relu_out, int* __restrict__ range_flag) {
    // ONE LDS object, carved by hand, weight buffers first: their fragment reads are then "lane * 16 + immediate"
    // (ds_read offsets are 16 bit; above 64 KB every fragment needed its own address register).  The DMA is issued
    // from inline asm, so there is no alias analysis to help by splitting the buffers into separate objects.
    __shared__ __attribute__((aligned(16))) char smem[2 * GS_CHUNK_STRIDE + GS_ECOMB_BYTES + GT_ROW_BYTES];
    char* const s_wa = smem;                    // even chunks
    char* const s_wb = smem + GS_CHUNK_STRIDE;  // odd chunks
    float* const s_ecomb = reinterpret_cast<float*>(smem + 2 * GS_CHUNK_STRIDE);
    float* const s_rows = reinterpret_cast<float*>(smem + 2 * GS_CHUNK_STRIDE + GS_ECOMB_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // ---- prologue: chunk 0, rows and CSR prefetch of the first tile, edge-embedding combos
    gt_issue_chunk(wchunks, s_wa, wave, lane);
    gt_issue_rows(h, s_rows, tile, n_tot, wave, lane);
    for (int i = threadIdx.x; i < GS_ECOMB_BYTES / 16; i += GT_WAVES * 64)
        reinterpret_cast<float4*>(s_ecomb)[i] = reinterpret_cast<const float4*>(ecomb)[i];
    int e_cur = 0, e_end = 0, su0 = 0, su1 = 0, su2 = 0, su3 = 0, sc0 = 0, sc1 = 0, sc2 = 0, sc3 = 0;
    {
        const long long node = (long long)tile * GT_TILE + wave * 16 + j;
        if (node < n_tot) {
            e_cur = row_ptr[node];
            e_end = row_ptr[node + 1];
            if (e_cur + 0 < e_end) { su0 = src[e_cur + 0]; sc0 = ecode[e_cur + 0]; }
            if (e_cur + 1 < e_end) { su1 = src[e_cur + 1]; sc1 = ecode[e_cur + 1]; }
            if (e_cur + 2 < e_end) { su2 = src[e_cur + 2]; sc2 = ecode[e_cur + 2]; }
            if (e_cur + 3 < e_end) { su3 = src[e_cur + 3]; sc3 = ecode[e_cur + 3]; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    float vmax = 0.0f;
    while (true) {
        // ---- gather (MP unit), LDS only: a = sum_e relu(h[src_e] + ecomb[code_e]) in CSR order, then + h[v]
        float bq[25];
#pragma unroll
        for (int k = 0; k < 25; k++) bq[k] = 0.0f;
        {
            const int tbase = tile * GT_TILE;
            for (int e = e_cur; __any(e < e_end); e++) {
                const int u = su0, code = sc0;
                su0 = su1; su1 = su2; su2 = su3;
                sc0 = sc1; sc1 = sc2; sc2 = sc3;
                if (e + 4 < e_end) { su3 = src[e + 4]; sc3 = ecode[e + 4]; }  // degrees above 4: four entries ahead
                if (e < e_end) {
                    const unsigned ul = (unsigned)(u - tbase);
                    const bool in = ul < (unsigned)GT_TILE;
                    const float* ur = s_rows + (in ? ul : 0u) * GS_D + 4 * g;
                    const float* er = s_ecomb + code * GS_D + 4 * g;
                    float4 x[6];
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        x[q] = *reinterpret_cast<const float4*>(ur + 16 * q);
                        // pinned: a `in ? lds : global` select would become flat loads with a full wait after each
                        asm volatile("" : "+v"(x[q].x), "+v"(x[q].y), "+v"(x[q].z), "+v"(x[q].w));
                    }
                    float xt = ur[96 - 3 * g];  // feature 96 + g
                    asm volatile("" : "+v"(xt));
                    if (!in) {  // neighbour outside the tile (graph straddling a tile boundary): from global memory
                        const float* gr = h + (size_t)u * GS_D + 4 * g;
#pragma unroll
                        for (int q = 0; q < 6; q++) x[q] = load_f4_rare(reinterpret_cast<const float4*>(gr + 16 * q));
                        xt = load_f32_rare(h + (size_t)u * GS_D + 96 + g);
                    }
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        const float4 w = *reinterpret_cast<const float4*>(er + 16 * q);
                        bq[4 * q + 0] += relu1(w.x + x[q].x);
                        bq[4 * q + 1] += relu1(w.y + x[q].y);
                        bq[4 * q + 2] += relu1(w.z + x[q].z);
                        bq[4 * q + 3] += relu1(w.w + x[q].w);
                    }
                    bq[24] += relu1(er[96 - 3 * g] + xt);
                }
            }
            const float* sr = s_rows + (wave * 16 + j) * GS_D + 4 * g;  // + (1 + eps) h[v], eps == 0
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const float4 x = *reinterpret_cast<const float4*>(sr + 16 * q);
                bq[4 * q + 0] += x.x; bq[4 * q + 1] += x.y; bq[4 * q + 2] += x.z; bq[4 * q + 3] += x.w;
            }
            bq[24] += sr[96 - 3 * g];
        }
        // B operands of the first linear layer
        uint4_t in_hi[1][3], in_lo[1][3];
        float in_t[1];
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            GS_SPLIT2(bq[8 * ks + 0], bq[8 * ks + 1], in_hi[0][ks].x, in_lo[0][ks].x);
            GS_SPLIT2(bq[8 * ks + 2], bq[8 * ks + 3], in_hi[0][ks].y, in_lo[0][ks].y);
            GS_SPLIT2(bq[8 * ks + 4], bq[8 * ks + 5], in_hi[0][ks].z, in_lo[0][ks].z);
            GS_SPLIT2(bq[8 * ks + 6], bq[8 * ks + 7], in_hi[0][ks].w, in_lo[0][ks].w);
        }
#pragma unroll
        for (int k = 0; k < 24; k += 2)
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(bq[k])), __builtin_fabsf(bq[k + 1]));
        asm volatile("" : "+v"(vmax));  // here, not sunk to its next use: that kept all of bq live across step 0 (spills)
        in_t[0] = bq[24];
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
        const long long nnode = (long long)next * GT_TILE + wave * 16 + j;
        const bool nvalid = has_next && nnode < n_tot;
        int n_rp0 = 0, n_rp1 = 0, nu0 = 0, nu1 = 0, nu2 = 0, nu3 = 0, nc0 = 0, nc1 = 0, nc2 = 0, nc3 = 0;

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
// the tile's outputs, stored inside step 7 so that the step's own wait covers them
#define GT_STORE()                                                                                                   \
    do {                                                                                                             \
        const long long node_ = (long long)tile * GT_TILE + wave * 16 + j;                                           \
        if (node_ < n_tot) {                                                                                         \
            float* row_ = hout + (size_t)node_ * GS_D;                                                               \
            _Pragma("unroll") for (int t2 = 0; t2 < GS_T2; t2++) {                                                   \
                const int col_ = 16 * t2 + 4 * g;                                                                    \
                if (col_ < GS_D) {                                                                                   \
                    float4_t r_ = acc2[0][t2] * oscale;                                                              \
                    if (relu_out) { r_.x = relu1(r_.x); r_.y = relu1(r_.y); r_.z = relu1(r_.z); r_.w = relu1(r_.w); } \
                    *reinterpret_cast<float4*>(row_ + col_) = make_float4(r_.x, r_.y, r_.z, r_.w);                   \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
        // ---- step 0: CSR row bounds of this lane's next node; chunk 1 -> B
        // (unconditional loads from clamped addresses, masked after the wait: a conditional load merges with the
        // default value in a copy, and hipcc waits for the load in front of that copy, i.e. right here)
        n_rp0 = row_ptr[nvalid ? nnode : 0];
        n_rp1 = row_ptr[nvalid ? nnode + 1 : 0];
        __builtin_amdgcn_sched_barrier(0);
        gt_issue_chunk(wchunks + 1 * GS_CHUNK_STRIDE, s_wb, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        gs_step_p<0>(s_wa, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax);
        GT_WAIT_ALL();
        GSP_BAR();  // every wave has finished its gather: s_rows may be overwritten
        // ---- step 1: its first four CSR entries; chunk 2 -> A; then the rows of the next tile -> s_rows
        if (!nvalid) n_rp1 = n_rp0;
        {   // entries past the row's end re-read its first entry (or entry 0 for an empty row); they are never used
            const int eb = n_rp0 < n_rp1 ? n_rp0 : 0;
            const int i1 = n_rp0 + 1 < n_rp1 ? n_rp0 + 1 : eb, i2 = n_rp0 + 2 < n_rp1 ? n_rp0 + 2 : eb,
                      i3 = n_rp0 + 3 < n_rp1 ? n_rp0 + 3 : eb;
            nu0 = src[eb]; nc0 = ecode[eb];
            nu1 = src[i1]; nc1 = ecode[i1];
            nu2 = src[i2]; nc2 = ecode[i2];
            nu3 = src[i3]; nc3 = ecode[i3];
        }
        __builtin_amdgcn_sched_barrier(0);
        gt_issue_chunk(wchunks + 2 * GS_CHUNK_STRIDE, s_wa, wave, lane);
        gt_issue_rows(h, s_rows, has_next ? next : tile, n_tot, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        gs_step_p<1>(s_wb, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax);
        // chunk 2 has landed, the rows stay in flight.  The CSR registers are NOT tied here: hipcc waits for a
        // pending load in front of its first use, with a count that ignores the asm-issued DMA, i.e. for the rows.
        asm volatile("s_waitcnt vmcnt(" GT_STR(GT_R) ")" ::: "memory");
        GSP_BAR();
        // ---- steps 2..7 (the wait of step 2 also covers the rows)
#define GT_STEP(S, CUR, NXT)                                                                          \
    gt_issue_chunk(wchunks + (size_t)(((S) + 1) & 7) * GS_CHUNK_STRIDE, NXT, wave, lane);             \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    gs_step_p<S>(CUR, lane, g, in_hi, in_lo, in_t, h_hi, h_lo, acc2, vmax);                           \
    if ((S) == 7) GT_STORE();                                                                         \
    GT_WAIT_ALL();                                                                                    \
    GSP_BAR()
        GT_STEP(2, s_wa, s_wb);
        GT_STEP(3, s_wb, s_wa);
        GT_STEP(4, s_wa, s_wb);
        GT_STEP(5, s_wb, s_wa);
        GT_STEP(6, s_wa, s_wb);
        GT_STEP(7, s_wb, s_wa);  // chunk 0 again, for the next tile
#undef GT_STEP
        if (!has_next) break;
        e_cur = n_rp0; e_end = n_rp1;
        su0 = nu0; su1 = nu1; su2 = nu2; su3 = nu3;
        sc0 = nc0; sc1 = nc1; sc2 = nc2; sc3 = nc3;
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
                            int nt, hipStream_t s) {
    if (e_tot == 0 && (nt == 3 || nt == 0)) nt = 1;  // the persistent kernels prefetch CSR entry 0 unconditionally
    if (nt == 3) {
        const int n_tiles = (int)ceil_div_ll(n_tot, GT_TILE);
        const int grid = n_tiles < 256 ? n_tiles : 256;  // persistent: one 12-wave workgroup per CU
        gin_layer_split_tiled_kernel<<<grid, GT_WAVES * 64, 0, s>>>(h, hout, row_ptr, src, ecode, ecomb, chunks, n_tot, n_tiles, relu_out,
                                                                    range_flag);
        return;
    }
    if (nt == 0) {
        const int n_tiles = (int)ceil_div_ll(n_tot, GSP_TILE);
        const int grid = n_tiles < 256 ? n_tiles : 256;  // persistent: one 12-wave workgroup per CU
        gin_layer_split_persistent_kernel<<<grid, GSP_WAVES * 64, 0, s>>>(h, hout, row_ptr, src, ecode, ecomb, chunks, n_tot, n_tiles,
                                                                         relu_out, range_flag, getenv("FLOWGNN_GS_ABL") ? atoi(getenv("FLOWGNN_GS_ABL")) : 0);
        return;
    }
    if (nt == 2) {
        const int blocks = (int)ceil_div_ll(n_tot, 128);
        gin_layer_split_kernel<2><<<blocks, 256, 0, s>>>(h, hout, row_ptr, src, ecode, ecomb, chunks, n_tot, relu_out, range_flag, getenv("FLOWGNN_GS_ABL") ? atoi(getenv("FLOWGNN_GS_ABL")) : 0);
    } else {
        const int blocks = (int)ceil_div_ll(n_tot, 64);
        gin_layer_split_kernel<1><<<blocks, 256, 0, s>>>(h, hout, row_ptr, src, ecode, ecomb, chunks, n_tot, relu_out, range_flag, getenv("FLOWGNN_GS_ABL") ? atoi(getenv("FLOWGNN_GS_ABL")) : 0);
    }
}

}  // namespace fg
