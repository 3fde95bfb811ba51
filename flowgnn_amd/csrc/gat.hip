// gat.hip -- GAT (5 layers, 4 heads x 16 dims) hot path for gfx950 (MI355X).
//
// Reference per graph (GAT/src/*.cc); per node 64 features, index f = dim * 4 + head (the reference's FM_VEC):
//   proj_0[v]   = W_lin0 feat(v)      (raw integer atom features, 9 inputs)            load_inputs.cc:184-201
//   skip_0[v]   = feat(v) in dims 0..8 of head 0, else 0                               load_inputs.cc:190-191
//   ssrc_l[v][h] = sum_d proj_l[v][d][h] a_src[l][h][d],  stgt likewise               load_inputs.cc:203-224, node_embedding.cc:235-268
//   per destination v, over its in-neighbours u plus v itself:
//     e[h] = exp(leaky_0.2(ssrc[v][h] + stgt[u][h]))   (no max subtraction)            message_passing.cc:122-128
//     msg[v][d][h] = sum_u e[h] proj[u][d][h] / sum_u e[h]                              message_passing.cc:130-141, conv_layer.cc:158-177
//   o = ELU(msg + W_skip_l skip_l),  skip_{l+1} = o,  proj_{l+1} = W_lin_{l+1} o       node_embedding.cc:157-195
//   last layer: emb[v][d] = mean_h (msg + W_skip_4 skip_4)[d][h];  out = pb + pw . mean_v emb   finalize.cc:46-112
//
// Here ONE fused kernel per layer: a wave owns 16 destination nodes; lane (j, g) gathers node j's attention
// message for dims {g, 4+g, 8+g, 12+g} x 4 heads straight into the MFMA accumulator layout, then runs both
// 64x64 contractions on fp32 MFMA chained through registers (as in gin.hip), and reduces the next scores with
// two shuffles.  Traffic per node-layer ~ 1 KB; the kernel is HBM-bound.
#include "common.h"
#include "device_common.h"
#include "dense_split.h"
#include "modelq.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace fg {

constexpr int GAT_D = 16;
constexpr int GAT_H = 4;
constexpr int GAT_F = GAT_D * GAT_H;  // 64
constexpr int GAT_L = 5;

// Layer 0 has no stored projection: proj_0[v][d][h] = sum_k feat_k(v) W_lin0[d][k][h] is 9 multiply-adds per value from the
// 36-byte integer feature row (load_inputs.cc:184-201), so the first layer kernel computes it for the rows of its tile
// straight into LDS (and for the rare neighbour outside the tile on the fly) instead of reading 256-byte rows an encoder
// kernel would have written: one launch and a 1.9 GB write + read less per pass.
// proj_0 of (row, dim) from the row's nine features, heads in the float4
__device__ __forceinline__ float4 gat_proj0(const int* f, const float4* lin0_d) {
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < ND_FEATURE; k++) {
        const float fk = (float)f[k];
        const float4 w = lin0_d[k];
        p.x += fk * w.x; p.y += fk * w.y; p.z += fk * w.z; p.w += fk * w.w;
    }
    return p;
}

// scores of layer 0: sum over the 16 dims in ascending order, the same code in the tile pass and in the rare path below, so a
// node's scores do not depend on which tile reads them
__device__ __forceinline__ void gat_score_acc(float4& s, const float4& pd, const float4& a) {
    s.x += pd.x * a.x; s.y += pd.y * a.y; s.z += pd.z * a.z; s.w += pd.w * a.w;
}

// A neighbour u outside the tile (layer 0): this lane's four projection rows from u's raw features (its scores come from
// the pre-pass below, like every other node's).
__device__ __forceinline__ void gat_layer0_rare(const int* __restrict__ node_feature, long long row, int g, const float4* s_lin0,
                                                const float4* score_ptr, float4& st, float4 (&p)[4]) {
    // the 36-byte feature row and u's target scores in ONE round trip (four loads, one wait)
    typedef int int4v __attribute__((ext_vector_type(4)));
    int4v fa, fb;
    int fc;
    const int* fp = node_feature + (size_t)row * ND_FEATURE;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\tglobal_load_dword %2, %4, off offset:32\n\t"
                 "global_load_dwordx4 %3, %5, off\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(fa), "=&v"(fb), "=&v"(fc), "=&v"(st)
                 : "v"(fp), "v"(score_ptr)
                 : "memory");
    static_assert(ND_FEATURE == 9, "feature row = 4 + 4 + 1 words");
    const int f[ND_FEATURE] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w, fc};
#pragma unroll 1
    for (int t = 0; t < 4; t++) {  // rolled on purpose: this is the rare path, it must not cost the common one registers
        const float4 pd = gat_proj0(f, s_lin0 + (4 * t + g) * ND_FEATURE);
        if (t == 0) p[0] = pd;
        else if (t == 1) p[1] = pd;
        else if (t == 2) p[2] = pd;
        else p[3] = pd;
    }
}

// Layer-0 attention scores of every node (32 B per node from its 36-byte feature row): ssrc[v][h] = sum_d proj_0[v][d][h]
// a_src[0][h][d], stgt likewise (load_inputs.cc:203-224), dims in ascending order.  A thread per node; the weights are
// wave-uniform (scalar loads).
__global__ __launch_bounds__(256) void gat_scores0_kernel(const int* __restrict__ node_feature, const int* __restrict__ feat_row,
                                                           const float* __restrict__ lin0, const float* __restrict__ a_src,
                                                           const float* __restrict__ a_tgt, float* __restrict__ scores, int n_tot) {
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_tot) return;
    const long long row = feat_row ? feat_row[v] : v;
    int f[ND_FEATURE];
#pragma unroll
    for (int k = 0; k < ND_FEATURE; k++) f[k] = node_feature[(size_t)row * ND_FEATURE + k];
    float4 ss = make_float4(0.f, 0.f, 0.f, 0.f), st = ss;
#pragma unroll
    for (int d = 0; d < GAT_D; d++) {
        const float4 pd = gat_proj0(f, reinterpret_cast<const float4*>(lin0) + d * ND_FEATURE);
        gat_score_acc(ss, pd, reinterpret_cast<const float4*>(a_src)[d]);
        gat_score_acc(st, pd, reinterpret_cast<const float4*>(a_tgt)[d]);
    }
    stream_store4(reinterpret_cast<float4*>(scores) + v * 2 + 0, ss);
    stream_store4(reinterpret_cast<float4*>(scores) + v * 2 + 1, st);
}

// local node index per node (reference quirk mode: every graph reads the first rows of the batch)
__global__ __launch_bounds__(256) void gat_local_rows_kernel(const int* __restrict__ node_off, int* __restrict__ feat_row,
                                                              int num_graphs) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= num_graphs) return;
    for (int v = node_off[g] + lane; v < node_off[g + 1]; v += 64) feat_row[v] = v - node_off[g];
}

struct GatLayerDev {
    // fp32 MFMA path:
    const float* wskip;  // fragments [4 t][4 q][64][4]: W_skip_l[f_o = 16 t + i][f_i = 16 q + 4 g + r]
    const float* wlin;   // fragments [4 t][4 t2][64][4]: W_lin_{l+1}[f_o = 16 t2 + i][f_i = 16 t + 4 g + r]
    // split-f16 path (the same two 64 x 64 matrices as f16 hi/lo fragments of v_mfma_f32_16x16x32_f16, 16 KiB each, in the
    // same LDS space): [out tile][K-step 0..1][hi, lo][64 lanes][8 halves]; slot e of lane (i, gk) of K-step ks is
    // W[16 t + i][16 (2 ks + (e >> 2)) + 4 gk + (e & 3)] scaled by a power of two; *_scale undoes it
    const float* wskip_split;
    const float* wlin_split;
    float wskip_scale, wlin_scale;
    int* range_flag;
    const float* a_src;  // [16 dim][4 head] of layer l+1
    const float* a_tgt;
};

// Persistent 8-wave workgroups walk tiles of 128 consecutive nodes.  Per tile the projections (128 x 256 B) and the
// attention scores (128 x 32 B) of the tile's nodes are copied into LDS by LDS-DMA, and the attention gather reads its
// neighbours from there: molecule batches are block diagonal with consecutive node ids, so almost every neighbour is a row
// of the same tile (the rare exception -- a graph straddling a tile boundary -- is fetched from global memory by loads that
// wait for themselves).  The layer's 32 KiB of weight fragments are staged in LDS once per workgroup instead of being read
// from L2 by every wave (13.7 GB per launch at 2^18 molhiv graphs), and the CSR entry of the next in-edge is requested one
// trip ahead.  68 KB of LDS: two workgroups per CU.
constexpr int GAT_TR = 128;
struct GatLayer0Dev {
    const float* lin0;   // [16 dim][9][4 head]
    const float* a_src;  // [16 dim][4 head] of layer 0
    const float* a_tgt;
};

template <bool FINAL, bool FIRST, bool SPLIT>
__global__ __launch_bounds__(512, 4) void gat_layer_kernel(const float* __restrict__ proj, const float* __restrict__ skipin,
                                                         const float* __restrict__ scores, float* __restrict__ proj_out,
                                                         float* __restrict__ skip_out, float* __restrict__ scores_out,
                                                         float* __restrict__ emb_out, const int* __restrict__ row_ptr,
                                                         const int* __restrict__ src, GatLayerDev w, int n_tot,
                                                         const int* __restrict__ node_feature, const int* __restrict__ feat_row,
                                                         GatLayer0Dev w0, const float* __restrict__ pool_w) {
    __shared__ __attribute__((aligned(16))) float4 s_wskip[16 * 64];
    __shared__ __attribute__((aligned(16))) float4 s_wlin[FINAL ? 1 : 16 * 64];
    // Staged projections, 256 B per row.  With rows stored as they are in memory every row starts in the same LDS banks and
    // the gather -- 16 node lanes reading the same 16-byte column of 16 different rows -- is a 16-way bank conflict.  So row r
    // is stored rotated by r columns (column c at slot (c + r) mod 16); LDS-DMA allows it because every lane supplies its own
    // global address while the LDS side stays lane-linear.
    __shared__ __attribute__((aligned(16))) float4 s_proj[GAT_TR * 16];
    __shared__ __attribute__((aligned(16))) float4 s_sc[GAT_TR * 2];
    __shared__ __attribute__((aligned(16))) float4 s_lin0[FIRST ? GAT_D * ND_FEATURE : 1];
    __shared__ int s_feat[FIRST ? GAT_TR * ND_FEATURE : 1];
    for (int i = threadIdx.x; i < 16 * 64; i += 512) {
        s_wskip[i] = reinterpret_cast<const float4*>(SPLIT ? w.wskip_split : w.wskip)[i];
        if (!FINAL) s_wlin[i] = reinterpret_cast<const float4*>(SPLIT ? w.wlin_split : w.wlin)[i];
    }
    float vmax = 0.0f;  // SPLIT: largest operand magnitude (range check of the f16 split, see gin_split.hip)
    if (FIRST) {
        for (int i = threadIdx.x; i < GAT_D * ND_FEATURE; i += 512) s_lin0[i] = reinterpret_cast<const float4*>(w0.lin0)[i];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const float4* proj4 = reinterpret_cast<const float4*>(proj);
    const float4* sc4 = reinterpret_cast<const float4*>(scores);
    const int n_tiles = (n_tot + GAT_TR - 1) / GAT_TR;
    const long long proj_last = (long long)n_tot * (GAT_F * 4) - 16, sc_last = (long long)n_tot * 32 - 16;
    // layer 0: this thread's words of a tile's feature rows (36 B per row; rows past the end repeat the last one)
    int fpre[3] = {0, 0, 0};
    auto fetch_features = [&](int t) {
        if (!FIRST || t >= n_tiles) return;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = threadIdx.x + 512 * k;
            if (i < GAT_TR * ND_FEATURE) {
                long long v = (long long)t * GAT_TR + i / ND_FEATURE;
                if (v >= n_tot) v = n_tot - 1;
                const long long row = feat_row ? feat_row[v] : v;
                fpre[k] = node_feature[(size_t)row * ND_FEATURE + (i % ND_FEATURE)];
            }
        }
    };
    fetch_features(blockIdx.x);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tbase = tile * GAT_TR;
    __syncthreads();  // the previous tile's rows are no longer read (first time: nothing to wait for)
    if (FIRST) {
        // the tile's feature rows (36 B each; rows past the end repeat the last one), then proj_0 and both scores per (row, dim)
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (threadIdx.x + 512 * k < GAT_TR * ND_FEATURE) s_feat[threadIdx.x + 512 * k] = fpre[k];
        __syncthreads();
        fetch_features(tile + gridDim.x);  // the next tile's, one tile ahead of their use
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // opaque per tile: this lane's dim is the same for every tile, and its nine W_lin0 rows would be kept across the loop
        {
            // this lane's dim is the same for its four rows: its nine W_lin0 rows are read once per tile into registers
            // (36 of them, alive only here) instead of once per value
            const int d = tid & 15;
            float4 wl[ND_FEATURE];
#pragma unroll
            for (int k = 0; k < ND_FEATURE; k++) wl[k] = s_lin0[d * ND_FEATURE + k];
#pragma unroll
            for (int k = 0; k < GAT_TR * GAT_D / 512; k++) {
                const int r = (tid + 512 * k) >> 4;
                s_proj[r * 16 + ((d + r) & 15)] = gat_proj0(&s_feat[r * ND_FEATURE], wl);
            }
        }
    } else {
#pragma unroll
    for (int p = 0; p < 4; p++) {  // 32 pieces of 1 KiB; bytes past the end of the array: its last 16 bytes (rows of no node)
        const int piece = wv + 8 * p;
        const int rr = piece * 4 + (lane >> 4);  // row inside the tile that this lane's LDS slot belongs to
        long long off = (long long)(tbase + rr) * (GAT_F * 4) + (((lane & 15) - rr) & 15) * 16;
        off = off < proj_last ? off : proj_last;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(proj) + off),
                                         (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(s_proj) + piece * 1024), 16, 0, 0);
    }
    }
    if (wv < 4) {  // the tile's scores (layer 0: written by gat_scores0_kernel)
        long long off = (long long)tbase * 32 + wv * 1024 + lane * 16;
        off = off < sc_last ? off : sc_last;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(scores) + off),
                                         (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(s_sc) + wv * 1024), 16, 0, 0);
    }
    int lds_lane = lane;
    asm volatile("" : "+v"(lds_lane));  // opaque per tile: otherwise all 32 fragment reads are hoisted out of the tile loop (256 VGPRs)
    long long node = (long long)tbase + wv * 16 + j;
    const bool valid = node < n_tot;
    if (!valid) node = n_tot - 1;
    int e = valid ? row_ptr[node] : 0;
    const int e_end = valid ? row_ptr[node + 1] : 0;
    int u = (int)node;  // the self edge
    int u_nx = e < e_end ? src[e] : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the tile's rows (and, first time, the weights) are in LDS

    // ---- attention gather (pull): self edge first, then the CSR row (ascending source)
    const float4 ssrc = s_sc[(wv * 16 + j) * 2 + 0];
    float4 den = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 num[4];
#pragma unroll
    for (int t = 0; t < 4; t++) num[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool more = true;
    while (__any(more)) {
        if (more) {
            const unsigned ul = (unsigned)(u - tbase);
            const bool in = ul < (unsigned)GAT_TR;
            const unsigned lr = in ? ul : 0u;
            // LDS reads unconditional (clamped) and pinned; global memory only in a branch, through asm loads (a
            // `cond ? lds : global` select makes hipcc emit flat loads with a full wait after each)
            float4 st = s_sc[lr * 2 + 1];
            asm volatile("" : "+v"(st.x), "+v"(st.y), "+v"(st.z), "+v"(st.w));
            float4 p[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                p[t] = s_proj[lr * 16 + ((4 * t + g + lr) & 15)];
                asm volatile("" : "+v"(p[t].x), "+v"(p[t].y), "+v"(p[t].z), "+v"(p[t].w));
            }
            if (!in) {
                const float4* sp = sc4 + (size_t)u * 2 + 1;  // target scores of u
                if (FIRST) {
                    gat_layer0_rare(node_feature, feat_row ? (long long)load_i32_rare(feat_row + u) : (long long)u, g, s_lin0, sp, st, p);
                } else {
                    // the scores and this lane's four projection rows of u in ONE round trip (five loads, one wait)
                    const float4* pp = proj4 + (size_t)u * 16 + g;
                    asm volatile("global_load_dwordx4 %0, %5, off\n\tglobal_load_dwordx4 %1, %6, off\n\t"
                                 "global_load_dwordx4 %2, %6, off offset:64\n\tglobal_load_dwordx4 %3, %6, off offset:128\n\t"
                                 "global_load_dwordx4 %4, %6, off offset:192\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(st), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3])
                                 : "v"(sp), "v"(pp)
                                 : "memory");
                }
            }
            more = e < e_end;
            u = u_nx;
            e++;
            if (e < e_end) u_nx = src[e];
            float4 s = make_float4(ssrc.x + st.x, ssrc.y + st.y, ssrc.z + st.z, ssrc.w + st.w);
            s.x = __expf(s.x < 0.f ? s.x * 0.2f : s.x); s.y = __expf(s.y < 0.f ? s.y * 0.2f : s.y);
            s.z = __expf(s.z < 0.f ? s.z * 0.2f : s.z); s.w = __expf(s.w < 0.f ? s.w * 0.2f : s.w);
            den.x += s.x; den.y += s.y; den.z += s.z; den.w += s.w;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                num[t].x += s.x * p[t].x; num[t].y += s.y * p[t].y; num[t].z += s.z * p[t].z; num[t].w += s.w * p[t].w;
            }
        }
    }

    // ---- o = msg + W_skip skip   (accumulators start from the message: rows 16 t + 4 g + r = dim 4 t + g, head r)
    float bq[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (FIRST) {
            // layer 0: the skip input is the raw integer feature vector (element 4 d of the 64-float row holds feature d < 9,
            // everything else is zero), read from the 36-byte feature row instead of a 256-byte row the encoder would write
            const int d = 4 * q + g;  // (the tile's feature rows are in LDS since the projection pass)
            bq[4 * q + 0] = d < ND_FEATURE ? (float)s_feat[(wv * 16 + j) * ND_FEATURE + d] : 0.0f;
            bq[4 * q + 1] = 0.0f; bq[4 * q + 2] = 0.0f; bq[4 * q + 3] = 0.0f;
        } else {
            const float4 x = *reinterpret_cast<const float4*>(skipin + (size_t)node * GAT_F + 16 * q + 4 * g);
            bq[4 * q + 0] = x.x; bq[4 * q + 1] = x.y; bq[4 * q + 2] = x.z; bq[4 * q + 3] = x.w;
        }
    }
    float4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (float4_t){num[t].x / den.x, num[t].y / den.y, num[t].z / den.z, num[t].w / den.w};
    const float4* ws4 = s_wskip;
    if constexpr (SPLIT) {
        // both 64 x 64 contractions of the layer as split-f16 products (3 x v_mfma_f32_16x16x32_f16 per fp32 product block,
        // dense_split.h): 24 MFMAs of 16 cycles per contraction instead of 64 fp32 MFMAs of 32
        ds_uint4_t b_hi[2], b_lo[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            DS_SPLIT2(bq[8 * ks + 0], bq[8 * ks + 1], b_hi[ks].x, b_lo[ks].x);
            DS_SPLIT2(bq[8 * ks + 2], bq[8 * ks + 3], b_hi[ks].y, b_lo[ks].y);
            DS_SPLIT2(bq[8 * ks + 4], bq[8 * ks + 5], b_hi[ks].z, b_lo[ks].z);
            DS_SPLIT2(bq[8 * ks + 6], bq[8 * ks + 7], b_hi[ks].w, b_lo[ks].w);
        }
#pragma unroll
        for (int k = 0; k < 16; k += 2) vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(bq[k])), __builtin_fabsf(bq[k + 1]));
        asm volatile("" : "+v"(vmax));
        const char* wb = reinterpret_cast<const char*>(s_wskip);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            float4_t sk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const ds_uint4_t a_hi = *reinterpret_cast<const ds_uint4_t*>(wb + ((t * 2 + ks) * 2 + 0) * 1024 + lds_lane * 16);
                const ds_uint4_t a_lo = *reinterpret_cast<const ds_uint4_t*>(wb + ((t * 2 + ks) * 2 + 1) * 1024 + lds_lane * 16);
                sk = DS_MFMA16(a_hi, b_hi[ks], sk);
                sk = DS_MFMA16(a_hi, b_lo[ks], sk);
                sk = DS_MFMA16(a_lo, b_hi[ks], sk);
            }
            acc[t] += sk * w.wskip_scale;
        }
    } else {
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float4 af = ws4[(t * 4 + q) * 64 + lds_lane];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bq[4 * q + 0], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float4 af = ws4[(t * 4 + q) * 64 + lds_lane];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bq[4 * q + 1], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float4 af = ws4[(t * 4 + q) * 64 + lds_lane];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bq[4 * q + 2], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float4 af = ws4[(t * 4 + q) * 64 + lds_lane];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bq[4 * q + 3], acc[t], 0, 0, 0);
        }
    }
    }

    if (FINAL) {
        if (pool_w != nullptr) {
            // readout folded in (as in gin_split.hip): the logit is mean_v(emb[v]) . w + b = mean_v(emb[v] . w) + b, so only the
            // per-node dot product leaves the kernel (emb_out is then a float[n_tot]; 4 B instead of a 64 B row to read back)
            float part = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; t++) part += (acc[t].x + acc[t].y + acc[t].z + acc[t].w) / (float)GAT_H * pool_w[4 * t + g];
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (valid && g == 0) emb_out[node] = part;
        } else if (valid) {
#pragma unroll
            for (int t = 0; t < 4; t++) emb_out[(size_t)node * GAT_D + 4 * t + g] = (acc[t].x + acc[t].y + acc[t].z + acc[t].w) / (float)GAT_H;
        }
        continue;
    }

    // ---- ELU, next skip input, next projection (chained through registers), next scores
#pragma unroll
    for (int t = 0; t < 4; t++) {
        acc[t].x = acc[t].x <= 0.f ? __expf(acc[t].x) - 1.0f : acc[t].x; acc[t].y = acc[t].y <= 0.f ? __expf(acc[t].y) - 1.0f : acc[t].y;
        acc[t].z = acc[t].z <= 0.f ? __expf(acc[t].z) - 1.0f : acc[t].z; acc[t].w = acc[t].w <= 0.f ? __expf(acc[t].w) - 1.0f : acc[t].w;
        if (valid)
            *reinterpret_cast<float4*>(skip_out + (size_t)node * GAT_F + 16 * t + 4 * g) = make_float4(acc[t].x, acc[t].y, acc[t].z, acc[t].w);
    }
    float4_t pr[4];
#pragma unroll
    for (int t2 = 0; t2 < 4; t2++) pr[t2] = (float4_t){0.f, 0.f, 0.f, 0.f};
    const float4* wl4 = s_wlin;
    if constexpr (SPLIT) {
        ds_uint4_t o_hi[2], o_lo[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {  // output tiles 2 ks, 2 ks + 1 of the skip contraction are K-step ks of this one
            DS_SPLIT2(acc[2 * ks].x, acc[2 * ks].y, o_hi[ks].x, o_lo[ks].x);
            DS_SPLIT2(acc[2 * ks].z, acc[2 * ks].w, o_hi[ks].y, o_lo[ks].y);
            DS_SPLIT2(acc[2 * ks + 1].x, acc[2 * ks + 1].y, o_hi[ks].z, o_lo[ks].z);
            DS_SPLIT2(acc[2 * ks + 1].z, acc[2 * ks + 1].w, o_hi[ks].w, o_lo[ks].w);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(acc[t].x)), __builtin_fabsf(acc[t].y));
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(acc[t].z)), __builtin_fabsf(acc[t].w));
        }
        asm volatile("" : "+v"(vmax));
        const char* wb = reinterpret_cast<const char*>(s_wlin);
#pragma unroll
        for (int t2 = 0; t2 < 4; t2++) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const ds_uint4_t a_hi = *reinterpret_cast<const ds_uint4_t*>(wb + ((t2 * 2 + ks) * 2 + 0) * 1024 + lds_lane * 16);
                const ds_uint4_t a_lo = *reinterpret_cast<const ds_uint4_t*>(wb + ((t2 * 2 + ks) * 2 + 1) * 1024 + lds_lane * 16);
                pr[t2] = DS_MFMA16(a_hi, o_hi[ks], pr[t2]);
                pr[t2] = DS_MFMA16(a_hi, o_lo[ks], pr[t2]);
                pr[t2] = DS_MFMA16(a_lo, o_hi[ks], pr[t2]);
            }
            pr[t2] *= w.wlin_scale;
        }
    } else {
#pragma unroll
    for (int t = 0; t < 4; t++) {
        float4 af[4];
#pragma unroll
        for (int t2 = 0; t2 < 4; t2++) af[t2] = wl4[(t * 4 + t2) * 64 + lds_lane];
#pragma unroll
        for (int t2 = 0; t2 < 4; t2++) pr[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t2].x, acc[t].x, pr[t2], 0, 0, 0);
#pragma unroll
        for (int t2 = 0; t2 < 4; t2++) pr[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t2].y, acc[t].y, pr[t2], 0, 0, 0);
#pragma unroll
        for (int t2 = 0; t2 < 4; t2++) pr[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t2].z, acc[t].z, pr[t2], 0, 0, 0);
#pragma unroll
        for (int t2 = 0; t2 < 4; t2++) pr[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t2].w, acc[t].w, pr[t2], 0, 0, 0);
    }
    }
    float4 ss = make_float4(0.f, 0.f, 0.f, 0.f), st = ss;
#pragma unroll
    for (int t2 = 0; t2 < 4; t2++) {
        if (valid)
            *reinterpret_cast<float4*>(proj_out + (size_t)node * GAT_F + 16 * t2 + 4 * g) = make_float4(pr[t2].x, pr[t2].y, pr[t2].z, pr[t2].w);
        const float4 as = reinterpret_cast<const float4*>(w.a_src)[4 * t2 + g], at = reinterpret_cast<const float4*>(w.a_tgt)[4 * t2 + g];
        ss.x += pr[t2].x * as.x; ss.y += pr[t2].y * as.y; ss.z += pr[t2].z * as.z; ss.w += pr[t2].w * as.w;
        st.x += pr[t2].x * at.x; st.y += pr[t2].y * at.y; st.z += pr[t2].z * at.z; st.w += pr[t2].w * at.w;
    }
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) {
        ss.x += __shfl_xor(ss.x, m, 64); ss.y += __shfl_xor(ss.y, m, 64); ss.z += __shfl_xor(ss.z, m, 64); ss.w += __shfl_xor(ss.w, m, 64);
        st.x += __shfl_xor(st.x, m, 64); st.y += __shfl_xor(st.y, m, 64); st.z += __shfl_xor(st.z, m, 64); st.w += __shfl_xor(st.w, m, 64);
    }
    if (valid && g == 0) {
        reinterpret_cast<float4*>(scores_out)[node * 2 + 0] = ss;
        reinterpret_cast<float4*>(scores_out)[node * 2 + 1] = st;
    }
  }  // tiles
    if constexpr (SPLIT) {
        if (__any(!(vmax < 6.0e4f))) {
            if (lane == 0) atomicOr(w.range_flag, 1);
        }
    }
}

// ---------------------------------------------------------------- graph-resident GAT: all five layers + readout in ONE launch
// The per-layer kernel above moves ~1.1 KB per node and layer (projection + skip rows in, projection + skip rows out): 1.9x what
// the layer needs, because o_l = ELU(msg + W_skip skip_l) has to reach the next launch twice (as skip_{l+1} and, projected, as
// proj_{l+1}).  Here a persistent 16-wave workgroup (one per CU) owns a tile of WHOLE graphs (GraphTiles: <= 256 rows, <= 1 280
// in-edges; molhiv's largest graph has 222 nodes) across all five layers, as the FPGA keeps one graph in BRAM across its layer loop (GAT/src/GAT_compute.cc:60-100):
//   * o_l never leaves the registers of the wave that owns the row (it is the B operand of both 64 x 64 contractions);
//   * proj_l and the attention scores of the tile live in LDS (rows rotated by their index, as above); the gather has no
//     out-of-tile case at all, because tiles are made of whole graphs;
//   * the layer's 32 KiB of split weight fragments stream L2 -> LDS under the layer's gather (which does not use them);
//   * per node the launch reads 36 B of features + 1 B per in-edge + row bounds, and writes 4 B per GRAPH.
// Two barriers per layer: gathers done -> projections may be overwritten; projections written -> the next gather may start.
constexpr int GATR_ROWS = 256;
constexpr int GATR_EDGES = 1280;
constexpr int GATR_WAVES = 16;

// per layer in device memory (GatModel::d_res_): [W_skip_l 16 KiB][W_lin_{l+1} 16 KiB][score tile 4 KiB], split-f16 fragments.  The
// score tile is the 16 x 64 matrix whose rows 0..3 are a_src[l+1][h] . W_lin_{l+1} (head h) and rows 4..7 the same with a_tgt: the
// next layer's attention scores (node_embedding.cc:235-268) come out of the same MFMA chain as the projection, already in the lanes
// that store them (g = 0: ssrc, g = 1: stgt), instead of 32 multiply-adds and 16 cross-lane shuffles per lane.
constexpr int GATR_LAYER_BYTES = 36 * 1024;
struct GatResidentDev {
    const uint8_t* layers;     // [5][GATR_LAYER_BYTES]
    const float* scales;       // [3][5]: what undoes the power-of-two scale of W_skip, W_lin, score tile (device memory: a run-time
                               // index into a kernel argument would go through scratch)
    const float* a_src;        // [5][16 dim][4 head] (layer 0's scores)
    const float* a_tgt;
    const float* lin0;         // [16 dim][9][4 head]
    const float* pool_w;       // [16]
    const float* pool_b;
    // the readout folded through the LAST layer's skip contraction: logit terms are linear in o_3, so sum_{d,h} (pw[d] / 4)
    // (W_skip_4 o_3)[d][h] = o_3 . u4 with u4 = W_skip_4^T v, v[(d, h)] = pw[d] / 4 -- the last layer needs no MFMA and no weights
    const float* u4;           // [64]
    int* range_flag;
};

#define GATR_ABSMAX(v, a, b) asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(v) : "v"(a), "v"(b))

constexpr int GATR_PS = 17;
constexpr float GATR_LOG2E = 1.4426950408889634f;

__global__ __launch_bounds__(GATR_WAVES * 64, 4) void gat_resident_kernel(const int* __restrict__ node_feature, const int* __restrict__ feat_row,
                                                                         const int* __restrict__ row_ptr, const int* __restrict__ src,
                                                                         const int* __restrict__ tile_row, const int* __restrict__ tile_graph,
                                                                         const int* __restrict__ node_off, float* __restrict__ out, int n_tiles,
                                                                         GatResidentDev w, int ablate_arg) {
    const int ablate = FG_ABLATE(ablate_arg);  // 0 in the shipped build: the branches below fold away (common.h)
    (void)ablate_arg;
    const bool sort_rows = !(ablate & 4);  // development aid: gat_ablate, -DFLOWGNN_DEV builds=4 keeps rows in natural order
    __shared__ __attribute__((aligned(16))) char s_w[GATR_LAYER_BYTES];  // this layer's fragments
    // projections: 16 float4 (dims, heads in the float4) per row at a stride of GATR_PS = 17 float4 -- row u starts in bank group
    // u mod 16, so the sixteen rows a gather instruction reads spread over the groups as the former per-row rotation did, but the
    // dim index is a compile-time offset (the rotation cost ~10 VALU instructions per edge and lane in a VALU-issue-bound kernel)
    __shared__ __attribute__((aligned(16))) float4 s_proj[(GATR_ROWS + 1) * GATR_PS];  // + the no-edge row (zeros; below)
    __shared__ __attribute__((aligned(16))) float4 s_sc[(GATR_ROWS + 1) * 2];         // + the no-edge row (score -inf)
    __shared__ __attribute__((aligned(16))) float4 s_lin0[GAT_D * ND_FEATURE];
    __shared__ int s_feat[GATR_ROWS * ND_FEATURE];
    __shared__ __attribute__((aligned(4))) uint8_t s_src[GATR_EDGES];
    __shared__ uint16_t s_rp[GATR_ROWS + 2];
    __shared__ float s_dot[GATR_ROWS];
    __shared__ __attribute__((aligned(16))) float4 s_att[2 * GAT_D];  // a_src | a_tgt of layer 0 (heads in the float4)
    __shared__ float s_pw[GAT_D];
    __shared__ __attribute__((aligned(16))) float s_u4[GAT_F];
    // Column owner table: the walk of a 16-lane group takes as many trips as its LONGEST row, so the tile's rows are dealt to the
    // groups in order of decreasing in-degree (counting sort per tile; order inside a degree class is whatever the LDS atomics
    // gave -- placement never changes a row's arithmetic: MFMA columns are independent and a row is summed in CSR order by one lane)
    __shared__ uint8_t s_perm[GATR_ROWS];
    __shared__ __attribute__((aligned(16))) int s_cnt[16], s_cur[16];
    constexpr int NT = GATR_WAVES * 64;
    int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < GAT_D * ND_FEATURE; i += NT) s_lin0[i] = reinterpret_cast<const float4*>(w.lin0)[i];
    if (threadIdx.x < 2 * GAT_D) s_att[threadIdx.x] = reinterpret_cast<const float4*>(threadIdx.x < GAT_D ? w.a_src : w.a_tgt)[threadIdx.x & (GAT_D - 1)];
    if (threadIdx.x < GAT_D) s_pw[threadIdx.x] = w.pool_w[threadIdx.x];
    if (threadIdx.x < GAT_F) s_u4[threadIdx.x] = w.u4[threadIdx.x];
    // The no-edge row GATR_ROWS: a lane whose row has no in-edge left keeps walking it -- score -inf, so exp2(leaky(s_v + -inf)) = +0
    // is added to the denominator and 0 * 0 to the numerators, which leaves every sum's bits as they were (written once per workgroup)
    if (threadIdx.x < GATR_PS) s_proj[GATR_ROWS * GATR_PS + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < 2) s_sc[GATR_ROWS * 2 + threadIdx.x] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    // the layers' three dequantisation scales, staged once: read from global memory where they are used, each cost its wave an
    // exposed L2 round trip (the load sits directly in front of its first use), two per layer and tile
    __shared__ float s_scales[3 * GAT_L + 1];
    if (threadIdx.x < 3 * GAT_L) s_scales[threadIdx.x] = w.scales[threadIdx.x];
    const float pool_bias = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w.pool_b[0])));  // (an SGPR)
    float vmax = 0.0f;
    const uint32_t sw_addr = lds_addr_of(s_w);
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    // ---- a tile's inputs travel through registers: requested during the previous tile's layers, stored to LDS when that tile is done
    int fpre[3] = {0, 0, 0};  // feature words (256 rows x 9)
    int spre[2] = {0, 0};     // CSR sources (row inside the tile)
    int rpre = 0;             // row offsets
    int t0 = tile_row[tile], rows = tile_row[tile + 1] - t0;
    if (rows > GATR_ROWS) rows = GATR_ROWS;
    int g0 = tile_graph[tile], g1 = tile_graph[tile + 1];
    int e0 = row_ptr[t0], ne = row_ptr[t0 + rows] - e0;
    if (ne > GATR_EDGES) ne = GATR_EDGES;  // cannot happen for a validated batch (the host packed by edge count)
    auto fetch_tile = [&](int ft0, int frows, int fe0, int fne) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = threadIdx.x + NT * k;
            if (i < GATR_ROWS * ND_FEATURE) {
                const int r = i / ND_FEATURE, c = i - r * ND_FEATURE;
                const long long v = (long long)ft0 + (r < frows ? r : 0);
                const long long row = feat_row ? (long long)feat_row[v] : v;
                fpre[k] = node_feature[(size_t)row * ND_FEATURE + c];
            }
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = threadIdx.x + NT * k;
            if (i < fne) spre[k] = src[fe0 + i];  // RAW: (source - tile start, offset - first edge, clamps) are applied where the words are
        }                                          // stored to LDS -- arithmetic on a value here puts its whole global round trip in front of it
        if ((int)threadIdx.x <= frows) rpre = row_ptr[ft0 + threadIdx.x];
    };
    fetch_tile(t0, rows, e0, ne);
    while (true) {
        asm volatile("" : "+v"(lane));  // (per tile: nothing computed from the lane id is hoisted out of the tile loop and spilled: -0.6 %)
        const int j = lane & 15, g = lane >> 4;
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < n_tiles;
        int nt0 = 0, nrows = 0, ng0 = 0, ng1 = 0, ne0 = 0, nne = 0;
        __syncthreads();  // the previous tile's readout has read s_dot; its LDS state is dead
        if (threadIdx.x < 16) { s_cnt[threadIdx.x] = 0; s_cur[threadIdx.x] = 0; }
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (threadIdx.x + NT * k < GATR_ROWS * ND_FEATURE) s_feat[threadIdx.x + NT * k] = fpre[k];
#pragma unroll
        for (int k = 0; k < 2; k++)
            if ((int)threadIdx.x + NT * k < ne) s_src[threadIdx.x + NT * k] = (uint8_t)((spre[k] - t0) & 255);
        if ((int)threadIdx.x <= rows) {
            const int o = rpre - e0;
            s_rp[threadIdx.x] = (uint16_t)(o < 0 ? 0 : (o > ne ? ne : o));
        }
        __syncthreads();
        int skey = 15;  // in-degree class of row threadIdx.x: 0 = longest (>= 14 in-edges) .. 14 = none, 15 = no such row
        if (threadIdx.x < GATR_ROWS) {
            if ((int)threadIdx.x < rows) {
                const int deg = (int)s_rp[threadIdx.x + 1] - (int)s_rp[threadIdx.x];
                skey = 14 - (deg < 14 ? deg : 14);
            }
            if (sort_rows) atomicAdd(&s_cnt[skey], 1);
        }
        // ---- layer 0: proj_0 of (row, dim) from the row's nine features (load_inputs.cc:184-201), heads in the float4
        {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));  // opaque per tile: this lane's nine W_lin0 rows must not stay in registers across the tile
            const int d = tid & 15;
            float4 wl[ND_FEATURE];
#pragma unroll
            for (int k = 0; k < ND_FEATURE; k++) wl[k] = s_lin0[d * ND_FEATURE + k];
#pragma unroll
            for (int k = 0; k < GATR_ROWS * GAT_D / NT; k++) {
                const int r = (tid + NT * k) >> 4;
                s_proj[r * GATR_PS + d] = gat_proj0(&s_feat[r * ND_FEATURE], wl);
            }
        }
        __syncthreads();
        if (threadIdx.x < GATR_ROWS) {  // layer-0 scores, dims in ascending order (load_inputs.cc:203-224)
            const int r = threadIdx.x;
            float4 ss = make_float4(0.f, 0.f, 0.f, 0.f), st = ss;
#pragma unroll
            for (int d = 0; d < GAT_D; d++) {
                const float4 pd = s_proj[r * GATR_PS + d];
                gat_score_acc(ss, pd, s_att[d]);
                gat_score_acc(st, pd, s_att[GAT_D + d]);
            }
            // scores are kept pre-multiplied by log2(e): leaky-ReLU is positively homogeneous, so exp(leaky(s)) = exp2(leaky(s log2 e))
            // and the gather's exponentials are bare v_exp_f32 (one multiply less per head, edge and lane)
            s_sc[r * 2 + 0] = make_float4(ss.x * GATR_LOG2E, ss.y * GATR_LOG2E, ss.z * GATR_LOG2E, ss.w * GATR_LOG2E);
            s_sc[r * 2 + 1] = make_float4(st.x * GATR_LOG2E, st.y * GATR_LOG2E, st.z * GATR_LOG2E, st.w * GATR_LOG2E);
            int pos = r;
            if (sort_rows) {
                pos = atomicAdd(&s_cur[skey], 1);
                // (the 15 class counts as four 16-byte reads and selects: `for (k < skey) pos += s_cnt[k]` was an LDS round trip per class)
                int cc[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int4 c4 = *reinterpret_cast<const int4*>(s_cnt + 4 * q);
                    cc[4 * q + 0] = c4.x; cc[4 * q + 1] = c4.y; cc[4 * q + 2] = c4.z; cc[4 * q + 3] = c4.w;
                }
#pragma unroll
                for (int k = 0; k < 15; k++) pos += k < skey ? cc[k] : 0;
            }
            s_perm[pos] = (uint8_t)r;
        }
        __syncthreads();
        // Which wave walks which 16 rows of the degree order: waves w, w + 4, w + 8, w + 12 share a SIMD, and dealt in order SIMD 0 walked
        // groups 0, 4, 8, 12 and SIMD 3 groups 3, 7, 11, 15.  Snake order over the SIMDs (groups 0 7 8 15 | 1 6 9 14 | 2 5 10 13 | 3 4 11 12)
        // gives every SIMD the same share of long and short walks: gat_resident -1.2 % (a Latin square: -0.8 %; scripts/dev/ab.py)
        const int grp = (wv >> 2) * 4 + (((wv >> 2) & 1) ? 3 - (wv & 3) : (wv & 3));
        const int r = s_perm[grp * 16 + j];
        const bool valid = r < rows;
        const int rr = valid ? r : 0;  // rows past the tile's end repeat row 0's self edge (finite values, never used)
        const int e_begin = valid ? (int)s_rp[r] : 0, e_end = valid && !(ablate & 1) ? (int)s_rp[r + 1] : e_begin;  // ablate: development aid (gat_ablate, -DFLOWGNN_DEV builds)
        // the skip input as split B operand; K-slot e of K-step ks <-> feature 16 (2 ks + (e >> 2)) + 4 g + (e & 3).  Layer 0: the raw
        // features (dims 0..8 of head 0, load_inputs.cc:190-191); later o_{l-1}, whose split for the projection IS this operand
        float head_pre = 0.0f;
        ds_uint4_t b_hi[2], b_lo[2];
        {
            float bq[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int d = 4 * q + g;
                bq[4 * q + 0] = d < ND_FEATURE ? (float)s_feat[rr * ND_FEATURE + d] : 0.0f;
                bq[4 * q + 1] = 0.0f; bq[4 * q + 2] = 0.0f; bq[4 * q + 3] = 0.0f;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                DS_SPLIT2(bq[8 * ks + 0], bq[8 * ks + 1], b_hi[ks].x, b_lo[ks].x);
                DS_SPLIT2(bq[8 * ks + 2], bq[8 * ks + 3], b_hi[ks].y, b_lo[ks].y);
                DS_SPLIT2(bq[8 * ks + 4], bq[8 * ks + 5], b_hi[ks].z, b_lo[ks].z);
                DS_SPLIT2(bq[8 * ks + 6], bq[8 * ks + 7], b_hi[ks].w, b_lo[ks].w);
            }
            GATR_ABSMAX(vmax, bq[0], bq[4]);
            GATR_ABSMAX(vmax, bq[8], bq[12]);
        }
        int ro_n0 = 0, ro_n1 = 1;
        // trips of the walk: the self edge + the wave's longest row, wave-uniform, once per tile (the rows are the same in every layer)
        int trips = e_end - e_begin;
#pragma unroll
        for (int mk = 1; mk < 64; mk <<= 1) trips = max(trips, __shfl_xor(trips, mk, 64));
        trips = __builtin_amdgcn_readfirstlane(trips) + 1;
#pragma unroll 1
        for (int l = 0; l < GAT_L; l++) {
            // this layer's fragments stream in under the gather, which does not use them: 36 pieces of 1 KiB (last layer: W_skip only)
            if (l < GAT_L - 1) {  // (the last layer needs none: its skip contraction is folded into the readout, u4)
                const uint8_t* gl = w.layers + (size_t)l * GATR_LAYER_BYTES;
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    const int piece = wv + GATR_WAVES * p;
                    if (piece < 36) lds_dma16(gl + piece * 1024, (uint32_t)lane * 16u, sw_addr + piece * 1024);
                }
            }
            if (l == 1 && has_next) {  // the next tile's descriptor (two dependent scalar round trips), used from layer 3 on
                nt0 = tile_row[ntile];
                nrows = tile_row[ntile + 1] - nt0;
                if (nrows > GATR_ROWS) nrows = GATR_ROWS;
                ng0 = tile_graph[ntile]; ng1 = tile_graph[ntile + 1];
                ne0 = row_ptr[nt0];
                nne = row_ptr[nt0 + nrows] - ne0;
                if (nne > GATR_EDGES) nne = GATR_EDGES;
            }
            if (l == 3 && has_next) fetch_tile(nt0, nrows, ne0, nne);  // lands during the last two layers
            // the readout's node range of "this lane's graph", a whole gather ahead of its use (requested at the readout, the L2 round
            // trip was paid by one wave while the other fifteen waited at the tile's first barrier)
            if (l == GAT_L - 1 && g0 + (int)threadIdx.x < g1) { ro_n0 = node_off[g0 + threadIdx.x]; ro_n1 = node_off[g0 + threadIdx.x + 1]; }
            // ---- attention gather (pull): self edge first, then the CSR row (ascending source); everything out of LDS
            const float4 ssrc = s_sc[rr * 2 + 0];
            float4 den = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 num[4];
#pragma unroll
            for (int t = 0; t < 4; t++) num[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            {
                // Branch-free trips over a wave-uniform count (as gcn_resident_kernel's and gin_resident_kernel's walks).  The former
                // `while (any lane has an edge) if (this lane has one) ...` cost, per trip, ten 64-bit moves of the accumulators (hipcc
                // keeps a copy of every loop-carried value for the lanes that sit a trip out), four branches and a dozen exec-mask
                // instructions beside the 26 VALU instructions that are the work -- in a kernel bound by VALU issue.
                int e = e_begin;
                int u = rr;  // trip 0: the self edge
                int u_nx = e < e_end ? (int)s_src[e] : GATR_ROWS;
                // accumulators as register PAIRS (heads 0,1 | 2,3): one v_pk_fma_f32 per pair, no operand shuffles
                float2_t dn[2] = {{0.f, 0.f}, {0.f, 0.f}}, nm[4][2];
#pragma unroll
                for (int q = 0; q < 4; q++) { nm[q][0] = (float2_t){0.f, 0.f}; nm[q][1] = (float2_t){0.f, 0.f}; }
#pragma unroll 1
                for (int t = 0; t < trips; t++) {
                    const float4_t st = *reinterpret_cast<const float4_t*>(&s_sc[u * 2 + 1]);
                    float4_t p[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) p[q] = *reinterpret_cast<const float4_t*>(&s_proj[u * GATR_PS + 4 * q + g]);
                    u = u_nx;
                    e++;
                    const bool more = e < e_end;
                    const int nx = (int)s_src[more ? e : 0];
                    u_nx = more ? nx : GATR_ROWS;
                    __builtin_amdgcn_sched_barrier(0);  // the trip's six reads are in flight before the first of them is waited for
                    float4_t sv = {ssrc.x + st.x, ssrc.y + st.y, ssrc.z + st.z, ssrc.w + st.w};
                    // leaky_0.2(x) = max(x, 0.2 x)
                    sv.x = __builtin_amdgcn_exp2f(__builtin_fmaxf(sv.x, sv.x * 0.2f)); sv.y = __builtin_amdgcn_exp2f(__builtin_fmaxf(sv.y, sv.y * 0.2f));
                    sv.z = __builtin_amdgcn_exp2f(__builtin_fmaxf(sv.z, sv.z * 0.2f)); sv.w = __builtin_amdgcn_exp2f(__builtin_fmaxf(sv.w, sv.w * 0.2f));
                    dn[0] += sv.lo; dn[1] += sv.hi;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        nm[q][0] = __builtin_elementwise_fma(sv.lo, p[q].lo, nm[q][0]);
                        nm[q][1] = __builtin_elementwise_fma(sv.hi, p[q].hi, nm[q][1]);
                    }
                }
                den = make_float4(dn[0].x, dn[0].y, dn[1].x, dn[1].y);
#pragma unroll
                for (int q = 0; q < 4; q++) num[q] = make_float4(nm[q][0].x, nm[q][0].y, nm[q][1].x, nm[q][1].y);
            }
            float4_t acc[4];  // rows 16 t + 4 g + r' = (dim 4 t + g, head r') of this lane's node
            {   // msg = num / den: one v_rcp_f32 per head (1 ulp; sixteen IEEE divisions are ~160 dependent instructions per lane and layer)
                const float4 rd = make_float4(__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y), __builtin_amdgcn_rcpf(den.z), __builtin_amdgcn_rcpf(den.w));
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = (float4_t){num[t].x * rd.x, num[t].y * rd.y, num[t].z * rd.z, num[t].w * rd.w};
            }
            if (l == GAT_L - 1) {
                // emb[v][d] = mean_h(msg + W_skip_4 o_3)[d][h]; the logit is mean_v(emb[v]) . w + b = mean_v(emb[v] . w) + b
                // (finalize.cc:46-112), and emb[v] . w = sum (pw[d] / 4) msg[d][h] + o_3 . u4 (head_pre, taken when o_3 was in registers)
                float part = head_pre;
#pragma unroll
                for (int t = 0; t < 4; t++) part += (acc[t].x + acc[t].y + acc[t].z + acc[t].w) / (float)GAT_H * s_pw[4 * t + g];
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                if (g == 0) s_dot[r] = part;
                break;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // #1: every gather of this layer is done (s_proj / s_sc may be rewritten); the fragments have landed
            const char* wb = s_w;
            const float sk_scale = s_scales[l];
            if (!(ablate & 2)) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    float4_t sk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) {
                        const ds_uint4_t a_hi = *reinterpret_cast<const ds_uint4_t*>(wb + ((t * 2 + ks) * 2 + 0) * 1024 + lane * 16);
                        const ds_uint4_t a_lo = *reinterpret_cast<const ds_uint4_t*>(wb + ((t * 2 + ks) * 2 + 1) * 1024 + lane * 16);
                        sk = DS_MFMA16(a_hi, b_hi[ks], sk);
                        sk = DS_MFMA16(a_hi, b_lo[ks], sk);
                        sk = DS_MFMA16(a_lo, b_hi[ks], sk);
                    }
                    acc[t] += sk * sk_scale;
                }
            }
            // ---- ELU; o_l split once: B operand of the projection now and of the next layer's skip contraction
#pragma unroll
            for (int t = 0; t < 4; t++) {
                acc[t].x = acc[t].x <= 0.f ? __expf(acc[t].x) - 1.0f : acc[t].x; acc[t].y = acc[t].y <= 0.f ? __expf(acc[t].y) - 1.0f : acc[t].y;
                acc[t].z = acc[t].z <= 0.f ? __expf(acc[t].z) - 1.0f : acc[t].z; acc[t].w = acc[t].w <= 0.f ? __expf(acc[t].w) - 1.0f : acc[t].w;
            }
            if (l == GAT_L - 2) {  // o_3 . u4: this lane's 16 features (16 t + 4 g + r'), the node's 4 lanes are summed with the message part
                head_pre = 0.0f;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float4 uu = *reinterpret_cast<const float4*>(s_u4 + 16 * t + 4 * g);
                    head_pre += acc[t].x * uu.x; head_pre += acc[t].y * uu.y; head_pre += acc[t].z * uu.z; head_pre += acc[t].w * uu.w;
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {  // output tiles 2 ks, 2 ks + 1 of the skip contraction are K-step ks of the next two
                DS_SPLIT2(acc[2 * ks].x, acc[2 * ks].y, b_hi[ks].x, b_lo[ks].x);
                DS_SPLIT2(acc[2 * ks].z, acc[2 * ks].w, b_hi[ks].y, b_lo[ks].y);
                DS_SPLIT2(acc[2 * ks + 1].x, acc[2 * ks + 1].y, b_hi[ks].z, b_lo[ks].z);
                DS_SPLIT2(acc[2 * ks + 1].z, acc[2 * ks + 1].w, b_hi[ks].w, b_lo[ks].w);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                GATR_ABSMAX(vmax, acc[t].x, acc[t].y);
                GATR_ABSMAX(vmax, acc[t].z, acc[t].w);
            }
            // ---- next projection and next scores (five output tiles of one MFMA chain)
            float4_t pr[5];
#pragma unroll
            for (int t2 = 0; t2 < 5; t2++) pr[t2] = (float4_t){0.f, 0.f, 0.f, 0.f};
            if (!(ablate & 2)) {
#pragma unroll
                for (int t2 = 0; t2 < 5; t2++) {
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) {
                        const ds_uint4_t a_hi = *reinterpret_cast<const ds_uint4_t*>(wb + 16384 + ((t2 * 2 + ks) * 2 + 0) * 1024 + lane * 16);
                        const ds_uint4_t a_lo = *reinterpret_cast<const ds_uint4_t*>(wb + 16384 + ((t2 * 2 + ks) * 2 + 1) * 1024 + lane * 16);
                        pr[t2] = DS_MFMA16(a_hi, b_hi[ks], pr[t2]);
                        pr[t2] = DS_MFMA16(a_hi, b_lo[ks], pr[t2]);
                        pr[t2] = DS_MFMA16(a_lo, b_hi[ks], pr[t2]);
                    }
                }
            }
            const float lin_scale = s_scales[GAT_L + l], sc_scale = s_scales[2 * GAT_L + l];
#pragma unroll
            for (int t2 = 0; t2 < 4; t2++)
                s_proj[r * GATR_PS + 4 * t2 + g] = make_float4(pr[t2].x * lin_scale, pr[t2].y * lin_scale, pr[t2].z * lin_scale, pr[t2].w * lin_scale);
            if (g < 2) {
                const float sl = sc_scale * GATR_LOG2E;  // (scores pre-multiplied by log2 e, see the layer-0 scores above)
                s_sc[r * 2 + g] = make_float4(pr[4].x * sl, pr[4].y * sl, pr[4].z * sl, pr[4].w * sl);
            }
            __syncthreads();  // #2: the next layer's projections and scores are complete
        }
        __syncthreads();  // the per-node readout terms are in s_dot
        if (g0 + (int)threadIdx.x < g1) out[g0 + threadIdx.x] = lds_sum_in_order(s_dot + (ro_n0 - t0), ro_n1 - ro_n0) / (float)(ro_n1 - ro_n0) + pool_bias;
        if (!has_next) break;
        tile = ntile; t0 = nt0; rows = nrows; g0 = ng0; g1 = ng1; e0 = ne0; ne = nne;
    }
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(w.range_flag, 1);
    }
}
#undef GATR_ABSMAX

class GatModel : public Model {
public:
    ~GatModel() override { free_all(); }
    int emb_dim() const override { return GAT_F; }
    int scratch_dim() const override { return 2 * GAT_F + 2 * 8 + GAT_D + 1; }  // skip x2, scores x2, emb, feat_row
    bool has_edge_attr() const override { return false; }
    int num_weight_tensors() const override { return 6; }
    bool weights_ready() const override { return ready_; }
    void configure(const Options& o) override {
        split_ = o.i("gat_mfma") != 32;
        reference_quirk_ = o.on("gat_reference_quirk");
        fold_readout_ = o.on("gat_fold_readout");
        resident_ = o.on("gat_resident");
        ablate_ = FG_ABLATE(o.i("gat_ablate"));
    }
    void set_exact(bool on) override { exact_ = on; }

    // host tensors (GAT/src/dcl.h:86-93): scoring_fn_target[5][4][16], scoring_fn_source[5][4][16],
    // linear_proj[5][4][16][4][16], skip_proj[5][4][16][4][16] (layer 0: only [ho][do][0][<9] is used), pred_w[1][16], pred_b[1]
    int set_numeric_mode(int mode) override {
        if (mode != 0 && mode != 1) return 8;
        qmode_ = mode == 1;
        return 0;
    }

    int set_weights(const float* const* t) override {
        {   // ap_fixed<16,6> copies of every tensor for the bit-faithful mode (modelq.hip)
            static const size_t elems[6] = {5 * 4 * 16, 5 * 4 * 16, 5 * 4 * 16 * 4 * 16, 5 * 4 * 16 * 4 * 16, 16, 1};
            if (int rc = q_.upload_all(6, t, elems, 10)) return rc;
        }
        const float *tgt = t[0], *srcw = t[1], *lin = t[2], *skip = t[3];
        auto W = [](const float* w, int l, int ho, int dout, int hi, int din) {
            return w[(((((size_t)l * GAT_H + ho) * GAT_D + dout) * GAT_H + hi) * GAT_D) + din];
        };
        std::vector<float> lin0((size_t)GAT_D * ND_FEATURE * GAT_H);
        for (int d = 0; d < GAT_D; d++)
            for (int k = 0; k < ND_FEATURE; k++)
                for (int h = 0; h < GAT_H; h++) lin0[((size_t)d * ND_FEATURE + k) * GAT_H + h] = W(lin, 0, h, d, 0, k);
        std::vector<float> asrc((size_t)GAT_L * GAT_D * GAT_H), atgt(asrc.size());
        for (int l = 0; l < GAT_L; l++)
            for (int d = 0; d < GAT_D; d++)
                for (int h = 0; h < GAT_H; h++) {
                    asrc[((size_t)l * GAT_D + d) * GAT_H + h] = srcw[((size_t)l * GAT_H + h) * GAT_D + d];
                    atgt[((size_t)l * GAT_D + d) * GAT_H + h] = tgt[((size_t)l * GAT_H + h) * GAT_D + d];
                }
        // dense 64 x 64 views, f = dim * 4 + head, then MFMA fragments
        std::vector<float> wskip((size_t)GAT_L * 16 * 64 * 4), wlin((size_t)GAT_L * 16 * 64 * 4, 0.0f);
        auto M = [&](const float* w, int l, int fo, int fi) { return W(w, l, fo & 3, fo >> 2, fi & 3, fi >> 2); };
        for (int l = 0; l < GAT_L; l++)
            for (int a = 0; a < 4; a++)
                for (int b = 0; b < 4; b++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int r = 0; r < 4; r++) {
                            const int i = lane & 15, g = lane >> 4;
                            // skip: (t = a, q = b): rows 16 t + i, cols 16 q + 4 g + r
                            wskip[((((size_t)l * 4 + a) * 4 + b) * 64 + lane) * 4 + r] = M(skip, l, 16 * a + i, 16 * b + 4 * g + r);
                            // lin of layer l+1, consumed by layer l: (t = a, t2 = b): rows 16 t2 + i, cols 16 t + 4 g + r
                            if (l + 1 < GAT_L)
                                wlin[((((size_t)l * 4 + a) * 4 + b) * 64 + lane) * 4 + r] = M(lin, l + 1, 16 * b + i, 16 * a + 4 * g + r);
                        }
        // split-f16 fragments of the same matrices (device layout in GatLayerDev), one power-of-two scale per matrix
        std::vector<float> wskip_s((size_t)GAT_L * 4096, 0.0f), wlin_s((size_t)GAT_L * 4096, 0.0f);
        auto pack_split = [&](const float* w, int l, float* out_f, float& inv_scale) {
            float mx = 0.0f;
            for (int fo = 0; fo < 64; fo++)
                for (int fi = 0; fi < 64; fi++) mx = std::fmax(mx, std::fabs(M(w, l, fo, fi)));
            const float sc = (mx > 0.0f && std::isfinite(mx)) ? std::ldexp(1.0f, -std::ilogb(mx)) : 1.0f;
            inv_scale = 1.0f / sc;
            uint8_t* out = reinterpret_cast<uint8_t*>(out_f);
            for (int tt = 0; tt < 4; tt++)
                for (int ks = 0; ks < 2; ks++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int e = 0; e < 8; e++) {
                            const int i = lane & 15, gk = lane >> 4;
                            const float v = M(w, l, 16 * tt + i, 16 * (2 * ks + (e >> 2)) + 4 * gk + (e & 3)) * sc;
                            const _Float16 hi = (_Float16)v;
                            const _Float16 lo = (_Float16)(v - (float)hi);
                            uint8_t* f = out + (size_t)((tt * 2 + ks) * 2) * 1024 + lane * 16 + e * 2;
                            std::memcpy(f, &hi, 2);
                            std::memcpy(f + 1024, &lo, 2);
                        }
        };
        for (int l = 0; l < GAT_L; l++) {
            pack_split(skip, l, &wskip_s[(size_t)l * 4096], wskip_scale_[l]);
            if (l + 1 < GAT_L) pack_split(lin, l + 1, &wlin_s[(size_t)l * 4096], wlin_scale_[l]);
        }
        std::vector<float> v_pw(t[4], t[4] + GAT_D), v_pb(t[5], t[5] + 1);
        int rc;
        if ((rc = upload(&d_wskip_s_, wskip_s))) return rc;
        {   // the graph-resident kernel's per-layer stream (GATR_LAYER_BYTES): W_skip | W_lin | score tile
            std::vector<uint8_t> res((size_t)GAT_L * GATR_LAYER_BYTES, 0);
            std::vector<float> sc(3 * GAT_L, 1.0f);
            for (int l = 0; l < GAT_L; l++) {
                uint8_t* base = res.data() + (size_t)l * GATR_LAYER_BYTES;
                std::memcpy(base, &wskip_s[(size_t)l * 4096], 16384);
                std::memcpy(base + 16384, &wlin_s[(size_t)l * 4096], 16384);
                sc[l] = wskip_scale_[l];
                sc[GAT_L + l] = wlin_scale_[l];
                if (l + 1 >= GAT_L) continue;
                // score tile: row h = a_src[l+1][h] . W_lin_{l+1} restricted to head h's outputs, row 4 + h the same with a_tgt
                double S[16][64] = {};
                double mx = 0.0;
                for (int i = 0; i < 8; i++) {
                    const int h = i & 3;
                    const float* av = (i < 4 ? srcw : tgt) + ((size_t)(l + 1) * GAT_H + h) * GAT_D;
                    for (int fi = 0; fi < 64; fi++) {
                        double a = 0.0;
                        for (int d = 0; d < GAT_D; d++) a += (double)av[d] * (double)M(lin, l + 1, d * 4 + h, fi);
                        S[i][fi] = a;
                        mx = std::fmax(mx, std::fabs(a));
                    }
                }
                const double scl = (mx > 0.0 && std::isfinite(mx)) ? std::ldexp(1.0, -std::ilogb(mx)) : 1.0;
                sc[2 * GAT_L + l] = (float)(1.0 / scl);
                uint8_t* outp = base + 32768;
                for (int ks = 0; ks < 2; ks++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int e = 0; e < 8; e++) {
                            const int i = lane & 15, gk = lane >> 4;
                            const float v = (float)(S[i][16 * (2 * ks + (e >> 2)) + 4 * gk + (e & 3)] * scl);
                            const _Float16 hi = (_Float16)v;
                            const _Float16 lo = (_Float16)(v - (float)hi);
                            uint8_t* f = outp + (size_t)(ks * 2) * 1024 + lane * 16 + e * 2;
                            std::memcpy(f, &hi, 2);
                            std::memcpy(f + 1024, &lo, 2);
                        }
            }
            if ((rc = upload(&d_scales_, sc))) return rc;
            if ((rc = upload(&d_res_, res))) return rc;
            std::vector<float> u4(GAT_F);
            for (int fi = 0; fi < GAT_F; fi++) {
                double a = 0.0;
                for (int fo = 0; fo < GAT_F; fo++) a += (double)t[4][fo >> 2] / (double)GAT_H * (double)M(skip, GAT_L - 1, fo, fi);
                u4[fi] = (float)a;
            }
            if ((rc = upload(&d_u4_, u4))) return rc;
        }
        if ((rc = upload(&d_wlin_s_, wlin_s))) return rc;
        if ((rc = upload(&d_lin0_, lin0))) return rc;
        if ((rc = upload(&d_asrc_, asrc))) return rc;
        if ((rc = upload(&d_atgt_, atgt))) return rc;
        if ((rc = upload(&d_wskip_, wskip))) return rc;
        if ((rc = upload(&d_wlin_, wlin))) return rc;
        if ((rc = upload(&d_pw_, v_pw))) return rc;
        if ((rc = upload(&d_pb_, v_pb))) return rc;
        ready_ = true;
        return 0;
    }

    // GAT/src/host_load.cc:20-91: eight files; layer 0 lives in the [ho][do][head_in = 0][dim_in < 9] corner
    int load_weights_dir(const char* dir) override {
        std::vector<float> pw(16), pb(1), tgt(5 * 4 * 16), srcw(5 * 4 * 16), l0(4 * 16 * 9), l1((size_t)4 * 4 * 16 * 4 * 16), s0(4 * 16 * 9),
            s1((size_t)4 * 4 * 16 * 4 * 16);
        int rc;
        if ((rc = read_floats(dir, "gat_ep1_pred_weights_layer5.bin", 0, pw.size(), pw.data()))) return rc;
        if ((rc = read_floats(dir, "gat_ep1_pred_bias_layer5.bin", 0, 1, pb.data()))) return rc;
        if ((rc = read_floats(dir, "gat_ep1_scoring_fn_target_layer5.bin", 0, tgt.size(), tgt.data()))) return rc;
        if ((rc = read_floats(dir, "gat_ep1_scoring_fn_source_layer5.bin", 0, srcw.size(), srcw.data()))) return rc;
        if ((rc = read_floats(dir, "gat_ep1_linear_proj_weight_0_layer5.bin", 0, l0.size(), l0.data()))) return rc;
        if ((rc = read_floats(dir, "gat_ep1_linear_proj_weight_1_layer5.bin", 0, l1.size(), l1.data()))) return rc;
        if ((rc = read_floats(dir, "gat_ep1_skip_proj_weight_0_layer5.bin", 0, s0.size(), s0.data()))) return rc;
        if ((rc = read_floats(dir, "gat_ep1_skip_proj_weight_1_layer5.bin", 0, s1.size(), s1.data()))) return rc;
        std::vector<float> lin((size_t)5 * 4096, 0.0f), skip((size_t)5 * 4096, 0.0f);
        for (int ho = 0; ho < 4; ho++)
            for (int d = 0; d < 16; d++)
                for (int k = 0; k < 9; k++) {
                    lin[(((size_t)ho * 16 + d) * 4 + 0) * 16 + k] = l0[((size_t)ho * 16 + d) * 9 + k];
                    skip[(((size_t)ho * 16 + d) * 4 + 0) * 16 + k] = s0[((size_t)ho * 16 + d) * 9 + k];
                }
        memcpy(&lin[4096], l1.data(), sizeof(float) * l1.size());
        memcpy(&skip[4096], s1.data(), sizeof(float) * s1.size());
        const float* t[6] = {tgt.data(), srcw.data(), lin.data(), skip.data(), pw.data(), pb.data()};
        return set_weights(t);
    }

    // graph-resident kernel (gat_resident_kernel): whole graphs packed into tiles of <= 256 rows / 1 280 in-edges by flowgnn_set_batch
    void graph_tile_limits(int& rows, int& edges) const override {
        rows = resident_ ? GATR_ROWS : 0;
        edges = resident_ ? GATR_EDGES : 0;
    }
    void set_keep_h(bool on) override { keep_h_ = on; }

    int forward(DeviceBatch& db, Profiler& prof, hipStream_t s) override {
        const int n = db.b.n_tot;
        if (n <= 0) return 0;
        float* skipb[2] = {db.scratch, db.scratch + (size_t)n * GAT_F};
        float* scoreb[2] = {db.scratch + (size_t)n * 2 * GAT_F, db.scratch + (size_t)n * (2 * GAT_F + 8)};
        float* emb = db.scratch + (size_t)n * (2 * GAT_F + 16);
        int* feat_row = nullptr;
        // gat_reference_quirk=1: node features read without the per-graph offset (GAT_compute.cc:72)
        if (reference_quirk_) {
            feat_row = reinterpret_cast<int*>(db.scratch + (size_t)n * (2 * GAT_F + 16 + GAT_D));
            gat_local_rows_kernel<<<(db.b.num_graphs + 3) / 4, 256, 0, s>>>(db.b.node_off, feat_row, db.b.num_graphs);
        }
        if (qmode_) return gatq_forward(q_, db, feat_row, prof, s);
        // all five layers in one launch when the batch packs into graph tiles (tiles under half full, e.g. graphs of 65..128
        // nodes, waste MFMA columns: the per-layer kernels take those); per-node taps (flowgnn_get_h) come from the per-layer path
        if (resident_ && !keep_h_ && fold_readout_ && split_ && !exact_ && db.gtiles.ok && db.gtiles.n_tiles > 0 && db.gtiles.fill >= 0.5) {
            GatResidentDev rw;
            rw.layers = d_res_;
            rw.scales = d_scales_;
            rw.a_src = d_asrc_;
            rw.a_tgt = d_atgt_;
            rw.lin0 = d_lin0_;
            rw.pool_w = d_pw_;
            rw.pool_b = d_pb_;
            rw.u4 = d_u4_;
            rw.range_flag = db.range_flag;
            ProfScope p(prof, "gat_resident", s);
            const int grid = db.gtiles.n_tiles < 256 ? db.gtiles.n_tiles : 256;  // persistent: one 16-wave workgroup per CU (118 KB of LDS)
            gat_resident_kernel<<<grid, GATR_WAVES * 64, 0, s>>>(db.b.node_feature, feat_row, db.csr.row_ptr, db.csr.src, db.gtiles.row_start,
                                                               db.gtiles.graph_start, db.b.node_off, db.out, db.gtiles.n_tiles, rw,
                                                               ablate_);
            db.final_h = 0;
            db.tap = nullptr;
            db.h_valid = false;  // no per-node tensor leaves the kernel: flowgnn_get_h repeats the pass on the per-layer kernels
            return 0;
        }
        const GatLayer0Dev w0{d_lin0_, d_asrc_, d_atgt_};
        {
            ProfScope p(prof, "gat_scores0", s);
            gat_scores0_kernel<<<(n + 255) / 256, 256, 0, s>>>(db.b.node_feature, feat_row, d_lin0_, d_asrc_, d_atgt_, scoreb[0], n);
        }
        const bool fold = fold_readout_;
        int cur = 0;
        for (int l = 0; l < GAT_L; l++) {
            GatLayerDev w;
            w.wskip = d_wskip_ + (size_t)l * 16 * 64 * 4;
            w.wlin = d_wlin_ + (size_t)l * 16 * 64 * 4;
            w.a_src = d_asrc_ + (size_t)(l + 1 < GAT_L ? l + 1 : l) * GAT_D * GAT_H;
            w.a_tgt = d_atgt_ + (size_t)(l + 1 < GAT_L ? l + 1 : l) * GAT_D * GAT_H;
            w.wskip_split = d_wskip_s_ + (size_t)l * 4096;
            w.wlin_split = d_wlin_s_ + (size_t)l * 4096;
            w.wskip_scale = wskip_scale_[l];
            w.wlin_scale = wlin_scale_[l];
            w.range_flag = db.range_flag;
            const bool sp = split_ && !exact_;
            ProfScope p(prof, "gat_layer", s);
            const int n_tiles = (n + GAT_TR - 1) / GAT_TR;
            const int layer_grid = n_tiles < 512 ? n_tiles : 512;  // persistent: two 8-wave workgroups per CU (68 KB of LDS each)
#define GAT_LAUNCH(FIN, FST, ...)                                                              \
    do {                                                                                          \
        if (sp) gat_layer_kernel<FIN, FST, true><<<layer_grid, 512, 0, s>>>(__VA_ARGS__);         \
        else gat_layer_kernel<FIN, FST, false><<<layer_grid, 512, 0, s>>>(__VA_ARGS__);           \
    } while (0)
            if (l == 0) {
                GAT_LAUNCH(false, true, db.h[cur], skipb[cur], scoreb[cur], db.h[cur ^ 1], skipb[cur ^ 1],
                                                                        scoreb[cur ^ 1], emb, db.csr.row_ptr, db.csr.src, w, n,
                                                                        db.b.node_feature, feat_row, w0, nullptr);
                cur ^= 1;
            } else if (l < GAT_L - 1) {
                GAT_LAUNCH(false, false, db.h[cur], skipb[cur], scoreb[cur], db.h[cur ^ 1], skipb[cur ^ 1],
                                                                         scoreb[cur ^ 1], emb, db.csr.row_ptr, db.csr.src, w, n, nullptr, nullptr, w0, nullptr);
                cur ^= 1;
            } else {
                GAT_LAUNCH(true, false, db.h[cur], skipb[cur], scoreb[cur], nullptr, nullptr, nullptr, emb,
                                                                        db.csr.row_ptr, db.csr.src, w, n, nullptr, nullptr, w0,
                                                                        fold ? d_pw_ : nullptr);
            }
        }
#undef GAT_LAUNCH
        db.final_h = cur;
        db.tap = skipb[cur];  // ELU output of layer 3
        db.tap_dim = GAT_F;
        {
            ProfScope p(prof, "mean_pool_linear", s);
            if (fold)
                segment_mean_bias_kernel<0><<<(db.b.num_graphs + 255) / 256, 256, 0, s>>>(emb, db.b.node_off, d_pb_, db.out, db.b.num_graphs);
            else
                mean_pool_linear_kernel<GAT_D><<<(db.b.num_graphs + 3) / 4, 256, 0, s>>>(emb, db.b.node_off, d_pw_, d_pb_, db.out,
                                                                                         db.b.num_graphs);
        }
        return 0;
    }

private:
    void free_all() {
        float** ptrs[] = {&d_lin0_, &d_asrc_, &d_atgt_, &d_wskip_, &d_wlin_, &d_pw_, &d_pb_, &d_wskip_s_, &d_wlin_s_, &d_scales_, &d_u4_};
        for (auto p : ptrs)
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        if (d_res_) { (void)hipFree(d_res_); d_res_ = nullptr; }
        q_.release();
    }
    bool ready_ = false;
    bool qmode_ = false;  // flowgnn_set_numeric_mode(FLOWGNN_NUMERIC_Q6_10)
    QPack q_;
    // the two 64 x 64 contractions per layer as split-f16 products unless FLOWGNN_GAT_MFMA=f32; exact_ = the engine asked for the
    // fp32 pipe after an operand left the split's accurate range (flowgnn_sync)
    bool split_ = true;
    bool reference_quirk_ = false;
    bool exact_ = false;
    float wskip_scale_[GAT_L] = {}, wlin_scale_[GAT_L] = {};
    float *d_wskip_s_ = nullptr, *d_wlin_s_ = nullptr, *d_scales_ = nullptr, *d_u4_ = nullptr;
    uint8_t* d_res_ = nullptr;  // per-layer fragment stream of gat_resident_kernel
    bool fold_readout_ = true;
    int ablate_ = 0;  // development aid (-DFLOWGNN_DEV builds only, option gat_ablate): per-phase timing (scripts/dev/pna_ablate.sh)
    bool resident_ = true;
    bool keep_h_ = false;
    float *d_lin0_ = nullptr, *d_asrc_ = nullptr, *d_atgt_ = nullptr, *d_wskip_ = nullptr, *d_wlin_ = nullptr, *d_pw_ = nullptr,
          *d_pb_ = nullptr;
};

Model* make_gat_model() { return new GatModel(); }

}  // namespace fg
