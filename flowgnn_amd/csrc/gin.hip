// gin.hip -- GIN / GIN-VN hot path for gfx950 (MI355X).
//
// What the reference does per graph (one NT unit + 4 MP units, GIN/src/*.cc):
//   h0[v]   = sum_{k<9} NodeEmb[off_k + feat_k(v)]                          load_inputs.cc:193-212
//   m_l[v]  = sum_{(u->v)} relu(h_l[u] + sum_{k<3} EdgeEmb_l[off_k+attr_k])  message_passing.cc:136-145
//   a       = m_l[v] + (1 + eps) h_l[v],  eps == 0 (never loaded)            node_embedding.cc:117
//   hid     = b1 + W1 a  (200x100),  h_{l+1} = b2 + W2 relu(hid) (100x200)   node_embedding.cc:124-191
//   out[g]  = pb + pw . mean_v h_5[v]                                        finalize.cc:36-113
//
// Here: the whole batch is one super-graph in HBM, h is [N_tot][100] fp32 (row stride
// 100 floats, rows of a graph contiguous), and per layer there are two kernels:
//   gin_aggregate : HBM-bound gather + per-destination ordered sum over the CSR
//   gin_mlp       : the dense update on fp32 MFMA (v_mfma_f32_16x16x4_f32), transposed so
//                   that nodes are MFMA columns and the MLP1 accumulators feed MLP2 directly
//                   as B operands (no LDS round trip, no transpose)
#include "common.h"
#include "device_common.h"
#include "gin_split.h"
#include "ginq.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>

namespace fg {

constexpr int GIN_D = 100;
constexpr int GIN_H = 200;
constexpr int GIN_L = 5;
constexpr int GIN_C = GIN_D / 4;   // float4 chunks per row
constexpr int GIN_T1 = 13;         // 16-row tiles of the hidden layer (208 >= 200)
constexpr int GIN_T2 = 7;          // 16-row tiles of the output layer (112 >= 100)

// ---------------------------------------------------------------- aggregation (MP unit)
// a[v] = h[v] + sum over in-edges (ascending source, ties in input order) of relu(h[u] + ecomb[code]).
// ecomb[code] = ((0 + E[a0]) + E[5+a1]) + E[11+a2], precombined on the host in the reference's
// accumulation order, so the per-edge value is bit-identical to message_passing.cc:136-145.
// Work item = (destination row, float4 chunk) in flattened order: every load/store of a row is a
// contiguous 400 B, a wave covers ~2.5 consecutive rows, the 24 KB combo table sits in LDS.
template <int D, bool ADD_SELF>
__global__ __launch_bounds__(256) void gin_aggregate_kernel(const float* __restrict__ h, float* __restrict__ a,
                                                             const int* __restrict__ row_ptr,
                                                             const int* __restrict__ src,
                                                             const uint8_t* __restrict__ ecode,
                                                             const float* __restrict__ ecomb, int n_tot) {
    constexpr int C = D / 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* s_ecomb = reinterpret_cast<float4*>(smem_raw);
    for (int i = threadIdx.x; i < EDGE_COMBOS * C; i += 256) s_ecomb[i] = reinterpret_cast<const float4*>(ecomb)[i];
    __syncthreads();
    const float4* h4 = reinterpret_cast<const float4*>(h);
    const long long total = (long long)n_tot * C;
    // each workgroup walks ONE contiguous span of rows: a row's neighbours (same graph, a few rows away)
    // are then re-read from this CU's L1 / this XCD's L2 instead of being fetched again by another XCD
    long long span = (total + gridDim.x - 1) / gridDim.x;
    span = (span + 255) / 256 * 256;
    const long long i_end = (span * (blockIdx.x + 1) < total) ? span * (blockIdx.x + 1) : total;
    for (long long i = span * blockIdx.x + threadIdx.x; i < i_end; i += 256) {
        const int v = (int)(i / C);
        const int c = (int)(i - (long long)v * C);
        const int beg = row_ptr[v], end = row_ptr[v + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int e = beg;
        for (; e + 1 < end; e += 2) {  // two gathers in flight
            const int u0 = src[e], u1 = src[e + 1];
            const int k0 = ecode[e], k1 = ecode[e + 1];
            const float4 x0 = h4[(size_t)u0 * C + c];
            const float4 x1 = h4[(size_t)u1 * C + c];
            const float4 w0 = s_ecomb[k0 * C + c];
            const float4 w1 = s_ecomb[k1 * C + c];
            acc.x += relu1(w0.x + x0.x); acc.y += relu1(w0.y + x0.y); acc.z += relu1(w0.z + x0.z); acc.w += relu1(w0.w + x0.w);
            acc.x += relu1(w1.x + x1.x); acc.y += relu1(w1.y + x1.y); acc.z += relu1(w1.z + x1.z); acc.w += relu1(w1.w + x1.w);
        }
        if (e < end) {
            const int u0 = src[e];
            const int k0 = ecode[e];
            const float4 x0 = h4[(size_t)u0 * C + c];
            const float4 w0 = s_ecomb[k0 * C + c];
            acc.x += relu1(w0.x + x0.x); acc.y += relu1(w0.y + x0.y); acc.z += relu1(w0.z + x0.z); acc.w += relu1(w0.w + x0.w);
        }
        if (ADD_SELF) {
            const float4 self = h4[i];
            acc.x += self.x; acc.y += self.y; acc.z += self.z; acc.w += self.w;
        }
        reinterpret_cast<float4*>(a)[i] = acc;
    }
}

// ---------------------------------------------------------------- aggregation with LDS-staged row tiles
// Same result as gin_aggregate_kernel.  Persistent workgroups (3 per CU) walk tiles of TR = 64 consecutive
// destination rows.  Per tile: (1) the tile's 25 KiB of h is DMA'd into LDS (global_load_lds_dwordx4, lane-linear
// because the rows are contiguous) together with its slice of row_ptr; (2) the tile's CSR entries -- contiguous in
// the CSR -- are copied into LDS; (3) every (row, float4 chunk) item sums its in-edges in CSR order reading indices,
// neighbour rows and edge-embedding combos from LDS; a neighbour outside the tile (rare: a graph's nodes are
// consecutive rows) is fetched from global memory.
// Measured motivation (PMC on the un-tiled kernel at 2^18 graphs, profiles/r01_g_*): 65.6 M L1->L2 read requests
// for 21.5 M row reads (L1 hit rate ~ 0) and an L2 hit rate of 54 %, i.e. 3.87 GB fetched for 2.78 GB of
// algorithmic reads, behind a three-deep dependent chain row_ptr -> src -> h[u] per item.
constexpr int GIN_TR = 64;

template <int D, bool ADD_SELF, int TR, int NTHR>
__global__ __launch_bounds__(NTHR) void gin_aggregate_tiled_kernel(const float* __restrict__ h, float* __restrict__ a,
                                                                   const int* __restrict__ row_ptr,
                                                                   const int* __restrict__ src,
                                                                   const uint8_t* __restrict__ ecode,
                                                                   const float* __restrict__ ecomb, int n_tot, int n_tiles) {
    constexpr int C = D / 4;
    constexpr int TE = 8 * TR;  // CSR entries of a tile kept in LDS (a molhiv tile of 64 rows has ~141)
    constexpr int NW = NTHR / 64;
    constexpr int TILE_BYTES = TR * D * 4;
    static_assert(TILE_BYTES % 1024 == 0, "tile must be whole 1 KiB DMA pieces");
    constexpr int PIECES = TILE_BYTES / 1024;
    __shared__ __attribute__((aligned(16))) float4 s_ecomb[EDGE_COMBOS * C];
    __shared__ __attribute__((aligned(16))) float4 s_h[TR * C];
    __shared__ int s_rp[TR + 1];
    __shared__ unsigned s_edge[TE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < EDGE_COMBOS * C; i += NTHR) s_ecomb[i] = reinterpret_cast<const float4*>(ecomb)[i];
    const float4* h4 = reinterpret_cast<const float4*>(h);
    auto load_rp = [&](int t) -> int {  // entry threadIdx.x of tile t's row_ptr slice (clamped to the array)
        if (t >= n_tiles || threadIdx.x > TR) return 0;
        const long long i = (long long)t * TR + threadIdx.x;
        return row_ptr[i <= n_tot ? i : n_tot];
    };
    int rp_next = load_rp(blockIdx.x);  // the row_ptr slice of a tile is fetched one tile ahead
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int t0 = tile * TR;
        const int rows = (n_tot - t0) < TR ? (n_tot - t0) : TR;
        __syncthreads();  // previous tile fully consumed (and, first time, the combos are in place)
        if (threadIdx.x <= TR) s_rp[threadIdx.x] = rp_next;
        __syncthreads();
        // one global round trip per tile: rows of h (DMA), the tile's CSR entries, the next tile's row_ptr slice
        const int e0 = s_rp[0];
        const int ne = s_rp[rows] - e0;
        const long long tile_bytes_left = ((long long)n_tot - t0) * D * 4;  // pieces past the last row are skipped
        for (int p = wave; p < PIECES && (long long)p * 1024 < tile_bytes_left; p += NW) {
            const char* g = reinterpret_cast<const char*>(h) + (size_t)t0 * D * 4 + p * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(s_h) + p * 1024), 16, 0, 0);
        }
        rp_next = load_rp(tile + gridDim.x);
        for (int i = threadIdx.x; i < ne && i < TE; i += NTHR) {
            s_edge[i] = (tile_pack_src(src[e0 + i], t0, TR) << 8) | ecode[e0 + i];
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rp_next) : : "memory");
        __syncthreads();
        // stage 3: ordered sums.  The common case (all CSR entries of the tile staged) runs a loop with no clamps
        // and no fall-back branches; LDS byte offsets are carried incrementally (256 = 10 * 25 + 6).
        const char* sh_b = reinterpret_cast<const char*>(s_h);
        const char* se_b = reinterpret_cast<const char*>(s_ecomb);
        int r = threadIdx.x / C, c = threadIdx.x - r * C;
        if (ne <= TE) {
            for (int idx = threadIdx.x; idx < rows * C; idx += NTHR) {
                const int beg = s_rp[r] - e0, end = s_rp[r + 1] - e0;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int e = beg; e < end; e++) {
                    unsigned pk = s_edge[e];
                    asm volatile("" : "+v"(pk));  // keep it a ds_read (see below)
                    const unsigned ul = pk >> 8;
                    const float4 w = *reinterpret_cast<const float4*>(se_b + (pk & 0xFFu) * (D * 4) + c * 16);
                    float4 x = *reinterpret_cast<const float4*>(sh_b + (ul < (unsigned)TR ? ul : 0u) * (D * 4) + c * 16);
                    asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w));
                    if (ul >= (unsigned)TR) x = load_f4_rare(h4 + (size_t)tile_far_row(ul, t0, src, (long long)e0 + e) * C + c);
                    acc.x += relu1(w.x + x.x); acc.y += relu1(w.y + x.y); acc.z += relu1(w.z + x.z); acc.w += relu1(w.w + x.w);
                }
                if (ADD_SELF) {
                    const float4 self = s_h[idx];
                    acc.x += self.x; acc.y += self.y; acc.z += self.z; acc.w += self.w;
                }
                stream_store4(reinterpret_cast<float4*>(a) + (size_t)t0 * C + idx, acc);
                c += NTHR % C;
                r += NTHR / C;
                if (c >= C) { c -= C; r++; }
            }
        } else {
            // dense tiles (kNN graphs, hub nodes): entries beyond the staged ones come from global memory.
            // LDS reads stay unconditional (clamped) and pinned with an empty asm, global memory is touched only in
            // branches: a `cond ? lds : global` select makes hipcc emit flat loads with a full wait after each.
            for (int idx = threadIdx.x; idx < rows * C; idx += NTHR) {
                const int beg = s_rp[r] - e0, end = s_rp[r + 1] - e0;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int e = beg; e < end; e++) {
                    unsigned pk = s_edge[e < TE ? e : TE - 1];
                    asm volatile("" : "+v"(pk));
                    if (e >= TE) {
                        pk = (tile_pack_src(src[e0 + e], t0, TR) << 8) | ecode[e0 + e];
                    }
                    const unsigned ul = pk >> 8;
                    const float4 w = *reinterpret_cast<const float4*>(se_b + (pk & 0xFFu) * (D * 4) + c * 16);
                    float4 x = *reinterpret_cast<const float4*>(sh_b + (ul < (unsigned)TR ? ul : 0u) * (D * 4) + c * 16);
                    asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w));
                    if (ul >= (unsigned)TR) x = load_f4_rare(h4 + (size_t)tile_far_row(ul, t0, src, (long long)e0 + e) * C + c);
                    acc.x += relu1(w.x + x.x); acc.y += relu1(w.y + x.y); acc.z += relu1(w.z + x.z); acc.w += relu1(w.w + x.w);
                }
                if (ADD_SELF) {
                    const float4 self = s_h[idx];
                    acc.x += self.x; acc.y += self.y; acc.z += self.z; acc.w += self.w;
                }
                stream_store4(reinterpret_cast<float4*>(a) + (size_t)t0 * C + idx, acc);
                c += NTHR % C;
                r += NTHR / C;
                if (c >= C) { c -= C; r++; }
            }
        }
    }
}

// ---------------------------------------------------------------- node MLP (NT unit) on fp32 MFMA
// Transposed formulation, one wavefront owns NT tiles of 16 nodes:
//   hid^T[o][node] = b1[o] + sum_k W1[o][k] a[node][k]        A = W1 fragment, B = a^T
//   out^T[d][node] = b2[d] + sum_k W2[d][k] relu(hid^T[k][node])   A = W2 fragment, B = hid^T
// v_mfma_f32_16x16x4_f32: lane l = (i = l & 15, g = l >> 4) supplies A[i][k=g], B[k=g][j=i] and
// receives D[row = 4 g + r][col = i] in register r.  So after MLP1 tile t, lane (j,g) holds hidden
// rows 16 t + 4 g + r of node j: exactly a B operand of MLP2 for the k-step "(t, r)" if the W2
// fragment for that step carries k = 16 t + 4 g + r in slot g.  Same trick on the input side:
// lane (j,g) loads a[j][16 q + 4 g .. +3] as float4 (q < 6) plus a[j][96 + g], and the W1
// fragments are packed with the matching k per slot.  Fragment packing: GinModel::set_weights.
struct GinLayerDev {
    const float* ecomb;    // [60][100]
    const float* w1f;      // [13][6][64][4]   (t, q, lane, r)
    const float* w1tail;   // [13][64]         k = 96 + g
    const float* b1p;      // [208]
    const float* w2f;      // [13][7][64][4]   (t, t2, lane, r)
    const float* b2p;      // [112]
};

template <int NT>
__global__ __launch_bounds__(256) void gin_mlp_kernel(const float* __restrict__ a, float* __restrict__ hout,
                                                       GinLayerDev w, int n_tot, int relu_out) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const long long node_base = (long long)wave * (16 * NT);
    if (node_base >= n_tot) return;

    float bq[NT][25];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        long long node = node_base + nt * 16 + j;
        if (node >= n_tot) node = n_tot - 1;  // clamp: computed, never stored
        const float* row = a + (size_t)node * GIN_D;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 x = *reinterpret_cast<const float4*>(row + 16 * q + 4 * g);
            bq[nt][4 * q + 0] = x.x; bq[nt][4 * q + 1] = x.y; bq[nt][4 * q + 2] = x.z; bq[nt][4 * q + 3] = x.w;
        }
        bq[nt][24] = row[96 + g];
    }

    float4_t acc2[NT][GIN_T2];
#pragma unroll
    for (int t2 = 0; t2 < GIN_T2; t2++) {
        const float4 b = *reinterpret_cast<const float4*>(w.b2p + 16 * t2 + 4 * g);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc2[nt][t2] = (float4_t){b.x, b.y, b.z, b.w};
    }

    const float4* w1f4 = reinterpret_cast<const float4*>(w.w1f);
    const float4* w2f4 = reinterpret_cast<const float4*>(w.w2f);
#pragma unroll 1
    for (int t = 0; t < GIN_T1; t++) {
        float4_t acc1[NT];
        {
            const float4 b = *reinterpret_cast<const float4*>(w.b1p + 16 * t + 4 * g);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc1[nt] = (float4_t){b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 af = w1f4[(size_t)(t * 6 + q) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bq[nt][4 * q + 0], acc1[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bq[nt][4 * q + 1], acc1[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bq[nt][4 * q + 2], acc1[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bq[nt][4 * q + 3], acc1[nt], 0, 0, 0);
        }
        {
            const float at = w.w1tail[t * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, bq[nt][24], acc1[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            acc1[nt].x = relu1(acc1[nt].x); acc1[nt].y = relu1(acc1[nt].y);
            acc1[nt].z = relu1(acc1[nt].z); acc1[nt].w = relu1(acc1[nt].w);
        }
#pragma unroll
        for (int t2 = 0; t2 < GIN_T2; t2++) {
            const float4 af = w2f4[(size_t)(t * GIN_T2 + t2) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc2[nt][t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, acc1[nt].x, acc2[nt][t2], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc2[nt][t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, acc1[nt].y, acc2[nt][t2], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc2[nt][t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, acc1[nt].z, acc2[nt][t2], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc2[nt][t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, acc1[nt].w, acc2[nt][t2], 0, 0, 0);
        }
    }

#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const long long node = node_base + nt * 16 + j;
        if (node >= n_tot) continue;
        float* row = hout + (size_t)node * GIN_D;
#pragma unroll
        for (int t2 = 0; t2 < GIN_T2; t2++) {
            const int col = 16 * t2 + 4 * g;
            if (col < GIN_D) {
                float4_t r = acc2[nt][t2];
                if (relu_out) { r.x = relu1(r.x); r.y = relu1(r.y); r.z = relu1(r.z); r.w = relu1(r.w); }
                *reinterpret_cast<float4*>(row + col) = make_float4(r.x, r.y, r.z, r.w);
            }
        }
    }
}

// ---------------------------------------------------------------- fused layer: MP + NT in one kernel
// One workgroup (4 waves) owns a tile of 64*NT consecutive destination nodes; each wave owns NT MFMA
// column tiles of 16 nodes and runs them end to end:
//   gather   lane (j, g) walks node j's CSR row and accumulates relu(h[u] + ecomb[code]) directly in the
//            register layout of the MLP1 B operand (k = 16 q + 4 g + r, plus k = 96 + g), adds h[v];
//            rows of a graph are neighbours in memory, so the re-reads are L1/L2 hits
//   MLP      as gin_mlp_kernel, but the weight fragments are streamed L2 -> LDS by LDS-DMA
//            (global_load_lds_dwordx4), double buffered, one barrier per step; the message m and the
//            hidden layer never leave registers.  Step c runs MLP1 of hidden tile c (25 dependent MFMAs
//            on one accumulator) interleaved with MLP2 of hidden tile c-1 (28 MFMAs on 7 accumulators),
//            so one wave alone can keep the matrix pipe issuing.
// Weight stream per layer: 14 chunks of 14 KiB (GinModel::set_weights packs them); chunk c holds
//   floats [0,1536) W1 tile c (q, lane, r) | [1536,1600) W1 tail tile c | [1600,3392) W2 tile c-1 (t2, lane, r)
//          [3392,3408) b1 slice tile c | [3408,3520) b2 padded | pad to 3584
// LDS: region A 24000 B = edge-embedding combos during the gather, then weight buffer for odd chunks;
//      region B 14336 B = weight buffer for even chunks.  38336 B per workgroup -> 4 workgroups per CU.
constexpr int GIN_CHUNKS = GIN_T1 + 1;
constexpr int GIN_CHUNK_FLOATS = 3584;
constexpr int GIN_CHUNK_BYTES = GIN_CHUNK_FLOATS * 4;   // 14 pieces of 1 KiB
constexpr int GIN_ECOMB_BYTES = EDGE_COMBOS * GIN_D * 4;  // 24000

__device__ inline void gin_issue_chunk(const float* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int piece = wave + 4 * p;
        if (piece < GIN_CHUNK_BYTES / 1024) {
            const char* g = reinterpret_cast<const char*>(gchunk) + piece * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds_buf + piece * 1024), 16, 0, 0);
        }
    }
}

#define GIN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// One pipeline step c of the node MLP: reads chunk c from `wb`, has already issued chunk c+1 into the
// other buffer.  MLP2 of hidden tile c-1 then MLP1 of hidden tile c; both blocks are written so that
// consecutive MFMAs never depend on each other (7 / 2 accumulators in turn).
template <int NT>
__device__ inline void gin_mlp_step(const float* wb, int c, int lane, int g, const float (&bq)[NT][25],
                                    float4_t (&hid)[NT], float4_t (&acc2)[NT][GIN_T2]) {
    // all fragment reads of this step up front: LDS latency is paid once, under other waves' MFMAs
    float4 a2[GIN_T2], a1[6];
#pragma unroll
    for (int t2 = 0; t2 < GIN_T2; t2++) a2[t2] = *reinterpret_cast<const float4*>(wb + 1600 + (t2 * 64 + lane) * 4);
#pragma unroll
    for (int q = 0; q < 6; q++) a1[q] = *reinterpret_cast<const float4*>(wb + (q * 64 + lane) * 4);
    const float at = wb[1536 + lane];
    const float4 b1v = *reinterpret_cast<const float4*>(wb + 3392 + 4 * g);
    if (c > 0) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
#pragma unroll
            for (int t2 = 0; t2 < GIN_T2; t2++) acc2[nt][t2] = GIN_MFMA(a2[t2].x, hid[nt].x, acc2[nt][t2]);
#pragma unroll
            for (int t2 = 0; t2 < GIN_T2; t2++) acc2[nt][t2] = GIN_MFMA(a2[t2].y, hid[nt].y, acc2[nt][t2]);
#pragma unroll
            for (int t2 = 0; t2 < GIN_T2; t2++) acc2[nt][t2] = GIN_MFMA(a2[t2].z, hid[nt].z, acc2[nt][t2]);
#pragma unroll
            for (int t2 = 0; t2 < GIN_T2; t2++) acc2[nt][t2] = GIN_MFMA(a2[t2].w, hid[nt].w, acc2[nt][t2]);
        }
    }
    if (c < GIN_T1) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            float4_t p0 = (float4_t){b1v.x, b1v.y, b1v.z, b1v.w};
            float4_t p1 = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 6; q++) {
                p0 = GIN_MFMA(a1[q].x, bq[nt][4 * q + 0], p0);
                p1 = GIN_MFMA(a1[q].y, bq[nt][4 * q + 1], p1);
                p0 = GIN_MFMA(a1[q].z, bq[nt][4 * q + 2], p0);
                p1 = GIN_MFMA(a1[q].w, bq[nt][4 * q + 3], p1);
            }
            p0 = GIN_MFMA(at, bq[nt][24], p0);
            hid[nt].x = relu1(p0.x + p1.x); hid[nt].y = relu1(p0.y + p1.y);
            hid[nt].z = relu1(p0.z + p1.z); hid[nt].w = relu1(p0.w + p1.w);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void gin_layer_fused_kernel(const float* __restrict__ h, float* __restrict__ hout,
                                                               const int* __restrict__ row_ptr,
                                                               const int* __restrict__ src,
                                                               const uint8_t* __restrict__ ecode,
                                                               const float* __restrict__ ecomb,
                                                               const float* __restrict__ wchunks, int n_tot,
                                                               int relu_out) {
    // Two DISTINCT LDS objects on purpose: the compiler can then prove that the LDS-DMA into one buffer
    // does not alias the ds_reads of the other and keeps the DMA in flight under the MFMAs (with one
    // object and a runtime-selected half it inserts s_waitcnt vmcnt(0) before the first ds_read).
    __shared__ __attribute__((aligned(16))) float s_a[GIN_ECOMB_BYTES / 4];   // combos, then odd chunks
    __shared__ __attribute__((aligned(16))) float s_b[GIN_CHUNK_FLOATS];      // even chunks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const long long node_base = (long long)blockIdx.x * (64 * NT) + wave * (16 * NT);

    gin_issue_chunk(wchunks, reinterpret_cast<char*>(s_b), wave, lane);  // chunk 0 in flight while we gather
    for (int i = threadIdx.x; i < GIN_ECOMB_BYTES / 16; i += 256)
        reinterpret_cast<float4*>(s_a)[i] = reinterpret_cast<const float4*>(ecomb)[i];
    __syncthreads();
    const float* s_ecomb = s_a;

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
                const float* hr = h + (size_t)u * GIN_D + 4 * g;
                const float* er = s_ecomb + code * GIN_D + 4 * g;
                float4 x[6];
#pragma unroll
                for (int q = 0; q < 6; q++) x[q] = *reinterpret_cast<const float4*>(hr + 16 * q);
                const float xt = h[(size_t)u * GIN_D + 96 + g];
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const float4 w = *reinterpret_cast<const float4*>(er + 16 * q);
                    bq[nt][4 * q + 0] += relu1(w.x + x[q].x);
                    bq[nt][4 * q + 1] += relu1(w.y + x[q].y);
                    bq[nt][4 * q + 2] += relu1(w.z + x[q].z);
                    bq[nt][4 * q + 3] += relu1(w.w + x[q].w);
                }
                bq[nt][24] += relu1(s_ecomb[code * GIN_D + 96 + g] + xt);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {  // + (1 + eps) h[v], eps == 0
        const float* hr = h + (size_t)self_row[nt] * GIN_D + 4 * g;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 x = *reinterpret_cast<const float4*>(hr + 16 * q);
            bq[nt][4 * q + 0] += x.x; bq[nt][4 * q + 1] += x.y; bq[nt][4 * q + 2] += x.z; bq[nt][4 * q + 3] += x.w;
        }
        bq[nt][24] += h[(size_t)self_row[nt] * GIN_D + 96 + g];
    }
    __syncthreads();  // every wave is done with the edge-embedding combos: s_a may be overwritten

    // ---- node MLP (NT unit) on fp32 MFMA, weights streamed through LDS
    float4_t acc2[NT][GIN_T2];
#pragma unroll
    for (int t2 = 0; t2 < GIN_T2; t2++) {
        const float4 b = *reinterpret_cast<const float4*>(s_b + 3408 + 16 * t2 + 4 * g);  // chunk 0 is resident
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc2[nt][t2] = (float4_t){b.x, b.y, b.z, b.w};
    }
    float4_t hid[NT];  // relu(hidden tile c-1), the MLP2 B operand of step c
#pragma unroll
    for (int nt = 0; nt < NT; nt++) hid[nt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < GIN_CHUNKS; c += 2) {
        // even step: compute from s_b while chunk c+1 streams into s_a
        gin_issue_chunk(wchunks + (size_t)(c + 1) * GIN_CHUNK_FLOATS, reinterpret_cast<char*>(s_a), wave, lane);
        gin_mlp_step<NT>(s_b, c, lane, g, bq, hid, acc2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk c+1 have landed
        __syncthreads();                                  // everyone's landed; everyone is done with s_b
        // odd step: compute from s_a while chunk c+2 streams into s_b
        if (c + 2 < GIN_CHUNKS)
            gin_issue_chunk(wchunks + (size_t)(c + 2) * GIN_CHUNK_FLOATS, reinterpret_cast<char*>(s_b), wave, lane);
        gin_mlp_step<NT>(s_a, c + 1, lane, g, bq, hid, acc2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const long long node = node_base + nt * 16 + j;
        if (node >= n_tot) continue;
        float* row = hout + (size_t)node * GIN_D;
#pragma unroll
        for (int t2 = 0; t2 < GIN_T2; t2++) {
            const int col = 16 * t2 + 4 * g;
            if (col < GIN_D) {
                float4_t r = acc2[nt][t2];
                if (relu_out) { r.x = relu1(r.x); r.y = relu1(r.y); r.z = relu1(r.z); r.w = relu1(r.w); }
                *reinterpret_cast<float4*>(row + col) = make_float4(r.x, r.y, r.z, r.w);
            }
        }
    }
}

// ---------------------------------------------------------------- host side: weights + forward
class GinModel : public Model {
public:
    explicit GinModel(bool virtual_node) : virtual_node_(virtual_node) {}
    ~GinModel() override { free_all(); }
    int emb_dim() const override { return GIN_D; }
    int scratch_dim() const override { return GIN_D; }
    int aggregate_dim() const override { return qmode_ ? 0 : GIN_D; }  // fixed-point modes have no float aggregation kernel
    bool has_edge_attr() const override { return true; }
    int num_weight_tensors() const override { return 8; }
    bool weights_ready() const override { return ready_; }

    // host tensors: node_emb[173][100], edge_emb[5][13][100], w1[5][200][100], b1[5][200],
    //               w2[5][100][200], b2[5][100], pred_w[1][100], pred_b[1]
    int set_weights(const float* const* t) override {
        const float *nemb = t[0], *eemb = t[1], *w1 = t[2], *b1 = t[3], *w2 = t[4], *b2 = t[5], *pw = t[6], *pb = t[7];
        std::vector<float> v_nemb(nemb, nemb + ND_FEATURE_TOTAL * GIN_D);
        std::vector<float> v_pw(pw, pw + (size_t)num_tasks_ * GIN_D), v_pb(pb, pb + num_tasks_);  // [NUM_TASK][100], [NUM_TASK]
        std::vector<float> ecomb((size_t)GIN_L * EDGE_COMBOS * GIN_D);
        std::vector<float> w1f((size_t)GIN_L * GIN_T1 * 6 * 64 * 4), w1tail((size_t)GIN_L * GIN_T1 * 64);
        std::vector<float> b1p((size_t)GIN_L * GIN_T1 * 16), b2p((size_t)GIN_L * GIN_T2 * 16);
        std::vector<float> w2f((size_t)GIN_L * GIN_T1 * GIN_T2 * 64 * 4);
        static const int ed_off[3] = {0, 5, 11};  // message_passing.cc:3
        for (int l = 0; l < GIN_L; l++) {
            const float* E = eemb + (size_t)l * ED_FEATURE_PER_LAYER * GIN_D;
            for (int a0 = 0; a0 < 5; a0++)
                for (int a1 = 0; a1 < 6; a1++)
                    for (int a2 = 0; a2 < 2; a2++) {
                        const int code = (a0 * 6 + a1) * 2 + a2;
                        for (int d = 0; d < GIN_D; d++) {
                            float s = 0.0f;  // same order as the reference's edge_embed loop
                            s += E[(ed_off[0] + a0) * GIN_D + d];
                            s += E[(ed_off[1] + a1) * GIN_D + d];
                            s += E[(ed_off[2] + a2) * GIN_D + d];
                            ecomb[((size_t)l * EDGE_COMBOS + code) * GIN_D + d] = s;
                        }
                    }
            const float* W1 = w1 + (size_t)l * GIN_H * GIN_D;
            const float* W2 = w2 + (size_t)l * GIN_D * GIN_H;
            for (int tt = 0; tt < GIN_T1; tt++) {
                for (int lane = 0; lane < 64; lane++) {
                    const int i = lane & 15, g = lane >> 4;
                    const int o = 16 * tt + i;
                    for (int q = 0; q < 6; q++)
                        for (int r = 0; r < 4; r++) {
                            const int k = 16 * q + 4 * g + r;
                            w1f[((((size_t)l * GIN_T1 + tt) * 6 + q) * 64 + lane) * 4 + r] = (o < GIN_H) ? W1[o * GIN_D + k] : 0.0f;
                        }
                    w1tail[((size_t)l * GIN_T1 + tt) * 64 + lane] = (o < GIN_H) ? W1[o * GIN_D + 96 + g] : 0.0f;
                    for (int t2 = 0; t2 < GIN_T2; t2++) {
                        const int d = 16 * t2 + i;
                        for (int r = 0; r < 4; r++) {
                            const int k = 16 * tt + 4 * g + r;
                            w2f[((((size_t)l * GIN_T1 + tt) * GIN_T2 + t2) * 64 + lane) * 4 + r] =
                                (d < GIN_D && k < GIN_H) ? W2[d * GIN_H + k] : 0.0f;
                        }
                    }
                }
                for (int x = 0; x < 16; x++) {
                    const int o = 16 * tt + x;
                    b1p[((size_t)l * GIN_T1 + tt) * 16 + x] = (o < GIN_H) ? b1[l * GIN_H + o] : 0.0f;
                }
            }
            for (int x = 0; x < GIN_T2 * 16; x++) b2p[(size_t)l * GIN_T2 * 16 + x] = (x < GIN_D) ? b2[l * GIN_D + x] : 0.0f;
        }
        // weight stream of the fused layer kernel: per layer 14 chunks of 14 KiB; chunk c = W1 of hidden
        // tile c (absent for c = 13) + W2 columns of hidden tile c-1 (absent for c = 0) + biases
        std::vector<float> chunks((size_t)GIN_L * GIN_CHUNKS * GIN_CHUNK_FLOATS, 0.0f);
        for (int l = 0; l < GIN_L; l++)
            for (int c = 0; c < GIN_CHUNKS; c++) {
                float* ck = &chunks[((size_t)l * GIN_CHUNKS + c) * GIN_CHUNK_FLOATS];
                if (c < GIN_T1) {
                    memcpy(ck, &w1f[(((size_t)l * GIN_T1 + c) * 6) * 64 * 4], sizeof(float) * 6 * 64 * 4);
                    memcpy(ck + 1536, &w1tail[((size_t)l * GIN_T1 + c) * 64], sizeof(float) * 64);
                    memcpy(ck + 3392, &b1p[((size_t)l * GIN_T1 + c) * 16], sizeof(float) * 16);
                }
                if (c >= 1)
                    memcpy(ck + 1600, &w2f[(((size_t)l * GIN_T1 + (c - 1)) * GIN_T2) * 64 * 4], sizeof(float) * GIN_T2 * 64 * 4);
                memcpy(ck + 3408, &b2p[(size_t)l * GIN_T2 * 16], sizeof(float) * GIN_T2 * 16);
            }
        // weight stream of the split-f16 layer kernel (gin_split.hip)
        std::vector<uint8_t> split((size_t)GIN_L * GS_LAYER_BYTES);
        for (int l = 0; l < GIN_L; l++)
            gin_split_pack_layer(w1 + (size_t)l * GIN_H * GIN_D, b1 + (size_t)l * GIN_H, w2 + (size_t)l * GIN_D * GIN_H,
                                 b2 + (size_t)l * GIN_D, split.data() + (size_t)l * GS_LAYER_BYTES);
        // ... and of the graph-resident kernel (its own chunk format)
        std::vector<uint8_t> rsplit((size_t)GIN_L * gin_resident_layer_bytes());
        for (int l = 0; l < GIN_L; l++)
            gin_resident_pack_layer(w1 + (size_t)l * GIN_H * GIN_D, b1 + (size_t)l * GIN_H, w2 + (size_t)l * GIN_D * GIN_H,
                                    b2 + (size_t)l * GIN_D, rsplit.data() + (size_t)l * gin_resident_layer_bytes());
#ifdef FLOWGNN_DEV
        // ... re-cut into the ping-pong kernel's pieces, with the edge-embedding tables as half tables (development builds only)
        std::vector<uint8_t> pp_pieces((size_t)GIN_L * gin_pp_layer_bytes());
        {
            std::vector<uint8_t> eight(gin_resident_layer_bytes());  // the eight-chunk form of the stream (chunk 7 = the packed K-step)
            for (int l = 0; l < GIN_L; l++) {
                gin_resident_pack_layer(w1 + (size_t)l * GIN_H * GIN_D, b1 + (size_t)l * GIN_H, w2 + (size_t)l * GIN_D * GIN_H,
                                        b2 + (size_t)l * GIN_D, eight.data(), false);
                gin_pp_pack_layer(eight.data(), pp_pieces.data() + (size_t)l * gin_pp_layer_bytes());
            }
        }
        std::vector<float> pp_tables(gin_pp_table_floats());
        gin_pp_pack_tables(ecomb.data(), pp_tables.data());
        if (int rc0 = upload(&d_pp_pieces_, pp_pieces)) return rc0;
        if (int rc0 = upload(&d_pp_tables_, pp_tables)) return rc0;
#endif
        int rc;
        if ((rc = ginq_upload(qw_, nemb, eemb, w1, b1, w2, b2, pw, pb))) return rc;  // Q6.10 copies (numeric mode 1)
        if ((rc = upload(&d_split_, split))) return rc;
        if ((rc = upload(&d_rsplit_, rsplit))) return rc;
        {   // the single-task readout folded through the last layer's second linear layer (gin_resident_kernel, gr_layer)
            std::vector<float> head(GIN_RESIDENT_HEAD_FLOATS);
            const int l = GIN_L - 1;
            gin_resident_head_fold(w1 + (size_t)l * GIN_H * GIN_D, w2 + (size_t)l * GIN_D * GIN_H, b2 + (size_t)l * GIN_D, pw, head.data());
            if ((rc = upload(&d_head_, head))) return rc;
        }
        {   // pre-combined encoder table of the one-pass tile loader (gin_tile_build_kernel + the resident kernel's ENC form)
            std::vector<float> etab(gin_resident_enc_table_floats());
            gin_resident_pack_enc_table(nemb, etab.data());
            if ((rc = upload(&d_enc_tab_, etab))) return rc;
        }
        if ((rc = upload(&d_chunks_, chunks))) return rc;
        if ((rc = upload(&d_nemb_, v_nemb))) return rc;
        if ((rc = upload(&d_pw_, v_pw))) return rc;
        if ((rc = upload(&d_pb_, v_pb))) return rc;
        if ((rc = upload(&d_ecomb_, ecomb))) return rc;
        {   // the resident kernel's copy of the edge-embedding combos, scaled by 2^-16: its walk forms relu(x + e) in that scaled domain
            // with one clamped packed FMA per two values (gin_split.hip, GR_MSG2).  Exact while |e| < 4 096 (x + e < 2^16 needs
            // x < 6e4, the range every operand is checked against anyway); tables beyond that take the per-layer kernels.
            float emax = 0.0f;
            std::vector<float> sc(ecomb.size() + (size_t)GIN_L * 6 * 4 * 64 * 4, 0.0f);
            for (size_t i = 0; i < ecomb.size(); i++) { sc[i] = ecomb[i] * (1.0f / 65536.0f); const float ae = std::fabs(ecomb[i]); emax = (ae > emax || ae != ae) ? ae : emax; }  // (not fmax: it would drop a NaN)
            // ... and once more in plane order [layer][quad q][quarter g][code, padded to 64] of 16 B, for the walk's reads through L1
            for (int l = 0; l < GIN_L; l++)
                for (int q = 0; q < 6; q++)
                    for (int gq = 0; gq < 4; gq++)
                        for (int c = 0; c < EDGE_COMBOS; c++)
                            for (int k = 0; k < 4; k++)
                                sc[ecomb.size() + ((((size_t)l * 6 + q) * 4 + gq) * 64 + c) * 4 + k] = sc[((size_t)l * EDGE_COMBOS + c) * GIN_D + 16 * q + 4 * gq + k];
            table_ok_ = emax < 4096.0f;  // (a NaN / Inf entry: emax is then NaN / Inf and the comparison false -> the per-layer kernels)
            if ((rc = upload(&d_ecomb_res_, sc))) return rc;
        }
        if ((rc = upload(&d_w1f_, w1f))) return rc;
        if ((rc = upload(&d_w1tail_, w1tail))) return rc;
        if ((rc = upload(&d_b1p_, b1p))) return rc;
        if ((rc = upload(&d_w2f_, w2f))) return rc;
        if ((rc = upload(&d_b2p_, b2p))) return rc;
        ready_ = true;
        return 0;
    }

    // GIN/src/host_load.cc:24-58 (file names, element counts, raw LE float32, no header)
    int load_weights_dir(const char* dir) override {
        std::vector<float> w1((size_t)GIN_L * GIN_H * GIN_D), b1((size_t)GIN_L * GIN_H), w2((size_t)GIN_L * GIN_D * GIN_H),
            b2((size_t)GIN_L * GIN_D), nemb((size_t)ND_FEATURE_TOTAL * GIN_D), eemb((size_t)GIN_L * ED_FEATURE_PER_LAYER * GIN_D),
            pw((size_t)num_tasks_ * GIN_D), pb(num_tasks_);
        int rc;
        if ((rc = read_floats(dir, "gin_ep1_mlp_1_weights_dim100.bin", 0, w1.size(), w1.data()))) return rc;
        if ((rc = read_floats(dir, "gin_ep1_mlp_1_bias_dim100.bin", 0, b1.size(), b1.data()))) return rc;
        if ((rc = read_floats(dir, "gin_ep1_mlp_2_weights_dim100.bin", 0, w2.size(), w2.data()))) return rc;
        if ((rc = read_floats(dir, "gin_ep1_mlp_2_bias_dim100.bin", 0, b2.size(), b2.data()))) return rc;
        // gin_ep1_eps_dim100.bin is read by the reference host and never used (host.cc:185-200)
        if ((rc = read_floats(dir, "gin_ep1_nd_embed_dim100.bin", 0, nemb.size(), nemb.data()))) return rc;
        if ((rc = read_floats(dir, "gin_ep1_ed_embed_dim100.bin", 0, eemb.size(), eemb.data()))) return rc;
        if ((rc = read_floats(dir, "gin_ep1_pred_weights_dim100.bin", 0, pw.size(), pw.data()))) return rc;
        if ((rc = read_floats(dir, "gin_ep1_pred_bias_dim100.bin", 0, pb.size(), pb.data()))) return rc;
        const float* t[8] = {nemb.data(), eemb.data(), w1.data(), b1.data(), w2.data(), b2.data(), pw.data(), pb.data()};
        return set_weights(t);
    }

    GinLayerDev layer_dev(int l) const {
        GinLayerDev w;
        w.ecomb = d_ecomb_ + (size_t)l * EDGE_COMBOS * GIN_D;
        w.w1f = d_w1f_ + (size_t)l * GIN_T1 * 6 * 64 * 4;
        w.w1tail = d_w1tail_ + (size_t)l * GIN_T1 * 64;
        w.b1p = d_b1p_ + (size_t)l * GIN_T1 * 16;
        w.w2f = d_w2f_ + (size_t)l * GIN_T1 * GIN_T2 * 64 * 4;
        w.b2p = d_b2p_ + (size_t)l * GIN_T2 * 16;
        return w;
    }

    void launch_aggregate(const DeviceBatch& db, int l, const float* hin, float* a, hipStream_t s) {
        const long long items = (long long)db.b.n_tot * GIN_C;
        const int grid = grid_for(items, 256, 256 * 6);
        const size_t lds = sizeof(float) * EDGE_COMBOS * GIN_D;
        // gin_agg_untiled=1 selects the first (un-tiled) kernel for A/B measurements
        if (!agg_untiled_) {
            const int n_tiles = (int)ceil_div_ll(db.b.n_tot, GIN_TR);
            int g2 = 256 * 3;  // persistent: three workgroups per CU (52 KB of LDS each)
            if (g2 > n_tiles) g2 = n_tiles;
            // tile = 128 rows per 512-thread workgroup, 2 workgroups per CU (78 KB of LDS each): 1.17 ms at 2^18
            // molhiv graphs vs 1.26-1.30 ms for 64 rows x 256 threads x 3 per CU (gin_agg_tile=64 / 256)
            const int tv = agg_tile_;
            if (tv == 128 || tv == 256) {
                const int nt2 = (int)ceil_div_ll(db.b.n_tot, tv);
                int g4 = tv == 128 ? 256 * 2 : 256;
                if (g4 > nt2) g4 = nt2;
                if (tv == 128)
                    gin_aggregate_tiled_kernel<GIN_D, true, 128, 512><<<g4, 512, 0, s>>>(hin, a, db.csr.row_ptr, db.csr.src, db.csr.ecode,
                                                                                         layer_dev(l).ecomb, db.b.n_tot, nt2);
                else
                    gin_aggregate_tiled_kernel<GIN_D, true, 256, 1024><<<g4, 1024, 0, s>>>(hin, a, db.csr.row_ptr, db.csr.src, db.csr.ecode,
                                                                                           layer_dev(l).ecomb, db.b.n_tot, nt2);
                return;
            }
            gin_aggregate_tiled_kernel<GIN_D, true, GIN_TR, 256><<<g2, 256, 0, s>>>(hin, a, db.csr.row_ptr, db.csr.src, db.csr.ecode,
                                                                                    layer_dev(l).ecomb, db.b.n_tot, n_tiles);
            return;
        }
        gin_aggregate_kernel<GIN_D, true><<<grid, 256, lds, s>>>(hin, a, db.csr.row_ptr, db.csr.src, db.csr.ecode,
                                                                 layer_dev(l).ecomb, db.b.n_tot);
    }

    // graph-resident path (gin_split.hip, gin_resident_kernel): whole graphs packed into 256-row tiles by flowgnn_set_batch
    void graph_tile_limits(int& rows, int& edges) const override {
        rows = resident_ ? GIN_RESIDENT_ROWS : 0;
        edges = resident_ ? GIN_RESIDENT_EDGES : 0;
    }
#ifdef FLOWGNN_DEV  // the ping-pong form (dev/gin_pp_device.inc: bit-identical, measured slower) exists in development builds only
    void sub_tile_limits(int& rows, int& edges) const override {
        const bool on = resident_ && pingpong_ && !virtual_node_;
        rows = on ? GIN_PP_ROWS : 0;
        edges = on ? GIN_PP_EDGES : 0;
    }
    // the batch's half-tiles on gin_pp_kernel, the few graphs beyond the half-tile limits on gin_resident_kernel.  (Decided from the
    // SHARD's own half-tile fill: unlike the shipped kernels' choices this one does not follow the job -- development only.)
    bool use_pingpong(const DeviceBatch& db) const {
        return pingpong_ && !virtual_node_ && use_resident(db) && !keep_h_ && num_tasks_ == 1 && fold_readout_ && head_fold_ && db.gtiles.sub_ok &&
               db.gtiles.n_sub > 0 && db.gtiles.sub_fill >= resident_min_fill_;
    }
#endif
    bool use_resident(const DeviceBatch& db) const {
        // tiles that are mostly empty (graphs of 130..256 nodes, or dense graphs that hit the edge limit first) waste the
        // MFMA columns of the absent rows: below half full the per-layer kernels are the better choice
        return resident_ && table_ok_ && fused_ && split_ && !exact_ && db.gtiles.ok && db.gtiles.n_tiles > 0 && db.gtiles.fill >= resident_min_fill_;
    }

    // One-pass form of the graph-resident path (default): gin_tile_build_kernel turns the caller's edge list / attributes / node
    // features of each tile into the tile descriptor + four table-row numbers per node, and the resident kernel's tile loader
    // computes h_0 itself -- no CSR, no h_0 rows and no separate encoder / index-build launches in HBM.  Needs the folded
    // single-task readout (the loader rides on the folded last layer's steps); gin_tile_build = 0 restores the three-kernel front end.
    bool one_pass(const DeviceBatch& db) const {
        // Measured (NOTEBOOK.md section 4): the tile build costs what index build + tile prep cost (0.27 ms at 2^18 molhiv graphs) and the
        // encoder inside the folded last layer's steps costs the resident kernel 0.32 ms (0.57 ms before the kernel lost its scratch
        // reloads, which made this form the slower one on large batches in round 3) where the separate, store-bound encoder launch costs
        // 0.51: ahead at every size now -- 8.49 vs 8.73 ms per step at 2^18 graphs, 1.23 vs 1.24 at 32 768, 0.206 vs 0.222 at 4 113.
        // -1 = default = on (development builds: an explicit gin_pingpong keeps the three-kernel front end -- the ping-pong kernel has no
        // encoder in its loader and is never used with a virtual node).
        const bool want = tile_build_ < 0 ? !(pingpong_ && !virtual_node_) : tile_build_ != 0;
        return want && use_resident(db) && !qmode_ && !keep_h_ && num_tasks_ == 1 && fold_readout_ && head_fold_ && db.b.edge_attr != nullptr;
    }
    bool needs_csr(const DeviceBatch& db) const override { return !one_pass(db); }
    // (asked at flowgnn_set_batch, before the batch is known: the lists are built whenever the one-pass path could take them)
    bool wants_packed_tile_lists() const override {
        return binpack_ && resident_ && !qmode_ && num_tasks_ == 1 && fold_readout_ && head_fold_ && tile_build_ != 0 && !pingpong_;
    }

    int forward(DeviceBatch& db, Profiler& prof, hipStream_t s) override {
        const int n = db.b.n_tot;
        if (n <= 0) return 0;
        if (qmode_) return ginq_forward(qw_, db, prof, s);
        if (one_pass(db)) {
            // bin-packed tile lists when flowgnn_set_batch made them (option gin_binpack): fewer, fuller tiles of the same graphs -- everything
            // the resident kernel reads is written by the tile build in tile order, and a row's sums depend on the row alone: the same bits
            const bool bp = binpack_ && db.gtiles.bp_tiles > 0;
            const int* t_row = bp ? db.gtiles.bp_row : db.gtiles.row_start;
            const int* t_graph = bp ? db.gtiles.bp_graph : db.gtiles.graph_start;
            const int n_tiles = bp ? db.gtiles.bp_tiles : db.gtiles.n_tiles;
            if (int rc = perm_.reserve((size_t)n_tiles * (GIN_RESIDENT_DESC_BYTES / 4))) return rc;
            if (int rc = enc_idx_.reserve((size_t)n)) return rc;
            GinTileBuild tb{db.b, enc_idx_.p, d_enc_tab_, db.csr.err, bp ? db.gtiles.bp_list : nullptr, bp ? db.gtiles.bp_lrow : nullptr};
            {
                ProfScope p(prof, "gin_tile_build", s);
                launch_gin_tile_build(tb, t_row, t_graph, reinterpret_cast<uint8_t*>(perm_.p), n_tiles, virtual_node_, resident_order_, s);
            }
            ProfScope p(prof, "gin_resident", s);  // the whole model
            launch_gin_resident(nullptr, nullptr, nullptr, nullptr, nullptr, d_ecomb_res_, d_rsplit_, d_pw_, d_pb_, t_row, t_graph,
                                reinterpret_cast<uint8_t*>(perm_.p), db.b.node_off, db.out, n_tiles,
                                db.range_flag, s, virtual_node_, d_head_, resident_order_, resident_prof_, &tb);
            db.final_h = 0;
            db.h_valid = false;
            h0_in_hbm_ = false;  // the tile loader computed h_0 on chip
            return 0;
        }
        h0_in_hbm_ = true;
        {
            ProfScope p(prof, "atom_encoder", s);
            atom_encoder_kernel<GIN_D><<<atom_encoder_grid(n, GIN_C), 512, 0, s>>>(db.b.node_feature, d_nemb_, db.h[0], n, db.csr.err);
        }
        const bool multi = num_tasks_ > 1;  // NUM_TASK > 1: the layers leave h_5 in HBM and a multi-task readout kernel follows
#ifdef FLOWGNN_DEV
        if (use_pingpong(db)) {
            const GraphTiles& gt = db.gtiles;
            const size_t sub_words = ((size_t)gt.n_sub * gin_pp_desc_bytes() + 3) / 4, big_words = (size_t)gt.n_big * (GIN_RESIDENT_DESC_BYTES / 4);
            if (int rc = perm_.reserve(sub_words + big_words)) return rc;
            {
                ProfScope p(prof, "gin_resident", s);  // all five layers + readout of every half-tile
                launch_gin_pp(db.h[0], db.csr.row_ptr, db.csr.src, db.csr.ecode, d_pp_tables_, d_pp_pieces_, d_pb_, gt.sub,
                              reinterpret_cast<uint8_t*>(perm_.p), db.b.node_off, db.out, gt.n_sub, db.range_flag, d_head_, s, resident_prof_,
                              pingpong_waves_);
            }
            if (gt.n_big > 0) {  // graphs of 129..256 nodes (or 641..1280 edges): one full tile each on the eight-wave resident kernel
                ProfScope p(prof, "gin_resident_big", s);
                launch_gin_resident(db.h[0], nullptr, db.csr.row_ptr, db.csr.src, db.csr.ecode, d_ecomb_res_, d_rsplit_, d_pw_, d_pb_, gt.big_row,
                                    gt.big_graph, reinterpret_cast<uint8_t*>(perm_.p + sub_words), db.b.node_off, db.out, gt.n_big,
                                    db.range_flag, s, false, d_head_, resident_order_, false, nullptr, 2);
            }
            db.final_h = 0;
            db.h_valid = false;
            return 0;
        }
#endif
        if (use_resident(db)) {
            // all five layers and the readout in one launch; h_5 rows are written (to h[1]) only for the flowgnn_get_h tap
            if (int rc = perm_.reserve((size_t)db.gtiles.n_tiles * (GIN_RESIDENT_DESC_BYTES / 4))) return rc;
            const bool rows = keep_h_ || multi;
            {
                ProfScope p(prof, "gin_resident", s);
                launch_gin_resident(db.h[0], rows ? db.h[1] : nullptr, db.csr.row_ptr, db.csr.src, db.csr.ecode, d_ecomb_res_, d_rsplit_, d_pw_, d_pb_,
                                    db.gtiles.row_start, db.gtiles.graph_start, reinterpret_cast<uint8_t*>(perm_.p), db.b.node_off,
                                    multi ? nullptr : db.out, db.gtiles.n_tiles, db.range_flag, s, virtual_node_,
                                    (!rows && fold_readout_ && head_fold_) ? d_head_ : nullptr, resident_order_, resident_prof_);
            }
            db.final_h = rows ? 1 : 0;
            db.h_valid = rows;
            if (multi) launch_readout_mt(db, db.h[1], prof, s);
            return 0;
        }
        int cur = 0;
        bool folded = false;
        for (int l = 0; l < GIN_L; l++) {
            if (fused_ && split_ && !exact_) {
                ProfScope p(prof, "gin_layer_fused", s);
                // last layer: the readout's per-node dot product h'[v] . w_pred is taken in the epilogue and only that
                // leaves the kernel (db.scratch as float[n]); the rows are written only for the flowgnn_get_h tap
                const bool fold = l == GIN_L - 1 && fold_readout_ && !keep_h_ && !multi;
                launch_gin_layer_split(db.h[cur], fold ? db.scratch : db.h[cur ^ 1], db.csr.row_ptr, db.csr.src, db.csr.ecode,
                                       layer_dev(l).ecomb, d_split_ + (size_t)l * GS_LAYER_BYTES, n, db.b.e_tot, l != GIN_L - 1,
                                       db.range_flag, split_nt_, s, fold ? d_pw_ : nullptr);
                if (fold) {
                    folded = true;
                    break;
                }
                cur ^= 1;
                continue;
            }
            if (fused_) {
                ProfScope p(prof, "gin_layer_fused", s);
                constexpr int NT = 1;
                const int blocks = (int)ceil_div_ll(n, 64 * NT);
                gin_layer_fused_kernel<NT><<<blocks, 256, 0, s>>>(
                    db.h[cur], db.h[cur ^ 1], db.csr.row_ptr, db.csr.src, db.csr.ecode, layer_dev(l).ecomb,
                    d_chunks_ + (size_t)l * GIN_CHUNKS * GIN_CHUNK_FLOATS, n, l != GIN_L - 1);
                cur ^= 1;
                continue;
            }
            {
                ProfScope p(prof, "gin_aggregate", s);
                launch_aggregate(db, l, db.h[cur], db.scratch, s);
            }
            {
                ProfScope p(prof, "gin_mlp", s);
                constexpr int NT = 2;
                const int waves = (int)ceil_div_ll(n, 16 * NT);
                gin_mlp_kernel<NT><<<(waves + 3) / 4, 256, 0, s>>>(db.scratch, db.h[cur ^ 1], layer_dev(l), n,
                                                                     l != GIN_L - 1);
            }
            cur ^= 1;
        }
        db.final_h = cur;
        db.h_valid = !folded;
        if (multi) {
            launch_readout_mt(db, db.h[cur], prof, s);
            return 0;
        }
        {
            ProfScope p(prof, "mean_pool_linear", s);
            if (folded)
                segment_mean_bias_kernel<0><<<(db.b.num_graphs + 255) / 256, 256, 0, s>>>(db.scratch, db.b.node_off, d_pb_, db.out,
                                                                                       db.b.num_graphs);
            else
                mean_pool_linear_kernel<GIN_D><<<(db.b.num_graphs + 3) / 4, 256, 0, s>>>(db.h[cur], db.b.node_off, d_pw_, d_pb_,
                                                                                         db.out, db.b.num_graphs);
        }
        return 0;
    }

    void launch_readout_mt(DeviceBatch& db, const float* h, Profiler& prof, hipStream_t s) {
        ProfScope p(prof, "mean_pool_linear", s);
        const int blocks = (db.b.num_graphs + 3) / 4;
        mean_pool_linear_mt_kernel<GIN_D><<<blocks < 512 ? blocks : 512, 256, 0, s>>>(h, db.b.node_off, d_pw_, d_pb_, db.out, db.b.num_graphs,
                                                                                      num_tasks_);
    }

    void configure(const Options& o) override {
        fused_ = !o.on("gin_unfused");
        split_ = o.i("gin_mfma") != 32;
        split_nt_ = o.i("gin_split_nt");
        agg_untiled_ = o.on("gin_agg_untiled");
        agg_tile_ = o.i("gin_agg_tile");
        resident_order_ = o.i("gin_resident_nosort");
        binpack_ = o.on("gin_binpack");
        resident_prof_ = o.on("gin_resident_prof");
        fold_readout_ = o.on("gin_fold_readout");
        resident_ = o.on("gin_resident");
        resident_min_fill_ = o.num("gin_resident_min_fill");
        tile_build_ = o.i("gin_tile_build");
#ifdef FLOWGNN_DEV
        pingpong_ = o.on("gin_pingpong");
        pingpong_waves_ = o.i("gin_pingpong") == 2 ? 16 : 8;  // 2: the sixteen-wave form (eight waves per half, one column tile each)
#endif
        head_fold_ = o.on("gin_head_fold");
    }
    void set_exact(bool on) override { exact_ = on; }
    void set_keep_h(bool on) override { keep_h_ = on; }
    int set_numeric_mode(int mode) override {
        if (mode != 0 && mode != 1) return 8;
        if (mode == 1 && num_tasks_ != 1) return 8;  // the Q6.10 readout is single-task (as the reference's)
        qmode_ = mode == 1;
        return 0;
    }
    int set_num_tasks(int t) override {
        if (t < 1 || (t != 1 && qmode_)) return 8;
        if (t != num_tasks_) ready_ = false;  // graph_pred_weights / bias change shape: set the weights again
        num_tasks_ = t;
        return 0;
    }

    int aggregation_only(DeviceBatch& db, int layer, hipStream_t s) override {
        if (qmode_) return 8;  // FLOWGNN_ERR_UNSUPPORTED: the fixed-point forward never builds the float kernels' inputs (tiles, h rows)
        if (layer < 0 || layer >= GIN_L) return 1;
        if (!h0_in_hbm_) {  // the last run was the one-pass resident path: the probe's input rows (h_0) were never written to HBM
            atom_encoder_kernel<GIN_D><<<atom_encoder_grid(db.b.n_tot, GIN_C), 512, 0, s>>>(db.b.node_feature, d_nemb_, db.h[0], db.b.n_tot, db.csr.err);
            h0_in_hbm_ = true;
        }
        launch_aggregate(db, layer, db.h[db.final_h], db.scratch, s);
        return 0;
    }

private:
    void free_all() {
        float** ptrs[] = {&d_chunks_, &d_nemb_, &d_pw_, &d_pb_, &d_ecomb_, &d_ecomb_res_, &d_w1f_, &d_w1tail_, &d_b1p_, &d_w2f_, &d_b2p_};
        for (auto p : ptrs)
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        if (d_split_) { (void)hipFree(d_split_); d_split_ = nullptr; }
        if (d_rsplit_) { (void)hipFree(d_rsplit_); d_rsplit_ = nullptr; }
        if (d_head_) { (void)hipFree(d_head_); d_head_ = nullptr; }
        if (d_enc_tab_) { (void)hipFree(d_enc_tab_); d_enc_tab_ = nullptr; }
        if (d_pp_pieces_) { (void)hipFree(d_pp_pieces_); d_pp_pieces_ = nullptr; }
        if (d_pp_tables_) { (void)hipFree(d_pp_tables_); d_pp_tables_ = nullptr; }
        enc_idx_.release();
        perm_.release();
        qw_.release();
    }
    bool ready_ = false;
    const bool virtual_node_;  // FLOWGNN_MODEL_GIN_VN: the batch carries one virtual node per graph
    // gin_unfused=1 keeps the two-kernel layer (aggregate + mlp) for A/B measurements
    bool fused_ = true;
    // gin_mfma=32 keeps the dense update on the fp32 matrix pipe (gin_layer_fused_kernel); the default (16) runs it
    // as three f16 MFMAs per product (gin_split.hip), with the engine falling back to fp32 when the range flag trips
    bool split_ = true;
    // 4 = eight-wave workgroups of 128 nodes (default), 1 / 2 = four waves x 1 / 2 node tiles
    int split_nt_ = 4;
    bool agg_untiled_ = false;  // gin_agg_untiled=1: the first (un-tiled) aggregation kernel, A/B measurements
    int agg_tile_ = 128;        // gin_agg_tile: 64 | 128 | 256 rows per tile of the stand-alone aggregation kernel
    bool binpack_ = true;       // gin_binpack: the one-pass resident path walks bin-packed tile lists (GraphTiles::bp_*)
    int resident_order_ = 0;    // gin_resident_nosort: column order of the resident kernel's tiles (0 degree-sorted, 1 natural, 2 bank-aware)
    bool resident_prof_ = false;  // gin_resident_prof: phase stamps printed per launch (synchronises)
    bool exact_ = false;
    bool keep_h_ = false;
    bool qmode_ = false;  // flowgnn_set_numeric_mode(FLOWGNN_NUMERIC_Q6_10)
    int num_tasks_ = 1;   // NUM_TASK (GIN/src/dcl.h:25) as a run-time dimension
    GinQWeights qw_;
    GrowBufI perm_;  // graph-resident path: per-tile descriptors (gin_tile_prep_kernel / gin_tile_build_kernel)
    GrowBufI enc_idx_;  // one-pass path: three table-row numbers per node in one word, written by gin_tile_build_kernel
    float* d_enc_tab_ = nullptr;  // ... and the pre-combined encoder table they index
    bool h0_in_hbm_ = false;      // db.h[0] holds h_0 of the resident batch (false after a one-pass run)
    int pingpong_waves_ = 8;
    bool pingpong_ = false;       // development builds, gin_pingpong = 1: gin_pp_kernel (two half-tiles per CU half a layer out of phase; measured slower)
    uint8_t* d_pp_pieces_ = nullptr;  // weight pieces of gin_pp_kernel
    float* d_pp_tables_ = nullptr;    // ... and its half tables
    int tile_build_ = -1;         // gin_tile_build: 1 = one-pass front end, 0 = CSR build + atom encoder + tile prep as separate launches, -1 = default = one-pass unless gin_pingpong selects the ping-pong kernel (no size rule)
    // gin_fold_readout=0 keeps the separate mean-pool + linear kernel (and the last layer's 2.7 GB of rows)
    bool fold_readout_ = true;
    // gin_resident=0 keeps one launch per layer (gin_layer_split_kernel).  GIN-VN runs the HUBS form of the resident kernel:
    // its virtual nodes are hub rows (in-degree = graph size), walked by the 16 lanes of their column tile together
    bool resident_ = true;
    double resident_min_fill_ = 0.5;
    uint8_t* d_split_ = nullptr;
    uint8_t* d_rsplit_ = nullptr;  // weight stream of the graph-resident kernel
    float* d_head_ = nullptr;      // gin_resident_head_fold
    bool head_fold_ = true;  // gin_head_fold=0: the last layer's second linear layer is computed (readout not folded through it)
    float* d_chunks_ = nullptr;
    float* d_ecomb_res_ = nullptr;  // ecomb * 2^-16: the resident kernel's walk (GR_MSG2, gin_split.hip)
    bool table_ok_ = true;          // |ecomb| < 4 096: the scaled walk is exact (set_weights)
    float *d_nemb_ = nullptr, *d_pw_ = nullptr, *d_pb_ = nullptr, *d_ecomb_ = nullptr, *d_w1f_ = nullptr,
          *d_w1tail_ = nullptr, *d_b1p_ = nullptr, *d_w2f_ = nullptr, *d_b2p_ = nullptr;
};

Model* make_gin_model(bool virtual_node) { return new GinModel(virtual_node); }

}  // namespace fg
