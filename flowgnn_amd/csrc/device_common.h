// device_common.h -- device code shared by the per-model translation units (included, not linked:
// every kernel here is a template or static).
#pragma once
#include "common.h"

namespace fg {

// reference tables
static __constant__ int c_nd_off[ND_FEATURE] = {0, 119, 123, 135, 147, 157, 163, 169, 171};  // load_inputs.cc:5
static __constant__ int c_nd_card[ND_FEATURE] = {119, 4, 12, 12, 10, 6, 6, 2, 2};            // host_load.cc:5

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// A global load that hipcc's wait-count tracking does not see: issued and waited for inside one asm statement.
// For RARE fall-back paths inside hot loops -- an ordinary load in a branch makes the compiler put an unconditional
// `s_waitcnt vmcnt(0)` at the merge point, which then waits for every outstanding store and prefetch on every trip.
__device__ inline float4 load_f4_rare(const float4* p) {
    float4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline float load_f32_rare(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// CSR word of the tiled kernels: (source row | edge code).  A source inside the tile is its row (< TR <= 2^15); one outside
// is bit 23 + its 23-bit signed distance from the tile start, so the rare path can form its address without first
// re-reading src[] (one dependent round trip instead of two); a distance beyond +-2^22 rows is the escape value.
constexpr unsigned TILE_FAR = 0x800000u, TILE_ESCAPE = 0xC00000u;  // escape = bit 23 | distance -2^22
__device__ __forceinline__ unsigned tile_pack_src(int u, int t0, int rows) {
    const int d = u - t0;
    if ((unsigned)d < (unsigned)rows) return (unsigned)d;
    return (d > -(1 << 22) && d < (1 << 22)) ? (TILE_FAR | ((unsigned)d & 0x7FFFFFu)) : TILE_ESCAPE;
}
// global row of an out-of-tile source (ul >= TILE_FAR); e = its CSR position, for the escape value
__device__ __forceinline__ long long tile_far_row(unsigned ul, int t0, const int* __restrict__ src, long long e);

__device__ inline int load_i32_rare(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ long long tile_far_row(unsigned ul, int t0, const int* __restrict__ src, long long e) {
    if (ul == TILE_ESCAPE) return load_i32_rare(src + e);
    return (long long)t0 + (((int)(ul << 9)) >> 9);  // sign-extend the 23-bit distance
}

// LDS-DMA of one 1 KiB piece (16 B per lane) from inline asm.  Not the builtin: hipcc's wait-count pass books a
// global_load ... lds as a FLAT access that may touch LDS, and while one is outstanding it turns every LDS wait of the wave
// into lgkmcnt(0) -- which serialises a software-pipelined fragment stream (each wait would also wait for the fragments just
// requested for the next unit).  Issued like this the pass does not see the transfer at all; the kernel orders it by hand
// (s_waitcnt vmcnt(0) + barrier before a buffer is read).  gbase and lds_addr are wave-uniform.
__device__ __forceinline__ void lds_dma16(const void* gbase, uint32_t voff, uint32_t lds_addr) {
    // both are wave-uniform by contract; readfirstlane makes that visible to the register allocator where it cannot prove it
    const uint64_t gb = (uint64_t)gbase;
    const uint64_t gu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb);
    const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gu), "s"(la) : "memory", "m0");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// Sum of n consecutive LDS floats in index order (the association of `for (v) sum += s[v]`), eight reads in flight at a time (sixteen: no faster): the plain
// loop costs an LDS round trip per element, and the readouts run it on one lane per graph while the rest of the workgroup waits.
__device__ __forceinline__ float lds_sum_in_order(const float* s, int n) {
    float sum = 0.0f;
    int v = 0;
    for (; v + 8 <= n; v += 8) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = s[v + i];
#pragma unroll
        for (int i = 0; i < 8; i++) sum += x[i];
    }
    if (v < n) {  // the last one to seven elements in one batch as well: absent ones read as +0, which leaves the sum's bits alone
        float x[8];
#pragma unroll
        for (int i = 0; i < 7; i++) x[i] = v + i < n ? s[v + i] : 0.0f;
#pragma unroll
        for (int i = 0; i < 7; i++) sum += x[i];
    }
    return sum;
}

// Streaming (nontemporal) 16-byte store for rows that the next kernel reads only after gigabytes of other rows have gone by:
// they need not displace the gather's working set (neighbour rows, weight stream) from the caches.
__device__ __forceinline__ void stream_store4(float* p, float a, float b, float c, float d) {
    __builtin_nontemporal_store((float4_t){a, b, c, d}, reinterpret_cast<float4_t*>(p));
}
__device__ __forceinline__ void stream_store4(float4* p, const float4& v) { stream_store4(reinterpret_cast<float*>(p), v.x, v.y, v.z, v.w); }

// one v_max_f32 (the compare + select form costs three issue slots); differs from `x < 0 ? 0 : x` only for NaN
__device__ inline float relu1(float x) { return __builtin_fmaxf(x, 0.0f); }

// ---------------------------------------------------------------- atom encoder
// One lane per (node, float4 chunk); fully coalesced 1 KiB stores per wave.  The 173-row table is staged in LDS once
// per (persistent) workgroup: read from global memory it cost 9 x 16 B of L2 traffic per 16 B written (24 GB per 2^18
// molhiv graphs, 1.58 ms); from LDS the kernel is bound by its 2.7 GB of stores.  Same summation order (k = 0..8).
template <int D>
__global__ __launch_bounds__(512) void atom_encoder_kernel(const int* __restrict__ node_feature,
                                                            const float* __restrict__ table,  // [173][D]
                                                            float* __restrict__ h, int n_tot, int* __restrict__ err) {
    constexpr int C = D / 4;
    constexpr int NB = 128;  // nodes per block iteration: their 9 features are staged once (coalesced), not re-read by 25 lanes
    __shared__ __attribute__((aligned(16))) float4 s_tab[ND_FEATURE_TOTAL * C];
    __shared__ int s_row[NB * ND_FEATURE];  // table row per (node, feature), validated
    for (int i = threadIdx.x; i < ND_FEATURE_TOTAL * C; i += 512) s_tab[i] = reinterpret_cast<const float4*>(table)[i];
    const int n_blocks = (n_tot + NB - 1) / NB;
    constexpr int PER = (NB * ND_FEATURE + 511) / 512;  // feature words per thread and block (3)
    int pre[PER];
    auto fetch = [&](int blk) {  // this thread's feature words of block blk (requested one block ahead of their use)
        const long long base = (long long)blk * NB * ND_FEATURE, lim = (long long)n_tot * ND_FEATURE;
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const long long i = base + threadIdx.x + 512 * p;
            pre[p] = (blk < n_blocks && threadIdx.x + 512 * p < NB * ND_FEATURE && i < lim) ? node_feature[i] : 0;
        }
    };
    fetch(blockIdx.x);
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int v0 = blk * NB;
        const int nv = (n_tot - v0) < NB ? (n_tot - v0) : NB;
        __syncthreads();  // previous iteration's readers are done (first time: the table is in place)
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const int i = threadIdx.x + 512 * p;
            if (i < nv * ND_FEATURE) {
                const int k = i % ND_FEATURE;
                int f = pre[p];
                if (f < 0 || f >= c_nd_card[k]) {
                    atomicMax(err, ERR_NODE_FEAT);
                    f = 0;
                }
                s_row[i] = (c_nd_off[k] + f) * C;
            }
        }
        __syncthreads();
        fetch(blk + gridDim.x);
        for (int i = threadIdx.x; i < nv * C; i += 512) {
            const int v = i / C;
            const int c = i - v * C;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < ND_FEATURE; k++) {
                const float4 w = s_tab[s_row[v * ND_FEATURE + k] + c];
                s.x += w.x; s.y += w.y; s.z += w.z; s.w += w.w;
            }
            // streaming store: the rows are next read by another kernel, after 2.7 GB of other rows have gone by
            __builtin_nontemporal_store((float4_t){s.x, s.y, s.z, s.w}, reinterpret_cast<float4_t*>(h) + (size_t)v0 * C + i);
        }
    }
}
// grid for atom_encoder_kernel: persistent, two 512-thread workgroups per CU (69 KB of LDS each at D = 100)
inline int atom_encoder_grid(long long n_tot, int C) {
    (void)C;
    const long long blocks = (n_tot + 127) / 128;
    return (int)(blocks < 512 ? (blocks > 0 ? blocks : 1) : 512);
}

// ---------------------------------------------------------------- readout with the linear head already applied per node
// out[g] = (sum of score[v] over the nodes of graph g, in node order) / n_g + bias: the second half of a mean-pool +
// linear readout whose per-node dot products were computed in the last layer's epilogue (gin_split.hip).
template <int UNUSED = 0>  // a template only for its linkage (the header is included by several translation units)
__global__ __launch_bounds__(256) void segment_mean_bias_kernel(const float* __restrict__ score, const int* __restrict__ node_off,
                                                                         const float* __restrict__ pb, float* __restrict__ out,
                                                                         int num_graphs) {
    const int gidx = blockIdx.x * 256 + threadIdx.x;
    if (gidx >= num_graphs) return;
    const int n0 = node_off[gidx], n1 = node_off[gidx + 1];
    float s = 0.0f;
    for (int v = n0; v < n1; v++) s += score[v];
    out[gidx] = s / (float)(n1 - n0) + pb[0];
}


// Sum of the rows v0, v0 + 2, ... < n1 of a graph (one float4 chunk c of each), in that order: four rows are requested before the first
// is added.  One at a time, every row was its own global round trip on the graph's wavefront (the readout kernels below).
template <int C>
__device__ __forceinline__ void pool_rows_in_order(float4& acc, const float* __restrict__ h, int v0, int n1, int c) {
    int v = v0;
    for (; v + 6 < n1; v += 8) {
        float4 x[4];
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = reinterpret_cast<const float4*>(h)[(size_t)(v + 2 * i) * C + c];
#pragma unroll
        for (int i = 0; i < 4; i++) { acc.x += x[i].x; acc.y += x[i].y; acc.z += x[i].z; acc.w += x[i].w; }
    }
    for (; v < n1; v += 2) {
        const float4 x = reinterpret_cast<const float4*>(h)[(size_t)v * C + c];
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
}

// ---------------------------------------------------------------- readout: mean pool + linear head
// One wavefront per graph; lanes 0..24 take even rows, lanes 32..56 odd rows (float4 chunks).
template <int D>
__global__ __launch_bounds__(256) void mean_pool_linear_kernel(const float* __restrict__ h,
                                                                const int* __restrict__ node_off,
                                                                const float* __restrict__ pw,
                                                                const float* __restrict__ pb,
                                                                float* __restrict__ out, int num_graphs) {
    constexpr int C = D / 4;
    static_assert(C <= 32, "row must fit half a wavefront in float4 chunks");
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= num_graphs) return;
    const int n0 = node_off[g], n1 = node_off[g + 1];
    const int half = lane >> 5, c = lane & 31;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) pool_rows_in_order<C>(acc, h, n0 + half, n1, c);
    acc.x += __shfl_down(acc.x, 32, 64); acc.y += __shfl_down(acc.y, 32, 64);
    acc.z += __shfl_down(acc.z, 32, 64); acc.w += __shfl_down(acc.w, 32, 64);
    float part = 0.f;
    if (half == 0 && c < C) {
        const float n = (float)(n1 - n0);
        const float4 w = reinterpret_cast<const float4*>(pw)[c];
        part = (acc.x / n) * w.x + (acc.y / n) * w.y + (acc.z / n) * w.z + (acc.w / n) * w.w;
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) part += __shfl_down(part, d, 64);
    if (lane == 0) out[g] = pb[0] + part;
}

// ---------------------------------------------------------------- readout: mean pool + linear head with NUM_TASK outputs
// out[g][t] = pb[t] + sum_d mean_v(h[v][d]) pw[t][d]   (finalize + linear<EMB_DIM, NUM_TASK, ...>: GIN/src/finalize.cc:14-34,
// GIN/src/linear.cc:26-47, with NUM_TASK -- 1 in the reference, GIN/src/dcl.h:25 -- as a run-time dimension; ogbg-molpcba has 128).
// Persistent workgroups; the head is kept in LDS transposed ([d][t], so the lanes of a wave read consecutive words), TCH tasks
// at a time; one wavefront per graph pools its rows (two half-waves over alternate rows, float4 chunks) and then takes the
// tasks t = lane, lane + 64, ...  Summation order per output: d = 0..D-1.
template <int D>
__global__ __launch_bounds__(256) void mean_pool_linear_mt_kernel(const float* __restrict__ h, const int* __restrict__ node_off,
                                                                   const float* __restrict__ pw, const float* __restrict__ pb,
                                                                   float* __restrict__ out, int num_graphs, int num_tasks) {
    constexpr int C = D / 4;
    constexpr int TCH = 128;
    static_assert(C <= 32, "row must fit half a wavefront in float4 chunks");
    __shared__ float s_w[D * TCH];
    __shared__ float s_hg[4][D];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t0 = 0; t0 < num_tasks; t0 += TCH) {
        const int nt = (num_tasks - t0) < TCH ? (num_tasks - t0) : TCH;
        __syncthreads();
        for (int i = threadIdx.x; i < nt * D; i += 256) {
            const int t = i / D, d = i - t * D;
            s_w[d * TCH + t] = pw[(size_t)(t0 + t) * D + d];
        }
        __syncthreads();
        for (int g = blockIdx.x * 4 + wv; g < num_graphs; g += gridDim.x * 4) {
            const int n0 = node_off[g], n1 = node_off[g + 1];
            const int half = lane >> 5, c = lane & 31;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) pool_rows_in_order<C>(acc, h, n0 + half, n1, c);
            acc.x += __shfl_down(acc.x, 32, 64); acc.y += __shfl_down(acc.y, 32, 64);
            acc.z += __shfl_down(acc.z, 32, 64); acc.w += __shfl_down(acc.w, 32, 64);
            if (half == 0 && c < C) {
                const float n = (float)(n1 - n0);
                s_hg[wv][4 * c + 0] = acc.x / n; s_hg[wv][4 * c + 1] = acc.y / n;
                s_hg[wv][4 * c + 2] = acc.z / n; s_hg[wv][4 * c + 3] = acc.w / n;
            }
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < nt; t += 64) {
                float sacc = pb[t0 + t];
                for (int d = 0; d < D; d++) sacc += s_hg[wv][d] * s_w[d * TCH + t];
                out[(size_t)g * num_tasks + t0 + t] = sacc;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

static inline int grid_for(long long items, int per_block, int cap) {
    long long nb = (items + per_block - 1) / per_block;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}


// ---------------------------------------------------------------- readout: mean pool + 3-layer head, one wavefront per graph (PNA D=80, DGN D=100 share it)
template <int D, int H1, int H2>
__global__ __launch_bounds__(256) void pool_mlp3_kernel(const float* __restrict__ h, const int* __restrict__ node_off,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         const float* __restrict__ w3, const float* __restrict__ b3,
                                                         float* __restrict__ out, int num_graphs) {
    constexpr int C = D / 4;
    static_assert(C <= 32 && H1 <= 64 && H2 <= 64, "sizes must fit one wavefront");
    __shared__ float s_hg[4][D];
    __shared__ float s_o1[4][H1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + wv;
    if (g >= num_graphs) return;
    const int n0 = node_off[g], n1 = node_off[g + 1];
    const int half = lane >> 5, c = lane & 31;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) pool_rows_in_order<C>(acc, h, n0 + half, n1, c);
    acc.x += __shfl_down(acc.x, 32, 64); acc.y += __shfl_down(acc.y, 32, 64);
    acc.z += __shfl_down(acc.z, 32, 64); acc.w += __shfl_down(acc.w, 32, 64);
    if (half == 0 && c < C) {
        const float n = (float)(n1 - n0);
        s_hg[wv][4 * c + 0] = acc.x / n; s_hg[wv][4 * c + 1] = acc.y / n;
        s_hg[wv][4 * c + 2] = acc.z / n; s_hg[wv][4 * c + 3] = acc.w / n;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < H1) {
        float s = b1[lane];
        for (int i = 0; i < D; i++) s = __builtin_fmaf(s_hg[wv][i], w1[lane * D + i], s);  // (spelled out: the resident kernels' heads round alike)
        s_o1[wv][lane] = relu1(s);
    }
    __builtin_amdgcn_wave_barrier();
    float part = 0.f;
    if (lane < H2) {
        float s = b2[lane];
        for (int i = 0; i < H1; i++) s = __builtin_fmaf(s_o1[wv][i], w2[lane * H1 + i], s);
        part = relu1(s) * w3[lane];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_down(part, d, 64);
    if (lane == 0) out[g] = b3[0] + part;
}

// ---------------------------------------------------------------- tiled aggregation (generic MP unit)
// The HBM-bound half of every two-kernel model (GCN, PNA, DGN; the GIN instance lives in gin.hip, where the scheme
// was developed and measured: 2.1 ms -> 1.13 ms = 62 % of HBM peak at 2^18 molhiv graphs).  Persistent workgroups of
// NTHR threads walk tiles of TR consecutive destination rows.  Per tile, ONE global round trip brings in
//   * the tile's rows of h            (LDS-DMA, lane-linear because the rows are contiguous),
//   * the tile's CSR entries          (contiguous in the CSR) as packed words (row inside the tile | 0xFFFFFF) << 8 | code,
//                                     plus an optional per-edge float derived from the SOURCE node (Policy::src_scalar),
//   * the next tile's row_ptr slice   (consumed one tile later);
// then every (row, float4 chunk) item folds its in-edges in CSR order out of LDS.  A source outside the tile is fetched
// with load_f4_rare(); CSR entries beyond the staged TE are read from global memory in a separate, slower loop.
//
// Policy: D, TR, NTHR, TE (CSR entries of a tile staged in LDS: ~2.2 per row for molecules, 16 per row for the kNN
//   graphs), TABLE_ROWS (rows of a [TABLE_ROWS][D] per-edge-code table kept in LDS, 0 = none),
//   HAS_SCALAR; struct Params; struct Acc;
//   NDST (0..2 per-DESTINATION floats staged per tile row), CONST_FLOATS (per-layer constants kept in LDS);
//   src_scalar(p, u) -> float, dst_stage(p, v, float[NDST]) (global reads allowed: both run at staging),
//   const_ptr(p) -> global pointer to CONST_FLOATS floats;
//   init(acc); edge(acc, x, w, s_src, dst[NDST]);
//   finish(p, acc, self, v, c, in_degree, dst[NDST], s_const, out_base)  (writes the item's outputs)
// Stage 3 touches global memory only for its stores (and the rare out-of-tile gather): any ordinary load inside
// the item loop would make hipcc wait for vmcnt(0), i.e. for the previous item's stores, on every trip.
// model-owned device buffer that grows to the largest batch seen (per-edge scalars)
struct GrowBuf {
    float* p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        FG_HIP_TRY(hipMalloc((void**)&p, n * sizeof(float)));
        cap = n;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct GrowBufI {
    int* p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        FG_HIP_TRY(hipMalloc((void**)&p, n * sizeof(int)));
        cap = n;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Graph-aligned tiles for the tiled kernels: tile t starts at the first row of the graph that contains row t * nominal if
// that is at most `slack` rows back, else at row t * nominal itself; so a tile has at most nominal + slack rows and, for
// graphs of up to `slack` nodes, never cuts a graph.  Why it matters: a neighbour outside the tile costs a global round
// trip inside the in-edge loop, and a wave stalls if ANY of its 64 lanes takes it -- with tiles on a fixed 128-row grid
// 1.8 % of molhiv's and 13.7 % of hep10k's edges cross a tile boundary, i.e. 69 % / ~100 % of the wave iterations stall.
static __global__ __launch_bounds__(256) void tile_bounds_kernel(const int* __restrict__ node_off, int num_graphs, int n_tot, int nominal,
                                                           int slack, int* __restrict__ tile_start, int n_tiles) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t > n_tiles) return;
    if (t == n_tiles) { tile_start[t] = n_tot; return; }
    const int r = t * nominal;
    int lo = 0, hi = num_graphs;  // largest g with node_off[g] <= r  (node_off[0] = 0, node_off[G] = n_tot > r)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (node_off[mid] <= r) lo = mid; else hi = mid;
    }
    const int cut = node_off[lo];
    tile_start[t] = (r - cut <= slack) ? cut : r;
}

// esc[e] = Policy::src_scalar(src[e]) for every CSR entry: once per forward pass (the scalar depends on the batch only)
template <class P>
__global__ __launch_bounds__(256) void edge_scalar_kernel(typename P::Params prm, const int* __restrict__ src,
                                                           float* __restrict__ esc, int e_tot) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < e_tot; e += (long long)gridDim.x * 256)
        esc[e] = P::src_scalar(prm, src[e]);
}

template <class P>
__global__ __launch_bounds__(P::NTHR) void tiled_aggregate_kernel(typename P::Params prm, const float* __restrict__ h,
                                                                   float* __restrict__ out, const int* __restrict__ row_ptr,
                                                                   const int* __restrict__ src,
                                                                   const uint8_t* __restrict__ ecode,
                                                                   const float* __restrict__ table, int n_tot, int n_tiles,
                                                                   const int* __restrict__ tile_start) {
    constexpr int D = P::D, TR = P::TR, NTHR = P::NTHR, C = D / 4, TE = P::TE, NW = NTHR / 64;
    constexpr int TILE_BYTES = TR * D * 4;
    static_assert(TILE_BYTES % 1024 == 0, "tile must be whole 1 KiB DMA pieces");
    constexpr int PIECES = TILE_BYTES / 1024;
    constexpr int TAB4 = P::TABLE_ROWS > 0 ? P::TABLE_ROWS * C : 1;
    __shared__ __attribute__((aligned(16))) float4 s_tab[TAB4];
    __shared__ __attribute__((aligned(16))) float4 s_h[TR * C];
    __shared__ int s_rp[TR + 1];
    __shared__ unsigned s_edge[TE];
    __shared__ float s_es[P::HAS_SCALAR ? TE : 1];
    __shared__ float s_dst[P::NDST > 0 ? P::NDST * TR : 1];
    __shared__ __attribute__((aligned(16))) float s_const[P::CONST_FLOATS > 0 ? P::CONST_FLOATS : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (P::TABLE_ROWS > 0)
        for (int i = threadIdx.x; i < P::TABLE_ROWS * C; i += NTHR) s_tab[i] = reinterpret_cast<const float4*>(table)[i];
    if (P::CONST_FLOATS > 0)
        for (int i = threadIdx.x; i < P::CONST_FLOATS; i += NTHR) s_const[i] = P::const_ptr(prm)[i];
    const float4* h4 = reinterpret_cast<const float4*>(h);
    auto load_rp = [&](int t) -> int {  // entry threadIdx.x of tile t's row_ptr slice (clamped to the array)
        if (t >= n_tiles || threadIdx.x > TR) return 0;
        const long long i = (long long)tile_start[t] + threadIdx.x;
        return row_ptr[i <= n_tot ? i : n_tot];
    };
    auto pack = [&](int u, int code, int t0, int rows) -> unsigned {  // rows: what the tile really holds (<= TR)
        return (tile_pack_src(u, t0, rows) << 8) | (unsigned)code;
    };
    int rp_next = load_rp(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int t0 = tile_start[tile];
        const int rows = tile_start[tile + 1] - t0;  // <= TR by construction (tile_bounds_kernel)
        __syncthreads();  // previous tile fully consumed (and, first time, the table is in place)
        if (threadIdx.x <= TR) s_rp[threadIdx.x] = rp_next;
        __syncthreads();
        const int e0 = s_rp[0];
        const int ne = s_rp[rows] - e0;
        const long long tile_bytes_left = (long long)rows * D * 4;  // pieces past the tile's last row are skipped
        for (int p = wave; p < PIECES && (long long)p * 1024 < tile_bytes_left; p += NW) {
            const char* g = reinterpret_cast<const char*>(h) + (size_t)t0 * D * 4 + p * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(s_h) + p * 1024), 16, 0, 0);
        }
        rp_next = load_rp(tile + gridDim.x);
        for (int i = threadIdx.x; i < ne && i < TE; i += NTHR) {
            const int u = src[e0 + i];
            s_edge[i] = pack(u, ecode ? (int)ecode[e0 + i] : 0, t0, rows);
            // per-edge source scalar, precomputed in CSR order (edge_scalar_kernel): reading Policy::src_scalar(u) here
            // would hang a second global round trip (the scalar of node u) behind the load of u itself
            if constexpr (P::HAS_SCALAR) s_es[i] = prm.esc[e0 + i];
        }
        if (P::NDST > 0 && threadIdx.x < rows) {
            float dv[P::NDST > 0 ? P::NDST : 1];
            P::dst_stage(prm, t0 + threadIdx.x, dv);
#pragma unroll
            for (int k = 0; k < P::NDST; k++) s_dst[k * TR + threadIdx.x] = dv[k];
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rp_next) : : "memory");
        __syncthreads();

        const char* sh_b = reinterpret_cast<const char*>(s_h);
        const char* st_b = reinterpret_cast<const char*>(s_tab);
        int r = threadIdx.x / C, c = threadIdx.x - r * C;
        for (int idx = threadIdx.x; idx < rows * C; idx += NTHR) {
            const int v = t0 + r;
            const int beg = s_rp[r] - e0, end = s_rp[r + 1] - e0;
            float sd[P::NDST > 0 ? P::NDST : 1];
#pragma unroll
            for (int k = 0; k < (P::NDST > 0 ? P::NDST : 1); k++) sd[k] = P::NDST > 0 ? s_dst[k * TR + r] : 0.f;
            typename P::Acc acc;
            P::init(acc);
            if (ne <= TE) {
                int e = beg;
                // four in-edges per trip: their CSR words, then their rows, are four INDEPENDENT LDS reads in flight
                // (one edge per trip is two dependent LDS round trips per edge, and on the kNN graphs -- 16 in-edges per
                // row -- that latency, not HBM, was the whole kernel); the folds stay in CSR order
                for (; e + 4 <= end; e += 4) {
                    unsigned pk[4];
                    float ss[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        pk[k] = s_edge[e + k];
                        ss[k] = P::HAS_SCALAR ? s_es[e + k] : 0.f;
                        asm volatile("" : "+v"(pk[k]), "+v"(ss[k]));
                    }
                    float4 x[4], w[4];
                    // a source outside the tile has bit 31 of its CSR word set: one wave-wide test picks the straight-line body
                    // (no clamps, no branches between the four folds) unless some lane has one
                    if (!__any((int)(pk[0] | pk[1] | pk[2] | pk[3]) < 0)) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (P::TABLE_ROWS > 0) w[k] = *reinterpret_cast<const float4*>(st_b + (pk[k] & 0xFFu) * (D * 4) + c * 16);
                            x[k] = *reinterpret_cast<const float4*>(sh_b + (pk[k] >> 8) * (D * 4) + c * 16);
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) P::edge(acc, x[k], w[k], ss[k], sd);
                        continue;
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const unsigned ul = pk[k] >> 8;
                        w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (P::TABLE_ROWS > 0) w[k] = *reinterpret_cast<const float4*>(st_b + (pk[k] & 0xFFu) * (D * 4) + c * 16);
                        x[k] = *reinterpret_cast<const float4*>(sh_b + (ul < (unsigned)TR ? ul : 0u) * (D * 4) + c * 16);
                        asm volatile("" : "+v"(x[k].x), "+v"(x[k].y), "+v"(x[k].z), "+v"(x[k].w));
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if ((pk[k] >> 8) >= (unsigned)TR) x[k] = load_f4_rare(h4 + (size_t)tile_far_row(pk[k] >> 8, t0, src, (long long)e0 + e + k) * C + c);
                        P::edge(acc, x[k], w[k], ss[k], sd);
                    }
                }
                for (; e < end; e++) {
                    unsigned pk = s_edge[e];
                    float ss = P::HAS_SCALAR ? s_es[e] : 0.f;
                    asm volatile("" : "+v"(pk), "+v"(ss));  // keep these ds_reads (no lds/global pointer select)
                    const unsigned ul = pk >> 8;
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (P::TABLE_ROWS > 0) w = *reinterpret_cast<const float4*>(st_b + (pk & 0xFFu) * (D * 4) + c * 16);
                    float4 x = *reinterpret_cast<const float4*>(sh_b + (ul < (unsigned)TR ? ul : 0u) * (D * 4) + c * 16);
                    asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w));
                    if (ul >= (unsigned)TR) x = load_f4_rare(h4 + (size_t)tile_far_row(ul, t0, src, (long long)e0 + e) * C + c);
                    P::edge(acc, x, w, ss, sd);
                }
            } else {
                for (int e = beg; e < end; e++) {
                    unsigned pk = s_edge[e < TE ? e : TE - 1];
                    float ss = P::HAS_SCALAR ? s_es[e < TE ? e : TE - 1] : 0.f;
                    asm volatile("" : "+v"(pk), "+v"(ss));
                    if (e >= TE) {
                        const int u = src[e0 + e];
                        pk = pack(u, ecode ? (int)ecode[e0 + e] : 0, t0, rows);
                        if (P::HAS_SCALAR) ss = P::src_scalar(prm, u);
                    }
                    const unsigned ul = pk >> 8;
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (P::TABLE_ROWS > 0) w = *reinterpret_cast<const float4*>(st_b + (pk & 0xFFu) * (D * 4) + c * 16);
                    float4 x = *reinterpret_cast<const float4*>(sh_b + (ul < (unsigned)TR ? ul : 0u) * (D * 4) + c * 16);
                    asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w));
                    if (ul >= (unsigned)TR) x = load_f4_rare(h4 + (size_t)tile_far_row(ul, t0, src, (long long)e0 + e) * C + c);
                    P::edge(acc, x, w, ss, sd);
                }
            }
            P::finish(prm, acc, s_h[idx], v, c, end - beg, sd, s_const, out);
            c += NTHR % C;
            r += NTHR / C;
            if (c >= C) { c -= C; r++; }
        }
    }
}

// nominal + slack <= P::TR.  tile_start must hold ceil(n_tot / nominal) + 1 ints (make_tile_bounds).
template <class P>
static inline void launch_tiled_aggregate(const typename P::Params& prm, const float* h, float* out, const CsrView& csr,
                                          const float* table, int n_tot, const int* tile_start, int nominal, hipStream_t s) {
    const int n_tiles = (int)ceil_div_ll(n_tot, nominal);
    if (n_tiles <= 0) return;
    constexpr int lds = (P::TABLE_ROWS > 0 ? P::TABLE_ROWS * P::D * 4 : 16) + P::TR * P::D * 4 + (P::TR + 1) * 4 + P::TE * 4 +
                        (P::HAS_SCALAR ? P::TE * 4 : 4) + P::NDST * P::TR * 4 + P::CONST_FLOATS * 4;
    int per_cu = 160 * 1024 / (lds + 256);
    if (per_cu < 1) per_cu = 1;
    if (per_cu * P::NTHR > 2048) per_cu = 2048 / P::NTHR;
    int grid = 256 * per_cu;  // persistent: as many workgroups per CU as the LDS admits
    if (grid > n_tiles) grid = n_tiles;
    tiled_aggregate_kernel<P><<<grid, P::NTHR, 0, s>>>(prm, h, out, csr.row_ptr, csr.src, P::TABLE_ROWS > 0 ? csr.ecode : nullptr, table,
                                                       n_tot, n_tiles, tile_start);
}
static inline int make_tile_bounds(GrowBufI& buf, const int* node_off, int num_graphs, int n_tot, int nominal, int slack, hipStream_t s) {
    const int n_tiles = (int)ceil_div_ll(n_tot, nominal);
    if (int rc = buf.reserve((size_t)n_tiles + 1)) return rc;
    tile_bounds_kernel<<<(n_tiles + 1 + 255) / 256, 256, 0, s>>>(node_off, num_graphs, n_tot, nominal, slack, buf.p, n_tiles);
    return 0;
}

// ---------------------------------------------------------------- dense layer on fp32 MFMA, input from HBM
// out[node][o] = bias[o] + sum_k W[o][k] in[node][k]   for K = 100 inputs, OUT = 16 * OT outputs (padded).
// Transposed formulation (nodes are MFMA columns), as described in gin.hip: lane (j, g) of a wave loads
// in[j][16 q + 4 g .. +3] (q < 6) and in[j][96 + g]; the W fragments carry the matching k per slot.
// Fragments: wf [OT][6][64][4] (t, q, lane, r), wtail [OT][64], bias padded to 16 * OT.
// pack_dense100() builds them on the host.
template <int OT, int NT, bool RELU_OUT>
__global__ __launch_bounds__(256) void dense100_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ wf, const float* __restrict__ wtail,
                                                        const float* __restrict__ biasp, int n_tot, int out_dim) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const long long node_base = (long long)wave * (16 * NT);
    if (node_base >= n_tot) return;
    float bq[NT][25];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        long long node = node_base + nt * 16 + j;
        if (node >= n_tot) node = n_tot - 1;
        const float* row = in + (size_t)node * 100;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 x = *reinterpret_cast<const float4*>(row + 16 * q + 4 * g);
            bq[nt][4 * q + 0] = x.x; bq[nt][4 * q + 1] = x.y; bq[nt][4 * q + 2] = x.z; bq[nt][4 * q + 3] = x.w;
        }
        bq[nt][24] = row[96 + g];
    }
    const float4* wf4 = reinterpret_cast<const float4*>(wf);
#pragma unroll 1
    for (int t = 0; t < OT; t++) {
        float4_t acc[NT];
        {
            const float4 b = *reinterpret_cast<const float4*>(biasp + 16 * t + 4 * g);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = (float4_t){b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 af = wf4[(size_t)(t * 6 + q) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bq[nt][4 * q + 0], acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bq[nt][4 * q + 1], acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bq[nt][4 * q + 2], acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bq[nt][4 * q + 3], acc[nt], 0, 0, 0);
        }
        {
            const float at = wtail[t * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, bq[nt][24], acc[nt], 0, 0, 0);
        }
        const int col = 16 * t + 4 * g;
        if (col < out_dim) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const long long node = node_base + nt * 16 + j;
                if (node < n_tot) {
                    float4_t r = acc[nt];
                    if (RELU_OUT) { r.x = relu1(r.x); r.y = relu1(r.y); r.z = relu1(r.z); r.w = relu1(r.w); }
                    *reinterpret_cast<float4*>(out + (size_t)node * out_dim + col) = make_float4(r.x, r.y, r.z, r.w);
                }
            }
        }
    }
}

// host: W [out_dim][100] row-major, b [out_dim]  ->  fragments for dense100_kernel<OT>
static inline void pack_dense100(const float* W, const float* b, int out_dim, int OT, std::vector<float>& wf,
                                 std::vector<float>& wtail, std::vector<float>& biasp) {
    wf.assign((size_t)OT * 6 * 64 * 4, 0.0f);
    wtail.assign((size_t)OT * 64, 0.0f);
    biasp.assign((size_t)OT * 16, 0.0f);
    for (int t = 0; t < OT; t++) {
        for (int lane = 0; lane < 64; lane++) {
            const int i = lane & 15, g = lane >> 4, o = 16 * t + i;
            if (o >= out_dim) continue;
            for (int q = 0; q < 6; q++)
                for (int r = 0; r < 4; r++) wf[(((size_t)t * 6 + q) * 64 + lane) * 4 + r] = W[(size_t)o * 100 + 16 * q + 4 * g + r];
            wtail[(size_t)t * 64 + lane] = W[(size_t)o * 100 + 96 + g];
        }
        for (int x = 0; x < 16; x++)
            if (16 * t + x < out_dim) biasp[(size_t)t * 16 + x] = b[16 * t + x];
    }
}

}  // namespace fg
