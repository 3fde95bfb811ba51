// pna.hip -- PNA hot path for gfx950 (MI355X).
//
// Reference per graph (PNA/src/*.cc), 4 layers, dim 80, no edge features:
//   h0[v]    = sum_{k<9} NodeEmb[off_k + feat_k(v)]                              load_inputs.cc:133-179
//   per (v, d): S = sum h[u][d], Q = sum h[u][d]^2, mn = min, mx = max over in-edges (u -> v);
//               mn / mx start from +31.999 / -32 (ap_fixed<16,6> limits) and keep them when v has no
//               in-edge                                                            message_passing.cc:127-147
//   mean = S / indeg (0 -> 1), std = sqrt(relu(Q / indeg - mean^2))               node_embedding.cc:123,143-145
//   t = log(outdeg+1) / avg_deg, scale = avg_deg / log(outdeg+1) (log == 0 -> 1)  node_embedding.cc:148-150
//   acc[o]   = b[o] + sum_i sum_{s,a} W[o][s][a][i] agg_a[i] sf_s,  sf = {1, t, scale}   node_embedding.cc:158-189
//   h'[v]    = h[v] + relu(acc)                                                   node_embedding.cc:205-213
//   out[g]   = head(mean_v h_4[v]),  head = 80 -> 40 (ReLU) -> 20 (ReLU) -> 1     finalize.cc:34-52
//
// Here: acc = b + Y_0 + t Y_1 + scale Y_2 with Y_s = W_s agg  (three 80 x 320 contractions per node on
// fp32 MFMA, per-node scalars applied to the accumulators), one HBM-bound aggregation kernel that writes
// agg[v] = [mean | min | max | std][80], and one wave-per-graph readout kernel.
#include "common.h"
#include "device_common.h"
#include "modelq.h"
#include "dense_split.h"
#include <cmath>
#include <cstring>

namespace fg {

constexpr int PNA_D = 80;
constexpr int PNA_L = 4;
constexpr int PNA_C = PNA_D / 4;   // 20 float4 chunks per row
constexpr int PNA_OT = 5;          // 16-row output tiles
constexpr int PNA_NA = 4;          // aggregators, reference enum order: mean, min, max, std
constexpr int PNA_NS = 3;          // scalers: none, t, scale
constexpr float PNA_SENT_MAX = 31.9990234375f;  // ap_fixed_max<ap_fixed<16,6>>, PNA/src/util.h:41-46
constexpr float PNA_SENT_MIN = -32.0f;          // ap_fixed_min, PNA/src/util.h:34-39

// agg[v][a][d], a = {mean, min, max, std}: policy of the generic tiled aggregation (device_common.h)
struct PnaAggPolicy {
    // TR = LDS capacity in rows: tiles are graph aligned, 112 nominal rows + up to 48 to reach the graph boundary behind them
    static constexpr int D = PNA_D, TR = 160, NTHR = 512, TE = 16 * 160, TABLE_ROWS = 0;
    static constexpr bool HAS_SCALAR = false;
    static constexpr int NDST = 0, CONST_FLOATS = 0;
    struct Params { int unused; };
    struct Acc { float4 S, Q, mn, mx; };
    __device__ static float src_scalar(const Params&, int) { return 0.f; }
    __device__ static void dst_stage(const Params&, int, float*) {}
    __device__ static const float* const_ptr(const Params&) { return nullptr; }
    __device__ static void init(Acc& a) {
        a.S = make_float4(0.f, 0.f, 0.f, 0.f);
        a.Q = a.S;
        a.mn = make_float4(PNA_SENT_MAX, PNA_SENT_MAX, PNA_SENT_MAX, PNA_SENT_MAX);
        a.mx = make_float4(PNA_SENT_MIN, PNA_SENT_MIN, PNA_SENT_MIN, PNA_SENT_MIN);
    }
    __device__ static void edge(Acc& a, const float4& x, const float4&, float, const float*) {
        a.S.x += x.x; a.S.y += x.y; a.S.z += x.z; a.S.w += x.w;
        a.Q.x += x.x * x.x; a.Q.y += x.y * x.y; a.Q.z += x.z * x.z; a.Q.w += x.w * x.w;
        a.mn.x = x.x < a.mn.x ? x.x : a.mn.x; a.mn.y = x.y < a.mn.y ? x.y : a.mn.y;
        a.mn.z = x.z < a.mn.z ? x.z : a.mn.z; a.mn.w = x.w < a.mn.w ? x.w : a.mn.w;
        a.mx.x = x.x > a.mx.x ? x.x : a.mx.x; a.mx.y = x.y > a.mx.y ? x.y : a.mx.y;
        a.mx.z = x.z > a.mx.z ? x.z : a.mx.z; a.mx.w = x.w > a.mx.w ? x.w : a.mx.w;
    }
    __device__ static void finish(const Params&, const Acc& a, const float4&, int v, int c, int indeg, const float*, const float*,
                                  float* out) {
        const float deg = (float)(indeg == 0 ? 1 : indeg);
        float4 mean, sd;
        mean.x = a.S.x / deg; mean.y = a.S.y / deg; mean.z = a.S.z / deg; mean.w = a.S.w / deg;
        sd.x = sqrtf(relu1(a.Q.x / deg - mean.x * mean.x)); sd.y = sqrtf(relu1(a.Q.y / deg - mean.y * mean.y));
        sd.z = sqrtf(relu1(a.Q.z / deg - mean.z * mean.z)); sd.w = sqrtf(relu1(a.Q.w / deg - mean.w * mean.w));
        float4* o = reinterpret_cast<float4*>(out) + (size_t)v * (PNA_NA * PNA_C) + c;
        stream_store4(o + 0 * PNA_C, mean); stream_store4(o + 1 * PNA_C, a.mn); stream_store4(o + 2 * PNA_C, a.mx); stream_store4(o + 3 * PNA_C, sd);
    }
};

// h'[v] = h[v] + relu(b + Y_0 + t Y_1 + scale Y_2),  Y_s = W_s agg[v]   (K = 320, 80 outputs)
// One wave = 16 nodes (MFMA columns).  Lane (j, g) holds agg[j][a][16 q + 4 g + r]; the W fragment for
// (s, t, a, q) carries W[16 t + i][s][a][16 q + 4 g + r] in slot g.  Fragments [3][5][4][5][64][4].
__global__ __launch_bounds__(256) void pna_dense_kernel(const float* __restrict__ agg, const float* __restrict__ h,
                                                         float* __restrict__ hout, const int* __restrict__ out_deg,
                                                         const float* __restrict__ wf, const float* __restrict__ bias,
                                                         float avg_deg, int n_tot) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const long long node_base = (long long)wave * 16;
    if (node_base >= n_tot) return;
    long long node = node_base + j;
    const bool valid = node < n_tot;
    if (!valid) node = n_tot - 1;

    float bq[PNA_NA][20];
#pragma unroll
    for (int a = 0; a < PNA_NA; a++) {
        const float* row = agg + ((size_t)node * PNA_NA + a) * PNA_D + 4 * g;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const float4 x = *reinterpret_cast<const float4*>(row + 16 * q);
            bq[a][4 * q + 0] = x.x; bq[a][4 * q + 1] = x.y; bq[a][4 * q + 2] = x.z; bq[a][4 * q + 3] = x.w;
        }
    }
    const float logd = logf((float)(out_deg[node] + 1));  // load_inputs.cc:110 (out-degree)
    const float sf_t = logd / avg_deg;
    const float sf_scale = (logd == 0.0f) ? 1.0f : avg_deg / logd;

    float4_t fin[PNA_OT];
#pragma unroll
    for (int t = 0; t < PNA_OT; t++) {
        const float4 b = *reinterpret_cast<const float4*>(bias + 16 * t + 4 * g);
        fin[t] = (float4_t){b.x, b.y, b.z, b.w};
    }
    const float4* wf4 = reinterpret_cast<const float4*>(wf);
#pragma unroll 1
    for (int s = 0; s < PNA_NS; s++) {
        const float sf = s == 0 ? 1.0f : (s == 1 ? sf_t : sf_scale);
#pragma unroll
        for (int t = 0; t < PNA_OT; t++) {
            float4_t y0 = (float4_t){0.f, 0.f, 0.f, 0.f}, y1 = y0;  // two accumulators: no back-to-back dependency
#pragma unroll
            for (int a = 0; a < PNA_NA; a++) {
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const float4 af = wf4[((((size_t)s * PNA_OT + t) * PNA_NA + a) * 5 + q) * 64 + lane];
                    y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bq[a][4 * q + 0], y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bq[a][4 * q + 1], y1, 0, 0, 0);
                    y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bq[a][4 * q + 2], y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bq[a][4 * q + 3], y1, 0, 0, 0);
                }
            }
            fin[t].x += sf * (y0.x + y1.x); fin[t].y += sf * (y0.y + y1.y);
            fin[t].z += sf * (y0.z + y1.z); fin[t].w += sf * (y0.w + y1.w);
        }
    }
    if (valid) {
#pragma unroll
        for (int t = 0; t < PNA_OT; t++) {
            const size_t off = (size_t)node * PNA_D + 16 * t + 4 * g;
            const float4 hv = *reinterpret_cast<const float4*>(h + off);
            *reinterpret_cast<float4*>(hout + off) =
                make_float4(hv.x + relu1(fin[t].x), hv.y + relu1(fin[t].y), hv.z + relu1(fin[t].z), hv.w + relu1(fin[t].w));
        }
    }
}

// The same update on the f16 matrix pipe, every fp32 product split into three f16 products (dense_split.h / DESIGN.md
// section 4).  pna_dense_kernel reads its 300 KiB of fp32 fragments per layer from L2 in every wave (30 GB per launch
// at 2^15 hep10k graphs: L2-bound at 3.3 ms); here a workgroup of 8 waves (128 nodes) streams them through LDS once,
// K-step by K-step: chunk ks holds the hi/lo fragments of all 15 (scaler, output tile) pairs for the 32 aggregate
// features of K-step ks (30 KiB, double buffered), and each wave keeps 15 accumulators (60 registers) while the B
// operand of a K-step (two float4 of the node's aggregate row, split on the fly) is loaded one K-step ahead.
// K = 320 = 10 K-steps exactly: slot e of K-step ks is aggregate feature [a][16 q + 4 g + (e & 3)] with
// 5 a + q = 2 ks + (e >> 2).  The scalers are applied to the accumulators in the epilogue, in the oracle's order.
constexpr int PNA_KS = 10;
constexpr int PNA_CHUNK = PNA_NS * PNA_OT * 2 * 1024;  // 30 KiB
constexpr size_t PNA_SPLIT_LAYER_BYTES = (size_t)PNA_KS * PNA_CHUNK;

__device__ __forceinline__ void pna_issue_chunk(const uint8_t* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int piece = wave + 8 * p;  // 30 pieces of 1 KiB over 8 waves
        if (piece < PNA_CHUNK / 1024) {
            const uint8_t* g = gchunk + piece * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds_buf + piece * 1024), 16, 0, 0);
        }
    }
}

__device__ __forceinline__ void pna_split_step(const char* wb, int lane, const float4& x0, const float4& x1, float4_t (&y)[PNA_NS * PNA_OT],
                                               float& vmax) {
    ds_uint4_t b_hi, b_lo;
    DS_SPLIT2(x0.x, x0.y, b_hi.x, b_lo.x);
    DS_SPLIT2(x0.z, x0.w, b_hi.y, b_lo.y);
    DS_SPLIT2(x1.x, x1.y, b_hi.z, b_lo.z);
    DS_SPLIT2(x1.z, x1.w, b_hi.w, b_lo.w);
    vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(x0.x)), __builtin_fabsf(x0.y));
    vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(x0.z)), __builtin_fabsf(x0.w));
    vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(x1.x)), __builtin_fabsf(x1.y));
    vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(x1.z)), __builtin_fabsf(x1.w));
    asm volatile("" : "+v"(vmax));  // computed here, not sunk to the kernel's end with all its inputs kept alive
#pragma unroll
    for (int st = 0; st < PNA_NS * PNA_OT; st++) {
        const ds_uint4_t a_hi = *reinterpret_cast<const ds_uint4_t*>(wb + (st * 2 + 0) * 1024 + lane * 16);
        const ds_uint4_t a_lo = *reinterpret_cast<const ds_uint4_t*>(wb + (st * 2 + 1) * 1024 + lane * 16);
        y[st] = DS_MFMA16(a_hi, b_hi, y[st]);
        y[st] = DS_MFMA16(a_hi, b_lo, y[st]);
        y[st] = DS_MFMA16(a_lo, b_hi, y[st]);
    }
}

__global__ __launch_bounds__(512) void pna_dense_split_kernel(const float* __restrict__ agg, const float* __restrict__ h,
                                                               float* __restrict__ hout, const int* __restrict__ out_deg,
                                                               const uint8_t* __restrict__ wpk, const float* __restrict__ bias,
                                                               float avg_deg, float oscale, int n_tot,
                                                               int* __restrict__ range_flag) {
    // two DISTINCT LDS objects: hipcc can then prove that the DMA into one does not alias the ds_reads of the other
    __shared__ __attribute__((aligned(16))) char s_a[PNA_CHUNK];  // even K-steps
    __shared__ __attribute__((aligned(16))) char s_b[PNA_CHUNK];  // odd K-steps
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    long long node = (long long)blockIdx.x * 128 + wave * 16 + j;
    const bool valid = node < n_tot;
    if (!valid) node = n_tot - 1;
    pna_issue_chunk(wpk, s_a, wave, lane);
    // aggregate row of this lane's node as K-step operands: Q = 5 a + q, float4 at [a][16 q + 4 g]
    const float* row = agg + (size_t)node * (PNA_NA * PNA_D) + 4 * g;
    auto qoff = [](int Q) { return (Q / 5) * PNA_D + (Q % 5) * 16; };
    float4 x0 = *reinterpret_cast<const float4*>(row + qoff(0)), x1 = *reinterpret_cast<const float4*>(row + qoff(1));
    float4_t y[PNA_NS * PNA_OT];
#pragma unroll
    for (int i = 0; i < PNA_NS * PNA_OT; i++) y[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    float vmax = 0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll  // fully: in a rolled loop hipcc rotates the 15 accumulators through registers and spills one per trip, and a
                // spill reload queues behind the chunk DMA in vmcnt order
    for (int ks = 0; ks < PNA_KS; ks += 2) {
        // even K-step from s_a while chunk ks+1 streams into s_b and its B operand into registers
        pna_issue_chunk(wpk + (size_t)(ks + 1) * PNA_CHUNK, s_b, wave, lane);
        const float4 n0 = *reinterpret_cast<const float4*>(row + qoff(2 * ks + 2)), n1 = *reinterpret_cast<const float4*>(row + qoff(2 * ks + 3));
        pna_split_step(s_a, lane, x0, x1, y, vmax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float4 m0 = n0, m1 = n1;
        if (ks + 2 < PNA_KS) {
            pna_issue_chunk(wpk + (size_t)(ks + 2) * PNA_CHUNK, s_a, wave, lane);
            m0 = *reinterpret_cast<const float4*>(row + qoff(2 * ks + 4));
            m1 = *reinterpret_cast<const float4*>(row + qoff(2 * ks + 5));
        }
        pna_split_step(s_b, lane, n0, n1, y, vmax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        x0 = m0; x1 = m1;
    }
    const float logd = logf((float)(out_deg[node] + 1));  // load_inputs.cc:110 (out-degree)
    const float sf_t = logd / avg_deg;
    const float sf_scale = (logd == 0.0f) ? 1.0f : avg_deg / logd;
    if (valid) {
#pragma unroll
        for (int t = 0; t < PNA_OT; t++) {
            const float4 b = *reinterpret_cast<const float4*>(bias + 16 * t + 4 * g);
            float4_t fin = {b.x, b.y, b.z, b.w};
            fin += y[0 * PNA_OT + t] * oscale;
            fin += sf_t * (y[1 * PNA_OT + t] * oscale);
            fin += sf_scale * (y[2 * PNA_OT + t] * oscale);
            const size_t off = (size_t)node * PNA_D + 16 * t + 4 * g;
            const float4 hv = *reinterpret_cast<const float4*>(h + off);
            *reinterpret_cast<float4*>(hout + off) =
                make_float4(hv.x + relu1(fin.x), hv.y + relu1(fin.y), hv.z + relu1(fin.z), hv.w + relu1(fin.w));
        }
    }
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

// ---------------------------------------------------------------- fused layer: aggregation + dense update in one kernel
// The aggregates (mean | min | max | std, 1 280 B per node) never go to HBM -- nor to LDS, nor do they ever exist as a whole.
// The contraction index is re-ordered FEATURE-major: K-step k (of 10) covers the four aggregates of features 8k .. 8k+7, lane
// (j, g) supplying both aggregate quadruples of features 8k + 2g, 8k + 2g + 1 of node j (K-slot e: feature 8k + 2g + (e >> 2),
// aggregator e & 3).  So a K-step's B operand needs one pass over node j's in-edges reading EIGHT bytes per neighbour row and
// eight accumulators; its 45 MFMAs then run while the same wave already gathers the next K-step's slice -- matrix pipe and
// VALU / LDS overlap inside every wave, and with ~110 registers sixteen waves fit a CU.
// A persistent 16-wave workgroup (one per CU) walks tiles of WHOLE graphs (GraphTiles: <= 256 rows, <= 4 608 in-edges): the
// tile's rows of h come into LDS by DMA (padded stride, below), its CSR slice as bytes (the first 16 in-edges of a row are kept
// packed in four registers and re-walked per K-step); the weight chunks (30 KiB per K-step, pna_pack_stream_layer) stream
// through two LDS buffers as in pna_dense_split_kernel.  In-edges are summed in CSR order as everywhere else.
#ifndef PNA_DMA_MID
#define PNA_DMA_MID 0  // -DPNA_DMA_MID=1 (scripts/dev/variant.sh): the chunk request between a wave's phases, compiled in without the DEV hooks
#endif
constexpr int PNA_FT_ROWS = 256;
constexpr int PNA_FT_EDGES = 4608;
// LDS row stride: 84 floats = 21 slots of 16 B.  A gather instruction reads 16 different rows at one column; with the rows'
// natural 320 B (80 dwords = 16 mod 64) they would fall on 4 of the 16 bank groups; 84 dwords = 20 mod 64 spreads 16
// consecutive rows over all of them.  The DMA places the padding: every lane picks its own global address.
constexpr int PNA_FT_STRIDE = 84;
constexpr int PNA_FT_WAVES = 16;

// Per-tile descriptor (pna_tile_build_kernel from the caller's edge list, or pna_tile_desc_kernel from the batch CSR; once per batch
// pass: the four layers share it): the tile's CSR slice as the layer kernels want it in LDS -- [0, 4608) source rows inside the tile,
// one byte each; [4608, 5632) u16 row offsets into them (rows + 1 used); [5632, 6144) u16 out-degrees of the tile's rows (the
// resident kernel's degree scalers) -- 6 pieces of 1 KiB that come by LDS-DMA.  (Staged from the batch CSR by the layer kernel itself -- load,
// subtract the tile's first row, store a byte -- hipcc waited for every word in turn: up to six serialized global round trips per tile
// and layer behind the row DMA, and two dependent scalar loads for the slice's bounds at the top of every tile.)
constexpr int PNA_DESC_RP = PNA_FT_EDGES;
constexpr int PNA_DESC_OD = PNA_FT_EDGES + 1024;
constexpr int PNA_DESC_BYTES = PNA_FT_EDGES + 1536;
__global__ __launch_bounds__(256) void pna_tile_desc_kernel(const int* __restrict__ row_ptr, const int* __restrict__ src, const int* __restrict__ out_deg,
                                                            const int* __restrict__ tile_row, uint8_t* __restrict__ desc, int n_tiles) {
    const int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const int t0 = tile_row[tile];
    int rows = tile_row[tile + 1] - t0;
    if (rows > PNA_FT_ROWS) rows = PNA_FT_ROWS;
    const int e0 = row_ptr[t0];
    int ne = row_ptr[t0 + rows] - e0;
    if (ne > PNA_FT_EDGES) ne = PNA_FT_EDGES;  // cannot happen for a validated batch (the host packed by edge count)
    uint8_t* d = desc + (size_t)tile * PNA_DESC_BYTES;
    for (int i = threadIdx.x * 4; i < PNA_FT_EDGES; i += 1024) {  // four source bytes per store
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (i + b < ne) w |= (uint32_t)((src[e0 + i + b] - t0) & 255) << (8 * b);
        *reinterpret_cast<uint32_t*>(d + i) = w;
    }
    uint16_t* rp = reinterpret_cast<uint16_t*>(d + PNA_DESC_RP);
    for (int i = threadIdx.x; i < 512; i += 256) {
        int o = i <= rows ? row_ptr[t0 + i] - e0 : ne;
        rp[i] = (uint16_t)(o < 0 ? 0 : (o > ne ? ne : o));
    }
    uint16_t* od = reinterpret_cast<uint16_t*>(d + PNA_DESC_OD);
    od[threadIdx.x] = (uint16_t)((int)threadIdx.x < rows ? out_deg[t0 + threadIdx.x] : 0);
}

// The same descriptor straight from the caller's edge list: what the resident kernel needs of load_graph (PNA/src/load_inputs.cc:87-131)
// WITHOUT the batch CSR.  One 256-thread workgroup per tile of whole graphs: the tile's edges are a contiguous slice of edge_list;
// each is validated as the index build validates it, turned into (source row, destination row) of the tile and ORed into a 256 x 256
// bit matrix in LDS.  Thread v then walks the set bits of row v in ascending order -- the CSR's order (sources ascending; copies of a
// duplicate edge are equal terms, so their order among themselves is immaterial) -- and emits the row's source bytes: no sort at all.
// An atomicOr that finds its bit set has found a duplicate edge: the copy goes to a list, and a row that has copies counts, per source,
// how many (rare; a short scan).  Bit-identical to launch_build_csr + pna_tile_desc_kernel.
// `list` (GraphTiles::bp_list, bin-packed tiles; null: tile t = the graphs tile_graph[t] .. and the batch's rows tile_row[t] ..): tile t =
// the graphs list[tile_graph[t]] .., one behind the other, tile_row[t] its first row in the tile-ordered row space.  fidx (that row
// space, 3 words per row): the row's nine encoder table rows offset_k + feature_k as bytes, validated here as atom_encoder_kernel
// validates them -- what the resident kernel's loader reads instead of the caller's 36 B of node features (it cannot find a list's rows).
__global__ __launch_bounds__(256) void pna_tile_build_kernel(BatchView b, const int* __restrict__ tile_row, const int* __restrict__ tile_graph,
                                                             int n_tiles, uint8_t* __restrict__ desc, int* __restrict__ err,
                                                             const int* __restrict__ list, uint32_t* __restrict__ fidx) {
    __shared__ uint32_t s_adj[PNA_FT_ROWS][8];
    __shared__ int s_cnt[PNA_FT_ROWS], s_odeg[PNA_FT_ROWS], s_wsum[4], s_next;
    __shared__ uint16_t s_ext[PNA_FT_EDGES];  // copies of duplicate edges: destination << 8 | source
    __shared__ __attribute__((aligned(16))) uint8_t s_out[PNA_DESC_BYTES];
    __shared__ int s_node[PNA_FT_ROWS];  // the batch's node behind every row of the tile
    const int tile = blockIdx.x, tid = threadIdx.x;
    if (tile >= n_tiles) return;
    const int t0 = tile_row[tile];
    int rows = tile_row[tile + 1] - t0;
    if (rows > PNA_FT_ROWS) rows = PNA_FT_ROWS;
    const int g0 = tile_graph[tile], g1 = tile_graph[tile + 1];
    s_node[tid] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s_adj[tid][i] = 0u;
    s_cnt[tid] = 0;
    s_odeg[tid] = 0;
    if (tid == 0) s_next = 0;
    for (int i = tid; i < PNA_DESC_BYTES / 4; i += 256) reinterpret_cast<uint32_t*>(s_out)[i] = 0u;
    __syncthreads();
    int run = 0;
    for (int gi = g0; gi < g1; gi++) {  // (a handful of graphs per tile; their headers are wave-uniform scalar loads)
        const int gph = list ? list[gi] : gi;
        const int n = b.nums_of_nodes[gph], first = b.node_off[gph], base = list ? run : first - t0, e0 = b.edge_off[gph], ne = b.edge_off[gph + 1] - e0;
        run += n;
        for (int k = tid; k < n; k += 256)
            if (base + k < PNA_FT_ROWS) s_node[base + k] = first + k;
        for (int e = tid; e < ne; e += 256) {
            const int2 uv = reinterpret_cast<const int2*>(b.edge_list)[e0 + e];
            int u = uv.x, v = uv.y;
            if (!((u >= 0) & (u < n) & (v >= 0) & (v < n))) {  // as the index build: flag it, then a self-loop on the graph's node 0
                atomicMax(err, ERR_EDGE_RANGE);
                u = 0;
                v = 0;
            }
            u += base; v += base;
            if (u >= PNA_FT_ROWS || v >= PNA_FT_ROWS) continue;  // (cannot happen: the host packed whole graphs into <= 256 rows)
            const uint32_t bit = 1u << (u & 31);
            const uint32_t old = atomicOr(&s_adj[v][u >> 5], bit);
            if (old & bit) {
                const int k = atomicAdd(&s_next, 1);
                if (k < PNA_FT_EDGES) s_ext[k] = (uint16_t)((v << 8) | u);
            }
            atomicAdd(&s_cnt[v], 1);
            atomicAdd(&s_odeg[u], 1);
        }
    }
    __syncthreads();
    // row offsets: exclusive scan of the in-degrees over the 256 rows (thread = row)
    const int lane = tid & 63, wv = tid >> 6;
    const int c = tid < rows ? s_cnt[tid] : 0;
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int off = incl - c;
    for (int w = 0; w < wv; w++) off += s_wsum[w];
    int total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    if (total > PNA_FT_EDGES) total = PNA_FT_EDGES;  // cannot happen for a validated batch (the host packed by edge count)
    uint16_t* rp = reinterpret_cast<uint16_t*>(s_out + PNA_DESC_RP);
    rp[tid] = (uint16_t)(tid <= rows ? (off < total ? off : total) : total);
    rp[256 + tid] = (uint16_t)total;
    reinterpret_cast<uint16_t*>(s_out + PNA_DESC_OD)[tid] = (uint16_t)(tid < rows ? s_odeg[tid] : 0);
    if (tid < rows) {
        const int v = tid;
        int nd = c;  // copies beyond the first of any (u -> v): in-degree - distinct sources
#pragma unroll
        for (int i = 0; i < 8; i++) nd -= __popc(s_adj[v][i]);
        const int next = nd > 0 ? (s_next < PNA_FT_EDGES ? s_next : PNA_FT_EDGES) : 0;
        int pos = off;
        for (int sb = 0; sb < 8; sb++) {
            uint32_t w = s_adj[v][sb];
            while (w) {
                const int u = 32 * sb + __ffs((int)w) - 1;
                w &= w - 1;
                int mult = 1;
                for (int k = 0; k < next; k++) mult += s_ext[k] == (uint16_t)((v << 8) | u);
                for (int m = 0; m < mult; m++) {
                    if (pos < PNA_FT_EDGES) s_out[pos] = (uint8_t)u;
                    pos++;
                }
            }
        }
    }
    if (fidx && tid < rows) {  // the row's nine encoder table rows (load_inputs.cc:133-179), as bytes
        constexpr int off[ND_FEATURE] = {0, 119, 123, 135, 147, 157, 163, 169, 171};  // load_inputs.cc:5
        constexpr int card[ND_FEATURE] = {119, 4, 12, 12, 10, 6, 6, 2, 2};            // host_load.cc:5
        const int* nf = b.node_feature + (size_t)s_node[tid] * ND_FEATURE;
        uint32_t fw[3] = {0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) {
            int f = nf[k];
            if (f < 0 || f >= card[k]) {
                atomicMax(err, ERR_NODE_FEAT);
                f = 0;
            }
            fw[k >> 2] |= (uint32_t)(off[k] + f) << (8 * (k & 3));
        }
        uint32_t* o = fidx + (size_t)(t0 + tid) * 3;
        o[0] = fw[0]; o[1] = fw[1]; o[2] = fw[2];
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(desc + (size_t)tile * PNA_DESC_BYTES);
    for (int i = tid; i < PNA_DESC_BYTES / 16; i += 256) dst[i] = reinterpret_cast<const uint4*>(s_out)[i];
}
// a tile's descriptor -> LDS: six pieces, dealt to the waves that issue one row piece fewer
__device__ __forceinline__ void pna_issue_desc(const uint8_t* __restrict__ desc, int tile, char* lds_buf, int wave, int lane) {
    const int piece = PNA_FT_WAVES - 1 - wave;  // waves 15, 14, ..., 10
    if (piece < PNA_DESC_BYTES / 1024)
        lds_dma16(desc + (size_t)tile * PNA_DESC_BYTES + piece * 1024, (uint32_t)lane * 16u, lds_addr_of(lds_buf) + piece * 1024);
}

__device__ __forceinline__ void pna_issue_chunk_asm(const uint8_t* __restrict__ gchunk, char* lds_buf, int wave, int lane) {
    const uint32_t lb = lds_addr_of(lds_buf);
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int piece = wave + PNA_FT_WAVES * p;  // 30 pieces of 1 KiB over 16 waves
        if (piece < PNA_CHUNK / 1024) lds_dma16(gchunk + piece * 1024, (uint32_t)lane * 16u, lb + piece * 1024);
    }
}
// rows of tile [t0, t0 + rows) -> LDS with the padded stride: LDS slot s = 21 row + c (c = 20 is padding) <- global slot
// 20 row + c; <= 84 pieces of 64 slots (a piece may run past the last row: the array has slack)
__device__ __forceinline__ void pna_issue_rows(const float* __restrict__ h, int t0, int rows, float* s_rows, int wave, int lane) {
    const int np = (rows * 21 + 63) >> 6;
    const uint32_t lb = lds_addr_of(s_rows);
    const char* gb = reinterpret_cast<const char*>(h) + (size_t)t0 * (PNA_D * 4);
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const int piece = wave + PNA_FT_WAVES * r;
        if (piece < np) {
            const int sl = piece * 64 + lane, row = sl / 21, c = sl - row * 21;
            lds_dma16(gb, (uint32_t)(row * 20 + (c < 20 ? c : 19)) * 16u, lb + piece * 1024);
        }
    }
}

// the 45 MFMAs of one K-step: fifteen (scaler, output tile) accumulators in five groups of three, product-major (no MFMA waits for
// its predecessor's accumulator).  Four waves per SIMD: the fragment reads of one wave hide under the MFMAs and gathers of the others.
__device__ __forceinline__ void pna_stream_mfma(const char* wb, int lane, const ds_uint4_t& b_hi, const ds_uint4_t& b_lo,
                                                float4_t (&y)[PNA_NS * PNA_OT]) {
#pragma unroll
    for (int G = 0; G < 8; G++) {  // pairs of accumulators (the last group has one): four fragment registers sets live at a time
        constexpr int NST = PNA_NS * PNA_OT;
        const int n = G * 2 + 1 < NST ? 2 : 1;
        ds_uint4_t f[4];
#pragma unroll
        for (int i = 0; i < 2 * n; i++) f[i] = *reinterpret_cast<const ds_uint4_t*>(wb + (G * 4 + i) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < n; i++) y[G * 2 + i] = DS_MFMA16(f[2 * i], b_hi, y[G * 2 + i]);
#pragma unroll
        for (int i = 0; i < n; i++) y[G * 2 + i] = DS_MFMA16(f[2 * i], b_lo, y[G * 2 + i]);
#pragma unroll
        for (int i = 0; i < n; i++) y[G * 2 + i] = DS_MFMA16(f[2 * i + 1], b_hi, y[G * 2 + i]);
    }
}

// h'[v] = h[v] + relu(b + Y_0 + t Y_1 + scale Y_2) for one float4 of outputs (node_embedding.cc:148-150,205-213), the accumulators still
// carrying the weights' power-of-two scale.  Every fusion is spelled out: the per-layer and the graph-resident kernel then round alike
// (left to -ffp-contract the same expression contracted differently in the two kernels: 1 ulp per layer, 10 ulp on a logit).
__device__ __forceinline__ float4 pna_update(const float4& hv, const float4& b, const float4_t& y0, const float4_t& y1, const float4_t& y2,
                                             float oscale, float sf_t, float sf_scale) {
    float o[4];
    const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float f = __builtin_fmaf(y0[c], oscale, bb[c]);
        f = __builtin_fmaf(sf_t, y1[c] * oscale, f);
        f = __builtin_fmaf(sf_scale, y2[c] * oscale, f);
        o[c] = hh[c] + relu1(f);
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

// one K-step's B operand: the aggregates of features f0 = 8k + 2g, f0 + 1 of this lane's node, gathered out of the LDS tile
struct PnaSlice { float S0, S1, Q0, Q1, mn0, mn1, mx0, mx1; };
__device__ __forceinline__ void pna_slice_add(PnaSlice& a, const float2& x) {
    a.S0 += x.x; a.S1 += x.y;
    a.Q0 = __builtin_fmaf(x.x, x.x, a.Q0); a.Q1 = __builtin_fmaf(x.y, x.y, a.Q1);  // explicit: the full and the ragged path round alike
    // min / max as bare instructions: fminf / fmaxf on a value read from memory cost a canonicalising v_max each (the value could be
    // a signalling NaN); rows of h are finite (the range flag catches what is not)
    asm("v_min_f32 %0, %1, %0" : "+v"(a.mn0) : "v"(x.x));
    asm("v_min_f32 %0, %1, %0" : "+v"(a.mn1) : "v"(x.y));
    asm("v_max_f32 %0, %1, %0" : "+v"(a.mx0) : "v"(x.x));
    asm("v_max_f32 %0, %1, %0" : "+v"(a.mx1) : "v"(x.y));
}
__device__ __forceinline__ void pna_slice_edge(PnaSlice& a, const float* __restrict__ s_h, int u, int col) {
    pna_slice_add(a, *reinterpret_cast<const float2*>(s_h + u * PNA_FT_STRIDE + col));
}
// a finished slice -> the K-step's B operand: mean = S / indeg (0 -> 1), std = sqrt(relu(Q / indeg - mean^2)) (node_embedding.cc:123,143-145)
__device__ __forceinline__ void pna_slice_finish(const PnaSlice& a, int indeg, ds_uint4_t& b_hi, ds_uint4_t& b_lo, float& vmax) {
    // mean = S / indeg (0 -> 1), std = sqrt(relu(Q / indeg - mean^2))   (node_embedding.cc:123,143-145)
    // 1 / indeg and the square root as single instructions (v_rcp_f32, v_sqrt_f32: 1 ulp): four IEEE divisions and two IEEE square
    // roots per slice are ~60 dependent VALU instructions, and at two waves per SIMD nothing hides their latency
    const float rdeg = __builtin_amdgcn_rcpf((float)(indeg == 0 ? 1 : indeg));
    const float m0 = a.S0 * rdeg, m1 = a.S1 * rdeg;
    const float sd0 = __builtin_amdgcn_sqrtf(relu1(__builtin_fmaf(-m0, m0, a.Q0 * rdeg))), sd1 = __builtin_amdgcn_sqrtf(relu1(__builtin_fmaf(-m1, m1, a.Q1 * rdeg)));
    // K-slots e = 0..7: (feature f0: mean, min, max, std), (feature f0 + 1: the same)
    DS_SPLIT2(m0, a.mn0, b_hi.x, b_lo.x);
    DS_SPLIT2(a.mx0, sd0, b_hi.y, b_lo.y);
    DS_SPLIT2(m1, a.mn1, b_hi.z, b_lo.z);
    DS_SPLIT2(a.mx1, sd1, b_hi.w, b_lo.w);
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(vmax) : "v"(m0), "v"(m1));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(vmax) : "v"(sd0), "v"(sd1));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(vmax) : "v"(a.mn0), "v"(a.mn1));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(vmax) : "v"(a.mx0), "v"(a.mx1));
    asm volatile("" : "+v"(vmax));
}

// wmask (wave-uniform, once per tile: pna_walk_mask): bit w = every row of the wave has the four in-edges of word w, bit 4 + w = some row
// has one of them, bit 8 = some row has more than sixteen.  (Evaluated inside the walk, each __all / __any was a v_cndmask + v_cmp +
// scalar compare per word and K-step: 20 of a gather's ~150 VALU instructions -- and VALU issue is kernel time, tools/coissue4.hip.)
__device__ __forceinline__ int pna_walk_mask(int indeg) {
    int m = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (__all(indeg >= 4 * w + 4)) m |= 1 << w;
        if (__any(indeg > 4 * w)) m |= 16 << w;
    }
    if (__any(indeg > 16)) m |= 256;
    return __builtin_amdgcn_readfirstlane(m);
}
__device__ __forceinline__ void pna_gather_slice(const float* __restrict__ s_h, const uint8_t* __restrict__ s_src, const uint32_t (&srcw)[4],
                                                 int e_base, int indeg, int wmask, int col, ds_uint4_t& b_hi, ds_uint4_t& b_lo, float& vmax) {
    PnaSlice a{0.f, 0.f, 0.f, 0.f, PNA_SENT_MAX, PNA_SENT_MAX, PNA_SENT_MIN, PNA_SENT_MIN};
#ifdef PNA_GATHER_PRIO
    __builtin_amdgcn_s_setprio(PNA_GATHER_PRIO);
#endif
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (wmask & (1 << w)) {  // every row of the wave has these four in-edges (kNN graphs): four reads in flight, no masks
            float2 x[4];
#pragma unroll
            for (int b = 0; b < 4; b++) x[b] = *reinterpret_cast<const float2*>(s_h + (int)((srcw[w] >> (8 * b)) & 0xFFu) * PNA_FT_STRIDE + col);
#pragma unroll
            for (int b = 0; b < 4; b++) {  // sums in CSR order, with the same instructions as pna_slice_add (min / max are order-free: two edges per v_min3 / v_max3)
                a.S0 += x[b].x; a.S1 += x[b].y;
                a.Q0 = __builtin_fmaf(x[b].x, x[b].x, a.Q0); a.Q1 = __builtin_fmaf(x[b].y, x[b].y, a.Q1);
            }
#pragma unroll
            for (int b = 0; b < 4; b += 2) {
                asm("v_min3_f32 %0, %1, %2, %0" : "+v"(a.mn0) : "v"(x[b].x), "v"(x[b + 1].x));
                asm("v_min3_f32 %0, %1, %2, %0" : "+v"(a.mn1) : "v"(x[b].y), "v"(x[b + 1].y));
                asm("v_max3_f32 %0, %1, %2, %0" : "+v"(a.mx0) : "v"(x[b].x), "v"(x[b + 1].x));
                asm("v_max3_f32 %0, %1, %2, %0" : "+v"(a.mx1) : "v"(x[b].y), "v"(x[b + 1].y));
            }
        } else if (wmask & (16 << w)) {  // ragged: whole words are skipped when no row of the wave has them
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (indeg > 4 * w + b) pna_slice_edge(a, s_h, (int)((srcw[w] >> (8 * b)) & 0xFFu), col);
        }
    }
    if (wmask & 256)
        for (int e = 16; __any(e < indeg); e++)  // rows with more than 16 in-edges: the rest comes from the LDS copy of the CSR slice
            if (e < indeg) pna_slice_edge(a, s_h, (int)s_src[e_base + e], col);
    pna_slice_finish(a, indeg, b_hi, b_lo, vmax);
#ifdef PNA_GATHER_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

__global__ __launch_bounds__(PNA_FT_WAVES * 64, 4) void pna_layer_fused_kernel(const float* __restrict__ h, float* __restrict__ hout,
                                                                  const uint8_t* __restrict__ desc,
                                                                  const int* __restrict__ out_deg, const uint8_t* __restrict__ wpk,
                                                                  const float* __restrict__ bias, float avg_deg, float oscale,
                                                                  const int* __restrict__ tile_row, int n_tiles, int* __restrict__ range_flag,
                                                                  int ablate_arg, unsigned long long* __restrict__ prof_out) {
    const int ablate = FG_ABLATE(ablate_arg);  // 0 in the shipped build: the branches below fold away (common.h)
    (void)ablate_arg;
    (void)prof_out;
    // development aid (-DFLOWGNN_DEV, pna_ablate bit 128): per-wave phase times in 10 ns ticks -> prof_out[(workgroup, wave)][8]:
    // 0 wait for the tile's rows / chunk 0, 1 tile set-up, 2 gathers, 3 MFMA phases, 4 closing wait + barrier of the K-steps,
    // 5 epilogue, 6 whole kernel, 7 tiles
#ifdef FLOWGNN_DEV
    const bool prof = (ablate & 128) && prof_out;
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tp = 0, tk0 = 0, ntl = 0;
    if (prof) tk0 = tp = wall_clock64();
#define PNA_STAMP(i) do { if (prof) { const unsigned long long t_ = wall_clock64(); tacc[i] += t_ - tp; tp = t_; } } while (0)
#else
#define PNA_STAMP(i) do { } while (0)
#endif
    __shared__ __attribute__((aligned(16))) char s_a[PNA_CHUNK];  // even K-steps
    __shared__ __attribute__((aligned(16))) char s_b[PNA_CHUNK];  // odd K-steps
    __shared__ __attribute__((aligned(16))) float s_h[PNA_FT_ROWS * PNA_FT_STRIDE];
    __shared__ __attribute__((aligned(16))) char s_desc[2][PNA_DESC_BYTES];  // the tile's CSR slice (pna_tile_desc_kernel), double buffered
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    // Waves 0..7 gather a K-step's slice and then multiply; waves 8..15 multiply first (with the slice they gathered one interval
    // earlier) and then gather the next one: between two barriers half of the waves of every SIMD are in the matrix pipe while
    // the other half is in the VALU / LDS gather.
    // (development: pna_ablate bit 256 = no half-step offset, every wave gathers then multiplies; bit 512 = the offset by SIMD pairs --
    //  waves 2, 3, 6, 7, ... late -- instead of by workgroup halves)
    const bool late = (ablate & 256) ? false : (ablate & 512) ? ((wave >> 1) & 1) != 0 : wave >= PNA_FT_WAVES / 2;
    float vmax = 0.0f;
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    // tile descriptor of the first tile; later ones are fetched one tile ahead
    int t0 = tile_row[tile], rows = tile_row[tile + 1] - t0;
    if (rows > PNA_FT_ROWS) rows = PNA_FT_ROWS;
    int buf = 0;
    pna_issue_desc(desc, tile, s_desc[0], wave, lane);
    pna_issue_rows(h, t0, rows, s_h, wave, lane);
    while (true) {
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < n_tiles;
        int nt0 = 0, nrows = 0;
        if (has_next) {
            nt0 = tile_row[ntile];
            nrows = tile_row[ntile + 1] - nt0;
            if (nrows > PNA_FT_ROWS) nrows = PNA_FT_ROWS;
        }
        pna_issue_chunk_asm(wpk, s_a, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's rows and CSR slice, chunk 0
        __syncthreads();
        PNA_STAMP(0);
        const int r = wave * 16 + j;
        const bool valid = r < rows;
        const uint8_t* csrc = reinterpret_cast<const uint8_t*>(s_desc[buf]);
        const uint16_t* crp = reinterpret_cast<const uint16_t*>(s_desc[buf] + PNA_DESC_RP);
        const int e_base = valid ? (int)crp[r] : 0;
        int indeg = valid ? (int)crp[r + 1] - e_base : 0;
        if (ablate & 1) indeg = 0;  // development aid (pna_ablate, -DFLOWGNN_DEV builds): timing without the gather
        const int wmask = pna_walk_mask(indeg);
        uint32_t srcw[4];  // the first 16 in-edges, one byte each (re-walked by every K-step)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) v |= (4 * w + b < indeg ? (uint32_t)csrc[e_base + 4 * w + b] : 0u) << (8 * b);
            srcw[w] = v;
        }
        // per-node scalars of the epilogue (node_embedding.cc:148-150), requested early
        const long long node = (long long)t0 + (valid ? r : 0);
        const int odeg = out_deg[node];
        float4_t y[PNA_NS * PNA_OT];
#pragma unroll
        for (int i = 0; i < PNA_NS * PNA_OT; i++) y[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
        ds_uint4_t b_hi = {0, 0, 0, 0}, b_lo = {0, 0, 0, 0};
        PNA_STAMP(1);
        if (late) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 2 * g, b_hi, b_lo, vmax);  // K-step 0's slice, ahead of the first interval
        PNA_STAMP(2);
        // Development variants of WHERE a K-step's chunk request sits (round-3 finding: requested between a wave's two phases the
        // kernel was 1.6 % faster and two-engine runs stopped being bit-identical; scripts/dev/pna_dma_race.py bisects it):
        //   ablate 8: the request between the wave's phases (early waves: gather, REQUEST, multiply; late: multiply, REQUEST, gather)
        //   + 16: s_waitcnt lgkmcnt(0) in front of the request   + 32: a workgroup barrier in front of it
        //   + 64: a second barrier (and a short sleep) behind the closing vmcnt(0) + barrier
        const bool mid = PNA_DMA_MID || (ablate & 8) != 0;
        auto issue_mid = [&](const uint8_t* gchunk, char* buf) {
            if (ablate & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ablate & 32) __syncthreads();
            pna_issue_chunk_asm(gchunk, buf, wave, lane);
        };
        auto close_step = [&]() {
            PNA_STAMP(2);  // (a late wave's gather sits in front of the closing wait; an early wave's MFMA phase was stamped already)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(ablate & 4)) __syncthreads();  // ablate 4 (development aid): timing without the K-step barriers (results are then wrong)
            if (ablate & 64) {
                asm volatile("s_sleep 2" ::: "memory");
                __syncthreads();
            }
            PNA_STAMP(4);
        };
#pragma unroll 1
        for (int ks = 0; ks < ((ablate & 2) ? 0 : PNA_KS); ks += 2) {
            // even K-step from s_a while chunk ks+1 streams into s_b
            if (!mid) pna_issue_chunk_asm(wpk + (size_t)(ks + 1) * PNA_CHUNK, s_b, wave, lane);
            if (!late) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 8 * ks + 2 * g, b_hi, b_lo, vmax);
            if (mid && !late) issue_mid(wpk + (size_t)(ks + 1) * PNA_CHUNK, s_b);
            PNA_STAMP(2);
            if (ablate & 1024) __builtin_amdgcn_s_setprio(2);
            pna_stream_mfma(s_a, lane, b_hi, b_lo, y);
            if (ablate & 1024) __builtin_amdgcn_s_setprio(0);
#ifdef FLOWGNN_DEV
            if (prof) asm volatile("" :: "v"(y[14].x), "v"(y[13].x));  // the stamp behind the MFMAs' results, not behind their issue
#endif
            PNA_STAMP(3);
            if (mid && late) issue_mid(wpk + (size_t)(ks + 1) * PNA_CHUNK, s_b);
            if (late) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 8 * (ks + 1) + 2 * g, b_hi, b_lo, vmax);
            close_step();
            if (!mid && ks + 2 < PNA_KS) pna_issue_chunk_asm(wpk + (size_t)(ks + 2) * PNA_CHUNK, s_a, wave, lane);
            if (!late) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 8 * (ks + 1) + 2 * g, b_hi, b_lo, vmax);
            if (mid && !late && ks + 2 < PNA_KS) issue_mid(wpk + (size_t)(ks + 2) * PNA_CHUNK, s_a);
            PNA_STAMP(2);
            if (ablate & 1024) __builtin_amdgcn_s_setprio(2);
            pna_stream_mfma(s_b, lane, b_hi, b_lo, y);
            if (ablate & 1024) __builtin_amdgcn_s_setprio(0);
#ifdef FLOWGNN_DEV
            if (prof) asm volatile("" :: "v"(y[14].x), "v"(y[13].x));
#endif
            PNA_STAMP(3);
            if (mid && late && ks + 2 < PNA_KS) issue_mid(wpk + (size_t)(ks + 2) * PNA_CHUNK, s_a);
            if (late && ks + 2 < PNA_KS) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 8 * (ks + 2) + 2 * g, b_hi, b_lo, vmax);
            close_step();
        }
        // ---- epilogue: h' = h + relu(b + Y_0 + t Y_1 + scale Y_2)   (node_embedding.cc:148-150,205-213); the residual rows come
        // out of the LDS tile, after which the tile is dead and the next one's rows can stream in
        float4 hv[PNA_OT];
#pragma unroll
        for (int t = 0; t < PNA_OT; t++) hv[t] = *reinterpret_cast<const float4*>(s_h + (valid ? r : 0) * PNA_FT_STRIDE + 16 * t + 4 * g);
        __syncthreads();  // every wave has its residual rows (and is done gathering): s_h may be overwritten
        if (has_next) {
            pna_issue_desc(desc, ntile, s_desc[buf ^ 1], wave, lane);  // ... and its CSR slice, into the other small buffer
            pna_issue_rows(h, nt0, nrows, s_h, wave, lane);
        }
        if (valid) {
            const float logd = logf((float)(odeg + 1));
            const float sf_t = logd / avg_deg;
            const float sf_scale = (logd == 0.0f) ? 1.0f : avg_deg / logd;
            // the bias slices are requested TOGETHER, ahead of the stores: a load issued between two stores is waited for with
            // vmcnt(0), i.e. together with the store before it -- five serialized global round trips per tile
            float4 bb[PNA_OT];
#pragma unroll
            for (int t = 0; t < PNA_OT; t++) bb[t] = *reinterpret_cast<const float4*>(bias + 16 * t + 4 * g);
#pragma unroll
            for (int t = 0; t < PNA_OT; t++)
                *reinterpret_cast<float4*>(hout + (size_t)node * PNA_D + 16 * t + 4 * g) =
                    pna_update(hv[t], bb[t], y[0 * PNA_OT + t], y[1 * PNA_OT + t], y[2 * PNA_OT + t], oscale, sf_t, sf_scale);
        }
        PNA_STAMP(5);
#ifdef FLOWGNN_DEV
        ntl++;
#endif
        if (!has_next) break;
        tile = ntile; t0 = nt0; rows = nrows;
        buf ^= 1;
    }
#ifdef FLOWGNN_DEV
    if (prof && lane == 0) {
        unsigned long long* o = prof_out + ((size_t)blockIdx.x * PNA_FT_WAVES + wave) * 8;
        for (int i = 0; i < 6; i++) o[i] = tacc[i];
        o[6] = wall_clock64() - tk0;
        o[7] = ntl;
    }
#endif
#undef PNA_STAMP
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

// ---------------------------------------------------------------- graph-resident kernel: encoder + four layers + readout in one launch
// The FPGA holds one graph on chip from load_graph to the logit (PNA/src/PNA_compute.cc:46-100).  Here a persistent 16-wave
// workgroup holds a tile of WHOLE graphs (GraphTiles: <= 256 rows / 4 608 in-edges) in LDS across all four layers:
//   loader    h_0 = the nine-term encoder sum (load_inputs.cc:133-179), in the order of atom_encoder_kernel, straight into the
//             tile's LDS rows (the 55 KB table stays L2-resident);
//   layers    the K-step schedule of pna_layer_fused_kernel, the weight stream running on across the layers (40 chunks);
//             h' = h + relu(...) is written IN PLACE (a wave owns its sixteen rows, and every gather of the layer is behind a
//             barrier by then), so no row of h ever goes to HBM;
//   readout   mean pool + 80 -> 40 -> 20 -> 1 head (finalize.cc:34-52), one wave per graph of the tile, the association of
//             pool_mlp3_kernel (even rows | odd rows, then the two halves).
// HBM traffic per tile: 36 B of node features per row, the 6 KiB descriptor (CSR slice + out-degrees), one logit per graph.
// Same bits as the per-layer path (atom_encoder + 4 x pna_layer_fused + pool_mlp3): same operations in the same order.
struct PnaResidentArgs {
    const uint32_t* fidx;      // [rows][3]: nine encoder table rows per row as bytes, tile order (pna_tile_build_kernel), or null: ...
    const int* node_feature;   // ... [N][9], the caller's features (batch-order tiles only)
    const int* list;           // GraphTiles::bp_list / bp_lrow (bin-packed tiles), or null: tile t = the graphs tile_graph[t] ..
    const int* lrow;
    const float* nemb;         // [173][80]
    const uint8_t* desc;       // pna_tile_build_kernel / pna_tile_desc_kernel
    const uint8_t* wpk;        // feature-major weight stream, 4 layers x 10 chunks
    const float* bias;         // [4][80]
    const int* tile_row;       // GraphTiles::row_start
    const int* tile_graph;     // GraphTiles::graph_start
    const int* node_off;       // [G + 1]
    const float *w1t, *b1, *w2t, *b2, *w3, *b3;  // head, w1 / w2 transposed ([in][out]: coalesced over the output lanes)
    float* out;                // [G]
    int* range_flag;
    int* err;
    float avg_deg;
    float oscale[PNA_L];
    int n_tiles;
};

// development only (-DFLOWGNN_DEV -DPNAR_TIMING=<bits>, scripts/dev/variant.sh): TIMING variants that leave a phase out and compute wrong
// results on purpose -- 1 no encoder (h_0 = 0), 2 no readout, 4 no gathers, 8 no MFMAs.  Never part of the shipped library.
#if defined(FLOWGNN_DEV) && defined(PNAR_TIMING)
#define PNAR_SKIP(bit) ((PNAR_TIMING & (bit)) != 0)
#else
#define PNAR_SKIP(bit) false
#endif
__global__ __launch_bounds__(PNA_FT_WAVES * 64, 4) void pna_resident_kernel(const PnaResidentArgs a) {
    __shared__ __attribute__((aligned(16))) char s_a[PNA_CHUNK];  // even K-steps
    __shared__ __attribute__((aligned(16))) char s_b[PNA_CHUNK];  // odd K-steps; the readout's scratch between two tiles
    __shared__ __attribute__((aligned(16))) float s_h[PNA_FT_ROWS * PNA_FT_STRIDE];
    __shared__ __attribute__((aligned(16))) char s_desc[2][PNA_DESC_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const bool late = wave >= PNA_FT_WAVES / 2;  // the half-step offset of pna_layer_fused_kernel
    float vmax = 0.0f;
    int tile = blockIdx.x;
    if (tile >= a.n_tiles) return;
    int buf = 0;
    pna_issue_desc(a.desc, tile, s_desc[0], wave, lane);
    while (true) {
        const int t0 = a.tile_row[tile];
        int rows = a.tile_row[tile + 1] - t0;
        if (rows > PNA_FT_ROWS) rows = PNA_FT_ROWS;
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < a.n_tiles;
        pna_issue_chunk_asm(a.wpk, s_a, wave, lane);
        if (has_next) pna_issue_desc(a.desc, ntile, s_desc[buf ^ 1], wave, lane);  // lands under this tile's first K-step at the latest
        {   // ---- loader: h_0 rows of the tile (atom_encoder_kernel's sum, k = 0..8)
            constexpr int off[ND_FEATURE] = {0, 119, 123, 135, 147, 157, 163, 169, 171};  // load_inputs.cc:5
            constexpr int card[ND_FEATURE] = {119, 4, 12, 12, 10, 6, 6, 2, 2};            // host_load.cc:5
            const float4* tab = reinterpret_cast<const float4*>(a.nemb);
#pragma unroll 1
            for (int it = 0; it < (PNA_FT_ROWS * PNA_C) / (PNA_FT_WAVES * 64); it++) {
                const int item = (int)threadIdx.x + PNA_FT_WAVES * 64 * it;
                const int row = item / PNA_C, c = item - row * PNA_C;
                if (row < rows && PNAR_SKIP(1)) *reinterpret_cast<float4*>(s_h + row * PNA_FT_STRIDE + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < rows && !PNAR_SKIP(1)) {
                    int trow[ND_FEATURE];  // the nine table rows of this node
                    if (a.fidx) {
                        const uint32_t* fi = a.fidx + (size_t)(t0 + row) * 3;
                        const uint32_t f0 = fi[0], f1 = fi[1], f2 = fi[2];
#pragma unroll
                        for (int k = 0; k < ND_FEATURE; k++) trow[k] = (int)(((k < 4 ? f0 : (k < 8 ? f1 : f2)) >> (8 * (k & 3))) & 0xFFu);
                    } else {
                        const int* nf = a.node_feature + (size_t)(t0 + row) * ND_FEATURE;
                        int f[ND_FEATURE];
#pragma unroll
                        for (int k = 0; k < ND_FEATURE; k++) f[k] = nf[k];
#pragma unroll
                        for (int k = 0; k < ND_FEATURE; k++) {
                            int fk = f[k];
                            if (fk < 0 || fk >= card[k]) {
                                atomicMax(a.err, ERR_NODE_FEAT);
                                fk = 0;
                            }
                            trow[k] = off[k] + fk;
                        }
                    }
                    float4 w[ND_FEATURE];
#pragma unroll
                    for (int k = 0; k < ND_FEATURE; k++) w[k] = tab[trow[k] * PNA_C + c];
                    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < ND_FEATURE; k++) { s.x += w[k].x; s.y += w[k].y; s.z += w[k].z; s.w += w[k].w; }
                    *reinterpret_cast<float4*>(s_h + row * PNA_FT_STRIDE + 4 * c) = s;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's CSR slice, chunk 0 of layer 0
        __syncthreads();                                  // ... and every row of h_0
        const int r = wave * 16 + j;
        const bool valid = r < rows;
        const uint8_t* csrc = reinterpret_cast<const uint8_t*>(s_desc[buf]);
        const uint16_t* crp = reinterpret_cast<const uint16_t*>(s_desc[buf] + PNA_DESC_RP);
        const int e_base = valid ? (int)crp[r] : 0;
        const int indeg = (valid && !PNAR_SKIP(4)) ? (int)crp[r + 1] - e_base : 0;
        const int wmask = pna_walk_mask(indeg);
        uint32_t srcw[4];  // the first 16 in-edges, one byte each (re-walked by every K-step of every layer)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) v |= (4 * w + b < indeg ? (uint32_t)csrc[e_base + 4 * w + b] : 0u) << (8 * b);
            srcw[w] = v;
        }
        const int odeg = (int)reinterpret_cast<const uint16_t*>(s_desc[buf] + PNA_DESC_OD)[valid ? r : 0];
        float* own = s_h + (valid ? r : 0) * PNA_FT_STRIDE + 4 * g;
#pragma unroll 1
        for (int l = 0; l < PNA_L; l++) {
            const uint8_t* wl = a.wpk + (size_t)l * PNA_SPLIT_LAYER_BYTES;
            float4_t y[PNA_NS * PNA_OT];
#pragma unroll
            for (int i = 0; i < PNA_NS * PNA_OT; i++) y[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
            ds_uint4_t b_hi = {0, 0, 0, 0}, b_lo = {0, 0, 0, 0};
            if (late) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 2 * g, b_hi, b_lo, vmax);
#pragma unroll 1
            for (int ks = 0; ks < PNA_KS; ks += 2) {
                pna_issue_chunk_asm(wl + (size_t)(ks + 1) * PNA_CHUNK, s_b, wave, lane);
                if (!late) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 8 * ks + 2 * g, b_hi, b_lo, vmax);
                if (!PNAR_SKIP(8)) pna_stream_mfma(s_a, lane, b_hi, b_lo, y); else y[0][0] += __builtin_bit_cast(float, b_hi.x ^ b_lo.y);
                if (late) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 8 * (ks + 1) + 2 * g, b_hi, b_lo, vmax);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // chunk ks + 2 of this layer, or chunk 0 of the next one: the stream runs on across the layers
                if (ks + 2 < PNA_KS || l + 1 < PNA_L) pna_issue_chunk_asm(wl + (size_t)(ks + 2) * PNA_CHUNK, s_a, wave, lane);
                if (!late) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 8 * (ks + 1) + 2 * g, b_hi, b_lo, vmax);
                if (!PNAR_SKIP(8)) pna_stream_mfma(s_b, lane, b_hi, b_lo, y); else y[0][0] += __builtin_bit_cast(float, b_hi.x ^ b_lo.y);
                if (late && ks + 2 < PNA_KS) pna_gather_slice(s_h, csrc, srcw, e_base, indeg, wmask, 8 * (ks + 2) + 2 * g, b_hi, b_lo, vmax);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // ---- epilogue: h' = h + relu(b + Y_0 + t Y_1 + scale Y_2) in place (node_embedding.cc:148-150,205-213).  No wave gathers
            // any more (the layer's last barrier is behind us) and a wave writes its own sixteen rows only.
            const float logd = logf((float)(odeg + 1));
            const float sf_t = logd / a.avg_deg;
            const float sf_scale = (logd == 0.0f) ? 1.0f : a.avg_deg / logd;
            const float oscale = a.oscale[l];
            const float* bl = a.bias + l * PNA_D + 4 * g;
            float4 bb[PNA_OT];
#pragma unroll
            for (int t = 0; t < PNA_OT; t++) bb[t] = *reinterpret_cast<const float4*>(bl + 16 * t);
#pragma unroll
            for (int t = 0; t < PNA_OT; t++) {
                const float4 hv = *reinterpret_cast<const float4*>(own + 16 * t);
                const float4 hn = pna_update(hv, bb[t], y[0 * PNA_OT + t], y[1 * PNA_OT + t], y[2 * PNA_OT + t], oscale, sf_t, sf_scale);
                if (valid) *reinterpret_cast<float4*>(own + 16 * t) = hn;
            }
            __syncthreads();  // h' of every row is in place
        }
        if (PNAR_SKIP(2)) {
            if (threadIdx.x == 0) a.out[a.list ? a.list[a.tile_graph[tile]] : a.tile_graph[tile]] = s_h[0];
        } else {   // ---- readout: one wave per graph of the tile (pool_mlp3_kernel's association)
            const int g0 = a.tile_graph[tile], g1 = a.tile_graph[tile + 1];
            float* s_hg = reinterpret_cast<float*>(s_b) + wave * 128;  // [0, 80) pooled row, [80, 120) first hidden layer
            float* s_o1 = s_hg + PNA_D;
            const int half = lane >> 5, c = lane & 31;
            for (int gp = g0 + wave; gp < g1; gp += PNA_FT_WAVES) {
                // a range of graphs, or (bin-packed tiles) list positions: the graph's id and its first row inside the tile
                const int gi = a.list ? a.list[gp] : gp;
                const int n0 = a.list ? a.lrow[gp] : a.node_off[gi] - t0, n1 = n0 + (a.node_off[gi + 1] - a.node_off[gi]);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < PNA_C)
                    for (int v = n0 + half; v < n1; v += 2) {
                        const float4 x = *reinterpret_cast<const float4*>(s_h + v * PNA_FT_STRIDE + 4 * c);
                        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
                    }
                acc.x += __shfl_down(acc.x, 32, 64); acc.y += __shfl_down(acc.y, 32, 64);
                acc.z += __shfl_down(acc.z, 32, 64); acc.w += __shfl_down(acc.w, 32, 64);
                if (half == 0 && c < PNA_C) {
                    const float n = (float)(n1 - n0);
                    s_hg[4 * c + 0] = acc.x / n; s_hg[4 * c + 1] = acc.y / n;
                    s_hg[4 * c + 2] = acc.z / n; s_hg[4 * c + 3] = acc.w / n;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < 40) {
                    float s = a.b1[lane];
                    for (int i = 0; i < PNA_D; i++) s = __builtin_fmaf(s_hg[i], a.w1t[i * 40 + lane], s);
                    s_o1[lane] = relu1(s);
                }
                __builtin_amdgcn_wave_barrier();
                float part = 0.f;
                if (lane < 20) {
                    float s = a.b2[lane];
                    for (int i = 0; i < 40; i++) s = __builtin_fmaf(s_o1[i], a.w2t[i * 20 + lane], s);
                    part = relu1(s) * a.w3[lane];
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_down(part, d, 64);
                if (lane == 0) a.out[gi] = a.b3[0] + part;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!has_next) break;
        __syncthreads();  // the readout is done with the rows and with s_b: the next tile may load
        tile = ntile;
        buf ^= 1;
    }
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(a.range_flag, 1);
    }
}

// host: conv_w of one layer [80][3][4][80] -> the FEATURE-major weight stream of pna_layer_fused_kernel (same chunk geometry as
// pna_pack_split_layer: chunk ks = fragments of the 15 (scaler, output tile) pairs, hi then lo, 1 KiB each); K-slot e of lane
// group gk in K-step ks is (feature 8 ks + 2 gk + (e >> 2), aggregator e & 3).  The scale is the one pna_pack_split_layer returns.
static void pna_pack_stream_layer(const float* cw, uint8_t* out) {
    float m = 0.0f;
    for (size_t i = 0; i < (size_t)PNA_D * PNA_NS * PNA_NA * PNA_D; i++) m = std::fmax(m, std::fabs(cw[i]));
    const float sc = (m > 0.0f && std::isfinite(m)) ? std::ldexp(1.0f, -std::ilogb(m)) : 1.0f;
    for (int ks = 0; ks < PNA_KS; ks++)
        for (int s = 0; s < PNA_NS; s++)
            for (int t = 0; t < PNA_OT; t++)
                for (int lane = 0; lane < 64; lane++) {
                    const int i = lane & 15, gk = lane >> 4, o = 16 * t + i;
                    uint8_t* f = out + (size_t)ks * PNA_CHUNK + (size_t)((s * PNA_OT + t) * 2) * 1024;
                    for (int e = 0; e < 8; e++) {
                        const int feat = 8 * ks + 2 * gk + (e >> 2), a = e & 3;
                        const float v = cw[(((size_t)o * PNA_NS + s) * PNA_NA + a) * PNA_D + feat] * sc;
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        std::memcpy(f + lane * 16 + e * 2, &hi, 2);
                        std::memcpy(f + 1024 + lane * 16 + e * 2, &lo, 2);
                    }
                }
}

// host: conv_w of one layer [80][3][4][80] -> PNA_SPLIT_LAYER_BYTES; returns 1 / scale
static float pna_pack_split_layer(const float* cw, uint8_t* out) {
    float m = 0.0f;
    for (size_t i = 0; i < (size_t)PNA_D * PNA_NS * PNA_NA * PNA_D; i++) m = std::fmax(m, std::fabs(cw[i]));
    const float sc = (m > 0.0f && std::isfinite(m)) ? std::ldexp(1.0f, -std::ilogb(m)) : 1.0f;
    for (int ks = 0; ks < PNA_KS; ks++)
        for (int s = 0; s < PNA_NS; s++)
            for (int t = 0; t < PNA_OT; t++)
                for (int lane = 0; lane < 64; lane++) {
                    const int i = lane & 15, gk = lane >> 4, o = 16 * t + i;
                    uint8_t* f = out + (size_t)ks * PNA_CHUNK + (size_t)((s * PNA_OT + t) * 2) * 1024;
                    for (int e = 0; e < 8; e++) {
                        const int Q = 2 * ks + (e >> 2), a = Q / 5, q = Q % 5, k = 16 * q + 4 * gk + (e & 3);
                        const float v = cw[(((size_t)o * PNA_NS + s) * PNA_NA + a) * PNA_D + k] * sc;
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        std::memcpy(f + lane * 16 + e * 2, &hi, 2);
                        std::memcpy(f + 1024 + lane * 16 + e * 2, &lo, 2);
                    }
                }
    return 1.0f / sc;
}

class PnaModel : public Model {
public:
    ~PnaModel() override { free_all(); }
    int emb_dim() const override { return PNA_D; }
    int scratch_dim() const override { return PNA_D * PNA_NA; }
    int aggregate_dim() const override { return qmode_ ? 0 : PNA_D * PNA_NA; }  // fixed-point modes have no float aggregation kernel
    bool has_edge_attr() const override { return false; }
    int num_weight_tensors() const override { return 10; }
    bool weights_ready() const override { return ready_; }

    // host tensors (PNA/src/dcl.h:98-110): node_emb[173][80], conv_w[4][80][3][4][80] (out, scaler, aggr, in),
    // conv_b[4][80], mlp1_w[40][80], mlp1_b[40], mlp2_w[20][40], mlp2_b[20], mlp3_w[1][20], mlp3_b[1], avg_deg[1]
    int set_numeric_mode(int mode) override {
        if (mode != 0 && mode != 1) return 8;
        qmode_ = mode == 1;
        return 0;
    }

    int set_weights(const float* const* t) override {
        {   // ap_fixed<16,6> copies of every tensor for the bit-faithful mode (modelq.hip)
            static const size_t elems[10] = {173 * 80, 4 * 80 * 3 * 4 * 80, 4 * 80, 40 * 80, 40, 20 * 40, 20, 20, 1, 1};
            if (int rc = q_.upload_all(10, t, elems, 10)) return rc;
        }
        const float *nemb = t[0], *cw = t[1], *cb = t[2];
        std::vector<float> v_nemb(nemb, nemb + ND_FEATURE_TOTAL * PNA_D), v_cb(cb, cb + PNA_L * PNA_D);
        std::vector<float> v_w1(t[3], t[3] + 40 * 80), v_b1(t[4], t[4] + 40), v_w2(t[5], t[5] + 20 * 40), v_b2(t[6], t[6] + 20),
            v_w3(t[7], t[7] + 20), v_b3(t[8], t[8] + 1);
        avg_deg_ = t[9][0];
        std::vector<float> wf((size_t)PNA_L * PNA_NS * PNA_OT * PNA_NA * 5 * 64 * 4);
        for (int l = 0; l < PNA_L; l++)
            for (int s = 0; s < PNA_NS; s++)
                for (int tt = 0; tt < PNA_OT; tt++)
                    for (int a = 0; a < PNA_NA; a++)
                        for (int q = 0; q < 5; q++)
                            for (int lane = 0; lane < 64; lane++)
                                for (int r = 0; r < 4; r++) {
                                    const int i = lane & 15, g = lane >> 4;
                                    const int o = 16 * tt + i, k = 16 * q + 4 * g + r;
                                    wf[((((((size_t)l * PNA_NS + s) * PNA_OT + tt) * PNA_NA + a) * 5 + q) * 64 + lane) * 4 + r] =
                                        cw[((((size_t)l * PNA_D + o) * PNA_NS + s) * PNA_NA + a) * PNA_D + k];
                                }
        std::vector<uint8_t> split((size_t)PNA_L * PNA_SPLIT_LAYER_BYTES);
        for (int l = 0; l < PNA_L; l++)
            oscale_[l] = pna_pack_split_layer(cw + (size_t)l * PNA_D * PNA_NS * PNA_NA * PNA_D, split.data() + (size_t)l * PNA_SPLIT_LAYER_BYTES);
        std::vector<uint8_t> stream((size_t)PNA_L * PNA_SPLIT_LAYER_BYTES);
        for (int l = 0; l < PNA_L; l++)
            pna_pack_stream_layer(cw + (size_t)l * PNA_D * PNA_NS * PNA_NA * PNA_D, stream.data() + (size_t)l * PNA_SPLIT_LAYER_BYTES);
        int rc;
        if ((rc = upload(&d_split_, split))) return rc;
        if ((rc = upload(&d_stream_, stream))) return rc;
        if ((rc = upload(&d_nemb_, v_nemb))) return rc;
        if ((rc = upload(&d_wf_, wf))) return rc;
        if ((rc = upload(&d_cb_, v_cb))) return rc;
        if ((rc = upload(&d_w1_, v_w1))) return rc;
        if ((rc = upload(&d_b1_, v_b1))) return rc;
        if ((rc = upload(&d_w2_, v_w2))) return rc;
        if ((rc = upload(&d_b2_, v_b2))) return rc;
        if ((rc = upload(&d_w3_, v_w3))) return rc;
        if ((rc = upload(&d_b3_, v_b3))) return rc;
        {   // the resident kernel's head reads w1 / w2 transposed (lane = output unit)
            std::vector<float> w1t(80 * 40), w2t(40 * 20);
            for (int o = 0; o < 40; o++)
                for (int i = 0; i < 80; i++) w1t[i * 40 + o] = v_w1[o * 80 + i];
            for (int o = 0; o < 20; o++)
                for (int i = 0; i < 40; i++) w2t[i * 20 + o] = v_w2[o * 40 + i];
            if ((rc = upload(&d_w1t_, w1t))) return rc;
            if ((rc = upload(&d_w2t_, w2t))) return rc;
        }
        ready_ = true;
        return 0;
    }

    // PNA/src/host_load.cc:23-68,127: one file, hard-coded float offsets; avg_deg is a host constant
    int load_weights_dir(const char* dir) override {
        const char* f = "pna_ep1_noBN_dim80.weights.all.bin";
        std::vector<float> nemb(173 * 80), cw((size_t)4 * 76800), cb(4 * 80), w1(3200), b1(40), w2(800), b2(20), w3(20), b3(1);
        int rc;
        if ((rc = read_floats(dir, f, 0, nemb.size(), nemb.data()))) return rc;
        for (int l = 0; l < PNA_L; l++) {
            const size_t base = 13840 + 76880 * (size_t)l;
            if ((rc = read_floats(dir, f, base, 76800, &cw[(size_t)l * 76800]))) return rc;
            if ((rc = read_floats(dir, f, base + 76800, 80, &cb[l * 80]))) return rc;
        }
        if ((rc = read_floats(dir, f, 321360, 3200, w1.data()))) return rc;
        if ((rc = read_floats(dir, f, 324560, 40, b1.data()))) return rc;
        if ((rc = read_floats(dir, f, 324600, 800, w2.data()))) return rc;
        if ((rc = read_floats(dir, f, 325400, 20, b2.data()))) return rc;
        if ((rc = read_floats(dir, f, 325420, 20, w3.data()))) return rc;
        if ((rc = read_floats(dir, f, 325440, 1, b3.data()))) return rc;
        const float avg = 6.885701656341553f;  // host_load.cc:127
        const float* t[10] = {nemb.data(), cw.data(), cb.data(), w1.data(), b1.data(), w2.data(), b2.data(), w3.data(), b3.data(), &avg};
        return set_weights(t);
    }

    void launch_aggregate(const DeviceBatch& db, const float* hin, hipStream_t s) {
        PnaAggPolicy::Params prm{0};
        launch_tiled_aggregate<PnaAggPolicy>(prm, hin, db.scratch, db.csr, nullptr, db.b.n_tot, tiles_.p, tile_nominal_, s);
    }

    // fused layer kernel (pna_layer_fused_kernel): whole graphs packed into tiles of <= 256 rows / 4 608 in-edges by flowgnn_set_batch
    void graph_tile_limits(int& rows, int& edges) const override {
        rows = fused_ ? PNA_FT_ROWS : 0;
        edges = fused_ ? PNA_FT_EDGES : 0;
    }

    // tiles that are mostly empty (graphs of 65..128 nodes) waste MFMA columns: below 40 % full the two-kernel layer is used
    bool use_fused(const DeviceBatch& db) const {
        return fused_ && !qmode_ && split_ && !exact_ && db.gtiles.ok && db.gtiles.n_tiles > 0 && db.gtiles.fill >= 0.4;
    }
    // the graph-resident kernel: every layer's h stays in LDS, nothing per node is written (flowgnn_get_h repeats the pass per layer)
    bool use_resident(const DeviceBatch& db) const { return resident_ && !keep_h_ && use_fused(db); }
    // pna_tile_build=0 keeps the index build + pna_tile_desc_kernel in front of the resident kernel
    bool needs_csr(const DeviceBatch& db) const override { return !(tile_build_ && use_resident(db)); }
    void set_keep_h(bool on) override { keep_h_ = on; }
    bool wants_packed_tile_lists() const override { return binpack_ && tile_build_ && resident_ && fused_ && !qmode_; }

    int forward_resident(DeviceBatch& db, Profiler& prof, hipStream_t s) {
        // bin-packed tile lists when flowgnn_set_batch made them (option pna_binpack; tile-build path only): fewer, fuller tiles of the
        // same graphs.  A row's aggregates and a graph's pooling depend on the row / the graph alone: the same bits.
        const bool bp = binpack_ && tile_build_ && db.gtiles.bp_tiles > 0;
        const int* t_row = bp ? db.gtiles.bp_row : db.gtiles.row_start;
        const int* t_graph = bp ? db.gtiles.bp_graph : db.gtiles.graph_start;
        const int n_tiles = bp ? db.gtiles.bp_tiles : db.gtiles.n_tiles;
        if (int rc = desc_.reserve(((size_t)n_tiles * PNA_DESC_BYTES + 3) / 4)) return rc;
        if (tile_build_) {  // two launches per step: descriptors from the caller's arrays, then everything else (no CSR in HBM)
            if (int rc = fidx_.reserve((size_t)db.b.n_tot * 3)) return rc;
            ProfScope p(prof, "pna_tile_build", s);
            pna_tile_build_kernel<<<n_tiles, 256, 0, s>>>(db.b, t_row, t_graph, n_tiles, reinterpret_cast<uint8_t*>(desc_.p), db.csr.err,
                                                          bp ? db.gtiles.bp_list : nullptr, reinterpret_cast<uint32_t*>(fidx_.p));
        } else {
            ProfScope p(prof, "pna_tile_desc", s);
            pna_tile_desc_kernel<<<n_tiles, 256, 0, s>>>(db.csr.row_ptr, db.csr.src, db.csr.out_deg, t_row, reinterpret_cast<uint8_t*>(desc_.p), n_tiles);
        }
        PnaResidentArgs a;
        a.fidx = tile_build_ ? reinterpret_cast<const uint32_t*>(fidx_.p) : nullptr;
        a.list = bp ? db.gtiles.bp_list : nullptr;
        a.lrow = bp ? db.gtiles.bp_lrow : nullptr;
        a.node_feature = db.b.node_feature; a.nemb = d_nemb_; a.desc = reinterpret_cast<const uint8_t*>(desc_.p);
        a.wpk = d_stream_; a.bias = d_cb_;
        a.tile_row = t_row; a.tile_graph = t_graph; a.node_off = db.b.node_off;
        a.w1t = d_w1t_; a.b1 = d_b1_; a.w2t = d_w2t_; a.b2 = d_b2_; a.w3 = d_w3_; a.b3 = d_b3_;
        a.out = db.out; a.range_flag = db.range_flag; a.err = db.csr.err;
        a.avg_deg = avg_deg_;
        for (int l = 0; l < PNA_L; l++) a.oscale[l] = oscale_[l];
        a.n_tiles = n_tiles;
        const int grid = n_tiles < 256 ? n_tiles : 256;  // persistent: one 16-wave workgroup per CU
        {
            ProfScope p(prof, "pna_resident", s);
            pna_resident_kernel<<<grid, PNA_FT_WAVES * 64, 0, s>>>(a);
        }
        db.final_h = 0;
        db.h_valid = false;  // no per-node tensor leaves the kernel: flowgnn_get_h repeats the pass on the per-layer kernels
        return 0;
    }

    int forward(DeviceBatch& db, Profiler& prof, hipStream_t s) override {
        const int n = db.b.n_tot;
        if (n <= 0) return 0;
        if (qmode_) return pnaq_forward(q_, db, prof, s);
        db.h_valid = true;
        if (use_resident(db)) return forward_resident(db, prof, s);
        {
            ProfScope p(prof, "atom_encoder", s);
            atom_encoder_kernel<PNA_D><<<atom_encoder_grid(n, PNA_C), 512, 0, s>>>(db.b.node_feature, d_nemb_,
                                                                                                    db.h[0], n, db.csr.err);
        }
        if (int rc = make_tile_bounds(tiles_, db.b.node_off, db.b.num_graphs, n, tile_nominal_, tile_slack_, s)) return rc;
        int cur = 0;
        const bool fused = use_fused(db);
        for (int l = 0; l < PNA_L; l++) {
            if (fused) {
                ProfScope p(prof, "pna_layer_fused", s);
                const int grid = db.gtiles.n_tiles < 256 ? db.gtiles.n_tiles : 256;  // persistent: one 16-wave workgroup per CU (151 KB of LDS)
                if (l == 0) {  // the tiles' CSR slices as the layer kernel stages them, once for the four layers
                    if (int rc = desc_.reserve(((size_t)db.gtiles.n_tiles * PNA_DESC_BYTES + 3) / 4)) return rc;
                    pna_tile_desc_kernel<<<db.gtiles.n_tiles, 256, 0, s>>>(db.csr.row_ptr, db.csr.src, db.csr.out_deg, db.gtiles.row_start,
                                                                          reinterpret_cast<uint8_t*>(desc_.p), db.gtiles.n_tiles);
                }
                pna_layer_fused_kernel<<<grid, PNA_FT_WAVES * 64, 0, s>>>(db.h[cur], db.h[cur ^ 1], reinterpret_cast<const uint8_t*>(desc_.p), db.csr.out_deg,
                                                            d_stream_ + (size_t)l * PNA_SPLIT_LAYER_BYTES, d_cb_ + (size_t)l * PNA_D, avg_deg_,
                                                            oscale_[l], db.gtiles.row_start, db.gtiles.n_tiles, db.range_flag,
                                                            ablate_, prof_buf(grid, l));
                cur ^= 1;
                continue;
            }
            {
                ProfScope p(prof, "pna_aggregate", s);
                launch_aggregate(db, db.h[cur], s);
            }
            {
                ProfScope p(prof, "pna_dense", s);
                const int waves = (int)ceil_div_ll(n, 16);
                if (split_ && !exact_) {
                    pna_dense_split_kernel<<<(int)ceil_div_ll(n, 128), 512, 0, s>>>(db.scratch, db.h[cur], db.h[cur ^ 1], db.csr.out_deg,
                                                                                    d_split_ + (size_t)l * PNA_SPLIT_LAYER_BYTES,
                                                                                    d_cb_ + (size_t)l * PNA_D, avg_deg_, oscale_[l], n,
                                                                                    db.range_flag);
                } else
                pna_dense_kernel<<<(waves + 3) / 4, 256, 0, s>>>(db.scratch, db.h[cur], db.h[cur ^ 1], db.csr.out_deg,
                                                                  d_wf_ + (size_t)l * PNA_NS * PNA_OT * PNA_NA * 5 * 64 * 4,
                                                                  d_cb_ + (size_t)l * PNA_D, avg_deg_, n);
            }
            cur ^= 1;
        }
        db.final_h = cur;
        {
            ProfScope p(prof, "pool_mlp3", s);
            pool_mlp3_kernel<PNA_D, 40, 20><<<(db.b.num_graphs + 3) / 4, 256, 0, s>>>(db.h[cur], db.b.node_off, d_w1_, d_b1_, d_w2_,
                                                                                      d_b2_, d_w3_, d_b3_, db.out, db.b.num_graphs);
        }
        return 0;
    }

    void configure(const Options& o) override {
        tile_nominal_ = o.i("tile_nominal") > 0 ? o.i("tile_nominal") : kTileNominal;  // <= 0 / < 0: back to the model's defaults
        tile_slack_ = o.i("tile_slack") >= 0 ? o.i("tile_slack") : kTileSlack;
        split_ = o.i("pna_mfma") != 32;
        fused_ = o.on("pna_fused");
        resident_ = o.on("pna_resident");
        tile_build_ = o.on("pna_tile_build");
        binpack_ = o.on("pna_binpack");
        ablate_ = FG_ABLATE(o.i("pna_ablate"));
    }
    void set_exact(bool on) override { exact_ = on; }

    int aggregation_only(DeviceBatch& db, int layer, hipStream_t s) override {
        if (qmode_) return 8;  // FLOWGNN_ERR_UNSUPPORTED: the fixed-point forward never builds the float kernels' inputs (tiles, h rows)
        if (layer < 0 || layer >= PNA_L) return 1;
        launch_aggregate(db, db.h[db.final_h], s);
        return 0;
    }

    // development aid (-DFLOWGNN_DEV, pna_ablate bit 128): phase times of the LAST layer launch of a pass, printed by the next one
    unsigned long long* prof_buf(int grid, int layer) {
#ifdef FLOWGNN_DEV
        if (!(ablate_ & 128)) return nullptr;
        const size_t n = (size_t)256 * PNA_FT_WAVES * 8;
        if (!d_prof_) { if (hipMalloc((void**)&d_prof_, n * 8) != hipSuccess) return nullptr; (void)hipMemset(d_prof_, 0, n * 8); }
        if (layer == 1) {  // layer 0's record is complete by stream order once we synchronise: read it, print the per-phase means
            std::vector<unsigned long long> h(n);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), d_prof_, n * 8, hipMemcpyDeviceToHost);
            double acc[8] = {0}, early[8] = {0}, late[8] = {0};
            int cnt = 0;
            for (int wg = 0; wg < grid; wg++)
                for (int w = 0; w < PNA_FT_WAVES; w++) {
                    const unsigned long long* o = &h[((size_t)wg * PNA_FT_WAVES + w) * 8];
                    if (!o[7]) continue;
                    for (int i = 0; i < 8; i++) { acc[i] += (double)o[i]; (w < 8 ? early : late)[i] += (double)o[i]; }
                    cnt++;
                }
            if (cnt) {
                static const char* nm[8] = {"rows_wait", "setup", "gather", "mfma", "close_wait", "epilogue", "kernel", "tiles"};
                fprintf(stderr, "pna phase stamps (us per wave, mean over %d waves; early waves | late waves):", cnt);
                for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.1f (%.1f | %.1f)", nm[i], acc[i] / cnt * (i < 7 ? 0.01 : 1.0), early[i] / (cnt / 2) * (i < 7 ? 0.01 : 1.0), late[i] / (cnt / 2) * (i < 7 ? 0.01 : 1.0));
                fprintf(stderr, "\n");
            }
        }
        return layer == 0 ? d_prof_ : nullptr;
#else
        (void)grid; (void)layer;
        return nullptr;
#endif
    }

private:
    unsigned long long* d_prof_ = nullptr;
    void free_all() {
        if (d_prof_) { (void)hipFree(d_prof_); d_prof_ = nullptr; }

        float** ptrs[] = {&d_nemb_, &d_wf_, &d_cb_, &d_w1_, &d_b1_, &d_w2_, &d_b2_, &d_w3_, &d_b3_, &d_w1t_, &d_w2t_};
        for (auto p : ptrs)
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        if (d_split_) { (void)hipFree(d_split_); d_split_ = nullptr; }
        if (d_stream_) { (void)hipFree(d_stream_); d_stream_ = nullptr; }
        tiles_.release();
        desc_.release();
        fidx_.release();
        q_.release();
    }
    bool ready_ = false;
    bool qmode_ = false;  // flowgnn_set_numeric_mode(FLOWGNN_NUMERIC_Q6_10)
    QPack q_;
    GrowBufI tiles_;  // graph-aligned tile starts of the resident batch (tile_bounds_kernel)
    GrowBufI desc_;   // pna_tile_desc_kernel: 5.5 KiB per graph tile
    GrowBufI fidx_;   // pna_tile_build_kernel: nine encoder table rows per node as bytes, tile order (12 B per node)
    bool binpack_ = true;  // pna_binpack: the resident kernel walks bin-packed tile lists (GraphTiles::bp_*)
    static constexpr int kTileNominal = 112, kTileSlack = 48;  // the model's defaults of the options tile_nominal / tile_slack
    int tile_nominal_ = kTileNominal, tile_slack_ = kTileSlack;
    // pna_mfma=32 keeps the dense update on the fp32 matrix pipe (pna_dense_kernel)
    bool split_ = true;
    // pna_fused=0 keeps aggregation and dense update as two kernels per layer (A/B measurements, the aggregation roofline probe)
    int ablate_ = 0;  // development aid (-DFLOWGNN_DEV builds only, option pna_ablate): per-phase timing (scripts/dev/pna_ablate.sh)
    bool fused_ = true;
    // pna_resident=0 keeps one launch per layer (encoder, four fused layers, readout) with h through HBM
    bool resident_ = true, tile_build_ = true, keep_h_ = false;
    bool exact_ = false;
    uint8_t* d_split_ = nullptr;
    uint8_t* d_stream_ = nullptr;  // feature-major weight stream of the fused layer kernel
    float oscale_[PNA_L] = {1.f, 1.f, 1.f, 1.f};
    float avg_deg_ = 1.0f;
    float *d_nemb_ = nullptr, *d_wf_ = nullptr, *d_cb_ = nullptr, *d_w1_ = nullptr, *d_b1_ = nullptr, *d_w2_ = nullptr,
          *d_b2_ = nullptr, *d_w3_ = nullptr, *d_b3_ = nullptr, *d_w1t_ = nullptr, *d_w2t_ = nullptr;
};

Model* make_pna_model() { return new PnaModel(); }

}  // namespace fg
