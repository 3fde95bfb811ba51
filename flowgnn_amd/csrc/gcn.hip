// gcn.hip -- GCN hot path for gfx950 (MI355X).
//
// Reference per graph (GCN/src/*.cc), x_l = output of the NT unit of layer l:
//   h0[v]   = sum_{k<9} NodeEmb[off_k + feat_k(v)]                                load_inputs.cc:168-215
//   a_0     = h0;   a_l = relu(BN_{l-1}(m_{l-1}[v] + relu(x_{l-1}[v] + root_{l-1}) / (deg(v)+1)))   node_embedding.cc:123-138
//   x_l     = b_l + W_l a_l   (100x100)                                           node_embedding.cc:140-146
//   m_l[v]  = sum_{(u->v)} norm(u,v) relu(x_l[u] + sum_k EdgeEmb_l[off_k+attr_k])   message_passing.cc:158-167
//   norm    = dinv[u] dinv[v], dinv[i] = outdeg(i) > 0 ? 1/sqrt(outdeg(i)+1) : 0   load_inputs.cc:122,163
//   out[g]  = pb + pw . mean_v BN_4(m_4[v] + relu(x_4[v] + root_4)/(deg(v)+1))       finalize.cc:79-113
//   deg(v) is the OUT-degree table of load_graph (load_inputs.cc:120), BN is eval-mode with
//   sqrt(var + 2^-10) (load_inputs.cc:32).
//
// Here, on the batched super-graph: per layer one HBM-bound aggregation kernel that also applies the
// root / degree / BatchNorm / ReLU epilogue (so a_l is written once), and one fp32-MFMA dense kernel.
#include "common.h"
#include "device_common.h"
#include "modelq.h"
#include "dense_split.h"
#include "gin_split.h"  // gin_resident_pack_enc_table: the pre-combined (three rows per node) form of a 173-row table
#include <cmath>
#include <cstring>

namespace fg {

constexpr int GCN_D = 100;
constexpr int GCN_L = 5;
constexpr int GCN_C = GCN_D / 4;
constexpr int GCN_OT = 7;


// a[v] = epilogue( sum_e norm_e relu(x[src_e] + ecomb[code_e]), x[v], outdeg[v] ), CSR order: policy of the generic
// tiled aggregation (device_common.h).  norm_e = dinv[u] dinv[v] with dinv from the out-degree table.
template <bool RELU_OUT>
struct GcnAggPolicy {
    // TE = 448 staged CSR entries (a molpcba tile has ~300): 81.6 KB of LDS, i.e. TWO workgroups per CU; with 1024 it was one
    static constexpr int D = GCN_D, TR = 128, NTHR = 512, TE = 448, TABLE_ROWS = EDGE_COMBOS;
    static constexpr bool HAS_SCALAR = true;
    static constexpr int NDST = 2;                 // dinv[v], 1 / (outdeg(v) + 1)
    static constexpr int CONST_FLOATS = 3 * GCN_D;  // root | folded BN scale | folded BN shift of the layer
    struct Params {
        const int* out_deg;
        const float* ep;   // the three epilogue vectors, contiguous
        const float* esc;  // [E] dinv[src_e] in CSR order (edge_scalar_kernel)
    };
    struct Acc { float4 m; };
    __device__ static float dinv(int d) { return d > 0 ? 1.0f / sqrtf((float)(d + 1)) : 0.0f; }  // load_inputs.cc:122
    __device__ static float src_scalar(const Params& p, int u) { return dinv(p.out_deg[u]); }
    __device__ static void dst_stage(const Params& p, int v, float* o) {
        const int d = p.out_deg[v];
        o[0] = dinv(d);
        o[1] = 1.0f / (float)(d + 1);
    }
    __device__ static const float* const_ptr(const Params& p) { return p.ep; }
    __device__ static void init(Acc& a) { a.m = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ static void edge(Acc& a, const float4& x, const float4& w, float ss, const float* sd) {
        const float norm = ss * sd[0];
        a.m.x += norm * relu1(w.x + x.x); a.m.y += norm * relu1(w.y + x.y);
        a.m.z += norm * relu1(w.z + x.z); a.m.w += norm * relu1(w.w + x.w);
    }
    __device__ static void finish(const Params&, const Acc& a, const float4& xs, int v, int c, int, const float* sd,
                                  const float* cst, float* out) {
        // BatchNorm folded on the host: (t - mean) / sqrt(var + 2^-10) * w + b  ==  t * scale + shift (one FMA instead of
        // a subtract, an IEEE divide, a multiply and an add); 1 / (deg + 1) is staged per row.
        const float4 rt = reinterpret_cast<const float4*>(cst)[c];
        const float4 sc = reinterpret_cast<const float4*>(cst + GCN_D)[c];
        const float4 sh = reinterpret_cast<const float4*>(cst + 2 * GCN_D)[c];
        const float idp1 = sd[1];
        float4 r;
        r.x = (a.m.x + relu1(xs.x + rt.x) * idp1) * sc.x + sh.x;
        r.y = (a.m.y + relu1(xs.y + rt.y) * idp1) * sc.y + sh.y;
        r.z = (a.m.z + relu1(xs.z + rt.z) * idp1) * sc.z + sh.z;
        r.w = (a.m.w + relu1(xs.w + rt.w) * idp1) * sc.w + sh.w;
        if (RELU_OUT) { r.x = relu1(r.x); r.y = relu1(r.y); r.z = relu1(r.z); r.w = relu1(r.w); }
        stream_store4(reinterpret_cast<float4*>(out) + (size_t)v * GCN_C + c, r);
    }
};

// x_0 = W_0 h0 + b_0 with h0 = atom encoder output computed on the fly: the 173-row embedding table (69 KB) and the
// layer's weight fragments (45 KB) both live in LDS, a lane sums the nine table rows of its node for the 25 features it
// owns (same order as atom_encoder_kernel, so h0 is bit-identical) and feeds them straight to the split MFMAs.  Saves
// the 2.8 GB write + 2.8 GB read of h0.  One persistent 16-wave workgroup per CU, no barrier after the initial fill.
__global__ __launch_bounds__(1024) void gcn_encoder_dense_kernel(const int* __restrict__ node_feature, const float* __restrict__ table,
                                                                  float* __restrict__ xout, const uint8_t* __restrict__ wpk, int n_tot,
                                                                  int* __restrict__ err, int* __restrict__ range_flag) {
    constexpr int OT = GCN_OT;
    constexpr int WBYTES = (int)dense100_split_bytes(OT);
    constexpr int TAIL_OFF = OT * 6 * 1024, BIAS_OFF = TAIL_OFF + OT * 256, SCALE_OFF = BIAS_OFF + OT * 64;
    __shared__ __attribute__((aligned(16))) char s_w[WBYTES];
    __shared__ __attribute__((aligned(16))) float s_tab[ND_FEATURE_TOTAL * GCN_D];
    for (int i = threadIdx.x; i < WBYTES / 16; i += 1024) reinterpret_cast<float4*>(s_w)[i] = reinterpret_cast<const float4*>(wpk)[i];
    for (int i = threadIdx.x; i < ND_FEATURE_TOTAL * GCN_C; i += 1024)
        reinterpret_cast<float4*>(s_tab)[i] = reinterpret_cast<const float4*>(table)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const float oscale = *reinterpret_cast<const float*>(s_w + SCALE_OFF);
    const long long n_tiles = ((long long)n_tot + 15) / 16;
    float vmax = 0.0f;
    // the feature row of this lane's node, requested one tile ahead of its use (the wave's only global round trip per tile)
    int fnext[ND_FEATURE];
    auto fetch = [&](long long t) {
        long long v = t * 16 + j;
        if (v >= n_tot) v = n_tot - 1;
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) fnext[k] = node_feature[(size_t)v * ND_FEATURE + k];
    };
    fetch((long long)blockIdx.x * 16 + wave < n_tiles ? (long long)blockIdx.x * 16 + wave : 0);
    for (long long tile = (long long)blockIdx.x * 16 + wave; tile < n_tiles; tile += (long long)gridDim.x * 16) {
        long long node = tile * 16 + j;
        const bool valid = node < n_tot;
        if (!valid) node = n_tot - 1;
        int rows[ND_FEATURE];
        int fcur[ND_FEATURE];
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) fcur[k] = fnext[k];
        {
            const long long tn = tile + (long long)gridDim.x * 16;
            fetch(tn < n_tiles ? tn : tile);
        }
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) {
            int f = fcur[k];
            if (f < 0 || f >= c_nd_card[k]) {
                atomicMax(err, ERR_NODE_FEAT);
                f = 0;
            }
            rows[k] = (c_nd_off[k] + f) * GCN_D;
        }
        float a[25];
#pragma unroll
        for (int i = 0; i < 25; i++) a[i] = 0.0f;
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) {
            const float* tr = s_tab + rows[k] + 4 * g;
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const float4 w = *reinterpret_cast<const float4*>(tr + 16 * q);
                a[4 * q + 0] += w.x; a[4 * q + 1] += w.y; a[4 * q + 2] += w.z; a[4 * q + 3] += w.w;
            }
            a[24] += s_tab[rows[k] + 96 + g];
        }
        ds_uint4_t b_hi[3], b_lo[3];
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            DS_SPLIT2(a[8 * ks + 0], a[8 * ks + 1], b_hi[ks].x, b_lo[ks].x);
            DS_SPLIT2(a[8 * ks + 2], a[8 * ks + 3], b_hi[ks].y, b_lo[ks].y);
            DS_SPLIT2(a[8 * ks + 4], a[8 * ks + 5], b_hi[ks].z, b_lo[ks].z);
            DS_SPLIT2(a[8 * ks + 6], a[8 * ks + 7], b_hi[ks].w, b_lo[ks].w);
        }
#pragma unroll
        for (int k = 0; k < 24; k += 2)
            vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(a[k])), __builtin_fabsf(a[k + 1]));
        asm volatile("" : "+v"(vmax));
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
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at, a[24], acc, 0, 0, 0);
            const int col = 16 * t + 4 * g;
            if (col < GCN_D && valid) {
                const float4_t r = acc * oscale;
                *reinterpret_cast<float4*>(xout + (size_t)node * GCN_D + col) = make_float4(r.x, r.y, r.z, r.w);
            }
        }
    }
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

// One GCN layer in one kernel: a = relu(BN(aggregate(x))) as in GcnAggPolicy<true>, then x' = W a + b on the f16 matrix pipe
// (split products, dense_split.h) -- the aggregate never goes to HBM (2 x 2.8 GB per layer at 2^18 molpcba graphs).
// Everything the layer needs besides the rows fits LDS for the whole kernel (45 KB of weight fragments, 24 KB of
// edge-embedding combos, the three epilogue vectors), so after the initial fill there is NO barrier: persistent
// workgroups, each wave walks its own 16-node tiles -- gather in the B-operand register layout (lane (j, g) owns node j's
// features 16 q + 4 g .. +3 and 96 + g), CSR entries and per-edge norms one edge ahead of the row loads, 70 MFMAs, stores --
// and the sixteen waves of a CU drift freely against each other instead of marching through barrier-delimited steps.
// FINAL: the last stage has no dense layer and no ReLU; the readout's linear head is applied per node instead
// (mean_v(a[v]) . w = mean_v(a[v] . w)) and xout is a float[n_tot] of per-node scores for segment_mean_bias_kernel.
template <bool FINAL>
__global__ __launch_bounds__(512) void gcn_layer_fused_kernel(const float* __restrict__ x, float* __restrict__ xout,
                                                               const int* __restrict__ row_ptr, const int* __restrict__ src,
                                                               const uint8_t* __restrict__ ecode, const float* __restrict__ esc,
                                                               const int* __restrict__ out_deg, const float* __restrict__ ecomb,
                                                               const float* __restrict__ ep, const uint8_t* __restrict__ wpk, int n_tot,
                                                               int* __restrict__ range_flag, const float* __restrict__ pool_w) {
    constexpr int OT = GCN_OT;
    constexpr int WBYTES = (int)dense100_split_bytes(OT);
    constexpr int TAIL_OFF = OT * 6 * 1024, BIAS_OFF = TAIL_OFF + OT * 256, SCALE_OFF = BIAS_OFF + OT * 64;
    __shared__ __attribute__((aligned(16))) char s_w[WBYTES];
    __shared__ __attribute__((aligned(16))) float s_ecomb[EDGE_COMBOS * GCN_D];
    __shared__ __attribute__((aligned(16))) float s_ep[3 * GCN_D];
    if (!FINAL)
        for (int i = threadIdx.x; i < WBYTES / 16; i += 512) reinterpret_cast<float4*>(s_w)[i] = reinterpret_cast<const float4*>(wpk)[i];
    for (int i = threadIdx.x; i < EDGE_COMBOS * GCN_C; i += 512)
        reinterpret_cast<float4*>(s_ecomb)[i] = reinterpret_cast<const float4*>(ecomb)[i];
    for (int i = threadIdx.x; i < 3 * GCN_D; i += 512) s_ep[i] = ep[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const float oscale = FINAL ? 1.0f : *reinterpret_cast<const float*>(s_w + SCALE_OFF);
    const long long n_tiles = ((long long)n_tot + 15) / 16;
    float vmax = 0.0f;
    for (long long tile = (long long)blockIdx.x * 8 + wave; tile < n_tiles; tile += (long long)gridDim.x * 8) {
        long long node = tile * 16 + j;
        const bool valid = node < n_tot;
        if (!valid) node = n_tot - 1;
        int e = row_ptr[node];
        const int e_end = valid ? row_ptr[node + 1] : e;
        const int dv = out_deg[node];
        const float* xr = x + (size_t)node * GCN_D + 4 * g;
        float4 xs[6];
#pragma unroll
        for (int q = 0; q < 6; q++) xs[q] = *reinterpret_cast<const float4*>(xr + 16 * q);
        const float xst = x[(size_t)node * GCN_D + 96 + g];
        const float dinv_v = dv > 0 ? 1.0f / sqrtf((float)(dv + 1)) : 0.0f;  // load_inputs.cc:122
        const float idp1 = 1.0f / (float)(dv + 1);
        float m[25];
#pragma unroll
        for (int k = 0; k < 25; k++) m[k] = 0.0f;
        int u_nx = 0, c_nx = 0;
        float n_nx = 0.0f;
        if (e < e_end) { u_nx = src[e]; c_nx = ecode[e]; n_nx = esc[e]; }
        while (__any(e < e_end)) {
            if (e < e_end) {
                const int u = u_nx, code = c_nx;
                const float norm = n_nx * dinv_v;
                e++;
                if (e < e_end) { u_nx = src[e]; c_nx = ecode[e]; n_nx = esc[e]; }
                const float* hr = x + (size_t)u * GCN_D + 4 * g;
                const float* er = s_ecomb + code * GCN_D + 4 * g;
                float4 xv[6];
#pragma unroll
                for (int q = 0; q < 6; q++) xv[q] = *reinterpret_cast<const float4*>(hr + 16 * q);
                const float xt = x[(size_t)u * GCN_D + 96 + g];
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const float4 w = *reinterpret_cast<const float4*>(er + 16 * q);
                    m[4 * q + 0] += norm * relu1(w.x + xv[q].x);
                    m[4 * q + 1] += norm * relu1(w.y + xv[q].y);
                    m[4 * q + 2] += norm * relu1(w.z + xv[q].z);
                    m[4 * q + 3] += norm * relu1(w.w + xv[q].w);
                }
                m[24] += norm * relu1(s_ecomb[code * GCN_D + 96 + g] + xt);
            }
        }
        // epilogue of the aggregation (GcnAggPolicy<true>::finish): root term, folded BatchNorm, ReLU
        // (this lane's 75 epilogue constants and, in the last layer, its 25 readout weights are the same for every tile: with
        //  the lane index left visible the compiler keeps all of them in registers across the tile loop -- 232 VGPRs and two
        //  waves per SIMD in the last layer -- so the index is made opaque per tile and the values are re-read from LDS / L1)
        int ge = g;
        if (FINAL) asm volatile("" : "+v"(ge));  // (the other layers stay under 128 registers as they are, and are faster with what the compiler keeps)
        float a[25];
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 rt = *reinterpret_cast<const float4*>(s_ep + 16 * q + 4 * ge);
            const float4 sc = *reinterpret_cast<const float4*>(s_ep + GCN_D + 16 * q + 4 * ge);
            const float4 sh = *reinterpret_cast<const float4*>(s_ep + 2 * GCN_D + 16 * q + 4 * ge);
            a[4 * q + 0] = (m[4 * q + 0] + relu1(xs[q].x + rt.x) * idp1) * sc.x + sh.x;
            a[4 * q + 1] = (m[4 * q + 1] + relu1(xs[q].y + rt.y) * idp1) * sc.y + sh.y;
            a[4 * q + 2] = (m[4 * q + 2] + relu1(xs[q].z + rt.z) * idp1) * sc.z + sh.z;
            a[4 * q + 3] = (m[4 * q + 3] + relu1(xs[q].w + rt.w) * idp1) * sc.w + sh.w;
        }
        a[24] = (m[24] + relu1(xst + s_ep[96 + ge]) * idp1) * s_ep[GCN_D + 96 + ge] + s_ep[2 * GCN_D + 96 + ge];
        if (FINAL) {  // no ReLU after the last BatchNorm; per-node score, fixed order: 25 terms in the lane, then the node's 4 lanes
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const float4 pw = *reinterpret_cast<const float4*>(pool_w + 16 * q + 4 * ge);
                part += a[4 * q + 0] * pw.x; part += a[4 * q + 1] * pw.y; part += a[4 * q + 2] * pw.z; part += a[4 * q + 3] * pw.w;
            }
            part += a[24] * pool_w[96 + ge];
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (g == 0 && valid) xout[node] = part;
            continue;
        }
#pragma unroll
        for (int k = 0; k < 25; k++) a[k] = relu1(a[k]);
        // dense layer on the split operands
        ds_uint4_t b_hi[3], b_lo[3];
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            DS_SPLIT2(a[8 * ks + 0], a[8 * ks + 1], b_hi[ks].x, b_lo[ks].x);
            DS_SPLIT2(a[8 * ks + 2], a[8 * ks + 3], b_hi[ks].y, b_lo[ks].y);
            DS_SPLIT2(a[8 * ks + 4], a[8 * ks + 5], b_hi[ks].z, b_lo[ks].z);
            DS_SPLIT2(a[8 * ks + 6], a[8 * ks + 7], b_hi[ks].w, b_lo[ks].w);
        }
#pragma unroll
        for (int k = 0; k < 24; k += 2) vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, a[k]), a[k + 1]);  // a >= 0 (ReLU)
        asm volatile("" : "+v"(vmax));
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
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at, a[24], acc, 0, 0, 0);
            const int col = 16 * t + 4 * g;
            if (col < GCN_D && valid) {
                const float4_t r = acc * oscale;
                *reinterpret_cast<float4*>(xout + (size_t)node * GCN_D + col) = make_float4(r.x, r.y, r.z, r.w);
            }
        }
    }
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

// ---------------------------------------------------------------- graph-resident GCN: the five aggregations + four dense layers + readout in ONE launch
// The counterpart of gin_resident_kernel (gin_split.hip) and of the FPGA keeping one graph in BRAM across its layer loop
// (GCN/src/GCN_compute.cc): a persistent 12-wave workgroup (one per CU) owns a tile of WHOLE graphs (GraphTiles: <= 192 rows,
// <= 960 in-edges) and keeps the tile's x rows in LDS across all layers:
//   * x_l never goes to HBM between layers: the gather reads neighbour rows from LDS, the dense layer writes x_{l+1} in place;
//   * the tile's CSR slice is staged once per tile as 16-bit words (row inside the tile << 6 | edge code); dinv and 1 / (deg + 1)
//     of the tile's rows are computed once per tile, so the per-edge norm is two LDS reads and a multiply;
//   * the two weight regions are never needed at the same time: W_{l+1} (45 KiB of split fragments) streams L2 -> LDS behind gather l
//     (requested when a wave's walk is done), the next layer's edge-embedding table + epilogue vectors (25 KiB) while dense l + 1 runs;
//   * per node the launch reads 400 B (x_0, from gcn_encoder_dense_kernel) + 5 B per in-edge + 8 B of row bounds / degree, and
//     writes 4 B per GRAPH.
// LDS: 76 800 (rows) + 46 080 (W) + 25 600 (table + epilogue) + 1 920 (edges) + ~2 700 = 153 KB.
constexpr int GCNR_ROWS = 192;
constexpr int GCNR_EDGES = 960;
constexpr int GCNR_WAVES = 12;
constexpr int GCNR_W_BYTES = 45 * 1024;     // dense100_split_bytes(GCN_OT) = 45 264, padded to whole 1 KiB DMA pieces
constexpr int GCNR_BLOB_BYTES = 25 * 1024;  // ecomb [60][100] | root | BN scale | BN shift (25 200 B)
constexpr int GCNR_LAYER_BYTES = GCNR_W_BYTES + GCNR_BLOB_BYTES;
// The edge-embedding table a second time, for the walk's reads through the vector-memory path: planes [quad q of 6][quarter g][code, padded
// to 64] of 16 B -- the sixteen lanes of a quarter wave (one q, one g, sixteen codes) fall into one plane of 1 KiB = 8 cache lines.
constexpr int GCNR_PLANES_OFF = GCN_L * GCNR_LAYER_BYTES + 4096, GCNR_PLANES_BYTES = 6 * 4 * 64 * 16;
constexpr int GCNR_EQ_VMEM = 4;  // quads of a table row the walk reads through L1 (the other two and the tail from LDS)
static_assert(dense100_split_bytes(GCN_OT) <= (size_t)GCNR_W_BYTES, "weight region");
static_assert((EDGE_COMBOS + 3) * GCN_D * 4 <= GCNR_BLOB_BYTES, "table region");

// ---------------------------------------------------------------- one-pass front end (round 5): the tile's descriptor straight from the caller's arrays
// gcn_tile_build_kernel = load_graph (GCN/src/load_inputs.cc:120-166: in-edge tables, out-degree table) + the index part of the atom
// encoder (:168-215) for ONE tile of whole graphs, as gin_tile_build_kernel is for GIN (gin_split.hip): no global CSR, no x_0 rows in
// HBM, no separate encoder / index-build launches (0.57 + 0.17 ms of a 5.6 ms step at 2^18 molpcba graphs; the FPGA keeps all of it on
// chip, GCN/src/GCN_compute.cc:65-98).  One 256-thread workgroup per tile:
//   * the tile's raw edges (a contiguous slice of edge_list / edge_attr) are validated, turned into (source row, destination row, edge
//     code) and counting-sorted by destination in LDS; inside a row they are rank-sorted by (source row, input index) -- the CSR's
//     order (graph_build.hip), whatever order the LDS atomics landed in; out-degrees are counted on the way;
//   * every node's nine features are validated and turned into three row numbers of the pre-combined PROJECTED table
//     (gin_resident_pack_enc_table applied to W_0 NodeEmb + b_0: T01 | T234 | T5678, 2 060 rows of 100 floats, L2-resident), so the
//     resident kernel's tile loader computes x_0 = (T01 + T234) + T5678 itself: the nine-term sum of the projected encoder
//     re-associated (fp32 rounding only; parity tolerance 1e-4).
// Descriptor (GCND_BYTES per tile, every byte written on every pass):
//   [0, 1920) in-edge words u16 x 960: source row << 6 | edge code, CSR order, 0 beyond the tile's edges
//   [1920, 2308) row offsets u16 x 194 (rows + 1 used; = edge count beyond)   [2320, 2704) out-degrees u16 x 192
//   [2704, 3472) encoder row numbers u32 x 192: T01 row | T234 row << 9 | T5678 row << 20
constexpr int GCND_RP = 1920, GCND_ODEG = 2320, GCND_ENC = 2704, GCND_BYTES = 3584;
constexpr int GCNB_T01 = 0, GCNB_T234 = 476, GCNB_T5678 = 1916;  // first rows of the table's parts (gin_resident_pack_enc_table)

__device__ __forceinline__ int gcnb_wave_inclusive_scan(int x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// `list` (GraphTiles::bp_list, bin-packed tiles; null: tile t = the graphs tile_graph[t] .. and the batch's rows tile_row[t] ..): tile t
// = the graphs list[tile_graph[t]] .., one behind the other; everything the resident kernel reads of a tile is in its descriptor.
__global__ __launch_bounds__(256) void gcn_tile_build_kernel(BatchView b, const int* __restrict__ tile_row, const int* __restrict__ tile_graph,
                                                             uint8_t* __restrict__ desc, int n_tiles, int* __restrict__ err,
                                                             const int* __restrict__ list) {
    constexpr int EPT = (GCNR_EDGES + 255) / 256;  // edges per thread
    __shared__ int s_eoff[GCNR_ROWS + 1], s_noff[GCNR_ROWS + 1];  // edge / row offsets of the tile's graphs, relative to the tile
    __shared__ int s_ebase[GCNR_ROWS + 1], s_nbase[GCNR_ROWS + 1];  // + these = the batch's edge / node behind a tile-local edge / row of that graph
    __shared__ int s_cnt[257], s_cur[256], s_odeg[256];
    __shared__ unsigned s_bucket[GCNR_EDGES];
    __shared__ int s_feat[GCNR_ROWS * ND_FEATURE];
    __shared__ int s_wtot[4];
    const int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const int r = threadIdx.x, lane = r & 63, wv = r >> 6;
    const int t0 = tile_row[tile];
    int rows = tile_row[tile + 1] - t0;
    if (rows > GCNR_ROWS) rows = GCNR_ROWS;
    const int g0 = tile_graph[tile];
    int ng = tile_graph[tile + 1] - g0;
    if (ng > GCNR_ROWS) ng = GCNR_ROWS;  // every graph has at least one node: a validated tile never has more graphs than rows
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
    } else {  // a LIST of graphs (bin-packed tiles): running sums of their counts (thread = graph, ng <= 192)
        const int gph = r < ng ? list[g0 + r] : 0;
        const int cn = r < ng ? b.nums_of_nodes[gph] : 0, ce = r < ng ? b.nums_of_edges[gph] : 0;
        const int in_n = gcnb_wave_inclusive_scan(cn, lane), in_e = gcnb_wave_inclusive_scan(ce, lane);
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
    if (ne > GCNR_EDGES) ne = GCNR_EDGES;  // cannot happen for a batch packed by flowgnn_set_batch; never overrun LDS
    s_cnt[r] = 0;
    s_cur[r] = 0;
    s_odeg[r] = 0;
    if (r == 0) s_cnt[256] = 0;
    if (list == nullptr)
        for (int i = r; i < GCNR_ROWS * ND_FEATURE; i += 256)  // the tile's node features, coalesced
            s_feat[i] = i < rows * ND_FEATURE ? b.node_feature[(size_t)t0 * ND_FEATURE + i] : 0;
    __syncthreads();
    if (list != nullptr && r < GCNR_ROWS) {  // (a list's rows are scattered over the batch: every row finds its graph, then its node)
        int lo = 0, hi = ng - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_noff[mid] <= r) lo = mid; else hi = mid - 1;
        }
        const int* nf = b.node_feature + (size_t)(s_nbase[lo] + (r < rows ? r : 0)) * ND_FEATURE;
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) s_feat[r * ND_FEATURE + k] = r < rows ? nf[k] : 0;
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
            const bool aok = (a0 >= 0) & (a0 < 5) & (a1 >= 0) & (a1 < 6) & (a2 >= 0) & (a2 < 2);  // cardinalities {5,6,2}: GCN/src/host_load.cc
            if (!aok) atomicMax(err, ERR_EDGE_ATTR);
            const unsigned code = aok ? (unsigned)((a0 * 6 + a1) * 2 + a2) : 0u;
            edst[k] = base + v;
            ekey[k] = ((unsigned)(base + u) << 17) | ((unsigned)i << 6) | code;
            atomicAdd(&s_cnt[base + v], 1);
            atomicAdd(&s_odeg[base + u], 1);  // degree_table: out-degree (load_inputs.cc:120)
        }
    }
    __syncthreads();
    const int deg = s_cnt[r];  // rows beyond the tile's last have none
    {
        const int incl = gcnb_wave_inclusive_scan(deg, lane);
        if (lane == 63) s_wtot[wv] = incl;
        __syncthreads();
        int start = incl - deg;
        for (int w = 0; w < wv; w++) start += s_wtot[w];
        s_cnt[r] = start;
        if (r == 255) s_cnt[256] = start + deg;  // = ne
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; k++)
        if (edst[k] >= 0) s_bucket[s_cnt[edst[k]] + atomicAdd(&s_cur[edst[k]], 1)] = ekey[k];
    __syncthreads();
    uint8_t* d = desc + (size_t)tile * GCND_BYTES;
    uint16_t* d_edge = reinterpret_cast<uint16_t*>(d);
    uint16_t* d_rp = reinterpret_cast<uint16_t*>(d + GCND_RP);
    uint16_t* d_od = reinterpret_cast<uint16_t*>(d + GCND_ODEG);
    uint32_t* d_enc = reinterpret_cast<uint32_t*>(d + GCND_ENC);
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int i = r + 256 * k;
        if (edst[k] >= 0) {
            const int beg = s_cnt[edst[k]], end = s_cnt[edst[k] + 1];
            const unsigned mine = ekey[k] >> 6;
            int rank = 0;
            for (int t = beg; t < end; t++) rank += (s_bucket[t] >> 6) < mine;
            d_edge[beg + rank] = (uint16_t)(((ekey[k] >> 17) << 6) | (ekey[k] & 63u));
        } else if (i < GCNR_EDGES) {
            d_edge[i] = 0;  // slots [ne, 960): every valid edge lands in [0, ne), so these are exactly the unused ones
        }
    }
    if (r < GCNR_ROWS + 2) d_rp[r] = (uint16_t)s_cnt[r < 256 ? r : 256];
    if (r < GCNR_ROWS) {
        d_od[r] = (uint16_t)s_odeg[r];
        unsigned word = 0;
        if (r < rows) {  // the node's rows of the pre-combined projected table (validated: table cardinalities, GCN/src/host_load.cc)
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
            word = i0 | (i1 << 9) | (i2 << 20);
        }
        d_enc[r] = word;
    }
}

#ifdef FLOWGNN_DEV
#include "dev/walk_timing_variants.h"  // development: timing variants of the walk (wrong sums on purpose), selected by extra -D flags
#endif
#ifndef GCN_WALK_WORD
#define GCN_WALK_WORD(W)
#endif

template <bool ONEPASS>
__global__ __launch_bounds__(GCNR_WAVES * 64, 3) void gcn_resident_kernel(const float* __restrict__ x0, const int* __restrict__ row_ptr,
                                                                         const int* __restrict__ src, const uint8_t* __restrict__ ecode,
                                                                         const int* __restrict__ out_deg, const uint8_t* __restrict__ layers,
                                                                         const float* __restrict__ pool_w, const float* __restrict__ pool_b,
                                                                         const int* __restrict__ tile_row, const int* __restrict__ tile_graph,
                                                                         const int* __restrict__ node_off, float* __restrict__ out, int n_tiles,
                                                                         int* __restrict__ range_flag, int ablate_arg,
                                                                         const uint8_t* __restrict__ desc, const float4* __restrict__ enc_tab,
                                                                         const int* __restrict__ list, const int* __restrict__ lrow) {
    // ONEPASS (the default front end since round 5): no x0 / row_ptr / src / ecode / out_deg -- the tile's CSR slice, out-degrees and
    // encoder row numbers come from gcn_tile_build_kernel's descriptor, and the loader computes the tile's x_0 rows itself from the
    // pre-combined projected table (three 400-B rows per node out of L2 instead of one out of HBM that another launch wrote).
    const int ablate = FG_ABLATE(ablate_arg);  // 0 in the shipped build: the branches below fold away (common.h)
    (void)ablate_arg;
    constexpr int OT = GCN_OT, NT = GCNR_WAVES * 64;
    constexpr int TAIL_OFF = OT * 6 * 1024, BIAS_OFF = TAIL_OFF + OT * 256, SCALE_OFF = BIAS_OFF + OT * 64;
    __shared__ __attribute__((aligned(16))) float s_x[GCNR_ROWS * GCN_D + 256];  // + slack: the last DMA piece of a tile may run past its rows
    __shared__ __attribute__((aligned(16))) char s_w[GCNR_W_BYTES];
    __shared__ __attribute__((aligned(16))) char s_blob[GCNR_BLOB_BYTES];
    __shared__ uint16_t s_edge[GCNR_EDGES];
    __shared__ uint16_t s_rp[GCNR_ROWS + 2];
    __shared__ float s_dinv[GCNR_ROWS], s_idp1[GCNR_ROWS], s_dot[GCNR_ROWS];
    // Column owner table: the walk of a 16-lane group takes as many trips as its LONGEST row, so the tile's rows are dealt to the
    // waves in order of decreasing in-degree (counting sort per tile; the order inside a degree class is whatever the LDS atomics
    // gave -- placement never changes a row's arithmetic: MFMA columns are independent and a row is summed in CSR order by one lane)
    __shared__ uint8_t s_perm[GCNR_ROWS];
    __shared__ __attribute__((aligned(16))) int s_cnt[16], s_cur[16];
    // the readout's weights, staged once per workgroup: read from global memory in the last layer, hipcc issued the seven loads one
    // at a time, each behind a vmcnt(0) -- seven serialized L2 round trips per tile
    __shared__ __attribute__((aligned(16))) float s_pw[GCN_D];
    __shared__ __attribute__((aligned(16))) uint32_t s_nidx[ONEPASS ? GCNR_ROWS : 4];  // ONEPASS: encoder row numbers of the NEXT tile's rows
    const bool sort_rows = !(ablate & 4);  // development aid: gcn_ablate, -DFLOWGNN_DEV builds=4 keeps rows in natural order
    const int lane0 = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* s_ecomb = reinterpret_cast<const float*>(s_blob);
    const float* s_ep = s_ecomb + EDGE_COMBOS * GCN_D;
    const uint32_t x_addr = lds_addr_of(s_x), w_addr = lds_addr_of(s_w), blob_addr = lds_addr_of(s_blob);
    float vmax = 0.0f;
#ifdef FLOWGNN_DEV  // gcn_ablate 64: per-phase clocks of workgroup 0's waves, printed at the end (scripts/dev: ab.py ... gcn_ablate=64)
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#define GCN_STAMP(i) do { if (ablate & 64) { const unsigned long long t_ = wall_clock64(); tph[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define GCN_STAMP(i) do { } while (0)
#endif
    // ONEPASS: the lane index is re-derived (opaque) at the top of every tile.  Left visible, hipcc computes every lane-dependent
    // address of the kernel once, in front of the tile loop, and holds ~60 of them in registers for the whole launch: nothing is
    // free at the tile boundary, where the loader wants 48 registers of table segments in flight (it spilled 568 B).
    int lane = lane0;
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    if ((int)threadIdx.x < GCN_D) s_pw[threadIdx.x] = pool_w[threadIdx.x];  // (read behind the tile loop's first barriers)
    const float pool_bias = pool_b[0];
    int t0 = tile_row[tile], rows = tile_row[tile + 1] - t0;
    if (rows > GCNR_ROWS) rows = GCNR_ROWS;
    int g0 = tile_graph[tile], g1 = tile_graph[tile + 1];
    int e0 = 0, ne = GCNR_EDGES;  // ONEPASS: the descriptor's 960 edge words are all valid to copy (zeros beyond the tile's edges)
    if (!ONEPASS) {
        e0 = row_ptr[t0];
        ne = row_ptr[t0 + rows] - e0;
        if (ne > GCNR_EDGES) ne = GCNR_EDGES;  // cannot happen for a validated batch (the host packed by edge count)
    }
    // a tile's rows of x_0 and layer 0's table come by LDS-DMA (requested as soon as the previous tile's last gather is done); its CSR
    // slice and degrees travel through registers: requested during the previous tile's middle layers, stored to LDS when that tile is done
    auto issue_rows = [&](int ft0, int frows) {
        if (!ONEPASS) {
            const int np = (frows * (GCN_D * 4) + 1023) >> 10;  // <= 75 pieces of 1 KiB
            const char* gb = reinterpret_cast<const char*>(x0) + (size_t)ft0 * (GCN_D * 4);
#pragma unroll
            for (int p = 0; p < 7; p++) {
                const int piece = wv + GCNR_WAVES * p;
                if (piece < np) lds_dma16(gb + (size_t)piece * 1024, (uint32_t)lane * 16u, x_addr + piece * 1024);
            }
        }
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int piece = wv + GCNR_WAVES * p;
            if (piece < GCNR_BLOB_BYTES / 1024) lds_dma16(layers + GCNR_W_BYTES + piece * 1024, (uint32_t)lane * 16u, blob_addr + piece * 1024);
        }
    };
    int epre0 = 0, epre1 = 0, rpre = 0, dpre = 0;
    // ONEPASS: the encoder row numbers of this thread's seven float4 slots of the next tile's rows (slot n = thread + 768 k: chunk
    // n mod 25 of row n / 25), and the 21 table segments they select
    constexpr int XK = (GCNR_ROWS * GCN_C + NT - 1) / NT;  // 7
    float4 xa[XK], xb[XK], xc[XK];
    auto fetch_csr = [&](int ft0, int frows, int fe0, int fne, int ftile) {
        if (ONEPASS) {
            const uint8_t* d = desc + (size_t)ftile * GCND_BYTES;
            const uint16_t* de = reinterpret_cast<const uint16_t*>(d);
            epre0 = de[threadIdx.x];
            if ((int)threadIdx.x + NT < GCNR_EDGES) epre1 = de[threadIdx.x + NT];
            if ((int)threadIdx.x < GCNR_ROWS + 2) rpre = reinterpret_cast<const uint16_t*>(d + GCND_RP)[threadIdx.x];
            if ((int)threadIdx.x < GCNR_ROWS) dpre = reinterpret_cast<const uint16_t*>(d + GCND_ODEG)[threadIdx.x];
            return;
        }
        if ((int)threadIdx.x < fne) epre0 = (((src[fe0 + threadIdx.x] - ft0) & 255) << 6) | (ecode[fe0 + threadIdx.x] & 63);
        if ((int)threadIdx.x + NT < fne) epre1 = (((src[fe0 + threadIdx.x + NT] - ft0) & 255) << 6) | (ecode[fe0 + threadIdx.x + NT] & 63);
        if ((int)threadIdx.x <= frows) {
            const int o = row_ptr[ft0 + threadIdx.x] - fe0;
            rpre = o < 0 ? 0 : (o > fne ? fne : o);
        }
        if ((int)threadIdx.x < frows) dpre = out_deg[ft0 + threadIdx.x];
    };
    // The row numbers come by LDS-DMA (768 B = 48 lanes of one piece) into s_nidx at the top of the tile BEFORE the one they belong to:
    // no register carries them through the layers (seven more live registers through the dense phases spilled), and everything the
    // loader defines it defines unconditionally (a conditional definition inside the tile loop is a phi with the previous tile's
    // value: 84 registers "live" around the whole loop).
    auto opaque_tid = [&]() {
        unsigned t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    };
    auto issue_idx = [&](int ftile) {
        if (wv == GCNR_WAVES - 1 && lane < 48)
            lds_dma16(desc + (size_t)ftile * GCND_BYTES + GCND_ENC, (uint32_t)lane * 16u, lds_addr_of(s_nidx));
    };
    // ONEPASS x_0: request the table segments of this thread's slots (L2 hits), then -- once the tile's rows are dead -- add and store.
    // In two halves (slots 0-3, then 4-6): all 21 segments at once are 84 registers across the tile's last barrier, which spilled.
    constexpr int XH = 4;
    auto x0_request = [&](int k0, int k1) {
        const unsigned tid = opaque_tid();
#pragma unroll
        for (int k = 0; k < XK; k++) {
            if (k < k0 || k >= k1) continue;
            const unsigned n = tid + NT * k;
            const unsigned row = (n * 5243u) >> 17, c = n - 25u * row;
            const uint32_t w = s_nidx[row < (unsigned)GCNR_ROWS ? row : 0u];
            xa[k] = enc_tab[((unsigned)GCNB_T01 + (w & 511u)) * 25u + c];
            xb[k] = enc_tab[((unsigned)GCNB_T234 + ((w >> 9) & 2047u)) * 25u + c];
            xc[k] = enc_tab[((unsigned)GCNB_T5678 + (w >> 20)) * 25u + c];
        }
    };
    auto x0_store = [&](int frows, int k0, int k1) {
        const int tid = (int)opaque_tid();
#pragma unroll
        for (int k = 0; k < XK; k++) {
            if (k < k0 || k >= k1) continue;
            const int n = tid + NT * k;
            if (n < frows * GCN_C)
                reinterpret_cast<float4*>(s_x)[n] = make_float4((xa[k].x + xb[k].x) + xc[k].x, (xa[k].y + xb[k].y) + xc[k].y,
                                                                (xa[k].z + xb[k].z) + xc[k].z, (xa[k].w + xb[k].w) + xc[k].w);
        }
    };
    static_assert(GCNR_EDGES <= 2 * NT, "two CSR words per thread");
    static_assert(GCNR_ROWS * GCN_C <= 5376 && XK * NT <= 5376 + NT, "n / 25 by multiply-shift");
    // (The CSR words are CONSUMED -- an empty asm that names them -- before the rows are requested: hipcc then waits for them there,
    // where they have long landed, and not at the top of the tile loop, where the same vmcnt(0) would also wait for the 75 KiB of
    // rows requested after them: the tile's CSR staging and row sort now run under that transfer.  Launch 4.92 -> see DESIGN section 4.)
    fetch_csr(t0, rows, e0, ne, tile);
    asm volatile("" : "+v"(epre0), "+v"(epre1), "+v"(rpre), "+v"(dpre));
    issue_rows(t0, rows);
    if (ONEPASS) {
        issue_idx(tile);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        x0_request(0, XH);
        x0_store(rows, 0, XH);
        x0_request(XH, XK);  // (stored at the top of the tile loop, as for every later tile)
        __syncthreads();     // every wave has read its row numbers: the loop's first act is to request the NEXT tile's into s_nidx
    }
    while (true) {
        if (ONEPASS) {
            lane = lane0;
            asm volatile("" : "+v"(lane));
        }
        const int j = lane & 15, g = lane >> 4;
        const int tid = ONEPASS ? wv * 64 + lane : (int)threadIdx.x;
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < n_tiles;
        int nt0 = 0, nrows = 0, ng0 = 0, ng1 = 0, ne0 = 0, nne = 0;
        if (tid < ne) s_edge[tid] = (uint16_t)epre0;  // (ONEPASS: ne = 960, the descriptor has zeros beyond the tile's edges)
        if (tid + NT < ne) s_edge[tid + NT] = (uint16_t)epre1;
        if (tid <= rows) s_rp[tid] = (uint16_t)rpre;
        if (tid < rows) {
            s_dinv[tid] = dpre > 0 ? 1.0f / sqrtf((float)(dpre + 1)) : 0.0f;  // load_inputs.cc:122
            s_idp1[tid] = 65536.0f / (float)(dpre + 1);  // 2^16 / (deg + 1): the self term arrives scaled by 2^-16 (epilogue below)
        }
        if (tid < 16) { s_cnt[tid] = 0; s_cur[tid] = 0; }
        if (ONEPASS) issue_idx(has_next ? ntile : tile);  // (its last readers were the previous tile's closing requests, a barrier ago)
        __syncthreads();
        int skey = 15;  // in-degree class of row tid: 0 = longest (>= 14 in-edges) .. 14 = none, 15 = no such row
        if (tid < GCNR_ROWS) {
            if (tid < rows) {
                const int deg = (int)s_rp[tid + 1] - (int)s_rp[tid];
                skey = 14 - (deg < 14 ? deg : 14);
            }
            if (sort_rows) atomicAdd(&s_cnt[skey], 1);
        }
        __syncthreads();
        if (tid < GCNR_ROWS) {
            int pos = tid;
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
            s_perm[pos] = (uint8_t)tid;
        }
        if (ONEPASS) x0_store(rows, XH, XK);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the tile's rows and of layer 0's table
        __syncthreads();
        // Which wave walks which 16 rows of the degree order.  Waves w, w + 4, w + 8 share a SIMD; dealt in order (wave w = group w) SIMD 0
        // walked groups 0, 4, 8 and SIMD 3 groups 3, 7, 11 -- the longest rows together.  Waves 0..2 now take groups 0, 4, 8, waves 3..5
        // groups 1, 5, 9, ...: every SIMD gets a long, a middle and a short walk (ranks 0+5+10, 4+9+3, 8+2+7, 1+6+11), and the waves that
        // carry the tile's set-up (threads 0..191) one of each: gcn_resident -1.0 %.  (The reverse order, long rows on the last
        // waves: +5 %; snake order over the SIMDs: +0.5 %; scripts/dev/ab.py, one box.)
        const int grp = (wv % 3) * 4 + wv / 3;
        const int r = s_perm[grp * 16 + j];
        const bool valid = r < rows;
        const int rr = valid ? r : 0;  // rows past the tile's end repeat row 0 without in-edges (finite values, never stored)
        const int e_begin = valid ? (int)s_rp[r] : 0, e_end = valid && !(ablate & 1) ? (int)s_rp[r + 1] : e_begin;  // ablate: development aid (gcn_ablate, -DFLOWGNN_DEV builds)
        const float dinv_v = s_dinv[rr], idp1 = s_idp1[rr];
        int trips = e_end - e_begin;  // the wave's longest row
#pragma unroll
        for (int mk = 1; mk < 64; mk <<= 1) trips = max(trips, __shfl_xor(trips, mk, 64));
        trips = __builtin_amdgcn_readfirstlane(trips);
        const int ro_gi = g0 + tid;
        int ro_n0 = 0, ro_n1 = 1, ro_g = 0;
#pragma unroll 1
        GCN_STAMP(0);  // tile set-up
        for (int l = 0; l < GCN_L; l++) {
            if (l == 1 && has_next) {  // the next tile's descriptor (two dependent scalar round trips), used from layer 2 on
                nt0 = tile_row[ntile];
                nrows = tile_row[ntile + 1] - nt0;
                if (nrows > GCNR_ROWS) nrows = GCNR_ROWS;
                ng0 = tile_graph[ntile]; ng1 = tile_graph[ntile + 1];
                if (!ONEPASS) {
                    ne0 = row_ptr[nt0];
                    nne = row_ptr[nt0 + nrows] - ne0;
                    if (nne > GCNR_EDGES) nne = GCNR_EDGES;
                } else {
                    nne = GCNR_EDGES;
                }
            }
            if (l == 2 && has_next) fetch_csr(nt0, nrows, ne0, nne, ntile);
            // ---- m_l[v] = sum over in-edges of norm relu(x_l[u] + ecomb[code]) (message_passing.cc:158-167), CSR order, all from LDS
            const float* xr = s_x + rr * GCN_D + 4 * g;
            float4 xs[6];
#pragma unroll
            for (int q = 0; q < 6; q++) xs[q] = *reinterpret_cast<const float4*>(xr + 16 * q);
            const float xst = s_x[rr * GCN_D + 96 + g];
            // The walk takes `trips` wave-uniform trips (the wave's longest row, found once per tile); a lane whose row is done keeps
            // walking with norm = 0 (source row 0, code 0: finite values), which adds +0 to its sums and leaves their bits alone -- so
            // the loop body has no per-lane branch.  (With `if (lane has an edge) m += ...` hipcc copies all the accumulators on both
            // paths of every trip: launch 5.40 -> 5.10 ms.  Spelling the fold as v_pk_add / v_pk_fma -- 65 VALU instructions per trip
            // instead of 100 -- changes nothing, 5.14 ms: the walk is bound by its LDS reads, not by VALU issue.)
            float2_t mq[12];
            float mt = 0.0f;
            const float2_t msg_c2 = {1.0f / 65536.0f, 1.0f / 65536.0f};
#pragma unroll
            for (int k = 0; k < 12; k++) mq[k] = (float2_t){0.0f, 0.0f};
            {
                int e = e_begin;
                int w_nx = e < e_end ? (int)s_edge[e] : 0;
                // Two memories feed the walk.  Per in-edge a lane group reads 400 B of x[u] and 400 B of the edge table, and the LDS array --
                // half of its cycles bank conflicts of sixteen unrelated rows per pass -- is what the walk waits for, while the CU's
                // vector-memory path (64 B per clock from L1) has nothing to do between the weight streams.  So GCNR_EQ_VMEM of the table
                // row's six quads come from the table's second copy in global memory (set_weights: planes [quad][quarter][code] of 16 B, so
                // that a quarter wave's sixteen codes fall into 8 cache lines; 24 KB per layer, L1 / L2 resident), requested one trip ahead.
                // Same values, same instructions on them: bit-identical.  Launch at 2^18 molhiv graphs, one box: 4.61 ms (all LDS),
                // 4.40 / 4.34 / 4.49 / 4.69 with 3 / 4 / 5 / 6 quads through L1 (row-major copy: best at 3 quads, 4.39).
                const float* etab_g = reinterpret_cast<const float*>(layers + GCNR_PLANES_OFF + (size_t)l * GCNR_PLANES_BYTES) + 256 * g;
                float4_t wg[GCNR_EQ_VMEM];
#pragma unroll
                for (int q = 0; q < GCNR_EQ_VMEM; q++) wg[q] = *reinterpret_cast<const float4_t*>(etab_g + (w_nx & 63) * 4 + 1024 * q);
#pragma unroll 1
                for (int t = 0; t < trips; t++) {
                    GCN_WALK_WORD(w_nx);
                    const int u = w_nx >> 6, code = w_nx & 63;
                    const float norm = e < e_end ? s_dinv[u] * dinv_v : 0.0f;
                    e++;
                    const bool more = e < e_end;
                    const int nw = (int)s_edge[more ? e : 0];
                    w_nx = more ? nw : 0;
                    const float* hr = s_x + u * GCN_D + 4 * g;
                    const float* er = s_ecomb + code * GCN_D + 4 * g;
                    float4_t xv[6], wv4[6];
#pragma unroll
                    for (int q = 0; q < GCNR_EQ_VMEM; q++) wv4[q] = wg[q];
                    {
                        const float* ergn = etab_g + (w_nx & 63) * 4;  // the next trip's row (code 0 behind the last edge)
#pragma unroll
                        for (int q = 0; q < GCNR_EQ_VMEM; q++) wg[q] = *reinterpret_cast<const float4_t*>(ergn + 1024 * q);
                    }
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        xv[q] = *reinterpret_cast<const float4_t*>(hr + 16 * q);
                        if (q >= GCNR_EQ_VMEM) wv4[q] = *reinterpret_cast<const float4_t*>(er + 16 * q);
                    }
                    const float xt = s_x[u * GCN_D + 96 + g];
                    const float wt = s_ecomb[code * GCN_D + 96 + g];
                    // relu(x + e) as ONE clamped packed FMA per two values (as gin_resident_kernel's walk, gin_split.hip GR_MSG2): the table is
                    // stored as e * 2^-16, v_pk_fma_f32(x, 2^-16, e') clamp = relu(x + e) * 2^-16 exactly while x + e < 2^16, and the norm
                    // carries the 2^16 back (exact): the same bits with two VALU instructions per two values instead of four.  x + e >= 2^16
                    // needs |x| > 6e4 (|e| < 4 096: set_weights), and every x this kernel writes is tracked by the range flag below.
                    const float ns = norm * 65536.0f;
                    const float2_t n2 = {ns, ns};
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        float2_t ta, tb;
                        { const float2_t xl = xv[q].lo, wl = wv4[q].lo, xh = xv[q].hi, wh = wv4[q].hi;
                          asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(ta) : "v"(xl), "s"(msg_c2), "v"(wl));
                          asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(tb) : "v"(xh), "s"(msg_c2), "v"(wh)); }
                        mq[2 * q + 0] += n2 * ta;
                        mq[2 * q + 1] += n2 * tb;
                    }
                    { float t1; asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t1) : "v"(xt), "s"(1.0f / 65536.0f), "v"(wt)); mt += ns * t1; }
                }
            }
            GCN_STAMP(1);  // self row + walk
            // W_{l+1} (45 pieces of 1 KiB) is requested BEHIND the walk and lands under the BatchNorm / split that follows and under the
            // slower waves' last trips.  Requested at the top of the walk -- where it used to be -- every wave paid four LDS-DMA issues
            // (100-150 cycles each) in front of its first trip and the 45 KiB landed through the LDS the walk is bound by: 5.18 ms per
            // launch against 5.04 now (same box, scripts/dev/ab.py); in the middle of the walk 5.05.
            if (l + 1 < GCN_L && !(ablate & 16)) {  // (ablate 16: timing without it)
                const uint8_t* gw = layers + (size_t)(l + 1) * GCNR_LAYER_BYTES;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int piece = wv + GCNR_WAVES * p;
                    if (piece < GCNR_W_BYTES / 1024) lds_dma16(gw + piece * 1024, (uint32_t)lane * 16u, w_addr + piece * 1024);
                }
            }
            float m[25];
#pragma unroll
            for (int k = 0; k < 12; k++) { m[2 * k] = mq[k].x; m[2 * k + 1] = mq[k].y; }
            m[24] = mt;
            // ---- a_{l+1} = BN_l(m_l + relu(x_l + root_l) / (deg + 1))   (node_embedding.cc:123-138); folded BatchNorm
            float a[25];
            // (the three epilogue vectors of quad q + 1 are requested before quad q is computed: read where they are used, hipcc waited
            // for them in a dozen small batches -- an LDS round trip each, with two other waves on the SIMD to cover it)
            float4 epv[2][3];
#pragma unroll
            for (int i = 0; i < 3; i++) epv[0][i] = *reinterpret_cast<const float4*>(s_ep + i * GCN_D + 4 * g);
            const float ept0 = s_ep[96 + g], ept1 = s_ep[GCN_D + 96 + g], ept2 = s_ep[2 * GCN_D + 96 + g];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                if (q + 1 < 6) {
#pragma unroll
                    for (int i = 0; i < 3; i++) epv[(q + 1) & 1][i] = *reinterpret_cast<const float4*>(s_ep + i * GCN_D + 16 * (q + 1) + 4 * g);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4 rt = epv[q & 1][0], sc = epv[q & 1][1], sh = epv[q & 1][2];
                // relu(x + root) as ONE clamped packed FMA per two values, as the walk's messages: the blob holds root * 2^-16, the clamp
                // gives relu(x + root) * 2^-16 exactly (x + root < 2^16: range flag and set_weights' table check), and idp1 carries
                // the 2^16 back inside the next FMA -- the same bits for 13 instructions instead of 50
                float2_t r01, r23;
                { const float2_t x01 = {xs[q].x, xs[q].y}, x23 = {xs[q].z, xs[q].w}, t01 = {rt.x, rt.y}, t23 = {rt.z, rt.w};
                  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r01) : "v"(x01), "s"(msg_c2), "v"(t01));
                  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r23) : "v"(x23), "s"(msg_c2), "v"(t23)); }
                a[4 * q + 0] = __builtin_fmaf(__builtin_fmaf(r01.x, idp1, m[4 * q + 0]), sc.x, sh.x);
                a[4 * q + 1] = __builtin_fmaf(__builtin_fmaf(r01.y, idp1, m[4 * q + 1]), sc.y, sh.y);
                a[4 * q + 2] = __builtin_fmaf(__builtin_fmaf(r23.x, idp1, m[4 * q + 2]), sc.z, sh.z);
                a[4 * q + 3] = __builtin_fmaf(__builtin_fmaf(r23.y, idp1, m[4 * q + 3]), sc.w, sh.w);
                __builtin_amdgcn_sched_barrier(0);
            }
            { float r24; asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r24) : "v"(xst), "s"(1.0f / 65536.0f), "v"(ept0));
              a[24] = __builtin_fmaf(__builtin_fmaf(r24, idp1, m[24]), ept1, ept2); }
            if (l == GCN_L - 1) {
                // the readout's node range of "this lane's graph": requested here, a BatchNorm ahead of its use, and consumed BEFORE the
                // next tile's rows are requested (below) -- behind them, its vmcnt wait would also wait for that whole transfer
                // (bin-packed tiles, ONEPASS only: ro_gi is a position in the tile list -- the graph's id and its first row inside the tile)
                if (ro_gi < g1) {
                    ro_g = list ? list[ro_gi] : ro_gi;
                    ro_n0 = node_off[ro_g]; ro_n1 = node_off[ro_g + 1];
                    if (list) { const int lr0 = lrow[ro_gi]; ro_n1 = lr0 + t0 + (ro_n1 - ro_n0); ro_n0 = lr0 + t0; }
                }
                // no ReLU after the last BatchNorm; the readout's linear head per node (mean_v(a[v]) . w = mean_v(a[v] . w),
                // finalize.cc:79-113): 25 terms in the lane, then the node's 4 lanes
                float part = 0.0f;
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const float4 pw = *reinterpret_cast<const float4*>(s_pw + 16 * q + 4 * g);
                    part += a[4 * q + 0] * pw.x; part += a[4 * q + 1] * pw.y; part += a[4 * q + 2] * pw.z; part += a[4 * q + 3] * pw.w;
                }
                part += a[24] * s_pw[96 + g];
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                if (g == 0) s_dot[r] = part;
                break;
            }
#pragma unroll
            for (int k = 0; k < 25; k++) a[k] = relu1(a[k]);
            ds_uint4_t b_hi[3], b_lo[3];
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                DS_SPLIT2(a[8 * ks + 0], a[8 * ks + 1], b_hi[ks].x, b_lo[ks].x);
                DS_SPLIT2(a[8 * ks + 2], a[8 * ks + 3], b_hi[ks].y, b_lo[ks].y);
                DS_SPLIT2(a[8 * ks + 4], a[8 * ks + 5], b_hi[ks].z, b_lo[ks].z);
                DS_SPLIT2(a[8 * ks + 6], a[8 * ks + 7], b_hi[ks].w, b_lo[ks].w);
            }
#pragma unroll
            for (int k = 0; k < 24; k += 2) asm("v_max3_f32 %0, %1, %2, %0" : "+v"(vmax) : "v"(a[k]), "v"(a[k + 1]));  // a >= 0 (ReLU)
            asm volatile("" : "+v"(vmax));
            const float a24 = a[24];
            GCN_STAMP(2);  // W request, BatchNorm, split
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // #1: every gather of this layer is done (rows may be rewritten, the table replaced); W_{l+1} has landed
            auto issue_blob = [&]() {  // the next layer's table + epilogue vectors stream in under the dense layer
                const uint8_t* gbl = layers + (size_t)(l + 1) * GCNR_LAYER_BYTES + GCNR_W_BYTES;
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    const int piece = wv + GCNR_WAVES * p;
                    if (piece < GCNR_BLOB_BYTES / 1024) lds_dma16(gbl + piece * 1024, (uint32_t)lane * 16u, blob_addr + piece * 1024);
                }
            };
            GCN_STAMP(3);  // wait + barrier #1
            if (!(ablate & 8)) issue_blob();  // (ablate 8: timing without it; issued after the third or the last column tile instead: no change / +1.3 %)
            // ---- x_{l+1} = b + W a on the f16 matrix pipe (split products, dense_split.h), written over the wave's own rows
            const float oscale = *reinterpret_cast<const float*>(s_w + SCALE_OFF);
            // The six fragments of column tile t + 1 are requested BEFORE the MFMAs of tile t issue (scheduling barriers pin the order):
            // left to itself hipcc reads one fragment, waits for it and issues one or two MFMAs, so every wave paid an LDS round
            // trip per fragment -- six per column tile -- with nothing of its own in flight.  Launch 5.06 -> 4.92 ms (same box).
            ds_uint4_t fr[2][6];
            float4_t accp = {0.f, 0.f, 0.f, 0.f};
            auto store_tile = [&](const float4_t& acc, int t) {
                const int col = 16 * t + 4 * g;
                if (col < GCN_D && valid) {
                    const float4_t o = acc * oscale;
                    *reinterpret_cast<float4*>(s_x + r * GCN_D + col) = make_float4(o.x, o.y, o.z, o.w);
                    // (the next walk's clamped messages need |x| < 6e4: beyond it the pass is repeated on the exact kernels)
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(vmax) : "v"(o.x), "v"(o.y));
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(vmax) : "v"(o.z), "v"(o.w));
                }
            };
#pragma unroll
            for (int i = 0; i < 6; i++) fr[0][i] = *reinterpret_cast<const ds_uint4_t*>(s_w + i * 1024 + lane * 16);
#pragma unroll
            for (int t = 0; t < OT; t++) {
                const float4 bv = *reinterpret_cast<const float4*>(s_w + BIAS_OFF + (16 * t + 4 * g) * 4);
                const float at = *reinterpret_cast<const float*>(s_w + TAIL_OFF + t * 256 + lane * 4);
                if (t + 1 < OT) {
#pragma unroll
                    for (int i = 0; i < 6; i++) fr[(t + 1) & 1][i] = *reinterpret_cast<const ds_uint4_t*>(s_w + ((t + 1) * 6 + i) * 1024 + lane * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                float4_t acc = {bv.x, bv.y, bv.z, bv.w};
                if (!(ablate & 2)) {  // (ablate 2: development aid, timing without the MFMAs)
#pragma unroll
                    for (int ks = 0; ks < 3; ks++) {
                        acc = DS_MFMA16(fr[t & 1][2 * ks], b_hi[ks], acc);
                        acc = DS_MFMA16(fr[t & 1][2 * ks], b_lo[ks], acc);
                        acc = DS_MFMA16(fr[t & 1][2 * ks + 1], b_hi[ks], acc);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at, a24, acc, 0, 0, 0);
                }
                // a column tile's results are stored one tile LATER, behind the next tile's MFMAs: stored at once, the wave stood
                // through the latency of its ten-MFMA chain (each waits for its predecessor's accumulator) seven times per layer
                __builtin_amdgcn_sched_barrier(0);
                if (t > 0) store_tile(accp, t - 1);
                accp = acc;
                __builtin_amdgcn_sched_barrier(0);
            }
            store_tile(accp, OT - 1);
            GCN_STAMP(4);  // dense layer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // #2: x_{l+1} is complete, the next table has landed
            GCN_STAMP(5);  // wait + barrier #2
        }
        // ONEPASS: this wave is done with the tile's rows; the next tile's table segments travel while the slower waves finish their
        // walks (the registers of the walk and the BatchNorm are free now), and are stored behind the barrier
        if (ONEPASS) x0_request(0, XH);  // (unconditional: behind the last tile the numbers are that tile's own again -- valid rows, never stored)
        __syncthreads();  // the per-node readout terms are in s_dot; the rows and the table are dead
        asm volatile("" : "+v"(epre0), "+v"(epre1), "+v"(rpre), "+v"(dpre), "+v"(ro_n0), "+v"(ro_n1), "+v"(ro_g));  // (see the prologue)
        if (has_next) issue_rows(nt0, nrows);
        if (ONEPASS) {
            x0_store(nrows, 0, XH);  // (nrows = 0 behind the last tile)
            x0_request(XH, XK);  // the second half travels under the readout, the CSR staging and the row sort of the next tile
        }
        if (ro_gi < g1) out[ro_g] = lds_sum_in_order(s_dot + (ro_n0 - t0), ro_n1 - ro_n0) / (float)(ro_n1 - ro_n0) + pool_bias;
        if (!has_next) break;
        tile = ntile; t0 = nt0; rows = nrows; g0 = ng0; g1 = ng1; e0 = ne0; ne = nne;
        GCN_STAMP(6);  // last layer's tail, readout, next tile's requests
        __syncthreads();  // the readout has read s_dot and s_rp's neighbours: the small arrays may be rewritten
    }
#ifdef FLOWGNN_DEV
    if ((ablate & 64) && blockIdx.x == 0 && (threadIdx.x & 63) == 0)
        printf("gcn wave %d clocks(100MHz): setup %llu walk %llu bn %llu bar1 %llu dense %llu bar2 %llu tail %llu\n", (int)(threadIdx.x >> 6), tph[0], tph[1], tph[2],
               tph[3], tph[4], tph[5], tph[6]);
#endif
#undef GCN_STAMP
    if (__any(!(vmax < 6.0e4f))) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
}

class GcnModel : public Model {
public:
    ~GcnModel() override { free_all(); }
    int emb_dim() const override { return GCN_D; }
    int scratch_dim() const override { return GCN_D; }
    int aggregate_dim() const override { return qmode_ ? 0 : GCN_D; }  // fixed-point modes have no float aggregation kernel
    bool has_edge_attr() const override { return true; }
    int num_weight_tensors() const override { return 11; }
    bool weights_ready() const override { return ready_; }

    // host tensors (GCN/src/dcl.h:83-96): node_emb[173][100], edge_emb[5][13][100], convs_weight[5][100][100],
    // convs_bias[5][100], root_emb[5][100], bn_weight/bias/mean/var[5][100], pred_w[1][100], pred_b[1]
    int set_numeric_mode(int mode) override {
        if (mode != 0 && mode != 1) return 8;
        if (mode == 1 && num_tasks_ != 1) return 8;  // the fixed-point readout is single-task, as the reference's
        qmode_ = mode == 1;
        return 0;
    }

    int set_weights(const float* const* t) override {
        {   // ap_fixed<16,6> copies of every tensor for the bit-faithful mode (modelq.hip)
            const size_t elems[11] = {173 * 100, 5 * 13 * 100, 5 * 100 * 100, 500, 500, 500, 500, 500, 500, (size_t)num_tasks_ * 100, (size_t)num_tasks_};
            if (int rc = q_.upload_all(11, t, elems, 10)) return rc;
            if (q_.extra) { (void)hipFree(q_.extra); q_.extra = nullptr; }  // derived tables follow the weights
        }
        const float *nemb = t[0], *eemb = t[1], *cw = t[2], *cb = t[3], *root = t[4], *bnw = t[5], *bnb = t[6], *bnm = t[7],
                    *bnv = t[8], *pw = t[9], *pb = t[10];
        std::vector<float> v_nemb(nemb, nemb + ND_FEATURE_TOTAL * GCN_D), v_pw(pw, pw + (size_t)num_tasks_ * GCN_D), v_pb(pb, pb + num_tasks_);
        std::vector<float> ecomb((size_t)GCN_L * EDGE_COMBOS * GCN_D), ep((size_t)GCN_L * 3 * GCN_D);
        std::vector<float> wf_all, wt_all, bp_all;
        std::vector<uint8_t> split_all;
        static const int ed_off[3] = {0, 5, 11};
        for (int l = 0; l < GCN_L; l++) {
            const float* E = eemb + (size_t)l * ED_FEATURE_PER_LAYER * GCN_D;
            for (int a0 = 0; a0 < 5; a0++)
                for (int a1 = 0; a1 < 6; a1++)
                    for (int a2 = 0; a2 < 2; a2++)
                        for (int d = 0; d < GCN_D; d++) {
                            float s = 0.0f;
                            s += E[(ed_off[0] + a0) * GCN_D + d];
                            s += E[(ed_off[1] + a1) * GCN_D + d];
                            s += E[(ed_off[2] + a2) * GCN_D + d];
                            ecomb[((size_t)l * EDGE_COMBOS + (a0 * 6 + a1) * 2 + a2) * GCN_D + d] = s;
                        }
            float* e = &ep[(size_t)l * 3 * GCN_D];
            for (int d = 0; d < GCN_D; d++) {
                const double sv = sqrt((double)(bnv[l * GCN_D + d] + 1.0f / 1024.0f));  // load_inputs.cc:32
                const double scale = (double)bnw[l * GCN_D + d] / sv;
                e[0 * GCN_D + d] = root[l * GCN_D + d];
                e[1 * GCN_D + d] = (float)scale;
                e[2 * GCN_D + d] = (float)((double)bnb[l * GCN_D + d] - (double)bnm[l * GCN_D + d] * scale);
            }
            std::vector<float> wf, wt, bp;
            pack_dense100(cw + (size_t)l * GCN_D * GCN_D, cb + (size_t)l * GCN_D, GCN_D, GCN_OT, wf, wt, bp);
            wf_all.insert(wf_all.end(), wf.begin(), wf.end());
            wt_all.insert(wt_all.end(), wt.begin(), wt.end());
            bp_all.insert(bp_all.end(), bp.begin(), bp.end());
            const size_t off = split_all.size();
            split_all.resize(off + dense100_split_bytes(GCN_OT));
            pack_dense100_split(cw + (size_t)l * GCN_D * GCN_D, cb + (size_t)l * GCN_D, GCN_D, GCN_OT, split_all.data() + off);
        }
        int rc;
        {   // x_0 = b_0 + W_0 (sum_k NodeEmb[off_k + f_k]) = sum_k (W_0 NodeEmb[off_k + f_k]) + b_0: the first dense layer is linear in the
            // looked-up rows, so it is applied to the TABLE once (in double), b_0 goes into the rows of feature 0 (every node reads exactly
            // one of them), and x_0 becomes nine lookups and adds per value (atom_encoder_kernel on the projected table): 0.55 ms instead
            // of the 1.1 ms of gcn_encoder_dense_kernel at 2^18 molpcba graphs.  Differs from the reference's order of operations by
            // fp32 reassociation only (~1e-6 relative; the parity tolerance is 1e-4).
            std::vector<float> proj((size_t)ND_FEATURE_TOTAL * GCN_D);
            const int card0 = 119;  // rows of feature 0 (atomic number), load_inputs.cc:168-215
            for (int r = 0; r < ND_FEATURE_TOTAL; r++)
                for (int o = 0; o < GCN_D; o++) {
                    double a = r < card0 ? (double)cb[o] : 0.0;
                    for (int i = 0; i < GCN_D; i++) a += (double)cw[(size_t)o * GCN_D + i] * (double)nemb[(size_t)r * GCN_D + i];
                    proj[(size_t)r * GCN_D + o] = (float)a;
                }
            proj_max_src_ = proj;
            if ((rc = upload(&d_nemb_proj_, proj))) return rc;
            // ... and its pre-combined form for the one-pass front end (x_0 = (T01 + T234) + T5678: gcn_tile_build_kernel)
            std::vector<float> comb(gin_resident_enc_table_floats());
            gin_resident_pack_enc_table(proj.data(), comb.data());
            if ((rc = upload(&d_enc_tab_, comb))) return rc;
        }
        {   // the graph-resident kernel's per-layer stream: [W_l split fragments, 45 KiB][ecomb_l | root_l | BN scale_l | BN shift_l, 25 KiB]
            std::vector<uint8_t> res((size_t)GCNR_PLANES_OFF + (size_t)GCN_L * GCNR_PLANES_BYTES, 0);
            float emax = 0.0f;
            for (int l = 0; l < GCN_L; l++) {
                uint8_t* base = res.data() + (size_t)l * GCNR_LAYER_BYTES;
                std::memcpy(base, split_all.data() + (size_t)l * dense100_split_bytes(GCN_OT), dense100_split_bytes(GCN_OT));
                {   // the edge-embedding combos scaled by 2^-16: the resident walk's clamped messages (gcn_resident_kernel)
                    float* dst = reinterpret_cast<float*>(base + GCNR_W_BYTES);
                    for (int i = 0; i < EDGE_COMBOS * GCN_D; i++) {
                        const float e = ecomb[(size_t)l * EDGE_COMBOS * GCN_D + i];
                        dst[i] = e * (1.0f / 65536.0f);
                        { const float ae = std::fabs(e); emax = (ae > emax || ae != ae) ? ae : emax; }  // (not fmax: it would drop a NaN)
                    }
                    float* pl = reinterpret_cast<float*>(res.data() + GCNR_PLANES_OFF + (size_t)l * GCNR_PLANES_BYTES);
                    for (int q = 0; q < 6; q++)
                        for (int gq = 0; gq < 4; gq++)
                            for (int c = 0; c < EDGE_COMBOS; c++)
                                for (int k = 0; k < 4; k++) pl[(((size_t)q * 4 + gq) * 64 + c) * 4 + k] = dst[c * GCN_D + 16 * q + 4 * gq + k];
                }
                std::memcpy(base + GCNR_W_BYTES + sizeof(float) * EDGE_COMBOS * GCN_D, &ep[(size_t)l * 3 * GCN_D], sizeof(float) * 3 * GCN_D);
                {   // ... and the root embedding scaled likewise: the epilogue's relu(x + root) is one clamped FMA too
                    float* rt = reinterpret_cast<float*>(base + GCNR_W_BYTES + sizeof(float) * EDGE_COMBOS * GCN_D);
                    for (int d = 0; d < GCN_D; d++) {
                        { const float ar = std::fabs(rt[d]); emax = (ar > emax || ar != ar) ? ar : emax; }
                        rt[d] *= 1.0f / 65536.0f;
                    }
                }
            }
            if ((rc = upload(&d_res_, res))) return rc;
            // the scaled walk is exact while x + e < 2^16: |e| < 4 096 here, |x| < 6e4 by the range flag (x_0: nine projected-table rows)
            float pmax = 0.0f;
            for (int r = 0; r < ND_FEATURE_TOTAL * GCN_D; r++) { const float ap = std::fabs(proj_max_src_[r]); pmax = (ap > pmax || ap != ap) ? ap : pmax; }
            table_ok_ = emax < 4096.0f && 9.0f * pmax < 6.0e4f;
        }
        if ((rc = upload(&d_split_, split_all))) return rc;
        if ((rc = upload(&d_nemb_, v_nemb))) return rc;
        if ((rc = upload(&d_pw_, v_pw))) return rc;
        if ((rc = upload(&d_pb_, v_pb))) return rc;
        if ((rc = upload(&d_ecomb_, ecomb))) return rc;
        if ((rc = upload(&d_ep_, ep))) return rc;
        if ((rc = upload(&d_wf_, wf_all))) return rc;
        if ((rc = upload(&d_wt_, wt_all))) return rc;
        if ((rc = upload(&d_bp_, bp_all))) return rc;
        ready_ = true;
        return 0;
    }

    // GCN/src/host_load.cc:31-170: one file, hard-coded float offsets
    int load_weights_dir(const char* dir) override {
        const char* f = "gcn_ep1_dim100.weights.all.bin";
        std::vector<float> nemb(173 * 100), eemb(5 * 13 * 100), cw(5 * 100 * 100), cb(500), root(500), bnw(500), bnb(500),
            bnm(500), bnv(500), pw((size_t)num_tasks_ * 100), pb(num_tasks_);
        int rc;
        if ((rc = read_floats(dir, f, 0, nemb.size(), nemb.data()))) return rc;
        for (int l = 0; l < GCN_L; l++) {
            const size_t base = 17300 + 11500 * (size_t)l;
            if ((rc = read_floats(dir, f, base, 10000, &cw[(size_t)l * 10000]))) return rc;
            if ((rc = read_floats(dir, f, base + 10000, 100, &cb[l * 100]))) return rc;
            if ((rc = read_floats(dir, f, base + 10100, 100, &root[l * 100]))) return rc;
            if ((rc = read_floats(dir, f, base + 10200, 1300, &eemb[(size_t)l * 1300]))) return rc;
            const size_t bn = 74800 + 401 * (size_t)l;  // 4 x 100 floats + one skipped counter per layer
            if ((rc = read_floats(dir, f, bn, 100, &bnw[l * 100]))) return rc;
            if ((rc = read_floats(dir, f, bn + 100, 100, &bnb[l * 100]))) return rc;
            if ((rc = read_floats(dir, f, bn + 200, 100, &bnm[l * 100]))) return rc;
            if ((rc = read_floats(dir, f, bn + 300, 100, &bnv[l * 100]))) return rc;
        }
        // graph_pred_linear: weight [NUM_TASK][100] then bias [NUM_TASK] (NUM_TASK = 1 in the reference's file: 76805, 76905)
        if ((rc = read_floats(dir, f, 76805, pw.size(), pw.data()))) return rc;
        if ((rc = read_floats(dir, f, 76805 + pw.size(), pb.size(), pb.data()))) return rc;
        const float* t[11] = {nemb.data(), eemb.data(), cw.data(), cb.data(), root.data(), bnw.data(),
                              bnb.data(),  bnm.data(),  bnv.data(), pw.data(), pb.data()};
        return set_weights(t);
    }


    template <bool RELU_OUT>
    void launch_aggregate(const DeviceBatch& db, int l, const float* x, float* a, hipStream_t s) {
        typename GcnAggPolicy<RELU_OUT>::Params prm{db.csr.out_deg, d_ep_ + (size_t)l * 3 * GCN_D, esc_.p};
        launch_tiled_aggregate<GcnAggPolicy<RELU_OUT>>(prm, x, a, db.csr, d_ecomb_ + (size_t)l * EDGE_COMBOS * GCN_D, db.b.n_tot, tiles_.p,
                                                       tile_nominal_, s);
    }

    void launch_dense(int l, const float* a, float* x, int n, int* range_flag, hipStream_t s) {
        if (split_ && !exact_) {
            const long long wgs = ceil_div_ll(n, 128);
            const int grid = (int)(wgs < 768 ? wgs : 768);  // persistent: three 8-wave workgroups per CU (45 KB of LDS each)
            dense100_split_kernel<GCN_OT, false><<<grid, 512, 0, s>>>(a, x, d_split_ + (size_t)l * dense100_split_bytes(GCN_OT), n,
                                                                      GCN_D, range_flag);
            return;
        }
        constexpr int NT = 2;
        const int waves = (int)ceil_div_ll(n, 16 * NT);
        dense100_kernel<GCN_OT, NT, false><<<(waves + 3) / 4, 256, 0, s>>>(
            a, x, d_wf_ + (size_t)l * GCN_OT * 6 * 64 * 4, d_wt_ + (size_t)l * GCN_OT * 64, d_bp_ + (size_t)l * GCN_OT * 16, n,
            GCN_D);
    }

    // row tiles + dinv[src_e] per CSR entry for the per-layer kernels, once per batch pass
    int prepare_aggregate(DeviceBatch& db, Profiler& prof, hipStream_t s) {
        if (int rc = make_tile_bounds(tiles_, db.b.node_off, db.b.num_graphs, db.b.n_tot, tile_nominal_, tile_slack_, s)) return rc;
        if (db.b.e_tot > 0) {
            if (int rc = esc_.reserve((size_t)db.b.e_tot)) return rc;
            ProfScope p(prof, "edge_scalar", s);
            typename GcnAggPolicy<true>::Params prm{db.csr.out_deg, nullptr, nullptr};
            edge_scalar_kernel<GcnAggPolicy<true>><<<grid_for(db.b.e_tot, 256, 256 * 8), 256, 0, s>>>(prm, db.csr.src, esc_.p, db.b.e_tot);
        }
        agg_ready_ = true;
        return 0;
    }

    // graph-resident kernel (gcn_resident_kernel): whole graphs packed into tiles of <= 192 rows / 960 in-edges by flowgnn_set_batch
    void graph_tile_limits(int& rows, int& edges) const override {
        rows = resident_ ? GCNR_ROWS : 0;
        edges = resident_ ? GCNR_EDGES : 0;
    }
    void set_keep_h(bool on) override { keep_h_ = on; }

    // x_0 by the encoder, then everything else in one launch when the batch packs into graph tiles (tiles under half full waste MFMA
    // columns: the per-layer kernels take those; so do per-node taps and the multi-task readout)
    bool use_resident(const DeviceBatch& db) const {
        return resident_ && table_ok_ && !qmode_ && !keep_h_ && split_ && !exact_ && fused_ && num_tasks_ == 1 && db.gtiles.ok && db.gtiles.n_tiles > 0 &&
               db.gtiles.fill >= 0.5;
    }
    // the one-pass front end (gcn_tile_build_kernel + the resident kernel's own encoder): the default; gcn_tile_build = 0 restores the
    // three-launch front end (index build, projected encoder, resident kernel)
    bool one_pass(const DeviceBatch& db) const { return tile_build_ && use_resident(db) && db.b.edge_attr != nullptr; }
    bool needs_csr(const DeviceBatch& db) const override { return !one_pass(db); }
    bool wants_packed_tile_lists() const override { return binpack_ && tile_build_ && resident_ && !qmode_ && num_tasks_ == 1; }

    int forward(DeviceBatch& db, Profiler& prof, hipStream_t s) override {
        const int n = db.b.n_tot;
        if (n <= 0) return 0;
        agg_ready_ = false;  // tiles_ / esc_ are rebuilt by whichever float path runs below; a fixed-point pass leaves none
        x0_in_hbm_ = !qmode_ && !(use_resident(db) && one_pass(db));
        if (qmode_) return gcnq_forward(q_, db, prof, s);
        if (use_resident(db)) {
            const int grid = db.gtiles.n_tiles < 256 ? db.gtiles.n_tiles : 256;  // persistent: one 12-wave workgroup per CU (153 KB of LDS)
            if (one_pass(db)) {  // two launches: descriptors from the caller's arrays, then everything else (no CSR, no x_0 in HBM)
                // bin-packed tile lists when flowgnn_set_batch made them (option gcn_binpack): fewer, fuller tiles of the same graphs; a
                // row's sums depend on the row alone, so the logits are the same bits
                const bool bp = binpack_ && db.gtiles.bp_tiles > 0;
                const int* t_row = bp ? db.gtiles.bp_row : db.gtiles.row_start;
                const int* t_graph = bp ? db.gtiles.bp_graph : db.gtiles.graph_start;
                const int n_tiles = bp ? db.gtiles.bp_tiles : db.gtiles.n_tiles;
                if (int rc = desc_.reserve(((size_t)n_tiles * GCND_BYTES + 3) / 4)) return rc;
                {
                    ProfScope p(prof, "gcn_tile_build", s);
                    gcn_tile_build_kernel<<<n_tiles, 256, 0, s>>>(db.b, t_row, t_graph, reinterpret_cast<uint8_t*>(desc_.p), n_tiles, db.csr.err,
                                                                  bp ? db.gtiles.bp_list : nullptr);
                }
                ProfScope p(prof, "gcn_resident", s);
                gcn_resident_kernel<true><<<n_tiles < 256 ? n_tiles : 256, GCNR_WAVES * 64, 0, s>>>(
                    nullptr, nullptr, nullptr, nullptr, nullptr, d_res_, d_pw_, d_pb_, t_row, t_graph, db.b.node_off, db.out, n_tiles, db.range_flag,
                    ablate_, reinterpret_cast<const uint8_t*>(desc_.p), reinterpret_cast<const float4*>(d_enc_tab_), bp ? db.gtiles.bp_list : nullptr,
                    bp ? db.gtiles.bp_lrow : nullptr);
            } else {
                {
                    ProfScope p(prof, "gcn_encoder_projected", s);  // x_0 from the projected table (set_weights)
                    atom_encoder_kernel<GCN_D><<<atom_encoder_grid(n, GCN_C), 512, 0, s>>>(db.b.node_feature, d_nemb_proj_, db.h[0], n, db.csr.err);
                }
                ProfScope p(prof, "gcn_resident", s);
                gcn_resident_kernel<false><<<grid, GCNR_WAVES * 64, 0, s>>>(db.h[0], db.csr.row_ptr, db.csr.src, db.csr.ecode, db.csr.out_deg, d_res_, d_pw_,
                                                                           d_pb_, db.gtiles.row_start, db.gtiles.graph_start, db.b.node_off, db.out,
                                                                           db.gtiles.n_tiles, db.range_flag, ablate_, nullptr, nullptr, nullptr, nullptr);
            }
            agg_ready_ = false;
            db.final_h = 0;
            db.h_valid = false;  // h[0] holds x_0, not x_4: flowgnn_get_h repeats the pass on the per-layer kernels
            return 0;
        }
        if (int rc = prepare_aggregate(db, prof, s)) return rc;
        int cur = 0;
        if (split_ && !exact_ && fused_) {
            ProfScope p(prof, "gcn_encoder_dense", s);  // x_0 = W_0 (atom encoder) + b_0 in one kernel
            const long long wgs = ceil_div_ll(n, 256);
            gcn_encoder_dense_kernel<<<(int)(wgs < 256 ? wgs : 256), 1024, 0, s>>>(db.b.node_feature, d_nemb_, db.h[cur], d_split_, n,
                                                                                   db.csr.err, db.range_flag);
        } else {
            {
                ProfScope p(prof, "atom_encoder", s);
                atom_encoder_kernel<GCN_D><<<atom_encoder_grid(n, GCN_C), 512, 0, s>>>(db.b.node_feature, d_nemb_, db.scratch, n, db.csr.err);
            }
            ProfScope p(prof, "gcn_dense", s);
            launch_dense(0, db.scratch, db.h[cur], n, db.range_flag, s);  // x_0 = W_0 h0 + b_0
        }
        for (int l = 1; l < GCN_L; l++) {
            if (split_ && !exact_ && fused_ && db.b.e_tot > 0) {
                ProfScope p(prof, "gcn_layer_fused", s);
                const long long wgs = ceil_div_ll(n, 128);
                const int grid = (int)(wgs < 512 ? wgs : 512);  // persistent: two 8-wave workgroups per CU (71 KB of LDS each)
                gcn_layer_fused_kernel<false><<<grid, 512, 0, s>>>(db.h[cur], db.h[cur ^ 1], db.csr.row_ptr, db.csr.src, db.csr.ecode, esc_.p,
                                                            db.csr.out_deg, d_ecomb_ + (size_t)(l - 1) * EDGE_COMBOS * GCN_D,
                                                            d_ep_ + (size_t)(l - 1) * 3 * GCN_D, d_split_ + (size_t)l * dense100_split_bytes(GCN_OT),
                                                            n, db.range_flag, nullptr);
                cur ^= 1;
                continue;
            }
            {
                ProfScope p(prof, "gcn_aggregate", s);
                launch_aggregate<true>(db, l - 1, db.h[cur], db.scratch, s);  // a_l
            }
            {
                ProfScope p(prof, "gcn_dense", s);
                launch_dense(l, db.scratch, db.h[cur ^ 1], n, db.range_flag, s);  // x_l
            }
            cur ^= 1;
        }
        db.final_h = cur;
        db.h_valid = true;
        if (split_ && !exact_ && fused_ && db.b.e_tot > 0 && num_tasks_ == 1) {
            // last stage: aggregation + BatchNorm with the readout's linear head folded in (per-node scores in db.scratch;
            // flowgnn_get_h returns x_4 = db.h[final_h], which is untouched by this)
            {
                ProfScope p(prof, "gcn_layer_fused", s);
                const long long wgs = ceil_div_ll(n, 128);
                const int grid = (int)(wgs < 512 ? wgs : 512);
                gcn_layer_fused_kernel<true><<<grid, 512, 0, s>>>(db.h[cur], db.scratch, db.csr.row_ptr, db.csr.src, db.csr.ecode, esc_.p,
                                                                  db.csr.out_deg, d_ecomb_ + (size_t)(GCN_L - 1) * EDGE_COMBOS * GCN_D,
                                                                  d_ep_ + (size_t)(GCN_L - 1) * 3 * GCN_D, nullptr, n, db.range_flag, d_pw_);
            }
            ProfScope p(prof, "mean_pool_linear", s);
            segment_mean_bias_kernel<0><<<(db.b.num_graphs + 255) / 256, 256, 0, s>>>(db.scratch, db.b.node_off, d_pb_, db.out, db.b.num_graphs);
            return 0;
        }
        {
            ProfScope p(prof, "gcn_aggregate", s);
            launch_aggregate<false>(db, GCN_L - 1, db.h[cur], db.scratch, s);  // BN_4(...), no ReLU
        }
        {
            ProfScope p(prof, "mean_pool_linear", s);
            if (num_tasks_ > 1) {  // NUM_TASK outputs per graph (linear_input_stationary over [NUM_TASK][100], GCN/src/finalize.cc:79-113)
                const int blocks = (db.b.num_graphs + 3) / 4;
                mean_pool_linear_mt_kernel<GCN_D><<<blocks < 512 ? blocks : 512, 256, 0, s>>>(db.scratch, db.b.node_off, d_pw_, d_pb_, db.out,
                                                                                              db.b.num_graphs, num_tasks_);
            } else
            mean_pool_linear_kernel<GCN_D><<<(db.b.num_graphs + 3) / 4, 256, 0, s>>>(db.scratch, db.b.node_off, d_pw_, d_pb_,
                                                                                     db.out, db.b.num_graphs);
        }
        return 0;
    }

    int set_num_tasks(int t) override {
        if (t < 1 || (t != 1 && qmode_)) return 8;
        if (t != num_tasks_) ready_ = false;  // graph_pred_weights / bias change shape: set the weights again
        num_tasks_ = t;
        return 0;
    }

    void configure(const Options& o) override {
        tile_nominal_ = o.i("tile_nominal") > 0 ? o.i("tile_nominal") : kTileNominal;  // <= 0 / < 0: back to the model's defaults
        tile_slack_ = o.i("tile_slack") >= 0 ? o.i("tile_slack") : kTileSlack;
        split_ = o.i("gcn_mfma") != 32;
        fused_ = !o.on("gcn_unfused");
        resident_ = o.on("gcn_resident");
        tile_build_ = o.i("gcn_tile_build") != 0;
        binpack_ = o.on("gcn_binpack");
        ablate_ = FG_ABLATE(o.i("gcn_ablate"));
        agg_ready_ = false;
    }
    void set_exact(bool on) override { exact_ = on; }

    int aggregation_only(DeviceBatch& db, int layer, hipStream_t s) override {
        if (qmode_) return 8;  // FLOWGNN_ERR_UNSUPPORTED: the fixed-point forward never builds the float kernels' inputs (tiles, h rows)
        if (layer < 0 || layer >= GCN_L) return 1;
        if (!agg_ready_) {  // the last forward ran the graph-resident kernel
            Profiler none;
            if (int rc = prepare_aggregate(db, none, s)) return rc;
        }
        if (!x0_in_hbm_) {  // ... behind the one-pass front end: no rows in HBM at all -- the probe reads x_0
            atom_encoder_kernel<GCN_D><<<atom_encoder_grid(db.b.n_tot, GCN_C), 512, 0, s>>>(db.b.node_feature, d_nemb_proj_, db.h[0], db.b.n_tot, db.csr.err);
            db.final_h = 0;
            x0_in_hbm_ = true;
        }
        launch_aggregate<true>(db, layer, db.h[db.final_h], db.scratch, s);
        return 0;
    }

private:
    void free_all() {
        float** ptrs[] = {&d_nemb_, &d_nemb_proj_, &d_enc_tab_, &d_pw_, &d_pb_, &d_ecomb_, &d_ep_, &d_wf_, &d_wt_, &d_bp_};
        for (auto p : ptrs)
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        if (d_split_) { (void)hipFree(d_split_); d_split_ = nullptr; }
        if (d_res_) { (void)hipFree(d_res_); d_res_ = nullptr; }
        esc_.release();
        tiles_.release();
        desc_.release();
        q_.release();
    }
    bool ready_ = false;
    bool qmode_ = false;  // flowgnn_set_numeric_mode(FLOWGNN_NUMERIC_Q6_10)
    QPack q_;
    GrowBufI desc_;               // gcn_tile_build_kernel: GCND_BYTES per graph tile
    float* d_enc_tab_ = nullptr;  // the projected table pre-combined into three rows per node (gin_resident_pack_enc_table)
    bool table_ok_ = true;             // the resident walk's scaled messages are exact for these weights (set_weights)
    std::vector<float> proj_max_src_;  // the projected table (host copy, for that check)
    bool x0_in_hbm_ = false;      // db.h[..] holds rows of the resident batch (false behind the one-pass front end and the fixed-point pass)
    bool binpack_ = true;         // gcn_binpack: the one-pass resident path walks bin-packed tile lists (GraphTiles::bp_*)
    bool tile_build_ = true;      // gcn_tile_build = 0: index build + projected encoder as separate launches in front of the resident kernel
    int num_tasks_ = 1;  // NUM_TASK (GCN/src/dcl.h) as a run-time dimension
    GrowBuf esc_;
    GrowBufI tiles_;  // graph-aligned tile starts of the resident batch (tile_bounds_kernel)
    static constexpr int kTileNominal = 96, kTileSlack = 32;  // the model's defaults of the options tile_nominal / tile_slack
    int tile_nominal_ = kTileNominal, tile_slack_ = kTileSlack;
    // gcn_mfma=32 keeps the dense layers on the fp32 matrix pipe (dense100_kernel); the default runs them as three
    // f16 MFMAs per product (dense_split.h), with the engine falling back to fp32 when the range flag trips
    bool split_ = true;
    bool exact_ = false;
    // gcn_unfused=1 keeps aggregate and dense as two kernels per layer (A/B measurements)
    bool fused_ = true;
    uint8_t* d_split_ = nullptr;
    uint8_t* d_res_ = nullptr;  // per-layer stream of gcn_resident_kernel
    int ablate_ = 0;  // development aid (-DFLOWGNN_DEV builds only, option gcn_ablate): per-phase timing (scripts/dev/pna_ablate.sh)
    bool resident_ = true;  // gcn_resident=0: one launch per layer
    bool keep_h_ = false;
    bool agg_ready_ = false;  // tiles_ / esc_ describe the batch of the last forward
    float* d_nemb_proj_ = nullptr;  // W_0 applied to the node-embedding table (+ b_0 in feature 0's rows): x_0 by lookups alone
    float *d_nemb_ = nullptr, *d_pw_ = nullptr, *d_pb_ = nullptr, *d_ecomb_ = nullptr, *d_ep_ = nullptr, *d_wf_ = nullptr,
          *d_wt_ = nullptr, *d_bp_ = nullptr;
};

Model* make_gcn_model() { return new GcnModel(); }

}  // namespace fg
