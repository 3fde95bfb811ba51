// dgn.hip -- DGN hot path for gfx950 (MI355X).
//
// Reference per graph (DGN/src/*.cc), 4 layers, dim 100, no edge features, eigenvector #1 per node:
//   h0[v]   = sum_{k<9} Table[k][feat_k(v)]              (dense [9][119][100] table)   load_inputs.cc:114-172
//   w_e     = eig1[u] - eig1[v]  for edge (u -> v);  abssum[v] = sum |w_e|,  wsum[v] = sum w_e   load_inputs.cc:92-111
//   m1[v]   = sum h[u],   m2[v] = sum h[u] w_e                                          message_passing.cc:148-149
//   a1      = m1 / outdeg(v)   (x / 0 = 0, see oracle/dgn_oracle.c),
//   a2      = | (m2 - wsum[v] h[v]) / abssum[v] |   (abssum 0 -> 2^-13)                node_embedding.cc:125-146
//   h'[v]   = h[v] + relu(b + W[:,0,:] a1 + W[:,1,:] a2)       W viewed as [100][2][100] node_embedding.cc:148-181
//   out[g]  = head(mean_v h_4[v]),  head = 100 -> 50 (ReLU) -> 25 (ReLU) -> 1            finalize.cc:28-52
//
// Here: one HBM-bound aggregation kernel per layer that writes z[v] = [a1 | a2] (the directional weights are
// recomputed from the eigenvector column, 4 B per node, instead of being stored per edge), one fp32-MFMA
// dense kernel (K = 200) with the residual in its epilogue, one wave-per-graph readout kernel.
#include "common.h"
#include "device_common.h"
#include "modelq.h"
#include "dense_split.h"
#include <cmath>
#include <cstring>

namespace fg {

constexpr int DGN_D = 100;
constexpr int DGN_L = 4;
constexpr int DGN_C = DGN_D / 4;
constexpr int DGN_OT = 7;
constexpr int DGN_TBL = 119;


// z[v] = [a1 | a2]: policy of the generic tiled aggregation (device_common.h).  The directional weight of an edge
// is eig1[u] - eig1[v]: the source half is staged per CSR entry, the destination half is read once per item.
struct DgnAggPolicy {
    static constexpr int D = DGN_D, TR = 128, NTHR = 512, TE = 20 * 128, TABLE_ROWS = 0;
    static constexpr bool HAS_SCALAR = true;
    static constexpr int NDST = 2, CONST_FLOATS = 0;  // eig1[v], outdeg(v)
    struct Params {
        const float* eig;     // [N][4]
        const int* out_deg;
        const float* esc;     // [E] eig1[src_e] in CSR order (edge_scalar_kernel)
    };
    struct Acc { float4 m1, m2; float wsum, abssum; };
    __device__ static float src_scalar(const Params& p, int u) { return p.eig[(size_t)u * 4 + 1]; }
    __device__ static void dst_stage(const Params& p, int v, float* o) {
        o[0] = p.eig[(size_t)v * 4 + 1];
        o[1] = (float)p.out_deg[v];
    }
    __device__ static const float* const_ptr(const Params&) { return nullptr; }
    __device__ static void init(Acc& a) {
        a.m1 = make_float4(0.f, 0.f, 0.f, 0.f);
        a.m2 = a.m1;
        a.wsum = 0.f;
        a.abssum = 0.f;
    }
    __device__ static void edge(Acc& a, const float4& x, const float4&, float ss, const float* sd) {
        const float w = ss - sd[0];
        a.wsum += w;
        a.abssum += fabsf(w);
        a.m1.x += x.x; a.m1.y += x.y; a.m1.z += x.z; a.m1.w += x.w;
        a.m2.x += x.x * w; a.m2.y += x.y * w; a.m2.z += x.z * w; a.m2.w += x.w * w;
    }
    __device__ static void finish(const Params&, const Acc& a, const float4& hv, int v, int c, int, const float* sd, const float*,
                                  float* out) {
        const float abssum = a.abssum == 0.0f ? 1.0f / 8192.0f : a.abssum;  // epsilon of ap_fixed<16,3>
        const float deg = sd[1];
        const bool dv = deg == 0.0f;
        float4 a1, a2;
        a1.x = dv ? 0.f : a.m1.x / deg; a1.y = dv ? 0.f : a.m1.y / deg;
        a1.z = dv ? 0.f : a.m1.z / deg; a1.w = dv ? 0.f : a.m1.w / deg;
        a2.x = fabsf((a.m2.x - a.wsum * hv.x) / abssum); a2.y = fabsf((a.m2.y - a.wsum * hv.y) / abssum);
        a2.z = fabsf((a.m2.z - a.wsum * hv.z) / abssum); a2.w = fabsf((a.m2.w - a.wsum * hv.w) / abssum);
        float4* o = reinterpret_cast<float4*>(out) + (size_t)v * (2 * DGN_C) + c;
        stream_store4(o, a1);
        stream_store4(o + DGN_C, a2);
    }
};

// h'[v] = h[v] + relu(b + W0 a1 + W1 a2), K = 2 x 100.  One wave = 16 nodes; fragments [2][7][6][64][4] + tails [2][7][64]
__global__ __launch_bounds__(256) void dgn_dense_kernel(const float* __restrict__ z, const float* __restrict__ h,
                                                         float* __restrict__ hout, const float* __restrict__ wf,
                                                         const float* __restrict__ wtail, const float* __restrict__ biasp,
                                                         int n_tot) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const long long node_base = (long long)wave * 16;
    if (node_base >= n_tot) return;
    long long node = node_base + j;
    const bool valid = node < n_tot;
    if (!valid) node = n_tot - 1;
    float bq[2][25];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const float* row = z + ((size_t)node * 2 + b) * DGN_D;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float4 x = *reinterpret_cast<const float4*>(row + 16 * q + 4 * g);
            bq[b][4 * q + 0] = x.x; bq[b][4 * q + 1] = x.y; bq[b][4 * q + 2] = x.z; bq[b][4 * q + 3] = x.w;
        }
        bq[b][24] = row[96 + g];
    }
    const float4* wf4 = reinterpret_cast<const float4*>(wf);
#pragma unroll 1
    for (int t = 0; t < DGN_OT; t++) {
        const float4 bb = *reinterpret_cast<const float4*>(biasp + 16 * t + 4 * g);
        float4_t y0 = (float4_t){bb.x, bb.y, bb.z, bb.w}, y1 = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 2; b++) {
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const float4 af = wf4[(((size_t)b * DGN_OT + t) * 6 + q) * 64 + lane];
                y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bq[b][4 * q + 0], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bq[b][4 * q + 1], y1, 0, 0, 0);
                y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bq[b][4 * q + 2], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bq[b][4 * q + 3], y1, 0, 0, 0);
            }
            const float at = wtail[((size_t)b * DGN_OT + t) * 64 + lane];
            y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(at, bq[b][24], y0, 0, 0, 0);
        }
        const int col = 16 * t + 4 * g;
        if (valid && col < DGN_D) {
            const size_t off = (size_t)node * DGN_D + col;
            const float4 hv = *reinterpret_cast<const float4*>(h + off);
            *reinterpret_cast<float4*>(hout + off) = make_float4(hv.x + relu1(y0.x + y1.x), hv.y + relu1(y0.y + y1.y),
                                                                 hv.z + relu1(y0.z + y1.z), hv.w + relu1(y0.w + y1.w));
        }
    }
}

class DgnModel : public Model {
public:
    ~DgnModel() override { free_all(); }
    int emb_dim() const override { return DGN_D; }
    int scratch_dim() const override { return 2 * DGN_D; }
    int aggregate_dim() const override { return 2 * DGN_D; }
    bool has_edge_attr() const override { return false; }
    int num_weight_tensors() const override { return 9; }
    bool weights_ready() const override { return ready_; }

    // host tensors (DGN/src/dcl.h:81-90): atom tables [9][119][100], layer W [4][100][200], b [4][100],
    // FC0 w [50][100] b [50], FC1 w [25][50] b [25], FC2 w [1][25] b [1]
    int set_numeric_mode(int mode) override {  // mode 1 = "the reference's own format": ap_fixed<16,3> for DGN (DGN/src/dcl.h:54-55)
        if (mode != 0 && mode != 1) return 8;
        qmode_ = mode == 1;
        return 0;
    }

    int set_weights(const float* const* t) override {
        {   // ap_fixed<16,3> copies of every tensor for the bit-faithful mode (modelq.hip)
            static const size_t elems[9] = {9 * 119 * 100, 4 * 100 * 200, 4 * 100, 50 * 100, 50, 25 * 50, 25, 25, 1};
            if (int rc = q_.upload_all(9, t, elems, 13)) return rc;
        }
        // The reference indexes a dense [9][119][100] table (DGN/src/load_inputs.cc:124-137), but a feature k only takes values
        // below its cardinality (validated on the device): the 173 rows that can be addressed are gathered into the compact
        // [173][100] table the other models use (row = offset_k + value), which fits LDS (69 KB) -- the encoder then reads its nine
        // rows per node from LDS instead of L2 (0.37 -> 0.13 ms for 2^15 hep10k graphs).
        static const int nd_card[ND_FEATURE] = {119, 4, 12, 12, 10, 6, 6, 2, 2}, nd_off[ND_FEATURE] = {0, 119, 123, 135, 147, 157, 163, 169, 171};
        std::vector<float> v_emb((size_t)ND_FEATURE_TOTAL * DGN_D);
        for (int k = 0; k < ND_FEATURE; k++)
            for (int f = 0; f < nd_card[k]; f++)
                memcpy(&v_emb[(size_t)(nd_off[k] + f) * DGN_D], t[0] + ((size_t)k * DGN_TBL + f) * DGN_D, sizeof(float) * DGN_D);
        std::vector<float> v_w0(t[3], t[3] + 50 * 100), v_b0(t[4], t[4] + 50), v_w1(t[5], t[5] + 25 * 50), v_b1(t[6], t[6] + 25),
            v_w2(t[7], t[7] + 25), v_b2(t[8], t[8] + 1);
        std::vector<float> wf_all, wt_all, bp_all;
        std::vector<uint8_t> split_all;
        std::vector<float> Wb((size_t)DGN_D * DGN_D), zero(DGN_D, 0.0f);
        for (int l = 0; l < DGN_L; l++) {
            const float* W = t[1] + (size_t)l * DGN_D * 2 * DGN_D;  // [out][2][in]
            std::vector<float> wf2, wt2;
            std::vector<float> bp;
            for (int b = 0; b < 2; b++) {
                for (int o = 0; o < DGN_D; o++)
                    for (int i = 0; i < DGN_D; i++) Wb[(size_t)o * DGN_D + i] = W[((size_t)o * 2 + b) * DGN_D + i];
                std::vector<float> wf, wt, bpp;
                pack_dense100(Wb.data(), b == 0 ? t[2] + (size_t)l * DGN_D : zero.data(), DGN_D, DGN_OT, wf, wt, bpp);
                wf2.insert(wf2.end(), wf.begin(), wf.end());
                wt2.insert(wt2.end(), wt.begin(), wt.end());
                if (b == 0) bp = bpp;
            }
            wf_all.insert(wf_all.end(), wf2.begin(), wf2.end());
            wt_all.insert(wt_all.end(), wt2.begin(), wt2.end());
            bp_all.insert(bp_all.end(), bp.begin(), bp.end());
            const size_t off = split_all.size();
            split_all.resize(off + dense200_split_bytes(DGN_OT));
            pack_dense200_split(W, t[2] + (size_t)l * DGN_D, DGN_D, DGN_OT, split_all.data() + off);
        }
        int rc;
        if ((rc = upload(&d_split_, split_all))) return rc;
        if ((rc = upload(&d_emb_, v_emb))) return rc;
        if ((rc = upload(&d_wf_, wf_all))) return rc;
        if ((rc = upload(&d_wt_, wt_all))) return rc;
        if ((rc = upload(&d_bp_, bp_all))) return rc;
        if ((rc = upload(&d_w0_, v_w0))) return rc;
        if ((rc = upload(&d_b0_, v_b0))) return rc;
        if ((rc = upload(&d_w1_, v_w1))) return rc;
        if ((rc = upload(&d_b1_, v_b1))) return rc;
        if ((rc = upload(&d_w2_, v_w2))) return rc;
        if ((rc = upload(&d_b2_, v_b2))) return rc;
        ready_ = true;
        return 0;
    }

    // DGN/src/host_load.cc:11-149: nine atom tables packed at 0, 11900, 12300, 13500, 14700, 15700, 16300, 16900,
    // 17100 into a [9][119][100] array; layer l at 17300 + 20100 l; head at 97700 ...
    int load_weights_dir(const char* dir) override {
        const char* f = "dgn_ep1_noBN_dim100.weights.all.bin";
        static const int card[9] = {119, 4, 12, 12, 10, 6, 6, 2, 2};
        static const size_t toff[9] = {0, 11900, 12300, 13500, 14700, 15700, 16300, 16900, 17100};
        std::vector<float> emb((size_t)9 * DGN_TBL * DGN_D, 0.0f), lw((size_t)4 * 20000), lb(400), w0(5000), b0(50), w1(1250),
            b1(25), w2(25), b2(1);
        int rc;
        for (int k = 0; k < 9; k++)
            if ((rc = read_floats(dir, f, toff[k], (size_t)card[k] * 100, &emb[(size_t)k * DGN_TBL * DGN_D]))) return rc;
        for (int l = 0; l < DGN_L; l++) {
            const size_t base = 17300 + 20100 * (size_t)l;
            if ((rc = read_floats(dir, f, base, 20000, &lw[(size_t)l * 20000]))) return rc;
            if ((rc = read_floats(dir, f, base + 20000, 100, &lb[l * 100]))) return rc;
        }
        if ((rc = read_floats(dir, f, 97700, 5000, w0.data()))) return rc;
        if ((rc = read_floats(dir, f, 102700, 50, b0.data()))) return rc;
        if ((rc = read_floats(dir, f, 102750, 1250, w1.data()))) return rc;
        if ((rc = read_floats(dir, f, 104000, 25, b1.data()))) return rc;
        if ((rc = read_floats(dir, f, 104025, 25, w2.data()))) return rc;
        if ((rc = read_floats(dir, f, 104050, 1, b2.data()))) return rc;
        const float* t[9] = {emb.data(), lw.data(), lb.data(), w0.data(), b0.data(), w1.data(), b1.data(), w2.data(), b2.data()};
        return set_weights(t);
    }

    void launch_aggregate(const DeviceBatch& db, const float* hin, hipStream_t s) {
        DgnAggPolicy::Params prm{db.node_eigen, db.csr.out_deg, esc_.p};
        launch_tiled_aggregate<DgnAggPolicy>(prm, hin, db.scratch, db.csr, nullptr, db.b.n_tot, tiles_.p, tile_nominal_, s);
    }

    int forward(DeviceBatch& db, Profiler& prof, hipStream_t s) override {
        const int n = db.b.n_tot;
        if (n <= 0) return 0;
        if (!db.node_eigen) return 1;
        if (qmode_) return dgnq_forward(q_, db, prof, s);
        {
            ProfScope p(prof, "atom_encoder", s);
            atom_encoder_kernel<DGN_D><<<atom_encoder_grid(n, DGN_C), 512, 0, s>>>(db.b.node_feature, d_emb_, db.h[0], n, db.csr.err);
        }
        if (int rc = make_tile_bounds(tiles_, db.b.node_off, db.b.num_graphs, n, tile_nominal_, tile_slack_, s)) return rc;
        if (db.b.e_tot > 0) {  // eig1[src_e] per CSR entry, once per pass
            if (int rc = esc_.reserve((size_t)db.b.e_tot)) return rc;
            ProfScope p(prof, "edge_scalar", s);
            DgnAggPolicy::Params prm{db.node_eigen, db.csr.out_deg, nullptr};
            edge_scalar_kernel<DgnAggPolicy><<<grid_for(db.b.e_tot, 256, 256 * 8), 256, 0, s>>>(prm, db.csr.src, esc_.p, db.b.e_tot);
        }
        int cur = 0;
        for (int l = 0; l < DGN_L; l++) {
            {
                ProfScope p(prof, "dgn_aggregate", s);
                launch_aggregate(db, db.h[cur], s);
            }
            {
                ProfScope p(prof, "dgn_dense", s);
                const int waves = (int)ceil_div_ll(n, 16);
                if (split_ && !exact_) {
                    const long long wgs = ceil_div_ll(n, 256);
                    dense200_res_relu_split_kernel<DGN_OT><<<(int)(wgs < 256 ? wgs : 256), 1024, 0, s>>>(
                        db.scratch, db.h[cur], db.h[cur ^ 1], d_split_ + (size_t)l * dense200_split_bytes(DGN_OT), n, DGN_D, db.range_flag);
                } else
                dgn_dense_kernel<<<(waves + 3) / 4, 256, 0, s>>>(db.scratch, db.h[cur], db.h[cur ^ 1],
                                                                  d_wf_ + (size_t)l * 2 * DGN_OT * 6 * 64 * 4,
                                                                  d_wt_ + (size_t)l * 2 * DGN_OT * 64, d_bp_ + (size_t)l * DGN_OT * 16, n);
            }
            cur ^= 1;
        }
        db.final_h = cur;
        {
            ProfScope p(prof, "pool_mlp3", s);
            pool_mlp3_kernel<DGN_D, 50, 25><<<(db.b.num_graphs + 3) / 4, 256, 0, s>>>(db.h[cur], db.b.node_off, d_w0_, d_b0_, d_w1_,
                                                                                      d_b1_, d_w2_, d_b2_, db.out, db.b.num_graphs);
        }
        return 0;
    }

    void set_exact(bool on) override { exact_ = on; }

    int aggregation_only(DeviceBatch& db, int layer, hipStream_t s) override {
        if (layer < 0 || layer >= DGN_L) return 1;
        launch_aggregate(db, db.h[db.final_h], s);
        return 0;
    }

private:
    void free_all() {
        float** ptrs[] = {&d_emb_, &d_wf_, &d_wt_, &d_bp_, &d_w0_, &d_b0_, &d_w1_, &d_b1_, &d_w2_, &d_b2_};
        for (auto p : ptrs)
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        esc_.release();
        tiles_.release();
        q_.release();
        if (d_split_) { (void)hipFree(d_split_); d_split_ = nullptr; }
    }
    bool ready_ = false;
    bool qmode_ = false;  // flowgnn_set_numeric_mode(FLOWGNN_NUMERIC_Q6_10): ap_fixed<16,3> arithmetic
    QPack q_;
    GrowBufI tiles_;  // graph-aligned tile starts of the resident batch (tile_bounds_kernel)
    int tile_nominal_ = getenv("FLOWGNN_TILE_NOMINAL") ? atoi(getenv("FLOWGNN_TILE_NOMINAL")) : 64;
    int tile_slack_ = getenv("FLOWGNN_TILE_SLACK") ? atoi(getenv("FLOWGNN_TILE_SLACK")) : 64;
    // FLOWGNN_DGN_MFMA=f32 keeps the dense update on the fp32 matrix pipe (dgn_dense_kernel)
    bool split_ = !(getenv("FLOWGNN_DGN_MFMA") && strcmp(getenv("FLOWGNN_DGN_MFMA"), "f32") == 0);
    bool exact_ = false;
    uint8_t* d_split_ = nullptr;
    GrowBuf esc_;
    float *d_emb_ = nullptr, *d_wf_ = nullptr, *d_wt_ = nullptr, *d_bp_ = nullptr, *d_w0_ = nullptr, *d_b0_ = nullptr,
          *d_w1_ = nullptr, *d_b1_ = nullptr, *d_w2_ = nullptr, *d_b2_ = nullptr;
};

Model* make_dgn_model() { return new DgnModel(); }

}  // namespace fg
