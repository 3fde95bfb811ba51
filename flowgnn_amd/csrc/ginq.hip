// GIN / GIN-VN forward in ap_fixed<16,6> (Q6.10), bit-faithful to the reference's arithmetic as restated in
// oracle/ginq_oracle.c (which lists the rules and why they are "parity unpinned": no Vitis headers here to run).
// Every value is a 16-bit two's-complement pattern with 10 fractional bits; the datapath reduces to
//     x + y -> (x + y) mod 2^16;   stored product a * w -> ((a * w) >> 10) mod 2^16 (arithmetic shift = floor);
//     r += a * w -> (r + ((a * w) >> 10)) mod 2^16;   relu = sign bit ? 0 : x;   mean = floor(sum / n).
// Sums are order independent (arithmetic mod 2^16), so the batched kernels below match the graph-at-a-time oracle
// bit for bit.  This is a fidelity mode, not the fast path: products are truncated one by one (no dot instructions,
// no MFMA), about 3 integer VALU operations per MAC -- 40 000 MACs per node and layer.
//   ginq_encoder_kernel   A5 : h0[v][d] = sum of 9 table rows                        (load_inputs.cc:203-209)
//   ginq_layer_kernel     A8 + A10: act = h[v] + sum_e relu(h[u] + ecomb[code]); hidden = b1 + sum floor(act W1);
//                         out = b2 + sum floor(relu(hidden) W2), relu unless last     (message_passing.cc:132-146,
//                                                                                      node_embedding.cc:103-201)
//   ginq_readout_kernel   A11: floor(sum_v h / n), then b + sum floor(hg w)           (finalize.cc:36-113, linear.cc)
#include "ginq.h"

#include <cmath>

#include "device_common.h"

namespace fg {

namespace {

constexpr int QD = 100, QH = 200, QL = 5, QKP = 104;  // K of the first linear layer padded to a multiple of 8
constexpr int QN = 32;                                  // nodes per workgroup tile

__device__ __forceinline__ int sx16(int x) { return (int)(short)x; }                 // low 16 bits, sign extended
__device__ __forceinline__ int lo16(unsigned p) { return (int)(short)(p & 0xFFFFu); }
__device__ __forceinline__ int hi16(unsigned p) { return (int)p >> 16; }
__device__ __forceinline__ unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16); }
__device__ __forceinline__ int relu16(int x) { return x < 0 ? 0 : x; }

// one thread per (node, pair of dims)
__global__ __launch_bounds__(256) void ginq_encoder_kernel(const int* __restrict__ node_feature, const int16_t* __restrict__ nemb,
                                                            int16_t* __restrict__ h, int n_tot, int* __restrict__ err) {
    const unsigned* t32 = reinterpret_cast<const unsigned*>(nemb);
    const long long total = (long long)n_tot * (QD / 2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int v = (int)(i / (QD / 2)), dp = (int)(i - (long long)v * (QD / 2));
        int lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) {
            int f = node_feature[(size_t)v * ND_FEATURE + k];
            if (f < 0 || f >= c_nd_card[k]) {
                atomicMax(err, ERR_NODE_FEAT);
                f = 0;
            }
            const unsigned p = t32[(size_t)(c_nd_off[k] + f) * (QD / 2) + dp];
            lo += lo16(p);
            hi += hi16(p);
        }
        reinterpret_cast<unsigned*>(h)[i] = pack16(lo, hi);  // mod 2^16 per half
    }
}

// Persistent 256-thread workgroups, tiles of 32 nodes.  Each thread keeps its row of W1 (thread o < 200) and of W2
// (thread d < 100 in each half of the workgroup) in registers for the whole kernel; activations go through LDS.
template <bool LAST>
__global__ __launch_bounds__(256) void ginq_layer_kernel(const int16_t* __restrict__ h, int16_t* __restrict__ hout,
                                                          const int* __restrict__ row_ptr, const int* __restrict__ src,
                                                          const uint8_t* __restrict__ ecode, const int16_t* __restrict__ ecomb,
                                                          const int16_t* __restrict__ w1, const int16_t* __restrict__ b1,
                                                          const int16_t* __restrict__ w2, const int16_t* __restrict__ b2,
                                                          int n_tot, int n_tiles) {
    __shared__ __attribute__((aligned(16))) int16_t s_act[QN][QKP];
    __shared__ __attribute__((aligned(16))) int16_t s_hid[QN][QH];
    const int tid = threadIdx.x;
    uint4 w1r[QKP / 8];  // this thread's row of W1: 104 patterns
    int bias1 = 0;
    if (tid < QH) {
#pragma unroll
        for (int i = 0; i < QKP / 8; i++) w1r[i] = reinterpret_cast<const uint4*>(w1 + (size_t)tid * QKP)[i];
        bias1 = b1[tid];
    }
    const int half = tid >> 7, dd = tid & 127;
    uint4 w2r[QH / 8];   // this thread's row of W2: 200 patterns
    int bias2 = 0;
    if (dd < QD) {
#pragma unroll
        for (int i = 0; i < QH / 8; i++) w2r[i] = reinterpret_cast<const uint4*>(w2 + (size_t)dd * QH)[i];
        bias2 = b2[dd];
    }
    const unsigned* h32 = reinterpret_cast<const unsigned*>(h);
    const unsigned* e32 = reinterpret_cast<const unsigned*>(ecomb);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int base = tile * QN;
        __syncthreads();  // the previous tile's readers are done
        // ---- message passing + the "+ h[v]" of the node transform: one thread per (node, pair of dims)
        for (int idx = tid; idx < QN * (QKP / 2); idx += 256) {
            const int v = idx / (QKP / 2), dp = idx - v * (QKP / 2);
            const int node = base + v;
            unsigned outp = 0;
            if (node < n_tot && dp < QD / 2) {
                const unsigned self = h32[(size_t)node * (QD / 2) + dp];
                int lo = lo16(self), hi = hi16(self);
                for (int e = row_ptr[node]; e < row_ptr[node + 1]; e++) {
                    const unsigned hu = h32[(size_t)src[e] * (QD / 2) + dp];
                    const unsigned ec = e32[(size_t)ecode[e] * (QD / 2) + dp];
                    lo += relu16(sx16(lo16(ec) + lo16(hu)));
                    hi += relu16(sx16(hi16(ec) + hi16(hu)));
                }
                outp = pack16(lo, hi);
            }
            reinterpret_cast<unsigned*>(&s_act[v][0])[dp] = outp;  // padding columns 100..103 and absent nodes: 0
        }
        __syncthreads();
        // ---- first linear layer: thread o, all 32 nodes
        if (tid < QH) {
#pragma unroll 1
            for (int v = 0; v < QN; v++) {
                int acc = bias1;
                const uint4* a4 = reinterpret_cast<const uint4*>(&s_act[v][0]);
#pragma unroll
                for (int i = 0; i < QKP / 8; i++) {
                    const uint4 a = a4[i];  // same address in every lane: an LDS broadcast
                    uint4 w = w1r[i];
                    asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w));  // keeps the 304 sign-extended halves of the two rows from being hoisted out of the tile loop (registers)
                    acc += (lo16(a.x) * lo16(w.x)) >> 10; acc += (hi16(a.x) * hi16(w.x)) >> 10;
                    acc += (lo16(a.y) * lo16(w.y)) >> 10; acc += (hi16(a.y) * hi16(w.y)) >> 10;
                    acc += (lo16(a.z) * lo16(w.z)) >> 10; acc += (hi16(a.z) * hi16(w.z)) >> 10;
                    acc += (lo16(a.w) * lo16(w.w)) >> 10; acc += (hi16(a.w) * hi16(w.w)) >> 10;
                }
                s_hid[v][tid] = (int16_t)relu16(sx16(acc));  // the relu of node_embedding.cc:180, applied once here
            }
        }
        __syncthreads();
        // ---- second linear layer: thread d of each half, 16 nodes per half
        if (dd < QD) {
#pragma unroll 1
            for (int v = half * (QN / 2); v < (half + 1) * (QN / 2); v++) {
                const int node = base + v;
                if (node >= n_tot) break;
                int r = bias2;
                const uint4* a4 = reinterpret_cast<const uint4*>(&s_hid[v][0]);
#pragma unroll
                for (int i = 0; i < QH / 8; i++) {
                    const uint4 a = a4[i];
                    uint4 w = w2r[i];
                    asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w));
                    r += (lo16(a.x) * lo16(w.x)) >> 10; r += (hi16(a.x) * hi16(w.x)) >> 10;
                    r += (lo16(a.y) * lo16(w.y)) >> 10; r += (hi16(a.y) * hi16(w.y)) >> 10;
                    r += (lo16(a.z) * lo16(w.z)) >> 10; r += (hi16(a.z) * hi16(w.z)) >> 10;
                    r += (lo16(a.w) * lo16(w.w)) >> 10; r += (hi16(a.w) * hi16(w.w)) >> 10;
                }
                r = sx16(r);
                if (!LAST) r = relu16(r);
                hout[(size_t)node * QD + dd] = (int16_t)r;
            }
        }
    }
}

// one wave per graph; lane l < 50 owns dims 2l, 2l+1
__global__ __launch_bounds__(256) void ginq_readout_kernel(const int16_t* __restrict__ h, const int* __restrict__ node_off,
                                                            const int16_t* __restrict__ pw, const int16_t* __restrict__ pb,
                                                            float* __restrict__ out, int num_graphs) {
    const int lane = threadIdx.x & 63;
    const int gidx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gidx >= num_graphs) return;
    const int n0 = node_off[gidx], n1 = node_off[gidx + 1], n = n1 - n0;
    int part = 0;
    if (lane < QD / 2) {
        int lo = 0, hi = 0;
        for (int v = n0; v < n1; v++) {
            const unsigned p = reinterpret_cast<const unsigned*>(h)[(size_t)v * (QD / 2) + lane];
            lo += lo16(p);
            hi += hi16(p);
        }
        lo = sx16(lo); hi = sx16(hi);
        // floor division by n > 0 (finalize.cc:112 as stored into FM_TYPE)
        int ql = lo / n, qh = hi / n;
        if (lo % n != 0 && lo < 0) ql--;
        if (hi % n != 0 && hi < 0) qh--;
        const unsigned w = reinterpret_cast<const unsigned*>(pw)[lane];
        part = ((sx16(ql) * lo16(w)) >> 10) + ((sx16(qh) * hi16(w)) >> 10);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if (lane == 0) out[gidx] = (float)sx16(part + (int)pb[0]) / 1024.0f;
}

int16_t q16_from_float(float x) {
    const double f = std::floor((double)x * 1024.0);
    return (int16_t)(uint16_t)(uint64_t)(long long)f;
}

}  // namespace

void GinQWeights::release() {
    int16_t** ptrs[] = {&nemb, &ecomb, &w1, &b1, &w2, &b2, &pw, &pb};
    for (auto p : ptrs)
        if (*p) { (void)hipFree(*p); *p = nullptr; }
}

int ginq_upload(GinQWeights& q, const float* nemb, const float* eemb, const float* w1, const float* b1, const float* w2, const float* b2,
                const float* pw, const float* pb) {
    std::vector<int16_t> v_nemb((size_t)ND_FEATURE_TOTAL * QD), v_ecomb((size_t)QL * EDGE_COMBOS * QD), v_w1((size_t)QL * QH * QKP, 0),
        v_b1((size_t)QL * QH), v_w2((size_t)QL * QD * QH), v_b2((size_t)QL * QD), v_pw(QD), v_pb(1);
    for (size_t i = 0; i < v_nemb.size(); i++) v_nemb[i] = q16_from_float(nemb[i]);
    static const int ed_off[3] = {0, 5, 11};  // message_passing.cc:3
    for (int l = 0; l < QL; l++) {
        const float* E = eemb + (size_t)l * ED_FEATURE_PER_LAYER * QD;
        for (int a0 = 0; a0 < 5; a0++)
            for (int a1 = 0; a1 < 6; a1++)
                for (int a2 = 0; a2 < 2; a2++) {
                    const int code = (a0 * 6 + a1) * 2 + a2;
                    for (int d = 0; d < QD; d++) {
                        int s = 0;  // FM_TYPE edge_embed accumulates the three quantised rows with wrap (message_passing.cc:136-141)
                        s += q16_from_float(E[(ed_off[0] + a0) * QD + d]);
                        s += q16_from_float(E[(ed_off[1] + a1) * QD + d]);
                        s += q16_from_float(E[(ed_off[2] + a2) * QD + d]);
                        v_ecomb[((size_t)l * EDGE_COMBOS + code) * QD + d] = (int16_t)(uint16_t)(unsigned)s;
                    }
                }
        for (int o = 0; o < QH; o++)
            for (int k = 0; k < QD; k++) v_w1[((size_t)l * QH + o) * QKP + k] = q16_from_float(w1[((size_t)l * QH + o) * QD + k]);
    }
    for (size_t i = 0; i < v_b1.size(); i++) v_b1[i] = q16_from_float(b1[i]);
    for (size_t i = 0; i < v_w2.size(); i++) v_w2[i] = q16_from_float(w2[i]);
    for (size_t i = 0; i < v_b2.size(); i++) v_b2[i] = q16_from_float(b2[i]);
    for (int i = 0; i < QD; i++) v_pw[i] = q16_from_float(pw[i]);
    v_pb[0] = q16_from_float(pb[0]);
    int rc;
    if ((rc = upload(&q.nemb, v_nemb))) return rc;
    if ((rc = upload(&q.ecomb, v_ecomb))) return rc;
    if ((rc = upload(&q.w1, v_w1))) return rc;
    if ((rc = upload(&q.b1, v_b1))) return rc;
    if ((rc = upload(&q.w2, v_w2))) return rc;
    if ((rc = upload(&q.b2, v_b2))) return rc;
    if ((rc = upload(&q.pw, v_pw))) return rc;
    if ((rc = upload(&q.pb, v_pb))) return rc;
    return 0;
}

int ginq_forward(const GinQWeights& q, DeviceBatch& db, Profiler& prof, hipStream_t s) {
    const int n = db.b.n_tot;
    if (n <= 0) return 0;
    int16_t* hq[2] = {reinterpret_cast<int16_t*>(db.h[0]), reinterpret_cast<int16_t*>(db.h[1])};  // [N][100] int16 in the float buffers
    {
        ProfScope p(prof, "ginq_encoder", s);
        ginq_encoder_kernel<<<grid_for((long long)n * (QD / 2), 256, 256 * 8), 256, 0, s>>>(db.b.node_feature, q.nemb, hq[0], n, db.csr.err);
    }
    const int n_tiles = (int)ceil_div_ll(n, QN);
    const int grid = n_tiles < 512 ? n_tiles : 512;
    int cur = 0;
    for (int l = 0; l < QL; l++) {
        ProfScope p(prof, "ginq_layer", s);
        const int16_t* ec = q.ecomb + (size_t)l * EDGE_COMBOS * QD;
        const int16_t *w1 = q.w1 + (size_t)l * QH * QKP, *b1 = q.b1 + (size_t)l * QH, *w2 = q.w2 + (size_t)l * QD * QH, *b2 = q.b2 + (size_t)l * QD;
        if (l == QL - 1)
            ginq_layer_kernel<true><<<grid, 256, 0, s>>>(hq[cur], hq[cur ^ 1], db.csr.row_ptr, db.csr.src, db.csr.ecode, ec, w1, b1, w2, b2, n, n_tiles);
        else
            ginq_layer_kernel<false><<<grid, 256, 0, s>>>(hq[cur], hq[cur ^ 1], db.csr.row_ptr, db.csr.src, db.csr.ecode, ec, w1, b1, w2, b2, n, n_tiles);
        cur ^= 1;
    }
    db.final_h = cur;
    db.h_valid = false;  // the node embeddings are int16 patterns, not the float rows flowgnn_get_h promises
    {
        ProfScope p(prof, "ginq_readout", s);
        ginq_readout_kernel<<<(db.b.num_graphs + 3) / 4, 256, 0, s>>>(hq[cur], db.b.node_off, q.pw, q.pb, db.out, db.b.num_graphs);
    }
    return 0;
}

}  // namespace fg
