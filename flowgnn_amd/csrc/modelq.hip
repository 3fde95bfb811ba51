// GCN, GAT, PNA (ap_fixed<16,6>, Q6.10) and DGN (ap_fixed<16,3>, Q3.13) in the reference's own number formats: the bit-faithful
// mode of SURVEY 8f rank 2 for the models ginq.hip does not cover.  The arithmetic rules (R0..R8: what is stored where, how
// ap_fixed division truncates, how hls::sqrt / recip / exp / log / abs are read) are written down in oracle/q_oracle.c, which is
// the CPU statement of the same thing; under them every value is a 16-bit two's-complement pattern, every statement integer
// arithmetic and every sum is taken mod 2^16 -- independent of its order, so the batched kernels below (one thread per output
// element, in-edges from the destination-major CSR) match the graph-at-a-time oracle bit for bit.
// This is a fidelity mode, not the fast path: products are truncated one by one, no MFMA, weights read through the caches.
#include "modelq.h"

#include <cmath>

#include "device_common.h"

namespace fg {

namespace {

__device__ __forceinline__ int sx16(int x) { return (int)(short)x; }
__device__ __forceinline__ int relu16(int x) { return x < 0 ? 0 : x; }
__device__ __forceinline__ int abs16(int x) { return sx16(x < 0 ? -x : x); }                        // R6
__device__ __forceinline__ int floordiv(int s, int n) { int q = s / n; if ((s % n != 0) && ((s < 0) != (n < 0))) q--; return q; }
__device__ __forceinline__ int div_int(int a, int n) { return n == 0 ? 0 : sx16(floordiv(a, n)); }  // R2
// R1: a_exact has fa fractional bits, b is a 16-bit pattern of a format with F fractional / 16 - F integer bits
__device__ __forceinline__ int qdiv(long long a_exact, int fa, int b, int F) {
    if (b == 0) return 0;
    const long long q = (a_exact * 65536ll) / (long long)b;  // F + I = 16
    return sx16((int)(q >> (fa + 16 - F - F)));
}
__device__ __forceinline__ unsigned isqrt64(unsigned long long x) {
    unsigned long long r = (unsigned long long)sqrt((double)x);
    while (r * r > x) r--;
    while ((r + 1) * (r + 1) <= x) r++;
    return (unsigned)r;
}
__device__ __forceinline__ int qsqrt(int pattern, int F) { return pattern <= 0 ? 0 : sx16((int)isqrt64((unsigned long long)pattern << F)); }  // R3
__device__ __forceinline__ int qrecip(int p, int F) { return p <= 0 ? 0 : sx16((int)((1ll << (2 * F)) / p)); }                                  // R4

// ---------------------------------------------------------------- shared kernels
// atom encoder: h[v][d] = sum of 9 table rows (wrap); rows = c_nd_off[k] + f (173-row table) or k * 119 + f (DGN's dense tables)
template <int D, bool DENSE9>
__global__ __launch_bounds__(256) void q_encoder_kernel(const int* __restrict__ nf, const int16_t* __restrict__ table, int16_t* __restrict__ h,
                                                         int n_tot, int* __restrict__ err) {
    const long long total = (long long)n_tot * D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int v = (int)(i / D), d = (int)(i - (long long)v * D);
        int s = 0;
#pragma unroll
        for (int k = 0; k < ND_FEATURE; k++) {
            int f = nf[(size_t)v * ND_FEATURE + k];
            if (f < 0 || f >= c_nd_card[k]) { atomicMax(err, ERR_NODE_FEAT); f = 0; }
            s += table[(size_t)((DENSE9 ? k * 119 : c_nd_off[k]) + f) * D + d];
        }
        h[i] = (int16_t)s;
    }
}
// out[v][o] = wrap(bias[o] + sum_i floor(in[v][i] w[o][i]))  (+ ReLU) (+ residual: res[v][o] + relu(...))
template <int F>
__global__ __launch_bounds__(256) void q_dense_kernel(const int16_t* __restrict__ in, int K, const int16_t* __restrict__ w,
                                                       const int16_t* __restrict__ bias, int16_t* __restrict__ out, int DO, int n_tot,
                                                       int relu, const int16_t* __restrict__ res) {
    const long long total = (long long)n_tot * DO;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx / DO), o = (int)(idx - (long long)v * DO);
        int acc = bias[o];
        const int16_t* a = in + (size_t)v * K;
        const int16_t* wr = w + (size_t)o * K;
        for (int i = 0; i < K; i++) acc += ((int)a[i] * (int)wr[i]) >> F;
        acc = sx16(acc);
        if (relu) acc = relu16(acc);
        if (res) acc = sx16((int)res[idx] + relu16(acc));
        out[idx] = (int16_t)acc;
    }
}
// readout: floor(sum_v h / n) per dim, then up to three linear layers (ReLU between), one workgroup of 128 threads per graph
template <int F, int D, int M1, int M2>
__global__ __launch_bounds__(128) void q_pool_head_kernel(const int16_t* __restrict__ h, const int* __restrict__ node_off,
                                                           const int16_t* __restrict__ w1, const int16_t* __restrict__ b1,
                                                           const int16_t* __restrict__ w2, const int16_t* __restrict__ b2,
                                                           const int16_t* __restrict__ w3, const int16_t* __restrict__ b3,
                                                           float* __restrict__ out, int num_graphs) {
    __shared__ int s_hg[D], s_o1[M1 > 0 ? M1 : 1], s_o2[M2 > 0 ? M2 : 1];
    const int g = blockIdx.x;
    if (g >= num_graphs) return;
    const int n0 = node_off[g], n1 = node_off[g + 1], n = n1 - n0;
    for (int d = threadIdx.x; d < D; d += 128) {
        int s = 0;
        for (int v = n0; v < n1; v++) s += h[(size_t)v * D + d];
        s_hg[d] = div_int(sx16(s), n);
    }
    __syncthreads();
    if constexpr (M1 == 0) {  // single linear head (GCN, GAT)
        if (threadIdx.x == 0) {
            int r = b1[0];
            for (int i = 0; i < D; i++) r += (s_hg[i] * (int)w1[i]) >> F;
            out[g] = (float)sx16(r) / (float)(1 << F);
        }
    } else {
        for (int o = threadIdx.x; o < M1; o += 128) {
            int r = b1[o];
            for (int i = 0; i < D; i++) r += (s_hg[i] * (int)w1[o * D + i]) >> F;
            s_o1[o] = relu16(sx16(r));
        }
        __syncthreads();
        for (int o = threadIdx.x; o < M2; o += 128) {
            int r = b2[o];
            for (int i = 0; i < M1; i++) r += (s_o1[i] * (int)w2[o * M1 + i]) >> F;
            s_o2[o] = relu16(sx16(r));
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int r = b3[0];
            for (int i = 0; i < M2; i++) r += (s_o2[i] * (int)w3[i]) >> F;
            out[g] = (float)sx16(r) / (float)(1 << F);
        }
    }
}

// ---------------------------------------------------------------- GCN (Q6.10)
constexpr int CD = 100, CL = 5;
// dinv[u] = recip(sqrt(FM(outdeg + 1))) for nodes with out-edges, 0 otherwise (GCN/src/load_inputs.cc:103,122)
__global__ __launch_bounds__(256) void gcnq_dinv_kernel(const int* __restrict__ out_deg, int16_t* __restrict__ dinv, int n_tot) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n_tot) return;
    const int d = out_deg[v];
    dinv[v] = (int16_t)(d > 0 ? qrecip(qsqrt(sx16((d + 1) << 10), 10), 10) : 0);
}
// BN-side activation of GCN/src/node_embedding.cc:123-138 (RELU) and finalize.cc:79-113 (no ReLU): per (v, i)
template <bool RELU>
__global__ __launch_bounds__(256) void gcnq_act_kernel(const int16_t* __restrict__ x, const int16_t* __restrict__ m, const int* __restrict__ out_deg,
                                                        const int16_t* __restrict__ root, const int16_t* __restrict__ bnw,
                                                        const int16_t* __restrict__ bnb, const int16_t* __restrict__ bnm,
                                                        const int16_t* __restrict__ bnsq, int16_t* __restrict__ act, int n_tot) {
    const long long total = (long long)n_tot * CD;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx / CD), i = (int)(idx - (long long)v * CD);
        const int t = relu16(sx16((int)x[idx] + (int)root[i]));
        const int a = sx16((int)m[idx] + div_int(t, out_deg[v] + 1));
        const long long num = (long long)a - (long long)bnm[i];
        const int sv = bnsq[i];
        const long long q16f = sv == 0 ? 0 : (num * 65536ll) / (long long)sv;
        const long long val = q16f * (long long)bnw[i] + (long long)bnb[i] * 65536ll;
        int r = sx16((int)(val >> 16));
        if (RELU) r = relu16(r);
        act[idx] = (int16_t)r;
    }
}
// m[v][d] = sum over in-edges of floor(norm relu(wrap(ee + x[u][d]))), norm = stored dinv[u] dinv[v]  (message_passing.cc:158-167)
__global__ __launch_bounds__(256) void gcnq_mp_kernel(const int16_t* __restrict__ x, const int16_t* __restrict__ dinv, const int* __restrict__ row_ptr,
                                                       const int* __restrict__ src, const uint8_t* __restrict__ ecode,
                                                       const int16_t* __restrict__ ecomb, int16_t* __restrict__ m, int n_tot) {
    const long long total = (long long)n_tot * CD;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx / CD), d = (int)(idx - (long long)v * CD);
        const int dv = dinv[v];
        int acc = 0;
        for (int e = row_ptr[v]; e < row_ptr[v + 1]; e++) {
            const int u = src[e];
            const int norm = sx16(((int)dinv[u] * dv) >> 10);
            const int tot = sx16((int)ecomb[(size_t)ecode[e] * CD + d] + (int)x[(size_t)u * CD + d]);
            acc += (norm * relu16(tot)) >> 10;
        }
        m[idx] = (int16_t)acc;
    }
}

// ---------------------------------------------------------------- GAT (Q6.10): 16 dims x 4 heads, layout [v][d][head]
constexpr int AD = 16, AH = 4, AF = 64, AL = 5;
#define GQW5(w, l, ho, dout, hi, din) (w)[(((((size_t)(l) * 4 + (ho)) * 16 + (dout)) * 4 + (hi)) * 16) + (din)]
// load_input_node_embeddings (GAT/src/load_inputs.cc:168-226): one thread per node
__global__ __launch_bounds__(128) void gatq_encoder_kernel(const int* __restrict__ nf, const int* __restrict__ feat_row, const int16_t* __restrict__ lin,
                                                            const int16_t* __restrict__ wsrc, const int16_t* __restrict__ wtgt,
                                                            int16_t* __restrict__ proj, int16_t* __restrict__ skipin, int16_t* __restrict__ ssrc,
                                                            int16_t* __restrict__ stgt, int n_tot) {
    const int v = blockIdx.x * 128 + threadIdx.x;
    if (v >= n_tot) return;
    const int fr = feat_row ? feat_row[v] : v;  // reference quirk: every graph reads the first rows of the batch (GAT_compute.cc:72)
    int pr[AD][AH];
#pragma unroll
    for (int d = 0; d < AD; d++)
#pragma unroll
        for (int h = 0; h < AH; h++) pr[d][h] = 0;
    for (int k = 0; k < AD; k++)
        for (int h = 0; h < AH; h++) skipin[((size_t)v * AD + k) * AH + h] = 0;
    for (int k = 0; k < ND_FEATURE; k++) {
        const int f = sx16(nf[(size_t)fr * ND_FEATURE + k] << 10);  // R8
        skipin[((size_t)v * AD + k) * AH + 0] = (int16_t)f;
#pragma unroll
        for (int d = 0; d < AD; d++)
#pragma unroll
            for (int ho = 0; ho < AH; ho++) pr[d][ho] += sx16((f * (int)GQW5(lin, 0, ho, d, 0, k)) >> 10);
    }
    int as[AH] = {0, 0, 0, 0}, at[AH] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < AD; d++)
#pragma unroll
        for (int h = 0; h < AH; h++) {
            const int r = sx16(pr[d][h]);
            proj[((size_t)v * AD + d) * AH + h] = (int16_t)r;
            as[h] += sx16((r * (int)wsrc[(0 * AH + h) * AD + d]) >> 10);
            at[h] += sx16((r * (int)wtgt[(0 * AH + h) * AD + d]) >> 10);
        }
    for (int h = 0; h < AH; h++) { ssrc[(size_t)v * AH + h] = (int16_t)as[h]; stgt[(size_t)v * AH + h] = (int16_t)at[h]; }
}
// message passing + the adapter's divide (message_passing.cc:83-151, conv_layer.cc:158-177): one thread per (v, d, head)
__global__ __launch_bounds__(256) void gatq_mp_kernel(const int16_t* __restrict__ proj, const int16_t* __restrict__ ssrc, const int16_t* __restrict__ stgt,
                                                       const int* __restrict__ row_ptr, const int* __restrict__ src, const int16_t* __restrict__ exptab,
                                                       int16_t* __restrict__ msg, int n_tot, int slope) {
    const long long total = (long long)n_tot * AF;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx >> 6), dh = (int)(idx & 63), h = dh & 3;
        const int sv = ssrc[(size_t)v * AH + h];
        int num = 0, den = 0;
        const int e0 = row_ptr[v], e1 = row_ptr[v + 1];
        for (int e = e0 - 1; e < e1; e++) {  // e0 - 1 stands for the self edge
            const int u = e < e0 ? v : src[e];
            int s = sx16(sv + (int)stgt[(size_t)u * AH + h]);
            if (s < 0) s = sx16((s * slope) >> 10);
            const int sc = exptab[(unsigned)s & 0xFFFFu];
            den += sc;
            num += sx16((sc * (int)proj[(size_t)u * AF + dh]) >> 10);
        }
        msg[idx] = (int16_t)qdiv((long long)sx16(num), 10, sx16(den), 10);
    }
}
// node transformation (node_embedding.cc:98-271), one thread per node; LAST: finalize.cc:89-112 (embedding = sum / NUM_HEADS)
template <bool LAST>
__global__ __launch_bounds__(128) void gatq_nt_kernel(const int16_t* __restrict__ msg, const int16_t* __restrict__ skipin, const int16_t* __restrict__ skipw,
                                                       const int16_t* __restrict__ linw, const int16_t* __restrict__ wsrc, const int16_t* __restrict__ wtgt,
                                                       const int16_t* __restrict__ exptab, int l, int16_t* __restrict__ skipout,
                                                       int16_t* __restrict__ projout, int16_t* __restrict__ ssrc, int16_t* __restrict__ stgt,
                                                       int16_t* __restrict__ emb, int n_tot) {
    const int v = blockIdx.x * 128 + threadIdx.x;
    if (v >= n_tot) return;
    int sk[AF];
#pragma unroll
    for (int i = 0; i < AF; i++) sk[i] = skipin[(size_t)v * AF + i];
    int acc[AD][AH];
#pragma unroll
    for (int d = 0; d < AD; d++)
#pragma unroll
        for (int h = 0; h < AH; h++) acc[d][h] = 0;
#pragma unroll 1
    for (int dout = 0; dout < AD; dout++) {
        int o[AH];
#pragma unroll
        for (int ho = 0; ho < AH; ho++) o[ho] = msg[((size_t)v * AD + dout) * AH + ho];
        if (LAST) {
            int f = 0;
            for (int ho = 0; ho < AH; ho++) f += o[ho];
#pragma unroll
            for (int din = 0; din < AD; din++)
#pragma unroll
                for (int ho = 0; ho < AH; ho++)
#pragma unroll
                    for (int hi = 0; hi < AH; hi++) f += (sk[din * AH + hi] * (int)GQW5(skipw, l, ho, dout, hi, din)) >> 10;
            emb[(size_t)v * AD + dout] = (int16_t)div_int(sx16(f), AH);
            continue;
        }
#pragma unroll
        for (int din = 0; din < AD; din++)
#pragma unroll
            for (int ho = 0; ho < AH; ho++)
#pragma unroll
                for (int hi = 0; hi < AH; hi++) o[ho] += (sk[din * AH + hi] * (int)GQW5(skipw, l, ho, dout, hi, din)) >> 10;
#pragma unroll
        for (int ho = 0; ho < AH; ho++) {
            o[ho] = sx16(o[ho]);
            if (o[ho] <= 0) o[ho] = sx16((int)exptab[(unsigned)o[ho] & 0xFFFFu] - 1024);  // ELU
            skipout[((size_t)v * AD + dout) * AH + ho] = (int16_t)o[ho];
        }
#pragma unroll
        for (int pd = 0; pd < AD; pd++)
#pragma unroll
            for (int ho = 0; ho < AH; ho++)
#pragma unroll
                for (int hi = 0; hi < AH; hi++) acc[pd][ho] += sx16((o[hi] * (int)GQW5(linw, l + 1, ho, pd, hi, dout)) >> 10);
    }
    if (LAST) return;
    int as[AH] = {0, 0, 0, 0}, at[AH] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < AD; d++)
#pragma unroll
        for (int h = 0; h < AH; h++) {
            const int r = sx16(acc[d][h]);
            projout[((size_t)v * AD + d) * AH + h] = (int16_t)r;
            as[h] += sx16((r * (int)wsrc[((l + 1) * AH + h) * AD + d]) >> 10);
            at[h] += sx16((r * (int)wtgt[((l + 1) * AH + h) * AD + d]) >> 10);
        }
    for (int h = 0; h < AH; h++) { ssrc[(size_t)v * AH + h] = (int16_t)as[h]; stgt[(size_t)v * AH + h] = (int16_t)at[h]; }
}

// ---------------------------------------------------------------- PNA (Q6.10)
constexpr int PD = 80, PL = 4;
// message passing + the per-(v, i) statistics of node_embedding.cc:123-145: stats[v][i] = {mean, sd, min, max}
__global__ __launch_bounds__(256) void pnaq_mp_kernel(const int16_t* __restrict__ h, const int* __restrict__ row_ptr, const int* __restrict__ src,
                                                       int16_t* __restrict__ stats, int n_tot) {
    const long long total = (long long)n_tot * PD;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx / PD), d = (int)(idx - (long long)v * PD);
        int sum = 0, sq = 0, mn = 0x7FFF, mx = -0x8000;
        const int e0 = row_ptr[v], e1 = row_ptr[v + 1];
        for (int e = e0; e < e1; e++) {
            const int x = h[(size_t)src[e] * PD + d];
            sum += x;
            sq += sx16((x * x) >> 10);
            mn = x < mn ? x : mn;
            mx = x > mx ? x : mx;
        }
        const int dg = e1 - e0 == 0 ? 1 : e1 - e0;
        const int mean = div_int(sx16(sum), dg);
        const int var = sx16(div_int(sx16(sq), dg) - sx16((mean * mean) >> 10));
        const int sd = qsqrt(relu16(var), 10);
        int16_t* o = stats + idx * 4;
        o[0] = (int16_t)mean; o[1] = (int16_t)sd; o[2] = (int16_t)mn; o[3] = (int16_t)mx;
    }
}
// h'[v][o] = h[v][o] + relu(bias + sum_i addend(i, o)), node_embedding.cc:148-213; weights [out][scaler][aggr][in]
__global__ __launch_bounds__(256) void pnaq_nt_kernel(const int16_t* __restrict__ stats, const int16_t* __restrict__ h, const int* __restrict__ out_deg,
                                                       const int16_t* __restrict__ logtab, const int16_t* __restrict__ w, const int16_t* __restrict__ bias,
                                                       int avg, int16_t* __restrict__ hout, int n_tot) {
    const long long total = (long long)n_tot * PD;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx / PD), o = (int)(idx - (long long)v * PD);
        const int od = (out_deg[v] + 1) & 63;            // FM_TYPE(degree + 1): (value << 10) mod 2^16 repeats with period 64 (R8) ...
        const int logd = (od >= 1 && od < 32) ? (int)logtab[od] : 0;  // ... and is <= 0 for 32..64: log -> 0 (R5)
        const int t = qdiv(logd, 10, avg, 10);
        int scale = qdiv(avg, 10, logd, 10);
        if (scale == 0) scale = 1024;
        int acc = bias[o];
        const int16_t* wo = w + (size_t)o * 3 * 4 * PD;
        for (int i = 0; i < PD; i++) {
            const int16_t* st = stats + ((size_t)v * PD + i) * 4;
            const int mean = st[0], sd = st[1], mn = st[2], mx = st[3];
            int g[3];
#pragma unroll
            for (int s = 0; s < 3; s++) {
                const int16_t* ws = wo + (size_t)s * 4 * PD + i;  // aggregator a at ws[a * PD]: MEAN 0, MIN 1, MAX 2, STD 3
                const int a = sx16(sx16((mean * (int)ws[0 * PD]) >> 10) + sx16((sd * (int)ws[3 * PD]) >> 10));
                const int b = sx16(sx16((mn * (int)ws[1 * PD]) >> 10) + sx16((mx * (int)ws[2 * PD]) >> 10));
                g[s] = sx16(a + b);
            }
            acc += sx16(g[0] + sx16(sx16((g[1] * t) >> 10) + sx16((g[2] * scale) >> 10)));
        }
        hout[idx] = (int16_t)sx16((int)h[idx] + relu16(sx16(acc)));
    }
}

// ---------------------------------------------------------------- DGN (Q3.13)
constexpr int GD = 100, GL = 4;
// per node: sum |eig[u] - eig[v]| and sum (eig[u] - eig[v]) over in-edges (DGN/src/load_inputs.cc:105-110)
__global__ __launch_bounds__(256) void dgnq_prep_kernel(const int16_t* __restrict__ eig, const int* __restrict__ row_ptr, const int* __restrict__ src,
                                                         int16_t* __restrict__ abssum, int16_t* __restrict__ wsum, int n_tot) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n_tot) return;
    int a = 0, w = 0;
    const int ev = eig[v];
    for (int e = row_ptr[v]; e < row_ptr[v + 1]; e++) {
        const int diff = sx16((int)eig[src[e]] - ev);
        a += abs16(diff);
        w += diff;
    }
    abssum[v] = (int16_t)a;
    wsum[v] = (int16_t)w;
}
__global__ __launch_bounds__(256) void dgnq_eig_kernel(const float* __restrict__ node_eigen, int16_t* __restrict__ eig, int n_tot) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n_tot) return;
    const double f = floor((double)node_eigen[(size_t)v * 4 + 1] * 8192.0);  // WT_TYPE from float, column 1
    eig[v] = (int16_t)(unsigned short)(unsigned long long)(long long)f;
}
// message passing + the two activations of node_embedding.cc:143-144: act[v][0][i] = a1, act[v][1][i] = a2
__global__ __launch_bounds__(256) void dgnq_mp_kernel(const int16_t* __restrict__ h, const int16_t* __restrict__ eig, const int* __restrict__ row_ptr,
                                                       const int* __restrict__ src, const int* __restrict__ out_deg, const int16_t* __restrict__ abssum,
                                                       const int16_t* __restrict__ wsum, int16_t* __restrict__ act, int n_tot) {
    const long long total = (long long)n_tot * GD;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx / GD), d = (int)(idx - (long long)v * GD);
        const int ev = eig[v];
        int m1 = 0, m2 = 0;
        for (int e = row_ptr[v]; e < row_ptr[v + 1]; e++) {
            const int u = src[e];
            const int ew = sx16((int)eig[u] - ev);
            const int x = h[(size_t)u * GD + d];
            m1 += x;
            m2 += (x * ew) >> 13;
        }
        m1 = sx16(m1); m2 = sx16(m2);
        const int as = abssum[v] == 0 ? 1 : (int)abssum[v];
        const int a1 = div_int(m1, out_deg[v]);
        const long long num = (long long)m2 * 8192ll - (long long)wsum[v] * (long long)h[idx];
        const int a2 = abs16(qdiv(num, 26, as, 13));
        act[((size_t)v * 2 + 0) * GD + d] = (int16_t)a1;
        act[((size_t)v * 2 + 1) * GD + d] = (int16_t)a2;
    }
}
// h'[v][o] = h[v][o] + relu(bias + sum_i floor(a1 W[o][0][i] + a2 W[o][1][i]))  (one store per pair of products)
__global__ __launch_bounds__(256) void dgnq_nt_kernel(const int16_t* __restrict__ act, const int16_t* __restrict__ h, const int16_t* __restrict__ w,
                                                       const int16_t* __restrict__ bias, int16_t* __restrict__ hout, int n_tot) {
    const long long total = (long long)n_tot * GD;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int v = (int)(idx / GD), o = (int)(idx - (long long)v * GD);
        const int16_t* a1 = act + ((size_t)v * 2 + 0) * GD;
        const int16_t* a2 = act + ((size_t)v * 2 + 1) * GD;
        const int16_t* w0 = w + ((size_t)o * 2 + 0) * GD;
        const int16_t* w1 = w + ((size_t)o * 2 + 1) * GD;
        int acc = bias[o];
        for (int i = 0; i < GD; i++) acc += ((int)a1[i] * (int)w0[i] + (int)a2[i] * (int)w1[i]) >> 13;
        hout[idx] = (int16_t)sx16((int)h[idx] + relu16(sx16(acc)));
    }
}

int16_t q_from_float_host(float x, int F) {
    const double f = std::floor((double)x * (double)(1 << F));
    return (int16_t)(uint16_t)(uint64_t)(long long)f;
}

}  // namespace

void QPack::release() {
    for (auto& p : dev)
        if (p) { (void)hipFree(p); p = nullptr; }
    dev.clear();
    if (exptab) { (void)hipFree(exptab); exptab = nullptr; }
    if (logtab) { (void)hipFree(logtab); logtab = nullptr; }
    if (extra) { (void)hipFree(extra); extra = nullptr; }
    work.release();
}

int QPack::upload_all(int ntens, const float* const* tens, const size_t* elems, int frac_bits) {
    F = frac_bits;
    if ((int)dev.size() != ntens) { release(); dev.assign((size_t)ntens, nullptr); }
    for (int i = 0; i < ntens; i++) {
        std::vector<int16_t> q(elems[i]);
        for (size_t k = 0; k < elems[i]; k++) q[k] = q_from_float_host(tens[i][k], F);
        if (int rc = upload(&dev[(size_t)i], q)) return rc;
    }
    if (!exptab) {  // R5: one entry per Q6.10 pattern, the real function floored to the grid and wrapped (q_oracle.c: orc_q_exp_table)
        std::vector<int16_t> t(65536);
        for (int p = -32768; p < 32768; p++) {
            const double v = std::floor(std::exp((double)p / 1024.0) * 1024.0);
            t[(uint16_t)p] = v >= 9.0e18 ? 0 : (int16_t)(uint16_t)(uint64_t)(long long)v;
        }
        if (int rc = upload(&exptab, t)) return rc;
        std::vector<int16_t> lt(32, 0);  // log(FM_TYPE(k)), k = 1..31 (R5; q_oracle.c: orc_q_log)
        for (int k = 1; k < 32; k++) lt[(size_t)k] = (int16_t)(uint16_t)(uint64_t)(long long)std::floor(std::log((double)k) * 1024.0);
        if (int rc = upload(&logtab, lt)) return rc;
    }
    return 0;
}

static inline int qgrid(long long items) { return grid_for(items, 256, 256 * 16); }

// ---------------------------------------------------------------- forward passes
int gcnq_forward(QPack& q, DeviceBatch& db, Profiler& prof, hipStream_t s) {
    const int n = db.b.n_tot;
    if (n <= 0) return 0;
    // tensors: 0 nemb 1 eemb 2 cw 3 cb 4 root 5 bnw 6 bnb 7 bnm 8 bnv 9 pw 10 pb
    if (!q.extra) {  // per-layer derived tables: ecomb [5][60][100] (wrapped sum of the three rows) and bn_sqrt_var [5][100]
        std::vector<int16_t> eemb((size_t)CL * 13 * CD), bnv((size_t)CL * CD), ex((size_t)CL * EDGE_COMBOS * CD + (size_t)CL * CD);
        FG_HIP_TRY(hipMemcpy(eemb.data(), q.dev[1], eemb.size() * 2, hipMemcpyDeviceToHost));
        FG_HIP_TRY(hipMemcpy(bnv.data(), q.dev[8], bnv.size() * 2, hipMemcpyDeviceToHost));
        static const int ed_off[3] = {0, 5, 11};
        for (int l = 0; l < CL; l++) {
            for (int a0 = 0; a0 < 5; a0++)
                for (int a1 = 0; a1 < 6; a1++)
                    for (int a2 = 0; a2 < 2; a2++)
                        for (int d = 0; d < CD; d++) {
                            const int16_t* E = &eemb[(size_t)l * 13 * CD];
                            const int sum = E[(ed_off[0] + a0) * CD + d] + E[(ed_off[1] + a1) * CD + d] + E[(ed_off[2] + a2) * CD + d];
                            ex[((size_t)l * EDGE_COMBOS + (a0 * 6 + a1) * 2 + a2) * CD + d] = (int16_t)(uint16_t)(unsigned)sum;
                        }
            for (int d = 0; d < CD; d++) {  // hls::sqrt(bn_var + epsilon), GCN/src/load_inputs.cc:32 (R3)
                const int p = (int)bnv[(size_t)l * CD + d] + 1;
                unsigned long long x = p <= 0 ? 0ull : ((unsigned long long)p << 10), r = (unsigned long long)std::sqrt((double)x);
                while (r * r > x) r--;
                while ((r + 1) * (r + 1) <= x) r++;
                ex[(size_t)CL * EDGE_COMBOS * CD + (size_t)l * CD + d] = (int16_t)(uint16_t)(p <= 0 ? 0u : (unsigned)r);
            }
        }
        if (int rc = upload(&q.extra, ex)) return rc;
    }
    const int16_t* ecomb = q.extra;
    const int16_t* bnsq = q.extra + (size_t)CL * EDGE_COMBOS * CD;
    if (int rc = q.work.reserve(((size_t)n * CD * 4 + n + 64) / 2 + 64)) return rc;  // int16 x, m, act, xn + dinv, in an int buffer
    int16_t* x = reinterpret_cast<int16_t*>(q.work.p);
    int16_t *m = x + (size_t)n * CD, *act = m + (size_t)n * CD, *xn = act + (size_t)n * CD, *dinv = xn + (size_t)n * CD;
    ProfScope p(prof, "gcnq_forward", s);
    const long long items = (long long)n * CD;
    gcnq_dinv_kernel<<<(n + 255) / 256, 256, 0, s>>>(db.csr.out_deg, dinv, n);
    q_encoder_kernel<CD, false><<<qgrid(items), 256, 0, s>>>(db.b.node_feature, q.dev[0], x, n, db.csr.err);
    for (int l = 0; l < CL; l++) {
        const int16_t* in = x;
        if (l > 0) {
            gcnq_act_kernel<true><<<qgrid(items), 256, 0, s>>>(x, m, db.csr.out_deg, q.dev[4] + (size_t)(l - 1) * CD, q.dev[5] + (size_t)(l - 1) * CD,
                                                               q.dev[6] + (size_t)(l - 1) * CD, q.dev[7] + (size_t)(l - 1) * CD, bnsq + (size_t)(l - 1) * CD, act, n);
            in = act;
        }
        q_dense_kernel<10><<<qgrid(items), 256, 0, s>>>(in, CD, q.dev[2] + (size_t)l * CD * CD, q.dev[3] + (size_t)l * CD, xn, CD, n, 0, nullptr);
        std::swap(x, xn);
        gcnq_mp_kernel<<<qgrid(items), 256, 0, s>>>(x, dinv, db.csr.row_ptr, db.csr.src, db.csr.ecode, ecomb + (size_t)l * EDGE_COMBOS * CD, m, n);
    }
    gcnq_act_kernel<false><<<qgrid(items), 256, 0, s>>>(x, m, db.csr.out_deg, q.dev[4] + (size_t)(CL - 1) * CD, q.dev[5] + (size_t)(CL - 1) * CD,
                                                        q.dev[6] + (size_t)(CL - 1) * CD, q.dev[7] + (size_t)(CL - 1) * CD, bnsq + (size_t)(CL - 1) * CD, act, n);
    q_pool_head_kernel<10, CD, 0, 0><<<db.b.num_graphs, 128, 0, s>>>(act, db.b.node_off, q.dev[9], q.dev[10], nullptr, nullptr, nullptr, nullptr, db.out,
                                                                     db.b.num_graphs);
    db.h_valid = false;
    return 0;
}

int gatq_forward(QPack& q, DeviceBatch& db, const int* feat_row, Profiler& prof, hipStream_t s) {
    const int n = db.b.n_tot;
    if (n <= 0) return 0;
    // tensors: 0 tgt 1 src 2 lin 3 skip 4 pw 5 pb
    if (int rc = q.work.reserve(((size_t)n * (AF * 5 + AH * 4 + AD) + 64) / 2 + 64)) return rc;
    int16_t* proj = reinterpret_cast<int16_t*>(q.work.p);
    int16_t *proj2 = proj + (size_t)n * AF, *skip = proj2 + (size_t)n * AF, *skip2 = skip + (size_t)n * AF, *msg = skip2 + (size_t)n * AF;
    int16_t *ssrc = msg + (size_t)n * AF, *stgt = ssrc + (size_t)n * AH, *ssrc2 = stgt + (size_t)n * AH, *stgt2 = ssrc2 + (size_t)n * AH;
    int16_t* emb = stgt2 + (size_t)n * AH;
    ProfScope p(prof, "gatq_forward", s);
    const int slope = q_from_float_host(0.2f, 10);  // FM_TYPE(0.2), GAT/src/message_passing.cc:127
    const int nb = (n + 127) / 128;
    gatq_encoder_kernel<<<nb, 128, 0, s>>>(db.b.node_feature, feat_row, q.dev[2], q.dev[1], q.dev[0], proj, skip, ssrc, stgt, n);
    for (int l = 0; l < AL; l++) {
        gatq_mp_kernel<<<qgrid((long long)n * AF), 256, 0, s>>>(proj, ssrc, stgt, db.csr.row_ptr, db.csr.src, q.exptab, msg, n, slope);
        if (l < AL - 1) {
            gatq_nt_kernel<false><<<nb, 128, 0, s>>>(msg, skip, q.dev[3], q.dev[2], q.dev[1], q.dev[0], q.exptab, l, skip2, proj2, ssrc2, stgt2, nullptr, n);
            std::swap(proj, proj2); std::swap(skip, skip2); std::swap(ssrc, ssrc2); std::swap(stgt, stgt2);
        } else {
            gatq_nt_kernel<true><<<nb, 128, 0, s>>>(msg, skip, q.dev[3], q.dev[2], q.dev[1], q.dev[0], q.exptab, l, nullptr, nullptr, nullptr, nullptr, emb, n);
        }
    }
    q_pool_head_kernel<10, AD, 0, 0><<<db.b.num_graphs, 128, 0, s>>>(emb, db.b.node_off, q.dev[4], q.dev[5], nullptr, nullptr, nullptr, nullptr, db.out,
                                                                     db.b.num_graphs);
    db.h_valid = false;
    return 0;
}

int pnaq_forward(QPack& q, DeviceBatch& db, Profiler& prof, hipStream_t s) {
    const int n = db.b.n_tot;
    if (n <= 0) return 0;
    // tensors: 0 nemb 1 cw 2 cb 3 w1 4 b1 5 w2 6 b2 7 w3 8 b3 9 avg_deg
    if (int rc = q.work.reserve(((size_t)n * PD * 6 + 64) / 2 + 64)) return rc;
    int16_t* h = reinterpret_cast<int16_t*>(q.work.p);
    int16_t *hn = h + (size_t)n * PD, *stats = hn + (size_t)n * PD;
    int16_t avg = 0;
    FG_HIP_TRY(hipMemcpy(&avg, q.dev[9], 2, hipMemcpyDeviceToHost));
    ProfScope p(prof, "pnaq_forward", s);
    const long long items = (long long)n * PD;
    q_encoder_kernel<PD, false><<<qgrid(items), 256, 0, s>>>(db.b.node_feature, q.dev[0], h, n, db.csr.err);
    for (int l = 0; l < PL; l++) {
        pnaq_mp_kernel<<<qgrid(items), 256, 0, s>>>(h, db.csr.row_ptr, db.csr.src, stats, n);
        pnaq_nt_kernel<<<qgrid(items), 256, 0, s>>>(stats, h, db.csr.out_deg, q.logtab, q.dev[1] + (size_t)l * PD * 3 * 4 * PD, q.dev[2] + (size_t)l * PD,
                                                    (int)avg, hn, n);
        std::swap(h, hn);
    }
    q_pool_head_kernel<10, PD, 40, 20><<<db.b.num_graphs, 128, 0, s>>>(h, db.b.node_off, q.dev[3], q.dev[4], q.dev[5], q.dev[6], q.dev[7], q.dev[8],
                                                                       db.out, db.b.num_graphs);
    db.h_valid = false;
    return 0;
}

int dgnq_forward(QPack& q, DeviceBatch& db, Profiler& prof, hipStream_t s) {
    const int n = db.b.n_tot;
    if (n <= 0) return 0;
    if (!db.node_eigen) return 1;
    // tensors: 0 emb [9][119][100] 1 lw 2 lb 3 w0 4 b0 5 w1 6 b1 7 w2 8 b2
    if (int rc = q.work.reserve(((size_t)n * (GD * 4 + 3) + 64) / 2 + 64)) return rc;
    int16_t* h = reinterpret_cast<int16_t*>(q.work.p);
    int16_t *hn = h + (size_t)n * GD, *act = hn + (size_t)n * GD, *eig = act + (size_t)n * GD * 2, *abssum = eig + n, *wsum = abssum + n;
    ProfScope p(prof, "dgnq_forward", s);
    const long long items = (long long)n * GD;
    dgnq_eig_kernel<<<(n + 255) / 256, 256, 0, s>>>(db.node_eigen, eig, n);
    dgnq_prep_kernel<<<(n + 255) / 256, 256, 0, s>>>(eig, db.csr.row_ptr, db.csr.src, abssum, wsum, n);
    q_encoder_kernel<GD, true><<<qgrid(items), 256, 0, s>>>(db.b.node_feature, q.dev[0], h, n, db.csr.err);
    for (int l = 0; l < GL; l++) {
        dgnq_mp_kernel<<<qgrid(items), 256, 0, s>>>(h, eig, db.csr.row_ptr, db.csr.src, db.csr.out_deg, abssum, wsum, act, n);
        dgnq_nt_kernel<<<qgrid(items), 256, 0, s>>>(act, h, q.dev[1] + (size_t)l * GD * 2 * GD, q.dev[2] + (size_t)l * GD, hn, n);
        std::swap(h, hn);
    }
    q_pool_head_kernel<13, GD, 50, 25><<<db.b.num_graphs, 128, 0, s>>>(h, db.b.node_off, q.dev[3], q.dev[4], q.dev[5], q.dev[6], q.dev[7], q.dev[8],
                                                                       db.out, db.b.num_graphs);
    db.h_valid = false;
    return 0;
}

}  // namespace fg
