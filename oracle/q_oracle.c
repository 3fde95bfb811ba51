/*
 * q_oracle.c -- CPU restatement of FlowGNN's GCN, GAT, PNA (ap_fixed<16,6>, Q6.10: GCN/src/dcl.h:58-59, GAT/src/dcl.h,
 * PNA/src/dcl.h:74-75) and DGN (ap_fixed<16,3>, Q3.13: DGN/src/dcl.h:54-55) in the reference's own number formats.
 * TEST INFRASTRUCTURE (see flowgnn_oracle.h).  GIN / GIN-VN: ginq_oracle.c.
 *
 * PARITY UNPINNED, doubly (as ginq_oracle.c): the Vitis headers that define ap_fixed, hls::vector and the hls math functions
 * are not in this image, so the rules below are a reading of their published semantics, not a run of them.  F = number of
 * fractional bits of the format (10 or 13); "pattern" = the 16-bit two's-complement integer, value = pattern / 2^F.
 *
 *  R0  (ginq_oracle.c) every expression is exact in a wider type; a value is quantised only where it is stored into an
 *      FM_TYPE / WT_TYPE object: floor to the 2^-F grid, keep the low 16 bits.  So  x + y -> (x + y) mod 2^16,
 *      stored a * w -> ((a * w) >> F) mod 2^16,  r += a * w -> (r + ((a * w) >> F)) mod 2^16,  relu = sign ? 0 : x;
 *      float weights are quantised the same way on load; sums are independent of their order.
 *  R1  a / b, both ap_fixed: the quotient type of the published operator/ keeps F_a + I_b fractional bits and the divide is an
 *      integer divide (truncation toward zero):  Q = trunc(a_exact * 2^(F_a + I_b) / b); storing floors Q to the 2^-F grid.
 *      A zero divisor gives 0 (the float path's convention too: DESIGN.md section 2).
 *  R2  ap_fixed / int: the int is ap_fixed<32,32>, the quotient keeps F + 32 fractional bits; stored, it is floor(a / n) on the
 *      grid (ginq_oracle.c: the truncation at 2^-(F+32) cannot move a quotient with n <= 2^15 across a grid point).
 *  R3  hls::sqrt(x) = floor(sqrt(x)) on x's grid = isqrt(pattern << F); 0 for x <= 0.
 *  R4  hls::recip(x) = floor(1 / x) on x's grid = floor(2^(2F) / pattern) for pattern > 0; 0 for pattern = 0.
 *  R5  hls::exp(x), hls::log(x): the real function floored to the grid and wrapped to 16 bits, evaluated in double precision
 *      (one table entry per 16-bit pattern; the engine builds the same tables with the same libm).  log(x <= 0) = 0.
 *  R6  hls::abs(x) = |pattern| wrapped (abs of the most negative pattern is itself).
 *  R7  hls::vector<T, N> arithmetic (GAT's FM_VEC) yields T per element: every product / quotient of a vector expression is
 *      stored (R0 / R1) before the next operation.
 *  R8  int -> ap_fixed: (value << F) mod 2^16 (GAT's raw atom features wrap at +-32: GAT/src/load_inputs.cc:190-191;
 *      FM_TYPE(degree + 1) in GCN/src/load_inputs.cc:122 and PNA/src/load_inputs.cc:110).
 *
 * Under these rules every statement is integer arithmetic on 16-bit patterns and every sum is taken mod 2^16, i.e. independent of
 * its order: the loops below walk the edge list directly (no per-PE tables), and a batched GPU kernel can match them bit for bit.
 * Each block cites the reference lines it follows (paths under /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "flowgnn_oracle.h"

typedef int16_t q16;

static inline q16 w16(int64_t x) { return (q16)(uint16_t)(uint64_t)x; }
static inline q16 qadd(q16 a, q16 b) { return w16((int32_t)a + (int32_t)b); }
static inline q16 qrelu(q16 a) { return a < 0 ? 0 : a; }
static inline q16 qabs(q16 a) { return w16(a < 0 ? -(int32_t)a : (int32_t)a); }                 /* R6 */
static inline int32_t mulf(q16 a, q16 w, int F) { return ((int32_t)a * (int32_t)w) >> F; }       /* floor(a w) on the grid */
static inline q16 qmul(q16 a, q16 w, int F) { return w16(mulf(a, w, F)); }                       /* stored product */
static inline int32_t floordiv(int32_t s, int32_t n) { int32_t q = s / n; if ((s % n != 0) && ((s < 0) != (n < 0))) q--; return q; }
static inline q16 qdiv_int(q16 a, int n) { return n == 0 ? 0 : w16(floordiv(a, n)); }           /* R2 */
/* R1: a has fa fractional bits (pattern a_exact), b is a 16-bit pattern with F fractional and I = 16 - F integer bits */
static inline q16 qdiv(int64_t a_exact, int fa, q16 b, int F)
{
    if (b == 0) return 0;
    const int I = 16 - F;
    int64_t q = (a_exact * ((int64_t)1 << (F + I))) / (int64_t)b;  /* trunc toward zero; units 2^-(fa + I) */
    /* store: floor to 2^-F  (q has fa + I fractional bits) */
    const int sh = fa + I - F;
    return w16(q >> sh);
}
static inline uint32_t isqrt64(uint64_t x)
{
    uint64_t r = (uint64_t)sqrt((double)x);
    while (r * r > x) r--;
    while ((r + 1) * (r + 1) <= x) r++;
    return (uint32_t)r;
}
static inline q16 qsqrt(int32_t pattern, int F) { return pattern <= 0 ? 0 : w16(isqrt64((uint64_t)pattern << F)); }   /* R3 */
static inline q16 qrecip(q16 p, int F) { return p <= 0 ? 0 : w16(((int64_t)1 << (2 * F)) / p); }                        /* R4 */
static inline q16 qfrom_int(int v, int F) { return w16((int64_t)v * ((int64_t)1 << F)); }                               /* R8 */
q16 orc_q_from_float(float x, int F)
{
    double f = floor((double)x * (double)(1 << F));
    return (q16)(uint16_t)(uint64_t)(long long)f;
}
/* R5: exp table over all 16-bit Q6.10 patterns */
static q16* g_exp10 = 0;
const int16_t* orc_q_exp_table(void)
{
    if (!g_exp10) {
        q16* t = (q16*)malloc(sizeof(q16) * 65536);
        for (int p = -32768; p < 32768; p++) {
            double v = floor(exp((double)p / 1024.0) * 1024.0);
            t[(uint16_t)p] = v >= 9.0e18 ? 0 : w16((int64_t)v);
        }
        g_exp10 = t;
    }
    return g_exp10;
}
static inline q16 qexp10(q16 x) { return orc_q_exp_table()[(uint16_t)x]; }
int16_t orc_q_log(int16_t pattern) { return pattern <= 0 ? 0 : w16((int64_t)floor(log((double)pattern / 1024.0) * 1024.0)); }

static const int nd_off[9] = {0, 119, 123, 135, 147, 157, 163, 169, 171};
static const int nd_card[9] = {119, 4, 12, 12, 10, 6, 6, 2, 2};
static const int ed_off[3] = {0, 5, 11};
static const int ed_card[3] = {5, 6, 2};

static q16* quantise(const float* src, size_t n, int F)
{
    q16* q = (q16*)malloc(sizeof(q16) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) q[i] = orc_q_from_float(src[i], F);
    return q;
}
static int check_graph(int n, int e, const int* nf, const int* el, const int* ea, int check_feat)
{
    for (int i = 0; i < e; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        if (u < 0 || u >= n || v < 0 || v >= n) return 2;
        if (ea)
            for (int k = 0; k < 3; k++)
                if (ea[i * 3 + k] < 0 || ea[i * 3 + k] >= ed_card[k]) return 3;
    }
    if (check_feat)
        for (int v = 0; v < n; v++)
            for (int k = 0; k < 9; k++)
                if (nf[v * 9 + k] < 0 || nf[v * 9 + k] >= nd_card[k]) return 4;
    return 0;
}
/* mean pooling: sums mod 2^16 (order free), then / n (R2) */
static void mean_pool(const q16* h, int n, int Dm, q16* hg)
{
    for (int d = 0; d < Dm; d++) {
        q16 s = 0;
        for (int v = 0; v < n; v++) s = qadd(s, h[v * Dm + d]);
        hg[d] = qdiv_int(s, n);
    }
}
/* linear / linear_output_stationary / linear_input_stationary (<M>/src/linear.cc): out = b + sum_i in_i w_oi, every product
   stored (R0), optional ReLU */
static void qlinear(const q16* in, int din, const q16* w, const q16* b, int dout, int relu, int F, q16* out)
{
    for (int o = 0; o < dout; o++) {
        q16 r = b[o];
        for (int i = 0; i < din; i++) r = w16((int32_t)r + mulf(in[i], w[o * din + i], F));
        out[o] = relu ? qrelu(r) : r;
    }
}

/* ===================================================================================================== GCN (Q6.10) */
typedef struct { const q16 *nemb, *eemb, *cw, *cb, *root, *bnw, *bnb, *bnm, *bnv, *pw, *pb; } gcnq_w;

static int gcnq_one_graph(int n, int e, const int* nf, const int* el, const int* ea, const gcnq_w* w, q16* out)
{
    enum { D = 100, L = 5, F = 10 };
    int rc = check_graph(n, e, nf, el, ea, 1);
    if (rc) return rc;
    size_t nn = (size_t)(n > 0 ? n : 1);
    int* deg = (int*)calloc(nn, sizeof(int));
    q16* dinv = (q16*)calloc(nn, sizeof(q16));
    q16* x = (q16*)malloc(sizeof(q16) * nn * D);
    q16* xn = (q16*)malloc(sizeof(q16) * nn * D);
    q16* m = (q16*)calloc(nn * D, sizeof(q16));
    q16 bnsq[5][100], acc[100];
    /* load_weights: bn_sqrt_var = hls::sqrt(bn_var + epsilon), GCN/src/load_inputs.cc:32 (R3; the sum is exact) */
    for (int l = 0; l < L; l++)
        for (int d = 0; d < D; d++) bnsq[l][d] = qsqrt((int32_t)w->bnv[l * D + d] + 1, F);
    /* load_graph: degree_inv_sqrt[u] = recip(sqrt(WT_TYPE(outdeg(u) + 1))) for nodes with out-edges, 0 otherwise
       (GCN/src/load_inputs.cc:103,122; R8, R3, R4) */
    for (int i = 0; i < e; i++) deg[el[2 * i]]++;
    for (int u = 0; u < n; u++) dinv[u] = deg[u] > 0 ? qrecip(qsqrt(qfrom_int(deg[u] + 1, F), F), F) : 0;
    /* atom encoder, GCN/src/load_inputs.cc:168-215 */
    for (int v = 0; v < n; v++)
        for (int d = 0; d < D; d++) {
            q16 s = 0;
            for (int k = 0; k < 9; k++) s = qadd(s, w->nemb[(nd_off[k] + nf[v * 9 + k]) * D + d]);
            x[v * D + d] = s;
        }
    for (int l = 0; l < L; l++) {
        /* NT(l), GCN/src/node_embedding.cc:93-148 */
        const q16* W = w->cw + (size_t)l * D * D;
        for (int v = 0; v < n; v++) {
            for (int i = 0; i < D; i++) {
                q16 act;
                if (l == 0) {
                    act = x[v * D + i];
                } else {
                    q16 t = qrelu(qadd(x[v * D + i], w->root[(l - 1) * D + i]));                 /* relu<FM_TYPE>(h + root) */
                    act = qadd(m[v * D + i], qdiv_int(t, deg[v] + 1));                           /* :135 (R2; t >= 0) */
                    /* :136  (act - mean) / sqrt_var * weight + bias: quotient with 16 fractional bits (R1), product and sum
                       exact, one store */
                    int64_t num = (int64_t)act - (int64_t)w->bnm[(l - 1) * D + i];
                    q16 sv = bnsq[l - 1][i];
                    int64_t q16f = sv == 0 ? 0 : (num * 65536) / (int64_t)sv;                    /* units 2^-16 */
                    int64_t val = q16f * (int64_t)w->bnw[(l - 1) * D + i] + (int64_t)w->bnb[(l - 1) * D + i] * 65536; /* 2^-26 */
                    act = qrelu(w16(val >> 16));
                }
                for (int o = 0; o < D; o++) acc[o] = w16(mulf(act, W[o * D + i], F) + (int32_t)(i == 0 ? w->cb[l * D + o] : acc[o]));
            }
            memcpy(&xn[v * D], acc, sizeof(q16) * D);
        }
        memcpy(x, xn, sizeof(q16) * nn * D);
        /* MP(l), GCN/src/message_passing.cc:124-170: m[v] += norm * relu(edge_embed + x[u]), norm = dinv[u] dinv[v] stored */
        memset(m, 0, sizeof(q16) * nn * D);
        const q16* ee_l = w->eemb + (size_t)l * 13 * D;
        for (int i = 0; i < e; i++) {
            int u = el[2 * i], v = el[2 * i + 1];
            q16 norm = qmul(dinv[u], dinv[v], F);                                                /* load_inputs.cc:163 */
            for (int d = 0; d < D; d++) {
                q16 ee = 0;
                for (int k = 0; k < 3; k++) ee = qadd(ee, ee_l[(ed_off[k] + ea[i * 3 + k]) * D + d]);
                q16 tot = qadd(ee, x[u * D + d]);
                m[v * D + d] = w16((int32_t)m[v * D + d] + mulf(norm, qrelu(tot), F));           /* :167 */
            }
        }
    }
    {   /* finalize, GCN/src/finalize.cc:79-113: BN_4 without ReLU, mean pool, linear head */
        q16 hg[100];
        for (int d = 0; d < D; d++) {
            q16 s = 0;
            for (int v = 0; v < n; v++) {
                q16 t = qrelu(qadd(x[v * D + d], w->root[(L - 1) * D + d]));
                q16 act = qadd(m[v * D + d], qdiv_int(t, deg[v] + 1));
                int64_t num = (int64_t)act - (int64_t)w->bnm[(L - 1) * D + d];
                q16 sv = bnsq[L - 1][d];
                int64_t q16f = sv == 0 ? 0 : (num * 65536) / (int64_t)sv;
                int64_t val = q16f * (int64_t)w->bnw[(L - 1) * D + d] + (int64_t)w->bnb[(L - 1) * D + d] * 65536;
                s = qadd(s, w16(val >> 16));
            }
            hg[d] = qdiv_int(s, n);
        }
        qlinear(hg, D, w->pw, w->pb, 1, 0, F, out);
    }
    free(deg); free(dinv); free(x); free(xn); free(m);
    return 0;
}

/* ===================================================================================================== GAT (Q6.10) */
typedef struct { const q16 *tgt, *src, *lin, *skip, *pw, *pb; } gatq_w;
#define GW5(w, l, ho, dout, hi, din) (w)[(((((size_t)(l) * 4 + (ho)) * 16 + (dout)) * 4 + (hi)) * 16) + (din)]

static int gatq_one_graph(int n, int e, const int* nf, const int* el, const gatq_w* w, q16* out)
{
    enum { D = 16, H = 4, L = 5, F = 10 };
    int rc = check_graph(n, e, nf, el, 0, 0);
    if (rc) return rc;
    size_t nn = (size_t)(n > 0 ? n : 1);
    q16* proj = (q16*)malloc(sizeof(q16) * nn * D * H);
    q16* proj2 = (q16*)malloc(sizeof(q16) * nn * D * H);
    q16* skipin = (q16*)calloc(nn * D * H, sizeof(q16));
    q16* skipin2 = (q16*)malloc(sizeof(q16) * nn * D * H);
    q16* ssrc = (q16*)malloc(sizeof(q16) * nn * H);
    q16* stgt = (q16*)malloc(sizeof(q16) * nn * H);
    q16* ssrc2 = (q16*)malloc(sizeof(q16) * nn * H);
    q16* stgt2 = (q16*)malloc(sizeof(q16) * nn * H);
    q16* num = (q16*)malloc(sizeof(q16) * nn * D * H);
    q16* den = (q16*)malloc(sizeof(q16) * nn * H);
    q16* emb = (q16*)malloc(sizeof(q16) * nn * D);
    const q16 slope = orc_q_from_float(0.2f, F);   /* FM_TYPE(0.2), GAT/src/message_passing.cc:127 */
    /* load_input_node_embeddings, GAT/src/load_inputs.cc:168-226: raw integer features as FM_TYPE (R8), scalar x vector (R7) */
    for (int v = 0; v < n; v++) {
        q16 pr[16][4];
        memset(pr, 0, sizeof(pr));
        for (int k = 0; k < 9; k++) {
            q16 f = qfrom_int(nf[v * 9 + k], F);
            skipin[(v * D + k) * H + 0] = f;                                                      /* :191 */
            for (int d = 0; d < D; d++)
                for (int ho = 0; ho < H; ho++) pr[d][ho] = qadd(pr[d][ho], qmul(f, GW5(w->lin, 0, ho, d, 0, k), F)); /* :198-200 */
        }
        q16 as[4] = {0, 0, 0, 0}, at[4] = {0, 0, 0, 0};
        for (int d = 0; d < D; d++)
            for (int hh = 0; hh < H; hh++) {
                q16 r = pr[d][hh];
                proj[(v * D + d) * H + hh] = r;
                as[hh] = qadd(as[hh], qmul(r, w->src[(0 * H + hh) * D + d], F));
                at[hh] = qadd(at[hh], qmul(r, w->tgt[(0 * H + hh) * D + d], F));
            }
        for (int hh = 0; hh < H; hh++) { ssrc[v * H + hh] = as[hh]; stgt[v * H + hh] = at[hh]; }
    }
    for (int l = 0; l < L; l++) {
        /* MP: per destination v, over the self edge and every in-edge (u -> v): GAT/src/message_passing.cc:83-151 */
        memset(num, 0, sizeof(q16) * nn * D * H);
        memset(den, 0, sizeof(q16) * nn * H);
        for (int i = -n; i < e; i++) {
            int u = i < 0 ? i + n : el[2 * i], v = i < 0 ? i + n : el[2 * i + 1];
            q16 sc[4];
            for (int hh = 0; hh < H; hh++) {
                q16 s = qadd(ssrc[v * H + hh], stgt[u * H + hh]);                                 /* :122 */
                if (s < 0) s = qmul(s, slope, F);                                                 /* :126-127 */
                sc[hh] = qexp10(s);                                                               /* :128 (R5) */
                den[v * H + hh] = qadd(den[v * H + hh], sc[hh]);
            }
            for (int d = 0; d < D; d++)
                for (int hh = 0; hh < H; hh++)
                    num[(v * D + d) * H + hh] = qadd(num[(v * D + d) * H + hh], qmul(sc[hh], proj[(u * D + d) * H + hh], F)); /* :133-141 (R7) */
        }
        for (int v = 0; v < n; v++) {
            q16 msg[16][4];
            for (int d = 0; d < D; d++)
                for (int hh = 0; hh < H; hh++) msg[d][hh] = qdiv(num[(v * D + d) * H + hh], F, den[v * H + hh], F); /* conv_layer.cc:177 (R1, R7) */
            if (l < L - 1) {
                /* NT, GAT/src/node_embedding.cc:98-271 */
                q16 acc[16][4];
                for (int dout = 0; dout < D; dout++) {
                    q16 o[4];
                    for (int ho = 0; ho < H; ho++) o[ho] = msg[dout][ho];
                    for (int din = 0; din < D; din++)
                        for (int ho = 0; ho < H; ho++)
                            for (int hi = 0; hi < H; hi++)
                                o[ho] = w16((int32_t)o[ho] + mulf(skipin[(v * D + din) * H + hi], GW5(w->skip, l, ho, dout, hi, din), F)); /* :157-169 */
                    for (int ho = 0; ho < H; ho++)
                        if (o[ho] <= 0) o[ho] = w16((int32_t)qexp10(o[ho]) - 1024);               /* ELU, :172-178 */
                    for (int ho = 0; ho < H; ho++) skipin2[(v * D + dout) * H + ho] = o[ho];
                    for (int pd = 0; pd < D; pd++)
                        for (int ho = 0; ho < H; ho++) {
                            q16 a = dout != 0 ? acc[pd][ho] : 0;
                            for (int hi = 0; hi < H; hi++) a = qadd(a, qmul(o[hi], GW5(w->lin, l + 1, ho, pd, hi, dout), F)); /* :182-195 (R7) */
                            acc[pd][ho] = a;
                        }
                }
                q16 as[4] = {0, 0, 0, 0}, at[4] = {0, 0, 0, 0};
                for (int d = 0; d < D; d++)
                    for (int hh = 0; hh < H; hh++) {
                        q16 r = acc[d][hh];
                        proj2[(v * D + d) * H + hh] = r;
                        as[hh] = qadd(as[hh], qmul(r, w->src[((l + 1) * H + hh) * D + d], F));    /* :235-268 */
                        at[hh] = qadd(at[hh], qmul(r, w->tgt[((l + 1) * H + hh) * D + d], F));
                    }
                for (int hh = 0; hh < H; hh++) { ssrc2[v * H + hh] = as[hh]; stgt2[v * H + hh] = at[hh]; }
            } else {
                /* last layer: (sum over heads of the message + skip projection) / NUM_HEADS, GAT/src/finalize.cc:89-112 */
                for (int dout = 0; dout < D; dout++) {
                    q16 f = 0;
                    for (int hh = 0; hh < H; hh++) f = qadd(f, msg[dout][hh]);
                    for (int din = 0; din < D; din++)
                        for (int ho = 0; ho < H; ho++)
                            for (int hi = 0; hi < H; hi++)
                                f = w16((int32_t)f + mulf(skipin[(v * D + din) * H + hi], GW5(w->skip, L - 1, ho, dout, hi, din), F));
                    emb[v * D + dout] = qdiv_int(f, H);
                }
            }
        }
        if (l < L - 1) {
            memcpy(proj, proj2, sizeof(q16) * nn * D * H);
            memcpy(skipin, skipin2, sizeof(q16) * nn * D * H);
            memcpy(ssrc, ssrc2, sizeof(q16) * nn * H);
            memcpy(stgt, stgt2, sizeof(q16) * nn * H);
        }
    }
    {
        q16 hg[16];
        mean_pool(emb, n, D, hg);
        qlinear(hg, D, w->pw, w->pb, 1, 0, F, out);
    }
    free(proj); free(proj2); free(skipin); free(skipin2); free(ssrc); free(stgt); free(ssrc2); free(stgt2); free(num); free(den); free(emb);
    return 0;
}

/* ===================================================================================================== PNA (Q6.10) */
typedef struct { const q16 *nemb, *cw, *cb, *w1, *b1, *w2, *b2, *w3, *b3; q16 avg; } pnaq_w;

static int pnaq_one_graph(int n, int e, const int* nf, const int* el, const pnaq_w* w, q16* out)
{
    enum { D = 80, L = 4, F = 10, NA = 4, NS = 3, M1 = 40, M2 = 20 };
    int rc = check_graph(n, e, nf, el, 0, 1);
    if (rc) return rc;
    size_t nn = (size_t)(n > 0 ? n : 1);
    int* indeg = (int*)calloc(nn, sizeof(int));
    int* outdeg = (int*)calloc(nn, sizeof(int));
    q16* h = (q16*)malloc(sizeof(q16) * nn * D);
    q16* hn = (q16*)malloc(sizeof(q16) * nn * D);
    q16* msg = (q16*)malloc(sizeof(q16) * nn * D * NA);  /* [v][d][mean-sum, std-sumsq, min, max] */
    q16 acc[80];
    for (int i = 0; i < e; i++) { outdeg[el[2 * i]]++; indeg[el[2 * i + 1]]++; }
    for (int v = 0; v < n; v++)
        for (int d = 0; d < D; d++) {
            q16 s = 0;
            for (int k = 0; k < 9; k++) s = qadd(s, w->nemb[(nd_off[k] + nf[v * 9 + k]) * D + d]);
            h[v * D + d] = s;
        }
    for (int l = 0; l <= L; l++) {
        if (l > 0) {
            /* NT(l-1), PNA/src/node_embedding.cc:106-214 */
            const q16* W = w->cw + (size_t)(l - 1) * D * NS * NA * D;  /* [out][scaler][aggr][in] */
            for (int v = 0; v < n; v++) {
                const int dg = indeg[v] == 0 ? 1 : indeg[v];                                     /* :123 */
                const q16 logd = orc_q_log(qfrom_int(outdeg[v] + 1, F));                          /* load_inputs.cc:110 (R8, R5) */
                const q16 t = qdiv(logd, F, w->avg, F);                                           /* :148 (R1) */
                q16 scale = qdiv(w->avg, F, logd, F);                                             /* :149 */
                if (scale == 0) scale = 1 << F;                                                   /* :150 */
                for (int i = 0; i < D; i++) {
                    const q16* mg = &msg[((size_t)v * D + i) * NA];
                    const q16 mean = qdiv_int(mg[0], dg);                                         /* :143 */
                    const q16 var = w16((int32_t)qdiv_int(mg[1], dg) - (int32_t)qmul(mean, mean, F));
                    const q16 sd = qsqrt(qrelu(var), F);                                          /* :144-145 (R3) */
                    const q16 mn = mg[2], mx = mg[3];
                    for (int o = 0; o < D; o++) {
                        const q16* wo = W + (size_t)o * NS * NA * D + i;
#define PW(s, a) wo[((s) * NA + (a)) * D]
#define PG(s) qadd(qadd(qmul(mean, PW(s, 0), F), qmul(sd, PW(s, 3), F)), qadd(qmul(mn, PW(s, 1), F), qmul(mx, PW(s, 2), F)))
                        /* aggregator order in the weights: AGGR_MEAN 0, AGGR_MIN 1, AGGR_MAX 2, AGGR_STD 3 (PNA/src/dcl.h:29-35) */
                        const q16 g0 = PG(0), g1 = PG(1), g2 = PG(2);
#undef PG
#undef PW
                        const q16 addend = qadd(g0, qadd(qmul(g1, t, F), qmul(g2, scale, F)));    /* :158-186 */
                        acc[o] = qadd(addend, i == 0 ? w->cb[(l - 1) * D + o] : acc[o]);
                    }
                }
                for (int d = 0; d < D; d++) hn[v * D + d] = qadd(h[v * D + d], qrelu(acc[d]));   /* :205-213 */
            }
            memcpy(h, hn, sizeof(q16) * nn * D);
        }
        if (l == L) break;
        /* MP, PNA/src/message_passing.cc:75-147 */
        for (size_t i = 0; i < nn * D; i++) { msg[i * NA + 0] = 0; msg[i * NA + 1] = 0; msg[i * NA + 2] = 0x7FFF; msg[i * NA + 3] = (q16)-0x8000; }
        for (int i = 0; i < e; i++) {
            int u = el[2 * i], v = el[2 * i + 1];
            for (int d = 0; d < D; d++) {
                q16 xv = h[u * D + d];
                q16* mg = &msg[((size_t)v * D + d) * NA];
                mg[0] = qadd(mg[0], xv);
                mg[1] = qadd(mg[1], qmul(xv, xv, F));
                if (xv < mg[2]) mg[2] = xv;
                if (xv > mg[3]) mg[3] = xv;
            }
        }
    }
    {
        q16 hg[80], o1[40], o2[20];
        mean_pool(h, n, D, hg);
        qlinear(hg, D, w->w1, w->b1, M1, 1, F, o1);
        qlinear(o1, M1, w->w2, w->b2, M2, 1, F, o2);
        qlinear(o2, M2, w->w3, w->b3, 1, 0, F, out);
    }
    free(indeg); free(outdeg); free(h); free(hn); free(msg);
    return 0;
}

/* ===================================================================================================== DGN (Q3.13) */
typedef struct { const q16 *emb, *lw, *lb, *w0, *b0, *w1, *b1, *w2, *b2; } dgnq_w;

static int dgnq_one_graph(int n, int e, const int* nf, const q16* eig /* column 1 */, const int* el, const dgnq_w* w, q16* out)
{
    enum { D = 100, L = 4, F = 13, TBL = 119, M1 = 50, M2 = 25 };
    int rc = check_graph(n, e, nf, el, 0, 1);
    if (rc) return rc;
    size_t nn = (size_t)(n > 0 ? n : 1);
    int* deg = (int*)calloc(nn, sizeof(int));
    q16* abssum = (q16*)calloc(nn, sizeof(q16));
    q16* wsum = (q16*)calloc(nn, sizeof(q16));
    q16* h = (q16*)malloc(sizeof(q16) * nn * D);
    q16* hn = (q16*)malloc(sizeof(q16) * nn * D);
    q16* msg = (q16*)malloc(sizeof(q16) * nn * 2 * D);
    q16 acc[100];
    /* load_graph, DGN/src/load_inputs.cc:92-111 */
    for (int i = 0; i < e; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        deg[u]++;
        q16 diff = w16((int32_t)eig[u] - (int32_t)eig[v]);
        abssum[v] = qadd(abssum[v], qabs(diff));
        wsum[v] = qadd(wsum[v], diff);
    }
    for (int v = 0; v < n; v++)
        for (int d = 0; d < D; d++) {
            q16 s = 0;
            for (int k = 0; k < 9; k++) s = qadd(s, w->emb[((size_t)k * TBL + nf[v * 9 + k]) * D + d]);
            h[v * D + d] = s;
        }
    for (int l = 0; l <= L; l++) {
        if (l > 0) {
            /* NT(l-1), DGN/src/node_embedding.cc:107-181; W viewed as [out][2][in] */
            const q16* W = w->lw + (size_t)(l - 1) * D * 2 * D;
            for (int v = 0; v < n; v++) {
                const q16 as = abssum[v] == 0 ? 1 : abssum[v];                                    /* :125-128 (epsilon = 2^-13) */
                for (int i = 0; i < D; i++) {
                    const q16 m1 = msg[(v * 2 + 0) * D + i], m2 = msg[(v * 2 + 1) * D + i], hv = h[v * D + i];
                    const q16 a1 = qdiv_int(m1, deg[v]);                                          /* :143 (R2; x / 0 = 0) */
                    const int64_t num = (int64_t)m2 * ((int64_t)1 << F) - (int64_t)wsum[v] * (int64_t)hv;  /* exact, 2F fractional bits */
                    const q16 a2 = qabs(qdiv(num, 2 * F, as, F));                                 /* :144 (R1, R6) */
                    for (int o = 0; o < D; o++) {
                        const int32_t both = ((int32_t)a1 * W[(o * 2 + 0) * D + i] + (int32_t)a2 * W[(o * 2 + 1) * D + i]) >> F; /* one store */
                        acc[o] = w16(both + (int32_t)(i == 0 ? w->lb[(l - 1) * D + o] : acc[o]));
                    }
                }
                for (int d = 0; d < D; d++) hn[v * D + d] = qadd(h[v * D + d], qrelu(acc[d]));   /* :176-181 */
            }
            memcpy(h, hn, sizeof(q16) * nn * D);
        }
        if (l == L) break;
        /* MP, DGN/src/message_passing.cc:120-152 */
        memset(msg, 0, sizeof(q16) * nn * 2 * D);
        for (int i = 0; i < e; i++) {
            int u = el[2 * i], v = el[2 * i + 1];
            q16 ew = w16((int32_t)eig[u] - (int32_t)eig[v]);
            for (int d = 0; d < D; d++) {
                msg[(v * 2 + 0) * D + d] = qadd(msg[(v * 2 + 0) * D + d], h[u * D + d]);
                msg[(v * 2 + 1) * D + d] = w16((int32_t)msg[(v * 2 + 1) * D + d] + mulf(h[u * D + d], ew, F));
            }
        }
    }
    {
        q16 hg[100], o1[50], o2[25];
        mean_pool(h, n, D, hg);
        qlinear(hg, D, w->w0, w->b0, M1, 1, F, o1);
        qlinear(o1, M1, w->w1, w->b1, M2, 1, F, o2);
        qlinear(o2, M2, w->w2, w->b2, 1, 0, F, out);
    }
    free(deg); free(abssum); free(wsum); free(h); free(hn); free(msg);
    return 0;
}

/* ===================================================================================================== batch drivers
 * Same arguments as the float entry points (float weights: quantised here as the reference host does); out_q receives the 16-bit
 * patterns, out (optional) pattern / 2^F.  model: 2 GCN, 3 GAT, 4 PNA, 5 DGN (the FLOWGNN_MODEL_* ids).  tens = the model's weight
 * tensors in entry-point order, elems = elements of ONE weight set per tensor. */
int orc_q_compute_graphs(int model, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges, const int* reload_weights,
                         float* out, int16_t* out_q, const int* node_feature_in, const float* node_eigen_in, const int* edge_list_in,
                         const int* edge_attr_in, int ntens, const float* const* tens, const long* elems, int gat_feature_offset_quirk,
                         int nthreads)
{
    const int F = model == 5 ? 13 : 10;
    long* noff = (long*)malloc(sizeof(long) * (size_t)(num_graphs + 1));
    long* eoff = (long*)malloc(sizeof(long) * (size_t)(num_graphs + 1));
    int* widx = (int*)malloc(sizeof(int) * (size_t)(num_graphs + 1));
    int wi = -1, rc = 0;
    noff[0] = eoff[0] = 0;
    for (int g = 0; g < num_graphs; g++) {
        if (reload_weights[g]) wi++;
        widx[g] = wi;
        noff[g + 1] = noff[g] + nums_of_nodes[g];
        eoff[g + 1] = eoff[g] + nums_of_edges[g];
    }
    if (num_graphs > 0 && widx[0] < 0) { free(noff); free(eoff); free(widx); return 1; }
    const size_t S = (size_t)(wi + 1 > 0 ? wi + 1 : 1);
    q16* qt[16];
    for (int i = 0; i < ntens; i++) qt[i] = quantise(tens[i], S * (size_t)elems[i], F);
    q16* eigq = 0;
    if (model == 5) {  /* node_eigen_t = array<WT_TYPE, 4>: column 1 is the one used (DGN/src/load_inputs.cc:105-106) */
        eigq = (q16*)malloc(sizeof(q16) * (size_t)(noff[num_graphs] > 0 ? noff[num_graphs] : 1));
        for (long v = 0; v < noff[num_graphs]; v++) eigq[v] = orc_q_from_float(node_eigen_in[v * 4 + 1], F);
    }
    (void)orc_q_exp_table();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int g = 0; g < num_graphs; g++) {
        const size_t s = (size_t)widx[g];
        const int n = nums_of_nodes[g], e = nums_of_edges[g];
        const int* nf = node_feature_in + noff[g] * 9;
        const int* el = edge_list_in + eoff[g] * 2;
        q16 o = 0;
        int r = 0;
#define T(i) (qt[i] + s * (size_t)elems[i])
        if (model == 2) {
            gcnq_w w = {T(0), T(1), T(2), T(3), T(4), T(5), T(6), T(7), T(8), T(9), T(10)};
            r = gcnq_one_graph(n, e, nf, el, edge_attr_in + eoff[g] * 3, &w, &o);
        } else if (model == 3) {
            gatq_w w = {T(0), T(1), T(2), T(3), T(4), T(5)};
            r = gatq_one_graph(n, e, gat_feature_offset_quirk ? node_feature_in : nf, el, &w, &o);
        } else if (model == 4) {
            pnaq_w w = {T(0), T(1), T(2), T(3), T(4), T(5), T(6), T(7), T(8), T(9)[0]};
            r = pnaq_one_graph(n, e, nf, el, &w, &o);
        } else if (model == 5) {
            dgnq_w w = {T(0), T(1), T(2), T(3), T(4), T(5), T(6), T(7), T(8)};
            r = dgnq_one_graph(n, e, nf, eigq + noff[g], el, &w, &o);
        } else {
            r = 8;
        }
#undef T
        if (out_q) out_q[g] = o;
        if (out) out[g] = (float)o / (float)(1 << F);
        if (r) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc = r;
        }
    }
    for (int i = 0; i < ntens; i++) free(qt[i]);
    free(eigq); free(noff); free(eoff); free(widx);
    return rc;
}
