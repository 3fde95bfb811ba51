/*
 * pna_oracle.c -- float restatement of FlowGNN PNA (TEST INFRASTRUCTURE, parity unpinned; see
 * flowgnn_oracle.h).  Each block cites the reference lines it follows (paths under /root/reference).
 */
#include "flowgnn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define D 80   /* EMB_DIM,    PNA/src/dcl.h:23 */
#define L 4    /* NUM_LAYERS, PNA/src/dcl.h:24 */
#define M1 40  /* GRAPH_MLP_1_OUT */
#define M2 20  /* GRAPH_MLP_2_OUT */
#define PE ORC_EDGE_PARALLEL
enum { A_MEAN = 0, A_MIN = 1, A_MAX = 2, A_STD = 3, NA = 4 };   /* PNA/src/dcl.h:29-35 */
enum { S_NONE = 0, S_T = 1, S_SCALE = 2, NS = 3 };             /* PNA/src/dcl.h:37-42 */

/* ap_fixed_max / ap_fixed_min of ap_fixed<16,6> (PNA/src/util.h:34-46): the min / max aggregators start
   from these and KEEP them for nodes without in-edges, where they enter the arithmetic. */
#define SENT_MAX 31.9990234375f
#define SENT_MIN (-32.0f)

static const int nd_off[ORC_ND_FEATURE] = {0, 119, 123, 135, 147, 157, 163, 169, 171}; /* PNA/src/load_inputs.cc:6 */
static const int nd_card[ORC_ND_FEATURE] = {119, 4, 12, 12, 10, 6, 6, 2, 2};

static inline float relu_f(float x) { return x < 0.0f ? 0.0f : x; }

typedef struct {
    const float *nemb, *cw, *cb, *w1, *b1, *w2, *b2, *w3, *b3;
    float avg_deg;
} pna_w;

static int pna_one_graph(int n, int e, const int* nf, const int* el, const pna_w* w, float* out, float* h_dump,
                         long n_tot, long node_off)
{
    size_t nn = (size_t)(n > 0 ? n : 1), ee = (size_t)(e > 0 ? e : 1);
    int* in_deg = (int*)calloc(nn, sizeof(int));
    int* out_deg = (int*)calloc(nn, sizeof(int));
    int* out_degs = (int*)calloc(nn * PE, sizeof(int));
    int* nto = (int*)calloc(nn * PE, sizeof(int));
    int* neighbor_tables = (int*)malloc(sizeof(int) * ee * PE);
    float* logd = (float*)malloc(sizeof(float) * nn);
    float* h = (float*)malloc(sizeof(float) * nn * D);
    float* hn = (float*)malloc(sizeof(float) * nn * D);
    float* msg = (float*)malloc(sizeof(float) * nn * D * NA);
    float acc[D];
    int epp[PE] = {0, 0, 0, 0};
    int rc = 0;

    for (int i = 0; i < e && !rc; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        if (u < 0 || u >= n || v < 0 || v >= n) rc = 2;
    }
    for (int v = 0; v < n && !rc; v++)
        for (int k = 0; k < ORC_ND_FEATURE; k++)
            if (nf[v * 9 + k] < 0 || nf[v * 9 + k] >= nd_card[k]) rc = 4;
    if (rc) goto done;

    /* load_graph, PNA/src/load_inputs.cc:48-131 */
    for (int i = 0; i < e; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        in_deg[v]++;
        out_deg[u]++;
        out_degs[(v % PE) * n + u]++;
    }
    for (int i = 0; i < n; i++) {
        logd[i] = logf((float)(out_deg[i] + 1)); /* :110, OUT-degree */
        for (int p = 0; p < PE; p++) {
            nto[p * n + i] = epp[p];
            epp[p] += out_degs[p * n + i];
        }
    }
    for (int i = 0; i < e; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        int p = v % PE;
        neighbor_tables[p * e + nto[p * n + u]++] = v / PE;
    }

    /* atom encoder, PNA/src/load_inputs.cc:133-179 */
    for (int v = 0; v < n; v++)
        for (int d = 0; d < D; d++) {
            float s = 0.0f;
            for (int k = 0; k < ORC_ND_FEATURE; k++) s += w->nemb[(nd_off[k] + nf[v * 9 + k]) * D + d];
            h[v * D + d] = s;
        }
    if (h_dump) memcpy(h_dump + (0 * n_tot + node_off) * D, h, sizeof(float) * (size_t)n * D);

    for (int l = 0; l <= L; l++) {
        if (l > 0) {
            /* NT(l-1), PNA/src/node_embedding.cc:106-214 */
            const float* W = w->cw + (size_t)(l - 1) * D * NS * NA * D; /* [out][scaler][aggr][in] */
            for (int v = 0; v < n; v++) {
                int indeg = in_deg[v] == 0 ? 1 : in_deg[v]; /* :123 */
                float t = logd[v] / w->avg_deg;             /* :148 */
                float scale = (logd[v] == 0.0f) ? 1.0f : w->avg_deg / logd[v]; /* :149-150: x/0 is 0 in ap_fixed, then 0 -> 1 */
                for (int i = 0; i < D; i++) {
                    const float* mg = &msg[((size_t)v * D + i) * NA];
                    float mean = mg[A_MEAN] / (float)indeg;
                    float sd = sqrtf(relu_f(mg[A_STD] / (float)indeg - mean * mean)); /* :144-145 */
                    float mn = mg[A_MIN], mx = mg[A_MAX];
                    for (int o = 0; o < D; o++) {
                        const float* wo = W + (size_t)o * NS * NA * D + i; /* + (s*NA + a)*D */
#define WT(s, a) wo[((s) * NA + (a)) * D]
                        float g0 = (mean * WT(S_NONE, A_MEAN) + sd * WT(S_NONE, A_STD)) + (mn * WT(S_NONE, A_MIN) + mx * WT(S_NONE, A_MAX));
                        float g1 = (mean * WT(S_T, A_MEAN) + sd * WT(S_T, A_STD)) + (mn * WT(S_T, A_MIN) + mx * WT(S_T, A_MAX));
                        float g2 = (mean * WT(S_SCALE, A_MEAN) + sd * WT(S_SCALE, A_STD)) + (mn * WT(S_SCALE, A_MIN) + mx * WT(S_SCALE, A_MAX));
#undef WT
                        float addend = g0 + (g1 * t + g2 * scale);           /* :158-186 */
                        acc[o] = addend + (i == 0 ? w->cb[(l - 1) * D + o] : acc[o]);
                    }
                }
                for (int d = 0; d < D; d++) hn[v * D + d] = h[v * D + d] + relu_f(acc[d]); /* :205-213 */
            }
            memcpy(h, hn, sizeof(float) * nn * D);
            if (h_dump) memcpy(h_dump + ((size_t)l * n_tot + node_off) * D, h, sizeof(float) * (size_t)n * D);
        }
        if (l == L) break;
        /* MP, PNA/src/message_passing.cc:75-147 (sentinels :140-147) */
        for (size_t i = 0; i < nn * D; i++) {
            msg[i * NA + A_MEAN] = 0.0f; msg[i * NA + A_STD] = 0.0f;
            msg[i * NA + A_MIN] = SENT_MAX; msg[i * NA + A_MAX] = SENT_MIN;
        }
        for (int p = 0; p < PE; p++) {
            int pos = 0;
            for (int u = 0; u < n; u++)
                for (int j = 0; j < out_degs[p * n + u]; j++, pos++) {
                    int v = neighbor_tables[p * e + pos] * PE + p;
                    for (int d = 0; d < D; d++) {
                        float x = h[u * D + d];
                        float* mg = &msg[((size_t)v * D + d) * NA];
                        mg[A_MEAN] += x;
                        float sq = x * x;
                        mg[A_STD] += sq;
                        if (x < mg[A_MIN]) mg[A_MIN] = x;
                        if (x > mg[A_MAX]) mg[A_MAX] = x;
                    }
                }
        }
    }

    /* readout: pair-ordered mean pool (PNA/src/finalize.cc:55-134) + 3-layer head (:34-52, linear.cc) */
    {
        float hg[D], o1[M1], o2[M2];
        int iters = (n + 1) / 2 - 1, tail = ((n - 1) % 2) + 1;
        for (int d = 0; d < D; d++) {
            float sum = 0.0f;
            for (int i = 0; i < iters; i++) {
                float s2 = 0.0f;
                s2 += h[(2 * i) * D + d];
                s2 += h[(2 * i + 1) * D + d];
                if (i != 0) s2 += sum;
                sum = s2;
            }
            float t = 0.0f;
            for (int k = 0; k < tail; k++) t += h[(2 * iters + k) * D + d];
            if (iters != 0) t += sum;
            hg[d] = t / (float)n;
        }
        for (int o = 0; o < M1; o++) { /* linear_output_stationary, ReLU */
            float s = w->b1[o];
            for (int i = 0; i < D; i++) s += hg[i] * w->w1[o * D + i];
            o1[o] = relu_f(s);
        }
        for (int o = 0; o < M2; o++) o2[o] = w->b2[o]; /* linear_input_stationary, PARALLEL = 2, ReLU */
        for (int i = 0; i < M1; i += 2)
            for (int o = 0; o < M2; o++) {
                float addend = 0.0f;
                addend += o1[i] * w->w2[o * M1 + i];
                addend += o1[i + 1] * w->w2[o * M1 + i + 1];
                o2[o] += addend;
            }
        for (int o = 0; o < M2; o++) o2[o] = relu_f(o2[o]);
        float r = w->b3[0]; /* linear, no ReLU */
        for (int i = 0; i < M2; i++) r += o2[i] * w->w3[i];
        out[0] = r;
    }
done:
    free(in_deg); free(out_deg); free(out_degs); free(nto); free(neighbor_tables); free(logd); free(h); free(hn); free(msg);
    return rc;
}

/* PNA_compute_graphs, PNA/src/PNA_compute.cc:7-101 (argument order of PNA/src/dcl.h:91-111).
   h_dump (optional): [5][N_tot][80], index 0 = encoder output, 1..4 = layer outputs. */
int orc_PNA_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out, const int* node_feature_in,
                           const int* edge_list_in, const float* node_embedding_weight_in,
                           const float* node_conv_weights_in, const float* node_conv_bias_in,
                           const float* graph_mlp_1_weights_in, const float* graph_mlp_1_bias_in,
                           const float* graph_mlp_2_weights_in, const float* graph_mlp_2_bias_in,
                           const float* graph_mlp_3_weights_in, const float* graph_mlp_3_bias_in,
                           const float* avg_deg_in, float* h_dump, int nthreads)
{
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
    long n_tot = noff[num_graphs];
    if (num_graphs > 0 && widx[0] < 0) { rc = 1; goto done; }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int g = 0; g < num_graphs; g++) {
        size_t s = (size_t)widx[g];
        pna_w w;
        w.nemb = node_embedding_weight_in + s * ORC_ND_FEATURE_TOTAL * D;
        w.cw = node_conv_weights_in + s * L * D * NS * NA * D;
        w.cb = node_conv_bias_in + s * L * D;
        w.w1 = graph_mlp_1_weights_in + s * M1 * D;
        w.b1 = graph_mlp_1_bias_in + s * M1;
        w.w2 = graph_mlp_2_weights_in + s * M2 * M1;
        w.b2 = graph_mlp_2_bias_in + s * M2;
        w.w3 = graph_mlp_3_weights_in + s * M2;
        w.b3 = graph_mlp_3_bias_in + s;
        w.avg_deg = avg_deg_in[s];
        int r = pna_one_graph(nums_of_nodes[g], nums_of_edges[g], node_feature_in + noff[g] * 9,
                              edge_list_in + eoff[g] * 2, &w, out + g, h_dump, n_tot, noff[g]);
        if (r) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc = r;
        }
    }
done:
    free(noff); free(eoff); free(widx);
    return rc;
}
