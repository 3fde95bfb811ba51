/*
 * dgn_oracle.c -- float restatement of FlowGNN DGN (TEST INFRASTRUCTURE, parity unpinned; see
 * flowgnn_oracle.h).  Each block cites the reference lines it follows (paths under /root/reference).
 *
 * Defined deviation: the reference divides the sum aggregate by the node's OUT-degree with no guard
 * (DGN/src/node_embedding.cc:143); ap_fixed x / 0 evaluates to 0 (the PNA code relies on exactly that,
 * PNA/src/node_embedding.cc:149-150), float would give inf / NaN.  The oracle and the GPU path both use
 * "x / 0 = 0" there.
 */
#include "flowgnn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define D 100  /* EMB_DIM,    DGN/src/dcl.h:17 */
#define L 4    /* NUM_LAYERS, DGN/src/dcl.h:21 */
#define M1 50
#define M2 25
#define PE ORC_EDGE_PARALLEL
#define TBL 119 /* rows per atom table, DGN/src/dcl.h:76 */

static const int nd_card[ORC_ND_FEATURE] = {119, 4, 12, 12, 10, 6, 6, 2, 2};
static inline float relu_f(float x) { return x < 0.0f ? 0.0f : x; }

typedef struct {
    const float *emb, *lw, *lb, *w0, *b0, *w1, *b1, *w2, *b2;
} dgn_w;

static int dgn_one_graph(int n, int e, const int* nf, const float* eig, const int* el, const dgn_w* w, float* out,
                         float* h_dump, long n_tot, long node_off)
{
    size_t nn = (size_t)(n > 0 ? n : 1), ee = (size_t)(e > 0 ? e : 1);
    int* degree_table = (int*)calloc(nn, sizeof(int));
    int* degree_tables = (int*)calloc(nn * PE, sizeof(int));
    int* nto = (int*)calloc(nn * PE, sizeof(int));
    int* neighbor_tables = (int*)malloc(sizeof(int) * ee * PE);
    float* eig_w = (float*)malloc(sizeof(float) * ee * PE);
    float* eig_abssums = (float*)calloc(nn, sizeof(float));
    float* eigw_sums = (float*)calloc(nn, sizeof(float));
    float* h = (float*)malloc(sizeof(float) * nn * D);
    float* hn = (float*)malloc(sizeof(float) * nn * D);
    float* msg = (float*)malloc(sizeof(float) * nn * 2 * D);
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

    /* load_graph, DGN/src/load_inputs.cc:28-112 */
    for (int i = 0; i < e; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        degree_table[u]++;
        degree_tables[(v % PE) * n + u]++;
    }
    for (int i = 0; i < n; i++)
        for (int p = 0; p < PE; p++) {
            nto[p * n + i] = epp[p];
            epp[p] += degree_tables[p * n + i];
        }
    for (int i = 0; i < e; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        int p = v % PE;
        int pos = nto[p * n + u]++;
        neighbor_tables[p * e + pos] = v / PE;
        float diff = eig[u * 4 + 1] - eig[v * 4 + 1]; /* eigenvector #1, :105-107 */
        eig_w[p * e + pos] = diff;
        eig_abssums[v] += fabsf(diff);
        eigw_sums[v] += diff;
    }

    /* atom encoder: dense [9][119][100] table indexed [k][feat_k], DGN/src/load_inputs.cc:114-172 */
    for (int v = 0; v < n; v++)
        for (int d = 0; d < D; d++) {
            float s = 0.0f;
            for (int k = 0; k < ORC_ND_FEATURE; k++) s += w->emb[((size_t)k * TBL + nf[v * 9 + k]) * D + d];
            h[v * D + d] = s;
        }
    if (h_dump) memcpy(h_dump + (0 * n_tot + node_off) * D, h, sizeof(float) * (size_t)n * D);

    for (int l = 0; l <= L; l++) {
        if (l > 0) {
            /* NT(l-1), DGN/src/node_embedding.cc:107-181; W viewed as [out][2][in] */
            const float* W = w->lw + (size_t)(l - 1) * D * 2 * D;
            for (int v = 0; v < n; v++) {
                float abssum = eig_abssums[v] == 0.0f ? (1.0f / 8192.0f) : eig_abssums[v]; /* epsilon of ap_fixed<16,3>, :125-128 */
                float deg = (float)degree_table[v];
                for (int i = 0; i < D; i++) {
                    float m1 = msg[(v * 2 + 0) * D + i], m2 = msg[(v * 2 + 1) * D + i];
                    float hv = h[v * D + i];
                    float a1 = degree_table[v] == 0 ? 0.0f : m1 / deg;         /* :143, see header */
                    float a2 = fabsf((m2 - eigw_sums[v] * hv) / abssum);       /* :144 */
                    for (int o = 0; o < D; o++) {
                        float addend = a1 * W[(o * 2 + 0) * D + i] + a2 * W[(o * 2 + 1) * D + i];
                        acc[o] = addend + (i == 0 ? w->lb[(l - 1) * D + o] : acc[o]);
                    }
                }
                for (int d = 0; d < D; d++) hn[v * D + d] = h[v * D + d] + relu_f(acc[d]); /* :176-181 */
            }
            memcpy(h, hn, sizeof(float) * nn * D);
            if (h_dump) memcpy(h_dump + ((size_t)l * n_tot + node_off) * D, h, sizeof(float) * (size_t)n * D);
        }
        if (l == L) break;
        /* MP, DGN/src/message_passing.cc:120-152 */
        memset(msg, 0, sizeof(float) * nn * 2 * D);
        for (int p = 0; p < PE; p++) {
            int pos = 0;
            for (int u = 0; u < n; u++)
                for (int j = 0; j < degree_tables[p * n + u]; j++, pos++) {
                    int v = neighbor_tables[p * e + pos] * PE + p;
                    float ew = eig_w[p * e + pos];
                    for (int d = 0; d < D; d++) {
                        msg[(v * 2 + 0) * D + d] += h[u * D + d];
                        msg[(v * 2 + 1) * D + d] += h[u * D + d] * ew;
                    }
                }
        }
    }

    /* readout: pair-ordered mean pool + head 100 -> 50 -> 25 -> 1, DGN/src/finalize.cc:28-52 */
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
        for (int o = 0; o < M1; o++) {
            float s = w->b0[o];
            for (int i = 0; i < D; i++) s += hg[i] * w->w0[o * D + i];
            o1[o] = relu_f(s);
        }
        for (int o = 0; o < M2; o++) o2[o] = w->b1[o];
        for (int i = 0; i < M1; i += 2)
            for (int o = 0; o < M2; o++) {
                float addend = 0.0f;
                addend += o1[i] * w->w1[o * M1 + i];
                addend += o1[i + 1] * w->w1[o * M1 + i + 1];
                o2[o] += addend;
            }
        for (int o = 0; o < M2; o++) o2[o] = relu_f(o2[o]);
        float r = w->b2[0];
        for (int i = 0; i < M2; i++) r += o2[i] * w->w2[i];
        out[0] = r;
    }
done:
    free(degree_table); free(degree_tables); free(nto); free(neighbor_tables); free(eig_w); free(eig_abssums);
    free(eigw_sums); free(h); free(hn); free(msg);
    return rc;
}

/* DGN_compute_graphs, DGN/src/DGN_compute.cc:6-104 (argument order of DGN/src/dcl.h:71-91).
   h_dump (optional): [5][N_tot][100]. */
int orc_DGN_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out, const int* node_feature_in,
                           const float* node_eigen_in, const int* edge_list_in,
                           const float* embedding_h_atom_embedding_list_weights_in,
                           const float* layers_posttrans_fully_connected_0_linear_weight_in,
                           const float* layers_posttrans_fully_connected_0_linear_bias_in,
                           const float* MLP_layer_FC_layers_0_weight_in, const float* MLP_layer_FC_layers_0_bias_in,
                           const float* MLP_layer_FC_layers_1_weight_in, const float* MLP_layer_FC_layers_1_bias_in,
                           const float* MLP_layer_FC_layers_2_weight_in, const float* MLP_layer_FC_layers_2_bias_in,
                           float* h_dump, int nthreads)
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
        dgn_w w;
        w.emb = embedding_h_atom_embedding_list_weights_in + s * 9 * TBL * D;
        w.lw = layers_posttrans_fully_connected_0_linear_weight_in + s * L * D * 2 * D;
        w.lb = layers_posttrans_fully_connected_0_linear_bias_in + s * L * D;
        w.w0 = MLP_layer_FC_layers_0_weight_in + s * M1 * D;
        w.b0 = MLP_layer_FC_layers_0_bias_in + s * M1;
        w.w1 = MLP_layer_FC_layers_1_weight_in + s * M2 * M1;
        w.b1 = MLP_layer_FC_layers_1_bias_in + s * M2;
        w.w2 = MLP_layer_FC_layers_2_weight_in + s * M2;
        w.b2 = MLP_layer_FC_layers_2_bias_in + s;
        int r = dgn_one_graph(nums_of_nodes[g], nums_of_edges[g], node_feature_in + noff[g] * 9, node_eigen_in + noff[g] * 4,
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
