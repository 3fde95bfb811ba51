/*
 * gcn_oracle.c -- float restatement of FlowGNN GCN (TEST INFRASTRUCTURE, parity unpinned; see
 * flowgnn_oracle.h).  Each block cites the reference lines it follows (paths under /root/reference).
 */
#include "flowgnn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define D 100 /* EMB_DIM,    GCN/src/dcl.h:23 */
#define L 5   /* NUM_LAYERS, GCN/src/dcl.h:24 */
#define PE ORC_EDGE_PARALLEL

static const int nd_off[ORC_ND_FEATURE] = {0, 119, 123, 135, 147, 157, 163, 169, 171}; /* GCN/src/load_inputs.cc:5 */
static const int nd_card[ORC_ND_FEATURE] = {119, 4, 12, 12, 10, 6, 6, 2, 2};
static const int ed_off[ORC_EDGE_ATTR] = {0, 5, 11};                                   /* GCN/src/message_passing.cc:3 */
static const int ed_card[ORC_EDGE_ATTR] = {5, 6, 2};

static inline float relu_f(float x) { return x < 0.0f ? 0.0f : x; }

typedef struct {
    const float *nemb, *eemb, *cw, *cb, *root, *bnw, *bnb, *bnm, *bnv, *pw, *pb;
    int nt; /* NUM_TASK (GCN/src/dcl.h): rows of graph_pred_weights, entries of graph_pred_bias and of out[] per graph */
} gcn_w;

static int gcn_one_graph(int n, int e, const int* nf, const int* el, const int* ea, const gcn_w* w, float* out,
                         float* x_dump, long n_tot, long node_off)
{
    size_t nn = (size_t)(n > 0 ? n : 1), ee = (size_t)(e > 0 ? e : 1);
    int* degree_table = (int*)calloc(nn, sizeof(int));
    int* degree_tables = (int*)calloc(nn * PE, sizeof(int));
    int* nto = (int*)calloc(nn * PE, sizeof(int));
    int* neighbor_tables = (int*)malloc(sizeof(int) * ee * PE);
    int* edge_attrs = (int*)malloc(sizeof(int) * ee * PE * 3);
    float* norms = (float*)malloc(sizeof(float) * ee * PE);
    float* dinv = (float*)calloc(nn, sizeof(float));
    float* h = (float*)malloc(sizeof(float) * nn * D);   /* h_node: encoder output, then x_l */
    float* m = (float*)malloc(sizeof(float) * nn * D);   /* message buffer */
    float* m2 = (float*)malloc(sizeof(float) * nn * D);
    float bn_sqrt_var[L][D];
    float acc[D];
    int epp[PE] = {0, 0, 0, 0};
    int rc = 0;

    for (int i = 0; i < e && !rc; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        if (u < 0 || u >= n || v < 0 || v >= n) rc = 2;
        for (int k = 0; k < 3; k++)
            if (ea[i * 3 + k] < 0 || ea[i * 3 + k] >= ed_card[k]) rc = 3;
    }
    for (int v = 0; v < n && !rc; v++)
        for (int k = 0; k < ORC_ND_FEATURE; k++)
            if (nf[v * 9 + k] < 0 || nf[v * 9 + k] >= nd_card[k]) rc = 4;
    if (rc) goto done;

    /* load_weights: bn_sqrt_var = sqrt(var + epsilon(ap_fixed<16,6>) = 2^-10), GCN/src/load_inputs.cc:32 */
    for (int l = 0; l < L; l++)
        for (int d = 0; d < D; d++) bn_sqrt_var[l][d] = sqrtf(w->bnv[l * D + d] + (1.0f / 1024.0f));

    /* load_graph, GCN/src/load_inputs.cc:99-166.  degree_inv_sqrt[u] is rewritten on every out-edge of u
       (:122), so it ends as 1/sqrt(outdeg(u)+1) for nodes WITH out-edges and stays 0 for the others. */
    for (int i = 0; i < e; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        degree_table[u]++;
        degree_tables[(v % PE) * n + u]++;
        dinv[u] = 1.0f / sqrtf((float)(degree_table[u] + 1));
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
        norms[p * e + pos] = dinv[u] * dinv[v]; /* :163 */
        for (int k = 0; k < 3; k++) edge_attrs[(p * e + pos) * 3 + k] = ea[i * 3 + k];
    }

    /* atom encoder, GCN/src/load_inputs.cc:168-215 */
    for (int v = 0; v < n; v++)
        for (int d = 0; d < D; d++) {
            float s = 0.0f;
            for (int k = 0; k < ORC_ND_FEATURE; k++) s += w->nemb[(nd_off[k] + nf[v * 9 + k]) * D + d];
            h[v * D + d] = s;
        }
    memset(m, 0, sizeof(float) * nn * D);

    for (int l = 0; l < L; l++) {
        /* NT(l), GCN/src/node_embedding.cc:93-148 */
        const float* W = w->cw + (size_t)l * D * D;
        for (int v = 0; v < n; v++) {
            for (int i = 0; i < D; i++) {
                float act;
                if (l == 0) {
                    act = h[v * D + i];
                } else {
                    act = m[v * D + i] + relu_f(h[v * D + i] + w->root[(l - 1) * D + i]) / (float)(degree_table[v] + 1); /* :135 */
                    act = (act - w->bnm[(l - 1) * D + i]) / bn_sqrt_var[l - 1][i] * w->bnw[(l - 1) * D + i] + w->bnb[(l - 1) * D + i]; /* :136 */
                    act = relu_f(act);
                }
                for (int o = 0; o < D; o++) {
                    float addend = act * W[o * D + i];
                    acc[o] = addend + (i == 0 ? w->cb[l * D + o] : acc[o]);
                }
            }
            memcpy(&m2[v * D], acc, sizeof(acc)); /* x_l[v]; h[] still holds x_{l-1} for later nodes */
        }
        memcpy(h, m2, sizeof(float) * nn * D);
        if (x_dump) memcpy(x_dump + ((size_t)l * n_tot + node_off) * D, h, sizeof(float) * (size_t)n * D);
        /* MP(l), GCN/src/message_passing.cc:124-170 */
        memset(m, 0, sizeof(float) * nn * D);
        const float* ee_l = w->eemb + (size_t)l * ORC_ED_FEATURE_PER_LAYER * D;
        for (int p = 0; p < PE; p++) {
            int pos = 0;
            for (int u = 0; u < n; u++)
                for (int j = 0; j < degree_tables[p * n + u]; j++, pos++) {
                    int v = neighbor_tables[p * e + pos] * PE + p;
                    float norm = norms[p * e + pos];
                    const int* at = &edge_attrs[(p * e + pos) * 3];
                    for (int d = 0; d < D; d++) {
                        float edge_embed = 0.0f;
                        for (int k = 0; k < 3; k++) edge_embed += ee_l[(ed_off[k] + at[k]) * D + d];
                        float total = edge_embed + h[u * D + d];
                        m[v * D + d] += norm * relu_f(total); /* :167 */
                    }
                }
        }
    }

    /* finalize, GCN/src/finalize.cc:39-113 + linear_input_stationary (GCN/src/linear.cc), PARALLEL = 2 */
    {
        float hg[D];
        for (int d = 0; d < D; d++) {
            float sum = 0.0f;
            for (int v = 0; v < n; v++) {
                float act = m[v * D + d];
                act += relu_f(h[v * D + d] + w->root[(L - 1) * D + d]) / (float)(degree_table[v] + 1);
                act = (act - w->bnm[(L - 1) * D + d]) / bn_sqrt_var[L - 1][d] * w->bnw[(L - 1) * D + d] + w->bnb[(L - 1) * D + d];
                sum += act;
            }
            hg[d] = sum / (float)n;
        }
        for (int task = 0; task < w->nt; task++) {
            float o = w->pb[task];
            for (int d = 0; d < D; d += 2) {
                float addend = 0.0f;
                addend += hg[d] * w->pw[task * D + d];
                addend += hg[d + 1] * w->pw[task * D + d + 1];
                o += addend;
            }
            out[task] = o;
        }
    }
done:
    free(degree_table); free(degree_tables); free(nto); free(neighbor_tables); free(edge_attrs); free(norms);
    free(dinv); free(h); free(m); free(m2);
    return rc;
}

/* GCN_compute_graphs, GCN/src/GCN_compute.cc:7-112 (argument order of GCN/src/dcl.h:75-97).
   x_dump (optional): [5][N_tot][100], x_l = output of NT(l). */
int orc_GCN_compute_graphs_mt(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                              const int* reload_weights, float* out, const int* node_feature_in,
                              const int* edge_list_in, const int* edge_attr_in,
                              const float* node_embedding_weight_in, const float* edge_embedding_weight_in,
                              const float* convs_weight_in, const float* convs_bias_in,
                              const float* convs_root_emb_weight_in, const float* bn_weight_in,
                              const float* bn_bias_in, const float* bn_mean_in, const float* bn_var_in,
                              const float* graph_pred_weights_in, const float* graph_pred_bias_in,
                              float* x_dump, int nthreads, int num_tasks);

int orc_GCN_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out, const int* node_feature_in,
                           const int* edge_list_in, const int* edge_attr_in,
                           const float* node_embedding_weight_in, const float* edge_embedding_weight_in,
                           const float* convs_weight_in, const float* convs_bias_in,
                           const float* convs_root_emb_weight_in, const float* bn_weight_in,
                           const float* bn_bias_in, const float* bn_mean_in, const float* bn_var_in,
                           const float* graph_pred_weights_in, const float* graph_pred_bias_in,
                           float* x_dump, int nthreads)
{
    return orc_GCN_compute_graphs_mt(num_graphs, nums_of_nodes, nums_of_edges, reload_weights, out, node_feature_in, edge_list_in,
                                     edge_attr_in, node_embedding_weight_in, edge_embedding_weight_in, convs_weight_in, convs_bias_in,
                                     convs_root_emb_weight_in, bn_weight_in, bn_bias_in, bn_mean_in, bn_var_in,
                                     graph_pred_weights_in, graph_pred_bias_in, x_dump, nthreads, 1);
}

/* The same with NUM_TASK as a run-time dimension: graph_pred_weights_in [S][num_tasks][100], graph_pred_bias_in [S][num_tasks],
   out [num_graphs][num_tasks]. */
int orc_GCN_compute_graphs_mt(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                              const int* reload_weights, float* out, const int* node_feature_in,
                              const int* edge_list_in, const int* edge_attr_in,
                              const float* node_embedding_weight_in, const float* edge_embedding_weight_in,
                              const float* convs_weight_in, const float* convs_bias_in,
                              const float* convs_root_emb_weight_in, const float* bn_weight_in,
                              const float* bn_bias_in, const float* bn_mean_in, const float* bn_var_in,
                              const float* graph_pred_weights_in, const float* graph_pred_bias_in,
                              float* x_dump, int nthreads, int num_tasks)
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
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int g = 0; g < num_graphs; g++) {
        size_t s = (size_t)widx[g];
        gcn_w w;
        w.nemb = node_embedding_weight_in + s * ORC_ND_FEATURE_TOTAL * D;
        w.eemb = edge_embedding_weight_in + s * L * ORC_ED_FEATURE_PER_LAYER * D;
        w.cw = convs_weight_in + s * L * D * D;
        w.cb = convs_bias_in + s * L * D;
        w.root = convs_root_emb_weight_in + s * L * D;
        w.bnw = bn_weight_in + s * L * D;
        w.bnb = bn_bias_in + s * L * D;
        w.bnm = bn_mean_in + s * L * D;
        w.bnv = bn_var_in + s * L * D;
        w.nt = num_tasks;
        w.pw = graph_pred_weights_in + s * num_tasks * D;
        w.pb = graph_pred_bias_in + s * num_tasks;
        int r = gcn_one_graph(nums_of_nodes[g], nums_of_edges[g], node_feature_in + noff[g] * 9,
                              edge_list_in + eoff[g] * 2, edge_attr_in + eoff[g] * 3, &w, out + (size_t)g * num_tasks, x_dump, n_tot, noff[g]);
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
