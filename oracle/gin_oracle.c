/*
 * gin_oracle.c -- float restatement of FlowGNN GIN (TEST INFRASTRUCTURE, parity
 * unpinned; see flowgnn_oracle.h).  Each block cites the reference lines it follows.
 */
#include "flowgnn_oracle.h"
#include <stdlib.h>
#include <string.h>

#define D 100          /* EMB_DIM,    GIN/src/dcl.h:23 */
#define H 200          /* MLP_1_OUT,  GIN/src/dcl.h:26 */
#define L 5            /* NUM_LAYERS, GIN/src/dcl.h:24 */
#define PE ORC_EDGE_PARALLEL

static const int nd_off[ORC_ND_FEATURE] = {0, 119, 123, 135, 147, 157, 163, 169, 171}; /* GIN/src/load_inputs.cc:5 */
static const int nd_card[ORC_ND_FEATURE] = {119, 4, 12, 12, 10, 6, 6, 2, 2};           /* GIN/src/host_load.cc:5 */
static const int ed_off[ORC_EDGE_ATTR] = {0, 5, 11};                                   /* GIN/src/message_passing.cc:3 */
static const int ed_card[ORC_EDGE_ATTR] = {5, 6, 2};                                   /* GIN/src/host_load.cc:6 */

static inline float relu_f(float x) { return x < 0.0f ? 0.0f : x; } /* GIN/src/util.h:22-27 */

/* GIN/src/load_inputs.cc:87-172 */
void orc_gin_load_graph(const int* edge_list, const int* edge_attr, int n, int e,
                        int* degree_table, int* degree_tables, int* neighbor_tables,
                        int* edge_attrs, int* num_of_edges_per_pe)
{
    int* nto = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1) * PE); /* neighbor_tables_offsets */
    for (int i = 0; i < n; i++) {
        degree_table[i] = 0;
        for (int p = 0; p < PE; p++) degree_tables[p * n + i] = 0;
    }
    /* pass 1: out-degree of u, total and per destination bank (:119-131) */
    for (int i = 0; i < e; i++) {
        int u = edge_list[2 * i], v = edge_list[2 * i + 1];
        degree_table[u]++;
        degree_tables[(v % PE) * n + u]++;
    }
    /* pass 2: exclusive prefix sums per PE over source id (:133-154) */
    for (int p = 0; p < PE; p++) num_of_edges_per_pe[p] = 0;
    for (int i = 0; i < n; i++)
        for (int p = 0; p < PE; p++) {
            nto[p * n + i] = num_of_edges_per_pe[p];
            num_of_edges_per_pe[p] += degree_tables[p * n + i];
        }
    /* pass 3: stable fill in input order (:156-171) */
    for (int i = 0; i < e; i++) {
        int u = edge_list[2 * i], v = edge_list[2 * i + 1];
        int p = v % PE;
        int pos = nto[p * n + u]++;
        neighbor_tables[p * e + pos] = v / PE;
        for (int k = 0; k < ORC_EDGE_ATTR; k++)
            edge_attrs[(p * e + pos) * ORC_EDGE_ATTR + k] = edge_attr[i * ORC_EDGE_ATTR + k];
    }
    free(nto);
}

typedef struct {
    const float *nemb, *eemb, *w1, *b1, *w2, *b2, *pw, *pb;
    int nt; /* NUM_TASK (GIN/src/dcl.h:25): rows of graph_pred_weights, entries of graph_pred_bias and of out[] per graph */
} gin_w;

static int gin_one_graph(int n, int e, const int* nf, const int* el, const int* ea,
                         const gin_w* w, float* out, float* h_dump, long n_tot, long node_off)
{
    size_t nn = (size_t)(n > 0 ? n : 1), ee = (size_t)(e > 0 ? e : 1);
    int* degree_table = (int*)malloc(sizeof(int) * nn);
    int* degree_tables = (int*)malloc(sizeof(int) * nn * PE);
    int* neighbor_tables = (int*)malloc(sizeof(int) * ee * PE);
    int* edge_attrs = (int*)malloc(sizeof(int) * ee * PE * ORC_EDGE_ATTR);
    int epp[PE];
    float* h = (float*)malloc(sizeof(float) * nn * D);
    float* m = (float*)malloc(sizeof(float) * nn * D);
    float acc[H];
    int rc = 0;

    for (int i = 0; i < e && !rc; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        if (u < 0 || u >= n || v < 0 || v >= n) rc = 2;
        for (int k = 0; k < ORC_EDGE_ATTR; k++)
            if (ea[i * 3 + k] < 0 || ea[i * 3 + k] >= ed_card[k]) rc = 3;
    }
    for (int v = 0; v < n && !rc; v++)
        for (int k = 0; k < ORC_ND_FEATURE; k++)
            if (nf[v * 9 + k] < 0 || nf[v * 9 + k] >= nd_card[k]) rc = 4;
    if (rc) goto done;

    orc_gin_load_graph(el, ea, n, e, degree_table, degree_tables, neighbor_tables, edge_attrs, epp);

    /* atom encoder, GIN/src/load_inputs.cc:193-212: h0[v][d] = sum_{k<9} NodeEmb[off_k+feat_k][d],
       accumulated from 0 in feature order; message row zeroed. */
    for (int v = 0; v < n; v++)
        for (int d = 0; d < D; d++) {
            float s = 0.0f;
            for (int k = 0; k < ORC_ND_FEATURE; k++) s += w->nemb[(nd_off[k] + nf[v * 9 + k]) * D + d];
            h[v * D + d] = s;
        }
    if (h_dump) memcpy(h_dump + (0 * n_tot + node_off) * D, h, sizeof(float) * (size_t)n * D);

    for (int l = 0; l < L; l++) {
        /* MP of layer l: GIN/src/message_passing.cc:96-148, one PE after another; PE p
           walks sources in ascending id, each replayed degree_tables[p][u] times. */
        memset(m, 0, sizeof(float) * nn * D);
        const float* ee_l = w->eemb + (size_t)l * ORC_ED_FEATURE_PER_LAYER * D;
        for (int p = 0; p < PE; p++) {
            int pos = 0;
            for (int u = 0; u < n; u++)
                for (int j = 0; j < degree_tables[p * n + u]; j++, pos++) {
                    int v = neighbor_tables[p * e + pos] * PE + p;
                    const int* at = &edge_attrs[(p * e + pos) * ORC_EDGE_ATTR];
                    for (int d = 0; d < D; d++) {
                        float edge_embed = 0.0f;
                        for (int k = 0; k < ORC_EDGE_ATTR; k++) edge_embed += ee_l[(ed_off[k] + at[k]) * D + d];
                        float total = edge_embed + h[u * D + d];   /* :144 */
                        m[v * D + d] += relu_f(total);              /* :145 */
                    }
                }
        }
        /* NT of layer l: GIN/src/node_embedding.cc:83-201.  eps is never loaded by the
           reference kernel (globals.cc:3, host.cc:185-200) => (1 + eps) == 1. */
        const float* w1 = w->w1 + (size_t)l * H * D;
        const float* b1 = w->b1 + (size_t)l * H;
        const float* w2 = w->w2 + (size_t)l * D * H;
        const float* b2 = w->b2 + (size_t)l * D;
        for (int v = 0; v < n; v++) {
            for (int i = 0; i < D; i++) {
                float a = m[v * D + i] + 1.0f * h[v * D + i];          /* :117 */
                for (int o = 0; o < H; o++) {
                    float addend = a * w1[o * D + i];                    /* :132 */
                    acc[o] = addend + (i == 0 ? b1[o] : acc[o]);        /* :133 */
                }
            }
            for (int d = 0; d < D; d++) {
                float r = b2[d];                                        /* :165-170 */
                for (int i = 0; i < H; i++) r += relu_f(acc[i]) * w2[d * H + i]; /* :180 */
                if (l != L - 1) r = relu_f(r);                          /* :189 */
                m[v * D + d] = r; /* reuse m as h' staging: row v of m is already consumed */
            }
        }
        memcpy(h, m, sizeof(float) * nn * D);
        if (h_dump) memcpy(h_dump + ((size_t)(l + 1) * n_tot + node_off) * D, h, sizeof(float) * (size_t)n * D);
    }

    /* readout: GIN/src/finalize.cc:36-113 sums nodes two at a time (NODE_PARALLEL=2):
       pair sum first, then the running sum is added; then / n; linear GIN/src/linear.cc:36-41 */
    {
        float hgv[D];
        int iters = (n + 1) / 2 - 1;
        int tail = ((n - 1) % 2) + 1;
        for (int d = 0; d < D; d++) {
            float sum = 0.0f;
            for (int i = 0; i < iters; i++) {
                float el2 = 0.0f;
                el2 += h[(2 * i) * D + d];
                el2 += h[(2 * i + 1) * D + d];
                if (i != 0) el2 += sum;
                sum = el2;
            }
            float t = 0.0f;
            for (int k = 0; k < tail; k++) t += h[(2 * iters + k) * D + d];
            if (iters != 0) t += sum;
            hgv[d] = t / (float)n;                                      /* finalize.cc:112 */
        }
        for (int task = 0; task < w->nt; task++) {                      /* linear<EMB_DIM, NUM_TASK, ...>, linear.cc:26-47 */
            float res = w->pb[task];
            for (int d = 0; d < D; d++) res += hgv[d] * w->pw[task * D + d];
            out[task] = res;
        }
    }
done:
    free(degree_table); free(degree_tables); free(neighbor_tables); free(edge_attrs); free(h); free(m);
    return rc;
}

int orc_GIN_compute_graphs_mt(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                              const int* reload_weights, float* out,
                              const int* node_feature_in, const int* edge_list_in,
                              const int* edge_attr_in,
                              const float* node_embedding_weight_in,
                              const float* edge_embedding_weight_in,
                              const float* node_mlp_1_weights, const float* node_mlp_1_bias,
                              const float* node_mlp_2_weights, const float* node_mlp_2_bias,
                              const float* graph_pred_weights_in, const float* graph_pred_bias_in,
                              float* h_dump, int nthreads, int num_tasks);

int orc_GIN_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out,
                           const int* node_feature_in, const int* edge_list_in,
                           const int* edge_attr_in,
                           const float* node_embedding_weight_in,
                           const float* edge_embedding_weight_in,
                           const float* node_mlp_1_weights, const float* node_mlp_1_bias,
                           const float* node_mlp_2_weights, const float* node_mlp_2_bias,
                           const float* graph_pred_weights_in, const float* graph_pred_bias_in,
                           float* h_dump, int nthreads)
{
    return orc_GIN_compute_graphs_mt(num_graphs, nums_of_nodes, nums_of_edges, reload_weights, out, node_feature_in, edge_list_in,
                                     edge_attr_in, node_embedding_weight_in, edge_embedding_weight_in, node_mlp_1_weights,
                                     node_mlp_1_bias, node_mlp_2_weights, node_mlp_2_bias, graph_pred_weights_in,
                                     graph_pred_bias_in, h_dump, nthreads, 1);
}

/* The same with NUM_TASK (GIN/src/dcl.h:25; 1 in the reference) as a run-time dimension:
   graph_pred_weights_in [S][num_tasks][100], graph_pred_bias_in [S][num_tasks], out [num_graphs][num_tasks]. */
int orc_GIN_compute_graphs_mt(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                              const int* reload_weights, float* out,
                              const int* node_feature_in, const int* edge_list_in,
                              const int* edge_attr_in,
                              const float* node_embedding_weight_in,
                              const float* edge_embedding_weight_in,
                              const float* node_mlp_1_weights, const float* node_mlp_1_bias,
                              const float* node_mlp_2_weights, const float* node_mlp_2_bias,
                              const float* graph_pred_weights_in, const float* graph_pred_bias_in,
                              float* h_dump, int nthreads, int num_tasks)
{
    /* prefix sums of node/edge offsets and the weight-set index per graph
       (GIN/src/GIN_compute.cc:44,51-53,96-97) */
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
        int s = widx[g];
        gin_w w;
        w.nemb = node_embedding_weight_in + (size_t)s * ORC_ND_FEATURE_TOTAL * D;
        w.eemb = edge_embedding_weight_in + (size_t)s * L * ORC_ED_FEATURE_PER_LAYER * D;
        w.w1 = node_mlp_1_weights + (size_t)s * L * H * D;
        w.b1 = node_mlp_1_bias + (size_t)s * L * H;
        w.w2 = node_mlp_2_weights + (size_t)s * L * D * H;
        w.b2 = node_mlp_2_bias + (size_t)s * L * D;
        w.nt = num_tasks;
        w.pw = graph_pred_weights_in + (size_t)s * num_tasks * D;
        w.pb = graph_pred_bias_in + (size_t)s * num_tasks;
        int r = gin_one_graph(nums_of_nodes[g], nums_of_edges[g],
                              node_feature_in + noff[g] * 9, edge_list_in + eoff[g] * 2,
                              edge_attr_in + eoff[g] * 3, &w, out + (size_t)g * num_tasks, h_dump, n_tot, noff[g]);
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
