/*
 * gat_oracle.c -- float restatement of FlowGNN GAT (TEST INFRASTRUCTURE, parity unpinned; see
 * flowgnn_oracle.h).  Each block cites the reference lines it follows (paths under /root/reference).
 *
 * Two documented reference quirks:
 *  - GAT_compute.cc:72 passes node_feature_in WITHOUT the per-graph node offset, so every graph reads the
 *    first num_of_nodes feature rows of the batch.  `feature_offset_quirk` != 0 reproduces that; 0 applies the
 *    offset (the intended semantics, and what the GPU engine does by default).
 *  - the raw integer atom features (0..118) are fed straight into ap_fixed<16,6>, which wraps at +-32; float
 *    semantics do not wrap.
 */
#include "flowgnn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define D 16  /* EMB_DIM,    GAT/src/dcl.h:23 */
#define H 4   /* NUM_HEADS,  GAT/src/dcl.h:24 */
#define L 5   /* NUM_LAYERS, GAT/src/dcl.h:25 */
#define PE ORC_EDGE_PARALLEL

typedef struct {
    const float *tgt, *src, *lin, *skip, *pw, *pb;
} gat_w;

/* weight index helpers: [l][head_out][dim_out][head_in][dim_in] */
#define W5(w, l, ho, dout, hi, din) (w)[(((((size_t)(l) * H + (ho)) * D + (dout)) * H + (hi)) * D) + (din)]

static int gat_one_graph(int n, int e, const int* nf, const int* el, const gat_w* w, float* out, float* dump,
                         long n_tot, long node_off)
{
    size_t nn = (size_t)(n > 0 ? n : 1), ee = (size_t)(e > 0 ? e : 1) + nn;
    int* degree_tables = (int*)calloc(nn * PE, sizeof(int));
    int* nto = (int*)calloc(nn * PE, sizeof(int));
    int* neighbor_tables = (int*)malloc(sizeof(int) * ee * PE);
    float* proj = (float*)malloc(sizeof(float) * nn * D * H);
    float* proj2 = (float*)malloc(sizeof(float) * nn * D * H);
    float* skipin = (float*)calloc(nn * D * H, sizeof(float));
    float* skipin2 = (float*)malloc(sizeof(float) * nn * D * H);
    float* ssrc = (float*)malloc(sizeof(float) * nn * H);
    float* stgt = (float*)malloc(sizeof(float) * nn * H);
    float* ssrc2 = (float*)malloc(sizeof(float) * nn * H);
    float* stgt2 = (float*)malloc(sizeof(float) * nn * H);
    float* pnum = (float*)malloc(sizeof(float) * PE * D * H);
    float* emb = (float*)malloc(sizeof(float) * nn * D);
    int epp[PE] = {0, 0, 0, 0};
    int rc = 0;

    for (int i = 0; i < e && !rc; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        if (u < 0 || u >= n || v < 0 || v >= n) rc = 2;
    }
    if (rc) goto done;

    /* load_graph, GAT/src/load_inputs.cc:87-166: keyed by destination, bank = u % 4, self edge first */
    for (int i = 0; i < n; i++) degree_tables[(i % PE) * n + i] = 1;
    for (int i = 0; i < e; i++) degree_tables[(el[2 * i] % PE) * n + el[2 * i + 1]]++;
    for (int i = 0; i < n; i++)
        for (int p = 0; p < PE; p++) {
            int a = epp[p];
            nto[p * n + i] = a;
            epp[p] = a + degree_tables[p * n + i];
            if (i % PE == p) {
                neighbor_tables[p * ee + a] = i / PE;
                nto[p * n + i] = a + 1;
            }
        }
    for (int i = 0; i < e; i++) {
        int u = el[2 * i], v = el[2 * i + 1];
        int p = u % PE;
        neighbor_tables[p * ee + nto[p * n + v]++] = u / PE;
    }

    /* load_input_node_embeddings, GAT/src/load_inputs.cc:168-226 */
    for (int v = 0; v < n; v++) {
        float pr[D][H];
        memset(pr, 0, sizeof(pr));
        for (int k = 0; k < ORC_ND_FEATURE; k++) {
            float f = (float)nf[v * 9 + k];
            skipin[(v * D + k) * H + 0] = f; /* :191 */
            for (int d = 0; d < D; d++)
                for (int ho = 0; ho < H; ho++) pr[d][ho] += f * W5(w->lin, 0, ho, d, 0, k); /* :198-200 */
        }
        float as[H] = {0, 0, 0, 0}, at[H] = {0, 0, 0, 0};
        for (int d = 0; d < D; d++)
            for (int hh = 0; hh < H; hh++) {
                float r = pr[d][hh];
                proj[(v * D + d) * H + hh] = r;
                as[hh] += r * w->src[(0 * H + hh) * D + d];
                at[hh] += r * w->tgt[(0 * H + hh) * D + d];
            }
        for (int hh = 0; hh < H; hh++) { ssrc[v * H + hh] = as[hh]; stgt[v * H + hh] = at[hh]; }
    }

    for (int l = 0; l < L; l++) {
        /* position of node v's first entry in every PE table (tables are v-major) */
        int pos[PE] = {0, 0, 0, 0};
        for (int v = 0; v < n; v++) {
            /* MP: per-PE partial numerators / denominators, GAT/src/message_passing.cc:83-151 */
            float msg[D][H], den[H] = {0, 0, 0, 0};
            memset(msg, 0, sizeof(msg));
            float pden[PE][H];
            memset(pnum, 0, sizeof(float) * PE * D * H);
            memset(pden, 0, sizeof(pden));
            for (int p = 0; p < PE; p++)
                for (int j = 0; j < degree_tables[p * n + v]; j++, pos[p]++) {
                    int u = neighbor_tables[p * ee + pos[p]] * PE + p;
                    float sc[H];
                    for (int hh = 0; hh < H; hh++) {
                        float s = ssrc[v * H + hh] + stgt[u * H + hh];     /* :122 */
                        if (s < 0) s = s * 0.2f;                            /* :126-127 */
                        sc[hh] = expf(s);                                   /* :128, no max subtraction */
                        pden[p][hh] += sc[hh];
                    }
                    for (int d = 0; d < D; d++)
                        for (int hh = 0; hh < H; hh++) pnum[(p * D + d) * H + hh] += sc[hh] * proj[(u * D + d) * H + hh]; /* :133-141 */
                }
            /* adapter: sum the four PEs, then divide, GAT/src/conv_layer.cc:158-177 */
            for (int p = 0; p < PE; p++)
                for (int hh = 0; hh < H; hh++) den[hh] += pden[p][hh];
            for (int d = 0; d < D; d++)
                for (int hh = 0; hh < H; hh++) {
                    float s = 0.0f;
                    for (int p = 0; p < PE; p++) s += pnum[(p * D + d) * H + hh];
                    msg[d][hh] = s / den[hh];
                }
            if (l < L - 1) {
                /* NT, GAT/src/node_embedding.cc:98-271 */
                float acc[D][H];
                for (int dout = 0; dout < D; dout++) {
                    float o[H];
                    for (int ho = 0; ho < H; ho++) o[ho] = msg[dout][ho];
                    for (int din = 0; din < D; din++)
                        for (int ho = 0; ho < H; ho++)
                            for (int hi = 0; hi < H; hi++) o[ho] += skipin[(v * D + din) * H + hi] * W5(w->skip, l, ho, dout, hi, din); /* :157-169 */
                    for (int ho = 0; ho < H; ho++)
                        if (o[ho] <= 0) o[ho] = expf(o[ho]) - 1.0f; /* ELU, :172-178 */
                    for (int ho = 0; ho < H; ho++) skipin2[(v * D + dout) * H + ho] = o[ho];
                    for (int pd = 0; pd < D; pd++) {
                        float a[H];
                        for (int ho = 0; ho < H; ho++) a[ho] = dout != 0 ? acc[pd][ho] : 0.0f;
                        for (int hi = 0; hi < H; hi++)
                            for (int ho = 0; ho < H; ho++) a[ho] += o[hi] * W5(w->lin, l + 1, ho, pd, hi, dout); /* :182-195 */
                        for (int ho = 0; ho < H; ho++) acc[pd][ho] = a[ho];
                    }
                }
                float as[H] = {0, 0, 0, 0}, at[H] = {0, 0, 0, 0};
                for (int d = 0; d < D; d++)
                    for (int hh = 0; hh < H; hh++) {
                        float r = acc[d][hh];
                        proj2[(v * D + d) * H + hh] = r;
                        as[hh] += r * w->src[((l + 1) * H + hh) * D + d]; /* :235-268 */
                        at[hh] += r * w->tgt[((l + 1) * H + hh) * D + d];
                    }
                for (int hh = 0; hh < H; hh++) { ssrc2[v * H + hh] = as[hh]; stgt2[v * H + hh] = at[hh]; }
            } else {
                /* last layer: mean over heads of (message + skip), GAT/src/finalize.cc:46-112 */
                for (int dout = 0; dout < D; dout++) {
                    float f = 0.0f;
                    for (int hh = 0; hh < H; hh++) f += msg[dout][hh];
                    for (int din = 0; din < D; din++)
                        for (int ho = 0; ho < H; ho++)
                            for (int hi = 0; hi < H; hi++) f += skipin[(v * D + din) * H + hi] * W5(w->skip, L - 1, ho, dout, hi, din);
                    emb[v * D + dout] = f / (float)H;
                }
            }
        }
        if (l < L - 1) {
            memcpy(proj, proj2, sizeof(float) * nn * D * H);
            memcpy(skipin, skipin2, sizeof(float) * nn * D * H);
            memcpy(ssrc, ssrc2, sizeof(float) * nn * H);
            memcpy(stgt, stgt2, sizeof(float) * nn * H);
            if (dump) memcpy(dump + ((size_t)l * n_tot + node_off) * D * H, skipin, sizeof(float) * (size_t)n * D * H);
        }
    }

    /* pair-ordered mean pool + linear (GAT/src/finalize.cc:114-, linear.cc) */
    {
        float r = w->pb[0];
        int iters = (n + 1) / 2 - 1, tail = ((n - 1) % 2) + 1;
        for (int d = 0; d < D; d++) {
            float sum = 0.0f;
            for (int i = 0; i < iters; i++) {
                float s2 = 0.0f;
                s2 += emb[(2 * i) * D + d];
                s2 += emb[(2 * i + 1) * D + d];
                if (i != 0) s2 += sum;
                sum = s2;
            }
            float t = 0.0f;
            for (int k = 0; k < tail; k++) t += emb[(2 * iters + k) * D + d];
            if (iters != 0) t += sum;
            r += (t / (float)n) * w->pw[d];
        }
        out[0] = r;
    }
done:
    free(degree_tables); free(nto); free(neighbor_tables); free(proj); free(proj2); free(skipin); free(skipin2);
    free(ssrc); free(stgt); free(ssrc2); free(stgt2); free(pnum); free(emb);
    return rc;
}

/* GAT_compute_graphs, GAT/src/GAT_compute.cc:7-112 (argument order of GAT/src/dcl.h:78-94).
   dump (optional): [4][N_tot][64], the ELU outputs (next skip inputs) of layers 0..3. */
int orc_GAT_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out, const int* node_feature_in,
                           const int* edge_list_in, const float* scoring_fn_target_in,
                           const float* scoring_fn_source_in, const float* linear_proj_weights_in,
                           const float* skip_proj_weights_in, const float* graph_pred_weights_in,
                           const float* graph_pred_bias_in, int feature_offset_quirk, float* dump, int nthreads)
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
        gat_w w;
        w.tgt = scoring_fn_target_in + s * L * H * D;
        w.src = scoring_fn_source_in + s * L * H * D;
        w.lin = linear_proj_weights_in + s * L * H * D * H * D;
        w.skip = skip_proj_weights_in + s * L * H * D * H * D;
        w.pw = graph_pred_weights_in + s * D;
        w.pb = graph_pred_bias_in + s;
        const int* nf = node_feature_in + (feature_offset_quirk ? 0 : noff[g] * 9);
        int r = gat_one_graph(nums_of_nodes[g], nums_of_edges[g], nf, edge_list_in + eoff[g] * 2, &w, out + g, dump, n_tot, noff[g]);
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
