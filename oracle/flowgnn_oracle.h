/*
 * flowgnn_oracle.h -- CPU restatement of FlowGNN's NT/MP hot path (TEST INFRASTRUCTURE).
 *
 * PARITY UNPINNED: the reference ships no golden vectors, no tests and no graph
 * packs, and its kernel sources cannot be built
 * in this image without Vitis HLS headers (ap_fixed.h, hls_stream.h, hls_math.h),
 * which are not vendored.  This oracle is therefore a from-reading restatement of
 * the reference algorithm with FM_TYPE = WT_TYPE = float, loop order and
 * summation order preserved; it is cross-checked by an independent float64 NumPy
 * restatement on the batched super-graph (tests/test_oracle_*.py against tests/numpy_ref.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call this library.  The product path (flowgnn_amd/) never does.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef FLOWGNN_ORACLE_H
#define FLOWGNN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Model constants, GIN/src/dcl.h:16-34 (same in GCN; PNA/DGN/GAT override some). */
#define ORC_ND_FEATURE 9
#define ORC_ND_FEATURE_TOTAL 173
#define ORC_EDGE_ATTR 3
#define ORC_ED_FEATURE_PER_LAYER 13
#define ORC_EDGE_PARALLEL 4

/*
 * Index bookkeeping of ONE graph exactly as GIN/src/load_inputs.cc:87-172 builds
 * it: per-PE (bank = v % 4) source-bucketed neighbour tables.  Caller allocates:
 *   degree_table[n], degree_tables[4*n] (pe-major), neighbor_tables[4*e] (pe-major,
 *   stride e), edge_attrs[4*e*3] (pe-major, stride e*3), num_of_edges_per_pe[4].
 * Integer work: the GPU CSR must reproduce the per-destination order this implies
 * bit-exactly.
 */
void orc_gin_load_graph(const int* edge_list, const int* edge_attr, int n, int e,
                        int* degree_table, int* degree_tables, int* neighbor_tables,
                        int* edge_attrs, int* num_of_edges_per_pe);

/*
 * GIN / GIN-VN forward over a concatenated batch, float semantics.
 * Mirrors GIN_compute_graphs, GIN/src/GIN_compute.cc:7-99 (argument order of
 * GIN/src/dcl.h:75-94 with float for FM_TYPE/WT_TYPE).
 * h_dump (optional, may be NULL): receives h after every NT stage,
 *   layout [6][N_tot][100] (index 0 = atom encoder output, 1..5 = layer outputs).
 * nthreads <= 1: scalar; > 1: OpenMP over graphs (results identical, graphs are
 * independent).  Returns 0, or nonzero on a feature index outside its table.
 */
int orc_GIN_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out,
                           const int* node_feature_in, const int* edge_list_in,
                           const int* edge_attr_in,
                           const float* node_embedding_weight_in,
                           const float* edge_embedding_weight_in,
                           const float* node_mlp_1_weights, const float* node_mlp_1_bias,
                           const float* node_mlp_2_weights, const float* node_mlp_2_bias,
                           const float* graph_pred_weights_in, const float* graph_pred_bias_in,
                           float* h_dump, int nthreads);

/* NUM_TASK (GIN/src/dcl.h:25, 1 in the reference) as a run-time dimension: graph_pred_weights_in [S][num_tasks][100],
 * graph_pred_bias_in [S][num_tasks], out [num_graphs][num_tasks] (linear<EMB_DIM, NUM_TASK, ...>, GIN/src/linear.cc:26-47). */
int orc_GIN_compute_graphs_mt(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                              const int* reload_weights, float* out,
                              const int* node_feature_in, const int* edge_list_in, const int* edge_attr_in,
                              const float* node_embedding_weight_in, const float* edge_embedding_weight_in,
                              const float* node_mlp_1_weights, const float* node_mlp_1_bias,
                              const float* node_mlp_2_weights, const float* node_mlp_2_bias,
                              const float* graph_pred_weights_in, const float* graph_pred_bias_in,
                              float* h_dump, int nthreads, int num_tasks);

/*
 * GCN forward, float semantics.  Mirrors GCN_compute_graphs, GCN/src/GCN_compute.cc:7-112
 * (argument order of GCN/src/dcl.h:75-97).  x_dump (optional): [5][N_tot][100], x_l = NT(l) output.
 */
/*
 * GIN / GIN-VN in the reference's own number format ap_fixed<16,6> (ginq_oracle.c: the rules it assumes, and why
 * it is "parity unpinned").  Float weights are quantised as the reference host does; out_q = 16-bit patterns,
 * out = out_q / 1024 (either may be NULL); h_dump (optional) int16 [6][N_tot][100].
 */
#include <stdint.h>
int orc_GIN_compute_graphs_q(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges, const int* reload_weights,
                             float* out, int16_t* out_q, const int* node_feature_in, const int* edge_list_in,
                             const int* edge_attr_in, const float* node_embedding_weight_in,
                             const float* edge_embedding_weight_in, const float* node_mlp_1_weights,
                             const float* node_mlp_1_bias, const float* node_mlp_2_weights, const float* node_mlp_2_bias,
                             const float* graph_pred_weights_in, const float* graph_pred_bias_in, int16_t* h_dump,
                             int nthreads);
int16_t orc_q16_from_float(float x);

int orc_GCN_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out, const int* node_feature_in,
                           const int* edge_list_in, const int* edge_attr_in,
                           const float* node_embedding_weight_in, const float* edge_embedding_weight_in,
                           const float* convs_weight_in, const float* convs_bias_in,
                           const float* convs_root_emb_weight_in, const float* bn_weight_in,
                           const float* bn_bias_in, const float* bn_mean_in, const float* bn_var_in,
                           const float* graph_pred_weights_in, const float* graph_pred_bias_in,
                           float* x_dump, int nthreads);

/*
 * PNA forward, float semantics.  Mirrors PNA_compute_graphs, PNA/src/PNA_compute.cc:7-101 (argument order
 * of PNA/src/dcl.h:91-111).  h_dump (optional): [5][N_tot][80] (0 = encoder, 1..4 = layer outputs).
 */
int orc_PNA_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out, const int* node_feature_in,
                           const int* edge_list_in, const float* node_embedding_weight_in,
                           const float* node_conv_weights_in, const float* node_conv_bias_in,
                           const float* graph_mlp_1_weights_in, const float* graph_mlp_1_bias_in,
                           const float* graph_mlp_2_weights_in, const float* graph_mlp_2_bias_in,
                           const float* graph_mlp_3_weights_in, const float* graph_mlp_3_bias_in,
                           const float* avg_deg_in, float* h_dump, int nthreads);

/*
 * DGN forward, float semantics.  Mirrors DGN_compute_graphs, DGN/src/DGN_compute.cc:6-104 (argument order of
 * DGN/src/dcl.h:71-91; node_eigen_in is float [N_tot][4]).  h_dump (optional): [5][N_tot][100].
 */
int orc_DGN_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out, const int* node_feature_in,
                           const float* node_eigen_in, const int* edge_list_in,
                           const float* embedding_h_atom_embedding_list_weights_in,
                           const float* layers_posttrans_fully_connected_0_linear_weight_in,
                           const float* layers_posttrans_fully_connected_0_linear_bias_in,
                           const float* MLP_layer_FC_layers_0_weight_in, const float* MLP_layer_FC_layers_0_bias_in,
                           const float* MLP_layer_FC_layers_1_weight_in, const float* MLP_layer_FC_layers_1_bias_in,
                           const float* MLP_layer_FC_layers_2_weight_in, const float* MLP_layer_FC_layers_2_bias_in,
                           float* h_dump, int nthreads);

/*
 * GAT forward, float semantics.  Mirrors GAT_compute_graphs, GAT/src/GAT_compute.cc:7-112 (argument order of
 * GAT/src/dcl.h:78-94).  feature_offset_quirk != 0 reproduces GAT_compute.cc:72 (node features read without
 * the per-graph offset).  dump (optional): [4][N_tot][64] = ELU outputs of layers 0..3, index dim*4 + head.
 */
int orc_GAT_compute_graphs(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                           const int* reload_weights, float* out, const int* node_feature_in,
                           const int* edge_list_in, const float* scoring_fn_target_in,
                           const float* scoring_fn_source_in, const float* linear_proj_weights_in,
                           const float* skip_proj_weights_in, const float* graph_pred_weights_in,
                           const float* graph_pred_bias_in, int feature_offset_quirk, float* dump, int nthreads);

/*
 * GCN / GAT / PNA (ap_fixed<16,6>) and DGN (ap_fixed<16,3>) in the reference's own number formats: q_oracle.c (the rules it
 * assumes, R0..R8, and why it is "parity unpinned").  model = FLOWGNN_MODEL_* id (2 GCN, 3 GAT, 4 PNA, 5 DGN); tens = the model's
 * float weight tensors in entry-point order ([S] leading), elems[i] = elements of ONE weight set of tensor i; out_q = 16-bit
 * patterns, out = pattern / 2^F (either may be NULL).
 */
int orc_q_compute_graphs(int model, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges, const int* reload_weights,
                         float* out, int16_t* out_q, const int* node_feature_in, const float* node_eigen_in, const int* edge_list_in,
                         const int* edge_attr_in, int ntens, const float* const* tens, const long* elems, int gat_feature_offset_quirk,
                         int nthreads);
int16_t orc_q_from_float(float x, int frac_bits);
const int16_t* orc_q_exp_table(void);   /* [65536], indexed by the Q6.10 pattern as uint16 */
int16_t orc_q_log(int16_t pattern);

#ifdef __cplusplus
}
#endif
#endif
