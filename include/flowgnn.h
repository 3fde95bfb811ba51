/*
 * flowgnn.h -- C ABI of the MI355X-native FlowGNN inference engine.
 *
 * This is the drop-in boundary for FlowGNN's NT/MP hot path.  Two layers:
 *
 *  (1) Reference-compatible entry points  <M>_compute_graphs(...)
 *      Same symbol names and argument order as the reference's HLS kernels
 *      (cited per function; paths relative to the reference repo), with `float` in
 *      place of FM_TYPE/WT_TYPE (ap_fixed<16,6>) and `int` status instead of `void`.
 *      All pointers are caller-owned HOST buffers; `out` is written on return.
 *      Graphs are concatenated with node ids LOCAL to each graph, exactly as the
 *      reference host builds them (GIN/src/host.cc:119-138).
 *
 *  (2) Handle API  flowgnn_*  (what (1) is implemented on)
 *      One engine per GPU, one HIP stream per engine; weights and the graph batch
 *      stay resident in HBM across runs, so a caller can time the device path
 *      alone, as the reference times kernel execution alone (run_experiments.sh:44).
 *
 * No torch / C++ types cross this boundary: plain pointers and sizes only.
 */
#ifndef FLOWGNN_H
#define FLOWGNN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the reference returns void and never checks) ---- */
#define FLOWGNN_OK 0
#define FLOWGNN_ERR_ARG 1          /* null pointer, negative count, num_nodes <= 0 */
#define FLOWGNN_ERR_EDGE_RANGE 2   /* edge endpoint outside [0, num_nodes) of its graph */
#define FLOWGNN_ERR_EDGE_ATTR 3    /* edge attribute outside its embedding table */
#define FLOWGNN_ERR_NODE_FEAT 4    /* node feature outside its embedding table */
#define FLOWGNN_ERR_HIP 5          /* HIP runtime error (see flowgnn_last_error) */
#define FLOWGNN_ERR_STATE 6        /* weights or batch not set */
#define FLOWGNN_ERR_IO 7           /* weight / graph file missing or short */
#define FLOWGNN_ERR_UNSUPPORTED 8  /* also: a graph larger than 2 048 nodes / 16 384 edges with a node of more than 16 384 in-edges */

/* ---- model ids ---- */
#define FLOWGNN_MODEL_GIN 0
#define FLOWGNN_MODEL_GIN_VN 1 /* same kernel as GIN; host appends a virtual node (GIN-VN/src/host_load.cc:125-153) */
#define FLOWGNN_MODEL_GCN 2
#define FLOWGNN_MODEL_GAT 3
#define FLOWGNN_MODEL_PNA 4
#define FLOWGNN_MODEL_DGN 5

/* ---- fixed model constants (GIN/src/dcl.h:16-26) ---- */
#define FLOWGNN_ND_FEATURE 9
#define FLOWGNN_ND_FEATURE_TOTAL 173
#define FLOWGNN_EDGE_ATTR 3
#define FLOWGNN_ED_FEATURE_PER_LAYER 13

/* =====================================================================
 * (1) Reference-compatible entry points
 * ===================================================================== */

/*
 * Replaces GIN_compute_graphs, GIN/src/dcl.h:75-94 (def. GIN/src/GIN_compute.cc:7-99);
 * also the GIN-VN kernel (GIN-VN/src/dcl.h, byte-identical).
 *   out                      [num_graphs][1]
 *   node_feature_in          int [N_tot][9]
 *   edge_list_in             int [E_tot][2]  (u, v): h[u] is sent to v
 *   edge_attr_in             int [E_tot][3]
 *   *_weight*                leading dimension = weight-set index, selected by the
 *                            running count of reload_weights[] (GIN_compute.cc:51-53)
 *   node_embedding_weight_in [S][173][100]   edge_embedding_weight_in [S][5][13][100]
 *   node_mlp_1_weights [S][5][200][100]  node_mlp_1_bias [S][5][200]
 *   node_mlp_2_weights [S][5][100][200]  node_mlp_2_bias [S][5][100]
 *   graph_pred_weights_in [S][1][100]    graph_pred_bias_in [S][1]
 */
int GIN_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges,
                       int* reload_weights, float* out,
                       int* node_feature_in, int* edge_list_in, int* edge_attr_in,
                       float* node_embedding_weight_in, float* edge_embedding_weight_in,
                       float* node_mlp_1_weights, float* node_mlp_1_bias,
                       float* node_mlp_2_weights, float* node_mlp_2_bias,
                       float* graph_pred_weights_in, float* graph_pred_bias_in);

/*
 * NUM_TASK (a compile-time constant of the reference build, GIN/src/dcl.h:25, 1 as shipped; ogbg-molpcba has 128) as an explicit
 * last argument: graph_pred_weights_in is then [S][num_tasks][100], graph_pred_bias_in [S][num_tasks], out [num_graphs][num_tasks]
 * (`FM_TYPE out[][NUM_TASK]`, GIN/src/dcl.h:80).  GIN_compute_graphs(...) == GIN_compute_graphs_mt(..., 1); nothing about the
 * shape of `out` is ever taken from the environment.
 */
int GIN_compute_graphs_mt(int num_graphs, int* nums_of_nodes, int* nums_of_edges,
                          int* reload_weights, float* out,
                          int* node_feature_in, int* edge_list_in, int* edge_attr_in,
                          float* node_embedding_weight_in, float* edge_embedding_weight_in,
                          float* node_mlp_1_weights, float* node_mlp_1_bias,
                          float* node_mlp_2_weights, float* node_mlp_2_bias,
                          float* graph_pred_weights_in, float* graph_pred_bias_in, int num_tasks);

/*
 * Replaces GCN_compute_graphs, GCN/src/dcl.h:75-97 (def. GCN/src/GCN_compute.cc:7-112).
 *   out [num_graphs]; graph arrays as for GIN
 *   convs_weight_in [S][5][100][100]  convs_bias_in [S][5][100]  convs_root_emb_weight_in [S][5][100]
 *   bn_weight_in / bn_bias_in / bn_mean_in / bn_var_in [S][5][100]   (eval-mode BatchNorm)
 */
int GCN_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges,
                       int* reload_weights, float* out,
                       int* node_feature_in, int* edge_list_in, int* edge_attr_in,
                       float* node_embedding_weight_in, float* edge_embedding_weight_in,
                       float* convs_weight_in, float* convs_bias_in, float* convs_root_emb_weight_in,
                       float* bn_weight_in, float* bn_bias_in, float* bn_mean_in, float* bn_var_in,
                       float* graph_pred_weights_in, float* graph_pred_bias_in);

/* GCN with NUM_TASK as an explicit last argument (see GIN_compute_graphs_mt). */
int GCN_compute_graphs_mt(int num_graphs, int* nums_of_nodes, int* nums_of_edges,
                          int* reload_weights, float* out,
                          int* node_feature_in, int* edge_list_in, int* edge_attr_in,
                          float* node_embedding_weight_in, float* edge_embedding_weight_in,
                          float* convs_weight_in, float* convs_bias_in, float* convs_root_emb_weight_in,
                          float* bn_weight_in, float* bn_bias_in, float* bn_mean_in, float* bn_var_in,
                          float* graph_pred_weights_in, float* graph_pred_bias_in, int num_tasks);

/*
 * Replaces PNA_compute_graphs, PNA/src/dcl.h:91-111 (def. PNA/src/PNA_compute.cc:7-101).  No edge features.
 *   node_embedding_weight_in [S][173][80]
 *   node_conv_weights_in [S][4][80][3][4][80]  (layer, out, scaler {none,t,scale}, aggregator {mean,min,max,std}, in)
 *   node_conv_bias_in [S][4][80]
 *   graph_mlp_1_weights_in [S][40][80] / _bias [S][40];  graph_mlp_2 [S][20][40] / [S][20];  graph_mlp_3 [S][1][20] / [S][1]
 *   avg_deg_in [S]   (the reference host passes 6.885701656341553, PNA/src/host_load.cc:127)
 */
int PNA_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges,
                       int* reload_weights, float* out,
                       int* node_feature_in, int* edge_list_in,
                       float* node_embedding_weight_in,
                       float* node_conv_weights_in, float* node_conv_bias_in,
                       float* graph_mlp_1_weights_in, float* graph_mlp_1_bias_in,
                       float* graph_mlp_2_weights_in, float* graph_mlp_2_bias_in,
                       float* graph_mlp_3_weights_in, float* graph_mlp_3_bias_in,
                       float* avg_deg_in);

/*
 * Replaces DGN_compute_graphs, DGN/src/dcl.h:71-91 (def. DGN/src/DGN_compute.cc:6-104).  No edge features.
 *   node_eigen_in float [N_tot][4]  (node_eigen_t, DGN/src/dcl.h:67; column 1 is the one used)
 *   embedding_h_atom_embedding_list_weights_in [S][9][119][100]  (dense per-feature tables)
 *   layers_posttrans_fully_connected_0_linear_weight_in [S][4][100][200] (= [out][2][in]) / _bias_in [S][4][100]
 *   MLP_layer_FC_layers_0 [S][50][100] / [S][50];  _1 [S][25][50] / [S][25];  _2 [S][1][25] / [S][1]
 */
int DGN_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges,
                       int* reload_weights, float* out,
                       int* node_feature_in, float* node_eigen_in, int* edge_list_in,
                       float* embedding_h_atom_embedding_list_weights_in,
                       float* layers_posttrans_fully_connected_0_linear_weight_in,
                       float* layers_posttrans_fully_connected_0_linear_bias_in,
                       float* MLP_layer_FC_layers_0_weight_in, float* MLP_layer_FC_layers_0_bias_in,
                       float* MLP_layer_FC_layers_1_weight_in, float* MLP_layer_FC_layers_1_bias_in,
                       float* MLP_layer_FC_layers_2_weight_in, float* MLP_layer_FC_layers_2_bias_in);

/*
 * Replaces GAT_compute_graphs, GAT/src/dcl.h:78-94 (def. GAT/src/GAT_compute.cc:7-112).  No edge features; the
 * node features are used as raw numbers (GAT/src/load_inputs.cc:190-191).
 *   scoring_fn_target_in / scoring_fn_source_in [S][5][4][16]
 *   linear_proj_weights_in / skip_proj_weights_in [S][5][4][16][4][16]  (layer, head_out, dim_out, head_in, dim_in);
 *       layer 0 uses only [head_out][dim_out][0][dim_in < 9] (GAT/src/host_load.cc:69-78)
 *   graph_pred_weights_in [S][1][16], graph_pred_bias_in [S][1]
 * Per-graph node-feature offsets ARE applied (the reference omits them, GAT_compute.cc:72); option
 * gat_reference_quirk = 1 (flowgnn_set_option / flowgnn_entry_set_option) reproduces the reference behaviour.
 */
int GAT_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges,
                       int* reload_weights, float* out,
                       int* node_feature_in, int* edge_list_in,
                       float* scoring_fn_target_in, float* scoring_fn_source_in,
                       float* linear_proj_weights_in, float* skip_proj_weights_in,
                       float* graph_pred_weights_in, float* graph_pred_bias_in);

/*
 * Devices and options of the entry points above (they have no handle to carry them).
 *  flowgnn_entry_set_devices: the HIP devices the six entry points run on.  With more than one, every run of constant weight
 *     set is cut into contiguous graph ranges balanced by sum(N + E) and the ranges run concurrently, one engine and one host
 *     thread per listed device (a device may be listed twice); `out` is filled in job order.  Default: device 0, or the list in
 *     the environment variable FLOWGNN_DEVICES (e.g. "0,1,2,3,4,5,6,7") read at the first call.
 *  flowgnn_entry_set_option: flowgnn_set_option for the engines behind the entry points of `model` (FLOWGNN_MODEL_*).
 *  flowgnn_entry_set_pipeline: the entry points take HOST arrays, so a large batch is cut into several ranges per engine and an
 *     engine's host -> device copy of its next range runs under the other engines' kernels (flowgnn_group_compute); with ONE listed
 *     device the entry points keep three engines on it for that.  0 (default): ranges of ~48 MB of host arrays, at most 8 per
 *     engine, small batches uncut on one engine; k >= 1: exactly k ranges per engine (1 = no pipelining, one engine per device).
 */
int flowgnn_entry_set_devices(int n_devices, const int* device_ids);
int flowgnn_entry_set_pipeline(int chunks_per_engine);
int flowgnn_entry_set_option(int model, const char* key, double value);

/* =====================================================================
 * (2) Handle API
 * ===================================================================== */
typedef struct flowgnn_engine flowgnn_engine;

/* Create an engine for `model` on HIP device `device_id`. */
int flowgnn_create(int model, int device_id, flowgnn_engine** out);
int flowgnn_destroy(flowgnn_engine* e);
/* Last HIP / IO error text for this engine (static storage, never NULL). */
const char* flowgnn_last_error(const flowgnn_engine* e);

/*
 * Weights, one weight set, host pointers, layouts as in the entry points above
 * without the leading [S].  Replaces the device-side load_weights
 * (GIN/src/load_inputs.cc:7-85).
 */
int flowgnn_set_weights_gin(flowgnn_engine* e,
                            const float* node_embedding_weight, const float* edge_embedding_weight,
                            const float* node_mlp_1_weights, const float* node_mlp_1_bias,
                            const float* node_mlp_2_weights, const float* node_mlp_2_bias,
                            const float* graph_pred_weights, const float* graph_pred_bias);

/*
 * Generic form: `count` host tensors of ONE weight set, in the argument order of the model's
 * <M>_compute_graphs entry point (GIN 8, GCN 11, GAT 6, PNA 10, DGN 9 tensors).
 */
int flowgnn_set_weights(flowgnn_engine* e, int count, const float* const* tensors);

/*
 * Read the reference's raw little-endian float32 .bin weight files from `dir`
 * (file names and offsets of <M>/src/host_load.cc; GIN: host_load.cc:24-58).
 */
int flowgnn_load_weights_dir(flowgnn_engine* e, const char* dir);

/*
 * Upload one concatenated batch (host pointers; copied to HBM, synchronous).
 * edge_attr may be NULL for models without edge features (GAT/PNA/DGN);
 * node_eigen ([N_tot][4] float, DGN/src/dcl.h:67) may be NULL except for DGN.
 * Replaces the host's flat-vector assembly + buffer migration
 * (GIN/src/host.cc:119-182).
 */
int flowgnn_set_batch(flowgnn_engine* e, int num_graphs,
                      const int* nums_of_nodes, const int* nums_of_edges,
                      const int* node_feature, const int* edge_list, const int* edge_attr,
                      const float* node_eigen);
/*
 * Declare the next flowgnn_set_batch batches to be SHARDS of a job of this many nodes and edges (a multi-process caller that cuts
 * one job over several GPUs, one engine each; flowgnn_group_* does it by itself).  The one choice between kernels that depends on the
 * batch size -- DGN's aggregation, by the density E / N (GIN's front end has had no size rule since round 4) -- is then made from
 * the job's totals, so every shard computes on the kernels a single engine would have chosen for the whole job and results do not
 * depend on the device count.  (-1, -1), the
 * default: each batch is its own job.  Totals smaller than a batch's own are raised to them.
 */
int flowgnn_set_job_totals(flowgnn_engine* e, long long job_nodes, long long job_edges);
/*
 * The other size-dependent choice: the graph-resident / fused kernels are used when the batch's graph tiles pack at least as full as
 * the model's threshold (GIN / GIN-VN: option "gin_resident_min_fill", default 50 %; GCN / GAT 50 %; PNA / DGN 40 %; the last tile does
 * not count).  flowgnn_graph_tile_fill computes that fill for any graph list under this engine's
 * model and options (host code, no device work; -1: the model has no graph tiles, 0: a graph exceeds the tile limits);
 * flowgnn_set_job_tile_fill makes the next flowgnn_set_batch batches take the side of the threshold the JOB's fill is on, whatever
 * their own graphs pack to (< 0: back to each batch's own packing).  flowgnn_group_* and the entry points hand both down by themselves.
 * (Development builds only -- make DEV=1, option "gin_pingpong": that kernel decides by the SHARD's own half-tile fill, which is not
 * handed down; the shipped library has no such path.)
 */
int flowgnn_graph_tile_fill(flowgnn_engine* e, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges, double* fill);
int flowgnn_set_job_tile_fill(flowgnn_engine* e, double fill);

/*
 * Enqueue one full forward of the resident batch on the engine's stream:
 * batched load_graph (CSR by destination) -> atom encoder -> conv layers ->
 * readout.  Asynchronous; results land in the engine's device result buffer.
 * Replaces one enqueueTask of <M>_compute_graphs (GIN/src/host.cc:203-210).
 */
int flowgnn_run(flowgnn_engine* e);
/* Wait for the engine's stream; returns the first validation / HIP error seen. */
int flowgnn_sync(flowgnn_engine* e);
/* Copy results [num_graphs] to host (synchronises the stream first). */
int flowgnn_get_results(flowgnn_engine* e, float* out_host);
/* Device pointer of the result buffer (float[num_graphs]); valid until next set_batch. */
int flowgnn_results_device(flowgnn_engine* e, void** d_out);
/*
 * Redirect results into a caller-owned DEVICE buffer of at least num_graphs floats (e.g. a
 * torch tensor's data_ptr(), so RCCL can all-gather it without a copy); NULL restores the
 * engine's own buffer.  Applies to the resident batch and is reset by flowgnn_set_batch.  Does not wait for the device: it
 * affects the runs enqueued after it (a caller alternating two buffers from step to step pays no host synchronisation).
 */
int flowgnn_set_results_buffer(flowgnn_engine* e, void* device_ptr);
/* The engine's hipStream_t as an opaque pointer (for event timing by a caller). */
int flowgnn_stream(flowgnn_engine* e, void** stream);

/*
 * Make the engine launch on a caller-owned hipStream_t (use_external != 0; `stream` may be the null stream) instead of
 * its own, e.g. the stream a collective library orders itself against, so that "forward, then all-gather the results"
 * needs no host synchronisation in between.  use_external == 0 restores the engine's own stream.  The caller keeps
 * the stream alive for as long as the engine uses it.
 */
int flowgnn_set_stream(flowgnn_engine* e, void* stream, int use_external);

/* Totals of the resident batch. */
int flowgnn_batch_info(const flowgnn_engine* e, long long* num_graphs,
                       long long* total_nodes, long long* total_edges);
/*
 * Graph tiles of the resident batch (0, 0 when the model keeps no whole graphs on chip or a graph exceeds its tile limits):
 * *batch_order = tiles cut in batch order (what the per-layer kernels and taps use), *packed = the bin-packed tile lists a
 * graph-resident kernel walks instead when the model asks for them (options <model>_binpack; 0 = not built).
 */
int flowgnn_batch_tiles(const flowgnn_engine* e, int* batch_order, int* packed);

/*
 * Number of forward passes this engine repeated on its exact-fp32 kernels because
 * an operand left the range in which the default kernels are fp32-accurate (GIN:
 * the dense update runs as three f16 MFMAs per product, accurate to 2^-20 relative
 * while 6e-5 < |activation| < 6e4 and to 6e-8 ABSOLUTE per operand below that -- no
 * flag is raised for tiny operands; the reference's own Q6.10 activations live in
 * [-32,32) on a 2^-10 grid).
 * The check happens in flowgnn_sync / flowgnn_get_results: device-side consumers
 * of flowgnn_results_device must call flowgnn_sync first.  -1 for a null handle.
 */
int flowgnn_exact_reruns(const flowgnn_engine* e);

/*
 * Launch-sequence replay (opt-in: option hipgraph = 1 for resident batches of up to 2^20 nodes, 2 for any size).
 * flowgnn_run then records its launch sequence (index build + forward pass) into a hipGraph on the second run of a
 * batch and replays it afterwards; any call that changes what the kernels read or write (weights, batch, result
 * buffer, numeric mode, an exact-fp32 re-run) drops the recording, and runs with the profiler enabled are never
 * replays.  Off by default: on the measured runtime plain asynchronous launches are as fast (NOTEBOOK.md section 5).
 * Returns how many runs of this engine were replays (-1 for a null handle).
 */
long long flowgnn_graph_replays(const flowgnn_engine* e);

/*
 * NUM_TASK of the readout as a run-time dimension (a compile-time constant in the reference, GIN/src/dcl.h:25, 1 as
 * shipped; ogbg-molpcba has 128 tasks).  graph_pred_weights is then [NUM_TASK][EMB_DIM], graph_pred_bias [NUM_TASK], and
 * the results are [num_graphs][NUM_TASK] (out[g][t], as `FM_TYPE out[][NUM_TASK]`, GIN/src/dcl.h:80) -- every buffer that
 * receives results (flowgnn_get_results, flowgnn_set_results_buffer) holds num_graphs * NUM_TASK floats.  Call it before
 * setting weights and batch (both must be set again afterwards).  GIN / GIN-VN / GCN; the other models' readouts are
 * single-task MLP heads and return FLOWGNN_ERR_UNSUPPORTED for NUM_TASK != 1.  The entry points take NUM_TASK as an explicit
 * argument (GIN_compute_graphs_mt / GCN_compute_graphs_mt).
 */
int flowgnn_set_num_tasks(flowgnn_engine* e, int num_tasks);
int flowgnn_num_tasks(const flowgnn_engine* e);

/*
 * Numeric mode of the engine.  FLOWGNN_NUMERIC_F32 (default): fp32 storage and accumulation.
 * FLOWGNN_NUMERIC_Q6_10: every value is the bit pattern of the reference's own number format -- ap_fixed<16,6> for GIN,
 * GIN-VN, GCN, GAT and PNA (GIN/src/dcl.h:58-59: 10 fractional bits, truncation toward -inf, wrap on overflow), ap_fixed<16,3>
 * for DGN (DGN/src/dcl.h:54-55: 13 fractional bits) -- weights are quantised from the float tensors as the reference host
 * does, and the outputs are pattern / 2^F (exact in float).  The rules assumed for ap_fixed division and the hls:: math
 * functions are written down in oracle/ginq_oracle.c and oracle/q_oracle.c.  A fidelity mode, one to two orders of
 * magnitude slower than the default path: products are truncated one at a time, as the reference does.
 * NOT validated against Vitis: the rules are taken from the published ap_fixed / hls_math semantics and the tests prove that
 * the GPU kernels and the CPU restatement agree bit for bit with EACH OTHER; no Vitis header or C-simulation output exists in
 * this repository's build environment to pin them to the reference's real bit patterns.
 */
#define FLOWGNN_NUMERIC_F32 0
#define FLOWGNN_NUMERIC_Q6_10 1
int flowgnn_set_numeric_mode(flowgnn_engine* e, int mode);

/*
 * Run-time switches, by name (the full list with defaults: the option table in flowgnn_amd/csrc/engine.hip, or
 * flowgnn_option_count / flowgnn_option_name).  They select between kernels that compute the SAME results -- e.g.
 * "gin_resident" 0 = one launch per layer, "gin_mfma" 32 = fp32 matrix pipe instead of three f16 products, "pna_fused" 0 =
 * separate aggregation and dense kernels, "hipgraph" 1 -- and exist for A/B measurements and parity tests.  Defaults come from the
 * table; the environment is read in exactly one place, when flowgnn_create builds an engine (FLOWGNN_<KEY IN UPPER CASE>, "f32"
 * reads as 32), and flowgnn_set_option overrides it.  Call it before flowgnn_set_batch: it invalidates the resident batch.
 * Unknown key: FLOWGNN_ERR_UNSUPPORTED.  No option changes the shape of any buffer.  ONE shipped option changes results, by
 * design: "gat_reference_quirk" (1 = read the node features without the per-graph offset, as GAT/src/GAT_compute.cc:72
 * does; 0, the default = with it); every other one selects between kernels that agree to
 * fp32 rounding (bit for bit where DESIGN.md says so).  The kernels' ablation hooks -- which do give wrong results, for per-phase
 * timing -- are compiled in only with -DFLOWGNN_DEV.  "gin_pingpong" != 0 implies the three-kernel front end ("gin_tile_build" 0).
 */
int flowgnn_set_option(flowgnn_engine* e, const char* key, double value);
int flowgnn_get_option(const flowgnn_engine* e, const char* key, double* value);
int flowgnn_option_count(void);
const char* flowgnn_option_name(int i);

/*
 * Several devices behind one handle (north_star: the batch dimension partitioned across the GPUs of a node; the reference has
 * one compute unit, GIN/config_slr.cfg:1-2).  One engine + one host thread per listed device; flowgnn_group_set_batch cuts the
 * batch into contiguous graph ranges balanced by sum(N + E) (flowgnn_shard_ranges: cuts[0..parts], cuts[r] = the first graph
 * whose cumulative node + edge count reaches r / parts of the total) and flowgnn_group_get_results writes [num_graphs][NUM_TASK]
 * in job order.  A device may appear more than once in the list.  flowgnn_group_engine(g, i) exposes member i for per-engine
 * calls (profiling, taps; a flowgnn_set_batch on a member makes flowgnn_group_run / get_results refuse until the next
 * flowgnn_group_set_batch).  Every engine has a persistent host thread: a group call costs microseconds of host time.
 * What a group's results are relative to ONE engine holding the whole job: graphs are independent and every kernel sums a row's
 * in-edges in an order that depends on the row alone, so results are BIT-IDENTICAL whenever the same kernels run -- and the
 * kernels are chosen from the JOB's totals, which the group (and the entry points' ranges) hand down to every member through
 * flowgnn_set_job_totals.  That covers GIN, GIN-VN, GCN, GAT and PNA with default options at any device count.  Two exceptions:
 * DGN's matrix-pipe aggregation (default on kNN-dense jobs, option "dgn_mfma_agg") sums in tile order -- equal to 1e-5 relative
 * under a different cut, bit-identical with "dgn_mfma_agg" 0; and a multi-PROCESS caller
 * that cuts a job of LARGE graphs whose tiles are about as full as the resident kernels' fill threshold must hand the job's fill down
 * itself (flowgnn_graph_tile_fill + flowgnn_set_job_tile_fill, as the group does), or a shard can pack to the other side of the
 * threshold and run the per-layer kernels: fp32 rounding differences.  flowgnn_group_run / get_results / shards answer FLOWGNN_ERR_STATE unless the engines hold
 * the shards of a flowgnn_group_set_batch job (flowgnn_group_compute and the entry points leave each engine on its last range).
 */
typedef struct flowgnn_group flowgnn_group;
int flowgnn_shard_ranges(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges, int parts, int* cuts);
int flowgnn_create_multi(int model, int n_devices, const int* device_ids, flowgnn_group** out);
int flowgnn_group_destroy(flowgnn_group* g);
int flowgnn_group_size(const flowgnn_group* g);
flowgnn_engine* flowgnn_group_engine(flowgnn_group* g, int i);
const char* flowgnn_group_last_error(const flowgnn_group* g);
int flowgnn_group_set_weights(flowgnn_group* g, int count, const float* const* tensors);
int flowgnn_group_load_weights_dir(flowgnn_group* g, const char* dir);
int flowgnn_group_set_option(flowgnn_group* g, const char* key, double value);
int flowgnn_group_set_num_tasks(flowgnn_group* g, int num_tasks);
int flowgnn_group_set_numeric_mode(flowgnn_group* g, int mode);
int flowgnn_group_set_batch(flowgnn_group* g, int num_graphs,
                            const int* nums_of_nodes, const int* nums_of_edges,
                            const int* node_feature, const int* edge_list, const int* edge_attr,
                            const float* node_eigen);
int flowgnn_group_shards(const flowgnn_group* g, int* cuts /* [size + 1] */);
int flowgnn_group_run(flowgnn_group* g);
int flowgnn_group_sync(flowgnn_group* g);
int flowgnn_group_get_results(flowgnn_group* g, float* out_host);
/* set_batch + run + get_results for a batch in HOST memory, cut into size x chunks_per_engine ranges; engine i takes ranges
 * i, i + size, ... in turn, so that one engine's copies overlap the others' kernels.  out_host: [num_graphs][NUM_TASK]. */
int flowgnn_group_compute(flowgnn_group* g, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                          const int* node_feature, const int* edge_list, const int* edge_attr, const float* node_eigen,
                          float* out_host, int chunks_per_engine);

/*
 * Debug / parity taps (device -> host copies; synchronise first).
 *  flowgnn_get_csr: the batched destination-major CSR built by load_graph:
 *     row_ptr[N_tot+1], src[E_tot] (global source id, ascending per row, ties in
 *     input order), eid[E_tot] (input edge index), out_deg[N_tot].
 *  flowgnn_get_h: node embeddings after the last executed stage, [N_tot][dim].
 */
int flowgnn_get_csr(flowgnn_engine* e, int* row_ptr, int* src, int* eid, int* out_deg);
int flowgnn_get_h(flowgnn_engine* e, float* h_host, int* dim);

/*
 * Per-kernel profile with HIP events on the engine's stream.
 *  flowgnn_profile_enable(e, 1) brackets every kernel launch of subsequent runs
 *  with events; flowgnn_profile_read fills, for kernel slot k < *count,
 *  total milliseconds and launch counts since enable, and the kernel names.
 */
#define FLOWGNN_MAX_PROFILE_SLOTS 32
int flowgnn_profile_enable(flowgnn_engine* e, int on);
int flowgnn_profile_read(flowgnn_engine* e, int* count, const char** names,
                         double* total_ms, long long* launches);

/*
 * Standalone aggregation kernel of layer `layer` on the resident batch (the
 * message-passing unit alone, m written to HBM): used to measure the HBM roofline
 * of the gather + segmented-sum path.  Requires a prior flowgnn_run.
 */
int flowgnn_run_aggregation_only(flowgnn_engine* e, int layer, int iters, float* avg_ms);
/*
 * Parity tap for that kernel: runs it once, on the node embeddings the last flowgnn_run left in the engine (the input
 * of the model's last stage when the readout was folded into it, the last layer's output otherwise), and copies to the
 * host the rows it read (h_in_host, [N_tot][*in_dim]) and what it wrote (agg_host, [N_tot][*agg_dim]; GIN: m + h,
 * GCN: relu(BN(...)), PNA: [mean|min|max|std] x 80, DGN: [mean | directional] x 100).  Either buffer may be NULL
 * (dims are still reported).  The next flowgnn_run rewrites everything this touches.
 */
int flowgnn_get_aggregate(flowgnn_engine* e, int layer, float* h_in_host, int* in_dim, float* agg_host, int* agg_dim);

#ifdef __cplusplus
}
#endif
#endif
