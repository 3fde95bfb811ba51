// common.h -- shared declarations for the MI355X (gfx950) FlowGNN engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace fg {

// ---- model constants (GIN/src/dcl.h:16-26 of the reference) ----
constexpr int ND_FEATURE = 9;
constexpr int ND_FEATURE_TOTAL = 173;
constexpr int EDGE_ATTR = 3;
constexpr int ED_FEATURE_PER_LAYER = 13;
constexpr int EDGE_COMBOS = 60;  // 5 * 6 * 2 distinct (attr0, attr1, attr2) triples
constexpr int WAVE = 64;

// error flag values written by validation code on the device (match flowgnn.h)
constexpr int ERR_EDGE_RANGE = 2;
constexpr int ERR_EDGE_ATTR = 3;
constexpr int ERR_NODE_FEAT = 4;
constexpr int ERR_UNSUPPORTED = 8;
// The per-row order of the CSR comes from a rank sort that costs indeg^2 per row (graph_build.hip).  Inside the per-graph LDS
// classes (<= 16 384 edges per graph) that is bounded by the class; on the flat path (graphs beyond every class, i.e. far beyond
// the reference's own 500-node / 5 500-edge caps, GIN/src/dcl.h:17-18) a row may have at most this many in-edges -- 2.7e8 key
// comparisons -- before the build refuses the batch with FLOWGNN_ERR_UNSUPPORTED instead of running for minutes.
constexpr int MAX_FLAT_INDEGREE = 16384;

__host__ __device__ inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------
// Batched graph bookkeeping in HBM (all arrays int32 unless noted)
// ---------------------------------------------------------------------------
struct BatchView {
    int num_graphs;
    int n_tot;               // total nodes
    int e_tot;               // total edges
    const int* nums_of_nodes;  // [G]
    const int* nums_of_edges;  // [G]
    const int* node_off;       // [G+1]
    const int* edge_off;       // [G+1]
    const int* node_feature;   // [N][9]
    const int* edge_list;      // [E][2] local ids
    const int* edge_attr;      // [E][3] or null
};

struct CsrView {
    int* row_ptr;    // [N+1] in-edges of destination v: [row_ptr[v], row_ptr[v+1])
    int* src;        // [E] global source id, ascending within a row, ties in input order
    int* eid;        // [E] input edge index
    uint8_t* ecode;  // [E] (attr0*6 + attr1)*2 + attr2, or null
    int* out_deg;    // [N] out-degree of each node (reference degree_table)
    // scratch
    int* gsrc;       // [E] global source of input edge e
    int* gdst;       // [E] global destination of input edge e
    int* cursor;     // [N]
    int* tmp;        // [E]
    int* block_sums; // scan scratch
    int* err;        // [1] device error flag
};

// graph_build.hip
// max_nodes / max_edges: largest graph of the batch (selects the per-graph LDS build or the flat global build)
void launch_build_csr(const BatchView& b, const CsrView& c, bool has_edge_attr, int max_nodes, int max_edges, hipStream_t s, const int* only_if = nullptr);

}  // namespace fg

// ---------------------------------------------------------------------------
// Engine-side plumbing shared by the per-model translation units
// ---------------------------------------------------------------------------
#include <string>
#include <vector>

namespace fg {

#define FG_HIP_TRY(expr)                                                            \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            fg::set_hip_error(#expr, _e, __FILE__, __LINE__);                       \
            return 5; /* FLOWGNN_ERR_HIP */                                         \
        }                                                                           \
    } while (0)

void set_hip_error(const char* what, hipError_t e, const char* file, int line);
const char* last_error_text();

// HIP-event profiler: brackets launches on the engine stream; read after a sync.
class Profiler {
public:
    bool enabled = false;
    int slot(const char* name);
    void begin(int slot, hipStream_t s);
    void end(int slot, hipStream_t s);
    void collect();  // after stream sync: fold finished event pairs into totals
    void reset();
    std::vector<std::string> names;
    std::vector<double> total_ms;
    std::vector<long long> launches;
    ~Profiler();
private:
    struct Pending { int slot; hipEvent_t a, b; };
    std::vector<Pending> pending_;
    std::vector<hipEvent_t> free_;
    hipEvent_t get_event();
};

struct ProfScope {
    Profiler& p; int slot; hipStream_t s;
    ProfScope(Profiler& p_, const char* name, hipStream_t s_) : p(p_), slot(-1), s(s_) {
        if (p.enabled) { slot = p.slot(name); p.begin(slot, s); }
    }
    ~ProfScope() { if (slot >= 0) p.end(slot, s); }
};

// Whole graphs packed greedily (in batch order) into tiles of at most `rows` rows and `edges` in-edges: what a kernel that
// keeps a tile of graphs on chip across all layers walks (the MI355X counterpart of the FPGA holding ONE graph in BRAM
// across its layer loop, GIN/src/GIN_compute.cc:72-94).  Built on the host in flowgnn_set_batch from the per-graph counts
// when the model asks for it (Model::graph_tile_limits).
struct GraphTiles {
    const int* row_start = nullptr;    // device [n_tiles + 1]: first row of tile t (row_start[n_tiles] = n_tot)
    const int* graph_start = nullptr;  // device [n_tiles + 1]: first graph of tile t
    int n_tiles = 0;
    int rows = 0, edges = 0;           // the limits it was built for
    bool ok = false;                   // every graph of the batch fits a tile
    double fill = 0.0;                 // n_tot / (n_tiles * rows)
    // Second packing for kernels that run two half-tiles per CU out of phase (Model::sub_tile_limits): runs of consecutive
    // graphs of at most sub_rows rows / sub_edges in-edges, {first row, rows, first graph, one past the last graph} each; a graph
    // beyond those limits (but within rows / edges above) ends the run and is listed in big_* instead -- pairs
    // (start, end) in the row_start / graph_start format read with stride 2, one graph per tile.
    const int* sub = nullptr;          // device [n_sub][4]
    int n_sub = 0, sub_rows = 0, sub_edges = 0;
    const int* big_row = nullptr;      // device [2 n_big]
    const int* big_graph = nullptr;    // device [2 n_big]
    int n_big = 0;
    bool sub_ok = false;
    double sub_fill = 0.0;             // rows of the sub-tiles' graphs / (n_sub * sub_rows)
    // Third packing (Model::wants_packed_tile_lists): the same graphs BIN-PACKED into tiles (best fit, largest first, inside windows
    // of 1 024 consecutive graphs) instead of cut in batch order -- tiles of graphs that are a large part of a tile (49-node graphs
    // in 128-row tiles) go from 80 % to 95 % full, and a resident kernel's time is per TILE.  A tile is then a LIST of graphs, not a
    // range: bp_list holds graph ids in tile order, bp_graph[t] .. bp_graph[t + 1] index it, bp_row[t] is the tile's first row in the
    // tile-ordered row space its own tile-build kernel writes (what the resident kernel reads is all tile-ordered).
    const int* bp_list = nullptr;      // device [num_graphs]
    const int* bp_lrow = nullptr;      // device [num_graphs]: the first row, inside its tile, of the graph at this list position
    const int* bp_graph = nullptr;     // device [bp_tiles + 1]
    const int* bp_row = nullptr;       // device [bp_tiles + 1]
    int bp_tiles = 0;                  // 0: not built
};

// Everything a model's forward needs about the resident batch.
struct DeviceBatch {
    BatchView b;
    long long job_n, job_e;   // nodes / edges of the JOB this batch is a shard of (= b.n_tot / b.e_tot for a batch that is the whole job;
                              // flowgnn_set_job_totals).  Kernel CHOICES by batch size read these, so that a job computes on the same
                              // kernels whether one engine holds all of it or a group of engines a shard each
    CsrView csr;
    const float* node_eigen;  // [N][4] or null
    float* h[2];              // ping/pong node embeddings [N][dim]
    float* scratch;           // [N][scratch_dim] model scratch (aggregates)
    float* out;               // [G][num_tasks]
    int num_tasks;            // NUM_TASK of the readout (1 unless flowgnn_set_num_tasks said otherwise)
    int final_h;              // which h[] holds the last stage's output (set by forward)
    const float* tap;         // optional debug tap returned by flowgnn_get_h instead of h[final_h]
    int tap_dim;
    bool h_valid;             // h[final_h] holds the last stage's node embeddings (false: the model folded the readout into its last
                              // layer and never wrote them; flowgnn_get_h then repeats the pass with Model::set_keep_h(true))
    bool csr_built;           // csr.* describe this batch (set by the engine when the index build has run; kernels that work from the
                              // caller's arrays directly -- Model::needs_csr() == false -- leave it as it is)
    int max_nodes, max_edges; // largest graph of the batch (what launch_build_csr selects its per-graph class by)
    GraphTiles gtiles;        // graph-aligned tiles (GraphTiles above), n_tiles == 0 if the model did not ask for them
    int* range_flag;          // [1] set by a reduced-range kernel whose operands left its accurate range (see Model::set_exact)
};

// Run-time switches of an engine (flowgnn_set_option / flowgnn_get_option).  Keys and defaults: the table in engine.hip.
// The environment is consulted in exactly ONE place -- Options::Options(), i.e. once per flowgnn_create -- as
// FLOWGNN_<KEY IN UPPER CASE> (e.g. gin_resident <- FLOWGNN_GIN_RESIDENT; "f32" reads as 32); flowgnn_set_option overrides.
// Switches that can change RESULTS (the ablation hooks of the resident / fused kernels) exist only in a -DFLOWGNN_DEV build.
class Options {
public:
    Options();
    bool set(const char* key, double v);         // false: unknown key
    bool get(const char* key, double* v) const;  // false: unknown key
    double num(const char* key) const;           // known keys only (0 for an unknown one)
    int i(const char* key) const { return (int)num(key); }
    bool on(const char* key) const { return num(key) != 0.0; }
private:
    std::vector<double> v_;
};

// Development-only ablation bits of the hot kernels: compiled out of the shipped library (the kernels see a constant 0, so the
// branches and the wrong-result modes behind them do not exist in it); `make DEV=1` builds them in.
#ifdef FLOWGNN_DEV
#define FG_ABLATE(x) (x)
#else
#define FG_ABLATE(x) 0
#endif

class Model {
public:
    virtual ~Model() {}
    // pull this model's switches out of the engine's options (called at create and after every flowgnn_set_option)
    virtual void configure(const Options&) {}
    // A model whose default kernels are fp32-accurate only inside an operand range (GIN: split-f16 MFMA) raises
    // DeviceBatch::range_flag when an input leaves it; the engine then calls set_exact(true) and repeats forward().
    virtual void set_exact(bool) {}
    // debug taps: make forward() materialise the last layer's node embeddings even if that costs a round trip
    virtual void set_keep_h(bool) {}
    // 0 = fp32 (default); 1 = the reference's ap_fixed<16,6> bit patterns (GIN / GIN-VN only: ginq.hip)
    virtual int set_numeric_mode(int mode) { return mode == 0 ? 0 : 8 /* FLOWGNN_ERR_UNSUPPORTED */; }
    // NUM_TASK of the readout (graph_pred_weights [NUM_TASK][EMB_DIM], out [G][NUM_TASK]); takes effect at the next set_weights
    virtual int set_num_tasks(int t) { return t == 1 ? 0 : 8; }
    // rows > 0: flowgnn_set_batch packs whole graphs into tiles of at most rows rows / edges in-edges (DeviceBatch::gtiles)
    virtual void graph_tile_limits(int& rows, int& edges) const { rows = 0; edges = 0; }
    // rows > 0: flowgnn_set_batch also packs the half-tile lists of GraphTiles (sub / big_*)
    virtual void sub_tile_limits(int& rows, int& edges) const { rows = 0; edges = 0; }
    // true: flowgnn_set_batch also bin-packs the graphs into tile LISTS (GraphTiles::bp_*)
    virtual bool wants_packed_tile_lists() const { return false; }
    virtual int emb_dim() const = 0;
    virtual int scratch_dim() const = 0;          // floats per node of scratch the forward needs
    virtual bool has_edge_attr() const = 0;
    virtual int num_weight_tensors() const = 0;    // host tensors expected by set_weights
    virtual int set_weights(const float* const* host_tensors) = 0;
    virtual int load_weights_dir(const char* dir) = 0;
    virtual bool weights_ready() const = 0;
    // false: the next forward() of this batch (under the current exact / keep_h / numeric state) reads the caller's arrays itself and
    // needs no destination-major CSR in HBM; the engine then skips the index build for that run (taps build it on demand)
    virtual bool needs_csr(const DeviceBatch&) const { return true; }
    virtual int forward(DeviceBatch& db, Profiler& prof, hipStream_t s) = 0;
    virtual int aggregation_only(DeviceBatch& db, int layer, hipStream_t s) { (void)db; (void)layer; (void)s; return 8; }
    // floats per node that aggregation_only() writes to db.scratch (0: the model has no standalone aggregation kernel)
    virtual int aggregate_dim() const { return 0; }
};

Model* make_gin_model(bool virtual_node);
Model* make_gcn_model();
Model* make_pna_model();
Model* make_dgn_model();
Model* make_gat_model();

// helpers
int read_floats(const char* dir, const char* file, size_t offset_floats, size_t count, float* dst);
template <typename T>
int upload(T** dptr, const std::vector<T>& host) {
    const size_t want = sizeof(T) * (host.empty() ? 1 : host.size());
    if (*dptr) {  // keep the buffer when its size is unchanged: a weight reload must not churn the allocator
        hipDeviceptr_t base = nullptr;
        size_t have = 0;
        if (hipMemGetAddressRange(&base, &have, (hipDeviceptr_t)*dptr) != hipSuccess || base != (hipDeviceptr_t)*dptr || have != want) {
            (void)hipGetLastError();
            (void)hipFree(*dptr);
            *dptr = nullptr;
        }
    }
    if (!*dptr) FG_HIP_TRY(hipMalloc((void**)dptr, want));
    if (!host.empty()) FG_HIP_TRY(hipMemcpy(*dptr, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace fg
