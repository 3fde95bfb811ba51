// engine.hip -- host-side engine and the C ABI declared in include/flowgnn.h.
//
// One engine = one GPU + one HIP stream + one model's weights + one resident batch.
// The reference host (GIN/src/host.cc) programs an FPGA, migrates flat buffers once and
// enqueues the kernel NUM_TRIALS times; the counterpart here is
//   flowgnn_create -> flowgnn_set_weights_* / flowgnn_load_weights_dir -> flowgnn_set_batch
//   -> N x flowgnn_run -> flowgnn_get_results.
#include "common.h"
#include "../../include/flowgnn.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <mutex>
#include <thread>
#include <atomic>
#include <limits>
#include <condition_variable>
#include <functional>
#include <memory>

namespace fg {

// ------------------------------------------------------------------ error text
static thread_local char g_err[512] = "";
void set_hip_error(const char* what, hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
}
const char* last_error_text() { return g_err; }
void set_last_error(const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); }

int read_floats(const char* dir, const char* file, size_t offset_floats, size_t count, float* dst) {
    std::string path = std::string(dir) + "/" + file;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        snprintf(g_err, sizeof(g_err), "cannot open %s", path.c_str());
        return FLOWGNN_ERR_IO;
    }
    int rc = 0;
    if (fseek(f, (long)(offset_floats * sizeof(float)), SEEK_SET) != 0 || fread(dst, sizeof(float), count, f) != count) {
        snprintf(g_err, sizeof(g_err), "short read of %zu floats at offset %zu from %s", count, offset_floats, path.c_str());
        rc = FLOWGNN_ERR_IO;
    }
    fclose(f);
    return rc;
}

// ------------------------------------------------------------------ profiler
Profiler::~Profiler() {
    for (auto& p : pending_) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : free_) (void)hipEventDestroy(e);
}
int Profiler::slot(const char* name) {
    for (size_t i = 0; i < names.size(); i++)
        if (names[i] == name) return (int)i;
    names.push_back(name);
    total_ms.push_back(0.0);
    launches.push_back(0);
    return (int)names.size() - 1;
}
hipEvent_t Profiler::get_event() {
    if (!free_.empty()) { hipEvent_t e = free_.back(); free_.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);  // a failure shows up as an invalid-handle error at the first record, reported by flowgnn_run
    return e;
}
void Profiler::begin(int slot, hipStream_t s) {
    Pending p{slot, get_event(), get_event()};
    (void)hipEventRecord(p.a, s);
    pending_.push_back(p);
}
void Profiler::end(int slot, hipStream_t s) {
    for (size_t i = pending_.size(); i-- > 0;)
        if (pending_[i].slot == slot) { (void)hipEventRecord(pending_[i].b, s); break; }
}
void Profiler::collect() {
    for (auto& p : pending_) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            total_ms[p.slot] += ms;
            launches[p.slot] += 1;
        }
        free_.push_back(p.a);
        free_.push_back(p.b);
    }
    pending_.clear();
}
void Profiler::reset() {
    collect();
    for (auto& x : total_ms) x = 0.0;
    for (auto& x : launches) x = 0;
}

// ------------------------------------------------------------------ options
// Every run-time switch of the library, with its default.  Keys ending in _ablate exist only in -DFLOWGNN_DEV builds.
struct OptionDef { const char* key; double dflt; };
static const OptionDef kOptionTable[] = {
    {"hipgraph", 0},              // 1: replay the launch sequence of resident batches of up to 2^20 nodes; 2: of any size
    {"csr_flat", 0},              // 1: global-memory index build for every graph (A/B)
    {"tile_nominal", -1},         // rows per tile of the tiled aggregation kernels (< 0: the model's default)
    {"tile_slack", -1},
    {"h2d_pack", 16},             // flowgnn_set_batch: host threads that narrow a large batch's int32 arrays for the transfer (h2d_pack.cpp); 0: plain copies
    {"tile_balance", 1},          // graph tiles of a batch of <= 8 rounds over the CUs: fewer rows per tile, whole rounds of tiles (flowgnn_set_batch)
    {"gin_resident", 1}, {"gin_binpack", 1}, {"gin_tile_build", -1}, {"gin_resident_min_fill", 0.5}, {"gin_resident_nosort", 0}, {"gin_resident_prof", 0},
    {"gin_unfused", 0}, {"gin_mfma", 16}, {"gin_split_nt", 4}, {"gin_fold_readout", 1}, {"gin_head_fold", 1},
    {"gin_agg_untiled", 0}, {"gin_agg_tile", 128},
    {"gcn_resident", 1}, {"gcn_binpack", 1}, {"gcn_tile_build", 1}, {"gcn_unfused", 0}, {"gcn_mfma", 16},
    {"gat_resident", 1}, {"gat_mfma", 16}, {"gat_fold_readout", 1}, {"gat_reference_quirk", 0},
    {"pna_resident", 1}, {"pna_binpack", 1}, {"pna_tile_build", 1}, {"pna_fused", 1}, {"pna_mfma", 16},
    {"pna_mfma_agg", 0},          // deprecated (removed in round 5, the kernel it selected is gone): accepted and ignored, so that callers' scripts keep working
    {"dgn_fused", 1}, {"dgn_mfma", 16}, {"dgn_mfma_agg", -1}, {"dgn_fold_readout", 1}, {"dgn_rowinfo_direct", 1}, {"dgn_resident", 1}, {"dgn_binpack", 1},
#ifdef FLOWGNN_DEV
    {"gcn_ablate", 0}, {"gat_ablate", 0}, {"pna_ablate", 0}, {"dgn_ablate", 0}, {"gin_pingpong", 0},
#endif
};
constexpr int kNumOptions = (int)(sizeof(kOptionTable) / sizeof(kOptionTable[0]));

static int option_index(const char* key) {
    if (!key) return -1;
    for (int i = 0; i < kNumOptions; i++)
        if (strcmp(kOptionTable[i].key, key) == 0) return i;
    return -1;
}

// THE place where the library reads its environment: FLOWGNN_<KEY> seeds option <key> of every engine created afterwards
// ("f32" reads as 32, for the *_mfma switches); FLOWGNN_DEVICES / FLOWGNN_DEVICE seed the device list of the
// <M>_compute_graphs entry points (entry_devices below).
static double env_number(const char* text) {
    if (strcmp(text, "f32") == 0) return 32.0;
    if (strcmp(text, "f16") == 0) return 16.0;
    return atof(text);
}
// stale_num_task: FLOWGNN_NUM_TASK (round 2's way to give the entry points NUM_TASK) is set to something other than 1.  It is no
// longer read -- NUM_TASK is an argument of the *_compute_graphs_mt symbols -- and a caller that still relies on it would get
// one task's worth of results for [T][100] weights, silently: the plain GIN / GCN entry points refuse to run instead.
static void read_environment(std::vector<double>* option_values, std::vector<int>* devices, bool* stale_num_task = nullptr) {
    if (stale_num_task) {
        const char* v = getenv("FLOWGNN_NUM_TASK");
        *stale_num_task = v && *v && atoi(v) != 1;
    }
    if (option_values) {
        option_values->resize(kNumOptions);
        for (int i = 0; i < kNumOptions; i++) {
            (*option_values)[i] = kOptionTable[i].dflt;
            std::string name = "FLOWGNN_";
            for (const char* c = kOptionTable[i].key; *c; c++) name += (char)toupper((unsigned char)*c);
            const char* v = getenv(name.c_str());
            if (v && *v) (*option_values)[i] = env_number(v);
        }
    }
    if (devices) {
        devices->clear();
        const char* v = getenv("FLOWGNN_DEVICES");
        if (!v || !*v) v = getenv("FLOWGNN_DEVICE");
        if (v && *v) {
            for (const char* c = v; *c;) {
                devices->push_back(atoi(c));
                while (*c && *c != ',') c++;
                if (*c == ',') c++;
            }
        }
        if (devices->empty()) devices->push_back(0);
    }
}

Options::Options() { read_environment(&v_, nullptr); }
bool Options::set(const char* key, double v) {
    const int i = option_index(key);
    if (i < 0) return false;
    v_[i] = v;
    return true;
}
bool Options::get(const char* key, double* v) const {
    const int i = option_index(key);
    if (i < 0) return false;
    if (v) *v = v_[i];
    return true;
}
double Options::num(const char* key) const {
    const int i = option_index(key);
    return i < 0 ? 0.0 : v_[i];
}

}  // namespace fg

using namespace fg;

namespace fg {
// h2d_pack.cpp
size_t h2d_pack_bytes(size_t n_nodes, size_t n_edges, bool attr, size_t* off_edges, size_t* off_attr);
void h2d_pack(const int* node_feature, const int* edge_list, const int* edge_attr, size_t n_nodes, size_t n_edges, uint8_t* dst, int threads);
void host_parallel_for(int parts, const std::function<void(int)>& fn);

// the packed arrays back into the reference's int32 layout (what every kernel reads): 255 / 65 535 = "did not fit" -> -1, which the
// validation on the device refuses as it would have refused the original value
__global__ __launch_bounds__(256) void unpack_batch_kernel(const uint8_t* __restrict__ nf8, const uint16_t* __restrict__ el16, const uint8_t* __restrict__ ea8,
                                                           int* __restrict__ nf, int* __restrict__ el, int* __restrict__ ea, long long n9, long long e2,
                                                           long long ne) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n9; i += stride) {  // four values per thread: n9 and e2 are padded to 16 B
        const uint32_t w = *reinterpret_cast<const uint32_t*>(nf8 + i);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + k < n9) { const int v = (int)((w >> (8 * k)) & 0xFFu); nf[i + k] = v == 255 ? -1 : v; }
    }
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < e2; i += stride) {
        const uint2 w = *reinterpret_cast<const uint2*>(el16 + i);
        const int v[4] = {(int)(w.x & 0xFFFFu), (int)(w.x >> 16), (int)(w.y & 0xFFFFu), (int)(w.y >> 16)};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + k < e2) el[i + k] = v[k] == 65535 ? -1 : v[k];
    }
    if (ea8)
        for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < ne; i += stride) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(ea8 + i);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + k < ne) {
                    const int c = (int)((w >> (8 * k)) & 0xFFu);
                    int* o = ea + 3 * (i + k);
                    if (c == 255) { o[0] = -1; o[1] = 0; o[2] = 0; }
                    else { o[0] = c / 12; o[1] = (c >> 1) % 6; o[2] = c & 1; }
                }
        }
}
static void launch_unpack_batch(const uint8_t* nf8, const uint8_t* el16, const uint8_t* ea8, int* nf, int* el, int* ea, long long n, long long e, hipStream_t s) {
    const long long work = n * 9 > e * 2 ? n * 9 : e * 2;
    long long blocks = (work / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    unpack_batch_kernel<<<(int)blocks, 256, 0, s>>>(nf8, reinterpret_cast<const uint16_t*>(el16), ea8, nf, el, ea, n * 9, e * 2, e);
}

// host threads a call may use: the option's count, capped by what this process may run on (affinity mask, cgroup CPU quota)
static int host_threads(int want) {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32];
        long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long cpus = (atoll(q) + period - 1) / period;
            if (cpus >= 1 && cpus < n) n = (int)cpus;
        }
        fclose(f);
    }
    return want < n ? (want < 1 ? 1 : want) : n;
}
}  // namespace fg

// ------------------------------------------------------------------ engine object
struct flowgnn_engine {
    int model_id = 0;
    int device = 0;
    hipStream_t stream = nullptr;      // the stream every launch goes to
    hipStream_t own_stream = nullptr;  // the engine's own stream (stream == own_stream unless flowgnn_set_stream redirected it)
    hipStream_t copy_stream = nullptr; // flowgnn_set_batch's host -> device copies: the engine's own queue, not the process's null stream
    Model* model = nullptr;
    Options opts;   // defaults <- environment (read once, here) <- flowgnn_set_option
    Profiler prof;
    std::string err;

    // resident batch
    bool batch_ready = false;
    bool ran = false;
    int num_tasks = 1;          // NUM_TASK of the readout: results are [G][num_tasks]
    bool force_exact = false;   // the resident batch tripped the range flag once: run it on the exact kernels
    int exact_reruns = 0;
    long long G = 0, N = 0, E = 0;
    double job_fill = -1.0;            // flowgnn_set_job_tile_fill: the graph-tile fill of the JOB (-1: the batch's own packing decides)
    long long job_n = -1, job_e = -1;  // flowgnn_set_job_totals: the job the next batches are shards of (-1: each batch is its own job)
    int max_nodes = 0, max_edges = 0;
    size_t capG = 0, capN = 0, capE = 0;
    int *d_nn = nullptr, *d_ne = nullptr, *d_noff = nullptr, *d_eoff = nullptr;
    int *d_nf = nullptr, *d_el = nullptr, *d_ea = nullptr;
    float* d_eig = nullptr;
    int *d_rowptr = nullptr, *d_src = nullptr, *d_eid = nullptr, *d_outdeg = nullptr, *d_gsrc = nullptr, *d_gdst = nullptr,
        *d_cursor = nullptr, *d_tmp = nullptr, *d_bsums = nullptr, *d_err = nullptr;
    uint8_t* d_ecode = nullptr;
    float *d_h0 = nullptr, *d_h1 = nullptr, *d_scratch = nullptr, *d_out = nullptr;
    int *d_trow = nullptr, *d_tgraph = nullptr;  // graph-aligned tiles (GraphTiles)
    size_t cap_tiles = 0;
    int* d_sub = nullptr;                        // GraphTiles::sub | big_row | big_graph in one allocation
    size_t cap_sub = 0;
    int* d_bp = nullptr;                         // GraphTiles::bp_list | bp_graph | bp_row in one allocation
    size_t cap_bp = 0;
    uint8_t *h_pack = nullptr, *d_pack = nullptr;  // packed host -> device transfer (h2d_pack.cpp): pinned staging + its device copy
    size_t cap_pack = 0;
    bool has_attr = false, has_eig = false;
    DeviceBatch db{};

    // hipGraph replay of the launch sequence (index build + forward), opt-in (FLOWGNN_HIPGRAPH=1; 2 = batches of any size).
    // Measured on this runtime it does not pay: asynchronous launches already pipeline, and a replay of the dozen kernels
    // of a step is 1-4 % SLOWER than launching them (4 113 molhiv graphs: 0.267 ms plain, 0.271 ms replayed; 512 graphs:
    // 0.099 vs 0.103 ms) -- so it is off by default and kept for hosts whose launch path is the bottleneck.
    // The first run of a batch is plain (models size their scratch buffers there), the second is captured, later ones
    // replay.  Every call that changes what the captured kernels would read or write drops the recording.
    hipGraphExec_t gexec = nullptr;
    bool graph_ok = false;
    bool graph_h_valid = true;   // what the captured forward left in db.h_valid / tap / tap_dim / final_h (host-side outputs)
    const float* graph_tap = nullptr;
    int graph_tap_dim = 0, graph_final_h = 0;
    int plain_runs = 0;
    int graph_mode = 0;  // option hipgraph
    long long graph_replays = 0;
    void drop_graph() {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        gexec = nullptr;
        graph_ok = false;
        plain_runs = 0;
    }

    void free_batch() {
        void* ptrs[] = {d_nn /* base of d_ne, d_noff, d_eoff too */, d_nf, d_el, d_ea, d_eig, d_rowptr, d_src, d_eid, d_outdeg, d_gsrc,
                        d_gdst, d_cursor, d_tmp, d_bsums, d_ecode, d_h0, d_h1, d_scratch, d_out};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        d_nn = d_ne = d_noff = d_eoff = d_nf = d_el = d_ea = nullptr;
        d_eig = nullptr;
        d_rowptr = d_src = d_eid = d_outdeg = d_gsrc = d_gdst = d_cursor = d_tmp = d_bsums = nullptr;
        d_ecode = nullptr;
        d_h0 = d_h1 = d_scratch = d_out = nullptr;
        if (d_trow) (void)hipFree(d_trow);  // (base of d_tgraph too)
        d_trow = d_tgraph = nullptr;
        cap_tiles = 0;
        if (d_sub) (void)hipFree(d_sub);
        d_sub = nullptr;
        cap_sub = 0;
        if (d_bp) (void)hipFree(d_bp);
        d_bp = nullptr;
        cap_bp = 0;
        if (h_pack) (void)hipHostFree(h_pack);
        if (d_pack) (void)hipFree(d_pack);
        h_pack = d_pack = nullptr;
        cap_pack = 0;
        capG = capN = capE = 0;
    }
};

#define ENGINE_TRY(e, expr)                                     \
    do {                                                        \
        int _rc = (expr);                                       \
        if (_rc) { (e)->err = fg::last_error_text(); return _rc; } \
    } while (0)

// HIP call made in an engine context: the failure text goes to the thread-local slot AND to the engine, so
// flowgnn_last_error(e) always reports the latest failure (never a stale earlier one)
#define EHIP_TRY(e, expr)                                                   \
    do {                                                                    \
        hipError_t _he = (expr);                                            \
        if (_he != hipSuccess) {                                            \
            fg::set_hip_error(#expr, _he, __FILE__, __LINE__);              \
            (e)->err = fg::last_error_text();                               \
            return FLOWGNN_ERR_HIP;                                         \
        }                                                                   \
    } while (0)

static int use_device(flowgnn_engine* e) {
    FG_HIP_TRY(hipSetDevice(e->device));
    return 0;
}

// index build (when the model's next forward needs the CSR) + forward pass, on the engine's stream; the state the model decides
// on (exact / keep_h / numeric mode) must be set before
static int engine_forward(flowgnn_engine* e) {
    if (e->model->needs_csr(e->db)) {
        ProfScope p(e->prof, "build_csr", e->stream);
        const bool flat = e->opts.on("csr_flat");  // A/B: force the global path
        launch_build_csr(e->db.b, e->db.csr, e->has_attr, flat ? (1 << 30) : e->max_nodes, flat ? (1 << 30) : e->max_edges, e->stream);
        e->db.csr_built = true;
    }
    return e->model->forward(e->db, e->prof, e->stream);
}
// taps that read the CSR (flowgnn_get_csr, the stand-alone aggregation kernels): build it if no run of this batch has
static void ensure_csr(flowgnn_engine* e) {
    if (e->db.csr_built || e->G == 0) return;
    launch_build_csr(e->db.b, e->db.csr, e->has_attr, e->max_nodes, e->max_edges, e->stream);
    e->db.csr_built = true;
}

extern "C" {

int flowgnn_create(int model, int device_id, flowgnn_engine** out) {
    if (!out) return FLOWGNN_ERR_ARG;
    *out = nullptr;
    Model* m = nullptr;
    switch (model) {
        case FLOWGNN_MODEL_GIN: m = make_gin_model(false); break;
        case FLOWGNN_MODEL_GIN_VN: m = make_gin_model(true); break;
        case FLOWGNN_MODEL_GCN: m = make_gcn_model(); break;
        case FLOWGNN_MODEL_PNA: m = make_pna_model(); break;
        case FLOWGNN_MODEL_DGN: m = make_dgn_model(); break;
        case FLOWGNN_MODEL_GAT: m = make_gat_model(); break;
        default: return FLOWGNN_ERR_UNSUPPORTED;
    }
    flowgnn_engine* e = new flowgnn_engine();
    e->model_id = model;
    e->device = device_id;
    e->model = m;
    e->graph_mode = e->opts.i("hipgraph");
    m->configure(e->opts);
    int rc = use_device(e);
    if (!rc) {
        hipError_t he = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
        e->stream = e->own_stream;
        // Copies on a stream of their own, at the highest priority (priorities have their own hardware queues): through the null stream
        // a copy can share a queue with ANOTHER engine's kernels -- which queue a stream lands on depends on how many streams the
        // process has had -- and then waits for them: the entry points' copy / kernel pipeline ran at 31 ms instead of 13 in such runs.
        if (he == hipSuccess) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (hipStreamCreateWithPriority(&e->copy_stream, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); e->copy_stream = nullptr; }
        }
        if (he == hipSuccess) he = hipMalloc((void**)&e->d_err, 2 * sizeof(int));  // [0] validation, [1] range flag
        if (he == hipSuccess) he = hipMemset(e->d_err, 0, 2 * sizeof(int));
        if (he != hipSuccess) {
            set_hip_error("engine init", he, __FILE__, __LINE__);
            rc = FLOWGNN_ERR_HIP;
        }
    }
    if (rc) {
        delete m;
        delete e;
        return rc;
    }
    *out = e;
    return FLOWGNN_OK;
}

int flowgnn_destroy(flowgnn_engine* e) {
    if (!e) return FLOWGNN_ERR_ARG;
    // best-effort teardown: nothing useful can be done with an error here
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    e->drop_graph();
    e->free_batch();
    if (e->d_err) (void)hipFree(e->d_err);
    delete e->model;
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    delete e;
    return FLOWGNN_OK;
}

const char* flowgnn_last_error(const flowgnn_engine* e) {
    if (e && !e->err.empty()) return e->err.c_str();
    return fg::last_error_text();
}

int flowgnn_set_weights_gin(flowgnn_engine* e, const float* node_embedding_weight, const float* edge_embedding_weight,
                            const float* node_mlp_1_weights, const float* node_mlp_1_bias,
                            const float* node_mlp_2_weights, const float* node_mlp_2_bias,
                            const float* graph_pred_weights, const float* graph_pred_bias) {
    if (!e || (e->model_id != FLOWGNN_MODEL_GIN && e->model_id != FLOWGNN_MODEL_GIN_VN)) return FLOWGNN_ERR_ARG;
    const float* t[8] = {node_embedding_weight, edge_embedding_weight, node_mlp_1_weights, node_mlp_1_bias,
                         node_mlp_2_weights,    node_mlp_2_bias,       graph_pred_weights, graph_pred_bias};
    for (auto p : t)
        if (!p) return FLOWGNN_ERR_ARG;
    ENGINE_TRY(e, use_device(e));
    e->drop_graph();
    if (e->stream) EHIP_TRY(e, hipStreamSynchronize(e->stream));
    ENGINE_TRY(e, e->model->set_weights(t));
    return FLOWGNN_OK;
}

int flowgnn_set_weights(flowgnn_engine* e, int count, const float* const* tensors) {
    if (!e || !tensors || count != e->model->num_weight_tensors()) return FLOWGNN_ERR_ARG;
    for (int i = 0; i < count; i++)
        if (!tensors[i]) return FLOWGNN_ERR_ARG;
    ENGINE_TRY(e, use_device(e));
    e->drop_graph();
    if (e->stream) EHIP_TRY(e, hipStreamSynchronize(e->stream));
    ENGINE_TRY(e, e->model->set_weights(tensors));
    return FLOWGNN_OK;
}

int flowgnn_load_weights_dir(flowgnn_engine* e, const char* dir) {
    if (!e || !dir) return FLOWGNN_ERR_ARG;
    ENGINE_TRY(e, use_device(e));
    e->drop_graph();
    if (e->stream) EHIP_TRY(e, hipStreamSynchronize(e->stream));
    ENGINE_TRY(e, e->model->load_weights_dir(dir));
    return FLOWGNN_OK;
}

static int alloc_batch(flowgnn_engine* e, size_t G, size_t N, size_t E, bool attr, bool eig) {
    const int D = e->model->emb_dim(), SD = e->model->scratch_dim();
    G *= (size_t)e->num_tasks;  // capG counts result slots (only d_out and the small per-graph arrays scale with it)
    if (G > e->capG || N > e->capN || E > e->capE || (attr && !e->d_ea) || (eig && !e->d_eig)) {
        // Capacities only grow, each on its own, and a regrow leaves an eighth of headroom: an engine that is handed the ranges of
        // a cut job one after the other (flowgnn_group_compute) sees counts that differ by a few percent from range to range, and
        // every reallocation is two dozen hipFree / hipMalloc pairs that wait for the device.
        const bool regrow = e->capN > 0 || e->capE > 0;
        if (G < e->capG) G = e->capG;
        if (N < e->capN) N = e->capN;
        if (E < e->capE) E = e->capE;
        if (regrow) { G += G / 8; N += N / 8; E += E / 8; }
        e->free_batch();
        const size_t g1 = G ? G : 1, n1 = N ? N : 1, e1 = E ? E : 1;
        // the four per-graph arrays share ONE allocation (d_nn is its base; flowgnn_set_batch places the other three behind it and
        // fills all four with one copy: every host -> device copy costs ~20 us whatever its size, which is what a dataset-sized
        // batch's flowgnn_set_batch is made of)
        EHIP_TRY(e, hipMalloc((void**)&e->d_nn, sizeof(int) * (4 * g1 + 2)));
        e->d_ne = e->d_noff = e->d_eoff = nullptr;
        EHIP_TRY(e, hipMalloc((void**)&e->d_nf, sizeof(int) * n1 * ND_FEATURE));
        EHIP_TRY(e, hipMalloc((void**)&e->d_el, sizeof(int) * e1 * 2));
        if (attr) EHIP_TRY(e, hipMalloc((void**)&e->d_ea, sizeof(int) * e1 * EDGE_ATTR));
        if (eig) EHIP_TRY(e, hipMalloc((void**)&e->d_eig, sizeof(float) * n1 * 4));
        EHIP_TRY(e, hipMalloc((void**)&e->d_rowptr, sizeof(int) * (n1 + 1)));
        EHIP_TRY(e, hipMalloc((void**)&e->d_src, sizeof(int) * e1));
        EHIP_TRY(e, hipMalloc((void**)&e->d_eid, sizeof(int) * e1));
        EHIP_TRY(e, hipMalloc((void**)&e->d_ecode, sizeof(uint8_t) * e1));
        EHIP_TRY(e, hipMalloc((void**)&e->d_outdeg, sizeof(int) * n1));
        EHIP_TRY(e, hipMalloc((void**)&e->d_gsrc, sizeof(int) * e1));
        EHIP_TRY(e, hipMalloc((void**)&e->d_gdst, sizeof(int) * e1));
        EHIP_TRY(e, hipMalloc((void**)&e->d_cursor, sizeof(int) * n1));
        EHIP_TRY(e, hipMalloc((void**)&e->d_tmp, sizeof(int) * e1 * 2));
        EHIP_TRY(e, hipMalloc((void**)&e->d_bsums, sizeof(int) * (n1 / 2048 + 2)));
        // + 4 KiB slack: tile loaders read whole 1 KiB pieces and may run past the last row
        EHIP_TRY(e, hipMalloc((void**)&e->d_h0, sizeof(float) * n1 * D + 4096));
        EHIP_TRY(e, hipMalloc((void**)&e->d_h1, sizeof(float) * n1 * D + 4096));
        EHIP_TRY(e, hipMalloc((void**)&e->d_scratch, sizeof(float) * n1 * (SD > 0 ? SD : 1) + 4096));
        EHIP_TRY(e, hipMalloc((void**)&e->d_out, sizeof(float) * g1));
        e->capG = G; e->capN = N; e->capE = E;
    }
    return 0;
}

int flowgnn_set_job_totals(flowgnn_engine* e, long long job_nodes, long long job_edges) {
    if (!e) return FLOWGNN_ERR_ARG;
    if ((job_nodes < 0) != (job_edges < 0)) {
        e->err = "flowgnn_set_job_totals: both totals, or -1 for both";
        fg::set_last_error(e->err.c_str());
        return FLOWGNN_ERR_ARG;
    }
    e->job_n = job_nodes < 0 ? -1 : job_nodes;
    e->job_e = job_edges < 0 ? -1 : job_edges;
    return FLOWGNN_OK;
}

// fill of the model's graph tiles when `num_graphs` graphs are packed greedily in order (the packing flowgnn_set_batch does), without
// the last tile; 1 for a one-tile batch, 0 when a graph exceeds the tile limits (no resident path), -1 when the model has no tiles
static double graph_tile_fill(fg::Model* model, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges) {
    int t_rows = 0, t_edges = 0;
    model->graph_tile_limits(t_rows, t_edges);
    if (t_rows <= 0 || num_graphs <= 0) return -1.0;
    long long rows_before_last = 0, rows = 0, tiles = 1;
    int cr = 0, ce = 0;
    for (int g = 0; g < num_graphs; g++) {
        const int n = nums_of_nodes[g], m = nums_of_edges[g];
        if (n > t_rows || m > t_edges) return 0.0;
        if (cr + n > t_rows || ce + m > t_edges) { tiles++; rows_before_last = rows; cr = 0; ce = 0; }
        cr += n; ce += m; rows += n;
    }
    return tiles > 1 ? (double)rows_before_last / ((double)(tiles - 1) * t_rows) : 1.0;
}

int flowgnn_set_job_tile_fill(flowgnn_engine* e, double fill) {
    if (!e) return FLOWGNN_ERR_ARG;
    e->job_fill = fill < 0.0 ? -1.0 : fill;
    return FLOWGNN_OK;
}

int flowgnn_graph_tile_fill(flowgnn_engine* e, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges, double* fill) {
    if (!e || !fill || num_graphs < 0 || (num_graphs > 0 && (!nums_of_nodes || !nums_of_edges))) return FLOWGNN_ERR_ARG;
    *fill = graph_tile_fill(e->model, num_graphs, nums_of_nodes, nums_of_edges);
    return FLOWGNN_OK;
}

// copy_mu (flowgnn_group_compute): held around the large host -> device copies only -- one copier per DEVICE at a time, while another
// engine of the same device packs its next range on the host or packs its tiles
static int set_batch_impl(flowgnn_engine* e, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                          const int* node_feature, const int* edge_list, const int* edge_attr, const float* node_eigen, std::mutex* copy_mu) {
    if (!e || num_graphs < 0) return FLOWGNN_ERR_ARG;
    if (num_graphs > 0 && (!nums_of_nodes || !nums_of_edges)) return FLOWGNN_ERR_ARG;
    const bool attr = e->model->has_edge_attr();
    const bool eig = (e->model_id == FLOWGNN_MODEL_DGN);
    // prefix sums of node / edge counts: what the reference carries as nodes_offset / edges_offset
    // (GIN/src/GIN_compute.cc:44,96-97)
    std::vector<int> noff((size_t)num_graphs + 1), eoff((size_t)num_graphs + 1);
    long long N = 0, E = 0;
    int mx_n = 0, mx_e = 0;
    for (int g = 0; g < num_graphs; g++) {
        if (nums_of_nodes[g] <= 0 || nums_of_edges[g] < 0) {
            e->err = "graph with num_of_nodes <= 0 or num_of_edges < 0";
            return FLOWGNN_ERR_ARG;
        }
        noff[g] = (int)N;
        eoff[g] = (int)E;
        if (nums_of_nodes[g] > mx_n) mx_n = nums_of_nodes[g];
        if (nums_of_edges[g] > mx_e) mx_e = nums_of_edges[g];
        N += nums_of_nodes[g];
        E += nums_of_edges[g];
        if (N > 0x7fffffffLL / 128 || E > 0x7fffffffLL / 4) {
            e->err = "batch too large for int32 indexing";
            return FLOWGNN_ERR_ARG;
        }
    }
    noff[num_graphs] = (int)N;
    eoff[num_graphs] = (int)E;
    if (N > 0 && !node_feature) return FLOWGNN_ERR_ARG;
    if (E > 0 && (!edge_list || (attr && !edge_attr))) return FLOWGNN_ERR_ARG;
    if (eig && N > 0 && !node_eigen) return FLOWGNN_ERR_ARG;

    ENGINE_TRY(e, use_device(e));
    e->drop_graph();
    if (e->stream) EHIP_TRY(e, hipStreamSynchronize(e->stream));
    e->batch_ready = false;
    e->ran = false;
    ENGINE_TRY(e, alloc_batch(e, (size_t)num_graphs, (size_t)N, (size_t)E, attr, eig));
    auto h2d = [&](void* dst, const void* src, size_t bytes) -> int {
        if (bytes == 0) return 0;
        if (e->copy_stream) {  // (the source is the caller's pageable memory: the call returns when the data has left it)
            EHIP_TRY(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->copy_stream));
            EHIP_TRY(e, hipStreamSynchronize(e->copy_stream));
        } else {
            EHIP_TRY(e, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        }
        return 0;
    };
    {   // counts and offsets: one staging vector, one copy
        const size_t G = (size_t)num_graphs;
        std::vector<int> meta(4 * G + 2);
        if (G) { memcpy(meta.data(), nums_of_nodes, sizeof(int) * G); memcpy(meta.data() + G, nums_of_edges, sizeof(int) * G); }
        memcpy(meta.data() + 2 * G, noff.data(), sizeof(int) * (G + 1));
        memcpy(meta.data() + 3 * G + 1, eoff.data(), sizeof(int) * (G + 1));
        e->d_ne = e->d_nn + G; e->d_noff = e->d_nn + 2 * G; e->d_eoff = e->d_nn + 3 * G + 1;
        ENGINE_TRY(e, h2d(e->d_nn, meta.data(), sizeof(int) * meta.size()));
    }
    // The three int32 arrays narrowed on the host (9 B per node, 5 B per edge: a quarter of the bytes), copied from pinned memory and
    // widened again on the GPU (h2d_pack.cpp; option h2d_pack, 0 = off).  Large batches only: below a few megabytes the plain copies
    // are latency, not bytes.  Not for GAT (its nine node features are NUMBERS, any integer is valid input) nor for graphs whose
    // node ids do not fit 16 bits.
    const size_t plain_bytes = sizeof(int) * ((size_t)N * ND_FEATURE + (size_t)E * 2 + (attr ? (size_t)E * EDGE_ATTR : 0));
    const int pack_threads = e->opts.i("h2d_pack") > 0 ? host_threads(e->opts.i("h2d_pack")) : 0;
    if (pack_threads > 0 && e->copy_stream && plain_bytes >= ((size_t)8 << 20) && mx_n <= 65535 && e->model_id != FLOWGNN_MODEL_GAT) {
        size_t off_e = 0, off_a = 0;
        const size_t pb = fg::h2d_pack_bytes((size_t)N, (size_t)E, attr, &off_e, &off_a);
        if (pb > e->cap_pack) {
            if (e->h_pack) (void)hipHostFree(e->h_pack);
            if (e->d_pack) (void)hipFree(e->d_pack);
            e->h_pack = e->d_pack = nullptr;
            e->cap_pack = 0;
            const size_t cap = pb + pb / 8;
            EHIP_TRY(e, hipHostMalloc((void**)&e->h_pack, cap, hipHostMallocDefault));
            EHIP_TRY(e, hipMalloc((void**)&e->d_pack, cap));
            e->cap_pack = cap;
        }
        fg::h2d_pack(node_feature, edge_list, attr ? edge_attr : nullptr, (size_t)N, (size_t)E, e->h_pack, pack_threads);
        std::unique_lock<std::mutex> lk;
        if (copy_mu) lk = std::unique_lock<std::mutex>(*copy_mu);
        EHIP_TRY(e, hipMemcpyAsync(e->d_pack, e->h_pack, pb, hipMemcpyHostToDevice, e->copy_stream));
        launch_unpack_batch(e->d_pack, e->d_pack + off_e, attr ? e->d_pack + off_a : nullptr, e->d_nf, e->d_el, attr ? e->d_ea : nullptr,
                            (long long)N, (long long)E, e->copy_stream);
        if (eig) EHIP_TRY(e, hipMemcpyAsync(e->d_eig, node_eigen, sizeof(float) * (size_t)N * 4, hipMemcpyHostToDevice, e->copy_stream));
        EHIP_TRY(e, hipStreamSynchronize(e->copy_stream));
    } else {
        std::unique_lock<std::mutex> lk;
        if (copy_mu) lk = std::unique_lock<std::mutex>(*copy_mu);
        ENGINE_TRY(e, h2d(e->d_nf, node_feature, sizeof(int) * (size_t)N * ND_FEATURE));
        ENGINE_TRY(e, h2d(e->d_el, edge_list, sizeof(int) * (size_t)E * 2));
        if (attr) ENGINE_TRY(e, h2d(e->d_ea, edge_attr, sizeof(int) * (size_t)E * EDGE_ATTR));
        if (eig) ENGINE_TRY(e, h2d(e->d_eig, node_eigen, sizeof(float) * (size_t)N * 4));
    }

    // graph-aligned tiles for kernels that keep whole graphs on chip across layers (GraphTiles, common.h)
    e->db.gtiles = GraphTiles{};
    {
        int t_rows = 0, t_edges = 0;
        e->model->graph_tile_limits(t_rows, t_edges);
        if (t_rows > 0 && num_graphs > 0) {
            std::vector<int> trow, tgraph;
            bool fits = true;
            // whole graphs, in batch order, into tiles of at most `cap` rows / t_edges in-edges
            auto pack = [&](int cap, std::vector<int>& tr, std::vector<int>& tg) {
                tr.clear(); tg.clear();
                tr.push_back(0);
                tg.push_back(0);
                int cr = 0, ce = 0;
                for (int g = 0; g < num_graphs; g++) {
                    const int n = nums_of_nodes[g], m = nums_of_edges[g];
                    if (n > cap || m > t_edges) return false;
                    if (cr + n > cap || ce + m > t_edges) {
                        tr.push_back(noff[g]);
                        tg.push_back(g);
                        cr = 0; ce = 0;
                    }
                    cr += n; ce += m;
                }
                return true;
            };
            fits = pack(t_rows, trow, tgraph);
            const int std_tiles = (int)trow.size();  // (the closing entry is appended below)
            int std_last_row = fits ? trow.back() : 0;
            // A batch of a few ROUNDS of tiles over the CUs (dataset-sized batches: 4 113 molhiv graphs pack to 410 GIN tiles of 256
            // rows -- two rounds over 256 CUs, the second 60 % full, at the price of two): the same graphs in tiles of fewer rows, as
            // many tiles as fill whole rounds (512 of ~203 rows), cost each CU two SHORTER tiles.  The smallest row cap whose greedy
            // packing needs no more than rounds x 256 tiles, by bisection; the fill the models' thresholds see stays that of the
            // full-size packing (it describes the graphs, not this choice).  Option tile_balance = 0: off.
            constexpr int kCUs = 256, kMaxRounds = 8;
            if (fits && e->opts.on("tile_balance") && std_tiles > 1) {
                const int rounds = (std_tiles + kCUs - 1) / kCUs, target = rounds * kCUs;
                if (rounds <= kMaxRounds && std_tiles < target) {
                    int lo = mx_n, hi = t_rows;  // pack(hi) <= target tiles holds; find the smallest cap that still does
                    std::vector<int> tr2, tg2;
                    while (lo < hi) {
                        const int mid = (lo + hi) / 2;
                        if (pack(mid, tr2, tg2) && (int)tr2.size() <= target) hi = mid; else lo = mid + 1;
                    }
                    if (hi < t_rows && pack(hi, tr2, tg2) && (int)tr2.size() <= target) { trow.swap(tr2); tgraph.swap(tg2); }
                }
            }
            if (fits) {
                trow.push_back((int)N);
                tgraph.push_back(num_graphs);
                const size_t cnt = trow.size();
                if (cnt > e->cap_tiles) {
                    if (e->d_trow) (void)hipFree(e->d_trow);
                    e->d_trow = e->d_tgraph = nullptr;
                    e->cap_tiles = 0;
                    const size_t cap = cnt + cnt / 8;  // (ranges of a cut job differ by a few tiles)
                    EHIP_TRY(e, hipMalloc((void**)&e->d_trow, sizeof(int) * 2 * cap));
                    e->cap_tiles = cap;
                }
                e->d_tgraph = e->d_trow + cnt;  // one allocation, one copy
                trow.insert(trow.end(), tgraph.begin(), tgraph.end());
                ENGINE_TRY(e, h2d(e->d_trow, trow.data(), sizeof(int) * 2 * cnt));
                GraphTiles& gt = e->db.gtiles;
                gt.row_start = e->d_trow; gt.graph_start = e->d_tgraph;
                gt.n_tiles = (int)cnt - 1; gt.rows = t_rows; gt.edges = t_edges; gt.ok = true;
                // how full the tiles are WITHOUT the last one (the tail of the batch, whatever is left over): a shard of a cut job
                // then sees the fill of its graphs' packing, not of its own tail -- a one-tile batch counts as full (one resident
                // launch beats the per-layer sequence on it anyway) -- and takes the path the whole job would take
                gt.fill = std_tiles > 1 ? (double)std_last_row / ((double)(std_tiles - 1) * t_rows) : 1.0;
                // ... and a shard that was told the fill of its JOB (flowgnn_set_job_tile_fill; the group and the entry points do)
                // takes the job's side of the models' thresholds whatever its own graphs pack to
                if (e->job_fill >= 0.0) gt.fill = e->job_fill;
            }
        }
        // bin-packed tile lists (GraphTiles::bp_*): best fit, largest graph first, inside windows of 1 024 consecutive graphs.  Bins are
        // kept in buckets by the rows they have left, so placing a graph is a scan over at most t_rows buckets, not over the bins.
        if (e->db.gtiles.ok && e->model->wants_packed_tile_lists() && num_graphs > 1) {
            constexpr int kWindow = 1024;
            // (flat arrays and a bitmap of the non-empty buckets, windows dealt to the pool of host threads: the packing runs inside every
            // flowgnn_set_batch -- the drop-in symbols call it per range -- so it must cost microseconds per thousand graphs)
            const int n_win = (num_graphs + kWindow - 1) / kWindow;
            std::vector<int> list((size_t)num_graphs), lrow((size_t)num_graphs);  // a window's graphs stay inside its span of the list
            std::vector<std::vector<int>> win_cnt((size_t)n_win), win_rows((size_t)n_win);  // per window: graphs / rows of each of its tiles
            int par = e->opts.i("h2d_pack") > 0 ? host_threads(e->opts.i("h2d_pack")) : 1;
            if (par > (n_win + 3) / 4) par = (n_win + 3) / 4;  // (at least four windows per thread)
            if (par < 1) par = 1;
            fg::host_parallel_for(par, [&](int part) {
                std::vector<int> bin_rows, bin_edges, bin_of((size_t)kWindow), bin_cnt, bin_pos;
                std::vector<std::vector<int>> bucket((size_t)t_rows + 1);  // bucket[r]: bins with r rows left
                std::vector<unsigned long long> nonempty(((size_t)t_rows + 64) / 64);
                std::vector<int> order((size_t)kWindow);
                std::vector<int> count((size_t)t_rows + 2);
                for (int wi_ = part; wi_ < n_win; wi_ += par) {
                    const int w0 = wi_ * kWindow;
                    const int w1 = w0 + kWindow < num_graphs ? w0 + kWindow : num_graphs, wn = w1 - w0;
                    // the window's graphs by node count, largest first (counting sort: n <= t_rows; ties in batch order)
                    std::fill(count.begin(), count.end(), 0);
                    for (int g = w0; g < w1; g++) count[(size_t)(t_rows - nums_of_nodes[g]) + 1]++;
                    for (int r = 0; r <= t_rows; r++) count[(size_t)r + 1] += count[(size_t)r];
                    for (int g = w0; g < w1; g++) order[(size_t)count[(size_t)(t_rows - nums_of_nodes[g])]++] = g;
                    bin_rows.clear();
                    bin_edges.clear();
                    for (int r = 0; r <= t_rows; r++)
                        if (!bucket[(size_t)r].empty()) bucket[(size_t)r].clear();
                    std::fill(nonempty.begin(), nonempty.end(), 0ull);
                    for (int oi = 0; oi < wn; oi++) {
                        const int g = order[(size_t)oi], n = nums_of_nodes[g], m = nums_of_edges[g];
                        int chosen = -1;
                        for (int r = n; r <= t_rows && chosen < 0;) {  // the fullest bin that still takes it: the next non-empty bucket from r = n up
                            const size_t wi = (size_t)r >> 6;
                            const unsigned long long bits = nonempty[wi] >> (r & 63);
                            if (!bits) { r = (int)((wi + 1) << 6); continue; }
                            r += __builtin_ctzll(bits);
                            if (r > t_rows) break;
                            std::vector<int>& bk = bucket[(size_t)r];
                            for (size_t k = bk.size(); k-- > 0;)
                                if (bin_edges[(size_t)bk[k]] + m <= t_edges) { chosen = bk[k]; bk[k] = bk.back(); bk.pop_back(); break; }
                            if (chosen >= 0 && bk.empty()) nonempty[wi] &= ~(1ull << (r & 63));
                            r++;
                        }
                        if (chosen < 0) { chosen = (int)bin_rows.size(); bin_rows.push_back(0); bin_edges.push_back(0); }
                        bin_rows[(size_t)chosen] += n;
                        bin_edges[(size_t)chosen] += m;
                        bin_of[(size_t)oi] = chosen;
                        const int left = t_rows - bin_rows[(size_t)chosen];
                        bucket[(size_t)left].push_back(chosen);
                        nonempty[(size_t)left >> 6] |= 1ull << (left & 63);
                    }
                    // the window's tiles, in the order the bins were opened; inside a tile the graphs largest first
                    const int nb = (int)bin_rows.size();
                    bin_cnt.assign((size_t)nb + 1, 0);
                    for (int oi = 0; oi < wn; oi++) bin_cnt[(size_t)bin_of[(size_t)oi] + 1]++;
                    for (int k = 0; k < nb; k++) bin_cnt[(size_t)k + 1] += bin_cnt[(size_t)k];
                    bin_pos.assign(bin_cnt.begin(), bin_cnt.end() - 1);
                    std::fill(bin_rows.begin(), bin_rows.end(), 0);  // (reused as the running row inside each tile)
                    for (int oi = 0; oi < wn; oi++) {
                        const int k = bin_of[(size_t)oi], g = order[(size_t)oi];
                        const size_t at = (size_t)w0 + (size_t)bin_pos[(size_t)k]++;
                        list[at] = g;
                        lrow[at] = bin_rows[(size_t)k];
                        bin_rows[(size_t)k] += nums_of_nodes[g];
                    }
                    win_cnt[(size_t)wi_].resize((size_t)nb);
                    win_rows[(size_t)wi_] = bin_rows;
                    for (int k = 0; k < nb; k++) win_cnt[(size_t)wi_][(size_t)k] = bin_cnt[(size_t)k + 1] - bin_cnt[(size_t)k];
                }
            });
            std::vector<int> tstart, trow2;
            tstart.push_back(0);
            trow2.push_back(0);
            {
                long long rows_done = 0;
                int graphs_done = 0;
                for (int wi_ = 0; wi_ < n_win; wi_++)
                    for (size_t k = 0; k < win_cnt[(size_t)wi_].size(); k++) {
                        graphs_done += win_cnt[(size_t)wi_][k];
                        rows_done += win_rows[(size_t)wi_][k];
                        tstart.push_back(graphs_done);
                        trow2.push_back((int)rows_done);
                    }
            }
            const size_t T1 = tstart.size(), total = 2 * list.size() + 2 * T1;
            if (total > e->cap_bp) {
                if (e->d_bp) (void)hipFree(e->d_bp);
                e->d_bp = nullptr;
                e->cap_bp = 0;
                const size_t cap = total + total / 8;
                EHIP_TRY(e, hipMalloc((void**)&e->d_bp, sizeof(int) * cap));
                e->cap_bp = cap;
            }
            list.insert(list.end(), lrow.begin(), lrow.end());
            list.insert(list.end(), tstart.begin(), tstart.end());
            list.insert(list.end(), trow2.begin(), trow2.end());
            ENGINE_TRY(e, h2d(e->d_bp, list.data(), sizeof(int) * total));
            GraphTiles& gt = e->db.gtiles;
            gt.bp_list = e->d_bp;
            gt.bp_lrow = e->d_bp + num_graphs;
            gt.bp_graph = e->d_bp + 2 * (size_t)num_graphs;
            gt.bp_row = e->d_bp + 2 * (size_t)num_graphs + T1;
            gt.bp_tiles = (int)T1 - 1;
        }
        int s_rows = 0, s_edges = 0;
        e->model->sub_tile_limits(s_rows, s_edges);
        if (e->db.gtiles.ok && s_rows > 0) {  // half-tile runs + the graphs beyond the half-tile limits (GraphTiles::sub / big_*)
            std::vector<int> sub, brow, bgraph;
            int cr = 0, ce = 0, g0 = -1;
            long long sub_rows_total = 0;
            auto close_run = [&](int g_end) {
                if (g0 >= 0) { sub.push_back(noff[g0]); sub.push_back(cr); sub.push_back(g0); sub.push_back(g_end); }
                g0 = -1; cr = 0; ce = 0;
            };
            for (int g = 0; g < num_graphs; g++) {
                const int n = nums_of_nodes[g], m = nums_of_edges[g];
                if (n > s_rows || m > s_edges) {  // within the full-tile limits (gtiles.ok), beyond the half tile: its own full tile
                    close_run(g);
                    brow.push_back(noff[g]); brow.push_back(noff[g] + n);
                    bgraph.push_back(g); bgraph.push_back(g + 1);
                    continue;
                }
                if (g0 >= 0 && (cr + n > s_rows || ce + m > s_edges)) close_run(g);
                if (g0 < 0) g0 = g;
                cr += n; ce += m;
                sub_rows_total += n;
            }
            close_run(num_graphs);
            const size_t cnt = sub.size() + brow.size() + bgraph.size();
            if (cnt > e->cap_sub) {
                if (e->d_sub) (void)hipFree(e->d_sub);
                e->d_sub = nullptr;
                e->cap_sub = 0;
                EHIP_TRY(e, hipMalloc((void**)&e->d_sub, sizeof(int) * (cnt ? cnt : 1)));
                e->cap_sub = cnt;
            }
            {
                std::vector<int> all(sub);
                all.insert(all.end(), brow.begin(), brow.end());
                all.insert(all.end(), bgraph.begin(), bgraph.end());
                ENGINE_TRY(e, h2d(e->d_sub, all.data(), sizeof(int) * all.size()));
            }
            GraphTiles& gt = e->db.gtiles;
            gt.sub = e->d_sub; gt.n_sub = (int)(sub.size() / 4); gt.sub_rows = s_rows; gt.sub_edges = s_edges;
            gt.big_row = e->d_sub + sub.size(); gt.big_graph = e->d_sub + sub.size() + brow.size(); gt.n_big = (int)(brow.size() / 2);
            gt.sub_ok = true;
            gt.sub_fill = gt.n_sub ? (double)sub_rows_total / ((double)gt.n_sub * s_rows) : 0.0;
        }
    }

    e->G = num_graphs; e->N = N; e->E = E;
    e->max_nodes = mx_n; e->max_edges = mx_e;
    e->has_attr = attr; e->has_eig = eig;
    DeviceBatch& db = e->db;
    db.b.num_graphs = num_graphs; db.b.n_tot = (int)N; db.b.e_tot = (int)E;
    db.job_n = e->job_n >= 0 ? std::max(e->job_n, N) : N;
    db.job_e = e->job_e >= 0 ? std::max(e->job_e, E) : E;
    db.b.nums_of_nodes = e->d_nn; db.b.nums_of_edges = e->d_ne;
    db.b.node_off = e->d_noff; db.b.edge_off = e->d_eoff;
    db.b.node_feature = e->d_nf; db.b.edge_list = e->d_el; db.b.edge_attr = attr ? e->d_ea : nullptr;
    db.csr.row_ptr = e->d_rowptr; db.csr.src = e->d_src; db.csr.eid = e->d_eid;
    db.csr.ecode = e->d_ecode; db.csr.out_deg = e->d_outdeg;
    db.csr.gsrc = e->d_gsrc; db.csr.gdst = e->d_gdst; db.csr.cursor = e->d_cursor; db.csr.tmp = e->d_tmp;
    db.csr.block_sums = e->d_bsums; db.csr.err = e->d_err;
    db.node_eigen = eig ? e->d_eig : nullptr;
    db.h[0] = e->d_h0; db.h[1] = e->d_h1; db.scratch = e->d_scratch; db.out = e->d_out;
    db.num_tasks = e->num_tasks;
    db.final_h = 0;
    db.tap = nullptr;
    db.tap_dim = 0;
    db.csr_built = false;
    db.max_nodes = mx_n; db.max_edges = mx_e;
    EHIP_TRY(e, hipMemset(e->d_err, 0, 2 * sizeof(int)));
    e->db.range_flag = e->d_err + 1;
    e->force_exact = false;
    e->batch_ready = true;
    return FLOWGNN_OK;
}

int flowgnn_set_batch(flowgnn_engine* e, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                      const int* node_feature, const int* edge_list, const int* edge_attr, const float* node_eigen) {
    return set_batch_impl(e, num_graphs, nums_of_nodes, nums_of_edges, node_feature, edge_list, edge_attr, node_eigen, nullptr);
}

int flowgnn_run(flowgnn_engine* e) {
    if (!e) return FLOWGNN_ERR_ARG;
    if (!e->batch_ready || !e->model->weights_ready()) {
        e->err = "flowgnn_run: weights or batch not set";
        return FLOWGNN_ERR_STATE;
    }
    ENGINE_TRY(e, use_device(e));
    if (e->G == 0) { e->ran = true; return FLOWGNN_OK; }
    const bool want_graph = e->graph_mode != 0 && !e->prof.enabled && (e->graph_mode > 1 || e->N <= (1ll << 20));
    if (want_graph && e->graph_ok) {
        e->db.tap = e->graph_tap;
        e->db.tap_dim = e->graph_tap_dim;
        e->db.final_h = e->graph_final_h;
        e->db.h_valid = e->graph_h_valid;
        EHIP_TRY(e, hipGraphLaunch(e->gexec, e->stream));
        e->graph_replays++;
        e->ran = true;
        return FLOWGNN_OK;
    }
    const bool capture = want_graph && e->plain_runs >= 1;
    if (capture) {
        hipError_t hc = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal);
        if (hc != hipSuccess) {
            (void)hipGetLastError();
            e->graph_mode = 0;  // this runtime cannot capture: stay on plain launches
            return flowgnn_run(e);
        }
    }
    int frc = FLOWGNN_OK;
    e->db.tap = nullptr;
    e->db.tap_dim = 0;
    e->db.h_valid = true;
    e->model->set_exact(e->force_exact);
    frc = engine_forward(e);
    if (capture) {
        hipGraph_t g = nullptr;
        hipError_t hc = hipStreamEndCapture(e->stream, &g);
        if (hc == hipSuccess && frc == FLOWGNN_OK && g) hc = hipGraphInstantiate(&e->gexec, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (hc != hipSuccess || frc != FLOWGNN_OK || !e->gexec) {
            (void)hipGetLastError();
            e->drop_graph();
            e->graph_mode = 0;  // something in this model's forward is not capturable: plain launches from now on
            return flowgnn_run(e);
        }
        e->graph_ok = true;
        e->graph_h_valid = e->db.h_valid;
        e->graph_tap = e->db.tap;
        e->graph_tap_dim = e->db.tap_dim;
        e->graph_final_h = e->db.final_h;
        EHIP_TRY(e, hipGraphLaunch(e->gexec, e->stream));  // the capture only recorded: this is the run itself
        e->graph_replays++;
        e->ran = true;
        return FLOWGNN_OK;
    }
    if (frc != FLOWGNN_OK) {
        e->err = fg::last_error_text();
        return frc;
    }
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) {
        set_hip_error("kernel launch", he, __FILE__, __LINE__);
        e->err = fg::last_error_text();
        return FLOWGNN_ERR_HIP;
    }
    e->plain_runs++;
    e->ran = true;
    return FLOWGNN_OK;
}

int flowgnn_sync(flowgnn_engine* e) {
    if (!e) return FLOWGNN_ERR_ARG;
    ENGINE_TRY(e, use_device(e));
    hipError_t he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess) {
        set_hip_error("hipStreamSynchronize", he, __FILE__, __LINE__);
        e->err = fg::last_error_text();
        return FLOWGNN_ERR_HIP;
    }
    e->prof.collect();
    int flags[2] = {0, 0};
    if (e->copy_stream) {  // (on the engine's own copy queue: the null stream may share a hardware queue with another engine's kernels)
        he = hipMemcpyAsync(flags, e->d_err, sizeof(flags), hipMemcpyDeviceToHost, e->copy_stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->copy_stream);
    } else {
        he = hipMemcpy(flags, e->d_err, sizeof(flags), hipMemcpyDeviceToHost);
    }
    if (he != hipSuccess) {
        set_hip_error("read error flag", he, __FILE__, __LINE__);
        e->err = fg::last_error_text();
        return FLOWGNN_ERR_HIP;
    }
    const int flag = flags[0];
    if (!flag && flags[1] && e->ran && !e->force_exact) {
        // an operand left the range in which the default kernels are fp32-accurate: repeat the pass on the exact
        // kernels, and keep this batch on them for later runs
        e->force_exact = true;
        e->exact_reruns++;
        e->drop_graph();  // the captured launches are the split-f16 ones
        EHIP_TRY(e, hipMemsetAsync(e->d_err + 1, 0, sizeof(int), e->stream));
        e->model->set_exact(true);
        ENGINE_TRY(e, engine_forward(e));
        he = hipStreamSynchronize(e->stream);
        if (he != hipSuccess) {
            set_hip_error("hipStreamSynchronize (exact re-run)", he, __FILE__, __LINE__);
            e->err = fg::last_error_text();
            return FLOWGNN_ERR_HIP;
        }
        e->prof.collect();
    }
    if (flag) {
        e->err = flag == FLOWGNN_ERR_UNSUPPORTED
                     ? "a node of a graph beyond the per-graph size classes has more than 16384 in-edges (the index build's rank sort is quadratic per row)"
                     : "input validation failed on device (edge endpoint / edge attribute / node feature out of range)";
        return flag;
    }
    return FLOWGNN_OK;
}

int flowgnn_get_results(flowgnn_engine* e, float* out_host) {
    if (!e || (!out_host && e->G > 0)) return FLOWGNN_ERR_ARG;
    if (!e->ran) { e->err = "flowgnn_get_results before flowgnn_run"; return FLOWGNN_ERR_STATE; }
    int rc = flowgnn_sync(e);
    if (rc) return rc;
    if (e->G > 0) {
        hipError_t he;
        if (e->copy_stream) {
            he = hipMemcpyAsync(out_host, e->db.out, sizeof(float) * (size_t)e->G * e->num_tasks, hipMemcpyDeviceToHost, e->copy_stream);
            if (he == hipSuccess) he = hipStreamSynchronize(e->copy_stream);
        } else {
            he = hipMemcpy(out_host, e->db.out, sizeof(float) * (size_t)e->G * e->num_tasks, hipMemcpyDeviceToHost);
        }
        if (he != hipSuccess) {
            set_hip_error("copy results", he, __FILE__, __LINE__);
            e->err = fg::last_error_text();
            return FLOWGNN_ERR_HIP;
        }
    }
    return FLOWGNN_OK;
}

int flowgnn_results_device(flowgnn_engine* e, void** d_out) {
    if (!e || !d_out) return FLOWGNN_ERR_ARG;
    if (!e->batch_ready) return FLOWGNN_ERR_STATE;
    *d_out = e->db.out;
    return FLOWGNN_OK;
}

int flowgnn_set_results_buffer(flowgnn_engine* e, void* device_ptr) {
    if (!e) return FLOWGNN_ERR_ARG;
    if (!e->batch_ready) return FLOWGNN_ERR_STATE;
    // no device synchronisation: launches are stream-ordered, the pointer only matters to launches enqueued after this call
    // (a recorded launch sequence bakes the old pointer in, so that is dropped)
    if (e->gexec) { ENGINE_TRY(e, use_device(e)); e->drop_graph(); }
    e->db.out = device_ptr ? (float*)device_ptr : e->d_out;
    return FLOWGNN_OK;
}

int flowgnn_stream(flowgnn_engine* e, void** stream) {
    if (!e || !stream) return FLOWGNN_ERR_ARG;
    *stream = (void*)e->stream;
    return FLOWGNN_OK;
}

int flowgnn_set_stream(flowgnn_engine* e, void* stream, int use_external) {
    if (!e) return FLOWGNN_ERR_ARG;
    ENGINE_TRY(e, use_device(e));
    e->drop_graph();
    EHIP_TRY(e, hipStreamSynchronize(e->stream));
    e->prof.collect();
    e->stream = use_external ? (hipStream_t)stream : e->own_stream;
    return FLOWGNN_OK;
}

int flowgnn_batch_info(const flowgnn_engine* e, long long* num_graphs, long long* total_nodes, long long* total_edges) {
    if (!e) return FLOWGNN_ERR_ARG;
    if (num_graphs) *num_graphs = e->G;
    if (total_nodes) *total_nodes = e->N;
    if (total_edges) *total_edges = e->E;
    return FLOWGNN_OK;
}

int flowgnn_batch_tiles(const flowgnn_engine* e, int* batch_order, int* packed) {
    if (!e) return FLOWGNN_ERR_ARG;
    if (batch_order) *batch_order = e->batch_ready && e->db.gtiles.ok ? e->db.gtiles.n_tiles : 0;
    if (packed) *packed = e->batch_ready && e->db.gtiles.ok ? e->db.gtiles.bp_tiles : 0;
    return FLOWGNN_OK;
}

int flowgnn_exact_reruns(const flowgnn_engine* e) { return e ? e->exact_reruns : -1; }

long long flowgnn_graph_replays(const flowgnn_engine* e) { return e ? e->graph_replays : -1; }

int flowgnn_set_num_tasks(flowgnn_engine* e, int num_tasks) {
    if (!e || num_tasks < 1) return FLOWGNN_ERR_ARG;
    ENGINE_TRY(e, use_device(e));
    e->drop_graph();
    if (e->stream) EHIP_TRY(e, hipStreamSynchronize(e->stream));
    const int rc = e->model->set_num_tasks(num_tasks);
    if (rc) { e->err = "flowgnn_set_num_tasks: this model's readout has a single task (multi-task readout exists for GIN / GIN-VN / GCN)"; return rc; }
    e->num_tasks = num_tasks;
    e->batch_ready = false;  // the result buffer is sized by the batch: set the batch again
    e->ran = false;
    return FLOWGNN_OK;
}

int flowgnn_num_tasks(const flowgnn_engine* e) { return e ? e->num_tasks : -1; }

int flowgnn_set_numeric_mode(flowgnn_engine* e, int mode) {
    if (!e) return FLOWGNN_ERR_ARG;
    e->drop_graph();
    const int rc = e->model->set_numeric_mode(mode);
    if (rc) e->err = "flowgnn_set_numeric_mode: unknown mode, or the fixed-point readout is single-task and NUM_TASK != 1";
    return rc;
}

int flowgnn_set_option(flowgnn_engine* e, const char* key, double value) {
    if (!e || !key) return FLOWGNN_ERR_ARG;
    ENGINE_TRY(e, use_device(e));
    e->drop_graph();
    if (e->stream) EHIP_TRY(e, hipStreamSynchronize(e->stream));
    if (!e->opts.set(key, value)) {
        e->err = std::string("flowgnn_set_option: unknown option '") + key + "'";
        return FLOWGNN_ERR_UNSUPPORTED;
    }
    e->graph_mode = e->opts.i("hipgraph");
    e->model->configure(e->opts);
    // the batch's graph tiles were packed for the limits the model asked for under the old options: set the batch again
    e->batch_ready = false;
    e->ran = false;
    return FLOWGNN_OK;
}

int flowgnn_get_option(const flowgnn_engine* e, const char* key, double* value) {
    if (!e || !key) return FLOWGNN_ERR_ARG;
    return e->opts.get(key, value) ? FLOWGNN_OK : FLOWGNN_ERR_UNSUPPORTED;
}

int flowgnn_option_count(void) { return kNumOptions; }
const char* flowgnn_option_name(int i) { return (i >= 0 && i < kNumOptions) ? kOptionTable[i].key : nullptr; }

int flowgnn_get_csr(flowgnn_engine* e, int* row_ptr, int* src, int* eid, int* out_deg) {
    if (!e) return FLOWGNN_ERR_ARG;
    if (!e->ran) return FLOWGNN_ERR_STATE;
    ENGINE_TRY(e, use_device(e));
    ensure_csr(e);  // the last run may have worked from the caller's arrays directly
    int rc = flowgnn_sync(e);
    if (rc) return rc;
    if (row_ptr && e->N == 0) row_ptr[0] = 0;  // empty batch: nothing was built (and nothing may have been allocated)
    if (row_ptr && e->N) EHIP_TRY(e, hipMemcpy(row_ptr, e->d_rowptr, sizeof(int) * ((size_t)e->N + 1), hipMemcpyDeviceToHost));
    if (src && e->E) EHIP_TRY(e, hipMemcpy(src, e->d_src, sizeof(int) * (size_t)e->E, hipMemcpyDeviceToHost));
    if (eid && e->E) EHIP_TRY(e, hipMemcpy(eid, e->d_eid, sizeof(int) * (size_t)e->E, hipMemcpyDeviceToHost));
    if (out_deg && e->N) EHIP_TRY(e, hipMemcpy(out_deg, e->d_outdeg, sizeof(int) * (size_t)e->N, hipMemcpyDeviceToHost));
    return FLOWGNN_OK;
}

// The last run kept no per-node rows (a graph-resident kernel, or a readout folded into the last layer): repeat the pass with the
// tap on, on the per-layer kernels -- which also leaves the model's per-batch state of that path (row tiles of the stand-alone
// aggregation kernels) describing THIS batch.  What flowgnn_get_h, flowgnn_get_aggregate and flowgnn_run_aggregation_only read.
static int ensure_rows(flowgnn_engine* e) {
    int rc = flowgnn_sync(e);
    if (rc) return rc;
    if (e->db.h_valid || e->db.tap) return FLOWGNN_OK;
    e->model->set_keep_h(true);
    e->model->set_exact(e->force_exact);
    rc = engine_forward(e);
    e->model->set_keep_h(false);
    if (rc) { e->err = fg::last_error_text(); return rc; }
    return flowgnn_sync(e);
}

int flowgnn_get_h(flowgnn_engine* e, float* h_host, int* dim) {
    if (!e) return FLOWGNN_ERR_ARG;
    if (!e->ran) return FLOWGNN_ERR_STATE;
    int rc = ensure_rows(e);
    if (rc) return rc;
    if (!e->db.h_valid && !e->db.tap) {
        e->err = "flowgnn_get_h: node embeddings are not available as float rows in this numeric mode";
        return FLOWGNN_ERR_UNSUPPORTED;
    }
    const int D = e->db.tap ? e->db.tap_dim : e->model->emb_dim();
    const float* srcp = e->db.tap ? e->db.tap : e->db.h[e->db.final_h];
    if (dim) *dim = D;
    if (h_host && e->N) EHIP_TRY(e, hipMemcpy(h_host, srcp, sizeof(float) * (size_t)e->N * D, hipMemcpyDeviceToHost));
    return FLOWGNN_OK;
}

int flowgnn_profile_enable(flowgnn_engine* e, int on) {
    if (!e) return FLOWGNN_ERR_ARG;
    ENGINE_TRY(e, use_device(e));
    EHIP_TRY(e, hipStreamSynchronize(e->stream));
    e->prof.reset();
    e->prof.enabled = on != 0;
    return FLOWGNN_OK;
}

int flowgnn_profile_read(flowgnn_engine* e, int* count, const char** names, double* total_ms, long long* launches) {
    if (!e || !count) return FLOWGNN_ERR_ARG;
    ENGINE_TRY(e, use_device(e));
    EHIP_TRY(e, hipStreamSynchronize(e->stream));
    e->prof.collect();
    int n = (int)e->prof.names.size();
    if (n > FLOWGNN_MAX_PROFILE_SLOTS) n = FLOWGNN_MAX_PROFILE_SLOTS;
    for (int i = 0; i < n; i++) {
        if (names) names[i] = e->prof.names[i].c_str();
        if (total_ms) total_ms[i] = e->prof.total_ms[i];
        if (launches) launches[i] = e->prof.launches[i];
    }
    *count = n;
    return FLOWGNN_OK;
}

int flowgnn_run_aggregation_only(flowgnn_engine* e, int layer, int iters, float* avg_ms) {
    if (!e || iters <= 0) return FLOWGNN_ERR_ARG;
    if (!e->ran) { e->err = "flowgnn_run_aggregation_only needs a prior flowgnn_run"; return FLOWGNN_ERR_STATE; }
    ENGINE_TRY(e, use_device(e));
    e->drop_graph();
    int rc = e->model->aggregate_dim() > 0 ? ensure_rows(e) : FLOWGNN_OK;  // the kernel's input rows (a resident run left none)
    if (rc) return rc;
    ensure_csr(e);
    rc = e->model->aggregation_only(e->db, layer, e->stream);  // warm-up; also the model's verdict on `layer`
    if (rc) {
        e->err = rc == FLOWGNN_ERR_UNSUPPORTED ? "no standalone aggregation kernel for this model / numeric mode (the fixed-point modes have none)" : "flowgnn_run_aggregation_only: bad layer";
        return rc;
    }
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t he = hipEventCreate(&a);
    if (he == hipSuccess) he = hipEventCreate(&b);
    if (he == hipSuccess) he = hipEventRecord(a, e->stream);
    for (int i = 0; i < iters && he == hipSuccess && rc == 0; i++) rc = e->model->aggregation_only(e->db, layer, e->stream);
    if (he == hipSuccess) he = hipEventRecord(b, e->stream);
    if (he == hipSuccess) he = hipEventSynchronize(b);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, a, b);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (he != hipSuccess) {
        set_hip_error("flowgnn_run_aggregation_only", he, __FILE__, __LINE__);
        e->err = fg::last_error_text();
        return FLOWGNN_ERR_HIP;
    }
    if (rc) { e->err = fg::last_error_text(); return rc; }
    if (avg_ms) *avg_ms = ms / iters;
    return FLOWGNN_OK;
}

int flowgnn_get_aggregate(flowgnn_engine* e, int layer, float* h_in_host, int* in_dim, float* agg_host, int* agg_dim) {
    if (!e) return FLOWGNN_ERR_ARG;
    if (!e->ran) { e->err = "flowgnn_get_aggregate needs a prior flowgnn_run"; return FLOWGNN_ERR_STATE; }
    int rc = flowgnn_sync(e);
    if (rc) return rc;
    e->drop_graph();
    const int D = e->model->emb_dim(), AD = e->model->aggregate_dim();
    if (in_dim) *in_dim = D;
    if (agg_dim) *agg_dim = AD;
    if (AD <= 0) { e->err = "no standalone aggregation kernel for this model / numeric mode (the fixed-point modes have none)"; return FLOWGNN_ERR_UNSUPPORTED; }
    if (e->N == 0 || (!h_in_host && !agg_host)) return FLOWGNN_OK;
    rc = ensure_rows(e);
    if (rc) return rc;
    ensure_csr(e);
    rc = e->model->aggregation_only(e->db, layer, e->stream);
    if (rc) { e->err = rc == FLOWGNN_ERR_UNSUPPORTED ? "flowgnn_get_aggregate: not available in this numeric mode" : "flowgnn_get_aggregate: bad layer"; return rc; }
    EHIP_TRY(e, hipStreamSynchronize(e->stream));
    if (h_in_host)
        EHIP_TRY(e, hipMemcpy(h_in_host, e->db.h[e->db.final_h], sizeof(float) * (size_t)e->N * D, hipMemcpyDeviceToHost));
    if (agg_host) EHIP_TRY(e, hipMemcpy(agg_host, e->db.scratch, sizeof(float) * (size_t)e->N * AD, hipMemcpyDeviceToHost));
    // the model's last launch may have left per-node readout terms in scratch: the next flowgnn_run rewrites them
    return FLOWGNN_OK;
}

// ------------------------------------------------------------------ several devices behind one handle
// north_star: "that batch dimension is partitioned across the 8 GPUs of one node".  A group = one engine (own stream, own
// resident shard) per listed device + one host thread per engine for every call that touches the device; the batch is cut
// into contiguous graph ranges balanced by sum(N + E) (flowgnn_shard_ranges, the C counterpart of flowgnn_amd/dist.py) and
// the results are written into the caller's buffer in job order.  A device may be listed more than once (two engines on
// one GPU: what the 1-GPU tests do) -- graphs are independent, so results are bit-identical to the single-engine run.
}  // extern "C"

// One persistent host thread per engine (engine 0 runs on the caller's thread): a call that touches the devices hands every worker the
// same function and waits for all of them.  (Creating and joining a std::thread per engine and call -- what this replaced -- costs
// 60-100 us per call with eight engines; a dataset-sized step is 190 us of GPU time.)  Workers spin briefly for the next job before
// they sleep on the condition variable, so the timed loop of `host --devices` (flowgnn_group_run back to back) never pays a wake-up.
class GroupWorkers {
public:
    ~GroupWorkers() { stop(); }
    void start(int n_engines) {
        n_ = n_engines;
        rc_.assign((size_t)n_engines, 0);
        for (int i = 1; i < n_engines; i++) th_.emplace_back([this, i] { loop(i); });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_go_.notify_all();
        for (auto& t : th_) t.join();
        th_.clear();
    }
    // fn(i) for every engine i; returns the per-engine status codes
    const std::vector<int>& each(const std::function<int(int)>& fn) {
        if (n_ > 1) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                fn_ = &fn;
                pending_.store(n_ - 1, std::memory_order_relaxed);
                gen_.fetch_add(1, std::memory_order_release);
            }
            cv_go_.notify_all();
        }
        // engine 0 runs here, on the caller's thread: its hipSetDevice must not outlive the call (the caller's current device is the
        // caller's business), and whatever fn(0) throws (std::bad_alloc from a staging vector) the workers still hold &fn and write
        // rc_ -- so the wait below runs before anything leaves this frame, and the exception becomes a status code (this is a C ABI)
        int caller_dev = -1;
        const bool have_dev = hipGetDevice(&caller_dev) == hipSuccess;
        try {
            rc_[0] = fn(0);
        } catch (...) {
            rc_[0] = FLOWGNN_ERR_HIP;
        }
        if (n_ > 1) {
            for (int spin = 0; spin < 20000 && pending_.load(std::memory_order_acquire) > 0; spin++) cpu_relax();
            if (pending_.load(std::memory_order_acquire) > 0) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_done_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) == 0; });
            }
            fn_ = nullptr;
        }
        if (have_dev) (void)hipSetDevice(caller_dev);
        return rc_;
    }

private:
    static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void loop(int i) {
        unsigned long long seen = 0;
        while (true) {
            // a short spin (a back-to-back caller is here again within microseconds), then sleep
            for (int spin = 0; spin < 4000 && gen_.load(std::memory_order_acquire) == seen; spin++) cpu_relax();
            if (gen_.load(std::memory_order_acquire) == seen) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_go_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
            }
            const std::function<int(int)>* fn;
            {
                std::lock_guard<std::mutex> lk(mu_);  // pairs with each(): fn_ and gen_ are published together
                seen = gen_.load(std::memory_order_acquire);
                if (quit_) return;
                fn = fn_;
            }
            try {
                rc_[(size_t)i] = (*fn)(i);
            } catch (...) {
                rc_[(size_t)i] = FLOWGNN_ERR_HIP;  // (an exception must not end the worker with the caller still waiting for it)
            }
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                cv_done_.notify_one();
            }
        }
    }
    int n_ = 0;
    std::vector<std::thread> th_;
    std::vector<int> rc_;
    std::mutex mu_;
    std::condition_variable cv_go_, cv_done_;
    const std::function<int(int)>* fn_ = nullptr;
    std::atomic<unsigned long long> gen_{0};
    std::atomic<int> pending_{0};
    bool quit_ = false;
};

struct flowgnn_group {
    int model_id = 0;
    GroupWorkers workers;
    std::mutex call_mu;  // one group call at a time (the workers hold one function)
    std::vector<flowgnn_engine*> eng;
    std::vector<int> cut;  // [n + 1] graph cuts of the resident batch
    bool batch_valid = false;  // the engines hold the shards `cut` describes (flowgnn_group_set_batch); flowgnn_group_compute and the
                               // entry points leave each engine on its LAST range and clear this
    std::vector<std::unique_ptr<std::mutex>> copy_mu;  // flowgnn_group_compute: one copier per device ...
    std::vector<int> copy_of;                          // ... engine i uses copy_mu[copy_of[i]] (the first engine on its device)
    int num_tasks = 1;
    std::string err;
};

namespace {
int group_each(flowgnn_group* g, const std::function<int(int)>& fn) {  // fn(i) on every engine, each on its own (persistent) host thread; first failure wins
    const int n = (int)g->eng.size();
    std::lock_guard<std::mutex> call(g->call_mu);
    const std::vector<int>& rc = g->workers.each(fn);
    for (int i = 0; i < n; i++)
        if (rc[(size_t)i]) {
            g->err = "engine " + std::to_string(i) + " (device " + std::to_string(g->eng[(size_t)i]->device) + "): " + flowgnn_last_error(g->eng[(size_t)i]);
            return rc[(size_t)i];
        }
    return FLOWGNN_OK;
}
// every flowgnn_group_* call starts with no error text of an earlier call; argument errors leave their own
int group_fail(flowgnn_group* g, int rc, const char* what) {
    if (g) g->err = what;
    fg::set_last_error(what);
    return rc;
}
}  // namespace

extern "C" {

int flowgnn_shard_ranges(int num_graphs, const int* nums_of_nodes, const int* nums_of_edges, int parts, int* cuts) {
    if (num_graphs < 0 || parts < 1 || !cuts || (num_graphs > 0 && (!nums_of_nodes || !nums_of_edges))) return FLOWGNN_ERR_ARG;
    long long total = 0;
    for (int g = 0; g < num_graphs; g++) total += (long long)nums_of_nodes[g] + nums_of_edges[g];
    // cut r = the first graph index whose cumulative work reaches r / parts of the total (exact integer comparison)
    cuts[0] = 0;
    long long cum = 0;
    int g = 0;
    for (int r = 1; r < parts; r++) {
        while (g < num_graphs && cum * parts < total * r) { cum += (long long)nums_of_nodes[g] + nums_of_edges[g]; g++; }
        cuts[r] = g;
    }
    cuts[parts] = num_graphs;
    return FLOWGNN_OK;
}

int flowgnn_create_multi(int model, int n_devices, const int* device_ids, flowgnn_group** out) {
    if (!out || n_devices < 1 || !device_ids) return FLOWGNN_ERR_ARG;
    *out = nullptr;
    flowgnn_group* g = new flowgnn_group();
    g->model_id = model;
    for (int i = 0; i < n_devices; i++) {
        flowgnn_engine* e = nullptr;
        const int rc = flowgnn_create(model, device_ids[i], &e);
        if (rc) {
            for (auto* p : g->eng) flowgnn_destroy(p);
            delete g;
            return rc;
        }
        g->eng.push_back(e);
    }
    g->cut.assign((size_t)n_devices + 1, 0);
    g->workers.start(n_devices);
    for (int i = 0; i < n_devices; i++) {
        int first = i;
        for (int k = 0; k < i; k++)
            if (device_ids[k] == device_ids[i]) { first = k; break; }
        g->copy_of.push_back(first);
        g->copy_mu.emplace_back(new std::mutex());
    }
    *out = g;
    return FLOWGNN_OK;
}

int flowgnn_group_destroy(flowgnn_group* g) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->workers.stop();
    for (auto* e : g->eng) flowgnn_destroy(e);
    delete g;
    return FLOWGNN_OK;
}

int flowgnn_group_size(const flowgnn_group* g) { return g ? (int)g->eng.size() : -1; }
flowgnn_engine* flowgnn_group_engine(flowgnn_group* g, int i) { return (g && i >= 0 && i < (int)g->eng.size()) ? g->eng[(size_t)i] : nullptr; }
const char* flowgnn_group_last_error(const flowgnn_group* g) { return (g && !g->err.empty()) ? g->err.c_str() : fg::last_error_text(); }

int flowgnn_group_set_weights(flowgnn_group* g, int count, const float* const* tensors) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    return group_each(g, [&](int i) { return flowgnn_set_weights(g->eng[(size_t)i], count, tensors); });
}
int flowgnn_group_load_weights_dir(flowgnn_group* g, const char* dir) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    return group_each(g, [&](int i) { return flowgnn_load_weights_dir(g->eng[(size_t)i], dir); });
}
int flowgnn_group_set_option(flowgnn_group* g, const char* key, double value) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    return group_each(g, [&](int i) { return flowgnn_set_option(g->eng[(size_t)i], key, value); });
}
int flowgnn_group_set_num_tasks(flowgnn_group* g, int num_tasks) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    const int rc = group_each(g, [&](int i) { return flowgnn_set_num_tasks(g->eng[(size_t)i], num_tasks); });
    if (!rc) g->num_tasks = num_tasks;
    return rc;
}
int flowgnn_group_set_numeric_mode(flowgnn_group* g, int mode) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    return group_each(g, [&](int i) { return flowgnn_set_numeric_mode(g->eng[(size_t)i], mode); });
}

int flowgnn_group_set_batch(flowgnn_group* g, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                            const int* node_feature, const int* edge_list, const int* edge_attr, const float* node_eigen) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    g->batch_valid = false;
    if (num_graphs < 0) return group_fail(g, FLOWGNN_ERR_ARG, "flowgnn_group_set_batch: negative graph count");
    if (num_graphs > 0 && (!nums_of_nodes || !nums_of_edges)) return group_fail(g, FLOWGNN_ERR_ARG, "flowgnn_group_set_batch: null count arrays");
    const int n = (int)g->eng.size();
    g->cut.assign((size_t)n + 1, 0);
    int rc = flowgnn_shard_ranges(num_graphs, nums_of_nodes, nums_of_edges, n, g->cut.data());
    if (rc) return group_fail(g, rc, "flowgnn_group_set_batch: flowgnn_shard_ranges refused the counts");
    // node / edge offsets of every cut (the reference's running nodes_offset / edges_offset, GIN/src/GIN_compute.cc:96-97)
    std::vector<long long> noff((size_t)n + 1, 0), eoff((size_t)n + 1, 0);
    {
        long long N = 0, E = 0;
        int r = 0;
        for (int gi = 0; gi <= num_graphs; gi++) {
            while (r <= n && g->cut[(size_t)r] == gi) { noff[(size_t)r] = N; eoff[(size_t)r] = E; r++; }
            if (gi < num_graphs) { N += nums_of_nodes[gi]; E += nums_of_edges[gi]; }
        }
    }
    // every shard chooses its kernels by the JOB's totals (flowgnn_set_job_totals): the same kernels as one engine holding all of it
    const long long job_n = noff[(size_t)n], job_e = eoff[(size_t)n];
    const double job_fill = graph_tile_fill(g->eng[0]->model, num_graphs, nums_of_nodes, nums_of_edges);  // (the members are one model with one option set)
    rc = group_each(g, [&](int i) {
        const int g0 = g->cut[(size_t)i], g1 = g->cut[(size_t)i + 1];
        const long long n0 = noff[(size_t)i], e0 = eoff[(size_t)i];
        flowgnn_engine* e = g->eng[(size_t)i];
        const long long keep_n = e->job_n, keep_e = e->job_e;
        flowgnn_set_job_totals(e, job_n, job_e);
        const double keep_fill = e->job_fill;
        flowgnn_set_job_tile_fill(e, job_fill);
        struct RestoreFill { flowgnn_engine* e; double f; ~RestoreFill() { e->job_fill = f; } } restore_fill{e, keep_fill};
        const int r = flowgnn_set_batch(e, g1 - g0, nums_of_nodes ? nums_of_nodes + g0 : nullptr,
                                        nums_of_edges ? nums_of_edges + g0 : nullptr, node_feature ? node_feature + n0 * 9 : nullptr,
                                        edge_list ? edge_list + e0 * 2 : nullptr, edge_attr ? edge_attr + e0 * 3 : nullptr,
                                        node_eigen ? node_eigen + n0 * 4 : nullptr);
        flowgnn_set_job_totals(e, keep_n, keep_e);  // a later flowgnn_set_batch on the member itself is its own job again
        return r;
    });
    g->batch_valid = rc == FLOWGNN_OK;
    return rc;
}

// the members still hold the shards flowgnn_group_set_batch gave them?  (flowgnn_group_engine hands the members out for per-engine
// calls: a flowgnn_set_batch on one of them would otherwise have its rows copied to the old cut's offset)
static int group_shards_intact(flowgnn_group* g, const char* who) {
    for (size_t i = 0; i < g->eng.size(); i++)
        if (g->eng[i]->G != g->cut[i + 1] - g->cut[i]) {
            g->batch_valid = false;
            const std::string msg = std::string(who) + ": engine " + std::to_string(i) + " no longer holds its shard of the group's batch (a per-engine flowgnn_set_batch?); call flowgnn_group_set_batch again";
            return group_fail(g, FLOWGNN_ERR_STATE, msg.c_str());
        }
    return FLOWGNN_OK;
}

int flowgnn_group_shards(const flowgnn_group* g, int* cuts) {
    if (!g || !cuts) return FLOWGNN_ERR_ARG;
    if (!g->batch_valid) { fg::set_last_error("flowgnn_group_shards: no batch set by flowgnn_group_set_batch"); return FLOWGNN_ERR_STATE; }
    for (size_t i = 0; i < g->cut.size(); i++) cuts[i] = g->cut[i];
    return FLOWGNN_OK;
}

int flowgnn_group_run(flowgnn_group* g) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    if (!g->batch_valid) return group_fail(g, FLOWGNN_ERR_STATE, "flowgnn_group_run: no batch set by flowgnn_group_set_batch (flowgnn_group_compute and the entry points leave none)");
    if (int rc = group_shards_intact(g, "flowgnn_group_run")) return rc;
    return group_each(g, [&](int i) { return flowgnn_run(g->eng[(size_t)i]); });
}
int flowgnn_group_sync(flowgnn_group* g) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    return group_each(g, [&](int i) { return flowgnn_sync(g->eng[(size_t)i]); });
}
int flowgnn_group_get_results(flowgnn_group* g, float* out_host) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    if (!g->batch_valid) return group_fail(g, FLOWGNN_ERR_STATE, "flowgnn_group_get_results: no batch set by flowgnn_group_set_batch (flowgnn_group_compute and the entry points leave none)");
    if (int rc = group_shards_intact(g, "flowgnn_group_get_results")) return rc;
    if (!out_host && g->cut.back() > 0) return group_fail(g, FLOWGNN_ERR_ARG, "flowgnn_group_get_results: null output");
    return group_each(g, [&](int i) {
        flowgnn_engine* e = g->eng[(size_t)i];
        if (e->G == 0) return flowgnn_sync(e);
        return flowgnn_get_results(e, out_host + (size_t)g->cut[(size_t)i] * g->num_tasks);
    });
}

// One call for a batch that lives in HOST memory: the job is cut into size x chunks_per_engine ranges (same rule), and engine i
// takes ranges i, i + size, ... one after the other -- set_batch (validation, tile packing, host -> device), run, results into
// out_host at the range's place.  While one engine's kernels run, the other engines' copies are in flight: with two engines on ONE
// device the PCIe transfer of range j + 1 hides under the kernels of range j (the entry points do exactly that).  The engines are
// left holding their last range.
int flowgnn_group_compute(flowgnn_group* g, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                          const int* node_feature, const int* edge_list, const int* edge_attr, const float* node_eigen,
                          float* out_host, int chunks_per_engine) {
    if (!g) return FLOWGNN_ERR_ARG;
    g->err.clear();
    if (num_graphs < 0 || chunks_per_engine < 1) return group_fail(g, FLOWGNN_ERR_ARG, "flowgnn_group_compute: negative graph count or chunks_per_engine < 1");
    if (num_graphs > 0 && (!nums_of_nodes || !nums_of_edges || !out_host)) return group_fail(g, FLOWGNN_ERR_ARG, "flowgnn_group_compute: null count arrays or output");
    const int n = (int)g->eng.size();
    const int S = n * chunks_per_engine;
    std::vector<int> cut((size_t)S + 1, 0);
    int rc = flowgnn_shard_ranges(num_graphs, nums_of_nodes, nums_of_edges, S, cut.data());
    if (rc) return group_fail(g, rc, "flowgnn_group_compute: flowgnn_shard_ranges refused the counts");
    // the engines end up holding their LAST range, not the shards of a flowgnn_group_set_batch job: run / get_results / shards
    // answer FLOWGNN_ERR_STATE until the next flowgnn_group_set_batch
    g->batch_valid = false;
    g->cut.assign((size_t)n + 1, 0);
    std::vector<long long> noff((size_t)S + 1, 0), eoff((size_t)S + 1, 0);
    {
        long long N = 0, E = 0;
        int r = 0;
        for (int gi = 0; gi <= num_graphs; gi++) {
            while (r <= S && cut[(size_t)r] == gi) { noff[(size_t)r] = N; eoff[(size_t)r] = E; r++; }
            if (gi < num_graphs) { N += nums_of_nodes[gi]; E += nums_of_edges[gi]; }
        }
    }
    const int T = g->num_tasks;
    const long long job_n = noff[(size_t)S], job_e = eoff[(size_t)S];  // every range chooses its kernels by the job's totals ...
    const double job_fill = graph_tile_fill(g->eng[0]->model, num_graphs, nums_of_nodes, nums_of_edges);  // ... and the job's tile fill
    return group_each(g, [&](int i) {
        flowgnn_engine* e = g->eng[(size_t)i];
        const long long keep_n = e->job_n, keep_e = e->job_e;
        struct Restore { flowgnn_engine* e; long long n, m; double f; ~Restore() { flowgnn_set_job_totals(e, n, m); e->job_fill = f; } } restore{e, keep_n, keep_e, e->job_fill};
        flowgnn_set_job_totals(e, job_n, job_e);
        flowgnn_set_job_tile_fill(e, job_fill);
        for (int j = i; j < S; j += n) {
            const int g0 = cut[(size_t)j], g1 = cut[(size_t)j + 1];
            if (g1 == g0) continue;
            const long long n0 = noff[(size_t)j], e0 = eoff[(size_t)j];
            int r;
            // one host -> device copy per DEVICE at a time (the mutex is taken inside, around the copies only: the host-side packing of
            // this range runs under the other engine's copy): two threads copying from pageable memory to the same GPU get a quarter
            // of the rate each (6.0 ms against 1.4 for a 67 MB range), and the ranges would then march in lockstep instead of
            // alternating copy / kernels
            r = set_batch_impl(e, g1 - g0, nums_of_nodes + g0, nums_of_edges + g0, node_feature ? node_feature + n0 * 9 : nullptr,
                               edge_list ? edge_list + e0 * 2 : nullptr, edge_attr ? edge_attr + e0 * 3 : nullptr,
                               node_eigen ? node_eigen + n0 * 4 : nullptr, g->copy_mu[(size_t)g->copy_of[(size_t)i]].get());
            if (!r) r = flowgnn_run(e);
            if (!r) r = flowgnn_get_results(e, out_host + (size_t)g0 * T);
            if (r) return r;
        }
        return (int)FLOWGNN_OK;
    });
}

// ------------------------------------------------------------------ reference-compatible entry points
// Split the batch into runs of constant weight set (reload_weights semantics of
// GIN/src/GIN_compute.cc:44,51-53) and run each through a process-wide group of engines per model: one engine on device 0
// unless flowgnn_entry_set_devices (or FLOWGNN_DEVICES=0,1,.. at the first call) lists more.
static std::mutex g_entry_mutex;
static flowgnn_group* g_entry_group[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
static std::vector<int> g_entry_devices;  // empty: not decided yet (the environment is asked at the first call)
static int g_entry_pipeline = 0;          // flowgnn_entry_set_pipeline: ranges per engine (0: by the size of the host arrays, 1: off)
static std::vector<std::pair<std::string, double>> g_entry_options[6];
// The weight set an entry-point group holds, kept on the host: a caller that reloads the SAME set on every graph
// (reload_weights = 1 everywhere is legal in the reference and cheap there) must not pay a repack + upload per graph.
// Compared with memcmp -- no hash, no collision to reason about.
static std::vector<float> g_entry_wcopy[6];
static bool same_weights(int model, int ntens, const float* const* t, const size_t* elems) {
    size_t total = 0;
    for (int i = 0; i < ntens; i++) total += elems[i];
    const std::vector<float>& c = g_entry_wcopy[model];
    if (c.size() != total) return false;
    size_t off = 0;
    for (int i = 0; i < ntens; i++) {
        if (memcmp(c.data() + off, t[i], elems[i] * sizeof(float)) != 0) return false;
        off += elems[i];
    }
    return true;
}
static void remember_weights(int model, int ntens, const float* const* t, const size_t* elems) {
    std::vector<float>& c = g_entry_wcopy[model];
    c.clear();
    for (int i = 0; i < ntens; i++) c.insert(c.end(), t[i], t[i] + elems[i]);
}

static void entry_drop_groups() {
    for (int m = 0; m < 6; m++) {
        if (g_entry_group[m]) flowgnn_group_destroy(g_entry_group[m]);
        g_entry_group[m] = nullptr;
        g_entry_wcopy[m].clear();
    }
}

int flowgnn_entry_set_devices(int n_devices, const int* device_ids) {
    if (n_devices < 1 || !device_ids) return FLOWGNN_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_entry_mutex);
    entry_drop_groups();
    g_entry_devices.assign(device_ids, device_ids + n_devices);
    return FLOWGNN_OK;
}

int flowgnn_entry_set_pipeline(int chunks_per_engine) {
    if (chunks_per_engine < 0 || chunks_per_engine > 64) return FLOWGNN_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_entry_mutex);
    if ((g_entry_pipeline == 1) != (chunks_per_engine == 1)) entry_drop_groups();  // the engine count of a one-device list changes
    g_entry_pipeline = chunks_per_engine;
    return FLOWGNN_OK;
}

int flowgnn_entry_set_option(int model, const char* key, double value) {
    if (model < 0 || model >= 6 || !key) return FLOWGNN_ERR_ARG;
    if (option_index(key) < 0) return FLOWGNN_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_entry_mutex);
    bool found = false;
    for (auto& kv : g_entry_options[model])
        if (kv.first == key) { kv.second = value; found = true; }
    if (!found) g_entry_options[model].emplace_back(key, value);
    if (g_entry_group[model]) {
        g_entry_wcopy[model].clear();
        return flowgnn_group_set_option(g_entry_group[model], key, value);
    }
    return FLOWGNN_OK;
}

static int compute_graphs_generic(int model, int num_graphs, const int* nums_of_nodes, const int* nums_of_edges,
                                  const int* reload_weights, float* out, const int* node_feature, const float* node_eigen,
                                  const int* edge_list, const int* edge_attr, int ntens, const float* const* tens,
                                  const size_t* tens_elems, int num_tasks = 1) {
    if (num_graphs < 0 || num_tasks < 1) return FLOWGNN_ERR_ARG;
    if (num_graphs == 0) return FLOWGNN_OK;
    if (!nums_of_nodes || !nums_of_edges || !reload_weights || !out || !node_feature) return FLOWGNN_ERR_ARG;
    for (int i = 0; i < ntens; i++)
        if (!tens[i]) return FLOWGNN_ERR_ARG;
    if (!reload_weights[0]) return FLOWGNN_ERR_ARG;  // the reference would index weight set -1
    std::lock_guard<std::mutex> lock(g_entry_mutex);
    flowgnn_group*& grp = g_entry_group[model];
    if (!grp) {
        if (g_entry_devices.empty()) read_environment(nullptr, &g_entry_devices);
        // one listed device: THREE engines on it, so that a large batch's host-side work (narrowing the arrays for the transfer, packing
        // tiles) and its host -> device copies run under the other engines' kernels (two engines: 13.4 ms per 2^18 molhiv graphs, three:
        // 11.8 -- one engine's host phase per range is longer than another's kernels for a range)
        std::vector<int> devs = g_entry_devices;
        if (devs.size() == 1 && g_entry_pipeline != 1) { devs.push_back(devs[0]); devs.push_back(devs[0]); }
        int rc = flowgnn_create_multi(model, (int)devs.size(), devs.data(), &grp);
        if (rc) return rc;
        for (auto& kv : g_entry_options[model]) {
            rc = flowgnn_group_set_option(grp, kv.first.c_str(), kv.second);
            if (rc) return rc;
        }
        g_entry_wcopy[model].clear();
    }
    if (grp->num_tasks != num_tasks) {
        int rc = flowgnn_group_set_num_tasks(grp, num_tasks);
        if (rc) return rc;
        g_entry_wcopy[model].clear();
    }
    long long noff = 0, eoff = 0;
    int set = -1, g = 0;
    const float* cur[16];
    while (g < num_graphs) {
        set++;
        int g1 = g + 1;
        while (g1 < num_graphs && !reload_weights[g1]) g1++;
        long long n = 0, m = 0;
        for (int i = g; i < g1; i++) { n += nums_of_nodes[i]; m += nums_of_edges[i]; }
        for (int i = 0; i < ntens; i++) cur[i] = tens[i] + (size_t)set * tens_elems[i];
        int rc = FLOWGNN_OK;
        if (!same_weights(model, ntens, cur, tens_elems)) {
            g_entry_wcopy[model].clear();
            rc = flowgnn_group_set_weights(grp, ntens, cur);
            if (rc) return rc;
            remember_weights(model, ntens, cur, tens_elems);
        }
        // ranges per engine: by the size of the host arrays (~48 MB per range, at most 8 per engine); a small batch is ONE range on
        // one engine (cutting it would only add launches and half-empty tiles)
        const int n_eng = flowgnn_group_size(grp);
        int chunks = g_entry_pipeline;
        bool whole = false;
        if (chunks == 0) {
            const double bytes = (double)n * (36.0 + (node_eigen ? 16.0 : 0.0)) + (double)m * (8.0 + (edge_attr ? 12.0 : 0.0));
            const int want = (int)(bytes / 48.0e6);  // ranges in all
            whole = want < 2 && (int)g_entry_devices.size() == 1;
            chunks = (want + n_eng - 1) / n_eng;
            if (chunks < 1) chunks = 1;
            if (chunks > 8) chunks = 8;
        }
        if (whole) {  // everything on engine 0
            flowgnn_engine* e0 = flowgnn_group_engine(grp, 0);
            grp->err.clear();
            grp->batch_valid = false;  // engine 0 is about to hold this range, whatever a flowgnn_group_set_batch left
            rc = flowgnn_set_batch(e0, g1 - g, nums_of_nodes + g, nums_of_edges + g, node_feature + noff * 9,
                                   edge_list ? edge_list + eoff * 2 : nullptr, edge_attr ? edge_attr + eoff * 3 : nullptr,
                                   node_eigen ? node_eigen + noff * 4 : nullptr);
            if (!rc) rc = flowgnn_run(e0);
            if (!rc) rc = flowgnn_get_results(e0, out + (size_t)g * num_tasks);
            if (rc) { grp->err = flowgnn_last_error(e0); return rc; }
        } else {
            rc = flowgnn_group_compute(grp, g1 - g, nums_of_nodes + g, nums_of_edges + g, node_feature + noff * 9,
                                       edge_list ? edge_list + eoff * 2 : nullptr, edge_attr ? edge_attr + eoff * 3 : nullptr,
                                       node_eigen ? node_eigen + noff * 4 : nullptr, out + (size_t)g * num_tasks, chunks);
            if (rc) return rc;
        }
        noff += n;
        eoff += m;
        g = g1;
    }
    return FLOWGNN_OK;
}

int GIN_compute_graphs_mt(int num_graphs, int* nums_of_nodes, int* nums_of_edges, int* reload_weights, float* out,
                          int* node_feature_in, int* edge_list_in, int* edge_attr_in, float* node_embedding_weight_in,
                          float* edge_embedding_weight_in, float* node_mlp_1_weights, float* node_mlp_1_bias,
                          float* node_mlp_2_weights, float* node_mlp_2_bias, float* graph_pred_weights_in,
                          float* graph_pred_bias_in, int num_tasks) {
    const float* t[8] = {node_embedding_weight_in, edge_embedding_weight_in, node_mlp_1_weights, node_mlp_1_bias,
                         node_mlp_2_weights,       node_mlp_2_bias,          graph_pred_weights_in, graph_pred_bias_in};
    if (num_tasks < 1) return FLOWGNN_ERR_ARG;
    const int T = num_tasks;
    const size_t sz[8] = {173 * 100, 5 * 13 * 100, 5 * 200 * 100, 5 * 200, 5 * 100 * 200, 5 * 100, (size_t)T * 100, (size_t)T};
    return compute_graphs_generic(FLOWGNN_MODEL_GIN, num_graphs, nums_of_nodes, nums_of_edges, reload_weights, out,
                                  node_feature_in, nullptr, edge_list_in, edge_attr_in, 8, t, sz, T);
}

// The reference's symbol is `void`: a caller that ignores the status must not read an untouched buffer as results, so a refusal also
// fills `out` with NaN and says why on stderr (once per process).  The environment is asked on every call (getenv is cheap), so
// unsetting the variable in the same process clears the refusal.
static int refuse_stale_num_task(const char* symbol, float* out, int num_graphs) {
    bool stale = false;
    read_environment(nullptr, nullptr, &stale);
    if (!stale) return FLOWGNN_OK;
    char msg[256];
    snprintf(msg, sizeof(msg), "%s: FLOWGNN_NUM_TASK is set in the environment but no longer read -- call %s_mt(..., num_tasks) (include/flowgnn.h) or unset it", symbol, symbol);
    fg::set_last_error(msg);
    static std::atomic<bool> said{false};
    if (!said.exchange(true)) fprintf(stderr, "flowgnn: %s; the output buffer is filled with NaN\n", msg);
    if (out)
        for (int g = 0; g < num_graphs; g++) out[g] = std::numeric_limits<float>::quiet_NaN();
    return FLOWGNN_ERR_UNSUPPORTED;
}

int GIN_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges, int* reload_weights, float* out,
                       int* node_feature_in, int* edge_list_in, int* edge_attr_in, float* node_embedding_weight_in,
                       float* edge_embedding_weight_in, float* node_mlp_1_weights, float* node_mlp_1_bias,
                       float* node_mlp_2_weights, float* node_mlp_2_bias, float* graph_pred_weights_in,
                       float* graph_pred_bias_in) {
    if (int rc = refuse_stale_num_task("GIN_compute_graphs", out, num_graphs)) return rc;
    return GIN_compute_graphs_mt(num_graphs, nums_of_nodes, nums_of_edges, reload_weights, out, node_feature_in, edge_list_in,
                                 edge_attr_in, node_embedding_weight_in, edge_embedding_weight_in, node_mlp_1_weights, node_mlp_1_bias,
                                 node_mlp_2_weights, node_mlp_2_bias, graph_pred_weights_in, graph_pred_bias_in, 1);
}

int GCN_compute_graphs_mt(int num_graphs, int* nums_of_nodes, int* nums_of_edges, int* reload_weights, float* out,
                          int* node_feature_in, int* edge_list_in, int* edge_attr_in, float* node_embedding_weight_in,
                          float* edge_embedding_weight_in, float* convs_weight_in, float* convs_bias_in,
                          float* convs_root_emb_weight_in, float* bn_weight_in, float* bn_bias_in, float* bn_mean_in,
                          float* bn_var_in, float* graph_pred_weights_in, float* graph_pred_bias_in, int num_tasks) {
    if (num_tasks < 1) return FLOWGNN_ERR_ARG;
    const float* t[11] = {node_embedding_weight_in, edge_embedding_weight_in, convs_weight_in, convs_bias_in,
                          convs_root_emb_weight_in, bn_weight_in, bn_bias_in, bn_mean_in, bn_var_in,
                          graph_pred_weights_in, graph_pred_bias_in};
    const int T = num_tasks;
    const size_t sz[11] = {173 * 100, 5 * 13 * 100, 5 * 100 * 100, 500, 500, 500, 500, 500, 500, (size_t)T * 100, (size_t)T};
    return compute_graphs_generic(FLOWGNN_MODEL_GCN, num_graphs, nums_of_nodes, nums_of_edges, reload_weights, out,
                                  node_feature_in, nullptr, edge_list_in, edge_attr_in, 11, t, sz, T);
}

int GCN_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges, int* reload_weights, float* out,
                       int* node_feature_in, int* edge_list_in, int* edge_attr_in, float* node_embedding_weight_in,
                       float* edge_embedding_weight_in, float* convs_weight_in, float* convs_bias_in,
                       float* convs_root_emb_weight_in, float* bn_weight_in, float* bn_bias_in, float* bn_mean_in,
                       float* bn_var_in, float* graph_pred_weights_in, float* graph_pred_bias_in) {
    if (int rc = refuse_stale_num_task("GCN_compute_graphs", out, num_graphs)) return rc;
    return GCN_compute_graphs_mt(num_graphs, nums_of_nodes, nums_of_edges, reload_weights, out, node_feature_in, edge_list_in,
                                 edge_attr_in, node_embedding_weight_in, edge_embedding_weight_in, convs_weight_in, convs_bias_in,
                                 convs_root_emb_weight_in, bn_weight_in, bn_bias_in, bn_mean_in, bn_var_in, graph_pred_weights_in,
                                 graph_pred_bias_in, 1);
}

int PNA_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges, int* reload_weights, float* out,
                       int* node_feature_in, int* edge_list_in, float* node_embedding_weight_in,
                       float* node_conv_weights_in, float* node_conv_bias_in, float* graph_mlp_1_weights_in,
                       float* graph_mlp_1_bias_in, float* graph_mlp_2_weights_in, float* graph_mlp_2_bias_in,
                       float* graph_mlp_3_weights_in, float* graph_mlp_3_bias_in, float* avg_deg_in) {
    const float* t[10] = {node_embedding_weight_in, node_conv_weights_in, node_conv_bias_in, graph_mlp_1_weights_in,
                          graph_mlp_1_bias_in, graph_mlp_2_weights_in, graph_mlp_2_bias_in, graph_mlp_3_weights_in,
                          graph_mlp_3_bias_in, avg_deg_in};
    static const size_t sz[10] = {173 * 80, 4 * 80 * 3 * 4 * 80, 4 * 80, 40 * 80, 40, 20 * 40, 20, 20, 1, 1};
    return compute_graphs_generic(FLOWGNN_MODEL_PNA, num_graphs, nums_of_nodes, nums_of_edges, reload_weights, out,
                                  node_feature_in, nullptr, edge_list_in, nullptr, 10, t, sz);
}

int DGN_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges, int* reload_weights, float* out,
                       int* node_feature_in, float* node_eigen_in, int* edge_list_in,
                       float* embedding_h_atom_embedding_list_weights_in,
                       float* layers_posttrans_fully_connected_0_linear_weight_in,
                       float* layers_posttrans_fully_connected_0_linear_bias_in, float* MLP_layer_FC_layers_0_weight_in,
                       float* MLP_layer_FC_layers_0_bias_in, float* MLP_layer_FC_layers_1_weight_in,
                       float* MLP_layer_FC_layers_1_bias_in, float* MLP_layer_FC_layers_2_weight_in,
                       float* MLP_layer_FC_layers_2_bias_in) {
    const float* t[9] = {embedding_h_atom_embedding_list_weights_in,
                         layers_posttrans_fully_connected_0_linear_weight_in,
                         layers_posttrans_fully_connected_0_linear_bias_in,
                         MLP_layer_FC_layers_0_weight_in, MLP_layer_FC_layers_0_bias_in, MLP_layer_FC_layers_1_weight_in,
                         MLP_layer_FC_layers_1_bias_in, MLP_layer_FC_layers_2_weight_in, MLP_layer_FC_layers_2_bias_in};
    static const size_t sz[9] = {9 * 119 * 100, 4 * 100 * 200, 4 * 100, 50 * 100, 50, 25 * 50, 25, 25, 1};
    if (num_graphs > 0 && !node_eigen_in) return FLOWGNN_ERR_ARG;
    return compute_graphs_generic(FLOWGNN_MODEL_DGN, num_graphs, nums_of_nodes, nums_of_edges, reload_weights, out,
                                  node_feature_in, node_eigen_in, edge_list_in, nullptr, 9, t, sz);
}

int GAT_compute_graphs(int num_graphs, int* nums_of_nodes, int* nums_of_edges, int* reload_weights, float* out,
                       int* node_feature_in, int* edge_list_in, float* scoring_fn_target_in, float* scoring_fn_source_in,
                       float* linear_proj_weights_in, float* skip_proj_weights_in, float* graph_pred_weights_in,
                       float* graph_pred_bias_in) {
    const float* t[6] = {scoring_fn_target_in, scoring_fn_source_in, linear_proj_weights_in, skip_proj_weights_in,
                         graph_pred_weights_in, graph_pred_bias_in};
    static const size_t sz[6] = {5 * 4 * 16, 5 * 4 * 16, 5 * 4 * 16 * 4 * 16, 5 * 4 * 16 * 4 * 16, 16, 1};
    return compute_graphs_generic(FLOWGNN_MODEL_GAT, num_graphs, nums_of_nodes, nums_of_edges, reload_weights, out,
                                  node_feature_in, nullptr, edge_list_in, nullptr, 6, t, sz);
}

}  // extern "C"
