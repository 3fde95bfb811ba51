// host_main.cpp -- command-line host for the MI355X engine, the counterpart of the reference's per-model
// OpenCL host (GIN/src/host.cc): load the model's .bin weights, read a graph pack in the reference's on-disk
// layout, run the whole dataset as ONE batched launch NUM_TRIALS times, write HLS_output.txt.
//
//   host <MODEL> [--graphs DIR] [--weights DIR] [--num-graphs N] [--trials T] [--out FILE] [--device D | --devices D0,D1,..] [--option key=value] [--numeric f32|q6.10] [XCLBIN]
//
//   MODEL        GIN | GIN-VN | GCN | GAT | PNA | DGN
//   --graphs     directory holding graph_info/ and graph_bin/      (default ../graphs, host.cc:14-15)
//   --weights    directory holding the model's .bin files          (default ., host_load.cc:24)
//   --num-graphs graph count; default = <graphs>/dataset_size.txt, else ../common/includes/dataset/dataset_size.txt
//                (the reference compiles it in: common/includes/dataset/dataset.hpp)
//   --trials     launches to time                                   (default 25 = NUM_TRIALS, host.h:8)
//   --out        result file                                        (default HLS_output.txt, host.cc:213)
//   XCLBIN       accepted and ignored, so `./host <xclbin>`-style command lines keep working
//
// Only the C ABI of include/flowgnn.h is used; no HIP or torch types here.
#include "../../include/flowgnn.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int model_id(const std::string& m) {
    if (m == "GIN") return FLOWGNN_MODEL_GIN;
    if (m == "GIN-VN" || m == "GIN_VN") return FLOWGNN_MODEL_GIN_VN;
    if (m == "GCN") return FLOWGNN_MODEL_GCN;
    if (m == "GAT") return FLOWGNN_MODEL_GAT;
    if (m == "PNA") return FLOWGNN_MODEL_PNA;
    if (m == "DGN") return FLOWGNN_MODEL_DGN;
    return -1;
}

template <typename T>
static bool read_exact(const std::string& path, std::vector<T>& dst, size_t start, size_t count) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = fread(dst.data() + start, sizeof(T), count, f) == count;
    fclose(f);
    return ok;
}

static long read_count_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    long n = -1;
    if (fscanf(f, "%ld", &n) != 1) n = -1;
    fclose(f);
    return n;
}

// DGN/src/host_load.cc:201-215: "tensor([[a, b,c,d],\n [..]])" -- take the numbers in order, 4 per node
static bool read_eig_txt(const std::string& path, std::vector<float>& eig, size_t start_node, int n) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    std::string txt;
    char buf[4096];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) txt.append(buf, got);
    fclose(f);
    const char* p = txt.c_str();
    size_t k = 0;
    while (*p && k < (size_t)n * 4) {
        if ((*p >= '0' && *p <= '9') || ((*p == '-' || *p == '+' || *p == '.') && p[1] >= '0' && p[1] <= '9')) {
            char* end;
            eig[start_node * 4 + k++] = strtof(p, &end);
            p = end;
        } else {
            p++;
        }
    }
    return k == (size_t)n * 4;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "Usage: %s <GIN|GIN-VN|GCN|GAT|PNA|DGN> [--graphs DIR] [--weights DIR] [--num-graphs N] [--trials T] "
                        "[--out FILE] [--device D | --devices D0,D1,..] [--option key=value] [--numeric f32|q6.10] [--num-tasks T] [XCLBIN File]\n", argv[0]);
        return EXIT_FAILURE;
    }
    const std::string model = argv[1];
    const int mid = model_id(model);
    if (mid < 0) { fprintf(stderr, "unknown model %s\n", model.c_str()); return EXIT_FAILURE; }
    std::string graphs = "../graphs", wdir = ".", out_path = "HLS_output.txt", eig_dir = "eig";
    long num_graphs = -1;
    int trials = 25, numeric = FLOWGNN_NUMERIC_F32, num_tasks = 1;
    std::vector<int> devices;
    std::vector<std::pair<std::string, double>> options;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(EXIT_FAILURE); }
            return argv[++i];
        };
        if (a == "--graphs") graphs = next("--graphs");
        else if (a == "--weights") wdir = next("--weights");
        else if (a == "--eig") eig_dir = next("--eig");
        else if (a == "--num-graphs") num_graphs = atol(next("--num-graphs"));
        else if (a == "--trials") trials = atoi(next("--trials"));
        else if (a == "--out") out_path = next("--out");
        else if (a == "--device") { devices.clear(); devices.push_back(atoi(next("--device"))); }
        else if (a == "--devices") {  // e.g. 0,1,2,3,4,5,6,7: the batch is cut by sum(N + E), one engine + host thread per device
            devices.clear();
            for (const char* c = next("--devices"); *c;) {
                devices.push_back(atoi(c));
                while (*c && *c != ',') c++;
                if (*c == ',') c++;
            }
        }
        else if (a == "--option") {  // key=value, flowgnn_set_option (e.g. gin_resident=0)
            const std::string kv = next("--option");
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) { fprintf(stderr, "--option wants key=value\n"); return EXIT_FAILURE; }
            options.emplace_back(kv.substr(0, eq), atof(kv.c_str() + eq + 1));
        }
        else if (a == "--num-tasks") num_tasks = atoi(next("--num-tasks"));  // NUM_TASK of the readout (GIN/src/dcl.h:25), GIN / GIN-VN / GCN
        else if (a == "--numeric") numeric = std::string(next("--numeric")) == "q6.10" ? FLOWGNN_NUMERIC_Q6_10 : FLOWGNN_NUMERIC_F32;
        // anything else (e.g. an .xclbin path) is ignored
    }
    if (num_graphs < 0) num_graphs = read_count_file(graphs + "/dataset_size.txt");
    if (num_graphs < 0) num_graphs = read_count_file("../common/includes/dataset/dataset_size.txt");
    if (num_graphs < 0) { fprintf(stderr, "graph count unknown: pass --num-graphs or provide dataset_size.txt\n"); return EXIT_FAILURE; }

    printf("\n******* This is the MI355X host for the %s model *******\n", model.c_str());
    if (devices.empty()) devices.push_back(0);
    flowgnn_group* eng = nullptr;  // one engine per listed device behind one handle (a single device is the common case)
    int rc = flowgnn_create_multi(mid, (int)devices.size(), devices.data(), &eng);
    if (rc) { fprintf(stderr, "flowgnn_create_multi failed: %d %s\n", rc, flowgnn_group_last_error(nullptr)); return EXIT_FAILURE; }
    for (auto& kv : options) {
        rc = flowgnn_group_set_option(eng, kv.first.c_str(), kv.second);
        if (rc) { fprintf(stderr, "--option %s: %d %s\n", kv.first.c_str(), rc, flowgnn_group_last_error(eng)); return EXIT_FAILURE; }
    }
    if (num_tasks != 1) {
        rc = flowgnn_group_set_num_tasks(eng, num_tasks);
        if (rc) { fprintf(stderr, "--num-tasks %d: %d %s\n", num_tasks, rc, flowgnn_group_last_error(eng)); return EXIT_FAILURE; }
    }
    rc = flowgnn_group_load_weights_dir(eng, wdir.c_str());
    if (rc) { fprintf(stderr, "loading weights failed: %d %s\n", rc, flowgnn_group_last_error(eng)); return EXIT_FAILURE; }
    if (numeric != FLOWGNN_NUMERIC_F32) {  // the reference's ap_fixed<16,6> bit patterns (GIN / GIN-VN)
        rc = flowgnn_group_set_numeric_mode(eng, numeric);
        if (rc) { fprintf(stderr, "numeric mode: %d %s\n", rc, flowgnn_group_last_error(eng)); return EXIT_FAILURE; }
    }
    printf("\n******* Weights loading done *******\n");

    const bool vn = mid == FLOWGNN_MODEL_GIN_VN, dgn = mid == FLOWGNN_MODEL_DGN;
    std::vector<int> nn(num_graphs), ne(num_graphs), nf, el, ea;
    std::vector<float> eig;
    size_t N = 0, E = 0;
    for (long g = 1; g <= num_graphs; g++) {
        char info[512];
        snprintf(info, sizeof(info), "%s/graph_info/g%ld_info.txt", graphs.c_str(), g);
        FILE* f = fopen(info, "r");
        int n = 0, e = 0;
        if (!f || fscanf(f, "%d\n%d", &n, &e) != 2) { fprintf(stderr, "cannot read %s\n", info); return EXIT_FAILURE; }
        fclose(f);
        char base[512];
        snprintf(base, sizeof(base), "%s/graph_bin/g%ld", graphs.c_str(), g);
        const int n2 = vn ? n + 1 : n, e2 = vn ? e + 2 * n : e;  // GIN-VN/src/host.cc:133-134
        nf.resize((N + n2) * 9, 0);
        el.resize((E + e2) * 2, 0);
        ea.resize((E + e2) * 3, 0);
        if (!read_exact(std::string(base) + "_node_feature.bin", nf, N * 9, (size_t)n * 9) ||
            !read_exact(std::string(base) + "_edge_list.bin", el, E * 2, (size_t)e * 2)) {
            fprintf(stderr, "cannot read graph %ld under %s\n", g, base);
            return EXIT_FAILURE;
        }
        read_exact(std::string(base) + "_edge_attr.bin", ea, E * 3, (size_t)e * 3);  // absent for some packs: zeros
        if (vn) {  // GIN-VN/src/host_load.cc:125-153: virtual node N, edges (nd, N) and (N, nd), attributes 0
            for (int nd = 0; nd < n; nd++) {
                el[(E + e + 2 * nd) * 2 + 0] = nd; el[(E + e + 2 * nd) * 2 + 1] = n;
                el[(E + e + 2 * nd + 1) * 2 + 0] = n; el[(E + e + 2 * nd + 1) * 2 + 1] = nd;
            }
        }
        if (dgn) {
            eig.resize((N + n2) * 4, 0.0f);
            char ep[512];
            snprintf(ep, sizeof(ep), "%s/g%ld.txt", eig_dir.c_str(), g);
            if (!read_eig_txt(ep, eig, N, n)) { fprintf(stderr, "cannot read %s\n", ep); return EXIT_FAILURE; }
        }
        nn[g - 1] = n2;
        ne[g - 1] = e2;
        N += n2;
        E += e2;
        if (g % 1000 == 0 || g == num_graphs) { printf("(%ld/%ld) Loading graphs ...\r", g, num_graphs); fflush(stdout); }
    }
    printf("\n******* Graphs loading done *******\n");

    rc = flowgnn_group_set_batch(eng, (int)num_graphs, nn.data(), ne.data(), nf.data(), el.data(), ea.data(), dgn ? eig.data() : nullptr);
    if (rc) { fprintf(stderr, "flowgnn_set_batch failed: %d %s\n", rc, flowgnn_group_last_error(eng)); return EXIT_FAILURE; }
    rc = flowgnn_group_run(eng);  // warm-up (first-touch, code load)
    if (!rc) rc = flowgnn_group_sync(eng);
    if (rc) { fprintf(stderr, "run failed: %d %s\n", rc, flowgnn_group_last_error(eng)); return EXIT_FAILURE; }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < trials; i++) {
        printf("(%d/%d) Computing %s ...\r", i + 1, trials, model.c_str());
        fflush(stdout);
        rc = flowgnn_group_run(eng);
        if (rc) break;
    }
    if (!rc) rc = flowgnn_group_sync(eng);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc) { fprintf(stderr, "run failed: %d %s\n", rc, flowgnn_group_last_error(eng)); return EXIT_FAILURE; }
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / (trials > 0 ? trials : 1);
    printf("\n******* Computation done *******\n");
    // the figure run_experiments.sh derives: kernel ms for the whole dataset / graphs (run_experiments.sh:44-46)
    printf("%s: %.6f ms per launch, %.6f ms per graph, %.1f graphs/s (%ld graphs, %zu nodes, %zu edges)\n", model.c_str(), ms,
           ms / num_graphs, num_graphs / (ms * 1e-3), num_graphs, N, E);

    std::vector<float> result((size_t)num_graphs * num_tasks);
    rc = flowgnn_group_get_results(eng, result.data());
    if (rc) { fprintf(stderr, "flowgnn_get_results failed: %d\n", rc); return EXIT_FAILURE; }
    FILE* o = fopen(out_path.c_str(), "w+");
    if (!o) { fprintf(stderr, "cannot write %s\n", out_path.c_str()); return EXIT_FAILURE; }
    for (long g = 1; g <= num_graphs; g++)  // one line per task, as host.cc:213-222
        for (int t = 0; t < num_tasks; t++) fprintf(o, "g%ld: %.8f\n", g, result[(size_t)(g - 1) * num_tasks + t]);
    fclose(o);
    flowgnn_group_destroy(eng);
    return 0;
}
