// h2d_pack.cpp -- host side of the packed host -> device transfer of flowgnn_set_batch (engine.hip).
//
// The reference's host hands the kernel int32 arrays (GIN/src/host.cc:119-138: 36 B of features per node, 8 B of endpoints and 12 B of
// attributes per edge) whose VALUES fit a byte (features < 119, attribute triples < 60 combinations) or 16 bits (node ids inside a
// graph).  Over PCIe those 536 MB per 2^18 molhiv graphs are what the drop-in symbols wait for (12.6 ms against 8.0 ms of kernels);
// narrowed on the host -- 9 B per node, 5 B per edge: 134 MB -- and widened again on the GPU (unpack_batch_kernel: the kernels keep
// reading the reference's int32 layout) the copy is a quarter of that.  Values that do not fit travel as the sentinel 255 / 65 535 and
// arrive as -1, which the device-side validation refuses with the same status code as the original value.
// Compiled by g++ (not hipcc) with function clones per ISA level: the loops are byte shuffles, memory-bound once they vectorise.
#include <cstddef>
#include <cstdint>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace fg {

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define FG_CLONES __attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))
#else
#define FG_CLONES
#endif

FG_CLONES static void narrow_u8(const int* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const unsigned v = (unsigned)in[i];
        out[i] = (uint8_t)(v < 255u ? v : 255u);
    }
}

FG_CLONES static void narrow_u16(const int* __restrict__ in, uint16_t* __restrict__ out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const unsigned v = (unsigned)in[i];
        out[i] = (uint16_t)(v < 65535u ? v : 65535u);
    }
}

// (attr0, attr1, attr2) -> (attr0 * 6 + attr1) * 2 + attr2, the edge code of graph_build.hip; 255 if any of them is outside its table
FG_CLONES static void narrow_attr(const int* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const unsigned a0 = (unsigned)in[3 * i], a1 = (unsigned)in[3 * i + 1], a2 = (unsigned)in[3 * i + 2];
        const bool ok = (a0 < 5u) & (a1 < 6u) & (a2 < 2u);
        out[i] = (uint8_t)(ok ? (a0 * 6u + a1) * 2u + a2 : 255u);
    }
}

size_t h2d_pack_bytes(size_t n_nodes, size_t n_edges, bool attr, size_t* off_edges, size_t* off_attr) {
    const size_t a = (n_nodes * 9 + 15) & ~(size_t)15;
    const size_t b = a + ((n_edges * 4 + 15) & ~(size_t)15);
    if (off_edges) *off_edges = a;
    if (off_attr) *off_attr = b;
    return b + (attr ? ((n_edges + 15) & ~(size_t)15) : 0);
}

// A process-wide pool of packing threads: a drop-in call cuts its batch into a dozen ranges, and starting sixteen std::threads per
// range (50 us each) cost more than the packing they did.  One parallel loop at a time (two engines of one device take turns: they
// would share the same cores anyway); workers spin briefly for the next loop, then sleep.
namespace {
class PackPool {
public:
    ~PackPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
            gen_++;
        }
        cv_go_.notify_all();
        for (auto& t : th_) t.join();
    }
    // fn(t) for t = 0 .. parts - 1, part 0 on the caller's thread
    void run(int parts, const std::function<void(int)>& fn) {
        std::lock_guard<std::mutex> one(call_mu_);
        while ((int)th_.size() < parts - 1) {
            const int id = (int)th_.size() + 1;
            th_.emplace_back([this, id] { loop(id); });
        }
        if (parts > 1) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                fn_ = &fn;
                parts_ = parts;
                pending_ = parts - 1;
                gen_++;
            }
            cv_go_.notify_all();
        }
        fn(0);
        if (parts > 1) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [this] { return pending_ == 0; });
            fn_ = nullptr;
        }
    }

private:
    void loop(int id) {
        unsigned long long seen = 0;
        while (true) {
            const std::function<void(int)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_go_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (quit_) return;
                if (id < parts_) fn = fn_;
            }
            if (!fn) continue;  // (a loop of fewer parts than there are workers)
            (*fn)(id);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) cv_done_.notify_one();
        }
    }
    std::mutex call_mu_, mu_;
    std::condition_variable cv_go_, cv_done_;
    std::vector<std::thread> th_;
    const std::function<void(int)>* fn_ = nullptr;
    int parts_ = 0, pending_ = 0;
    unsigned long long gen_ = 0;
    bool quit_ = false;
};
PackPool& pack_pool() {
    static PackPool* p = new PackPool();  // (never destroyed: its workers must not be joined from a static destructor after fork / at exit)
    return *p;
}
}  // namespace

// fn(t) for t = 0 .. parts - 1 on the pool (part 0 on the caller's thread): flowgnn_set_batch's other host loop, the bin packing of
// graph tiles, runs on it too
void host_parallel_for(int parts, const std::function<void(int)>& fn) {
    if (parts <= 1) { fn(0); return; }
    pack_pool().run(parts, fn);
}

// node_feature [N][9], edge_list [E][2], edge_attr [E][3] or null -> dst (h2d_pack_bytes bytes), on up to `threads` host threads (one
// per ~2 MB of input at least: a small range is not worth waking sixteen workers for)
void h2d_pack(const int* node_feature, const int* edge_list, const int* edge_attr, size_t n_nodes, size_t n_edges, uint8_t* dst, int threads) {
    size_t off_e = 0, off_a = 0;
    h2d_pack_bytes(n_nodes, n_edges, edge_attr != nullptr, &off_e, &off_a);
    const size_t in_bytes = 4 * (n_nodes * 9 + n_edges * (edge_attr ? 5 : 2));
    const int by_size = (int)(in_bytes / ((size_t)2 << 20)) + 1;
    if (threads > by_size) threads = by_size;
    if (threads < 1) threads = 1;
    const std::function<void(int)> part = [&](int t) {
        const size_t n0 = n_nodes * 9 * (size_t)t / threads, n1 = n_nodes * 9 * (size_t)(t + 1) / threads;
        narrow_u8(node_feature + n0, dst + n0, n1 - n0);
        const size_t e0 = n_edges * (size_t)t / threads, e1 = n_edges * (size_t)(t + 1) / threads;
        narrow_u16(edge_list + 2 * e0, reinterpret_cast<uint16_t*>(dst + off_e) + 2 * e0, 2 * (e1 - e0));
        if (edge_attr) narrow_attr(edge_attr + 3 * e0, dst + off_a + e0, e1 - e0);
    };
    pack_pool().run(threads, part);
}

}  // namespace fg
