// graph_build.hip -- batched load_graph for the whole super-graph, on the GPU.
//
// Replaces the reference's per-graph, serial load_graph (GIN/src/load_inputs.cc:87-172):
// there, an edge list is bucketed into four per-PE tables ordered by source id (stable in
// input order), and the scatter units replay them, so the messages of a destination v
// arrive in ascending source id, ties in input order.  Here the same ordering is produced
// for ALL graphs of the batch at once as one destination-major CSR over global node ids:
//
//   row v = { input edges (u -> v) } sorted by (u, input index)
//
// Integer work only; tests compare it bit-exactly with the oracle's tables.
//
// Kernels (all flat over edges / nodes, no per-graph size cap; the rank sort is quadratic in a row's in-degree, so the flat
// path refuses rows above MAX_FLAT_INDEGREE = 16 384 in-edges, three times the reference's per-GRAPH edge cap):
//   globalize_count : wave per graph; local -> global ids, in-degree / out-degree histograms,
//                     range validation of endpoints and edge attributes
//   exclusive scan  : in-degree -> row_ptr
//   place           : arbitrary-order placement with a per-row cursor
//   rank            : per-row rank sort by (u, input index) -> deterministic final order
#include "common.h"

namespace fg {

// ------------------------------------------------------------------ scan
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ inline int wave_inclusive_scan(int x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// out[i] = exclusive scan within the tile; block_sums[b] = tile total
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_kernel(const int* __restrict__ in, int* __restrict__ out,
                                                                  int* __restrict__ block_sums, int n) {
    __shared__ int wave_tot[SCAN_THREADS / 64];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        sum += v[i];
    }
    int incl = wave_inclusive_scan(sum);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int w = 0; w < wave; w++) wave_base += wave_tot[w];
    int run = wave_base + incl - sum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == SCAN_THREADS - 1) block_sums[blockIdx.x] = run;
}

// single block: exclusive scan of block_sums in place; writes grand total to *total
__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(int* __restrict__ block_sums, int nblocks,
                                                                  int* __restrict__ total) {
    __shared__ int wave_tot[SCAN_THREADS / 64];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += SCAN_THREADS) {
        int i = base + threadIdx.x;
        int x = (i < nblocks) ? block_sums[i] : 0;
        int incl = wave_inclusive_scan(x);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int wave_base = 0;
        for (int w = 0; w < wave; w++) wave_base += wave_tot[w];
        int carry = carry_s;
        if (i < nblocks) block_sums[i] = carry + wave_base + incl - x;
        __syncthreads();
        if (threadIdx.x == SCAN_THREADS - 1) carry_s = carry + wave_base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(int* __restrict__ out, const int* __restrict__ block_sums,
                                                                 int n) {
    const int off = block_sums[blockIdx.x];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) out[base + i] += off;
}

// in-place exclusive scan of data[0..n), total written to data[n]
static void exclusive_scan_inplace(int* data, int n, int* block_sums, hipStream_t s) {
    if (n <= 0) {
        (void)hipMemsetAsync(data, 0, sizeof(int), s);  // (errors of the async calls surface in flowgnn_run's hipGetLastError)
        return;
    }
    int nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    scan_tile_kernel<<<nblocks, SCAN_THREADS, 0, s>>>(data, data, block_sums, n);
    scan_sums_kernel<<<1, SCAN_THREADS, 0, s>>>(block_sums, nblocks, data + n);
    if (nblocks > 1) scan_add_kernel<<<nblocks, SCAN_THREADS, 0, s>>>(data, block_sums, n);
}

// ------------------------------------------------------------------ globalize + histograms
// One wavefront per graph: its edges are a contiguous slice of the edge list.
__global__ __launch_bounds__(256) void globalize_count_kernel(BatchView b, CsrView c, bool has_attr) {
    const int wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wave_in_block;
    if (g >= b.num_graphs) return;
    const int n = b.nums_of_nodes[g];
    const int noff = b.node_off[g];
    const int e0 = b.edge_off[g], e1 = b.edge_off[g + 1];
    for (int e = e0 + lane; e < e1; e += 64) {
        const int2 uv = reinterpret_cast<const int2*>(b.edge_list)[e];
        int u = uv.x, v = uv.y;
        const bool ok = (u >= 0) & (u < n) & (v >= 0) & (v < n);
        if (!ok) {  // flag it, then treat as a self-loop on node 0 so every later index stays in range
            atomicMax(c.err, ERR_EDGE_RANGE);
            u = 0;
            v = 0;
        }
        c.gsrc[e] = noff + u;
        c.gdst[e] = noff + v;
        atomicAdd(&c.row_ptr[noff + v], 1);   // in-degree histogram (scanned into row_ptr later)
        atomicAdd(&c.out_deg[noff + u], 1);   // reference degree_table[u] (load_inputs.cc:128)
        if (has_attr) {
            const int a0 = b.edge_attr[3 * (size_t)e], a1 = b.edge_attr[3 * (size_t)e + 1],
                      a2 = b.edge_attr[3 * (size_t)e + 2];
            // table cardinalities {5,6,2}: GIN/src/host_load.cc:6
            bool aok = (a0 >= 0) & (a0 < 5) & (a1 >= 0) & (a1 < 6) & (a2 >= 0) & (a2 < 2);
            if (!aok) atomicMax(c.err, ERR_EDGE_ATTR);
            c.tmp[e] = aok ? (a0 * 6 + a1) * 2 + a2 : 0;  // staged per input edge; permuted in rank
        }
    }
}

// arbitrary-order placement: slot = row_ptr[v] + cursor[v]++
__global__ __launch_bounds__(256) void place_kernel(CsrView c, int e_tot, int* __restrict__ slot_edge) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= e_tot) return;
    const int v = c.gdst[e];
    const int slot = c.row_ptr[v] + atomicAdd(&c.cursor[v], 1);
    slot_edge[slot] = e;
}

// rank sort inside each row by (source id, input index): unique keys => deterministic order
__global__ __launch_bounds__(256) void rank_kernel(CsrView c, int e_tot, const int* __restrict__ slot_edge,
                                                    const int* __restrict__ code_by_edge) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= e_tot) return;
    const int e = slot_edge[s];
    const int v = c.gdst[e], u = c.gsrc[e];
    const int beg = c.row_ptr[v], end = c.row_ptr[v + 1];
    if (end - beg > MAX_FLAT_INDEGREE) {  // quadratic beyond use (common.h): refuse, visibly (flowgnn_sync returns the flag)
        atomicMax(c.err, ERR_UNSUPPORTED);
        c.src[s] = u;  // unsorted but in range: the forward pass that is already queued behind this kernel must not read stale indices
        c.eid[s] = e;
        if (code_by_edge) c.ecode[s] = (uint8_t)code_by_edge[e];
        return;
    }
    int rank = 0;
    for (int t = beg; t < end; t++) {
        const int e2 = slot_edge[t];
        const int u2 = c.gsrc[e2];
        rank += (u2 < u) | ((u2 == u) & (e2 < e));
    }
    c.src[beg + rank] = u;
    c.eid[beg + rank] = e;
    if (code_by_edge) c.ecode[beg + rank] = (uint8_t)code_by_edge[e];
}

// ------------------------------------------------------------------ per-graph build in LDS
// A graph's edges are a contiguous slice of the edge list and its CSR rows a contiguous slice of the CSR, so one
// workgroup can build a whole graph's rows in LDS with no global atomics and no global scratch: histogram, scan,
// cursor placement, rank sort -- the same four steps as above, on LDS copies of the endpoints.  Global traffic is one
// read of the edge list / attributes and one write of row_ptr / src / eid / ecode / out_deg.  Used when every graph of
// the batch fits the class (the reference itself caps graphs at 500 nodes / 5500 edges, GIN/src/dcl.h:17-18);
// otherwise the flat global path above runs.  Same output, bit for bit (the keys are unique).
// LDS is sized at launch by the largest graph of the BATCH (nmax nodes, emax edges, both within the class limits NMAX /
// EMAX): a molhiv batch (largest graph 183 nodes / 378 edges) then needs 5 KB per graph instead of the class's 10 KB, and
// twice as many graphs are in flight per CU.
// DYN (the single-wave class): LDS is sized at launch by the largest graph of the BATCH (nmax nodes, emax edges, within the
// class limits) -- a molhiv batch (largest graph 183 nodes / 378 edges) then needs 5 KB per graph instead of the class's
// 10 KB and twice as many graphs are in flight per CU (0.32 -> 0.19 ms).  The multi-wave classes keep static arrays of the
// class size: sized by the batch they measured slower on the kNN graphs (0.23 -> 0.26 ms).
template <int NT, int NMAX, int EMAX, typename IdxT, bool DYN>
__global__ __launch_bounds__(NT) void build_csr_graph_kernel(BatchView b, CsrView c, bool has_attr, int nmax, int emax, const int* only_if) {
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    if (only_if != nullptr && *only_if == 0) return;  // a build that the launching model needs only when a device-side flag says so
    __shared__ int st_cnt[DYN ? 1 : NMAX + 1];
    __shared__ int st_cur[DYN ? 1 : NMAX];
    __shared__ int st_odeg[DYN ? 1 : NMAX];
    __shared__ int st_wtot[DYN ? 1 : NT / 64];
    __shared__ IdxT st_u[DYN ? 1 : EMAX], st_v[DYN ? 1 : EMAX], st_slot[DYN ? 1 : EMAX];
    __shared__ uint8_t st_code[DYN ? 1 : EMAX];
    int* s_cnt = DYN ? reinterpret_cast<int*>(s_dyn) : st_cnt;  // [nmax + 1] in-degree, then exclusive scan = row start
    int* s_cur = DYN ? s_cnt + (nmax + 1) : st_cur;             // [nmax]
    int* s_odeg = DYN ? s_cur + nmax : st_odeg;                 // [nmax]
    int* s_wtot = DYN ? s_odeg + nmax : st_wtot;                // [NT / 64]
    IdxT* s_u = DYN ? reinterpret_cast<IdxT*>(s_wtot + NT / 64) : st_u;  // [emax] each
    IdxT* s_v = DYN ? s_u + emax : st_v;
    IdxT* s_slot = DYN ? s_v + emax : st_slot;
    uint8_t* s_code = DYN ? reinterpret_cast<uint8_t*>(s_slot + emax) : st_code;
    const int g = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = b.nums_of_nodes[g];
    const int noff = b.node_off[g];
    const int e0 = b.edge_off[g];
    const int ne = b.edge_off[g + 1] - e0;
    for (int i = tid; i <= n; i += NT) {
        s_cnt[i] = 0;
        if (i < n) { s_cur[i] = 0; s_odeg[i] = 0; }
    }
    __syncthreads();
    for (int e = tid; e < ne; e += NT) {
        const int2 uv = reinterpret_cast<const int2*>(b.edge_list)[e0 + e];
        int u = uv.x, v = uv.y;
        if (!((u >= 0) & (u < n) & (v >= 0) & (v < n))) {
            atomicMax(c.err, ERR_EDGE_RANGE);
            u = 0;
            v = 0;
        }
        s_u[e] = (IdxT)u;
        s_v[e] = (IdxT)v;
        atomicAdd(&s_cnt[v], 1);
        atomicAdd(&s_odeg[u], 1);
        int code = 0;
        if (has_attr) {
            const int a0 = b.edge_attr[3 * (size_t)(e0 + e)], a1 = b.edge_attr[3 * (size_t)(e0 + e) + 1],
                      a2 = b.edge_attr[3 * (size_t)(e0 + e) + 2];
            const bool aok = (a0 >= 0) & (a0 < 5) & (a1 >= 0) & (a1 < 6) & (a2 >= 0) & (a2 < 2);
            if (!aok) atomicMax(c.err, ERR_EDGE_ATTR);
            code = aok ? (a0 * 6 + a1) * 2 + a2 : 0;
        }
        s_code[e] = (uint8_t)code;
    }
    __syncthreads();
    // exclusive scan of s_cnt[0..n) in place, s_cnt[n] = ne
    {
        constexpr int PER = (NMAX + NT - 1) / NT;
        int vals[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid * PER + k;
            vals[k] = i < n ? s_cnt[i] : 0;
            sum += vals[k];
        }
        const int incl = wave_inclusive_scan(sum);
        if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
        __syncthreads();
        int base = incl - sum;
        for (int w = 0; w < (tid >> 6); w++) base += s_wtot[w];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid * PER + k;
            if (i < n) s_cnt[i] = base;
            base += vals[k];
        }
        if (tid == 0) s_cnt[n] = ne;
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        c.row_ptr[noff + i] = e0 + s_cnt[i];
        c.out_deg[noff + i] = s_odeg[i];
    }
    if (g == b.num_graphs - 1 && tid == 0) c.row_ptr[b.n_tot] = b.e_tot;
    for (int e = tid; e < ne; e += NT) {
        const int v = s_v[e];
        s_slot[s_cnt[v] + atomicAdd(&s_cur[v], 1)] = (IdxT)e;
    }
    __syncthreads();
    for (int sl = tid; sl < ne; sl += NT) {
        const int e = s_slot[sl];
        const int v = s_v[e], u = s_u[e];
        const int beg = s_cnt[v], end = s_cnt[v + 1];
        int rank = 0;
        // four slots, then their four sources, per trip: the two LDS reads of a comparison depend on each other
        // (molecule class only: rows of two to four in-edges are done in one trip, build 0.185 -> 0.162 ms at 2^18 molhiv graphs; on
        // kNN rows of sixteen the same loop measured 11 % slower than the plain one, on GIN-VN's hubs 3 %)
        if constexpr (DYN) {
            for (int t = beg; t < end; t += 4) {
                int e2[4], u2[4];
#pragma unroll
                for (int i = 0; i < 4; i++) e2[i] = s_slot[t + i < end ? t + i : end - 1];
#pragma unroll
                for (int i = 0; i < 4; i++) u2[i] = s_u[e2[i]];
#pragma unroll
                for (int i = 0; i < 4; i++) rank += (t + i < end) & ((u2[i] < u) | ((u2[i] == u) & (e2[i] < e)));
            }
        } else {
            for (int t = beg; t < end; t++) {
                const int e2 = s_slot[t];
                const int u2 = s_u[e2];
                rank += (u2 < u) | ((u2 == u) & (e2 < e));
            }
        }
        const int pos = e0 + beg + rank;
        c.src[pos] = noff + u;
        c.eid[pos] = e0 + e;
        if (has_attr) c.ecode[pos] = s_code[e];
    }
}

// only_if (device int, or null): per-graph classes only -- every workgroup returns at once when *only_if == 0 (DGN's matrix-pipe path
// needs the CSR only for batches with duplicate edges, which its own index pass finds on the device)
void launch_build_csr(const BatchView& b, const CsrView& c, bool has_edge_attr, int max_nodes, int max_edges, hipStream_t s, const int* only_if) {
    if (b.num_graphs > 0 && max_nodes <= 256 && max_edges <= 1024) {
        // dynamic LDS of one graph: 3 nmax + 1 + NT / 64 ints, 3 emax indices, emax codes
        const int nmax = (max_nodes + 1) & ~1, emax = (max_edges + 3) & ~3;
        const size_t lds = (size_t)(3 * nmax + 1 + 1 + 1) * 4 + (size_t)emax * (3 * sizeof(uint16_t) + 1);
        build_csr_graph_kernel<64, 256, 1024, uint16_t, true><<<b.num_graphs, 64, lds, s>>>(b, c, has_edge_attr, nmax, emax, only_if);
        return;
    }
    // kNN graphs of the hep10k shape (<= ~100 nodes x 16 in-edges): 18 KB of LDS per graph, so nine workgroups share a CU;
    // in the class below one graph claims 139 KB and a CU holds a single 4-wave workgroup (0.93 -> 0.22 ms for 2^15 graphs)
    if (b.num_graphs > 0 && max_nodes <= 256 && max_edges <= 2048) {
        build_csr_graph_kernel<128, 256, 2048, uint16_t, false><<<b.num_graphs, 128, 0, s>>>(b, c, has_edge_attr, 256, 2048, only_if);
        return;
    }
    if (b.num_graphs > 0 && max_nodes <= 512 && max_edges <= 6144) {  // the reference's own caps: 500 nodes / 5500 edges
        build_csr_graph_kernel<256, 512, 6144, uint16_t, false><<<b.num_graphs, 256, 0, s>>>(b, c, has_edge_attr, 512, 6144, only_if);
        return;
    }
    if (b.num_graphs > 0 && max_nodes <= 2048 && max_edges <= 16384) {
        build_csr_graph_kernel<256, 2048, 16384, uint16_t, false><<<b.num_graphs, 256, 0, s>>>(b, c, has_edge_attr, 2048, 16384, only_if);
        return;
    }
    // flat global path: any graph size
    // c.tmp holds [E] edge codes (first half) and [E] slot->edge map (second half): sized 2E by the engine
    int* code_by_edge = c.tmp;
    int* slot_edge = c.tmp + b.e_tot;
    (void)hipMemsetAsync(c.row_ptr, 0, sizeof(int) * ((size_t)b.n_tot + 1), s);
    (void)hipMemsetAsync(c.out_deg, 0, sizeof(int) * (size_t)b.n_tot, s);
    (void)hipMemsetAsync(c.cursor, 0, sizeof(int) * (size_t)b.n_tot, s);
    if (b.num_graphs > 0)
        globalize_count_kernel<<<(b.num_graphs + 3) / 4, 256, 0, s>>>(b, c, has_edge_attr);
    exclusive_scan_inplace(c.row_ptr, b.n_tot, c.block_sums, s);
    if (b.e_tot > 0) {
        const int nb = (b.e_tot + 255) / 256;
        place_kernel<<<nb, 256, 0, s>>>(c, b.e_tot, slot_edge);
        rank_kernel<<<nb, 256, 0, s>>>(c, b.e_tot, slot_edge, has_edge_attr ? code_by_edge : nullptr);
    }
}

}  // namespace fg
