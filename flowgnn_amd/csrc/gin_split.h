// GIN layer with the dense 100->200->100 update on the f16 matrix pipe, fp32-accurate by operand splitting.
// See gin_split.hip for the scheme; gin.hip owns the model and decides which layer kernel runs.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

#include <cstddef>
#include <cstdint>

namespace fg {

// weight stream: 8 chunks per layer; chunk s = W1 fragments of hidden tiles 2s, 2s+1 (absent for s = 7) +
// W2 fragments of K-step s-1 (absent for s = 0, where the space holds b2 and the output scale)
constexpr int GS_STEPS = 8;
constexpr int GS_TAIL_OFF = 12288;     // bytes [0,12288): 2 hidden tiles x 3 K-steps x {hi,lo} x 1 KiB
constexpr int GS_B1_OFF = 12800;       // bytes [12288,12800): fp32 K-tail fragments, 2 x 64 floats
constexpr int GS_W2_OFF = 12928;       // bytes [12800,12928): b1 slices, 2 x 16 floats
constexpr int GS_CHUNK_BYTES = 27264;  // bytes [12928,27264): 7 output tiles x {hi,lo} x 1 KiB
constexpr int GS_CHUNK_STRIDE = 27648; // in global memory: 27 pieces of 1 KiB
constexpr size_t GS_LAYER_BYTES = (size_t)GS_STEPS * GS_CHUNK_STRIDE;

// w1[200][100], b1[200], w2[100][200], b2[100] (row-major, host) -> GS_LAYER_BYTES at `out` (host)
void gin_split_pack_layer(const float* w1, const float* b1, const float* w2, const float* b2, uint8_t* out);

// variant (FLOWGNN_GIN_SPLIT_NT): 4 = eight-wave workgroups of 128 nodes (default), 1 / 2 = four waves x 1 / 2 node tiles
// pool_w != null (last layer, readout folded in): hout is float[n_tot] and receives h'[v] . pool_w instead of the rows
// one GIN layer: hout = MLP(h[v] + sum_e relu(h[src_e] + ecomb[code_e])); *range_flag |= 1 if an operand left the
// range in which the split is fp32-accurate (the caller then repeats the forward pass on the fp32 MFMA kernel)
void launch_gin_layer_split(const float* h, float* hout, const int* row_ptr, const int* src, const uint8_t* ecode,
                            const float* ecomb, const uint8_t* chunks, int n_tot, int e_tot, int relu_out, int* range_flag,
                            int variant, hipStream_t s, const float* pool_w = nullptr);

// Graph-resident form (gin_split.hip, gin_resident_kernel): all five layers + readout in one launch; a persistent workgroup
// keeps a tile of whole graphs (GraphTiles: <= GIN_RESIDENT_ROWS rows, <= GIN_RESIDENT_EDGES in-edges) in LDS across the layers.
// h0 = atom-encoder output [N][100]; ecomb_all [5][60][100]; chunks_all = 5 x gin_resident_layer_bytes(); hout (nullable): h_5 rows for
// the flowgnn_get_h tap; out [G] receives the logits.
// weight stream of the resident kernel (its own chunk format: gin_split.hip "GR chunks"); chunks_all = 5 x gin_resident_layer_bytes()
size_t gin_resident_layer_bytes();
void gin_resident_pack_layer(const float* w1, const float* b1, const float* w2, const float* b2, uint8_t* out, bool merged = true);
constexpr int GIN_RESIDENT_ROWS = 256;
constexpr int GIN_RESIDENT_EDGES = 1280;
constexpr int GIN_RESIDENT_DESC_BYTES = 3584;  // per-tile descriptor built by gin_tile_prep_kernel (CSR slice as 16-bit words, row offsets, column owners)
#ifdef FLOWGNN_DEV
// development builds only (dev/gin_pp_*.inc) -- ping-pong form of the graph-resident kernel (gin_pp_kernel): two half-tiles of <= 128 rows / 640 in-edges per CU, half a layer out
// of phase -- one half multiplies while the other gathers and loads.  Single-task folded readout only; graphs beyond the half-tile
// limits go to launch_gin_resident with the (start, end) pair lists (tstride 2).
constexpr int GIN_PP_ROWS = 128, GIN_PP_EDGES = 640;
size_t gin_pp_layer_bytes();
size_t gin_pp_table_floats();
int gin_pp_desc_bytes();
void gin_pp_pack_layer(const uint8_t* resident_layer /* gin_resident_pack_layer's output */, uint8_t* out);
void gin_pp_pack_tables(const float* ecomb_all /* [5][60][100] */, float* out);
void launch_gin_pp(const float* h0, const int* row_ptr, const int* src, const uint8_t* ecode, const float* tables, const uint8_t* pieces,
                   const float* pool_b, const int* sub_tiles /* [n_sub][4] */, uint8_t* sub_desc, const int* node_off, float* out, int n_sub,
                   int* range_flag, const float* head_u, hipStream_t s, bool prof = false, int waves = 8);
#endif

// what the one-pass tile loader needs (launch_gin_resident, tb != null): the caller's arrays, the per-node table-row numbers it writes
// (8 B per node) and the pre-combined encoder table (gin_resident_pack_enc_table); err = the engine's validation flag
struct GinTileBuild {
    BatchView batch;
    void* enc_idx;         // device, [n_tot] x 4 B, written by gin_tile_build_kernel
    const float* enc_tab;  // device, gin_resident_enc_table_floats() floats
    int* err;
    // bin-packed tiles (GraphTiles::bp_list / bp_lrow; null: a tile is the range of graphs tile_graph[t] .. and of the batch's rows tile_row[t] ..):
    // a tile is the graphs list[tile_graph[t]] .. one behind the other, tile_row[t] its first row in the tile-ordered row space
    const int* list = nullptr;
    const int* lrow = nullptr;
};
size_t gin_resident_enc_table_floats();
void gin_resident_pack_enc_table(const float* node_embedding /* [173][100] */, float* out);
// the one-pass front end: descriptors + encoder row numbers of every tile from the caller's arrays (then launch_gin_resident with tb)
void launch_gin_tile_build(const GinTileBuild& tb, const int* tile_row, const int* tile_graph, uint8_t* tile_desc, int n_tiles, bool hubs,
                           int col_order, hipStream_t s);
void launch_gin_resident(const float* h0, float* hout, const int* row_ptr, const int* src, const uint8_t* ecode, const float* ecomb_all,
                         const uint8_t* chunks_all, const float* pool_w, const float* pool_b, const int* tile_row, const int* tile_graph,
                         uint8_t* tile_desc /* scratch, n_tiles x GIN_RESIDENT_DESC_BYTES */, const int* node_off, float* out, int n_tiles,
                         int* range_flag, hipStream_t s, bool hubs = false, const float* head_u = nullptr, int col_order = 0, bool prof = false,
                         const GinTileBuild* tb = nullptr, int tstride = 1);
// head_u for launch_gin_resident (GIN_RESIDENT_HEAD_FLOATS floats): the single-task readout folded through the LAST layer's second
// linear layer -- u = W2^T w_pred divided by the first layer's power-of-two weight scale, padded to 208, then c = b2 . w_pred
constexpr int GIN_RESIDENT_HEAD_FLOATS = 209;
void gin_resident_head_fold(const float* w1_last, const float* w2_last, const float* b2_last, const float* pool_w, float* out);


}  // namespace fg
