// GCN / GAT / PNA (ap_fixed<16,6>) and DGN (ap_fixed<16,3>) in the reference's own number formats: the bit-faithful mode of
// SURVEY 8f rank 2 for the models ginq.hip does not cover.  Selected per engine with flowgnn_set_numeric_mode(engine,
// FLOWGNN_NUMERIC_Q6_10) (DGN then computes in its own format, Q3.13); modelq.hip has the kernels, oracle/q_oracle.c the rules.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "common.h"
#include "device_common.h"

namespace fg {

struct QPack {  // quantised device copies of ALL weight tensors of a model, in entry-point order, + the function tables
    std::vector<int16_t*> dev;
    int16_t* exptab = nullptr;  // hls::exp over every Q6.10 pattern (rule R5)
    int16_t* logtab = nullptr;  // hls::log(FM_TYPE(k)), k = 1..31
    int16_t* extra = nullptr;   // model-specific derived tables (GCN: edge-embedding combos, bn_sqrt_var)
    GrowBufI work;              // activations of the resident batch
    int F = 10;
    int upload_all(int ntens, const float* const* tens, const size_t* elems, int frac_bits);
    void release();
};

int gcnq_forward(QPack& q, DeviceBatch& db, Profiler& prof, hipStream_t s);
int gatq_forward(QPack& q, DeviceBatch& db, const int* feat_row /* reference quirk, or null */, Profiler& prof, hipStream_t s);
int pnaq_forward(QPack& q, DeviceBatch& db, Profiler& prof, hipStream_t s);
int dgnq_forward(QPack& q, DeviceBatch& db, Profiler& prof, hipStream_t s);

}  // namespace fg
