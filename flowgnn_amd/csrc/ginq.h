// GIN / GIN-VN in the reference's own number format ap_fixed<16,6> ("Q6.10"): the bit-faithful mode of SURVEY 8f rank 2.
// Selected per engine with flowgnn_set_numeric_mode(engine, FLOWGNN_NUMERIC_Q6_10); ginq.hip has the arithmetic rules.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "common.h"

namespace fg {

struct GinQWeights {  // device copies, all int16 bit patterns
    int16_t* nemb = nullptr;   // [173][100]
    int16_t* ecomb = nullptr;  // [5 layers][60 codes][100]   (wrapped sum of the three edge-embedding rows)
    int16_t* w1 = nullptr;     // [5][200][104]  (K padded to 104 with zeros)
    int16_t* b1 = nullptr;     // [5][200]
    int16_t* w2 = nullptr;     // [5][100][200]
    int16_t* b2 = nullptr;     // [5][100]
    int16_t* pw = nullptr;     // [100]
    int16_t* pb = nullptr;     // [1]
    void release();
};

// host float tensors (the layout GinModel::set_weights receives) -> quantised device copies
int ginq_upload(GinQWeights& q, const float* nemb, const float* eemb, const float* w1, const float* b1, const float* w2, const float* b2,
                const float* pw, const float* pb);

// whole forward pass in Q6.10 on the resident batch; logits written to db.out as float = pattern / 1024
int ginq_forward(const GinQWeights& q, DeviceBatch& db, Profiler& prof, hipStream_t s);

}  // namespace fg
