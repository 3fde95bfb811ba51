// Software-pipelined gather of the GIN layer kernels: the per-lane state and the three macros that issue one slice
// of the NEXT tile's gather, wait for it, and fold it into the next B operand.  Shared by gin_layer_pipelined_kernel
// (gin.hip) and gin_layer_split_persistent_kernel (gin_split.hip).  The macros use these names from the enclosing
// scope: h, row_ptr, src, ecode, s_ecomb, g, nnode, nvalid, bqn[25], and the p_* / px* state declared by the kernel.
#pragma once

constexpr int GIN_PIPE_D = 100;

// gather-pipeline state of one lane (plain scalars on purpose: a struct captured by reference ended up in
// scratch memory, and scratch traffic shares vmcnt with the LDS-DMA)
#define GIN_PIPE_ISSUE(c)                                                                                     \
    do {                                                                                                      \
        p_mode = 0;                                                                                           \
        if ((c) == 0) {                                                                                       \
            if (nvalid) { p_rp0 = row_ptr[nnode]; p_rp1 = row_ptr[nnode + 1]; }                               \
            p_mode = 1;                                                                                       \
        }                                                                                                     \
        /* ONE load site for the own row (step 1) and for neighbour rows (steps 2..): the loads land      */ \
        /* directly in the registers the fold reads (no copies that would need an early wait).  At step 0 */ \
        /* the row bounds are still 0 == 0, so nothing is issued.                                         */ \
        {                                                                                                     \
            const bool self_ = ((c) == 1);                                                                    \
            if (self_ || p_ecur < p_eend) {                                                                   \
                const size_t row_ = self_ ? (size_t)nnode : (size_t)p_unx;                                    \
                const float* hr_ = h + row_ * GIN_PIPE_D + 4 * g;                                                  \
                px0 = *reinterpret_cast<const float4_t*>(hr_);      px1 = *reinterpret_cast<const float4_t*>(hr_ + 16); \
                px2 = *reinterpret_cast<const float4_t*>(hr_ + 32); px3 = *reinterpret_cast<const float4_t*>(hr_ + 48); \
                px4 = *reinterpret_cast<const float4_t*>(hr_ + 64); px5 = *reinterpret_cast<const float4_t*>(hr_ + 80); \
                pxt = h[row_ * GIN_PIPE_D + 96 + g];                                                               \
                p_code = p_cnx;                                                                               \
                const int ne_ = self_ ? p_ecur : p_ecur + 1;                                                  \
                if (ne_ < p_eend) { p_unew = src[ne_]; p_cnew = ecode[ne_]; }                                 \
                p_mode = self_ ? 2 : 3;                                                                       \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)

// The one wait of a step.  The in-flight registers are tied to the asm as read-write operands so that the
// compiler cannot hoist their consumers above it (it otherwise moves part of the fold into the MFMA block
// and inserts its own s_waitcnt vmcnt(0) there, draining the LDS-DMA and the gather in mid-step).
#define GIN_PIPE_WAIT() GIN_PIPE_WAIT_N("0")
#define GIN_PIPE_WAIT_N(N)                                                                                    \
    asm volatile("s_waitcnt vmcnt(" N ")"                                                                     \
                 : "+v"(px0), "+v"(px1), "+v"(px2), "+v"(px3), "+v"(px4), "+v"(px5), "+v"(pxt), "+v"(p_unew), \
                   "+v"(p_cnew), "+v"(p_rp0), "+v"(p_rp1)                                                     \
                 :                                                                                            \
                 : "memory")

#define GIN_PIPE_ADD4(q, X, W)                                                                                \
    bqn[4 * (q) + 0] += relu1((W).x + (X).x); bqn[4 * (q) + 1] += relu1((W).y + (X).y);                        \
    bqn[4 * (q) + 2] += relu1((W).z + (X).z); bqn[4 * (q) + 3] += relu1((W).w + (X).w)
#define GIN_PIPE_SET4(q, X)                                                                                   \
    bqn[4 * (q) + 0] = (X).x; bqn[4 * (q) + 1] = (X).y; bqn[4 * (q) + 2] = (X).z; bqn[4 * (q) + 3] = (X).w

#define GIN_PIPE_CONSUME()                                                                                    \
    do {                                                                                                      \
        if (p_mode == 1) {                                                                                    \
            p_ecur = nvalid ? p_rp0 : 0;                                                                      \
            p_eend = nvalid ? p_rp1 : 0;                                                                      \
        } else if (p_mode == 2) {                                                                             \
            GIN_PIPE_SET4(0, px0); GIN_PIPE_SET4(1, px1); GIN_PIPE_SET4(2, px2);                              \
            GIN_PIPE_SET4(3, px3); GIN_PIPE_SET4(4, px4); GIN_PIPE_SET4(5, px5);                              \
            bqn[24] = pxt;                                                                                    \
            p_unx = p_unew; p_cnx = p_cnew;                                                                   \
        } else if (p_mode == 3) {                                                                             \
            const float* er_ = s_ecomb + p_code * GIN_PIPE_D + 4 * g;                                              \
            const float4 w0_ = *reinterpret_cast<const float4*>(er_), w1_ = *reinterpret_cast<const float4*>(er_ + 16), \
                         w2_ = *reinterpret_cast<const float4*>(er_ + 32), w3_ = *reinterpret_cast<const float4*>(er_ + 48), \
                         w4_ = *reinterpret_cast<const float4*>(er_ + 64), w5_ = *reinterpret_cast<const float4*>(er_ + 80); \
            GIN_PIPE_ADD4(0, px0, w0_); GIN_PIPE_ADD4(1, px1, w1_); GIN_PIPE_ADD4(2, px2, w2_);               \
            GIN_PIPE_ADD4(3, px3, w3_); GIN_PIPE_ADD4(4, px4, w4_); GIN_PIPE_ADD4(5, px5, w5_);               \
            bqn[24] += relu1(s_ecomb[p_code * GIN_PIPE_D + 96 + g] + pxt);                                         \
            p_ecur++;                                                                                         \
            p_unx = p_unew; p_cnx = p_cnew;                                                                   \
        }                                                                                                     \
    } while (0)

