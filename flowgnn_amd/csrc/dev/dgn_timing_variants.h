// Development only (-DFLOWGNN_DEV plus -DDGNR_TIMING=<bits>, scripts/dev/variant.sh): TIMING variants of dgn_resident_kernel.  They leave
// out a phase and compute WRONG results on purpose, to price that phase (NOTEBOOK.md round 6).  Never part of the shipped library.
//   1  no s_barrier at the K-steps (the vmcnt waits stay)          -> what lock-stepping the eight waves costs
//   2  no weight stream (no chunk requests, no vmcnt waits)        -> what waiting for the chunks costs
//   4  no aggregation MFMAs (m1 = m2 = 0)
//   8  no dense MFMAs
//  16  no encoder (h_0 = 0: no table reads)
//  32  no readout
#pragma once
#ifndef DGNR_TIMING
#define DGNR_TIMING 0
#endif
#define DGNR_SKIP(bit) ((DGNR_TIMING & (bit)) != 0)
