// Development only (make DEV=1 plus one of the flags below, scripts/dev/variant.sh): TIMING variants of the resident kernels' in-edge walks.
// They compute WRONG sums on purpose -- the shipped instruction stream with the LDS bank conflicts of the walk taken away -- to price
// those conflicts (NOTEBOOK.md round 5: GIN -3.4 %, GCN -10.2 % as upper bounds).  Never part of the shipped library.
//   -DGR_CONFLICT_FREE_WALK  gin_resident_kernel: every lane of a column tile reads a row of its own bank class
//   -DGR_ONE_CODE_WALK       gin_resident_kernel: every lane reads the table row of code 0 (one broadcast; cf_zero is an opaque 0, so the
//                            reads stay in the loop)
//   -DGCN_CF_ROWS            gcn_resident_kernel: the source row out of the lane group's own 16 aligned rows (16 distinct bank quads)
//   -DGCN_CF_CODES           gcn_resident_kernel: edge code 0 on every lane (a literal 0 would let hipcc hoist the table reads out of the
//                            walk: that would time the walk WITHOUT them)
#pragma once
#ifdef GR_CONFLICT_FREE_WALK
#define GR_WALK_ROW(U) ((((U) & ~15u) | (unsigned)j) < (unsigned)GR_ROWS ? (((U) & ~15u) | (unsigned)j) : (unsigned)j)
#endif
#ifdef GR_ONE_CODE_WALK
#define GR_WALK_CODE(C) ((C) & cf_zero)
#define GR_WALK_TIMING_SETUP() unsigned cf_zero = 0; asm volatile("" : "+v"(cf_zero));
#endif
#if defined(GCN_CF_ROWS) || defined(GCN_CF_CODES)
#ifdef GCN_CF_ROWS
#define GCN_CF_U(U) (((((U) & ~15) | (lane & 15)) < GCNR_ROWS) ? (((U) & ~15) | (lane & 15)) : (lane & 15))
#else
#define GCN_CF_U(U) (U)
#endif
#ifdef GCN_CF_CODES
#define GCN_CF_C(C) ((C) & cf_zero_)
#else
#define GCN_CF_C(C) (C)
#endif
#define GCN_WALK_WORD(W) { int cf_zero_ = 0; asm volatile("" : "+v"(cf_zero_)); (void)cf_zero_; W = (GCN_CF_U((W) >> 6) << 6) | GCN_CF_C((W) & 63); }
#endif
