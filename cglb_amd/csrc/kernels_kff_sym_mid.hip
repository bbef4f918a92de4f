// The mid-width instances (32 < D <= 96, fp64) of the symmetric K_ff mat-vec: same templates as kernels_kff_sym.hip, compiled as a
// translation unit of their own so that the two sets of instances build in parallel.
#define CGLB_SYM_MID_TU 1
#include "kernels_kff_sym.hip"
