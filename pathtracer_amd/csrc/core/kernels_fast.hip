// kernels_fast.hip — the kernels of kernel.h a second time, with TOLERANCE arithmetic (namespace ptc::fast).
//
// Same source as the exact variant in ptcore.hip, one compile-time switch (PT_FAST, set by the Makefile together
// with -ffp-contract=fast for this file only). What changes: every a*b+c may fuse (the slab test becomes
// b*rinv - o*rinv, one fused multiply-add per plane; dots, crosses, Moeller-Trumbore), division and roots are the
// hardware's 1-ulp v_rcp_f32 / v_sqrt_f32 / v_rsq_f32, Box-Muller uses v_log_f32 / v_sin_f32 / v_cos_f32, the
// hidden double-precision spots are float, and the walk keeps no second-candidate distance and never starts over
// sequentially (an exact tie goes to whichever candidate is tested first). What does not change: the integer RNG
// stream, the order of random draws, every control-flow decision's FORM, the data layout, the launch machinery.
// It stands in for the arithmetic nvcc gives the reference by default (contraction on: bvh.cu:222-237,
// intersect.cu:12-81, mathops.h:114-121) and is graded by north_star's three checks (tests/test_gpu_tolerance.py),
// not by bit-identity; the exact variant stays the default and the gate.
#ifndef PT_FAST
#error "build kernels_fast.hip with -DPT_FAST=1 -ffp-contract=fast (see ../Makefile)"
#endif
#include "kernel.h"
#include "launch.h"
