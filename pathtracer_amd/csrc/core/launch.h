// launch.h — host-side launch wrappers of the kernels in kernel.h, one set per arithmetic variant.
//
// Include after kernel.h, once per translation unit. It defines ptc::<exact|fast>::launch_* (the template dispatch
// over stack size, tile shape and tree form) and the accessor ptc::launchers_<exact|fast>() that hands them to
// ptcore.hip as a table of plain function pointers, so that the C ABI picks a variant per context at run time
// (pt_set_arithmetic) without seeing either variant's templates.
#pragma once

#include <hip/hip_runtime.h>

#include "kernel.h"

namespace ptc {

struct KernelLaunchers {
    const char *name; // "exact" | "fast"
    // render_kernel<count, STACK(stack_need), rows, wide>; counters only with count
    void (*render)(bool count, int stack_need, int rows, bool wide, dim3 grid, const KernelParams &P, float *img,
                   unsigned long long *counters, hipStream_t st);
    // cost_probe_kernel: the same code without an image (tile costs, tree timing)
    void (*probe)(int stack_need, int rows, bool wide, dim3 grid, const KernelParams &P, hipStream_t st);
    // primary_hits_kernel
    void (*primary)(int stack_need, bool wide, dim3 grid, const KernelParams &P, int32_t *faces, float *dist, unsigned long long *redone,
                    hipStream_t st);
};

const KernelLaunchers &launchers_exact(); // ptcore.hip
const KernelLaunchers &launchers_fast();  // kernels_fast.hip

namespace PT_ARITH_NS {

template <bool COUNT, int ROWS, bool WIDE>
void launch_render_as(int need, dim3 grid, const KernelParams &P, float *img, unsigned long long *counters, hipStream_t st) {
    if (need <= 32)
        hipLaunchKernelGGL((render_kernel<COUNT, 32, ROWS, WIDE>), grid, dim3(kBlock), 0, st, P, img, counters);
    else if (need <= 40)
        hipLaunchKernelGGL((render_kernel<COUNT, 40, ROWS, WIDE>), grid, dim3(kBlock), 0, st, P, img, counters);
    else if (need <= 48)
        hipLaunchKernelGGL((render_kernel<COUNT, 48, ROWS, WIDE>), grid, dim3(kBlock), 0, st, P, img, counters);
    else
        hipLaunchKernelGGL((render_kernel<COUNT, 64, ROWS, WIDE>), grid, dim3(kBlock), 0, st, P, img, counters);
}

template <bool COUNT>
void launch_render_count(int need, int rows, bool wide, dim3 grid, const KernelParams &P, float *img, unsigned long long *counters, hipStream_t st) {
    if (rows == 16) {
        if (wide) launch_render_as<COUNT, 16, true>(need, grid, P, img, counters, st);
        else launch_render_as<COUNT, 16, false>(need, grid, P, img, counters, st);
    } else {
        if (wide) launch_render_as<COUNT, 8, true>(need, grid, P, img, counters, st);
        else launch_render_as<COUNT, 8, false>(need, grid, P, img, counters, st);
    }
}

inline void launch_render(bool count, int need, int rows, bool wide, dim3 grid, const KernelParams &P, float *img, unsigned long long *counters,
                          hipStream_t st) {
    if (count) launch_render_count<true>(need, rows, wide, grid, P, img, counters, st);
    else launch_render_count<false>(need, rows, wide, grid, P, img, counters, st);
}

template <int ROWS, bool WIDE> void launch_probe_as(int need, dim3 grid, const KernelParams &P, hipStream_t st) {
    if (need <= 32) hipLaunchKernelGGL((cost_probe_kernel<32, ROWS, WIDE>), grid, dim3(kBlock), 0, st, P);
    else if (need <= 40) hipLaunchKernelGGL((cost_probe_kernel<40, ROWS, WIDE>), grid, dim3(kBlock), 0, st, P);
    else if (need <= 48) hipLaunchKernelGGL((cost_probe_kernel<48, ROWS, WIDE>), grid, dim3(kBlock), 0, st, P);
    else hipLaunchKernelGGL((cost_probe_kernel<64, ROWS, WIDE>), grid, dim3(kBlock), 0, st, P);
}

inline void launch_probe(int need, int rows, bool wide, dim3 grid, const KernelParams &P, hipStream_t st) {
    if (rows == 16) {
        if (wide) launch_probe_as<16, true>(need, grid, P, st);
        else launch_probe_as<16, false>(need, grid, P, st);
    } else {
        if (wide) launch_probe_as<8, true>(need, grid, P, st);
        else launch_probe_as<8, false>(need, grid, P, st);
    }
}

template <bool WIDE>
void launch_primary_as(int need, dim3 grid, const KernelParams &P, int32_t *faces, float *dist, unsigned long long *redone, hipStream_t st) {
    if (need <= 32) hipLaunchKernelGGL((primary_hits_kernel<32, WIDE>), grid, dim3(kBlock), 0, st, P, faces, dist, redone);
    else if (need <= 48) hipLaunchKernelGGL((primary_hits_kernel<48, WIDE>), grid, dim3(kBlock), 0, st, P, faces, dist, redone);
    else hipLaunchKernelGGL((primary_hits_kernel<64, WIDE>), grid, dim3(kBlock), 0, st, P, faces, dist, redone);
}

inline void launch_primary(int need, bool wide, dim3 grid, const KernelParams &P, int32_t *faces, float *dist, unsigned long long *redone,
                           hipStream_t st) {
    if (wide) launch_primary_as<true>(need, grid, P, faces, dist, redone, st);
    else launch_primary_as<false>(need, grid, P, faces, dist, redone, st);
}

} // namespace PT_ARITH_NS

#if PT_FAST
const KernelLaunchers &launchers_fast() {
    static const KernelLaunchers k = {"fast", &fast::launch_render, &fast::launch_probe, &fast::launch_primary};
    return k;
}
#else
const KernelLaunchers &launchers_exact() {
    static const KernelLaunchers k = {"exact", &exact::launch_render, &exact::launch_probe, &exact::launch_primary};
    return k;
}
#endif

} // namespace ptc
