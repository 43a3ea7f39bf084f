// rocrand_check.cpp — TEST INFRASTRUCTURE ONLY. Runs rocRAND's own XORWOW engine
// (rocrand_device::xorwow_engine, /opt/rocm/include/rocrand/rocrand_xorwow.h, host side of its
// __host__ __device__ code, with rocRAND's precomputed jump tables) so that the oracle's whole generator
// — seed scramble structure, subsequence jump, Weyl counter, output — can be compared with an independent
// implementation once rocRAND's four scramble constants are substituted (orc_rng_set_scramble).
// cuRAND itself is not in this image: which four constants cuRAND uses stays unverified.
// Built by oracle/Makefile with hipcc (host code only; no kernel, no GPU needed).
#include <hip/hip_runtime.h>

#include <rocrand/rocrand_xorwow.h>

extern "C" void rr_xorwow_stream(unsigned long long seed, unsigned long long subsequence, unsigned long long offset,
                                 unsigned int *out, int n) {
    rocrand_device::xorwow_engine engine(seed, subsequence, offset);
    for (int i = 0; i < n; i++) out[i] = engine.next();
}

// rocRAND's scramble constants, read back from its engine: seed -> state of subsequence 0 is
// x0 = 123456789 + t0, x2 = 521288629 + t1 with t0 = M0 * (lo ^ C0), t1 = M1 * (hi ^ C1). The test takes them
// from the header text instead (rocrand_xorwow.h:113-116); this export only says which header was compiled.
extern "C" const char *rr_header(void) { return "/opt/rocm/include/rocrand/rocrand_xorwow.h"; }
