// bernstein.hpp -- the one piece of arithmetic the Bezier evaluators of kernels.hpp, ring.hpp and traj.hip share.
#pragma once
#include <hip/hip_runtime.h>

namespace pct {

// x^n for a small non-negative integer n, CORRECTLY ROUNDED: repeated multiplication in double-double with error-free products
// (p + e = hi * x exactly through one FMA), renormalised after every step; the accumulated error stays below 2^-100 of the value,
// so the returned double is the nearest one to the exact power.  The reference evaluates its Bernstein terms with libm's
// pow(u, j) (sim_planning_demo.cpp:715-727), which is correctly rounded on glibc >= 2.28 except within ~2^-15 ulp of a rounding
// boundary; ocml's pow (the device's) is not -- with it 5-10 % of the sampled positions differed from the host's in the last bit.
// (The explicit FMAs are intended: -ffp-contract=off only forbids the compiler from fusing on its own.)
__host__ __device__ __forceinline__ double pow_uint_cr(double x, int n)
{
    if (n <= 0) return 1.0;
    double hi = x, lo = 0.0;
    for (int i = 1; i < n; i++) {
        const double p = hi * x;
        const double e = __builtin_fma(hi, x, -p);
        const double l = __builtin_fma(lo, x, e);
        const double s = p + l;
        lo = l - (s - p);
        hi = s;
    }
    return hi;
}

// n choose j as an exact double by the multiplicative recurrence (bezier_base.cpp:33-48 / :256-266 hold them as doubles computed from
// integer factorials; every intermediate here is an integer below 2^53, the rounding guard only absorbs the division).  The ONE
// definition every Bernstein evaluator of the engine uses; pinned against the reference's compiled table (tests/golden/binomials.npz).
__host__ __device__ __forceinline__ double bernstein_binom(int n, int j)
{
    double b = 1.0;
    for (int i = 1; i <= j; i++) b = floor(b * (double)(n - i + 1) / (double)i + 0.5);
    return b;
}

}  // namespace pct
