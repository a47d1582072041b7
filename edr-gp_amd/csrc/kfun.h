// Stationary covariance as a function of the scaled squared distance r2:
// k(r) and h(r) = (dk/dr)/r  (GPy kern.K_of_r / dK_dr, stationary.py, rbf.py).
#pragma once
#include "../../include/edrgp_hip.h"

#define EDRGP_SQRT3 1.7320508075688772
#define EDRGP_SQRT5 2.23606797749979

// exp(x) for x <= 0 (every covariance argument here): n = rint(x log2 e), g = x - n ln2 (two-term ln2),
// degree-13 Taylor polynomial on |g| <= 0.347 (truncation 4e-18), ldexp.  ~20 VALU instructions against the
// library exp's special-case handling; max error 1 ulp-class (checked against libm in tests/test_hip_kernels.py).
__device__ __forceinline__ double edrgp_exp_neg(double x) {
    if (!(x > -708.0)) return 0.0;                       // also catches NaN -> 0 is never reached for finite inputs
    const double n = rint(x * 1.4426950408889634);
    double g = fma(n, -6.93147180369123816490e-01, x);
    g = fma(n, -1.90821492927058770002e-10, g);
    double p = 1.0 / 6227020800.0;
    p = fma(p, g, 1.0 / 479001600.0);
    p = fma(p, g, 1.0 / 39916800.0);
    p = fma(p, g, 1.0 / 3628800.0);
    p = fma(p, g, 1.0 / 362880.0);
    p = fma(p, g, 1.0 / 40320.0);
    p = fma(p, g, 1.0 / 5040.0);
    p = fma(p, g, 1.0 / 720.0);
    p = fma(p, g, 1.0 / 120.0);
    p = fma(p, g, 1.0 / 24.0);
    p = fma(p, g, 1.0 / 6.0);
    p = fma(p, g, 0.5);
    p = fma(p, g, 1.0);
    p = fma(p, g, 1.0);
    return ldexp(p, (int)n);
}

__device__ __forceinline__ void edrgp_kfun(int kind, double var, double r2, double& K, double& h) {
    if (kind == EDRGP_KERN_RBF) {
        K = var * edrgp_exp_neg(-0.5 * r2);
        h = -K;
    } else if (kind == EDRGP_KERN_MATERN32) {
        const double r = sqrt(r2);
        const double e = edrgp_exp_neg(-EDRGP_SQRT3 * r);
        K = var * (1.0 + EDRGP_SQRT3 * r) * e;
        h = -3.0 * var * e;
    } else {
        const double r = sqrt(r2);
        const double e = edrgp_exp_neg(-EDRGP_SQRT5 * r);
        K = var * (1.0 + EDRGP_SQRT5 * r + (5.0 / 3.0) * r2) * e;
        h = -(5.0 / 3.0) * var * (1.0 + EDRGP_SQRT5 * r) * e;
    }
}
