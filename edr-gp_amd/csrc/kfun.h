// Stationary covariance as a function of the scaled squared distance r2:
// k(r) and h(r) = (dk/dr)/r  (GPy kern.K_of_r / dK_dr, stationary.py, rbf.py).
#pragma once
#include "../../include/edrgp_hip.h"

#define EDRGP_SQRT3 1.7320508075688772
#define EDRGP_SQRT5 2.23606797749979

__device__ __forceinline__ void edrgp_kfun(int kind, double var, double r2, double& K, double& h) {
    if (kind == EDRGP_KERN_RBF) {
        K = var * exp(-0.5 * r2);
        h = -K;
    } else if (kind == EDRGP_KERN_MATERN32) {
        const double r = sqrt(r2);
        const double e = exp(-EDRGP_SQRT3 * r);
        K = var * (1.0 + EDRGP_SQRT3 * r) * e;
        h = -3.0 * var * e;
    } else {
        const double r = sqrt(r2);
        const double e = exp(-EDRGP_SQRT5 * r);
        K = var * (1.0 + EDRGP_SQRT5 * r + (5.0 / 3.0) * r2) * e;
        h = -(5.0 / 3.0) * var * (1.0 + EDRGP_SQRT5 * r) * e;
    }
}
