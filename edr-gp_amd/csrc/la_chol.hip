// Blocked right-looking Cholesky and recursive triangular inverse, fp64.
//
// potrf_lower: nb = 64.  Per step: (1) one single-wave workgroup factors the
// 64x64 diagonal block in LDS (left-looking column Cholesky, one lane per row)
// and inverts it; (2) panel  L21 = A21 * inv(L11)^T  as an MFMA GEMM;
// (3) trailing update  A22 -= L21 L21^T  on the lower tile set (MFMA GEMM).
// GPy reaches this through jitchol -> LAPACK dpotrf (GPy/util/linalg.py).
//
// trtri_lower: Linv = L^-1 by recursive doubling: the 64x64 diagonal blocks are
// inverted in one batched launch, then at level s every pair of adjacent
// s-blocks is merged with two batched MFMA GEMMs
//     inv([[L11, 0], [L21, L22]]) = [[I11, 0], [-I22 L21 I11, I22]].
#include "common.h"

namespace edrgp {

constexpr int NB = 64;

// Factor (do_chol) and/or invert one 64x64 lower block per workgroup (64 threads
// = one wave).  A is the block origin; jb <= 64 valid rows/cols, the rest is
// padded with the identity inside LDS.  Writes L (lower, zeros above) back when
// do_chol, and the inverse to Dinv (64x64 dense, row-major, ld 64) or, when
// inv_out != nullptr, into the matrix inv_out (ld ldi) restricted to jb x jb.
__global__ __launch_bounds__(64) void diag_block_kernel(double* __restrict__ Abase, int lda, int n,
                                                        int first_block, int do_chol,
                                                        double* __restrict__ Dinv,
                                                        double* __restrict__ inv_out, int ldi,
                                                        int* __restrict__ info) {
    __shared__ double S[NB][NB + 1];
    __shared__ double Xi[NB][NB + 1];
    const int lane = threadIdx.x;
    const int blk = first_block + blockIdx.x;
    const int j0 = blk * NB;
    const int jb = min(NB, n - j0);
    double* A = Abase + (size_t)j0 * lda + j0;

    for (int r = 0; r < NB; ++r) {
        double v = (r == lane) ? 1.0 : 0.0;
        if (r < jb && lane < jb && lane <= r) v = A[(size_t)r * lda + lane];
        S[r][lane] = v;
        Xi[r][lane] = 0.0;
    }
    __syncthreads();

    if (do_chol) {
        const int i = lane;
        int bad = 0;
        for (int j = 0; j < NB; ++j) {
            double s = S[i][j];
            int k = 0;
            for (; k + 4 <= j; k += 4) {
                s -= S[i][k] * S[j][k];
                s -= S[i][k + 1] * S[j][k + 1];
                s -= S[i][k + 2] * S[j][k + 2];
                s -= S[i][k + 3] * S[j][k + 3];
            }
            for (; k < j; ++k) s -= S[i][k] * S[j][k];
            const double d = __shfl(s, j, 64);
            if (!(d > 0.0) && bad == 0) bad = j0 + j + 1;
            const double dd = sqrt(d);               // NaN propagates when d <= 0
            __syncthreads();
            if (i == j) S[i][j] = dd;
            else if (i > j) S[i][j] = s / dd;
            __syncthreads();
        }
        if (bad && lane == 0) atomicCAS(info, 0, bad);
        // write L back (lower part, zeros above inside the block)
        for (int r = 0; r < jb; ++r)
            if (lane < jb) A[(size_t)r * lda + lane] = (lane <= r) ? S[r][lane] : 0.0;
    }

    // inverse by forward substitution, lane c owns column c of the inverse
    {
        const int cidx = lane;
        for (int i = 0; i < NB; ++i) {
            double s = (i == cidx) ? 1.0 : 0.0;
            int k = 0;
            for (; k + 4 <= i; k += 4) {
                s -= S[i][k] * Xi[k][cidx];
                s -= S[i][k + 1] * Xi[k + 1][cidx];
                s -= S[i][k + 2] * Xi[k + 2][cidx];
                s -= S[i][k + 3] * Xi[k + 3][cidx];
            }
            for (; k < i; ++k) s -= S[i][k] * Xi[k][cidx];
            Xi[i][cidx] = s / S[i][i];
            __syncthreads();
        }
    }
    if (Dinv) {
        double* D = Dinv + (size_t)blockIdx.x * NB * NB;
        for (int r = 0; r < NB; ++r) D[r * NB + lane] = Xi[r][lane];
    }
    if (inv_out) {
        double* O = inv_out + (size_t)j0 * ldi + j0;
        for (int r = 0; r < jb; ++r)
            if (lane < jb) O[(size_t)r * ldi + lane] = Xi[r][lane];
    }
}

__global__ void copy2d_kernel(const double* __restrict__ src, int lds, double* __restrict__ dst, int ldd,
                              int rows, int cols) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i < rows && j < cols) dst[(size_t)i * ldd + j] = src[(size_t)i * lds + j];
}

__global__ void zero_upper_kernel(double* __restrict__ A, int lda, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i < n && j < n && j > i) A[(size_t)i * lda + j] = 0.0;
}

__global__ void fill_kernel(double* __restrict__ A, size_t count, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) A[i] = v;
}

int potrf_lower(Ctx* c, double* A, int n, int lda, int* info_dev) {
    HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), c->stream));
    TRY(c->dinv.reserve((size_t)NB * NB * sizeof(double)));
    TRY(c->panel.reserve((size_t)n * NB * sizeof(double)));
    double* Dinv = c->dinv.d();
    double* P = c->panel.d();
    const int nblk = (n + NB - 1) / NB;
    for (int b = 0; b < nblk; ++b) {
        const int j0 = b * NB;
        const int jb = std::min(NB, n - j0);
        hipLaunchKernelGGL(diag_block_kernel, dim3(1), dim3(64), 0, c->stream, A, lda, n, b, 1, Dinv,
                           (double*)nullptr, 0, info_dev);
        const int rem = n - j0 - jb;
        if (rem > 0) {
            double* A21 = A + (size_t)(j0 + jb) * lda + j0;
            // P = A21 * Dinv^T   (rem x jb), Dinv is 64x64 with identity padding
            TRY(gemm(c, false, true, rem, jb, jb, 1.0, A21, lda, Dinv, NB, 0.0, P, NB));
            dim3 blk(64, 4), grd((jb + 63) / 64, (rem + 3) / 4);
            hipLaunchKernelGGL(copy2d_kernel, grd, blk, 0, c->stream, P, NB, A21, lda, rem, jb);
            // A22 -= P P^T on the lower tiles
            double* A22 = A + (size_t)(j0 + jb) * lda + (j0 + jb);
            GemmOpts o;
            o.tri = 1;
            o.force_tile = (rem >= 1024) ? 128 : 64;
            TRY(gemm(c, false, true, rem, rem, jb, -1.0, P, NB, P, NB, 1.0, A22, lda, o));
        }
    }
    dim3 blk(64, 4), grd((n + 63) / 64, (n + 3) / 4);
    hipLaunchKernelGGL(zero_upper_kernel, grd, blk, 0, c->stream, A, lda, n);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int trtri_lower(Ctx* c, const double* L, int ldl, double* Linv, int ldi, int n) {
    // zero the output, then diagonal-block inverses in one batched launch
    {
        // output may have ldi > n; clear row by row through a 2D fill
        if (ldi == n) {
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, c->stream, Linv, (size_t)n * n, 0.0);
        } else {
            for (int i = 0; i < n; ++i)
                HIP_TRY(hipMemsetAsync(Linv + (size_t)i * ldi, 0, sizeof(double) * n, c->stream));
        }
    }
    const int nblk = (n + NB - 1) / NB;
    hipLaunchKernelGGL(diag_block_kernel, dim3(nblk), dim3(64), 0, c->stream, const_cast<double*>(L), ldl, n,
                       0, 0, (double*)nullptr, Linv, ldi, (int*)nullptr);
    HIP_TRY(hipGetLastError());
    if (n <= NB) return EDRGP_OK;
    TRY(c->trtri_ws.reserve((size_t)n * n * sizeof(double)));
    double* Wk = c->trtri_ws.d();
    for (int s = NB; s < n; s *= 2) {
        const int nz = (n - s + 2 * s - 1) / (2 * s);      // pairs with (2z+1)s < n
        if (nz <= 0) break;
        GemmOpts o;
        o.batch = nz;
        o.limM_base = n - s;
        o.limM_stride = 2 * s;
        // T = L21 * I11  -> Wk (same coordinates as the 21 block), ld n
        o.strideA = (long long)2 * s * ldl + 2 * s;
        o.strideB = (long long)2 * s * ldi + 2 * s;
        o.strideC = (long long)2 * s * n + 2 * s;
        TRY(gemm(c, false, false, s, s, s, 1.0, L + (size_t)s * ldl, ldl, Linv, ldi, 0.0,
                 Wk + (size_t)s * n, n, o));
        // I21 = - I22 * T
        GemmOpts o2;
        o2.batch = nz;
        o2.limM_base = n - s;
        o2.limM_stride = 2 * s;
        o2.limK_base = n - s;
        o2.limK_stride = 2 * s;
        o2.strideA = (long long)2 * s * ldi + 2 * s;
        o2.strideB = (long long)2 * s * n + 2 * s;
        o2.strideC = (long long)2 * s * ldi + 2 * s;
        TRY(gemm(c, false, false, s, s, s, -1.0, Linv + (size_t)s * ldi + s, ldi, Wk + (size_t)s * n, n, 0.0,
                 Linv + (size_t)s * ldi, ldi, o2));
    }
    return EDRGP_OK;
}

}  // namespace edrgp
