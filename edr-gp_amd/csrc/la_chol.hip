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

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// sqrt(p) and 1/sqrt(p) by Goldschmidt iteration from v_rsq_f64 (p > 0, normal
// range): two coupled refinements + one residual correction, ~1 ulp.
__device__ __forceinline__ void sqrt_rsqrt(double p, double& sq, double& rs) {
    const double y = __builtin_amdgcn_rsq(p);
    double g = p * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, p);
    sq = fma(d, h, g);
    rs = h + h;
}

// Factor (DO_CHOL) and invert one 64x64 lower block per workgroup; the
// workgroup is ONE wave and lane i owns row i of the block in registers.
//   factor : right-looking, fully unrolled.  Column j: pivot broadcast by
//            v_readlane, Goldschmidt rsqrt, the finished column goes to LDS
//            (Lt[j][*], lane-contiguous) and is read back as wave-uniform
//            broadcasts for the rank-1 update of the lane's row.
//   invert : lane c solves L x = e_c by right-looking forward substitution, L
//            read from Lt as broadcasts, x in registers, no cross-lane traffic.
// jb <= 64 valid rows/cols; the rest is padded with the identity.  Outputs: L
// (lower, zeros above) written back when DO_CHOL; the inverse goes to Dinv
// (dense 64x64, ld 64) and/or into inv_out (ld ldi, jb x jb part).
template <bool DO_CHOL>
__global__ __launch_bounds__(64) void diag_block_kernel(double* __restrict__ Abase, int lda, int n,
                                                        int first_block, double* __restrict__ Dinv,
                                                        double* __restrict__ inv_out, int ldi,
                                                        int* __restrict__ info,
                                                        unsigned long long* __restrict__ stamps) {
    __shared__ double Lt[NB][NB + 1];      // Lt[k][i] = L[i][k]
#define EDRGP_STAMP(slot) do { if (stamps != nullptr && threadIdx.x == 0) stamps[slot] = __builtin_amdgcn_s_memtime(); } while (0)
    EDRGP_STAMP(0);
    __shared__ double rdiag[NB];           // 1 / L[i][i]
    const int lane = threadIdx.x;
    const int blk = first_block + blockIdx.x;
    const int j0 = blk * NB;
    const int jb = min(NB, n - j0);
    double* A = Abase + (size_t)j0 * lda + j0;

    // coalesced load of the block (all 64 row loads in flight at once), then
    // transposed into LDS: Lt[c][r] = A[r][c]
    {
        double t[NB];
        const int cl = min(lane, jb - 1);
#pragma unroll
        for (int r = 0; r < NB; ++r) t[r] = A[(size_t)min(r, jb - 1) * lda + cl];   // clamped, branch-free
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const bool in = (r < jb) && (lane < jb) && (lane <= r);
            t[r] = in ? t[r] : ((r == lane) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) Lt[lane][r] = t[r];
    }
    __syncthreads();
    EDRGP_STAMP(1);

    if (DO_CHOL) {
        double a[NB];                                          // row `lane` of A, then of L
#pragma unroll
        for (int k = 0; k < NB; ++k) a[k] = Lt[k][lane];
        __syncthreads();
        int bad = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double p = readlane_f64(a[j], j);
            if (!(p > 0.0) && bad == 0) bad = j0 + j + 1;
            // 1/sqrt(p): v_rsq_f64 + two Newton steps  y <- y (1.5 - 0.5 p y^2)
            const double hp = 0.5 * p;
            double y = __builtin_amdgcn_rsq(p);
            y = y * fma(-hp, y * y, 1.5);
            y = y * fma(-hp, y * y, 1.5);
            double l = a[j] * y;                    // lane j: p / sqrt(p) = sqrt(p)
            if (lane < j) l = 0.0;
            a[j] = l;
            if (j + 1 < NB) {
                // critical path first: the next pivot column needs only l[j+1] (one readlane)
                const double cn = readlane_f64(l, j + 1);
                a[j + 1] = fma(-l, cn, a[j + 1]);
            }
            Lt[j][lane] = l;
            if (lane == j) rdiag[j] = y;
            __syncthreads();
            // bulk of the rank-1 update from wave-uniform LDS reads (v_readlane per
            // element measured 4x slower than the LDS broadcast for a lone wave)
#pragma unroll
            for (int k = j + 2; k < NB; ++k) a[k] = fma(-l, Lt[j][k], a[k]);
        }
        if (bad && lane == 0 && info != nullptr) atomicCAS(info, 0, bad);
        __syncthreads();
        EDRGP_STAMP(2);
        // coalesced write-back of L (lower part, explicit zeros above)
        for (int r = 0; r < jb; ++r)
            if (lane < jb) A[(size_t)r * lda + lane] = (lane <= r) ? Lt[lane][r] : 0.0;
        EDRGP_STAMP(3);
    } else {
        rdiag[lane] = 1.0 / Lt[lane][lane];
        __syncthreads();
        EDRGP_STAMP(3);
    }

    // inverse: lane c owns column c of X = L^-1.  Right-looking forward
    // substitution: once x_k is final, every later row's accumulator takes its
    // L[i][k] x_k term -- 63-k independent FMAs per step; column k of L is read
    // from LDS as wave-uniform broadcasts, prefetched one column ahead.
    {
        double x[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) x[i] = (i == lane) ? 1.0 : 0.0;
        double la[NB], lb[NB];             // column k of L (rows k+1..63), two-stage ring
#pragma unroll
        for (int i = 1; i < NB; ++i) la[i] = Lt[0][i];
#pragma unroll
        for (int k = 0; k < NB; k += 2) {
            // prefetch column k+1 while column k is consumed, and vice versa
#pragma unroll
            for (int i = k + 2; i < NB; ++i) lb[i] = Lt[k + 1][i];
            __builtin_amdgcn_sched_barrier(0);
            x[k] *= rdiag[k];
            {
                const double xk = x[k];
#pragma unroll
                for (int i = k + 1; i < NB; ++i) x[i] = fma(-la[i], xk, x[i]);
            }
            if (k + 2 < NB) {
#pragma unroll
                for (int i = k + 3; i < NB; ++i) la[i] = Lt[k + 2][i];
            }
            __builtin_amdgcn_sched_barrier(0);
            x[k + 1] *= rdiag[k + 1];
            {
                const double xk = x[k + 1];
#pragma unroll
                for (int i = k + 2; i < NB; ++i) x[i] = fma(-lb[i], xk, x[i]);
            }
        }
        EDRGP_STAMP(4);
        if (Dinv) {
            double* D = Dinv + (size_t)blockIdx.x * NB * NB;
#pragma unroll
            for (int i = 0; i < NB; ++i) D[i * NB + lane] = x[i];
        }
        if (inv_out) {
            double* O = inv_out + (size_t)j0 * ldi + j0;
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (i < jb && lane < jb) O[(size_t)i * ldi + lane] = x[i];
        }
    }
    EDRGP_STAMP(5);
#undef EDRGP_STAMP
}

// one factor+invert of a 64x64 block with phase stamps (shader cycles): diagnostic only
int diag_block_stamps(Ctx* c, double* A64, double* Dinv64, unsigned long long* stamps_dev, int do_chol) {
    if (do_chol)
        hipLaunchKernelGGL(diag_block_kernel<true>, dim3(1), dim3(64), 0, c->stream, A64, 64, 64, 0, Dinv64,
                           (double*)nullptr, 0, (int*)nullptr, stamps_dev);
    else
        hipLaunchKernelGGL(diag_block_kernel<false>, dim3(1), dim3(64), 0, c->stream, A64, 64, 64, 0, Dinv64,
                           (double*)nullptr, 0, (int*)nullptr, stamps_dev);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
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
    double* Dinv = c->dinv.d();
    const int nblk = (n + NB - 1) / NB;
    for (int b = 0; b < nblk; ++b) {
        const int j0 = b * NB;
        const int jb = std::min(NB, n - j0);
        hipLaunchKernelGGL(diag_block_kernel<true>, dim3(1), dim3(64), 0, c->stream, A, lda, n, b, Dinv,
                           (double*)nullptr, 0, info_dev, (unsigned long long*)nullptr);
        const int rem = n - j0 - jb;
        if (rem > 0) {
            double* A21 = A + (size_t)(j0 + jb) * lda + j0;
            // L21 = A21 * Dinv^T in place: a workgroup owns whole rows of the 64-wide
            // panel (one column tile) and has consumed them before its epilogue stores.
            GemmOpts po;
            po.force_tile = 64;
            TRY(gemm(c, false, true, rem, jb, jb, 1.0, A21, lda, Dinv, NB, 0.0, A21, lda, po));
            // A22 -= L21 L21^T on the lower tiles
            double* A22 = A + (size_t)(j0 + jb) * lda + (j0 + jb);
            GemmOpts o;
            o.tri = 1;
            o.force_tile = (rem >= 1024) ? 128 : 64;
            TRY(gemm(c, false, true, rem, rem, jb, -1.0, A21, lda, A21, lda, 1.0, A22, lda, o));
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
    hipLaunchKernelGGL(diag_block_kernel<false>, dim3(nblk), dim3(64), 0, c->stream, const_cast<double*>(L), ldl,
                       n, 0, (double*)nullptr, Linv, ldi, (int*)nullptr, (unsigned long long*)nullptr);
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
        o.kmode = K_GE_COL;                                  // I11 is lower triangular
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
        o2.kmode = K_LE_ROW;                                 // I22 is lower triangular
        TRY(gemm(c, false, false, s, s, s, -1.0, Linv + (size_t)s * ldi + s, ldi, Wk + (size_t)s * n, n, 0.0,
                 Linv + (size_t)s * ldi, ldi, o2));
    }
    return EDRGP_OK;
}

}  // namespace edrgp
