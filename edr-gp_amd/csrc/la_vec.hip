// Small fp64 helpers around the MFMA kernels: gemv, reductions with a fixed
// (deterministic) summation order, element-wise combinations of M x M matrices.
#include "common.h"

namespace edrgp {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-wide sum (256 threads); result valid in thread 0
__device__ __forceinline__ double block_sum256(double v) {
    __shared__ double part[4];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) part[w] = v;
    __syncthreads();
    return (threadIdx.x == 0) ? (part[0] + part[1]) + (part[2] + part[3]) : 0.0;
}

// y[i] = alpha * sum_j A[i][j] x[j]; one wave per row
__global__ __launch_bounds__(256) void gemv_n_kernel(int M, int N, double alpha, const double* __restrict__ A,
                                                     int lda, const double* __restrict__ x,
                                                     double* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const double* a = A + (size_t)row * lda;
    double s = 0.0;
    for (int j = lane; j < N; j += 64) s += a[j] * x[j];
    s = wave_sum(s);
    if (lane == 0) y[row] = alpha * s;
}

// part[chunk][j] = sum_{i in chunk} A[i][j] x[i]; chunk = 128 rows, 64 columns per block
__global__ __launch_bounds__(256) void gemv_t_part_kernel(int M, int N, const double* __restrict__ A, int lda,
                                                          const double* __restrict__ x,
                                                          double* __restrict__ part) {
    __shared__ double red[4][64];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int i0 = blockIdx.y * 128;
    const int i1 = min(M, i0 + 128);
    double s = 0.0;
    if (j < N)
        for (int i = i0 + sub; i < i1; i += 4) s += A[(size_t)i * lda + j] * x[i];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && j < N)
        part[(size_t)blockIdx.y * N + j] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                           (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[j] = alpha * sum_p part[p][j].  64 columns x 16 row groups per workgroup: group g adds
// rows g, g+16, ... in order, then the 16 group sums are added in order (fixed summation order).
__global__ __launch_bounds__(1024) void col_reduce_kernel(const double* __restrict__ part, int nparts, int n,
                                                          double alpha, double* __restrict__ out) {
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    // eight loads in flight per thread (eight partial sums, folded in a fixed order): the dependent-add form was one
    // memory latency per row, 50 us for the 782 row tiles of the K_nm^T y partials in front of the Phi kernel
    double s = 0.0;
    if (j < n) {
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int p = grp;
        for (; p + 7 * 16 < nparts; p += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += part[(size_t)(p + 16 * u) * n + j];
        }
        for (int u = 0; p < nparts; p += 16, ++u) a[u] += part[(size_t)p * n + j];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && j < n) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][lane];
        out[j] = alpha * t;
    }
}

int gemv_n(Ctx* c, int M, int N, double alpha, const double* A, int lda, const double* x, double* y) {
    hipLaunchKernelGGL(gemv_n_kernel, dim3((M + 3) / 4), dim3(256), 0, c->stream, M, N, alpha, A, lda, x, y);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// y[j] = alpha * sum_i A[i][j] x[i]   (A is M x N)
int gemv_t(Ctx* c, int M, int N, double alpha, const double* A, int lda, const double* x, double* y) {
    const int nchunk = (M + 127) / 128;
    TRY(c->red.reserve((size_t)nchunk * N * sizeof(double)));
    hipLaunchKernelGGL(gemv_t_part_kernel, dim3((N + 63) / 64, nchunk), dim3(256), 0, c->stream, M, N, A, lda,
                       x, c->red.d());
    hipLaunchKernelGGL(col_reduce_kernel, dim3((N + 63) / 64), dim3(1024), 0, c->stream, c->red.d(), nchunk, N,
                       alpha, y);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// Column sums of a tall rows x n matrix (ld = n): 1024-row blocks leave partial sums (fixed order inside a block: four
// row groups of four accumulators each), col_reduce folds them.  (The Gram-matrix GEMM against a ones vector that did
// this for the mean Jacobian cost a 32 us launch and a 50 us slab reduction per call.)
__global__ __launch_bounds__(256) void col_partial_kernel(const double* __restrict__ A, int64_t rows, int n,
                                                         double* __restrict__ part) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int j = blockIdx.y * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.x * 1024, r1 = min(rows, r0 + 1024);
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (j < n) {
        int64_t r = r0 + grp;
        for (; r + 12 < r1; r += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += A[(size_t)(r + 4 * u) * n + j];
        }
        for (int u = 0; r < r1; r += 4, ++u) a[u] += A[(size_t)r * n + j];
    }
    red[grp][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (grp == 0 && j < n) part[(size_t)blockIdx.x * n + j] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

int col_sums(Ctx* c, const double* A, int64_t rows, int n, double* out) {
    const int nb = (int)((rows + 1023) / 1024);
    TRY(c->red.reserve((size_t)nb * n * sizeof(double)));
    hipLaunchKernelGGL(col_partial_kernel, dim3((unsigned)nb, (unsigned)((n + 63) / 64)), dim3(256), 0, c->stream, A, rows, n,
                       c->red.d());
    HIP_TRY(hipGetLastError());
    return col_reduce(c, c->red.d(), nb, n, out);
}

int col_reduce(Ctx* c, const double* part, int nparts, int n, double* out) {
    hipLaunchKernelGGL(col_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, c->stream, part, nparts, n, 1.0,
                       out);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

__global__ void add_diag_kernel(double* A, int n, int lda, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(size_t)i * lda + i] += v;
}

int add_diag(Ctx* c, double* A, int n, int lda, double v) {
    hipLaunchKernelGGL(add_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, A, n, lda, v);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// Right-hand side carried through a Cholesky factorisation as one extra matrix row:
//   row[j] = scale v[j] (j < n),   row[n] = 1 + sum_j row[j]^2
// Factoring [[B, .], [row, row[n]]] leaves L^-1 (scale v) in that row; with lambda_min(B) >= 1 the last
// pivot row[n] - |L^-1 scale v|^2 stays >= 1.
// Set-up of the bound-only factorisation in one launch: Lg (ld ldb, (n+1) x (n+1)) <- [[A + shift I, .], [scale v^T,
// 1 + |scale v|^2]] from A (n x n, ld n).  Block b copies rows 8 b .. 8 b + 7; the last block writes the extra row.
__global__ __launch_bounds__(256) void aug_setup_kernel(double* __restrict__ Lg, int ldb, const double* __restrict__ A, int n,
                                                        double shift, const double* __restrict__ v, double scale) {
    if ((int)blockIdx.x == (int)gridDim.x - 1) {
        __shared__ double sh[256];
        double* row = Lg + (size_t)n * ldb;
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) {
            const double t = scale * v[i];
            row[i] = t;
            s = fma(t, t, s);
        }
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) row[n] = 1.0 + sh[0];
        return;
    }
    const int r0 = blockIdx.x * 8;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = r0 + u;
        if (r >= n) break;
        for (int j = threadIdx.x; j < n; j += 256) Lg[(size_t)r * ldb + j] = A[(size_t)r * n + j] + ((j == r) ? shift : 0.0);
    }
}

int aug_setup(Ctx* c, double* Lg, int ldb, const double* A, int n, double shift, const double* v, double scale) {
    hipLaunchKernelGGL(aug_setup_kernel, dim3((n + 7) / 8 + 1), dim3(256), 0, c->stream, Lg, ldb, A, n, shift, v, scale);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// sum_i log L[i][i] (i < n) and sum_j L[n][j]^2 (j < n) of the factor of that matrix, one launch, fixed order
__global__ __launch_bounds__(256) void aug_tail_kernel(const double* __restrict__ L, int ldl, int n, double* __restrict__ out_logdet,
                                                       double* __restrict__ out_sumsq) {
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        s += log(L[(size_t)i * ldl + i]);
        const double t = L[(size_t)n * ldl + i];
        q = fma(t, t, q);
    }
    s = block_sum256(s);
    q = block_sum256(q);
    if (threadIdx.x == 0) { out_logdet[0] = s; out_sumsq[0] = q; }
}

int aug_tail(Ctx* c, const double* L, int ldl, int n, double* out_logdet, double* out_sumsq) {
    hipLaunchKernelGGL(aug_tail_kernel, dim3(1), dim3(256), 0, c->stream, L, ldl, n, out_logdet, out_sumsq);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// Generic single-block reduction with a fixed order: mode selects the summand.
//  0: x[i]^2   1: x[i]*y[i]   2: A[i][i]   3: log(A[i][i])   4: A[i][i]/n
__global__ __launch_bounds__(256) void reduce1_kernel(int mode, const double* __restrict__ x,
                                                      const double* __restrict__ y, int64_t n, int lda,
                                                      double* __restrict__ out) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        double v;
        if (mode == 0) v = x[i] * x[i];
        else if (mode == 1) v = x[i] * y[i];
        else if (mode == 2) v = x[(size_t)i * lda + i];
        else if (mode == 3) v = log(x[(size_t)i * lda + i]);
        else v = x[(size_t)i * lda + i] / (double)n;
        s += v;
    }
    s = block_sum256(s);
    if (threadIdx.x == 0) out[0] = s;
}

// two-stage sum of squares for long vectors / matrices: per-block partials then one block
__global__ __launch_bounds__(256) void sumsq_part_kernel(const double* __restrict__ x, int64_t n,
                                                         const double* __restrict__ y,
                                                         double* __restrict__ part) {
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        s += x[i] * (y ? y[i] : x[i]);
    s = block_sum256(s);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sum_part_kernel(const double* __restrict__ part, int n,
                                                       double* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum256(s);
    if (threadIdx.x == 0) out[0] = s;
}

static int long_dot(Ctx* c, const double* x, const double* y, int64_t n, double* out) {
    if (n <= 4096) {
        hipLaunchKernelGGL(reduce1_kernel, dim3(1), dim3(256), 0, c->stream, y ? 1 : 0, x, y, n, 0, out);
    } else {
        const int nb = (int)std::min<int64_t>(1024, (n + 1023) / 1024);
        TRY(c->red.reserve((size_t)nb * sizeof(double)));
        hipLaunchKernelGGL(sumsq_part_kernel, dim3(nb), dim3(256), 0, c->stream, x, n, y, c->red.d());
        hipLaunchKernelGGL(sum_part_kernel, dim3(1), dim3(256), 0, c->stream, c->red.d(), nb, out);
    }
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int sum_sq(Ctx* c, const double* x, int64_t n, double* out) { return long_dot(c, x, nullptr, n, out); }
int dot(Ctx* c, const double* x, const double* y, int64_t n, double* out) { return long_dot(c, x, y, n, out); }
int frob_sq(Ctx* c, const double* A, int n, int lda, double* out) {
    if (lda != n) { set_error("frob_sq: lda must equal n"); return EDRGP_ERR_INVALID; }
    return long_dot(c, A, nullptr, (int64_t)n * n, out);
}

int trace_of(Ctx* c, const double* A, int n, int lda, double* out) {
    hipLaunchKernelGGL(reduce1_kernel, dim3(1), dim3(256), 0, c->stream, 2, A, (const double*)nullptr,
                       (int64_t)n, lda, out);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}
int sum_log_diag(Ctx* c, const double* A, int n, int lda, double* out) {
    hipLaunchKernelGGL(reduce1_kernel, dim3(1), dim3(256), 0, c->stream, 3, A, (const double*)nullptr,
                       (int64_t)n, lda, out);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}
int diag_mean(Ctx* c, const double* A, int n, int lda, double* out) {
    hipLaunchKernelGGL(reduce1_kernel, dim3(1), dim3(256), 0, c->stream, 4, A, (const double*)nullptr,
                       (int64_t)n, lda, out);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

__global__ __launch_bounds__(256) void diag_minmax_kernel(const double* __restrict__ A, int n, int lda,
                                                         double* __restrict__ out2) {
    __shared__ double smin[256], smax[256];
    double mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double v = A[(size_t)i * lda + i];
        mn = fmin(mn, v);
        mx = fmax(mx, v);
    }
    smin[threadIdx.x] = mn;
    smax[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + o]);
            smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = smin[0]; out2[1] = smax[0]; }
}

int diag_minmax(Ctx* c, const double* A, int n, int lda, double* out2) {
    hipLaunchKernelGGL(diag_minmax_kernel, dim3(1), dim3(256), 0, c->stream, A, n, lda, out2);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// out = a*A + b*B + cc*Cm + g * w w^T   (n x n, ld n); null matrices are skipped
__global__ void combine3_kernel(int n, double a, const double* __restrict__ A, double b,
                                const double* __restrict__ B, double cc, const double* __restrict__ Cm,
                                double g, const double* __restrict__ w, double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= n || j >= n) return;
    const size_t idx = (size_t)i * n + j;
    double v = 0.0;
    if (A) v += a * A[idx];
    if (B) v += b * B[idx];
    if (Cm) v += cc * Cm[idx];
    if (w) v += g * w[i] * w[j];
    out[idx] = v;
}

int combine3(Ctx* c, int n, double a, const double* A, double b, const double* B, double cc, const double* Cm,
             double g, const double* w, double* out) {
    dim3 blk(64, 4), grd((n + 63) / 64, (n + 3) / 4);
    hipLaunchKernelGGL(combine3_kernel, grd, blk, 0, c->stream, n, a, A, b, B, cc, Cm, g, w, out);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// Device-generated operand for timing probes (edrgp_dbg_syrk_time): seed 0 = constant data,
// otherwise uniform [0, 1) values (like kernel-matrix entries) from a splitmix-style hash of the index.
__global__ void fill_pattern_kernel(double* __restrict__ K, size_t n, unsigned long long seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ULL + seed;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
        K[i] = (seed == 0ULL) ? 0.5 : (double)(x >> 11) * (1.0 / 9007199254740992.0);
    }
}

int fill_pattern(Ctx* c, double* K, size_t n, unsigned long long seed) {
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(2048), dim3(256), 0, c->stream, K, n, seed);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

}  // namespace edrgp
