// Stationary-kernel tile kernels: RBF / Matern32 / Matern52 covariance assembly
// (GPy kern.K, stationary.py/rbf.py), the element-wise factors of
// Stationary.gradients_X / update_gradients_full, and small finishing kernels.
//
// pair_kernel<MODE>: a workgroup (256 threads, 16x16, 4x4 outputs per thread)
// walks 64-row sub-tiles of X against one 64-column tile of X2.  Scaled inputs
// x/l are staged transposed in LDS ([q][row], 16-byte readable) so the inner
// loop is 2 ds_read_b128 per operand and 16 (sub, fma) pairs per dimension;
// global reads of X / X2 are coalesced along the feature dimension.
//   MODE_K : out = sum_p k_p(r_p)  (+ jitter on the diagonal when X2 is X)
//            and, when y is given, per-row-chunk partial sums of K^T y
//   MODE_G : out = wv[m] * h_p(r)            h = (dk/dr)/r, 0 where r == 0
//   MODE_T : out = Q .* h_p(r),  Q = W + ycoef * y[n] * wv[m];
//            dvpart[block] = sum Q .* k_p / var_p
#include "common.h"
#include "kfun.h"

namespace edrgp {

constexpr int MODE_K = 0, MODE_G = 1, MODE_T = 2;
constexpr int DT = 32;          // feature chunk staged in LDS
constexpr int TS = 64;          // tile side
constexpr int TP = TS + 2;      // padded row (16-byte aligned rows)

struct PairArgs {
    KSpec ks;
    int part;
    const double* X;
    int64_t rows;
    const double* X2;
    int M2;
    double* out;
    int ldo;
    int same;
    double jitter;
    const double* y;
    double* bpart;
    const double* W;
    int ldw;
    double ycoef;
    const double* wv;
    double* dvpart;
    int rows_per_block;
};

__device__ __forceinline__ void kfun(int kind, double var, double r2, double& K, double& h) {
    edrgp_kfun(kind, var, r2, K, h);
}

template <int MODE>
__global__ __launch_bounds__(256) void pair_kernel(PairArgs a) {
    __shared__ __attribute__((aligned(16))) double Xs[DT][TP];
    __shared__ __attribute__((aligned(16))) double Zs[DT][TP];
    __shared__ double red[16][TS];

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int D = a.ks.D;
    const int col0 = blockIdx.x * TS;
    const int64_t rb0 = (int64_t)blockIdx.y * a.rows_per_block;
    const int p_begin = (MODE == MODE_K) ? 0 : a.part;
    const int p_end = (MODE == MODE_K) ? a.ks.nparts : a.part + 1;

    double bacc[4] = {0.0, 0.0, 0.0, 0.0};
    double dvacc = 0.0;

    for (int sub = 0; sub * TS < a.rows_per_block; ++sub) {
        const int64_t row0 = rb0 + (int64_t)sub * TS;
        if (row0 >= a.rows) break;

        double outv[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) outv[i][j] = 0.0;

        double Kp[4][4];
        for (int p = p_begin; p < p_end; ++p) {
            double r2[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) r2[i][j] = 0.0;
            const double* invl = a.ks.invl + (size_t)p * D;
            for (int d0 = 0; d0 < D; d0 += DT) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < (TS * DT) / 256; ++e) {
                    const int idx = tid + 256 * e;
                    const int rr = idx / DT, q = idx % DT;
                    const int dq = d0 + q;
                    double xv = 0.0, zv = 0.0;
                    if (dq < D) {
                        const double il = invl[dq];
                        if (row0 + rr < a.rows) xv = a.X[(size_t)(row0 + rr) * D + dq] * il;
                        if (col0 + rr < a.M2) zv = a.X2[(size_t)(col0 + rr) * D + dq] * il;
                    }
                    Xs[q][rr] = xv;
                    Zs[q][rr] = zv;
                }
                __syncthreads();
                const int qn = min(DT, D - d0);
                for (int q = 0; q < qn; ++q) {
                    const double2 x01 = *reinterpret_cast<const double2*>(&Xs[q][ty * 4]);
                    const double2 x23 = *reinterpret_cast<const double2*>(&Xs[q][ty * 4 + 2]);
                    const double2 z01 = *reinterpret_cast<const double2*>(&Zs[q][tx * 4]);
                    const double2 z23 = *reinterpret_cast<const double2*>(&Zs[q][tx * 4 + 2]);
                    const double xv[4] = {x01.x, x01.y, x23.x, x23.y};
                    const double zv[4] = {z01.x, z01.y, z23.x, z23.y};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const double d = xv[i] - zv[j];
                            r2[i][j] = fma(d, d, r2[i][j]);
                        }
                }
            }
            const int kind = a.ks.kind[p];
            const double var = a.ks.var[p];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double K, h;
                    kfun(kind, var, r2[i][j], K, h);
                    if (MODE == MODE_K) {
                        outv[i][j] += K;
                    } else {
                        Kp[i][j] = K;
                        outv[i][j] = (r2[i][j] == 0.0) ? 0.0 : h;   // _inv_dist = 0 at r == 0
                    }
                }
        }

        // ---- epilogue for this 64x64 sub-tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + ty * 4 + i;
            if (row >= a.rows) continue;
            const double yr = (a.y != nullptr) ? a.y[row] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = col0 + tx * 4 + j;
                if (col >= a.M2) continue;
                double v = outv[i][j];
                if (MODE == MODE_K) {
                    if (a.same && row == (int64_t)col) v += a.jitter;
                    bacc[j] = fma(v, yr, bacc[j]);
                } else if (MODE == MODE_G) {
                    v *= a.wv[col];
                } else {
                    double Q = a.W[(size_t)row * a.ldw + col];
                    if (a.wv != nullptr) Q = fma(a.ycoef * yr, a.wv[col], Q);
                    dvacc = fma(Q, Kp[i][j], dvacc);
                    v *= Q;
                }
                a.out[(size_t)row * a.ldo + col] = v;
            }
        }
    }

    if (MODE == MODE_K && a.bpart != nullptr) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) red[ty][tx * 4 + j] = bacc[j];
        __syncthreads();
        if (tid < TS) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k][tid];
            if (col0 + tid < a.M2) a.bpart[(size_t)blockIdx.y * a.M2 + col0 + tid] = s;
        }
    }
    if (MODE == MODE_T && a.dvpart != nullptr) {
        __syncthreads();
        red[ty][tx] = dvacc;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int k = 0; k < 16; ++k)
                for (int l = 0; l < 16; ++l) s += red[k][l];
            a.dvpart[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s / a.ks.var[a.part];
        }
    }
}

static int pick_rows_per_block(int64_t rows, int col_tiles) {
    // enough workgroups to fill 256 CUs several times, at most 1024 rows each
    int rpb = 1024;
    while (rpb > 64 && (int64_t)col_tiles * ((rows + rpb - 1) / rpb) < 2048) rpb >>= 1;
    return rpb;
}

// Xs[n][q] = X[n][q] * invl[q], xn[n] = sum_q Xs[n][q]^2; one thread per row
__global__ void scale_norm_kernel(const double* __restrict__ X, int64_t rows, int D, const double* __restrict__ invl,
                                  double* __restrict__ Xs, double* __restrict__ xn) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= rows) return;
    double s = 0.0;
    for (int q = 0; q < D; ++q) {
        const double v = X[n * D + q] * invl[q];
        Xs[n * D + q] = v;
        s = fma(v, v, s);
    }
    xn[n] = s;
}

// Same result, coalesced: one workgroup per 64 consecutive rows (a contiguous run of 64 D doubles), scaled values
// written straight back and parked in LDS (row stride D + 1: conflict-free), then lane r sums row r in the
// same feature order as the kernel above.  D <= 64.  (The row-per-thread form reads with a stride of D doubles:
// 363 us for 200 000 x 64, three times per C4 evaluation.)
__global__ __launch_bounds__(256) void scale_norm_tile_kernel(const double* __restrict__ X, int64_t rows, int D,
                                                             const double* __restrict__ invl, double* __restrict__ Xs,
                                                             double* __restrict__ xn) {
    extern __shared__ double tile[];                    // 64 x (D + 1)
    const int lane = threadIdx.x;                       // 256 threads move the tile, the first 64 sum its rows
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int nr = (int)min((int64_t)64, rows - row0);
    const int count = nr * D;
    const double* src = X + (size_t)row0 * D;
    double* dst = Xs + (size_t)row0 * D;
    for (int e = lane; e < count; e += 256) {
        const int r = e / D, q = e - r * D;
        const double v = src[e] * invl[q];
        dst[e] = v;
        tile[r * (D + 1) + q] = v;
    }
    __syncthreads();
    if (lane < nr) {
        double s = 0.0;
        for (int q = 0; q < D; ++q) {
            const double v = tile[lane * (D + 1) + q];
            s = fma(v, v, s);
        }
        xn[row0 + lane] = s;
    }
}

static void launch_scale_norm(Ctx* c, const double* X, int64_t rows, int D, const double* invl, double* Xs, double* xn) {
    if (D <= 64)             // (also for the 2048 rows of Z: the row-per-thread form took 13-21 us there, latency-bound)
        hipLaunchKernelGGL(scale_norm_tile_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256),
                           (size_t)64 * (D + 1) * sizeof(double), c->stream, X, rows, D, invl, Xs, xn);
    else
        hipLaunchKernelGGL(scale_norm_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, X, rows, D, invl,
                           Xs, xn);
}

// Single-part kernels: inner products of the scaled inputs on the matrix pipe (NT GEMM over
// the feature dimension) + covariance epilogue.  |x|^2 + |z|^2 - 2 x.z is GPy's own formula.
static int kernel_matrix_mfma(Ctx* c, const KSpec& ks, const double* X, int64_t rows, const double* X2, int M2,
                              double* K, int ldk, int same, double jitter, const double* y, double* bvec) {
    const int D = ks.D;
    const bool self = (X2 == X) && (rows == M2);
    TRY(c->ks_x.reserve((size_t)rows * D * sizeof(double)));
    TRY(c->ks_xn.reserve((size_t)rows * sizeof(double)));
    launch_scale_norm(c, X, rows, D, ks.invl, c->ks_x.d(), c->ks_xn.d());
    const double* Zs = c->ks_x.d();
    const double* zn = c->ks_xn.d();
    if (!self) {
        TRY(c->ks_z.reserve((size_t)M2 * D * sizeof(double)));
        TRY(c->ks_zn.reserve((size_t)M2 * sizeof(double)));
        launch_scale_norm(c, X2, (int64_t)M2, D, ks.invl, c->ks_z.d(), c->ks_zn.d());
        Zs = c->ks_z.d();
        zn = c->ks_zn.d();
    }
    // 64x64 tiles: the product is only D deep, so a workgroup is a short load -> MFMA -> exp -> store chain;
    // four small workgroups per CU overlap those phases where two 128x128 ones serialise them (Cm: 1.08 -> 0.74 ms)
    const int tile = (c->kbuild_tile == 128 && rows >= 128 && M2 >= 128) ? 128 : 64;
    const int tiles_m = (int)((rows + tile - 1) / tile);
    FuseK f;
    f.xn = c->ks_xn.d(); f.zn = zn; f.y = y; f.bpart = nullptr;
    f.var = ks.var[0]; f.jitter = jitter; f.kind = ks.kind[0]; f.same = same; f.dbg = c->kbuild_r2 ? 1 : (same ? 0 : c->kbuild_dbg);
    f.H = c->kbuild_h; f.ldh = ldk;
    if (y != nullptr) {
        TRY(c->bpart.reserve((size_t)tiles_m * M2 * sizeof(double)));
        f.bpart = c->bpart.d();
    }
    GemmOpts o;
    o.tag = 3;
    o.force_tile = tile;
    o.fusek = &f;
    o.dyn_lds = (tile == 64) ? c->kbuild_pad_now : 0;
    TRY(gemm(c, false, true, (int)rows, M2, D, 1.0, c->ks_x.d(), D, Zs, D, 0.0, K, ldk, o));
    if (y != nullptr) TRY(col_reduce(c, c->bpart.d(), tiles_m, M2, bvec));
    return EDRGP_OK;
}

int kernel_matrix(Ctx* c, const KSpec& ks, const double* X, int64_t rows, const double* X2, int M2, double* K,
                  int ldk, int same, double jitter, const double* y, double* bvec) {
    if (rows <= 0 || M2 <= 0) return EDRGP_OK;
    if (c->mfma_kbuild && ks.nparts == 1 && rows <= 0x7fffffffLL)
        return kernel_matrix_mfma(c, ks, X, rows, X2, M2, K, ldk, same, jitter, y, bvec);
    PairArgs a{};
    a.ks = ks; a.part = 0; a.X = X; a.rows = rows; a.X2 = X2; a.M2 = M2; a.out = K; a.ldo = ldk;
    a.same = same; a.jitter = jitter; a.y = y; a.W = nullptr; a.wv = nullptr; a.dvpart = nullptr;
    const int ct = (M2 + TS - 1) / TS;
    a.rows_per_block = pick_rows_per_block(rows, ct);
    const int nchunk = (int)((rows + a.rows_per_block - 1) / a.rows_per_block);
    a.bpart = nullptr;
    if (y != nullptr) {
        TRY(c->bpart.reserve((size_t)nchunk * M2 * sizeof(double)));
        a.bpart = c->bpart.d();
    }
    hipLaunchKernelGGL(pair_kernel<MODE_K>, dim3(ct, nchunk), dim3(256), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    if (y != nullptr) TRY(col_reduce(c, c->bpart.d(), nchunk, M2, bvec));
    return EDRGP_OK;
}

// G = wv[m] * h_part(r)   (rows x Mp)
static int hmatrix(Ctx* c, const KSpec& ks, int part, const double* Xs, int64_t rows, const double* P, int Mp,
                   const double* wv, double* G, int ldg) {
    PairArgs a{};
    a.ks = ks; a.part = part; a.X = Xs; a.rows = rows; a.X2 = P; a.M2 = Mp; a.out = G; a.ldo = ldg;
    a.wv = wv;
    const int ct = (Mp + TS - 1) / TS;
    a.rows_per_block = pick_rows_per_block(rows, ct);
    const int nchunk = (int)((rows + a.rows_per_block - 1) / a.rows_per_block);
    hipLaunchKernelGGL(pair_kernel<MODE_G>, dim3(ct, nchunk), dim3(256), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

__global__ __launch_bounds__(256) void sum_add_kernel(const double* __restrict__ part, int n,
                                                      double* __restrict__ accum) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) accum[0] += sh[0];
}

int sum_into(Ctx* c, const double* part, int n, double* accum) {
    hipLaunchKernelGGL(sum_add_kernel, dim3(1), dim3(256), 0, c->stream, part, n, accum);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int grad_elementwise(Ctx* c, const KSpec& ks, int part, const double* X, const double* y, int64_t rows,
                     const double* Z, int M, const double* W, int ldw, double ycoef, const double* wv, double* T,
                     int ldt, double* dvar_accum) {
    PairArgs a{};
    a.ks = ks; a.part = part; a.X = X; a.rows = rows; a.X2 = Z; a.M2 = M; a.out = T; a.ldo = ldt;
    a.y = y; a.W = W; a.ldw = ldw; a.ycoef = ycoef; a.wv = wv;
    const int ct = (M + TS - 1) / TS;
    a.rows_per_block = pick_rows_per_block(rows, ct);
    const int nchunk = (int)((rows + a.rows_per_block - 1) / a.rows_per_block);
    TRY(c->red.reserve((size_t)nchunk * ct * sizeof(double)));
    a.dvpart = c->red.d();
    hipLaunchKernelGGL(pair_kernel<MODE_T>, dim3(ct, nchunk), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(sum_add_kernel, dim3(1), dim3(256), 0, c->stream, c->red.d(), nchunk * ct, dvar_accum);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// Kmm term of single-part kernels, element-wise: T = dK .* h(r(Z, Z)), dvar += sum dK .* k / var, with k and h as the
// MFMA build of K(Z, Z) stores them (h = 0 where r = 0, k = var on the diagonal).  Entries beyond the Mr real
// inducing points are written as zeros.  Replaces the pair kernel's second pass over the M^2 D differences
// (0.58 ms at M = 2048, D = 32 -- and 12 ms starved beside the persistent gradient contraction).
__global__ __launch_bounds__(256) void kmm_term_kernel(const double* __restrict__ dK, const double* __restrict__ Kz,
                                                       const double* __restrict__ Hz, int Mr, int M, double inv_var,
                                                       double* __restrict__ T, double* __restrict__ dvpart) {
    __shared__ double sh[256];
    const size_t total = (size_t)M * M;
    const size_t base = (size_t)blockIdx.x * 1024;
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t e = base + (size_t)u * 256 + threadIdx.x;
        if (e < total) {
            const int i = (int)(e / M), j = (int)(e - (size_t)i * M);
            double t = 0.0;
            if (i < Mr && j < Mr) {
                const double d = dK[e];
                t = d * Hz[e];
                s = fma(d, Kz[e], s);
            }
            T[e] = t;
        }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dvpart[blockIdx.x] = sh[0] * inv_var;
}

int kmm_term_elementwise(Ctx* c, const double* dK, const double* Kz, const double* Hz, int Mr, int M, double var,
                         double* T, double* dvar_accum) {
    const int nb = (int)(((size_t)M * M + 1023) / 1024);
    TRY(c->red.reserve((size_t)nb * sizeof(double)));
    hipLaunchKernelGGL(kmm_term_kernel, dim3(nb), dim3(256), 0, c->stream, dK, Kz, Hz, Mr, M, 1.0 / var, T, c->red.d());
    hipLaunchKernelGGL(sum_add_kernel, dim3(1), dim3(256), 0, c->stream, c->red.d(), nb, dvar_accum);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// Xt[n] = [x_n | 1 | x_n .* x_n]
__global__ void xtilde_kernel(const double* __restrict__ X, int64_t rows, int D, double* __restrict__ Xt) {
    const int W = 2 * D + 1;
    const int64_t total = rows * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / W;
        const int q = (int)(i - n * W);
        double v;
        if (q < D) v = X[n * D + q];
        else if (q == D) v = 1.0;
        else { const double x = X[n * D + (q - D - 1)]; v = x * x; }
        Xt[i] = v;
    }
}

// Xq[n] = [x_n | x_n .* x_n]  (rows x 2D)
__global__ void xq_kernel(const double* __restrict__ X, int64_t rows, int D, double* __restrict__ Xq) {
    const int W = 2 * D;
    const int64_t total = rows * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / W;
        const int q = (int)(i - n * W);
        const double x = X[n * D + (q < D ? q : q - D)];
        Xq[i] = (q < D) ? x : x * x;
    }
}

int build_xq(Ctx* c, const double* X, int64_t rows, int D, double* Xq) {
    const int64_t total = rows * 2 * D;
    const int nb = (int)std::min<int64_t>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(xq_kernel, dim3(std::max(nb, 1)), dim3(256), 0, c->stream, X, rows, D, Xq);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// Gt[m] = [G2[m][0:D] | cs[m] | G2[m][D:2D]]  (+= when beta_acc == 1)
__global__ void assemble_gt_kernel(const double* __restrict__ G2, const double* __restrict__ cs, int M, int D,
                                   double beta_acc, double* __restrict__ Gt) {
    const int W = 2 * D + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * W) return;
    const int m = i / W, q = i - m * W;
    const double v = (q < D) ? G2[(size_t)m * 2 * D + q] : (q == D ? cs[m] : G2[(size_t)m * 2 * D + q - 1]);
    Gt[i] = (beta_acc != 0.0) ? Gt[i] + v : v;
}

int assemble_gt(Ctx* c, const double* G2, const double* cs, int M, int D, double beta_acc, double* Gt) {
    const int total = M * (2 * D + 1);
    hipLaunchKernelGGL(assemble_gt_kernel, dim3((total + 255) / 256), dim3(256), 0, c->stream, G2, cs, M, D, beta_acc, Gt);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// Finishing of the sparse gradients on the device: dZ and the per-block partial sums of the length-scale terms out of
// the three M x (2D+1) tables [T^T X | colsum T | T^T X.^2] of every part (row-pass table, Kmm term transposed and
// not).  Saves the 3.2 MB device -> host copy of the tables at Cm and the host's pass over freshly DMA-ed memory
// (0.2 ms per evaluation).  Block = 64 x 4 threads over 32 rows; sums over rows in a fixed order.
//   s[p][q]  = sum_m  gn[D+1+q] + cn z^2 - 2 z gn[q]  +  gt[D+1+q] + ct z^2 - 2 z gt[q]
//   dZ[m][q] = sum_p invl2[p][q] ((z cn - gn[q]) + (z ct - gt[q]) + (z cN - gN[q]))
constexpr int GF_ROWS = 32;
__global__ __launch_bounds__(256) void grad_finish_kernel(int np, int M, int Mr, int D, const double* __restrict__ gt_nm,
                                                         const double* __restrict__ gt_mm, const double* __restrict__ Z,
                                                         const double* __restrict__ invl, double* __restrict__ dZ,
                                                         double* __restrict__ spart) {
    __shared__ double sh[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int Wd = 2 * D + 1;
    const int m0 = blockIdx.x * GF_ROWS;
    for (int q0 = 0; q0 < D; q0 += 64) {
        const int q = q0 + tx;
        for (int p = 0; p < np; ++p) {
            const double* Gn = gt_nm + (size_t)p * M * Wd;
            const double* GT = gt_mm + (size_t)(2 * p) * M * Wd;
            const double* GN = gt_mm + (size_t)(2 * p + 1) * M * Wd;
            double s = 0.0;
            if (q < D) {
                const double il = invl[(size_t)p * D + q], il2 = il * il;
                for (int r = ty; r < GF_ROWS; r += 4) {
                    const int m = m0 + r;
                    if (m >= Mr) break;
                    const double* gn = Gn + (size_t)m * Wd;
                    const double* gt = GT + (size_t)m * Wd;
                    const double* gN = GN + (size_t)m * Wd;
                    const double z = Z[(size_t)m * D + q];
                    const double cn = gn[D], ct = gt[D], cN = gN[D];
                    s += (gn[D + 1 + q] + cn * z * z - 2.0 * z * gn[q]) + (gt[D + 1 + q] + ct * z * z - 2.0 * z * gt[q]);
                    const double d = il2 * ((z * cn - gn[q]) + (z * ct - gt[q]) + (z * cN - gN[q]));
                    double* o = dZ + (size_t)m * D + q;
                    *o = (p == 0) ? d : *o + d;
                }
            }
            sh[ty][tx] = s;
            __syncthreads();
            if (ty == 0 && q < D) spart[((size_t)blockIdx.x * np + p) * D + q] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
            __syncthreads();
        }
    }
}

int grad_finish(Ctx* c, int np, int M, int Mr, int D, const double* gt_nm, const double* gt_mm, const double* Z,
                const double* invl, double* dZ, double* spart, int* nblocks) {
    const int nb = (Mr + GF_ROWS - 1) / GF_ROWS;
    *nblocks = nb;
    hipLaunchKernelGGL(grad_finish_kernel, dim3((unsigned)nb), dim3(256), 0, c->stream, np, M, Mr, D, gt_nm, gt_mm, Z, invl,
                       dZ, spart);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int build_xtilde(Ctx* c, const double* X, int64_t rows, int D, double* Xt) {
    const int64_t total = rows * (2 * D + 1);
    const int nb = (int)std::min<int64_t>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(xtilde_kernel, dim3(std::max(nb, 1)), dim3(256), 0, c->stream, X, rows, D, Xt);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// out[n] = sum_j A[n][j] * B[n][j]; one wave per row
__global__ __launch_bounds__(256) void rowdot_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                     int64_t rows, int cols, int ld, double* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const double* a = A + (size_t)row * ld;
    const double* b = B + (size_t)row * ld;
    double s = 0.0;
    for (int j = lane; j < cols; j += 64) s += a[j] * b[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) out[row] = s;
}

int rowdot(Ctx* c, const double* A, const double* B, int64_t rows, int cols, int ld, double* out) {
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream, A, B, rows,
                       cols, ld, out);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// jac[n][q] (+)= invl[q]^2 * (x[n][q] * GT[n][D] - GT[n][q]);  GT is rows x (2D+1)
__global__ void jac_finish_kernel(const double* __restrict__ Xs, const double* __restrict__ GT, int64_t rows,
                                  int D, const double* __restrict__ invl, int accumulate,
                                  double* __restrict__ jac) {
    const int W = 2 * D + 1;
    const int64_t total = rows * D;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / D;
        const int q = (int)(i - n * D);
        const double il = invl[q];
        const double v = il * il * (Xs[i] * GT[n * W + D] - GT[n * W + q]);
        jac[i] = accumulate ? jac[i] + v : v;
    }
}

// var[n] = max(kdiag - quad[n], 1e-15)
__global__ void var_finish_kernel(double* var, int64_t rows, double kdiag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) var[i] = fmax(kdiag - var[i], 1e-15);
}

// Posterior mean (K wv), latent variance and mean Jacobian on one row chunk of
// device-resident test inputs.  Buffers: Kbuf / Gbuf are rows x Mp scratch.
// Ptw[m][:] = -wv[m] Pt[m][:]   (Mp x Wd)
__global__ void scale_rows_neg_kernel(const double* __restrict__ Pt, const double* __restrict__ wv, int Mp, int Wd,
                                      double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Mp * Wd) out[i] = -wv[i / Wd] * Pt[i];
}

// out[m][q] = -wv[m] Z[m][q] (q < D), zero padding up to ldz; with_wv: column D carries wv[m] itself, so that the
// product K out also yields K wv (the posterior mean = - sum_m G) in its column D
__global__ void scale_z_neg_kernel(const double* __restrict__ Z, const double* __restrict__ wv, int Mp, int D, int ldz,
                                   double* __restrict__ out, int with_wv, double sign) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mp * ldz) return;
    const int m = i / ldz, q = i - m * ldz;
    out[i] = (q < D) ? sign * wv[m] * Z[(size_t)m * D + q] : ((with_wv && q == D) ? wv[m] : 0.0);
}

// Every stationary kernel: with H = h(r) = (dk/dr)/r of the part and GW = H [diag(wv) Z | wv]
// (rows x ldz, column D = H wv):  jac[n][q] (+)= invl[q]^2 (x[n][q] GW[n][D] - GW[n][q])
__global__ void jac_finish_h_kernel(const double* __restrict__ Xs, const double* __restrict__ GW, int64_t rows, int D,
                                    int ldz, const double* __restrict__ invl, int accumulate, double* __restrict__ jac) {
    const int64_t total = rows * D;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / D;
        const int q = (int)(i - n * D);
        const double il = invl[q];
        const double v = il * il * (Xs[i] * GW[n * ldz + D] - GW[n * ldz + q]);
        jac[i] = accumulate ? jac[i] + v : v;
    }
}

// RBF: jac[n][q] = invl[q]^2 (GZ[n][q] ... ) with G = -K diag(wv):  sum_m G = -(K wv)[n]
//   jac[n][q] = invl[q]^2 * (-x[n][q] * kwv[n] - GZ[n][q]),  GZ = K (-diag(wv) Z)   (rows x ldz)
__global__ void jac_finish_rbf_kernel(const double* __restrict__ Xs, const double* __restrict__ GZ,
                                      const double* __restrict__ kwv, int64_t kwv_stride, int64_t rows, int D, int ldz,
                                      const double* __restrict__ invl, double* __restrict__ jac) {
    const int64_t total = rows * D;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / D;
        const int q = (int)(i - n * D);
        const double il = invl[q];
        jac[i] = il * il * (-(Xs[i] * kwv[n * kwv_stride]) - GZ[n * ldz + q]);
    }
}

int predict_chunk(Ctx* c, const KSpec& ks, const double* Xs, int64_t rows, const double* P, const double* Pt,
                  int Mp, const double* wv, const double* winv, double kdiag, double* Kbuf, double* Gbuf,
                  double* GT, double* mean, double* var, double* jac, bool k_ready, double* Hbuf, bool h_ready) {
    const int D = ks.D;
    // RBF: h(r) = -k(r), so the Jacobian's G = wv .* h(r) is K itself with -wv folded into [P | 1 | P.^2]:
    // no second pass over the pairs.  Every other stationary kernel takes h(r) from the MFMA build, which stores it
    // beside k (or has stored it: the training rows right after an evaluation), part by part for sum kernels: the
    // Jacobian is the product H [diag(wv) Z | wv] on the matrix pipe.  Only without the MFMA build does
    // the pair kernel recompute the distances on the VALU.
    const bool rbf1 = (ks.nparts == 1 && ks.kind[0] == EDRGP_KERN_RBF);
    const bool hmm = (jac != nullptr) && !rbf1 && c->mfma_kbuild && Hbuf != nullptr && rows <= 0x7fffffffLL;
    const bool gen1 = hmm && ks.nparts == 1;          // single part: k and h out of one build
    if (mean != nullptr || var != nullptr || (jac != nullptr && rbf1) || gen1) {
        if (!k_ready || (gen1 && !h_ready)) {
            c->kbuild_h = gen1 ? Hbuf : nullptr;
            const int st = kernel_matrix(c, ks, Xs, rows, P, Mp, Kbuf, Mp, 0, 0.0, nullptr, nullptr);
            c->kbuild_h = nullptr;
            if (st != EDRGP_OK) return st;
        }
        if (mean != nullptr) TRY(gemv_n(c, (int)rows, Mp, 1.0, Kbuf, Mp, wv, mean));
        if (var != nullptr) {
            // quad[n] = sum_m (K Winv)[n][m] K[n][m]
            TRY(gemm(c, false, false, (int)rows, Mp, Mp, 1.0, Kbuf, Mp, winv, Mp, 0.0, Gbuf, Mp));
            TRY(rowdot(c, Gbuf, Kbuf, rows, Mp, Mp, var));
            hipLaunchKernelGGL(var_finish_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream,
                               var, rows, kdiag);
        }
    }
    if (jac != nullptr) {
        const int Wd = 2 * D + 1;
        for (int p = 0; p < ks.nparts; ++p) {
            if (rbf1) {
                // The Jacobian needs sum_m G[n][m] Z[m][:] and sum_m G[n][m] only (the Z.^2 columns of Pt serve
                // the length-scale gradient, not this), and with G = -K diag(wv) the second is minus the
                // posterior mean: one rows x D x Mp product on an even leading dimension instead of
                // rows x (2D+1) x Mp on an odd one (a second, one-column tile and the guarded loaders: C4,
                // d = 64: 2.3 -> 0.5 ms).
                // When the mean itself is not asked for and one more column still fits the 64-wide column tile,
                // wv rides along as column D of the right operand: K wv comes out of the same pass over K instead of
                // a second one (a rows x Mp gemv: 1.6 GB at Cm).
                const bool ride = (mean == nullptr) && (D + 2 <= 64);
                const int ldz = ride ? (D + 2 - (D & 1)) : D + (D & 1);       // even leading dimension
                TRY(c->ptw.reserve((size_t)Mp * ldz * sizeof(double)));
                hipLaunchKernelGGL(scale_z_neg_kernel, dim3((Mp * ldz + 255) / 256), dim3(256), 0, c->stream, P, wv, Mp, D,
                                   ldz, c->ptw.d(), ride ? 1 : 0, -1.0);
                TRY(gemm(c, false, false, (int)rows, ldz, Mp, 1.0, Kbuf, Mp, c->ptw.d(), ldz, 0.0, GT, ldz));
                const double* kwv = mean;
                int64_t kwv_stride = 1;
                if (ride) {
                    kwv = GT + D;
                    kwv_stride = ldz;
                } else if (kwv == nullptr) {
                    double* tmp = GT + (size_t)rows * ldz;          // GT holds rows x (2D+1) >= rows x (ldz + 1)
                    TRY(gemv_n(c, (int)rows, Mp, 1.0, Kbuf, Mp, wv, tmp));
                    kwv = tmp;
                }
                const int64_t total = rows * D;
                const int nb = (int)std::min<int64_t>(4096, (total + 255) / 256);
                hipLaunchKernelGGL(jac_finish_rbf_kernel, dim3(std::max(nb, 1)), dim3(256), 0, c->stream, Xs, GT, kwv,
                                   kwv_stride, rows, D, ldz, ks.invl, jac);
                continue;
            } else if (hmm) {
                if (!gen1) {
                    // part p alone: its k goes to the scratch (unused), its h(r) to Hbuf
                    KSpec k1 = ks;
                    k1.nparts = 1;
                    k1.kind[0] = ks.kind[p];
                    k1.var[0] = ks.var[p];
                    k1.invl = ks.invl + (size_t)p * D;
                    c->kbuild_h = Hbuf;
                    const int st = kernel_matrix(c, k1, Xs, rows, P, Mp, Gbuf, Mp, 0, 0.0, nullptr, nullptr);
                    c->kbuild_h = nullptr;
                    if (st != EDRGP_OK) return st;
                }
                const int ldz = D + 2 - (D & 1);                 // [diag(wv) Z | wv | (0)], even leading dimension
                TRY(c->ptw.reserve((size_t)Mp * ldz * sizeof(double)));
                hipLaunchKernelGGL(scale_z_neg_kernel, dim3((Mp * ldz + 255) / 256), dim3(256), 0, c->stream, P, wv, Mp, D,
                                   ldz, c->ptw.d(), 1, 1.0);
                TRY(gemm(c, false, false, (int)rows, ldz, Mp, 1.0, Hbuf, Mp, c->ptw.d(), ldz, 0.0, GT, ldz));
                const int64_t total = rows * D;
                const int nb = (int)std::min<int64_t>(4096, (total + 255) / 256);
                hipLaunchKernelGGL(jac_finish_h_kernel, dim3(std::max(nb, 1)), dim3(256), 0, c->stream, Xs, GT, rows, D, ldz,
                                   ks.invl + (size_t)p * D, p > 0 ? 1 : 0, jac);
                continue;
            } else {
                TRY(hmatrix(c, ks, p, Xs, rows, P, Mp, wv, Gbuf, Mp));
                TRY(gemm(c, false, false, (int)rows, Wd, Mp, 1.0, Gbuf, Mp, Pt, Wd, 0.0, GT, Wd));
            }
            const int64_t total = rows * D;
            const int nb = (int)std::min<int64_t>(4096, (total + 255) / 256);
            hipLaunchKernelGGL(jac_finish_kernel, dim3(std::max(nb, 1)), dim3(256), 0, c->stream, Xs, GT, rows, D,
                               ks.invl + (size_t)p * D, p > 0 ? 1 : 0, jac);
        }
    }
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

}  // namespace edrgp
