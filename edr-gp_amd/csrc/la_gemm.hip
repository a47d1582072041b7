// fp64 MFMA GEMM for gfx950 (v_mfma_f64_16x16x4_f64), row-major.
//
// One kernel template covers every dense contraction on the GP path:
//   * Phi = K_nm^T K_nm  (TN, lower-triangular tile set, split over rows)   -- the hot kernel
//   * W   = K_nm S       (NN, gradient contraction)
//   * the O(M^3) products around the two Choleskys (NN / NT / TN, batched for trtri)
//
// Workgroup = 256 threads = 4 waves in a 2x2 grid; a wave owns a (BM/2)x(BN/2)
// sub-tile made of 16x16 MFMA blocks.  Operand tiles are staged in LDS in
// k-major order ([k][m], m contiguous) so that an MFMA fragment read is
// 16 lanes x 8 B contiguous per k (ds_read_b64); the row stride BM+16 doubles
// puts the two k-rows of a 32-lane group on disjoint halves of the 64 banks.
// Global->register prefetch of tile t+1 overlaps the MFMAs of tile t; one
// barrier per k-tile (double-buffered LDS).
//
// MFMA fragment maps (cdna_hip_programming.md section 3, f64 exception):
//   A operand: lane l holds A[i = l&15][k = l>>4]
//   B operand: lane l holds B[k = l>>4][j = l&15]
//   C/D      : reg r of lane l is C[row = (l>>4) + 4r][col = l&15]
#include "common.h"

namespace edrgp {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;
constexpr int LPAD = 16;

struct GemmArgs {
    const double* A;
    const double* B;
    double* C;
    int M, N, K, lda, ldb, ldc;
    double alpha, beta;
    long long strideA, strideB, strideC;
    int limM_base, limM_stride, limK_base, limK_stride;
    int splits;
    long long slabStride;
    int tiles_m, tiles_n, tri;
    int vecA, vecB;
    int kmode;          // triangular operands: bit0 k <= row tile end, bit1 k >= row tile start,
                        //                      bit2 k <= col tile end, bit3 k >= col tile start
};

// Stage loader.  KM = true : source element (k, i) at P[k*ld + i]
//                KM = false: source element (k, i) at P[i*ld + k]
template <int BMN, bool KM>
struct TileLoader {
    static constexpr int NV = BMN * BK / 512;     // double2 per thread
    double2 r[NV];

    __device__ __forceinline__ void load(const double* __restrict__ P, int ld, int mn0, int k0,
                                         int mn_lim, int k_lim, bool vec, int tid) {
        if (KM) {
            constexpr int TPR = BMN / 2;          // threads per k-row
            constexpr int RPP = 256 / TPR;        // k-rows per pass
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int k = k0 + v * RPP + tid / TPR;
                const int i = mn0 + (tid % TPR) * 2;
                const double* p = P + (size_t)k * ld + i;
                if (vec && k < k_lim && i + 1 < mn_lim) {
                    r[v] = *reinterpret_cast<const double2*>(p);
                } else {
                    r[v].x = (k < k_lim && i < mn_lim) ? p[0] : 0.0;
                    r[v].y = (k < k_lim && i + 1 < mn_lim) ? p[1] : 0.0;
                }
            }
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = mn0 + v * 32 + tid / 8;
                const int k = k0 + (tid % 8) * 2;
                const double* p = P + (size_t)i * ld + k;
                if (vec && i < mn_lim && k + 1 < k_lim) {
                    r[v] = *reinterpret_cast<const double2*>(p);
                } else {
                    r[v].x = (i < mn_lim && k < k_lim) ? p[0] : 0.0;
                    r[v].y = (i < mn_lim && k + 1 < k_lim) ? p[1] : 0.0;
                }
            }
        }
    }

    __device__ __forceinline__ void store(double (*S)[BMN + LPAD], int tid) const {
        if (KM) {
            constexpr int TPR = BMN / 2;
            constexpr int RPP = 256 / TPR;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int kl = v * RPP + tid / TPR;
                const int il = (tid % TPR) * 2;
                *reinterpret_cast<double2*>(&S[kl][il]) = r[v];
            }
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int il = v * 32 + tid / 8;
                const int kl = (tid % 8) * 2;
                S[kl][il] = r[v].x;
                S[kl + 1][il] = r[v].y;
            }
        }
    }
};

// TAG only changes the kernel symbol, so the Phi syrk (1) and the gradient
// contraction W = K S (2) show up under their own names in rocprofv3 traces.
template <int BM, int BN, bool AKM, bool BKM, int TAG>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 16, TN = WN / 16;
    __shared__ __attribute__((aligned(16))) double As[2][BK][BM + LPAD];
    __shared__ __attribute__((aligned(16))) double Bs[2][BK][BN + LPAD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;

    // ---- tile id with an XCD-aware (bijective) remap: blocks b, b+8, ... share
    // an XCD (L2), give them consecutive tiles so they share operand panels.
    const int nt = gridDim.x;
    int t;
    {
        const int bid = blockIdx.x;
        const int q = nt >> 3, r = nt & 7, x = bid & 7, in = bid >> 3;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + in;
    }
    int ti, tj;
    if (g.tri) {
        ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while ((long long)ti * (ti + 1) / 2 > t) --ti;
        tj = t - (int)((long long)ti * (ti + 1) / 2);
    } else {
        ti = t / g.tiles_n;
        tj = t - ti * g.tiles_n;
    }

    const int z = blockIdx.z;
    int Meff = g.M, Keff = g.K;
    if (g.limM_stride) Meff = min(Meff, g.limM_base - z * g.limM_stride);
    if (g.limK_stride) Keff = min(Keff, g.limK_base - z * g.limK_stride);
    const int m0 = ti * BM, n0 = tj * BN;
    if (Meff <= 0 || m0 >= Meff || n0 >= g.N) return;

    const double* A = g.A + (long long)z * g.strideA;
    const double* B = g.B + (long long)z * g.strideB;
    double* C = g.C + (long long)z * g.strideC + (long long)blockIdx.y * g.slabStride;

    // ---- split-K range (in units of BK)
    const int kt_total = (Keff + BK - 1) / BK;
    const int kt_per = (kt_total + g.splits - 1) / g.splits;
    int kt0 = blockIdx.y * kt_per;
    int kt1 = min(kt_total, kt0 + kt_per);
    if (g.kmode) {          // skip k-tiles where a triangular operand is identically zero
        int k_lo = 0, k_hi = Keff;
        if (g.kmode & 1) k_hi = min(k_hi, m0 + BM);
        if (g.kmode & 2) k_lo = max(k_lo, m0);
        if (g.kmode & 4) k_hi = min(k_hi, n0 + BN);
        if (g.kmode & 8) k_lo = max(k_lo, n0);
        kt0 = max(kt0, k_lo / BK);
        kt1 = min(kt1, (k_hi + BK - 1) / BK);
    }

    d4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    if (kt0 < kt1) {
        // interior tiles may use 16-byte loads
        const bool vA = g.vecA, vB = g.vecB;
        TileLoader<BM, AKM> la;
        TileLoader<BN, BKM> lb;
        la.load(A, g.lda, m0, kt0 * BK, Meff, Keff, vA, tid);
        lb.load(B, g.ldb, n0, kt0 * BK, g.N, Keff, vB, tid);
        la.store(As[0], tid);
        lb.store(Bs[0], tid);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            const bool more = (kt + 1 < kt1);
            if (more) {
                la.load(A, g.lda, m0, (kt + 1) * BK, Meff, Keff, vA, tid);
                lb.load(B, g.ldb, n0, (kt + 1) * BK, g.N, Keff, vB, tid);
            }
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double a[TM], b[TN];
#pragma unroll
                for (int x = 0; x < TM; ++x) a[x] = As[buf][kk * 4 + lq][wm * WM + x * 16 + lr];
#pragma unroll
                for (int x = 0; x < TN; ++x) b[x] = Bs[buf][kk * 4 + lq][wn * WN + x * 16 + lr];
#pragma unroll
                for (int x = 0; x < TM; ++x)
#pragma unroll
                    for (int y = 0; y < TN; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
            }
            if (more) {
                la.store(As[buf ^ 1], tid);
                lb.store(Bs[buf ^ 1], tid);
            }
            __syncthreads();
        }
    }

    // ---- epilogue
    const double alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int x = 0; x < TM; ++x) {
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int col = n0 + wn * WN + y * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + x * 16 + lq + 4 * r;
                if (row < Meff && col < g.N) {
                    double* p = C + (size_t)row * g.ldc + col;
                    double v = alpha * acc[x][y][r];
                    if (beta != 0.0) v += beta * (*p);
                    *p = v;
                }
            }
        }
    }
}

// C[i][j] = alpha * sum_s slab[s][i][j] + beta * C[i][j]; mirror: computed for
// i >= j and written to both triangles.
__global__ void slab_reduce_kernel(const double* __restrict__ slabs, int splits, long long slabStride,
                                   int M, int N, int lds, double alpha, double beta,
                                   double* __restrict__ C, int ldc, int mirror) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= M || j >= N) return;
    if (mirror && j > i) return;
    double s = 0.0;
    const double* p = slabs + (size_t)i * lds + j;
    for (int k = 0; k < splits; ++k) s += p[(size_t)k * slabStride];
    double v = alpha * s;
    double* c = C + (size_t)i * ldc + j;
    if (beta != 0.0) v += beta * (*c);
    *c = v;
    if (mirror && i != j) C[(size_t)j * ldc + i] = v;
}

template <int BM, int BN>
static void launch_variant(bool akm, bool bkm, int tag, dim3 grid, hipStream_t s, const GemmArgs& g) {
    if (tag == 1 && akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, true, 1>), grid, dim3(256), 0, s, g);
    else if (tag == 2 && !akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, true, 2>), grid, dim3(256), 0, s, g);
    else if (akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, true, 0>), grid, dim3(256), 0, s, g);
    else if (akm && !bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, false, 0>), grid, dim3(256), 0, s, g);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, true, 0>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, false, 0>), grid, dim3(256), 0, s, g);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int gemm_impl(Ctx* c, bool transA, bool transB, int M, int N, int K, double alpha,
                     const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc,
                     const GemmOpts& o, int splits, double* slab, long long slabStride, int slab_ld) {
    if (M <= 0 || N <= 0) return EDRGP_OK;
    int tile = o.force_tile;
    if (tile == 0) {
        const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * o.batch * splits;
        tile = (M >= 128 && N >= 128 && t128 >= 96) ? 128 : 64;
    }
    GemmArgs g;
    g.A = A; g.B = B;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.strideA = o.strideA; g.strideB = o.strideB; g.strideC = o.strideC;
    g.limM_base = o.limM_base; g.limM_stride = o.limM_stride;
    g.limK_base = o.limK_base; g.limK_stride = o.limK_stride;
    g.splits = splits;
    g.tri = o.tri;
    g.kmode = o.kmode;
    g.tiles_m = (M + tile - 1) / tile;
    g.tiles_n = (N + tile - 1) / tile;
    g.vecA = aligned16(A) && (lda % 2 == 0) && (o.strideA % 2 == 0);
    g.vecB = aligned16(B) && (ldb % 2 == 0) && (o.strideB % 2 == 0);
    if (slab != nullptr) {
        g.C = slab; g.ldc = slab_ld; g.alpha = 1.0; g.beta = 0.0; g.slabStride = slabStride;
    } else {
        g.C = C; g.ldc = ldc; g.alpha = alpha; g.beta = beta; g.slabStride = 0;
    }
    const long long nt = o.tri ? (long long)g.tiles_m * (g.tiles_m + 1) / 2 : (long long)g.tiles_m * g.tiles_n;
    dim3 grid((unsigned)nt, (unsigned)splits, (unsigned)o.batch);
    // op(A) is M x K: not transposed -> stored [M][K] (m-major); transposed -> stored [K][M] (k-major)
    const bool akm = transA;
    // op(B) is K x N: not transposed -> stored [K][N] (k-major); transposed -> stored [N][K]
    const bool bkm = !transB;
    if (tile == 128) launch_variant<128, 128>(akm, bkm, o.tag, grid, c->stream, g);
    else launch_variant<64, 64>(akm, bkm, o.tag, grid, c->stream, g);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int gemm(Ctx* c, bool transA, bool transB, int M, int N, int K, double alpha, const double* A, int lda,
         const double* B, int ldb, double beta, double* C, int ldc, const GemmOpts& o) {
    // split-K only for long-K, small-output, unbatched products (e.g. T^T [X|1|X^2])
    int splits = 1;
    if (o.batch == 1 && !o.tri) {
        const int tile = (o.force_tile ? o.force_tile : 64);
        const long long nt = (long long)((M + tile - 1) / tile) * ((N + tile - 1) / tile);
        const int kt = (K + BK - 1) / BK;
        if (nt < 256 && kt >= 256) {
            splits = (int)std::min<long long>((1024 + nt - 1) / nt, kt / 32);
            if (splits < 1) splits = 1;
        }
    }
    if (splits > 1) {
        const long long slabStride = (long long)M * N;
        TRY(c->slabs.reserve((size_t)splits * slabStride * sizeof(double)));
        GemmOpts o2 = o;
        if (!o2.force_tile) o2.force_tile = 64;
        TRY(gemm_impl(c, transA, transB, M, N, K, 1.0, A, lda, B, ldb, 0.0, nullptr, 0, o2, splits,
                      c->slabs.d(), slabStride, N));
        dim3 blk(64, 4), grd((N + 63) / 64, (M + 3) / 4);
        hipLaunchKernelGGL(slab_reduce_kernel, grd, blk, 0, c->stream, c->slabs.d(), splits, slabStride, M, N,
                           N, alpha, beta, C, ldc, 0);
        HIP_TRY(hipGetLastError());
        return EDRGP_OK;
    }
    return gemm_impl(c, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, o, 1, nullptr, 0, 0);
}

__global__ void mirror_lower_kernel(double* __restrict__ A, int n, int lda) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i < n && j < i) A[(size_t)j * lda + i] = A[(size_t)i * lda + j];
}

int mirror_lower(Ctx* c, double* A, int n, int lda) {
    dim3 blk(64, 4), grd((n + 63) / 64, (n + 3) / 4);
    hipLaunchKernelGGL(mirror_lower_kernel, grd, blk, 0, c->stream, A, n, lda);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int slab_reduce_mirror(Ctx* c, const double* slabs, int splits, long long slabStride, int M, double* Phi) {
    dim3 blk(64, 4), grd((M + 63) / 64, (M + 3) / 4);
    hipLaunchKernelGGL(slab_reduce_kernel, grd, blk, 0, c->stream, slabs, splits, slabStride, M, M, M, 1.0, 0.0, Phi,
                       M, 1);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int syrk_tn(Ctx* c, const double* K, int64_t rows, int M, int ldk, double* Phi, int ldphi, bool accumulate) {
    if (rows > 0x7fffffffLL) { set_error("syrk_tn: rows chunk too large"); return EDRGP_ERR_INVALID; }
    const int tile = (M >= 256) ? 128 : 64;
    const int tm = (M + tile - 1) / tile;
    const long long nt = (long long)tm * (tm + 1) / 2;
    const int kt = (int)((rows + BK - 1) / BK);
    int splits = c->syrk_splits;
    if (splits <= 0) {
        // 2 workgroups per CU are resident: fill whole rounds of 2 * num_cu
        // workgroups (4 rounds) and never start a nearly empty extra round.
        const long long slots = 2LL * c->num_cu;
        splits = (int)std::max<long long>(1, (4 * slots) / nt);
        const int max_splits = std::max(1, kt / 16);          // >= 256 rows per split
        if (splits > max_splits) splits = max_splits;
    }
    if (splits < 1) splits = 1;
    if (splits > kt) splits = std::max(1, kt);
    const long long slabStride = (long long)M * M;
    TRY(c->slabs.reserve((size_t)splits * slabStride * sizeof(double)));
    GemmOpts o;
    o.tri = 1;
    o.tag = 1;
    o.force_tile = tile;
    {
        ProfScope ps(c, EDRGP_PROF_SYRK);
        // Phi_slab[s] = K[rows_s]^T K[rows_s]  (lower tiles)
        TRY(gemm_impl(c, true, false, M, M, (int)rows, 1.0, K, ldk, K, ldk, 0.0, nullptr, 0, o, splits,
                      c->slabs.d(), slabStride, M));
    }
    dim3 blk(64, 4), grd((M + 63) / 64, (M + 3) / 4);
    hipLaunchKernelGGL(slab_reduce_kernel, grd, blk, 0, c->stream, c->slabs.d(), splits, slabStride, M, M, M,
                       1.0, accumulate ? 1.0 : 0.0, Phi, ldphi, 1);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

}  // namespace edrgp
