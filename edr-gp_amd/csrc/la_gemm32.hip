// fp32 row pass (BASELINE config 5: "fp32 path, MFMA fp32"), gfx950 v_mfma_f32_16x16x4_f32.
//
// Mixed precision by construction: everything whose size grows with the number of rows N
// (scaled inputs, K_nm, the Phi = K_nm^T K_nm contraction, the gradient contraction
// T = (K_nm S2 + beta y wv^T) .* h(r) and T^T [X | X.^2]) is stored and multiplied in fp32 with
// fp32 MFMA accumulation inside one split-K slab; slabs, per-tile partial sums and every M x M
// object (Kmm, both Choleskys, the posterior factors) stay fp64.  The O(M^3) phase therefore
// keeps the fp64 code of the default path and only the N-proportional HBM traffic and MFMA
// time are halved.
//
// Kernel structure mirrors la_gemm.hip (4 waves in a 2x2 grid, double-buffered LDS k-tiles,
// interior fast-path loader with wave-uniform tile base, staging at raised priority); the
// differences are the element type (16-byte vectors hold 4 values, so BK = 32 keeps the tile
// bytes and the MFMA count per barrier) and the C/D fragment map:
//   A operand: lane l holds A[i = l&15][k = l>>4]        (same as f64)
//   B operand: lane l holds B[k = l>>4][j = l&15]        (same as f64)
//   C/D      : reg r of lane l is C[row = 4 (l>>4) + r][col = l&15]   (f64 uses (l>>4) + 4r)
#include "common.h"
#include "kfun.h"
#include <algorithm>
#include <cstdlib>

namespace edrgp {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BK32 = 32;
constexpr int PAD32 = 16;

__device__ __forceinline__ void kfun32(int kind, float var, float r2, float& K, float& h) {
    if (kind == EDRGP_KERN_RBF) {
        K = var * expf(-0.5f * r2);
        h = -K;
    } else if (kind == EDRGP_KERN_MATERN32) {
        const float r = sqrtf(r2);
        const float e = expf(-(float)EDRGP_SQRT3 * r);
        K = var * (1.0f + (float)EDRGP_SQRT3 * r) * e;
        h = -3.0f * var * e;
    } else {
        const float r = sqrtf(r2);
        const float e = expf(-(float)EDRGP_SQRT5 * r);
        K = var * (1.0f + (float)EDRGP_SQRT5 * r + (5.0f / 3.0f) * r2) * e;
        h = -(5.0f / 3.0f) * var * (1.0f + (float)EDRGP_SQRT5 * r) * e;
    }
}

struct Fuse32K {            // kernel-matrix epilogue (TAG 3)
    const float* xn;        // rows : |x/l|^2
    const float* zn;        // cols : |z/l|^2
    const double* y;        // rows or null
    double* bpart;          // tiles_m x N partial sums of K^T y
    float var;
    int kind;
};

struct Fuse32T {            // gradient epilogue (TAG 2)
    const float* Xs;        // rows x D scaled inputs (x / l)
    const float* Zs;        // N x D scaled inducing inputs
    const double* y;        // rows
    const double* wv;       // N
    double* dvpart;         // one partial per workgroup
    double* cspart;         // tiles_m x N partial column sums of T
    float ycoef, var;
    int D, kind;
};

struct G32Args {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, lda, ldb, ldc;
    int splits;
    long long slabStride;
    int tiles_n, tri;
    int vecA, vecB;
    int prio;
    Fuse32K fk;
    Fuse32T ft;
};

// Stage loader.  KM = true : source element (k, i) at P[k*ld + i]
//                KM = false: source element (k, i) at P[i*ld + k]
template <int BMN, bool KM>
struct Loader32 {
    static constexpr int NV = BMN * BK32 / 1024;      // float4 per thread
    static constexpr int TPR = BMN / 4;               // KM: threads per k-row
    static constexpr int RPP = 256 / TPR;             // KM: k-rows per pass
    f4 r[NV];

    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int mn0, int k0, int mn_lim,
                                         int k_lim, bool vec, int tid) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (KM) {
                const int k = k0 + v * RPP + tid / TPR;
                const int i = mn0 + (tid % TPR) * 4;
                const float* p = P + (size_t)k * ld + i;
                if (vec && k < k_lim && i + 3 < mn_lim) {
                    r[v] = *reinterpret_cast<const f4*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[v][e] = (k < k_lim && i + e < mn_lim) ? p[e] : 0.0f;
                }
            } else {
                const int i = mn0 + v * 32 + tid / 8;
                const int k = k0 + (tid % 8) * 4;
                const float* p = P + (size_t)i * ld + k;
                if (vec && i < mn_lim && k + 3 < k_lim) {
                    r[v] = *reinterpret_cast<const f4*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[v][e] = (i < mn_lim && k + e < k_lim) ? p[e] : 0.0f;
                }
            }
        }
    }

    // interior fast path: per-thread 32-bit byte offsets inside a k-tile, wave-uniform tile base
    __device__ __forceinline__ void offsets(unsigned (&off)[NV], int ld, int tid) const {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            off[v] = KM ? (unsigned)(((v * RPP + tid / TPR) * ld + (tid % TPR) * 4) * 4)
                        : (unsigned)(((v * 32 + tid / 8) * ld + (tid % 8) * 4) * 4);
    }
    static __device__ __forceinline__ long long tile_origin(int ld, int mn0, int kt) {
        return KM ? (long long)kt * BK32 * ld + mn0 : (long long)mn0 * ld + (long long)kt * BK32;
    }
    static __device__ __forceinline__ long long tile_step(int ld) { return KM ? (long long)BK32 * ld * 4 : (long long)BK32 * 4; }
    __device__ __forceinline__ void load_fast(const char* __restrict__ tile_base, const unsigned (&off)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; ++v) r[v] = *reinterpret_cast<const f4*>(tile_base + off[v]);
    }

    // LDS image of one k-tile: k-major sources keep [k][m] rows (stride BMN+16 floats: the four
    // k-rows of a fragment read fall on disjoint quarters of the 64 banks); m-major sources keep
    // [m][k] rows (stride BK+4 = 36 floats = 4 x odd: the 16 rows x 4 k of a fragment read hit
    // 64 distinct banks).  Both are filled with 16-byte stores.
    static constexpr int SIZE = KM ? BK32 * (BMN + PAD32) : BMN * (BK32 + 4);
    static __device__ __forceinline__ int idx(int k, int m) { return KM ? k * (BMN + PAD32) + m : m * (BK32 + 4) + k; }
    __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (KM) *reinterpret_cast<f4*>(&S[idx(v * RPP + tid / TPR, (tid % TPR) * 4)]) = r[v];
            else *reinterpret_cast<f4*>(&S[idx((tid % 8) * 4, v * 32 + tid / 8)]) = r[v];
        }
    }
};

// TAG: 0 plain store, 1 Phi syrk (symbol only), 2 gradient epilogue, 3 kernel-matrix epilogue
#ifndef EDRGP_FRAG_PIPE32
#define EDRGP_FRAG_PIPE32 1
#endif
template <int BM, int BN, bool AKM, bool BKM, int TAG>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(G32Args g) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr bool EARLY_STORE = (BM == 128);
    constexpr bool FRAG_PIPE32 = (EDRGP_FRAG_PIPE32 != 0) && BM == 128 && BN == 128;
    typedef Loader32<BM, AKM> LA;
    typedef Loader32<BN, BKM> LB;
    __shared__ __attribute__((aligned(16))) float As[2][LA::SIZE];
    __shared__ __attribute__((aligned(16))) float Bs[2][LB::SIZE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware bijective remap (blocks b, b+8, ... share an L2: give them consecutive tiles)
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
    const int m0 = ti * BM, n0 = tj * BN;
    if (m0 >= g.M || n0 >= g.N) return;
    const float* A = g.A;
    const float* B = g.B;
    float* C = g.C + (long long)blockIdx.y * g.slabStride;

    const int kt_total = (g.K + BK32 - 1) / BK32;
    const int kt_per = (kt_total + g.splits - 1) / g.splits;
    const int kt0 = blockIdx.y * kt_per;
    const int kt1 = min(kt_total, kt0 + kt_per);

    f4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    if (kt0 < kt1) {
        const bool vA = g.vecA, vB = g.vecB;
        LA la;
        LB lb;
        const bool fast = vA && vB && (m0 + BM <= g.M) && (n0 + BN <= g.N) && (kt1 * BK32 <= g.K) &&
                          ((long long)(AKM ? BK32 : BM) * g.lda < (1LL << 29)) &&
                          ((long long)(BKM ? BK32 : BN) * g.ldb < (1LL << 29));
        unsigned offA[LA::NV], offB[LB::NV];
        const char* baseA = reinterpret_cast<const char*>(A + LA::tile_origin(g.lda, m0, kt0));
        const char* baseB = reinterpret_cast<const char*>(B + LB::tile_origin(g.ldb, n0, kt0));
        const long long stepA = LA::tile_step(g.lda), stepB = LB::tile_step(g.ldb);
        if (fast) {
            la.offsets(offA, g.lda, tid);
            lb.offsets(offB, g.ldb, tid);
            la.load_fast(baseA, offA);
            lb.load_fast(baseB, offB);
        } else {
            la.load(A, g.lda, m0, kt0 * BK32, g.M, g.K, vA, tid);
            lb.load(B, g.ldb, n0, kt0 * BK32, g.N, g.K, vB, tid);
        }
        la.store(As[0], tid);
        lb.store(Bs[0], tid);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            const bool more = (kt + 1 < kt1);
            if (more) {
                if (fast) {
                    baseA += stepA;
                    baseB += stepB;
                    la.load_fast(baseA, offA);
                    lb.load_fast(baseB, offB);
                } else {
                    la.load(A, g.lda, m0, (kt + 1) * BK32, g.M, g.K, vA, tid);
                    lb.load(B, g.ldb, n0, (kt + 1) * BK32, g.N, g.K, vB, tid);
                }
            }
            if (g.prio) __builtin_amdgcn_s_setprio(0);
            if constexpr (FRAG_PIPE32) {
                // fragments of k-step kk + 1 read under the MFMAs of step kk: the two-stage register pipeline of the
                // fp64 kernels (la_gemm.hip), pinned with scheduling groups (fp32 contraction at config 5: -2.2 %;
                // groups of 4 MFMAs instead of 2: +1 %)
                float a[2][TM], b[2][TN];
#pragma unroll
                for (int x = 0; x < TM; ++x) a[0][x] = As[buf][LA::idx(lq, wm * WM + x * 16 + lr)];
#pragma unroll
                for (int x = 0; x < TN; ++x) b[0][x] = Bs[buf][LB::idx(lq, wn * WN + x * 16 + lr)];
#pragma unroll
                for (int kk = 0; kk < BK32 / 4; ++kk) {
                    const int cur = kk & 1, nxt = cur ^ 1;
                    if (kk + 1 < BK32 / 4) {
#pragma unroll
                        for (int x = 0; x < TM; ++x) a[nxt][x] = As[buf][LA::idx((kk + 1) * 4 + lq, wm * WM + x * 16 + lr)];
#pragma unroll
                        for (int x = 0; x < TN; ++x) b[nxt][x] = Bs[buf][LB::idx((kk + 1) * 4 + lq, wn * WN + x * 16 + lr)];
                    }
#pragma unroll
                    for (int x = 0; x < TM; ++x)
#pragma unroll
                        for (int y = 0; y < TN; ++y)
                            acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][x], b[cur][y], acc[x][y], 0, 0, 0);
                    if (kk + 1 < BK32 / 4) {
#pragma unroll
                        for (int i = 0; i < TM + TN; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
                    if (EARLY_STORE && kk == BK32 / 4 - 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) {
                            la.store(As[buf ^ 1], tid);
                            lb.store(Bs[buf ^ 1], tid);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else
#pragma unroll
            for (int kk = 0; kk < BK32 / 4; ++kk) {
                float a[TM], b[TN];
#pragma unroll
                for (int x = 0; x < TM; ++x) a[x] = As[buf][LA::idx(kk * 4 + lq, wm * WM + x * 16 + lr)];
#pragma unroll
                for (int x = 0; x < TN; ++x) b[x] = Bs[buf][LB::idx(kk * 4 + lq, wn * WN + x * 16 + lr)];
#pragma unroll
                for (int x = 0; x < TM; ++x)
#pragma unroll
                    for (int y = 0; y < TN; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], acc[x][y], 0, 0, 0);
                if (EARLY_STORE && kk == BK32 / 4 - 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        la.store(As[buf ^ 1], tid);
                        lb.store(Bs[buf ^ 1], tid);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (g.prio) __builtin_amdgcn_s_setprio(2);
            if (!EARLY_STORE && more) {
                la.store(As[buf ^ 1], tid);
                lb.store(Bs[buf ^ 1], tid);
            }
            __syncthreads();
        }
    }

    // ---- kernel-matrix epilogue: acc = <x/l, z/l>  ->  K = var k(r2), partial sums of K^T y
    if (TAG == 3) {
        const Fuse32K& f = g.fk;
        float zn[TN], bsum[TN];
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int col = n0 + wn * WN + y * 16 + lr;
            zn[y] = (col < g.N) ? f.zn[col] : 0.0f;
            bsum[y] = 0.0f;
        }
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + x * 16 + 4 * lq + r;
                const bool rok = row < g.M;
                const float xn = rok ? f.xn[row] : 0.0f;
                const float yr = (rok && f.y != nullptr) ? (float)f.y[row] : 0.0f;
#pragma unroll
                for (int y = 0; y < TN; ++y) {
                    const int col = n0 + wn * WN + y * 16 + lr;
                    const float r2 = fmaxf(fmaf(-2.0f, acc[x][y][r], xn + zn[y]), 0.0f);
                    float Kv, hv;
                    kfun32(f.kind, f.var, r2, Kv, hv);
                    if (rok && col < g.N) {
                        C[(size_t)row * g.ldc + col] = Kv;
                        bsum[y] = fmaf(Kv, yr, bsum[y]);
                    }
                }
            }
        if (f.y != nullptr) {
            __syncthreads();
            float* red = &As[0][0];
#pragma unroll
            for (int y = 0; y < TN; ++y) red[((wm * 4 + lq) * BN) + wn * WN + y * 16 + lr] = bsum[y];
            __syncthreads();
            for (int j = tid; j < BN; j += 256) {
                double sum = 0.0;
#pragma unroll
                for (int p = 0; p < 8; ++p) sum += (double)red[p * BN + j];
                if (n0 + j < g.N) f.bpart[(size_t)ti * g.N + n0 + j] = sum;
            }
        }
        return;
    }

    // ---- gradient epilogue: T = (acc + ycoef y wv^T) .* h(r), partial sums of Q .* k / var and of T's columns
    if (TAG == 2) {
        constexpr int DC = 64;                               // whole feature dimension (host checks D <= 64)
        static_assert(2 * LA::SIZE >= DC * BM && 2 * LB::SIZE >= DC * BN, "LDS too small for the fused epilogue");
        float* Xs = &As[0][0];
        float* Zs = &Bs[0][0];
        const Fuse32T& f = g.ft;
        const bool rbf = (f.kind == EDRGP_KERN_RBF);        // h(r) = -k(r): k is the A operand itself
        __syncthreads();
        if (!rbf)
        for (int e = tid; e < BM * f.D; e += 256) {
            const int i = e / f.D, q = e - i * f.D;
            Xs[q * BM + i] = (m0 + i < g.M) ? f.Xs[(size_t)(m0 + i) * f.D + q] : 0.0f;
        }
        if (!rbf)
        for (int e = tid; e < BN * f.D; e += 256) {
            const int j = e / f.D, q = e - j * f.D;
            Zs[q * BN + j] = (n0 + j < g.N) ? f.Zs[(size_t)(n0 + j) * f.D + q] : 0.0f;
        }
        __syncthreads();
        float wvc[TN], csum[TN];
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int col = n0 + wn * WN + y * 16 + lr;
            wvc[y] = (col < g.N) ? (float)f.wv[col] : 0.0f;
            csum[y] = 0.0f;
        }
        double dv = 0.0;
#pragma unroll
        for (int x = 0; x < TM; ++x) {
            float r2[TN][4];
#pragma unroll
            for (int y = 0; y < TN; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) r2[y][r] = 0.0f;
            for (int q = 0; !rbf && q < f.D; ++q) {
                float zc[TN];
#pragma unroll
                for (int y = 0; y < TN; ++y) zc[y] = Zs[q * BN + wn * WN + y * 16 + lr];
                const f4 xr = *reinterpret_cast<const f4*>(&Xs[q * BM + wm * WM + x * 16 + 4 * lq]);
#pragma unroll
                for (int y = 0; y < TN; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d = xr[r] - zc[y];
                        r2[y][r] = fmaf(d, d, r2[y][r]);
                    }
            }
            float dvx = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + x * 16 + 4 * lq + r;
                const float yr = (row < g.M) ? f.ycoef * (float)f.y[row] : 0.0f;
#pragma unroll
                for (int y = 0; y < TN; ++y) {
                    const int col = n0 + wn * WN + y * 16 + lr;
                    if (row < g.M && col < g.N) {
                        const float Q = fmaf(yr, wvc[y], acc[x][y][r]);
                        float Kv, hv;
                        bool zero;
                        if (rbf) {
                            Kv = A[(size_t)row * g.lda + col];
                            hv = -Kv;
                            zero = (Kv == f.var);
                        } else {
                            kfun32(f.kind, f.var, r2[y][r], Kv, hv);
                            zero = (r2[y][r] == 0.0f);
                        }
                        dvx = fmaf(Q, Kv, dvx);
                        const float tval = zero ? 0.0f : Q * hv;
                        C[(size_t)row * g.ldc + col] = tval;
                        csum[y] += tval;
                    }
                }
            }
            dv += (double)dvx;
        }
        __syncthreads();
        {
            float* redc = &Bs[0][0];
#pragma unroll
            for (int y = 0; y < TN; ++y) redc[((wm * 4 + lq) * BN) + wn * WN + y * 16 + lr] = csum[y];
            __syncthreads();
            for (int j = tid; j < BN; j += 256) {
                double sum = 0.0;
#pragma unroll
                for (int p = 0; p < 8; ++p) sum += (double)redc[p * BN + j];
                if (n0 + j < g.N) f.cspart[(size_t)ti * g.N + n0 + j] = sum;
            }
        }
        __syncthreads();
        double* red = reinterpret_cast<double*>(&As[0][0]);
        red[tid] = dv;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) f.dvpart[blockIdx.x] = red[0] / (double)f.var;
        return;
    }

    // ---- plain store (C or one split-K slab)
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int col = n0 + wn * WN + y * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + x * 16 + 4 * lq + r;
                if (row < g.M && col < g.N) C[(size_t)row * g.ldc + col] = acc[x][y][r];
            }
        }
}

// C (fp64) = sum_s slab[s] (fp32 slabs added in fp64, fixed order) (+ C if accumulate); mirror:
// computed for i >= j and written to both triangles
__global__ void slab_reduce32_kernel(const float* __restrict__ slabs, int splits, long long slabStride, int M, int N,
                                     int lds, int accumulate, double* __restrict__ C, int ldc, int mirror) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= M || j >= N) return;
    if (mirror && j > i) return;
    double s = 0.0;
    const float* p = slabs + (size_t)i * lds + j;
    for (int k = 0; k < splits; ++k) s += (double)p[(size_t)k * slabStride];
    double* c = C + (size_t)i * ldc + j;
    if (accumulate) s += *c;
    *c = s;
    if (mirror && i != j) C[(size_t)j * ldc + i] = s;
}

// Xs = (float)(X .* invl), xn = |Xs|^2 (norm of the rounded values, so that r2 = 0 for x == z)
__global__ void scale_norm32_kernel(const double* __restrict__ X, int64_t rows, int D, const double* __restrict__ invl,
                                    float* __restrict__ Xs, float* __restrict__ xn) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    double s = 0.0;
    for (int q = 0; q < D; ++q) {
        const float v = (float)(X[i * D + q] * invl[q]);
        Xs[i * D + q] = v;
        s = fma((double)v, (double)v, s);
    }
    xn[i] = (float)s;
}

__global__ void to_f32_kernel(const double* __restrict__ src, size_t n, float* __restrict__ dst) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i];
}

// Xq[n] = (float)[x_n | x_n .* x_n]
__global__ void xq32_kernel(const double* __restrict__ X, int64_t rows, int D, float* __restrict__ Xq) {
    const int W = 2 * D;
    const int64_t total = rows * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / W;
        const int q = (int)(i - n * W);
        const double x = X[n * D + (q < D ? q : q - D)];
        Xq[i] = (float)((q < D) ? x : x * x);
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int BM, int BN>
static void launch32(bool akm, bool bkm, int tag, dim3 grid, hipStream_t s, const G32Args& g) {
    if (tag == 1 && akm && bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true, 1>), grid, dim3(256), 0, s, g);
    else if (tag == 2 && !akm && bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true, 2>), grid, dim3(256), 0, s, g);
    else if (tag == 3 && !akm && !bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false, 3>), grid, dim3(256), 0, s, g);
    else if (akm && bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true, 0>), grid, dim3(256), 0, s, g);
    else if (akm && !bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false, 0>), grid, dim3(256), 0, s, g);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true, 0>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false, 0>), grid, dim3(256), 0, s, g);
}

// C (float, or `splits` float slabs) = op(A) op(B)
static int gemm32_impl(Ctx* c, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B,
                       int ldb, float* C, int ldc, int tag, int tile, int tri, int splits, long long slabStride,
                       const Fuse32K* fk, const Fuse32T* ft) {
    if (M <= 0 || N <= 0) return EDRGP_OK;
    if ((tag == 2 && ft == nullptr) || (tag == 3 && fk == nullptr) || ((tag == 2 || tag == 3) && (splits != 1 || tri))) {
        set_error("gemm32: fused epilogues need an unsplit full-grid product");
        return EDRGP_ERR_INVALID;
    }
    G32Args g;
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.splits = splits; g.slabStride = (splits > 1 || slabStride) ? slabStride : 0;
    g.tri = tri;
    g.prio = (c->gemm_dephase == 3);
    g.vecA = aligned16(A) && (lda % 4 == 0);
    g.vecB = aligned16(B) && (ldb % 4 == 0);
    if (fk) g.fk = *fk;
    if (ft) g.ft = *ft;
    const int tiles_m = (M + tile - 1) / tile;
    g.tiles_n = (N + tile - 1) / tile;
    const long long nt = tri ? (long long)tiles_m * (tiles_m + 1) / 2 : (long long)tiles_m * g.tiles_n;
    if (nt > 0x7fffffffLL) { set_error("gemm32: too many tiles"); return EDRGP_ERR_INVALID; }
    dim3 grid((unsigned)nt, (unsigned)splits, 1);
    const bool akm = transA, bkm = !transB;
    if (tile == 128) launch32<128, 128>(akm, bkm, tag, grid, c->stream, g);
    else launch32<64, 64>(akm, bkm, tag, grid, c->stream, g);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int gemm32(Ctx* c, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
           float* C, int ldc) {
    const int tile = (M >= 128 && N >= 128) ? 128 : 64;
    return gemm32_impl(c, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, 0, tile, 0, 1, 0, nullptr, nullptr);
}

// Phi (fp64, full symmetric) = K^T K (+ Phi), K fp32 rows x M: lower tiles, fp32 slabs over row
// ranges (one fp32 accumulation chain per slab), slabs added in fp64
int syrk32_tn(Ctx* c, const float* K, int64_t rows, int M, int ldk, double* Phi, int ldphi, bool accumulate) {
    if (rows > 0x7fffffffLL) { set_error("syrk32_tn: rows chunk too large"); return EDRGP_ERR_INVALID; }
    const int tile = (M >= 256) ? 128 : 64;
    const int tm = (M + tile - 1) / tile;
    const long long nt = (long long)tm * (tm + 1) / 2;
    const int kt = (int)((rows + BK32 - 1) / BK32);
    int splits = c->syrk_splits;
    if (splits <= 0) {
        // rounds-aware choice (la_gemm.hip), then at most 8192 rows per fp32 accumulation chain
        splits = pick_syrk_splits(c, nt, kt / 2, rows, M, tile, 4);
        const int by_len = (int)((rows + 8191) / 8192);
        if (splits < by_len) splits = by_len;
        const int max_splits = std::max(1, kt / 8);           // >= 256 rows per split
        if (splits > max_splits) splits = max_splits;
    }
    if (splits < 1) splits = 1;
    if (splits > kt) splits = std::max(1, kt);
    const long long slabStride = (long long)M * M;
    TRY(c->slabs.reserve((size_t)splits * slabStride * sizeof(float)));
    float* slabs = static_cast<float*>(c->slabs.p);
    {
        ProfScope ps(c, EDRGP_PROF_SYRK);
        TRY(gemm32_impl(c, true, false, M, M, (int)rows, K, ldk, K, ldk, slabs, M, 1, tile, 1, splits, slabStride,
                        nullptr, nullptr));
    }
    dim3 blk(64, 4), grd((M + 63) / 64, (M + 3) / 4);
    hipLaunchKernelGGL(slab_reduce32_kernel, grd, blk, 0, c->stream, slabs, splits, slabStride, M, M, M,
                       accumulate ? 1 : 0, Phi, ldphi, 1);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// C (fp64, M x N) = A^T B with A rows x M, B rows x N (fp32), split over rows
int gemm32_tn_rows(Ctx* c, int M, int N, int64_t rows, const float* A, int lda, const float* B, int ldb, double* C,
                   int ldc, bool accumulate) {
    if (rows > 0x7fffffffLL) { set_error("gemm32_tn_rows: rows chunk too large"); return EDRGP_ERR_INVALID; }
    const int tile = 64;
    const long long nt = (long long)((M + tile - 1) / tile) * ((N + tile - 1) / tile);
    const int kt = (int)((rows + BK32 - 1) / BK32);
    int splits = (int)std::min<long long>((1024 + nt - 1) / nt, std::max(1, kt / 16));
    splits = std::max(splits, (int)std::min<int64_t>((rows + 8191) / 8192, std::max(1, kt / 8)));
    if (splits < 1) splits = 1;
    const long long slabStride = (long long)M * N;
    TRY(c->slabs.reserve((size_t)splits * slabStride * sizeof(float)));
    float* slabs = static_cast<float*>(c->slabs.p);
    TRY(gemm32_impl(c, true, false, M, N, (int)rows, A, lda, B, ldb, slabs, N, 0, tile, 0, splits, slabStride, nullptr,
                    nullptr));
    dim3 blk(64, 4), grd((N + 63) / 64, (M + 3) / 4);
    hipLaunchKernelGGL(slab_reduce32_kernel, grd, blk, 0, c->stream, slabs, splits, slabStride, M, N, N,
                       accumulate ? 1 : 0, C, ldc, 0);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int scale_norm32(Ctx* c, const double* X, int64_t rows, int D, const double* invl, float* Xs, float* xn) {
    hipLaunchKernelGGL(scale_norm32_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, X, rows, D,
                       invl, Xs, xn);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int to_f32(Ctx* c, const double* src, size_t n, float* dst) {
    const int nb = (int)std::min<size_t>(8192, (n + 255) / 256);
    hipLaunchKernelGGL(to_f32_kernel, dim3(std::max(nb, 1)), dim3(256), 0, c->stream, src, n, dst);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

int build_xq32(Ctx* c, const double* X, int64_t rows, int D, float* Xq) {
    const int64_t total = rows * 2 * D;
    const int nb = (int)std::min<int64_t>(8192, (total + 255) / 256);
    hipLaunchKernelGGL(xq32_kernel, dim3(std::max(nb, 1)), dim3(256), 0, c->stream, X, rows, D, Xq);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// K32 (rows x M, fp32) = var k(|x/l - z/l|) for one single-part kernel, and b = K32^T y (fp64 out)
// when y is given.  Leaves the scaled inputs in c->ks32_x / ks32_z for the gradient epilogue.
int kernel_matrix32(Ctx* c, const KSpec& ks, const double* X, int64_t rows, const double* Z, int M, float* K32,
                    const double* y, double* bvec) {
    if (rows <= 0 || M <= 0) return EDRGP_OK;
    if (ks.nparts != 1 || rows > 0x7fffffffLL) {
        set_error("fp32 path: single-part kernels only");
        return EDRGP_ERR_INVALID;
    }
    const int D = ks.D;
    TRY(c->ks32_x.reserve((size_t)rows * D * sizeof(float)));
    TRY(c->ks32_xn.reserve((size_t)rows * sizeof(float)));
    TRY(c->ks32_z.reserve((size_t)M * D * sizeof(float)));
    TRY(c->ks32_zn.reserve((size_t)M * sizeof(float)));
    float* xs = static_cast<float*>(c->ks32_x.p);
    float* xn = static_cast<float*>(c->ks32_xn.p);
    float* zs = static_cast<float*>(c->ks32_z.p);
    float* zn = static_cast<float*>(c->ks32_zn.p);
    TRY(scale_norm32(c, X, rows, D, ks.invl, xs, xn));
    TRY(scale_norm32(c, Z, M, D, ks.invl, zs, zn));
    const int tile = (c->kbuild_tile == 128 && rows >= 128 && M >= 128) ? 128 : 64;     // see kernel_matrix_mfma
    const int tiles_m = (int)((rows + tile - 1) / tile);
    Fuse32K f;
    f.xn = xn; f.zn = zn; f.y = y; f.bpart = nullptr;
    f.var = (float)ks.var[0]; f.kind = ks.kind[0];
    if (y != nullptr) {
        TRY(c->bpart.reserve((size_t)tiles_m * M * sizeof(double)));
        f.bpart = c->bpart.d();
    }
    TRY(gemm32_impl(c, false, true, (int)rows, M, D, xs, D, zs, D, K32, M, 3, tile, 0, 1, 0, &f, nullptr));
    if (y != nullptr) TRY(col_reduce(c, c->bpart.d(), tiles_m, M, bvec));
    return EDRGP_OK;
}

// T32 = (K32 S2 + ycoef y wv^T) .* h(r) (fp32), dvar_accum += sum (..) .* k / var, cs = column sums of T32
int grad_contract32(Ctx* c, const KSpec& ks, const float* K32, int64_t rows, int M, const float* S2f, const double* y,
                    const double* wv, double ycoef, float* T32, double* dvar_accum, double* cs) {
    const int D = ks.D;
    if (ks.nparts != 1 || D > 64) {
        set_error("fp32 gradient path: single-part kernel, D <= 64");
        return EDRGP_ERR_INVALID;
    }
    const int tiles_m = (int)((rows + 127) / 128), tiles_n = (M + 127) / 128;
    const int ntiles = tiles_m * tiles_n;
    TRY(c->red.reserve((size_t)ntiles * sizeof(double)));
    TRY(c->bpart.reserve((size_t)tiles_m * M * sizeof(double)));
    Fuse32T f;
    f.Xs = static_cast<const float*>(c->ks32_x.p);
    f.Zs = static_cast<const float*>(c->ks32_z.p);
    f.y = y; f.wv = wv; f.dvpart = c->red.d(); f.cspart = c->bpart.d();
    f.ycoef = (float)ycoef; f.var = (float)ks.var[0]; f.D = D; f.kind = ks.kind[0];
    TRY(gemm32_impl(c, false, false, (int)rows, M, M, K32, M, S2f, M, T32, M, 2, 128, 0, 1, 0, nullptr, &f));
    TRY(sum_into(c, c->red.d(), ntiles, dvar_accum));
    TRY(col_reduce(c, c->bpart.d(), tiles_m, M, cs));
    return EDRGP_OK;
}

}  // namespace edrgp
