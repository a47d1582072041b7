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
#include "kfun.h"
#include "gemm_tile.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace edrgp {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

constexpr int BK = GemmScalar<double>::BK;        // 16: a k-tile is 128 bytes deep
constexpr int LPAD = GEMM_LPAD;
#ifndef EDRGP_UNMERGED_MMAJOR_READS
#define EDRGP_UNMERGED_MMAJOR_READS 1     // m-major fp64 images through volatile LDS pointers (gemm_tile.h: frag_read)
#endif
#ifndef EDRGP_FRAG_PIPE
#define EDRGP_FRAG_PIPE 3
#endif
#ifndef EDRGP_PIPE_M2
#define EDRGP_PIPE_M2 3         // gradient contraction: MFMAs between two fragment reads (6 read instructions per k-step;
                                // Cm bound+gradients with 1 / 2 / 3 / 4: 23.15 / 23.15 / 22.86 / 22.92 ms)
#endif
#ifndef EDRGP_PIPE_M
#define EDRGP_PIPE_M 4          // Phi: MFMAs between two fragment reads of the next k-step (1: 6.39, 2: 6.32, 3: 6.38, 4: 6.29 ms)
#endif
template <int BMN, bool KM> using TileLoader = TileLoaderT<double, BMN, KM>;      // gemm_tile.h

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
    int pair;               // TAG 4 with K_LE_COL: a workgroup owns column tiles (T-1-p, p): constant k-range per workgroup
    int vecA, vecB;
    unsigned long long* stamps;   // probe build (EDRGP_TILE_STAMPS) only
    FuseT fuse;                   // TAG 2 only, used when fuse_on
    FuseK fusek;                  // TAG 3 only
    int fuse_on;
    int dephase;        // 0 none; 1 s_setprio on odd wave slots; 2 half-tile start delay on odd wave slots
    int xmap;           // 1: split-major XCD map (see the kernel)
    int interleave;     // 1: tile = workgroup index (no XCD-contiguous remap): zero-skipping products
    int persist;        // TAG 2: > 0 = tile count of a persistent launch (grid = 2 workgroups per CU)
    int dbg_alias;      // TIMING PROBE ONLY (wrong results): bit0 A tiles alias 4 row tiles, bit1 B tiles alias column tile 0
    int diag_opt;       // 1: diagonal tiles of the Phi contraction compute their lower blocks only
    int kmode;          // triangular operands: bit0 k <= row tile end, bit1 k >= row tile start,
                        //                      bit2 k <= col tile end, bit3 k >= col tile start
};

// Diagonal tiles of the Phi contraction (128x128; the A and B panels are the same 128 columns of K_nm):
// only the 36 lower 16x16 blocks are needed.  Wave w takes block-rows w and 7-w (w+1 and 8-w blocks = 9
// each: perfectly balanced), so a diagonal tile issues 9 MFMAs per wave and k-step instead of 16, stages ONE
// panel instead of two, and never computes the upper halves (the slab reduction reads the lower triangle
// only).  Same pipeline as the main loop: register prefetch of tile t+1, mid-phase LDS staging, one barrier
// per k-tile, staging at raised priority.
template <int W, typename LAT>
__device__ __forceinline__ void diag_tile_run(const GemmArgs& g, const double* __restrict__ A, double* __restrict__ C,
                                              double (*As)[LAT::SIZE], int tid, int lr, int lq, int m0, int kt0, int kt1) {
    // k-major panel, staged exactly like TileLoader<128, true> (kept as plain locals: the struct form ended up in
    // scratch memory here)
    static_assert(LAT::NV == 4, "diag_tile_run is written for the 128-wide k-major panel");
    const int kr = tid >> 6, mc = (tid & 63) * 2;                 // thread -> k-row kr + 4 v, columns mc, mc + 1
    const unsigned o0 = (unsigned)(((kr) * g.lda + mc) * 8), o1 = (unsigned)(((kr + 4) * g.lda + mc) * 8);
    const unsigned o2 = (unsigned)(((kr + 8) * g.lda + mc) * 8), o3 = (unsigned)(((kr + 12) * g.lda + mc) * 8);
    const int l0 = LAT::idx(kr, mc), l1 = LAT::idx(kr + 4, mc), l2 = LAT::idx(kr + 8, mc), l3 = LAT::idx(kr + 12, mc);
    const char* base = reinterpret_cast<const char*>(A + LAT::tile_origin(g.lda, m0, kt0));
    const long long step = LAT::tile_step(g.lda);
    double2 r0 = {0.0, 0.0}, r1 = r0, r2 = r0, r3 = r0;
    if (kt0 < kt1) {                     // an empty row range (more splits than k-tiles) stores zeros
        r0 = *reinterpret_cast<const double2*>(base + o0);
        r1 = *reinterpret_cast<const double2*>(base + o1);
        r2 = *reinterpret_cast<const double2*>(base + o2);
        r3 = *reinterpret_cast<const double2*>(base + o3);
    }
    *reinterpret_cast<double2*>(&As[0][l0]) = r0;
    *reinterpret_cast<double2*>(&As[0][l1]) = r1;
    *reinterpret_cast<double2*>(&As[0][l2]) = r2;
    *reinterpret_cast<double2*>(&As[0][l3]) = r3;
    __syncthreads();
    d4 acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) acc[f] = d4{0.0, 0.0, 0.0, 0.0};
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        const bool more = (kt + 1 < kt1);
        if (more) {
            base += step;
            r0 = *reinterpret_cast<const double2*>(base + o0);
            r1 = *reinterpret_cast<const double2*>(base + o1);
            r2 = *reinterpret_cast<const double2*>(base + o2);
            r3 = *reinterpret_cast<const double2*>(base + o3);
        }
        if (g.dephase == 3) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int krow = kk * 4 + lq;
            const double a0 = As[buf][LAT::idx(krow, W * 16 + lr)];
            const double a1 = As[buf][LAT::idx(krow, (7 - W) * 16 + lr)];
            double b[8 - W];
#pragma unroll
            for (int j = 0; j < 8 - W; ++j) b[j] = As[buf][LAT::idx(krow, j * 16 + lr)];
#pragma unroll
            for (int j = 0; j <= W; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8 - W; ++j)
                acc[W + 1 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[j], acc[W + 1 + j], 0, 0, 0);
            if (kk == BK / 4 - 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    *reinterpret_cast<double2*>(&As[buf ^ 1][l0]) = r0;
                    *reinterpret_cast<double2*>(&As[buf ^ 1][l1]) = r1;
                    *reinterpret_cast<double2*>(&As[buf ^ 1][l2]) = r2;
                    *reinterpret_cast<double2*>(&As[buf ^ 1][l3]) = r3;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (g.dephase == 3) __builtin_amdgcn_s_setprio(2);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j <= W; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(size_t)(m0 + W * 16 + lq + 4 * r) * g.ldc + m0 + j * 16 + lr] = acc[j][r];
#pragma unroll
    for (int j = 0; j < 8 - W; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            C[(size_t)(m0 + (7 - W) * 16 + lq + 4 * r) * g.ldc + m0 + j * 16 + lr] = acc[W + 1 + j][r];
}

// Two diagonal tiles (column blocks d0 and d1) in one workgroup: 18 MFMAs per wave and k-step keep the MFMA
// phase of a k-tile as long as a regular tile's 16, so the one-tile-ahead prefetch still covers the load latency
// (a lone diagonal tile with 9 was load-latency-bound and gained nothing).  Panel d0 is staged in As, d1 in Bs.
template <int W, typename LAT>
__device__ __forceinline__ void diag_pair_run(const GemmArgs& g, const double* __restrict__ A, double* __restrict__ C,
                                              double (*As)[LAT::SIZE], double (*Bs)[LAT::SIZE], int tid, int lr, int lq,
                                              int c0, int c1, int kt0, int kt1) {
    static_assert(LAT::NV == 4, "diag_pair_run is written for the 128-wide k-major panel");
    const int kr = tid >> 6, mc = (tid & 63) * 2;
    const unsigned o0 = (unsigned)(((kr) * g.lda + mc) * 8), o1 = (unsigned)(((kr + 4) * g.lda + mc) * 8);
    const unsigned o2 = (unsigned)(((kr + 8) * g.lda + mc) * 8), o3 = (unsigned)(((kr + 12) * g.lda + mc) * 8);
    const int l0 = LAT::idx(kr, mc), l1 = LAT::idx(kr + 4, mc), l2 = LAT::idx(kr + 8, mc), l3 = LAT::idx(kr + 12, mc);
    const char* ba = reinterpret_cast<const char*>(A + LAT::tile_origin(g.lda, c0, kt0));
    const char* bb = reinterpret_cast<const char*>(A + LAT::tile_origin(g.lda, c1, kt0));
    const long long step = LAT::tile_step(g.lda);
    double2 r0 = {0.0, 0.0}, r1 = r0, r2 = r0, r3 = r0, q0 = r0, q1 = r0, q2 = r0, q3 = r0;
    if (kt0 < kt1) {                     // an empty row range (more splits than k-tiles) stores zeros
        r0 = *reinterpret_cast<const double2*>(ba + o0);
        r1 = *reinterpret_cast<const double2*>(ba + o1);
        r2 = *reinterpret_cast<const double2*>(ba + o2);
        r3 = *reinterpret_cast<const double2*>(ba + o3);
        q0 = *reinterpret_cast<const double2*>(bb + o0);
        q1 = *reinterpret_cast<const double2*>(bb + o1);
        q2 = *reinterpret_cast<const double2*>(bb + o2);
        q3 = *reinterpret_cast<const double2*>(bb + o3);
    }
    *reinterpret_cast<double2*>(&As[0][l0]) = r0;
    *reinterpret_cast<double2*>(&As[0][l1]) = r1;
    *reinterpret_cast<double2*>(&As[0][l2]) = r2;
    *reinterpret_cast<double2*>(&As[0][l3]) = r3;
    *reinterpret_cast<double2*>(&Bs[0][l0]) = q0;
    *reinterpret_cast<double2*>(&Bs[0][l1]) = q1;
    *reinterpret_cast<double2*>(&Bs[0][l2]) = q2;
    *reinterpret_cast<double2*>(&Bs[0][l3]) = q3;
    __syncthreads();
    d4 acc[18];
#pragma unroll
    for (int f = 0; f < 18; ++f) acc[f] = d4{0.0, 0.0, 0.0, 0.0};
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        const bool more = (kt + 1 < kt1);
        if (more) {
            ba += step;
            bb += step;
            r0 = *reinterpret_cast<const double2*>(ba + o0);
            r1 = *reinterpret_cast<const double2*>(ba + o1);
            r2 = *reinterpret_cast<const double2*>(ba + o2);
            r3 = *reinterpret_cast<const double2*>(ba + o3);
            q0 = *reinterpret_cast<const double2*>(bb + o0);
            q1 = *reinterpret_cast<const double2*>(bb + o1);
            q2 = *reinterpret_cast<const double2*>(bb + o2);
            q3 = *reinterpret_cast<const double2*>(bb + o3);
        }
        if (g.dephase == 3) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int krow = kk * 4 + lq;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double* S = h ? Bs[buf] : As[buf];
                const double a0 = S[LAT::idx(krow, W * 16 + lr)];
                const double a1 = S[LAT::idx(krow, (7 - W) * 16 + lr)];
                double b[8 - W];
#pragma unroll
                for (int j = 0; j < 8 - W; ++j) b[j] = S[LAT::idx(krow, j * 16 + lr)];
#pragma unroll
                for (int j = 0; j <= W; ++j)
                    acc[9 * h + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[j], acc[9 * h + j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8 - W; ++j)
                    acc[9 * h + W + 1 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[j], acc[9 * h + W + 1 + j], 0, 0, 0);
            }
            if (kk == BK / 4 - 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    *reinterpret_cast<double2*>(&As[buf ^ 1][l0]) = r0;
                    *reinterpret_cast<double2*>(&As[buf ^ 1][l1]) = r1;
                    *reinterpret_cast<double2*>(&As[buf ^ 1][l2]) = r2;
                    *reinterpret_cast<double2*>(&As[buf ^ 1][l3]) = r3;
                    *reinterpret_cast<double2*>(&Bs[buf ^ 1][l0]) = q0;
                    *reinterpret_cast<double2*>(&Bs[buf ^ 1][l1]) = q1;
                    *reinterpret_cast<double2*>(&Bs[buf ^ 1][l2]) = q2;
                    *reinterpret_cast<double2*>(&Bs[buf ^ 1][l3]) = q3;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (g.dephase == 3) __builtin_amdgcn_s_setprio(2);
        __syncthreads();
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = h ? c1 : c0;
#pragma unroll
        for (int j = 0; j <= W; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)(c + W * 16 + lq + 4 * r) * g.ldc + c + j * 16 + lr] = acc[9 * h + j][r];
#pragma unroll
        for (int j = 0; j < 8 - W; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(size_t)(c + (7 - W) * 16 + lq + 4 * r) * g.ldc + c + j * 16 + lr] = acc[9 * h + W + 1 + j][r];
    }
}

// Workgroup reduction at the end of the fused gradient epilogue: column sums of T (8 partials per column:
// 2 row-waves x 4 lane rows) and the d/dvariance partial, with TWO barriers -- a barrier beside the partner
// workgroup's MFMA stream costs ~2 us, and the first version spent eleven of them (0.5 ms per launch at Cm).
// Everything goes through LDS once; wave 2 folds the 256 d/dvariance partials (4 per lane, then a fixed
// shuffle tree), waves 0-1 the columns.
template <int BN, int WN, int TN>
__device__ __forceinline__ void fused_epilogue_reduce(const FuseT& f, const double (&csum)[TN], double dv,
                                                       double* __restrict__ lds, int tid, int wm, int wn, int lr, int lq,
                                                       int n0, int N, int ti, size_t dv_slot) {
    __syncthreads();                               // the operand tiles in LDS are dead from here
    double* redc = lds;                            // [8][BN] column partials
    double* redv = lds + 8 * BN;                   // [256] d/dvariance partials
#pragma unroll
    for (int y = 0; y < TN; ++y) redc[((wm * 4 + lq) * BN) + wn * WN + y * 16 + lr] = csum[y];
    redv[tid] = dv;
    __syncthreads();
    if (tid < BN) {
        if (f.cspart != nullptr) {
            double sum = 0.0;
#pragma unroll
            for (int p = 0; p < 8; ++p) sum += redc[p * BN + tid];
            if (n0 + tid < N) f.cspart[(size_t)ti * N + n0 + tid] = sum;
        }
    } else if (tid < BN + 64) {
        const int l = tid - BN;
        double v = (redv[l] + redv[l + 64]) + (redv[l + 128] + redv[l + 192]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (l == 0) f.dvpart[dv_slot] = v / f.var;
    }
}

// TAG only changes the kernel symbol, so the Phi syrk (1) and the gradient
// contraction W = K S (2) show up under their own names in rocprofv3 traces.
template <int BM, int BN, bool AKM, bool BKM, int TAG>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr bool EARLY_STORE = (BM == 128);      // long MFMA phase: stage the next tile mid-phase
    // m-major operand images (row stride 18 doubles) are read with unmerged ds_read_b64 (frag_read): 128-wide tiles
    constexpr bool VOL_FRAG = (EDRGP_UNMERGED_MMAJOR_READS != 0) && BM == 128 && BN == 128;
    // gradient contraction: 2 = RBF epilogue (h = -k, k from the resident K_nm), 7 = k and h(r) both resident (Matern),
    // 5 = distance-recomputing epilogue
    constexpr bool T25 = (TAG == 2 || TAG == 5 || TAG == 7);
    constexpr int PIPE_M = (BM != 128) ? 1 : (T25 ? EDRGP_PIPE_M2 : EDRGP_PIPE_M);   // the contraction has 6 read instructions per k-step, Phi 4
    constexpr bool FRAG_PIPE = (EDRGP_FRAG_PIPE != 0) && ((BM == 128 && BN == 128 && (TAG == 1 || (T25 && EDRGP_FRAG_PIPE >= 2))) ||
                                                           (EDRGP_FRAG_PIPE >= 3 && BM == 64 && BN == 64));
    // k-loop shape of this instance (gemm_tile.h); the next tile goes to LDS after k-step BK/4 - 2 of 4
    // (a k-loop per loader kind for Phi only: gemm_tile.h, gemm_kloop)
    typedef GemmLoopCfg<FRAG_PIPE, PIPE_M, EARLY_STORE, BK / 4 - 2, VOL_FRAG && !AKM, VOL_FRAG && !BKM, TAG == 1> LoopCfg;
    typedef TileLoader<BM, AKM> LA;
    typedef TileLoader<BN, BKM> LB;
    __shared__ __attribute__((aligned(16))) double As[2][LA::SIZE];
    __shared__ __attribute__((aligned(16))) double Bs[2][LB::SIZE];

    const int tid0 = threadIdx.x;
#ifdef EDRGP_TILE_STAMPS
    // probe build (tools/tile_stamps.py): wall-clock stamps (100 MHz) of this workgroup's phases, contraction only
    unsigned long long ts_entry = 0, ts_loop = 0, ts_loop_end = 0;
    if (T25 || TAG == 1) ts_entry = __builtin_amdgcn_s_memrealtime();
#endif

    // Two workgroups share a CU (two waves per SIMD).  Launched together and doing
    // identical work they run in lockstep, so their barrier / staging phases coincide
    // and the matrix pipe idles (~20 % measured).  De-phase them by the hardware wave
    // slot (HW_ID[3:0]), which differs between the two waves of a SIMD: speed only.
    if (g.dephase) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;
        if (slot) {
            if (g.dephase == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_sleep(48);
        }
    }

    // ---- tile id with an XCD-aware (bijective) remap: blocks b, b+8, ... share
    // an XCD (L2), give them consecutive tiles so they share operand panels.
    const int nt = gridDim.x;
    int t, split = blockIdx.y;
    if (g.xmap) {
        // split-major XCD map (Phi): workgroups are dispatched x-fastest and dealt to the 8 XCDs round-robin;
        // order them (split, tile) and give every XCD a contiguous eighth, so the workgroups running together
        // on an XCD are the tiles of ONE row range: each K_nm row block is fetched once into that XCD's L2 and
        // shared by all column panels, and each XCD reads ~1/8 of K_nm instead of all of it.
        const int total = nt * (int)gridDim.y;
        const int lin = (int)blockIdx.y * nt + (int)blockIdx.x;
        const int q = total >> 3, r = total & 7, x = lin & 7, in = lin >> 3;
        const int w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + in;
        split = w / nt;
        t = w - split * nt;
    } else if (TAG == 0 && g.interleave >= 2) {
        // one triangular operand, full output: tiles ranked by k-range (longest first) and dealt to the CUs in a
        // snake -- pass 0 over the CUs ascending, pass 1 descending, ... -- so the workgroups sharing a CU have the same
        // total.  Assumes the dispatcher's order (workgroup b -> XCD b % 8, CU (b / 8) % 32 of it, slot b / 256); any
        // other order only loses the balance, the map stays a bijection.
        const int b = blockIdx.x, x = b & 7, l = b >> 3, cu = l & 31, slot = l >> 5;
        const int G = cu * 8 + x;
        t = slot * 256 + ((slot & 1) ? 255 - G : G);
    } else if (TAG == 0 && g.interleave) {
        // zero-skipping products: a tile's k-range grows (or shrinks) with its index, so a contiguous eighth per XCD
        // hands one XCD twice the mean work and another almost none; dealt round-robin every XCD (and, 32 tiles apart,
        // every CU) gets the same mix
        t = blockIdx.x;
    } else {
        const int bid = blockIdx.x;
        const int q = nt >> 3, r = nt & 7, x = bid & 7, in = bid >> 3;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + in;
    }
    int ti, tj;
    int pair_d0 = -1;                  // Phi with diag_opt: >= 0 -> this workgroup owns diagonal tiles d0 (and d0 + 1)
    if (g.tri && g.diag_opt) {
        // strictly lower tiles first, then the diagonal tiles two per workgroup (measured: leading with the
        // pair workgroups instead is 2-3 % slower)
        const int n_strict = g.tiles_m * (g.tiles_m - 1) / 2;
        if (t < n_strict) {
            ti = (int)((sqrt(8.0 * (double)t + 1.0) + 1.0) * 0.5);
            while ((long long)ti * (ti + 1) / 2 <= t) ++ti;
            while ((long long)ti * (ti - 1) / 2 > t) --ti;
            tj = t - (int)((long long)ti * (ti - 1) / 2);
        } else {
            pair_d0 = 2 * (t - n_strict);
            ti = tj = pair_d0;
        }
    } else if (g.tri) {
        ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while ((long long)ti * (ti + 1) / 2 > t) --ti;
        tj = t - (int)((long long)ti * (ti + 1) / 2);
    } else if (TAG == 4 && g.pair) {
        const int half = (g.tiles_n + 1) / 2;
        ti = t / half;
        tj = g.tiles_n - 1 - (t - ti * half);          // the long column tile first, its short partner second
    } else if (TAG == 0 && g.interleave >= 2) {
        const int km = g.interleave >> 1;              // the single k-range restriction this product has
        if (km == 1)      { ti = g.tiles_m - 1 - t / g.tiles_n; tj = t % g.tiles_n; }      // k <= row end: last rows longest
        else if (km == 2) { ti = t / g.tiles_n; tj = t % g.tiles_n; }                        // k >= row start: first rows
        else if (km == 4) { tj = g.tiles_n - 1 - t / g.tiles_m; ti = t % g.tiles_m; }      // k <= col end: last columns
        else              { tj = t / g.tiles_m; ti = t % g.tiles_m; }                        // k >= col start: first columns
    } else {
        ti = t / g.tiles_n;
        tj = t - ti * g.tiles_n;
    }
    // Zero-skipping over a lower-triangular right operand gives column tile j a k-range of (j + 1) tiles: paired
    // (T-1-p, p), every workgroup runs T + 1 of them and half as many workgroups pay prologue and drain.
    const int tj_second = g.tiles_n - 1 - tj;
    int npass = (TAG == 4 && g.pair && tj_second != tj) ? 2 : 1;
    // Persistent contraction (TAG 2, g.persist = number of tiles): the grid is two workgroups per CU and each walks a
    // fixed list of tiles.  A fresh workgroup reaches a freed slot ~19 us after its predecessor left (measured,
    // tools/tile_stamps.py: 49 turnovers per CU = 0.9 ms with half the matrix pipe's feed), a resident one starts its
    // next tile at once.  Workgroups b, b + 8, ... share an XCD and take that XCD's contiguous eighth of the tiles 64
    // at a time, so the workgroups running together still share their K_nm row panels in the L2.
    // Persistent Phi (TAG 1, g.persist = tiles x splits): the same for the Gram launch -- its 1536 workgroups of 2.2 ms
    // in three rounds drifted apart by their turnovers and their unequal lengths (a diagonal-pair workgroup issues 18
    // MFMAs per k-step against 16): mean residency 484 of 512 (tools/phi_stamps.py).  Units in (split, tile) order, an
    // XCD's contiguous eighth 64 at a time: at most one diagonal-pair unit per workgroup.
    int p_start = 0, p_len = 0, p_per = 1, p_in = 0;
#ifdef EDRGP_NO_PERSIST_T1
    constexpr bool T1P = false;
#else
    constexpr bool T1P = (TAG == 1);
#endif
    if ((T25 || TAG == 4 || T1P) && g.persist) {
        const int total = g.persist;
        const int q = total >> 3, r = total & 7, x = (int)blockIdx.x & 7;
        p_in = (int)blockIdx.x >> 3;
        p_per = (int)gridDim.x >> 3;
        p_start = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q);
        p_len = q + (x < r ? 1 : 0);
        npass = (p_len - p_in + p_per - 1) / p_per;
        if (TAG == 4) npass *= 2;                      // a unit of the stable form's product = a pair of column tiles
    }
#pragma clang loop unroll(disable)
    for (int pass = 0; pass < npass; ++pass) {
    if (T1P && g.persist) {
        const int w = p_start + pass * p_per + p_in;
        const int per_split = g.persist / g.splits;
        split = w / per_split;
        t = w - split * per_split;
        const int n_strict = g.tiles_m * (g.tiles_m - 1) / 2;          // persistent Phi launches have diag_opt
        pair_d0 = -1;
        if (t < n_strict) {
            ti = (int)((sqrt(8.0 * (double)t + 1.0) + 1.0) * 0.5);
            while ((long long)ti * (ti + 1) / 2 <= t) ++ti;
            while ((long long)ti * (ti - 1) / 2 > t) --ti;
            tj = t - (int)((long long)ti * (ti - 1) / 2);
        } else {
            pair_d0 = 2 * (t - n_strict);
            ti = tj = pair_d0;
        }
        if (pass) __syncthreads();                     // the previous unit is done with the LDS
    } else if (T25 && g.persist) {
        t = p_start + pass * p_per + p_in;
        ti = t / g.tiles_n;
        tj = t - ti * g.tiles_n;
        if (pass) __syncthreads();                     // the previous tile's epilogue is done with the LDS
#ifdef EDRGP_TILE_STAMPS
        ts_entry = __builtin_amdgcn_s_memrealtime();
#endif
    } else if (TAG == 4 && g.persist) {
        // persistent stable-form product: units (row tile, column-tile pair) in an XCD's contiguous eighth, two
        // passes per unit (the long column tile, then its short partner); 140 us units paid a 19 us turnover each
        t = p_start + (pass >> 1) * p_per + p_in;
        const int half = (g.tiles_n + 1) / 2;
        ti = t / half;
        const int tj_long = g.tiles_n - 1 - (t - ti * half);
        const int tj_short = g.tiles_n - 1 - tj_long;
        if ((pass & 1) && tj_short == tj_long) continue;               // the middle column of an odd count has no partner
        tj = (pass & 1) ? tj_short : tj_long;
        if (pass) __syncthreads();                     // the previous tile's epilogue is done with the LDS
    } else if (pass == 1) {
        tj = tj_second;
        __syncthreads();                               // the first tile's epilogue is done with the LDS
    }

    // The thread index is re-derived per tile behind an opaque barrier (TAG 2): otherwise every per-thread address of
    // the loaders and the epilogue is hoisted out of the tile loop of a persistent launch and kept alive across it
    // (249 -> 256 VGPRs + 164 B of scratch per lane).
    int tid_l = tid0;
#ifdef EDRGP_NO_LAUNDER_T1
    if (T25 || TAG == 4) asm volatile("" : "+v"(tid_l));
#else
    if (TAG == 1 || T25 || TAG == 4) asm volatile("" : "+v"(tid_l));
#endif
    const int tid = tid_l;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;

    const int z = blockIdx.z;
    int Meff = g.M, Keff = g.K;
    if (g.limM_stride) Meff = min(Meff, g.limM_base - z * g.limM_stride);
    if (g.limK_stride) Keff = min(Keff, g.limK_base - z * g.limK_stride);
    const int m0 = ti * BM, n0 = tj * BN;
    if (Meff <= 0 || m0 >= Meff || n0 >= g.N) continue;

    const double* A = g.A + (long long)z * g.strideA;
    const double* B = g.B + (long long)z * g.strideB;
    double* C = g.C + (long long)z * g.strideC + (long long)split * g.slabStride;

    // ---- split-K range (in units of BK)
    const int kt_total = (Keff + BK - 1) / BK;
    const int kt_per = (kt_total + g.splits - 1) / g.splits;
    int kt0 = split * kt_per;
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

    if constexpr (TAG == 1 && BM == 128 && BN == 128 && AKM && BKM) {
        // Phi: the diagonal tiles run the 36-block loops, two tiles per workgroup
        // (the host enables diag_opt only for full tiles, full k-tiles and 16-byte aligned rows)
        if (pair_d0 >= 0) {
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);
            if (pair_d0 + 1 < g.tiles_m) {
                const int c0 = pair_d0 * BM, c1 = c0 + BM;
                if (wave_u == 0) diag_pair_run<0, LA>(g, A, C, As, Bs, tid, lr, lq, c0, c1, kt0, kt1);
                else if (wave_u == 1) diag_pair_run<1, LA>(g, A, C, As, Bs, tid, lr, lq, c0, c1, kt0, kt1);
                else if (wave_u == 2) diag_pair_run<2, LA>(g, A, C, As, Bs, tid, lr, lq, c0, c1, kt0, kt1);
                else diag_pair_run<3, LA>(g, A, C, As, Bs, tid, lr, lq, c0, c1, kt0, kt1);
            } else {                                       // odd tile count: the last diagonal tile alone
                if (wave_u == 0) diag_tile_run<0, LA>(g, A, C, As, tid, lr, lq, m0, kt0, kt1);
                else if (wave_u == 1) diag_tile_run<1, LA>(g, A, C, As, tid, lr, lq, m0, kt0, kt1);
                else if (wave_u == 2) diag_tile_run<2, LA>(g, A, C, As, tid, lr, lq, m0, kt0, kt1);
                else diag_tile_run<3, LA>(g, A, C, As, tid, lr, lq, m0, kt0, kt1);
            }
            continue;
        }
    }

    d4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    if (kt0 < kt1) {
        // the k-loop (gemm_tile.h: shared with the fp32 kernel below); timing probes may alias the operand tiles
        const int m0l = (TAG == 2 && (g.dbg_alias & 1)) ? m0 % (4 * BM) : m0, n0l = (TAG == 2 && (g.dbg_alias & 2)) ? 0 : n0;
#ifdef EDRGP_TILE_STAMPS
        if (T25) ts_loop = __builtin_amdgcn_s_memrealtime();
#endif
        gemm_mainloop<double, BM, BN, AKM, BKM, LoopCfg>(acc, A, B, g.lda, g.ldb, m0, n0, m0l, n0l, Meff, g.N, Keff, kt0, kt1,
                                                         g.vecA != 0, g.vecB != 0, As, Bs, tid, wm, wn, lr, lq, g.dephase == 3);
#ifdef EDRGP_TILE_STAMPS
        if (T25) ts_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
    }

    // ---- fused kernel-matrix epilogue (TAG 3): acc = <x/l, z/l>  ->  K = var k(r2)
    if (TAG == 3) {
        const FuseK& f = g.fusek;
        double zn[TN], bsum[TN];
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int col = n0 + wn * WN + y * 16 + lr;
            zn[y] = (col < g.N) ? f.zn[col] : 0.0;
            bsum[y] = 0.0;
        }
        // interior tiles that do not touch the diagonal of a square build: wave-uniform row base + one per-lane byte
        // offset, no guards (the build is bound by its epilogue's VALU work, not by the two k-tiles of its product)
        const bool interior = (m0 + BM <= Meff) && (n0 + BN <= g.N) && !(f.same && m0 < n0 + BN && n0 < m0 + BM) && f.dbg == 0 &&
                              ((long long)BM * g.ldc < (1LL << 28)) && (f.H == nullptr || (long long)BM * f.ldh < (1LL << 28));
        if (interior) {
            const int wu = __builtin_amdgcn_readfirstlane(wave);
            const int wmu = wu >> 1, wnu = wu & 1;
            char* Cb = reinterpret_cast<char*>(C + (size_t)(m0 + wmu * WM) * g.ldc + n0 + wnu * WN);
            char* Hb = (f.H != nullptr) ? reinterpret_cast<char*>(f.H + (size_t)(m0 + wmu * WM) * f.ldh + n0 + wnu * WN) : nullptr;
            const unsigned vc = (unsigned)((lq * g.ldc + lr) * 8), vh = (unsigned)((lq * f.ldh + lr) * 8);
            const double* xnb = f.xn + m0 + wmu * WM;
            const double* yb = (f.y != nullptr) ? f.y + m0 + wmu * WM : nullptr;
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    char* cr = Cb + (size_t)(x * 16 + 4 * r) * g.ldc * 8;
                    const double xn = xnb[x * 16 + 4 * r + lq];
                    const double yr = (yb != nullptr) ? yb[x * 16 + 4 * r + lq] : 0.0;
#pragma unroll
                    for (int y = 0; y < TN; ++y) {
                        const double r2 = fmax(fma(-2.0, acc[x][y][r], xn + zn[y]), 0.0);
                        double Kv, hv;
                        edrgp_kfun(f.kind, f.var, r2, Kv, hv);
                        *reinterpret_cast<double*>(cr + vc + y * 128) = Kv;
                        if (Hb != nullptr)
                            *reinterpret_cast<double*>(Hb + (size_t)(x * 16 + 4 * r) * f.ldh * 8 + vh + y * 128) = (r2 == 0.0) ? 0.0 : hv;
                        bsum[y] = fma(Kv, yr, bsum[y]);
                    }
                }
        } else
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + x * 16 + lq + 4 * r;
                const bool rok = row < Meff;
                const double xn = rok ? f.xn[row] : 0.0;
                const double yr = (rok && f.y != nullptr) ? f.y[row] : 0.0;
#pragma unroll
                for (int y = 0; y < TN; ++y) {
                    const int col = n0 + wn * WN + y * 16 + lr;
                    double r2 = fmax(fma(-2.0, acc[x][y][r], xn + zn[y]), 0.0);
                    const bool diag = f.same && (row == col);
                    if (diag) r2 = 0.0;
                    double Kv, hv;
                    if (f.dbg & 1) { Kv = r2; hv = 0.0; } else edrgp_kfun(f.kind, f.var, r2, Kv, hv);
                    if (diag) Kv += f.jitter;
                    if (rok && col < g.N) {
                        if (!(f.dbg & 2) || Kv == 12345.678) C[(size_t)row * g.ldc + col] = Kv;
                        // h(r) = (dk/dr) / r beside k for the gradient contraction of non-RBF kernels (0 where r = 0:
                        // T is 0 there upstream)
                        if (f.H != nullptr) f.H[(size_t)row * f.ldh + col] = (r2 == 0.0) ? 0.0 : hv;
                        bsum[y] = fma(Kv, yr, bsum[y]);
                    }
                }
            }
        if (f.y != nullptr) {
            // column sums over the tile's rows: 8 partials per column (2 row-waves x 4 lane rows)
            __syncthreads();
            double* red = &As[0][0];
#pragma unroll
            for (int y = 0; y < TN; ++y) red[((wm * 4 + lq) * BN) + wn * WN + y * 16 + lr] = bsum[y];
            __syncthreads();
            for (int j = tid; j < BN; j += 256) {
                double sum = 0.0;
#pragma unroll
                for (int p = 0; p < 8; ++p) sum += red[p * BN + j];
                if (n0 + j < g.N) f.bpart[(size_t)ti * g.N + n0 + j] = sum;
            }
        }
        continue;
    }

    // ---- fused gradient epilogue (TAG 2): T = (acc + ycoef y wv^T) .* h(r), sum Q .* k / var
    if (T25 && g.fuse_on) {
        // scaled inputs of this tile's rows / columns, staged in the (now idle) operand LDS:
        // Xs[q][BM] = X[m0 + i][q] * invl[q], Zs[q][BN] = Z[n0 + j][q] * invl[q]
        constexpr int DC = 32;                               // feature chunk
        static_assert(2 * LA::SIZE >= DC * BM && 2 * LB::SIZE >= DC * BN, "LDS too small for the fused epilogue");
        double* Xs = &As[0][0];
        double* Zs = &Bs[0][0];
        const FuseT& f = g.fuse;
        if (TAG == 2 || TAG == 7) {
            // (the host launches TAG 2 / 7 only with f.K (and f.H) set: one epilogue per kernel -- with both in one kernel the
            // contraction sat at 254-256 VGPRs and 221 spilled SGPRs, alone the RBF one takes 234 / 56)
            // RBF: h(r) = (dk/dr)/r = -k(r), and k is the resident K_nm tile this workgroup's rows and
            // columns address -- no distances, no exp.  (k == var  <=>  r == 0, where T is 0 upstream.)
            double wvc[TN], csum[TN];
#pragma unroll
            for (int y = 0; y < TN; ++y) {
                const int col = n0 + wn * WN + y * 16 + lr;
                wvc[y] = (col < g.N) ? f.wv[col] : 0.0;
                csum[y] = 0.0;
            }
            double dv = 0.0;
            const bool interior = (m0 + BM <= Meff) && (n0 + BN <= g.N) &&
                                  ((long long)BM * f.ldk < (1LL << 28)) && ((long long)BM * g.ldc < (1LL << 28));
            if (interior) {
                // Interior tiles: wave-uniform row bases (SGPRs) + one per-lane byte offset, so an element
                // costs a load, a store and ~5 VALU instructions -- beside the partner workgroup's MFMA stream
                // every further VALU instruction of this epilogue is paid at ~1 per MFMA issue slot.
                const int wu = __builtin_amdgcn_readfirstlane(wave);
                const int wmu = wu >> 1, wnu = wu & 1;
                const char* Kb = reinterpret_cast<const char*>(f.K + (size_t)(m0 + wmu * WM) * f.ldk + n0 + wnu * WN);
                char* Cb = reinterpret_cast<char*>(C + (size_t)(m0 + wmu * WM) * g.ldc + n0 + wnu * WN);
                const double* yb = f.y + m0 + wmu * WM;
                const unsigned vk = (unsigned)((lq * f.ldk + lr) * 8), vc = (unsigned)((lq * g.ldc + lr) * 8);
                const double nvar = f.var;
                if (TAG == 7) {
                    // k and h both from memory, loaded one row of the block at a time (8 values in flight per lane:
                    // two 4 x 4 register blocks would not fit beside the accumulators)
                    const char* Hb = reinterpret_cast<const char*>(f.H + (size_t)(m0 + wmu * WM) * f.ldk + n0 + wnu * WN);
#pragma unroll
                    for (int x = 0; x < TM; ++x) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const char* kr = Kb + (size_t)(x * 16 + 4 * r) * f.ldk * 8;
                            const char* hr = Hb + (size_t)(x * 16 + 4 * r) * f.ldk * 8;
                            char* cr = Cb + (size_t)(x * 16 + 4 * r) * g.ldc * 8;
                            double kv[TN], hv[TN];
#pragma unroll
                            for (int y = 0; y < TN; ++y) {
                                kv[y] = *reinterpret_cast<const double*>(kr + vk + y * 128);
                                hv[y] = *reinterpret_cast<const double*>(hr + vk + y * 128);
                            }
                            const double ys = f.ycoef * yb[x * 16 + 4 * r + lq];
#pragma unroll
                            for (int y = 0; y < TN; ++y) {
                                const double Q = fma(ys, wvc[y], acc[x][y][r]);
                                dv = fma(Q, kv[y], dv);
                                const double tval = Q * hv[y];
                                *reinterpret_cast<double*>(cr + vc + y * 128) = tval;
                                csum[y] += tval;
                            }
                        }
                    }
                } else
#pragma unroll
                for (int x = 0; x < TM; ++x) {
                    double kv[4][TN], yr[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const char* kr = Kb + (size_t)(x * 16 + 4 * r) * f.ldk * 8;
                        yr[r] = yb[x * 16 + 4 * r + lq];
#pragma unroll
                        for (int y = 0; y < TN; ++y) kv[r][y] = *reinterpret_cast<const double*>(kr + vk + y * 128);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        char* cr = Cb + (size_t)(x * 16 + 4 * r) * g.ldc * 8;
                        const double ys = f.ycoef * yr[r];
#pragma unroll
                        for (int y = 0; y < TN; ++y) {
                            const double Q = fma(ys, wvc[y], acc[x][y][r]);
                            dv = fma(Q, kv[r][y], dv);
                            const double tval = (kv[r][y] == nvar) ? 0.0 : -(Q * kv[r][y]);
                            *reinterpret_cast<double*>(cr + vc + y * 128) = tval;
                            csum[y] += tval;
                        }
                    }
                }
            } else
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * WM + x * 16 + lq + 4 * r;
                    const double yr = (row < Meff) ? f.ycoef * f.y[row] : 0.0;
#pragma unroll
                    for (int y = 0; y < TN; ++y) {
                        const int col = n0 + wn * WN + y * 16 + lr;
                        if (row < Meff && col < g.N) {
                            const double Kv = f.K[(size_t)row * f.ldk + col];
                            const double Q = fma(yr, wvc[y], acc[x][y][r]);
                            dv = fma(Q, Kv, dv);
                            const double tval = (TAG == 7) ? Q * f.H[(size_t)row * f.ldk + col]
                                                           : ((Kv == f.var) ? 0.0 : -(Q * Kv));
                            C[(size_t)row * g.ldc + col] = tval;
                            csum[y] += tval;
                        }
                    }
                }
            fused_epilogue_reduce<BN, WN, TN>(f, csum, dv, &Bs[0][0], tid, wm, wn, lr, lq, n0, g.N, ti,
                                              g.persist ? (size_t)t : (size_t)blockIdx.z * gridDim.x + blockIdx.x);
#ifdef EDRGP_TILE_STAMPS
            if (TAG == 2 && g.stamps != nullptr && tid == 0) {
                unsigned long long* o = g.stamps + (size_t)(g.persist ? t : (int)blockIdx.x) * 6;
                o[0] = ts_entry; o[1] = ts_loop; o[2] = ts_loop_end; o[3] = __builtin_amdgcn_s_memrealtime();
                o[4] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID
                o[5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);     // XCC_ID
            }
#endif
            continue;
        }
        if (TAG != 5) continue;                            // 2 / 7 carry the memory-resident epilogue only
        // one feature chunk (the host fuses only when D <= DC): stage, then one row block
        // of accumulators at a time to keep the register footprint at acc + 16 doubles
        __syncthreads();
        for (int e = tid; e < BM * DC; e += 256) {
            const int i = e / DC, q = e % DC;                // coalesced along the feature dimension
            double v = 0.0;
            if (q < f.D && m0 + i < Meff) v = f.X[(size_t)(m0 + i) * f.D + q] * f.invl[q];
            Xs[q * BM + i] = v;
        }
        for (int e = tid; e < BN * DC; e += 256) {
            const int j = e / DC, q = e % DC;
            double v = 0.0;
            if (q < f.D && n0 + j < g.N) v = f.Z[(size_t)(n0 + j) * f.D + q] * f.invl[q];
            Zs[q * BN + j] = v;
        }
        __syncthreads();
        double wvc[TN];
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int col = n0 + wn * WN + y * 16 + lr;
            wvc[y] = (col < g.N) ? f.wv[col] : 0.0;
        }
        double dv = 0.0;
        double csum[TN];
#pragma unroll
        for (int y = 0; y < TN; ++y) csum[y] = 0.0;
#pragma unroll
        for (int x = 0; x < TM; ++x) {
            double r2[TN][4];
#pragma unroll
            for (int y = 0; y < TN; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) r2[y][r] = 0.0;
            for (int q = 0; q < f.D; ++q) {
                double zc[TN], xr[4];
#pragma unroll
                for (int y = 0; y < TN; ++y) zc[y] = Zs[q * BN + wn * WN + y * 16 + lr];
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[r] = Xs[q * BM + wm * WM + x * 16 + lq + 4 * r];
#pragma unroll
                for (int y = 0; y < TN; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double d = xr[r] - zc[y];
                        r2[y][r] = fma(d, d, r2[y][r]);
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + x * 16 + lq + 4 * r;
                const double yr = (row < Meff) ? f.ycoef * f.y[row] : 0.0;
#pragma unroll
                for (int y = 0; y < TN; ++y) {
                    const int col = n0 + wn * WN + y * 16 + lr;
                    if (row < Meff && col < g.N) {
                        const double Q = fma(yr, wvc[y], acc[x][y][r]);
                        double Kv, hv;
                        edrgp_kfun(f.kind, f.var, r2[y][r], Kv, hv);
                        dv = fma(Q, Kv, dv);
                        const double tval = (r2[y][r] == 0.0) ? 0.0 : Q * hv;
                        C[(size_t)row * g.ldc + col] = tval;
                        csum[y] += tval;
                    }
                }
            }
        }
        fused_epilogue_reduce<BN, WN, TN>(f, csum, dv, &Bs[0][0], tid, wm, wn, lr, lq, n0, g.N, ti,
                                          g.persist ? (size_t)t : (size_t)blockIdx.z * gridDim.x + blockIdx.x);
#ifdef EDRGP_TILE_STAMPS
        if (TAG == 5 && g.stamps != nullptr && tid == 0) {
            unsigned long long* o = g.stamps + (size_t)(g.persist ? t : (int)blockIdx.x) * 6;
            o[0] = ts_entry; o[1] = ts_loop; o[2] = ts_loop_end; o[3] = __builtin_amdgcn_s_memrealtime();
            o[4] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            o[5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
        }
#endif
        continue;
    }

    // ---- TAG 4: plain store + column sums weighted by a row vector (W = K_nm Li^T and W^T y in one pass:
    //      the stable form's b, without reading W back)
    if (TAG == 4) {
        const FuseK& f = g.fusek;
        double bsum[TN];
#pragma unroll
        for (int y = 0; y < TN; ++y) bsum[y] = 0.0;
        if ((m0 + BM <= Meff) && (n0 + BN <= g.N) && ((long long)BM * g.ldc < (1LL << 28))) {
            // interior tiles: wave-uniform row base + one per-lane byte offset, a store and an fma per element (every
            // VALU instruction of an epilogue is paid beside the partner workgroup's MFMA stream: 15.6 us per pass with
            // the guarded form below)
            const int wu = __builtin_amdgcn_readfirstlane(wave);
            const int wmu = wu >> 1, wnu = wu & 1;
            char* Cb = reinterpret_cast<char*>(C + (size_t)(m0 + wmu * WM) * g.ldc + n0 + wnu * WN);
            const double* yb = f.y + m0 + wmu * WM;
            const unsigned vc = (unsigned)((lq * g.ldc + lr) * 8);
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    char* cr = Cb + (size_t)(x * 16 + 4 * r) * g.ldc * 8;
                    const double yr = yb[x * 16 + 4 * r + lq];
#pragma unroll
                    for (int y = 0; y < TN; ++y) {
                        const double v = acc[x][y][r];
                        *reinterpret_cast<double*>(cr + vc + y * 128) = v;
                        bsum[y] = fma(v, yr, bsum[y]);
                    }
                }
        } else
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + x * 16 + lq + 4 * r;
                const bool rok = row < Meff;
                const double yr = rok ? f.y[row] : 0.0;
#pragma unroll
                for (int y = 0; y < TN; ++y) {
                    const int col = n0 + wn * WN + y * 16 + lr;
                    if (rok && col < g.N) {
                        const double v = acc[x][y][r];
                        C[(size_t)row * g.ldc + col] = v;
                        bsum[y] = fma(v, yr, bsum[y]);
                    }
                }
            }
        // 8 partials per column (2 row-waves x 4 lane rows), summed in a fixed order
        __syncthreads();
        double* red = &As[0][0];
#pragma unroll
        for (int y = 0; y < TN; ++y) red[((wm * 4 + lq) * BN) + wn * WN + y * 16 + lr] = bsum[y];
        __syncthreads();
        for (int j = tid; j < BN; j += 256) {
            double sum = 0.0;
#pragma unroll
            for (int p = 0; p < 8; ++p) sum += red[p * BN + j];
            if (n0 + j < g.N) f.bpart[(size_t)ti * g.N + n0 + j] = sum;
        }
        continue;
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
    }   // pass
#ifdef EDRGP_TILE_STAMPS
    if (TAG == 1 && g.stamps != nullptr && threadIdx.x == 0) {      // Phi: entry / exit of every workgroup, pair flag
        unsigned long long* o = g.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 6;
        o[0] = ts_entry; o[1] = (unsigned long long)(pair_d0 >= 0); o[2] = 0; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    }
#endif
}

// C[i][j] = alpha * sum_s slab[s][i][j] + beta * C[i][j]; mirror: computed for
// i >= j and written to both triangles.
__global__ void slab_reduce_kernel(const double* __restrict__ slabs, int splits, long long slabStride,
                                   int M, int N, int lds, double alpha, double beta,
                                   double* __restrict__ C, int ldc, int mirror) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= M || j >= N) return;
    if (mirror && j > i) return;                   // 1: both triangles written; 2: lower triangle only
    // four partial sums, folded in a fixed order: four loads in flight instead of one dependent add per slab (the
    // D x D Gram matrix of the mean Jacobian has 390 slabs at config 4's N: 50 -> .. us)
    const double* p = slabs + (size_t)i * lds + j;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = 0;
    for (; k + 3 < splits; k += 4) {
        a0 += p[(size_t)k * slabStride];
        a1 += p[(size_t)(k + 1) * slabStride];
        a2 += p[(size_t)(k + 2) * slabStride];
        a3 += p[(size_t)(k + 3) * slabStride];
    }
    for (; k < splits; ++k) a0 += p[(size_t)k * slabStride];
    const double s = (a0 + a1) + (a2 + a3);
    double v = alpha * s;
    double* c = C + (size_t)i * ldc + j;
    if (beta != 0.0) v += beta * (*c);
    *c = v;
    if (mirror == 1 && i != j) C[(size_t)j * ldc + i] = v;
}

template <int BM, int BN>
static void launch_variant(bool akm, bool bkm, int tag, dim3 grid, hipStream_t s, const GemmArgs& g, unsigned dyn_lds) {
    if (tag == 1 && akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, true, 1>), grid, dim3(256), dyn_lds, s, g);
    else if (tag == 2 && !akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, true, 2>), grid, dim3(256), dyn_lds, s, g);
    else if (tag == 5 && !akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, true, 5>), grid, dim3(256), dyn_lds, s, g);
    else if (tag == 7 && !akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, true, 7>), grid, dim3(256), dyn_lds, s, g);
    else if (tag == 3 && !akm && !bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, false, 3>), grid, dim3(256), dyn_lds, s, g);
    else if (tag == 4 && !akm && !bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, false, 4>), grid, dim3(256), dyn_lds, s, g);
    else if (akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, true, 0>), grid, dim3(256), dyn_lds, s, g);
    else if (akm && !bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, false, 0>), grid, dim3(256), dyn_lds, s, g);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, true, 0>), grid, dim3(256), dyn_lds, s, g);
    else hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, false, 0>), grid, dim3(256), dyn_lds, s, g);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int gemm_impl(Ctx* c, bool transA, bool transB, int M, int N, int K, double alpha,
                     const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc,
                     const GemmOpts& o, int splits, double* slab, long long slabStride, int slab_ld) {
    if (M <= 0 || N <= 0) return EDRGP_OK;
    int tile = o.force_tile;
    if (tile < 0) {
        const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * o.batch / (o.tri ? 2 : 1);
        tile = (t128 >= 1024) ? 128 : o.small_tile;
    }
    if (tile == 0) {
        const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * o.batch * splits;
        // one 128 x 128 workgroup per CU runs at 47 TFLOP/s, four 64 x 64 ones at 54 (2048^3: 369 against 318 us): the
        // large tile needs two workgroups per CU to pay
        tile = (M >= 128 && N >= 128 && t128 >= 2LL * c->num_cu) ? 128 : 64;
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
    g.xmap = (o.tag == 1) && o.tri && splits > 1 && o.batch == 1 && c->syrk_xmap;
    g.diag_opt = 0;
    g.dephase = c->gemm_dephase;
    g.interleave = (o.kmode != 0 && c->gemm_interleave) ? 1 : 0;
    g.dbg_alias = (o.tag == 2) ? c->gemm_alias : 0;
    g.stamps = nullptr;
    g.fuse_on = 0;
    if (o.tag == 3 || o.tag == 4) {
        if (o.fusek == nullptr || splits != 1 || o.batch != 1 || o.tri) {
            set_error("kernel-matrix / column-sum epilogue needs an unsplit, unbatched, full-grid product");
            return EDRGP_ERR_INVALID;
        }
        if (o.tag == 4 && (transA || !transB || o.fusek->y == nullptr || o.fusek->bpart == nullptr || alpha != 1.0 || beta != 0.0)) {
            set_error("column-sum epilogue: NT product, alpha = 1, beta = 0, y and bpart required");
            return EDRGP_ERR_INVALID;
        }
        g.fusek = *o.fusek;
    }
    if (o.fuse != nullptr && o.tag == 2) {
        if (splits != 1 || o.batch != 1 || alpha != 1.0 || beta != 0.0) {
            set_error("fused gradient epilogue needs an unsplit, unbatched alpha=1 beta=0 product");
            return EDRGP_ERR_INVALID;
        }
        g.fuse = *o.fuse;
        g.fuse_on = 1;
    }
    // narrow outputs (T^T [X | X.^2] with 2D <= 32 columns, ...): a 64 x 32 tile -- the 64 x 64 tile would spend half its
    // MFMAs on padding (N = 100000, M = 2048: 496 -> 282 us for 32 columns)
    const bool narrow = (tile == 64 && N <= 32 && o.tag == 0 && !o.tri && o.kmode == 0 && transA && !transB && c->gemm_narrow);
    g.tiles_m = (M + tile - 1) / tile;
    g.tiles_n = narrow ? 1 : (N + tile - 1) / tile;
    g.vecA = aligned16(A) && (lda % 2 == 0) && (o.strideA % 2 == 0);
    g.vecB = aligned16(B) && (ldb % 2 == 0) && (o.strideB % 2 == 0);
    if (slab != nullptr) {
        g.C = slab; g.ldc = slab_ld; g.alpha = 1.0; g.beta = 0.0; g.slabStride = slabStride;
    } else {
        g.C = C; g.ldc = ldc; g.alpha = alpha; g.beta = beta; g.slabStride = 0;
    }
    // Phi only: diagonal tiles as 36-block loops, two per workgroup (full tiles, full k-tiles, aligned rows)
    g.diag_opt = (o.tag == 1) && o.tri && slab != nullptr && c->syrk_diag_opt && tile == 128 && transA && !transB &&
                 (M % 128 == 0) && (K % BK == 0) && g.vecA && ((long long)BK * lda < (1LL << 28)) && A == B && lda == ldb &&
                 o.batch == 1;
    long long nt = o.tri ? (long long)g.tiles_m * (g.tiles_m + 1) / 2 : (long long)g.tiles_m * g.tiles_n;
    if (g.interleave && !o.tri && o.tag == 0 && splits == 1 && o.batch == 1 && c->gemm_interleave >= 2 && nt % 256 == 0 &&
        (o.kmode == K_LE_ROW || o.kmode == K_GE_ROW || o.kmode == K_LE_COL || o.kmode == K_GE_COL) && c->num_cu == 256)
        g.interleave = 2 * o.kmode;
    g.pair = (o.tag == 4 && (o.kmode == K_LE_COL) && g.tiles_n >= 2 && c->gemm_pair) ? 1 : 0;
    if (g.pair) nt = (long long)g.tiles_m * ((g.tiles_n + 1) / 2);
    if (g.diag_opt) nt = (long long)g.tiles_m * (g.tiles_m - 1) / 2 + (g.tiles_m + 1) / 2;
    // the gradient contraction with its fused epilogue runs as a persistent grid when it has more than three rounds of tiles
    g.persist = 0;
    long long gx = nt;
    if (((o.tag == 2 && g.fuse_on) || (o.tag == 4 && g.pair && c->gemm_persist >= 2)) && tile == 128 && c->gemm_persist &&
        nt > 6LL * c->num_cu && nt < (1LL << 30)) {
        g.persist = (int)nt;
        gx = 2LL * c->num_cu;                          // a multiple of 8 (XCDs)
        if (gx % 8) { g.persist = 0; gx = nt; }
    }
    unsigned gy = (unsigned)splits;
    if (o.tag == 1 && g.diag_opt && g.xmap && c->syrk_persist && !c->syrk_shares_gpu && nt * splits >= 6LL * c->num_cu &&
        nt * splits < (1LL << 30) &&
        (2 * c->num_cu) % 8 == 0) {
        g.persist = (int)(nt * splits);
        gx = 2LL * c->num_cu;
        gy = 1;
    }
    dim3 grid((unsigned)gx, gy, (unsigned)o.batch);
    // op(A) is M x K: not transposed -> stored [M][K] (m-major); transposed -> stored [K][M] (k-major)
    const bool akm = transA;
    // op(B) is K x N: not transposed -> stored [K][N] (k-major); transposed -> stored [N][K]
    const bool bkm = !transB;
#ifdef EDRGP_TILE_STAMPS
    static unsigned long long* stamp_dev = nullptr;
        const bool stamp_phi = (o.tag == 1 && getenv("EDRGP_TILE_STAMPS_PHI") != nullptr);
    const size_t stamp_count = stamp_phi ? (size_t)nt * splits : (size_t)nt;
    if (((o.tag == 2 && g.fuse_on) || stamp_phi) && getenv("EDRGP_TILE_STAMPS_FILE") != nullptr) {
        if (stamp_dev == nullptr) HIP_TRY(hipMalloc(&stamp_dev, 64u << 20));
        if (stamp_count * 48 <= (64u << 20)) { HIP_TRY(hipMemsetAsync(stamp_dev, 0, stamp_count * 48, c->stream)); g.stamps = stamp_dev; }
    }
#endif
    // one fused epilogue per kernel: 2 = k from K_nm (RBF), 7 = k and h from memory, 5 = distances recomputed
    const int etag = (o.tag == 2 && g.fuse_on) ? (g.fuse.K == nullptr ? 5 : (g.fuse.H != nullptr ? 7 : 2)) : o.tag;
    if (narrow) hipLaunchKernelGGL((gemm_f64_kernel<64, 32, true, true, 0>), grid, dim3(256), 0, c->stream, g);
    else if (tile == 128) launch_variant<128, 128>(akm, bkm, etag, grid, c->stream, g, (unsigned)o.dyn_lds);
    else launch_variant<64, 64>(akm, bkm, etag, grid, c->stream, g, (unsigned)(o.dyn_lds ? o.dyn_lds : c->gemm_pad_now));
    HIP_TRY(hipGetLastError());
#ifdef EDRGP_TILE_STAMPS
    if ((o.tag == 2 || stamp_phi) && g.stamps == stamp_dev && stamp_dev != nullptr) {       // every launch overwrites the file: the last one stays
        std::vector<unsigned long long> hst(stamp_count * 6);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(hst.data(), stamp_dev, stamp_count * 48, hipMemcpyDeviceToHost));
        if (FILE* fp = fopen(getenv("EDRGP_TILE_STAMPS_FILE"), "wb")) { fwrite(hst.data(), 1, stamp_count * 48, fp); fclose(fp); }
    }
#endif
    return EDRGP_OK;
}

int gemm(Ctx* c, bool transA, bool transB, int M, int N, int K, double alpha, const double* A, int lda,
         const double* B, int ldb, double beta, double* C, int ldc, const GemmOpts& o) {
    // split-K only for long-K, small-output, unbatched products (e.g. T^T [X|1|X^2])
    int splits = 1;
    // (never for a product with a fused epilogue: its tiles are finished in one piece -- a gradient contraction's last,
    //  partly filled round at M >= 4096 has < 256 tiles and >= 256 k-tiles and used to be refused for that)
    if (o.batch == 1 && !o.tri && o.fuse == nullptr && o.fusek == nullptr) {
        const int tile = (o.force_tile > 0 ? o.force_tile : 64);
        const long long nt = (long long)((M + tile - 1) / tile) * ((N + tile - 1) / tile);
        const int kt = (K + BK - 1) / BK;
        if (nt < 256 && kt >= 256) {
            splits = (int)std::min<long long>((1024 + nt - 1) / nt, kt / 32);
            if (splits < 1) splits = 1;
        } else if (nt <= 128 && kt >= 64 && o.kmode == 0 && o.tag == 0 && c->skinny_split) {
            // skinny outputs with a medium k (the Kmm term's M x (2D+1) products at M = 2048: 64 workgroups of 128
            // k-tiles on 256 CUs, 190 us each): two rounds of shorter workgroups
            splits = (int)std::min<long long>((512 + nt - 1) / nt, kt / 16);
            if (splits < 1) splits = 1;
        }
    }
    // Lower-tiles-only products over triangular operands (A = Tm Li^T, R = LBi Li, P = R^T R, Kmm^-1 = Li^T Li at
    // M = 2048): a sixth of the full flops, but 528 workgroups whose longest walks all of K -- 160-185 us, the time of
    // its serial k-loop.  Cut along k into mid_split segments (empty ones leave at once) the chain is a quarter as
    // long; the segments meet in the slab reduction, which also mirrors (tools/mid_gemm_probe.py).
    int reduce_mode = 0;
    if (o.batch == 1 && o.tri && o.kmode != 0 && o.tag == 0 && c->mid_split > 1 && M == N && K >= 1024 && M >= 1024 &&
        true) {
        splits = std::min(c->mid_split, (K + BK - 1) / BK / 16);
        if (splits < 2) splits = 1;
        reduce_mode = 2;
    }
    if (splits > 1) {
        const long long slabStride = (long long)M * N;
        TRY(c->slabs.reserve((size_t)splits * slabStride * sizeof(double)));
        GemmOpts o2 = o;
        if (o2.force_tile <= 0) o2.force_tile = 64;
        TRY(gemm_impl(c, transA, transB, M, N, K, 1.0, A, lda, B, ldb, 0.0, nullptr, 0, o2, splits,
                      c->slabs.d(), slabStride, N));
        dim3 blk(64, 4), grd((N + 63) / 64, (M + 3) / 4);
        hipLaunchKernelGGL(slab_reduce_kernel, grd, blk, 0, c->stream, c->slabs.d(), splits, slabStride, M, N,
                           N, alpha, beta, C, ldc, reduce_mode ? (o.mirror_out ? 1 : 2) : 0);
        HIP_TRY(hipGetLastError());
        return EDRGP_OK;
    }
    TRY(gemm_impl(c, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, o, 1, nullptr, 0, 0));
    if (o.tri && o.mirror_out) TRY(mirror_lower(c, C, M, ldc));
    return EDRGP_OK;
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

// Split count of the Phi contraction.  2 workgroups per CU are resident, so the launch runs in
// rounds of 2 * num_cu workgroups and a nearly empty last round costs a whole round (5 % at Cm with a
// 25 %-full 5th round, 23 % at M = 4096 with 3 splits).  Cost model per candidate s:
//   rounds(s) * (work of one workgroup ~ 1/s)  +  slab traffic s * 2 M^2 elements,
// minimised over s with >= 256 rows per split.
int pick_syrk_splits(Ctx* c, long long nt, int kt, int64_t rows, int M, int tile, int elem_bytes) {
    const long long slots = 2LL * c->num_cu;
    const int max_splits = std::max(1, std::min(64, kt / 16));
    const double wg_rate = (elem_bytes == 8 ? 62e12 : 128e12) / (double)slots;        // flop/s per resident workgroup
    const double t_full = 2.0 * (double)rows * tile * tile / wg_rate;                  // one workgroup over all rows
    const double t_slab = 2.0 * (double)M * M * elem_bytes / 3.0e12;                   // write + read of one slab
    int best = 1;
    double best_cost = 1e300;
    for (int s = 1; s <= max_splits; ++s) {
        const long long rounds = (nt * s + slots - 1) / slots;
        // + 1/8 round: workgroups are not equally long (a diagonal-pair workgroup issues 18 MFMAs per k-step
        // against 16), so the last round overhangs; more, shorter rounds absorb that better
        const double cost = ((double)rounds + 0.125) * t_full / s + s * t_slab;
        if (cost < best_cost * (1.0 - 1e-3)) { best_cost = cost; best = s; }
    }
    return best;
}

int syrk_tn(Ctx* c, const double* K, int64_t rows, int M, int ldk, double* Phi, int ldphi, bool accumulate) {
    if (rows > 0x7fffffffLL) { set_error("syrk_tn: rows chunk too large"); return EDRGP_ERR_INVALID; }
    const int tile = (M >= 256) ? 128 : 64;
    const int tm = (M + tile - 1) / tile;
    long long nt = (long long)tm * (tm + 1) / 2;
    if (c->syrk_diag_opt && tile == 128 && M % 128 == 0 && rows % BK == 0 && ldk % 2 == 0)
        nt = (long long)tm * (tm - 1) / 2 + (tm + 1) / 2;          // diagonal tiles go two per workgroup (gemm_impl)
    const int kt = (int)((rows + BK - 1) / BK);
    int splits = c->syrk_splits;
    if (splits <= 0) splits = pick_syrk_splits(c, nt, kt, rows, M, tile, 8);
    if (splits < 1) splits = 1;
    if (splits > kt) splits = std::max(1, kt);
    {
        const int per = (kt + splits - 1) / splits;           // no empty trailing row ranges
        splits = (kt + per - 1) / per;
    }
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


// =====================================================================================================================
// fp32 row pass (BASELINE config 5: "fp32 path, MFMA fp32"), v_mfma_f32_16x16x4_f32: the same tile loader and k-loop
// (gemm_tile.h) instantiated for float.
//
// Mixed precision by construction: everything whose size grows with the number of rows N (scaled inputs, K_nm, the
// Phi = K_nm^T K_nm contraction, the gradient contraction T = (K_nm S2 + beta y wv^T) .* h(r) and T^T [X | X.^2]) is
// stored and multiplied in fp32 with fp32 MFMA accumulation inside one split-K slab; slabs, per-tile partial sums and
// every M x M object (Kmm, both Choleskys, the posterior factors) stay fp64.  The O(M^3) phase therefore keeps the
// fp64 code of the default path and only the N-proportional HBM traffic and MFMA time are halved.  Per type differ:
// the 16-byte vector (4 values, so BK = 32 keeps the tile bytes and the MFMA count per barrier), the C/D fragment map
// (reg r of lane l is C[row = 4 (l>>4) + r][col = l&15]; f64: (l>>4) + 4r) and the epilogues below (fp32 arithmetic on
// the tile, fp64 partial sums).
// =====================================================================================================================
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

template <int BMN, bool KM> using Loader32 = TileLoaderT<float, BMN, KM>;      // gemm_tile.h

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
        // the k-loop shared with the fp64 kernels (gemm_tile.h): two-stage fragment pipeline with 2 MFMAs between LDS reads
        // (fp32 contraction at config 5: -2.2 %; groups of 4: +1 %), next tile to LDS after k-step BK/4 - 2
        typedef GemmLoopCfg<FRAG_PIPE32, 2, EARLY_STORE, BK32 / 4 - 2, false, false> LoopCfg;
        gemm_mainloop<float, BM, BN, AKM, BKM, LoopCfg>(acc, A, B, g.lda, g.ldb, m0, n0, m0, n0, g.M, g.N, g.K, kt0, kt1,
                                                        g.vecA != 0, g.vecB != 0, As, Bs, tid, wm, wn, lr, lq, g.prio != 0);
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
