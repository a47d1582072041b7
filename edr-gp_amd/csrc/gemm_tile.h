// Scalar-type-generic pieces of the MFMA GEMM kernels (la_gemm.hip): operand tile loader, LDS tile images and the
// k-loop, written once for fp64 (v_mfma_f64_16x16x4_f64) and fp32 (v_mfma_f32_16x16x4_f32).  A 16-byte vector holds
// 2 doubles or 4 floats and a k-tile is 128 bytes deep (BK = 16 / 32), so the tile bytes, the number of loads per
// thread and the MFMA count per barrier are the same for both types; what differs is listed in GemmScalar<T>.
//
// MFMA fragment maps (cdna_hip_programming.md section 3):
//   A operand: lane l holds A[i = l&15][k = l>>4]                    (both types)
//   B operand: lane l holds B[k = l>>4][j = l&15]                    (both types)
//   C/D      : reg r of lane l is C[row][col = l&15], row = (l>>4) + 4r for f64, 4 (l>>4) + r for f32
#pragma once
#include <hip/hip_runtime.h>

namespace edrgp {

template <typename T> struct GemmScalar;
template <> struct GemmScalar<double> {
    static constexpr int VEC = 2, BK = 16;
    typedef double vec_t __attribute__((ext_vector_type(2)));
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int cd_row(int lq, int r) { return lq + 4 * r; }
};
template <> struct GemmScalar<float> {
    static constexpr int VEC = 4, BK = 32;
    typedef float vec_t __attribute__((ext_vector_type(4)));
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int cd_row(int lq, int r) { return 4 * lq + r; }
};

constexpr int GEMM_LPAD = 16;         // padding of a k-major image row, in elements

// Fragment reads of m-major operand images.  The compiler pairs adjacent ds_read_b64 of a fragment set into
// ds_read2_b64, which the LDS serves as two accesses of 4 x 16-lane groups with a 32-bank modulus
// (MI355X_MICROARCH.md, LDS).  On a k-major image (16 lanes = 128 contiguous bytes) that is conflict-free and
// halves the instruction count -- the Phi kernel lost 5 % when its reads were split (by hand-issued ds_read_b64 with
// explicit waits: -6 %; as volatile loads: -5 %).  On an m-major fp64 image (row stride 18 doubles) lanes l and l + 8 of
// a 16-lane group share a bank modulo 32: a 2-way conflict on every access (gradient contraction:
// SQ_LDS_BANK_CONFLICT = 15 % of SQ_LDS_IDX_ACTIVE, against 0 on the Phi kernel).  Plain ds_read_b64 (2 x 32 lanes,
// 64 banks) is conflict-free there, so m-major fp64 images are read through a volatile LDS pointer, which the load /
// store optimiser never merges and the scheduler still treats as an ordinary load (C2 bound+gradients -1.3 %,
// Cm -0.2 %: the conflicts were not what holds the contraction at 59 TFLOP/s).
template <bool VOL, typename T>
__device__ __forceinline__ T frag_read(const T* p) {
    if (VOL) {
        typedef const volatile __attribute__((address_space(3))) T* lds_vptr;     // keep it an LDS access (not flat)
        return *((lds_vptr)p);
    }
    return *p;
}

// Stage loader of one operand k-tile (256 threads).  KM = true : source element (k, i) at P[k*ld + i]
//                                                    KM = false: source element (k, i) at P[i*ld + k]
template <typename T, int BMN, bool KM>
struct TileLoaderT {
    typedef GemmScalar<T> S;
    typedef typename S::vec_t vec_t;
    static constexpr int VEC = S::VEC, BK = S::BK;
    static constexpr int NV = BMN * BK / (256 * VEC);     // 16-byte vectors per thread
    static constexpr int TPR = BMN / VEC;                 // KM: threads per k-row
    static constexpr int RPP = 256 / TPR;                 // KM: k-rows per pass
    static constexpr int TPK = BK / VEC;                  // m-major: threads per row (8 for both types)
    vec_t r[NV];

    __device__ __forceinline__ void load(const T* __restrict__ P, int ld, int mn0, int k0, int mn_lim, int k_lim, bool vec,
                                         int tid) {
        if (KM) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int k = k0 + v * RPP + tid / TPR;
                const int i = mn0 + (tid % TPR) * VEC;
                const T* p = P + (size_t)k * ld + i;
                if (vec && k < k_lim && i + VEC - 1 < mn_lim) {
                    r[v] = *reinterpret_cast<const vec_t*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) r[v][e] = (k < k_lim && i + e < mn_lim) ? p[e] : (T)0;
                }
            }
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = mn0 + v * (256 / TPK) + tid / TPK;
                const int k = k0 + (tid % TPK) * VEC;
                const T* p = P + (size_t)i * ld + k;
                if (vec && i < mn_lim && k + VEC - 1 < k_lim) {
                    r[v] = *reinterpret_cast<const vec_t*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) r[v][e] = (i < mn_lim && k + e < k_lim) ? p[e] : (T)0;
                }
            }
        }
    }

    // Interior fast path.  Measured on gfx950: while the SIMD's other wave streams f64 MFMAs, this wave's VALU
    // instructions issue about once per MFMA (~64 cycles each), so the ~100 address / bounds instructions of the
    // guarded loader cost ~5300 cycles per k-tile.  Here the per-thread byte offsets inside a k-tile are computed once
    // (32-bit), the tile base is wave-uniform (SGPR pair, advanced with SALU): the loop body is NV global_load_dwordx4
    // and nothing else.  Byte offsets relative to the tile origin (tile_origin below), < 2^31 by the caller's check.
    __device__ __forceinline__ void offsets(unsigned (&off)[NV], int ld, int tid) const {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            off[v] = KM ? (unsigned)(((v * RPP + tid / TPR) * ld + (tid % TPR) * VEC) * (int)sizeof(T))
                        : (unsigned)(((v * (256 / TPK) + tid / TPK) * ld + (tid % TPK) * VEC) * (int)sizeof(T));
    }
    // element offset of k-tile `kt` of the operand tile starting at row/col mn0
    static __device__ __forceinline__ long long tile_origin(int ld, int mn0, int kt) {
        return KM ? (long long)kt * BK * ld + mn0 : (long long)mn0 * ld + (long long)kt * BK;
    }
    // bytes from one k-tile to the next
    static __device__ __forceinline__ long long tile_step(int ld) {
        return KM ? (long long)BK * ld * (long long)sizeof(T) : (long long)BK * (long long)sizeof(T);
    }
    // rows of the source a k-tile (KM) / an operand tile (m-major) spans: the caller checks rows * ld * sizeof(T) < 2^31
    static constexpr int span_rows() { return KM ? BK : BMN; }

    __device__ __forceinline__ void load_fast(const char* __restrict__ tile_base, const unsigned (&off)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; ++v) r[v] = *reinterpret_cast<const vec_t*>(tile_base + off[v]);
    }

    // LDS image of one operand k-tile.  k-major sources keep [k][m] rows (stride BMN + 16 elements: the four k-rows of
    // a fragment read fall on disjoint parts of the banks); m-major sources keep their own [m][k] rows (stride
    // BK + VEC: 18 doubles / 36 floats), so both are filled with 16-byte stores and both give conflict-free fragment
    // reads (bank sets of the 16 rows x 4 k of a fragment are disjoint for either stride).
    static constexpr int SIZE = KM ? BK * (BMN + GEMM_LPAD) : BMN * (BK + VEC);
    static __device__ __forceinline__ int idx(int k, int m) { return KM ? k * (BMN + GEMM_LPAD) + m : m * (BK + VEC) + k; }
    static constexpr int cidx(int k, int m) { return KM ? k * (BMN + GEMM_LPAD) + m : m * (BK + VEC) + k; }   // linear in (k, m)

    __device__ __forceinline__ void store(T* __restrict__ Sm, int tid) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (KM) *reinterpret_cast<vec_t*>(&Sm[idx(v * RPP + tid / TPR, (tid % TPR) * VEC)]) = r[v];
            else *reinterpret_cast<vec_t*>(&Sm[idx((tid % TPK) * VEC, v * (256 / TPK) + tid / TPK)]) = r[v];
        }
    }
};

// What the k-loop of a kernel instance looks like: the fragment pipeline (fragments of k-step kk + 1 read while the
// MFMAs of step kk issue, PIPE_M MFMAs between two LDS reads), the k-step after whose MFMAs the next tile's registers
// go to LDS, which operand images are read through volatile pointers, and whether the interior and the guarded loader get
// a k-loop each (UNSWITCH) or share one with the choice inside it.
template <bool FRAG_PIPE_, int PIPE_M_, bool EARLY_STORE_, int EARLY_KK_, bool VOL_A_, bool VOL_B_, bool UNSWITCH_ = false>
struct GemmLoopCfg {
    static constexpr bool FRAG_PIPE = FRAG_PIPE_, EARLY_STORE = EARLY_STORE_, VOL_A = VOL_A_, VOL_B = VOL_B_;
    static constexpr bool UNSWITCH = UNSWITCH_;
    static constexpr int PIPE_M = PIPE_M_, EARLY_KK = EARLY_KK_;
};

// The k-loop of one output tile: acc += op(A)[m0 : m0 + BM, kt0*BK : kt1*BK] op(B)[.., n0 : n0 + BN].
// 4 waves in a 2 x 2 grid, a wave owns a (BM/2) x (BN/2) sub-tile of 16 x 16 MFMA blocks.  Tile t + 1 is prefetched
// global -> registers during the MFMAs of tile t and written to the free LDS buffer in the middle of the MFMA phase; one
// barrier per k-tile.  `prio`: the short staging / barrier phases run at raised priority so the partner wave's MFMA
// stream cannot starve them.  (m0l, n0l): where the operand tiles are actually read (timing probes alias them).
// The k-loop for one loader kind -- interior workgroup (full tile, full k-tiles, 16-byte aligned rows, 32-bit tile
// footprint: NV 16-byte loads per operand and k-tile, nothing else) or the guarded loader -- or, SPLIT = false, for
// both with the choice (`fast`) made inside the loop.  Which of the two forms a kernel takes is a measured choice
// (Cfg::UNSWITCH), because the two compile to different schedules and address forms (a loop per kind: wave-uniform base
// in an SGPR pair + 32-bit lane offsets; one loop: per-lane 64-bit addresses and a branch per k-tile):
//   Phi (same box, Cm):            a loop per kind 8.58 ms per bound evaluation, L2 hit rate 72 %; one loop 8.63 ms, 54 %
//   gradient contraction (Cm):     a loop per kind 22.38 ms per bound+gradient evaluation; one loop 22.08 ms
// (the hit rate: the co-running workgroups of an XCD share their K_nm rows in its L2 only while they walk them within
// ~16 k-tiles of each other, which the schedule influences; TCC_HIT / TCC_MISS of the Phi launch alone,
// tools/tcc_libs.sh -- neither kernel is bound by the fabric, 1-2 TB/s).
template <typename T, int BM, int BN, bool AKM, bool BKM, typename Cfg, bool SPLIT, bool FAST>
__device__ __forceinline__ void gemm_kloop(typename GemmScalar<T>::acc_t (&acc)[BM / 32][BN / 32], const T* __restrict__ A,
                                           const T* __restrict__ B, int lda, int ldb, int m0, int n0, int m0l, int n0l,
                                           int Mlim, int Nlim, int Klim, int kt0, int kt1, bool vA, bool vB,
                                           T (*As)[TileLoaderT<T, BM, AKM>::SIZE], T (*Bs)[TileLoaderT<T, BN, BKM>::SIZE],
                                           int tid, int wm, int wn, int lr, int lq, bool prio, bool fast_rt) {
    typedef GemmScalar<T> S;
    typedef TileLoaderT<T, BM, AKM> LA;
    typedef TileLoaderT<T, BN, BKM> LB;
    constexpr int BK = S::BK;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    LA la;
    LB lb;
    const bool fast = SPLIT ? FAST : fast_rt;
    unsigned offA[LA::NV], offB[LB::NV];
    const char* baseA = reinterpret_cast<const char*>(A + LA::tile_origin(lda, m0l, kt0));
    const char* baseB = reinterpret_cast<const char*>(B + LB::tile_origin(ldb, n0l, kt0));
    const long long stepA = LA::tile_step(lda), stepB = LB::tile_step(ldb);
    if (fast) {
        la.offsets(offA, lda, tid);
        lb.offsets(offB, ldb, tid);
        la.load_fast(baseA, offA);
        lb.load_fast(baseB, offB);
    } else {
        la.load(A, lda, m0, kt0 * BK, Mlim, Klim, vA, tid);
        lb.load(B, ldb, n0, kt0 * BK, Nlim, Klim, vB, tid);
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
                la.load(A, lda, m0, (kt + 1) * BK, Mlim, Klim, vA, tid);
                lb.load(B, ldb, n0, (kt + 1) * BK, Nlim, Klim, vB, tid);
            }
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        if constexpr (Cfg::FRAG_PIPE) {
            // Fragments of k-step kk + 1 are read while the MFMAs of step kk issue: written out as a two-stage register
            // pipeline and pinned with scheduling groups (one LDS read after every PIPE_M MFMAs), because the scheduler's
            // own choice flips with unrelated code changes between this order and "all reads after the last MFMA, then
            // wait" (1 % of the Phi launch, seen in the ISA of two builds).
            T a[2][TM], b[2][TN];
#pragma unroll
            for (int x = 0; x < TM; ++x) a[0][x] = frag_read<Cfg::VOL_A>(&As[buf][LA::idx(lq, wm * WM + x * 16 + lr)]);
#pragma unroll
            for (int x = 0; x < TN; ++x) b[0][x] = frag_read<Cfg::VOL_B>(&Bs[buf][LB::idx(lq, wn * WN + x * 16 + lr)]);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < BK / 4) {
#pragma unroll
                    for (int x = 0; x < TM; ++x)
                        a[nxt][x] = frag_read<Cfg::VOL_A>(&As[buf][LA::idx((kk + 1) * 4 + lq, wm * WM + x * 16 + lr)]);
#pragma unroll
                    for (int x = 0; x < TN; ++x)
                        b[nxt][x] = frag_read<Cfg::VOL_B>(&Bs[buf][LB::idx((kk + 1) * 4 + lq, wn * WN + x * 16 + lr)]);
                }
#pragma unroll
                for (int x = 0; x < TM; ++x)
#pragma unroll
                    for (int y = 0; y < TN; ++y) acc[x][y] = S::mfma(a[cur][x], b[cur][y], acc[x][y]);
                if (kk + 1 < BK / 4) {
#pragma unroll
                    for (int i = 0; i < TM + TN; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, Cfg::PIPE_M, 0);      // MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                // 1 LDS read
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                  // the rest of this step's MFMAs
                if (Cfg::EARLY_STORE && kk == Cfg::EARLY_KK) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        la.store(As[buf ^ 1], tid);
                        lb.store(Bs[buf ^ 1], tid);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                {
                    T a[TM], b[TN];
#pragma unroll
                    for (int x = 0; x < TM; ++x) a[x] = frag_read<Cfg::VOL_A>(&As[buf][LA::idx(kk * 4 + lq, wm * WM + x * 16 + lr)]);
#pragma unroll
                    for (int x = 0; x < TN; ++x) b[x] = frag_read<Cfg::VOL_B>(&Bs[buf][LB::idx(kk * 4 + lq, wn * WN + x * 16 + lr)]);
#pragma unroll
                    for (int x = 0; x < TM; ++x)
#pragma unroll
                        for (int y = 0; y < TN; ++y) acc[x][y] = S::mfma(a[x], b[y], acc[x][y]);
                }
                if (Cfg::EARLY_STORE && kk == BK / 4 - 2) {
                    // The next tile's registers go to the free LDS buffer in the MIDDLE of the MFMA phase (their loads
                    // were issued two k-steps ago), so the write latency hides under the remaining MFMAs and the tail
                    // of the iteration is the barrier only.
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        la.store(As[buf ^ 1], tid);
                        lb.store(Bs[buf ^ 1], tid);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (prio) __builtin_amdgcn_s_setprio(2);
        if (!Cfg::EARLY_STORE && more) {
            la.store(As[buf ^ 1], tid);
            lb.store(Bs[buf ^ 1], tid);
        }
        __syncthreads();
    }
}

template <typename T, int BM, int BN, bool AKM, bool BKM, typename Cfg>
__device__ __forceinline__ void gemm_mainloop(typename GemmScalar<T>::acc_t (&acc)[BM / 32][BN / 32], const T* __restrict__ A,
                                              const T* __restrict__ B, int lda, int ldb, int m0, int n0, int m0l, int n0l,
                                              int Mlim, int Nlim, int Klim, int kt0, int kt1, bool vA, bool vB,
                                              T (*As)[TileLoaderT<T, BM, AKM>::SIZE], T (*Bs)[TileLoaderT<T, BN, BKM>::SIZE],
                                              int tid, int wm, int wn, int lr, int lq, bool prio) {
    typedef TileLoaderT<T, BM, AKM> LA;
    typedef TileLoaderT<T, BN, BKM> LB;
    constexpr int BK = GemmScalar<T>::BK;
    // interior workgroup: full tile, full k-tiles, 16-byte aligned rows, 32-bit tile footprint
    const bool fast = vA && vB && (m0 + BM <= Mlim) && (n0 + BN <= Nlim) && (kt1 * BK <= Klim) &&
                      ((long long)LA::span_rows() * lda * (long long)sizeof(T) < (1LL << 31)) &&
                      ((long long)LB::span_rows() * ldb * (long long)sizeof(T) < (1LL << 31));
    if (!Cfg::UNSWITCH)
        gemm_kloop<T, BM, BN, AKM, BKM, Cfg, false, false>(acc, A, B, lda, ldb, m0, n0, m0l, n0l, Mlim, Nlim, Klim, kt0, kt1, vA,
                                                           vB, As, Bs, tid, wm, wn, lr, lq, prio, fast);
    else if (fast)
        gemm_kloop<T, BM, BN, AKM, BKM, Cfg, true, true>(acc, A, B, lda, ldb, m0, n0, m0l, n0l, Mlim, Nlim, Klim, kt0, kt1, vA,
                                                         vB, As, Bs, tid, wm, wn, lr, lq, prio, true);
    else
        gemm_kloop<T, BM, BN, AKM, BKM, Cfg, true, false>(acc, A, B, lda, ldb, m0, n0, m0l, n0l, Mlim, Nlim, Klim, kt0, kt1, vA,
                                                          vB, As, Bs, tid, wm, wn, lr, lq, prio, false);
}

}  // namespace edrgp
