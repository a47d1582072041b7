// Factorisation and inverse of ONE 64 x 64 diagonal block on one 256-thread workgroup (device code only).
// Shared by la_chol.hip (diag_block_kernel, and the fused step kernel whose workgroup 0 runs it on the block it has
// just finished) and by the standalone probe tools/diag_probe.hip.  GPy reaches this arithmetic through
// util.linalg.jitchol -> LAPACK dpotrf and pdinv -> dtrtri.
//
// The block is the critical path of the whole factorisation (2048 serial pivots at n = 2048, 32 block boundaries),
// so the body is organised around the chain of pivots:
//
//  * The 16 pivots of block column w (16 columns) run on wave w, which holds that panel as one row per lane.  The
//    trailing matrix lives in MFMA C/D tiles, row block i on wave i; when a panel's barrier falls every wave first
//    brings its tile of the NEXT block column up to date (4 MFMAs, fragments straight from the [k][row] image of L
//    in LDS) and writes it into the panel image, a second barrier, and the next wave reads its panel and starts.  On
//    the critical path between two panels: one tile update, one transposition through LDS (the single-wave form
//    updated all 6 / 3 / 1 remaining tiles in line).
//  * A pivot is a wave-uniform chain  pv -> 1/sqrt(pv) -> next pv  on values fetched with v_readlane BEFORE the
//    reciprocal square root is known (the entries (j+1, j) and (j+1, j+1) of the panel), so no cross-lane traffic and
//    no LDS round trip sits on it; the rank-1 update of the panel's other columns (one v_fma_f64 per column, the
//    multiplier broadcast from the column just written to LDS) fills the issue slots beside the chain.
//    1/sqrt: v_rsq_f64 (2^-23 relative, tools/diag_probe.hip measures it) + ONE third-order step
//    y = y0 + y0 e (1/2 + 3/8 e), e = 1 - pv y0^2: error ~ e^3 = 2^-69, four dependent operations instead of six.
//  * The inverse rides along.  Lanes 0..15 of wave w >= 1 carry no row of the panel (rows < 16 w are finished): they
//    hold the 16 x 16 identity, and the column operations of the 16 pivots turn it into inv(L_ww)^T -- lane c ends
//    with column c of T_w = inv(L_ww) at no additional instruction (wave 0, which has no idle lane, forms T_0 by
//    substitution while wave 1 works on its panel).  Block column j of the inverse belongs to wave j (j < i):
//    X_ij = -T_i sum_{k=j}^{i-1} L_ik X_kj on the matrix pipe, accumulator tiles used directly as the next
//    product's B operand; the sum is formed while panel i is still being factored (it needs L_ik, k < i, only), so
//    after the last pivot the whole inverse is 4 MFMAs and a store away (it used to be a separate phase of 7.5k cycles).
//  * Seven workgroup barriers, all of them on LDS counters only (outstanding global stores are not waited for).
#pragma once
#include <hip/hip_runtime.h>

namespace edrgp {

constexpr int NB = 64;
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// LDS footprint of the body, in doubles: Lt[64][65] | Pn[64][17] = Tl[4][16][17] (same storage: block w of Tl is rows
// 16 w .. 16 w + 15 of Pn, which only wave w itself ever uses as panel image) | rdiag[64] (inverse-only path): 42.5 KB,
// so that a workgroup carrying it fits into the LDS one retiring workgroup of the padded K_nm build leaves free on a
// CU -- see chol_step_kernel
constexpr int LTS = NB + 2;          // row stride of the image Lt[k][row]: even, so that pairs of broadcast multipliers are 16-byte aligned (ds_read_b128, immediate offsets)
constexpr int DIAG_LDS = NB * LTS + 4 * 16 * 17 + NB;

// workgroup barrier that orders LDS traffic only
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// same-wave LDS write -> read: a wave's LDS instructions execute in order, no s_barrier; the fence keeps the compiler
// from moving later reads above the store
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct DiagOut {
    double* Lw;          // factor block (j0, j0) of the output matrix
    int ldw;
    bool vec;            // 16-byte stores allowed (even ldw, aligned base)
    double* Dinv;        // dense 64 x 64 slab (ld 64) or null; upper blocks are never written
    double* inv;         // block (j0, j0) of an n x n inverse (ld ldi) or null
    int ldi;
    int jb;              // valid rows / columns of the block
    int* info;
    int info_base;       // info value of pivot 0 of this block minus 1
    unsigned long long* stamps;
};

// The 16 pivots of panel W on the wave that holds it as one row per lane (see the header comment)
template <int W>
__device__ __forceinline__ void diag_panel_pivots(double (*Lt)[LTS], const int lane, double (&p)[16], double (&yv)[16]) {
    constexpr int OFF = 16 * W;
    // Per pivot j (column jj of the panel): the wave-uniform chain, l = column jj scaled, and the rank-1 update of the
    // columns to its right, p[kk] -= l * L[OFF + kk][j].  The multipliers L[OFF + kk][j] are entries of l in other
    // lanes.  For kk = jj + 1, jj + 2 -- needed by the very next pivots -- they are formed from values fetched with
    // v_readlane before y is known (a1 y, a2 y: no dependence on l).  For kk >= jj + 3 they come back from the column
    // just written to LDS as broadcast reads, and are CONSUMED ONE PIVOT LATER (cm[], lprev): an LDS write -> read
    // round trip is 85 cycles and a broadcast read costs a lone wave ~19 cycles of LDS -> register traffic, so taking
    // them in the same iteration put 85 + 19 (15 - jj) cycles in front of the next v_rsq_f64 (370 cycles per pivot,
    // tools/diag_probe.hip); one iteration later they have arrived while the next chain was being issued (256).
    double pv = readlane_f64(p[0], OFF);
    double cm[16];                       // multipliers of the previous pivot for the columns it has not updated yet
    double lprev = 0.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = OFF + jj;
        double a1 = 0.0, b1 = 0.0, a2 = 0.0;
        if (jj < 15) {                   // entries (j+1, j), (j+1, j+1), (j+2, j): known before 1/sqrt(pv) is
            a1 = readlane_f64(p[jj], j + 1);
            b1 = readlane_f64(p[jj + 1], j + 1);
        }
        if (jj < 14) a2 = readlane_f64(p[jj], j + 2);
        const double y0 = __builtin_amdgcn_rsq(pv);
        const double t = pv * y0;
        const double e = fma(-t, y0, 1.0);
        const double c3 = fma(0.375, e, 0.5);
        const double ye = y0 * e;
        const double y = fma(ye, c3, y0);
        if (W == 0) yv[jj] = y;
        const double an1 = a1 * y, an2 = a2 * y;
        if (jj < 15) pv = fma(-an1, an1, b1);        // exactly what lane j+1 computes for its diagonal entry below
        const double l = p[jj] * y;
        p[jj] = l;
        Lt[j][lane] = l;                 // column j of L (lanes above the panel write zeros or, lanes 0..15, the
                                         // riding inverse into the unused upper triangle of the image)
        if (jj < 15) p[jj + 1] = fma(-l, an1, p[jj + 1]);
        if (jj < 14) p[jj + 2] = fma(-l, an2, p[jj + 2]);
        __builtin_amdgcn_sched_barrier(0);
        // the previous pivot's deferred columns (kk >= (jj - 1) + 3), then this pivot's broadcasts are requested
        if (jj > 0) {
#pragma unroll
            for (int kk = jj + 2; kk < 16; ++kk) p[kk] = fma(-lprev, cm[kk], p[kk]);
        }
        wave_lds_order();
        // (16-byte reads of aligned pairs: the image's row stride is even)
        if (((jj + 3) & 1) && jj + 3 < 16) cm[jj + 3] = Lt[j][OFF + jj + 3];
#pragma unroll
        for (int kk = (jj + 4) & ~1; kk < 16; kk += 2) {
            const double2 v = *reinterpret_cast<const double2*>(&Lt[j][OFF + kk]);
            cm[kk] = v.x;
            cm[kk + 1] = v.y;
        }
        lprev = l;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The program of wave W (see the header comment).  `Lt` holds the lower triangle of the block as Lt[col][row],
// identity padded beyond jb, zero above the diagonal.  Every wave runs the same seven barriers: B(s) when panel s is
// factored, B'(s) when the image of panel s + 1 is complete.
//   Wave W owns row block W of the trailing matrix -- the tiles (W, b), 1 <= b <= W, in MFMA C/D layout -- and block
//   column W of the inverse.  After B(s) every wave W > s first brings its tile of block column s + 1 up to date with
//   panel s (4 MFMAs, fragments from the [k][row] image of L) and writes it into the panel image; after B'(s) wave
//   s + 1 reads that image as one row per lane and starts on its pivots, while the others update their remaining
//   tiles.  On the critical path between two panels: 8 fragment reads, 4 MFMAs, 4 LDS writes, a barrier and 16 LDS
//   reads (when one wave updated all tiles of the next block column itself it was 16 + 12 + 12 + 16).
template <int W>
__device__ __forceinline__ void diag_wave_program(double (*Lt)[LTS], double (*Pn)[17], double (*Tl)[16][17],
                                                  const int lane, const DiagOut& o) {
    constexpr int OFF = 16 * W;
    const int lr = lane & 15, lq = lane >> 4;
    double p[16];
    double yv[16];                       // W == 0 only: 1 / L[j][j] for the substitution
    d4 T[4];                             // T[b] = tile (W, b), 1 <= b <= W
    d4 X[4];                             // X[i] = X_iW, i >= W, C/D layout: reg r = row lq + 4 r, col lr
    d4 S = {0.0, 0.0, 0.0, 0.0};
    if (W == 0) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) p[cc] = Lt[cc][lane];
    } else {
#pragma unroll
        for (int b = 1; b <= W; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[b][r] = Lt[16 * b + lr][OFF + lq + 4 * r];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        // ---- in front of barrier B(s)
        if (s == W) {
            diag_panel_pivots<W>(Lt, lane, p, yv);
            if (W > 0 && lane < 16) {
                // lane c holds column c of T_W = inv(L_WW): reg i = T[i][c]
#pragma unroll
                for (int i = 0; i < 16; ++i) Tl[W][i][lane] = p[i];
            }
        } else if (s > W) {
            // block (s, W) of the inverse, first half: S = sum_k L_sk X_kW (L_sk, k < s, is final since B(s - 1))
            S = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = W; k < s; ++k)
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const double af = Lt[16 * k + 4 * st + lq][16 * s + lr];        // L[16 s + lr][16 k + 4 st + lq]
                    S = __builtin_amdgcn_mfma_f64_16x16x4f64(af, X[k][st], S, 0, 0, 0);
                }
        }
        lds_barrier();                   // B(s): panel s (and T_s) are in LDS
        if (o.stamps != nullptr && lane == 0 && s == W) o.stamps[8 + W] = __builtin_amdgcn_s_memtime();
        double af[4];
        if (s < 3) {
            if (W > s) {
                // urgent: this wave's tile of block column s + 1, then into the panel image
#pragma unroll
                for (int st = 0; st < 4; ++st) af[st] = -Lt[16 * s + 4 * st + lq][OFF + lr];
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const double bf = Lt[16 * s + 4 * st + lq][16 * (s + 1) + lr];
                    T[s + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[st], bf, T[s + 1], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) Pn[OFF + lq + 4 * r][lr] = T[s + 1][r];
            }
            lds_barrier();               // B'(s): the image of panel s + 1 is complete
        }
        // ---- off the critical path
        if (s == W) {
            // the factor's block column W goes out (fire and forget: the barriers do not wait for global stores)
            const int row = lane;
            if (row < o.jb) {
                double* dst = o.Lw + (size_t)row * o.ldw + OFF;
                if (o.vec && OFF + 16 <= o.jb) {
#pragma unroll
                    for (int cc = 0; cc < 16; cc += 2) {
                        double2 v;
                        v.x = (row >= OFF + cc) ? p[cc] : 0.0;
                        v.y = (row >= OFF + cc + 1) ? p[cc + 1] : 0.0;
                        *reinterpret_cast<double2*>(dst + cc) = v;
                    }
                } else {
#pragma unroll
                    for (int cc = 0; cc < 16; ++cc)
                        if (OFF + cc < o.jb) dst[cc] = (row >= OFF + cc) ? p[cc] : 0.0;
                }
            }
            if (W == 0) {
                // T_0 = inv(L_00) by substitution (lane 16 b + c: column c; the four lane groups repeat the same work)
                double t[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = (i == lr) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    t[k] *= yv[k];
                    const double tk = t[k];
#pragma unroll
                    for (int i = k + 1; i < 16; ++i) t[i] = fma(-Lt[k][i], tk, t[i]);
                }
                if (lane < 16) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) Tl[0][i][lane] = t[i];
                }
                wave_lds_order();
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) X[W][r] = Tl[W][lq + 4 * r][lr];
        } else if (s > W) {
            // block (s, W) of the inverse, second half: X_sW = -T_s S
            d4 Y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const double tf = -Tl[s][lr][4 * st + lq];           // A fragment of -T_ss
                Y = __builtin_amdgcn_mfma_f64_16x16x4f64(tf, S[st], Y, 0, 0, 0);
            }
            X[s] = Y;
        }
        if (s < 3) {
            if (W == s + 1) {
                // the panel as one row per lane; lanes 0..15 take the identity (the inverse of the diagonal block
                // rides along)
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) p[cc] = (lane >= OFF) ? Pn[lane][cc] : ((lane == cc) ? 1.0 : 0.0);
            } else if (W > s + 1) {
                // the other tiles of this row block
#pragma unroll
                for (int b = s + 2; b <= W; ++b)
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const double bf = Lt[16 * s + 4 * st + lq][16 * b + lr];
                        T[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[st], bf, T[b], 0, 0, 0);
                    }
            }
        }
    }
    if (W == 3) {
        // all barriers have fallen: first non-positive pivot, if any.  A bad pivot makes its column NaN (rsq of a
        // non-positive number, or 0 * inf), and NaN spreads to every later pivot: the first lane whose diagonal entry
        // is not a positive number is the first bad pivot.
        const double d = Lt[lane][lane];
        const unsigned long long badm = __ballot(!(d > 0.0));
        if (badm != 0ull && lane == 0 && o.info != nullptr)
            atomicCAS(o.info, 0, o.info_base + (int)__builtin_ctzll(badm) + 1);
    }
    if (o.stamps != nullptr && lane == 0) o.stamps[12 + W] = __builtin_amdgcn_s_memtime();
    if (o.Dinv != nullptr) {
#pragma unroll
        for (int i = W; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) o.Dinv[(16 * i + lq + 4 * r) * NB + OFF + lr] = X[i][r];
    }
    if (o.inv != nullptr) {
#pragma unroll
        for (int i = W; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + lq + 4 * r, col = OFF + lr;
                if (row < o.jb && col < o.jb) o.inv[(size_t)row * o.ldi + col] = X[i][r];
            }
    }
}

// DO_CHOL: factor the block (jb <= 64 valid rows / columns, identity padding) and report the first non-positive
// pivot; then (both modes) invert it.  Outputs: the factor in place or, with Lbase, to the same block of Lbase; the
// inverse into Dinv_slab (dense 64 x 64, ld 64; upper blocks are never written and must be zero) and / or inv_out.
// 256 threads, `lds` = DIAG_LDS doubles, uniform call per workgroup.  PRELOADED: the caller has filled the image
// Lt[col][row] (identity padded, zero above the diagonal) and synchronised.
template <bool DO_CHOL, bool PRELOADED = false>
__device__ __forceinline__ void diag_block_body(double* __restrict__ Abase, int lda, int n, int blk,
                                                double* __restrict__ Dinv_slab, double* __restrict__ inv_out, int ldi,
                                                int* __restrict__ info, unsigned long long* __restrict__ stamps,
                                                double* __restrict__ Lbase, int ldlo, int info_off,
                                                double* __restrict__ lds) {
    lds = static_cast<double*>(__builtin_assume_aligned(lds, 16));
    double (*Lt)[LTS] = reinterpret_cast<double (*)[LTS]>(lds);                       // Lt[k][i] = L[i][k]
    double (*Tl)[16][17] = reinterpret_cast<double (*)[16][17]>(lds + NB * LTS);       // inverses of the 16x16 diagonal blocks
    double* rdiag = lds + NB * LTS + 4 * 16 * 17;                                      // 1 / L[i][i] (inverse-only path)
    double (*Pn)[17] = reinterpret_cast<double (*)[17]>(lds + NB * LTS);               // panel image (same storage as Tl)
#define EDRGP_STAMP(slot) do { if (!PRELOADED && stamps != nullptr && threadIdx.x == 0) stamps[slot] = __builtin_amdgcn_s_memtime(); } while (0)
    EDRGP_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: the wave programs branch uniformly
    const int j0 = blk * NB;
    const int jb = min(NB, n - j0);
    double* A = Abase + (size_t)j0 * lda + j0;
    if (!PRELOADED) {
        const int cl = min(lane, jb - 1);
        const int r0 = 16 * wave;                   // coalesced, clamped (branch-free), 16 rows in flight per wave
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = A[(size_t)min(r0 + u, jb - 1) * lda + cl];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = r0 + u;
            const bool in = (r < jb) && (lane < jb) && (lane <= r);
            Lt[lane][r] = in ? t[u] : ((r == lane) ? 1.0 : 0.0);
        }
        __syncthreads();
    }
    EDRGP_STAMP(1);

    if (DO_CHOL) {
        DiagOut o;
        o.Lw = (Lbase != nullptr) ? Lbase + (size_t)j0 * ldlo + j0 : A;
        o.ldw = (Lbase != nullptr) ? ldlo : lda;
        o.vec = ((reinterpret_cast<uintptr_t>(o.Lw) & 15) == 0) && ((o.ldw & 1) == 0);
        o.Dinv = Dinv_slab;
        o.inv = (inv_out != nullptr) ? inv_out + (size_t)j0 * ldi + j0 : nullptr;
        o.ldi = ldi;
        o.jb = jb;
        o.info = info;
        o.info_base = info_off + j0;
        o.stamps = stamps;
        if (wave == 0) diag_wave_program<0>(Lt, Pn, Tl, lane, o);
        else if (wave == 1) diag_wave_program<1>(Lt, Pn, Tl, lane, o);
        else if (wave == 2) diag_wave_program<2>(Lt, Pn, Tl, lane, o);
        else diag_wave_program<3>(Lt, Pn, Tl, lane, o);
        EDRGP_STAMP(5);
        return;
    }

    // ---- inverse only (recursive-doubling trtri of large matrices): rdiag, the four 16 x 16 diagonal blocks by
    // substitution on wave 0, then block substitution on the matrix pipe, wave w owning block column w
    if (wave == 0) rdiag[lane] = 1.0 / Lt[lane][lane];
    __syncthreads();
    EDRGP_STAMP(2);
    if (wave == 0) {
        const int b = lane >> 4, cidx = lane & 15, o = 16 * b;
        double t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = (i == cidx) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            t[k] *= rdiag[o + k];
            const double tk = t[k];
#pragma unroll
            for (int i = k + 1; i < 16; ++i) t[i] = fma(-Lt[o + k][o + i], tk, t[i]);    // L_bb[i][k]
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) Tl[b][i][cidx] = t[i];
    }
    __syncthreads();
    EDRGP_STAMP(3);
    const int lr = lane & 15, lq = lane >> 4;
    const int j = wave;
    d4 X[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) X[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
        if (jj == j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) X[jj][r] = Tl[jj][lq + 4 * r][lr];
        }
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        if (i > j) {
            d4 S = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < i; ++k) {
                if (k >= j) {
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const double af = Lt[16 * k + 4 * st + lq][16 * i + lr];
                        S = __builtin_amdgcn_mfma_f64_16x16x4f64(af, X[k][st], S, 0, 0, 0);
                    }
                }
            }
            d4 Y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const double tf = -Tl[i][lr][4 * st + lq];
                Y = __builtin_amdgcn_mfma_f64_16x16x4f64(tf, S[st], Y, 0, 0, 0);
            }
            X[i] = Y;
        }
    }
    EDRGP_STAMP(4);
    if (Dinv_slab) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i >= j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Dinv_slab[(16 * i + lq + 4 * r) * NB + 16 * j + lr] = X[i][r];
            }
    }
    if (inv_out) {
        double* O = inv_out + (size_t)j0 * ldi + j0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i >= j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + lq + 4 * r, col = 16 * j + lr;
                    if (row < jb && col < jb) O[(size_t)row * ldi + col] = X[i][r];
                }
            }
    }
    EDRGP_STAMP(5);
#undef EDRGP_STAMP
}

}  // namespace edrgp
