// Blocked right-looking Cholesky and recursive triangular inverse, fp64.
//
// potrf_lower: nb = 64.  Per step: (1) ONE single-wave workgroup factors the
// 64x64 diagonal block (rows in registers, right-looking, fully unrolled) and
// inverts it with fp64 MFMA (below); (2) panel  L21 = A21 inv(L11)^T  as an MFMA
// GEMM, in place; (3) trailing update  A22 -= L21 L21^T  on the lower 64x64 tile
// set (MFMA GEMM).  GPy reaches this through jitchol -> LAPACK dpotrf.
//
// trtri_lower: Linv = L^-1 by recursive doubling: the 64x64 diagonal blocks are
// inverted in one batched launch of the same kernel, then at level s every pair
// of adjacent s-blocks is merged with two batched MFMA GEMMs
//     inv([[L11, 0], [L21, L22]]) = [[I11, 0], [-I22 L21 I11, I22]].
//
// 64x64 inverse on one wave.  A lone wave gets a fraction of the LDS rate
// (measured: 61k cycles for the 2016 broadcast reads of a scalar substitution;
// v_readlane broadcasts 2x slower still; scalar-cache loads latency-bound), so
// the bulk goes through the matrix pipe: the four 16x16 diagonal blocks are
// inverted by a 16-step substitution (lane 16b+c owns column c of block b), the
// six off-diagonal 16x16 blocks by block substitution
//     X_ij = -T_ii sum_{k=j}^{i-1} L_ik X_kj
// as v_mfma_f64_16x16x4_f64 products.  An accumulator tile IS a B operand for
// the next product (C/D reg r holds row (lane>>4)+4r = k of k-step r), so the X
// tiles never leave registers; only A fragments (blocks of L and T) come from LDS.
#include "common.h"
#include "chol_diag.h"      // NB, d4, DIAG_LDS, diag_block_body: one 64 x 64 diagonal block on one workgroup

namespace edrgp {

#ifdef EDRGP_TILE_STAMPS
// probe build (tools/chol_stamps.py): phases of workgroup 0 of every step launch -- the critical path of the
// factorisation -- on the 100 MHz wall clock (slots 0..7) and, inside the diagonal-block body, in shader cycles
// (slots 8..15, see chol_diag.h); 16 slots per step
__device__ unsigned long long* g_chol_stamps = nullptr;
#define CHOL_STAMP(slot)                                                                                        \
    do {                                                                                                        \
        if (g_chol_stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)                 \
            g_chol_stamps[(size_t)k * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();                         \
    } while (0)
#else
#define CHOL_STAMP(slot) do { } while (0)
#endif

// Strides of a batch of independent matrices (blockIdx.y of the factorisation kernels); all zero = one matrix.
struct CholBatch {
    long long sA = 0, sL = 0, sD = 0, sC = 0, sI = 0;   // work matrix, factor workspace, Dinv slabs, C rows, inverse
    int sInfo = 0;
};

template <bool DO_CHOL>
__global__ __launch_bounds__(256) void diag_block_kernel(double* __restrict__ Abase, int lda, int n,
                                                        int first_block, double* __restrict__ Dinv,
                                                        double* __restrict__ inv_out, int ldi,
                                                        int* __restrict__ info,
                                                        unsigned long long* __restrict__ stamps,
                                                        double* __restrict__ Lbase = nullptr, int ldlo = 0,
                                                        int info_off = 0, CholBatch bt = CholBatch()) {
    __shared__ __attribute__((aligned(16))) double lds[DIAG_LDS];
    const long long z = blockIdx.y;
    diag_block_body<DO_CHOL>(Abase + z * bt.sA, lda, n, first_block + (int)blockIdx.x,
                             Dinv ? Dinv + z * bt.sD + (size_t)blockIdx.x * NB * NB : nullptr,
                             inv_out ? inv_out + z * bt.sI : nullptr, ldi, info ? info + z * bt.sInfo : nullptr, stamps,
                             Lbase ? Lbase + z * bt.sL : nullptr, ldlo, info_off, lds);
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

// One right-looking step behind diagonal block k, ONE launch: for every lower 64x64 tile (i, j),
// k < j <= i, of the trailing matrix
//     L_ik = A_ik Dinv^T,  L_jk = A_jk Dinv^T      (recomputed per tile: 2 x 64^3 flops on the matrix pipe
//                                                   instead of a separate panel launch + its kernel boundary)
//     A_ij -= L_ik L_jk^T
// and the tiles of the first trailing column (j = k+1) store L_ik to Lout.  The panel column of A is only
// read, so there is no ordering between workgroups.  256 threads = 2x2 waves, a wave owns a 32x32 quadrant.
// LDS: the two operand tiles as [row][k] images (stride 66: conflict-free ds_read_b64 per half-wave), reused
// for L_ik / L_jk; the Dinv fragments (same for both products) sit in registers.
//
// The inverse rides along (Cw != nullptr): the identity is appended below A as block rows C_0 .. C_{T-1} and goes
// through the same panel + trailing steps, which turn it into L^-T (forward substitution of the identity by the
// updates that run anyway, as the right-hand side row of bound-only evaluations does):
//     C_ik <- C_ik Dinv^T          (i <= k: C is upper triangular)       = final block (i, k) of L^-T
//     C_ij -= C_ik L_jk^T          (j > k)
// Workgroups [ntri, ntri + (k + 1) r) take these tiles (block row ci of C, trailing column tj); those of the first
// trailing column store the finished block TRANSPOSED into Linv (block (k, ci) of L^-1).  Block row k of C is still
// the identity when step k reads it and no tile of C is read before step `row` has written it, so C needs no
// initialisation.  A last launch (panel_only) finishes block column T - 1.  This replaces the recursive-doubling
// triangular inverse (12 launches, 0.34 ms at n = 2048) at no additional launch.
// LDS: ONE 64x66 tile image, refilled as the step proceeds, plus the diagonal-block body's extra 8.7 KB: 42.5 KB.
// That is less than what one retiring workgroup of the K_nm build frees on a CU that holds four of them, so the
// steps of the Kmm factorisation (high-priority side stream) get workgroup slots while the build streams beside them;
// with two tile images (67.6 KB) a step waited for two build workgroups of one CU to retire together, and the first
// two steps of phase 1 took 494 + 164 us instead of 2 x 38.
// (Every barrier of the kernel orders LDS traffic only -- lds_barrier() -- so that none of them waits for the global
// stores of L_ik / the updated tile that are still in flight: __syncthreads() drains them, ~1 us each time.)
template <bool CAPPED>
__global__ __launch_bounds__(256, CAPPED ? 1 : 2) void chol_step_kernel(double* __restrict__ A, int lda, double* __restrict__ Lout,
                                                           int ldl, const double* __restrict__ Dinv, int n, int k,
                                                           double* __restrict__ Dinv_next, int* __restrict__ info,
                                                           int info_off, int ntri, double* __restrict__ Cw, int ldc,
                                                           double* __restrict__ Linv, int ldi, int panel_only,
                                                           CholBatch bt, int ntiles) {
    constexpr int LS = NB + 2;
    constexpr int LDS_DOUBLES = (NB * LS > DIAG_LDS) ? NB * LS : DIAG_LDS;
    __shared__ __attribute__((aligned(16))) double Rbuf[LDS_DOUBLES];
    {   // blockIdx.y: one of a batch of independent factorisations advancing in the same launches
        const long long z = blockIdx.y;
        A += z * bt.sA;
        Lout += z * bt.sL;
        Dinv += z * bt.sD;
        if (Dinv_next != nullptr) Dinv_next += z * bt.sD;
        if (info != nullptr) info += z * bt.sInfo;
        if (Cw != nullptr) Cw += z * bt.sC;
        if (Linv != nullptr) Linv += z * bt.sI;
    }
    double* R = Rbuf;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: the k-step limits below branch uniformly
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lq = lane >> 4;
    // Column blocks of a wave: {wn, 3 - wn} (16 columns each), not {2 wn, 2 wn + 1}.  Dinv is lower triangular, so
    // the panel products  (tile) Dinv^T  need only the k-steps st < 4 (cb + 1) of column block cb; blocks {0, 3} and
    // {1, 2} give both wave columns 20 of the 32 (k-step, block) pairs: 40 MFMAs per product instead of 64.
    const int cb16[2] = {16 * wn, 16 * (3 - wn)};
    const int klim[2] = {4 * (wn + 1), 4 * (4 - wn)};

    // A launch never has more workgroups than CUs (the host caps the grid): a step took 37 us instead of 28 as soon
    // as its tiles outnumbered the CUs and workgroup 0 -- the critical path: its tile, then the factorisation and
    // inverse of the next diagonal block -- had to share its CU.  The surplus tiles are walked by the workgroups
    // other than 0 (each ~6 us against workgroup 0's ~25).
    const int G = (int)gridDim.x;
    CHOL_STAMP(0);
    d4 Cij[2][2];                        // the tile being updated (workgroup 0 keeps it for the diagonal-block body)
    bool next_diag = false;
    for (int tile = (int)blockIdx.x; tile < ntiles;) {
    const bool inv_tile = (tile >= ntri);
    int ti, tj;
    if (!inv_tile) {
        const int t = tile;
        ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        tj = t - ti * (ti + 1) / 2;
    } else {
        const int u = tile - ntri;
        ti = u % (k + 1);                // block row of C
        tj = u / (k + 1);                // trailing column (0 when panel_only)
    }
    const int k0 = k * NB;
    const int i0 = inv_tile ? ti * NB : (k + 1 + ti) * NB;
    const int j0 = (k + 1 + tj) * NB;
    const bool diag = !inv_tile && (ti == tj);
    const bool ident = inv_tile && (ti == k);          // block row k of C: still the identity, its tiles right of k zero
    const bool two = !diag && !panel_only;             // a second panel tile (row block j of A) takes part
    // the "i" panel tile and the tile being updated live in A (Cholesky tiles) or in C (inverse tiles)
    const double* Pi = inv_tile ? Cw : A;
    double* Ct = inv_tile ? Cw : A;
    const int ldp = inv_tile ? ldc : lda;

    // Dinv fragments (loaded below): b[st][y] = B[q = 4 st + lq][c = 32 wn + 16 y + lr] = Dinv[c][q]
    // Every global operand is requested up front -- the Dinv fragments, the tile being updated (used in the epilogue)
    // and both panel tiles (thread -> rows tid/32 + 8 v, columns 2 (tid % 32), 16-byte loads; the second one waits in
    // registers).  The common case -- whole 64 x 64 tiles, aligned rows -- is ONE basic block of unconditional loads:
    // operands a tile does not use (the identity block row of C, the second panel tile of a diagonal tile) are loaded
    // from their (valid) addresses anyway and replaced afterwards.  (The guarded form had branches between the load
    // groups and an LDS store per row group; the compiler waited for each group in turn -- up to ten memory latencies,
    // 6.8 us of a 26 us step on the critical path: tools/chol_stamps.py.)
    const int c2 = (tid & 31) * 2, r8 = tid >> 5;
    double db[16][2];
    double2 a1v[8], a2v[8];
    {
        const bool al = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((lda & 1) == 0);
        const bool alp = ((reinterpret_cast<uintptr_t>(Pi) & 15) == 0) && ((ldp & 1) == 0);
        const bool common = al && alp && !panel_only && (i0 + NB <= n) && (j0 + NB <= n);         // uniform
        if (common) {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                a1v[v] = *reinterpret_cast<const double2*>(Pi + (size_t)(i0 + r8 + 8 * v) * ldp + k0 + c2);
                a2v[v] = *reinterpret_cast<const double2*>(A + (size_t)(j0 + r8 + 8 * v) * lda + k0 + c2);
            }
            // Dinv arrives as 16-byte row pieces and goes through the LDS image first: its fragments are 8-byte pieces of
            // 16 different rows per load instruction, and 32 such loads per lane kept the CU's vector cache busy for
            // 3 us (entry -> operands staged 5.9 -> 2.9 us without them, tools/chol_stamps.py with -DEDRGP_PROBE_NODB)
            double2 dv[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) dv[v] = *reinterpret_cast<const double2*>(Dinv + (size_t)(r8 + 8 * v) * NB + c2);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cij[x][y][r] = Ct[(size_t)(i0 + 32 * wm + 16 * x + lq + 4 * r) * ldp + j0 + cb16[y] + lr];
            if (ident) {                                     // block row k of C: still the identity, zero right of block k
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const int r = r8 + 8 * v;
                    a1v[v].x = (r == c2) ? 1.0 : 0.0;
                    a1v[v].y = (r == c2 + 1) ? 1.0 : 0.0;
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) Cij[x][y] = d4{0.0, 0.0, 0.0, 0.0};
            }
#pragma unroll
            for (int v = 0; v < 8; ++v) *reinterpret_cast<double2*>(&R[(r8 + 8 * v) * LS + c2]) = dv[v];
            lds_barrier();
#pragma unroll
            for (int st = 0; st < 16; ++st)
#pragma unroll
                for (int y = 0; y < 2; ++y) db[st][y] = (st < klim[y]) ? R[(cb16[y] + lr) * LS + 4 * st + lq] : 0.0;
            lds_barrier();                                 // the image is free for tile i
        } else {
#pragma unroll
            for (int st = 0; st < 16; ++st)
#pragma unroll
                for (int y = 0; y < 2; ++y) db[st][y] = Dinv[(cb16[y] + lr) * NB + 4 * st + lq];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i0 + 32 * wm + 16 * x + lq + 4 * r, col = j0 + cb16[y] + lr;
                        Cij[x][y][r] = (!panel_only && !ident && row < n && col < n) ? Ct[(size_t)row * ldp + col] : 0.0;
                    }
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int r = r8 + 8 * v;
                double2 a1 = {0.0, 0.0};
                a2v[v] = double2{0.0, 0.0};
                if (ident) {
                    a1.x = (r == c2) ? 1.0 : 0.0;
                    a1.y = (r == c2 + 1) ? 1.0 : 0.0;
                } else if (i0 + r < n) {
                    // (the panel columns are whole except in the panel-only launch, whose block column may be ragged)
                    const double* p = Pi + (size_t)(i0 + r) * ldp + k0 + c2;
                    if (alp && k0 + c2 + 1 < n) a1 = *reinterpret_cast<const double2*>(p);
                    else { a1.x = (k0 + c2 < n) ? p[0] : 0.0; a1.y = (k0 + c2 + 1 < n) ? p[1] : 0.0; }
                }
                if (two && j0 + r < n) {
                    const double* p = A + (size_t)(j0 + r) * lda + k0 + c2;
                    if (al) a2v[v] = *reinterpret_cast<const double2*>(p); else { a2v[v].x = p[0]; a2v[v].y = p[1]; }
                }
                a1v[v] = a1;
            }
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) *reinterpret_cast<double2*>(&R[(r8 + 8 * v) * LS + c2]) = a1v[v];
    }
    lds_barrier();
    if (tile == 0) CHOL_STAMP(1);

    // panel products on the matrix pipe: L_i = (tile i) Dinv^T, then L_j likewise through the same LDS image
    d4 Li[2][2], Lj[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) { Li[x][y] = d4{0.0, 0.0, 0.0, 0.0}; Lj[x][y] = d4{0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        double a1[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) a1[x] = R[(32 * wm + 16 * x + lr) * LS + 4 * st + lq];
#pragma unroll
        for (int y = 0; y < 2; ++y)
            if (st < klim[y]) {
#pragma unroll
                for (int x = 0; x < 2; ++x) Li[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[x], db[st][y], Li[x][y], 0, 0, 0);
            }
    }
    lds_barrier();                     // everyone has read tile i
    if (tile == 0) CHOL_STAMP(2);
    if (two) {
#pragma unroll
        for (int v = 0; v < 8; ++v) *reinterpret_cast<double2*>(&R[(r8 + 8 * v) * LS + c2]) = a2v[v];
        lds_barrier();
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            double a2[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) a2[x] = R[(32 * wm + 16 * x + lr) * LS + 4 * st + lq];
#pragma unroll
            for (int y = 0; y < 2; ++y)
                if (st < klim[y]) {
#pragma unroll
                    for (int x = 0; x < 2; ++x) Lj[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[x], db[st][y], Lj[x][y], 0, 0, 0);
                }
        }
        lds_barrier();                 // everyone has read tile j
    }
    // L_i through LDS: its [row][k] image feeds the A fragments of the update (kept in registers) and the stores
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 32 * wm + 16 * x + lq + 4 * r, col = cb16[y] + lr;
                R[row * LS + col] = Li[x][y][r];
                if (!inv_tile && tj == 0 && i0 + row < n) Lout[(size_t)(i0 + row) * ldl + k0 + col] = Li[x][y][r];
            }
    lds_barrier();
    if (inv_tile && tj == 0) {
        // finished block (ci, k) of L^-T -> block (k, ci) of L^-1, transposed through LDS: consecutive lanes write
        // consecutive columns of one row of Linv
        const int cc = tid & 63;                     // column of Linv within the block = row of the C tile
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int rr = (tid >> 6) + 4 * v;         // row of Linv within the block = column of the C tile
            if (k0 + rr < n && i0 + cc < n) Linv[(size_t)(k0 + rr) * ldi + i0 + cc] = R[cc * LS + rr];
        }
    }
    if (panel_only) return;
    if (tile == 0) CHOL_STAMP(3);
    double af[16][2];
#pragma unroll
    for (int st = 0; st < 16; ++st)
#pragma unroll
        for (int x = 0; x < 2; ++x) af[st][x] = R[(32 * wm + 16 * x + lr) * LS + 4 * st + lq];
    if (two) {
        lds_barrier();                 // the A fragments (and the transposed store) are out of the image
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) R[(32 * wm + 16 * x + lq + 4 * r) * LS + cb16[y] + lr] = Lj[x][y][r];
        lds_barrier();
    }
    // the update: S = L_i L_j^T (diagonal tile: L_i L_i^T), B fragments from the image of L_j
    d4 S[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) S[x][y] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        double b[2];
#pragma unroll
        for (int y = 0; y < 2; ++y) b[y] = R[(cb16[y] + lr) * LS + 4 * st + lq];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) S[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[st][x], b[y], S[x][y], 0, 0, 0);
    }
    if (tile == 0) CHOL_STAMP(4);
    next_diag = (Dinv_next != nullptr && tile == 0);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + 32 * wm + 16 * x + lq + 4 * r, col = j0 + cb16[y] + lr;
                Cij[x][y][r] -= S[x][y][r];
                if (!next_diag && row < n && col < n) Ct[(size_t)row * ldp + col] = Cij[x][y][r];
            }
    // next tile of this workgroup (workgroup 0 has none; an uncapped launch has one tile per workgroup)
    if (!CAPPED || blockIdx.x == 0) break;
    tile = (tile < G) ? G + ((int)blockIdx.x - 1) : tile + (G - 1);
    lds_barrier();                     // the LDS image is reused
    }   // tile loop
    // The workgroup of tile (k+1, k+1) has just produced the next diagonal block: it factors and inverts it on the
    // spot, straight out of its registers (no round trip through memory: the work matrix never needs this tile
    // again), so the next step needs no separate diagonal-block launch on the critical path.
    if (next_diag) {
        lds_barrier();                 // every wave is done with the image of L_i
        double (*Lt)[LTS] = reinterpret_cast<double (*)[LTS]>(Rbuf);      // Lt[k][i] = block[i][k], identity padded
        const int jb = min(NB, n - (k + 1) * NB);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 32 * wm + 16 * x + lq + 4 * r, col = cb16[y] + lr;
                    const bool in = (row < jb) && (col < jb) && (col <= row);
                    Lt[col][row] = in ? Cij[x][y][r] : ((row == col) ? 1.0 : 0.0);
                }
        lds_barrier();
        CHOL_STAMP(5);
#ifdef EDRGP_TILE_STAMPS
        unsigned long long* body_stamps = (g_chol_stamps != nullptr && blockIdx.y == 0) ? g_chol_stamps + (size_t)k * 16 : nullptr;
#else
        unsigned long long* body_stamps = nullptr;
#endif
        diag_block_body<true, true>(A, lda, n, k + 1, Dinv_next, nullptr, 0, info, body_stamps, Lout, ldl, info_off, Rbuf);
        CHOL_STAMP(6);
    }
}

// A = lower triangle of L (explicit zeros above the diagonal)
__global__ void take_lower_kernel(double* __restrict__ A, int lda, const double* __restrict__ L, int ldl, int n,
                                  long long sA = 0, long long sL = 0) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    A += (long long)blockIdx.z * sA;
    L += (long long)blockIdx.z * sL;
    if (i < n && j < n) A[(size_t)i * lda + j] = (j <= i) ? L[(size_t)i * ldl + j] : 0.0;
}

// Fused-step variant of potrf_lower (same contract): the factor is assembled in the panel workspace and
// copied back, A is the work matrix of the trailing updates.  With Linv != nullptr the inverse L^-1 comes out of
// the same launches (see chol_step_kernel).
static int potrf_lower_fused(Ctx* c, double* A, int n, int lda, int* info_dev, bool reset_info = true,
                             int info_off = 0, double* Linv = nullptr, int ldi = 0, int batch = 1, long long strideA = 0,
                             long long strideI = 0, bool copy_back = true) {
    const size_t G = (size_t)std::max(1, batch);
    if (reset_info) HIP_TRY(hipMemsetAsync(info_dev, 0, G * sizeof(int), c->stream));
    if (c->dinv.bytes < G * 2 * NB * NB * sizeof(double)) {
        TRY(c->dinv.reserve(G * 2 * NB * NB * sizeof(double)));
        // blocks above the diagonal are never written by the kernel
        HIP_TRY(hipMemsetAsync(c->dinv.d(), 0, G * 2 * NB * NB * sizeof(double), c->stream));
    }
    TRY(c->panel.reserve(G * n * n * sizeof(double)));
    double* Dinv = c->dinv.d();                       // two slabs: step b reads [b & 1] and writes [(b + 1) & 1]
    double* Lw = c->panel.d();
    double* Cw = nullptr;
    CholBatch bt;
    if (batch > 1) {
        bt.sA = strideA;
        bt.sL = (long long)n * n;
        bt.sD = 2LL * NB * NB;
        bt.sC = (long long)n * n;
        bt.sI = strideI;
        bt.sInfo = 1;
    }
    if (Linv != nullptr) {
        TRY(c->trtri_ws.reserve(G * n * n * sizeof(double)));
        Cw = c->trtri_ws.d();                         // the appended identity rows (never initialised: see the kernel)
        for (size_t z = 0; z < G; ++z) {
            double* Lz = Linv + z * (size_t)strideI;
            if (ldi == n) {
                hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, c->stream, Lz, (size_t)n * n, 0.0);
            } else {
                HIP_TRY(hipMemset2DAsync(Lz, (size_t)ldi * sizeof(double), 0, (size_t)n * sizeof(double), (size_t)n, c->stream));
            }
        }
    }
    const int nblk = (n + NB - 1) / NB;
#ifdef EDRGP_TILE_STAMPS
    static unsigned long long* stamp_dev = nullptr;
    const bool stamping = getenv("EDRGP_CHOL_STAMPS_FILE") != nullptr && nblk <= 64;
    if (stamping) {
        if (stamp_dev == nullptr) HIP_TRY(hipMalloc(&stamp_dev, 64 * 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(stamp_dev, 0, 64 * 16 * sizeof(unsigned long long), c->stream));
        HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_chol_stamps), &stamp_dev, sizeof(stamp_dev), 0, hipMemcpyHostToDevice, c->stream));
    }
#endif
    hipLaunchKernelGGL(diag_block_kernel<true>, dim3(1, (unsigned)G), dim3(256), 0, c->stream, A, lda, n, 0, Dinv,
                       (double*)nullptr, 0, info_dev, (unsigned long long*)nullptr, Lw, n, info_off, bt);
    for (int b = 0; b + 1 < nblk; ++b) {
        const int r = nblk - b - 1;
        const int ntri = r * (r + 1) / 2;
        const int ninv = (Cw != nullptr) ? (b + 1) * r : 0;
        // one launch per step: panel + trailing update of every tile (of A and of the appended identity rows), and
        // the factor + inverse of the next diagonal block by the workgroup that has just finished it
        const int ntiles = ntri + ninv;
        // one matrix with the GPU to itself: never more workgroups than CUs (see the kernel; in isolation 1094 ->
        // 1001 us at n = 2048).  A batch fills the chip anyway, and beside the K_nm build the capped variant (one
        // workgroup per CU by its register budget) finds slots later than the plain one (+0.25 ms per evaluation)
        const int gx = (G == 1 && c->chol_cap_grid && !c->chol_shares_gpu && ntiles > c->num_cu) ? c->num_cu : ntiles;
        if (gx < ntiles)
            hipLaunchKernelGGL(chol_step_kernel<true>, dim3((unsigned)gx, (unsigned)G), dim3(256), 0, c->stream, A, lda, Lw,
                               n, Dinv + (size_t)(b & 1) * NB * NB, n, b, Dinv + (size_t)((b + 1) & 1) * NB * NB, info_dev,
                               info_off, ntri, Cw, n, Linv, ldi, 0, bt, ntiles);
        else
            hipLaunchKernelGGL(chol_step_kernel<false>, dim3((unsigned)gx, (unsigned)G), dim3(256), 0, c->stream, A, lda, Lw,
                               n, Dinv + (size_t)(b & 1) * NB * NB, n, b, Dinv + (size_t)((b + 1) & 1) * NB * NB, info_dev,
                               info_off, ntri, Cw, n, Linv, ldi, 0, bt, ntiles);
    }
    if (Cw != nullptr)                                // block column nblk - 1 of L^-T: panel products only
        hipLaunchKernelGGL(chol_step_kernel<false>, dim3((unsigned)nblk, (unsigned)G), dim3(256), 0, c->stream, A, lda, Lw, n,
                           Dinv + (size_t)((nblk - 1) & 1) * NB * NB, n, nblk - 1, (double*)nullptr, info_dev, info_off, 0, Cw,
                           n, Linv, ldi, 1, bt, nblk);
    if (copy_back) {
        dim3 blk(64, 4), grd((n + 63) / 64, (n + 3) / 4, (unsigned)G);
        hipLaunchKernelGGL(take_lower_kernel, grd, blk, 0, c->stream, A, lda, Lw, n, n, bt.sA, bt.sL);
    }
    HIP_TRY(hipGetLastError());
#ifdef EDRGP_TILE_STAMPS
    if (stamping) {         // every factorisation overwrites the file: the last one stays
        std::vector<unsigned long long> hst(64 * 16);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(hst.data(), stamp_dev, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE* fp = fopen(getenv("EDRGP_CHOL_STAMPS_FILE"), "wb")) { fwrite(hst.data(), 8, hst.size(), fp); fclose(fp); }
    }
#endif
    return EDRGP_OK;
}

// `batch` independent n x n matrices (stride strideA) factored and inverted in the SAME launches: the steps are
// latency-bound chains that leave the chip almost idle, so a batch costs about what one matrix does (BASELINE
// config 4: the 16 length-scales of a candidate).  n in (64, 3072]; info_dev holds `batch` words.
int potrf_inv_lower_batched(Ctx* c, double* A, int n, int lda, long long strideA, int* info_dev, double* Linv, int ldi,
                            long long strideI, int batch) {
    if (batch < 1 || n <= NB || n > 3072) { set_error("potrf_inv_lower_batched: n must be in (64, 3072]"); return EDRGP_ERR_INVALID; }
    return potrf_lower_fused(c, A, n, lda, info_dev, true, 0, Linv, ldi, batch, strideA, strideI);
}

// Cholesky factor and its inverse: A <- L (lower, zeros above), Linv <- L^-1.  n in (64, 3072]: out of the same
// launches; otherwise factorisation, then the recursive-doubling inverse.
int potrf_inv_lower(Ctx* c, double* A, int n, int lda, int* info_dev, double* Linv, int ldi) {
    if (c->chol_fused && c->chol_inv_fused && n > NB && n <= 3072)
        return potrf_lower_fused(c, A, n, lda, info_dev, true, 0, Linv, ldi);
    TRY(potrf_lower(c, A, n, lda, info_dev));
    return trtri_lower(c, A, lda, Linv, ldi, n);
}

// Factorisation whose caller only reads the factor: it stays in the workspace it was assembled in (*L, leading
// dimension *ldl, zeros above the diagonal are NOT guaranteed there) and A holds the updated work matrix; sizes the
// fused path does not cover are factored in place.
int potrf_lower_ws(Ctx* c, double* A, int n, int lda, int* info_dev, const double** L, int* ldl) {
    if (c->chol_fused && n > NB && n <= 3072) {
        TRY(potrf_lower_fused(c, A, n, lda, info_dev, true, 0, nullptr, 0, 1, 0, 0, false));
        *L = c->panel.d();
        *ldl = n;
        return EDRGP_OK;
    }
    TRY(potrf_lower(c, A, n, lda, info_dev));
    *L = A;
    *ldl = lda;
    return EDRGP_OK;
}

int potrf_lower(Ctx* c, double* A, int n, int lda, int* info_dev) {
    // one launch per step wins while the steps are latency-bound (M = 2048: -3 % on the O(M^3) phase); its
    // recomputed panel tiles triple the MFMA work, which costs +13 % at n = 4096 and +28 % at 8192
    if (c->chol_fused && n > NB && n <= 3072) return potrf_lower_fused(c, A, n, lda, info_dev);
    HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), c->stream));
    if (c->dinv.bytes < (size_t)NB * NB * sizeof(double)) {
        TRY(c->dinv.reserve((size_t)NB * NB * sizeof(double)));
        // blocks above the diagonal are never written by the kernel
        HIP_TRY(hipMemsetAsync(c->dinv.d(), 0, (size_t)NB * NB * sizeof(double), c->stream));
    }
    double* Dinv = c->dinv.d();
    const int nblk = (n + NB - 1) / NB;
    for (int b = 0; b < nblk; ++b) {
        const int j0 = b * NB;
        const int jb = std::min(NB, n - j0);
        hipLaunchKernelGGL(diag_block_kernel<true>, dim3(1), dim3(256), 0, c->stream, A, lda, n, b, Dinv,
                           (double*)nullptr, 0, info_dev, (unsigned long long*)nullptr);
        const int rem = n - j0 - jb;
        if (rem > 0) {
            double* A21 = A + (size_t)(j0 + jb) * lda + j0;
            // L21 = A21 * Dinv^T in place: a workgroup owns whole rows of the 64-wide
            // panel (one column tile) and has consumed them before its epilogue stores.
            GemmOpts po;
            po.force_tile = 64;
            po.kmode = K_LE_COL;                 // Dinv is lower triangular
            TRY(gemm(c, false, true, rem, jb, jb, 1.0, A21, lda, Dinv, NB, 0.0, A21, lda, po));
            // A22 -= L21 L21^T on the lower 64x64 tiles (K = 64: many small workgroups
            // beat few large ones here -- 14 us vs 38 us per update)
            double* A22 = A + (size_t)(j0 + jb) * lda + (j0 + jb);
            GemmOpts o;
            o.tri = 1;
            o.force_tile = 64;
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
    hipLaunchKernelGGL(diag_block_kernel<false>, dim3(nblk), dim3(256), 0, c->stream, const_cast<double*>(L), ldl,
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
