// Shared by the evaluation pipelines (sparse_eval.hip, api.hip): workspace indices, device scalar slots,
// the side-stream scope and a few small helpers around the GEMM layer.
#pragma once
#include "common.h"
#include <algorithm>
#include <cmath>

namespace edrgp {

enum {  // c->mats
    MAT_LM = 0,   // Kmm + jitter, overwritten by its Cholesky factor
    MAT_LI,       // Lm^-1
    MAT_PHI,      // [Phi | b]  (M*M + M)
    MAT_T,        // scratch
    MAT_A,        // beta Lm^-1 Phi Lm^-T
    MAT_LB,       // I + A, overwritten by its Cholesky factor
    MAT_LBI,      // LB^-1
    MAT_R,        // LB^-1 Lm^-1
    MAT_P,        // R^T R
    MAT_KMMI,     // Lm^-T Lm^-1
    MAT_S2,       // 2 dL/dpsi2
    MAT_C1,       // scratch
    MAT_DKMM,     // dL/dKmm
    MAT_TMM,      // element-wise factor for the Kmm term
    MAT_KZ,       // K(Z, Z) without jitter and
    MAT_HZ,       // h(r(Z, Z)) for the Kmm term of single-part kernels
};

enum {  // device scalar slots (c->scal)
    SC_TRA = 0, SC_SUMLOGLB, SC_DATAFIT, SC_USQ, SC_LBIFROB, SC_YY,
    SC_DVAR_NM = 6,          // + part (4)
    SC_DVAR_MM = 10,         // + part (4)
    SC_YALPHA = 14, SC_SUMLOGL, SC_ALPHASQ, SC_TRKYI,
    SC_DIAGMIN = 18, SC_DIAGMAX = 19,          // of diag(Lm): written by phase 1, survive phase 3's reset
    SC_LBDIAGMIN = 20, SC_LBDIAGMAX = 21, SC_PHIFROB = 22,
    SC_COUNT = 32
};

// Runs the enclosed launches on the context's side stream (with its own scratch
// buffers); join_into_main() makes the main stream wait for them.
struct SideStream {
    Ctx* c;
    explicit SideStream(Ctx* c_) : c(c_) { swap(); }
    ~SideStream() { swap(); }
    void swap() {
        std::swap(c->stream, c->stream2);
        c->red.swap(c->red2);
        c->slabs.swap(c->slabs2);
        c->ks_x.swap(c->ks_x2);          // the side stream builds Kmm while the main one builds K_nm
        c->ks_xn.swap(c->ks_xn2);
    }
    // while active c->stream IS the side stream and c->stream2 the main one
    int start_after_main() {
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream2));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_fork, 0));
        return EDRGP_OK;
    }
    int mark_done() {                      // side work queued so far is what the main stream will wait for
        HIP_TRY(hipEventRecord(c->ev_join, c->stream));
        return EDRGP_OK;
    }
    int join_into_main() {
        TRY(mark_done());
        HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_join, 0));
        return EDRGP_OK;
    }
};

inline int main_waits_side(Ctx* c) {       // outside a SideStream scope: c->stream is the main stream
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return EDRGP_OK;
}

extern int g_tri_tile;

inline GemmOpts kopt(int kmode, int tri = 0, int tag = 0) {
    GemmOpts o;
    o.kmode = kmode;
    o.tri = tri;
    o.tag = tag;
    // zero-skipping makes the k-range depend on the tile: with one 128x128 tile per CU the
    // longest tile sets the time, 64x64 tiles (4 per CU, mixed lengths) balance it
    // (at M >= 4096 there are >= 1024 tiles of 128x128: the larger tile's MFMA efficiency wins again --
    //  gemm_impl resolves force_tile = -1 by the tile count)
    if (kmode != 0 && tag == 0) { o.force_tile = -1; o.small_tile = g_tri_tile; }
    return o;
}

// C (n x n, symmetric) = L^T L for lower-triangular L: lower tiles, zero-skipping, mirrored
inline int gram_of_lower(Ctx* c, int n, const double* L, double* C) {
    GemmOpts o = kopt(K_GE_ROW | K_GE_COL, 1);
    o.mirror_out = 1;
    return gemm(c, true, false, n, n, n, 1.0, L, n, L, n, 0.0, C, n, o);
}

inline KSpec make_kspec(const Ctx* c) {
    KSpec ks;
    ks.nparts = c->nparts;
    ks.D = c->D;
    for (int p = 0; p < EDRGP_MAX_PARTS; ++p) {
        ks.kind[p] = (p < c->nparts) ? c->kind[p] : 0;
        ks.var[p] = (p < c->nparts) ? c->var[p] : 0.0;
    }
    ks.invl = c->invl.d();
    return ks;
}

inline double var_sum(const Ctx* c) {
    double s = 0.0;
    for (int p = 0; p < c->nparts; ++p) s += c->var[p];
    return s;
}

int ensure_mats(Ctx* c, int n, int count);
// W = K Li^T and b = W^T y of the stable form (sparse_eval.hip)
int stable_product(Ctx* c, const double* K, int64_t rows, int M, const double* Li, const double* y, double* W, double* b);
int dense_kernel_gradients(Ctx* c, const double* dK, double* Tn, std::vector<double>* dvar, std::vector<double>* dlen);

// The pivot-ratio limit of the automatic statistics form for M inducing points: "pivot_ratio_max" up to M = 2048, tighter
// above it.  The Gram form's bound error was fitted over 237 large random shapes (tests/fuzz/fuzz_shapes.py) as
// proxy^2.1 M^1.2 (N^0.2 beta^0.3): equal error means a limit ~ M^-0.6.  (M = 4096: 79 -- config 5, proxy 27, keeps the
// Gram form; the fuzz's largest error, 7e-10 against the oracle at M = 4216 and proxy 86, now takes the stable one.)
inline double pivot_ratio_limit(const Ctx* c, int M) {
    return (M > 2048) ? c->pivot_ratio_max * std::pow(2048.0 / (double)M, 0.6) : c->pivot_ratio_max;
}

// jitchol's jitter ladder: level -1 = none, level k = mean(diag) * 1e-6 * 10^k
inline double jit_amount(int level, double diag_mean_) {
    if (level < 0) return 0.0;
    return diag_mean_ * 1e-6 * std::pow(10.0, (double)level);
}

struct SparseOut {
    double lml = 0.0;
    std::vector<double> dvar, dlen, dZ;
    double dnoise = 0.0;
};

// One evaluation of the Titsias bound (VarDTC.inference) with optional gradients / posterior
// (sparse_eval.hip).  EDRGP_ERR_NOT_PD when a factorisation fails at every jitter level.
int sparse_eval(Ctx* c, bool need_grad, bool need_post, SparseOut* out);
int posterior_from_last_gradient(Ctx* c);

// [Phi | b] summed over the ranks: the lower triangle and b travel (M (M + 1) / 2 + M values), the upper
// triangle is restored by mirroring (comm.hip)
int allreduce_phi(Ctx* c, double* Phi, int M);

// EP classifier (ep_eval.hip)
int ep_begin(Ctx* c, const double* y_host);
int ep_sweep(Ctx* c, const int* order, double eta, double delta, double* tau_diff, double* v_diff);
int ep_finish(Ctx* c, double* log_Z_tilde);
int ep_sites(Ctx* c, double* tau, double* v, const double* tau_in, const double* v_in, double logZt_in);
int ep_eval(Ctx* c, bool need_grad, bool need_post, double* lml, std::vector<double>* dvar, std::vector<double>* dlen);
// sparse EP classifier (EPDTC, ep_eval.hip); ep_finish / ep_sites serve both
int epdtc_begin(Ctx* c, const double* y_host);
int epdtc_sweep(Ctx* c, const int* order, double eta, double delta, double* tau_diff, double* v_diff);
int epdtc_eval(Ctx* c, bool need_grad, bool need_post, double* lml, std::vector<double>* dvar, std::vector<double>* dlen,
               std::vector<double>* dZ);

}  // namespace edrgp
