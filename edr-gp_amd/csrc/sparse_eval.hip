// One evaluation of the Titsias bound -- GPy VarDTC.inference as reached from
// edrgp/gp_model/base.py:69 (fit: one inference per optimiser iteration) and :187 / :222 (posterior for
// predict / predictive_gradients) -- in the row-statistics form of DESIGN.md section 3, with optional gradients
// (SparseGP._update_gradients) and posterior (woodbury vector / inverse).
//
// The evaluation is a sequence of phases, each a function of SparseEval below:
//   phase 1   Kmm + (1e-8 + jitter) I, Lm = chol, Lm^-1                      O(M^3), independent of the rows
//   phase 2   row statistics [Phi | b] (Gram or stable form), one all-reduce  O(N M^2)
//   phase 3   A, B = I + A, LB = chol, (LB^-1), the M-vectors and scalars     O(M^3)
//   posterior / gradients                                                      O(M^3) + O(N M^2)
// Two drivers run them:
//   * the SYNC-FREE driver (default) queues everything for the mode it assumes (the form the previous evaluation
//     ended in) and reads the device's verdict -- Cholesky info words, pivot-ratio proxy of cond(Kmm) -- together
//     with the results at the single synchronisation at the end.  If the verdict disagrees (a factorisation needs
//     jitter, or the conditioning asks for the other form) the evaluation is repeated by
//   * the CAREFUL driver, which synchronises after phase 1 and phase 3 and climbs GPy's jitchol ladder on the host.
// Either way the returned numbers are those of the form the pivot-ratio rule selects for THIS Kmm: results do not
// depend on the history of calls.  Every decision that changes the number or content of collectives (the form,
// the jitter level, a repeat) derives from replicated quantities (Kmm, the all-reduced Phi), never from a rank's
// local row count: all ranks of a communicator issue the same sequence of all-reduces.
#include "eval_common.h"
#include <chrono>
#include <cstring>

namespace edrgp {

int g_tri_tile = 64;

int ensure_mats(Ctx* c, int n, int count) {
    for (int i = 0; i < count; ++i) {
        size_t need = (size_t)n * n * sizeof(double);
        if (i == MAT_PHI) need += ((size_t)n + 8) * sizeof(double);
        TRY(c->mats[i].reserve(need));
    }
    TRY(c->scal.reserve(SC_COUNT * sizeof(double)));
    TRY(c->info.reserve(4 * sizeof(int)));
    for (int i = 0; i < 6; ++i) TRY(c->tmpvec[i].reserve((size_t)(n + 8) * sizeof(double)));
    return EDRGP_OK;
}

// The stable form's product W = K Li^T (rows x M; Li lower triangular: column tile j has a k-range of j + 1 tiles) with
// b = W^T y out of the same launch's epilogue (per-row-tile partials, summed in a fixed order).  Shared by the
// evaluation and the batched sweep.  (Measured: it runs at the Phi kernel's rate for this M once the tile-granular
// k-ranges are counted -- 1.125 N M^2 flops at M = 1024, 59 vs 61 TFLOP/s; giving the rows of the last, partly
// filled round of its persistent grid to small tiles instead changed nothing, 4.55 -> 4.52 ms at C2.)
int stable_product(Ctx* c, const double* K, int64_t rows, int M, const double* Li, const double* y, double* W, double* b) {
    if (rows < 128 || M < 128) {
        TRY(gemm(c, false, true, (int)rows, M, M, 1.0, K, M, Li, M, 0.0, W, M, kopt(K_LE_COL, 0, 2)));
        return gemv_t(c, (int)rows, M, 1.0, W, M, y, b);
    }
    const int tiles_m = (int)((rows + 127) / 128);
    TRY(c->bpart.reserve((size_t)tiles_m * M * sizeof(double)));
    FuseK fk{};
    fk.y = y;
    fk.bpart = c->bpart.d();
    GemmOpts ow = kopt(K_LE_COL, 0, 4);
    ow.force_tile = 128;
    ow.fusek = &fk;
    TRY(gemm(c, false, true, (int)rows, M, M, 1.0, K, M, Li, M, 0.0, W, M, ow));
    return col_reduce(c, c->bpart.d(), tiles_m, M, b);
}

namespace {

// globals over ranks (N, yy) -- data-dependent only, cached
struct Globals { double N, yy; };
int get_globals(Ctx* c, Globals* g) {
    g->N = (double)c->N;
    g->yy = c->yy_local;
    if (c->sharded() && c->glob_valid) {        // all-reduced once per data set (every rank takes this branch together)
        g->N = c->glob_N;
        g->yy = c->glob_yy;
        return EDRGP_OK;
    }
    if (c->sharded()) {
        TRY(c->red.reserve(2 * sizeof(double)));
        double h[2] = {g->N, g->yy};
        HIP_TRY(hipMemcpyAsync(c->red.d(), h, sizeof(h), hipMemcpyHostToDevice, c->stream));
        TRY(allreduce_sum(c, c->red.d(), 2));
        HIP_TRY(hipMemcpyAsync(h, c->red.d(), sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        g->N = h[0];
        g->yy = h[1];
        c->glob_N = h[0];
        c->glob_yy = h[1];
        c->glob_valid = true;
    }
    return EDRGP_OK;
}

struct HostTrace {          // option "trace": host wall-clock of the phases of one evaluation, to stderr
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    explicit HostTrace(bool on_) : on(on_) { t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char* what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[edrgp trace] %-28s +%8.3f ms (total %8.3f)\n", what,
                std::chrono::duration<double, std::milli>(now - last).count(),
                std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

constexpr int RETRY_CAREFUL = 1;      // sync-free driver: the device's verdict asks for a careful repeat
constexpr int RETRY_JITTER = 2;       // ... because a factorisation needs jitter

struct SparseEval {
    Ctx* c;
    const bool need_grad, need_post;
    SparseOut* out;
    HostTrace tr;

    // sizes.  The matrix kernels run their fast paths on whole 128-wide tiles and whole k-tiles only (M = 1000 ran
    // 1.4x slower than M = 1024).  An inducing set whose size is not a multiple of 128 is therefore embedded in the
    // next multiple: Mr real points plus M - Mr phantom points that are decoupled from everything -- K_nm gets zero
    // columns, Kmm the block [[Kmm, 0], [0, I]].  Then Lm, Lm^-1, B, LB, LB^-1, Kmm^-1, P are block diagonal with an
    // identity block, Phi / A / S2 / dL/dKmm have zero phantom blocks, and the bound and every gradient of the real
    // parameters are those of the Mr-point problem (the log-determinants gain log 1, the traces 0).  Only the
    // builds know about Mr; everything between them works on M.
    int D = 0, Wd = 0, np = 0, Mr = 0, M = 0;
    bool padded = false;
    int64_t N = 0, chunk = 0, nchunks = 0;
    double beta = 0.0, kmm_diag_mean = 0.0;
    KSpec ks;
    Globals gl;

    double *Lm = nullptr, *Li = nullptr, *Phi = nullptr, *bvec = nullptr, *Tm = nullptr, *A = nullptr, *LB = nullptr,
           *LBi = nullptr, *sc = nullptr;
    int* info = nullptr;
    double *v_v = nullptr, *v_c = nullptr, *v_u = nullptr, *v_wv = nullptr, *v_tmp = nullptr;

    bool k_resident = false;      // c->Knm holds K(X, Z) of all local rows
    bool prebuilt = false;        // K_nm (and b) of the single row chunk were built beside phase 1
    bool kmmi_done = false;       // Kmm^-1 = Li^T Li was formed on the side stream
    bool kmmi_pending = false;    // ... and the main stream has not waited for it yet
    bool fin_dev = false;         // dZ / d lengthscale finished on the device (grad_finish)
    int fin_blocks = 0;
    bool h_built = false;         // the K_nm chunk in c->Knm has its h(r) companion in c->Hnm
    bool tma_done = false;        // Lm^-T A Lm^-1 (Kmm term) was formed on the side stream beside the factorisation of B
    bool kz_done = false;         // K(Z, Z) and h(r) for the element-wise Kmm term are in MAT_KZ / MAT_HZ
    bool kmm_chain_queued = false;
    bool k32_resident = false;
    // precision 32 asks for the fp32 row pass; it is used while the Gram form is safe in fp32 (error ~ eps32
    // cond(Kmm), judged by the pivot-ratio proxy after phase 1), otherwise this evaluation falls back to the fp64
    // row pass (last_mode: 1 gram fp64, 2 stable fp64, 3 gram fp32)
    //   Phi in fp32 (f32): opt-in ("f32_phi" = 1; the bound is then only 1e-4..1e-3 accurate): attempted, kept only
    //     if the a-posteriori guard accepts it; a rejected attempt is remembered (phi32_bad) until the data, Z or
    //     the precision change.  Default: Phi always from the fp64 row pass, the bound is exact.
    //   gradient contraction in fp32 (g32): its rounding reaches the gradients only, never the positive
    //     definiteness of I + A.
    bool want32 = false, g32 = false, f32 = false;
    bool rhs_row = false;         // bound only: beta v rides through the second factorisation, no explicit LB^-1
    bool stable = false;          // form of the row statistics this evaluation uses
    double h_sc[SC_COUNT];
    int h_info[4];

    SparseEval(Ctx* c_, bool g, bool p, SparseOut* o) : c(c_), need_grad(g), need_post(p), out(o), tr(c_->trace) {
        std::memset(h_sc, 0, sizeof(h_sc));
        std::memset(h_info, 0, sizeof(h_info));
    }

    // ------------------------------------------------------------------ set-up
    int init() {
        if (!c->have_data || !c->have_Z || !c->have_kernel) {
            set_error("sparse evaluation needs set_data, set_inducing and set_kernel first");
            return EDRGP_ERR_STATE;
        }
        D = c->D;
        N = c->N;
        Wd = 2 * D + 1;
        np = c->nparts;
        c->grad_serial = 0;                 // the workspaces are about to be overwritten
        Mr = c->M;
        padded = c->pad_inducing && Mr > 256 && (Mr % 128) != 0 && c->precision != 32;
        M = padded ? (Mr + 127) / 128 * 128 : Mr;
        c->M_work = M;
        c->knm_train_valid = false;
        c->hnm_train_valid = false;
        beta = 1.0 / std::max(c->noise_var, 1e-8);
        ks = make_kspec(c);
        TRY(get_globals(c, &gl));
        TRY(ensure_mats(c, M, need_grad ? 16 : (need_post ? 10 : 7)));
        chunk = std::min<int64_t>(N, std::max<int64_t>(64, c->chunk_rows_for(M)));
        nchunks = (N + chunk - 1) / chunk;
        // 16 spare rows: the row count handed to the Phi kernel is rounded up to whole k-tiles with zero rows
        TRY(c->Knm.reserve((size_t)(chunk + 16) * M * sizeof(double)));
        Lm = c->mats[MAT_LM].d();
        Li = c->mats[MAT_LI].d();
        Phi = c->mats[MAT_PHI].d();
        bvec = Phi + (size_t)M * M;
        Tm = c->mats[MAT_T].d();
        A = c->mats[MAT_A].d();
        LB = c->mats[MAT_LB].d();
        LBi = c->mats[MAT_LBI].d();
        sc = c->scal.d();
        info = static_cast<int*>(c->info.p);
        v_v = c->tmpvec[0].d();
        v_c = c->tmpvec[1].d();
        v_u = c->tmpvec[2].d();
        v_wv = c->tmpvec[3].d();
        v_tmp = c->tmpvec[4].d();
        kmm_diag_mean = var_sum(c) + 1e-8;
        want32 = (c->precision == 32) && (np == 1);
        g32 = want32 && need_grad && D <= 64;      // and a Kmm conditioned for the Gram form, decided after phase 1
        f32 = want32 && c->f32_phi && !c->phi32_bad;
        // the fused one-launch-per-step factorisation covers n in (64, 3072]; outside it the explicit inverse stays
        rhs_row = !need_grad && !need_post && c->chol_rhs && c->chol_fused && M > 64 && M + 1 <= 3072;
        k_resident = prebuilt = kmmi_done = kmmi_pending = tma_done = k32_resident = h_built = kz_done = kmm_chain_queued = false;
        return EDRGP_OK;
    }

    // device scalars and info words -> host (through pinned staging: asynchronous copies); the one place an
    // evaluation waits for the device
    int fetch() {
        static_assert(SC_COUNT + 4 <= 64, "scalar block of the pinned staging");
        TRY(c->ensure_pinned(64));
        HIP_TRY(hipMemcpyAsync(c->h_pin, sc, sizeof(h_sc), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(c->h_pin + SC_COUNT, info, sizeof(h_info), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::memcpy(h_sc, c->h_pin, sizeof(h_sc));
        std::memcpy(h_info, c->h_pin + SC_COUNT, sizeof(h_info));
        return EDRGP_OK;
    }

    // ------------------------------------------------------------------ phase 1
    // Kmm + (1e-8 + jitter) I, Lm (info word 0), the pivot extremes of Lm, Lm^-1 -- queued, not waited for
    int phase1_queue(int lvl) {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        if (padded) HIP_TRY(hipMemsetAsync(Lm, 0, (size_t)M * M * sizeof(double), c->stream));
        TRY(kernel_matrix(c, ks, c->Z.d(), Mr, c->Z.d(), Mr, Lm, M, 1, 1e-8 + jit_amount(lvl, kmm_diag_mean), nullptr,
                          nullptr));
        if (padded) TRY(add_diag(c, Lm + (size_t)Mr * M + Mr, M - Mr, M, 1.0));
        {
            ProfScope pc(c, EDRGP_PROF_CHOL);
            TRY(potrf_inv_lower(c, Lm, M, M, info + 0, Li, M));
        }
        TRY(diag_minmax(c, Lm, Mr, M, sc + SC_DIAGMIN));
        // gradient evaluations: K(Z, Z) and h(r) for the element-wise Kmm term, behind the factorisation (phase 1 ends
        // before the K_nm build it runs beside; beside the factorisation of B the same launch cost that chain 0.25 ms,
        // in the tail of the evaluation it took 420 us beside T^T [X | X.^2])
        if (need_grad && np == 1 && c->mfma_kbuild && !kz_done) TRY(build_kz());
        return EDRGP_OK;
    }
    // careful driver: GPy's jitchol ladder, one synchronisation per attempt
    int phase1_careful() {
        for (int lvl = -1;;) {
            TRY(phase1_queue(lvl));
            TRY(fetch());
            if (h_info[0] == 0) return EDRGP_OK;
            if (++lvl > 4) { set_error("Kmm not positive definite, even with jitter"); return EDRGP_ERR_NOT_PD; }
        }
    }
    int phase1(bool careful) { return careful ? phase1_careful() : phase1_queue(-1); }

    // ------------------------------------------------------------------ phase 2
    // K(X[r0 : r0 + rows], Z) into `outK` (ld M) and, when yv is given, K^T yv into bt; phantom columns are zero
    // Non-RBF single-part kernels in a gradient evaluation: the build stores h(r) = (dk/dr)/r beside k, and the gradient
    // contraction reads both instead of recomputing distances, sqrt and exp per element in its epilogue -- fp64 VALU
    // work crawls beside the partner workgroup's MFMA stream (Matern-5/2, D = 32: 108 us of epilogue per 470 us tile,
    // tools/tile_stamps.py).  One more N x M array (1.6 GB at Cm, written once per evaluation).
    bool want_h() const {
        // (posterior evaluations store it too: predict_gradient on the training rows -- what the EDR loop asks next,
        // edrgp/base.py:274 -- then needs no second pass over the pairs)
        return (need_grad || need_post) && np == 1 && c->kind[0] != EDRGP_KERN_RBF && c->fuse_grad && c->h_resident &&
               c->mfma_kbuild && !g32 && c->rbf_epilogue;
    }
    int build_knm(int64_t r0, int64_t rows, double* outK, const double* yv, double* bt) {
        const bool with_h = want_h() && outK == c->Knm.d();
        if (with_h) {
            TRY(c->Hnm.reserve((size_t)chunk * M * sizeof(double)));
            c->kbuild_h = c->Hnm.d();
        }
        const int st = kernel_matrix(c, ks, c->X.d() + (size_t)r0 * D, rows, c->Z.d(), Mr, outK, M, 0, 0.0, yv, bt);
        c->kbuild_h = nullptr;
        if (st != EDRGP_OK) return st;
        h_built = with_h;
        if (padded && with_h)
            HIP_TRY(hipMemset2DAsync(c->Hnm.d() + Mr, (size_t)M * sizeof(double), 0, (size_t)(M - Mr) * sizeof(double), (size_t)rows,
                                     c->stream));
        if (padded) {
            HIP_TRY(hipMemset2DAsync(outK + Mr, (size_t)M * sizeof(double), 0, (size_t)(M - Mr) * sizeof(double), (size_t)rows,
                                     c->stream));
            if (bt != nullptr) HIP_TRY(hipMemsetAsync(bt + Mr, 0, (size_t)(M - Mr) * sizeof(double), c->stream));
        }
        return EDRGP_OK;
    }

    // row statistics.  gram: Phi = K^T K, b = K^T y.
    //                  stable: Phi := (K Li^T)^T (K Li^T), b := (K Li^T)^T y  (= Li b)
    // (Gram form: A = beta Li (K^T K) Li^T, rounding error ~ eps cond(Kmm);
    //  stable form: A = beta (K Li^T)^T (K Li^T), error ~ eps sqrt(cond).)
    // Exactly one all-reduce of [Phi | b] per call, whatever the local chunking.
    int phase2(bool stable_form) {
        tma_done = false;                          // A follows Phi
        if (f32) {
            // fp32 row pass: K_nm stored and multiplied in fp32, Phi / b accumulated into fp64
            TRY(c->Knm32.reserve((size_t)chunk * M * sizeof(float)));
            float* K32 = static_cast<float*>(c->Knm32.p);
            for (int64_t ch = 0; ch < nchunks; ++ch) {
                const int64_t r0 = ch * chunk;
                const int64_t rows = std::min<int64_t>(chunk, N - r0);
                double* btarget = (ch == 0) ? bvec : v_tmp;
                {
                    ProfScope ps(c, EDRGP_PROF_KBUILD);
                    TRY(kernel_matrix32(c, ks, c->X.d() + (size_t)r0 * D, rows, c->Z.d(), M, K32, c->y.d() + r0, btarget));
                }
                if (ch > 0) TRY(vec_axpy(c, M, 1.0, v_tmp, bvec));
                TRY(syrk32_tn(c, K32, rows, M, M, Phi, M, ch > 0));
            }
            k32_resident = (nchunks == 1);
            TRY(allreduce_phi(c, Phi, M));
            return EDRGP_OK;
        }
        if (stable_form) TRY(c->W.reserve((size_t)(chunk + 16) * M * sizeof(double)));
        for (int64_t ch = 0; ch < nchunks; ++ch) {
            const int64_t r0 = ch * chunk;
            const int64_t rows = std::min<int64_t>(chunk, N - r0);
            double* btarget = (ch == 0) ? bvec : v_tmp;
            if (!(prebuilt && ch == 0)) {
                ProfScope ps(c, EDRGP_PROF_KBUILD);
                TRY(build_knm(r0, rows, c->Knm.d(), stable_form ? nullptr : c->y.d() + r0, stable_form ? nullptr : btarget));
            }
            const double* S = c->Knm.d();
            if (stable_form) {
                ProfScope ps(c, EDRGP_PROF_GRADGEMM);
                // W = K_nm Li^T with b = W^T y out of the same pass (per-row-tile partials, fixed order)
                TRY(stable_product(c, c->Knm.d(), rows, M, Li, c->y.d() + r0, c->W.d(), btarget));
                S = c->W.d();
            }
            if (ch > 0) TRY(vec_axpy(c, M, 1.0, v_tmp, bvec));
            // whole k-tiles for the Phi kernel (its 16-byte loaders and the paired diagonal tiles need them):
            // zero rows add nothing to K^T K
            const int64_t rows_pad = (rows + 15) & ~(int64_t)15;
            if (rows_pad > rows)
                HIP_TRY(hipMemsetAsync(const_cast<double*>(S) + (size_t)rows * M, 0, (size_t)(rows_pad - rows) * M * sizeof(double),
                                       c->stream));
            TRY(syrk_tn(c, S, rows_pad, M, M, Phi, M, ch > 0));
        }
        k_resident = (nchunks == 1);
        // predictions address K_nm with ld = number of real points (the stable form keeps K_nm in Knm, K_nm Li^T in W)
        c->knm_train_valid = k_resident && !padded;
        c->hnm_train_valid = c->knm_train_valid && h_built;
        // sharded: what does not depend on the summed Phi (Kmm^-1 for the posterior / gradients) is released on the side
        // stream NOW, so that it runs while the packed triangle is on the fabric; the sequence of collectives is the same
        if (c->sharded()) TRY(kmmi_beside_phase3());
        TRY(allreduce_phi(c, Phi, M));
        return EDRGP_OK;
    }

    // ------------------------------------------------------------------ phase 3
    // A, B = I + A (+ jitter at ladder level lvl, which needs trA of the failed attempt in h_sc), LB (info word 1),
    // LB^-1, the M-vectors and the scalars of the bound -- queued, not waited for
    int phase3_queue(int lvl) {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        // every slot but the pivot extremes of Lm, which phase 1 wrote and the verdict still needs
        HIP_TRY(hipMemsetAsync(sc, 0, SC_DIAGMIN * sizeof(double), c->stream));
        HIP_TRY(hipMemsetAsync(sc + SC_LBDIAGMIN, 0, (SC_COUNT - SC_LBDIAGMIN) * sizeof(double), c->stream));
        if (stable) {
            TRY(combine3(c, M, beta, Phi, 0.0, nullptr, 0.0, nullptr, 0.0, nullptr, A));
            HIP_TRY(hipMemcpyAsync(v_v, bvec, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        } else {
            // Tm = Li Phi: A's lower tiles (i, j) read Tm[i][k] for k <= j <= i only, so the lower tiles of Tm suffice
            TRY(gemm(c, false, false, M, M, M, 1.0, Li, M, Phi, M, 0.0, Tm, M, kopt(K_LE_ROW, 1)));
            // A = beta (Li Phi) Li^T is symmetric: lower tiles only, then mirrored
            GemmOpts oa = kopt(K_LE_COL, 1);
            oa.mirror_out = 1;
            TRY(gemm(c, false, true, M, M, M, beta, Tm, M, Li, M, 0.0, A, M, oa));
            TRY(gemv_n(c, M, M, 1.0, Li, M, bvec, v_v));
        }
        TRY(trace_of(c, A, M, M, sc + SC_TRA));
        TRY(kmm_products_beside_chol());
        const double bjit = 1.0 + (lvl >= 0 ? jit_amount(lvl, 1.0 + h_sc[SC_TRA] / Mr) : 0.0);
        if (rhs_row) {
            // bound only: the bound needs |c|^2, c = beta LB^-1 v, and log det LB, not LB^-1.  beta v rides
            // through the factorisation as row M of the (M+1) x (M+1) matrix [[B, .], [beta v^T, alpha]]: the
            // panel steps turn it into c^T (forward substitution by the trailing updates), one extra step
            // instead of the explicit inverse and its products
            const int ldb = M + 8;
            TRY(c->lbaug.reserve((size_t)(M + 1) * ldb * sizeof(double)));
            double* Lg = c->lbaug.d();
            // one launch each for the set-up (copy, jitter, right-hand-side row) and for log det / |c|^2, and the factor
            // is read where the factorisation assembled it (no copy back)
            TRY(aug_setup(c, Lg, ldb, A, M, bjit, v_v, beta));
            const double* Lf = nullptr;
            int ldf = 0;
            TRY(potrf_lower_ws(c, Lg, M + 1, ldb, info + 1, &Lf, &ldf));
            TRY(aug_tail(c, Lf, ldf, M, sc + SC_SUMLOGLB, sc + SC_DATAFIT));
            if (f32) {
                TRY(diag_minmax(c, Lf, M, ldf, sc + SC_LBDIAGMIN));
                TRY(frob_sq(c, Phi, M, M, sc + SC_PHIFROB));
            }
            return EDRGP_OK;
        }
        HIP_TRY(hipMemcpyAsync(LB, A, (size_t)M * M * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        // jitter on B: mean(diag(B)) = 1 + trA / M, trA known from the failed pass
        TRY(add_diag(c, LB, M, M, bjit));
        {
            ProfScope pc(c, EDRGP_PROF_CHOL);
            TRY(potrf_inv_lower(c, LB, M, M, info + 1, LBi, M));
        }
        TRY(sum_log_diag(c, LB, M, M, sc + SC_SUMLOGLB));
        // c = beta LBi v ; u = LBi^T c ; wv = Li^T u     (v = Lm^-1 K_mn y)
        TRY(gemv_n(c, M, M, beta, LBi, M, v_v, v_c));
        TRY(gemv_t(c, M, M, 1.0, LBi, M, v_c, v_u));
        TRY(gemv_t(c, M, M, 1.0, Li, M, v_u, v_wv));
        TRY(sum_sq(c, v_c, M, sc + SC_DATAFIT));
        TRY(sum_sq(c, v_u, M, sc + SC_USQ));
        TRY(frob_sq(c, LBi, M, M, sc + SC_LBIFROB));
        if (f32) {
            TRY(diag_minmax(c, LB, M, M, sc + SC_LBDIAGMIN));
            TRY(frob_sq(c, Phi, M, M, sc + SC_PHIFROB));
        }
        return EDRGP_OK;
    }
    // careful driver: jitchol ladder on B.  no_jitter: EDRGP_ERR_NOT_PD at the first failed factorisation
    int phase3_careful(bool no_jitter) {
        for (int lvl = -1;;) {
            TRY(phase3_queue(lvl));
            TRY(fetch());
            if (h_info[1] == 0) return EDRGP_OK;
            if (no_jitter) { set_error("I + A not positive definite"); return EDRGP_ERR_NOT_PD; }
            if (++lvl > 4) { set_error("I + A not positive definite, even with jitter"); return EDRGP_ERR_NOT_PD; }
        }
    }

    // the pivot-ratio rule on the fetched scalars: does THIS Kmm ask for the stable form?
    double pivot_ratio2() const {
        const double pr = h_sc[SC_DIAGMAX] / h_sc[SC_DIAGMIN];
        return pr * pr;
    }
    bool rule_says_stable() const {
        return (c->stats_mode == 2) || (c->stats_mode == 0 && pivot_ratio2() > pivot_ratio_limit(c, Mr));
    }

    void set_bound() {
        const double psi0sum = gl.N * var_sum(c);
        out->lml = -0.5 * gl.N * (std::log(2.0 * M_PI) - std::log(beta)) - 0.5 * beta * gl.yy -
                   0.5 * (beta * psi0sum - h_sc[SC_TRA]) - h_sc[SC_SUMLOGLB] + 0.5 * h_sc[SC_DATAFIT];
    }

    // ------------------------------------------------------------------ posterior / gradient pieces
    // Kmm^-1 = Li^T Li is needed by the posterior / gradients only and depends on phase 1 alone: it runs on the side
    // stream beside phase 3, whose factorisation is a chain of latency-bound steps that leave the chip almost idle
    // (it used to sit in front of the Phi kernel: 0.21 ms on the critical path at M = 2048)
    int kmmi_beside_phase3() {
        if (!(need_grad || need_post) || !c->overlap || kmmi_done) return EDRGP_OK;
        SideStream side(c);
        TRY(side.start_after_main());              // after the row pass: the Phi kernel keeps the GPU to itself
        {
            ProfScope ps(c, EDRGP_PROF_SMALL);
            c->gemm_pad_now = c->side_gemm_pad;
            const int st = gram_of_lower(c, M, Li, c->mats[MAT_KMMI].d());
            c->gemm_pad_now = 0;
            if (st != EDRGP_OK) return st;
        }
        TRY(side.mark_done());
        kmmi_done = kmmi_pending = true;
        return EDRGP_OK;
    }

    // K(Z, Z) without jitter and its h(r) for the element-wise Kmm term (35 us alone; queued at the end of phase 1)
    int build_kz() {
        c->kbuild_h = c->mats[MAT_HZ].d();
        const int st = kernel_matrix(c, ks, c->Z.d(), Mr, c->Z.d(), Mr, c->mats[MAT_KZ].d(), M, 1, 0.0, nullptr, nullptr);
        c->kbuild_h = nullptr;
        if (st == EDRGP_OK) kz_done = true;
        return st;
    }

    // The two M^3 products of the Kmm term that need A but not LB (C1 = Li^T A, Lm^-T A Lm^-1 = C1 Li) join Kmm^-1 on
    // the side stream beside the factorisation of B.  They used to run beside the gradient contraction, which as a
    // persistent grid leaves no slot to a second kernel: the whole Kmm chain then ran after it (1.2 ms at Cm).
    int kmm_products_beside_chol() {
        if (!need_grad || !c->overlap || !c->kmm_early || tma_done) return EDRGP_OK;
        SideStream side(c);
        TRY(side.start_after_main());
        {
            ProfScope ps(c, EDRGP_PROF_SMALL);
            double* C1 = c->mats[MAT_C1].d();
            // padded like the K_nm build beside phase 1: the LDS hole a finished workgroup leaves must take a
            // factorisation step's workgroup
            c->gemm_pad_now = c->side_gemm_pad;
            int st = gemm(c, true, false, M, M, M, 1.0, Li, M, A, M, 0.0, C1, M, kopt(K_GE_ROW));
            if (st == EDRGP_OK) st = gemm(c, false, false, M, M, M, 1.0, C1, M, Li, M, 0.0, c->mats[MAT_TMM].d(), M, kopt(K_GE_COL));
            c->gemm_pad_now = 0;
            if (st != EDRGP_OK) return st;
        }
        TRY(side.mark_done());                     // stream order: also covers Kmm^-1 queued before
        tma_done = true;
        kmmi_pending = true;                       // factors_queue waits for the side stream
        return EDRGP_OK;
    }

    // R = LBi Li, P = R^T R, Kmm^-1 = Li^T Li
    int factors_queue() {
        if (kmmi_pending) {
            TRY(main_waits_side(c));
            kmmi_pending = false;
        }
        ProfScope ps(c, EDRGP_PROF_SMALL);
        double* R = c->mats[MAT_R].d();
        // R is lower triangular: lower tiles only, upper part stays zero
        HIP_TRY(hipMemsetAsync(R, 0, (size_t)M * M * sizeof(double), c->stream));
        TRY(gemm(c, false, false, M, M, M, 1.0, LBi, M, Li, M, 0.0, R, M, kopt(K_LE_ROW | K_GE_COL, 1)));
        TRY(gram_of_lower(c, M, R, c->mats[MAT_P].d()));
        if (!kmmi_done) TRY(gram_of_lower(c, M, Li, c->mats[MAT_KMMI].d()));
        return EDRGP_OK;
    }
    // the posterior is kept compact: Mr entries of wv, the Mr x Mr block of Kmm^-1 - P
    int posterior_queue() {
        const double* Kmmi = c->mats[MAT_KMMI].d();
        const double* P = c->mats[MAT_P].d();
        TRY(c->wv.reserve((size_t)Mr * sizeof(double)));
        TRY(c->winv.reserve((size_t)Mr * Mr * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(c->wv.d(), v_wv, (size_t)Mr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        if (padded) {
            TRY(combine3(c, M, 1.0, Kmmi, -1.0, P, 0.0, nullptr, 0.0, nullptr, Tm));        // Tm is free here
            HIP_TRY(hipMemcpy2DAsync(c->winv.d(), (size_t)Mr * sizeof(double), Tm, (size_t)M * sizeof(double),
                                     (size_t)Mr * sizeof(double), (size_t)Mr, hipMemcpyDeviceToDevice, c->stream));
        } else {
            TRY(combine3(c, M, 1.0, Kmmi, -1.0, P, 0.0, nullptr, 0.0, nullptr, c->winv.d()));
        }
        return EDRGP_OK;
    }

    // Kmm term (X2 = None upstream): dL/dKmm, its element-wise factor, then T^T Zt and T Zt.  O(M^3) and
    // independent of the rows: runs on the side stream beside the N x M x M gradient contraction.
    int kmm_term_queue() {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        const double* Kmmi = c->mats[MAT_KMMI].d();
        const double* P = c->mats[MAT_P].d();
        double* C1 = c->mats[MAT_C1].d();
        double* dKmm = c->mats[MAT_DKMM].d();
        double* Tmm = c->mats[MAT_TMM].d();
        // C1 = Li^T A ; Tm = C1 Li = Lm^-T A Lm^-1 (already in Tmm when formed beside the factorisation of B)
        const double* LtAL = Tm;
        if (tma_done) {
            LtAL = Tmm;
        } else {
            TRY(gemm(c, true, false, M, M, M, 1.0, Li, M, A, M, 0.0, C1, M, kopt(K_GE_ROW)));
            TRY(gemm(c, false, false, M, M, M, 1.0, C1, M, Li, M, 0.0, Tm, M, kopt(K_GE_COL)));
        }
        // dL/dKmm = 1/2 Kmm^-1 - 1/2 P - 1/2 wv wv^T - 1/2 Lm^-T A Lm^-1
        TRY(combine3(c, M, 0.5, Kmmi, -0.5, P, -0.5, LtAL, -0.5, v_wv, dKmm));
        for (int p = 0; p < np; ++p) {
            if (np == 1 && c->mfma_kbuild) {
                // k and h(r) of K(Z, Z) from the MFMA build, then one element-wise pass
                if (!kz_done) TRY(build_kz());
                TRY(kmm_term_elementwise(c, dKmm, c->mats[MAT_KZ].d(), c->mats[MAT_HZ].d(), Mr, M, c->var[0], Tmm,
                                         sc + SC_DVAR_MM));
            } else
            TRY(grad_elementwise(c, ks, p, c->Z.d(), nullptr, M, c->Z.d(), M, dKmm, M, 0.0, nullptr, Tmm, M,
                                 sc + SC_DVAR_MM + p));
            TRY(gemm(c, true, false, M, Wd, M, 1.0, Tmm, M, c->Zt.d(), Wd, 0.0,
                     c->gt_mm.d() + (size_t)(2 * p) * M * Wd, Wd));
            TRY(gemm(c, false, false, M, Wd, M, 1.0, Tmm, M, c->Zt.d(), Wd, 0.0,
                     c->gt_mm.d() + (size_t)(2 * p + 1) * M * Wd, Wd));
        }
        return EDRGP_OK;
    }

    // rows of the leading row tiles that fill whole rounds of the persistent contraction grid (2 workgroups per CU);
    // `rows` when there is nothing to split off
    int64_t contraction_whole_rounds(int64_t rows) const {
        if (!c->overlap || !c->gemm_persist || M < 128) return rows;
        const int64_t tn = (M + 127) / 128, tm = (rows + 127) / 128, grid = 2 * (int64_t)c->num_cu;
        const int64_t rounds = tm * tn / grid;
        const int64_t tmA = rounds * grid / tn;
        if (rounds < 2 || tmA <= 0 || tmA >= tm) return rows;
        return tmA * 128;
    }

    // The Kmm-term chain on the side stream, released by whatever the main stream has queued so far
    int queue_kmm_chain() {
        if (!c->overlap || kmm_chain_queued) return EDRGP_OK;
        SideStream side(c);
        TRY(side.start_after_main());
        TRY(kmm_term_queue());
        TRY(side.mark_done());
        kmm_chain_queued = true;
        return EDRGP_OK;
    }

    // fp32 gradient row pass: T = (K32 S2 + beta y wv^T) .* h(r) out of the contraction's epilogue,
    // then T^T [X | X.^2]; partial sums and the M x 2D result are fp64
    int grad_rows32_queue(const double* S2) {
        TRY(c->Knm32.reserve((size_t)chunk * M * sizeof(float)));
        TRY(c->T32.reserve((size_t)chunk * M * sizeof(float)));
        TRY(c->S2f.reserve((size_t)M * M * sizeof(float)));
        TRY(c->g2.reserve(((size_t)M * 2 * D + M + 8) * sizeof(double)));
        if (!c->xq32_valid) {
            TRY(c->Xq32.reserve((size_t)N * 2 * D * sizeof(float)));
            TRY(build_xq32(c, c->X.d(), N, D, static_cast<float*>(c->Xq32.p)));
            c->xq32_valid = true;
        }
        float* K32 = static_cast<float*>(c->Knm32.p);
        float* T32 = static_cast<float*>(c->T32.p);
        float* S2f = static_cast<float*>(c->S2f.p);
        const float* Xq32 = static_cast<const float*>(c->Xq32.p);
        double* G2 = c->g2.d();
        double* csv = G2 + (size_t)M * 2 * D;
        TRY(to_f32(c, S2, (size_t)M * M, S2f));
        for (int64_t ch = 0; ch < nchunks; ++ch) {
            const int64_t r0 = ch * chunk;
            const int64_t rows = std::min<int64_t>(chunk, N - r0);
            if (!k32_resident) {
                ProfScope ps(c, EDRGP_PROF_KBUILD);
                TRY(kernel_matrix32(c, ks, c->X.d() + (size_t)r0 * D, rows, c->Z.d(), M, K32, nullptr, nullptr));
            }
            ProfScope ps(c, EDRGP_PROF_GRADGEMM);
            TRY(grad_contract32(c, ks, K32, rows, M, S2f, c->y.d() + r0, v_wv, beta, T32, sc + SC_DVAR_NM, csv));
            TRY(gemm32_tn_rows(c, M, 2 * D, rows, T32, M, Xq32 + (size_t)r0 * 2 * D, 2 * D, G2, 2 * D, false));
            TRY(assemble_gt(c, G2, csv, M, D, ch > 0 ? 1.0 : 0.0, c->gt_nm.d()));
        }
        return EDRGP_OK;
    }

    // fp64 gradient row pass: Q = K_nm S2 + beta y wv^T, T = Q .* h(r), Gt = T^T [X | 1 | X.^2]
    int grad_rows64_queue(const double* S2) {
        TRY(c->W.reserve((size_t)chunk * M * sizeof(double)));
        TRY(c->T.reserve((size_t)chunk * M * sizeof(double)));
        for (int64_t ch = 0; ch < nchunks; ++ch) {
            const int64_t r0 = ch * chunk;
            const int64_t rows = std::min<int64_t>(chunk, N - r0);
            if (!k_resident) {
                ProfScope ps(c, EDRGP_PROF_KBUILD);
                TRY(build_knm(r0, rows, c->Knm.d(), nullptr, nullptr));
            }
            // the distance-recomputing epilogue stages D <= 32 features; the RBF epilogue reads k from K_nm: any D
            const bool fused = (np == 1) && c->fuse_grad &&
                               (D <= 32 || (c->kind[0] == EDRGP_KERN_RBF && c->rbf_epilogue) || h_built);
            if (fused) {
                // T = (K_nm S2 + beta y wv^T) .* h(r) and the d/dvariance partials straight out of
                // the contraction's epilogue: no W round trip, no separate element-wise pass
                ProfScope ps(c, EDRGP_PROF_GRADGEMM);
                const int ntiles = (int)(((rows + 127) / 128) * ((M + 127) / 128));
                TRY(c->red.reserve((size_t)ntiles * sizeof(double)));
                FuseT f;
                f.X = c->X.d() + (size_t)r0 * D; f.Z = c->Z.d(); f.y = c->y.d() + r0; f.wv = v_wv;
                f.invl = c->invl.d(); f.dvpart = c->red.d(); f.ycoef = beta; f.var = c->var[0]; f.D = D; f.kind = c->kind[0];
                // the "ones" column of T^T [X | 1 | X.^2] comes out of the epilogue as column sums, so the
                // trailing product is T^T [X | X.^2]: 2D = 64 columns = one column tile instead of two
                const int tiles_m = (int)((rows + 127) / 128);
                TRY(c->bpart.reserve((size_t)tiles_m * M * sizeof(double)));
                TRY(c->g2.reserve(((size_t)M * 2 * D + M + 8) * sizeof(double)));
                f.cspart = c->bpart.d();
                // RBF: the epilogue reads k from the resident K_nm chunk instead of recomputing distances
                f.K = (c->kind[0] == EDRGP_KERN_RBF && c->rbf_epilogue) ? c->Knm.d() : nullptr;
                if (h_built) {                        // k and h(r) both resident: any D, no distances in the epilogue
                    f.K = c->Knm.d();
                    f.H = c->Hnm.d();
                }
                f.ldk = M;
                double* G2 = c->g2.d();
                double* csv = G2 + (size_t)M * 2 * D;
                GemmOpts o;
                o.tag = 2;
                o.force_tile = 128;
                o.fuse = &f;
                // The launch is a persistent grid of 2 workgroups per CU that holds every workgroup slot; its last,
                // partial round of tiles leaves half the CUs idle (Cm: 224 tiles, 0.46 ms).  That round is a launch
                // of its own -- the last row tiles -- and the Kmm-term chain (side stream, 0.2 ms) is released when
                // the whole rounds are done: it runs in the idle half beside the remainder instead of queueing for
                // slots through the whole contraction (its first kernel showed 12 ms in the kernel statistics).
                const int64_t rowsA = (ch == nchunks - 1) ? contraction_whole_rounds(rows) : rows;
                TRY(gemm(c, false, false, (int)rowsA, M, M, 1.0, c->Knm.d(), M, S2, M, 0.0, c->T.d(), M, o));
                if (ch == nchunks - 1) TRY(queue_kmm_chain());
                if (rowsA < rows) {
                    FuseT fb = f;
                    const size_t tA = (size_t)(rowsA / 128);
                    fb.X += (size_t)rowsA * D; fb.y += rowsA;
                    fb.dvpart += tA * ((M + 127) / 128);
                    fb.cspart += tA * M;
                    if (fb.K != nullptr) fb.K += (size_t)rowsA * M;
                    if (fb.H != nullptr) fb.H += (size_t)rowsA * M;
                    o.fuse = &fb;
                    TRY(gemm(c, false, false, (int)(rows - rowsA), M, M, 1.0, c->Knm.d() + (size_t)rowsA * M, M, S2, M, 0.0,
                             c->T.d() + (size_t)rowsA * M, M, o));
                }
                TRY(sum_into(c, c->red.d(), ntiles, sc + SC_DVAR_NM));
                TRY(col_reduce(c, c->bpart.d(), tiles_m, M, csv));
                TRY(gemm(c, true, false, M, 2 * D, (int)rows, 1.0, c->T.d(), M, c->Xq.d() + (size_t)r0 * 2 * D, 2 * D, 0.0, G2,
                         2 * D));
                TRY(assemble_gt(c, G2, csv, M, D, ch > 0 ? 1.0 : 0.0, c->gt_nm.d()));
                continue;
            }
            {
                ProfScope ps(c, EDRGP_PROF_GRADGEMM);
                GemmOpts o;
                o.tag = 2;
                TRY(gemm(c, false, false, (int)rows, M, M, 1.0, c->Knm.d(), M, S2, M, 0.0, c->W.d(), M, o));
            }
            for (int p = 0; p < np; ++p) {
                TRY(grad_elementwise(c, ks, p, c->X.d() + (size_t)r0 * D, c->y.d() + r0, rows, c->Z.d(), M, c->W.d(), M,
                                     beta, v_wv, c->T.d(), M, sc + SC_DVAR_NM + p));
                TRY(gemm(c, true, false, M, Wd, (int)rows, 1.0, c->T.d(), M, c->Xt.d() + (size_t)r0 * Wd, Wd,
                         ch > 0 ? 1.0 : 0.0, c->gt_nm.d() + (size_t)p * M * Wd, Wd));
            }
        }
        return EDRGP_OK;
    }

    // SparseGP._update_gradients: everything up to and including the device -> host copies of the gradient
    // tables and the scalars, and the wait for them
    int gradients_run() {
        double* S2 = c->mats[MAT_S2].d();
        {
            ProfScope ps(c, EDRGP_PROF_SMALL);
            // S2 = 2 dL/dpsi2 = beta (Kmm^-1 - P - wv wv^T)
            TRY(combine3(c, M, beta, c->mats[MAT_KMMI].d(), -beta, c->mats[MAT_P].d(), 0.0, nullptr, -beta, v_wv, S2));
        }
        TRY(c->gt_mm.reserve((size_t)2 * np * M * Wd * sizeof(double)));
        TRY(c->gt_nm.reserve((size_t)np * M * Wd * sizeof(double)));
        {
            // the fused fp64 row pass releases the Kmm-term chain beside its last launch; every other path queues it
            // in front of the row pass
            const bool fused_expected = !g32 && (np == 1) && c->fuse_grad &&
                                        (D <= 32 || (c->kind[0] == EDRGP_KERN_RBF && c->rbf_epilogue) || want_h());
            if (!fused_expected) TRY(queue_kmm_chain());
        }
        if (g32) TRY(grad_rows32_queue(S2));
        else TRY(grad_rows64_queue(S2));
        tr.mark("grad queued");
        if (c->overlap) {
            TRY(queue_kmm_chain());              // (paths that did not release it beside their last launch)
            TRY(main_waits_side(c));
        } else {
            TRY(kmm_term_queue());
        }
        if (c->sharded()) {
            TRY(allreduce_sum(c, c->gt_nm.d(), (size_t)np * M * Wd));
            TRY(allreduce_sum(c, sc + SC_DVAR_NM, EDRGP_MAX_PARTS));
        }
        // pinned host staging owned by the context (no per-evaluation allocation / zero-fill; true async copies);
        // Z itself is the host copy kept by edrgp_set_inducing
        const size_t n_nm = (size_t)np * M * Wd, n_mm = (size_t)2 * np * M * Wd;
        fin_dev = c->grad_finish_dev != 0;
        if (c->h_stage_cap < n_nm + n_mm) {
            if (c->h_stage) (void)hipHostFree(c->h_stage);
            c->h_stage = nullptr;
            c->h_stage_cap = 0;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), (n_nm + n_mm) * sizeof(double), hipHostMallocDefault));
            c->h_stage_cap = n_nm + n_mm;
        }
        if (fin_dev) {
            // dZ and the length-scale partial sums leave the device instead of the three tables
            const size_t n_dz = (size_t)Mr * D;
            const size_t n_sp = (size_t)((Mr + 31) / 32) * np * D;
            TRY(c->gfin.reserve((n_dz + n_sp) * sizeof(double)));
            TRY(grad_finish(c, np, M, Mr, D, c->gt_nm.d(), c->gt_mm.d(), c->Z.d(), c->invl.d(), c->gfin.d(), c->gfin.d() + n_dz,
                            &fin_blocks));
            HIP_TRY(hipMemcpyAsync(c->h_stage, c->gfin.d(), (n_dz + n_sp) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(c->h_stage, c->gt_nm.d(), n_nm * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(c->h_stage + n_nm, c->gt_mm.d(), n_mm * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        tr.mark("grad D2H queued");
        TRY(fetch());
        c->drain_events();
        tr.mark("grad synced");
        return EDRGP_OK;
    }

    // finishing on the host: O(M D) arithmetic on the fetched tables
    void gradients_finish() {
        const size_t n_nm = (size_t)np * M * Wd;
        const double* g_nm = c->h_stage;
        const double* g_mm = c->h_stage + n_nm;
        const std::vector<double>& Zh = c->Z_host;
        const double psi0sum = gl.N * var_sum(c);
        const double trA = h_sc[SC_TRA], data_fit = h_sc[SC_DATAFIT];
        out->dvar.assign(np, 0.0);
        out->dlen.assign((size_t)np * D, 0.0);
        out->dZ.assign((size_t)Mr * D, 0.0);
        std::vector<double> s_acc((size_t)D), il2v((size_t)D);
        if (fin_dev) {
            const size_t n_dz = (size_t)Mr * D;
            const double* sp = c->h_stage + n_dz;
            std::memcpy(out->dZ.data(), c->h_stage, n_dz * sizeof(double));
            for (int p = 0; p < np; ++p) {
                out->dvar[p] = h_sc[SC_DVAR_NM + p] + h_sc[SC_DVAR_MM + p] - 0.5 * beta * gl.N;   // + sum dL_dKdiag
                for (int q = 0; q < D; ++q) {
                    double acc = 0.0;
                    for (int b = 0; b < fin_blocks; ++b) acc += sp[((size_t)b * np + p) * D + q];     // fixed order
                    const double l = c->lens[(size_t)p * D + q];
                    out->dlen[(size_t)p * D + q] = -(1.0 / (l * l) / l) * acc;
                }
            }
        } else
        for (int p = 0; p < np; ++p) {
            out->dvar[p] = h_sc[SC_DVAR_NM + p] + h_sc[SC_DVAR_MM + p] - 0.5 * beta * gl.N;   // + sum dL_dKdiag
            const double* Gn = &g_nm[(size_t)p * M * Wd];
            const double* GmT = &g_mm[(size_t)(2 * p) * M * Wd];
            const double* GmN = &g_mm[(size_t)(2 * p + 1) * M * Wd];
            for (int q = 0; q < D; ++q) {
                const double l = c->lens[(size_t)p * D + q];
                il2v[q] = 1.0 / (l * l);
                s_acc[q] = 0.0;
            }
            // rows of the M x (2D+1) tables are walked once (m outer, q inner); every per-q sum still runs over
            // m in ascending order, so the result is the one of the q-outer form
            double* __restrict__ sa = s_acc.data();
            const double* __restrict__ il2 = il2v.data();
            for (int m = 0; m < Mr; ++m) {            // phantom rows of the tables are zero
                const double* __restrict__ gn = Gn + (size_t)m * Wd;
                const double* __restrict__ gt = GmT + (size_t)m * Wd;
                const double* __restrict__ gN = GmN + (size_t)m * Wd;
                const double* __restrict__ zr = &Zh[(size_t)m * D];
                double* __restrict__ dz = &out->dZ[(size_t)m * D];
                const double cn = gn[D], ct = gt[D], cN = gN[D];
#pragma clang loop vectorize(enable) interleave(enable)
                for (int q = 0; q < D; ++q) {
                    const double z = zr[q];
                    double s_nm = gn[D + 1 + q] + cn * z * z - 2.0 * z * gn[q];
                    double s_mm = gt[D + 1 + q] + ct * z * z - 2.0 * z * gt[q];
                    sa[q] += s_nm + s_mm;
                    dz[q] += il2[q] * ((z * cn - gn[q]) + (z * ct - gt[q]) + (z * cN - gN[q]));
                }
            }
            for (int q = 0; q < D; ++q) out->dlen[(size_t)p * D + q] = -(il2v[q] / c->lens[(size_t)p * D + q]) * s_acc[q];
        }
        // _compute_dL_dR (homoscedastic); sum(A .* DBi_plus_BiPBi) = M - |LBi|_F^2 + |c|^2 - |u|^2
        const double sumA_DBi = (double)M - h_sc[SC_LBIFROB] + data_fit - h_sc[SC_USQ];
        double dR = -0.5 * gl.N * beta + 0.5 * gl.yy * beta * beta;
        dR += 0.5 * (psi0sum * beta * beta - trA * beta);
        dR += beta * (0.5 * sumA_DBi - data_fit);
        out->dnoise = dR;
        tr.mark("host finishing");
    }

    // ------------------------------------------------------------------ drivers
    // Phase 1 and whatever of the row pass does not depend on it, placed on the two streams.
    // The O(M^3), latency-bound phase 1 does not depend on the rows.  Level 3 (default): with one resident chunk
    // the K_nm build (needed by either form) runs on the main stream beside phase 1 on the side stream and the Phi
    // kernel then keeps the GPU to itself (Cm bound-only: 10.4 ms; the whole row pass beside phase 1: 10.9 ms, the
    // Phi kernel stretched from 6.6 to 8.2 ms by the side stream's 70 launches; no overlap: 10.8 ms).  Chunked row
    // passes of bound-only evaluations in the Gram form run whole beside phase 1.  Level 2 forces the latter.
    // Returns whether the row pass has been queued already (in the Gram form, `f32` as it stood).
    int stage_phase1(bool careful, bool assume_stable, bool* rows_done) {
        *rows_done = false;
        const bool can_prebuild = c->overlap && nchunks == 1 && !f32 && c->prebuild;
        // `can_prebuild` looks at the LOCAL chunk count, and a row pass queued here all-reduces Phi before the verdict
        // on the form is known (a second all-reduce follows when the rule then asks for the stable form): with a
        // communicator the choice must not depend on it, or a chunked rank would issue two collectives where a
        // single-chunk rank issues one.  Sharded contexts therefore never take the local-count branch;
        // `overlap` = 2 is an option every rank shares.
        const bool beside_rows = ((c->overlap == 2) || (c->overlap == 3 && !need_grad && !can_prebuild && !c->sharded())) &&
                                 !assume_stable;
        // Z and the hyper-parameters reach the device by asynchronous copies on the main stream
        // (edrgp_set_inducing / edrgp_set_kernel do not wait): the side stream starts behind them
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        if (beside_rows) {
            c->syrk_shares_gpu = true;
            const int st2 = phase2(false);
            c->syrk_shares_gpu = false;
            if (st2 != EDRGP_OK) return st2;
            *rows_done = true;
            SideStream side(c);
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_fork, 0));
            c->chol_shares_gpu = true;
            int st = phase1(careful);
            c->chol_shares_gpu = false;
            if (st == EDRGP_OK) st = side.join_into_main();
            return st;
        }
        if (can_prebuild) {
            {
                // A workgroup of the build holds 36.9 KB of LDS, four per CU; a Cholesky step workgroup (42.5 KB)
                // fits neither into the 12 KB left over nor -- LDS is allocated contiguously -- into the hole one
                // retiring build workgroup leaves, so the steps of phase 1 starved until the build had drained
                // (first step 456 us instead of 38).  Padded by `kbuild_pad` (16 KB: 53 KB per workgroup) the build runs
                // three per CU and every hole it leaves takes a step workgroup of the high-priority side stream; the
                // two then share the CUs (steps ~55 us, build 1.2 ms instead of 0.73) and the Phi kernel starts 0.15 ms
                // earlier (Cm bound-only 10.25 -> 9.8 ms; padding 6 / 16 / 20 / 26 / 44 KB: 10.05 / 9.81 / 9.91 / 10.10 / 9.98).
                ProfScope ps(c, EDRGP_PROF_KBUILD);
                c->kbuild_pad_now = c->kbuild_pad;
                const int st = build_knm(0, N, c->Knm.d(), c->y.d(), bvec);
                c->kbuild_pad_now = 0;
                if (st != EDRGP_OK) return st;
            }
            prebuilt = true;
            SideStream side(c);
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_fork, 0));
            c->chol_shares_gpu = true;
            int st = phase1(careful);
            c->chol_shares_gpu = false;
            if (st == EDRGP_OK) st = side.join_into_main();
            return st;
        }
        return phase1(careful);
    }

    int finish_after_bound() {          // posterior / gradients once phase 3 is queued (and, careful driver, verified)
        if (!need_grad && !need_post) return EDRGP_OK;
        TRY(factors_queue());
        if (need_post) {
            TRY(posterior_queue());
            c->have_post = true;
            c->post_dense = false;
            c->P = Mr;
        }
        return EDRGP_OK;
    }

    void remember_for_posterior() {
        // Kmm^-1, P and wv of this (data, Z, theta) stay in the workspaces: a posterior request that follows
        // without any change of state is served from them (GPy's inference returns the posterior with every
        // evaluation)
        c->grad_serial = c->state_serial;
        c->grad_lml = out->lml;
        c->grad_M = M;
        c->grad_Mr = Mr;
    }

    // CAREFUL driver: synchronises after phase 1 (verdict on the form, jitchol ladder) and after phase 3.
    int run_careful() {
        TRY(init());
        const bool assume_stable = (c->stats_mode == 2) || (c->stats_mode == 0 && c->last_mode == 2);
        const bool ran32 = f32;
        bool rows_done = false;
        tr.mark("setup");
        TRY(stage_phase1(true, assume_stable, &rows_done));
        const double pr2 = pivot_ratio2();
        const double dmin_lm = h_sc[SC_DIAGMIN];
        c->last_pivot_ratio2 = pr2;
        f32 = want32 && c->f32_phi && !c->phi32_bad && (pr2 <= c->f32_ratio_max);
        const bool stable64 = rule_says_stable();
        // K_nm S2 in fp32 cancels catastrophically once Kmm is ill conditioned (S2 carries Kmm^-1): the fp32
        // gradient contraction is used only where the Gram form itself is considered safe
        if (stable64 || pr2 > pivot_ratio_limit(c, Mr)) g32 = false;
        stable = !f32 && stable64;
        c->last_mode = f32 ? 3 : (stable ? 2 : 1);
        tr.mark("phase1 (+queued phase2)");
        if (!rows_done || stable || f32 != ran32) TRY(phase2(stable));
        tr.mark("phase2 queued");
        TRY(kmmi_beside_phase3());
        {
            int st3 = phase3_careful(f32);
            if (f32) {
                // a-posteriori guard of the fp32 Gram form: the rounding error of Phi (relative gamma32 per
                // entry, ||dPhi||_2 ~ 2 gamma32 ||Phi||_F / sqrt(M)) reaches A = beta Li Phi Li^T amplified by
                // ||Li||_2^2 ~ 1 / min diag(Lm)^2 and must stay small against lambda_min(I + A) ~ min diag(LB)^2
                bool bad = (st3 == EDRGP_ERR_NOT_PD);
                if (st3 != EDRGP_OK && !bad) return st3;
                if (!bad) {
                    const double gamma32 = 3e-7;
                    const double dphi = 2.0 * gamma32 * std::sqrt(h_sc[SC_PHIFROB] / M);
                    const double dlb = h_sc[SC_LBDIAGMIN];
                    const double rho = beta * dphi / (dmin_lm * dmin_lm * dlb * dlb);
                    c->f32_rho = rho;
                    bad = !(rho <= c->f32_guard);
                }
                if (bad) {          // this evaluation (and the following ones) take Phi from the fp64 row pass
                    f32 = false;
                    c->phi32_bad = true;
                    stable = stable64;
                    c->last_mode = stable ? 2 : 1;
                    TRY(phase2(stable));
                    TRY(phase3_careful(false));
                }
            } else if (st3 != EDRGP_OK) {
                return st3;
            }
        }
        c->drain_events();
        tr.mark("phase3 synced");
        set_bound();
        if (!need_grad && !need_post) return EDRGP_OK;
        TRY(finish_after_bound());
        if (!need_grad) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->drain_events();
            return EDRGP_OK;
        }
        TRY(gradients_run());
        gradients_finish();
        remember_for_posterior();
        return EDRGP_OK;
    }

    // SYNC-FREE driver: one wait, at the end.  Returns RETRY_* when the device's verdict disagrees with what
    // was assumed; the caller then repeats the evaluation with the careful driver.
    int run_sync_free() {
        TRY(init());
        const bool assume_stable = (c->stats_mode == 2) || (c->stats_mode == 0 && c->last_mode == 2);
        bool rows_done = false;
        tr.mark("setup");
        TRY(stage_phase1(false, assume_stable, &rows_done));
        stable = assume_stable;
        if (!rows_done) TRY(phase2(stable));
        tr.mark("phase 1 + 2 queued");
        TRY(kmmi_beside_phase3());
        TRY(phase3_queue(-1));
        TRY(finish_after_bound());
        if (need_grad) TRY(gradients_run());
        else TRY(fetch());
        c->drain_events();
        tr.mark("synced");
        // the verdict
        if (h_info[0] != 0 || h_info[1] != 0) { c->have_post = false; return RETRY_JITTER; }
        c->last_pivot_ratio2 = pivot_ratio2();
        if (rule_says_stable() != stable) {
            // the careful repeat starts from the form the rule has just asked for (no wasted row pass in the other one)
            c->last_mode = rule_says_stable() ? 2 : 1;
            c->have_post = false;
            return RETRY_CAREFUL;
        }
        c->last_mode = stable ? 2 : 1;
        set_bound();
        if (need_grad) {
            gradients_finish();
            remember_for_posterior();
        }
        return EDRGP_OK;
    }
};

}  // namespace

int sparse_eval(Ctx* c, bool need_grad, bool need_post, SparseOut* out) {
    // precision 32 decides between its fp32 and fp64 row passes on the verdict of phase 1: careful driver
    const bool want32 = (c->precision == 32) && (c->nparts == 1);
    if (c->sync_free && !want32 && c->careful_left == 0) {
        SparseEval ev(c, need_grad, need_post, out);
        const int st = ev.run_sync_free();
        if (st <= 0) return st;                       // done, or a hard error
        // a factorisation that needs jitter tends to need it again at the optimiser's next iterates: skip the
        // optimistic attempt for a while
        if (st == RETRY_JITTER) c->careful_left = 8;
        ++c->retries;
    } else if (c->careful_left > 0) {
        --c->careful_left;
    }
    SparseEval ev(c, need_grad, need_post, out);
    return ev.run_careful();
}

// Posterior (woodbury vector / inverse) out of the workspaces of the bound+gradient evaluation that has just run
// for the same state.
int posterior_from_last_gradient(Ctx* c) {
    const int M = c->grad_M, Mr = c->grad_Mr;
    const double* Kmmi = c->mats[MAT_KMMI].d();
    const double* P = c->mats[MAT_P].d();
    double* Tm = c->mats[MAT_T].d();
    TRY(c->wv.reserve((size_t)Mr * sizeof(double)));
    TRY(c->winv.reserve((size_t)Mr * Mr * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->wv.d(), c->tmpvec[3].d(), (size_t)Mr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (M != Mr) {
        TRY(combine3(c, M, 1.0, Kmmi, -1.0, P, 0.0, nullptr, 0.0, nullptr, Tm));
        HIP_TRY(hipMemcpy2DAsync(c->winv.d(), (size_t)Mr * sizeof(double), Tm, (size_t)M * sizeof(double),
                                 (size_t)Mr * sizeof(double), (size_t)Mr, hipMemcpyDeviceToDevice, c->stream));
    } else {
        TRY(combine3(c, M, 1.0, Kmmi, -1.0, P, 0.0, nullptr, 0.0, nullptr, c->winv.d()));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_post = true;
    c->post_dense = false;
    c->P = Mr;
    return EDRGP_OK;
}

}  // namespace edrgp
