// C-ABI entry points and the host-side orchestration of the GP inference
// pipelines (include/edrgp_hip.h).  The operation sequence restates GPy
// VarDTC.inference / ExactGaussianInference.inference / Posterior._raw_predict
// in the lean Gram form: everything O(N) is a tile build
// plus MFMA contractions, everything O(M^3) is MFMA GEMMs around two blocked
// Choleskys with explicit triangular inverses.
#include "common.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <cstdlib>

namespace edrgp {

static thread_local std::string g_err;
static int g_tri_tile = 64;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int DBuf::reserve(size_t need) {
    if (need <= bytes && p != nullptr) return EDRGP_OK;
    if (need == 0) need = 8;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e));
        return EDRGP_ERR_HIP;
    }
    bytes = need;
    return EDRGP_OK;
}

void DBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

ProfScope::ProfScope(Ctx* c_, int slot_) : c(c_), slot(slot_) {
    if (!c->profile) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
    (void)hipEventRecord(e0, c->stream);
}

ProfScope::~ProfScope() {
    if (!e0) return;
    (void)hipEventRecord(e1, c->stream);
    c->pending_events.push_back({slot, {e0, e1}});
}

int Ctx::drain_events() {
    for (auto& pe : pending_events) {
        float ms = 0.f;
        if (hipEventSynchronize(pe.second.second) == hipSuccess &&
            hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) {
            prof[pe.first].total_ms += ms;
            prof[pe.first].launches += 1;
        }
        (void)hipEventDestroy(pe.second.first);
        (void)hipEventDestroy(pe.second.second);
    }
    pending_events.clear();
    return EDRGP_OK;
}

__global__ void axpy_kernel(int64_t n, double a, const double* __restrict__ x, double* __restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] += a * x[i];
}

int vec_axpy(Ctx* c, int64_t n, double a, const double* x, double* y) {
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (n + 255) / 256));
    hipLaunchKernelGGL(axpy_kernel, dim3(nb), dim3(256), 0, c->stream, n, a, x, y);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

// ----------------------------------------------------------------- indices
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
};

enum {  // device scalar slots (c->scal)
    SC_TRA = 0, SC_SUMLOGLB, SC_DATAFIT, SC_USQ, SC_LBIFROB, SC_YY,
    SC_DVAR_NM = 6,          // + part (4)
    SC_DVAR_MM = 10,         // + part (4)
    SC_YALPHA = 14, SC_SUMLOGL, SC_ALPHASQ, SC_TRKYI,
    SC_DIAGMIN = 18, SC_DIAGMAX = 19,
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
        c->ks_x.swap(c->ks_x2);         // the side stream builds Kmm while the main one builds K_nm
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

// CU-masked streams (hipExtStreamCreateWithCUMask).  Mask bit b is CU b/8 of XCD b%8 on this part, so the
// low `side_cus` bits give the side stream side_cus/8 CUs in every XCD.
static bool ensure_masked_streams(Ctx* c) {
    if (c->masked_state != 0) return c->masked_state > 0;
    c->masked_state = -1;
    const int ncu = c->num_cu, side = c->side_cus;
    if (side <= 0 || side >= ncu / 2 || ncu > 512) return false;
    uint32_t m_side[16] = {0}, m_main[16] = {0};
    // mask_layout 1 (default): the low `side` bits.  Measured on MI355X: consecutive mask bits go to
    //   consecutive XCDs, so this gives every XCD side/8 CUs (side = 32: 11.8 ms per bound evaluation at Cm
    //   vs 12.2 unmasked; 8 or 16 CUs are too few for the O(M^3) phase: 14-15 ms).
    // mask_layout 0: side/8 bits at the start of each block of ncu/8 bits (15 ms: unbalances the XCDs).
    const int per_xcd = std::max(1, ncu / 8), per_side = std::max(1, side / 8);
    for (int b = 0; b < ncu; ++b) {
        const bool is_side = (c->mask_layout == 0) ? ((b % per_xcd) < per_side && (b / per_xcd) < 8) : (b < side);
        (is_side ? m_side : m_main)[b >> 5] |= (1u << (b & 31));
    }
    const uint32_t words = (uint32_t)((ncu + 31) / 32);
    if (hipExtStreamCreateWithCUMask(&c->stream_mm, words, m_main) != hipSuccess) { (void)hipGetLastError(); c->stream_mm = nullptr; return false; }
    if (hipExtStreamCreateWithCUMask(&c->stream_ms, words, m_side) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamDestroy(c->stream_mm);
        c->stream_mm = c->stream_ms = nullptr;
        return false;
    }
    c->masked_state = 1;
    return true;
}

struct MaskedScope {                    // while active, c->stream / c->stream2 are the masked pair
    Ctx* c;
    bool on;
    MaskedScope(Ctx* c_, bool on_) : c(c_), on(on_) { if (on) swap(+1); }
    ~MaskedScope() { if (on) swap(-1); }
    void swap(int dir) {
        std::swap(c->stream, c->stream_mm);
        std::swap(c->stream2, c->stream_ms);
        c->num_cu -= dir * c->side_cus;
    }
};

static int main_waits_side(Ctx* c) {       // outside a SideStream scope: c->stream is the main stream
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return EDRGP_OK;
}

static GemmOpts kopt(int kmode, int tri = 0, int tag = 0) {
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
static int gram_of_lower(Ctx* c, int n, const double* L, double* C) {
    TRY(gemm(c, true, false, n, n, n, 1.0, L, n, L, n, 0.0, C, n, kopt(K_GE_ROW | K_GE_COL, 1)));
    return mirror_lower(c, C, n, n);
}

static KSpec make_kspec(const Ctx* c) {
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

static double var_sum(const Ctx* c) {
    double s = 0.0;
    for (int p = 0; p < c->nparts; ++p) s += c->var[p];
    return s;
}

static int ensure_mats(Ctx* c, int n, int count, size_t extra_first = 0) {
    for (int i = 0; i < count; ++i) {
        size_t need = (size_t)n * n * sizeof(double);
        if (i == MAT_PHI) need += ((size_t)n + 8) * sizeof(double);
        (void)extra_first;
        TRY(c->mats[i].reserve(need));
    }
    TRY(c->scal.reserve(SC_COUNT * sizeof(double)));
    TRY(c->info.reserve(4 * sizeof(int)));
    for (int i = 0; i < 6; ++i) TRY(c->tmpvec[i].reserve((size_t)(n + 8) * sizeof(double)));
    return EDRGP_OK;
}

// globals over ranks (N, yy) -- data-dependent only, cached
struct Globals { double N, yy; };
static int get_globals(Ctx* c, Globals* g) {
    g->N = (double)c->N;
    g->yy = c->yy_local;
    if (c->comm && c->glob_valid) {        // all-reduced once per data set (every rank takes this branch together)
        g->N = c->glob_N;
        g->yy = c->glob_yy;
        return EDRGP_OK;
    }
    if (c->comm) {
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

// jitchol's jitter ladder: level -1 = none, level k = mean(diag) * 1e-6 * 10^k
static double jit_amount(int level, double diag_mean_) {
    if (level < 0) return 0.0;
    return diag_mean_ * 1e-6 * std::pow(10.0, (double)level);
}

struct SparseOut {
    double lml = 0.0;
    std::vector<double> dvar, dlen, dZ;
    double dnoise = 0.0;
};

// One evaluation of the Titsias bound (VarDTC.inference) with optional
// gradients / posterior.  Returns EDRGP_ERR_NOT_PD when a factorisation fails
// at every jitter level.
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

static int sparse_eval(Ctx* c, bool need_grad, bool need_post, SparseOut* out) {
    HostTrace tr(c->trace);
    if (!c->have_data || !c->have_Z || !c->have_kernel) {
        set_error("sparse evaluation needs set_data, set_inducing and set_kernel first");
        return EDRGP_ERR_STATE;
    }
    const int D = c->D;
    const int64_t N = c->N;
    const int Wd = 2 * D + 1;
    const int np = c->nparts;
    c->grad_serial = 0;                 // the workspaces are about to be overwritten
    // Working size.  The matrix kernels run their fast paths on whole 128-wide tiles and whole k-tiles only
    // (M = 1000 ran 1.4x slower than M = 1024).  An inducing set whose size is not a multiple of 128 is therefore
    // embedded in the next multiple: Mr real points plus M - Mr phantom points that are decoupled from
    // everything -- K_nm gets zero columns, Kmm the block [[Kmm, 0], [0, I]].  Then Lm, Lm^-1, B, LB, LB^-1,
    // Kmm^-1, P are block diagonal with an identity block, Phi / A / S2 / dL/dKmm have zero phantom blocks, and
    // the bound and every gradient of the real parameters are those of the Mr-point problem (the log-determinants
    // gain log 1, the traces 0).  Only the builds know about Mr; everything between them works on M.
    const int Mr = c->M;
    const bool padded = c->pad_inducing && Mr > 256 && (Mr % 128) != 0 && c->precision != 32;
    const int M = padded ? (Mr + 127) / 128 * 128 : Mr;
    c->M_work = M;
    c->knm_train_valid = false;
    const double beta = 1.0 / std::max(c->noise_var, 1e-8);
    const KSpec ks = make_kspec(c);
    Globals gl;
    TRY(get_globals(c, &gl));

    TRY(ensure_mats(c, M, need_grad ? 14 : (need_post ? 10 : 7)));
    const int64_t chunk = std::min<int64_t>(N, std::max<int64_t>(64, c->chunk_rows));
    const int64_t nchunks = (N + chunk - 1) / chunk;
    // 16 spare rows: the row count handed to the Phi kernel is rounded up to whole k-tiles with zero rows
    TRY(c->Knm.reserve((size_t)(chunk + 16) * M * sizeof(double)));
    double* Lm = c->mats[MAT_LM].d();
    double* Li = c->mats[MAT_LI].d();
    double* Phi = c->mats[MAT_PHI].d();
    double* bvec = Phi + (size_t)M * M;
    double* Tm = c->mats[MAT_T].d();
    double* A = c->mats[MAT_A].d();
    double* LB = c->mats[MAT_LB].d();
    double* LBi = c->mats[MAT_LBI].d();
    double* sc = c->scal.d();
    int* info = static_cast<int*>(c->info.p);
    double* v_v = c->tmpvec[0].d();
    double* v_c = c->tmpvec[1].d();
    double* v_u = c->tmpvec[2].d();
    double* v_wv = c->tmpvec[3].d();
    double* v_tmp = c->tmpvec[4].d();

    double h_sc[SC_COUNT];
    int h_info[4];

    // ---- phase 1: Kmm + (1e-8 + jitter) I, Lm (jitchol ladder), Lm^-1
    const double kmm_diag_mean = var_sum(c) + 1e-8;
    auto phase1 = [&]() -> int {
        for (int lvl = -1;;) {
            ProfScope ps(c, EDRGP_PROF_SMALL);
            if (padded) HIP_TRY(hipMemsetAsync(Lm, 0, (size_t)M * M * sizeof(double), c->stream));
            TRY(kernel_matrix(c, ks, c->Z.d(), Mr, c->Z.d(), Mr, Lm, M, 1, 1e-8 + jit_amount(lvl, kmm_diag_mean),
                              nullptr, nullptr));
            if (padded) TRY(add_diag(c, Lm + (size_t)Mr * M + Mr, M - Mr, M, 1.0));
            TRY(potrf_lower(c, Lm, M, M, info + 0));
            TRY(diag_minmax(c, Lm, Mr, M, sc + SC_DIAGMIN));
            HIP_TRY(hipMemcpyAsync(h_sc, sc, sizeof(h_sc), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(h_info, info, sizeof(h_info), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (h_info[0] == 0) break;
            if (++lvl > 4) { set_error("Kmm not positive definite, even with jitter"); return EDRGP_ERR_NOT_PD; }
        }
        TRY(trtri_lower(c, Lm, M, Li, M, M));
        return EDRGP_OK;
    };

    // ---- phase 2: row statistics.  gram: Phi = K^T K, b = K^T y.
    //      stable: Phi := (K Li^T)^T (K Li^T), b := (K Li^T)^T y  (= Li b)
    bool k_resident = false;
    bool prebuilt = false;             // K_nm (and b) of the single row chunk were built beside phase 1
    bool kmmi_done = false;            // Kmm^-1 = Li^T Li was formed on the side stream
    // precision 32 asks for the fp32 row pass; it is used while the Gram form is safe in fp32
    // (error ~ eps32 cond(Kmm), judged by the pivot-ratio proxy after phase 1), otherwise this
    // evaluation falls back to the fp64 row pass (last_mode: 1 gram fp64, 2 stable fp64, 3 gram fp32)
    //   Phi in fp32 (f32): opt-in ("f32_phi" = 1; the bound is then only 1e-4..1e-3 accurate): attempted, kept only
    //     if the a-posteriori guard below accepts it; a rejected attempt is remembered (phi32_bad) until the
    //     data, Z or the precision change.  Default: Phi always from the fp64 row pass, the bound is exact.
    //   gradient contraction in fp32 (g32): always -- its rounding reaches the gradients only, never
    //     the positive definiteness of I + A.
    const bool want32 = (c->precision == 32) && (np == 1);
    bool g32 = want32 && need_grad && D <= 64;      // and a Kmm conditioned for the Gram form, decided after phase 1
    bool f32 = want32 && c->f32_phi && !c->phi32_bad;
    bool k32_resident = false;
    // K(X[r0 : r0 + rows], Z) into `out` (ld M) and, when yv is given, K^T yv into bt; phantom columns are zero
    auto build_knm = [&](int64_t r0, int64_t rows, double* outK, const double* yv, double* bt) -> int {
        TRY(kernel_matrix(c, ks, c->X.d() + (size_t)r0 * D, rows, c->Z.d(), Mr, outK, M, 0, 0.0, yv, bt));
        if (padded) {
            HIP_TRY(hipMemset2DAsync(outK + Mr, (size_t)M * sizeof(double), 0, (size_t)(M - Mr) * sizeof(double), (size_t)rows,
                                     c->stream));
            if (bt != nullptr) HIP_TRY(hipMemsetAsync(bt + Mr, 0, (size_t)(M - Mr) * sizeof(double), c->stream));
        }
        return EDRGP_OK;
    };
    auto phase2 = [&](bool stable) -> int {
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
            if (c->comm) TRY(allreduce_sum(c, Phi, (size_t)M * M + M));
            return EDRGP_OK;
        }
        if (stable) TRY(c->W.reserve((size_t)(chunk + 16) * M * sizeof(double)));
        for (int64_t ch = 0; ch < nchunks; ++ch) {
            const int64_t r0 = ch * chunk;
            const int64_t rows = std::min<int64_t>(chunk, N - r0);
            double* btarget = (ch == 0) ? bvec : v_tmp;
            if (!(prebuilt && ch == 0)) {
                ProfScope ps(c, EDRGP_PROF_KBUILD);
                TRY(build_knm(r0, rows, c->Knm.d(), stable ? nullptr : c->y.d() + r0, stable ? nullptr : btarget));
            }
            const double* S = c->Knm.d();
            if (stable) {
                ProfScope ps(c, EDRGP_PROF_GRADGEMM);
                // W = K_nm Li^T with b = W^T y out of the same launch's epilogue (per-row-tile partials, fixed order)
                const int tiles_m = (int)((rows + 127) / 128);
                TRY(c->bpart.reserve((size_t)tiles_m * M * sizeof(double)));
                FuseK fk{};
                fk.y = c->y.d() + r0;
                fk.bpart = c->bpart.d();
                GemmOpts ow = kopt(K_LE_COL, 0, 4);
                ow.force_tile = 128;
                ow.fusek = &fk;
                if (rows >= 128 && M >= 128) {
                    TRY(gemm(c, false, true, (int)rows, M, M, 1.0, c->Knm.d(), M, Li, M, 0.0, c->W.d(), M, ow));
                    TRY(col_reduce(c, c->bpart.d(), tiles_m, M, btarget));
                } else {
                    TRY(gemm(c, false, true, (int)rows, M, M, 1.0, c->Knm.d(), M, Li, M, 0.0, c->W.d(), M,
                             kopt(K_LE_COL, 0, 2)));
                    TRY(gemv_t(c, (int)rows, M, 1.0, c->W.d(), M, c->y.d() + r0, btarget));
                }
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
        c->knm_train_valid = k_resident && !padded;          // predictions address K_nm with ld = number of real points          // (the stable form keeps K_nm in Knm and K_nm Li^T in W)
        if (c->comm) TRY(allreduce_sum(c, Phi, (size_t)M * M + M));
        return EDRGP_OK;
    };

    // The O(M^3), latency-bound phase 1 does not depend on the rows: when the
    // Gram form is expected (it ran last time, or is forced) the row pass is
    // queued first on the main stream and phase 1 runs beside it on the side
    // stream.  The pivot-ratio proxy of cond(Kmm) is known only after phase 1;
    // if it then asks for the stable form, the row pass is redone that way.
    // (Gram form: A = beta Li (K^T K) Li^T, rounding error ~ eps cond(Kmm);
    //  stable form: A = beta (K Li^T)^T (K Li^T), error ~ eps sqrt(cond).)
    tr.mark("setup");
    // level 3 (default): with one resident chunk, the K_nm build runs beside phase 1 and the Phi kernel keeps
    // the GPU to itself (Cm bound-only: 10.4 ms; the whole row pass beside phase 1: 10.9 ms, the Phi kernel
    // stretched from 6.6 to 8.2 ms by the side stream's 70 launches; no overlap: 10.8 ms).  Chunked row passes of
    // bound-only evaluations overlap phase 1 with the whole pass.  Level 2 forces the latter everywhere.
    const bool can_prebuild = c->overlap && nchunks == 1 && !f32 && c->prebuild;
    const bool want_p1 = (c->overlap == 2) || (c->overlap == 3 && !need_grad && !can_prebuild);
    // bound-only evaluations (default policy): masked stream pair, phase 1 gets its own CUs
    MaskedScope masked(c, c->overlap == 3 && !need_grad && c->side_cus > 0 && !c->comm && ensure_masked_streams(c));
    const bool optimistic = want_p1 && (c->stats_mode == 1 || (c->stats_mode == 0 && c->last_mode != 2));
    const bool ran32 = f32;
    if (optimistic) {
        TRY(phase2(false));
        {
            SideStream side(c);
            int st = phase1();
            if (st == EDRGP_OK) st = side.join_into_main();
            if (st != EDRGP_OK) return st;
        }
    } else if (can_prebuild) {
        // The mode (Gram / stable) is only known after phase 1, but K_nm is needed either
        // way.  Its build (bandwidth / exp-bound) runs on the main stream while the latency-bound phase 1 -- and
        // Kmm^-1 = Li^T Li for the gradient -- run on the side stream; the Phi kernel then starts alone.
        {
            ProfScope ps(c, EDRGP_PROF_KBUILD);
            TRY(build_knm(0, N, c->Knm.d(), c->y.d(), bvec));
        }
        prebuilt = true;
        SideStream side(c);
        int st = phase1();
        if (st == EDRGP_OK && (need_grad || need_post)) {
            ProfScope ps(c, EDRGP_PROF_SMALL);
            st = gram_of_lower(c, M, Li, c->mats[MAT_KMMI].d());
            kmmi_done = (st == EDRGP_OK);
        }
        if (st == EDRGP_OK) st = side.join_into_main();
        if (st != EDRGP_OK) return st;
    } else {
        TRY(phase1());
    }
    const double pr = h_sc[SC_DIAGMAX] / h_sc[SC_DIAGMIN];
    const double dmin_lm = h_sc[SC_DIAGMIN];
    c->last_pivot_ratio2 = pr * pr;
    f32 = want32 && c->f32_phi && !c->phi32_bad && (pr * pr <= c->f32_ratio_max);
    const bool stable64 = (c->stats_mode == 2) || (c->stats_mode == 0 && pr * pr > c->pivot_ratio_max);
    // K_nm S2 in fp32 cancels catastrophically once Kmm is ill conditioned (S2 carries Kmm^-1): the fp32
    // gradient contraction is used only where the Gram form itself is considered safe
    if (stable64 || pr * pr > c->pivot_ratio_max) g32 = false;
    bool stable = !f32 && stable64;
    c->last_mode = f32 ? 3 : (stable ? 2 : 1);
    tr.mark("phase1 (+queued phase2)");
    if (!optimistic || stable || f32 != ran32) TRY(phase2(stable));
    tr.mark("phase2 queued");

    // ---- phase 3: A, B = I + A (jitchol ladder), LB^-1, the M-vectors and scalars
    // no_jitter: return EDRGP_ERR_NOT_PD at the first failed factorisation instead of climbing the ladder
    // the fused one-launch-per-step factorisation covers n in (64, 3072]; outside it the explicit inverse stays
    const bool rhs_row = !need_grad && !need_post && c->chol_rhs && c->chol_fused && M > 64 && M + 1 <= 3072;
    auto phase3 = [&](bool no_jitter) -> int {
      for (int lvl = -1;;) {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        HIP_TRY(hipMemsetAsync(sc, 0, SC_COUNT * sizeof(double), c->stream));
        if (stable) {
            TRY(combine3(c, M, beta, Phi, 0.0, nullptr, 0.0, nullptr, 0.0, nullptr, A));
            HIP_TRY(hipMemcpyAsync(v_v, bvec, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        } else {
            TRY(gemm(c, false, false, M, M, M, 1.0, Li, M, Phi, M, 0.0, Tm, M, kopt(K_LE_ROW)));
            // A = beta (Li Phi) Li^T is symmetric: lower tiles only, then mirrored
            TRY(gemm(c, false, true, M, M, M, beta, Tm, M, Li, M, 0.0, A, M, kopt(K_LE_COL, 1)));
            TRY(mirror_lower(c, A, M, M));
            TRY(gemv_n(c, M, M, 1.0, Li, M, bvec, v_v));
        }
        TRY(trace_of(c, A, M, M, sc + SC_TRA));
        const double bjit = 1.0 + (lvl >= 0 ? jit_amount(lvl, 1.0 + h_sc[SC_TRA] / Mr) : 0.0);
        if (rhs_row) {
            // bound only: the bound needs |c|^2, c = beta LB^-1 v, and log det LB, not LB^-1.  beta v rides
            // through the factorisation as row M of the (M+1) x (M+1) matrix [[B, .], [beta v^T, alpha]]: the
            // panel steps turn it into c^T (forward substitution by the trailing updates), one extra step
            // instead of the explicit inverse (10 launches) and its products
            const int ldb = M + 8;
            TRY(c->lbaug.reserve((size_t)(M + 1) * ldb * sizeof(double)));
            double* Lg = c->lbaug.d();
            HIP_TRY(hipMemcpy2DAsync(Lg, (size_t)ldb * sizeof(double), A, (size_t)M * sizeof(double),
                                     (size_t)M * sizeof(double), (size_t)M, hipMemcpyDeviceToDevice, c->stream));
            TRY(add_diag(c, Lg, M, ldb, bjit));
            TRY(aug_rhs_row(c, Lg + (size_t)M * ldb, v_v, M, beta));
            TRY(potrf_lower(c, Lg, M + 1, ldb, info + 1));
            TRY(sum_log_diag(c, Lg, M, ldb, sc + SC_SUMLOGLB));
            TRY(sum_sq(c, Lg + (size_t)M * ldb, M, sc + SC_DATAFIT));
            if (f32) {
                TRY(diag_minmax(c, Lg, M, ldb, sc + SC_LBDIAGMIN));
                TRY(frob_sq(c, Phi, M, M, sc + SC_PHIFROB));
            }
        } else {
        HIP_TRY(hipMemcpyAsync(LB, A, (size_t)M * M * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        // jitter on B: mean(diag(B)) = 1 + trA / M, trA known from the failed pass
        TRY(add_diag(c, LB, M, M, bjit));
        TRY(potrf_lower(c, LB, M, M, info + 1));
        TRY(trtri_lower(c, LB, M, LBi, M, M));
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
        }
        HIP_TRY(hipMemcpyAsync(h_sc, sc, sizeof(h_sc), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h_info, info, sizeof(h_info), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (h_info[1] == 0) break;
        if (no_jitter) { set_error("I + A not positive definite"); return EDRGP_ERR_NOT_PD; }
        if (++lvl > 4) { set_error("I + A not positive definite, even with jitter"); return EDRGP_ERR_NOT_PD; }
      }
      return EDRGP_OK;
    };
    {
        int st3 = phase3(f32);
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
                TRY(phase3(false));
            }
        } else if (st3 != EDRGP_OK) {
            return st3;
        }
    }
    c->drain_events();
    tr.mark("phase3 synced");

    const double psi0sum = gl.N * var_sum(c);
    const double trA = h_sc[SC_TRA], data_fit = h_sc[SC_DATAFIT];
    out->lml = -0.5 * gl.N * (std::log(2.0 * M_PI) - std::log(beta)) - 0.5 * beta * gl.yy -
               0.5 * (beta * psi0sum - trA) - h_sc[SC_SUMLOGLB] + 0.5 * data_fit;
    if (!need_grad && !need_post) return EDRGP_OK;

    double* R = c->mats[MAT_R].d();
    double* P = c->mats[MAT_P].d();
    double* Kmmi = c->mats[MAT_KMMI].d();
    {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        // R = LBi Li is lower triangular: lower tiles only, upper part stays zero
        HIP_TRY(hipMemsetAsync(R, 0, (size_t)M * M * sizeof(double), c->stream));
        TRY(gemm(c, false, false, M, M, M, 1.0, LBi, M, Li, M, 0.0, R, M, kopt(K_LE_ROW | K_GE_COL, 1)));
        TRY(gram_of_lower(c, M, R, P));
        if (!kmmi_done) TRY(gram_of_lower(c, M, Li, Kmmi));
    }
    if (need_post) {
        // the posterior is kept compact: Mr entries of wv, the Mr x Mr block of Kmm^-1 - P
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
        c->have_post = true;
        c->post_dense = false;
        c->P = Mr;
    }
    if (!need_grad) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->drain_events();
        return EDRGP_OK;
    }

    // ---- gradients: SparseGP._update_gradients
    double* S2 = c->mats[MAT_S2].d();
    double* C1 = c->mats[MAT_C1].d();
    double* dKmm = c->mats[MAT_DKMM].d();
    double* Tmm = c->mats[MAT_TMM].d();
    {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        // S2 = 2 dL/dpsi2 = beta (Kmm^-1 - P - wv wv^T)
        TRY(combine3(c, M, beta, Kmmi, -beta, P, 0.0, nullptr, -beta, v_wv, S2));
    }
    TRY(c->gt_mm.reserve((size_t)2 * np * M * Wd * sizeof(double)));
    // Kmm term (X2 = None upstream): dL/dKmm, its element-wise factor, then T^T Zt and
    // T Zt.  O(M^3) and independent of the rows: runs on the side stream beside the
    // N x M x M gradient contraction.
    auto kmm_term = [&]() -> int {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        // C1 = Li^T A ; Tm = C1 Li = Lm^-T A Lm^-1
        TRY(gemm(c, true, false, M, M, M, 1.0, Li, M, A, M, 0.0, C1, M, kopt(K_GE_ROW)));
        TRY(gemm(c, false, false, M, M, M, 1.0, C1, M, Li, M, 0.0, Tm, M, kopt(K_GE_COL)));
        // dL/dKmm = 1/2 Kmm^-1 - 1/2 P - 1/2 wv wv^T - 1/2 Lm^-T A Lm^-1
        TRY(combine3(c, M, 0.5, Kmmi, -0.5, P, -0.5, Tm, -0.5, v_wv, dKmm));
        for (int p = 0; p < np; ++p) {
            TRY(grad_elementwise(c, ks, p, c->Z.d(), nullptr, M, c->Z.d(), M, dKmm, M, 0.0, nullptr, Tmm, M,
                                 sc + SC_DVAR_MM + p));
            TRY(gemm(c, true, false, M, Wd, M, 1.0, Tmm, M, c->Zt.d(), Wd, 0.0,
                     c->gt_mm.d() + (size_t)(2 * p) * M * Wd, Wd));
            TRY(gemm(c, false, false, M, Wd, M, 1.0, Tmm, M, c->Zt.d(), Wd, 0.0,
                     c->gt_mm.d() + (size_t)(2 * p + 1) * M * Wd, Wd));
        }
        return EDRGP_OK;
    };
    if (c->overlap) {
        SideStream side(c);
        int st = side.start_after_main();
        if (st == EDRGP_OK) st = kmm_term();
        if (st == EDRGP_OK) st = side.mark_done();
        if (st != EDRGP_OK) return st;
    }
    TRY(c->gt_nm.reserve((size_t)np * M * Wd * sizeof(double)));
    if (g32) {
        // fp32 gradient row pass: T = (K32 S2 + beta y wv^T) .* h(r) out of the contraction's epilogue,
        // then T^T [X | X.^2]; partial sums and the M x 2D result are fp64
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
    }
    if (!g32) {
        TRY(c->W.reserve((size_t)chunk * M * sizeof(double)));
        TRY(c->T.reserve((size_t)chunk * M * sizeof(double)));
    }
    for (int64_t ch = 0; !g32 && ch < nchunks; ++ch) {
        const int64_t r0 = ch * chunk;
        const int64_t rows = std::min<int64_t>(chunk, N - r0);
        if (!k_resident) {
            ProfScope ps(c, EDRGP_PROF_KBUILD);
            TRY(build_knm(r0, rows, c->Knm.d(), nullptr, nullptr));
        }
        // the distance-recomputing epilogue stages D <= 32 features; the RBF epilogue reads k from K_nm: any D
        const bool fused = (np == 1) && c->fuse_grad &&
                           (D <= 32 || (c->kind[0] == EDRGP_KERN_RBF && c->rbf_epilogue));
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
            f.ldk = M;
            double* G2 = c->g2.d();
            double* csv = G2 + (size_t)M * 2 * D;
            GemmOpts o;
            o.tag = 2;
            o.force_tile = 128;
            o.fuse = &f;
            TRY(gemm(c, false, false, (int)rows, M, M, 1.0, c->Knm.d(), M, S2, M, 0.0, c->T.d(), M, o));
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
    tr.mark("grad queued");
    if (c->overlap) TRY(main_waits_side(c));
    else TRY(kmm_term());
    if (c->comm) {
        TRY(allreduce_sum(c, c->gt_nm.d(), (size_t)np * M * Wd));
        TRY(allreduce_sum(c, sc + SC_DVAR_NM, EDRGP_MAX_PARTS));
    }
    // pinned host staging owned by the context (no per-evaluation allocation / zero-fill; true async copies);
    // Z itself is the host copy kept by edrgp_set_inducing
    const size_t n_nm = (size_t)np * M * Wd, n_mm = (size_t)2 * np * M * Wd;
    if (c->h_stage_cap < n_nm + n_mm) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), (n_nm + n_mm) * sizeof(double), hipHostMallocDefault));
        c->h_stage_cap = n_nm + n_mm;
    }
    double* g_nm = c->h_stage;
    double* g_mm = c->h_stage + n_nm;
    const std::vector<double>& Zh = c->Z_host;
    HIP_TRY(hipMemcpyAsync(h_sc, sc, sizeof(h_sc), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(g_nm, c->gt_nm.d(), n_nm * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(g_mm, c->gt_mm.d(), n_mm * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    tr.mark("grad D2H queued");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->drain_events();
    tr.mark("grad synced");

    // ---- finishing on the host: O(M D) arithmetic
    out->dvar.assign(np, 0.0);
    out->dlen.assign((size_t)np * D, 0.0);
    out->dZ.assign((size_t)Mr * D, 0.0);
    std::vector<double> s_acc((size_t)D), il2v((size_t)D);
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
        for (int m = 0; m < Mr; ++m) {            // phantom rows of the tables are zero
            const double* gn = Gn + (size_t)m * Wd;
            const double* gt = GmT + (size_t)m * Wd;
            const double* gN = GmN + (size_t)m * Wd;
            const double* zr = &Zh[(size_t)m * D];
            double* dz = &out->dZ[(size_t)m * D];
            const double cn = gn[D], ct = gt[D], cN = gN[D];
            for (int q = 0; q < D; ++q) {
                const double z = zr[q];
                double s_nm = gn[D + 1 + q] + cn * z * z - 2.0 * z * gn[q];
                double s_mm = gt[D + 1 + q] + ct * z * z - 2.0 * z * gt[q];
                s_acc[q] += s_nm + s_mm;
                dz[q] += il2v[q] * ((z * cn - gn[q]) + (z * ct - gt[q]) + (z * cN - gN[q]));
            }
        }
        for (int q = 0; q < D; ++q) out->dlen[(size_t)p * D + q] = -(il2v[q] / c->lens[(size_t)p * D + q]) * s_acc[q];
    }
    {
        // _compute_dL_dR (homoscedastic); sum(A .* DBi_plus_BiPBi) = M - |LBi|_F^2 + |c|^2 - |u|^2
        const double sumA_DBi = (double)M - h_sc[SC_LBIFROB] + data_fit - h_sc[SC_USQ];
        double dR = -0.5 * gl.N * beta + 0.5 * gl.yy * beta * beta;
        dR += 0.5 * (psi0sum * beta * beta - trA * beta);
        dR += beta * (0.5 * sumA_DBi - data_fit);
        out->dnoise = dR;
    }
    tr.mark("host finishing");
    // Kmm^-1, P and wv of this (data, Z, theta) stay in the workspaces: a posterior request that follows without
    // any change of state is served from them (GPy's inference returns the posterior with every evaluation)
    c->grad_serial = c->state_serial;
    c->grad_lml = out->lml;
    c->grad_M = M;
    c->grad_Mr = Mr;
    return EDRGP_OK;
}

// Posterior (woodbury vector / inverse) out of the workspaces of the bound+gradient evaluation that has just run
// for the same state.
static int posterior_from_last_gradient(Ctx* c) {
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

struct DenseOut {
    double lml = 0.0;
    std::vector<double> dvar, dlen;
    double dnoise = 0.0;
};

// ExactGaussianInference.inference on the resident (X, y).
static int dense_eval(Ctx* c, bool need_grad, bool need_post, DenseOut* out) {
    if (!c->have_data || !c->have_kernel) {
        set_error("dense evaluation needs set_data and set_kernel first");
        return EDRGP_ERR_STATE;
    }
    if (c->comm) { set_error("the dense exact GP does not shard: replicas only"); return EDRGP_ERR_STATE; }
    c->grad_serial = 0;                 // shares the M x M workspaces with the sparse path
    if (c->N > 46000) { set_error("dense exact GP limited to N <= 46000"); return EDRGP_ERR_INVALID; }
    const int n = (int)c->N, D = c->D, np = c->nparts, Wd = 2 * D + 1;
    const KSpec ks = make_kspec(c);
    TRY(ensure_mats(c, n, need_grad ? 5 : 3));
    double* L = c->mats[0].d();
    double* Li = c->mats[1].d();
    double* Kyi = c->mats[2].d();
    double* dK = c->mats[3].d();
    double* Tn = c->mats[4].d();
    double* sc = c->scal.d();
    int* info = static_cast<int*>(c->info.p);
    double* alpha = c->tmpvec[0].d();
    const double diag_mean_ = var_sum(c) + c->noise_var + 1e-8;
    int lvl = -1;
    double h_sc[SC_COUNT];
    int h_info[4];
    for (;;) {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        HIP_TRY(hipMemsetAsync(sc, 0, SC_COUNT * sizeof(double), c->stream));
        TRY(kernel_matrix(c, ks, c->X.d(), n, c->X.d(), n, L, n, 1,
                          c->noise_var + 1e-8 + jit_amount(lvl, diag_mean_), nullptr, nullptr));
        TRY(potrf_lower(c, L, n, n, info));
        TRY(trtri_lower(c, L, n, Li, n, n));
        TRY(gram_of_lower(c, n, Li, Kyi));
        TRY(gemv_n(c, n, n, 1.0, Kyi, n, c->y.d(), alpha));
        TRY(sum_log_diag(c, L, n, n, sc + SC_SUMLOGL));
        TRY(dot(c, c->y.d(), alpha, n, sc + SC_YALPHA));
        TRY(sum_sq(c, alpha, n, sc + SC_ALPHASQ));
        TRY(trace_of(c, Kyi, n, n, sc + SC_TRKYI));
        HIP_TRY(hipMemcpyAsync(h_sc, sc, sizeof(h_sc), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h_info, info, sizeof(h_info), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (h_info[0] != 0) {
            if (++lvl > 4) { set_error("K + noise not positive definite, even with jitter"); return EDRGP_ERR_NOT_PD; }
            continue;
        }
        break;
    }
    c->drain_events();
    out->lml = 0.5 * (-(double)n * std::log(2.0 * M_PI) - 2.0 * h_sc[SC_SUMLOGL] - h_sc[SC_YALPHA]);
    if (need_post) {
        TRY(c->wv.reserve((size_t)n * sizeof(double)));
        TRY(c->winv.reserve((size_t)n * n * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(c->wv.d(), alpha, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->winv.d(), Kyi, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        c->have_post = true;
        c->post_dense = true;
        c->P = n;
    }
    if (!need_grad) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        return EDRGP_OK;
    }
    // dL/dK = 1/2 (alpha alpha^T - Ky^-1)
    TRY(combine3(c, n, -0.5, Kyi, 0.0, nullptr, 0.0, nullptr, 0.5, alpha, dK));
    TRY(c->gt_nm.reserve((size_t)np * n * Wd * sizeof(double)));
    for (int p = 0; p < np; ++p) {
        TRY(grad_elementwise(c, ks, p, c->X.d(), nullptr, n, c->X.d(), n, dK, n, 0.0, nullptr, Tn, n,
                             sc + SC_DVAR_NM + p));
        TRY(gemm(c, true, false, n, Wd, n, 1.0, Tn, n, c->Xt.d(), Wd, 0.0, c->gt_nm.d() + (size_t)p * n * Wd, Wd));
    }
    std::vector<double> g((size_t)np * n * Wd), Xh((size_t)n * D);
    HIP_TRY(hipMemcpyAsync(h_sc, sc, sizeof(h_sc), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(g.data(), c->gt_nm.d(), g.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(Xh.data(), c->X.d(), Xh.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out->dvar.assign(np, 0.0);
    out->dlen.assign((size_t)np * D, 0.0);
    for (int p = 0; p < np; ++p) {
        out->dvar[p] = h_sc[SC_DVAR_NM + p];
        const double* G = &g[(size_t)p * n * Wd];
        for (int q = 0; q < D; ++q) {
            const double l = c->lens[(size_t)p * D + q];
            const double il3 = 1.0 / (l * l * l);
            double s = 0.0;
            for (int m = 0; m < n; ++m) {
                const double x = Xh[(size_t)m * D + q];
                const double* gm = G + (size_t)m * Wd;
                s += gm[D + 1 + q] + gm[D] * x * x - 2.0 * x * gm[q];
            }
            out->dlen[(size_t)p * D + q] = -il3 * s;
        }
    }
    out->dnoise = 0.5 * (h_sc[SC_ALPHASQ] - h_sc[SC_TRKYI]);
    return EDRGP_OK;
}

// Prediction on a device-resident block of inputs; outputs to host pointers.
__global__ void fill_ones_kernel(double* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 1.0;
}

// gram_h / gsum_h (optional): the D x D outer-product matrix J^T J of the posterior-mean gradients and
// their column sums, accumulated on the device chunk by chunk -- what the EDR step and the candidate
// sweep (BASELINE config 4) consume -- so the N x D gradient matrix need not cross PCIe.
static int predict_rows(Ctx* c, const double* Xs_dev, int64_t Ns, double* mean_h, double* var_h, double* jac_h,
                        double* gram_h = nullptr, double* gsum_h = nullptr) {
    if (!c->have_post) { set_error("no posterior: call *_posterior first"); return EDRGP_ERR_STATE; }
    const int D = c->D, Mp = c->P, Wd = 2 * D + 1;
    const KSpec ks = make_kspec(c);
    const double* Pp = c->post_dense ? c->X.d() : c->Z.d();
    const double* Pt = c->post_dense ? c->Xt.d() : c->Zt.d();
    int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(c->chunk_rows, (int64_t)(1LL << 28) / std::max(1, Mp)));
    chunk = std::min<int64_t>(chunk, Ns);
    TRY(c->Knm.reserve((size_t)chunk * Mp * sizeof(double)));
    TRY(c->W.reserve((size_t)chunk * Mp * sizeof(double)));
    TRY(c->T.reserve((size_t)chunk * (Wd + D + 2) * sizeof(double)));
    double* GT = c->T.d();
    double* o_jac = GT + (size_t)chunk * Wd;
    double* o_mean = o_jac + (size_t)chunk * D;
    double* o_var = o_mean + chunk;
    const double kdiag = var_sum(c);
    // the training rows right after a sparse evaluation with the same theta / Z: K(X, Z) is still resident
    const bool k_train = c->knm_train_valid && !c->post_dense && Xs_dev == c->X.d() && Ns == c->N && Mp == c->M;
    if (!k_train) c->knm_train_valid = false;          // Knm is about to be overwritten chunk by chunk
    const bool want_gram = (gram_h != nullptr);
    double* d_gram = nullptr;
    double* d_ones = nullptr;
    if (want_gram) {
        TRY(c->g2.reserve(((size_t)D * D + D + (size_t)chunk + 8) * sizeof(double)));
        d_gram = c->g2.d();
        d_ones = d_gram + (size_t)D * D + D;
        hipLaunchKernelGGL(fill_ones_kernel, dim3((unsigned)((chunk + 255) / 256)), dim3(256), 0, c->stream, d_ones, chunk);
    }
    for (int64_t r0 = 0; r0 < Ns; r0 += chunk) {
        const int64_t rows = std::min<int64_t>(chunk, Ns - r0);
        TRY(predict_chunk(c, ks, Xs_dev + (size_t)r0 * D, rows, Pp, Pt, Mp, c->wv.d(), c->winv.d(), kdiag,
                          k_train ? c->Knm.d() + (size_t)r0 * Mp : c->Knm.d(), c->W.d(), GT, mean_h ? o_mean : nullptr,
                          var_h ? o_var : nullptr, (jac_h || want_gram) ? o_jac : nullptr, k_train));
        if (want_gram) {
            TRY(gemm(c, true, false, D, D, (int)rows, 1.0, o_jac, D, o_jac, D, r0 > 0 ? 1.0 : 0.0, d_gram, D));
            TRY(gemm(c, true, false, D, 1, (int)rows, 1.0, o_jac, D, d_ones, 1, r0 > 0 ? 1.0 : 0.0, d_gram + (size_t)D * D, 1));
        }
        if (mean_h) HIP_TRY(hipMemcpyAsync(mean_h + r0, o_mean, rows * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (var_h) HIP_TRY(hipMemcpyAsync(var_h + r0, o_var, rows * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (jac_h)
            HIP_TRY(hipMemcpyAsync(jac_h + (size_t)r0 * D, o_jac, (size_t)rows * D * sizeof(double),
                                   hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (want_gram) {
        HIP_TRY(hipMemcpyAsync(gram_h, d_gram, (size_t)D * D * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (gsum_h) HIP_TRY(hipMemcpyAsync(gsum_h, d_gram + (size_t)D * D, (size_t)D * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return EDRGP_OK;
}

}  // namespace edrgp

// ===========================================================================
using namespace edrgp;

extern "C" {

const char* edrgp_last_error(void) { return g_err.c_str(); }
int edrgp_version(void) { return 100; }

int edrgp_device_count(int* count) {
    if (!count) return EDRGP_ERR_INVALID;
    HIP_TRY(hipGetDeviceCount(count));
    return EDRGP_OK;
}

int edrgp_ctx_create(int device, edrgp_ctx** ctx) {
    if (!ctx) return EDRGP_ERR_INVALID;
    *ctx = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0 || device < 0 || device >= n) {
        set_error("no HIP device %d (device count %d): the MI355X path has no CPU fallback", device, n);
        return EDRGP_ERR_HIP;
    }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return EDRGP_ERR_HIP;
    }
    Ctx* c = new Ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        // side stream at the highest priority: its short, latency-bound kernels should
        // take the next free workgroup slot ahead of the row pass' queue
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    *ctx = reinterpret_cast<edrgp_ctx*>(c);
    return EDRGP_OK;
}

int edrgp_ctx_destroy(edrgp_ctx* ctx) {
    if (!ctx) return EDRGP_OK;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream2);
    c->drain_events();
    if (c->comm) edrgp_comm_destroy(ctx);
    DBuf* bufs[] = {&c->ptw, &c->Xbase, &c->X, &c->y, &c->Z, &c->Xt, &c->Zt, &c->Xq, &c->g2, &c->Knm32, &c->T32, &c->S2f, &c->Xq32,
                    &c->ks32_x, &c->ks32_z, &c->ks32_xn, &c->ks32_zn, &c->invl, &c->wv, &c->winv, &c->Knm, &c->W, &c->T, &c->slabs,
                    &c->bpart, &c->scal, &c->info, &c->gt_nm, &c->gt_mm, &c->panel, &c->trtri_ws, &c->dinv, &c->red, &c->lbaug, &c->xs_in};
    for (DBuf* b : bufs) b->release();
    for (auto& m : c->mats) m.release();
    for (auto& v : c->tmpvec) v.release();
    c->red2.release();
    c->slabs2.release();
    for (DBuf* b : {&c->ks_x, &c->ks_z, &c->ks_xn, &c->ks_zn, &c->ks_x2, &c->ks_xn2}) b->release();
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->stream_mm) (void)hipStreamDestroy(c->stream_mm);
    if (c->stream_ms) (void)hipStreamDestroy(c->stream_ms);
    (void)hipStreamDestroy(c->stream2);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return EDRGP_OK;
}

#define CTX(ctx)                                                   \
    Ctx* c = reinterpret_cast<Ctx*>(ctx);                          \
    if (!c) { set_error("null context"); return EDRGP_ERR_INVALID; } \
    HIP_TRY(hipSetDevice(c->device))

int edrgp_sync(edrgp_ctx* ctx) {
    CTX(ctx);
    HIP_TRY(hipStreamSynchronize(c->stream2));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->drain_events();
    return EDRGP_OK;
}

int edrgp_set_option(edrgp_ctx* ctx, const char* key, double value) {
    CTX(ctx);
    ++c->state_serial;
    if (!key) return EDRGP_ERR_INVALID;
    if (!std::strcmp(key, "chunk_rows")) c->chunk_rows = std::max<int64_t>(64, (int64_t)value);
    else if (!std::strcmp(key, "syrk_splits")) c->syrk_splits = (int)value;
    else if (!std::strcmp(key, "profile")) c->profile = (value != 0.0);
    else if (!std::strcmp(key, "stats_mode")) c->stats_mode = (int)value;          // 0 auto, 1 gram, 2 stable
    else if (!std::strcmp(key, "pivot_ratio_max")) c->pivot_ratio_max = value;
    else if (!std::strcmp(key, "overlap")) c->overlap = (int)value;
    else if (!std::strcmp(key, "trace")) c->trace = (value != 0.0);
    else if (!std::strcmp(key, "gemm_dephase")) c->gemm_dephase = (int)value;
    else if (!std::strcmp(key, "fuse_grad")) c->fuse_grad = (value != 0.0);
    else if (!std::strcmp(key, "tri_tile")) g_tri_tile = (int)value;
    else if (!std::strcmp(key, "mfma_kbuild")) c->mfma_kbuild = (value != 0.0);
    else if (!std::strcmp(key, "kbuild_dbg")) c->kbuild_dbg = (int)value;
    else if (!std::strcmp(key, "kbuild_tile")) c->kbuild_tile = (int)value;
    else if (!std::strcmp(key, "rbf_epilogue")) c->rbf_epilogue = (value != 0.0);
    else if (!std::strcmp(key, "chol_fused")) c->chol_fused = (value != 0.0);
    else if (!std::strcmp(key, "chol_rhs")) c->chol_rhs = (value != 0.0);
    else if (!std::strcmp(key, "pad_inducing")) c->pad_inducing = (value != 0.0);
    else if (!std::strcmp(key, "gemm_pair")) c->gemm_pair = (value != 0.0);
    else if (!std::strcmp(key, "posterior_reuse")) c->posterior_reuse = (value != 0.0);
    else if (!std::strcmp(key, "prebuild")) c->prebuild = (value != 0.0);
    else if (!std::strcmp(key, "syrk_xmap")) c->syrk_xmap = (value != 0.0);
    else if (!std::strcmp(key, "syrk_diag_opt")) c->syrk_diag_opt = (value != 0.0);
    else if (!std::strcmp(key, "mask_layout")) c->mask_layout = (int)value;
    else if (!std::strcmp(key, "side_cus")) {
        // CUs reserved for the side stream of bound-only evaluations (0 = plain streams); takes effect
        // before the masked streams are first created
        if (c->masked_state == 0 || (int)value == 0) c->side_cus = (int)value;
    }
    else if (!std::strcmp(key, "f32_ratio_max")) c->f32_ratio_max = value;
    else if (!std::strcmp(key, "f32_guard")) c->f32_guard = value;
    else if (!std::strcmp(key, "f32_phi")) { c->f32_phi = (value != 0.0); c->phi32_bad = false; }
    else if (!std::strcmp(key, "precision")) {
        // 64 (default): fp64 throughout.  32: the N-proportional row pass in fp32 (la_gemm32.hip)
        if ((int)value != 32 && (int)value != 64) { set_error("precision must be 32 or 64"); return EDRGP_ERR_INVALID; }
        c->precision = (int)value;
        c->have_post = false;
        c->last_mode = 0;
        c->phi32_bad = false;
    }
    else { set_error("unknown option %s", key); return EDRGP_ERR_INVALID; }
    return EDRGP_OK;
}

int edrgp_set_data(edrgp_ctx* ctx, const double* X, const double* y, int64_t N, int D) {
    CTX(ctx);
    ++c->state_serial;
    if (!X || !y || N <= 0 || D <= 0 || D > EDRGP_MAX_DIM) { set_error("set_data: bad shape N=%lld D=%d", (long long)N, D); return EDRGP_ERR_INVALID; }
    if (c->have_Z && c->D != D) c->have_Z = false;
    if (c->have_kernel && c->D != D) c->have_kernel = false;
    c->have_post = false;
    TRY(c->X.reserve((size_t)N * D * sizeof(double)));
    TRY(c->y.reserve((size_t)N * sizeof(double)));
    TRY(c->Xt.reserve((size_t)N * (2 * D + 1) * sizeof(double)));
    TRY(c->Xq.reserve((size_t)N * 2 * D * sizeof(double)));
    TRY(c->scal.reserve(SC_COUNT * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->X.d(), X, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->y.d(), y, (size_t)N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->N = N;
    c->D = D;
    c->xq32_valid = false;
    c->phi32_bad = false;
    c->glob_valid = false;
    c->knm_train_valid = false;
    TRY(build_xtilde(c, c->X.d(), N, D, c->Xt.d()));
    TRY(build_xq(c, c->X.d(), N, D, c->Xq.d()));
    TRY(sum_sq(c, c->y.d(), N, c->scal.d() + SC_YY));
    double yy = 0.0;
    HIP_TRY(hipMemcpyAsync(&yy, c->scal.d() + SC_YY, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->yy_local = yy;
    c->have_data = true;
    return EDRGP_OK;
}

// Training inputs as a projection of resident base rows: X = Xbase W (N x d), computed on the device.
// Xbase (N x Dbase) and y are uploaded when given and stay resident for later calls with Xbase == NULL
// (candidate projections of the sweep / the EDR loop's X C^T, edrgp/base.py:459-517): a new candidate
// costs the upload of W and one N x d x Dbase product instead of a host product and an N x d upload.
int edrgp_set_data_projected(edrgp_ctx* ctx, const double* Xbase, const double* y, int64_t N, int Dbase,
                             const double* W, int d) {
    CTX(ctx);
    ++c->state_serial;
    if (!W || d <= 0 || d > EDRGP_MAX_DIM || Dbase <= 0 || N <= 0) { set_error("set_data_projected: bad shape"); return EDRGP_ERR_INVALID; }
    if (Xbase != nullptr) {
        if (!y) { set_error("set_data_projected: y is required with Xbase"); return EDRGP_ERR_INVALID; }
        TRY(c->Xbase.reserve((size_t)N * Dbase * sizeof(double)));
        TRY(c->y.reserve((size_t)N * sizeof(double)));
        TRY(c->scal.reserve(SC_COUNT * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(c->Xbase.d(), Xbase, (size_t)N * Dbase * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->y.d(), y, (size_t)N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        TRY(sum_sq(c, c->y.d(), N, c->scal.d() + SC_YY));
        double yy = 0.0;
        HIP_TRY(hipMemcpyAsync(&yy, c->scal.d() + SC_YY, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->yy_local = yy;
        c->glob_valid = false;
        c->base_N = N;
        c->base_D = Dbase;
    } else if (c->base_N != N || c->base_D != Dbase) {
        set_error("set_data_projected: no resident base rows of shape %lld x %d", (long long)N, Dbase);
        return EDRGP_ERR_STATE;
    }
    if (c->have_Z && c->D != d) c->have_Z = false;
    if (c->have_kernel && c->D != d) c->have_kernel = false;
    c->have_post = false;
    TRY(c->X.reserve((size_t)N * d * sizeof(double)));
    TRY(c->Xt.reserve((size_t)N * (2 * d + 1) * sizeof(double)));
    TRY(c->Xq.reserve((size_t)N * 2 * d * sizeof(double)));
    TRY(c->tmpvec[5].reserve((size_t)Dbase * d * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->tmpvec[5].d(), W, (size_t)Dbase * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (N > 0x7fffffffLL) { set_error("set_data_projected: N too large"); return EDRGP_ERR_INVALID; }
    TRY(gemm(c, false, false, (int)N, d, Dbase, 1.0, c->Xbase.d(), Dbase, c->tmpvec[5].d(), d, 0.0, c->X.d(), d));
    c->N = N;
    c->D = d;
    c->knm_train_valid = false;
    c->xq32_valid = false;
    c->phi32_bad = false;
    TRY(build_xtilde(c, c->X.d(), N, d, c->Xt.d()));
    TRY(build_xq(c, c->X.d(), N, d, c->Xq.d()));
    HIP_TRY(hipStreamSynchronize(c->stream));      // W is a borrowed host pointer
    c->have_data = true;
    return EDRGP_OK;
}

int edrgp_set_inducing(edrgp_ctx* ctx, const double* Z, int M) {
    CTX(ctx);
    ++c->state_serial;
    if (!Z || M <= 0) { set_error("set_inducing: bad M=%d", M); return EDRGP_ERR_INVALID; }
    if (!c->have_data) { set_error("set_inducing: call set_data first (fixes D)"); return EDRGP_ERR_STATE; }
    const int D = c->D;
    // rows up to the next multiple of 128 stay zero: coordinates of the phantom points of a padded evaluation
    const size_t Mcap = ((size_t)M + 127) / 128 * 128;
    TRY(c->Z.reserve(Mcap * D * sizeof(double)));
    TRY(c->Zt.reserve(Mcap * (2 * D + 1) * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->Z.d(), 0, Mcap * D * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->Zt.d(), 0, Mcap * (2 * D + 1) * sizeof(double), c->stream));
    HIP_TRY(hipMemcpyAsync(c->Z.d(), Z, (size_t)M * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->Z_host.assign(Z, Z + (size_t)M * D);
    TRY(build_xtilde(c, c->Z.d(), M, D, c->Zt.d()));
    HIP_TRY(hipStreamSynchronize(c->stream));   // Z is a borrowed host pointer
    if (c->M != M) c->phi32_bad = false;
    c->M_work = 0;
    c->knm_train_valid = false;
    c->M = M;
    c->have_Z = true;
    if (!c->post_dense) c->have_post = false;
    return EDRGP_OK;
}

int edrgp_set_kernel(edrgp_ctx* ctx, int nparts, const int* kinds, const double* variances,
                     const double* lengthscales, double noise_var) {
    CTX(ctx);
    ++c->state_serial;
    if (!c->have_data) { set_error("set_kernel: call set_data first (fixes D)"); return EDRGP_ERR_STATE; }
    if (nparts < 1 || nparts > EDRGP_MAX_PARTS || !kinds || !variances || !lengthscales) {
        set_error("set_kernel: nparts must be 1..%d", EDRGP_MAX_PARTS);
        return EDRGP_ERR_INVALID;
    }
    const int D = c->D;
    std::vector<double> invl((size_t)nparts * D);
    for (int p = 0; p < nparts; ++p) {
        if (kinds[p] < 0 || kinds[p] > 2) { set_error("set_kernel: unknown kind %d", kinds[p]); return EDRGP_ERR_INVALID; }
        if (!(variances[p] > 0.0)) { set_error("set_kernel: variance must be positive"); return EDRGP_ERR_INVALID; }
        for (int q = 0; q < D; ++q) {
            const double l = lengthscales[(size_t)p * D + q];
            if (!(l > 0.0)) { set_error("set_kernel: lengthscale must be positive"); return EDRGP_ERR_INVALID; }
            invl[(size_t)p * D + q] = 1.0 / l;
        }
    }
    if (!(noise_var >= 0.0)) { set_error("set_kernel: noise variance must be non-negative"); return EDRGP_ERR_INVALID; }
    c->nparts = nparts;
    for (int p = 0; p < nparts; ++p) { c->kind[p] = kinds[p]; c->var[p] = variances[p]; }
    c->lens.assign(lengthscales, lengthscales + (size_t)nparts * D);
    c->noise_var = noise_var;
    TRY(c->invl.reserve((size_t)EDRGP_MAX_PARTS * EDRGP_MAX_DIM * sizeof(double)));
    // synchronous copy: the staging vector dies at return, and kernels queued
    // earlier on the stream may still read the previous values
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->invl.d(), invl.data(), invl.size() * sizeof(double), hipMemcpyHostToDevice));
    c->have_kernel = true;
    c->have_post = false;
    c->knm_train_valid = false;
    return EDRGP_OK;
}

int edrgp_sparse_lml(edrgp_ctx* ctx, double* lml) {
    CTX(ctx);
    SparseOut o;
    TRY(sparse_eval(c, false, false, &o));
    if (lml) *lml = o.lml;
    return EDRGP_OK;
}

int edrgp_sparse_lml_grad(edrgp_ctx* ctx, double* lml, double* dvar, double* dlen, double* dnoise, double* dZ) {
    CTX(ctx);
    SparseOut o;
    TRY(sparse_eval(c, true, false, &o));
    if (lml) *lml = o.lml;
    if (dvar) std::memcpy(dvar, o.dvar.data(), o.dvar.size() * sizeof(double));
    if (dlen) std::memcpy(dlen, o.dlen.data(), o.dlen.size() * sizeof(double));
    if (dnoise) *dnoise = o.dnoise;
    if (dZ) std::memcpy(dZ, o.dZ.data(), o.dZ.size() * sizeof(double));
    return EDRGP_OK;
}

int edrgp_sparse_posterior(edrgp_ctx* ctx, double* lml) {
    CTX(ctx);
    if (c->posterior_reuse && c->grad_serial != 0 && c->grad_serial == c->state_serial) {
        TRY(posterior_from_last_gradient(c));
        if (lml) *lml = c->grad_lml;
        return EDRGP_OK;
    }
    SparseOut o;
    TRY(sparse_eval(c, false, true, &o));
    if (lml) *lml = o.lml;
    return EDRGP_OK;
}

int edrgp_dense_lml_grad(edrgp_ctx* ctx, double* lml, double* dvar, double* dlen, double* dnoise) {
    CTX(ctx);
    DenseOut o;
    const bool need_grad = (dvar || dlen || dnoise);
    TRY(dense_eval(c, need_grad, false, &o));
    if (lml) *lml = o.lml;
    if (need_grad) {
        if (dvar) std::memcpy(dvar, o.dvar.data(), o.dvar.size() * sizeof(double));
        if (dlen) std::memcpy(dlen, o.dlen.data(), o.dlen.size() * sizeof(double));
        if (dnoise) *dnoise = o.dnoise;
    }
    return EDRGP_OK;
}

int edrgp_dense_posterior(edrgp_ctx* ctx, double* lml) {
    CTX(ctx);
    DenseOut o;
    TRY(dense_eval(c, false, true, &o));
    if (lml) *lml = o.lml;
    return EDRGP_OK;
}

// test inputs go through a grow-only buffer owned by the context (no hipMalloc / hipFree per call)
static int upload_inputs(Ctx* c, const double* Xs, int64_t Ns) {
    if (!c->have_data) { set_error("predict: no data (the input dimension is fixed by set_data)"); return EDRGP_ERR_STATE; }
    TRY(c->xs_in.reserve((size_t)Ns * c->D * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->xs_in.d(), Xs, (size_t)Ns * c->D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return EDRGP_OK;
}

int edrgp_predict(edrgp_ctx* ctx, const double* Xs, int64_t Ns, double* mean, double* var) {
    CTX(ctx);
    if (!Xs || Ns <= 0) { set_error("predict: bad input"); return EDRGP_ERR_INVALID; }
    TRY(upload_inputs(c, Xs, Ns));
    const int s = predict_rows(c, c->xs_in.d(), Ns, mean, var, nullptr);
    (void)hipStreamSynchronize(c->stream);          // Xs is a borrowed host pointer
    return s;
}

int edrgp_predict_grad(edrgp_ctx* ctx, const double* Xs, int64_t Ns, double* grad) {
    CTX(ctx);
    if (!Xs || Ns <= 0 || !grad) { set_error("predict_grad: bad input"); return EDRGP_ERR_INVALID; }
    TRY(upload_inputs(c, Xs, Ns));
    const int s = predict_rows(c, c->xs_in.d(), Ns, nullptr, nullptr, grad);
    (void)hipStreamSynchronize(c->stream);
    return s;
}

int edrgp_predict_train(edrgp_ctx* ctx, double* mean, double* grad) {
    CTX(ctx);
    if (!c->have_data) { set_error("predict_train: no data"); return EDRGP_ERR_STATE; }
    return predict_rows(c, c->X.d(), c->N, mean, nullptr, grad);
}

int edrgp_predict_train_gram(edrgp_ctx* ctx, double* mean, double* gram, double* gsum) {
    CTX(ctx);
    if (!c->have_data) { set_error("predict_train_gram: no data"); return EDRGP_ERR_STATE; }
    if (!gram) { set_error("predict_train_gram: gram is required"); return EDRGP_ERR_INVALID; }
    return predict_rows(c, c->X.d(), c->N, mean, nullptr, nullptr, gram, gsum);
}

int edrgp_get(edrgp_ctx* ctx, int which, double* buf, int64_t cap) {
    CTX(ctx);
    if (!buf) return EDRGP_ERR_INVALID;
    const size_t M = (size_t)c->M;
    // the last sparse evaluation may have worked on a padded inducing set (ld Mw): the caller gets the real block
    const size_t Mw = (c->M_work >= c->M) ? (size_t)c->M_work : M;
    const double* src = nullptr;
    size_t n = 0;
    size_t rows2d = 0;          // > 0: rows2d x M block of an ld-Mw matrix
    switch (which) {
        case EDRGP_GET_PHI: src = c->mats[MAT_PHI].d(); n = M * M; rows2d = M; break;
        case EDRGP_GET_BVEC: src = c->mats[MAT_PHI].d() ? c->mats[MAT_PHI].d() + Mw * Mw : nullptr; n = M; break;
        case EDRGP_GET_LM: case EDRGP_GET_KMM: src = c->mats[MAT_LM].d(); n = M * M; rows2d = M; break;
        case EDRGP_GET_LB: src = c->mats[MAT_LB].d(); n = M * M; rows2d = M; break;
        case EDRGP_GET_A: src = c->mats[MAT_A].d(); n = M * M; rows2d = M; break;
        case EDRGP_GET_WV: src = c->wv.d(); n = (size_t)c->P; break;
        case EDRGP_GET_WINV: src = c->winv.d(); n = (size_t)c->P * c->P; break;
        case EDRGP_GET_KNM:
            rows2d = (size_t)std::min<int64_t>(c->N, c->chunk_rows);
            n = rows2d * M;
            if (c->last_mode == 3) {          // fp32 row pass: widen the resident chunk
                if ((int64_t)n > cap) { set_error("edrgp_get: buffer too small"); return EDRGP_ERR_INVALID; }
                std::vector<float> tmp(n);
                HIP_TRY(hipStreamSynchronize(c->stream));
                HIP_TRY(hipMemcpy(tmp.data(), c->Knm32.p, n * sizeof(float), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < n; ++i) buf[i] = (double)tmp[i];
                return EDRGP_OK;
            }
            src = c->Knm.d();
            break;
        case EDRGP_GET_LMI: src = c->mats[MAT_LI].d(); n = M * M; rows2d = M; break;
        case EDRGP_GET_LBI: src = c->mats[MAT_LBI].d(); n = M * M; rows2d = M; break;
        case EDRGP_GET_DLDKMM: src = c->mats[MAT_DKMM].d(); n = M * M; rows2d = M; break;
        case EDRGP_GET_SCALARS: src = c->scal.d(); n = 16; break;
        case EDRGP_GET_MODE:
            buf[0] = (double)c->last_mode;
            if (cap >= 2) buf[1] = c->f32_rho;      // last fp32 guard estimate (0 if the fp32 pass never ran)
            if (cap >= 3) buf[2] = c->last_pivot_ratio2;   // (max / min diag Lm)^2 of the last evaluation
            return EDRGP_OK;
        default: set_error("edrgp_get: unknown selector %d", which); return EDRGP_ERR_INVALID;
    }
    if (!src) { set_error("edrgp_get: object %d not computed yet", which); return EDRGP_ERR_STATE; }
    if ((int64_t)n > cap) { set_error("edrgp_get: buffer too small (%lld < %zu)", (long long)cap, n); return EDRGP_ERR_INVALID; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (rows2d > 0 && Mw != M)
        HIP_TRY(hipMemcpy2D(buf, M * sizeof(double), src, Mw * sizeof(double), M * sizeof(double), rows2d, hipMemcpyDeviceToHost));
    else
        HIP_TRY(hipMemcpy(buf, src, n * sizeof(double), hipMemcpyDeviceToHost));
    return EDRGP_OK;
}

int edrgp_profile_reset(edrgp_ctx* ctx) {
    CTX(ctx);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->drain_events();
    for (auto& p : c->prof) p = ProfSlot();
    return EDRGP_OK;
}

int edrgp_profile_get(edrgp_ctx* ctx, int slot, double* total_ms, int64_t* launches) {
    CTX(ctx);
    if (slot < 0 || slot >= EDRGP_PROF_NSLOTS) return EDRGP_ERR_INVALID;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->drain_events();
    if (total_ms) *total_ms = c->prof[slot].total_ms;
    if (launches) *launches = c->prof[slot].launches;
    return EDRGP_OK;
}

// ------------------------------------------------------------- debug hooks
int edrgp_dbg_gemm(edrgp_ctx* ctx, int transA, int transB, int M, int N, int K, double alpha, const double* A,
                   int lda, const double* B, int ldb, double beta, double* C, int ldc) {
    CTX(ctx);
    const size_t ra = transA ? K : M, rb = transB ? N : K;
    DBuf dA, dB, dC;
    TRY(dA.reserve(ra * lda * sizeof(double)));
    TRY(dB.reserve(rb * ldb * sizeof(double)));
    TRY(dC.reserve((size_t)M * ldc * sizeof(double)));
    HIP_TRY(hipMemcpy(dA.d(), A, ra * lda * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dB.d(), B, rb * ldb * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dC.d(), C, (size_t)M * ldc * sizeof(double), hipMemcpyHostToDevice));
    int s = gemm(c, transA != 0, transB != 0, M, N, K, alpha, dA.d(), lda, dB.d(), ldb, beta, dC.d(), ldc);
    if (s == EDRGP_OK && hipStreamSynchronize(c->stream) != hipSuccess) { set_error("dbg_gemm: kernel failed"); s = EDRGP_ERR_HIP; }
    if (s == EDRGP_OK) HIP_TRY(hipMemcpy(C, dC.d(), (size_t)M * ldc * sizeof(double), hipMemcpyDeviceToHost));
    dA.release(); dB.release(); dC.release();
    return s;
}

// fp32 building blocks on host float buffers: C = op(A) op(B) (fp32 MFMA), and Phi = K^T K with fp64 slab sums
int edrgp_dbg_gemm32(edrgp_ctx* ctx, int transA, int transB, int M, int N, int K, const float* A, int lda,
                     const float* B, int ldb, float* C, int ldc) {
    CTX(ctx);
    const size_t ra = transA ? K : M, rb = transB ? N : K;
    DBuf dA, dB, dC;
    TRY(dA.reserve(ra * lda * sizeof(float)));
    TRY(dB.reserve(rb * ldb * sizeof(float)));
    TRY(dC.reserve((size_t)M * ldc * sizeof(float)));
    HIP_TRY(hipMemcpy(dA.p, A, ra * lda * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dB.p, B, rb * ldb * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dC.p, 0, (size_t)M * ldc * sizeof(float)));
    int s = gemm32(c, transA != 0, transB != 0, M, N, K, static_cast<const float*>(dA.p), lda,
                   static_cast<const float*>(dB.p), ldb, static_cast<float*>(dC.p), ldc);
    if (s == EDRGP_OK && hipStreamSynchronize(c->stream) != hipSuccess) { set_error("dbg_gemm32: kernel failed"); s = EDRGP_ERR_HIP; }
    if (s == EDRGP_OK) HIP_TRY(hipMemcpy(C, dC.p, (size_t)M * ldc * sizeof(float), hipMemcpyDeviceToHost));
    dA.release(); dB.release(); dC.release();
    return s;
}

int edrgp_dbg_syrk32(edrgp_ctx* ctx, const float* K, int64_t N, int M, double* Phi) {
    CTX(ctx);
    if (!K || !Phi || N <= 0 || M <= 0) { set_error("dbg_syrk32: bad input"); return EDRGP_ERR_INVALID; }
    DBuf dK, dP;
    TRY(dK.reserve((size_t)N * M * sizeof(float)));
    TRY(dP.reserve((size_t)M * M * sizeof(double)));
    HIP_TRY(hipMemcpy(dK.p, K, (size_t)N * M * sizeof(float), hipMemcpyHostToDevice));
    int s = syrk32_tn(c, static_cast<const float*>(dK.p), N, M, M, dP.d(), M, false);
    if (s == EDRGP_OK && hipStreamSynchronize(c->stream) != hipSuccess) { set_error("dbg_syrk32: kernel failed"); s = EDRGP_ERR_HIP; }
    if (s == EDRGP_OK) HIP_TRY(hipMemcpy(Phi, dP.d(), (size_t)M * M * sizeof(double), hipMemcpyDeviceToHost));
    dK.release(); dP.release();
    return s;
}

int edrgp_dbg_potrf(edrgp_ctx* ctx, double* A, int n, int* info) {
    CTX(ctx);
    DBuf dA;
    TRY(dA.reserve((size_t)n * n * sizeof(double)));
    TRY(c->info.reserve(4 * sizeof(int)));
    HIP_TRY(hipMemcpy(dA.d(), A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
    int s = potrf_lower(c, dA.d(), n, n, static_cast<int*>(c->info.p));
    if (s == EDRGP_OK && hipStreamSynchronize(c->stream) != hipSuccess) { set_error("dbg_potrf: kernel failed"); s = EDRGP_ERR_HIP; }
    if (s == EDRGP_OK) {
        HIP_TRY(hipMemcpy(A, dA.d(), (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
        int h = 0;
        HIP_TRY(hipMemcpy(&h, c->info.p, sizeof(int), hipMemcpyDeviceToHost));
        if (info) *info = h;
    }
    dA.release();
    return s;
}

// diagnostic: phase stamps (shader cycles) of one diagonal-block kernel launch
int edrgp_dbg_diag_stamps(edrgp_ctx* ctx, const double* A64x64, int do_chol, double* stamps6) {
    CTX(ctx);
    DBuf dA, dD, dS;
    TRY(dA.reserve(64 * 64 * sizeof(double)));
    TRY(dD.reserve(64 * 64 * sizeof(double)));
    TRY(dS.reserve(8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpy(dA.d(), A64x64, 64 * 64 * sizeof(double), hipMemcpyHostToDevice));
    int s = diag_block_stamps(c, dA.d(), dD.d(), static_cast<unsigned long long*>(dS.p), do_chol);
    if (s == EDRGP_OK && hipStreamSynchronize(c->stream) != hipSuccess) s = EDRGP_ERR_HIP;
    unsigned long long h[8] = {0};
    if (s == EDRGP_OK) HIP_TRY(hipMemcpy(h, dS.p, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) stamps6[i] = (double)(h[i] - h[0]);
    dA.release(); dD.release(); dS.release();
    return s;
}

int edrgp_dbg_trtri(edrgp_ctx* ctx, const double* L, double* Linv, int n) {
    CTX(ctx);
    DBuf dL, dI;
    TRY(dL.reserve((size_t)n * n * sizeof(double)));
    TRY(dI.reserve((size_t)n * n * sizeof(double)));
    HIP_TRY(hipMemcpy(dL.d(), L, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
    int s = trtri_lower(c, dL.d(), n, dI.d(), n, n);
    if (s == EDRGP_OK && hipStreamSynchronize(c->stream) != hipSuccess) { set_error("dbg_trtri: kernel failed"); s = EDRGP_ERR_HIP; }
    if (s == EDRGP_OK) HIP_TRY(hipMemcpy(Linv, dI.d(), (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
    dL.release(); dI.release();
    return s;
}

int edrgp_dbg_syrk(edrgp_ctx* ctx, const double* K, int64_t N, int M, double* Phi) {
    CTX(ctx);
    DBuf dK, dP;
    TRY(dK.reserve((size_t)N * M * sizeof(double)));
    TRY(dP.reserve((size_t)M * M * sizeof(double)));
    HIP_TRY(hipMemcpy(dK.d(), K, (size_t)N * M * sizeof(double), hipMemcpyHostToDevice));
    int s = syrk_tn(c, dK.d(), N, M, M, dP.d(), M, false);
    if (s == EDRGP_OK && hipStreamSynchronize(c->stream) != hipSuccess) { set_error("dbg_syrk: kernel failed"); s = EDRGP_ERR_HIP; }
    c->drain_events();
    if (s == EDRGP_OK) HIP_TRY(hipMemcpy(Phi, dP.d(), (size_t)M * M * sizeof(double), hipMemcpyDeviceToHost));
    dK.release(); dP.release();
    return s;
}

// Times `reps` launches of the production Phi = K^T K kernel alone on a device-generated K
// (uniform [0,1) pattern; splits < 0: constant data).  Phi_out (M x M) optional.
int edrgp_dbg_syrk_time(edrgp_ctx* ctx, int64_t N, int M, int splits, int reps, double* avg_ms, double* Phi_out) {
    CTX(ctx);
    if (N <= 0 || N % 16 || M % 128 || reps < 1) { set_error("dbg_syrk_time: need N %% 16 == 0, M %% 128 == 0"); return EDRGP_ERR_INVALID; }
    TRY(c->Knm.reserve((size_t)N * M * sizeof(double)));
    TRY(c->mats[MAT_PHI].reserve(((size_t)M * M + M + 8) * sizeof(double)));
    c->knm_train_valid = false;
    c->grad_serial = 0;
    TRY(fill_pattern(c, c->Knm.d(), (size_t)N * M, splits < 0 ? 0ULL : 12345ULL));
    if (splits < 0) splits = 0;
    double* Phi = c->mats[MAT_PHI].d();
    const int saved = c->syrk_splits;
    c->syrk_splits = splits;
    int s = syrk_tn(c, c->Knm.d(), N, M, M, Phi, M, false);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s == EDRGP_OK) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, c->stream));
        for (int r = 0; r < reps && s == EDRGP_OK; ++r) s = syrk_tn(c, c->Knm.d(), N, M, M, Phi, M, false);
        HIP_TRY(hipEventRecord(e1, c->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (avg_ms) *avg_ms = ms / reps;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    c->syrk_splits = saved;
    c->drain_events();
    if (s == EDRGP_OK && Phi_out) HIP_TRY(hipMemcpy(Phi_out, Phi, (size_t)M * M * sizeof(double), hipMemcpyDeviceToHost));
    return s;
}

// kern.K(X, X2) of a kernel given by value (GPy kern.K): nothing resident is touched -- the
// training data, Z, the hyper-parameters and the posterior of the context stay as they are.
int edrgp_kernel_matrix(edrgp_ctx* ctx, int nparts, const int* kinds, const double* variances,
                        const double* lengthscales, const double* X, int64_t N, const double* X2, int64_t N2,
                        int D, double* K) {
    CTX(ctx);
    if (nparts < 1 || nparts > EDRGP_MAX_PARTS || !kinds || !variances || !lengthscales || !X || !K || N <= 0 ||
        D <= 0 || D > EDRGP_MAX_DIM || (X2 != nullptr && N2 <= 0)) {
        set_error("kernel_matrix: bad argument");
        return EDRGP_ERR_INVALID;
    }
    const bool same = (X2 == nullptr);
    const int64_t n2 = same ? N : N2;
    if (n2 > 0x7fffffffLL) { set_error("kernel_matrix: too many columns"); return EDRGP_ERR_INVALID; }
    KSpec ks;
    ks.nparts = nparts;
    ks.D = D;
    std::vector<double> invl((size_t)nparts * D);
    for (int p = 0; p < EDRGP_MAX_PARTS; ++p) { ks.kind[p] = 0; ks.var[p] = 0.0; }
    for (int p = 0; p < nparts; ++p) {
        if (kinds[p] < 0 || kinds[p] > 2 || !(variances[p] > 0.0)) { set_error("kernel_matrix: bad kernel part %d", p); return EDRGP_ERR_INVALID; }
        ks.kind[p] = kinds[p];
        ks.var[p] = variances[p];
        for (int q = 0; q < D; ++q) {
            const double l = lengthscales[(size_t)p * D + q];
            if (!(l > 0.0)) { set_error("kernel_matrix: lengthscale must be positive"); return EDRGP_ERR_INVALID; }
            invl[(size_t)p * D + q] = 1.0 / l;
        }
    }
    DBuf dX, dX2, dK, dL;
    TRY(dX.reserve((size_t)N * D * sizeof(double)));
    TRY(dK.reserve((size_t)N * n2 * sizeof(double)));
    TRY(dL.reserve(invl.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(dX.d(), X, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dL.d(), invl.data(), invl.size() * sizeof(double), hipMemcpyHostToDevice));
    if (!same) {
        TRY(dX2.reserve((size_t)N2 * D * sizeof(double)));
        HIP_TRY(hipMemcpy(dX2.d(), X2, (size_t)N2 * D * sizeof(double), hipMemcpyHostToDevice));
    }
    ks.invl = dL.d();
    TRY(kernel_matrix(c, ks, dX.d(), N, same ? dX.d() : dX2.d(), (int)n2, dK.d(), (int)n2, same ? 1 : 0, 0.0, nullptr, nullptr));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(K, dK.d(), (size_t)N * n2 * sizeof(double), hipMemcpyDeviceToHost));
    return EDRGP_OK;
}

int edrgp_dbg_mfma_layout(edrgp_ctx* ctx, const double* A16x4, const double* B4x16, double* C16x16) {
    CTX(ctx);
    DBuf dA, dB, dC;
    TRY(dA.reserve(64 * sizeof(double)));
    TRY(dB.reserve(64 * sizeof(double)));
    TRY(dC.reserve(256 * sizeof(double)));
    HIP_TRY(hipMemcpy(dA.d(), A16x4, 64 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dB.d(), B4x16, 64 * sizeof(double), hipMemcpyHostToDevice));
    int s = mfma_layout_probe(c, dA.d(), dB.d(), dC.d());
    if (s == EDRGP_OK && hipStreamSynchronize(c->stream) != hipSuccess) s = EDRGP_ERR_HIP;
    if (s == EDRGP_OK) HIP_TRY(hipMemcpy(C16x16, dC.d(), 256 * sizeof(double), hipMemcpyDeviceToHost));
    dA.release(); dB.release(); dC.release();
    return s;
}

int edrgp_calib(edrgp_ctx* ctx, double* mfma_f64_tflops, double* hbm_copy_gbps) {
    CTX(ctx);
    if (getenv("EDRGP_CALIB_PATTERNS")) {
        double t[10] = {0};
        TRY(calib_mfma_patterns(c, t));
        fprintf(stderr, "[calib] fp64 MFMA TFLOP/s, 2 waves/SIMD: registers only %.1f | AGPR acc %.1f | one A/B pair %.1f | "
                        "+LDS fragment reads %.1f | +barrier/k-tile %.1f | +staging ds_write (no barrier) %.1f | "
                        "+streaming global loads (no barrier) %.1f | full loop skeleton %.1f || 1 wave/SIMD 128x256 skeleton %.1f, "
                        "with fragment prefetch %.1f\n",
                t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9]);
    }
    if (getenv("EDRGP_CALIB_ORDERS")) {
        double t[12] = {0};
        for (int rep = 0; rep < 2; ++rep) {
            TRY(calib_mfma_orders(c, t));
            fprintf(stderr, "[calib] fp64 MFMA TFLOP/s by issue order of a k-step's 16 MFMAs (x-major | y-major | snake | rotated): "
                            "registers only %.1f %.1f %.1f %.1f ; full loop skeleton %.1f %.1f %.1f %.1f ; one wave per SIMD, registers only: "
                            "4x4 grid %.1f, one A/B pair %.1f ; loop skeleton with fragment prefetch: staging after k-step 1 %.1f, after k-step 3 %.1f\n",
                    t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11]);
        }
    }
    if (mfma_f64_tflops) TRY(calib_mfma_f64(c, mfma_f64_tflops));
    if (hbm_copy_gbps) TRY(calib_hbm_copy(c, hbm_copy_gbps));
    return EDRGP_OK;
}

}  // extern "C"
