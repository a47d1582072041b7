// Internal declarations shared by the HIP translation units of libedrgp_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts, v_mfma_f64_16x16x4_f64.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>
#include <vector>
#include "../../include/edrgp_hip.h"

namespace edrgp {

void set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            edrgp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,         \
                             hipGetErrorString(_e));                               \
            return EDRGP_ERR_HIP;                                                  \
        }                                                                          \
    } while (0)

#define TRY(expr)                                                                  \
    do {                                                                           \
        int _s = (expr);                                                           \
        if (_s != EDRGP_OK) return _s;                                             \
    } while (0)

// ---------------------------------------------------------------- kernels spec
struct KSpec {                       // passed by value to device kernels
    int nparts;
    int D;
    int kind[EDRGP_MAX_PARTS];
    double var[EDRGP_MAX_PARTS];
    const double* invl;              // device, nparts x D : 1 / lengthscale
};

// ------------------------------------------------------------ grow-only buffer
struct DBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { release(); }           // every exit path of a function holding a local DBuf frees it
    void swap(DBuf& o) { void* tp = p; p = o.p; o.p = tp; size_t tb = bytes; bytes = o.bytes; o.bytes = tb; }
    int reserve(size_t need);        // grow-only (hipMalloc)
    void release();
    double* d() const { return static_cast<double*>(p); }
};

// ------------------------------------------------------------------- profiling
struct ProfSlot {
    double total_ms = 0.0;
    int64_t launches = 0;
};

struct Ctx;

struct ProfScope {                   // records start/stop events on ctx->stream
    Ctx* c;
    int slot;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(Ctx* c, int slot);
    ~ProfScope();
};

// ------------------------------------------------------------------------ GEMM
// Fused epilogue of the gradient contraction (single-part kernels): instead of
// W = K_nm S2 the kernel stores T = (W + ycoef y wv^T) .* h(r) and per-workgroup
// partial sums of (W + ycoef y wv^T) .* k / var  (Stationary.update_gradients_full).
struct FuseT {
    const double* X;        // rows x D   (rows of this chunk)
    const double* Z;        // M x D
    const double* y;        // rows
    const double* wv;       // M
    const double* invl;     // D
    double* dvpart;         // one partial per workgroup
    double* cspart;         // tiles_m x M partial column sums of T (or null)
    const double* K;        // (or null) the resident K_nm chunk: RBF takes h(r) = -k(r) from it, no distances
    const double* H = nullptr;  // (or null) h(r) = (dk/dr)/r of the same chunk, written by the build (other kernels), ld = ldk
    int ldk;
    double ycoef, var;
    int D, kind;
};

// Fused epilogue of the kernel-matrix build (single-part kernels): the GEMM computes the
// inner products of the scaled inputs, the epilogue turns them into covariances
//   r2 = max(|x|^2 + |z|^2 - 2 x.z, 0)   (GPy Stationary._unscaled_dist, same clip)
//   K  = var k(r2)  (+ jitter and r2 := 0 on the diagonal when X2 is X)
// and, when y is given, per-row-tile partial sums of K^T y.
struct FuseK {
    const double* xn;       // rows : |x/l|^2
    const double* zn;       // cols : |z/l|^2
    const double* y;        // rows or null
    double* bpart;          // tiles_m x N partial sums of K^T y (when y)
    double var, jitter;
    int kind, same;
    int dbg;                // diagnostic: 1 no exp, 2 no store
    double* H = nullptr;    // (or null) h(r) = (dk/dr)/r stored beside K (0 where r = 0)
    int ldh = 0;
};

struct GemmOpts {
    const FuseT* fuse = nullptr;             // tag 2 only
    const FuseK* fusek = nullptr;            // tag 3 only
    int tag = 0;                             // kernel symbol tag: 0 generic, 1 Phi syrk, 2 gradient W = K S
    int batch = 1;
    long long strideA = 0, strideB = 0, strideC = 0;
    int limM_base = 0, limM_stride = 0;      // M_eff(z) = min(M, limM_base - z*limM_stride)
    int limK_base = 0, limK_stride = 0;      // K_eff(z) likewise
    int tri = 0;                             // 1: only tiles with ti >= tj (M == N)
    int mirror_out = 0;                      // with tri: the upper triangle is filled from the lower one (symmetric result)
    int kmode = 0;                           // zero-skipping for triangular operands, see K_* below
    int force_tile = 0;                      // 64 or 128; 0 = heuristic; -1 = 128 if that gives >= 1024 tiles, else small_tile
    int small_tile = 64;
    int dyn_lds = 0;                         // tag 3 only: extra LDS bytes per workgroup (occupancy control, see sparse_eval.hip)
};

// k-range restrictions (GemmOpts::kmode) for triangular operands
constexpr int K_LE_ROW = 1;   // op(A)[i][k] == 0 for k > i   (A lower triangular, not transposed)
constexpr int K_GE_ROW = 2;   // op(A)[i][k] == 0 for k < i   (A = L^T)
constexpr int K_LE_COL = 4;   // op(B)[k][j] == 0 for k > j   (B = L^T)
constexpr int K_GE_COL = 8;   // op(B)[k][j] == 0 for k < j   (B lower triangular, not transposed)
int mirror_lower(Ctx* c, double* A, int n, int lda);          // copy the lower triangle onto the upper

// C = alpha * op(A) op(B) + beta * C, row-major fp64, MFMA f64 16x16x4.
int gemm(Ctx* c, bool transA, bool transB, int M, int N, int K, double alpha,
         const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc,
         const GemmOpts& o = GemmOpts());

// Phi (M x M, full symmetric) = K^T K (+ Phi if accumulate), K is rows x M with
// leading dimension ldk.  Lower-triangular tiles, split over rows into slabs,
// deterministic slab reduction + mirror.
int syrk_tn(Ctx* c, const double* K, int64_t rows, int M, int ldk, double* Phi, int ldphi,
            bool accumulate);

int pick_syrk_splits(Ctx* c, long long nt, int kt, int64_t rows, int M, int tile, int elem_bytes);
int slab_reduce_mirror(Ctx* c, const double* slabs, int splits, long long slabStride, int M, double* Phi);
int fill_pattern(Ctx* c, double* K, size_t n, unsigned long long seed);   // device-generated test operand (la_vec.hip)

// ------------------------------------------------------- Cholesky / triangular
// In-place lower Cholesky of the n x n matrix A (row-major, lda); the strict
// upper triangle is zeroed.  *info_dev (device int) receives 0 or the 1-based
// index of the first non-positive pivot.
int potrf_lower(Ctx* c, double* A, int n, int lda, int* info_dev);
// ... the factor left in the library's workspace (*L, ld *ldl; only its lower triangle is meaningful)
int potrf_lower_ws(Ctx* c, double* A, int n, int lda, int* info_dev, const double** L, int* ldl);
int aug_setup(Ctx* c, double* Lg, int ldb, const double* A, int n, double shift, const double* v, double scale);
int aug_tail(Ctx* c, const double* L, int ldl, int n, double* out_logdet, double* out_sumsq);
// Linv = L^-1 for lower-triangular L (explicit zeros above the diagonal of the output).
int trtri_lower(Ctx* c, const double* L, int ldl, double* Linv, int ldi, int n);
// both at once (one launch per 64-column step for n in (64, 3072]: la_chol.hip)
int potrf_inv_lower(Ctx* c, double* A, int n, int lda, int* info_dev, double* Linv, int ldi);
int potrf_inv_lower_batched(Ctx* c, double* A, int n, int lda, long long strideA, int* info_dev, double* Linv, int ldi,
                            long long strideI, int batch);

// -------------------------------------------------------------- vector helpers
int gemv_n(Ctx* c, int M, int N, double alpha, const double* A, int lda, const double* x, double* y);
int gemv_t(Ctx* c, int M, int N, double alpha, const double* A, int lda, const double* x, double* y);
int add_diag(Ctx* c, double* A, int n, int lda, double v);
int sum_sq(Ctx* c, const double* x, int64_t n, double* out);               // out[0] = sum x^2
int dot(Ctx* c, const double* x, const double* y, int64_t n, double* out);
int trace_of(Ctx* c, const double* A, int n, int lda, double* out);
int sum_log_diag(Ctx* c, const double* A, int n, int lda, double* out);
int diag_mean(Ctx* c, const double* A, int n, int lda, double* out);
int diag_minmax(Ctx* c, const double* A, int n, int lda, double* out2);   // out2 = {min, max} of diag
int frob_sq(Ctx* c, const double* A, int n, int lda, double* out);       // sum of squares n x n
int col_reduce(Ctx* c, const double* part, int nparts, int n, double* out); // out[j] = sum_p part[p][j]
int col_sums(Ctx* c, const double* A, int64_t rows, int n, double* out);      // out[j] = sum_r A[r][j], A rows x n (uses c->red)
// C = a*A + b*B + g * w w^T  (n x n, all ld = n)
int combine3(Ctx* c, int n, double a, const double* A, double b, const double* B, double cc,
             const double* Cm, double g, const double* w, double* out);

// ----------------------------------------------------------------- GP kernels
// K (rows x M2, ld = ldk) = k(X[r0:r0+rows], X2); if X2 == X and same==1 the
// diagonal r2 is exactly 0 and `jitter` is added there.  If y != nullptr also
// accumulates bpart[chunk][m] = sum_rows K[r][m] y[r] (partial sums; nchunk out).
int kernel_matrix(Ctx* c, const KSpec& ks, const double* X, int64_t rows, const double* X2, int M2,
                  double* K, int ldk, int same, double jitter,
                  const double* y, double* bvec /* M2, overwritten */);
// T = (W + ycoef * y wv^T) .* h_part(r) (0 where r == 0); dvar_out += sum (..) .* K_part / var
int grad_elementwise(Ctx* c, const KSpec& ks, int part, const double* X, const double* y, int64_t rows,
                     const double* Z, int M, const double* W, int ldw, double ycoef, const double* wv,
                     double* T, int ldt, double* dvar_accum /* device scalar, += */);
// Kmm term, single-part kernels: T = dK .* Hz, dvar_accum += sum dK .* Kz / var (Kz, Hz: K(Z, Z) and h(r) of the MFMA build)
int kmm_term_elementwise(Ctx* c, const double* dK, const double* Kz, const double* Hz, int Mr, int M, double var,
                         double* T, double* dvar_accum);
// Xt = [X | 1 | X.*X]  (rows x (2D+1))
int build_xtilde(Ctx* c, const double* X, int64_t rows, int D, double* Xt);
int build_xq(Ctx* c, const double* X, int64_t rows, int D, double* Xq);               // [X | X.^2]
int assemble_gt(Ctx* c, const double* G2, const double* cs, int M, int D, double beta_acc, double* Gt);
int grad_finish(Ctx* c, int np, int M, int Mr, int D, const double* gt_nm, const double* gt_mm, const double* Z,
                const double* invl, double* dZ, double* spart, int* nblocks);
int sum_into(Ctx* c, const double* part, int n, double* accum);   // accum[0] += sum(part), fixed order
int rowdot(Ctx* c, const double* A, const double* B, int64_t rows, int cols, int ld, double* out);
// posterior mean / latent variance / mean Jacobian for one device-resident row chunk
int predict_chunk(Ctx* c, const KSpec& ks, const double* Xs, int64_t rows, const double* P, const double* Pt,
                  int Mp, const double* wv, const double* winv, double kdiag, double* Kbuf, double* Gbuf,
                  double* GT, double* mean, double* var, double* jac, bool k_ready = false, double* Hbuf = nullptr,
                  bool h_ready = false);
int vec_axpy(Ctx* c, int64_t n, double a, const double* x, double* y);   // y += a x

// ------------------------------------------------------- fp32 row pass (la_gemm.hip, float instantiation)
int gemm32(Ctx* c, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
           float* C, int ldc);
int syrk32_tn(Ctx* c, const float* K, int64_t rows, int M, int ldk, double* Phi, int ldphi, bool accumulate);
int gemm32_tn_rows(Ctx* c, int M, int N, int64_t rows, const float* A, int lda, const float* B, int ldb, double* C,
                   int ldc, bool accumulate);
int to_f32(Ctx* c, const double* src, size_t n, float* dst);
int build_xq32(Ctx* c, const double* X, int64_t rows, int D, float* Xq);
int kernel_matrix32(Ctx* c, const KSpec& ks, const double* X, int64_t rows, const double* Z, int M, float* K32,
                    const double* y, double* bvec);
int grad_contract32(Ctx* c, const KSpec& ks, const float* K32, int64_t rows, int M, const float* S2f, const double* y,
                    const double* wv, double ycoef, float* T32, double* dvar_accum, double* cs);
int calib_mfma_f64(Ctx* c, double* tflops);
int calib_mfma_patterns(Ctx* c, double* tflops4);
int calib_mfma_orders(Ctx* c, double* tflops12);
int calib_hbm_copy(Ctx* c, double* gbps);
int mfma_layout_probe(Ctx* c, const double* A, const double* B, double* C);  // device ptrs

// --------------------------------------------------------------------- context
struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;        // side stream: latency-bound O(M^3) work beside the row pass
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_pred[2] = {nullptr, nullptr};   // predict_rows: a chunk's outputs are ready (two output slots)
    DBuf red2, slabs2;                    // scratch used while launching on the side stream
    DBuf ks_x, ks_z, ks_xn, ks_zn, ks_x2, ks_xn2;   // scaled inputs and squared norms for the MFMA kernel build
    bool mfma_kbuild = true;
    int kbuild_dbg = 0;
    int kbuild_tile = 64;                // tile of the MFMA kernel build (64 or 128)
    int kbuild_pad = 16384;              // LDS padding of the K_nm build while phase 1 runs beside it (bytes; 0 = none)
    int kbuild_pad_now = 0;              // padding in force for the next build launch
    int overlap = 3;                  // 0 none; 1 Kmm gradient term beside the gradient contraction;
                                      // 2 also phase 1 beside the row pass (stretches the Phi kernel);
                                      // 3 = 2 for bound-only evaluations, 1 for gradient evaluations
    bool syrk_diag_opt = true;        // Phi: diagonal tiles compute their 36 lower 16x16 blocks only
    bool syrk_xmap = true;            // Phi: split-major workgroup -> XCD map (la_gemm.hip)
    bool prebuild = true;             // gradient evaluations: K_nm build on the main stream beside phase 1 on the side stream
    // posterior reuse: every mutator bumps state_serial; a bound+gradient evaluation that completed leaves
    // Kmm^-1, P, wv of its (theta, Z, data) in the workspaces and records the serial it was computed for
    uint64_t state_serial = 1, grad_serial = 0;
    double grad_lml = 0.0;
    int grad_M = 0, grad_Mr = 0;
    bool posterior_reuse = true;
    bool gemm_pair = true;            // stable form's K_nm Li^T: column tiles paired (long, short) per workgroup
    bool pad_inducing = true;         // sparse evaluations: embed M inducing points in the next multiple of 128 (api.hip)
    int M_work = 0;                   // working size of the last sparse evaluation (>= M)
    bool chol_rhs = true;             // bound-only evaluations: L_B^-1 (beta v) as an extra row of the factorisation, no explicit L_B^-1
    bool chol_fused = true;           // Cholesky: panel + trailing update in one launch per step (la_chol.hip)
    bool chol_cap_grid = false;       // a step launch never has more workgroups than CUs (la_chol.hip); off since round 3: the bulk tiles, not workgroup 0, bound the wide steps
    bool chol_shares_gpu = false;     // set while a factorisation runs beside another kernel (phase 1 beside the K_nm build)
    bool chol_inv_fused = true;       // ... and the inverse of the factor out of the same launches
    bool rbf_epilogue = true;         // RBF: fused gradient epilogue takes h(r) = -k(r) from the resident K_nm
    bool fuse_grad = true;            // fused Q .* h epilogue of the gradient contraction (single-part kernels, D <= 32)
    int dbg_gemm_kmode = 0, dbg_gemm_tri = 0, dbg_gemm_tile = 0, dbg_gemm_reps = 1;   // edrgp_dbg_gemm only
    int mid_split = 4;                // k-segments of lower-tiles-only products over triangular operands (la_gemm.hip)
    int gemm_interleave = 2;          // zero-skipping products: 1 tiles dealt round-robin to the XCDs, 2 also length-ranked snake over the CUs (la_gemm.hip)
    int grad_finish_dev = 1;          // finish dZ / d lengthscale on the device (sparse_eval.hip)
    int gemm_narrow = 1;              // 64 x 32 tiles for TN products with at most 32 columns (la_gemm.hip)
    int h_resident = 1;               // non-RBF kernels: the build stores h(r) beside k, the contraction reads it (sparse_eval.hip)
    int skinny_split = 1;             // split-K for skinny outputs with 64 <= k-tiles < 256 (la_gemm.hip)
    int side_gemm_pad = 0;            // see gemm_pad_now
    int gemm_pad_now = 0;             // dynamic LDS padding of the 64 x 64 products queued beside a factorisation (bytes)
    int kmm_early = 1;                // Kmm-term products that need A only run beside the factorisation of B
    bool syrk_shares_gpu = false;     // set while the row pass is queued beside phase 1 (a persistent grid would starve it)
    int syrk_persist = 1;             // Phi launch as a persistent grid (la_gemm.hip)
    int gemm_persist = 2;             // 1: gradient contraction as a persistent grid, 2: the stable form's product too (la_gemm.hip)
    int gemm_alias = 0;               // timing probe of the contraction (wrong results): la_gemm.hip dbg_alias
    int gemm_dephase = 3;             // see la_gemm.hip: 3 = staging / barrier phases at raised priority
    bool trace = false;
    bool sync_free = true;            // sparse evaluations wait for the device once, at the end (sparse_eval.hip)
    int careful_left = 0;             // evaluations that skip the sync-free attempt (after one needed jitter)
    int64_t retries = 0;              // sync-free attempts repeated by the careful driver
    int num_cu = 256;

    // data
    int64_t N = 0;
    int D = 0;
    int M = 0;
    DBuf Xbase;                       // resident base rows of edrgp_set_data_projected (N x base_D)
    int64_t base_N = 0;
    int base_D = 0;
    bool base_y_valid = false;        // c->y still holds the targets uploaded with Xbase (edrgp_set_data overwrites them)
    DBuf X, y, Z, Xt, Zt, Xq;         // Xt = [X|1|X.^2], Zt likewise for Z, Xq = [X|X.^2]
    bool have_data = false, have_Z = false, have_kernel = false;
    double yy_local = 0.0;
    bool glob_valid = false;          // [N, y'y] summed over the ranks, once per data set
    double glob_N = 0.0, glob_yy = 0.0;

    // kernel (host copy)
    int nparts = 0;
    int kind[EDRGP_MAX_PARTS];
    double var[EDRGP_MAX_PARTS];
    std::vector<double> lens;         // nparts x D
    double noise_var = 1.0;
    DBuf invl;

    // fp32 row pass (precision == 32): N-proportional storage and MFMA in fp32, M x M work in fp64
    int precision = 64;
    double f32_guard = 1.0;           // a-posteriori bound on the relative perturbation of I + A (see sparse_eval)
    double f32_rho = 0.0;             // last value of that estimate
    double f32_ratio_max = 1e4;       // fp32 Gram form while (max/min diag Lm)^2 <= this, else the fp64 row pass
    DBuf Knm32, T32, S2f, Xq32, ks32_x, ks32_z, ks32_xn, ks32_zn;
    bool xq32_valid = false;
    bool f32_phi = false;             // also try Phi in fp32 (guarded); default: fp32 only in the gradient contraction
    bool phi32_bad = false;           // the guard rejected the fp32 Phi for the current data: use the fp64 row pass

    // options
    int64_t chunk_rows = 0;           // rows of K_nm resident at a time; 0 = automatic (chunk_rows_for below)
    int64_t auto_chunk = 0, auto_chunk_N = -1;    // the automatic choice and the (N, M) it was made for
    int auto_chunk_M = -1;
    int64_t chunk_rows_for(int M);    // api.hip
    int syrk_splits = 0;              // 0 = auto
    bool profile = false;
    int stats_mode = 0;               // 0 auto, 1 gram (Phi = K^T K), 2 stable (K Li^T first)
    double pivot_ratio_max = 120.0;   // auto: stable when (max/min diag Lm)^2 exceeds this (configs: C2 248 -> stable; Cm 61, C3 11, C5 27 -> Gram)
    int last_mode = 0;
    double last_pivot_ratio2 = 0.0;

    // posterior (for predict)
    bool have_post = false;
    bool post_dense = false;
    int P = 0;                        // number of posterior basis points
    DBuf wv, winv;
    DBuf xs_in;                       // test inputs of edrgp_predict / edrgp_predict_grad (grow-only)

    // workspace
    DBuf Knm, W, T;                   // chunk x M
    DBuf mats[16];                    // M x M (or N x N dense) matrices
    DBuf ptw;                         // -diag(wv) [P | 1 | P.^2] of the RBF predict-gradient path
    bool knm_train_valid = false;     // c->Knm holds K(X, Z) of ALL training rows for the current theta / Z
    bool hnm_train_valid = false;     // ... and c->Hnm its h(r) (non-RBF single-part kernels)
    DBuf Hnm;                         // h(r) of the resident K_nm chunk (non-RBF single-part kernels, gradient evaluations)
    DBuf gfin;                        // dZ (Mr x D) and per-block partial sums of the length-scale terms (grad_finish)
    DBuf g2;                          // T^T [X | X.^2] and column sums of T (fused gradient path)
    DBuf slabs, bpart, scal, info, gt_nm, gt_mm, tmpvec[6], panel, trtri_ws, dinv, red, lbaug;
    DBuf pack;                        // lower triangle of Phi and b, packed for the all-reduce
    DBuf sw_small, sw_r2, sw_mats, sw_kall;   // batched length-scale sweep (sweep_eval.hip)
    DBuf ep_K, ep_Sigma, ep_vec;      // EP classifiers (ep_eval.hip): K, Sigma (N x N; EPDTC: K_nm, U = K_nm Lm^-T, N x M), site / cavity / moment vectors
    bool ep_begun = false, ep_valid = false;
    bool ep_sparse = false, ep_fresh = false;   // the EP run in progress is EPDTC's; no site of it has been updated yet
    double ep_logZt = 0.0;            // log Z_tilde of the fitted sites
    double* kbuild_h = nullptr;       // the next MFMA build also stores h(r) here (ld = its K's): set around build_knm
    bool kbuild_r2 = false;           // the MFMA build stores r^2 instead of k(r^2) (set around the sweep's distance pass)
    std::vector<double> host_scal;
    std::vector<double> Z_host;       // host copy of the inducing inputs (gradient finishing)
    double* h_stage = nullptr;        // pinned staging of the gradient tables
    size_t h_stage_cap = 0;
    double* h_pin = nullptr;          // pinned: [0, 64) scalars + info words of an evaluation; then Z / 1/lengthscale staging
    size_t h_pin_cap = 0;             // doubles
    int ensure_pinned(size_t doubles);

    // comm
    void* comm = nullptr;             // ncclComm_t
    edrgp_allreduce_fn host_ar = nullptr;   // or: the launcher's own sum-all-reduce on host buffers (edrgp_comm_init_host)
    void* host_ar_user = nullptr;
    double* h_ar = nullptr;           // pinned staging of that transport
    size_t h_ar_cap = 0;
    int nranks = 1, rank = 0;
    bool sharded() const { return comm != nullptr || host_ar != nullptr; }

    // profiling
    ProfSlot prof[EDRGP_PROF_NSLOTS];
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending_events;
    int drain_events();
};

int allreduce_sum(Ctx* c, double* buf, size_t count);
// bound / woodbury vector / gradient Gram matrix for G isotropic length-scales of the resident rows and Z as one batch
int sweep_scales(Ctx* c, int G, const double* lengthscales, double var, double noise_var, int kind, double* bounds,
                 double* grams, double* gsums, int* modes);

}  // namespace edrgp
