/*
 * edrgp_hip.h -- C ABI of the MI355X (gfx950) sparse-GP inference core.
 *
 * The reference (neuro-ml/edr-gp) is pure Python and has no FFI: its GP
 * arithmetic is delegated to GPy through
 *     edrgp/gp_model/regression.py:76-77   (GPRegression)
 *     edrgp/gp_model/regression.py:153-157 (SparseGPRegression)
 *     edrgp/gp_model/base.py:69            (model.optimize -> one inference per iteration)
 *     edrgp/gp_model/base.py:187,206       (model.predict)
 *     edrgp/gp_model/base.py:222           (model.predictive_gradients)
 * Each entry point below replaces the GPy unit those call sites reach; the
 * Python side (edr-gp_amd/gp_model) binds them with ctypes and keeps the
 * reference's estimator API.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns an int
 *     status: 0 = ok, negative = error (text via edrgp_last_error()).
 *   - All matrices are row-major fp64.  Host pointers are borrowed for the
 *     duration of the call only.  Device state lives in the opaque context.
 *   - One context <-> one HIP device <-> one HIP stream.  A context is not
 *     thread-safe; different contexts may be used from different threads.
 *   - Nothing here falls back to the CPU: without a gfx950 device
 *     edrgp_ctx_create fails with EDRGP_ERR_HIP.
 */
#ifndef EDRGP_HIP_H
#define EDRGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct edrgp_ctx edrgp_ctx;

/* status codes */
#define EDRGP_OK            0
#define EDRGP_ERR_INVALID  -1   /* bad argument / shape */
#define EDRGP_ERR_NOT_PD   -2   /* Cholesky failed even after the jitchol retries */
#define EDRGP_ERR_HIP      -3   /* HIP runtime error (no device, OOM, launch failure) */
#define EDRGP_ERR_STATE    -4   /* call order: data / inducing / kernel / posterior missing */
#define EDRGP_ERR_COMM     -5   /* RCCL error */

/* stationary kernel kinds: GPy.kern.{RBF,Matern32,Matern52}
 * (names resolved by _make_kernel, edrgp/gp_model/base.py:134-146) */
#define EDRGP_KERN_RBF       0
#define EDRGP_KERN_MATERN32  1
#define EDRGP_KERN_MATERN52  2
#define EDRGP_MAX_PARTS      4   /* additive parts of a sum kernel (base.py:146) */
#define EDRGP_MAX_DIM        256 /* input dimension limit */

/* edrgp_get() selectors (debug / parity accessors) */
#define EDRGP_GET_PHI        1   /* K_mn K_nm (Gram form) or Lm^-1 K_mn K_nm Lm^-T (stable form) */
#define EDRGP_GET_BVEC       2   /* K_mn y (Gram form) or Lm^-1 K_mn y (stable form) */
#define EDRGP_GET_LM         3   /* chol(Kmm + 1e-8 I)   (M x M, lower) */
#define EDRGP_GET_LB         4   /* chol(I + A)          (M x M, lower) */
#define EDRGP_GET_A          5   /* beta Lm^-1 Phi Lm^-T (M x M) */
#define EDRGP_GET_WV         6   /* woodbury vector      (P)     */
#define EDRGP_GET_WINV       7   /* woodbury inverse     (P x P) */
#define EDRGP_GET_KMM        8   /* Kmm + jitter         (M x M) */
#define EDRGP_GET_KNM        9   /* first row chunk of K_nm (rows x M) */
#define EDRGP_GET_LMI       10   /* Lm^-1                (M x M) */
#define EDRGP_GET_LBI       11   /* LB^-1                (M x M) */
#define EDRGP_GET_DLDKMM    12   /* dL/dKmm              (M x M) */
#define EDRGP_GET_SCALARS   13   /* device scalar slots (16 doubles) */
#define EDRGP_GET_MODE      14   /* buf[0]: 1 Gram form was used, 2 stable form, 3 fp32 Gram; [1] fp32 guard estimate; [2] (max/min diag Lm)^2;
                                    [3] number of sync-free evaluations repeated by the careful driver */

/* profile slots: device time of kernel families measured with HIP events on
 * the context's stream (bench.py roofline) */
#define EDRGP_PROF_SYRK      0   /* Phi = K_mn K_nm fp64 MFMA kernel */
#define EDRGP_PROF_KBUILD    1   /* K_nm tile build */
#define EDRGP_PROF_GRADGEMM  2   /* W = K_nm S (gradient contraction) */
#define EDRGP_PROF_SMALL     3   /* O(M^3) part: Cholesky, trtri, M x M GEMMs */
#define EDRGP_PROF_CHOL      4   /* factorisation + inverse of one M x M matrix (Kmm, or I + A): 2 M^3 / 3 flop */
#define EDRGP_PROF_NSLOTS    8

const char* edrgp_last_error(void);
int  edrgp_version(void);
int  edrgp_device_count(int* count);

int  edrgp_ctx_create(int device, edrgp_ctx** ctx);
int  edrgp_ctx_destroy(edrgp_ctx* ctx);
int  edrgp_sync(edrgp_ctx* ctx);
/* Options (twelve; every other key is EDRGP_ERR_INVALID in the product library -- A/B switches of settled design
 * decisions and timing probes exist in probe builds only, see edr-gp_amd/build.py and tools/README.md):
 *  "chunk_rows"       rows of K_nm resident at a time; 0 (default) = automatic: all local rows when five N x M fp64 work
 *                     arrays fit into half of the free device memory (and N M < 2^31), else 131072
 *  "stats_mode"       form of the row statistics: 0 automatic (pivot-ratio rule), 1 Gram form Phi = K_mn K_nm,
 *                     2 stable form via K_nm Lm^-T (GPy's operation order)
 *  "pivot_ratio_max"  automatic mode takes the stable form when (max / min diag Lm)^2 exceeds this (default 120; above
 *                     2048 inducing points the limit is this times (2048 / M)^0.6)
 *  "precision"        64 (default) or 32 = BASELINE config 5: K_nm stored in fp32 and the gradient contraction
 *                     T = (K_nm S2 + ..) .* h(r), T^T [X | X.^2] on the fp32 MFMA with fp64 slab sums, for single-part
 *                     kernels with D <= 64 while Kmm is conditioned for the Gram form; the bound itself stays fp64
 *  "f32_phi"          with precision 32: also try the Gram matrix Phi in fp32; kept only if the factorisation of I + A
 *                     succeeds without jitter and an error estimate stays below "f32_guard" (EDRGP_GET_MODE reports 3
 *                     then; the bound is 1e-4..1e-3 accurate in that mode); "f32_ratio_max": fp32 Gram form only while
 *                     (max / min diag Lm)^2 stays below this
 *  "overlap"          side-stream use: 0 none, 1 the Kmm gradient term beside the gradient contraction and the K_nm
 *                     build beside the Kmm factorisation, 2 the whole row pass beside the Kmm factorisation,
 *                     3 (default) = 1, and 2 only for bound-only evaluations whose row pass is chunked (never with a
 *                     communicator attached).  Same results in every setting
 *  "sync_free"        (default 1) a sparse evaluation queues all its phases for the statistics form the previous one
 *                     ended in and waits for the device once, at the end, where it also reads the device's verdict
 *                     (Cholesky info words, pivot-ratio proxy); if the verdict disagrees -- jitter needed, or the other
 *                     form -- the evaluation is repeated with a synchronisation after each factorisation (0 = always
 *                     that way).  Results are those of the form the rule selects for the current Kmm either way
 *  "posterior_reuse"  (default 1) edrgp_sparse_posterior right after a bound+gradient evaluation of the same state is
 *                     served from that evaluation's factors
 *  "pad_inducing"     (default 1) M > 256 not a multiple of 128: decoupled phantom inducing points up to the next
 *                     multiple (exact: same bound, gradients and posterior)
 *  "profile"          (0/1) HIP-event timing of the kernel families (edrgp_profile_get) */
int  edrgp_set_option(edrgp_ctx* ctx, const char* key, double value);

/* Training data (already validated / normalised by the host: check_X_y at
 * gp_model/base.py:87, Standardize at regression.py:48,124).  Copied to HBM
 * once and kept resident. */
int  edrgp_set_data(edrgp_ctx* ctx, const double* X, const double* y, int64_t N, int D);
/* Training inputs as a projection of resident base rows: X = Xbase W (N x d) computed on the
 * device.  Xbase (N x Dbase) and y are uploaded when Xbase != NULL and stay resident for later
 * calls with Xbase == NULL, y == NULL -- the candidate projections of BASELINE config 4 and the
 * EDR loop's X C^T (edrgp/base.py:459-517; edrgp/edr.py:215-241 maps gradients back with W). */
int  edrgp_set_data_projected(edrgp_ctx* ctx, const double* Xbase, const double* y, int64_t N,
                              int Dbase, const double* W, int d);
/* Inducing inputs Z (M x D): GPy SparseGPRegression's Z parameter
 * (regression.py:153-157).  Re-sent whenever the optimiser moves Z. */
int  edrgp_set_inducing(edrgp_ctx* ctx, const double* Z, int M);
/* Kernel hyper-parameters: GPy kern.{variance,lengthscale} + Gaussian_noise.variance.
 * lengthscales is nparts x D (an isotropic part repeats its value D times). */
int  edrgp_set_kernel(edrgp_ctx* ctx, int nparts, const int* kinds,
                      const double* variances, const double* lengthscales,
                      double noise_var);

/* GPy kern.K(X, X2) (X2 == NULL: K(X, X), exact zeros of r on the diagonal) for a kernel passed by value:
 * what edrgp's tests build their targets with (edrgp/tests/test_edr.py:40-42, k.K(X)).  Stateless: the
 * resident data, inducing inputs, hyper-parameters and posterior of the context are not touched. */
int  edrgp_kernel_matrix(edrgp_ctx* ctx, int nparts, const int* kinds, const double* variances,
                         const double* lengthscales, const double* X, int64_t N, const double* X2, int64_t N2,
                         int D, double* K);

/* GPy VarDTC.inference (sparse fit, one call per optimiser iteration):
 * Titsias bound only. */
int  edrgp_sparse_lml(edrgp_ctx* ctx, double* lml);
/* ... bound + gradients: SparseGP._update_gradients.  dvar[nparts],
 * dlen[nparts*D] (per input dimension; the host sums them for an isotropic
 * part), dnoise[1], dZ[M*D]. */
int  edrgp_sparse_lml_grad(edrgp_ctx* ctx, double* lml, double* dvar, double* dlen,
                           double* dnoise, double* dZ);
/* Builds woodbury_vector / woodbury_inv for predict (GPy Posterior). */
int  edrgp_sparse_posterior(edrgp_ctx* ctx, double* lml);

/* GPy ExactGaussianInference.inference (dense fit, config 1): LML and, when
 * the pointers are non-NULL, its gradients. */
int  edrgp_dense_lml_grad(edrgp_ctx* ctx, double* lml, double* dvar, double* dlen,
                          double* dnoise);
int  edrgp_dense_posterior(edrgp_ctx* ctx, double* lml);

/* GPy Posterior._raw_predict (latent mean / variance, normalised space, no
 * likelihood noise; var clipped at 1e-15).  var may be NULL. */
int  edrgp_predict(edrgp_ctx* ctx, const double* Xs, int64_t Ns, double* mean, double* var);
/* Mean part of GPy GP.predictive_gradients: d mean / d x*, (Ns x D). */
int  edrgp_predict_grad(edrgp_ctx* ctx, const double* Xs, int64_t Ns, double* grad);
/* Same as edrgp_predict + edrgp_predict_grad on the resident training X
 * (what EDR calls: edrgp/base.py:274 with X = training set); outputs host. */
int  edrgp_predict_train(edrgp_ctx* ctx, double* mean, double* grad);
/* The same pass over the training rows, reduced on the device: gram (D x D) = J^T J with
 * J = d mean / d x (N x D), gsum (D, may be NULL) = column sums of J, mean (N, may be NULL).
 * This is what the EDR step (SVD / PCA of the gradients, edrgp/utils.py:27-55 via
 * edrgp/base.py:459-463) and the candidate sweep of BASELINE config 4 consume; the N x D
 * gradient matrix stays in HBM. */
int  edrgp_predict_train_gram(edrgp_ctx* ctx, double* mean, double* gram, double* gsum);

/* GP classification: GPy GPClassification = Bernoulli likelihood (probit link) + expectation propagation
 * (edrgp/gp_model/classification.py:63-64), dense, on the resident training inputs.  GPy's EP in its default
 * "alt" mode fits the site parameters once per optimize() call and then holds them fixed while the kernel
 * hyper-parameters move:
 *   edrgp_ep_begin(labels 0/1)                        K = kern.K(X) for the current hyper-parameters, sites zero
 *   edrgp_ep_sweep(order[N], eta, delta, &dt, &dv)    one sweep of site updates in the given order (upstream: a random
 *                                                     permutation per sweep) + recomputation of the posterior; returns
 *                                                     the mean squared change of tau and v (stop when both < 1e-6)
 *   edrgp_ep_finish(&log_Z_tilde)                     fixes the sites
 *   edrgp_ep_lml_grad / edrgp_ep_posterior            EP._inference for the fixed sites and the CURRENT hyper-parameters:
 *                                                     log marginal, d/dvariance, d/dlengthscale; woodbury vector / inverse
 *                                                     for edrgp_predict (latent mean / variance) and edrgp_predict_grad
 *   edrgp_ep_get_sites / edrgp_ep_set_sites           parity accessor / restoring a saved model */
int  edrgp_ep_begin(edrgp_ctx* ctx, const double* y);
int  edrgp_ep_sweep(edrgp_ctx* ctx, const int* order, double eta, double delta, double* tau_diff, double* v_diff);
int  edrgp_ep_finish(edrgp_ctx* ctx, double* log_Z_tilde);
int  edrgp_ep_get_sites(edrgp_ctx* ctx, double* tau, double* v);
int  edrgp_ep_set_sites(edrgp_ctx* ctx, const double* tau, const double* v, double log_Z_tilde);
int  edrgp_ep_lml_grad(edrgp_ctx* ctx, double* lml, double* dvar, double* dlen);
int  edrgp_ep_posterior(edrgp_ctx* ctx, double* lml);

/* Sparse GP classification (GPy SparseGPClassification = SparseGP + Bernoulli/probit + EPDTC, built at
 * edrgp/gp_model/classification.py:147-149; optimised at gp_model/base.py:69, predicted at :187 / :222).
 * Needs edrgp_set_data, edrgp_set_inducing and edrgp_set_kernel.  edrgp_epdtc_begin / edrgp_epdtc_sweep run
 * EPDTC.expectation_propagation (one sweep per call, site order from the caller; tau_diff / v_diff = mean squared change
 * of the sites against the previous sweep, EPBase._stop_criteria); edrgp_ep_finish, edrgp_ep_get_sites and
 * edrgp_ep_set_sites serve both EP variants.  With the sites fixed, edrgp_epdtc_lml_grad is EPDTC.inference =
 * VarDTC.inference(Y = v / tau, precision = tau, Z_tilde): bound, dvar[nparts], dlen[nparts * D], dZ[M * D] (any may be
 * NULL; all NULL = bound only) and edrgp_epdtc_posterior leaves the woodbury vector / inverse resident for
 * edrgp_predict*.  M <= 1024; replicas only (no row sharding). */
int  edrgp_epdtc_begin(edrgp_ctx* ctx, const double* y);
int  edrgp_epdtc_sweep(edrgp_ctx* ctx, const int* order, double eta, double delta, double* tau_diff, double* v_diff);
int  edrgp_epdtc_lml_grad(edrgp_ctx* ctx, double* lml, double* dvar, double* dlen, double* dZ);
int  edrgp_epdtc_posterior(edrgp_ctx* ctx, double* lml);

/* BASELINE config 4 as a batch: for the resident training rows (edrgp_set_data / edrgp_set_data_projected) and
 * inducing inputs, the unit of work of the EDR loop at fixed hyper-parameters (edrgp/base.py:460-461: fit on the
 * projected rows, posterior-mean gradients at the training rows) for `nscales` isotropic length-scales of one stationary kernel (`kind`:
 * RBF, Matern32 or Matern52) at one variance / noise: bounds[nscales] (Titsias bound), grams[nscales * D * D] (J^T J of the mean Jacobian at the
 * training rows; may be NULL), gsums[nscales * D] (column sums of J; may be NULL), modes[nscales] (1 Gram form,
 * 2 stable form, -1: this scale needs jitchol jitter -- its bound is NaN and the caller evaluates it with
 * edrgp_set_kernel + edrgp_sparse_posterior + edrgp_predict_train_gram, which give the same numbers as the batch does
 * for the others).  The squared distances are formed once, the factorisations of all scales advance in the same
 * launches.  Needs 64 < M <= 3072, one row chunk, no communicator. */
int  edrgp_sweep_scales(edrgp_ctx* ctx, int nscales, const double* lengthscales, double variance, double noise_var,
                        int kind, double* bounds, double* grams, double* gsums, int* modes);

/* debug / parity accessor */
int  edrgp_get(edrgp_ctx* ctx, int which, double* buf, int64_t cap_elems);

/* profiling (HIP events on the context's stream) */
int  edrgp_profile_reset(edrgp_ctx* ctx);
int  edrgp_profile_get(edrgp_ctx* ctx, int slot, double* total_ms, int64_t* launches);

/* Row-sharded multi-GPU (one process per GPU): all ranks hold the same Z and
 * hyper-parameters and a row block of (X, y); Phi, K_mn y and the scalars are
 * sum-all-reduced over RCCL once per evaluation -- the lower triangle of Phi and b, M (M + 1) / 2 + M doubles --
 * (GPy's own unused precedent: VarDTC_minibatch + mpi4py Allreduce).  Every rank must make the same sequence of
 * calls; the number and content of the collectives never depend on a rank's local row count. */
int  edrgp_comm_unique_id(void* id128);                 /* 128-byte ncclUniqueId */
int  edrgp_comm_init(edrgp_ctx* ctx, int nranks, int rank, const void* id128);
/* The same sharding over a collective the LAUNCHER provides (MPI, gloo, torch.distributed ...): `fn` must sum
 * `count` doubles in `buf` (host memory) over all ranks in place and return 0.  The library stages the exchanged
 * statistics through pinned host memory.  Slower than RCCL over xGMI; meant for hosts without a usable RCCL fabric
 * and for exercising the sharded control flow with several processes on one device. */
typedef int (*edrgp_allreduce_fn)(void* user, double* buf, int64_t count);
int  edrgp_comm_init_host(edrgp_ctx* ctx, int nranks, int rank, edrgp_allreduce_fn fn, void* user);
int  edrgp_comm_destroy(edrgp_ctx* ctx);

/* Low-level device linear algebra on host buffers -- test hooks for the
 * building blocks (row-major fp64). */
int  edrgp_dbg_gemm(edrgp_ctx* ctx, int transA, int transB, int M, int N, int K,
                    double alpha, const double* A, int lda, const double* B, int ldb,
                    double beta, double* C, int ldc);
/* fp32 building blocks of the "precision" = 32 row pass (fp32 MFMA; Phi slabs summed in fp64) */
int  edrgp_dbg_gemm32(edrgp_ctx* ctx, int transA, int transB, int M, int N, int K,
                      const float* A, int lda, const float* B, int ldb, float* C, int ldc);
int  edrgp_dbg_syrk32(edrgp_ctx* ctx, const float* K, int64_t N, int M, double* Phi);
int  edrgp_dbg_potrf(edrgp_ctx* ctx, double* A, int n, int* info);
/* factor and inverse of the factor out of the same launches: A <- L, Linv <- L^-1 */
int  edrgp_dbg_potrf_inv(edrgp_ctx* ctx, double* A, int n, int* info, double* Linv);
/* average time (ms) of `reps` runs of that factorisation-with-inverse alone on the SPD matrix A (bench.py) */
int  edrgp_dbg_potrf_time(edrgp_ctx* ctx, const double* A, int n, int reps, double* avg_ms);
int  edrgp_dbg_trtri(edrgp_ctx* ctx, const double* L, double* Linv, int n);
int  edrgp_dbg_syrk(edrgp_ctx* ctx, const double* K, int64_t N, int M, double* Phi);
/* times the production Phi = K^T K kernel alone on a device-generated K (bench.py "isolated" figure) */
int  edrgp_dbg_syrk_time(edrgp_ctx* ctx, int64_t N, int M, int splits, int reps, double* avg_ms, double* Phi_out);
int  edrgp_dbg_mfma_layout(edrgp_ctx* ctx, const double* A16x4, const double* B4x16, double* C16x16);
/* fp64 MFMA issue-rate microbenchmark: achieved TFLOP/s of back-to-back
 * v_mfma_f64_16x16x4_f64 on every SIMD, and HBM copy bandwidth in GB/s. */
int  edrgp_calib(edrgp_ctx* ctx, double* mfma_f64_tflops, double* hbm_copy_gbps);

#ifdef __cplusplus
}
#endif
#endif /* EDRGP_HIP_H */
