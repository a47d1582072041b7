// GP classification: GPy GPClassification = GP + Bernoulli likelihood (probit link) + expectation propagation, as
// built by edrgp/gp_model/classification.py:63-64 and driven by edrgp/gp_model/base.py:69 (optimize), :187 (predict),
// :222 (predictive_gradients).  SURVEY 8f-4: last row of the scope table.
//
// GPy's EP (inference/latent_function_inference/expectation_propagation.py, class EP, defaults epsilon = 1e-6,
// eta = delta = 1, ep_mode "alt", sequential updates):
//   * one EP run per optimize() call: sweeps of site updates in a random order (the host draws the permutation, as
//     upstream does with np.random.permutation), each site followed by the rank-one update of Sigma and mu = Sigma v;
//     after every sweep the posterior is recomputed from scratch through chol(B), B = I + S^1/2 K S^1/2; stop when the
//     mean squared change of tau and of v is below epsilon;
//   * then the sites (tau, v) and log Z_tilde are FIXED while the kernel hyper-parameters are optimised: every
//     objective evaluation is _ep_marginal + _inference for those sites -- a Gaussian evaluation with pseudo-targets,
//     dL/dK = (alpha alpha^T - Wi) / 2, Wi = S^1/2 B^-1 S^1/2 = (K + S^-1)^-1.
// Device layout: K, Sigma and the M x M workspaces of the dense path are N x N fp64; a site costs two launches (scalars
// + column copy; row-parallel rank-one update fused with mu = Sigma v), a sweep ends with two N^3 products.
#include "eval_common.h"
#include <cstring>

namespace edrgp {

namespace {

enum { EP_TAU = 0, EP_V, EP_TAUC, EP_VC, EP_LOGZ, EP_MUHAT, EP_S2HAT, EP_MU, EP_Y, EP_TAUOLD, EP_VOLD, EP_S, EP_T1, EP_T2,
       EP_T3, EP_ROOT, EP_NVEC };

// log Phi(z) and phi(z) / Phi(z), stable in both tails (GPy util.univariate_Gaussian logCdfNormal / derivLogCdfNormal)
__device__ __forceinline__ void probit_terms(double z, double& log_Phi, double& phi_div_Phi) {
    if (z < 0.0) {
        const double t = -z * 0.70710678118654752440;
        const double e = erfcx(t);                       // Phi(z) = e exp(-t^2) / 2
        log_Phi = log(0.5 * e) - t * t;
        phi_div_Phi = 0.79788456080286535588 / e;        // sqrt(2 / pi) / erfcx(t)
    } else {
        const double Phi = 1.0 - 0.5 * erfc(z * 0.70710678118654752440);
        log_Phi = log1p(-0.5 * erfc(z * 0.70710678118654752440));
        phi_div_Phi = 0.39894228040143267794 * exp(-0.5 * z * z) / Phi;
    }
}

// Site i, first half: cavity (cavityParams._update_i), tilted moments (Bernoulli.moments_match_ep), site update with
// the positivity floor on tau (gaussianApproximation._update_i); column i of Sigma is copied for the second half.
__global__ __launch_bounds__(256) void ep_site_prepare_kernel(int i, int n, const double* __restrict__ Sigma,
                                                             double* __restrict__ vec, double eta, double delta) {
    double* tau = vec + (size_t)EP_TAU * n;
    double* v = vec + (size_t)EP_V * n;
    const double* mu = vec + (size_t)EP_MU * n;
    double* svec = vec + (size_t)EP_S * n;
    double* t1 = vec + (size_t)EP_T1 * n;               // [0] = ci
    for (int j = threadIdx.x; j < n; j += 256) svec[j] = Sigma[(size_t)i * n + j];
    if (threadIdx.x == 0) {
        const double sii = Sigma[(size_t)i * n + i], mi = mu[i];
        const double tc = 1.0 / sii - eta * tau[i];
        const double vc = mi / sii - eta * v[i];
        const double sign = (vec[(size_t)EP_Y * n + i] == 1.0) ? 1.0 : -1.0;
        const double den = sqrt(tc * tc + tc);
        const double z = sign * vc / den;
        double lz, r;
        probit_terms(z, lz, r);
        const double mh = vc / tc + sign * r / den;
        const double s2 = 1.0 / tc - (r / (tc * tc + tc)) * (z + r);
        double d_tau = delta / eta * (1.0 / s2 - 1.0 / sii);
        const double d_v = delta / eta * (mh / s2 - mi / sii);
        const double prev = tau[i];
        double tn = prev + d_tau;
        if (tn < 2.220446049250313e-16) {
            tn = 2.220446049250313e-16;
            d_tau = tn - prev;
        }
        tau[i] = tn;
        v[i] += d_v;
        vec[(size_t)EP_TAUC * n + i] = tc;
        vec[(size_t)EP_VC * n + i] = vc;
        vec[(size_t)EP_LOGZ * n + i] = lz;
        vec[(size_t)EP_MUHAT * n + i] = mh;
        vec[(size_t)EP_S2HAT * n + i] = s2;
        t1[0] = d_tau / (1.0 + d_tau * sii);
    }
}

// Site i, second half (posteriorParams._update_rank1): Sigma -= ci s s^T, mu = Sigma v.  One wave per row.
__global__ __launch_bounds__(256) void ep_site_update_kernel(int n, double* __restrict__ Sigma, double* __restrict__ vec) {
    const int lane = threadIdx.x & 63, row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double* v = vec + (size_t)EP_V * n;
    const double* svec = vec + (size_t)EP_S * n;
    const double ci = vec[(size_t)EP_T1 * n];
    const double cs = ci * svec[row];
    double acc = 0.0;
    double* Sr = Sigma + (size_t)row * n;
    for (int c = lane; c < n; c += 64) {
        const double val = Sr[c] - cs * svec[c];
        Sr[c] = val;
        acc = fma(val, v[c], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) vec[(size_t)EP_MU * n + row] = acc;
}

// B = I + (s s^T) .* K ; SrK = s[i] K[i][j]
__global__ void ep_bmat_kernel(int n, const double* __restrict__ K, const double* __restrict__ root, double* __restrict__ B,
                               double* __restrict__ SrK) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= n || j >= n) return;
    const double k = K[(size_t)i * n + j], si = root[i];
    B[(size_t)i * n + j] = si * k * root[j] + (i == j ? 1.0 : 0.0);
    if (SrK != nullptr) SrK[(size_t)i * n + j] = si * k;
}

// A[i][j] *= s[j]  (LWi = LBi diag(s))
__global__ void ep_scale_cols_kernel(int n, double* __restrict__ A, const double* __restrict__ s) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i < n && j < n) A[(size_t)i * n + j] *= s[j];
}

// element-wise vector helpers: op 0: out = sqrt(a); 1: out = a .* b; 2: out = a - b .* c; 3: out = a - b
__global__ void ep_vec_kernel(int op, int n, const double* __restrict__ a, const double* __restrict__ b,
                              const double* __restrict__ cc, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    if (op == 0) r = sqrt(a[i]);
    else if (op == 1) r = a[i] * b[i];
    else if (op == 2) r = a[i] - b[i] * cc[i];
    else r = a[i] - b[i];
    out[i] = r;
}

// log Z_tilde (EP._log_Z_tilde), summed in a fixed order by one workgroup
__global__ __launch_bounds__(256) void ep_logz_kernel(int n, const double* __restrict__ vec, double* __restrict__ out) {
    __shared__ double sh[256];
    const double* tau = vec + (size_t)EP_TAU * n;
    const double* v = vec + (size_t)EP_V * n;
    const double* tc = vec + (size_t)EP_TAUC * n;
    const double* vc = vec + (size_t)EP_VC * n;
    const double* lz = vec + (size_t)EP_LOGZ * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double sum = tc[i] + tau[i];
        s += lz[i] + 0.91893853320467274178 + 0.5 * log(1.0 + tau[i] / tc[i]) - 0.5 * v[i] * v[i] / sum +
             0.5 * (vc[i] * ((tau[i] / tc[i]) * vc[i] - 2.0 * v[i]) / sum);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

int vec_op(Ctx* c, int op, int n, const double* a, const double* b, const double* cc, double* out) {
    hipLaunchKernelGGL(ep_vec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, op, n, a, b, cc, out);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

struct EpWs {
    int n;
    double *K, *Sigma, *vec;
    double* v(int which) const { return vec + (size_t)which * n; }
};

int ep_ws(Ctx* c, EpWs* w, bool either_variant = false) {
    if (!c->ep_begun || (c->ep_sparse && !either_variant)) { set_error("EP: call edrgp_ep_begin first"); return EDRGP_ERR_STATE; }
    w->n = (int)c->N;
    w->K = c->ep_K.d();
    w->Sigma = c->ep_Sigma.d();
    w->vec = c->ep_vec.d();
    return EDRGP_OK;
}

// chol(B), B = I + S^1/2 K S^1/2, and LB^-1 for the current sites; SrK optional.  mats[0] = LB, mats[1] = LBi.
int ep_factor(Ctx* c, const EpWs& w, double* SrK) {
    const int n = w.n;
    TRY(ensure_mats(c, n, 5));
    double* LB = c->mats[0].d();
    double* LBi = c->mats[1].d();
    int* info = static_cast<int*>(c->info.p);
    TRY(vec_op(c, 0, n, w.v(EP_TAU), nullptr, nullptr, w.v(EP_ROOT)));
    dim3 blk(64, 4), grd((n + 63) / 64, (n + 3) / 4);
    hipLaunchKernelGGL(ep_bmat_kernel, grd, blk, 0, c->stream, n, w.K, w.v(EP_ROOT), LB, SrK);
    HIP_TRY(hipGetLastError());
    TRY(potrf_inv_lower(c, LB, n, n, info, LBi, n));          // B >= I: no jitter ladder needed
    return EDRGP_OK;
}

// posteriorParams._recompute: Sigma = K - V^T V, V = LB^-1 S^1/2 K ; mu = K alpha, alpha = v - s .* B^-1 (s .* K v)
int ep_recompute(Ctx* c, const EpWs& w) {
    const int n = w.n;
    TRY(ensure_mats(c, n, 5));
    double* SrK = c->mats[2].d();
    double* V = c->mats[3].d();
    TRY(ep_factor(c, w, SrK));
    double* LBi = c->mats[1].d();
    TRY(gemm(c, false, false, n, n, n, 1.0, LBi, n, SrK, n, 0.0, V, n, kopt(K_LE_ROW)));
    HIP_TRY(hipMemcpyAsync(w.Sigma, w.K, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    {
        GemmOpts o;
        o.tri = 1;
        TRY(gemm(c, true, false, n, n, n, -1.0, V, n, V, n, 1.0, w.Sigma, n, o));
        TRY(mirror_lower(c, w.Sigma, n, n));
    }
    // alpha, mu
    TRY(gemv_n(c, n, n, 1.0, w.K, n, w.v(EP_V), w.v(EP_T1)));                    // K v
    TRY(vec_op(c, 1, n, w.v(EP_ROOT), w.v(EP_T1), nullptr, w.v(EP_T2)));         // s .* K v
    TRY(gemv_n(c, n, n, 1.0, LBi, n, w.v(EP_T2), w.v(EP_T3)));                   // LBi (.)
    TRY(gemv_t(c, n, n, 1.0, LBi, n, w.v(EP_T3), w.v(EP_T2)));                   // LBi^T (.) = B^-1 (.)
    TRY(vec_op(c, 2, n, w.v(EP_V), w.v(EP_ROOT), w.v(EP_T2), w.v(EP_T3)));       // alpha
    TRY(gemv_n(c, n, n, 1.0, w.K, n, w.v(EP_T3), w.v(EP_MU)));
    return EDRGP_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
int ep_begin(Ctx* c, const double* y_host) {
    if (!c->have_data || !c->have_kernel) { set_error("EP needs set_data and set_kernel first"); return EDRGP_ERR_STATE; }
    if (c->sharded()) { set_error("the dense EP classifier does not shard: replicas only"); return EDRGP_ERR_STATE; }
    if (c->N > 20000) { set_error("dense EP classifier limited to N <= 20000"); return EDRGP_ERR_INVALID; }
    const int n = (int)c->N;
    ++c->state_serial;
    c->grad_serial = 0;
    c->have_post = false;
    c->ep_valid = false;
    TRY(c->ep_K.reserve((size_t)n * n * sizeof(double)));
    TRY(c->ep_Sigma.reserve((size_t)n * n * sizeof(double)));
    TRY(c->ep_vec.reserve((size_t)EP_NVEC * n * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->ep_vec.d(), 0, (size_t)EP_NVEC * n * sizeof(double), c->stream));
    std::vector<double> y(y_host, y_host + n);
    for (int i = 0; i < n; ++i)
        if (!(y[i] == 1.0 || y[i] == 0.0 || y[i] == -1.0)) { set_error("EP: labels must be 0 / 1 (or -1 / 1)"); return EDRGP_ERR_INVALID; }
    HIP_TRY(hipMemcpyAsync(c->ep_vec.d() + (size_t)EP_Y * n, y.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const KSpec ks = make_kspec(c);
    // K = kern.K(X): exact zeros of r on the diagonal, no jitter (upstream passes K as it is)
    TRY(kernel_matrix(c, ks, c->X.d(), n, c->X.d(), n, c->ep_K.d(), n, 1, 0.0, nullptr, nullptr));
    HIP_TRY(hipMemcpyAsync(c->ep_Sigma.d(), c->ep_K.d(), (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));             // y is a local
    c->ep_begun = true;
    c->ep_sparse = false;
    return EDRGP_OK;
}

int ep_sweep(Ctx* c, const int* order, double eta, double delta, double* tau_diff, double* v_diff) {
    EpWs w;
    TRY(ep_ws(c, &w));
    const int n = w.n;
    if (!order || !(eta > 0.0) || !(delta > 0.0)) { set_error("EP sweep: bad arguments"); return EDRGP_ERR_INVALID; }
    for (int k = 0; k < n; ++k) {
        const int i = order[k];
        if (i < 0 || i >= n) { set_error("EP sweep: site index out of range"); return EDRGP_ERR_INVALID; }
        hipLaunchKernelGGL(ep_site_prepare_kernel, dim3(1), dim3(256), 0, c->stream, i, n, w.Sigma, w.vec, eta, delta);
        hipLaunchKernelGGL(ep_site_update_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, n, w.Sigma, w.vec);
    }
    HIP_TRY(hipGetLastError());
    TRY(ep_recompute(c, w));
    // mean squared change of the site parameters against the previous sweep (EP._stop_criteria)
    double* sc = c->scal.d();
    TRY(vec_op(c, 3, n, w.v(EP_TAU), w.v(EP_TAUOLD), nullptr, w.v(EP_T1)));
    TRY(vec_op(c, 3, n, w.v(EP_V), w.v(EP_VOLD), nullptr, w.v(EP_T2)));
    TRY(sum_sq(c, w.v(EP_T1), n, sc + 0));
    TRY(sum_sq(c, w.v(EP_T2), n, sc + 1));
    HIP_TRY(hipMemcpyAsync(w.v(EP_TAUOLD), w.v(EP_TAU), (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(w.v(EP_VOLD), w.v(EP_V), (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    double h[2] = {0.0, 0.0};
    HIP_TRY(hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (tau_diff) *tau_diff = h[0] / n;
    if (v_diff) *v_diff = h[1] / n;
    return EDRGP_OK;
}

int ep_finish(Ctx* c, double* log_Z_tilde) {
    EpWs w;
    TRY(ep_ws(c, &w, true));                               // the site / cavity / moment vectors are laid out alike
    double* sc = c->scal.d();
    hipLaunchKernelGGL(ep_logz_kernel, dim3(1), dim3(256), 0, c->stream, w.n, w.vec, sc);
    HIP_TRY(hipGetLastError());
    double h = 0.0;
    HIP_TRY(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->ep_logZt = h;
    c->ep_valid = true;
    ++c->state_serial;
    if (log_Z_tilde) *log_Z_tilde = h;
    return EDRGP_OK;
}

int ep_sites(Ctx* c, double* tau, double* v, const double* tau_in, const double* v_in, double logZt_in) {
    if (tau_in != nullptr) {                 // restore sites (a loaded model): no EP state needed beyond the vectors
        if (!c->have_data) { set_error("EP: set_data first"); return EDRGP_ERR_STATE; }
        const int n = (int)c->N;
        TRY(c->ep_vec.reserve((size_t)EP_NVEC * n * sizeof(double)));
        HIP_TRY(hipStreamSynchronize(c->stream));          // nothing in flight may still read the old sites
        HIP_TRY(hipMemcpy(c->ep_vec.d() + (size_t)EP_TAU * n, tau_in, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->ep_vec.d() + (size_t)EP_V * n, v_in, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        c->ep_logZt = logZt_in;
        c->ep_valid = true;
        c->have_post = false;
        ++c->state_serial;
        return EDRGP_OK;
    }
    if (!c->ep_valid && !c->ep_begun) { set_error("EP: no sites"); return EDRGP_ERR_STATE; }
    const int n = (int)c->N;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (tau) HIP_TRY(hipMemcpy(tau, c->ep_vec.d() + (size_t)EP_TAU * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (v) HIP_TRY(hipMemcpy(v, c->ep_vec.d() + (size_t)EP_V * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return EDRGP_OK;
}

// EP._ep_marginal + EP._inference for the resident sites and the CURRENT kernel hyper-parameters.
int ep_eval(Ctx* c, bool need_grad, bool need_post, double* lml, std::vector<double>* dvar, std::vector<double>* dlen) {
    if (!c->have_data || !c->have_kernel || !c->ep_valid) { set_error("EP evaluation needs data, kernel and fitted sites"); return EDRGP_ERR_STATE; }
    if (c->sharded()) { set_error("the dense EP classifier does not shard: replicas only"); return EDRGP_ERR_STATE; }
    const int n = (int)c->N;
    c->grad_serial = 0;
    TRY(c->ep_K.reserve((size_t)n * n * sizeof(double)));
    EpWs w;
    w.n = n; w.K = c->ep_K.d(); w.Sigma = nullptr; w.vec = c->ep_vec.d();
    const KSpec ks = make_kspec(c);
    ProfScope ps(c, EDRGP_PROF_SMALL);
    TRY(kernel_matrix(c, ks, c->X.d(), n, c->X.d(), n, w.K, n, 1, 0.0, nullptr, nullptr));
    TRY(ep_factor(c, w, nullptr));
    double* LB = c->mats[0].d();
    double* LWi = c->mats[1].d();
    double* Wi = c->mats[2].d();
    double* dK = c->mats[3].d();
    double* Tn = c->mats[4].d();
    double* sc = c->scal.d();
    HIP_TRY(hipMemsetAsync(sc, 0, SC_COUNT * sizeof(double), c->stream));
    TRY(sum_log_diag(c, LB, n, n, sc + SC_SUMLOGL));
    // LWi = LB^-1 S^1/2 ; Wi = LWi^T LWi = (K + S^-1)^-1
    {
        dim3 blk(64, 4), grd((n + 63) / 64, (n + 3) / 4);
        hipLaunchKernelGGL(ep_scale_cols_kernel, grd, blk, 0, c->stream, n, LWi, w.v(EP_ROOT));
        HIP_TRY(hipGetLastError());
    }
    TRY(gram_of_lower(c, n, LWi, Wi));
    // alpha = v - Wi K v ;  v^T Sigma v = v^T K v - |LWi K v|^2
    TRY(gemv_n(c, n, n, 1.0, w.K, n, w.v(EP_V), w.v(EP_T1)));
    TRY(gemv_n(c, n, n, 1.0, Wi, n, w.v(EP_T1), w.v(EP_T2)));
    TRY(vec_op(c, 3, n, w.v(EP_V), w.v(EP_T2), nullptr, w.v(EP_T3)));            // alpha
    TRY(gemv_n(c, n, n, 1.0, LWi, n, w.v(EP_T1), w.v(EP_T2)));
    TRY(dot(c, w.v(EP_V), w.v(EP_T1), n, sc + SC_YALPHA));
    TRY(sum_sq(c, w.v(EP_T2), n, sc + SC_ALPHASQ));
    double h_sc[SC_COUNT];
    int h_info[4];
    HIP_TRY(hipMemcpyAsync(h_sc, sc, sizeof(h_sc), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_info, c->info.p, sizeof(h_info), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h_info[0] != 0) { set_error("EP: I + S^1/2 K S^1/2 not positive definite"); return EDRGP_ERR_NOT_PD; }
    if (lml)
        *lml = 0.5 * (-(double)n * std::log(2.0 * M_PI) - 2.0 * h_sc[SC_SUMLOGL] + (h_sc[SC_YALPHA] - h_sc[SC_ALPHASQ])) + c->ep_logZt;
    if (need_post) {
        TRY(c->wv.reserve((size_t)n * sizeof(double)));
        TRY(c->winv.reserve((size_t)n * n * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(c->wv.d(), w.v(EP_T3), (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->winv.d(), Wi, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->have_post = true;
        c->post_dense = true;
        c->P = n;
    }
    if (need_grad) {
        TRY(combine3(c, n, -0.5, Wi, 0.0, nullptr, 0.0, nullptr, 0.5, w.v(EP_T3), dK));
        TRY(dense_kernel_gradients(c, dK, Tn, dvar, dlen));
    }
    return EDRGP_OK;
}


// =====================================================================================================================
// Sparse GP classification: GPy SparseGPClassification = SparseGP + Bernoulli (probit) + EPDTC
// (edrgp/gp_model/classification.py:147-149; driven by gp_model/base.py:69, :187, :222 like every other model).
//
// EPDTC (expectation_propagation.py, class EPDTC / posteriorParamsDTC) runs the same site updates as EP on the DTC
// posterior  Sigma = K_nm (K_mm + K_mn T K_nm)^-1 K_mn,  T = diag(tau).  Upstream refactors the M x M matrix
// LLT = K_mm + K_mn T K_nm and solves against all of K_mn after EVERY site (O(M^3 + N M^2) per site).  Here, with
// Lm = jitchol(K_mm) and the whitened rows U = K_nm Lm^-T (N x M, fixed during an EP run):
//     LLT = Lm (I + U^T T U) Lm^T = Lm B Lm^T,      Sigma = U B^-1 U^T,
// so a site's rank-one change of LLT is B += d_tau u_i u_i^T and, by Sherman-Morrison on the well-conditioned B
// (eigenvalues >= 1),  B^-1 -= c w w^T  with  w = B^-1 u_i,  c = d_tau / (1 + d_tau u_i^T w):  O(M^2 + N M) per site,
// the same posterior.  Only Sigma_ii of the site being visited is ever read (u_i^T B^-1 u_i, formed on the spot) and
// mu follows upstream's increment  mu += (d_v - d_tau mu_i) Sigma'[:, i].  After every sweep B is rebuilt and
// factored from scratch (posteriorParamsDTC._recompute), which also removes the drift of the rank-one updates.
// A site costs two launches: one workgroup (w, the scalars, the update of B^-1), then one wave per row (mu).
//
// With the sites fixed, every objective evaluation is VarDTC.inference with the pseudo-targets v / tau, the
// heteroscedastic precision tau and Z_tilde = log Z_tilde (var_dtc.py, het_noise branches) -- written out below in the
// stable form (A = U^T T U) on the generic products; the gradient tables are finished on the host as in
// dense_kernel_gradients.  Sizes are a classifier's (upstream's default is 10 inducing points): nothing here is tuned.
// =====================================================================================================================
namespace {

constexpr int EPD_MAX_M = 1024;

__device__ __forceinline__ double block_sum_256(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// Site i: w = Bi u_i, Sigma_ii = u_i^T w; cavity, tilted moments and site update (as ep_site_prepare_kernel); then
// Bi -= c w w^T.  wbuf[0 .. M) = w, wbuf[M] = the coefficient of mu's increment.
__global__ __launch_bounds__(256) void epdtc_site_kernel(int i, int n, int M, const double* __restrict__ Ut,
                                                        double* __restrict__ Bi, double* __restrict__ vec,
                                                        double* __restrict__ wbuf, double eta, double delta, int first) {
    __shared__ double u_s[EPD_MAX_M], w_s[EPD_MAX_M], red[256];
    __shared__ double ci_s;
    for (int j = threadIdx.x; j < M; j += 256) u_s[j] = Ut[(size_t)i * M + j];
    __syncthreads();
    double part = 0.0;
    for (int j = threadIdx.x; j < M; j += 256) {
        double acc = 0.0;
        for (int k = 0; k < M; ++k) acc = fma(Bi[(size_t)k * M + j], u_s[k], acc);       // Bi is symmetric
        w_s[j] = acc;
        wbuf[j] = acc;
        part = fma(acc, u_s[j], part);
    }
    const double sii = block_sum_256(part, red);
    if (threadIdx.x == 0) {
        double* tau = vec + (size_t)EP_TAU * n;
        double* v = vec + (size_t)EP_V * n;
        const double* mu = vec + (size_t)EP_MU * n;
        // upstream's Sigma_diag: Qnn + 1e-8 before the first update of a fresh run, max(diag, eps) afterwards
        const double sd = first ? sii + 1e-8 : fmax(sii, 2.220446049250313e-16);
        const double mi = mu[i];
        const double tc = 1.0 / sd - eta * tau[i];
        const double vc = mi / sd - eta * v[i];
        const double sign = (vec[(size_t)EP_Y * n + i] == 1.0) ? 1.0 : -1.0;
        const double den = sqrt(tc * tc + tc);
        const double z = sign * vc / den;
        double lz, r;
        probit_terms(z, lz, r);
        const double mh = vc / tc + sign * r / den;
        const double s2 = 1.0 / tc - (r / (tc * tc + tc)) * (z + r);
        double d_tau = delta / eta * (1.0 / s2 - 1.0 / sd);
        const double d_v = delta / eta * (mh / s2 - mi / sd);
        const double prev = tau[i];
        double tn = prev + d_tau;
        if (tn < 2.220446049250313e-16) {
            tn = 2.220446049250313e-16;
            d_tau = tn - prev;
        }
        tau[i] = tn;
        v[i] += d_v;
        vec[(size_t)EP_TAUC * n + i] = tc;
        vec[(size_t)EP_VC * n + i] = vc;
        vec[(size_t)EP_LOGZ * n + i] = lz;
        vec[(size_t)EP_MUHAT * n + i] = mh;
        vec[(size_t)EP_S2HAT * n + i] = s2;
        const double den1 = 1.0 + d_tau * sii;
        ci_s = d_tau / den1;
        wbuf[M] = (d_v - d_tau * mi) / den1;          // Sigma'[:, i] = U w / (1 + d_tau Sigma_ii)
    }
    __syncthreads();
    const double ci = ci_s;
    const size_t total = (size_t)M * M;
    for (size_t e = threadIdx.x; e < total; e += 256) {
        const int a = (int)(e / M), b = (int)(e - (size_t)a * M);
        Bi[e] = fma(-ci * w_s[a], w_s[b], Bi[e]);
    }
}

// mu[row] += coef * U[row] . w.  One wave per row.
__global__ __launch_bounds__(256) void epdtc_mu_kernel(int n, int M, const double* __restrict__ Ut,
                                                      const double* __restrict__ wbuf, double* __restrict__ vec) {
    const int lane = threadIdx.x & 63, row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double* ur = Ut + (size_t)row * M;
    double acc = 0.0;
    for (int k = lane; k < M; k += 64) acc = fma(ur[k], wbuf[k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) vec[(size_t)EP_MU * n + row] += wbuf[M] * acc;
}

// out[r][:] = f(s[r]) in[r][:],  f = sqrt (mode 0) or identity (mode 1)
__global__ void epdtc_scale_rows_kernel(int64_t rows, int M, const double* __restrict__ in, const double* __restrict__ s,
                                        int mode, double* __restrict__ out) {
    const int64_t total = rows * M;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const double f = s[e / M];
        out[e] = (mode == 0 ? sqrt(f) : f) * in[e];
    }
}

int scale_rows(Ctx* c, int64_t rows, int M, const double* in, const double* s, int mode, double* out) {
    const int nb = (int)std::min<int64_t>(4096, (rows * M + 255) / 256);
    hipLaunchKernelGGL(epdtc_scale_rows_kernel, dim3(std::max(nb, 1)), dim3(256), 0, c->stream, rows, M, in, s, mode, out);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

enum { EM_LM = 0, EM_LI, EM_A, EM_LB, EM_LBI, EM_BI, EM_E1, EM_E2, EM_T1, EM_DKMM, EM_S2, EM_TMM, EM_COUNT };

struct EpdWs {
    int n, M;
    double *Knm, *Ut, *vec;
    double* v(int which) const { return vec + (size_t)which * n; }
};

int epd_ws(Ctx* c, EpdWs* w) {
    if (!c->have_data || !c->have_Z || !c->have_kernel) { set_error("EPDTC needs set_data, set_inducing and set_kernel first"); return EDRGP_ERR_STATE; }
    if (c->sharded()) { set_error("the sparse EP classifier does not shard: replicas only"); return EDRGP_ERR_STATE; }
    if (c->M > EPD_MAX_M) { set_error("sparse EP classifier limited to 1024 inducing points"); return EDRGP_ERR_INVALID; }
    if (c->N > 0x7fffffffLL / std::max(c->M, 1)) { set_error("sparse EP classifier: N x M too large"); return EDRGP_ERR_INVALID; }
    w->n = (int)c->N;
    w->M = c->M;
    TRY(c->ep_K.reserve((size_t)w->n * w->M * sizeof(double)));
    TRY(c->ep_Sigma.reserve((size_t)w->n * w->M * sizeof(double)));
    TRY(c->ep_vec.reserve((size_t)EP_NVEC * w->n * sizeof(double)));
    TRY(ensure_mats(c, w->M, EM_COUNT));
    w->Knm = c->ep_K.d();
    w->Ut = c->ep_Sigma.d();
    w->vec = c->ep_vec.d();
    return EDRGP_OK;
}

// K_nm, Lm = jitchol(K(Z, Z)) (GPy's ladder, NO const_jitter: EPDTC.inference factors kern.K(Z) itself), Lm^-1 and
// the whitened rows U = K_nm Lm^-T
int epd_kernel_side(Ctx* c, const EpdWs& w) {
    const KSpec ks = make_kspec(c);
    const int M = w.M;
    double* Lm = c->mats[EM_LM].d();
    double* Li = c->mats[EM_LI].d();
    int* info = static_cast<int*>(c->info.p);
    TRY(kernel_matrix(c, ks, c->X.d(), w.n, c->Z.d(), M, w.Knm, M, 0, 0.0, nullptr, nullptr));
    const double dmean = var_sum(c);
    for (int lvl = -1;; ++lvl) {
        if (lvl > 4) { set_error("EPDTC: K(Z, Z) not positive definite, even with jitter"); return EDRGP_ERR_NOT_PD; }
        TRY(kernel_matrix(c, ks, c->Z.d(), M, c->Z.d(), M, Lm, M, 1, jit_amount(lvl, dmean), nullptr, nullptr));
        TRY(potrf_inv_lower(c, Lm, M, M, info, Li, M));
        int h = 0;
        HIP_TRY(hipMemcpyAsync(&h, info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (h == 0) break;
    }
    TRY(gemm(c, false, true, w.n, M, M, 1.0, w.Knm, M, Li, M, 0.0, w.Ut, M));
    return EDRGP_OK;
}

// B = I + U^T T U -> LB (EM_LB), LB^-1 (EM_LBI); A = U^T T U stays in EM_A.  scratch: N x M
int epd_factor_b(Ctx* c, const EpdWs& w, double* scratch) {
    const int M = w.M;
    double* A = c->mats[EM_A].d();
    double* LB = c->mats[EM_LB].d();
    double* LBi = c->mats[EM_LBI].d();
    int* info = static_cast<int*>(c->info.p);
    TRY(scale_rows(c, w.n, M, w.Ut, w.v(EP_TAU), 0, scratch));
    TRY(gemm(c, true, false, M, M, w.n, 1.0, scratch, M, scratch, M, 0.0, A, M));
    HIP_TRY(hipMemcpyAsync(LB, A, (size_t)M * M * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    TRY(add_diag(c, LB, M, M, 1.0));
    TRY(potrf_inv_lower(c, LB, M, M, info, LBi, M));          // B >= I: no jitter ladder needed
    return EDRGP_OK;
}

// posteriorParamsDTC._recompute: B^-1 and mu = U B^-1 U^T v
int epd_recompute(Ctx* c, const EpdWs& w) {
    const int M = w.M;
    TRY(c->W.reserve((size_t)w.n * M * sizeof(double)));
    TRY(epd_factor_b(c, w, c->W.d()));
    double* Bi = c->mats[EM_BI].d();
    TRY(gram_of_lower(c, M, c->mats[EM_LBI].d(), Bi));
    double* t1 = c->tmpvec[0].d();
    double* t2 = c->tmpvec[1].d();
    TRY(gemv_t(c, w.n, M, 1.0, w.Ut, M, w.v(EP_V), t1));
    TRY(gemv_n(c, M, M, 1.0, Bi, M, t1, t2));
    TRY(gemv_n(c, w.n, M, 1.0, w.Ut, M, t2, w.v(EP_MU)));
    return EDRGP_OK;
}

}  // namespace

int epdtc_begin(Ctx* c, const double* y_host) {
    EpdWs w;
    TRY(epd_ws(c, &w));
    const int n = w.n, M = w.M;
    ++c->state_serial;
    c->grad_serial = 0;
    c->have_post = false;
    c->ep_valid = false;
    c->ep_begun = false;
    c->knm_train_valid = false;
    c->hnm_train_valid = false;
    HIP_TRY(hipMemsetAsync(w.vec, 0, (size_t)EP_NVEC * n * sizeof(double), c->stream));
    std::vector<double> y(y_host, y_host + n);
    for (int i = 0; i < n; ++i)
        if (!(y[i] == 1.0 || y[i] == 0.0 || y[i] == -1.0)) { set_error("EP: labels must be 0 / 1 (or -1 / 1)"); return EDRGP_ERR_INVALID; }
    HIP_TRY(hipMemcpyAsync(w.v(EP_Y), y.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    TRY(epd_kernel_side(c, w));
    // no sites yet: B = I
    double* Bi = c->mats[EM_BI].d();
    HIP_TRY(hipMemsetAsync(Bi, 0, (size_t)M * M * sizeof(double), c->stream));
    TRY(add_diag(c, Bi, M, M, 1.0));
    HIP_TRY(hipStreamSynchronize(c->stream));             // y is a local
    c->ep_begun = true;
    c->ep_sparse = true;
    c->ep_fresh = true;
    return EDRGP_OK;
}

int epdtc_sweep(Ctx* c, const int* order, double eta, double delta, double* tau_diff, double* v_diff) {
    if (!c->ep_begun || !c->ep_sparse) { set_error("EPDTC: call edrgp_epdtc_begin first"); return EDRGP_ERR_STATE; }
    EpdWs w;
    TRY(epd_ws(c, &w));
    const int n = w.n, M = w.M;
    if (!order || !(eta > 0.0) || !(delta > 0.0)) { set_error("EP sweep: bad arguments"); return EDRGP_ERR_INVALID; }
    TRY(c->ptw.reserve((size_t)(M + 8) * sizeof(double)));
    double* wbuf = c->ptw.d();
    double* Bi = c->mats[EM_BI].d();
    for (int k = 0; k < n; ++k) {
        const int i = order[k];
        if (i < 0 || i >= n) { set_error("EP sweep: site index out of range"); return EDRGP_ERR_INVALID; }
        hipLaunchKernelGGL(epdtc_site_kernel, dim3(1), dim3(256), 0, c->stream, i, n, M, w.Ut, Bi, w.vec, wbuf, eta, delta,
                           (c->ep_fresh && k == 0) ? 1 : 0);
        hipLaunchKernelGGL(epdtc_mu_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, n, M, w.Ut, wbuf, w.vec);
    }
    HIP_TRY(hipGetLastError());
    c->ep_fresh = false;
    TRY(epd_recompute(c, w));
    double* sc = c->scal.d();
    TRY(vec_op(c, 3, n, w.v(EP_TAU), w.v(EP_TAUOLD), nullptr, w.v(EP_T1)));
    TRY(vec_op(c, 3, n, w.v(EP_V), w.v(EP_VOLD), nullptr, w.v(EP_T2)));
    TRY(sum_sq(c, w.v(EP_T1), n, sc + 0));
    TRY(sum_sq(c, w.v(EP_T2), n, sc + 1));
    HIP_TRY(hipMemcpyAsync(w.v(EP_TAUOLD), w.v(EP_TAU), (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(w.v(EP_VOLD), w.v(EP_V), (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    double h[2] = {0.0, 0.0};
    int h_info = 0;
    HIP_TRY(hipMemcpyAsync(h, sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&h_info, c->info.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h_info != 0) { set_error("EPDTC: I + U^T T U not positive definite"); return EDRGP_ERR_NOT_PD; }
    if (tau_diff) *tau_diff = h[0] / n;
    if (v_diff) *v_diff = h[1] / n;
    return EDRGP_OK;
}

// EPDTC.inference for the resident sites and the CURRENT kernel hyper-parameters / inducing inputs.
int epdtc_eval(Ctx* c, bool need_grad, bool need_post, double* lml, std::vector<double>* dvar, std::vector<double>* dlen,
               std::vector<double>* dZ) {
    if (!c->ep_valid) { set_error("EPDTC evaluation needs fitted sites"); return EDRGP_ERR_STATE; }
    EpdWs w;
    TRY(epd_ws(c, &w));
    const int n = w.n, M = w.M, D = c->D, np = c->nparts, Wd = 2 * D + 1;
    c->grad_serial = 0;
    c->knm_train_valid = false;
    c->hnm_train_valid = false;
    c->ep_begun = false;                                   // U of the EP run is overwritten below
    ProfScope ps(c, EDRGP_PROF_SMALL);
    TRY(c->W.reserve((size_t)n * M * sizeof(double)));
    TRY(epd_kernel_side(c, w));
    TRY(epd_factor_b(c, w, c->W.d()));
    double* Li = c->mats[EM_LI].d();
    double* A = c->mats[EM_A].d();
    double* LB = c->mats[EM_LB].d();
    double* LBi = c->mats[EM_LBI].d();
    double* sc = c->scal.d();
    double* t1 = c->tmpvec[0].d();
    double* c1 = c->tmpvec[1].d();
    double* t2 = c->tmpvec[2].d();
    double* wv = c->tmpvec[3].d();
    double* pv = c->tmpvec[4].d();
    HIP_TRY(hipMemsetAsync(sc, 0, SC_COUNT * sizeof(double), c->stream));
    TRY(sum_log_diag(c, LB, M, M, sc + SC_SUMLOGL));
    TRY(trace_of(c, A, M, M, sc + SC_TRA));
    // psi1Vf = K_mn v ; c1 = LB^-1 Lm^-1 psi1Vf ; wv = Lm^-T LB^-T c1
    TRY(gemv_t(c, n, M, 1.0, w.Knm, M, w.v(EP_V), pv));
    TRY(gemv_n(c, M, M, 1.0, Li, M, pv, t1));
    TRY(gemv_n(c, M, M, 1.0, LBi, M, t1, c1));
    TRY(gemv_t(c, M, M, 1.0, LBi, M, c1, t2));
    TRY(gemv_t(c, M, M, 1.0, Li, M, t2, wv));
    TRY(sum_sq(c, c1, M, sc + SC_ALPHASQ));
    std::vector<double> h_tau(n), h_v(n);
    double h_sc[SC_COUNT];
    int h_info[4];
    HIP_TRY(hipMemcpyAsync(h_sc, sc, sizeof(h_sc), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_info, c->info.p, sizeof(h_info), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_tau.data(), w.v(EP_TAU), (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_v.data(), w.v(EP_V), (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h_info[0] != 0) { set_error("EPDTC: I + A not positive definite"); return EDRGP_ERR_NOT_PD; }
    double sum_log_tau = 0.0, sum_tau = 0.0, sum_v2 = 0.0;
    for (int i = 0; i < n; ++i) {
        sum_log_tau += std::log(h_tau[i]);
        sum_tau += h_tau[i];
        sum_v2 += h_v[i] * h_v[i] / h_tau[i];
    }
    if (lml)
        *lml = -0.5 * (double)n * std::log(2.0 * M_PI) + 0.5 * sum_log_tau - 0.5 * sum_v2 -
               0.5 * (var_sum(c) * sum_tau - h_sc[SC_TRA]) - h_sc[SC_SUMLOGL] + 0.5 * h_sc[SC_ALPHASQ] + c->ep_logZt;
    double* Bi = c->mats[EM_BI].d();
    double* E1 = c->mats[EM_E1].d();
    double* E2 = c->mats[EM_E2].d();
    double* T1 = c->mats[EM_T1].d();
    if (need_post || need_grad) TRY(gram_of_lower(c, M, LBi, Bi));
    if (need_post) {
        // woodbury_inv = Lm^-T (I - B^-1) Lm^-1
        TRY(c->wv.reserve((size_t)M * sizeof(double)));
        TRY(c->winv.reserve((size_t)M * M * sizeof(double)));
        HIP_TRY(hipMemsetAsync(E1, 0, (size_t)M * M * sizeof(double), c->stream));
        TRY(add_diag(c, E1, M, M, 1.0));
        TRY(combine3(c, M, 1.0, E1, -1.0, Bi, 0.0, nullptr, 0.0, nullptr, E2));
        TRY(gemm(c, false, false, M, M, M, 1.0, E2, M, Li, M, 0.0, T1, M));
        TRY(gemm(c, true, false, M, M, M, 1.0, Li, M, T1, M, 0.0, c->winv.d(), M));
        HIP_TRY(hipMemcpyAsync(c->wv.d(), wv, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->have_post = true;
        c->post_dense = false;
        c->P = M;
    }
    if (!need_grad) return EDRGP_OK;

    // DB = B^-1 + (LB^-T c1)(LB^-T c1)^T ; dL/dKmm = Lm^-T (I/2 - DB/2 - A/2) Lm^-1 ; S2 = Lm^-T (I - DB) Lm^-1 / 2
    double* dKmm = c->mats[EM_DKMM].d();
    double* S2 = c->mats[EM_S2].d();
    double* Tmm = c->mats[EM_TMM].d();
    HIP_TRY(hipMemsetAsync(T1, 0, (size_t)M * M * sizeof(double), c->stream));
    TRY(add_diag(c, T1, M, M, 1.0));                                                  // I
    TRY(combine3(c, M, 0.5, T1, -0.5, Bi, -0.5, A, -0.5, t2, E1));
    TRY(combine3(c, M, 0.5, T1, -0.5, Bi, 0.0, nullptr, -0.5, t2, E2));
    TRY(gemm(c, false, false, M, M, M, 1.0, E1, M, Li, M, 0.0, T1, M));
    TRY(gemm(c, true, false, M, M, M, 1.0, Li, M, T1, M, 0.0, dKmm, M));
    TRY(gemm(c, false, false, M, M, M, 1.0, E2, M, Li, M, 0.0, T1, M));
    TRY(gemm(c, true, false, M, M, M, 1.0, Li, M, T1, M, 0.0, S2, M));
    // dL/dKnm = v wv^T + 2 diag(tau) K_nm S2 : the second term into W, the first joins in the element-wise pass
    TRY(c->T.reserve((size_t)n * std::max(M, Wd) * sizeof(double)));
    double* Wn = c->W.d();
    double* Tn = c->T.d();
    TRY(scale_rows(c, n, M, w.Knm, w.v(EP_TAU), 1, Tn));
    TRY(gemm(c, false, false, n, M, M, 2.0, Tn, M, S2, M, 0.0, Wn, M));
    const KSpec ks = make_kspec(c);
    HIP_TRY(hipMemsetAsync(sc + SC_DVAR_NM, 0, EDRGP_MAX_PARTS * sizeof(double), c->stream));
    TRY(c->gt_nm.reserve((size_t)np * M * Wd * sizeof(double)));
    TRY(c->gt_mm.reserve((size_t)2 * np * M * Wd * sizeof(double)));
    for (int p = 0; p < np; ++p) {
        TRY(grad_elementwise(c, ks, p, c->X.d(), w.v(EP_V), n, c->Z.d(), M, Wn, M, 1.0, wv, Tn, M, sc + SC_DVAR_NM + p));
        TRY(gemm(c, true, false, M, Wd, n, 1.0, Tn, M, c->Xt.d(), Wd, 0.0, c->gt_nm.d() + (size_t)p * M * Wd, Wd));
        TRY(grad_elementwise(c, ks, p, c->Z.d(), nullptr, M, c->Z.d(), M, dKmm, M, 0.0, nullptr, Tmm, M, sc + SC_DVAR_NM + p));
        TRY(gemm(c, true, false, M, Wd, M, 1.0, Tmm, M, c->Zt.d(), Wd, 0.0, c->gt_mm.d() + (size_t)(2 * p) * M * Wd, Wd));
        TRY(gemm(c, false, false, M, Wd, M, 1.0, Tmm, M, c->Zt.d(), Wd, 0.0, c->gt_mm.d() + (size_t)(2 * p + 1) * M * Wd, Wd));
    }
    std::vector<double> gn((size_t)np * M * Wd), gm((size_t)2 * np * M * Wd), Zh((size_t)M * D);
    double h_dv[EDRGP_MAX_PARTS];
    HIP_TRY(hipMemcpyAsync(h_dv, sc + SC_DVAR_NM, sizeof(h_dv), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(gn.data(), c->gt_nm.d(), gn.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(gm.data(), c->gt_mm.d(), gm.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(Zh.data(), c->Z.d(), Zh.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    dvar->assign(np, 0.0);
    dlen->assign((size_t)np * D, 0.0);
    dZ->assign((size_t)M * D, 0.0);
    for (int p = 0; p < np; ++p) {
        (*dvar)[p] = h_dv[p] - 0.5 * sum_tau;                 // + sum dL/dKdiag, dL/dKdiag = -tau / 2
        const double* Gn = &gn[(size_t)p * M * Wd];
        const double* Gt = &gm[(size_t)(2 * p) * M * Wd];
        const double* GN = &gm[(size_t)(2 * p + 1) * M * Wd];
        for (int q = 0; q < D; ++q) {
            const double l = c->lens[(size_t)p * D + q];
            const double il2 = 1.0 / (l * l), il3 = il2 / l;
            double s = 0.0;
            for (int m = 0; m < M; ++m) {
                const double z = Zh[(size_t)m * D + q];
                const double *a = Gn + (size_t)m * Wd, *b = Gt + (size_t)m * Wd, *d = GN + (size_t)m * Wd;
                s += (a[D + 1 + q] + a[D] * z * z - 2.0 * z * a[q]) + (b[D + 1 + q] + b[D] * z * z - 2.0 * z * b[q]);
                (*dZ)[(size_t)m * D + q] += il2 * ((z * a[D] - a[q]) + (z * b[D] - b[q]) + (z * d[D] - d[q]));
            }
            (*dlen)[(size_t)p * D + q] = -il3 * s;
        }
    }
    return EDRGP_OK;
}

}  // namespace edrgp
