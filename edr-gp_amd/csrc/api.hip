// C-ABI entry points (include/edrgp_hip.h), the dense exact-GP pipeline (GPy
// ExactGaussianInference.inference) and the prediction pass (Posterior._raw_predict,
// GP.predictive_gradients).  The sparse pipeline (VarDTC.inference) lives in sparse_eval.hip.
#include "eval_common.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <cstdlib>

namespace edrgp {

static thread_local std::string g_err;

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

// Pinned host staging owned by the context: copies to and from it are truly asynchronous (a pageable buffer makes
// hipMemcpyAsync stage and wait: ~90 us in front of the result copies of every evaluation).
int Ctx::ensure_pinned(size_t doubles) {
    if (h_pin != nullptr && h_pin_cap >= doubles) return EDRGP_OK;
    HIP_TRY(hipStreamSynchronize(stream));                 // nothing in flight may still use the old block
    if (stream2) HIP_TRY(hipStreamSynchronize(stream2));
    if (h_pin) (void)hipHostFree(h_pin);
    h_pin = nullptr;
    h_pin_cap = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_pin), doubles * sizeof(double), hipHostMallocDefault));
    h_pin_cap = doubles;
    return EDRGP_OK;
}

// Rows of K_nm kept resident at a time.  An explicit "chunk_rows" is taken as given.  Automatic: ALL local rows when five
// N x M fp64 work arrays (K_nm, h(r), K_nm Lm^-T, the gradient contraction's T, one spare) fit into half of the memory
// this context can still get and N M stays below 2^31 elements -- one Phi launch per evaluation, K_nm (and h) stay
// resident for the gradient pass and for predictions on the training rows (config 3 whole on one GPU, 1M x 2048:
// bound+gradients 212 -> 201 ms, predict_gradient's Gram matrix 19.8 -> 9.6 ms) -- otherwise chunks of 131072 rows.
int64_t Ctx::chunk_rows_for(int M) {
    if (chunk_rows > 0) return chunk_rows;
    if (auto_chunk_N == N && auto_chunk_M == M && auto_chunk > 0) return auto_chunk;
    auto_chunk = 131072;
    if (N > auto_chunk && M > 0) {
        size_t free_b = 0, total_b = 0;
        const size_t held = Knm.bytes + Hnm.bytes + W.bytes + T.bytes;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double need = 5.0 * (double)N * M * sizeof(double);
            if (need <= 0.5 * (double)(free_b + held) && (double)(N + 16) * M < 2147483648.0 - 16777216.0) auto_chunk = N;
        }
        (void)hipGetLastError();
    }
    auto_chunk_N = N;
    auto_chunk_M = M;
    return auto_chunk;
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

struct DenseOut {
    double lml = 0.0;
    std::vector<double> dvar, dlen;
    double dnoise = 0.0;
};

// kern.update_gradients_full(dL_dK, X) for the resident rows: d/dvariance and d/dlengthscale of every additive part
// from dL/dK (n x n; Tn is an n x n scratch).  Shared by the exact GP and the EP classifier (ep_eval.hip).
int dense_kernel_gradients(Ctx* c, const double* dK, double* Tn, std::vector<double>* dvar, std::vector<double>* dlen) {
    const int n = (int)c->N, D = c->D, np = c->nparts, Wd = 2 * D + 1;
    const KSpec ks = make_kspec(c);
    double* sc = c->scal.d();
    HIP_TRY(hipMemsetAsync(sc + SC_DVAR_NM, 0, EDRGP_MAX_PARTS * sizeof(double), c->stream));
    TRY(c->gt_nm.reserve((size_t)np * n * Wd * sizeof(double)));
    for (int p = 0; p < np; ++p) {
        TRY(grad_elementwise(c, ks, p, c->X.d(), nullptr, n, c->X.d(), n, dK, n, 0.0, nullptr, Tn, n,
                             sc + SC_DVAR_NM + p));
        TRY(gemm(c, true, false, n, Wd, n, 1.0, Tn, n, c->Xt.d(), Wd, 0.0, c->gt_nm.d() + (size_t)p * n * Wd, Wd));
    }
    std::vector<double> g((size_t)np * n * Wd), Xh((size_t)n * D);
    double h_dv[EDRGP_MAX_PARTS];
    HIP_TRY(hipMemcpyAsync(h_dv, sc + SC_DVAR_NM, sizeof(h_dv), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(g.data(), c->gt_nm.d(), g.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(Xh.data(), c->X.d(), Xh.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    dvar->assign(np, 0.0);
    dlen->assign((size_t)np * D, 0.0);
    for (int p = 0; p < np; ++p) {
        (*dvar)[p] = h_dv[p];
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
            (*dlen)[(size_t)p * D + q] = -il3 * s;
        }
    }
    return EDRGP_OK;
}

// ExactGaussianInference.inference on the resident (X, y).
static int dense_eval(Ctx* c, bool need_grad, bool need_post, DenseOut* out) {
    if (!c->have_data || !c->have_kernel) {
        set_error("dense evaluation needs set_data and set_kernel first");
        return EDRGP_ERR_STATE;
    }
    if (c->sharded()) { set_error("the dense exact GP does not shard: replicas only"); return EDRGP_ERR_STATE; }
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
        TRY(potrf_inv_lower(c, L, n, n, info, Li, n));
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
    TRY(dense_kernel_gradients(c, dK, Tn, &out->dvar, &out->dlen));
    out->dnoise = 0.5 * (h_sc[SC_ALPHASQ] - h_sc[SC_TRKYI]);
    return EDRGP_OK;
}

// Prediction on a device-resident block of inputs; outputs to host pointers.
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
    int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(c->chunk_rows_for(Mp), (int64_t)(1LL << 28) / std::max(1, Mp)));
    chunk = std::min<int64_t>(chunk, Ns);
    TRY(c->Knm.reserve((size_t)chunk * Mp * sizeof(double)));
    TRY(c->W.reserve((size_t)chunk * Mp * sizeof(double)));
    // two output slots (mean Jacobian, mean, variance of a chunk): chunk k + 1 is computed while chunk k's outputs cross
    // PCIe on the side stream (config 3 whole on one GPU: 256 MB of Jacobian, 48 -> .. ms for the 1M rows)
    TRY(c->T.reserve((size_t)chunk * (Wd + 2 * (D + 2)) * sizeof(double)));
    double* GT = c->T.d();
    const size_t slot_doubles = (size_t)chunk * (D + 2);
    for (int i = 0; i < 2; ++i)
        if (c->ev_pred[i] == nullptr) HIP_TRY(hipEventCreateWithFlags(&c->ev_pred[i], hipEventDisableTiming));
    const double kdiag = var_sum(c);
    // the training rows right after a sparse evaluation with the same theta / Z: K(X, Z) is still resident
    const bool k_train = c->knm_train_valid && !c->post_dense && Xs_dev == c->X.d() && Ns == c->N && Mp == c->M;
    if (!k_train) c->knm_train_valid = false;          // Knm is about to be overwritten chunk by chunk
    // h(r) beside k for the mean Jacobian of non-RBF kernels (one more chunk x Mp array)
    const bool rbf1 = (ks.nparts == 1 && ks.kind[0] == EDRGP_KERN_RBF);
    const bool want_h = (jac_h != nullptr || gram_h != nullptr) && !rbf1 && c->mfma_kbuild;
    const bool h_train = k_train && c->hnm_train_valid;
    if (want_h && !h_train) {
        c->hnm_train_valid = false;
        TRY(c->Hnm.reserve((size_t)chunk * Mp * sizeof(double)));
    }
    const bool want_gram = (gram_h != nullptr);
    double* d_gram = nullptr;
    double* d_ones = nullptr;
    if (want_gram) {
        TRY(c->g2.reserve(((size_t)D * D + 2 * (size_t)D + 8) * sizeof(double)));
        d_gram = c->g2.d();
        d_ones = d_gram + (size_t)D * D + D;              // D doubles of scratch: a later chunk's column sums
    }
    auto slot_jac = [&](int64_t k) { return GT + (size_t)chunk * Wd + (size_t)(k & 1) * slot_doubles; };
    auto compute = [&](int64_t k) -> int {
        const int64_t r0 = k * chunk, rows = std::min<int64_t>(chunk, Ns - r0);
        double* o_jac = slot_jac(k);
        double* o_mean = o_jac + (size_t)chunk * D;
        double* o_var = o_mean + chunk;
        TRY(predict_chunk(c, ks, Xs_dev + (size_t)r0 * D, rows, Pp, Pt, Mp, c->wv.d(), c->winv.d(), kdiag,
                          k_train ? c->Knm.d() + (size_t)r0 * Mp : c->Knm.d(), c->W.d(), GT, mean_h ? o_mean : nullptr,
                          var_h ? o_var : nullptr, (jac_h || want_gram) ? o_jac : nullptr, k_train,
                          want_h ? (h_train ? c->Hnm.d() + (size_t)r0 * Mp : c->Hnm.d()) : nullptr, h_train));
        if (want_gram) {
            TRY(gemm(c, true, false, D, D, (int)rows, 1.0, o_jac, D, o_jac, D, r0 > 0 ? 1.0 : 0.0, d_gram, D));
            // column sums of J: a two-stage reduction (la_vec.hip) instead of a GEMM against a ones vector
            if (r0 == 0) {
                TRY(col_sums(c, o_jac, rows, D, d_gram + (size_t)D * D));
            } else {
                TRY(col_sums(c, o_jac, rows, D, d_ones));
                TRY(vec_axpy(c, D, 1.0, d_ones, d_gram + (size_t)D * D));
            }
        }
        HIP_TRY(hipEventRecord(c->ev_pred[k & 1], c->stream));
        return EDRGP_OK;
    };
    const int64_t nck = (Ns + chunk - 1) / chunk;
    const bool any_copy = (mean_h != nullptr) || (var_h != nullptr) || (jac_h != nullptr);
    TRY(compute(0));
    for (int64_t k = 0; k < nck; ++k) {
        // chunk k + 1 is queued first (its slot was drained two iterations ago: the copies below block the host)
        if (k + 1 < nck) TRY(compute(k + 1));
        if (!any_copy) continue;
        const int64_t r0 = k * chunk, rows = std::min<int64_t>(chunk, Ns - r0);
        const double* o_jac = slot_jac(k);
        const double* o_mean = o_jac + (size_t)chunk * D;
        const double* o_var = o_mean + chunk;
        HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_pred[k & 1], 0));
        if (mean_h) HIP_TRY(hipMemcpyAsync(mean_h + r0, o_mean, rows * sizeof(double), hipMemcpyDeviceToHost, c->stream2));
        if (var_h) HIP_TRY(hipMemcpyAsync(var_h + r0, o_var, rows * sizeof(double), hipMemcpyDeviceToHost, c->stream2));
        if (jac_h)
            HIP_TRY(hipMemcpyAsync(jac_h + (size_t)r0 * D, o_jac, (size_t)rows * D * sizeof(double),
                                   hipMemcpyDeviceToHost, c->stream2));
        HIP_TRY(hipStreamSynchronize(c->stream2));      // slot k & 1 is free again
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
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
    if (c->sharded()) edrgp_comm_destroy(ctx);
    if (c->h_ar) (void)hipHostFree(c->h_ar);
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
    for (int i = 0; i < 2; ++i)
        if (c->ev_pred[i]) (void)hipEventDestroy(c->ev_pred[i]);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
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
    // The product's options (include/edrgp_hip.h documents each).  None of them changes WHAT is computed beyond the
    // documented numerical choices (statistics form, precision of the row pass).
    if (!std::strcmp(key, "chunk_rows")) c->chunk_rows = (value <= 0.0) ? 0 : std::max<int64_t>(64, (int64_t)value);
    else if (!std::strcmp(key, "stats_mode")) c->stats_mode = (int)value;          // 0 auto, 1 gram, 2 stable
    else if (!std::strcmp(key, "pivot_ratio_max")) c->pivot_ratio_max = value;
    else if (!std::strcmp(key, "overlap")) c->overlap = (int)value;
    else if (!std::strcmp(key, "sync_free")) c->sync_free = (value != 0.0);
    else if (!std::strcmp(key, "posterior_reuse")) c->posterior_reuse = (value != 0.0);
    else if (!std::strcmp(key, "pad_inducing")) c->pad_inducing = (value != 0.0);
    else if (!std::strcmp(key, "profile")) c->profile = (value != 0.0);
    else if (!std::strcmp(key, "f32_ratio_max")) c->f32_ratio_max = value;
    else if (!std::strcmp(key, "f32_guard")) c->f32_guard = value;
    else if (!std::strcmp(key, "f32_phi")) { c->f32_phi = (value != 0.0); c->phi32_bad = false; }
    else if (!std::strcmp(key, "precision")) {
        // 64 (default): fp64 throughout.  32: the N-proportional row pass in fp32 (la_gemm.hip, float instantiation)
        if ((int)value != 32 && (int)value != 64) { set_error("precision must be 32 or 64"); return EDRGP_ERR_INVALID; }
        c->precision = (int)value;
        c->have_post = false;
        c->last_mode = 0;
        c->phi32_bad = false;
    }
#ifdef EDRGP_PROBE_BUILD
    // Probe builds only (EDRGP_BUILD_VARIANT=<name> python edr-gp_amd/build.py -> libedrgp_hip_<name>.so, tools/):
    // A/B switches of settled design decisions, launch-geometry knobs, and timing probes whose results are wrong
    // (gemm_alias, kbuild_dbg, dbg_gemm_*).  The product library does not know these keys.
    else if (!std::strcmp(key, "syrk_splits")) c->syrk_splits = (int)value;
    else if (!std::strcmp(key, "trace")) c->trace = (value != 0.0);
    else if (!std::strcmp(key, "gemm_dephase")) c->gemm_dephase = (int)value;
    else if (!std::strcmp(key, "gemm_alias")) c->gemm_alias = (int)value;
    else if (!std::strcmp(key, "gemm_persist")) c->gemm_persist = (int)value;
    else if (!std::strcmp(key, "syrk_persist")) c->syrk_persist = (int)value;
    else if (!std::strcmp(key, "kmm_early")) c->kmm_early = (int)value;
    else if (!std::strcmp(key, "side_gemm_pad")) c->side_gemm_pad = (int)value;
    else if (!std::strcmp(key, "skinny_split")) c->skinny_split = (int)value;
    else if (!std::strcmp(key, "h_resident")) c->h_resident = (int)value;
    else if (!std::strcmp(key, "gemm_narrow")) c->gemm_narrow = (int)value;
    else if (!std::strcmp(key, "grad_finish_dev")) c->grad_finish_dev = (int)value;
    else if (!std::strcmp(key, "gemm_interleave")) c->gemm_interleave = (int)value;
    else if (!std::strcmp(key, "mid_split")) c->mid_split = (int)value;
    else if (!std::strcmp(key, "dbg_gemm_kmode")) c->dbg_gemm_kmode = (int)value;
    else if (!std::strcmp(key, "dbg_gemm_tri")) c->dbg_gemm_tri = (int)value;
    else if (!std::strcmp(key, "dbg_gemm_tile")) c->dbg_gemm_tile = (int)value;
    else if (!std::strcmp(key, "dbg_gemm_reps")) c->dbg_gemm_reps = (int)value;
    else if (!std::strcmp(key, "fuse_grad")) c->fuse_grad = (value != 0.0);
    else if (!std::strcmp(key, "tri_tile")) g_tri_tile = (int)value;
    else if (!std::strcmp(key, "mfma_kbuild")) c->mfma_kbuild = (value != 0.0);
    else if (!std::strcmp(key, "kbuild_dbg")) c->kbuild_dbg = (int)value;
    else if (!std::strcmp(key, "kbuild_tile")) c->kbuild_tile = (int)value;
    else if (!std::strcmp(key, "kbuild_pad")) c->kbuild_pad = std::max(0, std::min(65536, (int)value));
    else if (!std::strcmp(key, "rbf_epilogue")) c->rbf_epilogue = (value != 0.0);
    else if (!std::strcmp(key, "chol_fused")) c->chol_fused = (value != 0.0);
    else if (!std::strcmp(key, "chol_cap_grid")) c->chol_cap_grid = (value != 0.0);
    else if (!std::strcmp(key, "chol_inv_fused")) c->chol_inv_fused = (value != 0.0);
    else if (!std::strcmp(key, "chol_rhs")) c->chol_rhs = (value != 0.0);
    else if (!std::strcmp(key, "gemm_pair")) c->gemm_pair = (value != 0.0);
    else if (!std::strcmp(key, "prebuild")) c->prebuild = (value != 0.0);
    else if (!std::strcmp(key, "syrk_xmap")) c->syrk_xmap = (value != 0.0);
    else if (!std::strcmp(key, "syrk_diag_opt")) c->syrk_diag_opt = (value != 0.0);
#endif
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
    c->ep_begun = c->ep_valid = false;
    c->base_y_valid = false;                    // y no longer belongs to the resident base rows of set_data_projected
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
        c->base_y_valid = true;
    } else if (c->base_N != N || c->base_D != Dbase || !c->base_y_valid) {
        // (edrgp_set_data overwrites y: the resident targets are then another data set's, even at the same N)
        set_error("set_data_projected: no resident base rows / targets of shape %lld x %d", (long long)N, Dbase);
        return EDRGP_ERR_STATE;
    }
    c->ep_begun = c->ep_valid = false;          // EP sites belong to the rows they were fitted on
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
    // Z is a borrowed host pointer: it goes through the context's pinned staging, so the upload is asynchronous and
    // the call returns without waiting for the device (the optimiser sends Z before every evaluation)
    const size_t zoff = 64 + (size_t)EDRGP_MAX_PARTS * EDRGP_MAX_DIM;
    TRY(c->ensure_pinned(zoff + (size_t)M * D));
    std::memcpy(c->h_pin + zoff, Z, (size_t)M * D * sizeof(double));
    HIP_TRY(hipMemsetAsync(c->Z.d(), 0, Mcap * D * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->Zt.d(), 0, Mcap * (2 * D + 1) * sizeof(double), c->stream));
    HIP_TRY(hipMemcpyAsync(c->Z.d(), c->h_pin + zoff, (size_t)M * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->Z_host.assign(Z, Z + (size_t)M * D);
    TRY(build_xtilde(c, c->Z.d(), M, D, c->Zt.d()));
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
    // stream-ordered copy out of pinned staging: kernels queued earlier on the stream still read the previous
    // values, later ones the new ones; no wait for the device
    TRY(c->ensure_pinned(64 + (size_t)EDRGP_MAX_PARTS * EDRGP_MAX_DIM));
    std::memcpy(c->h_pin + 64, invl.data(), invl.size() * sizeof(double));
    HIP_TRY(hipMemcpyAsync(c->invl.d(), c->h_pin + 64, invl.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
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

int edrgp_ep_begin(edrgp_ctx* ctx, const double* y) {
    CTX(ctx);
    if (!y) { set_error("ep_begin: labels required"); return EDRGP_ERR_INVALID; }
    return ep_begin(c, y);
}

int edrgp_ep_sweep(edrgp_ctx* ctx, const int* order, double eta, double delta, double* tau_diff, double* v_diff) {
    CTX(ctx);
    return ep_sweep(c, order, eta, delta, tau_diff, v_diff);
}

int edrgp_ep_finish(edrgp_ctx* ctx, double* log_Z_tilde) {
    CTX(ctx);
    return ep_finish(c, log_Z_tilde);
}

int edrgp_ep_get_sites(edrgp_ctx* ctx, double* tau, double* v) {
    CTX(ctx);
    return ep_sites(c, tau, v, nullptr, nullptr, 0.0);
}

int edrgp_ep_set_sites(edrgp_ctx* ctx, const double* tau, const double* v, double log_Z_tilde) {
    CTX(ctx);
    if (!tau || !v) { set_error("ep_set_sites: tau and v required"); return EDRGP_ERR_INVALID; }
    return ep_sites(c, nullptr, nullptr, tau, v, log_Z_tilde);
}

int edrgp_ep_lml_grad(edrgp_ctx* ctx, double* lml, double* dvar, double* dlen) {
    CTX(ctx);
    std::vector<double> gv, gl;
    const bool need_grad = (dvar || dlen);
    TRY(ep_eval(c, need_grad, false, lml, &gv, &gl));
    if (need_grad) {
        if (dvar) std::memcpy(dvar, gv.data(), gv.size() * sizeof(double));
        if (dlen) std::memcpy(dlen, gl.data(), gl.size() * sizeof(double));
    }
    return EDRGP_OK;
}

int edrgp_ep_posterior(edrgp_ctx* ctx, double* lml) {
    CTX(ctx);
    return ep_eval(c, false, true, lml, nullptr, nullptr);
}

int edrgp_epdtc_begin(edrgp_ctx* ctx, const double* y) {
    CTX(ctx);
    if (!y) { set_error("epdtc_begin: labels required"); return EDRGP_ERR_INVALID; }
    return epdtc_begin(c, y);
}

int edrgp_epdtc_sweep(edrgp_ctx* ctx, const int* order, double eta, double delta, double* tau_diff, double* v_diff) {
    CTX(ctx);
    return epdtc_sweep(c, order, eta, delta, tau_diff, v_diff);
}

int edrgp_epdtc_lml_grad(edrgp_ctx* ctx, double* lml, double* dvar, double* dlen, double* dZ) {
    CTX(ctx);
    std::vector<double> gv, gl, gz;
    const bool need_grad = (dvar || dlen || dZ);
    TRY(epdtc_eval(c, need_grad, false, lml, &gv, &gl, &gz));
    if (need_grad) {
        if (dvar) std::memcpy(dvar, gv.data(), gv.size() * sizeof(double));
        if (dlen) std::memcpy(dlen, gl.data(), gl.size() * sizeof(double));
        if (dZ) std::memcpy(dZ, gz.data(), gz.size() * sizeof(double));
    }
    return EDRGP_OK;
}

int edrgp_epdtc_posterior(edrgp_ctx* ctx, double* lml) {
    CTX(ctx);
    return epdtc_eval(c, false, true, lml, nullptr, nullptr, nullptr);
}

int edrgp_sweep_scales(edrgp_ctx* ctx, int nscales, const double* lengthscales, double variance, double noise_var,
                       int kind, double* bounds, double* grams, double* gsums, int* modes) {
    CTX(ctx);
    return sweep_scales(c, nscales, lengthscales, variance, noise_var, kind, bounds, grams, gsums, modes);
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
            rows2d = (size_t)std::min<int64_t>(c->N, c->chunk_rows_for(c->M_work > 0 ? c->M_work : c->M));
            n = rows2d * M;
            if (c->last_mode == 3) {          // fp32 row pass: widen the resident chunk
                if ((int64_t)n > cap) { set_error("edrgp_get: buffer too small"); return EDRGP_ERR_INVALID; }
                std::vector<float> tmp(n);
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
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
            if (cap >= 4) buf[3] = (double)c->retries;     // sync-free evaluations repeated by the careful driver so far
            return EDRGP_OK;
        default: set_error("edrgp_get: unknown selector %d", which); return EDRGP_ERR_INVALID;
    }
    if (!src) { set_error("edrgp_get: object %d not computed yet", which); return EDRGP_ERR_STATE; }
    if ((int64_t)n > cap) { set_error("edrgp_get: buffer too small (%lld < %zu)", (long long)cap, n); return EDRGP_ERR_INVALID; }
    // both streams: some of these objects (Lm, Lm^-1, Kmm^-1 ...) are written on the side stream; every evaluation joins
    // it into the main stream before it returns, the second wait makes this accessor independent of that
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
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
    // options dbg_gemm_kmode / dbg_gemm_tri / dbg_gemm_tile: the zero-skipping / lower-tiles-only / tile-size variants
    // the evaluation uses for its M x M x M products (tools/mid_gemm_probe.py)
    GemmOpts go = kopt(c->dbg_gemm_kmode, c->dbg_gemm_tri);
    if (c->dbg_gemm_tile > 0) go.force_tile = c->dbg_gemm_tile;
    int s = gemm(c, transA != 0, transB != 0, M, N, K, alpha, dA.d(), lda, dB.d(), ldb, beta, dC.d(), ldc, go);
    for (int rep = 1; rep < c->dbg_gemm_reps && s == EDRGP_OK; ++rep)
        s = gemm(c, transA != 0, transB != 0, M, N, K, alpha, dA.d(), lda, dB.d(), ldb, beta, dC.d(), ldc, go);
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
    // on the context's stream: a null-stream memset is not ordered against this (non-blocking) stream's kernels
    HIP_TRY(hipMemsetAsync(dC.p, 0, (size_t)M * ldc * sizeof(float), c->stream));
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

// Cholesky factor and its inverse out of the same launches (la_chol.hip::potrf_inv_lower)
int edrgp_dbg_potrf_inv(edrgp_ctx* ctx, double* A, int n, int* info, double* Linv) {
    CTX(ctx);
    if (!A || !Linv || n <= 0) { set_error("dbg_potrf_inv: bad input"); return EDRGP_ERR_INVALID; }
    DBuf dA, dI;
    TRY(dA.reserve((size_t)n * n * sizeof(double)));
    TRY(dI.reserve((size_t)n * n * sizeof(double)));
    TRY(c->info.reserve(4 * sizeof(int)));
    HIP_TRY(hipMemcpy(dA.d(), A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
    // NaN pattern: every entry must be written.  On the context's stream: a null-stream memset is not ordered against
    // this (non-blocking) stream's kernels and could land on top of entries the factorisation had already written (seen
    // once as a failing test_potrf_with_inverse[3072] in a long test process)
    HIP_TRY(hipMemsetAsync(dI.d(), 0xff, (size_t)n * n * sizeof(double), c->stream));
    TRY(potrf_inv_lower(c, dA.d(), n, n, static_cast<int*>(c->info.p), dI.d(), n));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(A, dA.d(), (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(Linv, dI.d(), (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
    int h = 0;
    HIP_TRY(hipMemcpy(&h, c->info.p, sizeof(int), hipMemcpyDeviceToHost));
    if (info) *info = h;
    return EDRGP_OK;
}

// Times `reps` runs of the production factorisation-with-inverse (potrf_inv_lower) alone on the n x n SPD matrix A:
// each run starts from a fresh device copy of A, events bracket the factorisation only.
int edrgp_dbg_potrf_time(edrgp_ctx* ctx, const double* A, int n, int reps, double* avg_ms) {
    CTX(ctx);
    if (!A || n <= 0 || reps < 1 || !avg_ms) { set_error("dbg_potrf_time: bad input"); return EDRGP_ERR_INVALID; }
    DBuf d0, dA, dI;
    const size_t bytes = (size_t)n * n * sizeof(double);
    TRY(d0.reserve(bytes));
    TRY(dA.reserve(bytes));
    TRY(dI.reserve(bytes));
    TRY(c->info.reserve(4 * sizeof(int)));
    HIP_TRY(hipMemcpy(d0.d(), A, bytes, hipMemcpyHostToDevice));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    double total = 0.0;
    int s = EDRGP_OK;
    for (int r = -1; r < reps && s == EDRGP_OK; ++r) {      // r = -1: untimed warm-up
        HIP_TRY(hipMemcpyAsync(dA.d(), d0.d(), bytes, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipEventRecord(e0, c->stream));
        s = potrf_inv_lower(c, dA.d(), n, n, static_cast<int*>(c->info.p), dI.d(), n);
        HIP_TRY(hipEventRecord(e1, c->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 0) total += ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->drain_events();
    *avg_ms = total / reps;
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
#ifdef EDRGP_PROBE_BUILD
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
#endif
    if (mfma_f64_tflops) TRY(calib_mfma_f64(c, mfma_f64_tflops));
    if (hbm_copy_gbps) TRY(calib_hbm_copy(c, hbm_copy_gbps));
    return EDRGP_OK;
}

}  // extern "C"
