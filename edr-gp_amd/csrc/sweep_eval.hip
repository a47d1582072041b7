// BASELINE config 4 -- one candidate projection x a grid of isotropic length-scales, evaluated as a BATCH.
//
// The unit of work of the sweep is one pass of edrgp/base.py:460-461 (fit the GP on the projected rows, read the
// posterior-mean gradients at the training rows) at fixed hyper-parameters: Titsias bound + woodbury vector + the
// d x d Gram matrix J^T J of the mean Jacobian J.  edrgp_amd.sweep used to run the G length-scales of a candidate one
// after another, each paying the whole latency-bound O(M^3) chain (two factorisations of 16 dependent steps at
// M = 1024) and a full kernel-matrix build.  Here, for the resident rows and inducing inputs:
//   * the UNSCALED squared distances R2 = |x - z|^2 (N x M and M x M) are formed ONCE (an isotropic length-scale
//     only rescales them: GPy Stationary._scaled_dist divides the unscaled distance by l); every K_g is then one
//     element-wise pass  var k(R2 / l_g^2)  with the column sums K_g^T y out of the same pass;
//   * the G matrices Kmm_g are factored and inverted in the SAME launches (potrf_inv_lower_batched), and so are the
//     G matrices B_g = I + A_g later: a batch costs about what one matrix does;
//   * the row passes (Phi_g, or K_g Lm_g^-T and its Gram matrix where the pivot-ratio rule asks for the stable form)
//     run back to back with no host synchronisation in between; there are two waits in all.
// Results are those of the per-unit path (same kernels, same statistics form per scale): tests/test_sweep.py.
#include "eval_common.h"
#include "kfun.h"
#include <cstring>

namespace edrgp {

namespace {

// K[z][n][m] = var k(R2[n][m] il2[z]) (+ jitter where n == m, for square matrices), z = blockIdx.z.
// A workgroup owns 64 rows x 512 columns (thread: two adjacent columns, 16-byte accesses) and, when y is given,
// leaves the partial column sums  sum_rows K y  of its rows in bpart[row tile][column].  H (optional, same layout as
// one K): h(r) = k'(r) / r of the same pass, what the mean Jacobian of a non-RBF kernel multiplies (kfun.h); K may
// then be null (the gradient pass of a grid whose K_g were not kept).
__global__ __launch_bounds__(256) void k_from_r2_kernel(const double* __restrict__ R2, int64_t rows, int M,
                                                       const double* __restrict__ il2, double var, int kind,
                                                       double jitter, const double* __restrict__ y,
                                                       double* __restrict__ K, long long strideK,
                                                       double* __restrict__ bpart, double* __restrict__ H) {
    const int c0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 2;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const double s = il2[blockIdx.z];
    double* Kz = (K != nullptr) ? K + (long long)blockIdx.z * strideK : nullptr;
    if (c0 >= M) return;
    const bool two = (c0 + 1 < M);
    const bool vec = two && ((M & 1) == 0);
    double b0 = 0.0, b1 = 0.0;
    const int nr = (int)min((int64_t)64, rows - r0);
    for (int r = 0; r < nr; ++r) {
        const size_t off = (size_t)(r0 + r) * M + c0;
        double q0, q1 = 0.0;
        if (vec) {
            const double2 q = *reinterpret_cast<const double2*>(R2 + off);
            q0 = q.x; q1 = q.y;
        } else {
            q0 = R2[off];
            if (two) q1 = R2[off + 1];
        }
        double k0, k1 = 0.0, h0, h1 = 0.0;
        edrgp_kfun(kind, var, q0 * s, k0, h0);
        if (two) edrgp_kfun(kind, var, q1 * s, k1, h1);
        if (jitter != 0.0) {
            if (r0 + r == c0) k0 += jitter;
            if (r0 + r == c0 + 1) k1 += jitter;
        }
        if (Kz != nullptr) {
            if (vec) {
                *reinterpret_cast<double2*>(Kz + off) = double2{k0, k1};
            } else {
                Kz[off] = k0;
                if (two) Kz[off + 1] = k1;
            }
        }
        if (H != nullptr) {
            if (vec) {
                *reinterpret_cast<double2*>(H + off) = double2{h0, h1};
            } else {
                H[off] = h0;
                if (two) H[off + 1] = h1;
            }
        }
        if (y != nullptr) {
            const double yr = y[r0 + r];
            b0 = fma(k0, yr, b0);
            b1 = fma(k1, yr, b1);
        }
    }
    if (bpart != nullptr) {
        bpart[(size_t)blockIdx.y * M + c0] = b0;
        if (two) bpart[(size_t)blockIdx.y * M + c0 + 1] = b1;
    }
}

int k_from_r2(Ctx* c, const double* R2, int64_t rows, int M, const double* il2_dev, int G, double var, int kind,
              double jitter, const double* y, double* K, long long strideK, double* bvec, double* H = nullptr) {
    const int tiles_r = (int)((rows + 63) / 64);
    double* bpart = nullptr;
    if (y != nullptr) {
        TRY(c->bpart.reserve((size_t)tiles_r * M * sizeof(double)));
        bpart = c->bpart.d();
    }
    dim3 grid((unsigned)((M + 511) / 512), (unsigned)tiles_r, (unsigned)G);
    hipLaunchKernelGGL(k_from_r2_kernel, grid, dim3(256), 0, c->stream, R2, rows, M, il2_dev, var, kind, jitter, y, K, strideK,
                       bpart, H);
    HIP_TRY(hipGetLastError());
    if (y != nullptr) TRY(col_reduce(c, bpart, tiles_r, M, bvec));
    return EDRGP_OK;
}

// The same pass for a GROUP of up to SW_GZ length-scales: R2 is read once, every K_z of the group written and its
// column sums K_z^T y left in bpart[z][row tile][column].  (One R2 read per scale made the pass 3.2 GB per scale at
// config 4's unit shape; four scales per read: 2.0 GB.)  RBF grids whose K_g are all kept.
constexpr int SW_GZ = 4;
__global__ __launch_bounds__(256) void k_from_r2_group_kernel(const double* __restrict__ R2, int64_t rows, int M,
                                                             const double* __restrict__ il2, int nz, double var, int kind,
                                                             const double* __restrict__ y, double* __restrict__ K,
                                                             long long strideK, double* __restrict__ bpart,
                                                             long long strideB) {
    const int c0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 2;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    if (c0 >= M) return;
    const bool two = (c0 + 1 < M);
    const bool vec = two && ((M & 1) == 0);
    double sz[SW_GZ], b0[SW_GZ], b1[SW_GZ];
#pragma unroll
    for (int z = 0; z < SW_GZ; ++z) {
        sz[z] = (z < nz) ? il2[z] : 0.0;
        b0[z] = b1[z] = 0.0;
    }
    const int nr = (int)min((int64_t)64, rows - r0);
    for (int r = 0; r < nr; ++r) {
        const size_t off = (size_t)(r0 + r) * M + c0;
        double q0, q1 = 0.0;
        if (vec) {
            const double2 q = *reinterpret_cast<const double2*>(R2 + off);
            q0 = q.x; q1 = q.y;
        } else {
            q0 = R2[off];
            if (two) q1 = R2[off + 1];
        }
        const double yr = y[r0 + r];
#pragma unroll
        for (int z = 0; z < SW_GZ; ++z) {
            if (z >= nz) break;
            double k0, k1 = 0.0, h;
            edrgp_kfun(kind, var, q0 * sz[z], k0, h);
            if (two) edrgp_kfun(kind, var, q1 * sz[z], k1, h);
            double* Kz = K + (long long)z * strideK;
            if (vec) {
                *reinterpret_cast<double2*>(Kz + off) = double2{k0, k1};
            } else {
                Kz[off] = k0;
                if (two) Kz[off + 1] = k1;
            }
            b0[z] = fma(k0, yr, b0[z]);
            b1[z] = fma(k1, yr, b1[z]);
        }
    }
#pragma unroll
    for (int z = 0; z < SW_GZ; ++z) {
        if (z >= nz) break;
        double* bp = bpart + (long long)z * strideB + (size_t)blockIdx.y * M + c0;
        bp[0] = b0[z];
        if (two) bp[1] = b1[z];
    }
}

// K_z (z < nz <= SW_GZ; stride strideK) and b_z = K_z^T y (bvecs + z * strideBvec) out of one pass over R2
int k_from_r2_group(Ctx* c, const double* R2, int64_t rows, int M, const double* il2_dev, int nz, double var, int kind,
                    const double* y, double* K, long long strideK, double* bvecs, long long strideBvec) {
    const int tiles_r = (int)((rows + 63) / 64);
    const long long strideB = (long long)tiles_r * M;
    TRY(c->bpart.reserve((size_t)nz * strideB * sizeof(double)));
    dim3 grid((unsigned)((M + 511) / 512), (unsigned)tiles_r, 1);
    hipLaunchKernelGGL(k_from_r2_group_kernel, grid, dim3(256), 0, c->stream, R2, rows, M, il2_dev, nz, var, kind, y, K, strideK,
                       c->bpart.d(), strideB);
    HIP_TRY(hipGetLastError());
    for (int z = 0; z < nz; ++z) TRY(col_reduce(c, c->bpart.d() + (size_t)z * strideB, tiles_r, M, bvecs + (long long)z * strideBvec));
    return EDRGP_OK;
}

__global__ void fill_ones_kernel(double* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 1.0;
}

enum { SW_TRA = 0, SW_SUMLOG, SW_DATAFIT, SW_DIAGMIN, SW_DIAGMAX, SW_SLOTS = 8 };

}  // namespace

int sweep_scales(Ctx* c, int G, const double* ls, double var, double noise_var, int kind, double* bounds, double* grams,
                 double* gsums, int* modes) {
    if (!c->have_data || !c->have_Z) { set_error("sweep_scales needs set_data (or set_data_projected) and set_inducing first"); return EDRGP_ERR_STATE; }
    if (c->sharded()) { set_error("sweep_scales: the sweep partitions its evaluations over the ranks, detach the communicator"); return EDRGP_ERR_STATE; }
    const int D = c->D, M = c->M;
    const int64_t N = c->N;
    if (G < 1 || G > 64 || !ls || !bounds || !(var > 0.0) || !(noise_var >= 0.0)) { set_error("sweep_scales: bad arguments"); return EDRGP_ERR_INVALID; }
    if (kind != EDRGP_KERN_RBF && kind != EDRGP_KERN_MATERN32 && kind != EDRGP_KERN_MATERN52) { set_error("sweep_scales: RBF, Matern32 or Matern52"); return EDRGP_ERR_INVALID; }
    const bool rbf = (kind == EDRGP_KERN_RBF);
    if (!c->mfma_kbuild) { set_error("sweep_scales needs the MFMA kernel build (option mfma_kbuild)"); return EDRGP_ERR_STATE; }
    if (M <= 64 || M > 3072 || N > 0x7fffffffLL || N < M) { set_error("sweep_scales: needs 64 < M <= 3072 and N >= M"); return EDRGP_ERR_INVALID; }
    for (int g = 0; g < G; ++g) if (!(ls[g] > 0.0)) { set_error("sweep_scales: lengthscale must be positive"); return EDRGP_ERR_INVALID; }
    ++c->state_serial;
    c->grad_serial = 0;
    c->have_post = false;
    c->knm_train_valid = false;
    const double beta = 1.0 / std::max(noise_var, 1e-8);
    const size_t MM = (size_t)M * M;
    const size_t phi_stride = MM + (size_t)M + 8;
    const int64_t Npad = (N + 15) & ~(int64_t)15;

    // ---- workspaces
    TRY(c->sw_small.reserve(((size_t)EDRGP_MAX_DIM + (size_t)G * (2 + D + SW_SLOTS + 4 * (size_t)M)) * sizeof(double) + 2 * G * sizeof(int) + 64));
    double* ones = c->sw_small.d();                             // invl = 1: unscaled distances
    double* il2 = ones + EDRGP_MAX_DIM;                         // [G] 1 / l^2
    double* invl_g = il2 + G;                                   // [G][D] 1 / l (per-scale invl of the predict pass)
    double* scal = invl_g + (size_t)G * D;                      // [G][SW_SLOTS]
    double* vecs = scal + (size_t)G * SW_SLOTS;                 // [G][4][M]: v, c, u, wv
    int* info = reinterpret_cast<int*>(vecs + (size_t)G * 4 * M + G);   // [2][G]
    TRY(c->sw_r2.reserve(((size_t)(N + 16) * M + MM) * sizeof(double)));
    double* R2nm = c->sw_r2.d();
    double* R2mm = R2nm + (size_t)(N + 16) * M;
    TRY(c->sw_mats.reserve((size_t)G * (6 * MM + phi_stride) * sizeof(double)));
    double* Lm = c->sw_mats.d();                                // [G] Kmm_g -> its factor
    double* Li = Lm + (size_t)G * MM;                           // [G] Lm^-1
    double* Aw = Li + (size_t)G * MM;                           // [G] A_g
    double* Tw = Aw + (size_t)G * MM;                           // [G] Li Phi (scratch)
    double* LB = Tw + (size_t)G * MM;                           // [G] B_g -> its factor
    double* LBi = LB + (size_t)G * MM;                          // [G] LB^-1
    double* PhiB = LBi + (size_t)G * MM;                        // [G] [Phi | b]
    TRY(c->Knm.reserve((size_t)(N + 16) * M * sizeof(double)));
    // keep every K_g for the gradient pass when it fits (G x 1.6 GB at N = 200 000, M = 1024), else rebuild it
    bool keep_k = false;
    if (grams != nullptr) {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)G * (size_t)(N + 16) * M * sizeof(double);
        if (c->sw_kall.bytes >= need) keep_k = true;
        else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + c->sw_kall.bytes > need + (8ull << 30))
            keep_k = (c->sw_kall.reserve(need) == EDRGP_OK);
        (void)hipGetLastError();
    }
    const size_t k_stride = (size_t)(N + 16) * M;

    {
        std::vector<double> h((size_t)G * (1 + D));
        for (int g = 0; g < G; ++g) {
            h[g] = 1.0 / (ls[g] * ls[g]);
            for (int q = 0; q < D; ++q) h[(size_t)G + (size_t)g * D + q] = 1.0 / ls[g];
        }
        hipLaunchKernelGGL(fill_ones_kernel, dim3(1), dim3(256), 0, c->stream, ones, (int64_t)EDRGP_MAX_DIM);
        HIP_TRY(hipMemcpyAsync(il2, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));             // h is a local
    }

    // ---- unscaled squared distances, once
    {
        ProfScope ps(c, EDRGP_PROF_KBUILD);
        KSpec ks1;
        ks1.nparts = 1; ks1.D = D;
        for (int p = 0; p < EDRGP_MAX_PARTS; ++p) { ks1.kind[p] = 0; ks1.var[p] = 0.0; }
        ks1.var[0] = 1.0;
        ks1.invl = ones;
        c->kbuild_r2 = true;                                   // the build's epilogue stores r^2 instead of k(r^2)
        int st = kernel_matrix(c, ks1, c->Z.d(), M, c->Z.d(), M, R2mm, M, 1, 0.0, nullptr, nullptr);
        if (st == EDRGP_OK) st = kernel_matrix(c, ks1, c->X.d(), N, c->Z.d(), M, R2nm, M, 0, 0.0, nullptr, nullptr);
        c->kbuild_r2 = false;
        if (st != EDRGP_OK) return st;
        if (Npad > N) HIP_TRY(hipMemsetAsync(R2nm + (size_t)N * M, 0, (size_t)(Npad - N) * M * sizeof(double), c->stream));
    }

    // ---- phase 1 for every scale at once: Kmm_g + 1e-8 I, Lm_g, Lm_g^-1, pivot extremes
    {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        TRY(k_from_r2(c, R2mm, M, M, il2, G, var, kind, 1e-8, nullptr, Lm, (long long)MM, nullptr));
        TRY(potrf_inv_lower_batched(c, Lm, M, M, (long long)MM, info, Li, M, (long long)MM, G));
        for (int g = 0; g < G; ++g) TRY(diag_minmax(c, Lm + (size_t)g * MM, M, M, scal + (size_t)g * SW_SLOTS + SW_DIAGMIN));
    }
    std::vector<double> h_scal((size_t)G * SW_SLOTS);
    std::vector<int> h_info((size_t)2 * G);
    HIP_TRY(hipMemcpyAsync(h_scal.data(), scal, h_scal.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_info.data(), info, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // the pivot-ratio rule per scale; a Kmm_g that needs jitter is left to the per-unit path (mode -1)
    std::vector<int> mode((size_t)G, 1);
    for (int g = 0; g < G; ++g) {
        const double pr = h_scal[(size_t)g * SW_SLOTS + SW_DIAGMAX] / h_scal[(size_t)g * SW_SLOTS + SW_DIAGMIN];
        if (h_info[g] != 0) mode[g] = -1;
        else mode[g] = ((c->stats_mode == 2) || (c->stats_mode == 0 && pr * pr > pivot_ratio_limit(c, M))) ? 2 : 1;
    }

    // ---- row passes, back to back
    for (int g = 0; g < G; ++g) {
        // RBF: h(r) = -k(r), the kept K_g serves the gradient pass; Matern: K_g is scratch and h(r) is what is kept
        double* Kg = (keep_k && rbf) ? c->sw_kall.d() + (size_t)g * k_stride : c->Knm.d();
        double* Hg = (keep_k && !rbf) ? c->sw_kall.d() + (size_t)g * k_stride : nullptr;
        double* Phi = PhiB + (size_t)g * phi_stride;
        double* bvec = Phi + MM;
        // the kept K_g of an RBF grid are built SW_GZ scales per pass over R2, a scale that is left to the per-unit path
        // included (b_g = K_g^T y with them: used by the Gram-form scales, overwritten by the stable form's product for
        // the others)
        if (keep_k && rbf && g % SW_GZ == 0) {
            ProfScope ps(c, EDRGP_PROF_KBUILD);
            TRY(k_from_r2_group(c, R2nm, N, M, il2 + g, std::min(SW_GZ, G - g), var, kind, c->y.d(), Kg, (long long)k_stride,
                                bvec, (long long)phi_stride));
        }
        if (mode[g] < 0) continue;
        if (!(keep_k && rbf)) {
            ProfScope ps(c, EDRGP_PROF_KBUILD);
            TRY(k_from_r2(c, R2nm, N, M, il2 + g, 1, var, kind, 0.0, mode[g] == 1 ? c->y.d() : nullptr, Kg, 0,
                          mode[g] == 1 ? bvec : nullptr, Hg));
        }
        const double* S = Kg;
        if (mode[g] == 2) {
            // W = K_g Lm_g^-T with b = W^T y out of the same launch (sparse_eval.hip, phase 2)
            ProfScope ps(c, EDRGP_PROF_GRADGEMM);
            TRY(c->W.reserve((size_t)(N + 16) * M * sizeof(double)));
            TRY(stable_product(c, Kg, N, M, Li + (size_t)g * MM, c->y.d(), c->W.d(), bvec));
            S = c->W.d();
        }
        if (Npad > N) HIP_TRY(hipMemsetAsync(const_cast<double*>(S) + (size_t)N * M, 0, (size_t)(Npad - N) * M * sizeof(double), c->stream));
        TRY(syrk_tn(c, S, Npad, M, M, Phi, M, false));
    }

    // ---- phase 3 for every scale: A_g, B_g = I + A_g factored and inverted in one batch, the M-vectors, scalars
    {
        ProfScope ps(c, EDRGP_PROF_SMALL);
        HIP_TRY(hipMemsetAsync(scal, 0, (size_t)G * SW_SLOTS * sizeof(double), c->stream));
        // Gram form: A = beta Li (Phi Li^T); computed for the whole batch (uniform strides), the stable-form scales
        // then take A = beta Phi
        {
            GemmOpts o1 = kopt(K_LE_ROW);
            o1.batch = G; o1.strideA = (long long)MM; o1.strideB = (long long)phi_stride; o1.strideC = (long long)MM;
            TRY(gemm(c, false, false, M, M, M, 1.0, Li, M, PhiB, M, 0.0, Tw, M, o1));
            GemmOpts o2 = kopt(K_LE_COL, 1);
            o2.batch = G; o2.strideA = (long long)MM; o2.strideB = (long long)MM; o2.strideC = (long long)MM;
            TRY(gemm(c, false, true, M, M, M, beta, Tw, M, Li, M, 0.0, Aw, M, o2));
        }
        for (int g = 0; g < G; ++g) {
            if (mode[g] < 0) {                                  // keep the batch factorisation well defined: B = I
                HIP_TRY(hipMemsetAsync(LB + (size_t)g * MM, 0, MM * sizeof(double), c->stream));
                TRY(add_diag(c, LB + (size_t)g * MM, M, M, 1.0));
                continue;
            }
            double* Ag = Aw + (size_t)g * MM;
            double* Phi = PhiB + (size_t)g * phi_stride;
            double* v = vecs + (size_t)g * 4 * M;
            if (mode[g] == 2) {
                TRY(combine3(c, M, beta, Phi, 0.0, nullptr, 0.0, nullptr, 0.0, nullptr, Ag));
                HIP_TRY(hipMemcpyAsync(v, Phi + MM, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            } else {
                TRY(mirror_lower(c, Ag, M, M));
                TRY(gemv_n(c, M, M, 1.0, Li + (size_t)g * MM, M, Phi + MM, v));
            }
            TRY(trace_of(c, Ag, M, M, scal + (size_t)g * SW_SLOTS + SW_TRA));
            HIP_TRY(hipMemcpyAsync(LB + (size_t)g * MM, Ag, MM * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            TRY(add_diag(c, LB + (size_t)g * MM, M, M, 1.0));
        }
        TRY(potrf_inv_lower_batched(c, LB, M, M, (long long)MM, info + G, LBi, M, (long long)MM, G));
        for (int g = 0; g < G; ++g) {
            if (mode[g] < 0) continue;
            double* v = vecs + (size_t)g * 4 * M;
            double *vc = v + M, *vu = v + 2 * M, *vw = v + 3 * M;
            const double* LBg = LB + (size_t)g * MM;
            const double* LBig = LBi + (size_t)g * MM;
            TRY(sum_log_diag(c, LBg, M, M, scal + (size_t)g * SW_SLOTS + SW_SUMLOG));
            // c = beta LBi v ; u = LBi^T c ; wv = Li^T u
            TRY(gemv_n(c, M, M, beta, LBig, M, v, vc));
            TRY(gemv_t(c, M, M, 1.0, LBig, M, vc, vu));
            TRY(gemv_t(c, M, M, 1.0, Li + (size_t)g * MM, M, vu, vw));
            TRY(sum_sq(c, vc, M, scal + (size_t)g * SW_SLOTS + SW_DATAFIT));
        }
    }

    // ---- gradient Gram matrices: J = d mean / d x at the training rows (one N x D x M product on K_g, or on h_g(r))
    double* d_gram = nullptr;
    if (grams != nullptr) {
        TRY(c->g2.reserve(((size_t)G * ((size_t)D * D + D) + 8) * sizeof(double)));
        d_gram = c->g2.d();
        const int Wd = 2 * D + 1;
        TRY(c->W.reserve((size_t)(N + 16) * M * sizeof(double)));
        TRY(c->T.reserve((size_t)N * (Wd + D + 2) * sizeof(double)));
        double* GT = c->T.d();
        double* o_jac = GT + (size_t)N * Wd;
        for (int g = 0; g < G; ++g) {
            double* Gg = d_gram + (size_t)g * ((size_t)D * D + D);
            if (mode[g] < 0) { HIP_TRY(hipMemsetAsync(Gg, 0, ((size_t)D * D + D) * sizeof(double), c->stream)); continue; }
            double* Kg = keep_k ? c->sw_kall.d() + (size_t)g * k_stride : c->Knm.d();      // K_g (RBF) or h_g (Matern)
            if (!keep_k) {
                ProfScope ps(c, EDRGP_PROF_KBUILD);
                TRY(k_from_r2(c, R2nm, N, M, il2 + g, 1, var, kind, 0.0, nullptr, rbf ? Kg : nullptr, 0, nullptr,
                              rbf ? nullptr : Kg));
            }
            KSpec ks;
            ks.nparts = 1; ks.D = D;
            for (int p = 0; p < EDRGP_MAX_PARTS; ++p) { ks.kind[p] = 0; ks.var[p] = 0.0; }
            ks.kind[0] = kind; ks.var[0] = var;
            ks.invl = invl_g + (size_t)g * D;
            // (Matern: the Jacobian is H_g [diag(wv) Z | wv] on the matrix pipe, K_g itself is not read)
            TRY(predict_chunk(c, ks, c->X.d(), N, c->Z.d(), c->Zt.d(), M, vecs + (size_t)g * 4 * M + 3 * M, nullptr, var, Kg,
                              c->W.d(), GT, nullptr, nullptr, o_jac, true, rbf ? nullptr : Kg, !rbf));
            TRY(gemm(c, true, false, D, D, (int)N, 1.0, o_jac, D, o_jac, D, 0.0, Gg, D));
            TRY(col_sums(c, o_jac, N, D, Gg + (size_t)D * D));
        }
    }

    // ---- results
    HIP_TRY(hipMemcpyAsync(h_scal.data(), scal, h_scal.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_info.data(), info, (size_t)2 * G * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    std::vector<double> h_gram;
    if (grams != nullptr) {
        h_gram.resize((size_t)G * ((size_t)D * D + D));
        HIP_TRY(hipMemcpyAsync(h_gram.data(), d_gram, h_gram.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->drain_events();
    const double yy = c->yy_local, Nd = (double)N;
    for (int g = 0; g < G; ++g) {
        if (mode[g] > 0 && h_info[(size_t)G + g] != 0) mode[g] = -1;      // B_g needs jitter: per-unit path
        if (modes != nullptr) modes[g] = mode[g];
        if (mode[g] < 0) {
            bounds[g] = NAN;
        } else {
            const double* sg = &h_scal[(size_t)g * SW_SLOTS];
            bounds[g] = -0.5 * Nd * (std::log(2.0 * M_PI) - std::log(beta)) - 0.5 * beta * yy -
                        0.5 * (beta * Nd * var - sg[SW_TRA]) - sg[SW_SUMLOG] + 0.5 * sg[SW_DATAFIT];
        }
        if (grams != nullptr) {
            const double* src = &h_gram[(size_t)g * ((size_t)D * D + D)];
            std::memcpy(grams + (size_t)g * D * D, src, (size_t)D * D * sizeof(double));
            if (gsums != nullptr) std::memcpy(gsums + (size_t)g * D, src + (size_t)D * D, (size_t)D * sizeof(double));
        }
    }
    return EDRGP_OK;
}

}  // namespace edrgp
