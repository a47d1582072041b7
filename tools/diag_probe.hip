// Standalone probe of the 64 x 64 diagonal-block kernel body (edr-gp_amd/csrc/chol_diag.h): correctness against a
// CPU Cholesky / triangular inverse in the same program, phase stamps in shader cycles, the accuracy of v_rsq_f64 and
// of the third-order refinement the pivots use.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I edr-gp_amd/csrc tools/diag_probe.hip -o tools/_bin/diag_probe
//   (on the GPU box)  tools/_bin/diag_probe [jb=64] [reps=200]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "chol_diag.h"

using namespace edrgp;

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe_kernel(double* A, int lda, int n, double* Lout, double* Dinv, double* inv_out,
                                                    int* info, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) double lds[DIAG_LDS];
    diag_block_body<true>(A, lda, n, 0, Dinv, inv_out, n, info, stamps, Lout, n, 0, lds);
}

__global__ void rsq_probe(const double* x, double* y0, double* y1, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double pv = x[i];
    const double a = __builtin_amdgcn_rsq(pv);
    const double t = pv * a;
    const double e = fma(-t, a, 1.0);
    const double c3 = fma(0.375, e, 0.5);
    const double ye = a * e;
    y0[i] = a;
    y1[i] = fma(ye, c3, a);
}

// dependent-chain latencies of the operations a pivot is made of, one wave alone on a CU (cycles per operation)
// s_memtime pinned in the instruction stream: the value chain `x` passes through the stamp
__device__ __forceinline__ unsigned long long stamp_now(double& x) {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 0" : "+v"(x));
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(x));
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

__global__ void latency_probe(unsigned long long* out, double* sink, double seed) {
    __shared__ double buf[128];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-9, acc = 0.0;
    buf[lane] = x; buf[lane + 64] = x;
    __syncthreads();
    unsigned long long t0, t1;
    constexpr int N = 256;
    // v_fma_f64 chain
    t0 = stamp_now(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = fma(x, 0.999999, 1e-9);
    asm volatile("" : "+v"(x));
    t1 = stamp_now(x);
    if (lane == 0) out[0] = (t1 - t0);
    // v_mul_f64 chain
    t0 = stamp_now(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * 1.0000001;
    asm volatile("" : "+v"(x));
    t1 = stamp_now(x);
    if (lane == 0) out[1] = (t1 - t0);
    // v_rsq_f64 chain
    t0 = stamp_now(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x);
    asm volatile("" : "+v"(x));
    t1 = stamp_now(x);
    if (lane == 0) out[2] = (t1 - t0);
    // readlane (2 x b32) -> fma with the scalar
    t0 = stamp_now(x);
#pragma unroll
    for (int i = 0; i < N; ++i) { const double s = readlane_f64(x, i & 63); x = fma(x, 0.5, s * 1e-3); }
    asm volatile("" : "+v"(x));
    t1 = stamp_now(x);
    if (lane == 0) out[3] = (t1 - t0);
    // LDS write -> broadcast read -> fma
    t0 = stamp_now(x);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        buf[lane] = x;
        wave_lds_order();
        const double s = buf[(i * 7) & 63];
        x = fma(x, 0.5, s * 1e-3);
        wave_lds_order();
    }
    asm volatile("" : "+v"(x));
    t1 = stamp_now(x);
    if (lane == 0) out[4] = (t1 - t0);
    // independent FMAs (issue rate): 8 accumulators
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    t0 = stamp_now(x);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
        a0 = fma(a0, 0.999, 1e-9); a1 = fma(a1, 0.999, 1e-9); a2 = fma(a2, 0.999, 1e-9); a3 = fma(a3, 0.999, 1e-9);
        a4 = fma(a4, 0.999, 1e-9); a5 = fma(a5, 0.999, 1e-9); a6 = fma(a6, 0.999, 1e-9); a7 = fma(a7, 0.999, 1e-9);
    }
    x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    t1 = stamp_now(x);
    if (lane == 0) out[5] = (t1 - t0);
    // dependent MFMA chain
    d4 c = {x, x, x, x};
    t0 = stamp_now(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, c, 0, 0, 0);
    x = c[0] + c[1] + c[2] + c[3];
    t1 = stamp_now(x);
    if (lane == 0) out[6] = (t1 - t0) * 4;      // scaled to N = 256
    // LDS broadcast reads of old data (no dependent write), 8 in flight -> fma
    t0 = stamp_now(x);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
        double s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] = buf[64 + ((i * 8 + u) & 63)];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(s[u], 1e-3, acc);
    }
    x += acc;
    t1 = stamp_now(x);
    if (lane == 0) out[7] = (t1 - t0);
    // ds_read_b128 broadcast (two values per instruction), 8 in flight
    {
        const double2* b2 = reinterpret_cast<const double2*>(buf + 64);
        t0 = stamp_now(x);
#pragma unroll
        for (int i = 0; i < N / 16; ++i) {
            double2 s[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = b2[(i * 8 + u) & 31];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc = fma(s[u].x, 1e-3, acc); acc = fma(s[u].y, 1e-3, acc); }
        }
        x += acc;
        t1 = stamp_now(x);
        if (lane == 0) out[8] = (t1 - t0);       // per VALUE: / 256
    }
    // v_readlane_b32 alone: 8 independent per group, then one dependent VALU
    {
        t0 = stamp_now(x);
        int xi = __double2loint(x), acci = 0;
#pragma unroll
        for (int i = 0; i < N / 8; ++i) {
            int s[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = __builtin_amdgcn_readlane(xi, (i * 8 + u) & 63);
#pragma unroll
            for (int u = 0; u < 8; ++u) acci += s[u];
            xi ^= acci;
        }
        x += (double)xi * 1e-30;
        t1 = stamp_now(x);
        if (lane == 0) out[9] = (t1 - t0);
    }
    // DPP row_newbcast mov (b32), dependent chain
    {
        t0 = stamp_now(x);
        int xi = __double2loint(x);
#pragma unroll
        for (int i = 0; i < N; ++i) xi = __builtin_amdgcn_update_dpp(0, xi, 0x155, 0xf, 0xf, false) + 1;
        x += (double)xi * 1e-30;
        t1 = stamp_now(x);
        if (lane == 0) out[10] = (t1 - t0);
    }
    // ds_write_b64 (lane-contiguous) alone, 8 in flight
    {
        t0 = stamp_now(x);
#pragma unroll
        for (int i = 0; i < N / 8; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) buf[(lane + u) & 63] = x + u;
            wave_lds_order();
        }
        t1 = stamp_now(x);
        if (lane == 0) out[11] = (t1 - t0);
    }
    sink[lane] = x + acc + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c[0] + c[1] + c[2] + c[3];
}

int main(int argc, char** argv) {
    {
        unsigned long long* dl;
        double* ds;
        CHECK(hipMalloc(&dl, 16 * 8)); CHECK(hipMalloc(&ds, 64 * 8));
        unsigned long long hl[16];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(latency_probe, dim3(1), dim3(64), 0, 0, dl, ds, 1.2345);
            CHECK(hipMemcpy(hl, dl, 128, hipMemcpyDeviceToHost));
        }
        printf("cycles per op, one wave (s_memtime ticks / 256): dependent v_fma_f64 %.1f | v_mul_f64 %.1f | v_rsq_f64 %.1f | readlane x2 + 2 dependent VALU %.1f | "
               "LDS write -> broadcast read -> 2 VALU %.1f | independent v_fma_f64 %.1f | dependent MFMA f64 16x16x4 %.1f | LDS broadcast read (8 in flight) + fma %.1f\n",
               hl[0] / 256.0, hl[1] / 256.0, hl[2] / 256.0, hl[3] / 256.0, hl[4] / 256.0, hl[5] / 256.0, hl[6] / 256.0, hl[7] / 256.0);
        printf("ds_read_b128 broadcast + 2 fma, per value %.1f | v_readlane_b32 (8 independent) + add %.1f | DPP row_newbcast mov + add, dependent %.1f | ds_write_b64 (8 in flight) %.1f\n",
               hl[8] / 256.0, hl[9] / 256.0, hl[10] / 256.0, hl[11] / 256.0);
    }

    const int jb = argc > 1 ? atoi(argv[1]) : 64;
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    const int n = jb;
    // --- accuracy of v_rsq_f64 and of the refined value
    {
        const int m = 1 << 20;
        std::vector<double> x(m), y0(m), y1(m);
        srand(1);
        for (int i = 0; i < m; ++i) x[i] = std::exp((rand() / (double)RAND_MAX - 0.5) * 60.0) * (1.0 + rand() / (double)RAND_MAX);
        double *dx, *d0, *d1;
        CHECK(hipMalloc(&dx, m * 8)); CHECK(hipMalloc(&d0, m * 8)); CHECK(hipMalloc(&d1, m * 8));
        CHECK(hipMemcpy(dx, x.data(), m * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(rsq_probe, dim3(m / 256), dim3(256), 0, 0, dx, d0, d1, m);
        CHECK(hipMemcpy(y0.data(), d0, m * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(y1.data(), d1, m * 8, hipMemcpyDeviceToHost));
        double e0 = 0, e1 = 0;
        for (int i = 0; i < m; ++i) {
            const long double ref = 1.0L / sqrtl((long double)x[i]);
            e0 = std::max(e0, (double)fabsl((y0[i] - ref) / ref));
            e1 = std::max(e1, (double)fabsl((y1[i] - ref) / ref));
        }
        printf("v_rsq_f64 max rel err %.3e (2^%.1f); after the third-order step %.3e (2^%.1f)\n", e0, std::log2(e0), e1, std::log2(e1));
        hipFree(dx); hipFree(d0); hipFree(d1);
    }
    // --- the block
    std::vector<double> A(n * n), L(n * n, 0.0), Li(n * n, 0.0);
    srand(7);
    {
        std::vector<double> Q(n * n);
        for (auto& v : Q) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0;
                for (int k = 0; k < n; ++k) s += Q[i * n + k] * Q[j * n + k];
                A[i * n + j] = s / n + (i == j ? 0.05 : 0.0);
            }
    }
    // CPU reference (long double accumulation)
    for (int j = 0; j < n; ++j) {
        long double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= (long double)L[j * n + k] * L[j * n + k];
        L[j * n + j] = (double)sqrtl(d);
        for (int i = j + 1; i < n; ++i) {
            long double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= (long double)L[i * n + k] * L[j * n + k];
            L[i * n + j] = (double)(s / L[j * n + j]);
        }
    }
    for (int c = 0; c < n; ++c)
        for (int i = c; i < n; ++i) {
            long double s = (i == c) ? 1.0L : 0.0L;
            for (int k = c; k < i; ++k) s -= (long double)L[i * n + k] * Li[k * n + c];
            Li[i * n + c] = (double)(s / L[i * n + i]);
        }
    double *dA, *dL, *dD, *dI;
    int* dinfo;
    unsigned long long* dS;
    CHECK(hipMalloc(&dA, n * n * 8)); CHECK(hipMalloc(&dL, n * n * 8)); CHECK(hipMalloc(&dD, 64 * 64 * 8));
    CHECK(hipMalloc(&dI, n * n * 8)); CHECK(hipMalloc(&dinfo, 4)); CHECK(hipMalloc(&dS, 16 * 8));
    CHECK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemset(dL, 0xff, n * n * 8)); CHECK(hipMemset(dD, 0, 64 * 64 * 8)); CHECK(hipMemset(dI, 0, n * n * 8));
    CHECK(hipMemset(dinfo, 0, 4)); CHECK(hipMemset(dS, 0, 16 * 8));
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 0, 0, dA, n, n, dL, dD, dI, dinfo, dS);
    CHECK(hipDeviceSynchronize());
    std::vector<double> gL(n * n), gD(64 * 64), gI(n * n);
    int info = -1;
    unsigned long long st[16];
    CHECK(hipMemcpy(gL.data(), dL, n * n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(gD.data(), dD, 64 * 64 * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(gI.data(), dI, n * n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(st, dS, 16 * 8, hipMemcpyDeviceToHost));
    double eL = 0, eI = 0, eD = 0, up = 0, mL = 0, mI = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (j <= i) {
                eL = std::max(eL, std::fabs(gL[i * n + j] - L[i * n + j]));
                eI = std::max(eI, std::fabs(gI[i * n + j] - Li[i * n + j]));
                eD = std::max(eD, std::fabs(gD[i * 64 + j] - Li[i * n + j]));
                mL = std::max(mL, std::fabs(L[i * n + j]));
                mI = std::max(mI, std::fabs(Li[i * n + j]));
            } else {
                up = std::max(up, std::fabs(gL[i * n + j]));
                // blocks of the inverse above the diagonal are never written (zero-filled by the host); inside the
                // diagonal 16 x 16 blocks the kernel writes the (zero) upper halves
                eI = std::max(eI, std::fabs(gI[i * n + j]));
            }
        }
    printf("jb=%d info=%d  |L - ref| %.2e (max |L| %.2e)  |inv - ref| %.2e  |Dinv - ref| %.2e (max %.2e)  upper(L) %.1e\n", jb, info, eL,
           mL, eI, eD, mI, up);
    printf("stamps (cycles from entry): loaded %llu | barrier w0..w3 %llu %llu %llu %llu | inverse done w0..w3 %llu %llu %llu %llu | exit %llu\n",
           st[1] - st[0], st[8] - st[0], st[9] - st[0], st[10] - st[0], st[11] - st[0], st[12] - st[0], st[13] - st[0],
           st[14] - st[0], st[15] - st[0], st[5] - st[0]);
    // a non-positive pivot must be reported at its position
    {
        std::vector<double> B = A;
        const int badj = std::min(37, n - 1);
        B[badj * n + badj] = -1.0;
        CHECK(hipMemcpy(dA, B.data(), n * n * 8, hipMemcpyHostToDevice));
        CHECK(hipMemset(dinfo, 0, 4));
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 0, 0, dA, n, n, dL, dD, dI, dinfo, (unsigned long long*)nullptr);
        CHECK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
        printf("bad pivot at %d -> info %d (expected %d)\n", badj, info, badj + 1);
        CHECK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
    }
    // launch time (events over back-to-back launches)
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 0, 0, dA, n, n, dL, dD, dI, dinfo, (unsigned long long*)nullptr);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 0, 0, dA, n, n, dL, dD, dI, dinfo, (unsigned long long*)nullptr);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%.2f us per launch (%d back-to-back launches, includes the ~1.5 us kernel boundary)\n", ms * 1e3 / reps, reps);
    return 0;
}
