// Calibration micro-benchmarks: fp64 MFMA issue rate (the local hardware guide
// lists no fp64 matrix peak) and HBM copy bandwidth; plus the fragment-layout
// probe the parity tests use to pin the v_mfma_f64_16x16x4_f64 lane maps.
#include "common.h"

namespace edrgp {

typedef double d4 __attribute__((ext_vector_type(4)));

// Inline asm pins the accumulators in VGPRs; hipcc otherwise shuttles the
// loop-carried tiles between AGPRs and VGPRs every iteration.
#define MFMA_F64_ASM(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

__global__ __launch_bounds__(256) void mfma_f64_rate_kernel(int iters, double* sink) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + 1e-3 * lane, b = 1.0 - 1e-3 * lane;
    d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        MFMA_F64_ASM(c0); MFMA_F64_ASM(c1); MFMA_F64_ASM(c2); MFMA_F64_ASM(c3);
        MFMA_F64_ASM(c0); MFMA_F64_ASM(c1); MFMA_F64_ASM(c2); MFMA_F64_ASM(c3);
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const d4 s = c0 + c1 + c2 + c3;
    if (s[0] + s[1] + s[2] + s[3] == -1.0) sink[0] = s[0];   // never true; keeps the chain live
}

int calib_mfma_f64(Ctx* c, double* tflops) {
    TRY(c->red.reserve(64));
    const int iters = 10000;
    const int blocks = c->num_cu * 2;              // 2 x 256 threads per CU = 2 waves per SIMD
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, c->stream, 200, c->red.d());   // warm-up
    HIP_TRY(hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, c->stream, iters, c->red.d());
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double flops = (double)blocks * 4.0 * iters * 8.0 * (2.0 * 16 * 16 * 4);
    *tflops = flops / (ms * 1e-3) * 1e-12;
    return EDRGP_OK;
}

__global__ __launch_bounds__(256) void copy16_kernel(const double2* __restrict__ src, double2* __restrict__ dst,
                                                     size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

int calib_hbm_copy(Ctx* c, double* gbps) {
    const size_t bytes = (size_t)1 << 30;          // 1 GiB each way, beyond the 256 MiB Infinity Cache
    DBuf a, b;
    TRY(a.reserve(bytes));
    TRY(b.reserve(bytes));
    HIP_TRY(hipMemsetAsync(a.p, 1, bytes, c->stream));
    const size_t n = bytes / sizeof(double2);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(copy16_kernel, dim3(c->num_cu * 8), dim3(256), 0, c->stream,
                       static_cast<const double2*>(a.p), static_cast<double2*>(b.p), n);
    HIP_TRY(hipEventRecord(e0, c->stream));
    const int reps = 5;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(copy16_kernel, dim3(c->num_cu * 8), dim3(256), 0, c->stream,
                           static_cast<const double2*>(a.p), static_cast<double2*>(b.p), n);
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    a.release();
    b.release();
    *gbps = 2.0 * bytes * reps / (ms * 1e-3) * 1e-9;
    return EDRGP_OK;
}

// One MFMA on caller-supplied operands: A is 16x4 (row-major), B is 4x16,
// C (16x16) = A B written through the documented C/D lane map.
__global__ void mfma_probe_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                  double* __restrict__ C) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];       // A[i = l&15][k = l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)];      // B[k = l>>4][j = l&15]
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

int mfma_layout_probe(Ctx* c, const double* A, const double* B, double* C) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, c->stream, A, B, C);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

}  // namespace edrgp
