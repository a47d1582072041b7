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

// Issue-pattern probes: P=1: 4x4 accumulator grid, distinct A/B registers, VGPR accumulators;
// P=2: same with AGPR accumulators; P=3: 16 accumulators, one A/B pair.
#define MFMA_V(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
// Issue-pattern probes of the GEMM inner loop.  Bits of F add one ingredient each per
// 64 MFMAs (= one k-tile): 1 = LDS fragment reads (8 per 16 MFMAs), 2 = workgroup barrier,
// 4 = 8 ds_write_b128 (staging stores), 8 = 8 streaming global_load_dwordx4.
// AG = 1: accumulators in AGPRs; ONE = 1: a single A/B pair.
// ORD: order of the 16 MFMAs of a k-step over the accumulator grid: 0 x-major, 1 y-major, 2 snake, 3 rotated
// (consecutive instructions share neither operand)
// The loop skeleton (F = 15) with the fragments of k-step s+1 read during the MFMAs of step s (register
// double-buffered fragments, +16 VGPRs); WR = k-step after which the staging stores are issued.
#ifdef EDRGP_PROBE_BUILD
// ---- loop-skeleton and issue-order microbenchmarks (probe builds only: DESIGN section 5 cites their numbers)
template <int WR>
__global__ __launch_bounds__(256, 2) void mfma_prefetch_kernel(int iters, double* sink, const double* __restrict__ gsrc,
                                                              long long gstride) {
    __shared__ __attribute__((aligned(16))) double lds[2][16][144];
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    for (int i = threadIdx.x; i < 2 * 16 * 144; i += 256) (&lds[0][0][0])[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    double a[2][4], b[2][4];
    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = d4{0, 0, 0, 0};
    const char* gbase = reinterpret_cast<const char*>(gsrc) + (long long)blockIdx.x * gstride;
    const unsigned goff = threadIdx.x * 16;
    double2 st[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) st[v] = double2{1.0, 2.0};
#pragma unroll
    for (int x = 0; x < 4; ++x) { a[0][x] = lds[0][lq][x * 16 + lr]; b[0][x] = lds[0][lq][64 + x * 16 + lr]; }
    for (int i = 0; i < iters; ++i) {
        const int buf = i & 1;
#pragma unroll
        for (int v = 0; v < 8; ++v) st[v] = *reinterpret_cast<const double2*>(gbase + goff + v * 4096);
        gbase += 32768;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk < 3) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    a[nxt][x] = lds[buf][(kk + 1) * 4 + lq][x * 16 + lr];
                    b[nxt][x] = lds[buf][(kk + 1) * 4 + lq][64 + x * 16 + lr];
                }
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) MFMA_V(acc[x][y], a[cur][x], b[cur][y]);
            if (kk == WR) {
#pragma unroll
                for (int v = 0; v < 8; ++v)
                    *reinterpret_cast<double2*>(&lds[buf ^ 1][(v * 2 + (threadIdx.x >> 7)) & 15][(threadIdx.x & 63) * 2]) = st[v];
            }
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 4; ++x) { a[0][x] = lds[buf ^ 1][lq][x * 16 + lr]; b[0][x] = lds[buf ^ 1][lq][64 + x * 16 + lr]; }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    d4 s = {0, 0, 0, 0};
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) s += acc[x][y];
    if (s[0] + s[1] + s[2] + s[3] + st[3].x == -1.0) sink[0] = s[0];
}

template <int F, int AG, int ONE, int ORD = 0>
__global__ __launch_bounds__(256, 2) void mfma_pattern_kernel(int iters, double* sink, const double* __restrict__ gsrc,
                                                             long long gstride) {
    __shared__ __attribute__((aligned(16))) double lds[2][16][144];
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    for (int i = threadIdx.x; i < 2 * 16 * 144; i += 256) (&lds[0][0][0])[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = 1.0 + 1e-3 * (lane + i); b[i] = 1.0 - 1e-3 * (lane + 2 * i); }
    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = d4{0, 0, 0, 0};
    const char* gbase = reinterpret_cast<const char*>(gsrc) + (long long)blockIdx.x * gstride;
    const unsigned goff = threadIdx.x * 16;
    double2 st[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) st[v] = double2{1.0, 2.0};
    for (int i = 0; i < iters; ++i) {            // one iteration = one k-tile = 64 MFMAs per wave
        const int buf = i & 1;
        if (F & 8) {
#pragma unroll
            for (int v = 0; v < 8; ++v) st[v] = *reinterpret_cast<const double2*>(gbase + goff + v * 4096);
            gbase += 32768;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (F & 1) {
#pragma unroll
                for (int x = 0; x < 4; ++x) { a[x] = lds[buf][kk * 4 + lq][x * 16 + lr]; b[x] = lds[buf][kk * 4 + lq][64 + x * 16 + lr]; }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int x = (ORD == 1) ? (i & 3) : (ORD == 3) ? (i & 3) : (i >> 2);
                const int y = (ORD == 1) ? (i >> 2) : (ORD == 2) ? (((i >> 2) & 1) ? 3 - (i & 3) : (i & 3))
                            : (ORD == 3) ? (((i >> 2) + (i & 3)) & 3) : (i & 3);
                if (AG) MFMA_A(acc[x][y], a[x], b[y]);
                else if (ONE) MFMA_V(acc[x][y], a[0], b[0]);
                else MFMA_V(acc[x][y], a[x], b[y]);
            }
        }
        if (F & 4) {
#pragma unroll
            for (int v = 0; v < 8; ++v)
                *reinterpret_cast<double2*>(&lds[buf ^ 1][(v * 2 + (threadIdx.x >> 7)) & 15][(threadIdx.x & 63) * 2]) = st[v];
        }
        if (F & 2) __syncthreads();
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    d4 s = {0, 0, 0, 0};
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) s += acc[x][y];
    if (s[0] + s[1] + s[2] + s[3] + st[3].x == -1.0) sink[0] = s[0];
}

// Big-tile skeleton: one wave per SIMD, 4x8 accumulator grid (64x128 per wave, 256 VGPRs of
// accumulators), per k-tile 128 MFMAs, 12 fragment reads per k-step, 12 streaming loads,
// 12 staging stores, one barrier.  PF = 1: fragments of k-step s+1 are read during step s.
template <int PF>
__global__ __launch_bounds__(256, 1) void mfma_bigtile_kernel(int iters, double* sink, const double* __restrict__ gsrc,
                                                             long long gstride) {
    __shared__ __attribute__((aligned(16))) double lds[2][16][400];
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    for (int i = threadIdx.x; i < 2 * 16 * 400; i += 256) (&lds[0][0][0])[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    d4 acc[4][8];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 8; ++y) acc[x][y] = d4{0, 0, 0, 0};
    const char* gbase = reinterpret_cast<const char*>(gsrc) + (long long)blockIdx.x * gstride;
    const unsigned goff = threadIdx.x * 16;
    double2 st[12];
#pragma unroll
    for (int v = 0; v < 12; ++v) st[v] = double2{1.0, 2.0};
    double a[2][4], b[2][8];
#pragma unroll
    for (int x = 0; x < 4; ++x) a[0][x] = lds[0][lq][x * 16 + lr];
#pragma unroll
    for (int y = 0; y < 8; ++y) b[0][y] = lds[0][lq][128 + y * 16 + lr];
    for (int i = 0; i < iters; ++i) {
        const int buf = i & 1;
#pragma unroll
        for (int v = 0; v < 12; ++v) st[v] = *reinterpret_cast<const double2*>(gbase + goff + v * 4096);
        gbase += 49152;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = PF ? (kk & 1) : 0, nxt = PF ? ((kk + 1) & 1) : 0;
            if (!PF) {
#pragma unroll
                for (int x = 0; x < 4; ++x) a[0][x] = lds[buf][kk * 4 + lq][x * 16 + lr];
#pragma unroll
                for (int y = 0; y < 8; ++y) b[0][y] = lds[buf][kk * 4 + lq][128 + y * 16 + lr];
            } else if (kk < 3) {
#pragma unroll
                for (int x = 0; x < 4; ++x) a[nxt][x] = lds[buf][(kk + 1) * 4 + lq][x * 16 + lr];
#pragma unroll
                for (int y = 0; y < 8; ++y) b[nxt][y] = lds[buf][(kk + 1) * 4 + lq][128 + y * 16 + lr];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 8; ++y) MFMA_V(acc[x][y], a[cur][x], b[cur][y]);
            if (kk == 1) {
#pragma unroll
                for (int v = 0; v < 12; ++v)
                    *reinterpret_cast<double2*>(&lds[buf ^ 1][(v + (threadIdx.x >> 7)) & 15][(threadIdx.x & 127) * 2]) = st[v];
            }
        }
        __syncthreads();
        if (PF) {
#pragma unroll
            for (int x = 0; x < 4; ++x) a[0][x] = lds[buf ^ 1][lq][x * 16 + lr];
#pragma unroll
            for (int y = 0; y < 8; ++y) b[0][y] = lds[buf ^ 1][lq][128 + y * 16 + lr];
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    d4 s = {0, 0, 0, 0};
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 8; ++y) s += acc[x][y];
    if (s[0] + s[1] + s[2] + s[3] + st[3].x == -1.0) sink[0] = s[0];
}

template <int F, int AG, int ONE, int ORD = 0>
static int run_pattern(Ctx* c, int blocks, int iters, const double* gs, long long gstride, double* tf) {
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) HIP_TRY(hipEventRecord(e0, c->stream));
        hipLaunchKernelGGL((mfma_pattern_kernel<F, AG, ONE, ORD>), dim3(blocks), dim3(256), 0, c->stream, iters, c->red.d(),
                           gs, gstride);
    }
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *tf = (double)blocks * 4.0 * iters * 64.0 * 2048.0 / (ms * 1e-3) * 1e-12;
    return EDRGP_OK;
}

int calib_mfma_patterns(Ctx* c, double* t) {
    TRY(c->red.reserve(64));
    const int iters = 600, blocks = c->num_cu * 2;
    // each workgroup streams its own 20 MB slice: (iters + 1) * 32 KiB
    const long long gstride = (long long)(iters + 2) * 32768;
    DBuf g;
    TRY(g.reserve((size_t)blocks * gstride));
    HIP_TRY(hipMemsetAsync(g.p, 0, g.bytes, c->stream));
    const double* gs = g.d();
    TRY((run_pattern<0, 0, 0>(c, blocks, iters, gs, gstride, &t[0])));      // 4x4 grid, registers only
    TRY((run_pattern<0, 1, 0>(c, blocks, iters, gs, gstride, &t[1])));      // AGPR accumulators
    TRY((run_pattern<0, 0, 1>(c, blocks, iters, gs, gstride, &t[2])));      // one A/B pair
    TRY((run_pattern<1, 0, 0>(c, blocks, iters, gs, gstride, &t[3])));      // + LDS fragment reads
    TRY((run_pattern<3, 0, 0>(c, blocks, iters, gs, gstride, &t[4])));      // + barrier per k-tile
    TRY((run_pattern<5, 0, 0>(c, blocks, iters, gs, gstride, &t[5])));      // + staging stores (no barrier)
    TRY((run_pattern<9, 0, 0>(c, blocks, iters, gs, gstride, &t[6])));      // + streaming global loads (no barrier)
    TRY((run_pattern<15, 0, 0>(c, blocks, iters, gs, gstride, &t[7])));     // everything: the GEMM loop skeleton
    for (int pf = 0; pf < 2; ++pf) {                                          // big-tile skeleton, 1 wave per SIMD
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        const int it2 = iters / 2;
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) HIP_TRY(hipEventRecord(e0, c->stream));
            if (pf) hipLaunchKernelGGL((mfma_bigtile_kernel<1>), dim3(c->num_cu), dim3(256), 0, c->stream, it2, c->red.d(), gs, gstride * 2);
            else hipLaunchKernelGGL((mfma_bigtile_kernel<0>), dim3(c->num_cu), dim3(256), 0, c->stream, it2, c->red.d(), gs, gstride * 2);
        }
        HIP_TRY(hipEventRecord(e1, c->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        t[8 + pf] = (double)c->num_cu * 4.0 * it2 * 128.0 * 2048.0 / (ms * 1e-3) * 1e-12;
    }
    g.release();
    return EDRGP_OK;
}

// issue order of the 16 MFMAs of a k-step: registers only and the full loop skeleton, per order
int calib_mfma_orders(Ctx* c, double* t) {
    TRY(c->red.reserve(64));
    const int iters = 600, blocks = c->num_cu * 2;
    const long long gstride = (long long)(iters + 2) * 32768;
    DBuf g;
    TRY(g.reserve((size_t)blocks * gstride));
    HIP_TRY(hipMemsetAsync(g.p, 0, g.bytes, c->stream));
    const double* gs = g.d();
    TRY((run_pattern<0, 0, 0, 0>(c, blocks, iters, gs, gstride, &t[0])));
    TRY((run_pattern<0, 0, 0, 1>(c, blocks, iters, gs, gstride, &t[1])));
    TRY((run_pattern<0, 0, 0, 2>(c, blocks, iters, gs, gstride, &t[2])));
    TRY((run_pattern<0, 0, 0, 3>(c, blocks, iters, gs, gstride, &t[3])));
    TRY((run_pattern<15, 0, 0, 0>(c, blocks, iters, gs, gstride, &t[4])));
    TRY((run_pattern<15, 0, 0, 1>(c, blocks, iters, gs, gstride, &t[5])));
    TRY((run_pattern<15, 0, 0, 2>(c, blocks, iters, gs, gstride, &t[6])));
    TRY((run_pattern<15, 0, 0, 3>(c, blocks, iters, gs, gstride, &t[7])));
    for (int wr = 0; wr < 2; ++wr) {                 // fragment prefetch; staging stores after k-step 1 or 3
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) HIP_TRY(hipEventRecord(e0, c->stream));
            if (wr == 0) hipLaunchKernelGGL((mfma_prefetch_kernel<1>), dim3(blocks), dim3(256), 0, c->stream, iters, c->red.d(), gs, gstride);
            else hipLaunchKernelGGL((mfma_prefetch_kernel<3>), dim3(blocks), dim3(256), 0, c->stream, iters, c->red.d(), gs, gstride);
        }
        HIP_TRY(hipEventRecord(e1, c->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        t[10 + wr] = (double)blocks * 4.0 * iters * 64.0 * 2048.0 / (ms * 1e-3) * 1e-12;
    }
    TRY((run_pattern<0, 0, 0, 0>(c, c->num_cu, iters, gs, gstride, &t[8])));       // one workgroup per CU: a lone wave per SIMD
    TRY((run_pattern<0, 0, 1, 0>(c, c->num_cu, iters, gs, gstride, &t[9])));
    g.release();
    return EDRGP_OK;
}

#endif  // EDRGP_PROBE_BUILD

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
