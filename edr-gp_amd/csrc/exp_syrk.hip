// Experimental Phi = K^T K kernels for A/B measurements against the production
// LDS-staged kernel (la_gemm.hip).  Reached only through edrgp_dbg_syrk_variant.
// Aligned shapes only: M % 128 == 0, rows % 16 == 0, ld even.
#include "common.h"
#include <cstdlib>

namespace edrgp {

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void fill_pattern_kernel(double* __restrict__ K, size_t n, unsigned long long seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ULL + seed;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
        // seed 0: constant operands (lowest switching activity); otherwise uniform [0, 1) like kernel values
        K[i] = (seed == 0ULL) ? 0.5 : (double)(x >> 11) * (1.0 / 9007199254740992.0);
    }
}

__device__ __forceinline__ void tri_tile(int t, int& ti, int& tj) {
    ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while ((long long)ti * (ti + 1) / 2 > t) --ti;
    tj = t - (int)((long long)ti * (ti + 1) / 2);
}

__device__ __forceinline__ int xcd_remap(int bid, int nt) {
    const int q = nt >> 3, r = nt & 7, x = bid & 7, in = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + in;
}

// ---- variant 1: no LDS, no barriers.  Every wave streams its own MFMA fragments
// straight from global memory (L1/L2) with a two-stage register ring.
template <int KS>   // k-steps (of 4 rows) per stage
__global__ __launch_bounds__(256, 2) void syrk_direct_kernel(const double* __restrict__ K, int rows, int M, int ld,
                                                             double* __restrict__ slabs, long long slabStride,
                                                             int splits, unsigned long long* __restrict__ clk) {
    unsigned long long t_cyc0 = 0, t_real0 = 0;
    const bool stamp = (clk != nullptr) && (threadIdx.x == 0) && (blockIdx.x == 3) && (blockIdx.y == 1);
    if (stamp) { t_cyc0 = __builtin_amdgcn_s_memtime(); t_real0 = __builtin_amdgcn_s_memrealtime(); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
    int ti, tj;
    tri_tile(xcd_remap(blockIdx.x, gridDim.x), ti, tj);
    const int kt_total = rows / 4;                       // k-steps
    const int per = ((kt_total + splits - 1) / splits + KS - 1) / KS * KS;
    const int ks0 = blockIdx.y * per;
    const int ks1 = min(kt_total, ks0 + per);
    const double* pa = K + (size_t)(ti * 128 + wm * 64 + lr);
    const double* pb = K + (size_t)(tj * 128 + wn * 64 + lr);

    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

    double a0[KS][4], b0[KS][4], a1[KS][4], b1[KS][4];
    auto load = [&](double (&a)[KS][4], double (&b)[KS][4], int ks) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = (ks + s) * 4 + lq;
            const bool ok = (ks + s) < ks1;
            const size_t off = (size_t)(ok ? k : 0) * ld;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                a[s][x] = ok ? pa[off + x * 16] : 0.0;
                b[s][x] = ok ? pb[off + x * 16] : 0.0;
            }
        }
    };
    auto compute = [&](const double (&a)[KS][4], const double (&b)[KS][4]) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][x], b[s][y], acc[x][y], 0, 0, 0);
    };
    if (ks0 < ks1) {
        load(a0, b0, ks0);
        for (int ks = ks0; ks < ks1; ks += 2 * KS) {
            load(a1, b1, ks + KS);
            compute(a0, b0);
            load(a0, b0, ks + 2 * KS);
            compute(a1, b1);
        }
    }
    if (stamp) {       // diagnostic: shader cycles and 100 MHz real-time ticks over the k-loop of one workgroup
        clk[0] = __builtin_amdgcn_s_memtime() - t_cyc0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - t_real0;
    }
    double* C = slabs + (long long)blockIdx.y * slabStride;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int col = tj * 128 + wn * 64 + y * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 128 + wm * 64 + x * 16 + lq + 4 * r;
                C[(size_t)row * M + col] = acc[x][y][r];
            }
        }
}

// ---- variant 3: the production tiling (128x128 tile, BK = 16, 2x2 waves, [k][m] LDS images with row
// stride 144 doubles, double-buffered) with LDS-DMA staging: one global_load_lds_dwordx4 wave-instruction
// moves one k-row of one operand (128 doubles = 1 KB = 64 lanes x 16 B) straight into its padded LDS row.
// No staging registers, no ds_write phase, no per-thread address arithmetic in the loop.
constexpr int GBK = 16, GLD = 128 + 16;
__global__ __launch_bounds__(256, 2) void syrk_glds_kernel(const double* __restrict__ K, int rows, int M, int ld,
                                                           double* __restrict__ slabs, long long slabStride, int splits) {
    __shared__ __attribute__((aligned(16))) double As[2][GBK * GLD];
    __shared__ __attribute__((aligned(16))) double Bs[2][GBK * GLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lq = lane >> 4;
    int ti, tj;
    tri_tile(xcd_remap(blockIdx.x, gridDim.x), ti, tj);
    const int kt_total = rows / GBK;
    const int kt_per = (kt_total + splits - 1) / splits;
    const int kt0 = blockIdx.y * kt_per, kt1 = min(kt_total, kt0 + kt_per);

    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

    // this wave's four k-rows of each operand: rows wave, wave+4, wave+8, wave+12 of the k-tile
    const char* ga = reinterpret_cast<const char*>(K + (size_t)kt0 * GBK * ld + ti * 128) + lane * 16;
    const char* gb = reinterpret_cast<const char*>(K + (size_t)kt0 * GBK * ld + tj * 128) + lane * 16;
    const size_t rowB = (size_t)ld * 8, tileB = (size_t)GBK * ld * 8;
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kr = wave + 4 * q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + kr * rowB),
                                             (__attribute__((address_space(3))) void*)(&As[buf][kr * GLD]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + kr * rowB),
                                             (__attribute__((address_space(3))) void*)(&Bs[buf][kr * GLD]), 16, 0, 0);
        }
    };
    if (kt0 < kt1) {
        stage(0);
        __syncthreads();                              // vmcnt(0) + barrier: tile kt0 is in LDS
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < kt1) {
                ga += tileB;
                gb += tileB;
                stage(buf ^ 1);
            }
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int kk = 0; kk < GBK / 4; ++kk) {
                double a[4], b[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) a[x] = As[buf][(kk * 4 + lq) * GLD + wm * 64 + x * 16 + lr];
#pragma unroll
                for (int x = 0; x < 4; ++x) b[x] = Bs[buf][(kk * 4 + lq) * GLD + wn * 64 + x * 16 + lr];
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(2);
            __syncthreads();                          // next tile landed (vmcnt(0)), everyone done with `buf`
        }
    }
    double* C = slabs + (long long)blockIdx.y * slabStride;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int col = tj * 128 + wn * 64 + y * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 128 + wm * 64 + x * 16 + lq + 4 * r;
                C[(size_t)row * M + col] = acc[x][y][r];
            }
        }
}

int syrk_variant(Ctx* c, const double* K, int rows, int M, int ld, int variant, int splits, double* Phi) {
    const int tm = M / 128;
    const long long nt = (long long)tm * (tm + 1) / 2;
    if (splits <= 0) splits = (variant == 3) ? pick_syrk_splits(c, nt, rows / 16, rows, M, 128, 8) : (int)std::max<long long>(1, (8LL * c->num_cu) / nt);
    const long long slabStride = (long long)M * M;
    TRY(c->slabs.reserve((size_t)splits * slabStride * sizeof(double)));
    dim3 grid((unsigned)nt, (unsigned)splits);
    TRY(c->red.reserve(64));
    unsigned long long* clk = static_cast<unsigned long long*>(c->red.p);
    if (variant == 1) hipLaunchKernelGGL((syrk_direct_kernel<2>), grid, dim3(256), 0, c->stream, K, rows, M, ld, c->slabs.d(), slabStride, splits, clk);
    else if (variant == 2) hipLaunchKernelGGL((syrk_direct_kernel<1>), grid, dim3(256), 0, c->stream, K, rows, M, ld, c->slabs.d(), slabStride, splits, clk);
    else if (variant == 3) hipLaunchKernelGGL(syrk_glds_kernel, grid, dim3(256), 0, c->stream, K, rows, M, ld, c->slabs.d(), slabStride, splits);
    else { set_error("unknown syrk variant %d", variant); return EDRGP_ERR_INVALID; }
    HIP_TRY(hipGetLastError());
    if (Phi) TRY(slab_reduce_mirror(c, c->slabs.d(), splits, slabStride, M, Phi));
    if (getenv("EDRGP_CLK")) {
        unsigned long long h[2] = {0, 0};
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        const int ksteps = (rows / 4 + splits - 1) / splits;
        fprintf(stderr, "[clk] variant %d: %llu shader cycles, %llu x10ns -> %.3f GHz; %d k-steps -> %.1f cycles per k-step (16 MFMA per wave)\n",
                variant, h[0], h[1], h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0, ksteps, (double)h[0] / ksteps);
    }
    return EDRGP_OK;
}

int fill_pattern(Ctx* c, double* K, size_t n, unsigned long long seed) {
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(2048), dim3(256), 0, c->stream, K, n, seed);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

}  // namespace edrgp
