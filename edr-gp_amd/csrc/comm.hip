// Row-sharded multi-GPU exchange: one sum-all-reduce of [Phi | K_mn y] (and of
// the gradient partials) per evaluation over RCCL / xGMI.  One process per GPU;
// the ncclUniqueId travels through the host launcher (torch.distributed store,
// MPI, a file ...), never through this library.
#include "eval_common.h"
#include <rccl/rccl.h>
#include <cstring>

namespace edrgp {

#define NCCL_TRY(expr)                                                               \
    do {                                                                             \
        ncclResult_t _r = (expr);                                                    \
        if (_r != ncclSuccess) {                                                     \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
            return EDRGP_ERR_COMM;                                                   \
        }                                                                            \
    } while (0)

int allreduce_sum(Ctx* c, double* buf, size_t count) {
    if (c->host_ar != nullptr) {
        // launcher-provided collective (edrgp_comm_init_host): staged through pinned host memory.  The next
        // call's device -> host copy is stream-ordered behind this call's host -> device copy.
        if (c->h_ar_cap < count) {
            if (c->h_ar) (void)hipHostFree(c->h_ar);
            c->h_ar = nullptr;
            c->h_ar_cap = 0;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_ar), count * sizeof(double), hipHostMallocDefault));
            c->h_ar_cap = count;
        }
        HIP_TRY(hipMemcpyAsync(c->h_ar, buf, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const int st = c->host_ar(c->host_ar_user, c->h_ar, (int64_t)count);
        if (st != 0) { set_error("host all-reduce callback failed with status %d", st); return EDRGP_ERR_COMM; }
        HIP_TRY(hipMemcpyAsync(buf, c->h_ar, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
        return EDRGP_OK;
    }
    if (!c->comm) return EDRGP_OK;
    NCCL_TRY(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream));
    return EDRGP_OK;
}

// Phi is symmetric: only its lower triangle (row i: i + 1 values at offset i (i + 1) / 2) and b travel,
// M (M + 1) / 2 + M values instead of M^2 + M (16.8 MB instead of 33.6 MB at M = 2048).
__global__ void pack_lower_kernel(const double* __restrict__ Phi, int M, double* __restrict__ buf) {
    const int i = blockIdx.y;
    const size_t off = (size_t)i * (i + 1) / 2;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j <= i; j += gridDim.x * blockDim.x) buf[off + j] = Phi[(size_t)i * M + j];
    if (i == 0) {                        // b follows the triangle
        const size_t tri = (size_t)M * (M + 1) / 2;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) buf[tri + j] = Phi[(size_t)M * M + j];
    }
}

__global__ void unpack_lower_kernel(const double* __restrict__ buf, int M, double* __restrict__ Phi) {
    const int i = blockIdx.y;
    const size_t off = (size_t)i * (i + 1) / 2;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j <= i; j += gridDim.x * blockDim.x) {
        const double v = buf[off + j];
        Phi[(size_t)i * M + j] = v;
        Phi[(size_t)j * M + i] = v;      // mirrored: Phi stays bitwise symmetric
    }
    if (i == 0) {
        const size_t tri = (size_t)M * (M + 1) / 2;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) Phi[(size_t)M * M + j] = buf[tri + j];
    }
}

int allreduce_phi(Ctx* c, double* Phi, int M) {
    if (!c->sharded()) return EDRGP_OK;
    const size_t count = (size_t)M * (M + 1) / 2 + M;
    TRY(c->pack.reserve(count * sizeof(double)));
    dim3 grd((unsigned)std::max(1, std::min(8, (M + 255) / 256)), (unsigned)M);
    hipLaunchKernelGGL(pack_lower_kernel, grd, dim3(256), 0, c->stream, Phi, M, c->pack.d());
    HIP_TRY(hipGetLastError());
    TRY(allreduce_sum(c, c->pack.d(), count));
    hipLaunchKernelGGL(unpack_lower_kernel, grd, dim3(256), 0, c->stream, c->pack.d(), M, Phi);
    HIP_TRY(hipGetLastError());
    return EDRGP_OK;
}

}  // namespace edrgp

using namespace edrgp;

extern "C" {

int edrgp_comm_unique_id(void* id128) {
    if (!id128) return EDRGP_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return EDRGP_OK;
}

int edrgp_comm_init(edrgp_ctx* ctx, int nranks, int rank, const void* id128) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) { set_error("comm_init: bad arguments"); return EDRGP_ERR_INVALID; }
    HIP_TRY(hipSetDevice(c->device));
    if (c->sharded()) { set_error("comm_init: communicator already attached"); return EDRGP_ERR_STATE; }
    ++c->state_serial;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm;
    NCCL_TRY(ncclCommInitRank(&comm, nranks, id, rank));
    c->comm = comm;
    c->glob_valid = false;
    c->nranks = nranks;
    c->rank = rank;
    return EDRGP_OK;
}

int edrgp_comm_init_host(edrgp_ctx* ctx, int nranks, int rank, edrgp_allreduce_fn fn, void* user) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !fn || nranks < 1 || rank < 0 || rank >= nranks) { set_error("comm_init_host: bad arguments"); return EDRGP_ERR_INVALID; }
    if (c->sharded()) { set_error("comm_init_host: communicator already attached"); return EDRGP_ERR_STATE; }
    ++c->state_serial;
    c->host_ar = fn;
    c->host_ar_user = user;
    c->glob_valid = false;
    c->nranks = nranks;
    c->rank = rank;
    return EDRGP_OK;
}

int edrgp_comm_destroy(edrgp_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !c->sharded()) return EDRGP_OK;
    ++c->state_serial;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->host_ar != nullptr) {
        c->host_ar = nullptr;
        c->host_ar_user = nullptr;
        c->glob_valid = false;
        c->nranks = 1;
        c->rank = 0;
        return EDRGP_OK;
    }
    ncclResult_t r = ncclCommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    c->glob_valid = false;
    c->nranks = 1;
    c->rank = 0;
    if (r != ncclSuccess) { set_error("ncclCommDestroy: %s", ncclGetErrorString(r)); return EDRGP_ERR_COMM; }
    return EDRGP_OK;
}

}  // extern "C"
