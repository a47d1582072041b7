// Row-sharded multi-GPU exchange: one sum-all-reduce of [Phi | K_mn y] (and of
// the gradient partials) per evaluation over RCCL / xGMI.  One process per GPU;
// the ncclUniqueId travels through the host launcher (torch.distributed store,
// MPI, a file ...), never through this library.
#include "common.h"
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
    if (!c->comm) return EDRGP_OK;
    NCCL_TRY(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream));
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
    if (c->comm) { set_error("comm_init: communicator already attached"); return EDRGP_ERR_STATE; }
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

int edrgp_comm_destroy(edrgp_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !c->comm) return EDRGP_OK;
    ++c->state_serial;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    ncclResult_t r = ncclCommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    c->glob_valid = false;
    c->nranks = 1;
    c->rank = 0;
    if (r != ncclSuccess) { set_error("ncclCommDestroy: %s", ncclGetErrorString(r)); return EDRGP_ERR_COMM; }
    return EDRGP_OK;
}

}  // extern "C"
