// hpn_comm.hip -- the one collective of the path: a sum all-reduce of the small
// per-GPU count vectors over xGMI.  The reference has no
// counterpart (it is single-process); the nearest seam is reduceStats
// (fastq_count_kthread.c:180-210), an element-wise sum of per-file accumulators.
//
// RCCL is bound at run time with dlopen so that the library loads on machines
// (and in processes) that never reduce across GPUs; inside a PyTorch process this
// resolves to the librccl.so.1 torch already loaded.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "hpn_ctx.hpp"

using namespace hpn;

namespace {

// The slice of rccl.h this file needs (ABI-stable NCCL 2.x definitions).
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclUint64 = 5 };
enum { ncclSum = 0 };

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

void rccl_load(Rccl &r);

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;  // contexts of several host threads may initialise concurrently
    std::call_once(once, [] { rccl_load(r); });
    return r;
}

void rccl_load(Rccl &r)
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        r.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (r.h) break;
    }
    if (!r.h) return;
    r.GetUniqueId = (int (*)(ncclUniqueId *))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(ncclComm_t *, int, ncclUniqueId, int))dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (int (*)(ncclComm_t))dlsym(r.h, "ncclCommDestroy");
    r.AllReduce = (int (*)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t))dlsym(r.h, "ncclAllReduce");
    r.GetErrorString = (const char *(*)(int))dlsym(r.h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
}

}  // namespace

static_assert(sizeof(ncclUniqueId) == HPN_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" {

int hpn_comm_unique_id(uint8_t id[HPN_UNIQUE_ID_BYTES])
{
    if (!id) return HPN_E_ARG;
    Rccl &r = rccl();
    if (!r.ok) return HPN_E_RCCL;
    ncclUniqueId u;
    if (r.GetUniqueId(&u) != ncclSuccess) return HPN_E_RCCL;
    memcpy(id, u.internal, HPN_UNIQUE_ID_BYTES);
    return HPN_OK;
}

int hpn_comm_init(hpn_ctx *c, int rank, int n_ranks, const uint8_t id[HPN_UNIQUE_ID_BYTES])
{
    if (!c || !id || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return HPN_E_ARG;
    if (c->comm) return fail(c, HPN_E_STATE, "communicator already initialised");
    Rccl &r = rccl();
    if (!r.ok) return fail(c, HPN_E_RCCL, "librccl.so.1 not loadable: %s", dlerror());
    HPN_HIP(c, hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(u.internal, id, HPN_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    int e = r.CommInitRank(&comm, n_ranks, u, rank);
    if (e != ncclSuccess) return fail(c, HPN_E_RCCL, "ncclCommInitRank: %s", r.GetErrorString ? r.GetErrorString(e) : "?");
    c->comm = comm;
    return HPN_OK;
}

int hpn_comm_destroy(hpn_ctx *c)
{
    if (!c) return HPN_E_ARG;
    if (c->comm) {
        rccl().CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    return HPN_OK;
}

int hpn_allreduce_u64(hpn_ctx *c, uint64_t *d_vec, size_t n)
{
    if (!c || !d_vec) return HPN_E_ARG;
    if (!c->comm) return fail(c, HPN_E_STATE, "hpn_comm_init has not been called");
    HPN_HIP(c, hipSetDevice(c->device));
    // One call, no chunking: 515 .. 68k words is latency-bound on any xGMI ring.
    int e = rccl().AllReduce(d_vec, d_vec, n, ncclUint64, ncclSum, (ncclComm_t)c->comm, c->stream);
    if (e != ncclSuccess) return fail(c, HPN_E_RCCL, "ncclAllReduce: %s", rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
    return HPN_OK;
}

}  // extern "C"
