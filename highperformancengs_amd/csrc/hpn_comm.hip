// hpn_comm.hip -- the one collective of the path: a sum all-reduce of the small
// per-GPU count vectors over xGMI.  The reference has no
// counterpart (it is single-process); the nearest seam is reduceStats
// (fastq_count_kthread.c:180-210), an element-wise sum of per-file accumulators.
//
// RCCL is bound at run time with dlopen so that the library loads on machines
// (and in processes) that never reduce across GPUs; inside a PyTorch process this
// resolves to the librccl.so.1 torch already loaded.
#include <dlfcn.h>
#include <stdlib.h>
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
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    char path[512] = {0};   // where the library really came from (dladdr of one of its symbols)
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
    // A process that already holds an RCCL (PyTorch loads its own torch/lib/librccl.so) must not get a second
    // one beside it: RTLD_NOLOAD first asks for the copy that is mapped under either soname, and only a process
    // without any loads the system library.  HPN_RCCL_LIB names a file outright.
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (const char *e = test_env("HPN_RCCL_LIB")) r.h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    for (int i = 0; !r.h && i < 2; ++i) r.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!r.h && dlsym(RTLD_DEFAULT, "ncclAllReduce")) r.h = dlopen(nullptr, RTLD_NOW);   // linked into the process under another name
    for (int i = 0; !r.h && i < 3; ++i) r.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!r.h) return;
    r.GetUniqueId = (int (*)(ncclUniqueId *))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(ncclComm_t *, int, ncclUniqueId, int))dlsym(r.h, "ncclCommInitRank");
    r.CommInitAll = (int (*)(ncclComm_t *, int, const int *))dlsym(r.h, "ncclCommInitAll");
    r.GroupStart = (int (*)())dlsym(r.h, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(r.h, "ncclGroupEnd");
    r.CommDestroy = (int (*)(ncclComm_t))dlsym(r.h, "ncclCommDestroy");
    r.AllReduce = (int (*)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t))dlsym(r.h, "ncclAllReduce");
    r.CommCount = (int (*)(const ncclComm_t, int *))dlsym(r.h, "ncclCommCount");
    r.GetErrorString = (const char *(*)(int))dlsym(r.h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
    Dl_info di;
    if (r.AllReduce && dladdr((void *)r.AllReduce, &di) && di.dli_fname) snprintf(r.path, sizeof r.path, "%s", di.dli_fname);
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

const char *hpn_comm_library(void) { return rccl().path; }

// HPN_COMM_SHARED_DEVICE=1 (tests only): contexts that share a device are passed on to ncclCommInitAll.  RCCL itself refuses
// them ("duplicate GPU"), so the switch changes nothing in production; with the test stand-in named by HPN_RCCL_LIB
// (tests/stub/rccl_stub.cpp) it lets the grouped collective run with n = 2..8 "ranks" on a one-GPU box.
static bool shared_device_allowed()
{
    const char *e = test_env("HPN_COMM_SHARED_DEVICE");
    return e && e[0] == '1';
}

int hpn_comm_init_all(hpn_ctx **ctxs, int n)
{
    if (!ctxs || n < 1 || n > 64) return HPN_E_ARG;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return HPN_E_ARG;
        if (ctxs[i]->comm) return fail(ctxs[i], HPN_E_STATE, "communicator already initialised");
        for (int j = 0; j < i && !shared_device_allowed(); ++j)
            if (ctxs[j]->device == ctxs[i]->device)
                return fail(ctxs[0], HPN_E_ARG, "contexts %d and %d share device %d: one RCCL rank per device", j, i, ctxs[i]->device);
    }
    Rccl &r = rccl();
    if (!r.ok || !r.CommInitAll || !r.GroupStart || !r.GroupEnd) return fail(ctxs[0], HPN_E_RCCL, "RCCL not loadable: %s", dlerror());
    ncclComm_t comms[64];
    int devs[64];
    for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device, comms[i] = nullptr;
    const int e = r.CommInitAll(comms, n, devs);
    if (e != ncclSuccess) return fail(ctxs[0], HPN_E_RCCL, "ncclCommInitAll: %s", r.GetErrorString ? r.GetErrorString(e) : "?");
    for (int i = 0; i < n; ++i) ctxs[i]->comm = comms[i];
    return HPN_OK;
}

int hpn_comm_count(hpn_ctx *c, int *n_ranks)
{
    if (!c || !n_ranks) return HPN_E_ARG;
    if (!c->comm) return fail(c, HPN_E_STATE, "no communicator on this context");
    Rccl &r = rccl();
    if (!r.CommCount) return fail(c, HPN_E_RCCL, "ncclCommCount not exported by %s", r.path);
    const int e = r.CommCount((ncclComm_t)c->comm, n_ranks);
    if (e != ncclSuccess) return fail(c, HPN_E_RCCL, "ncclCommCount: %s", r.GetErrorString ? r.GetErrorString(e) : "?");
    return HPN_OK;
}

int hpn_allreduce_u64_all(hpn_ctx **ctxs, uint64_t **d_vecs, int n, size_t n_words)
{
    if (!ctxs || !d_vecs || n < 1 || n > 64) return HPN_E_ARG;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !d_vecs[i]) return HPN_E_ARG;
        if (!ctxs[i]->comm) return fail(ctxs[i], HPN_E_STATE, "hpn_comm_init_all has not been called");
    }
    Rccl &r = rccl();
    // one group: a single thread drives all ranks, so the n calls must be fused or the first would wait for peers forever.
    // Nothing returns between GroupStart and GroupEnd: a group left open on this thread would swallow every later collective.
    int e = r.GroupStart();
    if (e != ncclSuccess) return fail(ctxs[0], HPN_E_RCCL, "ncclGroupStart: %s", r.GetErrorString ? r.GetErrorString(e) : "?");
    hipError_t he = hipSuccess;
    int queued = 0;
    for (int i = 0; e == ncclSuccess && he == hipSuccess && i < n; ++i) {
        he = hipSetDevice(ctxs[i]->device);
        if (he != hipSuccess) break;
        e = r.AllReduce(d_vecs[i], d_vecs[i], n_words, ncclUint64, ncclSum, (ncclComm_t)ctxs[i]->comm, ctxs[i]->stream);
        if (e == ncclSuccess) ++queued;
    }
    const int e2 = r.GroupEnd();
    // with some of the all-reduces enqueued the vectors may or may not hold sums: HPN_E_PARTIAL tells the caller not to add them again
    if (he != hipSuccess)
        return fail(ctxs[0], queued ? HPN_E_PARTIAL : HPN_E_HIP, "hipSetDevice inside the group failed: %s (%d of %d all-reduces enqueued)",
                    hipGetErrorString(he), queued, n);
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess)
        return fail(ctxs[0], queued ? HPN_E_PARTIAL : HPN_E_RCCL, "grouped ncclAllReduce: %s (%d of %d enqueued)",
                    r.GetErrorString ? r.GetErrorString(e) : "?", queued, n);
    // (all n in-place all-reduces are enqueued or done by now: a failure from here on leaves sums in the vectors, too)
    for (int i = 0; i < n; ++i) {
        hipError_t hs = hipSetDevice(ctxs[i]->device);
        if (hs == hipSuccess) hs = hipStreamSynchronize(ctxs[i]->stream);
        if (hs != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctxs[i], HPN_E_PARTIAL, "waiting for rank %d's all-reduce failed: %s (all %d were enqueued)", i, hipGetErrorString(hs), n);
        }
    }
    return HPN_OK;
}

}  // extern "C"
