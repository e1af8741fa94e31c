// rccl_stub.cpp -- a TEST-ONLY stand-in for librccl, selected with HPN_RCCL_LIB=<this .so> (csrc/hpn_comm.hip binds RCCL by
// dlopen).  A one-GPU box cannot run RCCL with more than one rank -- the real library refuses two ranks on one device -- so the
// grouped single-process collective of the C tools (hpn_comm_init_all + hpn_allreduce_u64_all: GroupStart, n x AllReduce,
// GroupEnd) and everything above it had never executed with n > 1.  This file implements just the calls hpn_comm.hip makes,
// with the semantics rccl.h documents for them:
//   ncclCommInitAll(comms, n, devs)   n communicators of one clique (devices may repeat here: that is the point)
//   ncclCommInitRank                  cliques of ONE rank only (several processes are not emulated)
//   ncclGroupStart / ncclGroupEnd     calls between them are queued; the outermost GroupEnd runs them
//   ncclAllReduce(ncclUint64, ncclSum) at GroupEnd: every rank of a clique must have queued one call of the same count; the
//                                     streams are drained, the vectors summed, the sum written to every recv buffer
//   ncclCommCount / ncclCommDestroy / ncclGetUniqueId / ncclGetErrorString
// The sum is made through the host (hipMemcpy): correctness of the call pattern is what is tested, not bandwidth.
// RCCL_STUB_FAIL_AT=k: the k-th ncclAllReduce of the process (0-based) returns ncclInternalError -- how the tests reach
// hpn_allreduce_u64_all's "failed half-way" branch.  Nothing of this is linked into or shipped with libhpngs.so.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
enum { ncclUint64 = 5, ncclSum = 0 };

struct Clique {
    int n = 0;
};
struct Comm {
    std::shared_ptr<Clique> clique;
    int rank = 0, dev = 0;
};
struct Call {
    const void *send;
    void *recv;
    size_t count;
    Comm *comm;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Call> g_queue;
std::atomic<long> g_calls{0};

int run(std::vector<Call> &q)
{
    std::map<Clique *, std::vector<Call *>> by;
    for (Call &c : q) by[c.comm->clique.get()].push_back(&c);
    int rc = ncclSuccess, dev0 = 0;
    (void)hipGetDevice(&dev0);
    for (auto &kv : by) {
        std::vector<Call *> &v = kv.second;
        if ((int)v.size() != kv.first->n) {   // a rank of the clique did not take part: the real library would hang here
            fprintf(stderr, "[rccl-stub] %zu of %d ranks entered the all-reduce\n", v.size(), kv.first->n);
            rc = ncclInvalidUsage;
            continue;
        }
        const size_t count = v[0]->count;
        std::vector<uint64_t> sum(count, 0), tmp(count);
        bool ok = true;
        for (Call *c : v) {
            if (c->count != count) ok = false;
            if (!ok) break;
            ok = hipSetDevice(c->comm->dev) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
                 hipMemcpy(tmp.data(), c->send, count * 8, hipMemcpyDeviceToHost) == hipSuccess;
            for (size_t i = 0; ok && i < count; ++i) sum[i] += tmp[i];
        }
        for (Call *c : v)
            if (ok) ok = hipSetDevice(c->comm->dev) == hipSuccess && hipMemcpy(c->recv, sum.data(), count * 8, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) rc = ncclUnhandledCudaError;
        if (getenv("RCCL_STUB_LOG")) fprintf(stderr, "[rccl-stub] all-reduce: %d ranks x %zu words\n", kv.first->n, count);
    }
    (void)hipSetDevice(dev0);
    q.clear();
    return rc;
}
}  // namespace

extern "C" {
typedef Comm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    for (int i = 0; i < 128; ++i) id->internal[i] = (char)(i * 7 + 1);
    return ncclSuccess;
}
int ncclCommInitRank(ncclComm_t *comm, int n, ncclUniqueId, int rank)
{
    if (!comm || rank < 0 || rank >= n) return ncclInvalidArgument;
    if (n != 1) return ncclInvalidUsage;   // ranks in other processes are not emulated
    Comm *c = new Comm;
    c->clique = std::make_shared<Clique>();
    c->clique->n = 1;
    (void)hipGetDevice(&c->dev);
    *comm = c;
    return ncclSuccess;
}
int ncclCommInitAll(ncclComm_t *comms, int n, const int *devs)
{
    if (!comms || n < 1) return ncclInvalidArgument;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) return ncclUnhandledCudaError;
    auto cl = std::make_shared<Clique>();
    cl->n = n;
    for (int i = 0; i < n; ++i) {
        const int d = devs ? devs[i] : i;
        if (d < 0 || d >= have) return ncclInvalidArgument;
        comms[i] = new Comm;
        comms[i]->clique = cl, comms[i]->rank = i, comms[i]->dev = d;
    }
    return ncclSuccess;
}
int ncclCommDestroy(ncclComm_t c)
{
    delete c;
    return ncclSuccess;
}
int ncclCommCount(const ncclComm_t c, int *n)
{
    if (!c || !n) return ncclInvalidArgument;
    *n = c->clique->n;
    return ncclSuccess;
}
int ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}
int ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth) return ncclSuccess;
    return run(g_queue);
}
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t comm, hipStream_t stream)
{
    const long k = g_calls++;
    if (const char *e = getenv("RCCL_STUB_FAIL_AT"))
        if (atol(e) == k) return ncclInternalError;
    if (!send || !recv || !comm) return ncclInvalidArgument;
    if (dtype != ncclUint64 || op != ncclSum) return ncclInvalidArgument;
    g_queue.push_back(Call{send, recv, count, comm, stream});
    if (g_depth) return ncclSuccess;
    return run(g_queue);
}
const char *ncclGetErrorString(int e)
{
    switch (e) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "stub: a HIP call failed";
        case ncclInternalError: return "stub: injected failure";
        case ncclInvalidArgument: return "stub: invalid argument";
        case ncclInvalidUsage: return "stub: invalid usage";
    }
    return "stub: unknown";
}
}
