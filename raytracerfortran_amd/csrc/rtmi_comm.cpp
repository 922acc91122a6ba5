// rtmi_comm.cpp -- the one exchange step of the path: RCCL over xGMI.
//
// The reference's tempering loop moves chain states between MPI ranks through rank 0 (prjmh_temper_rf.f90:331-349,
// 1329-1357).  Here chains are sharded over the GPUs of a node, one process per GPU, and a swap round needs exactly
// one collective: an all-gather of the chains' logL (8 B per chain; 256 chains = 2 KB), after which every rank takes
// the same decisions and swaps TEMPERATURES in place (rtmi_pt_rules.h).  This file wraps the RCCL communicator
// behind the C ABI so that a C or Fortran host can run the exchange without torch: the 128-byte unique id is created
// on rank 0, broadcast by whatever the host program already has (MPI_Bcast, a file, torch.distributed) and handed
// to rtmi_comm_create on every rank.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>

#include "rtmi_host.h"

using namespace rtmi_host;

struct rtmi_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
};

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess)                                                                           \
            return fail(RTMI_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(r_));                  \
    } while (0)

extern "C" {

int rtmi_comm_unique_id(char id[RTMI_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) <= RTMI_COMM_ID_BYTES, "RTMI_COMM_ID_BYTES too small for ncclUniqueId");
    if (!id)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_comm_unique_id: NULL");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    memset(id, 0, RTMI_COMM_ID_BYTES);
    memcpy(id, &u, sizeof u);
    return RTMI_OK;
}

rtmi_comm *rtmi_comm_create(int nranks, int rank, const char id[RTMI_COMM_ID_BYTES])
{
    if (nranks < 1 || rank < 0 || rank >= nranks || !id) {
        fail(RTMI_ERR_BAD_ARG, "rtmi_comm_create: bad arguments");
        return nullptr;
    }
    Context &c = ctx();
    int device = 0;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (ensure_ready_locked(c) != RTMI_OK)
            return nullptr;
        device = c.device;
    }
    if (hipSetDevice(device) != hipSuccess)
        return nullptr;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    rtmi_comm *k = new rtmi_comm;
    k->nranks = nranks;
    k->rank = rank;
    k->device = device;
    const ncclResult_t r = ncclCommInitRank(&k->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        fail(RTMI_ERR_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
        delete k;
        return nullptr;
    }
    return k;
}

void rtmi_comm_destroy(rtmi_comm *comm)
{
    if (!comm)
        return;
    if (comm->comm)
        (void)ncclCommDestroy(comm->comm);
    delete comm;
}

int rtmi_comm_size(const rtmi_comm *comm) { return comm ? comm->nranks : 1; }

// Which RCCL this library's communicator is: a process that also holds torch carries torch's bundled librccl next to the
// system one this library is linked against, and which of them ncclCommInitRank / ncclAllGather resolve to INSIDE librtmi.so is
// decided by load order (raymod.py imports torch first, so that one copy serves both).  Version as ncclGetVersion reports it;
// path of the shared object the symbol lives in (dladdr).
int rtmi_comm_rccl_version(void)
{
    int v = 0;
    return ncclGetVersion(&v) == ncclSuccess ? v : -1;
}

int rtmi_comm_rccl_library(char *path, int cap)
{
    if (!path || cap < 2)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_comm_rccl_library: bad argument");
    Dl_info info;
    memset(&info, 0, sizeof info);
    if (!dladdr(reinterpret_cast<const void *>(&ncclAllGather), &info) || !info.dli_fname)
        return fail(RTMI_ERR_HIP, "rtmi_comm_rccl_library: dladdr found no object for ncclAllGather");
    snprintf(path, (size_t)cap, "%s", info.dli_fname);
    return RTMI_OK;
}

int rtmi_comm_rank(const rtmi_comm *comm) { return comm ? comm->rank : 0; }

int rtmi_comm_allgather_f64(rtmi_comm *comm, const double *d_send, double *d_recv, int64_t count, void *stream)
{
    if (!comm || !d_send || !d_recv || count < 0)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_comm_allgather_f64: bad argument");
    NCCL_TRY(ncclAllGather(d_send, d_recv, (size_t)count, ncclDouble, comm->comm, (hipStream_t)stream));
    return RTMI_OK;
}

}  // extern "C"
