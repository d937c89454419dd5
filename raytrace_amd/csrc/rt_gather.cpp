// rt_gather.cpp -- the frame gather over RCCL behind the C ABI of include/rt_gather.h (librt_rccl.so).
// One process, one communicator per GPU (ncclCommInitAll), one host thread per GPU: each rank sends its packed
// share to rank 0 and rank 0 posts one receive per rank, all inside one group, so the transfers of the (up to seven)
// peers run concurrently, each over its own xGMI link to the root.  This is the exchange that replaces
// MPI_Isend / MPI_Recv of raytrace.cpp:211, 219-227.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "rt_gather.h"
#include "rt_hip.h"

namespace {
thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
} // namespace

struct rt_gather {
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
};

#define NCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) return fail(RT_E_DEVICE, "%s: %s", #expr, ncclGetErrorString(r_));   \
    } while (0)

extern "C" {

const char *rt_gather_last_error(void) { return g_err.c_str(); }

int rt_gather_create(int32_t n_ranks, const int32_t *devs, rt_gather **out) {
    if (!out) return fail(RT_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n_ranks < 1 || !devs) return fail(RT_E_INVALID, "no devices");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return fail(RT_E_DEVICE, "hipGetDeviceCount failed");
    for (int i = 0; i < n_ranks; i++) {
        if (devs[i] < 0 || devs[i] >= ndev) return fail(RT_E_INVALID, "rank %d: device %d not in [0,%d)", i, devs[i], ndev);
        for (int j = 0; j < i; j++)
            if (devs[j] == devs[i]) return fail(RT_E_INVALID, "device %d is listed twice", devs[i]);
    }
    rt_gather *g = new rt_gather();
    g->devs.assign(devs, devs + n_ranks);
    g->comms.resize((size_t)n_ranks);
    const ncclResult_t r = ncclCommInitAll(g->comms.data(), n_ranks, g->devs.data());
    if (r != ncclSuccess) {
        delete g;
        return fail(RT_E_DEVICE, "ncclCommInitAll over %d GPU(s): %s", n_ranks, ncclGetErrorString(r));
    }
    *out = g;
    return RT_OK;
}

int rt_gather_shares(rt_gather *g, int32_t rank, const double *d_send, double *d_recv, const uint64_t *counts, void *stream_) {
    if (!g || !counts) return fail(RT_E_INVALID, "NULL argument");
    const int n = (int)g->comms.size();
    if (rank < 0 || rank >= n) return fail(RT_E_INVALID, "rank %d not in [0,%d)", rank, n);
    if (counts[rank] > 0 && !d_send) return fail(RT_E_INVALID, "d_send is NULL");
    if (rank == 0 && !d_recv) return fail(RT_E_INVALID, "rank 0 needs d_recv");
    if (hipSetDevice(g->devs[(size_t)rank]) != hipSuccess) return fail(RT_E_DEVICE, "hipSetDevice(%d) failed", g->devs[(size_t)rank]);
    hipStream_t stream = (hipStream_t)stream_;
    ncclComm_t comm = g->comms[(size_t)rank];
    NCCL_TRY(ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    if (rank == 0) { // one receive per rank, its own share included
        uint64_t off = 0;
        for (int k = 0; k < n && r == ncclSuccess; k++) {
            if (counts[k]) r = ncclRecv(d_recv + off, (size_t)counts[k], ncclDouble, k, comm, stream);
            off += counts[k];
        }
    }
    if (r == ncclSuccess && counts[rank]) r = ncclSend(d_send, (size_t)counts[rank], ncclDouble, 0, comm, stream);
    const ncclResult_t e = ncclGroupEnd(); // (always closed, also after a failed call inside the group)
    if (r != ncclSuccess) return fail(RT_E_DEVICE, "ncclSend/ncclRecv: %s", ncclGetErrorString(r));
    if (e != ncclSuccess) return fail(RT_E_DEVICE, "ncclGroupEnd: %s", ncclGetErrorString(e));
    return RT_OK;
}

int rt_gather_destroy(rt_gather *g) {
    if (!g) return RT_OK;
    for (size_t i = 0; i < g->comms.size(); i++) {
        (void)hipSetDevice(g->devs[i]);
        (void)ncclCommDestroy(g->comms[i]);
    }
    delete g;
    return RT_OK;
}

} // extern "C"
