/*
 * rt_gather.h -- C ABI of the frame gather over RCCL (librt_rccl.so).
 *
 * It replaces the reference's exchange step: every rank MPI_Isends its row band to rank 0, which MPI_Recvs them
 * one after the other and averages replicas (raytrace.cpp:211, 219-227).  Here one process drives all GPUs of a
 * node (one host thread per GPU, as raytrace_hip does): every GPU holds its share of the frame as one packed device
 * buffer (rt_render_tiles_device) and sends it to the root GPU in ONE grouped exchange; tiles are disjoint, so
 * nothing is averaged.  The library depends on librccl and on nothing of librt_hip.so.
 * (The one-process-per-GPU driver does the same exchange with torch.distributed.gather.)
 *
 * Functions return 0 or a negative RT_E_* code of rt_hip.h; rt_gather_last_error() gives the calling thread's message.
 */
#ifndef RT_GATHER_H
#define RT_GATHER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_gather rt_gather; /* one communicator per listed GPU, all in this process */

/* replaces MPI_Init / MPI_Comm_rank / MPI_Comm_size (raytrace.cpp:99-104): rank r of the group is GPU devs[r] */
int rt_gather_create(int32_t n_ranks, const int32_t *devs, rt_gather **out);

/*
 * replaces MPI_Isend + the MPI_Recv loop (raytrace.cpp:211, 219-227).  Called once per rank, each from the host
 * thread that drives that rank's GPU (the calls of one exchange must all be made: it completes collectively).
 *   d_send   this rank's packed share on its GPU, counts[rank] doubles
 *   d_recv   root only (rank 0): room for all shares back to back in rank order, sum(counts) doubles, on GPU devs[0]
 *   counts   [n_ranks] doubles per rank, the same array on every rank
 *   stream   hipStream_t of the calling rank's GPU (NULL = default); the exchange is ordered after the work queued
 *            on it, and asynchronous: synchronise the stream before reading d_recv
 */
int rt_gather_shares(rt_gather *g, int32_t rank, const double *d_send, double *d_recv, const uint64_t *counts, void *stream);

int rt_gather_destroy(rt_gather *g); /* replaces MPI_Finalize (raytrace.cpp:238) */

const char *rt_gather_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_GATHER_H */
