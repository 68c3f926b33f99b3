/* gn2v_rccl.h -- a gn2v_comm (gn2v.h) filled with RCCL calls, shipped in libgn2v.so.
 *
 * What a non-Python binding of the multi-GPU fit needs besides gn2v_train_world: one process per
 * GPU, rank 0 makes the job's id and hands its 128 bytes to the other ranks by whatever the host
 * has (a file, a pipe, MPI, a socket), every rank creates its communicator and passes it to
 * gn2v_train_world.  The reference has no counterpart (ensmallen is one process:
 * embedders/ensmallen_embedders/node2vec.py:99).
 *
 * RCCL is loaded when the first of these functions is called (dlopen of librccl.so; GN2V_RCCL_LIB
 * names another file): libgn2v.so itself does not link against it, so a host that brings its own
 * communicator -- the Python side passes torch.distributed's through callbacks -- never loads a
 * second copy.
 *
 * The part exchanges run on a stream of the communicator's own, ordered after the caller's stream
 * at sendrecv_start and awaited by it at sendrecv_wait (they hide behind the training of the next
 * part); all-gathers and broadcasts run on the caller's stream. */
#ifndef GN2V_RCCL_H
#define GN2V_RCCL_H

#include "gn2v.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GN2V_RCCL_ID_BYTES 128 /* sizeof(ncclUniqueId) */

/* rank 0: a new job id into out[GN2V_RCCL_ID_BYTES] (ncclGetUniqueId) */
int gn2v_rccl_unique_id(void *out);
/* every rank, collectively: the communicator of `rank` of `world` on HIP device `device`
 * (ncclCommInitRank) -- fills *comm (ctx owns the RCCL communicator, a stream and an event) */
int gn2v_rccl_comm_create(const void *unique_id, uint32_t rank, uint32_t world, int device,
                          gn2v_comm *comm);
/* releases what gn2v_rccl_comm_create made (ncclCommDestroy); clears *comm */
int gn2v_rccl_comm_destroy(gn2v_comm *comm);

#ifdef __cplusplus
}
#endif
#endif
