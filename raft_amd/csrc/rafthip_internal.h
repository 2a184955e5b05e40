/*
 * rafthip_internal.h — entry points of librafthip.so that are NOT part of the drop-in boundary
 * (include/rafthip.h): test seams.  A host never binds these.
 */
#ifndef RAFTHIP_INTERNAL_H
#define RAFTHIP_INTERNAL_H

#include "rafthip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A loopback communicator: the n engines, all on ONE GPU, become ranks 0..n-1 of one communicator
 * whose all-gather is n x n device-to-device copies ordered with events instead of RCCL (which
 * refuses two ranks on one GPU).  Every other step of the exchange — the reporting pass with
 * group_base, the speculative payload width and its redo, the packing of the gathered slots,
 * clipping at the per-rank capacity, landing buffers, the dense form with ranks of different plane
 * strides — is the code the RCCL communicators run, so a one-GPU box can execute the N > 1 branch
 * of rafthip_commit_allgather_many / _dense_many (tests/test_gpu_loopback.py).  Creation follows
 * rafthip_comm_create's per-rank path: rank i asks for max_pairs_per_rank[i] pairs (NULL or 0 = its
 * group count) and every rank ends up with the largest.  Use only through the _many calls, with all
 * n communicators in rank order; destroy each with rafthip_comm_destroy. */
int rafthip_internal_comm_create_loopback(rafthip_engine **engines, int n, const uint64_t *group_base,
                                          const uint64_t *max_pairs_per_rank, rafthip_comm **out);

#ifdef __cplusplus
}
#endif
#endif /* RAFTHIP_INTERNAL_H */
