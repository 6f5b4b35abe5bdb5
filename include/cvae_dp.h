/* cvae_dp.h — the data-parallel gradient exchange of the train step as a C ABI over RCCL (libcvae_dp.so).
 *
 * The reference is single-process (causal_cascade/main.py:11 picks ONE device); the north star adds plain data parallelism: one process per GPU,
 * the per-rank gradients SUMMED (the losses are sums over the batch, causal_cascade/train.py:7,10,13 — never averaged) once per step over xGMI.
 * SURVEY.md §8(b) names this second tiny ABI: cvae_dp_init(rank, world, uniqueId), cvae_dp_allreduce_sum(flat_grads, n, dtype, stream).
 *
 * Conventions as in cvae_hip.h: plain pointers and sizes, enqueue-only on the caller's HIP stream, no device allocation, negative error codes
 * (CVAE_DP_E_*; cvae_dp_strerror), no hidden global state — a communicator is an opaque handle owned by the caller.  One communicator per process
 * and GPU; the calls of one communicator are made from one thread at a time.
 *
 * Exchange form: the flat bucket is reduced IN PLACE as reduce-scatter + all-gather over `world` equal slices (both collectives use every xGMI link
 * of the fully connected node; with `world` not dividing n the tail goes through one small all-reduce).  world == 1 is the identity (no launch). */
#ifndef CVAE_DP_H
#define CVAE_DP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CVAE_DP_OK 0
#define CVAE_DP_E_BADARG (-1)
#define CVAE_DP_E_NULLPTR (-2)
#define CVAE_DP_E_DTYPE (-3)
#define CVAE_DP_E_RCCL (-4)      /* an RCCL call failed: cvae_dp_last_rccl_error() has its text */
#define CVAE_DP_UNIQUE_ID_BYTES 128
#define CVAE_DP_F32 0            /* same codes as CVAE_F32 / CVAE_BF16 */
#define CVAE_DP_BF16 1

int cvae_dp_version(void);
const char* cvae_dp_strerror(int code);
const char* cvae_dp_last_rccl_error(void);

/* Rank 0 creates the id (128 bytes) and hands it to every rank by any out-of-band means (the Python side: one torch.distributed broadcast of 128 bytes,
 * or the launcher's store). */
int cvae_dp_unique_id(void* id128);
/* Collective over all ranks: builds this rank's communicator on the CURRENT HIP device.  *comm receives the handle. */
int cvae_dp_init(int rank, int world, const void* id128, void** comm);
int cvae_dp_world(const void* comm);
int cvae_dp_rank(const void* comm);
/* flat[0..n) <- sum over ranks, in place, enqueued on `stream` (capturable in a HIP graph as far as RCCL's kernels are).  dtype CVAE_DP_F32 / _BF16. */
int cvae_dp_allreduce_sum(void* comm, void* flat, size_t n, int dtype, void* stream);
/* The two halves separately (overlap: reduce-scatter the first bucket under the rest of the backward, all-gather later): slice r of n / world elements
 * (n % world == 0 required) holds the sum after the first call; the second call fills every other slice.  These two always go through RCCL, also on a
 * one-rank communicator (where cvae_dp_allreduce_sum returns at once): the one-rank tests run ncclReduceScatter / ncclAllGather through them. */
int cvae_dp_reduce_scatter_sum(void* comm, void* flat, size_t n, int dtype, void* stream);
int cvae_dp_all_gather(void* comm, void* flat, size_t n, int dtype, void* stream);
int cvae_dp_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* CVAE_DP_H */
