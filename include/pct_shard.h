/*
 * pct_shard.h -- C ABI of the multi-GPU exchange step (libpct_shard.so = libpct_engine.so + RCCL).
 *
 * One process per GPU.  The obstacle cloud is split into contiguous index ranges (rank r of W owns
 * [r*N/W, (r+1)*N/W), global index = range begin + local index), every rank holds its range in its own HBM as an ordinary
 * pct_cloud and answers the whole, replicated query batch on it; ONE exchange step over xGMI merges the per-shard winners:
 *
 *     d2*  = ncclAllReduce(min) of the per-shard squared distances               (fp64: exact)
 *     idx* = ncclAllReduce(min) of  (global index  if  d2_local == d2*  else  INT32_MAX)   -> lowest index on exact ties
 *     count = ncclAllReduce(sum) of the per-shard radius counts
 *
 * with one small kernel between the two reductions (pct_merge_mask_dev) and one behind them (pct_merge_finish_dev).
 * Everything is queued on the caller's stream: no host synchronisation inside the *_dev entry points.
 *
 * The reference has no multi-GPU code; the contract is SURVEY.md section 8(e) / BASELINE.json config 4.
 * The planner process (a C++ ROS node, Planner/src/sim_planning_demo.cpp:89-90) links this library next to libpct_engine.so;
 * examples/shard_client.cpp is such a process.  Global indices must stay below 2^31 - 1 (they travel as int32).
 */
#ifndef PCT_SHARD_H
#define PCT_SHARD_H

#include <stdint.h>

#include "pct_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pct_shard pct_shard;     /* opaque: this rank's communicator + exchange workspaces */

#define PCT_SHARD_ID_BYTES 128          /* = NCCL_UNIQUE_ID_BYTES */

/* rank 0 makes the rendezvous token (ncclGetUniqueId) and hands its 128 bytes to every rank by any channel it has */
int pct_shard_unique_id(void *id_out);
/* selects `device` for this process (pct_init) and joins the communicator (ncclCommInitRank): collective over all ranks */
int pct_shard_init(const void *id, int rank, int world, int device, pct_shard **out);
/* the same over a communicator the caller already owns (an ncclComm_t; not destroyed by pct_shard_destroy) */
int pct_shard_init_comm(void *nccl_comm, int rank, int world, pct_shard **out);
int pct_shard_destroy(pct_shard *s);
int pct_shard_rank(const pct_shard *s);
int pct_shard_world(const pct_shard *s);

/* [begin, end) of this rank's contiguous index range of an n_total-point cloud */
int pct_shard_range(const pct_shard *s, int64_t n_total, int64_t *begin, int64_t *end);
/* this rank's shard as a pct_cloud: capacity end - begin (at least 1), index base = begin; upload / index it like any cloud */
int pct_shard_cloud_create(pct_shard *s, int64_t n_total, pct_cloud **out);

/* Q replicated queries (device, Q x 3 fp32) against the sharded cloud: per-shard kernels (pct_nn_batch_dev with `algo`) + the
 * exchange step, all on `stream`.  d_idx[Q]: global index of the nearest point of the WHOLE cloud (lowest index on exact
 * ties, PCT_NO_INDEX when every shard is empty), d_d2[Q]: its fp64 squared distance.  Every rank gets the same answers. */
int pct_shard_nn_dev(pct_shard *s, pct_cloud *local, int algo, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, void *stream);
/* d_count[Q] = number of points of the whole cloud with d2 <= r*r */
int pct_shard_radius_count_dev(pct_shard *s, pct_cloud *local, int algo, const float *d_q, const float *d_r, int64_t Q,
                               uint32_t *d_count, void *stream);
/* host buffers, synchronous: upload the batch, run, download */
int pct_shard_nn(pct_shard *s, pct_cloud *local, int algo, const float *q, int64_t Q, uint32_t *idx, double *d2);

/* ---- Routed form: slab ownership (the N > 1 throughput path).
 * Under index-range sharding every rank answers every query, and a cell-pruned query costs what the local point density makes it
 * cost, whatever the shard's size: W ranks do W times the work.  pct_shard_route_build re-distributes the cloud ONCE into W slabs
 * of equal point count along its longest axis (plus a halo of `halo_spacings` mean point spacings on both sides; rows keep their
 * ascending global order), after which every query is answered by the one rank that owns its slab (1/W of the batch per rank), the
 * owned answers are exchanged as 16-byte records {query, global index, d2} -- an all-gather of variable-sized slices: grouped
 * ncclSend / ncclRecv over xGMI, Q x 16 B x (W-1)/W received per rank instead of two all-reduces over the whole batch -- and an
 * answer the owner cannot certify (the point found is not strictly nearer than the edge of its halo) is answered by everybody in
 * a second round merged like the index-range form.  Identical results to the single cloud (lowest global index on exact ties).
 * The entry point synchronises `stream` twice per batch (the owned share sizes the next launch and the exchange; the
 * number of uncertified answers decides whether there is a second round). ---- */
typedef struct pct_route pct_route;
/* collective over all ranks.  local_points: this rank's rows [index_begin, index_begin + n_local) of the global cloud, host memory,
 * x,y,z fp32 at the start of every stride_bytes-byte record */
int pct_shard_route_build(pct_shard *s, const void *local_points, int64_t n_local, int64_t stride_bytes, int64_t index_begin,
                          double halo_spacings, pct_route **out);
/* Q replicated queries (device, the same on every rank); every rank gets all Q answers */
int pct_shard_route_nn_dev(pct_route *r, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, void *stream);
int pct_shard_route_stats(const pct_route *r, int64_t *slab_points, uint64_t *owned, uint64_t *uncertified, uint64_t *batches);
int pct_shard_route_destroy(pct_route *r);

/* PARTITIONED batches: every rank brings its OWN Q queries (Q may differ per rank) and gets its own Q answers; the queries travel to the
 * rank that owns their slab and the answers travel back (two variable-sized all-to-alls of 16 bytes per query, grouped ncclSend /
 * ncclRecv); nothing is replicated and nobody answers a query twice -- weak scaling: the job answers the sum of all batches per step.
 * Uncertified answers of all ranks are gathered and answered by everybody in a second round.  Collective; three stream synchronises. */
int pct_shard_route_nn_partitioned_dev(pct_route *r, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, void *stream);

/* Several ranks in ONE process, on the process's one device: the same phases, the bytes moved between the ranks' buffers by plain
 * copies instead of RCCL (which refuses two ranks on one device).  A rehearsal of W ranks on one card (tests), and the building
 * block for a single process that keeps several slabs.  out: `world` handles, to be released with pct_shard_destroy. */
int pct_shard_local_world(int world, pct_shard **out);
int pct_shard_route_build_world(pct_shard *const *ranks, int world, const void *const *local_points, const int64_t *n_local, int64_t stride_bytes,
                                const int64_t *index_begin, double halo_spacings, pct_route **out);
/* d_idx[k] / d_d2[k]: rank k's result arrays (Q entries each; they may all be the same arrays) */
int pct_shard_route_nn_world(pct_route *const *routes, int world, const float *d_q, int64_t Q, uint32_t *const *d_idx, double *const *d_d2, void *stream);

/* partitioned batches with every rank in this process: d_q[k] / Q[k] = rank k's own batch, d_idx[k] / d_d2[k] = its Q[k] answers */
int pct_shard_route_nn_partitioned_world(pct_route *const *routes, int world, const float *const *d_q, const int64_t *Q, uint32_t *const *d_idx,
                                         double *const *d_d2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCT_SHARD_H */
