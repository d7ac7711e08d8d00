/*
 * kdtree/kdtree_ext.h -- batch extensions of the drop-in libkdtree.so (NOT part of the reference API).
 *
 * The planner's RRT* loop asks one kd_nearestf and one kd_nearest_rangef per sample
 * (corridor_finder.cpp:428-437, 464).  These entry points answer K of them with one GPU launch each
 * against a SNAPSHOT of the tree, and let the caller complete the answers on the host for the few
 * nodes inserted after the snapshot -- so a speculative, batched expansion can reproduce the
 * sequential results exactly (include/pct_corridor_finder.hpp).
 * Node numbers are insertion indices since the last kd_clear (0 = first inserted).
 */
#ifndef PCT_KDTREE_EXT_H
#define PCT_KDTREE_EXT_H
#include <stdint.h>
#include "kdtree/kdtree.h"
#include "pct_engine.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Single-query dispatch of kd_nearest* / kd_nearest_range*: node sets of up to `nodes` nodes are answered from the host copy of the
 * node list (BASELINE.json config C1: "CPU Utils/kdtree path (plumbing, no GPU)" -- a launch costs ~11 us, the reference's walk
 * under one), larger ones by the HIP kernels; same arithmetic, same tie winner, same range order either way (the tests run every
 * fixture both ways).  Default 4096 (environment PCT_KD_HOST_MAX); 0 = always the device; negative = back to the default. */
void kdx_set_host_threshold(int64_t nodes);
int64_t kdx_host_threshold(void);
int kdx_size(struct kdtree *tree);
void *kdx_node_data(struct kdtree *tree, int32_t node);
/* stored (fp64) position of a node; returns 0 on success */
int kdx_node_pos(struct kdtree *tree, int32_t node, double pos[3]);

/* K x kd_nearestf in one launch: node_out[i] = nearest node of pos[3i..3i+2] (lowest index on exact ties), -1 for an
 * empty tree.  Returns 0 on success. */
int kdx_nearestf_batch(struct kdtree *tree, const float *pos, int k, int32_t *node_out);

/* K range queries in one launch: for query i the nodes with d2 <= range[i]^2, UNORDERED and not yet filtered by the
 * reference's traversal rule, at ids[i*cap_per_query ...]; counts[i] >= 0 hits stored, < 0 list truncated. */
int kdx_range_candidates_batch(struct kdtree *tree, const float *pos, const float *range, int k, uint32_t *ids,
                               int cap_per_query, int32_t *counts);

/* The result set kd_nearest_rangef(tree, pos, range) returns NOW, built on the host from a candidate list that holds
 * every in-range node with number < n_snapshot (e.g. from kdx_range_candidates_batch); nodes numbered >= n_snapshot
 * are tested here.  Same hits, same iteration order as the reference (kdtree.c:262-293, 810-828). */
struct kdres *kdx_range_from_candidates(struct kdtree *tree, const float *pos, float range, const uint32_t *ids, int n_ids,
                                        int32_t n_snapshot);

/* Per-node planner data for the fused expansion step: aux = {x, y, z, radius} -- the node's fp64 centre and its sphere
 * radius as the steer step reads them (corridor_finder.cpp:387-404).  Defaults to the stored position and radius 0. */
int kdx_set_node_aux(struct kdtree *tree, int32_t node, const double aux[4]);

/* K RRT* iterations' queries in ONE launch against the tree as it is now (pct_rrt_expand_batch): for sample i the nearest
 * node (as kd_nearestf on the fp32-narrowed sample), the steered centre, its inflation radius against `obstacles`, and the
 * candidate list kdx_range_from_candidates() needs for kd_nearest_rangef(centre, 2 * float(radius)).
 * Returns 0 on success, -1 when the fused path cannot serve this tree (more than 65536 nodes, obstacle cloud without its
 * cell index): the caller then issues the three queries separately. */
int kdx_expand_batch(struct kdtree *tree, pct_cloud *obstacles, const pct_inflate_params *prm, const double *samples, int k,
                     int cap_per_query, pct_expand_result *out, uint32_t *ids);

#ifdef __cplusplus
}
#endif
#endif
