/*
 * pct_corridor.h -- C ABI of libpct_corridor.so: the safe-region RRT* corridor finder
 * (include/pct_corridor_finder.hpp) for callers that cannot include C++ (tests, bench, cgo/ctypes).
 *
 * Replaces, call for call, the planner node's use of safeRegionRrtStar
 * (Planner/src/sim_planning_demo.cpp:143, 167, 346-347, 350, 354, 367, 399, 412-413, 416, 487, 761;
 * class surface Planner/include/pointcloudTraj/corridor_finder.h:81-149).  SafeRegionExpansion / Refine / Evaluate exist in both
 * forms: `_timed` takes the reference's wall-clock limit in seconds (corridor_finder.h:97-99, read before every iteration as at
 * corridor_finder.cpp:721-722, 774-775, 900-901, 950-951); the plain ones take iteration counts (deterministic runs).  A timed run
 * reports the iterations it consumed; the plain run of that many iterations gives the identical tree.
 */
#ifndef PCT_CORRIDOR_H
#define PCT_CORRIDOR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pct_corridor pct_corridor;

int pct_corridor_create(int64_t cloud_capacity, int device, pct_corridor **out);   /* 0 = ok */
void pct_corridor_destroy(pct_corridor *c);
const char *pct_corridor_last_error(void);

int pct_corridor_set_param(pct_corridor *c, double safety_margin, double search_margin, double max_radius, double sample_range);
int pct_corridor_reset(pct_corridor *c);
/* samples evaluated per GPU round trip in Expansion/Refine: 1 = one by one (the reference's loop); K > 1 = speculative batches
 * that are replayed in order and give the identical corridor */
int pct_corridor_set_speculation(pct_corridor *c, int k);
int pct_corridor_set_input(pct_corridor *c, const void *points, int64_t n, int64_t stride_bytes, int build_index);
int pct_corridor_set_pt(pct_corridor *c, const double start[3], const double end[3], double xl, double xh, double yl, double yh,
                        double zl, double zh, double local_range, int max_iter, double sample_portion, double goal_portion);
int pct_corridor_set_start_pt(pct_corridor *c, const double start[3], const double end[3]);
int pct_corridor_reset_root(pct_corridor *c, const double target[3]);
int pct_corridor_expansion(pct_corridor *c, int64_t iterations);
int pct_corridor_refine(pct_corridor *c, int64_t iterations);
int pct_corridor_evaluate(pct_corridor *c);
/* SafeRegionExpansion / Refine / Evaluate (double time_limit), corridor_finder.h:97-99; iterations_done may be NULL */
int pct_corridor_expansion_timed(pct_corridor *c, double time_limit, int64_t *iterations_done);
int pct_corridor_refine_timed(pct_corridor *c, double time_limit, int64_t *iterations_done);
int pct_corridor_evaluate_timed(pct_corridor *c, double time_limit);
int pct_corridor_check_traj_pt_col(pct_corridor *c, const double p[3], int *collides);
/* Path (k x 3, root first) and Radius (k); k through *n_out (may exceed cap; only cap rows written).
 * No path: the reference's placeholder, a 3x3 identity and three zero radii. */
int pct_corridor_get_path(pct_corridor *c, double *path, double *radius, int64_t cap, int64_t *n_out);
int pct_corridor_status(pct_corridor *c, int *path_exists, int *global_navi, int64_t *n_nodes, uint64_t *inflation_queries);
int pct_corridor_speculation_stats(pct_corridor *c, uint64_t *replayed_from_batch, uint64_t *fell_back);
/* on (default): one fused launch per speculative batch (nearest node -> steer -> inflation -> neighbourhood);
 * off: the same three stages as three batched launches.  The corridor does not depend on it. */
int pct_corridor_set_fused_expansion(pct_corridor *c, int on);
int pct_corridor_expansion_launches(pct_corridor *c, uint64_t *launches);
/* GPU round trips treeRepair (corridor_finder.cpp:938-1021) has made: two per pass -- one launch for every failed node's neighbourhood,
 * one for the sphere inflation of every node that may be re-checked -- where the reference's loop asks two per neighbour */
int pct_corridor_repair_batches(pct_corridor *c, uint64_t *batches);

#ifdef __cplusplus
}
#endif
#endif
