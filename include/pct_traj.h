/* pct_traj.h -- C ABI of the Bezier-trajectory evaluators next to the collision check (libpct_engine.so).
 *
 * the steps right after the corridor/optimizer that share the device-side Bernstein
 * evaluator with pct_bezier_check (include/pct_engine.h):
 *
 *   pct_bezier_state_batch    getStateFromBezier           Planner/src/sim_planning_demo.cpp:688-713
 *   pct_traj_wire_from_matrix getBezierTraj's packing      Planner/src/sim_planning_demo.cpp:543-562, 578-581
 *   pct_traj_wire_sample      the 1001-samples-per-segment loops of
 *   pct_traj_segm_index         get_segm_index             Planner/src/traj_postprocessing.cpp:29-57
 *   pct_traj_nearest_voxels     to_nearest_traj            Planner/src/traj_postprocessing.cpp:59-90
 *   pct_traj_end_yaws         to_poly_traj's yaw block     Planner/src/traj_postprocessing.cpp:152-179
 *
 * Arithmetic: fp64 in the reference's accumulation order, no FMA; pow() is the device library's, which can differ from
 * glibc's in the last ulp (positions are held to 1e-12 relative by the tests, everything discrete -- sample counts,
 * segment index, voxel keys away from rounding boundaries -- exactly).  The sequential `len -= step` loops run on the
 * host over the step lengths the kernel produced, in the reference's order.  Orders up to 12 (binomial_coefs.h:11 holds
 * MAX_N = 13 rows).  Not thread-safe (one workspace per process), like the rest of the library.
 */
#ifndef PCT_TRAJ_H
#define PCT_TRAJ_H

#include <stdint.h>

#include "pct_engine.h"
#include "pct_voxel.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The trajectory as it travels in quadrotor_msgs/PolynomialTrajectoryExtra (msg/PolynomialTrajectoryExtra.msg:19-31):
 * control points of all segments concatenated, order[s] + 1 of them for segment s, one array per axis. */
typedef struct pct_traj_wire {
    const double *coef_x, *coef_y, *coef_z;   /* ncoef each */
    int64_t ncoef;                            /* sum over segments of order[s] + 1 */
    const double *time;                       /* num_segment */
    const uint32_t *order;                    /* num_segment, each <= 12 */
    int32_t num_segment;
} pct_traj_wire;

/* getStateFromBezier for n (segment, u) pairs: state9[9*i + 0..2] = position sums, +3..5 = velocity sums, +6..8 = acceleration
 * sums, exactly the vector `ret` of the reference (the caller scales by the segment time as trajGeneration does, :223-227). */
int pct_bezier_state_batch(const pct_bezier_traj *traj, const int32_t *seg, const double *u, int64_t n, double *state9);

/* PolyCoeff matrix rows -> wire arrays (getBezierTraj).  coef_* must hold sum(orders[s] + 1) doubles; returns that count. */
int pct_traj_wire_from_matrix(const pct_bezier_traj *traj, double *coef_x, double *coef_y, double *coef_z, int64_t cap, int64_t *ncoef);

/* Positions of samples i = 0..samples-1 (t = i / (samples - 1.0); the reference uses samples = 1001 -> t = i / 1000.0) of every
 * segment, in segment order: pos[(s * samples + i) * 3 + d], and the distance of each sample to its predecessor
 * (the first sample's predecessor is coef(0) * time[0], traj_postprocessing.cpp:32/62).  Either output may be NULL. */
int pct_traj_wire_sample(const pct_traj_wire *w, int32_t samples, double *pos, double *step_len);

/* get_segm_index: walk the samples until `twirl_len` of arc length is used up -> (segment, 0 | 1 for first/second half) */
int pct_traj_segm_index(const pct_traj_wire *w, double twirl_len, int32_t *segm, int32_t *part);

/* to_nearest_traj: voxelise (m's resolution) the samples covering the first `twirl_len` of arc length, in order.  `m` is
 * cleared first and then holds the voxel cloud (read it with pct_voxel_map_get_f32 / _soa_dev); points_used = samples added. */
int pct_traj_nearest_voxels(const pct_traj_wire *w, double twirl_len, pct_voxel_map *m, int64_t *points_used);

/* to_poly_traj's end_yaws (= middle_yaws) for n = radii.size() corridor spheres: heading of path[i] -> path[i+1] when they are
 * more than 0.01 apart in the xy-plane, else 10 (the reference's "no yaw" marker); the last one repeats its predecessor, or,
 * when n == 1, comes from the first two control points.  Host arithmetic (atan2), n values out. */
int pct_traj_end_yaws(const double *path_x, const double *path_y, int64_t n, const double *coef_x, const double *coef_y, double *end_yaws);

/* test hook: the two device-side statements of "n choose k" (Pascal's rule of the trajectory evaluators, the multiplicative
 * recurrence of the collision-check kernels) as 13 x 13 tables of doubles, 0 above the diagonal */
int pct_debug_binomials(double *pascal, double *recurrence);

#ifdef __cplusplus
}
#endif
#endif
