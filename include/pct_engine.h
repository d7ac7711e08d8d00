/*
 * pct_engine.h -- C ABI of the MI355X obstacle-cloud engine (libpct_engine.so).
 *
 * This is the batched extension SURVEY.md section 8(b) specifies next to the kd_* functions
 * (include/kdtree/kdtree.h): an opaque cloud handle in HBM plus batch queries that replace
 * the per-point PCL/FLANN calls on the planner's hot path.  Plain pointers and sizes only;
 * int status codes; no exceptions cross the boundary; host buffers are caller-owned.
 *
 * Reference interface each entry point replaces (paths relative to /root/reference/):
 *   pct_cloud_upload_*      safeRegionRrtStar::setInput           Planner/src/corridor_finder.cpp:93-99
 *                           (rcvPointCloudCallBack                Planner/src/sim_planning_demo.cpp:159-167)
 *   pct_nn_batch            kdtreeForMap.nearestKSearch(p,1,..)   Planner/src/corridor_finder.cpp:130
 *                           with kd_nearestf arithmetic           Utils/kdtree/src/kdtree.c:345-491
 *   pct_radius_count_batch  kd_nearest_rangef + kd_res_size       Utils/kdtree/src/kdtree.c:262-293,561-593,620-623
 *   pct_inflate_batch       safeRegionRrtStar::radiusSearch       Planner/src/corridor_finder.cpp:113-133
 *                           (+ checkRadius :656-659, checkTrajPtCol :412-416), batched over the loops at
 *                           :829-835 (SafeRegionEvaluate) and :958-974 (treeRepair)
 *   pct_bezier_check        checkSafeTrajectory/getPosFromBezier  Planner/src/sim_planning_demo.cpp:715-781
 *   pct_cloud_ring_index +  the per-frame rebuild of the search tree: rcvPointCloudCallBack -> setInput
 *   pct_cloud_append_aos    Planner/src/sim_planning_demo.cpp:159-167 -> Planner/src/corridor_finder.cpp:93-99 (rolling map, config C5)
 *   pct_ctrl_points_check   the containment the optimizer enforces on the control points, Planner/src/traj_optimizer.cpp:624-648,
 *                           as the threshold test of checkTrajPtCol (corridor_finder.cpp:412-416) on control point * T_i
 *   pct_plan_create_replan  one replan tick's query side in one captured graph: SafeRegionEvaluate's re-check loop
 *   pct_plan_replan_run     corridor_finder.cpp:829-835 + checkSafeTrajectory sim_planning_demo.cpp:729-781 + the control points
 *   pct_*_dev               the same batches on device buffers and the caller's stream (multi-GPU: include/pct_shard.h)
 *
 * Arithmetic contract (what "parity" means): coordinates are fp32 in HBM; every distance is
 * computed in fp64 from the float-widened operands as ((dx*dx + dy*dy) + dz*dz) with one
 * rounding per operation and no fused multiply-add -- bit-identical to kdtree.c:379-382.
 * Nearest neighbour: the minimum of that value; among fp64-equal minima the LOWEST index
 * wins (the reference's winner on exact ties depends on tree shape).  Radius count:
 * d2 <= (double)r * (double)r, inclusive (kdtree.c:273).
 *
 * All entry points need a HIP device; there is no host fallback.
 */
#ifndef PCT_ENGINE_H
#define PCT_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pct_cloud pct_cloud;     /* opaque: an obstacle cloud resident in HBM */
typedef struct pct_plan pct_plan;       /* opaque: a hipGraph-captured fixed-shape query batch */

enum pct_status {
    PCT_OK = 0,
    PCT_ERR_NO_DEVICE = 1,   /* no HIP device / runtime failure at init */
    PCT_ERR_INVALID = 2,     /* bad argument */
    PCT_ERR_ALLOC = 3,       /* host or device allocation failed */
    PCT_ERR_HIP = 4,         /* a HIP call failed; see pct_last_error() */
    PCT_ERR_EMPTY = 5,       /* query against an empty cloud (outputs are still filled: idx=PCT_NO_INDEX, d2=+inf) */
    PCT_ERR_CAPACITY = 6,    /* more points than the cloud's capacity */
    PCT_ERR_INTERNAL = 7     /* a self-check of the library failed (e.g. a built index whose records are not a permutation of the cloud) */
};

#define PCT_NO_INDEX 0xFFFFFFFFu

enum pct_algo {
    PCT_ALGO_AUTO = 0,       /* grid kernel when a grid is built, streaming kernel otherwise */
    PCT_ALGO_STREAM = 1,     /* brute-force SoA streaming kernels: fp32 filter + exact fp64 recheck (no index needed) */
    PCT_ALGO_GRID = 2,       /* cell-pruned kernel (needs pct_cloud_build_grid) */
    PCT_ALGO_STREAM_EXACT = 3 /* brute force with every pair in fp64 (the filter's reference; same results) */
};

/* ---- process / device ---------------------------------------------------------------- */
int pct_init(int device);                       /* select the HIP device for this process */
const char *pct_last_error(void);               /* thread-local text of the last failure */
int pct_device_count(void);
int pct_sync(void);                             /* wait for everything queued by this library */

/* ---- cloud lifecycle ----------------------------------------------------------------- */
int pct_cloud_create(int64_t capacity, pct_cloud **out);
/* a cloud whose coordinates live in host-mapped memory: appends are plain host stores (no launch, no copy),
 * kernels read over the bus.  For small, frequently growing point sets (the RRT* node tree behind kd_*). */
int pct_cloud_create_small(int64_t capacity, pct_cloud **out);
int pct_cloud_destroy(pct_cloud *c);
int64_t pct_cloud_size(const pct_cloud *c);
int64_t pct_cloud_capacity(const pct_cloud *c);
/* index reported for local point i is base + i (multi-GPU shards; default 0) */
int pct_cloud_set_index_base(pct_cloud *c, int64_t base);

/* Replace the cloud.  `pts` is array-of-structures: x,y,z fp32 at byte offsets 0,4,8 of
 * each `stride_bytes` record (12 = packed, 16 = pcl::PointXYZ).  De-interleaved to SoA on
 * the device.  Drops any grid. */
int pct_cloud_upload_aos(pct_cloud *c, const void *pts, int64_t n, int64_t stride_bytes);
/* Same for a sensor_msgs/PointCloud2 payload: n records of point_step bytes, FLOAT32 fields at the given byte
 * offsets (msg.fields[i].offset), any order, any padding (sim_planning_demo.cpp:159-167). */
int pct_cloud_upload_fields(pct_cloud *c, const void *data, int64_t n, int64_t point_step, int64_t off_x, int64_t off_y, int64_t off_z);
/* Same, from three device arrays (already SoA, e.g. produced on the GPU). */
int pct_cloud_upload_soa_dev(pct_cloud *c, const float *d_x, const float *d_y, const float *d_z, int64_t n);
/* Rolling map: append n points, overwriting the oldest once capacity is reached (ring).
 * Index of a point = its slot in the ring.  Drops any cell-sorted grid; updates the rolling-map index in place.
 * `pts` is the caller's again when the call returns.  On a rolling-map cloud a frame of up to 4 MB is copied to a staging buffer
 * and the call returns once the launches are queued (every later call on the cloud is ordered behind them on the library's
 * stream); an error of those launches is reported by the next call on the cloud. */
int pct_cloud_append_aos(pct_cloud *c, const void *pts, int64_t n, int64_t stride_bytes);

/* Rolling-map index (config C5: the obstacle map is a sliding window fed one sensor frame at a time, where the reference
 * rebuilds its search tree per frame -- safeRegionRrtStar::setInput, Planner/src/corridor_finder.cpp:93-99 called from
 * rcvPointCloudCallBack, Planner/src/sim_planning_demo.cpp:159-167).  After this call pct_cloud_append_aos no longer drops an
 * index: it retires the points it overwrites from a world-anchored bucket table and files the new frame, in place, and
 * pct_nn_batch / pct_inflate_batch / pct_bezier_check / pct_ctrl_points_check / the replan plan search that table (ALGO_AUTO
 * and ALGO_GRID; ALGO_STREAM still scans the whole window).  Results are the same as on any other cloud: exact fp64
 * distances, lowest ring slot on ties.  cell_size <= 0: chosen from the first data (about 6 points per cell at capacity);
 * extent (may be NULL): the window's size per axis, when the caller knows it (e.g. the sensing range) -- the table is then
 * allocated at once.  Memory: 512 B per bucket (32 records of 16 B), buckets = the extent / cell_size per axis plus a quarter,
 * rounded up to powers of two; a cell holding more records than a bucket has room for spills to a queue every query scans
 * exhaustively (pct_cloud_ring_info reports its length), and when that queue holds more than about 1 % of the window the buckets
 * are doubled -- 64, 128, at most 256 records, 4 KiB per bucket, the table never beyond 32 GiB -- and the window is filed again
 * (pct_cloud_ring_bucket_records).  Not available for small (host-mapped) clouds; excludes pct_cloud_build_grid. */
int pct_cloud_ring_index(pct_cloud *c, float cell_size, const float extent[3]);
int pct_cloud_ring_drop(pct_cloud *c);
int pct_cloud_has_ring_index(const pct_cloud *c);
int pct_cloud_ring_info(pct_cloud *c, int32_t dims[3], double *cell_size, int64_t *overflow_entries);
/* records per bucket of the rolling-map index: 32 to begin with; doubled (up to 256, the window filed again) when more than ~1 % of the window
 * sits in the overflow queue although the cells were sized from the window -- surfaces on a lattice finer than the cell, the same points sensed
 * frame after frame (the reference's rgbd mode, camera_sensor.cpp:160-166).  0 = no rolling-map index. */
int pct_cloud_ring_bucket_records(const pct_cloud *c);
/* Zero-copy ingest.  pct_cloud_frame_buffer hands out a host-mapped staging buffer of at least `bytes` bytes (valid until the next
 * call that asks for a larger one, or pct_cloud_destroy); the producer -- a sensor driver, the deserialiser of a
 * sensor_msgs/PointCloud2 -- writes the frame's records there (x, y, z floats at the start of each stride-byte record) and
 * pct_cloud_append_frame appends the first n of them exactly as pct_cloud_append_aos would (rcvPointCloudCallBack,
 * sim_planning_demo.cpp:159-178), minus the host-side copy: on a rolling-map cloud the insert kernel reads the buffer over the bus. */
int pct_cloud_frame_buffer(pct_cloud *c, int64_t bytes, void **host_ptr);
int pct_cloud_append_frame(pct_cloud *c, int64_t n, int64_t stride_bytes);

/* diagnostics (tests): where ring slot `slot`'s record is filed: out = {where word, bucket of the slot's coordinates, head, tail of
 * that bucket (or of the overflow queue, bit 31 of the where word), id word stored at the filed position, overflow queue length} */
int pct_debug_ring_slot(pct_cloud *c, int64_t slot, uint32_t out[6]);

/* Build / drop the uniform-cell index used by PCT_ALGO_GRID.  cell_size <= 0 picks one from
 * the bounding box and point count (about `pct` points per cell). */
int pct_cloud_build_grid(pct_cloud *c, float cell_size);
int pct_cloud_drop_grid(pct_cloud *c);
int pct_cloud_has_grid(const pct_cloud *c);
/* grid facts for tests/bench: dims[3], cell size, origin[3], number of cells */
int pct_cloud_grid_info(const pct_cloud *c, int32_t dims[3], float *cell_size, float origin[3], int64_t *ncells);

/* ---- batch queries, host buffers, synchronous ------------------------------------------ */
/* q: Q x 3 fp32.  idx[Q] (index_base + local index), d2[Q] fp64. */
int pct_nn_batch(pct_cloud *c, const float *q, int64_t Q, uint32_t *idx, double *d2);
int pct_nn_batch_algo(pct_cloud *c, int algo, const float *q, int64_t Q, uint32_t *idx, double *d2);
/* same with fp64 query coordinates (kd_nearest's double positions); streaming kernel */
int pct_nn_batch_q64(pct_cloud *c, const double *q, int64_t Q, uint32_t *idx, double *d2);
/* the same, also reporting how many points attain the minimum: ties[i] >= 1 where counted (single queries against clouds of up
 * to 16384 points -- the RRT* node sets of the kd_* drop-in), 0 = not counted on the path taken.  ties may be NULL. */
int pct_nn_batch_q64_ties(pct_cloud *c, const double *q, int64_t Q, uint32_t *idx, double *d2, uint32_t *ties);
/* count[Q] = #points with d2 <= r*r */
int pct_radius_count_batch(pct_cloud *c, const float *q, const float *r, int64_t Q, uint32_t *count);
int pct_radius_count_batch_algo(pct_cloud *c, int algo, const float *q, const float *r, int64_t Q, uint32_t *count);
/* lidar-style crop (camera_sensor.cpp:133-145): indices of all points within r of ONE centre,
 * ascending index order; returns the count through *n_out (may exceed cap; only cap written). */
int pct_radius_indices(pct_cloud *c, const float q[3], float r, uint32_t *idx_out, int64_t cap, int64_t *n_out);
int pct_radius_indices_q64(pct_cloud *c, const double q[3], double r, uint32_t *idx_out, int64_t cap, int64_t *n_out);
/* the same with the SQUARED radius given exactly (d2 <= r2): with r2 = the d2 a nearest-neighbour query returned it lists every
 * point tied at the minimum */
int pct_radius_indices_r2_q64(pct_cloud *c, const double q[3], double r2, uint32_t *idx_out, int64_t cap, int64_t *n_out);
/* The same crop with everything its consumer builds from it: idx_out[cap], d2_out[cap] (fp64, kdtree.c arithmetic) and
 * xyz_out[cap*3] (the cropped cloud = pcl::PointCloud(cloud, indices)); any of the three may be NULL.  Order: ascending
 * index, or -- sort_by_distance != 0 -- nearest first with ties in ascending index, the order pcl's radiusSearch returns
 * (PCL's own fp32 distances are parity-unpinned; cap must then hold every hit).  *n_out = number of hits. */
int pct_radius_crop(pct_cloud *c, const double q[3], double r, int sort_by_distance, int64_t cap, uint32_t *idx_out, double *d2_out,
                    float *xyz_out, int64_t *n_out);
/* dst := the points of src within r of q, in src's order, device to device (camera_sensor.cpp:398-401 known_map_pcl;
 * a lidar frame cropped out of a resident world map).  dst's cell index is dropped; rebuild it with pct_cloud_build_grid. */
int pct_cloud_crop_to(pct_cloud *src, const double q[3], double r, pct_cloud *dst);
/* K range queries against a SMALL cloud (<= 65536 points) in one launch.  ids_out[k*cap_per_query + j] in arrival order;
 * counts_out[k] >= 0: number of hits, all stored; < 0: -(number of hits), list truncated -- ask that query alone. */
int pct_radius_indices_batch_q64(pct_cloud *c, const double *q, const double *r, int64_t K, uint32_t *ids_out, int64_t cap_per_query,
                                 int64_t *counts_out);

/* ---- node sets of any dimension with fp64 coordinates (the general form of the kd_* API: kd_create(k), double positions --
 * Utils/kdtree/src/kdtree.c:112-131, 167-209).  Rows of `dim` doubles in HBM, insertion order = node number; every distance is
 * the reference's sum  s = 0; for i < dim: s += (row[i] - q[i])^2  in that order, in fp64, without contraction
 * (kdtree.c:267-272, 379-382, 420-423).  Exhaustive kernels: these sets are search trees of a planner (10^3 .. 10^6 nodes), not
 * obstacle clouds -- 3-D fp32 clouds belong in a pct_cloud. */
typedef struct pct_nodeset pct_nodeset;
int pct_nodeset_create(int dim, int64_t capacity, pct_nodeset **out);      /* 1 <= dim <= 1024; capacity grows on append */
int pct_nodeset_destroy(pct_nodeset *s);
int pct_nodeset_clear(pct_nodeset *s);
int64_t pct_nodeset_size(const pct_nodeset *s);
int pct_nodeset_dim(const pct_nodeset *s);
int pct_nodeset_append(pct_nodeset *s, const double *rows, int64_t n);     /* n rows of dim doubles, host memory */
/* nearest node of ONE query (dim doubles): idx = the LOWEST node number at the minimum distance, d2 = that distance, ties = how
 * many nodes attain it.  Empty set: idx = PCT_NO_INDEX, d2 = +inf, ties = 0. */
int pct_nodeset_nearest(pct_nodeset *s, const double *q, uint32_t *idx, double *d2, uint32_t *ties);
/* node numbers with d2 <= r2, ascending; *n_out = number of hits (may exceed cap; only cap written) */
int pct_nodeset_radius_indices_r2(pct_nodeset *s, const double *q, double r2, uint32_t *idx_out, int64_t cap, int64_t *n_out);

typedef struct pct_inflate_params {
    double start[3];        /* start_pt */
    double sample_range;    /* early-out: |p - start| > sample_range + max_radius */
    double search_margin;
    double max_radius;
} pct_inflate_params;

/* pts: Q x 3 fp64 (the planner's Vector3d).  radius[Q] as radiusSearch returns it;
 * idx/d2 (optional, may be NULL) = NN, or PCT_NO_INDEX / +inf where the early-out fired. */
int pct_inflate_batch(pct_cloud *c, const pct_inflate_params *p, const double *pts, int64_t Q,
                      double *radius, uint32_t *idx, double *d2);

/* ---- one RRT* iteration's three dependent queries in ONE launch (corridor_finder.cpp:385-410 genNewNode, :428-437
 * findNearstVertex, :464 treeRewire's neighbourhood), for K samples at once: nearest tree node of the fp32-narrowed
 * sample -> steer -> sphere inflation of the steered centre against `obstacles` -> tree nodes within 2*float(radius) of the
 * centre.  `nodes` is the small (host-mapped) cloud of node coordinates; pct_cloud_small_aux() hands out its per-node
 * planner data, 4 doubles per node {x, y, z, radius} (the node's fp64 centre and float radius, as the steer step reads
 * them), which the caller keeps current with plain stores.  ids[k*cap_per_query ...] = candidate node numbers (unordered);
 * out[k].count < 0 means -(count) hits of which only part were stored.  near_idx = -1 for an empty node set (centre = sample). */
typedef struct pct_expand_result { double center[3]; double radius; int32_t near_idx; int32_t count; } pct_expand_result;
int pct_cloud_small_aux(pct_cloud *nodes, double **host_aux);
int pct_rrt_expand_batch(pct_cloud *nodes, pct_cloud *obstacles, const pct_inflate_params *p, const double *samples, int64_t K,
                         int64_t cap_per_query, pct_expand_result *out, uint32_t *ids);

typedef struct pct_bezier_traj {
    const double *polycoef;   /* nseg rows of row_stride doubles: [x_0..x_n, y_0..y_n, z_0..z_n], n = orders[seg] */
    int64_t row_stride;       /* 3 * (max_order + 1) */
    const double *seg_time;   /* nseg */
    const int32_t *orders;    /* nseg, each <= 12 */
    int32_t nseg;
} pct_bezier_traj;

/* Sample the trajectory every dt from t_start over stop_time (reference loop semantics),
 * inflate every sample, report the first one with negative radius (NN distance <
 * search_margin).  first_hit = -1 when none.  Optional per-sample outputs (capacity cap):
 * pos (cap x 3 fp64), radius, d2, idx. */
int pct_bezier_check(pct_cloud *c, const pct_bezier_traj *traj, const pct_inflate_params *p,
                     double t_start, double stop_time, double dt,
                     int64_t *first_hit, int64_t *nsamples,
                     int64_t cap, double *pos, double *radius, double *d2, uint32_t *idx);

/* Control-point check (SURVEY.md 3.3, config C5's build extension): the threshold test of checkTrajPtCol
 * (Planner/src/corridor_finder.cpp:412-416) applied to the raw control points of the committed trajectory in world units --
 * control point j of segment i is polycoef[i][d*(n+1)+j] * seg_time[i] (layout Planner/src/traj_optimizer.cpp:739-751), the
 * point the optimizer's cone constraint keeps inside corridor sphere i (Planner/src/traj_optimizer.cpp:624-648).  Segments
 * from the one holding t_start on (the segment search of checkSafeTrajectory, sim_planning_demo.cpp:735-741), j ascending.
 * first_hit = index into that list of the first control point with radiusSearch(point) < 0, -1 when none; nctrl = its length;
 * optional per-point outputs (capacity cap): pos (cap x 3), radius, d2, idx. */
int pct_ctrl_points_check(pct_cloud *c, const pct_bezier_traj *traj, const pct_inflate_params *p, double t_start,
                          int64_t *first_hit, int64_t *nctrl, int64_t cap, double *pos, double *radius, double *d2, uint32_t *idx);

/* ---- batch queries, DEVICE buffers, asynchronous on `stream` (a hipStream_t; NULL = HIP's null
 * stream, as in any HIP call -- that is also PyTorch's default stream).  Every kernel of the batch is
 * ordered on that stream and nothing else, so work the caller queues behind it (a collective, a copy)
 * sees the results.  One batch at a time per cloud: the cloud's scratch buffers are shared, so do not
 * issue batches on the same cloud from two streams concurrently.  A rolling-map append that is still running on the library's
 * own stream (pct_cloud_append_aos returns once its launches are queued) is waited for through an event, so a batch issued right
 * after it on `stream` sees the appended frame.  For torch.distributed sharding and
 * graph capture.  An empty shard yields idx=PCT_NO_INDEX, d2=+inf and PCT_OK. ---------------------- */
int pct_nn_batch_dev(pct_cloud *c, int algo, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, void *stream);
int pct_radius_count_batch_dev(pct_cloud *c, int algo, const float *d_q, const float *d_r, int64_t Q, uint32_t *d_count, void *stream);
/* Stream variants of the planner arithmetic (same results as pct_inflate_batch / pct_bezier_check, nothing crosses the bus but the
 * trajectory's coefficients): d_pts = Q x 3 fp64 planner points on the device; d_radius[Q] required, d_idx / d_d2 optional.
 * Reserve the batch size first (pct_cloud_reserve_queries). */
int pct_inflate_batch_dev(pct_cloud *c, const pct_inflate_params *p, const double *d_pts, int64_t Q, double *d_radius, uint32_t *d_idx,
                          double *d_d2, void *stream);
/* traj's arrays are host memory (copied on the stream: keep them alive until the stream has passed the call); outputs on the device:
 * d_radius[cap] required, d_pos[3*cap] / d_d2[cap] / d_idx[cap] optional, *d_first_hit (int64, -1 = none), *d_nsamples (int32: the
 * number of samples the reference would evaluate; the first min(nsamples, cap) slots are valid).  cap <= 4096 and <= the reserved
 * batch size. */
int pct_bezier_check_dev(pct_cloud *c, const pct_bezier_traj *traj, const pct_inflate_params *p, double t_start, double stop_time, double dt,
                         int64_t cap, double *d_pos, double *d_radius, double *d_d2, uint32_t *d_idx, long long *d_first_hit, int32_t *d_nsamples,
                         void *stream);
/* Exchange step of a sharded cloud (one process per GPU): between all_reduce(min) on the squared distances and all_reduce(min) on
 * the indices, a rank offers its global index only where its own d2 equals the reduced minimum (and is finite), INT32_MAX
 * elsewhere -- so the second reduction returns the lowest global index among the ranks that tie.  Device pointers, async on
 * `stream`; global indices must be < 2^31 - 1. */
int pct_merge_mask_dev(const double *d_d2_local, const double *d_d2_best, const uint32_t *d_idx_local, int32_t *d_cand, int64_t Q, void *stream);
/* behind the second reduction: merged int32 candidates -> u32 indices (INT32_MAX -> PCT_NO_INDEX).  include/pct_shard.h wraps the
 * whole exchange step (these two kernels + the RCCL calls) for C / C++ callers. */
int pct_merge_finish_dev(const int32_t *d_cand, uint32_t *d_idx, int64_t Q, void *stream);
/* ---- device helpers of the spatially routed multi-GPU form (include/pct_shard.h; the host logic lives in libpct_shard.so).
 * cuts: world + 1 ascending slab boundaries along `axis` (cuts[0] = -inf, cuts[world] = +inf); an answer record is
 * {uint32 query, uint32 global index, double d2} = 16 bytes, d2 < 0 = "not certified by its owner". ---- */
int pct_cloud_upload_aos_dev(pct_cloud *c, const void *d_pts, int64_t n, int64_t stride_bytes);      /* setInput from device memory */
int pct_route_owner_dev(const double *cuts, int world, int axis, int rank, const float *d_q, int64_t Q, uint32_t *d_counts /* [world] */,
                        uint32_t *d_mine_ids, float *d_mine_q, void *stream);
/* partitioned batches (every rank brings its own queries): owner + counts per owner, then the queries grouped by owner with their slots */
int pct_route_owner_all_dev(const double *cuts, int world, int axis, const float *d_q, int64_t Q, uint32_t *d_counts /* [world] */, unsigned char *d_owner, void *stream);
int pct_route_partition_dev(const uint32_t *offsets /* host, [world] */, int world, const unsigned char *d_owner, const float *d_q, int64_t Q,
                            uint32_t *d_cursors /* [world] */, float *d_out_xyz, uint32_t *d_out_slot, void *stream);
int pct_route_certify_dev(int axis, double lo_edge, double hi_edge, const float *d_mine_q, const uint32_t *d_mine_ids, int64_t m,
                          const uint32_t *d_lidx, const double *d_ld2, const uint32_t *d_gid, void *d_answers, void *stream);
int pct_route_scatter_dev(const void *d_answers, int64_t n, uint32_t *d_idx, double *d_d2, uint32_t *d_flag_count, uint32_t *d_flag_ids, void *stream);
int pct_route_gather_queries_dev(const float *d_q, const uint32_t *d_ids, int64_t n, float *d_out, void *stream);
int pct_route_to_global_dev(uint32_t *d_lidx, int64_t n, const uint32_t *d_gid, void *stream);
int pct_route_put_back_dev(const uint32_t *d_ids, int64_t n, const uint32_t *d_idx, const double *d_d2, uint32_t *d_out_idx, double *d_out_d2, void *stream);
/* make sure workspaces for batches up to Q exist (call before capturing a graph) */
int pct_cloud_reserve_queries(pct_cloud *c, int64_t Q);

/* ---- hipGraph-captured fixed-shape batches (config C5: 20 Hz replan) ----------------------- */
/* Captures H2D(queries) -> NN kernels -> D2H(idx,d2) once; pct_plan_run replays it.
 * A plan belongs to its cloud and must be destroyed before it.  The captured kernels hold the cloud's point count, index and
 * workspace pointers; whenever one of them changes (upload / append on a cloud without the ring index, pct_cloud_build_grid /
 * drop_grid, pct_cloud_ring_index, a larger batch that grows the workspaces) the next run captures the graph again by itself
 * (one capture costs a few hundred microseconds).  Appends on a ring-indexed cloud change none of them. */
int pct_plan_create_nn(pct_cloud *c, int algo, int64_t Q, pct_plan **out);
int pct_plan_run(pct_plan *p, const float *q, uint32_t *idx, double *d2);
int pct_plan_destroy(pct_plan *p);

/* The whole query side of one replan tick as ONE captured graph (config C5; the reference's chain is
 * rcvPointCloudCallBack -> checkSafeTrajectory, Planner/src/sim_planning_demo.cpp:159-178, 729-781, next to
 * SafeRegionEvaluate's re-check of the corridor nodes, Planner/src/corridor_finder.cpp:829-835):
 *   arguments in (corridor node centres, trajectory, sample times) -> ONE kernel with a block per planner point: sphere
 *   inflation of every corridor node, getPosFromBezier + inflation of every sample of the committed trajectory, inflation of
 *   every control point (pct_ctrl_points_check) -> first colliding sample / control point -> results out.
 * The cloud needs an index: the rolling-map one (pct_cloud_ring_index; appends then never invalidate the plan) or the
 * cell-sorted one (pct_cloud_build_grid; re-captured after a rebuild).  Capacities are fixed at creation: at most max_nodes
 * corridor nodes, max_samples trajectory samples (further ones are counted in nsamples but not evaluated), max_segments
 * segments (13 control points each).  want_nn != 0: exact nearest neighbour for every point (idx / d2 outputs meaningful);
 * 0: the search may stop once everything unseen is beyond max_radius + search_margin (radii identical, idx / d2 then only
 * say "some point at least this near"). */
typedef struct pct_replan_out {
    double *node_radius; uint32_t *node_idx; double *node_d2;                      /* n_nodes each; any may be NULL */
    double *sample_pos, *sample_radius, *sample_d2; uint32_t *sample_idx;          /* up to max_samples (pos: x3); any may be NULL */
    double *ctrl_pos, *ctrl_radius, *ctrl_d2; uint32_t *ctrl_idx;                  /* nctrl (pos: x3); any may be NULL */
    int64_t nsamples, first_hit_sample;                                            /* filled: as pct_bezier_check */
    int64_t nctrl, first_hit_ctrl;                                                 /* filled: as pct_ctrl_points_check */
} pct_replan_out;
int pct_plan_create_replan(pct_cloud *c, int32_t max_nodes, int32_t max_samples, int32_t max_segments, pct_plan **out);
/* traj may be NULL (corridor nodes only) */
int pct_plan_replan_run(pct_plan *p, const pct_inflate_params *prm, const double *nodes, int64_t n_nodes, const pct_bezier_traj *traj,
                        double t_start, double stop_time, double dt, int want_nn, pct_replan_out *out);
/* host wall time of the last pct_plan_replan_run in microseconds: {argument fill, hipGraphLaunch, wait for results, read-out} */
int pct_plan_last_run_us(pct_plan *p, double us[4]);

/* ---- measurement hooks (bench.py): HIP events recorded on the stream the kernels ran on.
 * pct_last_kernel_ms: the last batch's DOMINANT kernel alone (nn_grid_kernel, nn_tile_filter_kernel
 * or the nn_stream_kernel passes) -- the same quantity rocprofv3 --kernel-trace averages;
 * pct_last_batch_ms: every kernel of the batch (binning, bounds, reduction included). ---------- */
int pct_last_kernel_ms(pct_cloud *c, float *ms);
int pct_last_batch_ms(pct_cloud *c, float *ms);
/* which HIP events the batch entry points record: 0 = none, 1 = around the batch's dominant kernel (default; what
 * pct_last_kernel_ms / pct_kernel_ms_history read), 2 = also around the whole batch (pct_last_batch_ms).  An event pair
 * costs 5-9 us per batch. */
int pct_set_timing(pct_cloud *c, int level);
/* dominant-kernel durations of the most recent batches (up to 64 are kept, oldest first): K batches can be queued back to
 * back without a host sync and every launch's duration read afterwards */
int pct_kernel_ms_history(pct_cloud *c, float *ms, int cap, int *n);
/* Sampling: with stride n > 1 the index path (PCT_ALGO_GRID batches) times only every n-th launch, the first one after this call
 * included -- the kernel's own begin / end timestamps (hipExtLaunchKernel), no marker packets on the stream; timing every launch
 * costs ~4 us of a 160 us step.  pct_kernel_ms_samples: how many durations have been recorded so far (host counter, no sync), so
 * that a caller can tell how many of them fall into a region it brackets. */
int pct_set_timing_stride(pct_cloud *c, int stride);
int pct_kernel_ms_samples(pct_cloud *c, uint64_t *count);
/* algorithmic work of the last batch: points examined (sum over queries), cells examined */
int pct_last_work(pct_cloud *c, uint64_t *points_scanned, uint64_t *cells_scanned);
int pct_set_work_counters(pct_cloud *c, int enabled);
/* diagnostics / tests: which form of the brute-force fp32 filter runs -- -1 automatic (expanded |p|^2 - 2 p.q form while its error
 * band is small against the cloud's point spacing, clouds of 200 000 points or more), 0 always the direct (p - q)^2 form,
 * 1 the expanded form whenever a bounding box exists.  Results are identical; only speed differs. */
int pct_debug_set_filter_mode(int mode);
/* diagnostics: the fp32 upper bounds the streaming filter used for the last batch */
int pct_debug_read_bounds(pct_cloud *c, float *out, int64_t Q);
/* test hook: copies of the cell index as built -- cell_start[ncells + 1] and the cell-ordered records (4 floats per point:
 * x, y, z, bit-cast original index); either pointer may be NULL */
int pct_debug_read_grid(pct_cloud *c, uint32_t *cell_start, float *records);
/* the same index checked on the device, through the caches the query kernels read it through:
 * out = {ids out of range, duplicated ids, records outside their cell's run, decreasing cell_start steps, cell_start[0],
 * cell_start[ncells]} -- a sound index gives {0, 0, 0, 0, 0, n} */
int pct_debug_verify_grid(pct_cloud *c, uint64_t out[6]);
/* work counters of the last instrumented batch: {points scanned, cell runs scanned, pyramid node visits (8 boxes of 32 B each)} */
int pct_last_work_ex(pct_cloud *c, uint64_t out[3]);
/* bounding-box pyramid over the cell index (built for sparsely occupied clouds; PCT_PYRAMID=0/1 never / always):
 * levels = 0 when the cloud has none; empty_fraction = empty cells / cells of the index as built */
int pct_cloud_pyramid_info(const pct_cloud *c, int32_t *levels, int64_t *nodes, double *empty_fraction);

#ifdef __cplusplus
}
#endif
#endif /* PCT_ENGINE_H */
