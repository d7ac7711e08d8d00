/* pct_voxel.h -- C ABI of the voxel de-duplication stage in front of the obstacle cloud (libpct_engine.so).
 *
 * Replaces, for large clouds, the reference's host containers
 *     voxel_map<Cont>::add_point_cloud / add_point / get_voxel_cloud / to_voxel_cloud   (Planner/src/voxel_map.cpp:22-57)
 *     voxel_value_map::add_point / get_voxel_cloud                                      (Planner/src/voxel_map.cpp:59-76)
 * (declared in Planner/include/pointcloudTraj/voxel_map.h:10-45), which keep a std::set / std::map of integer voxel
 * coordinates and append the centre of every voxel seen for the first time to a point list.
 *
 * Arithmetic (voxel_map.cpp:5-16, 29-31): voxel coordinate = (int) round(coordinate / res) -- the division is fp64 (a float
 * coordinate is widened first), round() is half-away-from-zero; voxel centre = i * res in fp64, narrowed to fp32 when the
 * container holds floats (pcl::PointXYZ, Eigen::Vector3f).  Voxels are numbered in the order they are first seen, which is
 * also voxel_value_map's value for the voxel; the engine reproduces exactly that order for a batch (the first occurrence in
 * INPUT order wins), so the voxel cloud is identical to the sequential container's, element by element.
 *
 * Differences from the reference: voxel coordinates must lie in [-2^20, 2^20) per axis (+-104 km at res = 0.1; the three
 * of them are packed into one 64-bit hash key) -- anything outside is rejected with PCT_ERR_INVALID and nothing is added;
 * at most 2^31 voxels per map.  Status codes and pct_last_error() are those of pct_engine.h; pct_init() must have run.
 */
#ifndef PCT_VOXEL_H
#define PCT_VOXEL_H

#include <stdint.h>

#include "pct_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pct_voxel_map pct_voxel_map;

/* voxel_map(double res) (voxel_map.cpp:18-21).  capacity_hint = voxels expected (storage grows on demand). */
int pct_voxel_map_create(double res, int64_t capacity_hint, pct_voxel_map **out);
int pct_voxel_map_destroy(pct_voxel_map *m);
int pct_voxel_map_clear(pct_voxel_map *m);
int pct_voxel_map_size(const pct_voxel_map *m, int64_t *n_voxels);

/* add_point_cloud (voxel_map.cpp:24-33) for n points in a HOST array: records of `stride_bytes` bytes whose first three
 * fields are x, y, z -- float (is_f64 = 0: pcl::PointXYZ has stride 16, Eigen::Vector3f 12) or double (is_f64 = 1).
 * Optional outputs (NULL to skip): n_new = voxels added; is_new[n] = what add_point (:35-44) would have returned for
 * each point in sequence; voxel_index[n] = what voxel_value_map::add_point (:62-72) would have returned. */
int pct_voxel_map_add(pct_voxel_map *m, const void *pts, int64_t n, int64_t stride_bytes, int is_f64,
                      int64_t *n_new, uint8_t *is_new, int32_t *voxel_index);
/* same with the points already in DEVICE memory (a sensor simulation or a previous stage on the GPU); is_new / voxel_index
 * are device pointers too.  Synchronous with respect to the host (n_new is returned). */
int pct_voxel_map_add_dev(pct_voxel_map *m, const void *d_pts, int64_t n, int64_t stride_bytes, int is_f64,
                          int64_t *n_new, uint8_t *d_is_new, int32_t *d_voxel_index);

/* get_voxel_cloud (voxel_map.cpp:46-49, 74-76): voxels [first, first+count) in first-seen order.
 * f32: centres as floats, `stride_floats` (3 or 4) floats per record; f64: x, y, z doubles; keys: integer coordinates. */
int pct_voxel_map_get_f32(const pct_voxel_map *m, int64_t first, int64_t count, float *out, int64_t stride_floats);
int pct_voxel_map_get_f64(const pct_voxel_map *m, int64_t first, int64_t count, double *out);
int pct_voxel_map_get_keys(const pct_voxel_map *m, int64_t first, int64_t count, int32_t *out_xyz);

/* device view of the float voxel cloud (SoA), valid until the next add/clear/destroy: feed it to
 * pct_cloud_upload_soa_dev() to make the de-duplicated cloud the obstacle cloud without touching the host. */
int pct_voxel_map_soa_dev(const pct_voxel_map *m, const float **d_x, const float **d_y, const float **d_z, int64_t *n);

/* kernel time of the last add (HIP events on the library's stream), for bench/probe code */
int pct_voxel_map_last_ms(const pct_voxel_map *m, float *ms);

#ifdef __cplusplus
}
#endif
#endif
