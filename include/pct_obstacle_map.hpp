// pct_obstacle_map.hpp -- C++ host-side mirror of the planner's obstacle-cloud seam, header only,
// over the C ABI of pct_engine.h.  Method names, argument meaning and return conventions follow
// the reference so that the call sites listed below change by one line each.
//
//   reference (paths relative to /root/reference/Planner)                      this class
//   ---------------------------------------------------------------------     -----------------------------
//   safeRegionRrtStar::setInput(pcl::PointCloud<PointXYZ>)  src/corridor_finder.cpp:93-99    setInput(points, n, stride)
//   safeRegionRrtStar::setParam(safety, search, max_r, range) :17-23; setPt :52-91           setParam(...), setStartPt(...)
//   safeRegionRrtStar::radiusSearch(Vector3d&)              :113-133                          radiusSearch(p)
//   safeRegionRrtStar::checkRadius(Vector3d&)               :656-659                          checkRadius(p)
//   safeRegionRrtStar::checkTrajPtCol(Vector3d&)            :412-416                          checkTrajPtCol(p)
//   loops over checkRadius in SafeRegionEvaluate :829-835 and treeRepair :958-974             checkRadiusBatch(pts, n, out)
//   checkSafeTrajectory(double) + getPosFromBezier          src/sim_planning_demo.cpp:715-781 checkSafeTrajectory(...)
//
// No Eigen/PCL types appear: points are plain double[3] / float records, exactly the bytes
// Eigen::Vector3d and pcl::PointXYZ hold (INTEGRATION.md shows the adapter lines).
// Errors: engine failures throw std::runtime_error carrying pct_last_error(); there is no CPU path.
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "pct_engine.h"

namespace pct {

class ObstacleMap {
public:
    explicit ObstacleMap(int64_t capacity = 1 << 20, int device = 0) : capacity_(capacity)
    {
        check(pct_init(device), "pct_init");
        check(pct_cloud_create(capacity, &cloud_), "pct_cloud_create");
    }
    ~ObstacleMap() { if (cloud_) pct_cloud_destroy(cloud_); }
    ObstacleMap(const ObstacleMap &) = delete;
    ObstacleMap &operator=(const ObstacleMap &) = delete;

    // corridor_finder.cpp:17-23
    void setParam(double safety_margin, double search_margin, double max_radius, double sample_range)
    {
        safety_margin_ = safety_margin;
        prm_.search_margin = search_margin;
        prm_.max_radius = max_radius;
        prm_.sample_range = sample_range;
    }
    // corridor_finder.cpp:43-50 (start_pt) and :87 (setPt overwrites sample_range with local_range)
    void setStartPt(const double start[3]) { for (int i = 0; i < 3; i++) prm_.start[i] = start[i]; }
    void setSampleRange(double local_range) { prm_.sample_range = local_range; }

    // corridor_finder.cpp:93-99.  stride_bytes = 16 for pcl::PointXYZ (cloud.points.data()), 12 for packed xyz.
    // build_index: also build the cell index used by the batched queries (worth it for static clouds).
    void setInput(const void *points, int64_t n, int64_t stride_bytes = 16, bool build_index = true)
    {
        if (n > capacity_) {   // grow: the reference accepts any cloud size.  The larger cloud is created FIRST: if that fails
            pct_cloud *bigger = nullptr;   // (check throws) the map keeps its old, valid cloud
            const int64_t cap = n + n / 2;
            check(pct_cloud_create(cap, &bigger), "pct_cloud_create");
            if (rolling_ && pct_cloud_ring_index(bigger, 0.0f, nullptr) != PCT_OK) {
                pct_cloud_destroy(bigger);
                check(PCT_ERR_HIP, "pct_cloud_ring_index");
            }
            pct_cloud_destroy(cloud_);
            cloud_ = bigger;
            capacity_ = cap;
        }
        check(pct_cloud_upload_aos(cloud_, points, n, stride_bytes), "pct_cloud_upload_aos");
        cloud_empty_ = (n == 0);
        if (build_index && n > 0 && !rolling_) check(pct_cloud_build_grid(cloud_, 0.0f), "pct_cloud_build_grid");
    }
    // rolling map with the in-place index (config C5): call once before the first appendInput; appends then update the index
    // instead of dropping it, and every query below searches it
    void enableRollingIndex(float cell_size = 0.0f, const float *extent = nullptr)
    {
        check(pct_cloud_ring_index(cloud_, cell_size, extent), "pct_cloud_ring_index");
        rolling_ = true;
    }
    // rolling map (config C5): append the newest sensor frame, evicting the oldest points
    void appendInput(const void *points, int64_t n, int64_t stride_bytes = 16)
    {
        check(pct_cloud_append_aos(cloud_, points, n, stride_bytes), "pct_cloud_append_aos");
        cloud_empty_ = pct_cloud_size(cloud_) == 0;
    }

    // corridor_finder.cpp:113-133
    double radiusSearch(const double p[3])
    {
        double r = 0;
        check(pct_inflate_batch(cloud_, &prm_, p, 1, &r, nullptr, nullptr), "pct_inflate_batch");
        return r;
    }
    double checkRadius(const double p[3]) { return radiusSearch(p); }               // :656-659
    bool checkTrajPtCol(const double p[3]) { return radiusSearch(p) < 0.0; }        // :412-416

    // the independent re-checks of SafeRegionEvaluate (:829-835) / treeRepair (:958-974) as ONE launch
    void checkRadiusBatch(const double *pts, int64_t n, double *radius, uint32_t *nn_index = nullptr, double *nn_d2 = nullptr)
    {
        check(pct_inflate_batch(cloud_, &prm_, pts, n, radius, nn_index, nn_d2), "pct_inflate_batch");
    }
    // corridor_finder.cpp:661-669
    int checkNodeUpdate(double new_radius, double old_radius) const
    {
        if (new_radius < safety_margin_) return -1;
        if (new_radius < old_radius) return 0;
        return 1;
    }

    // sim_planning_demo.cpp:729-781.  poly_coeff: segments x row_stride row-major (the MatrixXd the optimiser
    // returns, traj_optimizer.cpp:739-751, is column-major: pass PolyCoeff.transpose().eval().data() or a
    // RowMajor copy); seg_time = _Time; orders = _poly_orderList; t_now = max(0, odom stamp - start time).
    // Returns true on collision like the reference; *first_hit_sample (optional) is the index of that sample.
    bool checkSafeTrajectory(const double *poly_coeff, int64_t row_stride, const double *seg_time, const int32_t *orders,
                             int32_t segment_num, double t_now, double stop_time, int64_t *first_hit_sample = nullptr,
                             int64_t *samples = nullptr)
    {
        pct_bezier_traj tr{ poly_coeff, row_stride, seg_time, orders, segment_num };
        int64_t fh = -1, ns = 0;
        check(pct_bezier_check(cloud_, &tr, &prm_, t_now, stop_time, 0.02, &fh, &ns, 4096, nullptr, nullptr, nullptr, nullptr),
              "pct_bezier_check");
        if (first_hit_sample) *first_hit_sample = fh;
        if (samples) *samples = ns;
        return fh >= 0;
    }

    // SURVEY 3.3 (config C5's build extension): the same threshold test on the raw control points of the committed trajectory
    // (control point j of segment i = poly_coeff[i][d*(n+1)+j] * T_i, the point traj_optimizer.cpp:624-648 keeps inside sphere i)
    bool checkControlPoints(const double *poly_coeff, int64_t row_stride, const double *seg_time, const int32_t *orders,
                            int32_t segment_num, double t_now, int64_t *first_hit_point = nullptr, int64_t *points = nullptr)
    {
        pct_bezier_traj tr{ poly_coeff, row_stride, seg_time, orders, segment_num };
        int64_t fh = -1, nc = 0;
        check(pct_ctrl_points_check(cloud_, &tr, &prm_, t_now, &fh, &nc, 0, nullptr, nullptr, nullptr, nullptr), "pct_ctrl_points_check");
        if (first_hit_point) *first_hit_point = fh;
        if (points) *points = nc;
        return fh >= 0;
    }

    // raw batched NN for other consumers (status_inspector.cpp:33-46 collision test, camera_sensor.cpp:133-145 crop)
    void nearest(const float *queries, int64_t n, uint32_t *index, double *d2)
    {
        check(pct_nn_batch(cloud_, queries, n, index, d2), "pct_nn_batch");
    }
    std::vector<uint32_t> radiusIndices(const float center[3], float radius)
    {
        std::vector<uint32_t> out((size_t)std::max<int64_t>(pct_cloud_size(cloud_), 1));
        int64_t n = 0;
        check(pct_radius_indices(cloud_, center, radius, out.data(), (int64_t)out.size(), &n), "pct_radius_indices");
        out.resize((size_t)n);
        return out;
    }

    // lidar sensor (camera_sensor.cpp:133-145): the points within max_dist of the sensor become this frame's observed map
    void cropTo(const double center[3], double radius, ObstacleMap &observed)
    {
        check(pct_cloud_crop_to(cloud_, center, radius, observed.cloud_), "pct_cloud_crop_to");
        observed.cloud_empty_ = pct_cloud_size(observed.cloud_) == 0;
        if (!observed.cloud_empty_) check(pct_cloud_build_grid(observed.cloud_, 0.0f), "pct_cloud_build_grid");
    }
    // supervisor (status_inspector.cpp:33-46): nearestKSearch(point, 1) and sqrt(d2) < col_rad
    bool collides(const double p[3], double col_rad)
    {
        if (cloud_empty_) return false;
        const float q[3] = { (float)p[0], (float)p[1], (float)p[2] };       // pcl::PointXYZ(cmd.position.x, ...)
        uint32_t idx;
        double d2;
        check(pct_nn_batch(cloud_, q, 1, &idx, &d2), "pct_nn_batch");
        return std::sqrt(d2) < col_rad;
    }

    int64_t size() const { return pct_cloud_size(cloud_); }
    bool empty() const { return cloud_empty_; }
    pct_cloud *handle() { return cloud_; }
    const pct_inflate_params &params() const { return prm_; }

private:
    static void check(int status, const char *what)
    {
        if (status != PCT_OK && status != PCT_ERR_EMPTY)
            throw std::runtime_error(std::string(what) + ": " + pct_last_error());
    }
    pct_cloud *cloud_ = nullptr;
    int64_t capacity_ = 0;
    bool cloud_empty_ = true;
    bool rolling_ = false;
    double safety_margin_ = 0.0;
    pct_inflate_params prm_{ { 0, 0, 0 }, 0.0, 0.0, 0.0 };
};

}  // namespace pct
