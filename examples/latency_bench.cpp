// latency_bench.cpp -- host-observed latency of the planner's single calls, measured from C++ (what the ROS node would see;
// scripts/probe_latency.py measures the same entry points through Python, which adds several microseconds per call):
//   kd_nearestf / kd_nearest_rangef on a 1000-node RRT* tree        (corridor_finder.cpp:428-437, 464)
//   ObstacleMap::radiusSearch on a 10 M-point indexed cloud          (corridor_finder.cpp:113-133)
//   ObstacleMap::checkSafeTrajectory, 99 samples                      (sim_planning_demo.cpp:729-781)
//   the replan plan on a rolling map (64 nodes + 99 samples + 21 control points, ONE captured graph)
// Build: pointcloudtraj_amd/build.py.  Prints p50 / p99 microseconds.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "kdtree/kdtree.h"
#include "pct_obstacle_map.hpp"

static uint64_t sm64(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static float u01(uint64_t &s) { return (float)(sm64(s) >> 40) * 0x1p-24f; }

template <class F>
static void lat(const char *what, F fn, int n = 2000)
{
    for (int i = 0; i < 50; i++) fn();
    std::vector<double> t((size_t)n);
    for (int i = 0; i < n; i++) {
        const auto a = std::chrono::steady_clock::now();
        fn();
        t[(size_t)i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
    }
    std::sort(t.begin(), t.end());
    std::printf("%-78s p50 %7.2f us   p99 %7.2f us\n", what, t[(size_t)n / 2], t[(size_t)(n * 99 / 100)]);
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? std::atoll(argv[1]) : 10000000;
    uint64_t seed = 7;
    std::vector<float> cloud((size_t)3 * N);
    for (auto &v : cloud) v = 100.f * u01(seed);
    pct::ObstacleMap map(N);
    const double start[3] = { 50, 50, 50 };
    map.setParam(0.6, 0.25, 1.5, 1e9);
    map.setStartPt(start);
    map.setInput(cloud.data(), N, 12);

    kdtree *t = kd_create(3);
    static int payload[1000];
    for (int i = 0; i < 1000; i++) { const float p[3] = { 10.f * u01(seed), 10.f * u01(seed), 10.f * u01(seed) }; kd_insertf(t, p, &payload[i]); }
    const float q[3] = { 5, 5, 5 };
    lat("kd_nearestf + kd_res_item_data + kd_res_free, 1000-node tree", [&] { kdres *r = kd_nearestf(t, q); (void)kd_res_item_data(r); kd_res_free(r); });
    lat("kd_nearest_rangef(1.5) + kd_res_free, 1000-node tree", [&] { kdres *r = kd_nearest_rangef(t, q, 1.5f); kd_res_free(r); });
    double p[3] = { 41.3, 52.9, 47.1 };
    lat("ObstacleMap::radiusSearch, indexed cloud", [&] { p[0] += 1e-3; (void)map.radiusSearch(p); });
    std::vector<double> batch(3 * 64), rad(64);
    for (auto &v : batch) v = 20.0 + 60.0 * u01(seed);
    lat("ObstacleMap::checkRadiusBatch, 64 corridor nodes", [&] { map.checkRadiusBatch(batch.data(), 64, rad.data()); });
    const int32_t orders[3] = { 6, 6, 6 };
    const double seg_time[3] = { 1, 1, 1 };
    double coef[3 * 21];
    for (int s = 0; s < 3; s++)
        for (int d = 0; d < 3; d++)
            for (int j = 0; j < 7; j++) coef[s * 21 + d * 7 + j] = 40.0 + 4.0 * (s + j / 6.0) + (d == 1 ? 0.3 : 0.0);
    lat("ObstacleMap::checkSafeTrajectory, 99 samples, indexed cloud", [&] { (void)map.checkSafeTrajectory(coef, 21, seg_time, orders, 3, 0.0, 2.0); });
    lat("ObstacleMap::checkControlPoints, 21 points, indexed cloud", [&] { (void)map.checkControlPoints(coef, 21, seg_time, orders, 3, 0.0); });

    // rolling map + the captured replan batch
    {
        const int64_t W = std::min<int64_t>(N, 5000000), F = 50000;
        pct_cloud *ring = nullptr;
        if (pct_cloud_create(W, &ring) != PCT_OK || pct_cloud_ring_index(ring, 0.0f, nullptr) != PCT_OK) { std::printf("ring cloud: %s\n", pct_last_error()); return 1; }
        for (int64_t off = 0; off + F <= W; off += F) pct_cloud_append_aos(ring, cloud.data() + 3 * off, F, 12);
        pct_plan *plan = nullptr;
        if (pct_plan_create_replan(ring, 64, 128, 3, &plan) != PCT_OK) { std::printf("plan: %s\n", pct_last_error()); return 1; }
        pct_bezier_traj tr{ coef, 21, seg_time, orders, 3 };
        pct_replan_out out{};
        std::vector<double> nr(64);
        out.node_radius = nr.data();
        const pct_inflate_params prm = map.params();
        int64_t k = 0;
        lat("pct_cloud_append_aos, 50,000-point frame into the 5 M-point rolling map", [&] { pct_cloud_append_aos(ring, cloud.data() + 3 * ((k++ * F) % (W - F)), F, 12); }, 300);
        lat("pct_plan_replan_run: 64 nodes + 99 samples + 21 control points, one graph", [&] { pct_plan_replan_run(plan, &prm, batch.data(), 64, &tr, 0.0, 2.0, 0.02, 0, &out); });
        // one whole tick: the append returns once its launches are queued, the graph queues behind the insert kernel
        lat("C5 tick: pct_cloud_append_aos (50,000 points) + pct_plan_replan_run", [&] {
            pct_cloud_append_aos(ring, cloud.data() + 3 * ((k++ * F) % (W - F)), F, 12);
            pct_plan_replan_run(plan, &prm, batch.data(), 64, &tr, 0.0, 2.0, 0.02, 0, &out);
        }, 300);
        pct_plan_destroy(plan);
        pct_cloud_destroy(ring);
    }
    kd_free(t);
    return 0;
}
