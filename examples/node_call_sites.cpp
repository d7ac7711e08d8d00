// node_call_sites.cpp -- the planner node's own statements against the replacement class.
//
// The body of planInitialTraj / planIncrementalTraj / rcvPointCloudCallBack below is Planner/src/sim_planning_demo.cpp:159-178,
// 344-356, 381-416 with `safeRegionRrtStar _rrtPathPlaner` declared as `pct::SafeRegionRrtStar` and nothing else changed in the
// calls that carry the wall-clock budgets: SafeRegionExpansion(_path_find_limit), SafeRegionRefine(_time_limit_1),
// SafeRegionEvaluate(_time_limit_2) take seconds as doubles exactly as corridor_finder.h:97-99 declares them.  What does change
// is where Eigen / PCL types cross the boundary (setInput's cloud, setPt's vectors, getPath's matrices): INTEGRATION.md 3b.
// Runs one planning cycle on a synthetic pillar world and prints what the node would log.  Exit code 0 = a corridor was found and
// kept through a refine / evaluate cycle.
#include <cstdio>
#include <tuple>
#include <vector>

#include "pct_corridor_finder.hpp"

struct PointXYZ { float x, y, z, pad; };              // pcl::PointXYZ's 16-byte layout

// ---- the node's globals (sim_planning_demo.cpp:40-90), same names ----
static double _x_l = -15, _x_h = 15, _y_l = -15, _y_h = 15, _z_l = 0, _z_h = 4;
static double _sensing_range = 30.0, _sample_portion = 0.3, _goal_portion = 0.1, _path_find_limit = 0.05, _plan_rate = 10.0;
static double _safety_margin = 0.6, _search_margin = 0.25, _max_radius = 1.5, _refine_portion = 0.7;
static int _max_samples = 200000;
static double _time_limit_1, _time_limit_2;
static pct::Vec3 _start_pos(-10, -10, 2), _end_pos(9, 9, 2), _commit_target;
static std::vector<pct::Vec3> _Path;
static std::vector<double> _Radius;
static bool _is_traj_exist = false;
static pct::SafeRegionRrtStar _rrtPathPlaner(1 << 20);          // was: safeRegionRrtStar _rrtPathPlaner;  (:89)

static void rcvPointCloudCallBack(const std::vector<PointXYZ> &cloud_input)             // :159-178
{
    if (cloud_input.empty()) return;
    _rrtPathPlaner.setInput(cloud_input.data(), (int64_t)cloud_input.size(), sizeof(PointXYZ));      // :167
}

static bool planInitialTraj()                                                                      // :344-379
{
    _rrtPathPlaner.reset();
    _rrtPathPlaner.setPt(_start_pos, _end_pos, _x_l, _x_h, _y_l, _y_h, _z_l, _z_h, _sensing_range, _max_samples, _sample_portion, _goal_portion);
    _rrtPathPlaner.SafeRegionExpansion(_path_find_limit);                                          // :350, unchanged
    std::tie(_Path, _Radius) = _rrtPathPlaner.getPath();                                           // :354
    if (_rrtPathPlaner.getPathExistStatus() == false) {
        std::printf("[Demo] Can't find a path, mission stall, please reset the target\n");
        return false;
    }
    std::printf("[Demo] initial corridor: %zu spheres after %lld samples in %.3f s\n", _Path.size(), (long long)_rrtPathPlaner.lastIterations(), _path_find_limit);
    _is_traj_exist = true;
    return true;
}

static void planIncrementalTraj()                                                                  // :381-422
{
    if (_rrtPathPlaner.getGlobalNaviStatus() == true) return;
    _rrtPathPlaner.SafeRegionRefine  ( _time_limit_1 ); // add samples to the tree                  :412, unchanged
    _rrtPathPlaner.SafeRegionEvaluate( _time_limit_2 ); // ensure that the path is collision-free   :413, unchanged
    if (_rrtPathPlaner.getPathExistStatus() == true) {
        std::tie(_Path, _Radius) = _rrtPathPlaner.getPath();
        std::printf("[Demo] refined corridor: %zu spheres, %zu tree nodes, refine consumed %lld samples\n", _Path.size(), _rrtPathPlaner.getTree().size(),
                    (long long)_rrtPathPlaner.lastIterations());
    }
}

int main()
{
    // a pillar world like map_generator's: square pillars on a 0.1 lattice, the start and goal neighbourhoods kept free
    std::vector<PointXYZ> cloud;
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0; };
    for (int k = 0; k < 40; k++) {
        const double cx = -13 + 26 * rnd(), cy = -13 + 26 * rnd(), w = 0.4 + 0.8 * rnd(), h = 1 + 6 * rnd();
        if ((cx + 10) * (cx + 10) + (cy + 10) * (cy + 10) < 9 || (cx - 9) * (cx - 9) + (cy - 9) * (cy - 9) < 9) continue;
        for (double z = 0.1; z < h; z += 0.1)
            for (double t = -w; t <= w; t += 0.1) {
                cloud.push_back({ (float)(cx + t), (float)(cy - w), (float)z, 0 }); cloud.push_back({ (float)(cx + t), (float)(cy + w), (float)z, 0 });
                cloud.push_back({ (float)(cx - w), (float)(cy + t), (float)z, 0 }); cloud.push_back({ (float)(cx + w), (float)(cy + t), (float)z, 0 });
            }
    }
    _time_limit_1 = _refine_portion * 1.0 / _plan_rate * 0.1;                                      // :476-477 (scaled down: a demo, not 70 ms)
    _time_limit_2 = (1 - _refine_portion) * 1.0 / _plan_rate * 0.1;
    _rrtPathPlaner.setParam(_safety_margin, _search_margin, _max_radius, _sensing_range);           // :487
    rcvPointCloudCallBack(cloud);
    if (!planInitialTraj()) return 1;
    for (int tick = 0; tick < 3; tick++) { rcvPointCloudCallBack(cloud); planIncrementalTraj(); }
    if (!_rrtPathPlaner.getPathExistStatus()) { std::printf("corridor lost\n"); return 2; }
    const double p[3] = { _Path[0].x, _Path[0].y, _Path[0].z };
    std::printf("checkTrajPtCol(start) = %d\nall checks passed\n", (int)_rrtPathPlaner.checkTrajPtCol(pct::Vec3(p[0], p[1], p[2])));   // :761
    return 0;
}
