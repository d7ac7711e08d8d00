// corridor.cpp -- libpct_corridor.so: C ABI (include/pct_corridor.h) over pct::SafeRegionRrtStar.
// Host-only C++; the GPU is reached through libkdtree.so / libpct_engine.so.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>

#include "pct_corridor.h"
#include "pct_corridor_finder.hpp"

struct pct_corridor { pct::SafeRegionRrtStar *impl; };

namespace {
thread_local char g_err[512] = "";
template <typename F>
int guarded(F &&f)
{
    try { f(); return 0; }
    catch (const std::exception &e) { std::snprintf(g_err, sizeof g_err, "%s", e.what()); return 1; }
    catch (...) { std::snprintf(g_err, sizeof g_err, "unknown exception"); return 1; }
}
pct::Vec3 v3(const double p[3]) { return pct::Vec3(p[0], p[1], p[2]); }
}  // namespace

extern "C" {

const char *pct_corridor_last_error(void) { return g_err; }

int pct_corridor_create(int64_t cloud_capacity, int device, pct_corridor **out)
{
    if (!out) return 2;
    return guarded([&] {
        pct_corridor *c = new pct_corridor{ nullptr };
        try { c->impl = new pct::SafeRegionRrtStar(cloud_capacity, device); }
        catch (...) { delete c; throw; }
        *out = c;
    });
}
void pct_corridor_destroy(pct_corridor *c)
{
    if (!c) return;
    delete c->impl;
    delete c;
}
int pct_corridor_set_param(pct_corridor *c, double safety_margin, double search_margin, double max_radius, double sample_range)
{
    return guarded([&] { c->impl->setParam(safety_margin, search_margin, max_radius, sample_range); });
}
int pct_corridor_reset(pct_corridor *c) { return guarded([&] { c->impl->reset(); }); }
int pct_corridor_set_speculation(pct_corridor *c, int k) { return guarded([&] { c->impl->setSpeculation(k); }); }
int pct_corridor_speculation_stats(pct_corridor *c, uint64_t *hit, uint64_t *miss)
{
    return guarded([&] { if (hit) *hit = c->impl->speculativeHits(); if (miss) *miss = c->impl->speculativeFallbacks(); });
}
int pct_corridor_set_fused_expansion(pct_corridor *c, int on) { return guarded([&] { c->impl->setFusedExpansion(on != 0); }); }
int pct_corridor_expansion_launches(pct_corridor *c, uint64_t *launches)
{
    return guarded([&] { if (launches) *launches = c->impl->expansionLaunches(); });
}
int pct_corridor_repair_batches(pct_corridor *c, uint64_t *batches)
{
    return guarded([&] { if (batches) *batches = c->impl->repairBatches(); });
}
int pct_corridor_set_input(pct_corridor *c, const void *points, int64_t n, int64_t stride_bytes, int build_index)
{
    return guarded([&] { c->impl->setInput(points, n, stride_bytes, build_index != 0); });
}
int pct_corridor_set_pt(pct_corridor *c, const double start[3], const double end[3], double xl, double xh, double yl, double yh,
                        double zl, double zh, double local_range, int max_iter, double sample_portion, double goal_portion)
{
    return guarded([&] { c->impl->setPt(v3(start), v3(end), xl, xh, yl, yh, zl, zh, local_range, max_iter, sample_portion, goal_portion); });
}
int pct_corridor_set_start_pt(pct_corridor *c, const double start[3], const double end[3])
{
    return guarded([&] { c->impl->setStartPt(v3(start), v3(end)); });
}
int pct_corridor_reset_root(pct_corridor *c, const double target[3]) { return guarded([&] { c->impl->resetRoot(v3(target)); }); }
int pct_corridor_expansion(pct_corridor *c, int64_t iterations) { return guarded([&] { c->impl->ExpansionIterations(iterations); }); }
int pct_corridor_refine(pct_corridor *c, int64_t iterations) { return guarded([&] { c->impl->RefineIterations(iterations); }); }
int pct_corridor_evaluate(pct_corridor *c) { return guarded([&] { c->impl->EvaluateOnce(); }); }
// the reference's own signatures: seconds of wall clock (corridor_finder.h:97-99)
int pct_corridor_expansion_timed(pct_corridor *c, double time_limit, int64_t *iterations_done)
{
    return guarded([&] { c->impl->SafeRegionExpansion(time_limit); if (iterations_done) *iterations_done = c->impl->lastIterations(); });
}
int pct_corridor_refine_timed(pct_corridor *c, double time_limit, int64_t *iterations_done)
{
    return guarded([&] { c->impl->SafeRegionRefine(time_limit); if (iterations_done) *iterations_done = c->impl->lastIterations(); });
}
int pct_corridor_evaluate_timed(pct_corridor *c, double time_limit) { return guarded([&] { c->impl->SafeRegionEvaluate(time_limit); }); }
int pct_corridor_check_traj_pt_col(pct_corridor *c, const double p[3], int *collides)
{
    return guarded([&] { *collides = c->impl->checkTrajPtCol(v3(p)) ? 1 : 0; });
}
int pct_corridor_get_path(pct_corridor *c, double *path, double *radius, int64_t cap, int64_t *n_out)
{
    return guarded([&] {
        const auto pr = c->impl->getPath();
        const int64_t k = (int64_t)pr.first.size();
        for (int64_t i = 0; i < std::min(k, cap); i++) {
            path[3 * i] = pr.first[i].x; path[3 * i + 1] = pr.first[i].y; path[3 * i + 2] = pr.first[i].z;
            radius[i] = pr.second[i];
        }
        *n_out = k;
    });
}
int pct_corridor_status(pct_corridor *c, int *path_exists, int *global_navi, int64_t *n_nodes, uint64_t *inflation_queries)
{
    return guarded([&] {
        if (path_exists) *path_exists = c->impl->getPathExistStatus() ? 1 : 0;
        if (global_navi) *global_navi = c->impl->getGlobalNaviStatus() ? 1 : 0;
        if (n_nodes) *n_nodes = (int64_t)c->impl->treeSize();
        if (inflation_queries) *inflation_queries = c->impl->inflationQueries();
    });
}

}  // extern "C"
