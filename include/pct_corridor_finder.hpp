// pct_corridor_finder.hpp -- "safe-region RRT*" flight-corridor finder on the MI355X engine.
//
// Host-side mirror of the reference's safeRegionRrtStar (Planner/include/pointcloudTraj/corridor_finder.h:17-150,
// Planner/src/corridor_finder.cpp) with the same public surface (setParam, reset, setInput, setPt, setStartPt,
// resetRoot, SafeRegionExpansion / Refine / Evaluate, checkTrajPtCol, getPath, getTree, getPathExistStatus,
// getGlobalNaviStatus) and the same bookkeeping, but:
//   * the obstacle cloud lives in HBM behind pct::ObstacleMap -- every radiusSearch / checkRadius is the HIP
//     inflation path (pct_inflate_batch); the independent re-checks of SafeRegionEvaluate run as ONE batch per pass;
//   * the RRT* node set is the drop-in kd_* API (libkdtree.so), i.e. also answered on the GPU;
//   * the wall-clock boxes (ros::Time checks at corridor_finder.cpp:721-722, 774-775, 900-901, 950-951) are
//     replaced by iteration counts so that a run is deterministic and comparable;
//   * Eigen is replaced by a 3-double struct; std::default_random_engine / uniform_real_distribution by an own
//     minstd_rand0 + generate_canonical<double,53> (what libstdc++ does), so results do not depend on the host library.
//   * setSpeculation(K): the sampling loop may run K samples ahead on the GPU (three batched launches per K samples)
//     and replay them in order on the host, falling back to the one-by-one path whenever an earlier sample of the batch
//     changed what a later one would have seen -- the accepted nodes, and so the corridor, are identical to K = 1.
// Numeric types follow data_type.h:12-51 exactly (Node::radius, g, f, rel_dis are float; coordinates double).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "kdtree/kdtree.h"
#include "kdtree/kdtree_ext.h"
#include "pct_obstacle_map.hpp"

namespace pct {

struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    Vec3 operator+(const Vec3 &o) const { return { x + o.x, y + o.y, z + o.z }; }
    Vec3 operator-(const Vec3 &o) const { return { x - o.x, y - o.y, z - o.z }; }
    Vec3 operator*(double s) const { return { x * s, y * s, z * s }; }
    Vec3 operator/(double s) const { return { x / s, y / s, z / s }; }
    double norm() const { return std::sqrt(x * x + y * y + z * z); }
    Vec3 normalized() const { const double n2 = x * x + y * y + z * z; return n2 > 0 ? *this / std::sqrt(n2) : *this; }
    Vec3 cross(const Vec3 &o) const { return { y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x }; }
};

// data_type.h:12-51
struct CorridorNode {
    Vec3 coord;
    float radius = 0;
    bool valid = true, best = false, change = false;
    int rel_id = -2;
    float rel_dis = -1.0f;
    CorridorNode *preNode_ptr = nullptr;
    std::vector<CorridorNode *> nxtNode_ptr;
    float g = 0, f = 0;
    int32_t kd_index = -1;          // node number in the kd tree (since its last rebuild): where the node's steer data lives
    CorridorNode() = default;
    CorridorNode(const Vec3 &c, float r, float g_, float f_) : coord(c), radius(r), g(g_), f(f_) {}
};

// std::minstd_rand0 + the libstdc++ recipe of uniform_real_distribution<double> (two draws per double)
class MinStdRand0 {
public:
    explicit MinStdRand0(uint32_t seed = 0) { x_ = seed % 2147483647u; if (x_ == 0) x_ = 1; }
    uint32_t next() { x_ = (uint32_t)(((uint64_t)x_ * 16807ull) % 2147483647ull); return x_; }
    double canonical()
    {
        const double r = 2147483646.0;
        double s = (double)(next() - 1u);
        s += (double)(next() - 1u) * r;
        double ret = s / (r * r);
        if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
        return ret;
    }
    double uniform(double a, double b) { return canonical() * (b - a) + a; }
    uint32_t state() const { return x_; }
    void setState(uint32_t s) { x_ = s; }
private:
    uint32_t x_;
};

class SafeRegionRrtStar {
public:
    using NodePtr = CorridorNode *;
    static constexpr double kInf = 9999999.0;       // data_type.h:6

    explicit SafeRegionRrtStar(int64_t cloud_capacity = 1 << 20, int device = 0) : map_(cloud_capacity, device), eng_(0) {}
    ~SafeRegionRrtStar() { treeDestruct(); delete best_end_owned_; delete root_owned_; }
    SafeRegionRrtStar(const SafeRegionRrtStar &) = delete;
    SafeRegionRrtStar &operator=(const SafeRegionRrtStar &) = delete;

    // corridor_finder.cpp:17-23
    void setParam(double safety_margin, double search_margin, double max_radius, double sample_range)
    {
        safety_margin_ = safety_margin; search_margin_ = search_margin; max_radius_ = max_radius; sample_range_ = sample_range;
        syncMap();
    }
    // :25-41
    void reset()
    {
        treeDestruct();
        NodeList.clear(); EndList.clear(); invalidSet.clear(); PathList.clear();
        delete best_end_owned_; delete root_owned_;
        best_end_ptr = best_end_owned_ = new CorridorNode();
        root_node = root_owned_ = new CorridorNode();
        path_exist_status = true; inform_status = false; global_navi_status = false;
        best_distance = kInf;
    }
    // :43-50
    void setStartPt(const Vec3 &startPt, const Vec3 &endPt)
    {
        start_pt = startPt; end_pt = endPt;
        x_in_lo = start_pt.x - sample_range_; x_in_hi = start_pt.x + sample_range_;
        y_in_lo = start_pt.y - sample_range_; y_in_hi = start_pt.y + sample_range_;
        syncMap();
    }
    // :52-91
    void setPt(const Vec3 &startPt, const Vec3 &endPt, double xl, double xh, double yl, double yh, double zl, double zh,
               double local_range, int max_iter, double sample_portion, double goal_portion)
    {
        start_pt = startPt; end_pt = endPt;
        x_l = xl; x_h = xh; y_l = yl; y_h = yh; z_l = zl; z_h = zh;
        z_lo = z_l + safety_margin_; z_hi = z_h;
        x_in_lo = start_pt.x - sample_range_; x_in_hi = start_pt.x + sample_range_;    // uses the OLD sample_range, as the reference does
        y_in_lo = start_pt.y - sample_range_; y_in_hi = start_pt.y + sample_range_;
        min_distance = std::sqrt(std::pow(start_pt.x - end_pt.x, 2) + std::pow(start_pt.y - end_pt.y, 2) + std::pow(start_pt.z - end_pt.z, 2));
        updateEllipsoid(end_pt, (start_pt + end_pt) / 2.0);
        sample_range_ = local_range;
        max_samples = max_iter;
        inlier_ratio = sample_portion;
        goal_ratio = goal_portion;
        syncMap();
    }
    // :93-99.  pcl::PointXYZ records are 16 bytes; build_index = build the cell index (static clouds)
    void setInput(const void *points, int64_t n, int64_t stride_bytes = 16, bool build_index = true)
    {
        map_.setInput(points, n, stride_bytes, build_index);
    }

    bool checkTrajPtCol(const Vec3 &pt) { return radiusSearch(pt) < 0.0; }                 // :412-416
    std::pair<std::vector<Vec3>, std::vector<double>> getPath() const { return { Path, Radius }; }
    const std::vector<NodePtr> &getTree() const { return NodeList; }
    bool getPathExistStatus() const { return path_exist_status; }
    bool getGlobalNaviStatus() const { return global_navi_status; }
    ObstacleMap &obstacleMap() { return map_; }
    uint64_t inflationQueries() const { return n_inflate_; }
    // samples evaluated per GPU round trip in Expansion/Refine (1 = the reference's one-by-one loop)
    void setSpeculation(int k) { spec_k_ = std::max(1, std::min(k, 256)); }
    uint64_t speculativeHits() const { return n_spec_hit_; }
    uint64_t speculativeFallbacks() const { return n_spec_miss_; }
    uint64_t expansionLaunches() const { return n_launch_; }
    uint64_t repairBatches() const { return n_repair_batches_; }      // GPU round trips of treeRepair: two per pass (was two per neighbour)
    void setFusedExpansion(bool on) { fused_ok_ = on; }          // off = nearest / inflation / range as three batched launches

    // :226-270
    void resetRoot(const Vec3 &target_coord)
    {
        NodePtr lstNode = PathList.front();
        if (getDis(lstNode->coord, target_coord) < lstNode->radius) { global_navi_status = true; return; }
        double cost_reduction = 0;
        commit_root = target_coord;
        std::vector<NodePtr> cutList;
        for (auto n : NodeList) n->best = false;
        bool delete_root = false;
        for (auto n : PathList) {
            if (!delete_root && getDis(n->coord, target_coord) < (n->radius - 0.1)) {
                delete_root = true;
                n->best = true;
                n->preNode_ptr = nullptr;
                cost_reduction = n->g;
                root_node = n;
                continue;
            }
            if (delete_root) { n->best = false; n->valid = false; cutList.push_back(n); }
        }
        solutionUpdate(cost_reduction, target_coord);
        for (auto n : cutList) { invalidSet.push_back(n); clearBranchW(n); }
        removeInvalid();
    }

    // :704-763 -- `iterations` replaces the wall-clock limit (and is capped by max_samples like the reference's loop)
    void SafeRegionExpansion(int64_t iterations)
    {
        kdTree_ = kd_create(3);
        if (!kdTree_) throw std::runtime_error(std::string("kd_create: ") + pct_last_error());
        commit_root = start_pt;
        root_node = new CorridorNode(start_pt, (float)radiusSearch(start_pt), 0.0f, (float)min_distance);
        recordNode(root_node);
        insertKd(root_node);
        const int64_t limit = std::min<int64_t>(iterations, max_samples);
        growMany(limit, false);
        removeInvalid();
        tracePath();
    }
    // :765-815
    void SafeRegionRefine(int64_t iterations)
    {
        growMany(iterations, true);
        removeInvalid();
        tracePath();
    }
    // :817-936 (no time limit: the pass loop ends on its own conditions)
    void SafeRegionEvaluate()
    {
        if (!path_exist_status) return;
        std::vector<std::pair<Vec3, double>> fail_node_list;
        while (true) {
            // The reference calls checkRadius node by node (:835); the radii only depend on the node coordinates and the
            // cloud, so the whole pass is one batched inflation here and the per-node logic below consumes the results.
            std::vector<double> coords;
            std::vector<int> slot(PathList.size(), -1);
            for (size_t i = 0; i < PathList.size(); i++)
                if (PathList[i]->preNode_ptr != nullptr) {
                    slot[i] = (int)(coords.size() / 3);
                    coords.push_back(PathList[i]->coord.x); coords.push_back(PathList[i]->coord.y); coords.push_back(PathList[i]->coord.z);
                }
            std::vector<double> radii(coords.size() / 3);
            if (!radii.empty()) { map_.checkRadiusBatch(coords.data(), (int64_t)radii.size(), radii.data()); n_inflate_ += radii.size(); }

            for (size_t i = 0; i < PathList.size(); i++) {
                NodePtr ptr = PathList[i];
                NodePtr pre_ptr = ptr->preNode_ptr;
                if (pre_ptr == nullptr) continue;
                const double update_radius = radii[slot[i]];
                const int ret = checkNodeUpdate(update_radius, ptr->radius);
                const double old_radius = ptr->radius;
                ptr->radius = (float)update_radius;
                syncNodeAux(ptr);
                if (ret == -1) {
                    ptr->valid = false;
                    invalidSet.push_back(ptr);
                    clearBranchS(ptr);
                    fail_node_list.push_back({ ptr->coord, old_radius });
                } else if (checkNodeRelation(getDis(ptr->coord, pre_ptr->coord), ptr, pre_ptr) != -1) {
                    if (ptr->valid) {
                        ptr->valid = false;
                        invalidSet.push_back(ptr);
                        clearBranchS(ptr);
                        fail_node_list.push_back({ ptr->coord, old_radius });
                    }
                } else {
                    const std::vector<NodePtr> childList = ptr->nxtNode_ptr;
                    for (auto child : childList)
                        if (checkNodeRelation(getDis(ptr->coord, child->coord), ptr, child) != -1 && child->valid) {
                            child->valid = false;
                            invalidSet.push_back(child);
                            clearBranchS(child);
                            fail_node_list.push_back({ child->coord, (double)child->radius });
                        }
                }
            }
            bool isBreak = true;
            for (auto p : PathList) isBreak = isBreak && p->valid;
            if (isBreak) break;

            std::vector<NodePtr> feasibleEndList;
            for (auto e : EndList)
                if (e->valid && checkEnd(e)) feasibleEndList.push_back(e);
            EndList = feasibleEndList;
            if (feasibleEndList.empty()) {
                path_exist_status = false; inform_status = false; best_distance = kInf;
                break;
            }
            best_end_ptr = feasibleEndList[0];
            double best_cost = kInf;
            for (auto n : feasibleEndList) {
                const double cost = n->g + getDis(n->coord, end_pt) + getDis(root_node->coord, commit_root);
                if (cost < best_cost) { best_end_ptr = n; best_cost = cost; best_distance = best_cost; }
            }
            PathList.clear();
            for (NodePtr p = best_end_ptr; p != nullptr; p = p->preNode_ptr) PathList.push_back(p);
        }
        removeInvalid();
        treeRepair(fail_node_list);
        tracePath();
    }

private:
    // ---- geometry helpers (:101-111) ----
    static double getDis(const Vec3 &a, const Vec3 &b)
    {
        return std::sqrt(std::pow(a.x - b.x, 2) + std::pow(a.y - b.y, 2) + std::pow(a.z - b.z, 2));
    }
    void syncMap()
    {
        map_.setParam(safety_margin_, search_margin_, max_radius_, sample_range_);
        const double s[3] = { start_pt.x, start_pt.y, start_pt.z };
        map_.setStartPt(s);
    }
    // kd_* return NULL on a device failure (and kd_nearest* on an empty tree): surface it as an exception the C ABI wrapper
    // (csrc/corridor.cpp guarded()) turns into an error code, instead of dereferencing it
    static kdres *must(kdres *r, const char *what)
    {
        if (!r) throw std::runtime_error(std::string(what) + " returned no result set: " + pct_last_error());
        return r;
    }
    double radiusSearch(const Vec3 &p)                      // :113-133 -> HIP inflation
    {
        const double q[3] = { p.x, p.y, p.z };
        n_inflate_++;
        return map_.radiusSearch(q);
    }
    void updateEllipsoid(const Vec3 &toward, const Vec3 &centre)         // :77-85, :285-295
    {
        translation_inf = centre;
        const Vec3 downward(0, 0, -1);
        const Vec3 xtf = (toward - translation_inf).normalized();
        const Vec3 ytf = xtf.cross(downward).normalized();
        const Vec3 ztf = xtf.cross(ytf);
        rot_c0 = xtf; rot_c1 = ytf; rot_c2 = ztf;
    }
    void solutionUpdate(double cost_reduction, const Vec3 &target)       // :272-296
    {
        for (auto n : NodeList) n->g = (float)((double)n->g - cost_reduction);     // float -= double, as written in the reference
        min_distance = getDis(target, end_pt);
        updateEllipsoid(target, (target + end_pt) / 2.0);
        best_distance -= cost_reduction;
    }
    void updateHeuristicRegion(NodePtr update_end_node)                  // :298-331
    {
        const double update_cost = update_end_node->g + getDis(update_end_node->coord, end_pt) + getDis(root_node->coord, commit_root);
        if (update_cost < best_distance) {
            best_distance = update_cost;
            elli_l = best_distance;
            elli_s = std::sqrt(best_distance * best_distance - min_distance * min_distance);
            if (inform_status) for (auto p : NodeList) p->best = false;
            for (NodePtr p = update_end_node; p != nullptr; p = p->preNode_ptr) p->best = true;
            best_end_ptr = update_end_node;
            sample_epoch_++;          // the sampling ellipsoid changed
        }
    }
    Vec3 genSample()                                                     // :333-383
    {
        const double bias = eng_.uniform(0.0, 1.0);
        if (bias <= goal_ratio) return end_pt;
        Vec3 pt;
        if (!inform_status) {
            if (bias > goal_ratio && bias <= (goal_ratio + inlier_ratio)) {
                pt.x = eng_.uniform(x_in_lo, x_in_hi); pt.y = eng_.uniform(y_in_lo, y_in_hi); pt.z = eng_.uniform(z_lo, z_hi);
            } else {
                pt.x = eng_.uniform(x_l, x_h); pt.y = eng_.uniform(y_l, y_h); pt.z = eng_.uniform(z_lo, z_hi);
            }
        } else {
            const double us = eng_.uniform(0.0, 1.0), vs = eng_.uniform(0.0, 1.0), phis = eng_.uniform(0.0, 2 * M_PI);
            const double as = elli_l / 2.0 * std::cbrt(us), bs = elli_s / 2.0 * std::cbrt(us);
            const double thetas = std::acos(1 - 2 * vs);
            const Vec3 e(as * std::sin(thetas) * std::cos(phis), bs * std::sin(thetas) * std::sin(phis), bs * std::cos(thetas));
            pt = rot_c0 * e.x + rot_c1 * e.y + rot_c2 * e.z + translation_inf;
            pt.x = std::min(std::max(pt.x, x_l), x_h);
            pt.y = std::min(std::max(pt.y, y_l), y_h);
            pt.z = std::min(std::max(pt.z, z_l), z_h);
        }
        return pt;
    }
    NodePtr findNearstVertex(const Vec3 &pt)                             // :428-437
    {
        float pos[3] = { (float)pt.x, (float)pt.y, (float)pt.z };
        kdres *nearest = must(kd_nearestf(kdTree_, pos), "kd_nearestf");
        NodePtr n = (NodePtr)kd_res_item_data(nearest);
        kd_res_free(nearest);
        return n;
    }
    NodePtr genNewNode(const Vec3 &pt_sample, NodePtr nearest)           // :385-410
    {
        const Vec3 center = steer(pt_sample, nearest);
        const double radius_ = radiusSearch(center);
        const double h_dis_ = getDis(center, end_pt);
        return new CorridorNode(center, (float)radius_, (float)kInf, (float)h_dis_);
    }
    bool checkEnd(NodePtr p) const { return getDis(p->coord, end_pt) + 0.1 < p->radius; }       // :418-426
    static int checkNodeRelation(double dis, NodePtr n1, NodePtr n2)                             // :439-454
    {
        if ((dis + n2->radius) == n1->radius) return 1;
        if ((dis + 0.1) < 0.95 * (n1->radius + n2->radius)) return -1;
        return 0;
    }
    int checkNodeUpdate(double new_radius, double old_radius) const                             // :661-669
    {
        if (new_radius < safety_margin_) return -1;
        if (new_radius < old_radius) return 0;
        return 1;
    }
    static bool isSuccessor(NodePtr cur, NodePtr near)                                          // :670-683
    {
        for (NodePtr p = near->preNode_ptr; p != nullptr; p = p->preNode_ptr) if (p == cur) return true;
        return false;
    }
    bool checkValidEnd(NodePtr endPtr) const                                                    // :685-702
    {
        for (NodePtr p = endPtr; p != nullptr; p = p->preNode_ptr) {
            if (!p->valid) return false;
            if (getDis(p->coord, root_node->coord) < p->radius) return true;
        }
        return false;
    }
    void insertKd(NodePtr n)
    {
        float pos[3] = { (float)n->coord.x, (float)n->coord.y, (float)n->coord.z };
        n->kd_index = kdx_size(kdTree_);
        if (kd_insertf(kdTree_, pos, n) != 0) throw std::runtime_error(std::string("kd_insertf: ") + pct_last_error());
        syncNodeAux(n);
    }
    // what the fused expansion kernel's steer step reads for this node: fp64 centre, float radius (widened)
    void syncNodeAux(NodePtr n)
    {
        if (n->kd_index < 0 || n->kd_index >= kdx_size(kdTree_) || kdx_node_data(kdTree_, n->kd_index) != (void *)n) return;
        const double aux[4] = { n->coord.x, n->coord.y, n->coord.z, (double)n->radius };
        kdx_set_node_aux(kdTree_, n->kd_index, aux);
    }
    void recordNode(NodePtr n) { NodeList.push_back(n); }                                       // :569-573

    // one iteration of the Expansion (:719-756) / Refine (:772-808) loop body
    void growOnce(bool refine) { growWithSample(genSample(), refine); }
    void growWithSample(const Vec3 &pt_sample, bool refine)
    {
        NodePtr nearest = findNearstVertex(pt_sample);
        if (nearest == nullptr || !nearest->valid) return;
        NodePtr fresh = genNewNode(pt_sample, nearest);
        finishGrow(fresh, nearest, refine, nullptr);
    }
    // everything after genNewNode; presults != nullptr = the neighbourhood set was prepared by the caller
    void finishGrow(NodePtr fresh, NodePtr nearest, bool refine, kdres *presults)
    {
        if (fresh->coord.z < z_l || fresh->radius < safety_margin_) { if (presults) kd_res_free(presults); discarded_.push_back(fresh); return; }
        treeRewire(fresh, nearest, presults);
        if (!fresh->valid) { discarded_.push_back(fresh); return; }
        if (checkEnd(fresh)) {
            if (!inform_status) { best_end_ptr = fresh; sample_epoch_++; }      // genSample switches to the ellipsoid from now on
            EndList.push_back(fresh);
            if (refine) updateHeuristicRegion(fresh);
            inform_status = true;
        }
        insertKd(fresh);
        recordNode(fresh);
        treePrune(fresh);
        if ((int)invalidSet.size() >= cach_size) removeInvalid();
    }

    static double kdDist2(const Vec3 &node_coord, const float q[3])      // the kd tree sees float-narrowed node positions and queries
    {
        const double dx = (double)(float)node_coord.x - (double)q[0], dy = (double)(float)node_coord.y - (double)q[1],
                     dz = (double)(float)node_coord.z - (double)q[2];
        double s = dx * dx;
        s = s + dy * dy;
        s = s + dz * dz;
        return s;
    }

    void growMany(int64_t iterations, bool refine)
    {
        int64_t done = 0;
        while (done < iterations) {
            if (spec_k_ <= 1 && !fused_ok_) { growOnce(refine); done++; continue; }      // three single queries per iteration
            // (with the fused kernel even K = 1 is one launch per iteration instead of three)
            done += growBatch((int)std::min<int64_t>(std::max(spec_k_, 1), iterations - done), refine);
        }
    }

    // Speculative batch: evaluate K samples against a snapshot of the tree with three batched GPU calls, then replay them
    // in order.  A sample is replayed from the precomputed answers only if nothing an earlier sample of the batch did could
    // have changed them; otherwise it takes the one-by-one path, and if the sampling distribution or the kd tree itself
    // changed (path found / improved, removeInvalid rebuilt the tree) the rest of the batch is discarded and the generator
    // is rewound, so the sequence of samples is exactly the sequential one.  Returns the number of samples consumed.
    int growBatch(int K, bool refine) { return fused_ok_ ? growBatchFused(K, refine) : growBatchStaged(K, refine); }

    // The same speculation with ONE launch per batch: kdx_expand_batch answers nearest node -> steer -> inflation ->
    // neighbourhood candidates of every sample in a single kernel (the three stages of growBatchStaged are dependent, so
    // staged they cost three launch + sync round trips).  When a sample's nearest node turns out to be one added earlier
    // in the batch, the REST of the batch is re-evaluated against the tree as it is now -- one launch again, and the
    // conflicting sample is then first in line and cannot conflict -- instead of walking that sample through three
    // single queries.  Samples, acceptance order and results are exactly the sequential ones.
    int growBatchFused(int K, bool refine)
    {
        const uint64_t epoch0 = sample_epoch_, kd_epoch0 = kd_epoch_;
        std::vector<Vec3> sample((size_t)K);
        std::vector<uint32_t> rng_before((size_t)K + 1);
        for (int i = 0; i < K; i++) { rng_before[i] = eng_.state(); sample[i] = genSample(); }
        rng_before[K] = eng_.state();
        const int cap = 256;
        std::vector<double> sflat((size_t)3 * K);
        std::vector<pct_expand_result> res((size_t)K);
        std::vector<uint32_t> ids((size_t)K * cap);
        int pos = 0;
        while (pos < K) {
            const int32_t n0 = kdx_size(kdTree_);
            const int m = K - pos;
            for (int i = 0; i < m; i++) { sflat[3 * i] = sample[pos + i].x; sflat[3 * i + 1] = sample[pos + i].y; sflat[3 * i + 2] = sample[pos + i].z; }
            if (kdx_expand_batch(kdTree_, map_.handle(), &map_.params(), sflat.data(), m, cap, res.data(), ids.data()) != 0) {
                fused_ok_ = false;                                   // e.g. obstacle cloud without its cell index: staged path from here on
                eng_.setState(rng_before[pos]);
                return pos > 0 ? pos : growBatchStaged(K, refine);
            }
            n_launch_++;
            int i = pos;
            for (; i < K; i++) {
                if (sample_epoch_ != epoch0 || kd_epoch_ != kd_epoch0) {   // what sample i would be, or the tree it would see, changed
                    eng_.setState(rng_before[i]);
                    return i;
                }
                const pct_expand_result &e = res[(size_t)(i - pos)];
                const int32_t n_now = kdx_size(kdTree_);
                int32_t best = e.near_idx;
                if (best < 0) {                                       // the tree was empty at the snapshot
                    if (n_now == 0) continue;                         // findNearstVertex finds nothing: the sample is skipped
                    break;
                }
                const float qf[3] = { (float)sample[i].x, (float)sample[i].y, (float)sample[i].z };
                double best_d2 = kdDist2(((NodePtr)kdx_node_data(kdTree_, best))->coord, qf);
                bool conflict = false;
                for (int32_t j = n0; j < n_now && !conflict; j++)     // a node added during this batch is strictly closer?
                    conflict = kdDist2(((NodePtr)kdx_node_data(kdTree_, j))->coord, qf) < best_d2;
                if (conflict) break;
                NodePtr nearest = (NodePtr)kdx_node_data(kdTree_, best);
                if (!nearest->valid) continue;                        // as the reference: skip the sample
                n_inflate_++;
                if (i > pos) n_spec_hit_++;
                const Vec3 center(e.center[0], e.center[1], e.center[2]);
                NodePtr fresh = new CorridorNode(center, (float)e.radius, (float)kInf, (float)getDis(center, end_pt));
                kdres *pre = nullptr;
                if (!(fresh->coord.z < z_l || fresh->radius < safety_margin_)) {
                    const float cposf[3] = { (float)center.x, (float)center.y, (float)center.z };
                    const float r = fresh->radius * 2.0f;
                    pre = e.count >= 0 ? kdx_range_from_candidates(kdTree_, cposf, r, &ids[(size_t)(i - pos) * cap], e.count, n0)
                                       : kd_nearest_rangef(kdTree_, cposf, r);
                }
                finishGrow(fresh, nearest, refine, pre);
            }
            if (i == K) break;
            if (i == pos) { growWithSample(sample[i], refine); i++; }   // cannot happen (nothing is younger than the snapshot); never loop
            else n_spec_miss_++;
            pos = i;
        }
        return K;
    }

    int growBatchStaged(int K, bool refine)
    {
        const uint64_t epoch0 = sample_epoch_, kd_epoch0 = kd_epoch_;
        const int32_t n0 = kdx_size(kdTree_);
        std::vector<Vec3> sample((size_t)K);
        std::vector<uint32_t> rng_before((size_t)K + 1);
        std::vector<float> posf((size_t)3 * K);
        for (int i = 0; i < K; i++) {
            rng_before[i] = eng_.state();
            sample[i] = genSample();
            posf[3 * i] = (float)sample[i].x; posf[3 * i + 1] = (float)sample[i].y; posf[3 * i + 2] = (float)sample[i].z;
        }
        rng_before[K] = eng_.state();
        // stage A: nearest tree node of every sample
        std::vector<int32_t> near_idx((size_t)K, -1);
        kdx_nearestf_batch(kdTree_, posf.data(), K, near_idx.data());
        // stage B: steer + inflate the candidate centres
        std::vector<Vec3> center((size_t)K);
        std::vector<double> cflat((size_t)3 * K), radius((size_t)K, 0.0);
        for (int i = 0; i < K; i++) {
            NodePtr nearest = near_idx[i] >= 0 ? (NodePtr)kdx_node_data(kdTree_, near_idx[i]) : nullptr;
            center[i] = nearest ? steer(sample[i], nearest) : sample[i];
            cflat[3 * i] = center[i].x; cflat[3 * i + 1] = center[i].y; cflat[3 * i + 2] = center[i].z;
        }
        map_.checkRadiusBatch(cflat.data(), K, radius.data());
        // stage C: neighbourhood candidates for treeRewire (range = 2 * float radius, centre narrowed to float)
        std::vector<float> cposf((size_t)3 * K), range((size_t)K);
        for (int i = 0; i < K; i++) {
            cposf[3 * i] = (float)center[i].x; cposf[3 * i + 1] = (float)center[i].y; cposf[3 * i + 2] = (float)center[i].z;
            range[i] = std::max((float)radius[i], 0.0f) * 2.0f;
        }
        const int cap = 256;
        std::vector<uint32_t> ids((size_t)K * cap);
        std::vector<int32_t> counts((size_t)K, -1);
        kdx_range_candidates_batch(kdTree_, cposf.data(), range.data(), K, ids.data(), cap, counts.data());

        // replay, in order
        for (int i = 0; i < K; i++) {
            if (sample_epoch_ != epoch0 || kd_epoch_ != kd_epoch0) {      // what sample i would be, or the tree it would see, changed
                eng_.setState(rng_before[i]);
                return i;
            }
            // exact nearest = snapshot winner unless a node added during this batch is strictly closer
            int32_t best = near_idx[i];
            if (best < 0) { growWithSample(sample[i], refine); n_spec_miss_++; continue; }
            const float *qf = &posf[3 * i];
            double best_d2 = kdDist2(((NodePtr)kdx_node_data(kdTree_, best))->coord, qf);
            const int32_t n_now = kdx_size(kdTree_);
            for (int32_t j = n0; j < n_now; j++) {
                const double d2 = kdDist2(((NodePtr)kdx_node_data(kdTree_, j))->coord, qf);
                if (d2 < best_d2) { best_d2 = d2; best = j; }
            }
            if (best != near_idx[i]) { growWithSample(sample[i], refine); n_spec_miss_++; continue; }   // centre differs: one-by-one
            NodePtr nearest = (NodePtr)kdx_node_data(kdTree_, best);
            if (!nearest->valid) continue;                                                               // as the reference: skip the sample
            n_inflate_++;
            n_spec_hit_++;
            NodePtr fresh = new CorridorNode(center[i], (float)radius[i], (float)kInf, (float)getDis(center[i], end_pt));
            kdres *pre = nullptr;
            if (!(fresh->coord.z < z_l || fresh->radius < safety_margin_)) {
                const float r = fresh->radius * 2.0f;
                pre = counts[i] >= 0 ? kdx_range_from_candidates(kdTree_, &cposf[3 * i], r, &ids[(size_t)i * cap], counts[i], n0)
                                     : kd_nearest_rangef(kdTree_, &cposf[3 * i], r);
            }
            finishGrow(fresh, nearest, refine, pre);
        }
        return K;
    }

    Vec3 steer(const Vec3 &pt_sample, NodePtr nearest) const             // the first half of genNewNode (:387-404)
    {
        const double dis = getDis(nearest->coord, pt_sample);
        if (dis > nearest->radius) {
            const double steer_dis = nearest->radius / dis;
            return Vec3(nearest->coord.x + (pt_sample.x - nearest->coord.x) * steer_dis,
                        nearest->coord.y + (pt_sample.y - nearest->coord.y) * steer_dis,
                        nearest->coord.z + (pt_sample.z - nearest->coord.z) * steer_dis);
        }
        return pt_sample;
    }

    void clearBranchW(NodePtr node)                                                             // :135-149
    {
        for (auto n : node->nxtNode_ptr) {
            if (n->best) continue;
            if (n->valid) invalidSet.push_back(n);
            n->valid = false;
            clearBranchW(n);
        }
    }
    void clearBranchS(NodePtr node)                                                             // :151-159
    {
        for (auto n : node->nxtNode_ptr) {
            if (n->valid) invalidSet.push_back(n);
            n->valid = false;
            clearBranchS(n);
        }
    }
    void treePrune(NodePtr p)                                                                   // :161-169
    {
        if (p->g + p->f > best_distance) {
            p->valid = false;
            invalidSet.push_back(p);
            clearBranchS(p);
        }
    }
    void removeInvalid()                                                                        // :170-231
    {
        std::vector<NodePtr> keep, ends;
        kd_epoch_++;                  // node numbers of the kd tree change
        kd_clear(kdTree_);
        for (auto n : NodeList)
            if (n->valid) {
                insertKd(n);
                keep.push_back(n);
                if (checkEnd(n)) ends.push_back(n);
            }
        NodeList = keep;
        EndList = ends;
        for (auto n : invalidSet)
            if (n->preNode_ptr != nullptr) {
                n->change = true;
                const std::vector<NodePtr> child = n->preNode_ptr->nxtNode_ptr;
                n->preNode_ptr->nxtNode_ptr.clear();
                for (auto c : child) if (!c->change) n->preNode_ptr->nxtNode_ptr.push_back(c);
            }
        std::vector<NodePtr> deleteList;
        for (auto n : invalidSet) {
            for (auto c : n->nxtNode_ptr) if (c->valid) c->preNode_ptr = nullptr;
            deleteList.push_back(n);
        }
        invalidSet.clear();
        for (auto n : deleteList) delete n;
    }
    void treeRewire(NodePtr newPtr, NodePtr nearestPtr, kdres *presults = nullptr)              // :457-567
    {
        if (!presults) {
            const float range = newPtr->radius * 2.0f;
            float pos[3] = { (float)newPtr->coord.x, (float)newPtr->coord.y, (float)newPtr->coord.z };
            presults = must(kd_nearest_rangef(kdTree_, pos, range), "kd_nearest_rangef");
        }
        std::vector<NodePtr> nearPtrList;
        bool isInvalid = false;
        while (!kd_res_end(presults)) {
            NodePtr nearPtr = (NodePtr)kd_res_item_data(presults);
            const double dis = getDis(nearPtr->coord, newPtr->coord);
            const int res = checkNodeRelation(dis, nearPtr, newPtr);
            nearPtr->rel_id = res;
            nearPtr->rel_dis = (float)dis;
            nearPtrList.push_back(nearPtr);
            if (res == 1) { newPtr->valid = false; isInvalid = true; break; }
            kd_res_next(presults);
        }
        kd_res_free(presults);
        if (isInvalid) {
            for (auto n : nearPtrList) { n->rel_id = -2; n->rel_dis = -1.0f; }
            return;
        }
        double min_cost = nearestPtr->g + getDis(nearestPtr->coord, newPtr->coord);
        newPtr->preNode_ptr = nearestPtr;
        newPtr->g = (float)min_cost;
        nearestPtr->nxtNode_ptr.push_back(newPtr);
        NodePtr lstParentPtr = nearestPtr;
        std::vector<NodePtr> nearVertex;
        for (auto nearPtr : nearPtrList) {
            const int res = nearPtr->rel_id;
            const double dis = nearPtr->rel_dis;
            const double cost = nearPtr->g + dis;
            if (res == -1) {
                if (cost < min_cost) {
                    min_cost = cost;
                    newPtr->preNode_ptr = nearPtr;
                    newPtr->g = (float)min_cost;
                    lstParentPtr->nxtNode_ptr.pop_back();
                    lstParentPtr = nearPtr;
                    lstParentPtr->nxtNode_ptr.push_back(newPtr);
                }
                nearVertex.push_back(nearPtr);
            }
            nearPtr->rel_id = -2;
            nearPtr->rel_dis = -1.0f;
        }
        for (auto nearPtr : nearVertex) {
            if (!nearPtr->valid) continue;
            const double dis = getDis(nearPtr->coord, newPtr->coord);
            const double cost = dis + newPtr->g;
            if (cost < nearPtr->g) {
                if (isSuccessor(nearPtr, newPtr->preNode_ptr)) continue;
                if (nearPtr->preNode_ptr == nullptr) {
                    nearPtr->preNode_ptr = newPtr;
                    nearPtr->g = (float)cost;
                } else {
                    NodePtr lstNearParent = nearPtr->preNode_ptr;
                    nearPtr->preNode_ptr = newPtr;
                    nearPtr->g = (float)cost;
                    nearPtr->change = true;
                    const std::vector<NodePtr> child = lstNearParent->nxtNode_ptr;
                    lstNearParent->nxtNode_ptr.clear();
                    for (auto c : child) if (!c->change) lstNearParent->nxtNode_ptr.push_back(c);
                    nearPtr->change = false;
                }
                newPtr->nxtNode_ptr.push_back(nearPtr);
            }
        }
    }
    void tracePath()                                                                            // :575-643
    {
        std::vector<NodePtr> feasibleEndList;
        for (auto e : EndList)
            if (checkValidEnd(e) && checkEnd(e) && e->valid) feasibleEndList.push_back(e);
        if (feasibleEndList.empty()) {
            path_exist_status = false;
            best_distance = kInf;
            inform_status = false;
            EndList.clear();
            Path = { Vec3(1, 0, 0), Vec3(0, 1, 0), Vec3(0, 0, 1) };          // MatrixXd::Identity(3,3)
            Radius = { 0.0, 0.0, 0.0 };
            return;
        }
        EndList = feasibleEndList;
        best_end_ptr = feasibleEndList[0];
        double best_cost = kInf;
        for (auto n : feasibleEndList) {
            const double cost = n->g + getDis(n->coord, end_pt) + getDis(root_node->coord, commit_root);
            if (cost < best_cost) { best_end_ptr = n; best_cost = cost; best_distance = best_cost; }
        }
        PathList.clear();
        for (NodePtr p = best_end_ptr; p != nullptr; p = p->preNode_ptr) PathList.push_back(p);
        const size_t k = PathList.size();
        Path.assign(k, Vec3());
        Radius.assign(k, 0.0);
        for (size_t i = 0; i < k; i++) { Path[k - 1 - i] = PathList[i]->coord; Radius[k - 1 - i] = PathList[i]->radius; }
        path_exist_status = true;
    }
    // One batch per pass (corridor_finder.cpp:938-1021).  The reference asks, per failed node, one kd_nearest_rangef and then one
    // radiusSearch per valid neighbour, each a launch + a host round trip here.  Neither the node tree nor the obstacle cloud changes
    // inside the loop (nodes are only marked invalid; removeInvalid runs after it), and radiusSearch is a pure function of the
    // neighbour's centre, so: ONE launch finds the neighbourhood candidates of every failed node, ONE launch inflates every node
    // that could be re-checked, and the reference's per-node logic then runs on the host in its own order with those answers.
    void treeRepair(std::vector<std::pair<Vec3, double>> &node_list)                            // :938-1021
    {
        const int K = (int)node_list.size();
        if (K == 0) { removeInvalid(); return; }
        const int32_t n0 = kdx_size(kdTree_);
        const int cap = 256;
        std::vector<float> posf((size_t)3 * K), range((size_t)K);
        for (int i = 0; i < K; i++) {
            const Vec3 &c = node_list[(size_t)i].first;
            posf[3 * (size_t)i] = (float)c.x; posf[3 * (size_t)i + 1] = (float)c.y; posf[3 * (size_t)i + 2] = (float)c.z;
            range[(size_t)i] = (float)node_list[(size_t)i].second * 2.0f;
        }
        std::vector<kdres *> sets((size_t)K, nullptr);
        struct Free { std::vector<kdres *> &v; ~Free() { for (auto r : v) if (r) kd_res_free(r); } } guard{ sets };
        {
            std::vector<uint32_t> ids((size_t)K * cap);
            std::vector<int32_t> counts((size_t)K);
            for (int b0 = 0; b0 < K; b0 += 1024) {             // kdx batches hold at most 1024 queries
                const int m = std::min(1024, K - b0);
                const bool ok = kdx_range_candidates_batch(kdTree_, &posf[3 * (size_t)b0], &range[(size_t)b0], m, &ids[(size_t)b0 * cap], cap, &counts[(size_t)b0]) == 0;
                n_repair_batches_++;
                for (int i = b0; i < b0 + m; i++)
                    sets[(size_t)i] = must(ok && counts[(size_t)i] >= 0 ? kdx_range_from_candidates(kdTree_, &posf[3 * (size_t)i], range[(size_t)i], &ids[(size_t)i * cap], counts[(size_t)i], n0)
                                                                        : kd_nearest_rangef(kdTree_, &posf[3 * (size_t)i], range[(size_t)i]), "kd_nearest_rangef");
            }
        }
        // every node the loop below may re-check: in some failed node's neighbourhood, still valid, not the root nor its child
        std::vector<NodePtr> cand;
        for (int i = 0; i < K; i++) {
            for (kd_res_rewind(sets[(size_t)i]); !kd_res_end(sets[(size_t)i]); kd_res_next(sets[(size_t)i])) {
                NodePtr ptr = (NodePtr)kd_res_item_data(sets[(size_t)i]);
                if (!ptr->valid || ptr->preNode_ptr == root_node || ptr == root_node || ptr->rel_id == -3) continue;
                ptr->rel_id = -3;                              // scratch mark (restored below): listed once
                cand.push_back(ptr);
            }
            kd_res_rewind(sets[(size_t)i]);
        }
        std::vector<double> coords(3 * cand.size()), radii(cand.size());
        for (size_t k = 0; k < cand.size(); k++) {
            cand[k]->rel_id = -2;
            coords[3 * k] = cand[k]->coord.x; coords[3 * k + 1] = cand[k]->coord.y; coords[3 * k + 2] = cand[k]->coord.z;
        }
        if (!cand.empty()) { map_.checkRadiusBatch(coords.data(), (int64_t)cand.size(), radii.data()); n_repair_batches_++; }
        std::unordered_map<NodePtr, double> fresh;
        fresh.reserve(cand.size() * 2);
        for (size_t k = 0; k < cand.size(); k++) fresh.emplace(cand[k], radii[k]);

        for (int i = 0; i < K; i++) {
            kdres *presults = sets[(size_t)i];
            while (!kd_res_end(presults)) {
                NodePtr ptr = (NodePtr)kd_res_item_data(presults);
                kd_res_next(presults);
                if (!ptr->valid) continue;
                NodePtr pre_ptr = ptr->preNode_ptr;
                if (pre_ptr == root_node || ptr == root_node) continue;
                const auto it = fresh.find(ptr);
                const double update_radius = it != fresh.end() ? it->second : radiusSearch(ptr->coord);
                if (it != fresh.end()) n_inflate_++;           // counted where the reference calls radiusSearch
                const int ret = checkNodeUpdate(update_radius, ptr->radius);
                ptr->radius = (float)update_radius;
                syncNodeAux(ptr);
                if (ret == -1) {
                    if (ptr->valid) { ptr->valid = false; invalidSet.push_back(ptr); clearBranchS(ptr); }
                    continue;
                }
                if (pre_ptr == nullptr) continue;      // the reference dereferences a NULL parent here; a parentless non-root node has nothing to re-check
                const double dis = getDis(pre_ptr->coord, ptr->coord);
                if (checkNodeRelation(dis, pre_ptr, ptr) != -1 && pre_ptr->valid) {
                    pre_ptr->valid = false;
                    invalidSet.push_back(pre_ptr);
                    clearBranchS(pre_ptr);
                    continue;
                }
                const std::vector<NodePtr> childList = ptr->nxtNode_ptr;
                for (auto child : childList)
                    if (checkNodeRelation(getDis(ptr->coord, child->coord), ptr, child) != -1 && child->valid) {
                        child->valid = false;
                        invalidSet.push_back(child);
                        clearBranchS(child);
                    }
            }
        }
        removeInvalid();
    }
    void treeDestruct()                                                                         // :645-654
    {
        if (kdTree_) { kd_free(kdTree_); kdTree_ = nullptr; }
        for (auto n : NodeList) delete n;
        NodeList.clear();
        for (auto n : discarded_) delete n;      // the reference leaks rejected nodes; they are reclaimed here
        discarded_.clear();
    }

    ObstacleMap map_;
    kdtree *kdTree_ = nullptr;
    std::vector<NodePtr> NodeList, EndList, PathList, invalidSet, discarded_;
    NodePtr best_end_ptr = nullptr, root_node = nullptr, best_end_owned_ = nullptr, root_owned_ = nullptr;
    Vec3 start_pt, end_pt, commit_root, translation_inf, rot_c0, rot_c1, rot_c2;
    int cach_size = 10;                      // corridor_finder.cpp:8
    int max_samples = 30000;
    double x_l = 0, x_h = 0, y_l = 0, y_h = 0, z_l = 0, z_h = 0, inlier_ratio = 0, goal_ratio = 0;
    double x_in_lo = 0, x_in_hi = 0, y_in_lo = 0, y_in_hi = 0, z_lo = 0, z_hi = 0;
    double safety_margin_ = 0, max_radius_ = 0, search_margin_ = 0, sample_range_ = 0;
    double min_distance = 0, best_distance = kInf, elli_l = 0, elli_s = 0;
    bool inform_status = false, path_exist_status = true, global_navi_status = false;
    std::vector<Vec3> Path;
    std::vector<double> Radius;
    MinStdRand0 eng_;
    uint64_t n_inflate_ = 0, n_spec_hit_ = 0, n_spec_miss_ = 0, sample_epoch_ = 0, kd_epoch_ = 0, n_launch_ = 0, n_repair_batches_ = 0;
    bool fused_ok_ = true;          // one-launch expansion batches (kdx_expand_batch); false = the three-stage form
    int spec_k_ = 64;               // results do not depend on it (tests/test_corridor.py); 1 = the reference's one-by-one loop
};

}  // namespace pct
