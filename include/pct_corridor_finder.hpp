// pct_corridor_finder.hpp -- "safe-region RRT*" flight-corridor finder on the MI355X engine.
//
// Same public surface and the same decisions, sample for sample, as the reference's safeRegionRrtStar
// (Planner/include/pointcloudTraj/corridor_finder.h:17-150, Planner/src/corridor_finder.cpp): setParam, reset, setInput,
// setPt, setStartPt, resetRoot, SafeRegionExpansion / Refine / Evaluate, checkTrajPtCol, getPath, getTree,
// getPathExistStatus, getGlobalNaviStatus.  The machinery underneath is this library's own:
//   * the search tree is an ARENA of spheres addressed by 32-bit ids (detail::SphereArena: centre / radius / cost arrays,
//     parent + intrusive sibling links, a free list) instead of heap nodes holding pointer vectors -- O(1) re-parenting,
//     no per-node scratch fields, nothing to leak or free twice, and the arrays are what a batch hands to the GPU;
//   * the obstacle cloud lives in HBM behind pct::ObstacleMap: every clearance query (the reference's radiusSearch /
//     checkRadius, corridor_finder.cpp:113-133) is the HIP inflation path; SafeRegionEvaluate's re-checks and the repair
//     pass after it run as ONE batch per pass;
//   * the tree's nearest / range queries go through the kd_* drop-in (libkdtree.so), i.e. they are answered on the GPU too;
//   * the sampling loop may run K samples ahead (setSpeculation): one fused launch answers nearest node -> steer ->
//     clearance -> neighbourhood for K samples against a snapshot of the tree, the host replays them in order and falls back
//     whenever an earlier sample of the batch changed what a later one would have seen -- the accepted spheres, and so the
//     corridor, are identical to K = 1;
//   * the wall-clock limits (ros::Time checks at corridor_finder.cpp:721-722, 774-775, 900-901, 950-951) are iteration counts,
//     so a run is deterministic and comparable; Eigen is a 3-double struct; std::default_random_engine +
//     uniform_real_distribution are an own minstd_rand0 + the two-draw generate_canonical recipe of libstdc++.
// Numeric types follow data_type.h:12-51 (radius, g, f are float; coordinates double): that is part of the behaviour.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
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

// what getTree() hands out per sphere (data_type.h:12-51 by value; parent = position of the parent in the same list, -1 = none)
struct CorridorNode {
    Vec3 coord;
    float radius = 0, g = 0, f = 0;
    bool valid = true, best = false;
    int32_t parent = -1;
};

namespace detail {

// The search tree's storage: spheres addressed by id, structure-of-arrays, children as an intrusive doubly linked sibling list
// (the order of siblings never influences a decision of the algorithm, so attaching is a push-front).
class SphereArena {
public:
    static constexpr int32_t kNone = -1;
    std::vector<Vec3> centre;
    std::vector<float> radius, cost, heur;          // data_type.h: radius, g (cost from the root), f (distance to the goal)
    std::vector<int32_t> parent, kd_slot;           // kd_slot: node number in the kd tree since its last rebuild
    std::vector<uint8_t> alive, on_best, condemned;

    int32_t create(const Vec3 &c, float r, float g, float f)
    {
        int32_t id;
        if (!free_.empty()) { id = free_.back(); free_.pop_back(); }
        else {
            id = (int32_t)centre.size();
            centre.emplace_back(); radius.push_back(0); cost.push_back(0); heur.push_back(0);
            parent.push_back(kNone); kd_slot.push_back(-1); alive.push_back(0); on_best.push_back(0); condemned.push_back(0);
            head_.push_back(kNone); next_.push_back(kNone); prev_.push_back(kNone);
        }
        centre[id] = c; radius[id] = r; cost[id] = g; heur[id] = f;
        parent[id] = kNone; kd_slot[id] = -1; alive[id] = 1; on_best[id] = 0; condemned[id] = 0;
        head_[id] = next_[id] = prev_[id] = kNone;
        return id;
    }
    void destroy(int32_t id) { alive[id] = 0; free_.push_back(id); }
    void clear()
    {
        centre.clear(); radius.clear(); cost.clear(); heur.clear(); parent.clear(); kd_slot.clear();
        alive.clear(); on_best.clear(); condemned.clear(); head_.clear(); next_.clear(); prev_.clear(); free_.clear();
    }
    size_t capacity() const { return centre.size(); }

    void attach(int32_t child, int32_t to)
    {
        parent[child] = to;
        prev_[child] = kNone;
        next_[child] = head_[to];
        if (head_[to] != kNone) prev_[head_[to]] = child;
        head_[to] = child;
    }
    // takes `child` out of its parent's sibling list (no-op for a parentless sphere); the parent field becomes kNone
    void detach(int32_t child)
    {
        const int32_t p = parent[child];
        if (p == kNone) return;
        if (prev_[child] != kNone) next_[prev_[child]] = next_[child];
        else if (head_[p] == child) head_[p] = next_[child];
        if (next_[child] != kNone) prev_[next_[child]] = prev_[child];
        prev_[child] = next_[child] = kNone;
        parent[child] = kNone;
    }
    // the parent link alone (the child stays listed under its old parent until that one is swept): resetRoot
    void forget_parent(int32_t child) { detach(child); }
    void children_of(int32_t id, std::vector<int32_t> &out) const
    {
        out.clear();
        for (int32_t c = head_[id]; c != kNone; c = next_[c]) out.push_back(c);
    }
    int32_t first_child(int32_t id) const { return head_[id]; }
    int32_t next_sibling(int32_t id) const { return next_[id]; }
    // is `a` strictly above `below` (reachable from parent(below) upwards)?
    bool above_parent_of(int32_t a, int32_t below) const
    {
        for (int32_t p = parent[below]; p != kNone; p = parent[p]) if (p == a) return true;
        return false;
    }
private:
    std::vector<int32_t> head_, next_, prev_, free_;
};

}  // namespace detail

class SafeRegionRrtStar {
public:
    static constexpr double kInf = 9999999.0;       // data_type.h:6

    explicit SafeRegionRrtStar(int64_t cloud_capacity = 1 << 20, int device = 0) : map_(cloud_capacity, device), rng_(0) {}
    ~SafeRegionRrtStar() { drop_tree(); }
    SafeRegionRrtStar(const SafeRegionRrtStar &) = delete;
    SafeRegionRrtStar &operator=(const SafeRegionRrtStar &) = delete;

    // ---- configuration (corridor_finder.cpp:17-99) ----
    void setParam(double safety_margin, double search_margin, double max_radius, double sample_range)
    {
        safety_margin_ = safety_margin; search_margin_ = search_margin; max_radius_ = max_radius; sample_range_ = sample_range;
        push_params();
    }
    void reset()
    {
        drop_tree();
        goal_ids_.clear(); route_.clear();
        root_ = best_goal_ = kNone;
        path_exists_ = true; informed_ = false; reached_commit_ = false;
        best_cost_ = kInf;
    }
    void setStartPt(const Vec3 &startPt, const Vec3 &endPt)
    {
        start_ = startPt; goal_ = endPt;
        near_x_ = { start_.x - sample_range_, start_.x + sample_range_ };
        near_y_ = { start_.y - sample_range_, start_.y + sample_range_ };
        push_params();
    }
    void setPt(const Vec3 &startPt, const Vec3 &endPt, double xl, double xh, double yl, double yh, double zl, double zh,
               double local_range, int max_iter, double sample_portion, double goal_portion)
    {
        start_ = startPt; goal_ = endPt;
        box_x_ = { xl, xh }; box_y_ = { yl, yh }; box_z_ = { zl, zh };
        sample_z_ = { zl + safety_margin_, zh };
        // the window around the start still uses the sampling range of the PREVIOUS call, as the reference's does (:64-68 before :87)
        near_x_ = { start_.x - sample_range_, start_.x + sample_range_ };
        near_y_ = { start_.y - sample_range_, start_.y + sample_range_ };
        direct_ = std::sqrt(std::pow(start_.x - goal_.x, 2) + std::pow(start_.y - goal_.y, 2) + std::pow(start_.z - goal_.z, 2));
        aim_ellipsoid(goal_, (start_ + goal_) / 2.0);
        sample_range_ = local_range;
        max_samples_ = max_iter;
        near_share_ = sample_portion;
        goal_share_ = goal_portion;
        push_params();
    }
    // pcl::PointXYZ records are 16 bytes; build_index = build the cell index (static clouds)
    void setInput(const void *points, int64_t n, int64_t stride_bytes = 16, bool build_index = true)
    {
        map_.setInput(points, n, stride_bytes, build_index);
    }

    // ---- queries ----
    bool checkTrajPtCol(const Vec3 &pt) { return clearance(pt) < 0.0; }                     // :412-416
    std::pair<std::vector<Vec3>, std::vector<double>> getPath() const { return { path_centres_, path_radii_ }; }
    std::vector<CorridorNode> getTree() const
    {
        std::vector<int32_t> where(T_.capacity(), -1);
        for (size_t i = 0; i < order_.size(); i++) where[(size_t)order_[i]] = (int32_t)i;
        std::vector<CorridorNode> out(order_.size());
        for (size_t i = 0; i < order_.size(); i++) {
            const int32_t id = order_[i];
            out[i].coord = T_.centre[id]; out[i].radius = T_.radius[id]; out[i].g = T_.cost[id]; out[i].f = T_.heur[id];
            out[i].valid = T_.alive[id] != 0; out[i].best = T_.on_best[id] != 0;
            out[i].parent = T_.parent[id] == kNone ? -1 : where[(size_t)T_.parent[id]];
        }
        return out;
    }
    size_t treeSize() const { return order_.size(); }
    bool getPathExistStatus() const { return path_exists_; }
    bool getGlobalNaviStatus() const { return reached_commit_; }
    ObstacleMap &obstacleMap() { return map_; }
    uint64_t inflationQueries() const { return n_clearance_; }
    // samples evaluated per GPU round trip in Expansion / Refine (1 = the reference's one-by-one loop)
    void setSpeculation(int k) { ahead_ = std::max(1, std::min(k, 256)); }
    uint64_t speculativeHits() const { return n_replayed_; }
    uint64_t speculativeFallbacks() const { return n_restarts_; }
    uint64_t expansionLaunches() const { return n_fused_launches_; }
    uint64_t repairBatches() const { return n_repair_trips_; }        // GPU round trips of the repair pass: two per pass
    void setFusedExpansion(bool on) { fused_ = on; }                  // off = nearest / clearance / range as three batched launches

    // ---- corridor_finder.cpp:226-270: the drone has committed to `target`; the sphere of the current route that holds it
    // becomes the root, everything between it and the old root is cut ----
    void resetRoot(const Vec3 &target)
    {
        const int32_t tip = route_.front();
        if (dist(T_.centre[tip], target) < T_.radius[tip]) { reached_commit_ = true; return; }
        committed_ = target;
        for (int32_t id : order_) T_.on_best[id] = 0;
        double spent = 0;
        bool found = false;
        std::vector<int32_t> behind;
        for (int32_t id : route_) {                                   // goal end first
            if (!found) {
                if (dist(T_.centre[id], target) < (T_.radius[id] - 0.1)) {
                    found = true;
                    T_.on_best[id] = 1;
                    T_.forget_parent(id);
                    spent = T_.cost[id];
                    root_ = id;
                }
                continue;
            }
            T_.alive[id] = 0;
            behind.push_back(id);
        }
        rebase_costs(spent, target);
        for (int32_t id : behind) { note_condemned(id); condemn_below(id, /*spare_best=*/true); }
        sweep();
    }

    // ---- the reference's entry points, wall-clock boxed exactly as there (corridor_finder.h:97-99; the planner node calls them
    // with seconds: sim_planning_demo.cpp:350, 412-413).  The clock is read before every iteration (corridor_finder.cpp:721-722,
    // 774-775): inside a speculative batch that is before every REPLAYED sample, and a sample whose turn comes after the deadline
    // is handed back to the generator, so a boxed run is, sample for sample, the iteration-count run of lastIterations(). ----
    void SafeRegionExpansion(double time_limit)
    {
        start_clock(time_limit);
        expand(max_samples_);
        timed_ = false;
    }
    void SafeRegionRefine(double time_limit)
    {
        start_clock(time_limit);
        last_iterations_ = grow(std::numeric_limits<int64_t>::max(), true);
        timed_ = false;
        sweep();
        choose_route();
    }
    void SafeRegionEvaluate(double time_limit)
    {
        start_clock(time_limit);
        evaluate();
        timed_ = false;
    }
    // ---- the same three with iteration counts instead of the clock: deterministic runs (tests, benchmarks) ----
    void ExpansionIterations(int64_t iterations) { timed_ = false; expand(std::min<int64_t>(iterations, max_samples_)); }   // capped by max_samples like :719
    void RefineIterations(int64_t iterations)
    {
        timed_ = false;
        last_iterations_ = grow(iterations, true);
        sweep();
        choose_route();
    }
    void EvaluateOnce() { timed_ = false; evaluate(); }
    // samples consumed by the last Expansion / Refine call (the reference's iter_count)
    int64_t lastIterations() const { return last_iterations_; }

private:
    using Clock = std::chrono::steady_clock;
    void start_clock(double time_limit)
    {
        timed_ = true;
        t_begin_ = Clock::now();
        time_limit_ = time_limit;
    }
    double elapsed() const { return std::chrono::duration<double>(Clock::now() - t_begin_).count(); }
    bool out_of_time() const { return timed_ && elapsed() > time_limit_; }                 // `(now - before).toSec() > time_limit`

    // ---- :704-763 ----
    void expand(int64_t iterations)
    {
        kd_ = kd_create(3);
        if (!kd_) throw std::runtime_error(std::string("kd_create: ") + pct_last_error());
        committed_ = start_;
        root_ = T_.create(start_, (float)clearance(start_), 0.0f, (float)direct_);
        order_.push_back(root_);
        kd_add(root_);
        last_iterations_ = grow(iterations, false);
        sweep();
        choose_route();
    }
    // ---- :817-936: a new cloud has arrived; re-measure the route's spheres and cut what no longer holds.  The reference asks
    // checkRadius sphere by sphere (:835); the radii depend only on the centres and the cloud, so each pass is one batched
    // inflation whose answers the per-sphere logic then consumes in the reference's order. ----
    void evaluate()
    {
        if (!path_exists_) return;
        std::vector<std::pair<Vec3, double>> broken;                  // (centre, radius before the update) of every sphere cut here
        std::vector<double> xyz, fresh;
        std::vector<int32_t> kids;
        for (;;) {
            xyz.clear();
            for (int32_t id : route_)
                if (T_.parent[id] != kNone) { xyz.push_back(T_.centre[id].x); xyz.push_back(T_.centre[id].y); xyz.push_back(T_.centre[id].z); }
            fresh.assign(xyz.size() / 3, 0.0);
            if (!fresh.empty()) { map_.checkRadiusBatch(xyz.data(), (int64_t)fresh.size(), fresh.data()); n_clearance_ += fresh.size(); }
            size_t k = 0;
            for (int32_t id : route_) {
                const int32_t up = T_.parent[id];
                if (up == kNone) continue;
                const double now = fresh[k++], before = T_.radius[id];
                const int verdict = judge_radius(now, before);
                T_.radius[id] = (float)now;
                kd_refresh(id);
                if (verdict < 0) {
                    T_.alive[id] = 0;
                    note_condemned(id);
                    condemn_below(id, false);
                    broken.push_back({ T_.centre[id], before });
                } else if (overlap(dist(T_.centre[id], T_.centre[up]), id, up) != kLinked) {
                    if (T_.alive[id]) {
                        T_.alive[id] = 0;
                        note_condemned(id);
                        condemn_below(id, false);
                        broken.push_back({ T_.centre[id], before });
                    }
                } else {
                    T_.children_of(id, kids);
                    for (int32_t c : kids)
                        if (overlap(dist(T_.centre[id], T_.centre[c]), id, c) != kLinked && T_.alive[c]) {
                            T_.alive[c] = 0;
                            note_condemned(c);
                            condemn_below(c, false);
                            broken.push_back({ T_.centre[c], (double)T_.radius[c] });
                        }
                }
            }
            bool intact = true;
            for (int32_t id : route_) intact = intact && T_.alive[id];
            if (intact) break;
            std::vector<int32_t> usable;
            for (int32_t e : goal_ids_) if (T_.alive[e] && touches_goal(e)) usable.push_back(e);
            goal_ids_ = usable;
            if (usable.empty() || out_of_time()) { path_exists_ = false; informed_ = false; best_cost_ = kInf; break; }      // :900-905
            pick_cheapest(usable);
            route_.clear();
            for (int32_t p = best_goal_; p != kNone; p = T_.parent[p]) route_.push_back(p);
        }
        // what is left of the box goes to the repair pass (:931-935); it looks at the clock before every broken sphere (:950-951)
        const double repair_limit = timed_ ? time_limit_ - elapsed() : 0.0;
        sweep();
        repair_around(broken, repair_limit);
        choose_route();
    }

    static constexpr int32_t kNone = detail::SphereArena::kNone;
    // how two spheres relate (corridor_finder.cpp:439-454): one inside the other / overlapping enough to fly through / neither
    static constexpr int kInside = 1, kLinked = -1, kApart = 0;

    static double dist(const Vec3 &a, const Vec3 &b)                  // :101-105, with pow() as there
    {
        return std::sqrt(std::pow(a.x - b.x, 2) + std::pow(a.y - b.y, 2) + std::pow(a.z - b.z, 2));
    }
    int overlap(double d, int32_t a, int32_t b) const
    {
        if ((d + T_.radius[b]) == T_.radius[a]) return kInside;
        if ((d + 0.1) < 0.95 * (T_.radius[a] + T_.radius[b])) return kLinked;
        return kApart;
    }
    int judge_radius(double now, double before) const                 // :661-669
    {
        if (now < safety_margin_) return -1;
        return now < before ? 0 : 1;
    }
    bool touches_goal(int32_t id) const { return dist(T_.centre[id], goal_) + 0.1 < T_.radius[id]; }        // :418-426
    bool chain_holds(int32_t tip) const                               // :685-702: alive all the way up to a sphere that holds the root
    {
        for (int32_t p = tip; p != kNone; p = T_.parent[p]) {
            if (!T_.alive[p]) return false;
            if (dist(T_.centre[p], root_centre()) < T_.radius[p]) return true;
        }
        return false;
    }
    Vec3 root_centre() const { return root_ == kNone ? Vec3() : T_.centre[root_]; }
    double route_cost(int32_t tip) const { return T_.cost[tip] + dist(T_.centre[tip], goal_) + dist(root_centre(), committed_); }

    void push_params()
    {
        map_.setParam(safety_margin_, search_margin_, max_radius_, sample_range_);
        const double s[3] = { start_.x, start_.y, start_.z };
        map_.setStartPt(s);
    }
    // kd_* return NULL on a device failure (and kd_nearest* on an empty tree): an exception the C ABI wrapper
    // (csrc/corridor.cpp guarded()) turns into an error code, instead of a NULL dereference
    static kdres *must(kdres *r, const char *what)
    {
        if (!r) throw std::runtime_error(std::string(what) + " returned no result set: " + pct_last_error());
        return r;
    }
    static void *tag(int32_t id) { return reinterpret_cast<void *>((intptr_t)id + 1); }
    static int32_t untag(void *p) { return (int32_t)(reinterpret_cast<intptr_t>(p) - 1); }

    double clearance(const Vec3 &p)                                   // radiusSearch (:113-133) on the HIP inflation path
    {
        const double q[3] = { p.x, p.y, p.z };
        n_clearance_++;
        return map_.radiusSearch(q);
    }

    // ---- sampling (:77-85, :272-383) ----
    void aim_ellipsoid(const Vec3 &toward, const Vec3 &centre)
    {
        ell_centre_ = centre;
        const Vec3 down(0, 0, -1);
        ell_u_ = (toward - ell_centre_).normalized();
        ell_v_ = ell_u_.cross(down).normalized();
        ell_w_ = ell_u_.cross(ell_v_);
    }
    void rebase_costs(double spent, const Vec3 &target)               // solutionUpdate
    {
        for (int32_t id : order_) T_.cost[id] = (float)((double)T_.cost[id] - spent);      // float -= double, as written there
        direct_ = dist(target, goal_);
        aim_ellipsoid(target, (target + goal_) / 2.0);
        best_cost_ -= spent;
    }
    void tighten_ellipsoid(int32_t tip)                               // updateHeuristicRegion
    {
        const double c = route_cost(tip);
        if (!(c < best_cost_)) return;
        best_cost_ = c;
        ell_long_ = best_cost_;
        ell_short_ = std::sqrt(best_cost_ * best_cost_ - direct_ * direct_);
        if (informed_) for (int32_t id : order_) T_.on_best[id] = 0;
        for (int32_t p = tip; p != kNone; p = T_.parent[p]) T_.on_best[p] = 1;
        best_goal_ = tip;
        sampler_version_++;
    }
    Vec3 draw()
    {
        const double u = rng_.uniform(0.0, 1.0);
        if (u <= goal_share_) return goal_;
        Vec3 pt;
        if (!informed_) {
            const bool near = u > goal_share_ && u <= (goal_share_ + near_share_);
            const auto &bx = near ? near_x_ : box_x_, &by = near ? near_y_ : box_y_;
            pt.x = rng_.uniform(bx.first, bx.second);
            pt.y = rng_.uniform(by.first, by.second);
            pt.z = rng_.uniform(sample_z_.first, sample_z_.second);
            return pt;
        }
        const double us = rng_.uniform(0.0, 1.0), vs = rng_.uniform(0.0, 1.0), phi = rng_.uniform(0.0, 2 * M_PI);
        const double a = ell_long_ / 2.0 * std::cbrt(us), b = ell_short_ / 2.0 * std::cbrt(us);
        const double theta = std::acos(1 - 2 * vs);
        const Vec3 e(a * std::sin(theta) * std::cos(phi), b * std::sin(theta) * std::sin(phi), b * std::cos(theta));
        pt = ell_u_ * e.x + ell_v_ * e.y + ell_w_ * e.z + ell_centre_;
        pt.x = std::min(std::max(pt.x, box_x_.first), box_x_.second);
        pt.y = std::min(std::max(pt.y, box_y_.first), box_y_.second);
        pt.z = std::min(std::max(pt.z, box_z_.first), box_z_.second);
        return pt;
    }

    // ---- the kd tree over the spheres' centres ----
    void kd_add(int32_t id)
    {
        float pos[3] = { (float)T_.centre[id].x, (float)T_.centre[id].y, (float)T_.centre[id].z };
        T_.kd_slot[id] = kdx_size(kd_);
        if (kd_insertf(kd_, pos, tag(id)) != 0) throw std::runtime_error(std::string("kd_insertf: ") + pct_last_error());
        kd_refresh(id);
    }
    // what the fused expansion kernel's steer step reads for a sphere: fp64 centre, float radius (widened)
    void kd_refresh(int32_t id)
    {
        const int32_t slot = T_.kd_slot[id];
        if (slot < 0 || slot >= kdx_size(kd_) || kdx_node_data(kd_, slot) != tag(id)) return;
        const double aux[4] = { T_.centre[id].x, T_.centre[id].y, T_.centre[id].z, (double)T_.radius[id] };
        kdx_set_node_aux(kd_, slot, aux);
    }
    int32_t nearest_sphere(const Vec3 &pt)                            // :428-437
    {
        float pos[3] = { (float)pt.x, (float)pt.y, (float)pt.z };
        kdres *r = must(kd_nearestf(kd_, pos), "kd_nearestf");
        void *d = kd_res_item_data(r);
        kd_res_free(r);
        return d ? untag(d) : kNone;
    }
    Vec3 steer(const Vec3 &sample, int32_t from) const                // the first half of genNewNode (:387-404)
    {
        const Vec3 &c = T_.centre[from];
        const double d = dist(c, sample);
        if (d > T_.radius[from]) {
            const double t = T_.radius[from] / d;
            return Vec3(c.x + (sample.x - c.x) * t, c.y + (sample.y - c.y) * t, c.z + (sample.z - c.z) * t);
        }
        return sample;
    }
    // the kd tree sees float-narrowed centres and queries
    static double kd_d2(const Vec3 &c, const float q[3])
    {
        const double dx = (double)(float)c.x - (double)q[0], dy = (double)(float)c.y - (double)q[1], dz = (double)(float)c.z - (double)q[2];
        double s = dx * dx;
        s = s + dy * dy;
        s = s + dz * dz;
        return s;
    }

    // ---- invalidation bookkeeping.  The reference pushes a pointer per invalidation and sweeps when ten have piled up (:8, :759);
    // here a sphere is queued once (condemned flag) and the EVENTS are counted, so the sweeps happen at the same moments. ----
    void note_condemned(int32_t id)
    {
        condemn_events_++;
        if (!T_.condemned[id]) { T_.condemned[id] = 1; doomed_.push_back(id); }
    }
    void condemn_below(int32_t top, bool spare_best)                  // clearBranchS (:151-159) / clearBranchW (:135-149)
    {
        walk_.clear();
        walk_.push_back(top);
        while (!walk_.empty()) {
            const int32_t at = walk_.back();
            walk_.pop_back();
            for (int32_t c = T_.first_child(at); c != kNone; c = T_.next_sibling(c)) {
                if (spare_best && T_.on_best[c]) continue;
                if (T_.alive[c]) note_condemned(c);
                T_.alive[c] = 0;
                walk_.push_back(c);
            }
        }
    }
    // removeInvalid (:170-231): rebuild the kd tree from the surviving spheres (in list order: the tree's shape, and with it the
    // order of range results, depends on it), recompute the goal-touching set, unhook and release the condemned ones
    void sweep()
    {
        kd_version_++;
        kd_clear(kd_);
        size_t w = 0;
        goal_ids_.clear();
        for (int32_t id : order_) {
            if (!T_.alive[id]) continue;
            kd_add(id);
            order_[w++] = id;
            if (touches_goal(id)) goal_ids_.push_back(id);
        }
        order_.resize(w);
        for (int32_t id : doomed_) T_.detach(id);
        for (int32_t id : doomed_) {
            for (int32_t c = T_.first_child(id); c != kNone;) {       // whatever still hangs here survives as a parentless sphere
                const int32_t nxt = T_.next_sibling(c);
                T_.detach(c);
                c = nxt;
            }
            T_.destroy(id);
        }
        doomed_.clear();
        condemn_events_ = 0;
    }
    void drop_tree()                                                  // :645-654
    {
        if (kd_) { kd_free(kd_); kd_ = nullptr; }
        T_.clear();
        order_.clear(); doomed_.clear();
        condemn_events_ = 0;
    }

    // ---- accepting a candidate sphere: everything after genNewNode in the loop bodies (:728-756, :781-808) ----
    void accept(const Vec3 &centre, double radius, int32_t nearest, bool refine, kdres *neighbourhood)
    {
        const float r = (float)radius;
        if (centre.z < box_z_.first || r < safety_margin_) { if (neighbourhood) kd_res_free(neighbourhood); return; }
        const int32_t id = T_.create(centre, r, (float)kInf, (float)dist(centre, goal_));
        if (!wire_in(id, nearest, neighbourhood)) { T_.destroy(id); return; }
        if (touches_goal(id)) {
            if (!informed_) { best_goal_ = id; sampler_version_++; }  // draw() switches to the ellipsoid from now on
            goal_ids_.push_back(id);
            if (refine) tighten_ellipsoid(id);
            informed_ = true;
        }
        kd_add(id);
        order_.push_back(id);
        const float through = T_.cost[id] + T_.heur[id];              // float + float, as in treePrune (:161-169)
        if (through > best_cost_) { T_.alive[id] = 0; note_condemned(id); condemn_below(id, false); }
        if (condemn_events_ >= sweep_after_) sweep();
    }
    // treeRewire (:457-567).  Returns false when some neighbour's sphere contains the candidate (it adds nothing).
    bool wire_in(int32_t id, int32_t nearest, kdres *hits)
    {
        if (!hits) {
            const float range = T_.radius[id] * 2.0f;
            float pos[3] = { (float)T_.centre[id].x, (float)T_.centre[id].y, (float)T_.centre[id].z };
            hits = must(kd_nearest_rangef(kd_, pos, range), "kd_nearest_rangef");
        }
        near_.clear();
        bool swallowed = false;
        for (; !kd_res_end(hits); kd_res_next(hits)) {
            const int32_t nb = untag(kd_res_item_data(hits));
            const double d = dist(T_.centre[nb], T_.centre[id]);
            const int rel = overlap(d, nb, id);
            near_.push_back({ nb, rel, (float)d });
            if (rel == kInside) { swallowed = true; break; }
        }
        kd_res_free(hits);
        if (swallowed) return false;
        // cheapest parent among the linked neighbours, the nearest sphere being the one to beat (strict <, first wins)
        int32_t up = nearest;
        double cheapest = T_.cost[nearest] + dist(T_.centre[nearest], T_.centre[id]);
        for (const auto &n : near_)
            if (n.rel == kLinked) {
                const double via = T_.cost[n.id] + (double)n.d;       // the distance went through a float field there (rel_dis)
                if (via < cheapest) { cheapest = via; up = n.id; }
            }
        T_.cost[id] = (float)cheapest;
        T_.attach(id, up);
        // and the other way round: linked neighbours that get cheaper through the new sphere move under it
        for (const auto &n : near_) {
            if (n.rel != kLinked || !T_.alive[n.id]) continue;
            const double via = dist(T_.centre[n.id], T_.centre[id]) + T_.cost[id];
            if (!(via < T_.cost[n.id])) continue;
            if (T_.above_parent_of(n.id, up)) continue;               // isSuccessor (:670-683): it sits above the new sphere's parent
            T_.detach(n.id);
            T_.cost[n.id] = (float)via;
            T_.attach(n.id, id);
        }
        return true;
    }

    // ---- route selection (tracePath, :575-643) ----
    void pick_cheapest(const std::vector<int32_t> &tips)
    {
        best_goal_ = tips[0];
        double lowest = kInf;
        for (int32_t t : tips) {
            const double c = route_cost(t);
            if (c < lowest) { best_goal_ = t; lowest = c; best_cost_ = c; }
        }
    }
    void choose_route()
    {
        std::vector<int32_t> usable;
        for (int32_t e : goal_ids_) if (chain_holds(e) && touches_goal(e) && T_.alive[e]) usable.push_back(e);
        if (usable.empty()) {
            path_exists_ = false; best_cost_ = kInf; informed_ = false;
            goal_ids_.clear();
            path_centres_ = { Vec3(1, 0, 0), Vec3(0, 1, 0), Vec3(0, 0, 1) };                // MatrixXd::Identity(3,3)
            path_radii_ = { 0.0, 0.0, 0.0 };
            return;
        }
        goal_ids_ = usable;
        pick_cheapest(usable);
        route_.clear();
        for (int32_t p = best_goal_; p != kNone; p = T_.parent[p]) route_.push_back(p);
        const size_t k = route_.size();
        path_centres_.assign(k, Vec3());
        path_radii_.assign(k, 0.0);
        for (size_t i = 0; i < k; i++) { path_centres_[k - 1 - i] = T_.centre[route_[i]]; path_radii_[k - 1 - i] = T_.radius[route_[i]]; }
        path_exists_ = true;
    }

    // ---- treeRepair (:938-1021), one batch per pass.  The reference asks, per broken sphere, one kd_nearest_rangef and then one
    // radiusSearch per live neighbour -- each a launch + a host round trip here.  Neither the kd tree nor the cloud changes inside
    // that loop (spheres are only marked; the sweep runs after it) and a clearance is a pure function of the centre, so: ONE launch
    // finds the neighbourhood candidates of every broken sphere, ONE launch measures every sphere the loop could re-check, and the
    // per-sphere logic then runs on the host in the reference's order with those answers. ----
    void repair_around(const std::vector<std::pair<Vec3, double>> &broken, double repair_limit)
    {
        const int K = (int)broken.size();
        if (K == 0) { sweep(); return; }
        if (timed_ && repair_limit < 0.0) { sweep(); return; }       // the box was spent before the repair began (:950-951 at i = 0): no launch at all
        const Clock::time_point repair_begin = Clock::now();
        const int32_t n0 = kdx_size(kd_);
        const int cap = 256;
        std::vector<float> posf((size_t)3 * K), range((size_t)K);
        for (int i = 0; i < K; i++) {
            const Vec3 &c = broken[(size_t)i].first;
            posf[3 * (size_t)i] = (float)c.x; posf[3 * (size_t)i + 1] = (float)c.y; posf[3 * (size_t)i + 2] = (float)c.z;
            range[(size_t)i] = (float)broken[(size_t)i].second * 2.0f;
        }
        std::vector<kdres *> sets((size_t)K, nullptr);
        struct Release { std::vector<kdres *> &v; ~Release() { for (auto r : v) if (r) kd_res_free(r); } } release{ sets };
        {
            std::vector<uint32_t> ids((size_t)K * cap);
            std::vector<int32_t> counts((size_t)K);
            for (int b0 = 0; b0 < K; b0 += 1024) {                    // kdx batches hold at most 1024 queries
                const int m = std::min(1024, K - b0);
                const bool ok = kdx_range_candidates_batch(kd_, &posf[3 * (size_t)b0], &range[(size_t)b0], m, &ids[(size_t)b0 * cap], cap, &counts[(size_t)b0]) == 0;
                n_repair_trips_++;
                for (int i = b0; i < b0 + m; i++)
                    sets[(size_t)i] = must(ok && counts[(size_t)i] >= 0 ? kdx_range_from_candidates(kd_, &posf[3 * (size_t)i], range[(size_t)i], &ids[(size_t)i * cap], counts[(size_t)i], n0)
                                                                        : kd_nearest_rangef(kd_, &posf[3 * (size_t)i], range[(size_t)i]), "kd_nearest_rangef");
            }
        }
        // every sphere the loop below may re-measure: in some broken sphere's neighbourhood, alive, neither the root nor a child of it
        std::vector<int32_t> pending;
        std::vector<int32_t> slot_of(T_.capacity(), -1);
        for (int i = 0; i < K; i++) {
            kdres *s = sets[(size_t)i];
            for (kd_res_rewind(s); !kd_res_end(s); kd_res_next(s)) {
                const int32_t id = untag(kd_res_item_data(s));
                if (!T_.alive[id] || T_.parent[id] == root_ || id == root_ || slot_of[(size_t)id] >= 0) continue;
                slot_of[(size_t)id] = (int32_t)pending.size();
                pending.push_back(id);
            }
            kd_res_rewind(s);
        }
        std::vector<double> xyz(3 * pending.size()), measured(pending.size());
        for (size_t k = 0; k < pending.size(); k++) {
            const Vec3 &c = T_.centre[pending[k]];
            xyz[3 * k] = c.x; xyz[3 * k + 1] = c.y; xyz[3 * k + 2] = c.z;
        }
        if (!pending.empty()) { map_.checkRadiusBatch(xyz.data(), (int64_t)pending.size(), measured.data()); n_repair_trips_++; }

        std::vector<int32_t> kids;
        for (int i = 0; i < K; i++) {
            if (timed_ && std::chrono::duration<double>(Clock::now() - repair_begin).count() > repair_limit) break;        // :950-951
            kdres *s = sets[(size_t)i];
            while (!kd_res_end(s)) {
                const int32_t id = untag(kd_res_item_data(s));
                kd_res_next(s);
                if (!T_.alive[id]) continue;
                const int32_t up = T_.parent[id];
                if (up == root_ || id == root_) continue;
                double now;
                if (slot_of[(size_t)id] >= 0) { now = measured[(size_t)slot_of[(size_t)id]]; n_clearance_++; }   // counted where the reference asks
                else now = clearance(T_.centre[id]);
                const int verdict = judge_radius(now, T_.radius[id]);
                T_.radius[id] = (float)now;
                kd_refresh(id);
                if (verdict < 0) {
                    if (T_.alive[id]) { T_.alive[id] = 0; note_condemned(id); condemn_below(id, false); }
                    continue;
                }
                if (up == kNone) continue;     // the reference dereferences a NULL parent here; a parentless non-root sphere has nothing to re-check
                if (overlap(dist(T_.centre[up], T_.centre[id]), up, id) != kLinked && T_.alive[up]) {
                    T_.alive[up] = 0;
                    note_condemned(up);
                    condemn_below(up, false);
                    continue;
                }
                T_.children_of(id, kids);
                for (int32_t c : kids)
                    if (overlap(dist(T_.centre[id], T_.centre[c]), id, c) != kLinked && T_.alive[c]) {
                        T_.alive[c] = 0;
                        note_condemned(c);
                        condemn_below(c, false);
                    }
            }
        }
        sweep();
    }

    // ---- the sampling loops ----
    int64_t grow(int64_t iterations, bool refine)
    {
        int64_t done = 0;
        while (done < iterations && !out_of_time()) {
            if (ahead_ <= 1 && !fused_) { grow_one(draw(), refine); done++; continue; }      // three single queries per iteration
            // (with the fused kernel even K = 1 is one launch per iteration instead of three)
            const int K = (int)std::min<int64_t>(std::max(ahead_, 1), iterations - done);
            done += fused_ ? grow_fused(K, refine) : grow_staged(K, refine);
        }
        return done;
    }
    void grow_one(const Vec3 &sample, bool refine)                     // one iteration of :719-756 / :772-808
    {
        const int32_t nearest = nearest_sphere(sample);
        if (nearest == kNone || !T_.alive[nearest]) return;
        const Vec3 c = steer(sample, nearest);
        accept(c, clearance(c), nearest, refine, nullptr);
    }
    struct Lookahead {
        std::vector<Vec3> sample;
        std::vector<uint32_t> rng_before;       // generator state before sample i was drawn (entry K: after the last)
        void draw_all(SafeRegionRrtStar &f, int K)
        {
            sample.resize((size_t)K);
            rng_before.resize((size_t)K + 1);
            for (int i = 0; i < K; i++) { rng_before[(size_t)i] = f.rng_.state(); sample[(size_t)i] = f.draw(); }
            rng_before[(size_t)K] = f.rng_.state();
        }
    };
    // the prepared neighbourhood of an accepted-looking candidate: from the batch's candidate ids when they were complete
    kdres *neighbourhood_from(const Vec3 &centre, double radius, const uint32_t *ids, int32_t count, int32_t n0)
    {
        const float r = (float)radius;
        if (centre.z < box_z_.first || r < safety_margin_) return nullptr;          // accept() will reject it before looking
        const float cposf[3] = { (float)centre.x, (float)centre.y, (float)centre.z };
        return count >= 0 ? kdx_range_from_candidates(kd_, cposf, r * 2.0f, ids, count, n0) : kd_nearest_rangef(kd_, cposf, r * 2.0f);
    }

    // Speculative batch with ONE launch: kdx_expand_batch answers nearest sphere -> steer -> clearance -> neighbourhood candidates
    // of every sample in a single kernel against the tree as it is now.  The host replays the samples in order; a sample whose
    // nearest sphere turns out to be one added earlier in the batch stops the replay, and the REST of the batch is re-evaluated
    // against the tree as it is then (one launch again; the conflicting sample is first in line and cannot conflict).  If the
    // sampling distribution or the kd tree itself changed (a route was found / improved, a sweep rebuilt the tree) the remaining
    // samples are dropped and the generator is rewound, so the sequence of samples is exactly the sequential one.  Returns the
    // number of samples consumed.
    int grow_fused(int K, bool refine)
    {
        const uint64_t sampler0 = sampler_version_, kd0 = kd_version_;
        Lookahead la;
        la.draw_all(*this, K);
        const int cap = 256;
        std::vector<double> flat((size_t)3 * K);
        std::vector<pct_expand_result> res((size_t)K);
        std::vector<uint32_t> ids((size_t)K * cap);
        int pos = 0;
        while (pos < K) {
            const int32_t n0 = kdx_size(kd_);
            const int m = K - pos;
            for (int i = 0; i < m; i++) { const Vec3 &s = la.sample[(size_t)(pos + i)]; flat[3 * (size_t)i] = s.x; flat[3 * (size_t)i + 1] = s.y; flat[3 * (size_t)i + 2] = s.z; }
            if (kdx_expand_batch(kd_, map_.handle(), &map_.params(), flat.data(), m, cap, res.data(), ids.data()) != 0) {
                fused_ = false;                                       // e.g. obstacle cloud without its cell index: staged path from here on
                rng_.setState(la.rng_before[(size_t)pos]);
                return pos > 0 ? pos : grow_staged(K, refine);
            }
            n_fused_launches_++;
            int i = pos;
            for (; i < K; i++) {
                if (sampler_version_ != sampler0 || kd_version_ != kd0 || out_of_time()) { rng_.setState(la.rng_before[(size_t)i]); return i; }
                const pct_expand_result &e = res[(size_t)(i - pos)];
                const int32_t n_now = kdx_size(kd_);
                const int32_t slot = e.near_idx;
                if (slot < 0) {                                       // the tree was empty at the snapshot
                    if (n_now == 0) continue;                         // nothing to be nearest to: the sample is skipped
                    break;
                }
                const float qf[3] = { (float)la.sample[(size_t)i].x, (float)la.sample[(size_t)i].y, (float)la.sample[(size_t)i].z };
                const int32_t nearest = untag(kdx_node_data(kd_, slot));
                const double d2 = kd_d2(T_.centre[nearest], qf);
                bool overtaken = false;                               // a sphere added during this batch is strictly closer?
                for (int32_t j = n0; j < n_now && !overtaken; j++) overtaken = kd_d2(T_.centre[untag(kdx_node_data(kd_, j))], qf) < d2;
                // (answering just this sample on its own and replaying on -- as grow_staged does -- was measured on config C1: 9 launches
                // instead of 43, but 333 single clearance launches for the samples a stale snapshot keeps losing to the spheres the batch
                // itself adds: 7.7 ms against 2.4)
                if (overtaken) break;
                if (!T_.alive[nearest]) continue;                     // as the reference: skip the sample
                n_clearance_++;
                if (i > pos) n_replayed_++;
                const Vec3 c(e.center[0], e.center[1], e.center[2]);
                accept(c, e.radius, nearest, refine, neighbourhood_from(c, e.radius, &ids[(size_t)(i - pos) * cap], e.count, n0));
            }
            if (i == K) break;
            if (i == pos) { grow_one(la.sample[(size_t)i], refine); i++; }      // cannot happen (nothing is younger than the snapshot); never loop
            else n_restarts_++;
            pos = i;
        }
        return K;
    }
    // the same speculation with three batched launches (nearest, clearance, neighbourhood) and a one-by-one fallback per sample
    int grow_staged(int K, bool refine)
    {
        const uint64_t sampler0 = sampler_version_, kd0 = kd_version_;
        const int32_t n0 = kdx_size(kd_);
        Lookahead la;
        la.draw_all(*this, K);
        std::vector<float> posf((size_t)3 * K);
        for (int i = 0; i < K; i++) { const Vec3 &s = la.sample[(size_t)i]; posf[3 * (size_t)i] = (float)s.x; posf[3 * (size_t)i + 1] = (float)s.y; posf[3 * (size_t)i + 2] = (float)s.z; }
        std::vector<int32_t> slot((size_t)K, -1);
        kdx_nearestf_batch(kd_, posf.data(), K, slot.data());
        std::vector<Vec3> centre((size_t)K);
        std::vector<double> flat((size_t)3 * K), radius((size_t)K, 0.0);
        for (int i = 0; i < K; i++) {
            centre[(size_t)i] = slot[(size_t)i] >= 0 ? steer(la.sample[(size_t)i], untag(kdx_node_data(kd_, slot[(size_t)i]))) : la.sample[(size_t)i];
            flat[3 * (size_t)i] = centre[(size_t)i].x; flat[3 * (size_t)i + 1] = centre[(size_t)i].y; flat[3 * (size_t)i + 2] = centre[(size_t)i].z;
        }
        map_.checkRadiusBatch(flat.data(), K, radius.data());
        std::vector<float> cposf((size_t)3 * K), range((size_t)K);
        for (int i = 0; i < K; i++) {
            cposf[3 * (size_t)i] = (float)centre[(size_t)i].x; cposf[3 * (size_t)i + 1] = (float)centre[(size_t)i].y; cposf[3 * (size_t)i + 2] = (float)centre[(size_t)i].z;
            range[(size_t)i] = std::max((float)radius[(size_t)i], 0.0f) * 2.0f;
        }
        const int cap = 256;
        std::vector<uint32_t> ids((size_t)K * cap);
        std::vector<int32_t> counts((size_t)K, -1);
        kdx_range_candidates_batch(kd_, cposf.data(), range.data(), K, ids.data(), cap, counts.data());
        for (int i = 0; i < K; i++) {
            if (sampler_version_ != sampler0 || kd_version_ != kd0 || out_of_time()) { rng_.setState(la.rng_before[(size_t)i]); return i; }
            if (slot[(size_t)i] < 0) { grow_one(la.sample[(size_t)i], refine); n_restarts_++; continue; }
            // the snapshot's winner stands unless a sphere added during this batch is strictly closer
            const float *qf = &posf[3 * (size_t)i];
            int32_t best = slot[(size_t)i];
            double d2 = kd_d2(T_.centre[untag(kdx_node_data(kd_, best))], qf);
            const int32_t n_now = kdx_size(kd_);
            for (int32_t j = n0; j < n_now; j++) {
                const double dj = kd_d2(T_.centre[untag(kdx_node_data(kd_, j))], qf);
                if (dj < d2) { d2 = dj; best = j; }
            }
            if (best != slot[(size_t)i]) { grow_one(la.sample[(size_t)i], refine); n_restarts_++; continue; }    // the centre would differ
            const int32_t nearest = untag(kdx_node_data(kd_, best));
            if (!T_.alive[nearest]) continue;
            n_clearance_++;
            n_replayed_++;
            accept(centre[(size_t)i], radius[(size_t)i], nearest, refine,
                   neighbourhood_from(centre[(size_t)i], radius[(size_t)i], &ids[(size_t)i * cap], counts[(size_t)i], n0));
        }
        return K;
    }

    ObstacleMap map_;
    kdtree *kd_ = nullptr;
    detail::SphereArena T_;
    std::vector<int32_t> order_;             // live spheres in insertion order (the reference's NodeList): kd rebuilds follow it
    std::vector<int32_t> goal_ids_;          // spheres that touch the goal (EndList), in discovery order: ties go to the first
    std::vector<int32_t> route_;             // the chosen route, goal end first (PathList)
    std::vector<int32_t> doomed_, walk_;
    struct NearRec { int32_t id; int rel; float d; };
    std::vector<NearRec> near_;
    int condemn_events_ = 0;
    static constexpr int sweep_after_ = 10;  // cach_size, corridor_finder.cpp:8
    int32_t root_ = kNone, best_goal_ = kNone;
    Vec3 start_, goal_, committed_, ell_centre_, ell_u_, ell_v_, ell_w_;
    std::pair<double, double> box_x_{ 0, 0 }, box_y_{ 0, 0 }, box_z_{ 0, 0 }, near_x_{ 0, 0 }, near_y_{ 0, 0 }, sample_z_{ 0, 0 };
    int max_samples_ = 30000;
    double near_share_ = 0, goal_share_ = 0;
    double safety_margin_ = 0, max_radius_ = 0, search_margin_ = 0, sample_range_ = 0;
    double direct_ = 0, best_cost_ = kInf, ell_long_ = 0, ell_short_ = 0;
    bool informed_ = false, path_exists_ = true, reached_commit_ = false;
    std::vector<Vec3> path_centres_;
    std::vector<double> path_radii_;
    MinStdRand0 rng_;
    uint64_t n_clearance_ = 0, n_replayed_ = 0, n_restarts_ = 0, sampler_version_ = 0, kd_version_ = 0, n_fused_launches_ = 0, n_repair_trips_ = 0;
    bool timed_ = false;            // the current call is wall-clock boxed
    Clock::time_point t_begin_{};
    double time_limit_ = 0.0;
    int64_t last_iterations_ = 0;
    bool fused_ = true;             // one-launch expansion batches (kdx_expand_batch); false = the three-stage form
    int ahead_ = 256;               // results do not depend on it (tests/test_corridor.py); 1 = the reference's one-by-one loop.
                                    // Same box, C1 scenario: K = 16 / 64 / 256 -> 5.9 / 3.5 / 3.1 ms per replan (profiles/r02_corridor_probe.txt)
};

}  // namespace pct
