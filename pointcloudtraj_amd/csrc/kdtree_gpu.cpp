// kdtree_gpu.cpp -- libkdtree.so: the reference's kd_* C API (26 functions)
// (include/kdtree/kdtree.h; reference Utils/kdtree/include/kdtree/kdtree.h:39-122) served by the
// MI355X engine.
//
// What lives where:
//   device : the points, as an fp32 SoA cloud in HBM (pct_cloud).  kd_nearest* and
//            kd_nearest_range* are answered by the HIP streaming kernels -- every distance is
//            computed on the GPU, in the reference's fp64 arithmetic.
//   host   : payload pointers, fp64 copies of the coordinates handed back by kd_res_item*, and
//            the insertion topology (child links + split axis per node, kdtree.c:167-194).
//            The topology is NOT used to search.  It exists because the reference's result
//            ORDER for range queries is observable (corridor_finder.cpp:464-488 breaks on the
//            first containing neighbour) and is a property of the insertion-ordered tree:
//            hits come out in reverse pre-order of a near-child-first walk (kdtree.c:262-293,
//            810-828), and a hit behind a split plane with fabs(dx) == range is dropped
//            (:283).  The host replays exactly that filter and order over the GPU's hit list.
//
// Exact distance ties in kd_nearest* resolve to the node the reference's own walk returns (reference_tie_winner below;
// the batched kdx_* extensions and the obstacle-cloud engine keep "lowest index").
// Any k and any double: a 3-D tree whose coordinates are all representable in fp32 (always true for the *f entry points, the only
// ones the planner uses) is the fast case above; a tree of another dimension, or one that was handed a double fp32 cannot hold,
// keeps its rows as fp64 columns in a pct_nodeset (csrc/nodeset.hip: exhaustive fp64 kernels, the same sums in the same order) and
// is answered from there -- same tie rule, same range order.  The batched kdx_* extensions serve the fast case only.
// Deliberate differences (documented in include/kdtree/kdtree.h): 1 <= k <= 1024; no host fallback.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "kdtree/kdtree.h"
#include "kdtree/kdtree_ext.h"
#include "pct_engine.h"

namespace {
constexpr int32_t NIL = -1;
constexpr int kMaxDim = 1024;

void three_only(const char *what)
{
    std::fprintf(stderr, "[libkdtree/pct] %s: the x,y,z forms serve trees of at most 3 dimensions\n", what);
}

void complain(const char *what)
{
    std::fprintf(stderr, "[libkdtree/pct] %s: %s\n", what, pct_last_error());
}
}  // namespace

struct kdtree {
    int dim = 3;
    bool generic = false;         // dim != 3, or some coordinate is not an fp32 value: the device copy is a pct_nodeset of fp64 columns
    std::vector<double> pos;      // dim per node, insertion order (fp64 as the reference stores them)
    std::vector<float> posf;      // 3-D fp32 trees: the same, narrowed (exact) -- what is mirrored to HBM
    pct_nodeset *nodes = nullptr; // generic trees
    int64_t nsynced = 0;          // rows [0, nsynced) are in the node set
    std::vector<void *> data;
    std::vector<int32_t> lo, hi, parent;
    std::vector<uint16_t> axis;
    void (*destr)(void *) = nullptr;
    pct_cloud *cloud = nullptr;
    int64_t synced = 0;           // nodes [0, synced) are in HBM
    // optional per-node planner data {x, y, z, radius} for the fused expansion step (kdx_set_node_aux / kdx_expand_batch):
    // master copy here, mirrored into the cloud's host-mapped array
    std::vector<double> aux;
    double *aux_mapped = nullptr;
    int64_t count() const { return (int64_t)data.size(); }
};

struct kdres {
    kdtree *tree = nullptr;
    std::vector<int32_t> items;   // node ids in ITERATION order
    size_t cursor = 0;
    int size = 0;
};

namespace {

// bring HBM up to date with the host-side node list
int sync_device(kdtree *t)
{
    const int64_t n = t->count();
    if (t->cloud && n > pct_cloud_capacity(t->cloud)) {
        pct_cloud_destroy(t->cloud);
        t->cloud = nullptr;
        t->synced = 0;
        t->aux_mapped = nullptr;
    }
    if (!t->cloud) {
        int64_t cap = 1024;
        while (cap < n) cap *= 2;
        // node sets up to 64k live in host-mapped memory (kd_insert* costs no launch; single queries take the
        // one-launch express kernels); larger trees move to a device-resident cloud
        const int st = cap <= 65536 ? pct_cloud_create_small(cap, &t->cloud) : pct_cloud_create(cap, &t->cloud);
        if (st != PCT_OK) { complain("pct_cloud_create"); t->cloud = nullptr; return -1; }
        t->synced = 0;
    }
    if (t->synced < n) {
        int st = (t->synced == 0) ? pct_cloud_upload_aos(t->cloud, t->posf.data(), n, 12)
                                  : pct_cloud_append_aos(t->cloud, t->posf.data() + 3 * t->synced, n - t->synced, 12);
        if (st != PCT_OK) { complain("cloud upload"); return -1; }
        if (!t->aux.empty() && (t->aux_mapped || pct_cloud_small_aux(t->cloud, &t->aux_mapped) == PCT_OK))
            std::memcpy(t->aux_mapped + 4 * t->synced, t->aux.data() + 4 * t->synced, sizeof(double) * 4 * (size_t)(n - t->synced));
        t->synced = n;
    }
    return 0;
}

// generic trees: bring the fp64 columns up to date
int sync_nodes(kdtree *t)
{
    const int64_t n = t->count();
    if (!t->nodes) {
        if (pct_nodeset_create(t->dim, std::max<int64_t>(n, 1024), &t->nodes) != PCT_OK) { complain("pct_nodeset_create"); t->nodes = nullptr; return -1; }
        t->nsynced = 0;
    }
    if (t->nsynced < n) {
        if (pct_nodeset_append(t->nodes, t->pos.data() + (size_t)t->dim * t->nsynced, n - t->nsynced) != PCT_OK) { complain("node set upload"); return -1; }
        t->nsynced = n;
    }
    return 0;
}

// a 3-D fp32 tree met a double that fp32 cannot hold: from here on its device copy is the fp64 node set
void make_generic(kdtree *t)
{
    t->generic = true;
    if (t->cloud) { pct_cloud_destroy(t->cloud); t->cloud = nullptr; }
    t->synced = 0;
    t->aux_mapped = nullptr;
    t->posf.clear(); t->posf.shrink_to_fit();
    t->aux.clear(); t->aux.shrink_to_fit();
}

// kdtree.c:136-148: destructor order = left subtree, right subtree, node
void run_destructors(kdtree *t)
{
    if (!t->destr || t->data.empty()) return;
    std::vector<int32_t> stack{ 0 };
    std::vector<uint8_t> state(t->data.size(), 0);
    while (!stack.empty()) {
        const int32_t n = stack.back();
        if (state[n] == 0) { state[n] = 1; if (t->lo[n] != NIL) stack.push_back(t->lo[n]); }
        else if (state[n] == 1) { state[n] = 2; if (t->hi[n] != NIL) stack.push_back(t->hi[n]); }
        else { t->destr(t->data[n]); stack.pop_back(); }
    }
}

// Is node n visited by the reference's range walk for (q, range), and where?  The walk position of a node is its root path, one step
// per edge: NEAR = "into the nearer child", FAR = "into the farther one".  Returns false when some far-side step has
// fabs(dx) >= range (kdtree.c:283: the walk never gets there).  Steps come out leaf-to-root in `steps`; their number in `depth`.
enum : uint8_t { kStepNear = 1, kStepFar = 2 };       // 0 closes a path: an ancestor (a proper prefix) sorts before its descendants
constexpr int kPackedDepth = 128;                      // steps that fit the packed key below
// depth (>= 0, steps[0 .. depth) filled leaf-to-root, at most `cap` of them stored), or -1: pruned
int walk_steps(const kdtree *t, int32_t n, const double *q, double range, uint8_t *steps, int cap)
{
    int depth = 0;
    for (int32_t c = n, a = t->parent[n]; a != NIL; c = a, a = t->parent[a]) {
        const int ax = t->axis[a];
        const double dx = q[ax] - t->pos[(size_t)t->dim * a + ax];
        const int32_t near_child = dx <= 0.0 ? t->lo[a] : t->hi[a];
        uint8_t step = kStepNear;
        if (c != near_child) {
            if (!(std::fabs(dx) < range)) return -1;
            step = kStepFar;
        }
        if (depth < cap) steps[depth] = step;
        depth++;
    }
    return depth;
}

// replay the reference walk's pruning (kdtree.c:283) and visit order over a list of in-range nodes.
// Visit order = pre-order, nearer child before farther child = lexicographic order of the root paths with "path ends" < NEAR < FAR.
// A path of up to 128 steps packs into four 64-bit words, two bits per step from the top: comparing the words compares the paths, and
// an ordinary call (a few dozen hits) allocates nothing but its result -- the planner asks this once per RRT* sample
// (corridor_finder.cpp:464).  Everything lives on the stack: queries on one tree may come from several threads, as with the
// reference's double-precision entry points.  Deeper trees (a sorted insertion order can make them) take the byte-string form of the
// same comparison.
kdres *build_range_result(kdtree *t, const double *q, double range, const uint32_t *hits, int64_t nh)
{
    kdres *r = new (std::nothrow) kdres();
    if (!r) return nullptr;
    r->tree = t;
    struct Key { uint64_t w[4]; int32_t id; };
    constexpr int kSmall = 96;
    Key small[kSmall];
    std::vector<Key> large;
    Key *kept = small;
    if (nh > kSmall) { large.resize((size_t)nh); kept = large.data(); }
    int64_t nk = 0;
    uint8_t steps[kPackedDepth];
    bool deep = false;
    for (int64_t i = 0; i < nh; i++) {
        const int depth = walk_steps(t, (int32_t)hits[i], q, range, steps, kPackedDepth);
        if (depth < 0) continue;
        if (depth > kPackedDepth) { deep = true; break; }
        Key &k = kept[nk++];
        k.w[0] = k.w[1] = k.w[2] = k.w[3] = 0;
        k.id = (int32_t)hits[i];
        for (int pos = 0; pos < depth; pos++) k.w[pos >> 5] |= (uint64_t)steps[depth - 1 - pos] << (62 - 2 * (pos & 31));    // step `pos` from the root
    }
    if (!deep) {
        std::sort(kept, kept + nk, [](const Key &a, const Key &b) {
            for (int i = 0; i < 4; i++) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
            return false;
        });
        r->items.resize((size_t)nk);
        for (int64_t i = 0; i < nk; i++) r->items[(size_t)i] = kept[nk - 1 - i].id;       // head insertion => reverse visit order
        r->size = (int)nk;
        return r;
    }
    struct Hit { int32_t id; std::vector<uint8_t> path; };
    std::vector<Hit> all;
    all.reserve((size_t)nh);
    std::vector<uint8_t> longsteps((size_t)t->count());
    for (int64_t i = 0; i < nh; i++) {
        const int depth = walk_steps(t, (int32_t)hits[i], q, range, longsteps.data(), (int)longsteps.size());
        if (depth >= 0) all.push_back(Hit{ (int32_t)hits[i], std::vector<uint8_t>(longsteps.rend() - depth, longsteps.rend()) });
    }
    std::sort(all.begin(), all.end(), [](const Hit &a, const Hit &b) {
        return std::lexicographical_compare(a.path.begin(), a.path.end(), b.path.begin(), b.path.end());
    });
    r->items.reserve(all.size());
    for (size_t i = all.size(); i-- > 0;) r->items.push_back(all[i].id);
    r->size = (int)r->items.size();
    return r;
}

// The reference's winner among several nodes at exactly the minimum distance (kdtree.c:345-402, 432-436).  kd_nearest starts with the
// root as its guess and replaces it only on a STRICTLY smaller distance, walking "nearer subtree, then the node, then the
// farther subtree (if its box is strictly closer than the best so far)".  So: a tied root keeps the answer; otherwise the tied node
// that this walk reaches first wins -- and pruning cannot hide it, because until a tied node has been seen the best distance is
// larger than the minimum, which is at least the box distance of any subtree holding a tied node.  The walk position of a node
// is its root path written as 0 = "into the nearer child", 2 = "into the farther child", closed by 1 = "the node itself".
int32_t reference_tie_winner(const kdtree *t, const double *q, const uint32_t *tied, int64_t n)
{
    std::vector<uint8_t> best, cur;
    int32_t win = NIL;
    for (int64_t k = 0; k < n; k++) {
        const int32_t id = (int32_t)tied[k];
        if (id == 0) return 0;                                   // the root is the initial guess and is never displaced by an equal distance
        cur.clear();
        for (int32_t c = id, a = t->parent[id]; a != NIL; c = a, a = t->parent[a]) {
            const int ax = t->axis[a];
            const double dx = q[ax] - t->pos[(size_t)t->dim * a + ax];
            const int32_t near_child = dx <= 0.0 ? t->lo[a] : t->hi[a];
            cur.push_back(c == near_child ? 0 : 2);
        }
        std::reverse(cur.begin(), cur.end());
        cur.push_back(1);
        if (win == NIL || std::lexicographical_compare(cur.begin(), cur.end(), best.begin(), best.end())) { win = id; best = cur; }
    }
    return win;
}

// ---- small node sets: the host answers ------------------------------------------------------------------------------------------
// The planner's RRT* tree holds tens to a few thousand spheres and asks ONE question at a time (corridor_finder.cpp:428-437, 464):
// a launch plus the bus round trip is ~11 us whatever the tree's size, the reference's own walk well under one.  BASELINE.json
// states this use as config C1, "CPU Utils/kdtree path (plumbing, no GPU)".  Up to PCT_KD_HOST_MAX nodes (default 4096; 0 = always
// the device) the single-query calls below therefore scan the host copy of the node list -- the same fp64 arithmetic
// ((dx^2 + dy^2) + dz^2 on the stored doubles, this file is built with -ffp-contract=off), the same tie rule (reference_tie_winner)
// and the same range order (build_range_result) as the device path, which the tests run on the same fixtures with the threshold at
// 0.  The node set still lives in host-mapped memory for the kernels that read it (the fused expansion batch, kdx_*), the obstacle
// cloud is never answered here, and kd_create still refuses to run without a device: this is a size dispatch, not a fallback.
int64_t g_host_max = -1;              // -1: not read yet
int64_t host_max_nodes()
{
    if (g_host_max < 0) { const char *e = std::getenv("PCT_KD_HOST_MAX"); g_host_max = e ? std::max<int64_t>(std::atoll(e), 0) : 4096; }
    return g_host_max;
}

// kdtree.c:379-382
// the scan over the host copy, with the dimension known to the compiler (the planner's trees: DIM = 3)
template <int DIM>
inline double row_d2(const double *p, const double *q, int dim)
{
    if (DIM == 3) {
        const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        double s = dx * dx;
        s = s + dy * dy;
        s = s + dz * dz;
        return s;
    }
    double s = 0.0;
    for (int k = 0; k < dim; k++) { const double d = p[k] - q[k]; s = s + d * d; }
    return s;
}

template <int DIM>
int32_t host_nearest_dim(const kdtree *t, const double *q)
{
    const int64_t n = t->count();
    const int dim = t->dim;
    const double *P = t->pos.data();
    double best = row_d2<DIM>(P, q, dim);
    int64_t bi = 0, ties = 1;
    for (int64_t i = 1; i < n; i++) {
        const double d = row_d2<DIM>(P + (size_t)dim * i, q, dim);
        if (d < best) { best = d; bi = i; ties = 1; }
        else if (d == best) ties++;
    }
    if (ties == 1) return (int32_t)bi;
    std::vector<uint32_t> tied;
    for (int64_t i = 0; i < n; i++) if (row_d2<DIM>(P + (size_t)dim * i, q, dim) == best) tied.push_back((uint32_t)i);
    return reference_tie_winner(t, q, tied.data(), (int64_t)tied.size());
}
int32_t host_nearest(const kdtree *t, const double *q) { return t->dim == 3 ? host_nearest_dim<3>(t, q) : host_nearest_dim<0>(t, q); }

// node numbers with d2 <= r2 into buf[0 .. cap); returns how many there are (more than cap: the caller repeats with room for all)
template <int DIM>
int64_t host_in_range_dim(const kdtree *t, const double *q, double r2, uint32_t *buf, int64_t cap)
{
    const int64_t n = t->count();
    const int dim = t->dim;
    const double *P = t->pos.data();
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++)
        if (row_d2<DIM>(P + (size_t)dim * i, q, dim) <= r2) { if (m < cap) buf[m] = (uint32_t)i; m++; }
    return m;
}
inline int64_t host_in_range(const kdtree *t, const double *q, double r2, uint32_t *buf, int64_t cap)
{
    return t->dim == 3 ? host_in_range_dim<3>(t, q, r2, buf, cap) : host_in_range_dim<0>(t, q, r2, buf, cap);
}

}  // namespace

extern "C" {

struct kdtree *kd_create(int k)
{
    if (k < 1 || k > 1024) {
        std::fprintf(stderr, "[libkdtree/pct] kd_create(%d): 1 <= k <= 1024\n", k);
        return nullptr;
    }
    if (pct_device_count() <= 0) {
        std::fprintf(stderr, "[libkdtree/pct] kd_create: no HIP device; this library has no host fallback\n");
        return nullptr;
    }
    kdtree *t = new (std::nothrow) kdtree();
    if (t) { t->dim = k; t->generic = (k != 3); }
    return t;
}

void kd_clear(struct kdtree *t)
{
    run_destructors(t);
    t->pos.clear(); t->posf.clear(); t->data.clear(); t->aux.clear();
    t->lo.clear(); t->hi.clear(); t->parent.clear(); t->axis.clear();
    t->synced = 0;
    if (t->cloud) pct_cloud_upload_aos(t->cloud, nullptr, 0, 12);
    t->nsynced = 0;
    if (t->nodes) pct_nodeset_clear(t->nodes);
    t->generic = (t->dim != 3);                       // an emptied 3-D tree is an fp32 tree again until told otherwise
}

void kd_free(struct kdtree *t)
{
    if (!t) return;
    kd_clear(t);
    if (t->cloud) pct_cloud_destroy(t->cloud);
    if (t->nodes) pct_nodeset_destroy(t->nodes);
    delete t;
}

void kd_data_destructor(struct kdtree *t, void (*destr)(void *)) { t->destr = destr; }

// kdtree.c:167-209: strictly smaller on the split axis -> negative side, ties and larger ->
// positive side; a new leaf splits on (parent axis + 1) % dim, the root on axis 0.
int kd_insert(struct kdtree *t, const double *p, void *data)
{
    const int dim = t->dim;
    float pf[3] = { 0.0f, 0.0f, 0.0f };
    if (!t->generic) {
        for (int i = 0; i < 3; i++) {
            pf[i] = (float)p[i];
            if (!((double)pf[i] == p[i])) { make_generic(t); break; }     // a double fp32 cannot hold (or a NaN): fp64 columns from here on
        }
    }
    const int32_t id = (int32_t)t->count();
    int ax = 0;
    int32_t par = NIL;
    try {
        if (id > 0) {
            int32_t cur = 0;
            for (;;) {
                const int a = t->axis[cur];
                int32_t &link = (p[a] < t->pos[(size_t)dim * cur + a]) ? t->lo[cur] : t->hi[cur];
                if (link == NIL) { link = id; ax = (a + 1) % dim; par = cur; break; }
                cur = link;
            }
        }
        t->pos.insert(t->pos.end(), p, p + dim);
        if (!t->generic) {
            t->posf.insert(t->posf.end(), pf, pf + 3);
            const double a[4] = { p[0], p[1], p[2], 0.0 };               // until kdx_set_node_aux says otherwise
            t->aux.insert(t->aux.end(), a, a + 4);
        }
        t->data.push_back(data);
        t->lo.push_back(NIL); t->hi.push_back(NIL); t->parent.push_back(par);
        t->axis.push_back((uint16_t)ax);
    } catch (const std::bad_alloc &) {
        return -1;
    }
    return 0;
}

int kd_insertf(struct kdtree *t, const float *p, void *data)          // kdtree.c:211-241: dim floats widened
{
    double w[kMaxDim];
    for (int i = 0; i < t->dim; i++) w[i] = p[i];
    return kd_insert(t, w, data);
}
// kdtree.c:243-259 hand a 3-element buffer to kd_insert whatever the tree's dimension (it reads past the buffer for k > 3): refused here
int kd_insert3(struct kdtree *t, double x, double y, double z, void *data)
{
    if (t->dim > 3) return three_only("kd_insert3"), -1;
    const double w[3] = { x, y, z };
    return kd_insert(t, w, data);
}
int kd_insert3f(struct kdtree *t, float x, float y, float z, void *data)
{
    if (t->dim > 3) return three_only("kd_insert3f"), -1;
    const double w[3] = { x, y, z };
    return kd_insert(t, w, data);
}

// kdtree.c:404-457 -- NULL for a NULL or empty tree; otherwise a one-element set
struct kdres *kd_nearest(struct kdtree *t, const double *q)
{
    if (!t || t->count() == 0) return nullptr;
    if (t->count() <= host_max_nodes()) {                 // small node set: scanned on the host (see host_max_nodes)
        kdres *r = new (std::nothrow) kdres();
        if (!r) return nullptr;
        r->tree = t;
        r->items.push_back(host_nearest(t, q));
        r->size = 1;
        return r;
    }
    uint32_t idx = PCT_NO_INDEX, ties = 0;
    double d2 = 0;
    if (t->generic) {                                     // fp64 columns of any dimension (csrc/nodeset.hip)
        if (sync_nodes(t)) return nullptr;
        if (pct_nodeset_nearest(t->nodes, q, &idx, &d2, &ties) != PCT_OK) { complain("kd_nearest"); return nullptr; }
        if (ties > 1) {
            std::vector<uint32_t> tied((size_t)ties);
            int64_t nt = 0;
            if (pct_nodeset_radius_indices_r2(t->nodes, q, d2, tied.data(), (int64_t)tied.size(), &nt) != PCT_OK) { complain("kd_nearest (tie set)"); return nullptr; }
            if (nt > 1) idx = (uint32_t)reference_tie_winner(t, q, tied.data(), std::min<int64_t>(nt, (int64_t)tied.size()));
        }
        kdres *r = new (std::nothrow) kdres();
        if (!r) return nullptr;
        r->tree = t;
        r->items.push_back((int32_t)idx);
        r->size = 1;
        return r;
    }
    if (sync_device(t)) return nullptr;
    if (pct_nn_batch_q64_ties(t->cloud, q, 1, &idx, &d2, &ties) != PCT_OK) { complain("kd_nearest"); return nullptr; }
    if (ties != 1) {
        // several nodes at exactly the minimum distance (ties > 1), or a path that does not count them (0: trees beyond 16384
        // nodes): fetch the tied set and pick the node the reference's walk would return
        std::vector<uint32_t> tied((size_t)t->count());
        int64_t nt = 0;
        if (pct_radius_indices_r2_q64(t->cloud, q, d2, tied.data(), (int64_t)tied.size(), &nt) != PCT_OK) { complain("kd_nearest (tie set)"); return nullptr; }
        if (nt > 1) idx = (uint32_t)reference_tie_winner(t, q, tied.data(), std::min<int64_t>(nt, (int64_t)tied.size()));
    }
    kdres *r = new (std::nothrow) kdres();
    if (!r) return nullptr;
    r->tree = t;
    r->items.push_back((int32_t)idx);
    r->size = 1;
    return r;
}
struct kdres *kd_nearestf(struct kdtree *t, const float *q)
{
    if (!t) return nullptr;
    double w[kMaxDim];
    for (int i = 0; i < t->dim; i++) w[i] = q[i];
    return kd_nearest(t, w);
}
struct kdres *kd_nearest3(struct kdtree *t, double x, double y, double z)
{
    if (t && t->dim > 3) return three_only("kd_nearest3"), nullptr;
    const double w[3] = { x, y, z };
    return kd_nearest(t, w);
}
struct kdres *kd_nearest3f(struct kdtree *t, float x, float y, float z)
{
    if (t && t->dim > 3) return three_only("kd_nearest3f"), nullptr;
    const double w[3] = { x, y, z };
    return kd_nearest(t, w);
}

// kdtree.c:537-559 -- an empty tree yields a valid empty set
struct kdres *kd_nearest_range(struct kdtree *t, const double *q, double range)
{
    kdres *r = new (std::nothrow) kdres();
    if (!r) return nullptr;
    r->tree = t;
    const int64_t n = t->count();
    if (n == 0) return r;
    if (n <= host_max_nodes()) {                          // small node set: the in-range nodes from a host scan, then the same replay
        const double r2 = range * range;                  // kdtree.c:273: dist_sq <= SQ(range)
        uint32_t few[256];                                // (one call per RRT* sample: the ordinary call allocates nothing here)
        int64_t m = host_in_range(t, q, r2, few, 256);
        kdres *out;
        if (m <= 256) out = build_range_result(t, q, range, few, m);
        else {
            std::vector<uint32_t> many((size_t)m);
            m = host_in_range(t, q, r2, many.data(), m);
            out = build_range_result(t, q, range, many.data(), m);
        }
        delete r;
        return out;
    }
    std::vector<uint32_t> hits((size_t)n);
    int64_t nh = 0;
    if (t->generic) {
        if (sync_nodes(t)) { delete r; return nullptr; }
        if (pct_nodeset_radius_indices_r2(t->nodes, q, range * range, hits.data(), n, &nh) != PCT_OK) { complain("kd_nearest_range"); delete r; return nullptr; }
    } else {
        if (sync_device(t)) { delete r; return nullptr; }
        if (pct_radius_indices_q64(t->cloud, q, range, hits.data(), n, &nh) != PCT_OK) { complain("kd_nearest_range"); delete r; return nullptr; }
    }
    kdres *out = build_range_result(t, q, range, hits.data(), nh);
    delete r;
    return out;
}
struct kdres *kd_nearest_rangef(struct kdtree *t, const float *q, float range)
{
    double w[kMaxDim];
    for (int i = 0; i < t->dim; i++) w[i] = q[i];
    return kd_nearest_range(t, w, range);
}
struct kdres *kd_nearest_range3(struct kdtree *t, double x, double y, double z, double range)
{
    if (t->dim > 3) return three_only("kd_nearest_range3"), nullptr;
    const double w[3] = { x, y, z };
    return kd_nearest_range(t, w, range);
}
struct kdres *kd_nearest_range3f(struct kdtree *t, float x, float y, float z, float range)
{
    if (t->dim > 3) return three_only("kd_nearest_range3f"), nullptr;
    const double w[3] = { x, y, z };
    return kd_nearest_range(t, w, range);
}

// kdtree.c:613-639.  A NULL result set (kd_nearest* on an empty tree, or a device failure reported through pct_last_error)
// reads as an empty, exhausted set here; the reference dereferences it (kd_res_free(NULL) and kd_res_next on an exhausted
// cursor crash there) -- no correct caller can tell the difference.
void kd_res_free(struct kdres *r) { delete r; }
int kd_res_size(struct kdres *r) { return r ? r->size : 0; }
void kd_res_rewind(struct kdres *r) { if (r) r->cursor = 0; }
int kd_res_end(struct kdres *r) { return !r || r->cursor >= r->items.size(); }
int kd_res_next(struct kdres *r)
{
    if (!r) return 0;
    if (r->cursor < r->items.size()) r->cursor++;
    return r->cursor < r->items.size();
}

// kdtree.c:641-664
void *kd_res_item(struct kdres *r, double *pos)
{
    if (!r || r->cursor >= r->items.size()) return nullptr;
    const int32_t n = r->items[r->cursor];
    if (pos) std::memcpy(pos, &r->tree->pos[(size_t)r->tree->dim * n], (size_t)r->tree->dim * sizeof(double));
    return r->tree->data[n];
}
void *kd_res_itemf(struct kdres *r, float *pos)
{
    if (!r || r->cursor >= r->items.size()) return nullptr;
    const int32_t n = r->items[r->cursor];
    if (pos) for (int i = 0; i < r->tree->dim; i++) pos[i] = (float)r->tree->pos[(size_t)r->tree->dim * n + i];
    return r->tree->data[n];
}
// kdtree.c:666-684: tests the pointee, never returns the payload -- kept as is
void *kd_res_item3(struct kdres *r, double *x, double *y, double *z)
{
    if (r && r->cursor < r->items.size() && r->tree->dim >= 3) {         // (the reference reads three coordinates whatever the dimension)
        const double *p = &r->tree->pos[(size_t)r->tree->dim * r->items[r->cursor]];
        if (*x) *x = p[0];
        if (*y) *y = p[1];
        if (*z) *z = p[2];
    }
    return nullptr;
}
void *kd_res_item3f(struct kdres *r, float *x, float *y, float *z)
{
    if (r && r->cursor < r->items.size() && r->tree->dim >= 3) {
        const double *p = &r->tree->pos[(size_t)r->tree->dim * r->items[r->cursor]];
        if (*x) *x = (float)p[0];
        if (*y) *y = (float)p[1];
        if (*z) *z = (float)p[2];
    }
    return nullptr;
}
void *kd_res_item_data(struct kdres *r) { return kd_res_item(r, nullptr); }

// ---- batch extensions (include/kdtree/kdtree_ext.h) ---------------------------------------------------------
// node sets up to `nodes` answer single queries from the host copy (0 = always the device; negative = back to PCT_KD_HOST_MAX / 4096)
void kdx_set_host_threshold(int64_t nodes) { g_host_max = nodes; }
int64_t kdx_host_threshold(void) { return host_max_nodes(); }
int kdx_size(struct kdtree *t) { return t ? (int)t->count() : 0; }
void *kdx_node_data(struct kdtree *t, int32_t node) { return (t && node >= 0 && node < t->count()) ? t->data[node] : nullptr; }
int kdx_node_pos(struct kdtree *t, int32_t node, double pos[3])
{
    if (!t || t->dim != 3 || node < 0 || node >= t->count()) return -1;
    std::memcpy(pos, &t->pos[3 * (size_t)node], 3 * sizeof(double));
    return 0;
}

int kdx_nearestf_batch(struct kdtree *t, const float *pos, int k, int32_t *node_out)
{
    if (!t || t->generic || k < 0 || (k > 0 && (!pos || !node_out))) return -1;      // fp32 3-D trees only
    if (t->count() == 0) { for (int i = 0; i < k; i++) node_out[i] = -1; return 0; }
    if (sync_device(t)) return -1;
    std::vector<double> q((size_t)3 * k), d2((size_t)k);
    std::vector<uint32_t> idx((size_t)k);
    for (int i = 0; i < 3 * k; i++) q[i] = pos[i];
    if (pct_nn_batch_q64(t->cloud, q.data(), k, idx.data(), d2.data()) != PCT_OK) { complain("kdx_nearestf_batch"); return -1; }
    for (int i = 0; i < k; i++) node_out[i] = (int32_t)idx[i];
    return 0;
}

int kdx_range_candidates_batch(struct kdtree *t, const float *pos, const float *range, int k, uint32_t *ids, int cap_per_query,
                               int32_t *counts)
{
    if (!t || t->generic || k < 0 || (k > 0 && (!pos || !range || !ids || !counts)) || cap_per_query <= 0) return -1;
    for (int i = 0; i < k; i++) counts[i] = 0;
    if (t->count() == 0 || k == 0) return 0;
    if (sync_device(t)) return -1;
    std::vector<double> q((size_t)3 * k), r((size_t)k);
    std::vector<int64_t> cnt((size_t)k);
    for (int i = 0; i < 3 * k; i++) q[i] = pos[i];
    for (int i = 0; i < k; i++) r[i] = range[i];
    if (pct_radius_indices_batch_q64(t->cloud, q.data(), r.data(), k, ids, cap_per_query, cnt.data()) != PCT_OK) {
        for (int i = 0; i < k; i++) counts[i] = -1;          // e.g. a tree too large for the batched kernel: ask one by one
        return 0;
    }
    for (int i = 0; i < k; i++) counts[i] = (int32_t)cnt[i];
    return 0;
}

int kdx_set_node_aux(struct kdtree *t, int32_t node, const double aux[4])
{
    if (!t || t->generic || node < 0 || node >= t->count() || !aux) return -1;
    std::memcpy(&t->aux[4 * (size_t)node], aux, 4 * sizeof(double));
    if (node < t->synced && t->aux_mapped) std::memcpy(t->aux_mapped + 4 * (size_t)node, aux, 4 * sizeof(double));   // plain store into mapped memory
    return 0;
}

int kdx_expand_batch(struct kdtree *t, pct_cloud *obstacles, const pct_inflate_params *prm, const double *samples, int k, int cap_per_query,
                     pct_expand_result *out, uint32_t *ids)
{
    if (!t || t->generic || !obstacles || !prm || k < 0 || (k > 0 && (!samples || !out || !ids)) || cap_per_query <= 0) return -1;
    if (k == 0) return 0;
    if (t->count() > 65536) return -1;                      // the fused kernel serves host-mapped node sets
    if (sync_device(t)) return -1;
    if (t->count() > 0 && !t->aux_mapped) return -1;
    if (pct_rrt_expand_batch(t->cloud, obstacles, prm, samples, k, cap_per_query, out, ids) != PCT_OK) return -1;
    return 0;
}

struct kdres *kdx_range_from_candidates(struct kdtree *t, const float *pos, float range, const uint32_t *ids, int n_ids, int32_t n_snapshot)
{
    if (!t || t->dim != 3) return nullptr;
    const double q[3] = { pos[0], pos[1], pos[2] };
    const double rng = range, r2 = rng * rng;
    std::vector<uint32_t> hits(ids, ids + n_ids);
    for (int64_t j = std::max<int32_t>(n_snapshot, 0); j < t->count(); j++) {      // nodes younger than the snapshot: host test, same arithmetic
        const double *p = &t->pos[3 * (size_t)j];
        const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        double d2 = dx * dx;
        d2 = d2 + dy * dy;
        d2 = d2 + dz * dz;
        if (d2 <= r2) hits.push_back((uint32_t)j);
    }
    return build_range_result(t, q, rng, hits.data(), (int64_t)hits.size());
}

}  // extern "C"
