/*
 * kdtree/kdtree.h -- public C API of the drop-in libkdtree.so built by this repository.
 *
 * Same 26 entry points (SURVEY.md counts them as "22"; the header declares 26), names, signatures and calling conventions as the reference's
 * Utils/kdtree/include/kdtree/kdtree.h:39-122, so that Planner/src/corridor_finder.cpp
 * (call sites :174,179,431-434,464-488,647,709,716,750,802,956-1015) links against it
 * unchanged.  The implementation behind it is NOT a pointer tree on the host: points are
 * mirrored into an SoA cloud in HBM and kd_nearest* / kd_nearest_range* run the HIP
 * kernels of pct_engine (see include/pct_engine.h); the host keeps only the insertion
 * topology needed to reproduce the reference's result ORDER for range queries.
 *
 * Behaviour kept from the reference (reference file: Utils/kdtree/src/kdtree.c):
 *   - coordinates are stored as double; the *f variants widen float -> double (:211-242)
 *   - kd_nearest*: exact 1-NN, d2 = ((dx*dx + dy*dy) + dz*dz) in fp64 without FMA
 *     (:379-382); NULL for a NULL or empty tree (:412-413)
 *   - kd_nearest_range*: hit iff d2 <= range*range (:273), far side of a split pruned
 *     unless fabs(dx) < range (:283), iteration order = reverse visit order (:810-828);
 *     an empty tree gives a valid empty set (:537-559)
 *   - result items alias live tree nodes: a set is invalidated by kd_clear/kd_free
 *   - kd_res_item3/kd_res_item3f test the pointee, not the pointer, and return NULL (:666-684)
 *   - kd_nearest* on exact distance ties: the node the reference's own walk returns -- the root if it is
 *     among the tied nodes (it is the initial guess and only a strictly smaller distance displaces it,
 *     :432-436), otherwise the tied node reached first by "nearer subtree, node, farther subtree" (:345-402)
 *   - any dimension, any double: a 3-D tree whose coordinates are all fp32 values (what the *f entry points insert, the
 *     planner's only use) is mirrored into an fp32 cloud; a tree of another dimension, or one that was handed a double
 *     fp32 cannot hold, keeps fp64 columns in HBM and is answered by exhaustive fp64 kernels with the same sums in the
 *     same order (:267-272, :379-382) -- same tie rule, same range order
 * Documented differences:
 *   - kd_create(k) serves 1 <= k <= 1024 (NULL otherwise)
 *   - the x,y,z forms (kd_insert3*, kd_nearest3*, kd_nearest_range3*) refuse trees of more than 3 dimensions, where the
 *     reference reads past its 3-element buffer (:243-259, :493-535); kd_res_item3* leave their arguments alone on trees of
 *     fewer than 3 dimensions
 *   - every query needs a HIP device; there is no host fallback (queries return NULL and
 *     print a diagnostic when the device is missing)
 */
#ifndef PCT_KDTREE_KDTREE_H
#define PCT_KDTREE_KDTREE_H

#ifdef __cplusplus
extern "C" {
#endif

struct kdtree;   /* opaque */
struct kdres;    /* opaque */

/* lifecycle -- kdtree.c:112-164 */
struct kdtree *kd_create(int dimensions);
void kd_free(struct kdtree *kd);
void kd_clear(struct kdtree *kd);
void kd_data_destructor(struct kdtree *kd, void (*on_clear)(void *payload));

/* insertion, 0 on success / -1 on allocation failure -- kdtree.c:167-260 */
int kd_insert(struct kdtree *kd, const double *xyz, void *payload);
int kd_insertf(struct kdtree *kd, const float *xyz, void *payload);
int kd_insert3(struct kdtree *kd, double px, double py, double pz, void *payload);
int kd_insert3f(struct kdtree *kd, float px, float py, float pz, void *payload);

/* exact nearest neighbour: result set of size 1, or NULL -- kdtree.c:345-509 */
struct kdres *kd_nearest(struct kdtree *kd, const double *xyz);
struct kdres *kd_nearestf(struct kdtree *kd, const float *xyz);
struct kdres *kd_nearest3(struct kdtree *kd, double px, double py, double pz);
struct kdres *kd_nearest3f(struct kdtree *kd, float px, float py, float pz);

/* all nodes within `range` (inclusive) -- kdtree.c:262-293, 537-611 */
struct kdres *kd_nearest_range(struct kdtree *kd, const double *xyz, double radius);
struct kdres *kd_nearest_rangef(struct kdtree *kd, const float *xyz, float radius);
struct kdres *kd_nearest_range3(struct kdtree *kd, double px, double py, double pz, double radius);
struct kdres *kd_nearest_range3f(struct kdtree *kd, float px, float py, float pz, float radius);

/* result-set cursor -- kdtree.c:613-689 */
void kd_res_free(struct kdres *rs);
int kd_res_size(struct kdres *rs);
void kd_res_rewind(struct kdres *rs);
int kd_res_end(struct kdres *rs);
int kd_res_next(struct kdres *rs);
void *kd_res_item(struct kdres *rs, double *xyz_out);
void *kd_res_itemf(struct kdres *rs, float *xyz_out);
void *kd_res_item3(struct kdres *rs, double *px, double *py, double *pz);
void *kd_res_item3f(struct kdres *rs, float *px, float *py, float *pz);
void *kd_res_item_data(struct kdres *rs);

#ifdef __cplusplus
}
#endif
#endif /* PCT_KDTREE_KDTREE_H */
