/*
 * oracle/kdtree_port.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's Utils/kdtree (insertion-ordered k-d tree,
 * exact 1-NN, inclusive radius query, cursor-style result sets).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; nothing under pointcloudtraj_amd/ links or calls it.
 *
 * Parity status: PINNED.  tests/golden/make_golden.py drives this port and the
 * reference's own kdtree.c (compiled unmodified into oracle/_ref/ by
 * oracle/Makefile) on the same inputs and the committed fixtures hold the
 * reference's outputs; tests/test_oracle_golden.py replays them.
 *
 * Every exported function cites the reference lines whose observable behaviour
 * it reproduces (paths relative to /root/reference/Utils/kdtree/src/kdtree.c).
 * The data structure is deliberately different (index-addressed node pool,
 * array-backed result sets, explicit stacks) -- only the observable results
 * (which node wins, which nodes are reported and in which order, return codes)
 * are kept.
 *
 * Symbols carry the prefix okd_ so the oracle can sit in the same process as
 * the product library's kd_* symbols.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OKD_NIL (-1)

typedef struct okd_tree {
    int dim;
    int32_t count, cap;
    double *coord;      /* count * dim, insertion order */
    void **payload;     /* count */
    int32_t *lo, *hi;   /* child links: lo = "negative side", hi = "positive side" */
    uint8_t *axis;      /* split axis of each node */
    int has_box;
    double *box_min, *box_max; /* dim each, valid when has_box */
    void (*destr)(void *);
} okd_tree;

typedef struct okd_res {
    okd_tree *tree;
    int32_t *hit;       /* node ids in VISIT order */
    int32_t nhit, cap;
    int32_t cursor;     /* counts down: iteration is reverse visit order; -1 = end */
    int32_t size;
} okd_res;

/* kdtree.c:112-126 */
okd_tree *okd_create(int k)
{
    okd_tree *t = (okd_tree *)calloc(1, sizeof *t);
    if (!t) return 0;
    t->dim = k;
    return t;
}

static void okd_release_nodes(okd_tree *t)
{
    /* kdtree.c:136-148: destructor runs left subtree, right subtree, then node
     * (post-order).  Reproduce that call order with an explicit stack. */
    if (t->destr && t->count > 0) {
        int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)t->count + 2);
        uint8_t *state = (uint8_t *)calloc((size_t)t->count, 1);
        int32_t sp = 0;
        stack[sp++] = 0;
        while (sp > 0) {
            int32_t n = stack[sp - 1];
            if (state[n] == 0) {
                state[n] = 1;
                if (t->lo[n] != OKD_NIL) stack[sp++] = t->lo[n];
            } else if (state[n] == 1) {
                state[n] = 2;
                if (t->hi[n] != OKD_NIL) stack[sp++] = t->hi[n];
            } else {
                t->destr(t->payload[n]);
                sp--;
            }
        }
        free(stack);
        free(state);
    }
    free(t->coord); free(t->payload); free(t->lo); free(t->hi); free(t->axis);
    t->coord = 0; t->payload = 0; t->lo = t->hi = 0; t->axis = 0;
    t->count = t->cap = 0;
}

/* kdtree.c:150-159 */
void okd_clear(okd_tree *t)
{
    okd_release_nodes(t);
    if (t->has_box) {
        free(t->box_min); free(t->box_max);
        t->box_min = t->box_max = 0;
        t->has_box = 0;
    }
}

/* kdtree.c:128-134 */
void okd_free(okd_tree *t)
{
    if (t) { okd_clear(t); free(t); }
}

/* kdtree.c:161-164 */
void okd_data_destructor(okd_tree *t, void (*destr)(void *)) { t->destr = destr; }

static int okd_grow(okd_tree *t)
{
    int32_t ncap = t->cap ? t->cap * 2 : 64;
    double *c = (double *)realloc(t->coord, sizeof(double) * (size_t)ncap * t->dim);
    if (!c) return -1; t->coord = c;
    void **p = (void **)realloc(t->payload, sizeof(void *) * (size_t)ncap);
    if (!p) return -1; t->payload = p;
    int32_t *l = (int32_t *)realloc(t->lo, sizeof(int32_t) * (size_t)ncap);
    if (!l) return -1; t->lo = l;
    int32_t *h = (int32_t *)realloc(t->hi, sizeof(int32_t) * (size_t)ncap);
    if (!h) return -1; t->hi = h;
    uint8_t *a = (uint8_t *)realloc(t->axis, (size_t)ncap);
    if (!a) return -1; t->axis = a;
    t->cap = ncap;
    return 0;
}

/* kdtree.c:167-209: descend from the root; strictly smaller on the node's
 * split axis goes to the negative side, everything else (ties included) to the
 * positive side; a new leaf splits on (parent axis + 1) mod dim, the root on 0;
 * the bounding box is created from / extended by the point. */
int okd_insert(okd_tree *t, const double *pos, void *data)
{
    const int dim = t->dim;
    if (t->count == t->cap && okd_grow(t)) return -1;
    int32_t id = t->count;
    int ax = 0;
    if (id > 0) {
        int32_t cur = 0;
        for (;;) {
            int a = t->axis[cur];
            int32_t *link = (pos[a] < t->coord[(size_t)cur * dim + a]) ? &t->lo[cur] : &t->hi[cur];
            if (*link == OKD_NIL) { *link = id; ax = (a + 1) % dim; break; }
            cur = *link;
        }
    }
    memcpy(t->coord + (size_t)id * dim, pos, sizeof(double) * dim);
    t->payload[id] = data;
    t->lo[id] = t->hi[id] = OKD_NIL;
    t->axis[id] = (uint8_t)ax;
    t->count++;

    if (!t->has_box) {
        t->box_min = (double *)malloc(sizeof(double) * dim);
        t->box_max = (double *)malloc(sizeof(double) * dim);
        memcpy(t->box_min, pos, sizeof(double) * dim);
        memcpy(t->box_max, pos, sizeof(double) * dim);
        t->has_box = 1;
    } else {
        for (int i = 0; i < dim; i++) {           /* kdtree.c:729-741 */
            if (pos[i] < t->box_min[i]) t->box_min[i] = pos[i];
            if (pos[i] > t->box_max[i]) t->box_max[i] = pos[i];
        }
    }
    return 0;
}

#define OKD_MAXDIM 1024        /* the *f forms widen onto the stack; kd_create(k) beyond it is refused */

/* kdtree.c:211-242: float coordinates are widened to double, nothing else. */
int okd_insertf(okd_tree *t, const float *pos, void *data)
{
    double w[OKD_MAXDIM];
    if (t->dim > OKD_MAXDIM) return -1;
    for (int i = 0; i < t->dim; i++) w[i] = pos[i];
    return okd_insert(t, w, data);
}
/* kdtree.c:244-260 */
int okd_insert3(okd_tree *t, double x, double y, double z, void *data)
{ double w[3] = { x, y, z }; return okd_insert(t, w, data); }
int okd_insert3f(okd_tree *t, float x, float y, float z, void *data)
{ double w[3] = { x, y, z }; return okd_insert(t, w, data); }

/* d2 accumulated from 0 in axis order, one rounding per op, no fused multiply-add
 * (kdtree.c:269-272, 379-382, 434-436; this file is built with -ffp-contract=off). */
static inline double okd_dist2(const okd_tree *t, int32_t n, const double *q)
{
    const double *p = t->coord + (size_t)n * t->dim;
    double s = 0;
    for (int i = 0; i < t->dim; i++) { double d = p[i] - q[i]; s += d * d; }
    return s;
}

/* kdtree.c:743-757 */
static inline double okd_box_dist2(int dim, const double *bmin, const double *bmax, const double *q)
{
    double s = 0;
    for (int i = 0; i < dim; i++) {
        if (q[i] < bmin[i]) { double d = bmin[i] - q[i]; s += d * d; }
        else if (q[i] > bmax[i]) { double d = bmax[i] - q[i]; s += d * d; }
    }
    return s;
}

struct okd_nn_ctx {
    const okd_tree *t; const double *q;
    double *bmin, *bmax;
    int32_t best; double best_d2;
};

/* kdtree.c:345-402: nearer child first (q[axis] - node[axis] <= 0 -> negative
 * side), with the box sliced at the node; then the node itself under strict <;
 * then the farther child only when the sliced box is strictly closer than the
 * current best. */
static void okd_nn_visit(struct okd_nn_ctx *c, int32_t n)
{
    const okd_tree *t = c->t;
    const int a = t->axis[n];
    const double split = t->coord[(size_t)n * t->dim + a];
    int32_t near_c, far_c; double *near_side, *far_side;
    if (c->q[a] - split <= 0) { near_c = t->lo[n]; far_c = t->hi[n]; near_side = c->bmax + a; far_side = c->bmin + a; }
    else                      { near_c = t->hi[n]; far_c = t->lo[n]; near_side = c->bmin + a; far_side = c->bmax + a; }

    if (near_c != OKD_NIL) {
        double keep = *near_side; *near_side = split;
        okd_nn_visit(c, near_c);
        *near_side = keep;
    }
    double d2 = okd_dist2(t, n, c->q);
    if (d2 < c->best_d2) { c->best = n; c->best_d2 = d2; }
    if (far_c != OKD_NIL) {
        double keep = *far_side; *far_side = split;
        if (okd_box_dist2(t->dim, c->bmin, c->bmax, c->q) < c->best_d2) okd_nn_visit(c, far_c);
        *far_side = keep;
    }
}

static okd_res *okd_res_new(okd_tree *t)
{
    okd_res *r = (okd_res *)calloc(1, sizeof *r);
    if (!r) return 0;
    r->tree = t; r->cursor = -1;
    return r;
}

static int okd_res_push(okd_res *r, int32_t n)
{
    if (r->nhit == r->cap) {
        int32_t ncap = r->cap ? r->cap * 2 : 16;
        int32_t *h = (int32_t *)realloc(r->hit, sizeof(int32_t) * (size_t)ncap);
        if (!h) return -1;
        r->hit = h; r->cap = ncap;
    }
    r->hit[r->nhit++] = n;
    return 0;
}

/* Core of kd_nearest that also reports the winning squared distance (the
 * reference API never exposes d2; the oracle needs it for the fixtures). */
int okd_nearest_id(okd_tree *t, const double *q, int32_t *id_out, double *d2_out)
{
    if (!t || !t->has_box) return -1;            /* kdtree.c:412-413 */
    double bmin[OKD_MAXDIM], bmax[OKD_MAXDIM];
    if (t->dim > OKD_MAXDIM) return -1;
    memcpy(bmin, t->box_min, sizeof(double) * t->dim);
    memcpy(bmax, t->box_max, sizeof(double) * t->dim);
    struct okd_nn_ctx c = { t, q, bmin, bmax, 0, 0 };
    c.best_d2 = okd_dist2(t, 0, q);              /* kdtree.c:432-436: root is the first guess */
    okd_nn_visit(&c, 0);
    if (id_out) *id_out = c.best;
    if (d2_out) *d2_out = c.best_d2;
    return 0;
}

/* kdtree.c:404-457 */
okd_res *okd_nearest(okd_tree *t, const double *q)
{
    int32_t id;
    if (okd_nearest_id(t, q, &id, 0)) return 0;
    okd_res *r = okd_res_new(t);
    if (!r) return 0;
    if (okd_res_push(r, id)) { free(r); return 0; }
    r->size = 1; r->cursor = 0;
    return r;
}
/* kdtree.c:459-509 */
okd_res *okd_nearestf(okd_tree *t, const float *q)
{
    double w[OKD_MAXDIM];
    if (t->dim > OKD_MAXDIM) return 0;
    for (int i = 0; i < t->dim; i++) w[i] = q[i];
    return okd_nearest(t, w);
}
okd_res *okd_nearest3(okd_tree *t, double x, double y, double z)
{ double w[3] = { x, y, z }; return okd_nearest(t, w); }
okd_res *okd_nearest3f(okd_tree *t, float x, float y, float z)
{ double w[3] = { x, y, z }; return okd_nearest(t, w); }

/* kdtree.c:262-293: pre-order.  A node is reported when d2 <= range*range
 * (inclusive); the nearer child is always entered, the farther one only when
 * fabs(q[axis] - node[axis]) < range (strict).  Explicit stack instead of
 * recursion so degenerate (sorted-insert) trees cannot overflow the C stack. */
static int okd_range_walk(okd_tree *t, const double *q, double range, okd_res *r)
{
    if (t->count == 0) return 0;
    int32_t cap = 64, sp = 0;
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    if (!stack) return -1;
    const double r2 = range * range;
    stack[sp++] = 0;
    while (sp > 0) {
        int32_t n = stack[--sp];
        if (okd_dist2(t, n, q) <= r2 && okd_res_push(r, n)) { free(stack); return -1; }
        const int a = t->axis[n];
        const double dx = q[a] - t->coord[(size_t)n * t->dim + a];
        int32_t near_c = dx <= 0.0 ? t->lo[n] : t->hi[n];
        int32_t far_c  = dx <= 0.0 ? t->hi[n] : t->lo[n];
        if (sp + 2 > cap) {
            cap *= 2;
            int32_t *s2 = (int32_t *)realloc(stack, sizeof(int32_t) * (size_t)cap);
            if (!s2) { free(stack); return -1; }
            stack = s2;
        }
        /* far pushed first so near is popped (visited) first */
        if (far_c != OKD_NIL && fabs(dx) < range) stack[sp++] = far_c;
        if (near_c != OKD_NIL) stack[sp++] = near_c;
    }
    free(stack);
    return 0;
}

/* kdtree.c:537-559: an empty tree yields a valid empty set. */
okd_res *okd_nearest_range(okd_tree *t, const double *q, double range)
{
    okd_res *r = okd_res_new(t);
    if (!r) return 0;
    if (okd_range_walk(t, q, range, r)) { free(r->hit); free(r); return 0; }
    r->size = r->nhit;
    r->cursor = r->nhit - 1;     /* kdtree.c:810-828: head insertion => last visited comes out first */
    return r;
}
/* kdtree.c:561-611 */
okd_res *okd_nearest_rangef(okd_tree *t, const float *q, float range)
{
    double w[OKD_MAXDIM];
    if (t->dim > OKD_MAXDIM) return 0;
    for (int i = 0; i < t->dim; i++) w[i] = q[i];
    return okd_nearest_range(t, w, range);
}
okd_res *okd_nearest_range3(okd_tree *t, double x, double y, double z, double range)
{ double w[3] = { x, y, z }; return okd_nearest_range(t, w, range); }
okd_res *okd_nearest_range3f(okd_tree *t, float x, float y, float z, float range)
{ double w[3] = { x, y, z }; return okd_nearest_range(t, w, range); }

/* kdtree.c:613-639 */
void okd_res_free(okd_res *r) { free(r->hit); free(r); }
int okd_res_size(okd_res *r) { return r->size; }
void okd_res_rewind(okd_res *r) { r->cursor = r->nhit - 1; }
int okd_res_end(okd_res *r) { return r->cursor < 0; }
int okd_res_next(okd_res *r) { r->cursor--; return r->cursor >= 0; }

/* kdtree.c:641-664 */
void *okd_res_item(okd_res *r, double *pos)
{
    if (r->cursor < 0) return 0;
    int32_t n = r->hit[r->cursor];
    if (pos) memcpy(pos, r->tree->coord + (size_t)n * r->tree->dim, sizeof(double) * r->tree->dim);
    return r->tree->payload[n];
}
void *okd_res_itemf(okd_res *r, float *pos)
{
    if (r->cursor < 0) return 0;
    int32_t n = r->hit[r->cursor];
    if (pos) for (int i = 0; i < r->tree->dim; i++) pos[i] = (float)r->tree->coord[(size_t)n * r->tree->dim + i];
    return r->tree->payload[n];
}
/* kdtree.c:666-684: the reference tests the POINTEE (*x), not the pointer, so an
 * output that currently holds 0 is left untouched, and the payload is never
 * returned (always NULL).  Kept verbatim as observable behaviour. */
void *okd_res_item3(okd_res *r, double *x, double *y, double *z)
{
    if (r->cursor >= 0) {
        const double *p = r->tree->coord + (size_t)r->hit[r->cursor] * r->tree->dim;
        if (*x) *x = p[0];
        if (*y) *y = p[1];
        if (*z) *z = p[2];
    }
    return 0;
}
void *okd_res_item3f(okd_res *r, float *x, float *y, float *z)
{
    if (r->cursor >= 0) {
        const double *p = r->tree->coord + (size_t)r->hit[r->cursor] * r->tree->dim;
        if (*x) *x = (float)p[0];
        if (*y) *y = (float)p[1];
        if (*z) *z = (float)p[2];
    }
    return 0;
}
/* kdtree.c:686-689 */
void *okd_res_item_data(okd_res *r) { return okd_res_item(r, 0); }

/* ---- batch helpers for the test harness / CPU baseline (not in the reference API) ---- */

/* node id currently under the cursor (insertion index), -1 at end */
int32_t okd_res_item_id(okd_res *r) { return r->cursor < 0 ? -1 : r->hit[r->cursor]; }

/* Insert n float points in the given order with payload = (index+1). */
int okd_insertf_batch(okd_tree *t, const float *xyz, int64_t n)
{
    for (int64_t i = 0; i < n; i++)
        if (okd_insertf(t, xyz + 3 * i, (void *)(intptr_t)(i + 1))) return -1;
    return 0;
}

/* kd_nearestf over nq float queries; writes insertion index and fp64 d2. */
int okd_nearestf_batch(okd_tree *t, const float *q, int64_t nq, int32_t *idx, double *d2)
{
    for (int64_t i = 0; i < nq; i++) {
        double w[3] = { q[3 * i], q[3 * i + 1], q[3 * i + 2] };
        if (okd_nearest_id(t, w, idx + i, d2 + i)) return -1;
    }
    return 0;
}

/* kd_nearest_rangef result sizes over nq queries (the reference's "radius count"). */
int okd_range_countf_batch(okd_tree *t, const float *q, const float *range, int64_t nq, int32_t *count)
{
    for (int64_t i = 0; i < nq; i++) {
        okd_res *r = okd_nearest_rangef(t, q + 3 * i, range[i]);
        if (!r) return -1;
        count[i] = okd_res_size(r);
        okd_res_free(r);
    }
    return 0;
}

/* Exhaustive fp64 scan in index order, lowest index wins ties: the engine's
 * documented tie rule, used to label ties in fixtures.  Same d2 arithmetic. */
int okd_brute_nearestf(const float *xyz, int64_t n, const float *q, int64_t nq, int32_t *idx, double *d2)
{
    for (int64_t j = 0; j < nq; j++) {
        double qx = q[3 * j], qy = q[3 * j + 1], qz = q[3 * j + 2];
        double best = INFINITY; int32_t bi = -1;
        for (int64_t i = 0; i < n; i++) {
            double dx = (double)xyz[3 * i] - qx, dy = (double)xyz[3 * i + 1] - qy, dz = (double)xyz[3 * i + 2] - qz;
            double s = 0; s += dx * dx; s += dy * dy; s += dz * dz;
            if (s < best) { best = s; bi = (int32_t)i; }
        }
        idx[j] = bi; d2[j] = best;
    }
    return 0;
}

int okd_brute_countf(const float *xyz, int64_t n, const float *q, const float *range, int64_t nq, int32_t *count)
{
    for (int64_t j = 0; j < nq; j++) {
        double qx = q[3 * j], qy = q[3 * j + 1], qz = q[3 * j + 2];
        double r = range[j], r2 = r * r; int32_t c = 0;
        for (int64_t i = 0; i < n; i++) {
            double dx = (double)xyz[3 * i] - qx, dy = (double)xyz[3 * i + 1] - qy, dz = (double)xyz[3 * i + 2] - qz;
            double s = 0; s += dx * dx; s += dy * dy; s += dz * dz;
            c += (s <= r2);
        }
        count[j] = c;
    }
    return 0;
}
