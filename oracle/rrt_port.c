/*
 * oracle/rrt_port.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's safe-region RRT* corridor finder
 * (/root/reference/Planner/src/corridor_finder.cpp, class in include/pointcloudTraj/corridor_finder.h:17-150,
 * node type include/pointcloudTraj/data_type.h:12-51), written independently of the product's C++ class
 * (include/pct_corridor_finder.hpp): plain C, index-addressed node pool, int vectors, the okd_* port for the
 * RRT* node tree and ocor_radius_search (corridor_port.c) for the obstacle cloud.
 *
 * Parity status: PARITY UNPINNED for the planner logic -- corridor_finder.cpp cannot be compiled here (Eigen,
 * PCL 1.10, roscpp absent) and the reference holds no expected corridors.  What the tests establish with this
 * file is that the GPU-backed corridor finder and this CPU one, which share no code, produce the same Path /
 * Radius for the same seed and iteration counts.  Deliberate differences from the reference, identical in both
 * implementations: wall-clock limits become iteration counts; std::default_random_engine(0) and
 * uniform_real_distribution<double> are spelled out (minstd_rand0, two draws per double, as libstdc++ does);
 * rejected nodes are reclaimed instead of leaked; treeRepair skips the parent test for a parentless node
 * (the reference would dereference NULL there).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct okd_tree okd_tree;
typedef struct okd_res okd_res;
okd_tree *okd_create(int k);
void okd_free(okd_tree *t);
void okd_clear(okd_tree *t);
int okd_insertf(okd_tree *t, const float *pos, void *data);
int okd_insertf_batch(okd_tree *t, const float *xyz, int64_t n);
okd_res *okd_nearestf(okd_tree *t, const float *q);
okd_res *okd_nearest_rangef(okd_tree *t, const float *q, float range);
void okd_res_free(okd_res *r);
int okd_res_end(okd_res *r);
int okd_res_next(okd_res *r);
void *okd_res_item_data(okd_res *r);

typedef struct { double start[3]; double sample_range, search_margin, max_radius; int cloud_empty; } ocor_params;
double ocor_radius_search(const ocor_params *p, okd_tree *map, const double *pt, int32_t *idx_out, double *d2_out);

#define RRT_INF 9999999.0
#define NONE (-1)

typedef struct { int *v; int n, cap; } ivec;
static void iv_push(ivec *a, int x)
{
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 8; a->v = (int *)realloc(a->v, sizeof(int) * (size_t)a->cap); }
    a->v[a->n++] = x;
}
static void iv_clear(ivec *a) { a->n = 0; }
static void iv_free(ivec *a) { free(a->v); a->v = 0; a->n = a->cap = 0; }
static void iv_copy(ivec *dst, const ivec *src) { iv_clear(dst); for (int i = 0; i < src->n; i++) iv_push(dst, src->v[i]); }

typedef struct {
    double c[3];
    float radius, g, f, rel_dis;
    int valid, best, change, rel_id, pre, alive;
    ivec kids;
} rnode;

typedef struct orrt {
    okd_tree *map;            /* obstacle cloud (owned) */
    int64_t map_n;
    okd_tree *tree;           /* RRT* node tree */
    rnode *nodes; int nn, ncap;
    ivec NodeList, EndList, PathList, invalidSet;
    int best_end, root;
    double start[3], end[3], commit_root[3], trans[3], r0[3], r1[3], r2[3];
    int cach_size, max_samples;
    double x_l, x_h, y_l, y_h, z_l, z_h, inlier_ratio, goal_ratio;
    double xin_lo, xin_hi, yin_lo, yin_hi, z_lo, z_hi;
    double safety_margin, max_radius, search_margin, sample_range;
    double min_distance, best_distance, elli_l, elli_s;
    int inform_status, path_exist_status, global_navi_status;
    double *Path, *Radius; int npath;
    uint32_t rng;
    uint64_t n_inflate;
} orrt;

/* ---- minstd_rand0 + libstdc++'s uniform_real_distribution<double> ---- */
static uint32_t rng_next(orrt *s) { s->rng = (uint32_t)(((uint64_t)s->rng * 16807ull) % 2147483647ull); return s->rng; }
static double rng_uniform(orrt *s, double a, double b)
{
    const double r = 2147483646.0;
    double acc = (double)(rng_next(s) - 1u);
    acc += (double)(rng_next(s) - 1u) * r;
    double u = acc / (r * r);
    if (u >= 1.0) u = nextafter(1.0, 0.0);
    return u * (b - a) + a;
}

static double dis3(const double *a, const double *b)       /* corridor_finder.cpp:101-111 */
{
    return sqrt(pow(a[0] - b[0], 2) + pow(a[1] - b[1], 2) + pow(a[2] - b[2], 2));
}
static void normalize3(double *v)
{
    double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (n2 > 0) { double n = sqrt(n2); v[0] /= n; v[1] /= n; v[2] /= n; }
}
static void cross3(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static void update_ellipsoid(orrt *s, const double *toward, const double *centre)     /* :77-85, :285-295 */
{
    const double down[3] = { 0, 0, -1 };
    memcpy(s->trans, centre, sizeof s->trans);
    for (int i = 0; i < 3; i++) s->r0[i] = toward[i] - s->trans[i];
    normalize3(s->r0);
    cross3(s->r0, down, s->r1);
    normalize3(s->r1);
    cross3(s->r0, s->r1, s->r2);
}

static int node_new(orrt *s, const double *c, float radius, float g, float f)
{
    if (s->nn == s->ncap) { s->ncap = s->ncap ? s->ncap * 2 : 256; s->nodes = (rnode *)realloc(s->nodes, sizeof(rnode) * (size_t)s->ncap); }
    rnode *n = &s->nodes[s->nn];
    memset(n, 0, sizeof *n);
    memcpy(n->c, c, sizeof n->c);
    n->radius = radius; n->g = g; n->f = f;
    n->rel_id = -2; n->rel_dis = -1.0f; n->valid = 1; n->pre = NONE; n->alive = 1;
    return s->nn++;
}
static void node_delete(orrt *s, int i) { iv_free(&s->nodes[i].kids); s->nodes[i].alive = 0; }

static double radius_search(orrt *s, const double *p)     /* :113-133 */
{
    ocor_params prm;
    memcpy(prm.start, s->start, sizeof prm.start);
    prm.sample_range = s->sample_range; prm.search_margin = s->search_margin; prm.max_radius = s->max_radius;
    prm.cloud_empty = s->map_n == 0;
    s->n_inflate++;
    return ocor_radius_search(&prm, s->map, p, 0, 0);
}

/* ---------------------------------------------------------------- public ---- */
orrt *orrt_create(void)
{
    orrt *s = (orrt *)calloc(1, sizeof *s);
    s->cach_size = 10;                 /* corridor_finder.cpp:8 */
    s->max_samples = 30000;
    s->rng = 1;                        /* default_random_engine(0): a zero seed becomes 1 */
    s->best_distance = RRT_INF;
    s->path_exist_status = 1;
    s->best_end = s->root = NONE;
    s->map = okd_create(3);
    return s;
}

static void tree_destruct(orrt *s)                         /* :645-654 */
{
    if (s->tree) { okd_free(s->tree); s->tree = 0; }
    for (int i = 0; i < s->nn; i++) if (s->nodes[i].alive) node_delete(s, i);
    s->nn = 0;
}

void orrt_destroy(orrt *s)
{
    tree_destruct(s);
    okd_free(s->map);
    free(s->nodes);
    iv_free(&s->NodeList); iv_free(&s->EndList); iv_free(&s->PathList); iv_free(&s->invalidSet);
    free(s->Path); free(s->Radius);
    free(s);
}

void orrt_set_param(orrt *s, double safety_margin, double search_margin, double max_radius, double sample_range)   /* :17-23 */
{
    s->safety_margin = safety_margin; s->search_margin = search_margin; s->max_radius = max_radius; s->sample_range = sample_range;
}

void orrt_reset(orrt *s)                                   /* :25-41 */
{
    tree_destruct(s);
    iv_clear(&s->NodeList); iv_clear(&s->EndList); iv_clear(&s->invalidSet); iv_clear(&s->PathList);
    s->best_end = s->root = NONE;
    s->path_exist_status = 1; s->inform_status = 0; s->global_navi_status = 0;
    s->best_distance = RRT_INF;
}

void orrt_set_input(orrt *s, const float *xyz, int64_t n)  /* :93-99; xyz packed, inserted in the given order */
{
    okd_clear(s->map);
    s->map_n = n;
    if (n) okd_insertf_batch(s->map, xyz, n);
}

void orrt_set_start_pt(orrt *s, const double *start, const double *end)   /* :43-50 */
{
    memcpy(s->start, start, sizeof s->start); memcpy(s->end, end, sizeof s->end);
    s->xin_lo = start[0] - s->sample_range; s->xin_hi = start[0] + s->sample_range;
    s->yin_lo = start[1] - s->sample_range; s->yin_hi = start[1] + s->sample_range;
}

void orrt_set_pt(orrt *s, const double *start, const double *end, double xl, double xh, double yl, double yh, double zl, double zh,
                 double local_range, int max_iter, double sample_portion, double goal_portion)   /* :52-91 */
{
    double mid[3];
    memcpy(s->start, start, sizeof s->start); memcpy(s->end, end, sizeof s->end);
    s->x_l = xl; s->x_h = xh; s->y_l = yl; s->y_h = yh; s->z_l = zl; s->z_h = zh;
    s->z_lo = zl + s->safety_margin; s->z_hi = zh;
    s->xin_lo = start[0] - s->sample_range; s->xin_hi = start[0] + s->sample_range;   /* old sample_range, as in the reference */
    s->yin_lo = start[1] - s->sample_range; s->yin_hi = start[1] + s->sample_range;
    s->min_distance = sqrt(pow(start[0] - end[0], 2) + pow(start[1] - end[1], 2) + pow(start[2] - end[2], 2));
    for (int i = 0; i < 3; i++) mid[i] = (start[i] + end[i]) / 2.0;
    update_ellipsoid(s, end, mid);
    s->sample_range = local_range;
    s->max_samples = max_iter;
    s->inlier_ratio = sample_portion;
    s->goal_ratio = goal_portion;
}

static int check_end(const orrt *s, int i) { return dis3(s->nodes[i].c, s->end) + 0.1 < s->nodes[i].radius; }   /* :418-426 */
static int node_relation(const orrt *s, double dis, int a, int b)                                               /* :439-454 */
{
    const rnode *n1 = &s->nodes[a], *n2 = &s->nodes[b];
    if ((dis + n2->radius) == n1->radius) return 1;
    if ((dis + 0.1) < 0.95 * (n1->radius + n2->radius)) return -1;
    return 0;
}
static int node_update(const orrt *s, double nr, double orad) { return nr < s->safety_margin ? -1 : (nr < orad ? 0 : 1); }   /* :661-669 */
static int is_successor(const orrt *s, int cur, int near_)                                                      /* :670-683 */
{
    for (int p = s->nodes[near_].pre; p != NONE; p = s->nodes[p].pre) if (p == cur) return 1;
    return 0;
}
static int check_valid_end(const orrt *s, int e)                                                                /* :685-702 */
{
    for (int p = e; p != NONE; p = s->nodes[p].pre) {
        if (!s->nodes[p].valid) return 0;
        if (dis3(s->nodes[p].c, s->nodes[s->root].c) < s->nodes[p].radius) return 1;
    }
    return 0;
}
static void kd_put(orrt *s, int i)
{
    float pos[3] = { (float)s->nodes[i].c[0], (float)s->nodes[i].c[1], (float)s->nodes[i].c[2] };
    okd_insertf(s->tree, pos, (void *)(intptr_t)(i + 1));
}
static void clear_branch_s(orrt *s, int i)                 /* :151-159 */
{
    for (int k = 0; k < s->nodes[i].kids.n; k++) {
        int c = s->nodes[i].kids.v[k];
        if (s->nodes[c].valid) iv_push(&s->invalidSet, c);
        s->nodes[c].valid = 0;
        clear_branch_s(s, c);
    }
}
static void clear_branch_w(orrt *s, int i)                 /* :135-149 */
{
    for (int k = 0; k < s->nodes[i].kids.n; k++) {
        int c = s->nodes[i].kids.v[k];
        if (s->nodes[c].best) continue;
        if (s->nodes[c].valid) iv_push(&s->invalidSet, c);
        s->nodes[c].valid = 0;
        clear_branch_w(s, c);
    }
}
static void remove_change_flagged(orrt *s, int parent)
{
    ivec *k = &s->nodes[parent].kids;
    int w = 0;
    for (int r = 0; r < k->n; r++) if (!s->nodes[k->v[r]].change) k->v[w++] = k->v[r];
    k->n = w;
}
static void remove_invalid(orrt *s)                        /* :170-231 */
{
    ivec keep = { 0, 0, 0 }, ends = { 0, 0, 0 };
    okd_clear(s->tree);
    for (int k = 0; k < s->NodeList.n; k++) {
        int i = s->NodeList.v[k];
        if (s->nodes[i].valid) {
            kd_put(s, i);
            iv_push(&keep, i);
            if (check_end(s, i)) iv_push(&ends, i);
        }
    }
    iv_copy(&s->NodeList, &keep); iv_copy(&s->EndList, &ends);
    iv_free(&keep); iv_free(&ends);
    for (int k = 0; k < s->invalidSet.n; k++) {
        int i = s->invalidSet.v[k];
        if (s->nodes[i].pre != NONE) { s->nodes[i].change = 1; remove_change_flagged(s, s->nodes[i].pre); }
    }
    for (int k = 0; k < s->invalidSet.n; k++) {
        int i = s->invalidSet.v[k];
        for (int c = 0; c < s->nodes[i].kids.n; c++) { int ch = s->nodes[i].kids.v[c]; if (s->nodes[ch].valid) s->nodes[ch].pre = NONE; }
    }
    for (int k = 0; k < s->invalidSet.n; k++) node_delete(s, s->invalidSet.v[k]);
    iv_clear(&s->invalidSet);
}
static void tree_prune(orrt *s, int i)                     /* :161-169 */
{
    if (s->nodes[i].g + s->nodes[i].f > s->best_distance) { s->nodes[i].valid = 0; iv_push(&s->invalidSet, i); clear_branch_s(s, i); }
}
static void update_heuristic(orrt *s, int e)               /* :298-331 */
{
    double cost = s->nodes[e].g + dis3(s->nodes[e].c, s->end) + dis3(s->nodes[s->root].c, s->commit_root);
    if (cost < s->best_distance) {
        s->best_distance = cost;
        s->elli_l = cost;
        s->elli_s = sqrt(cost * cost - s->min_distance * s->min_distance);
        if (s->inform_status) for (int k = 0; k < s->NodeList.n; k++) s->nodes[s->NodeList.v[k]].best = 0;
        for (int p = e; p != NONE; p = s->nodes[p].pre) s->nodes[p].best = 1;
        s->best_end = e;
    }
}
static void gen_sample(orrt *s, double *pt)                /* :333-383 */
{
    double bias = rng_uniform(s, 0.0, 1.0);
    if (bias <= s->goal_ratio) { memcpy(pt, s->end, sizeof(double) * 3); return; }
    if (!s->inform_status) {
        if (bias > s->goal_ratio && bias <= (s->goal_ratio + s->inlier_ratio)) {
            pt[0] = rng_uniform(s, s->xin_lo, s->xin_hi); pt[1] = rng_uniform(s, s->yin_lo, s->yin_hi); pt[2] = rng_uniform(s, s->z_lo, s->z_hi);
        } else {
            pt[0] = rng_uniform(s, s->x_l, s->x_h); pt[1] = rng_uniform(s, s->y_l, s->y_h); pt[2] = rng_uniform(s, s->z_lo, s->z_hi);
        }
    } else {
        double us = rng_uniform(s, 0.0, 1.0), vs = rng_uniform(s, 0.0, 1.0), phis = rng_uniform(s, 0.0, 2 * M_PI);
        double as = s->elli_l / 2.0 * cbrt(us), bs = s->elli_s / 2.0 * cbrt(us), th = acos(1 - 2 * vs);
        double e0 = as * sin(th) * cos(phis), e1 = bs * sin(th) * sin(phis), e2 = bs * cos(th);
        for (int i = 0; i < 3; i++) pt[i] = s->r0[i] * e0 + s->r1[i] * e1 + s->r2[i] * e2 + s->trans[i];
        pt[0] = fmin(fmax(pt[0], s->x_l), s->x_h);
        pt[1] = fmin(fmax(pt[1], s->y_l), s->y_h);
        pt[2] = fmin(fmax(pt[2], s->z_l), s->z_h);
    }
}
static int find_nearest_vertex(orrt *s, const double *pt)  /* :428-437 */
{
    float pos[3] = { (float)pt[0], (float)pt[1], (float)pt[2] };
    okd_res *r = okd_nearestf(s->tree, pos);
    if (!r) return NONE;
    int i = (int)((intptr_t)okd_res_item_data(r) - 1);
    okd_res_free(r);
    return i;
}
static int gen_new_node(orrt *s, const double *sample, int nearest)      /* :385-410 */
{
    const rnode *nn = &s->nodes[nearest];
    double d = dis3(nn->c, sample), c[3];
    if (d > nn->radius) {
        double k = nn->radius / d;
        for (int i = 0; i < 3; i++) c[i] = nn->c[i] + (sample[i] - nn->c[i]) * k;
    } else memcpy(c, sample, sizeof c);
    double rad = radius_search(s, c);
    double h = dis3(c, s->end);
    return node_new(s, c, (float)rad, (float)RRT_INF, (float)h);
}
static void tree_rewire(orrt *s, int nw, int nearest)      /* :457-567 */
{
    float range = s->nodes[nw].radius * 2.0f;
    float pos[3] = { (float)s->nodes[nw].c[0], (float)s->nodes[nw].c[1], (float)s->nodes[nw].c[2] };
    okd_res *res = okd_nearest_rangef(s->tree, pos, range);
    ivec near = { 0, 0, 0 }, vert = { 0, 0, 0 };
    int invalid = 0;
    while (!okd_res_end(res)) {
        int p = (int)((intptr_t)okd_res_item_data(res) - 1);
        double d = dis3(s->nodes[p].c, s->nodes[nw].c);
        int rel = node_relation(s, d, p, nw);
        s->nodes[p].rel_id = rel;
        s->nodes[p].rel_dis = (float)d;
        iv_push(&near, p);
        if (rel == 1) { s->nodes[nw].valid = 0; invalid = 1; break; }
        okd_res_next(res);
    }
    okd_res_free(res);
    if (invalid) {
        for (int k = 0; k < near.n; k++) { s->nodes[near.v[k]].rel_id = -2; s->nodes[near.v[k]].rel_dis = -1.0f; }
        iv_free(&near);
        return;
    }
    double min_cost = s->nodes[nearest].g + dis3(s->nodes[nearest].c, s->nodes[nw].c);
    s->nodes[nw].pre = nearest;
    s->nodes[nw].g = (float)min_cost;
    iv_push(&s->nodes[nearest].kids, nw);
    int last_parent = nearest;
    for (int k = 0; k < near.n; k++) {
        int p = near.v[k];
        int rel = s->nodes[p].rel_id;
        double d = s->nodes[p].rel_dis;
        double cost = s->nodes[p].g + d;
        if (rel == -1) {
            if (cost < min_cost) {
                min_cost = cost;
                s->nodes[nw].pre = p;
                s->nodes[nw].g = (float)min_cost;
                s->nodes[last_parent].kids.n--;            /* pop_back */
                last_parent = p;
                iv_push(&s->nodes[last_parent].kids, nw);
            }
            iv_push(&vert, p);
        }
        s->nodes[p].rel_id = -2;
        s->nodes[p].rel_dis = -1.0f;
    }
    for (int k = 0; k < vert.n; k++) {
        int p = vert.v[k];
        if (!s->nodes[p].valid) continue;
        double d = dis3(s->nodes[p].c, s->nodes[nw].c);
        double cost = d + s->nodes[nw].g;
        if (cost < s->nodes[p].g) {
            if (is_successor(s, p, s->nodes[nw].pre)) continue;
            if (s->nodes[p].pre == NONE) {
                s->nodes[p].pre = nw;
                s->nodes[p].g = (float)cost;
            } else {
                int old_parent = s->nodes[p].pre;
                s->nodes[p].pre = nw;
                s->nodes[p].g = (float)cost;
                s->nodes[p].change = 1;
                remove_change_flagged(s, old_parent);
                s->nodes[p].change = 0;
            }
            iv_push(&s->nodes[nw].kids, p);
        }
    }
    iv_free(&near); iv_free(&vert);
}
static void trace_path(orrt *s)                            /* :575-643 */
{
    ivec feas = { 0, 0, 0 };
    for (int k = 0; k < s->EndList.n; k++) {
        int e = s->EndList.v[k];
        if (check_valid_end(s, e) && check_end(s, e) && s->nodes[e].valid) iv_push(&feas, e);
    }
    free(s->Path); free(s->Radius);
    if (feas.n == 0) {
        s->path_exist_status = 0; s->best_distance = RRT_INF; s->inform_status = 0;
        iv_clear(&s->EndList);
        s->npath = 3;
        s->Path = (double *)calloc(9, sizeof(double)); s->Radius = (double *)calloc(3, sizeof(double));
        s->Path[0] = s->Path[4] = s->Path[8] = 1.0;
        iv_free(&feas);
        return;
    }
    iv_copy(&s->EndList, &feas);
    s->best_end = feas.v[0];
    double best_cost = RRT_INF;
    for (int k = 0; k < feas.n; k++) {
        int e = feas.v[k];
        double cost = s->nodes[e].g + dis3(s->nodes[e].c, s->end) + dis3(s->nodes[s->root].c, s->commit_root);
        if (cost < best_cost) { s->best_end = e; best_cost = cost; s->best_distance = best_cost; }
    }
    iv_clear(&s->PathList);
    for (int p = s->best_end; p != NONE; p = s->nodes[p].pre) iv_push(&s->PathList, p);
    int k = s->PathList.n;
    s->npath = k;
    s->Path = (double *)malloc(sizeof(double) * 3 * (size_t)k); s->Radius = (double *)malloc(sizeof(double) * (size_t)k);
    for (int i = 0; i < k; i++) {
        const rnode *n = &s->nodes[s->PathList.v[i]];
        memcpy(s->Path + 3 * (k - 1 - i), n->c, sizeof(double) * 3);
        s->Radius[k - 1 - i] = n->radius;
    }
    s->path_exist_status = 1;
    iv_free(&feas);
}
static void grow_once(orrt *s, int refine)                 /* loop bodies :719-756 / :772-808 */
{
    double sample[3];
    gen_sample(s, sample);
    int nearest = find_nearest_vertex(s, sample);
    if (nearest == NONE || !s->nodes[nearest].valid) return;
    int nw = gen_new_node(s, sample, nearest);
    if (s->nodes[nw].c[2] < s->z_l || s->nodes[nw].radius < s->safety_margin) { node_delete(s, nw); return; }
    tree_rewire(s, nw, nearest);
    if (!s->nodes[nw].valid) { node_delete(s, nw); return; }
    if (check_end(s, nw)) {
        if (!s->inform_status) s->best_end = nw;
        iv_push(&s->EndList, nw);
        if (refine) update_heuristic(s, nw);
        s->inform_status = 1;
    }
    kd_put(s, nw);
    iv_push(&s->NodeList, nw);
    tree_prune(s, nw);
    if (s->invalidSet.n >= s->cach_size) remove_invalid(s);
}

void orrt_expansion(orrt *s, int64_t iterations)           /* :704-763 */
{
    s->tree = okd_create(3);
    memcpy(s->commit_root, s->start, sizeof s->commit_root);
    s->root = node_new(s, s->start, (float)radius_search(s, s->start), 0.0f, (float)s->min_distance);
    iv_push(&s->NodeList, s->root);
    kd_put(s, s->root);
    int64_t limit = iterations < s->max_samples ? iterations : s->max_samples;
    for (int64_t it = 0; it < limit; it++) grow_once(s, 0);
    remove_invalid(s);
    trace_path(s);
}
void orrt_refine(orrt *s, int64_t iterations)              /* :765-815 */
{
    for (int64_t it = 0; it < iterations; it++) grow_once(s, 1);
    remove_invalid(s);
    trace_path(s);
}

typedef struct { double c[3]; double r; } failrec;

/* The reference boxes Evaluate / treeRepair by wall clock (:900-901, :950-951).  A restatement cannot share a clock with the thing
 * it checks, so it states the box at its two deterministic ends: 0 = the clock never runs out (iteration-count form), 1 = it has
 * run out at every check -- the route is lost unless the first pass leaves it intact, and no broken sphere is repaired. */
static int g_clock_exhausted = 0;

static void tree_repair(orrt *s, failrec *fails, int nf)   /* :938-1021 */
{
    for (int i = 0; i < nf; i++) {
        if (g_clock_exhausted) break;                      /* :950-951 */
        float range = (float)fails[i].r * 2.0f;
        float pos[3] = { (float)fails[i].c[0], (float)fails[i].c[1], (float)fails[i].c[2] };
        okd_res *res = okd_nearest_rangef(s->tree, pos, range);
        while (!okd_res_end(res)) {
            int p = (int)((intptr_t)okd_res_item_data(res) - 1);
            okd_res_next(res);
            if (!s->nodes[p].valid) continue;
            int pre = s->nodes[p].pre;
            if (pre == s->root || p == s->root) continue;
            double nr = radius_search(s, s->nodes[p].c);
            int ret = node_update(s, nr, s->nodes[p].radius);
            s->nodes[p].radius = (float)nr;
            if (ret == -1) {
                if (s->nodes[p].valid) { s->nodes[p].valid = 0; iv_push(&s->invalidSet, p); clear_branch_s(s, p); }
                continue;
            }
            if (pre == NONE) continue;
            if (node_relation(s, dis3(s->nodes[pre].c, s->nodes[p].c), pre, p) != -1 && s->nodes[pre].valid) {
                s->nodes[pre].valid = 0; iv_push(&s->invalidSet, pre); clear_branch_s(s, pre);
                continue;
            }
            ivec kids = { 0, 0, 0 };
            iv_copy(&kids, &s->nodes[p].kids);
            for (int k = 0; k < kids.n; k++) {
                int ch = kids.v[k];
                if (node_relation(s, dis3(s->nodes[p].c, s->nodes[ch].c), p, ch) != -1 && s->nodes[ch].valid) {
                    s->nodes[ch].valid = 0; iv_push(&s->invalidSet, ch); clear_branch_s(s, ch);
                }
            }
            iv_free(&kids);
        }
        okd_res_free(res);
    }
    remove_invalid(s);
}

void orrt_evaluate(orrt *s)                                /* :817-936 */
{
    if (!s->path_exist_status) return;
    failrec *fails = 0; int nf = 0, fcap = 0;
#define PUSH_FAIL(cc, rr) do { if (nf == fcap) { fcap = fcap ? fcap * 2 : 16; fails = (failrec *)realloc(fails, sizeof(failrec) * (size_t)fcap); } \
                               memcpy(fails[nf].c, (cc), sizeof fails[nf].c); fails[nf].r = (rr); nf++; } while (0)
    for (;;) {
        for (int i = 0; i < s->PathList.n; i++) {
            int p = s->PathList.v[i];
            int pre = s->nodes[p].pre;
            if (pre == NONE) continue;
            double nr = radius_search(s, s->nodes[p].c);
            int ret = node_update(s, nr, s->nodes[p].radius);
            double old_r = s->nodes[p].radius;
            s->nodes[p].radius = (float)nr;
            if (ret == -1) {
                s->nodes[p].valid = 0; iv_push(&s->invalidSet, p); clear_branch_s(s, p);
                PUSH_FAIL(s->nodes[p].c, old_r);
            } else if (node_relation(s, dis3(s->nodes[p].c, s->nodes[pre].c), p, pre) != -1) {
                if (s->nodes[p].valid) {
                    s->nodes[p].valid = 0; iv_push(&s->invalidSet, p); clear_branch_s(s, p);
                    PUSH_FAIL(s->nodes[p].c, old_r);
                }
            } else {
                ivec kids = { 0, 0, 0 };
                iv_copy(&kids, &s->nodes[p].kids);
                for (int k = 0; k < kids.n; k++) {
                    int ch = kids.v[k];
                    if (node_relation(s, dis3(s->nodes[p].c, s->nodes[ch].c), p, ch) != -1 && s->nodes[ch].valid) {
                        s->nodes[ch].valid = 0; iv_push(&s->invalidSet, ch); clear_branch_s(s, ch);
                        PUSH_FAIL(s->nodes[ch].c, (double)s->nodes[ch].radius);
                    }
                }
                iv_free(&kids);
            }
        }
        int all_valid = 1;
        for (int i = 0; i < s->PathList.n; i++) all_valid = all_valid && s->nodes[s->PathList.v[i]].valid;
        if (all_valid) break;
        ivec feas = { 0, 0, 0 };
        for (int k = 0; k < s->EndList.n; k++) { int e = s->EndList.v[k]; if (s->nodes[e].valid && check_end(s, e)) iv_push(&feas, e); }
        iv_copy(&s->EndList, &feas);
        if (feas.n == 0 || g_clock_exhausted) {            /* :900-905 */
            s->path_exist_status = 0; s->inform_status = 0; s->best_distance = RRT_INF;
            iv_free(&feas);
            break;
        }
        s->best_end = feas.v[0];
        double best_cost = RRT_INF;
        for (int k = 0; k < feas.n; k++) {
            int e = feas.v[k];
            double cost = s->nodes[e].g + dis3(s->nodes[e].c, s->end) + dis3(s->nodes[s->root].c, s->commit_root);
            if (cost < best_cost) { s->best_end = e; best_cost = cost; s->best_distance = best_cost; }
        }
        iv_clear(&s->PathList);
        for (int p = s->best_end; p != NONE; p = s->nodes[p].pre) iv_push(&s->PathList, p);
        iv_free(&feas);
    }
    remove_invalid(s);
    tree_repair(s, fails, nf);
    trace_path(s);
    free(fails);
#undef PUSH_FAIL
}

void orrt_evaluate_exhausted(orrt *s)                      /* SafeRegionEvaluate(time_limit) with the limit already spent */
{
    g_clock_exhausted = 1;
    orrt_evaluate(s);
    g_clock_exhausted = 0;
}

static void solution_update(orrt *s, double cost_reduction, const double *target)     /* :272-296 */
{
    double mid[3];
    for (int k = 0; k < s->NodeList.n; k++) { rnode *n = &s->nodes[s->NodeList.v[k]]; n->g = (float)((double)n->g - cost_reduction); }
    s->min_distance = dis3(target, s->end);
    for (int i = 0; i < 3; i++) mid[i] = (target[i] + s->end[i]) / 2.0;
    update_ellipsoid(s, target, mid);
    s->best_distance -= cost_reduction;
}

void orrt_reset_root(orrt *s, const double *target)        /* :226-270 */
{
    int lst = s->PathList.v[0];
    if (dis3(s->nodes[lst].c, target) < s->nodes[lst].radius) { s->global_navi_status = 1; return; }
    double cost_reduction = 0;
    memcpy(s->commit_root, target, sizeof s->commit_root);
    ivec cut = { 0, 0, 0 };
    for (int k = 0; k < s->NodeList.n; k++) s->nodes[s->NodeList.v[k]].best = 0;
    int delete_root = 0;
    for (int k = 0; k < s->PathList.n; k++) {
        int p = s->PathList.v[k];
        if (!delete_root && dis3(s->nodes[p].c, target) < (s->nodes[p].radius - 0.1)) {
            delete_root = 1;
            s->nodes[p].best = 1;
            s->nodes[p].pre = NONE;
            cost_reduction = s->nodes[p].g;
            s->root = p;
            continue;
        }
        if (delete_root) { s->nodes[p].best = 0; s->nodes[p].valid = 0; iv_push(&cut, p); }
    }
    solution_update(s, cost_reduction, target);
    for (int k = 0; k < cut.n; k++) { iv_push(&s->invalidSet, cut.v[k]); clear_branch_w(s, cut.v[k]); }
    remove_invalid(s);
    iv_free(&cut);
}

int orrt_check_traj_pt_col(orrt *s, const double *p) { return radius_search(s, p) < 0.0; }   /* :412-416 */

int64_t orrt_get_path(orrt *s, double *path, double *radius, int64_t cap)
{
    for (int64_t i = 0; i < s->npath && i < cap; i++) { memcpy(path + 3 * i, s->Path + 3 * i, sizeof(double) * 3); radius[i] = s->Radius[i]; }
    return s->npath;
}
void orrt_status(orrt *s, int *path_exists, int *global_navi, int64_t *n_nodes, uint64_t *inflations)
{
    if (path_exists) *path_exists = s->path_exist_status;
    if (global_navi) *global_navi = s->global_navi_status;
    if (n_nodes) *n_nodes = s->NodeList.n;
    if (inflations) *inflations = s->n_inflate;
}
