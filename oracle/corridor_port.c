/*
 * oracle/corridor_port.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the planner-side arithmetic that sits on the hot path:
 * sphere inflation (safeRegionRrtStar::radiusSearch and its two wrappers) and
 * the sampled Bezier collision check (checkSafeTrajectory / getPosFromBezier).
 * Paths cited are relative to /root/reference/Planner/.
 *
 * Parity status: the NN inside is the pinned okd_* port (see kdtree_port.c).
 * The planner files themselves cannot be compiled here (Eigen, PCL 1.10, roscpp
 * are absent) and the reference holds no fixtures for them, so the *planner
 * arithmetic around the NN* is PARITY UNPINNED: it follows the reference text
 * line by line but was never diffed against a run of the reference.  One
 * deliberate, documented difference: the reference's obstacle NN runs in
 * PCL/FLANN fp32; here (as the north star specifies) it has Utils/kdtree
 * semantics -- query narrowed to fp32, distances in fp64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct okd_tree okd_tree;
int okd_nearest_id(okd_tree *t, const double *q, int32_t *id_out, double *d2_out);

typedef struct {
    double start[3];       /* start_pt */
    double sample_range;   /* sample_range (== sensing range after setPt) */
    double search_margin;
    double max_radius;
    int cloud_empty;
} ocor_params;

/* src/corridor_finder.cpp:109-111 -- sqrt(pow(.,2)+pow(.,2)+pow(.,2)) */
static double ocor_dist(const double *a, const double *b)
{
    double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return sqrt(dx * dx + dy * dy + dz * dz);
}

/*
 * src/corridor_finder.cpp:113-133.
 *  - farther than sample_range + max_radius from the start, or empty cloud:
 *    max_radius - search_margin (not clamped again);
 *  - otherwise the query is narrowed to float (searchPoint.x = search_Pt(0)),
 *    1-NN, r = sqrt(d2) - search_margin, result min(r, max_radius).
 * idx_out/d2_out receive the NN (or -1 / +inf when the early-out fires).
 */
double ocor_radius_search(const ocor_params *p, okd_tree *map, const double *pt, int32_t *idx_out, double *d2_out)
{
    if (idx_out) *idx_out = -1;
    if (d2_out) *d2_out = INFINITY;
    if (ocor_dist(pt, p->start) > p->sample_range + p->max_radius) return p->max_radius - p->search_margin;
    if (p->cloud_empty) return p->max_radius - p->search_margin;
    double q[3] = { (double)(float)pt[0], (double)(float)pt[1], (double)(float)pt[2] };
    int32_t id; double d2;
    if (okd_nearest_id(map, q, &id, &d2)) return p->max_radius - p->search_margin;
    if (idx_out) *idx_out = id;
    if (d2_out) *d2_out = d2;
    double r = sqrt(d2) - p->search_margin;
    return r < p->max_radius ? r : p->max_radius;   /* std::min(radius, max_radius) */
}

/* src/corridor_finder.cpp:656-659 (checkRadius) and :412-416 (checkTrajPtCol), batched */
void ocor_inflate_batch(const ocor_params *p, okd_tree *map, const double *pts, int64_t n,
                        double *radius, int32_t *idx, double *d2, uint8_t *collide)
{
    for (int64_t i = 0; i < n; i++) {
        radius[i] = ocor_radius_search(p, map, pts + 3 * i, idx ? idx + i : 0, d2 ? d2 + i : 0);
        if (collide) collide[i] = radius[i] < 0.0;
    }
}

/* src/bezier_base.cpp:256-270: C(k) = n choose k, held as doubles */
static double ocor_binom(int n, int k)
{
    double c = 1.0;
    for (int i = 1; i <= k; i++) c = c * (double)(n - k + i) / (double)i;
    return floor(c + 0.5);
}

double ocor_binom_public(int n, int k) { return ocor_binom(n, k); }

/*
 * src/sim_planning_demo.cpp:715-727.  coef is row `seg` of PolyCoeff, laid out
 * [x_0..x_n, y_0..y_n, z_0..z_n] with that segment's own n
 * (src/traj_optimizer.cpp:739-751).  Accumulation: j ascending,
 * ((C * c) * pow(u, j)) * pow(1 - u, n - j).
 */
void ocor_bezier_pos(const double *coef, int order, double u, double *out3)
{
    int m = order + 1;
    for (int d = 0; d < 3; d++) {
        double acc = 0.0;
        for (int j = 0; j < m; j++)
            acc += ocor_binom(order, j) * coef[d * m + j] * pow(u, j) * pow(1.0 - u, order - j);
        out3[d] = acc;
    }
}

/*
 * src/sim_planning_demo.cpp:729-781.  t_start is max(0, now - traj start).
 * Enumerates the sample times exactly like the reference's two nested loops
 * (sequential += 0.02 on both t and the accumulated horizon) and writes them to
 * seg_out/t_out/pos_out (capacity cap).  Returns the number of samples the
 * reference would have evaluated if no collision stopped it.
 */
int64_t ocor_bezier_samples(const double *polycoef, int64_t row_stride, const double *seg_time, const int32_t *orders,
                            int32_t nseg, double t_start, double stop_time, double dt,
                            int32_t *seg_out, double *t_out, double *pos_out, int64_t cap)
{
    double t_s = t_start;
    int idx;
    for (idx = 0; idx < nseg; ++idx) {
        if (t_s > seg_time[idx] && idx + 1 < nseg) t_s -= seg_time[idx];
        else break;
    }
    int64_t n = 0;
    double t_accu = 0.0;
    for (int i = idx; i < nseg; i++) {
        double t_ss = (i == idx) ? t_s : 0.0;
        for (double t = t_ss; t < seg_time[i]; t += dt) {
            t_accu += dt;
            if (t_accu > stop_time) break;       /* leaves only the inner loop, as in the reference */
            if (n < cap) {
                double p[3];
                ocor_bezier_pos(polycoef + (size_t)i * row_stride, orders[i], t / seg_time[i], p);
                seg_out[n] = i; t_out[n] = t;
                pos_out[3 * n] = p[0] * seg_time[i];
                pos_out[3 * n + 1] = p[1] * seg_time[i];
                pos_out[3 * n + 2] = p[2] * seg_time[i];
            }
            n++;
        }
    }
    return n;
}

/*
 * Full check: first sample whose inflation radius is negative (NN distance <
 * search_margin) => returns its index, else -1.  radius_out/d2_out/idx_out (may
 * be NULL) receive per-sample values for ALL enumerated samples (the reference
 * stops at the first hit; the extra values are for test diagnostics).
 */
int64_t ocor_check_safe_trajectory(const ocor_params *p, okd_tree *map,
                                   const double *polycoef, int64_t row_stride, const double *seg_time, const int32_t *orders,
                                   int32_t nseg, double t_start, double stop_time, double dt,
                                   int64_t cap, int64_t *nsamples_out, double *pos_out, double *radius_out, double *d2_out, int32_t *idx_out)
{
    int32_t *seg = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    double *tt = (double *)malloc(sizeof(double) * (size_t)cap);
    double *pos = pos_out ? pos_out : (double *)malloc(sizeof(double) * 3 * (size_t)cap);
    int64_t n = ocor_bezier_samples(polycoef, row_stride, seg_time, orders, nseg, t_start, stop_time, dt, seg, tt, pos, cap);
    if (n > cap) n = cap;
    int64_t first = -1;
    for (int64_t i = 0; i < n; i++) {
        int32_t id; double d2;
        double r = ocor_radius_search(p, map, pos + 3 * i, &id, &d2);
        if (radius_out) radius_out[i] = r;
        if (d2_out) d2_out[i] = d2;
        if (idx_out) idx_out[i] = id;
        if (r < 0.0 && first < 0) first = i;
    }
    if (nsamples_out) *nsamples_out = n;
    free(seg); free(tt);
    if (!pos_out) free(pos);
    return first;
}
