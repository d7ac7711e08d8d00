/* traj_port.c -- TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the reference's Bezier evaluators next to the
 * collision check; glibc pow/sqrt/atan2, x86-64 baseline arithmetic (no FMA).
 *
 * PARITY UNPINNED: the sources need Eigen/roscpp (absent) and the reference holds no fixture for them; each function
 * follows the lines it cites.  Eigen's `.norm()` of a 3-vector is taken as sqrt((x*x + y*y) + z*z); its reduction
 * order for fixed size 3 cannot be confirmed without the library (a last-ulp matter for the arc-length walks).
 *
 *   otraj_state             getStateFromBezier        Planner/src/sim_planning_demo.cpp:688-713
 *   otraj_wire_from_matrix  getBezierTraj             Planner/src/sim_planning_demo.cpp:543-562
 *   otraj_wire_sample       sampling loops            Planner/src/traj_postprocessing.cpp:34-43, 64-73
 *   otraj_segm_index        get_segm_index            Planner/src/traj_postprocessing.cpp:29-57
 *   otraj_nearest_traj      to_nearest_traj           Planner/src/traj_postprocessing.cpp:59-90   (voxels via voxel_port.c)
 *   otraj_end_yaws          to_poly_traj              Planner/src/traj_postprocessing.cpp:152-179
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* binomial_coefs.cpp:3-17: c(n, k) = factorial_from(n, k + 1) / factorial_from(n - k, 2), int arithmetic */
static int factorial_from(int n, int k)
{
    int a = 1;
    for (int i = k; i <= n; i++) a *= i;
    return a;
}
static int binom_int(int n, int k) { return factorial_from(n, k + 1) / factorial_from(n - k, 2); }
/* the three ways this oracle writes "n choose k" (pinned against the reference's compiled table, tests/golden/binomials.npz):
 * out3 = { binomial_coefs' integer form, bezier_base's double form (traj_port), the same in corridor_port } */
double ocor_binom_public(int n, int k);
static double binom_d(int n, int k);
void otraj_binomials(int n, int k, double *out3) { out3[0] = (double)binom_int(n, k); out3[1] = binom_d(n, k); out3[2] = ocor_binom_public(n, k); }

/* bezier_base.cpp:256-266: C(k) = combinatorial(n, k), C_v over n-1, C_a over n-2, held as doubles */
static double binom_d(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    double c = 1.0;
    for (int i = 1; i <= k; i++) c = c * (double)(n - k + i) / (double)i;
    return floor(c + 0.5);
}

void otraj_state(const double *ctrl, int order, double t, double *ret9)
{
    const int m = order + 1;
    for (int i = 0; i < 9; i++) ret9[i] = 0.0;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < m; j++) {
            ret9[i] += binom_d(order, j) * ctrl[i * m + j] * pow(t, j) * pow(1 - t, order - j);
            if (j < m - 1)
                ret9[i + 3] += binom_d(order - 1, j) * order * (ctrl[i * m + j + 1] - ctrl[i * m + j]) * pow(t, j) * pow(1 - t, order - j - 1);
            if (j < m - 2)
                ret9[i + 6] += binom_d(order - 2, j) * order * (order - 1) *
                               (ctrl[i * m + j + 2] - 2 * ctrl[i * m + j + 1] + ctrl[i * m + j]) * pow(t, j) * pow(1 - t, order - j - 2);
        }
    }
}

int64_t otraj_wire_from_matrix(const double *polycoef, int64_t row_stride, const int32_t *orders, int32_t nseg, double *cx, double *cy, double *cz)
{
    int64_t idx = 0;
    for (int i = 0; i < nseg; i++) {
        const int m = orders[i] + 1;
        for (int j = 0; j < m; j++, idx++) {
            cx[idx] = polycoef[i * row_stride + j];
            cy[idx] = polycoef[i * row_stride + m + j];
            cz[idx] = polycoef[i * row_stride + 2 * m + j];
        }
    }
    return idx;
}

static void wire_point(const double *cx, const double *cy, const double *cz, const double *time, const uint32_t *order, int segm, int shift,
                       double t, double *p)
{
    const int n = (int)order[segm];
    p[0] = p[1] = p[2] = 0.0;
    for (int k = 0; k < n + 1; k++) {
        const double c = (double)binom_int(n, k), a = pow(t, k), b = pow(1 - t, n - k);
        p[0] += time[segm] * cx[shift + k] * c * a * b;
        p[1] += time[segm] * cy[shift + k] * c * a * b;
        p[2] += time[segm] * cz[shift + k] * c * a * b;
    }
}

/* every sample of every segment: pos[(s*samples + i)*3], step[(s*samples + i)] = |p - last_p| */
void otraj_wire_sample(const double *cx, const double *cy, const double *cz, const double *time, const uint32_t *order, int32_t nseg,
                       int32_t samples, double *pos, double *step)
{
    if (nseg <= 0) return;
    double last[3] = {cx[0] * time[0], cy[0] * time[0], cz[0] * time[0]};
    int shift = 0;
    for (int s = 0; s < nseg; s++) {
        for (int i = 0; i < samples; i++) {
            const double t = i / (samples - 1.0);
            double p[3];
            wire_point(cx, cy, cz, time, order, s, shift, t, p);
            const int64_t g = (int64_t)s * samples + i;
            const double dx = p[0] - last[0], dy = p[1] - last[1], dz = p[2] - last[2];
            if (pos) { pos[3 * g] = p[0]; pos[3 * g + 1] = p[1]; pos[3 * g + 2] = p[2]; }
            if (step) step[g] = sqrt((dx * dx + dy * dy) + dz * dz);
            last[0] = p[0]; last[1] = p[1]; last[2] = p[2];
        }
        shift += (int)order[s] + 1;
    }
}

void otraj_segm_index(const double *cx, const double *cy, const double *cz, const double *time, const uint32_t *order, int32_t nseg,
                      double twirl_len, int32_t *segm_out, int32_t *part_out)
{
    double len = twirl_len;
    double last[3] = {cx[0] * time[0], cy[0] * time[0], cz[0] * time[0]};
    int shift = 0;
    for (int segm = 0; segm < nseg; segm++) {
        for (int i = 0; i < 1001; i++) {
            const double t = i / 1000.0;
            double p[3];
            wire_point(cx, cy, cz, time, order, segm, shift, t, p);
            const double dx = p[0] - last[0], dy = p[1] - last[1], dz = p[2] - last[2];
            len -= sqrt((dx * dx + dy * dy) + dz * dz);
            last[0] = p[0]; last[1] = p[1]; last[2] = p[2];
            if (len < 0) { *segm_out = segm; *part_out = t > 0.5 ? 1 : 0; return; }
        }
        shift += (int)order[segm] + 1;
    }
    *segm_out = nseg - 1;
    *part_out = 1;
}

/* the points to_nearest_traj hands to traj_voxels.add_point, in order; returns how many (out holds up to cap of them) */
int64_t otraj_nearest_points(const double *cx, const double *cy, const double *cz, const double *time, const uint32_t *order, int32_t nseg,
                             double twirl_len, double *out, int64_t cap)
{
    double len = twirl_len;
    int64_t n = 0;
    if (nseg <= 0) return 0;
    double last[3] = {cx[0] * time[0], cy[0] * time[0], cz[0] * time[0]};
    int shift = 0;
    for (int segm = 0; segm < nseg && len >= 0; segm++) {
        for (int i = 0; i < 1001 && len >= 0; i++) {
            const double t = i / 1000.0;
            double p[3];
            wire_point(cx, cy, cz, time, order, segm, shift, t, p);
            if (n < cap) { out[3 * n] = p[0]; out[3 * n + 1] = p[1]; out[3 * n + 2] = p[2]; }
            n++;
            const double dx = p[0] - last[0], dy = p[1] - last[1], dz = p[2] - last[2];
            len -= sqrt((dx * dx + dy * dy) + dz * dz);
            last[0] = p[0]; last[1] = p[1]; last[2] = p[2];
        }
        shift += (int)order[segm] + 1;
    }
    return n;
}

void otraj_end_yaws(const double *path_x, const double *path_y, int64_t n, const double *coef_x, const double *coef_y, double *end_yaws)
{
    for (int64_t i = 0; i < n; i++) {
        if (i < n - 1) {
            const double vx = path_x[i + 1] - path_x[i], vy = path_y[i + 1] - path_y[i];
            end_yaws[i] = sqrt(vx * vx + vy * vy) > 0.01 ? atan2(vy, vx) : 10;
        } else if (i > 0) {
            end_yaws[i] = end_yaws[i - 1];
        } else {
            const double vx = coef_x[1] - coef_x[0], vy = coef_y[1] - coef_y[0];
            end_yaws[i] = sqrt(vx * vx + vy * vy) > 0.01 ? atan2(vy, vx) : 10;
        }
    }
}
