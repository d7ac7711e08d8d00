// traj.hip -- Bezier-trajectory evaluators beside the collision check (include/pct_traj.h), part of libpct_engine.so.
//
// Two small kernels share the Bernstein evaluation with pct_bezier_check: one thread per (segment, u) sample, fp64, the
// reference's accumulation order, -ffp-contract=off.  The work is tiny (10^4 samples x <= 13 control points); what the GPU
// buys is that the samples stay on the device for the voxel map / the collision check that consume them, and that the
// 1001-per-segment walks of traj_postprocessing.cpp cost one launch instead of ~2x10^5 libm pow calls.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/pct_traj.h"
#include "engine_internal.hpp"
#include "bernstein.hpp"

using pct_internal::fail;

namespace {

constexpr int kMaxOrder = 12;
constexpr int kMaxCtrl = kMaxOrder + 1;

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(PCT_ERR_HIP, "%s -> %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define PCTCHK(call)                    \
    do {                                \
        int s_ = (call);                \
        if (s_ != PCT_OK) return s_;    \
    } while (0)

// n choose k as an exact double (bezier_base.cpp:256-266 `combinatorial`, binomial_coefs.cpp:11-17): Pascal's rule in integers
__host__ __device__ inline double binom(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    unsigned long long row[kMaxCtrl + 1] = {1};
    for (int i = 1; i <= n; i++)
        for (int j = i; j >= 1; j--) row[j] = (j == i ? 0ull : row[j]) + row[j - 1];
    return (double)row[k];
}

// test hook: both ways the device writes "n choose k", for every 0 <= k <= n <= 12
__global__ void binomial_tables_kernel(double *__restrict__ pascal, double *__restrict__ recurrence)
{
    const int n = (int)threadIdx.x / 13, k = (int)threadIdx.x % 13;
    if (n < 13) {
        pascal[n * 13 + k] = k <= n ? binom(n, k) : 0.0;
        recurrence[n * 13 + k] = k <= n ? pct::bernstein_binom(n, k) : 0.0;
    }
}

// getStateFromBezier (sim_planning_demo.cpp:688-713).  ctrl = row seg of PolyCoeff: [x_0..x_n, y_0..y_n, z_0..z_n].
__global__ __launch_bounds__(256) void bezier_state_kernel(const double *__restrict__ polycoef, int64_t row_stride,
                                                           const int32_t *__restrict__ orders, const int32_t *__restrict__ seg,
                                                           const double *__restrict__ u, int64_t n, double *__restrict__ state9)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = seg[i];
    const int order = orders[s], m = order + 1;
    const double t = u[i];
    const double *ctrl = polycoef + (int64_t)s * row_stride;
    double C[kMaxCtrl], Cv[kMaxCtrl], Ca[kMaxCtrl];
    for (int j = 0; j < m; j++) { C[j] = binom(order, j); Cv[j] = binom(order - 1, j); Ca[j] = binom(order - 2, j); }
    for (int d = 0; d < 3; d++) {
        double p = 0.0, v = 0.0, a = 0.0;
        for (int j = 0; j < m; j++) {
            const double tj = pct::pow_uint_cr(t, j);
            p += C[j] * ctrl[d * m + j] * tj * pct::pow_uint_cr(1.0 - t, order - j);
            if (j < m - 1)
                v += Cv[j] * (double)order * (ctrl[d * m + j + 1] - ctrl[d * m + j]) * tj * pct::pow_uint_cr(1.0 - t, order - j - 1);
            if (j < m - 2)
                a += Ca[j] * (double)order * (double)(order - 1) * (ctrl[d * m + j + 2] - 2.0 * ctrl[d * m + j + 1] + ctrl[d * m + j]) * tj *
                     pct::pow_uint_cr(1.0 - t, order - j - 2);
        }
        state9[9 * i + d] = p;
        state9[9 * i + 3 + d] = v;
        state9[9 * i + 6 + d] = a;
    }
}

// traj_postprocessing.cpp:36-43 / :66-73: p += time[segm] * coef_vec(shift + k) * c(order, k) * pow(t, k) * pow(1 - t, order - k)
// (per component: (((time * c) * C) * pow) * pow, k ascending), t = i / (samples - 1.0)
__global__ __launch_bounds__(256) void wire_sample_kernel(const double *__restrict__ cx, const double *__restrict__ cy,
                                                          const double *__restrict__ cz, const double *__restrict__ time,
                                                          const uint32_t *__restrict__ order, const uint32_t *__restrict__ shift,
                                                          int32_t nseg, int32_t samples, double *__restrict__ pos)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)nseg * samples) return;
    const int s = (int)(g / samples), i = (int)(g % samples);
    const int n = (int)order[s];
    const uint32_t sh = shift[s];
    const double t = (double)i / ((double)samples - 1.0);
    const double T = time[s];
    double px = 0.0, py = 0.0, pz = 0.0;
    for (int k = 0; k <= n; k++) {
        const double c = binom(n, k), a = pct::pow_uint_cr(t, k), b = pct::pow_uint_cr(1.0 - t, n - k);
        px += T * cx[sh + k] * c * a * b;
        py += T * cy[sh + k] * c * a * b;
        pz += T * cz[sh + k] * c * a * b;
    }
    pos[3 * g] = px; pos[3 * g + 1] = py; pos[3 * g + 2] = pz;
}

// (p - last_p).norm() for every sample; the very first predecessor is coef_vec(0) * time[0] (:32 / :62)
__global__ __launch_bounds__(256) void wire_step_kernel(const double *__restrict__ pos, int64_t total, double x0, double y0, double z0,
                                                        double *__restrict__ step)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const double lx = g ? pos[3 * (g - 1)] : x0, ly = g ? pos[3 * (g - 1) + 1] : y0, lz = g ? pos[3 * (g - 1) + 2] : z0;
    const double dx = pos[3 * g] - lx, dy = pos[3 * g + 1] - ly, dz = pos[3 * g + 2] - lz;
    step[g] = sqrt((dx * dx + dy * dy) + dz * dz);
}

struct Workspace {
    double *d_a = nullptr;      // general fp64 scratch (inputs)
    size_t a_cap = 0;
    double *d_b = nullptr;      // general fp64 scratch (outputs)
    size_t b_cap = 0;
    int32_t *d_i = nullptr;     // int scratch
    size_t i_cap = 0;
} g_ws;

template <typename T>
int grow(T **p, size_t *cap, size_t need)
{
    if (need <= *cap) return PCT_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    void *v = nullptr;
    const size_t n = std::max<size_t>(need, 4096);
    if (hipMalloc(&v, n * sizeof(T)) != hipSuccess) return fail(PCT_ERR_ALLOC, "hipMalloc(%zu bytes) failed", n * sizeof(T));
    *p = static_cast<T *>(v);
    *cap = n;
    return PCT_OK;
}

int check_matrix_traj(const pct_bezier_traj *t)
{
    if (!t || !t->polycoef || !t->seg_time || !t->orders || t->nseg <= 0) return fail(PCT_ERR_INVALID, "bad trajectory");
    for (int s = 0; s < t->nseg; s++) {
        if (t->orders[s] < 1 || t->orders[s] > kMaxOrder) return fail(PCT_ERR_INVALID, "segment %d has order %d (supported: 1..%d)", s, t->orders[s], kMaxOrder);
        if (3 * (int64_t)(t->orders[s] + 1) > t->row_stride) return fail(PCT_ERR_INVALID, "row_stride %lld too small for order %d", (long long)t->row_stride, t->orders[s]);
    }
    return PCT_OK;
}

int check_wire(const pct_traj_wire *w, std::vector<uint32_t> &shift)
{
    if (!w || w->num_segment < 0) return fail(PCT_ERR_INVALID, "bad wire trajectory");
    if (w->num_segment > 0 && (!w->coef_x || !w->coef_y || !w->coef_z || !w->time || !w->order)) return fail(PCT_ERR_INVALID, "null wire array");
    shift.resize((size_t)w->num_segment);
    int64_t acc = 0;
    for (int s = 0; s < w->num_segment; s++) {
        if (w->order[s] < 1 || w->order[s] > (uint32_t)kMaxOrder) return fail(PCT_ERR_INVALID, "segment %d has order %u (supported: 1..%d)", s, w->order[s], kMaxOrder);
        shift[(size_t)s] = (uint32_t)acc;
        acc += w->order[s] + 1;
    }
    if (acc > w->ncoef) return fail(PCT_ERR_INVALID, "wire trajectory holds %lld control points, its orders need %lld", (long long)w->ncoef, (long long)acc);
    return PCT_OK;
}

// device samples of the whole wire trajectory: positions in g_ws.d_b[0 .. 3*total), step lengths behind them
int sample_wire_device(const pct_traj_wire *w, int32_t samples, int64_t *total_out, double **d_pos, double **d_step)
{
    std::vector<uint32_t> shift;
    PCTCHK(check_wire(w, shift));
    if (samples < 2) return fail(PCT_ERR_INVALID, "need at least 2 samples per segment");
    PCTCHK(pct_internal::require_init());
    hipStream_t s = pct_internal::stream();
    const int nseg = w->num_segment;
    const int64_t total = (int64_t)nseg * samples;
    *total_out = total;
    if (total == 0) return PCT_OK;
    // inputs: cx | cy | cz | time  (doubles), order | shift (u32)
    const size_t nc = (size_t)w->ncoef;
    PCTCHK(grow(&g_ws.d_a, &g_ws.a_cap, 3 * nc + (size_t)nseg));
    PCTCHK(grow(&g_ws.d_i, &g_ws.i_cap, 2 * (size_t)nseg));
    PCTCHK(grow(&g_ws.d_b, &g_ws.b_cap, 4 * (size_t)total));
    HIPCHK(hipMemcpyAsync(g_ws.d_a, w->coef_x, sizeof(double) * nc, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g_ws.d_a + nc, w->coef_y, sizeof(double) * nc, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g_ws.d_a + 2 * nc, w->coef_z, sizeof(double) * nc, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g_ws.d_a + 3 * nc, w->time, sizeof(double) * nseg, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g_ws.d_i, w->order, sizeof(uint32_t) * nseg, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g_ws.d_i + nseg, shift.data(), sizeof(uint32_t) * nseg, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));      // `shift` (and the caller's arrays) may go away
    double *pos = g_ws.d_b, *step = g_ws.d_b + 3 * total;
    const int nb = (int)((total + 255) / 256);
    wire_sample_kernel<<<nb, 256, 0, s>>>(g_ws.d_a, g_ws.d_a + nc, g_ws.d_a + 2 * nc, g_ws.d_a + 3 * nc,
                                          reinterpret_cast<const uint32_t *>(g_ws.d_i), reinterpret_cast<const uint32_t *>(g_ws.d_i + nseg),
                                          nseg, samples, pos);
    // last_p starts as coef_vec(traj_ext, 0) * traj_ext.time[0]
    wire_step_kernel<<<nb, 256, 0, s>>>(pos, total, w->coef_x[0] * w->time[0], w->coef_y[0] * w->time[0], w->coef_z[0] * w->time[0], step);
    HIPCHK(hipGetLastError());
    *d_pos = pos;
    *d_step = step;
    return PCT_OK;
}

}  // namespace

extern "C" {

int pct_bezier_state_batch(const pct_bezier_traj *traj, const int32_t *seg, const double *u, int64_t n, double *state9)
{
    PCTCHK(check_matrix_traj(traj));
    if (n < 0 || (n > 0 && (!seg || !u || !state9))) return fail(PCT_ERR_INVALID, "bad sample arrays");
    for (int64_t i = 0; i < n; i++)
        if (seg[i] < 0 || seg[i] >= traj->nseg) return fail(PCT_ERR_INVALID, "sample %lld names segment %d of %d", (long long)i, seg[i], traj->nseg);
    if (n == 0) return PCT_OK;
    PCTCHK(pct_internal::require_init());
    hipStream_t s = pct_internal::stream();
    const size_t ncoef = (size_t)traj->nseg * (size_t)traj->row_stride;
    PCTCHK(grow(&g_ws.d_a, &g_ws.a_cap, ncoef + (size_t)n));
    PCTCHK(grow(&g_ws.d_i, &g_ws.i_cap, (size_t)traj->nseg + (size_t)n));
    PCTCHK(grow(&g_ws.d_b, &g_ws.b_cap, 9 * (size_t)n));
    HIPCHK(hipMemcpyAsync(g_ws.d_a, traj->polycoef, sizeof(double) * ncoef, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g_ws.d_a + ncoef, u, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g_ws.d_i, traj->orders, sizeof(int32_t) * traj->nseg, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g_ws.d_i + traj->nseg, seg, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    bezier_state_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(g_ws.d_a, traj->row_stride, g_ws.d_i, g_ws.d_i + traj->nseg, g_ws.d_a + ncoef, n, g_ws.d_b);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(state9, g_ws.d_b, sizeof(double) * 9 * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return PCT_OK;
}

int pct_traj_wire_from_matrix(const pct_bezier_traj *traj, double *coef_x, double *coef_y, double *coef_z, int64_t cap, int64_t *ncoef)
{
    PCTCHK(check_matrix_traj(traj));
    int64_t total = 0;
    for (int s = 0; s < traj->nseg; s++) total += traj->orders[s] + 1;
    if (ncoef) *ncoef = total;
    if (!coef_x || !coef_y || !coef_z || cap < total) return fail(PCT_ERR_INVALID, "wire arrays need room for %lld control points", (long long)total);
    int64_t idx = 0;
    for (int s = 0; s < traj->nseg; s++) {           // sim_planning_demo.cpp:552-562
        const int m = traj->orders[s] + 1;
        const double *row = traj->polycoef + (int64_t)s * traj->row_stride;
        for (int j = 0; j < m; j++, idx++) { coef_x[idx] = row[j]; coef_y[idx] = row[m + j]; coef_z[idx] = row[2 * m + j]; }
    }
    return PCT_OK;
}

int pct_traj_wire_sample(const pct_traj_wire *w, int32_t samples, double *pos, double *step_len)
{
    int64_t total = 0;
    double *d_pos = nullptr, *d_step = nullptr;
    PCTCHK(sample_wire_device(w, samples, &total, &d_pos, &d_step));
    if (total == 0) return PCT_OK;
    hipStream_t s = pct_internal::stream();
    if (pos) HIPCHK(hipMemcpyAsync(pos, d_pos, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, s));
    if (step_len) HIPCHK(hipMemcpyAsync(step_len, d_step, sizeof(double) * total, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return PCT_OK;
}

int pct_traj_segm_index(const pct_traj_wire *w, double twirl_len, int32_t *segm, int32_t *part)
{
    if (!segm || !part) return fail(PCT_ERR_INVALID, "null output");
    const int32_t samples = 1001;
    int64_t total = 0;
    double *d_pos = nullptr, *d_step = nullptr;
    PCTCHK(sample_wire_device(w, samples, &total, &d_pos, &d_step));
    *segm = w->num_segment - 1;                       // traj_postprocessing.cpp:56
    *part = 1;
    if (total == 0) return PCT_OK;
    std::vector<double> step((size_t)total);
    hipStream_t s = pct_internal::stream();
    HIPCHK(hipMemcpyAsync(step.data(), d_step, sizeof(double) * total, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    double len = twirl_len;
    for (int64_t g = 0; g < total; g++) {             // :34-52, sequential on purpose (fp64 subtraction order)
        len -= step[(size_t)g];
        if (len < 0) {
            const double t = (double)(g % samples) / 1000.0;
            *segm = (int32_t)(g / samples);
            *part = t > 0.5 ? 1 : 0;
            return PCT_OK;
        }
    }
    return PCT_OK;
}

int pct_traj_nearest_voxels(const pct_traj_wire *w, double twirl_len, pct_voxel_map *m, int64_t *points_used)
{
    if (!m) return fail(PCT_ERR_INVALID, "null voxel map");
    if (points_used) *points_used = 0;
    PCTCHK(pct_voxel_map_clear(m));
    const int32_t samples = 1001;
    int64_t total = 0;
    double *d_pos = nullptr, *d_step = nullptr;
    PCTCHK(sample_wire_device(w, samples, &total, &d_pos, &d_step));
    if (total == 0) return PCT_OK;
    std::vector<double> step((size_t)total);
    hipStream_t s = pct_internal::stream();
    HIPCHK(hipMemcpyAsync(step.data(), d_step, sizeof(double) * total, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    // :64-78 -- both loops run while len >= 0: a sample is added, THEN its step is subtracted
    double len = twirl_len;
    int64_t used = 0;
    for (int64_t g = 0; g < total && len >= 0; g++) {
        used++;
        len -= step[(size_t)g];
    }
    if (points_used) *points_used = used;
    if (used == 0) return PCT_OK;
    return pct_voxel_map_add_dev(m, d_pos, used, 3 * sizeof(double), 1, nullptr, nullptr, nullptr);
}

int pct_traj_end_yaws(const double *path_x, const double *path_y, int64_t n, const double *coef_x, const double *coef_y, double *end_yaws)
{
    if (n < 0 || (n > 0 && (!path_x || !path_y || !end_yaws))) return fail(PCT_ERR_INVALID, "bad arguments");
    for (int64_t i = 0; i < n; i++) {                 // traj_postprocessing.cpp:154-177
        if (i < n - 1) {
            const double vx = path_x[i + 1] - path_x[i], vy = path_y[i + 1] - path_y[i];
            end_yaws[i] = std::sqrt(vx * vx + vy * vy) > 0.01 ? std::atan2(vy, vx) : 10.0;
        } else if (i > 0) {
            end_yaws[i] = end_yaws[i - 1];
        } else {
            if (!coef_x || !coef_y) return fail(PCT_ERR_INVALID, "a single-sphere corridor needs the first two control points");
            const double vx = coef_x[1] - coef_x[0], vy = coef_y[1] - coef_y[0];
            end_yaws[i] = std::sqrt(vx * vx + vy * vy) > 0.01 ? std::atan2(vy, vx) : 10.0;
        }
    }
    return PCT_OK;
}

int pct_debug_binomials(double *pascal, double *recurrence)
{
    if (!pascal || !recurrence) return fail(PCT_ERR_INVALID, "null output");
    double *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, sizeof(double) * 2 * 169));
    binomial_tables_kernel<<<1, 192>>>(d, d + 169);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(pascal, d, sizeof(double) * 169, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(recurrence, d + 169, sizeof(double) * 169, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PCT_ERR_HIP, "binomial tables: %s", hipGetErrorString(e));
    return PCT_OK;
}

}  // extern "C"
