// kernels.hpp -- gfx950 device code of the obstacle-cloud engine.
//
// Arithmetic contract (include/pct_engine.h): every distance is fp64 on float-widened
// operands, ((dx*dx + dy*dy) + dz*dz), one rounding per operation, NO fused multiply-add,
// so results are bit-identical to Utils/kdtree/src/kdtree.c:379-382 compiled for x86-64.
// The whole translation unit is built with -ffp-contract=off and this pragma repeats it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace pct {

constexpr int kWave = 64;
constexpr uint32_t kNoIndex = 0xFFFFFFFFu;

// ((dx^2 + dy^2) + dz^2), point minus query as in kdtree.c (the sign is squared away).
__device__ __forceinline__ double dist2(double px, double py, double pz, double qx, double qy, double qz)
{
    double dx = px - qx, dy = py - qy, dz = pz - qz;
    double s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return s;
}

// total order used everywhere a winner is picked: smaller d2, then lower index
__device__ __forceinline__ bool better(double d2a, uint32_t ia, double d2b, uint32_t ib)
{
    return d2a < d2b || (d2a == d2b && ia < ib);
}

__device__ __forceinline__ void wave_argmin(double &d, uint32_t &i)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double od = __shfl_xor(d, off, kWave);
        uint32_t oi = (uint32_t)__shfl_xor((int)i, off, kWave);
        if (better(od, oi, d, i)) { d = od; i = oi; }
    }
}

// =====================================================================================
// 1. Streaming kernels: lanes own POINTS, queries are wave-uniform (scalar registers).
//    HBM-bound for small query tiles: the SoA cloud is read exactly once per pass with
//    16-byte-per-lane loads (three 1-KiB wave transactions per 256 points).
// =====================================================================================

// q64: [Q][3] doubles (float-widened queries).  One pass handles queries q0 .. q0+QT-1.
// Output: per-block partial winners part_d2/part_idx[(q0+j) * nparts + block].
template <int QT>
__global__ __launch_bounds__(256) void nn_stream_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ z, uint32_t n,
                                                        const double *__restrict__ q64, int q0, int qcount,
                                                        double *__restrict__ part_d2, uint32_t *__restrict__ part_idx,
                                                        int nparts)
{
    double qx[QT], qy[QT], qz[QT];
#pragma unroll
    for (int j = 0; j < QT; j++) {
        // tiles past the end re-use the last query (results discarded); keeps loads uniform
        int qi = q0 + (j < qcount ? j : qcount - 1);
        qx[j] = q64[3 * qi + 0];
        qy[j] = q64[3 * qi + 1];
        qz[j] = q64[3 * qi + 2];
    }
    double bd[QT];
    uint32_t bi[QT];
#pragma unroll
    for (int j = 0; j < QT; j++) { bd[j] = __builtin_huge_val(); bi[j] = kNoIndex; }

    const uint32_t ngroups = n >> 2;
    const uint32_t stride = gridDim.x * blockDim.x;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const float4 *y4 = reinterpret_cast<const float4 *>(y);
    const float4 *z4 = reinterpret_cast<const float4 *>(z);

    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        const float4 X = x4[g], Y = y4[g], Z = z4[g];
        const float xs[4] = { X.x, X.y, X.z, X.w };
        const float ys[4] = { Y.x, Y.y, Y.z, Y.w };
        const float zs[4] = { Z.x, Z.y, Z.z, Z.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double px = (double)xs[k], py = (double)ys[k], pz = (double)zs[k];
            const uint32_t id = 4u * g + (uint32_t)k;
#pragma unroll
            for (int j = 0; j < QT; j++) {
                const double d2 = dist2(px, py, pz, qx[j], qy[j], qz[j]);
                // indices grow within a thread, so strict < keeps the lowest index on ties
                if (d2 < bd[j]) { bd[j] = d2; bi[j] = id; }
            }
        }
    }
    // tail (n % 4 points): one lane each, in block 0
    if (blockIdx.x == 0) {
        const uint32_t id = 4u * ngroups + threadIdx.x;
        if (threadIdx.x < (n & 3u)) {
            const double px = (double)x[id], py = (double)y[id], pz = (double)z[id];
#pragma unroll
            for (int j = 0; j < QT; j++) {
                const double d2 = dist2(px, py, pz, qx[j], qy[j], qz[j]);
                if (better(d2, id, bd[j], bi[j])) { bd[j] = d2; bi[j] = id; }
            }
        }
    }

    __shared__ double s_d[4][QT];
    __shared__ uint32_t s_i[4][QT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < QT; j++) {
        double d = bd[j];
        uint32_t i = bi[j];
        wave_argmin(d, i);
        if (lane == 0) { s_d[wave][j] = d; s_i[wave][j] = i; }
    }
    __syncthreads();
    if (threadIdx.x < QT && (int)threadIdx.x < qcount) {
        const int j = threadIdx.x;
        double d = s_d[0][j];
        uint32_t i = s_i[0][j];
#pragma unroll
        for (int w = 1; w < 4; w++)
            if (better(s_d[w][j], s_i[w][j], d, i)) { d = s_d[w][j]; i = s_i[w][j]; }
        part_d2[(size_t)(q0 + j) * nparts + blockIdx.x] = d;
        part_idx[(size_t)(q0 + j) * nparts + blockIdx.x] = i;
    }
}

// one wave per query folds the per-block partials; adds the shard's index base
__global__ __launch_bounds__(64) void nn_reduce_partials_kernel(const double *__restrict__ part_d2,
                                                                const uint32_t *__restrict__ part_idx, int nparts,
                                                                uint32_t index_base, uint32_t *__restrict__ out_idx,
                                                                double *__restrict__ out_d2)
{
    const int q = blockIdx.x;
    double d = __builtin_huge_val();
    uint32_t i = kNoIndex;
    for (int p = threadIdx.x; p < nparts; p += 64) {
        const double pd = part_d2[(size_t)q * nparts + p];
        const uint32_t pi = part_idx[(size_t)q * nparts + p];
        if (better(pd, pi, d, i)) { d = pd; i = pi; }
    }
    wave_argmin(d, i);
    if (threadIdx.x == 0) {
        out_idx[q] = (i == kNoIndex) ? kNoIndex : i + index_base;
        out_d2[q] = d;
    }
}

// radius count, same streaming shape.  r2[j] = (double)r * (double)r (kdtree.c:273).
template <int QT>
__global__ __launch_bounds__(256) void count_stream_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ z, uint32_t n,
                                                           const double *__restrict__ q64, const double *__restrict__ r2,
                                                           int q0, int qcount, uint32_t *__restrict__ count)
{
    double qx[QT], qy[QT], qz[QT], rr[QT];
#pragma unroll
    for (int j = 0; j < QT; j++) {
        int qi = q0 + (j < qcount ? j : qcount - 1);
        qx[j] = q64[3 * qi + 0];
        qy[j] = q64[3 * qi + 1];
        qz[j] = q64[3 * qi + 2];
        rr[j] = r2[qi];
    }
    uint32_t c[QT];
#pragma unroll
    for (int j = 0; j < QT; j++) c[j] = 0;

    const uint32_t ngroups = n >> 2;
    const uint32_t stride = gridDim.x * blockDim.x;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const float4 *y4 = reinterpret_cast<const float4 *>(y);
    const float4 *z4 = reinterpret_cast<const float4 *>(z);
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        const float4 X = x4[g], Y = y4[g], Z = z4[g];
        const float xs[4] = { X.x, X.y, X.z, X.w };
        const float ys[4] = { Y.x, Y.y, Y.z, Y.w };
        const float zs[4] = { Z.x, Z.y, Z.z, Z.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double px = (double)xs[k], py = (double)ys[k], pz = (double)zs[k];
#pragma unroll
            for (int j = 0; j < QT; j++) c[j] += dist2(px, py, pz, qx[j], qy[j], qz[j]) <= rr[j] ? 1u : 0u;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
        const uint32_t id = 4u * ngroups + threadIdx.x;
        const double px = (double)x[id], py = (double)y[id], pz = (double)z[id];
#pragma unroll
        for (int j = 0; j < QT; j++) c[j] += dist2(px, py, pz, qx[j], qy[j], qz[j]) <= rr[j] ? 1u : 0u;
    }
    __shared__ uint32_t s_c[4][QT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < QT; j++) {
        uint32_t v = c[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, kWave);
        if (lane == 0) s_c[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < QT && (int)threadIdx.x < qcount) {
        const int j = threadIdx.x;
        const uint32_t v = s_c[0][j] + s_c[1][j] + s_c[2][j] + s_c[3][j];
        if (v) atomicAdd(&count[q0 + j], v);
    }
}

// lidar crop: indices within r of ONE centre.  Two passes share this kernel: flags -> scan is
// avoided by a wave-aggregated atomic cursor; the host sorts the (unordered) result ascending.
__global__ __launch_bounds__(256) void radius_collect_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, uint32_t n, double qx, double qy,
                                                             double qz, double r2, uint32_t index_base,
                                                             uint32_t *__restrict__ out, uint32_t cap,
                                                             uint32_t *__restrict__ cursor)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool hit = dist2((double)x[i], (double)y[i], (double)z[i], qx, qy, qz) <= r2;
        if (hit) {
            const uint32_t pos = atomicAdd(cursor, 1u);
            if (pos < cap) out[pos] = i + index_base;
        }
    }
}

// =====================================================================================
// 2. Host-layout plumbing
// =====================================================================================

// AoS(stride) staging buffer -> SoA slots [dst0, dst0+n) (ring wrap handled by the caller
// issuing two launches).  Each thread moves one point.
__global__ __launch_bounds__(256) void deinterleave_kernel(const unsigned char *__restrict__ aos, uint32_t stride_bytes,
                                                           uint32_t n, float *__restrict__ x, float *__restrict__ y,
                                                           float *__restrict__ z, uint32_t dst0)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = reinterpret_cast<const float *>(aos + (size_t)i * stride_bytes);
    x[dst0 + i] = p[0];
    y[dst0 + i] = p[1];
    z[dst0 + i] = p[2];
}

// packed xyz (12 B) fast path: 4 points = 3 float4 loads per lane -> three float4 stores
__global__ __launch_bounds__(256) void deinterleave12_kernel(const float4 *__restrict__ aos4, uint32_t ngroups,
                                                             float4 *__restrict__ x4, float4 *__restrict__ y4,
                                                             float4 *__restrict__ z4)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const float4 a = aos4[3 * g], b = aos4[3 * g + 1], c = aos4[3 * g + 2];
    x4[g] = make_float4(a.x, a.w, b.z, c.y);
    y4[g] = make_float4(a.y, b.x, b.w, c.z);
    z4[g] = make_float4(a.z, b.y, c.x, c.w);
}

__global__ __launch_bounds__(256) void widen_queries_kernel(const float *__restrict__ q, uint32_t n3,
                                                            double *__restrict__ q64)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) q64[i] = (double)q[i];
}

__global__ __launch_bounds__(256) void square_radii_kernel(const float *__restrict__ r, uint32_t n, double *__restrict__ r2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const double w = (double)r[i]; r2[i] = w * w; }
}

__global__ __launch_bounds__(256) void fill_empty_kernel(uint32_t *__restrict__ idx, double *__restrict__ d2, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { idx[i] = kNoIndex; d2[i] = __builtin_huge_val(); }
}

// =====================================================================================
// 3. Uniform-cell index: counting sort of the cloud into cells (x fastest, then y, then z)
// =====================================================================================
struct GridDesc {
    float ox, oy, oz, inv_h;   // fp32 cell assignment (points and queries)
    double oxd, oyd, ozd, hd;  // fp64 face positions for the termination bound
    int gx, gy, gz;
    uint32_t ncells;
};

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int g)
{
    float t = floorf((v - o) * inv_h);
    t = fminf(fmaxf(t, 0.0f), (float)(g - 1));
    return (int)t;
}

__device__ __forceinline__ uint32_t cell_lin(const GridDesc &G, int cx, int cy, int cz)
{
    return ((uint32_t)cz * (uint32_t)G.gy + (uint32_t)cy) * (uint32_t)G.gx + (uint32_t)cx;
}

// per-block min/max -> partials[block][6]
__global__ __launch_bounds__(256) void bbox_partial_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ z, uint32_t n,
                                                           float *__restrict__ partials)
{
    float lo[3] = { __builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf() };
    float hi[3] = { -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf() };
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v[3] = { x[i], y[i], z[i] };
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], v[k]); hi[k] = fmaxf(hi[k], v[k]); }
    }
    __shared__ float s[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, kWave));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, kWave));
        }
        if (lane == 0) { s[wave][k] = lo[k]; s[wave][3 + k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = s[0][k];
        for (int w = 1; w < 4; w++) v = (k < 3) ? fminf(v, s[w][k]) : fmaxf(v, s[w][k]);
        partials[blockIdx.x * 6 + k] = v;
    }
}

__global__ __launch_bounds__(256) void cell_histogram_kernel(GridDesc G, const float *__restrict__ x,
                                                             const float *__restrict__ y, const float *__restrict__ z,
                                                             uint32_t n, uint32_t *__restrict__ cell_count,
                                                             uint32_t *__restrict__ point_cell)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t c = cell_lin(G, cell_coord(x[i], G.ox, G.inv_h, G.gx), cell_coord(y[i], G.oy, G.inv_h, G.gy),
                                    cell_coord(z[i], G.oz, G.inv_h, G.gz));
        point_cell[i] = c;
        atomicAdd(&cell_count[c], 1u);
    }
}

// exclusive scan, three launches: per-block scan of 1024-element tiles, scan of the tile sums
// (single block), then add.  cell_start has ncells+1 entries; entry ncells = n.
constexpr int kScanTile = 1024;

__global__ __launch_bounds__(256) void scan_tiles_kernel(const uint32_t *__restrict__ in, uint32_t n,
                                                         uint32_t *__restrict__ out, uint32_t *__restrict__ tile_sum)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (base + k < n) ? in[base + k] : 0u;
    const uint32_t tsum = v[0] + v[1] + v[2] + v[3];
    // inclusive wave scan of per-thread sums
    uint32_t inc = tsum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off, kWave);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += s_wave[w];
    uint32_t run = wave_off + inc - tsum;   // exclusive prefix of this thread within the tile
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255) tile_sum[blockIdx.x] = wave_off + inc;
}

// single block: exclusive scan of tile sums in place (ntiles arbitrary; serial over chunks of 256)
__global__ __launch_bounds__(256) void scan_tile_sums_kernel(uint32_t *__restrict__ tile_sum, uint32_t ntiles)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < ntiles; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < ntiles) ? tile_sum[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off, kWave);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wave_off = s_carry;
        for (int w = 0; w < wave; w++) wave_off += s_wave[w];
        if (i < ntiles) tile_sum[i] = wave_off + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = wave_off + inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *__restrict__ out, uint32_t n,
                                                       const uint32_t *__restrict__ tile_sum, uint32_t total)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += tile_sum[i / kScanTile];
    if (i == 0) out[n] = total;
}

// sorted[pos] = (x, y, z, bit-cast original index); order inside a cell is arbitrary, which
// is harmless because every consumer picks winners by (d2, index) or counts.
__global__ __launch_bounds__(256) void cell_scatter_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ z, uint32_t n,
                                                           const uint32_t *__restrict__ point_cell,
                                                           const uint32_t *__restrict__ cell_start,
                                                           uint32_t *__restrict__ cell_fill, float4 *__restrict__ sorted)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t c = point_cell[i];
        const uint32_t pos = cell_start[c] + atomicAdd(&cell_fill[c], 1u);
        sorted[pos] = make_float4(x[i], y[i], z[i], __uint_as_float(i));
    }
}

// =====================================================================================
// 4. Cell-pruned kernels: one lane per query walks an expanding cube of cells.
//    Termination is exact: a point outside the scanned cube of cells is at least
//    `bound` away (distance to the cube's faces, minus a slack covering the fp32 cell
//    assignment rounding), so the search stops once best_d2 <= bound^2.
// =====================================================================================
struct WorkCounters { unsigned long long points, cells; };

template <bool COUNT>
__device__ __forceinline__ void scan_run(const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                         uint32_t lin0, uint32_t lin1, double qx, double qy, double qz, double &bd,
                                         uint32_t &bi, uint32_t &npts, uint32_t &nruns)
{
    const uint32_t s = cell_start[lin0], e = cell_start[lin1 + 1];
    if (COUNT) { npts += e - s; nruns += 1; }
    for (uint32_t p = s; p < e; p++) {
        const float4 P = pts[p];
        const double d2 = dist2((double)P.x, (double)P.y, (double)P.z, qx, qy, qz);
        const uint32_t id = __float_as_uint(P.w);
        if (better(d2, id, bd, bi)) { bd = d2; bi = id; }
    }
}

template <bool COUNT>
__global__ __launch_bounds__(256) void nn_grid_kernel(GridDesc G, const float4 *__restrict__ pts,
                                                      const uint32_t *__restrict__ cell_start,
                                                      const float *__restrict__ q, uint32_t Q, uint32_t index_base,
                                                      uint32_t *__restrict__ out_idx, double *__restrict__ out_d2,
                                                      WorkCounters *__restrict__ work)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t npts = 0, nruns = 0;
    if (t < Q) {
        const float qxf = q[3 * t], qyf = q[3 * t + 1], qzf = q[3 * t + 2];
        const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
        const int cx = cell_coord(qxf, G.ox, G.inv_h, G.gx);
        const int cy = cell_coord(qyf, G.oy, G.inv_h, G.gy);
        const int cz = cell_coord(qzf, G.oz, G.inv_h, G.gz);
        double bd = __builtin_huge_val();
        uint32_t bi = kNoIndex;
        const double slack = G.hd * (1.0 / 256.0);
        for (int r = 1;; r++) {
            const int x0 = max(cx - r, 0), x1 = min(cx + r, G.gx - 1);
            const int y0 = max(cy - r, 0), y1 = min(cy + r, G.gy - 1);
            const int z0 = max(cz - r, 0), z1 = min(cz + r, G.gz - 1);
            for (int zz = z0; zz <= z1; zz++) {
                const bool zface = (zz == cz - r) || (zz == cz + r);
                for (int yy = y0; yy <= y1; yy++) {
                    const uint32_t row = cell_lin(G, 0, yy, zz);
                    if (r == 1 || zface || yy == cy - r || yy == cy + r) {
                        scan_run<COUNT>(pts, cell_start, row + x0, row + x1, qx, qy, qz, bd, bi, npts, nruns);
                    } else {
                        if (cx - r >= 0) scan_run<COUNT>(pts, cell_start, row + cx - r, row + cx - r, qx, qy, qz, bd, bi, npts, nruns);
                        if (cx + r <= G.gx - 1) scan_run<COUNT>(pts, cell_start, row + cx + r, row + cx + r, qx, qy, qz, bd, bi, npts, nruns);
                    }
                }
            }
            // distance from q to the nearest face of the scanned cube that still has cells behind it
            double bound = __builtin_huge_val();
            if (cx - r > 0) bound = fmin(bound, qx - (G.oxd + (double)(cx - r) * G.hd));
            if (cx + r < G.gx - 1) bound = fmin(bound, (G.oxd + (double)(cx + r + 1) * G.hd) - qx);
            if (cy - r > 0) bound = fmin(bound, qy - (G.oyd + (double)(cy - r) * G.hd));
            if (cy + r < G.gy - 1) bound = fmin(bound, (G.oyd + (double)(cy + r + 1) * G.hd) - qy);
            if (cz - r > 0) bound = fmin(bound, qz - (G.ozd + (double)(cz - r) * G.hd));
            if (cz + r < G.gz - 1) bound = fmin(bound, (G.ozd + (double)(cz + r + 1) * G.hd) - qz);
            if (bound == __builtin_huge_val()) break;          // the cube covers the whole grid
            bound -= slack;
            if (bound > 0.0 && bd <= bound * bound) break;
        }
        out_idx[t] = (bi == kNoIndex) ? kNoIndex : bi + index_base;
        out_d2[t] = bd;
    }
    if (COUNT) {
        unsigned long long a = npts, b = nruns;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += (unsigned long long)__shfl_xor((long long)a, off, kWave);
            b += (unsigned long long)__shfl_xor((long long)b, off, kWave);
        }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&work->points, a); atomicAdd(&work->cells, b); }
    }
}

// radius count through the grid: every cell row overlapping the ball's bounding box
// (one extra cell of margin for the fp32 cell assignment) is scanned.
template <bool COUNT>
__global__ __launch_bounds__(256) void count_grid_kernel(GridDesc G, const float4 *__restrict__ pts,
                                                         const uint32_t *__restrict__ cell_start,
                                                         const float *__restrict__ q, const float *__restrict__ rad,
                                                         uint32_t Q, uint32_t *__restrict__ count,
                                                         WorkCounters *__restrict__ work)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t npts = 0, nruns = 0;
    if (t < Q) {
        const float qxf = q[3 * t], qyf = q[3 * t + 1], qzf = q[3 * t + 2], rf = rad[t];
        const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
        const double r2 = (double)rf * (double)rf;
        uint32_t c = 0;
        if (rf >= 0.0f) {
            const float pad = rf + 0.01f * (1.0f / G.inv_h);
            int x0 = cell_coord(qxf - pad, G.ox, G.inv_h, G.gx), x1 = cell_coord(qxf + pad, G.ox, G.inv_h, G.gx);
            int y0 = cell_coord(qyf - pad, G.oy, G.inv_h, G.gy), y1 = cell_coord(qyf + pad, G.oy, G.inv_h, G.gy);
            int z0 = cell_coord(qzf - pad, G.oz, G.inv_h, G.gz), z1 = cell_coord(qzf + pad, G.oz, G.inv_h, G.gz);
            x0 = max(x0 - 1, 0); x1 = min(x1 + 1, G.gx - 1);
            y0 = max(y0 - 1, 0); y1 = min(y1 + 1, G.gy - 1);
            z0 = max(z0 - 1, 0); z1 = min(z1 + 1, G.gz - 1);
            for (int zz = z0; zz <= z1; zz++)
                for (int yy = y0; yy <= y1; yy++) {
                    const uint32_t row = cell_lin(G, 0, yy, zz);
                    const uint32_t s = cell_start[row + x0], e = cell_start[row + x1 + 1];
                    if (COUNT) { npts += e - s; nruns += 1; }
                    for (uint32_t p = s; p < e; p++) {
                        const float4 P = pts[p];
                        c += dist2((double)P.x, (double)P.y, (double)P.z, qx, qy, qz) <= r2 ? 1u : 0u;
                    }
                }
        }
        count[t] = c;
    }
    if (COUNT) {
        unsigned long long a = npts, b = nruns;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += (unsigned long long)__shfl_xor((long long)a, off, kWave);
            b += (unsigned long long)__shfl_xor((long long)b, off, kWave);
        }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&work->points, a); atomicAdd(&work->cells, b); }
    }
}

// =====================================================================================
// 5. Planner arithmetic around the NN: sphere inflation and the sampled Bezier check
// =====================================================================================
struct InflateParams { double sx, sy, sz, sample_range, search_margin, max_radius; };

// corridor_finder.cpp:113-126: early-out test in fp64 on the planner's Vector3d, then the
// query is narrowed to fp32 (searchPoint.x = search_Pt(0)).  skip[i] = 1 when the early-out fires.
__global__ __launch_bounds__(256) void inflate_prologue_kernel(InflateParams P, const double *__restrict__ pts,
                                                               uint32_t n, float *__restrict__ q,
                                                               unsigned char *__restrict__ skip)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const double dx = px - P.sx, dy = py - P.sy, dz = pz - P.sz;
    const double dist = sqrt(dx * dx + dy * dy + dz * dz);     // getDis, :109-111
    skip[i] = dist > P.sample_range + P.max_radius ? 1 : 0;
    q[3 * i] = (float)px;
    q[3 * i + 1] = (float)py;
    q[3 * i + 2] = (float)pz;
}

// corridor_finder.cpp:130-132: r = sqrt(d2) - search_margin, min(r, max_radius);
// early-out rows get max_radius - search_margin, idx = none, d2 = +inf.
__global__ __launch_bounds__(256) void inflate_epilogue_kernel(InflateParams P, uint32_t n,
                                                               const unsigned char *__restrict__ skip, int cloud_empty,
                                                               uint32_t *__restrict__ idx, double *__restrict__ d2,
                                                               double *__restrict__ radius)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (skip[i] || cloud_empty) {
        radius[i] = P.max_radius - P.search_margin;
        idx[i] = kNoIndex;
        d2[i] = __builtin_huge_val();
        return;
    }
    const double r = sqrt(d2[i]) - P.search_margin;
    radius[i] = r < P.max_radius ? r : P.max_radius;
}

constexpr int kMaxBezierOrder = 12;

struct BezierDesc {
    const double *coef;      // device copy of PolyCoeff, nseg x row_stride
    const double *seg_time;  // device
    const int *orders;       // device
    int row_stride, nseg;
    double t_start, stop_time, dt;
    int cap;
};

// sim_planning_demo.cpp:729-771.  Thread 0 enumerates the sample times with the same
// sequential additions as the reference's nested loops; then one thread per sample
// evaluates getPosFromBezier (:715-727): acc += C(n,j) * c * pow(u,j) * pow(1-u,n-j), j ascending,
// scaled by the segment time (:752-753).  Outputs: pos (fp64) and nsamples.
__global__ __launch_bounds__(256) void bezier_samples_kernel(BezierDesc B, double *__restrict__ pos,
                                                             int *__restrict__ nsamples)
{
    extern __shared__ unsigned char smem[];
    double *s_t = reinterpret_cast<double *>(smem);
    int *s_seg = reinterpret_cast<int *>(s_t + B.cap);
    __shared__ int s_n;
    if (threadIdx.x == 0) {
        double t_s = B.t_start;
        int idx;
        for (idx = 0; idx < B.nseg; ++idx) {
            if (t_s > B.seg_time[idx] && idx + 1 < B.nseg) t_s -= B.seg_time[idx];
            else break;
        }
        int n = 0;
        double t_accu = 0.0;
        for (int i = idx; i < B.nseg; i++) {
            const double T = B.seg_time[i];
            for (double t = (i == idx) ? t_s : 0.0; t < T; t += B.dt) {
                t_accu += B.dt;
                if (t_accu > B.stop_time) break;
                if (n < B.cap) { s_t[n] = t; s_seg[n] = i; }
                n++;
            }
        }
        s_n = n;
        *nsamples = n;
    }
    __syncthreads();
    const int n = min(s_n, B.cap);
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        const int seg = s_seg[s];
        const int order = B.orders[seg], m = order + 1;
        const double T = B.seg_time[seg];
        const double u = s_t[s] / T;
        const double *c = B.coef + (size_t)seg * B.row_stride;
        // binomials as exact doubles (bezier_base.cpp:33-48 computes them with integer factorials)
        double binom[kMaxBezierOrder + 1];
        binom[0] = 1.0;
        for (int j = 1; j <= order; j++) binom[j] = floor(binom[j - 1] * (double)(order - j + 1) / (double)j + 0.5);
        for (int d = 0; d < 3; d++) {
            double acc = 0.0;
            for (int j = 0; j < m; j++) acc += binom[j] * c[d * m + j] * pow(u, (double)j) * pow(1.0 - u, (double)(order - j));
            pos[3 * s + d] = acc * T;
        }
    }
}

// first sample with negative radius (checkTrajPtCol, corridor_finder.cpp:412-416); -1 if none
__global__ __launch_bounds__(256) void first_hit_kernel(const double *__restrict__ radius, const int *__restrict__ nsamples,
                                                        int cap, long long *__restrict__ first_hit)
{
    __shared__ int s_min;
    if (threadIdx.x == 0) s_min = 0x7FFFFFFF;
    __syncthreads();
    const int n = min(*nsamples, cap);
    int best = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (radius[i] < 0.0 && i < best) best = i;
    atomicMin(&s_min, best);
    __syncthreads();
    if (threadIdx.x == 0) *first_hit = (s_min == 0x7FFFFFFF) ? -1ll : (long long)s_min;
}

}  // namespace pct
