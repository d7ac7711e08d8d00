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

#include "bernstein.hpp"

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

// -------------------------------------------------------------------------------------
// fp32 filter + exact fp64 recheck, LDS-staged (the default brute-force path for Q > 4).
//
// Measured on MI355X: an fp64 vector op costs ~8 cycles per wave and a plain
// fp32 op ~4; only the packed v_pk_{add,mul,fma}_f32 forms reach the fp32 peak (2 lanes-worth per
// 4 cycles).  The all-fp64 kernel above needs 12 slow ops per (point, query) pair and tops out at
// 1.8e12 pairs/s.  Exactness only matters for the few points that can win, so:
//   1. nn_sample_bounds_kernel: packed-fp32 minimum over a 1/16 sample of the cloud per query.
//      ANY upper bound of the true minimum is a valid threshold; the sample only makes it tight.
//   2. nn_tile_filter_kernel: each block stages its contiguous chunk of the SoA cloud in LDS
//      ONCE (HBM is read once per launch whatever Q is), then walks the query batch in tiles of
//      QT wave-uniform queries.  Every pair is evaluated in packed fp32 on two points at a time
//      (per 4 points and query: 6 pk_add, 2 pk_mul, 4 pk_fma, min3+min, one compare).  When the
//      smallest of the four fp32 distances is <= thr = bound * (1 + 2^-19) + 2^-90 the group is
//      re-evaluated in the exact fp64 arithmetic and competes by (d2, index).
// Why nothing is lost: the fp32 value d32 of a pair differs from the exact d2 by < 4e-7 relative
// (correctly rounded differences, three non-negative products, two sums: no cancellation) plus
// underflow (< 2^-120 absolute).  The true winner w satisfies d2(w) <= d2(p) for the sample point
// p that set the bound, so d32(w) <= d2(w)(1+e) <= d2(p)(1+e) <= d32(p)(1+e)/(1-e) < thr.  Hence w
// and every exact tie of it always reach the fp64 path: results equal nn_stream_kernel's bit for bit.
// -------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kSampleStride = 16;     // sample one 1024-point chunk out of every 16
constexpr int kTileQ = 8;             // wave-uniform queries per tile (7 scalar registers each)
constexpr int kSampleGroups = 4;      // sampled 256-group chunks per block of the bound kernel

struct PointGroup { v2f x01, x23, y01, y23, z01, z23; };

__device__ __forceinline__ PointGroup make_group(const float4 X, const float4 Y, const float4 Z)
{
    PointGroup g;
    g.x01 = v2f{ X.x, X.y }; g.x23 = v2f{ X.z, X.w };
    g.y01 = v2f{ Y.x, Y.y }; g.y23 = v2f{ Y.z, Y.w };
    g.z01 = v2f{ Z.x, Z.y }; g.z23 = v2f{ Z.z, Z.w };
    return g;
}

// smallest fp32 squared distance from the 4 points of g to (qx,qy,qz): 12 packed ops + min3 + min
__device__ __forceinline__ float group_min_d32(const PointGroup &g, float qx, float qy, float qz)
{
    const v2f qx2 = v2f{ qx, qx }, qy2 = v2f{ qy, qy }, qz2 = v2f{ qz, qz };
    const v2f dx01 = g.x01 - qx2, dx23 = g.x23 - qx2;
    const v2f dy01 = g.y01 - qy2, dy23 = g.y23 - qy2;
    const v2f dz01 = g.z01 - qz2, dz23 = g.z23 - qz2;
    v2f d01 = dx01 * dx01, d23 = dx23 * dx23;
    d01 = __builtin_elementwise_fma(dy01, dy01, d01);
    d23 = __builtin_elementwise_fma(dy23, dy23, d23);
    d01 = __builtin_elementwise_fma(dz01, dz01, d01);
    d23 = __builtin_elementwise_fma(dz23, dz23, d23);
    return fminf(fminf(d01.x, d01.y), fminf(d23.x, d23.y));
}

// Block-wide minimum of QT floats per thread through LDS (one transposed pass + a 32-lane
// butterfly instead of 6 cross-lane steps per value).  s_red: QT*256 floats.  Result for query j
// is returned by threads with (tid >> 5) == j, lane bit pattern (tid & 31) == 0.
template <int QT>
__device__ __forceinline__ float block_min_transposed(const float (&m)[QT], float *s_red)
{
    static_assert(QT <= 8, "one 32-lane group per query");
#pragma unroll
    for (int j = 0; j < QT; j++) s_red[j * 256 + threadIdx.x] = m[j];
    __syncthreads();
    const int j = threadIdx.x >> 5, l = threadIdx.x & 31;
    float v = __builtin_huge_valf();
    if (j < QT) {
#pragma unroll
        for (int i = 0; i < 8; i++) v = fminf(v, s_red[j * 256 + l + 32 * i]);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, kWave));
    __syncthreads();
    return v;
}

// grid = sampled chunks / kSampleGroups; block b looks at kSampleGroups chunks of 256 groups, one
// every `sample_stride` chunks (the per-tile block reduction is amortised over them).  One launch
// covers the whole query batch: bound_part[q * nsblocks + b].
__global__ __launch_bounds__(256) void nn_sample_bounds_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                               const float *__restrict__ z, uint32_t n, uint32_t sample_stride,
                                                               const float *__restrict__ qf, int Q, int qslice,
                                                               float *__restrict__ bound_part, int nsblocks)
{
    __shared__ float s_red[kTileQ * 256];
    const uint32_t ngroups = n >> 2;
    const float big = __builtin_huge_valf();   // filler for slots past the end: its d32 is +inf and never lowers a bound
    PointGroup pg[kSampleGroups];
#pragma unroll
    for (int c = 0; c < kSampleGroups; c++) {
        const uint32_t g = (blockIdx.x * kSampleGroups + c) * sample_stride * 256u + threadIdx.x;
        pg[c] = make_group(make_float4(big, big, big, big), make_float4(big, big, big, big), make_float4(big, big, big, big));
        if (g < ngroups) pg[c] = make_group(reinterpret_cast<const float4 *>(x)[g], reinterpret_cast<const float4 *>(y)[g],
                                            reinterpret_cast<const float4 *>(z)[g]);
    }
    // blockIdx.y takes a slice of the batch (a multiple of kTileQ queries): small clouds have few point blocks, and the
    // tile loop below is sequential, so the batch is what fills the chip (4096 queries on 1 M points: 0.8 -> 0.1 ms)
    const int q_end = min(Q, ((int)blockIdx.y + 1) * qslice);
    for (int q0 = (int)blockIdx.y * qslice; q0 < q_end; q0 += kTileQ) {
        const int qcount = min(kTileQ, q_end - q0);
        float m[kTileQ];
#pragma unroll
        for (int j = 0; j < kTileQ; j++) {
            const int qi = q0 + (j < qcount ? j : qcount - 1);
            const float qx = qf[3 * qi], qy = qf[3 * qi + 1], qz = qf[3 * qi + 2];
            float v = group_min_d32(pg[0], qx, qy, qz);
#pragma unroll
            for (int c = 1; c < kSampleGroups; c++) v = fminf(v, group_min_d32(pg[c], qx, qy, qz));
            m[j] = v;
        }
        const float v = block_min_transposed<kTileQ>(m, s_red);
        const int j = threadIdx.x >> 5;
        if ((threadIdx.x & 31) == 0 && j < qcount) bound_part[(size_t)(q0 + j) * nsblocks + blockIdx.x] = v;
    }
}

// one block per query: fold the sample partials into the fp32 bound (bit pattern; FLT_MAX when
// the sample was empty, which makes the filter recheck everything)
__global__ __launch_bounds__(256) void bound_reduce_kernel(const float *__restrict__ bound_part, int nsblocks,
                                                           uint32_t *__restrict__ bound_bits)
{
    const float *p = bound_part + (size_t)blockIdx.x * nsblocks;
    float v = 3.402823466e+38f;
    for (int i = threadIdx.x; i < nsblocks; i += 256) v = fminf(v, p[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, kWave));
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) bound_bits[blockIdx.x] = __float_as_uint(fminf(fminf(s[0], s[1]), fminf(s[2], s[3])));
}

// grid = chunks of `chunk_groups` 4-point groups; dynamic LDS = 3 * chunk_groups float4.
// Queries [qbase, qbase+Q) of the batch; partials at part[(q - qbase) * nparts + block].
__global__ __launch_bounds__(256) void nn_tile_filter_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, uint32_t n, uint32_t chunk_groups,
                                                             const float *__restrict__ qf, const double *__restrict__ q64,
                                                             const uint32_t *__restrict__ bound_bits, int qbase, int Q,
                                                             double *__restrict__ part_d2, uint32_t *__restrict__ part_idx,
                                                             int nparts)
{
    extern __shared__ float4 s_pts[];                 // [3][chunk_groups]
    __shared__ double s_d[4][kTileQ];
    __shared__ uint32_t s_i[4][kTileQ];
    const uint32_t ngroups = n >> 2;
    const uint32_t g0 = blockIdx.x * chunk_groups;
    const uint32_t ng = min(chunk_groups, ngroups > g0 ? ngroups - g0 : 0u);
    float4 *sx = s_pts, *sy = s_pts + chunk_groups, *sz = s_pts + 2 * chunk_groups;
    for (uint32_t i = threadIdx.x; i < ng; i += 256) {
        sx[i] = reinterpret_cast<const float4 *>(x)[g0 + i];
        sy[i] = reinterpret_cast<const float4 *>(y)[g0 + i];
        sz[i] = reinterpret_cast<const float4 *>(z)[g0 + i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool tail_owner = (blockIdx.x == gridDim.x - 1) && threadIdx.x < (n & 3u);   // n % 4 leftover points

    for (int q0 = 0; q0 < Q; q0 += kTileQ) {
        const int qcount = min(kTileQ, Q - q0);
        float qx[kTileQ], qy[kTileQ], qz[kTileQ], thr[kTileQ];
#pragma unroll
        for (int j = 0; j < kTileQ; j++) {
            const int qi = qbase + q0 + (j < qcount ? j : qcount - 1);
            qx[j] = qf[3 * qi]; qy[j] = qf[3 * qi + 1]; qz[j] = qf[3 * qi + 2];
            thr[j] = __uint_as_float(bound_bits[qi]) * (1.0f + 0x1p-19f) + 0x1p-90f;   // FLT_MAX bound -> +inf: recheck everything
        }
        double bd[kTileQ];
        uint32_t bi[kTileQ];
#pragma unroll
        for (int j = 0; j < kTileQ; j++) { bd[j] = __builtin_huge_val(); bi[j] = kNoIndex; }

        for (uint32_t i = threadIdx.x; i < ng; i += 256) {
            const float4 X = sx[i], Y = sy[i], Z = sz[i];
            const PointGroup pg = make_group(X, Y, Z);
            // the compare results stay in scalar registers (one s_or per query, no per-lane flags)
            unsigned long long hit = 0ull;
#pragma unroll
            for (int j = 0; j < kTileQ; j++) hit |= __builtin_amdgcn_ballot_w64(group_min_d32(pg, qx[j], qy[j], qz[j]) <= thr[j]);
            // wave-uniform branch: taken only when some lane holds a possible winner for some query
            if (hit != 0ull) {
                if ((hit >> lane) & 1ull) {
                    const float xs[4] = { X.x, X.y, X.z, X.w }, ys[4] = { Y.x, Y.y, Y.z, Y.w }, zs[4] = { Z.x, Z.y, Z.z, Z.w };
#pragma unroll
                    for (int j = 0; j < kTileQ; j++) {
                        const int qi = qbase + q0 + (j < qcount ? j : qcount - 1);
                        const double Qx = q64[3 * qi], Qy = q64[3 * qi + 1], Qz = q64[3 * qi + 2];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const double d2 = dist2((double)xs[k], (double)ys[k], (double)zs[k], Qx, Qy, Qz);
                            if (d2 < bd[j]) { bd[j] = d2; bi[j] = 4u * (g0 + i) + (uint32_t)k; }   // ids grow within a thread
                        }
                    }
                }
            }
        }
        if (tail_owner) {
            const uint32_t id = 4u * ngroups + threadIdx.x;
            const double px = (double)x[id], py = (double)y[id], pz = (double)z[id];
#pragma unroll
            for (int j = 0; j < kTileQ; j++) {
                const int qi = qbase + q0 + (j < qcount ? j : qcount - 1);
                const double d2 = dist2(px, py, pz, q64[3 * qi], q64[3 * qi + 1], q64[3 * qi + 2]);
                if (better(d2, id, bd[j], bi[j])) { bd[j] = d2; bi[j] = id; }
            }
        }
        // block winner per query; almost every lane holds (+inf, none), so skip the exchange then
#pragma unroll
        for (int j = 0; j < kTileQ; j++) {
            double d = bd[j];
            uint32_t ix = bi[j];
            if (__builtin_amdgcn_ballot_w64(ix != kNoIndex) != 0ull) wave_argmin(d, ix);
            if (lane == 0) { s_d[wave][j] = d; s_i[wave][j] = ix; }
        }
        __syncthreads();
        if ((int)threadIdx.x < qcount) {
            const int j = threadIdx.x;
            double d = s_d[0][j];
            uint32_t ix = s_i[0][j];
#pragma unroll
            for (int w = 1; w < 4; w++)
                if (better(s_d[w][j], s_i[w][j], d, ix)) { d = s_d[w][j]; ix = s_i[w][j]; }
            part_d2[(size_t)(q0 + j) * nparts + blockIdx.x] = d;
            part_idx[(size_t)(q0 + j) * nparts + blockIdx.x] = ix;
        }
        __syncthreads();
    }
}

// Candidate-list form of the filter (default).  The kernel above keeps a running (d2, index) per lane and query and folds it
// per block and per tile of 8 queries -- two __syncthreads, a wave argmin and 8 partial writes for every tile, which at
// 4096 queries is 512 block reductions per block and, for clouds under a few million points, most of the run time; it also
// needs the [query][block] partial arrays and a second kernel to fold them.  But survivors of the fp32 bound are rare (about
// as many points as the sample stride lie closer than the closest SAMPLED point), so here a lane that holds one evaluates it
// exactly and appends (d2, index) to its query's candidate list in global memory; nothing else leaves the tile loop.
// nn_reduce_candidates_kernel then folds each list by (d2, index).  A list that overflows kCandCap entries (exact ties in
// bulk: duplicate points, lattice clouds seen from a lattice point) is queued for nn_overflow_scan_kernel, an exact scan by the
// whole grid -- rare, and still exact.
constexpr uint32_t kCandCap = 256;

__global__ __launch_bounds__(256) void nn_tile_candidates_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                 const float *__restrict__ z, uint32_t n, uint32_t chunk_groups,
                                                                 const float *__restrict__ qf, const double *__restrict__ q64,
                                                                 const uint32_t *__restrict__ bound_bits, int Q, int qslice,
                                                                 uint32_t *__restrict__ cand_count, double *__restrict__ cand_d2,
                                                                 uint32_t *__restrict__ cand_idx)
{
    extern __shared__ float4 s_pts[];                 // [3][chunk_groups]
    const uint32_t ngroups = n >> 2;
    const uint32_t g0 = blockIdx.x * chunk_groups;
    const uint32_t ng = min(chunk_groups, ngroups > g0 ? ngroups - g0 : 0u);
    float4 *sx = s_pts, *sy = s_pts + chunk_groups, *sz = s_pts + 2 * chunk_groups;
    for (uint32_t i = threadIdx.x; i < ng; i += 256) {
        sx[i] = reinterpret_cast<const float4 *>(x)[g0 + i];
        sy[i] = reinterpret_cast<const float4 *>(y)[g0 + i];
        sz[i] = reinterpret_cast<const float4 *>(z)[g0 + i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool tail_owner = (blockIdx.x == gridDim.x - 1) && threadIdx.x < (n & 3u);   // n % 4 leftover points

    const int q_end = min(Q, ((int)blockIdx.y + 1) * qslice);      // blockIdx.y = slice of the batch (see nn_sample_bounds_kernel)
    for (int q0 = (int)blockIdx.y * qslice; q0 < q_end; q0 += kTileQ) {
        const int qcount = min(kTileQ, q_end - q0);
        float qx[kTileQ], qy[kTileQ], qz[kTileQ], thr[kTileQ];
#pragma unroll
        for (int j = 0; j < kTileQ; j++) {
            const int qi = q0 + (j < qcount ? j : qcount - 1);
            qx[j] = qf[3 * qi]; qy[j] = qf[3 * qi + 1]; qz[j] = qf[3 * qi + 2];
            thr[j] = __uint_as_float(bound_bits[qi]) * (1.0f + 0x1p-19f) + 0x1p-90f;   // FLT_MAX bound -> +inf: everything is a candidate
        }
        for (uint32_t i = threadIdx.x; i < ng; i += 256) {
            const float4 X = sx[i], Y = sy[i], Z = sz[i];
            const PointGroup pg = make_group(X, Y, Z);
            unsigned long long hit = 0ull;
#pragma unroll
            for (int j = 0; j < kTileQ; j++) hit |= __builtin_amdgcn_ballot_w64(group_min_d32(pg, qx[j], qy[j], qz[j]) <= thr[j]);
            if (hit != 0ull) {                                 // wave-uniform: some lane holds a possible winner for some query
                if ((hit >> lane) & 1ull) {
                    const float xs[4] = { X.x, X.y, X.z, X.w }, ys[4] = { Y.x, Y.y, Y.z, Y.w }, zs[4] = { Z.x, Z.y, Z.z, Z.w };
#pragma unroll 1
                    for (int j = 0; j < qcount; j++) {
                        if (!(group_min_d32(pg, qx[j], qy[j], qz[j]) <= thr[j])) continue;
                        const int qi = q0 + j;
                        const double Qx = q64[3 * qi], Qy = q64[3 * qi + 1], Qz = q64[3 * qi + 2];
                        double bd = __builtin_huge_val();
                        uint32_t bi = kNoIndex;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const double d2 = dist2((double)xs[k], (double)ys[k], (double)zs[k], Qx, Qy, Qz);
                            if (d2 < bd) { bd = d2; bi = 4u * (g0 + i) + (uint32_t)k; }       // ids grow with k: lowest index on ties
                        }
                        const uint32_t slot = atomicAdd(&cand_count[qi], 1u);
                        if (slot < kCandCap) { cand_d2[(size_t)qi * kCandCap + slot] = bd; cand_idx[(size_t)qi * kCandCap + slot] = bi; }
                    }
                }
            }
        }
        if (tail_owner) {
            const uint32_t id = 4u * ngroups + threadIdx.x;
            const double px = (double)x[id], py = (double)y[id], pz = (double)z[id];
            for (int j = 0; j < qcount; j++) {
                const int qi = q0 + j;
                const double d2 = dist2(px, py, pz, q64[3 * qi], q64[3 * qi + 1], q64[3 * qi + 2]);
                // the true nearest neighbour is never farther than the nearest SAMPLED point, whose fp32 distance is the bound
                if (d2 <= (double)thr[j] * (1.0 + 0x1p-18) + 1e-300) {
                    const uint32_t slot = atomicAdd(&cand_count[qi], 1u);
                    if (slot < kCandCap) { cand_d2[(size_t)qi * kCandCap + slot] = d2; cand_idx[(size_t)qi * kCandCap + slot] = id; }
                }
            }
        }
    }
}

// one 256-thread block per query: fold its candidate list by (d2, index); re-zero the counter for the next slice.  A query
// whose list overflowed is queued (ovf[0] = count, ovf[1..] = queries) for the two kernels below.
__global__ __launch_bounds__(256) void nn_reduce_candidates_kernel(uint32_t *__restrict__ cand_count, const double *__restrict__ cand_d2,
                                                                   const uint32_t *__restrict__ cand_idx, uint32_t index_base,
                                                                   uint32_t *__restrict__ ovf, uint32_t *__restrict__ out_idx,
                                                                   double *__restrict__ out_d2)
{
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    const int q = blockIdx.x;
    const uint32_t count = cand_count[q];
    if (count > kCandCap) {                                   // block-uniform
        if (threadIdx.x == 0) { ovf[1 + atomicAdd(&ovf[0], 1u)] = (uint32_t)q; cand_count[q] = 0; }
        return;
    }
    double d = __builtin_huge_val();
    uint32_t i = kNoIndex;
    if (threadIdx.x < count) { d = cand_d2[(size_t)q * kCandCap + threadIdx.x]; i = cand_idx[(size_t)q * kCandCap + threadIdx.x]; }
    wave_argmin(d, i);
    if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = d; s_i[threadIdx.x >> 6] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (better(s_d[w], s_i[w], d, i)) { d = s_d[w]; i = s_i[w]; }
        out_idx[q] = (i == kNoIndex) ? kNoIndex : i + index_base;
        out_d2[q] = d;
        cand_count[q] = 0;
    }
}

// Overflowed candidate lists (bulk exact ties: a sensor that accumulates the same points frame after frame, lattice clouds
// seen from a lattice point): the whole grid scans the cloud once more for just those queries, in exact fp64, each block its
// own contiguous range of points; a second kernel folds the per-block winners.  Both are launched after every slice and
// return at once when nothing overflowed (ovf[0] == 0), so the host never has to read the count back.
constexpr int kOvfBlocks = 1024;

__global__ __launch_bounds__(256) void nn_overflow_scan_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                               const float *__restrict__ z, uint32_t n, const double *__restrict__ q64,
                                                               const uint32_t *__restrict__ ovf, double *__restrict__ part_d2,
                                                               uint32_t *__restrict__ part_idx)
{
    const uint32_t count = ovf[0];
    if (count == 0) return;
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t begin = min(n, blockIdx.x * per), end = min(n, begin + per);
    for (uint32_t m = 0; m < count; m++) {
        const uint32_t q = ovf[1 + m];
        const double qx = q64[3 * q], qy = q64[3 * q + 1], qz = q64[3 * q + 2];
        double d = __builtin_huge_val();
        uint32_t i = kNoIndex;
        for (uint32_t p = begin + threadIdx.x; p < end; p += 256) {
            const double d2 = dist2((double)x[p], (double)y[p], (double)z[p], qx, qy, qz);
            if (d2 < d) { d = d2; i = p; }                    // ids grow within a thread
        }
        wave_argmin(d, i);
        __syncthreads();                                      // previous round's s_d / s_i have been read
        if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = d; s_i[threadIdx.x >> 6] = i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; w++)
                if (better(s_d[w], s_i[w], d, i)) { d = s_d[w]; i = s_i[w]; }
            part_d2[(size_t)m * gridDim.x + blockIdx.x] = d;
            part_idx[(size_t)m * gridDim.x + blockIdx.x] = i;
        }
    }
}

__global__ __launch_bounds__(256) void nn_overflow_fold_kernel(const uint32_t *__restrict__ ovf, const double *__restrict__ part_d2,
                                                               const uint32_t *__restrict__ part_idx, int nparts, uint32_t index_base,
                                                               uint32_t *__restrict__ out_idx, double *__restrict__ out_d2)
{
    if (blockIdx.x >= ovf[0]) return;
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    const uint32_t q = ovf[1 + blockIdx.x];
    double d = __builtin_huge_val();
    uint32_t i = kNoIndex;
    for (int p = threadIdx.x; p < nparts; p += 256) {
        const double pd = part_d2[(size_t)blockIdx.x * nparts + p];
        const uint32_t pi = part_idx[(size_t)blockIdx.x * nparts + p];
        if (better(pd, pi, d, i)) { d = pd; i = pi; }
    }
    wave_argmin(d, i);
    if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = d; s_i[threadIdx.x >> 6] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (better(s_d[w], s_i[w], d, i)) { d = s_d[w]; i = s_i[w]; }
        out_idx[q] = (i == kNoIndex) ? kNoIndex : i + index_base;
        out_d2[q] = d;
    }
}

// Exchange step of the sharded cloud, between its two all_reduce(min) calls: a rank stays in the race for a query only if its own
// squared distance IS the global minimum; everybody else offers INT32_MAX.  One pass instead of five elementwise kernels.
__global__ __launch_bounds__(256) void merge_mask_kernel(const double *__restrict__ d2_local, const double *__restrict__ d2_best,
                                                         const uint32_t *__restrict__ idx_local, int32_t *__restrict__ cand, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = d2_local[i];
    const uint32_t ix = idx_local[i];
    cand[i] = (d == d2_best[i] && d < __builtin_huge_val() && ix <= 0x7FFFFFFEu) ? (int32_t)ix : 0x7FFFFFFF;
}

// ---- spatially routed multi-GPU queries (include/pct_shard.h, "routed form"): every rank owns a slab of the cloud along one axis
// and answers only the queries that fall into it.  An answer travels as one 16-byte record.
struct RouteAnswer { uint32_t query, gid; double d2; };        // d2 < 0: the owner could not certify it (everybody answers it again)
constexpr int kRouteMaxWorld = 64;
struct RouteCuts { int world, axis; double cut[kRouteMaxWorld + 1]; };       // slab k = [cut[k], cut[k+1]) along `axis`; cut[0] = -inf, cut[world] = +inf

__device__ __forceinline__ int route_owner(const RouteCuts &C, double x)
{
    int k = 0;
    for (int j = 1; j < C.world; j++) k += x >= C.cut[j] ? 1 : 0;             // cuts ascend: the number of cuts at or below x
    return k;
}

// owner of every query; counts per owner (one LDS histogram per block); the queries of `rank` are compacted (any order: an answer
// carries its query's index)
__global__ __launch_bounds__(256) void route_owner_kernel(RouteCuts C, int rank, const float *__restrict__ q, uint32_t Q,
                                                          uint32_t *__restrict__ counts, uint32_t *__restrict__ mine_ids, float *__restrict__ mine_q)
{
    __shared__ uint32_t h[kRouteMaxWorld];
    __shared__ uint32_t s_base;
    if (threadIdx.x < (uint32_t)kRouteMaxWorld) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int own = -1;
    float qx = 0, qy = 0, qz = 0;
    uint32_t slot = 0;
    if (i < Q) {
        qx = q[3 * i]; qy = q[3 * i + 1]; qz = q[3 * i + 2];
        own = route_owner(C, (double)(C.axis == 0 ? qx : C.axis == 1 ? qy : qz));
        slot = atomicAdd(&h[own], 1u);
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)C.world && h[threadIdx.x]) {
        const uint32_t b = atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
        if ((int)threadIdx.x == rank) s_base = b;
    }
    __syncthreads();
    if (own == rank) {
        const uint32_t p = s_base + slot;
        mine_ids[p] = i;
        mine_q[3 * p] = qx; mine_q[3 * p + 1] = qy; mine_q[3 * p + 2] = qz;
    }
}

// partitioned batches (every rank brings its OWN queries): owner of every query + counts per owner ...
__global__ __launch_bounds__(256) void route_owner_all_kernel(RouteCuts C, const float *__restrict__ q, uint32_t Q, uint32_t *__restrict__ counts,
                                                              unsigned char *__restrict__ owner)
{
    __shared__ uint32_t h[kRouteMaxWorld];
    if (threadIdx.x < (uint32_t)kRouteMaxWorld) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Q) {
        const int own = route_owner(C, (double)q[3 * i + C.axis]);
        owner[i] = (unsigned char)own;
        atomicAdd(&h[own], 1u);
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)C.world && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}
// ... then the queries grouped by owner (segment k starts at off.v[k]; order inside a segment is arrival order of the atomics: every
// record carries the slot it came from)
struct RouteOffsets { uint32_t v[kRouteMaxWorld]; };
__global__ __launch_bounds__(256) void route_partition_kernel(RouteOffsets off, const unsigned char *__restrict__ owner, const float *__restrict__ q, uint32_t Q,
                                                              uint32_t *__restrict__ cursors, float *__restrict__ out_xyz, uint32_t *__restrict__ out_slot)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const uint32_t o = owner[i];
    const uint32_t p = off.v[o] + atomicAdd(&cursors[o], 1u);
    out_xyz[3 * p] = q[3 * i]; out_xyz[3 * p + 1] = q[3 * i + 1]; out_xyz[3 * p + 2] = q[3 * i + 2];
    out_slot[p] = i;
}

// the owner's answers as records: certified when the point found is STRICTLY nearer than the edge of the slab's halo (a point just
// outside the halo at exactly that distance could tie with a lower index), otherwise flagged with d2 = -1
__global__ __launch_bounds__(256) void route_certify_kernel(int axis, double lo_edge, double hi_edge, const float *__restrict__ mine_q,
                                                            const uint32_t *__restrict__ mine_ids, uint32_t m, const uint32_t *__restrict__ lidx,
                                                            const double *__restrict__ ld2, const uint32_t *__restrict__ gid,
                                                            RouteAnswer *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double x = (double)mine_q[3 * i + axis];
    const double margin = fmin(x - lo_edge, hi_edge - x);             // +inf at the outer slabs' outer sides
    const double d2 = ld2[i];
    const bool found = lidx[i] != kNoIndex;
    const bool cert = found && margin > 0.0 && d2 < margin * margin;
    RouteAnswer a;
    a.query = mine_ids[i];
    a.gid = cert ? gid[lidx[i]] : kNoIndex;
    a.d2 = cert ? d2 : -1.0;
    out[i] = a;
}

// records -> the result arrays (each query appears exactly once); flagged queries are counted and listed
__global__ __launch_bounds__(256) void route_scatter_kernel(const RouteAnswer *__restrict__ rec, uint32_t n, uint32_t *__restrict__ out_idx,
                                                            double *__restrict__ out_d2, uint32_t *__restrict__ flag_count, uint32_t *__restrict__ flag_ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RouteAnswer a = rec[i];
    out_idx[a.query] = a.gid;
    out_d2[a.query] = a.d2;
    if (a.d2 < 0.0) flag_ids[atomicAdd(flag_count, 1u)] = a.query;
}

// second round: gather the flagged queries / map local answers to global ids / put the merged answers back
__global__ __launch_bounds__(256) void route_gather_queries_kernel(const float *__restrict__ q, const uint32_t *__restrict__ ids, uint32_t n, float *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = ids[i];
    out[3 * i] = q[3 * k]; out[3 * i + 1] = q[3 * k + 1]; out[3 * i + 2] = q[3 * k + 2];
}
__global__ __launch_bounds__(256) void route_to_global_kernel(uint32_t *__restrict__ lidx, uint32_t n, const uint32_t *__restrict__ gid)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && lidx[i] != kNoIndex) lidx[i] = gid[lidx[i]];
}
__global__ __launch_bounds__(256) void route_put_back_kernel(const uint32_t *__restrict__ ids, uint32_t n, const uint32_t *__restrict__ idx, const double *__restrict__ d2,
                                                             uint32_t *__restrict__ out_idx, double *__restrict__ out_d2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_idx[ids[i]] = idx[i];
    out_d2[ids[i]] = d2[i];
}

// behind the second reduction: the merged int32 candidates back to the engine's index convention (INT32_MAX = no shard holds a point)
__global__ __launch_bounds__(256) void merge_finish_kernel(const int32_t *__restrict__ cand, uint32_t *__restrict__ idx, uint32_t Q)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Q) idx[i] = cand[i] == 0x7FFFFFFF ? kNoIndex : (uint32_t)cand[i];
}

// one 256-thread block per query folds the per-block partials (loads issued back to back,
// compared afterwards); adds the shard's index base
__global__ __launch_bounds__(256) void nn_reduce_partials_kernel(const double *__restrict__ part_d2,
                                                                 const uint32_t *__restrict__ part_idx, int nparts,
                                                                 uint32_t index_base, uint32_t *__restrict__ out_idx,
                                                                 double *__restrict__ out_d2)
{
    const int q = blockIdx.x;
    const double *pd = part_d2 + (size_t)q * nparts;
    const uint32_t *pi = part_idx + (size_t)q * nparts;
    double d = __builtin_huge_val();
    uint32_t i = kNoIndex;
    for (int base = 0; base < nparts; base += 1024) {
        double ld[4];
        uint32_t li[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int p = base + k * 256 + (int)threadIdx.x;
            ld[k] = p < nparts ? pd[p] : __builtin_huge_val();
            li[k] = p < nparts ? pi[p] : kNoIndex;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (better(ld[k], li[k], d, i)) { d = ld[k]; i = li[k]; }
    }
    wave_argmin(d, i);
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = d; s_i[threadIdx.x >> 6] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; w++)
            if (better(s_d[w], s_i[w], d, i)) { d = s_d[w]; i = s_i[w]; }
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

// Order-preserving crop (lidar sensor: camera_sensor.cpp:133-145 radiusSearch + PointCloud(cloud, indices)).  A first
// version appended hits through one global cursor: it serialised on that address (~12 ns per hit, 20 K hits = 0.25 ms) and
// returned arrival order; this pair keeps insertion order and has no contended atomic:
//   crop_count_kernel    hits per 1024-point tile                                    (12 B/point read)
//   (scan_tile_sums_kernel, one block: exclusive scan of the tile counts)
//   crop_scatter_kernel  recompute the test, rank inside the tile, write {index, d2, x, y, z}  (12 B/point + 32 B/hit)
constexpr int kCropTile = 1024;

__device__ __forceinline__ uint32_t crop_tile_rank(const uint32_t f[4], uint32_t &tile_total)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t tsum = f[0] + f[1] + f[2] + f[3];
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
    tile_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return wave_off + inc - tsum;          // exclusive rank of this thread's first element inside the tile
}

__global__ __launch_bounds__(256) void crop_count_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                         const float *__restrict__ z, uint32_t n, double qx, double qy, double qz,
                                                         double r2, uint32_t *__restrict__ tile_sum)
{
    const uint32_t first = blockIdx.x * kCropTile + threadIdx.x * 4;
    uint32_t f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = first + k;
        f[k] = (i < n && dist2((double)x[i], (double)y[i], (double)z[i], qx, qy, qz) <= r2) ? 1u : 0u;
    }
    uint32_t total;
    (void)crop_tile_rank(f, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void crop_scatter_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ z, uint32_t n, double qx, double qy, double qz,
                                                           double r2, uint32_t index_base, const uint32_t *__restrict__ tile_off,
                                                           uint32_t cap, uint32_t *__restrict__ out_idx, double *__restrict__ out_d2,
                                                           float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz)
{
    const uint32_t first = blockIdx.x * kCropTile + threadIdx.x * 4;
    uint32_t f[4];
    double d[4];
    float px[4], py[4], pz[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = first + k;
        f[k] = 0;
        if (i < n) {
            px[k] = x[i]; py[k] = y[i]; pz[k] = z[i];
            d[k] = dist2((double)px[k], (double)py[k], (double)pz[k], qx, qy, qz);
            f[k] = d[k] <= r2 ? 1u : 0u;
        }
    }
    uint32_t total;
    uint32_t pos = tile_off[blockIdx.x] + crop_tile_rank(f, total);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (f[k]) {
            if (pos < cap) {
                if (out_idx) out_idx[pos] = first + k + index_base;
                if (out_d2) out_d2[pos] = d[k];
                if (ox) { ox[pos] = px[k]; oy[pos] = py[k]; oz[pos] = pz[k]; }
            }
            pos++;
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
    int octant_first;          // cooperative NN: try the 2x2x2 block on the query's side before the 3x3x3 cube
    // block table (optional, nullptr = none): for every corner (u, v, w) of the cell lattice, u in [0, gx] etc., the four x-runs of the
    // 2x2x2 block of cells around it -- entry 2*i = the runs' first records, 2*i + 1 = one past their last (block_corner_kernel)
    const uint4 *blocks;
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

// The block table (opt-in: PCT_BLOCK_TABLE=1; measured slower than what it replaces, see engine.hip).  Stage 0 of the batch search reads
// the four x-runs of the 2x2x2 block of cells on the query's side of its cell: eight words of cell_start in four different rows of the
// table = four cache lines per query, 40 % of the line look-ups of the dense batch kernel.  The block only depends on the lattice corner
// the query is nearest to, so the build can lay the eight words of every corner side by side: one 32-byte entry, one line, no cross-lane
// broadcast -- but a table eight times the size of cell_start, which no longer lives in the L2s.  Corner (u, v, w), u in [0, gx]: cells [max(u-1, 0), min(u, gx-1)] along x, the same along y and z; a row that
// coincides with another one at the border of the grid is stored as an empty run (first == last), exactly as stage 0 computes it.
__global__ __launch_bounds__(256) void block_corner_kernel(GridDesc G, const uint32_t *__restrict__ cell_start, uint4 *__restrict__ blocks, uint32_t ncorners)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncorners) return;
    const uint32_t ux = (uint32_t)G.gx + 1u, uy = (uint32_t)G.gy + 1u;
    const int u = (int)(i % ux), v = (int)((i / ux) % uy), w = (int)(i / (ux * uy));
    const int xa = max(u - 1, 0), xb = min(u, G.gx - 1), ya = max(v - 1, 0), yb = min(v, G.gy - 1), za = max(w - 1, 0), zb = min(w, G.gz - 1);
    uint32_t s[4], e[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const bool ok = !((r >> 1) && zb == za) && !((r & 1) && yb == ya);
        const uint32_t row = cell_lin(G, 0, (r & 1) ? yb : ya, (r >> 1) ? zb : za);
        s[r] = cell_start[row + xa];
        e[r] = ok ? cell_start[row + xb + 1] : s[r];
    }
    blocks[2u * i] = make_uint4(s[0], s[1], s[2], s[3]);
    blocks[2u * i + 1u] = make_uint4(e[0], e[1], e[2], e[3]);
}

// OPTIONAL coarser copies of the index (cell size x4 per level, same origin), used only by the block-per-query
// express kernel: a query that finds nothing decisive in its 3x3x3 fine cells tries the 3x3x3 cells of each coarser
// level before expanding shell by shell at the coarsest level.  n = 0: no pyramid (the default, see engine.hip).
constexpr int kMaxCoarse = 3;
struct CoarseLevels {
    int n;
    GridDesc G[kMaxCoarse];
    const float4 *pts[kMaxCoarse];
    const uint32_t *cell_start[kMaxCoarse];
};

__global__ __launch_bounds__(256) void count_empty_cells_kernel(const uint32_t *__restrict__ cell_start, uint32_t ncells,
                                                                uint32_t *__restrict__ n_empty)
{
    uint32_t c = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ncells; i += stride) c += cell_start[i + 1] == cell_start[i] ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, kWave);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(n_empty, c);
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

// Query binning for the cell-pruned kernels: a counting sort of the batch by coarse cell
// ((cx,cy,cz) >> shift, x fastest) so that neighbouring lanes walk neighbouring cells and
// share cache lines.  Random 1M-query batches ran 2.5x faster pre-sorted.
// Bin order = (y-strip, z, y inside the strip, x): walking the batch in this order sweeps a strip
// of `strip` bin rows through all z before moving to the next strip, so the planes a stretch of
// queries re-uses (z-1, z, z+1) are a strip wide, not a whole plane wide, and stay inside one
// XCD's 4 MiB L2 (a full-plane sweep re-fetched every plane ~3x, measured).
struct BinDesc { int shift, bx, by, bz, strip; uint32_t nbins; };

__device__ __forceinline__ uint32_t query_bin(const GridDesc &G, const BinDesc &B, float qx, float qy, float qz)
{
    const int cx = cell_coord(qx, G.ox, G.inv_h, G.gx) >> B.shift;
    const int cy = cell_coord(qy, G.oy, G.inv_h, G.gy) >> B.shift;
    const int cz = cell_coord(qz, G.oz, G.inv_h, G.gz) >> B.shift;
    const uint32_t s = (uint32_t)cy / (uint32_t)B.strip, yin = (uint32_t)cy % (uint32_t)B.strip;
    return ((s * (uint32_t)B.bz + (uint32_t)cz) * (uint32_t)B.strip + yin) * (uint32_t)B.bx + (uint32_t)cx;
}

__global__ __launch_bounds__(256) void query_bin_count_kernel(GridDesc G, BinDesc B, const float *__restrict__ q, uint32_t Q,
                                                              uint32_t *__restrict__ bin_count, uint32_t *__restrict__ qbin)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Q) return;
    const uint32_t b = query_bin(G, B, q[3 * t], q[3 * t + 1], q[3 * t + 2]);
    qbin[t] = b;
    atomicAdd(&bin_count[b], 1u);
}

// perm[pos] = original query id; qsorted[pos] = {qx, qy, qz, bitcast(id)} so the NN kernel gets a
// query and its output slot with ONE 16-byte load instead of a perm -> q dependent pair
__global__ __launch_bounds__(256) void query_bin_scatter_kernel(const uint32_t *__restrict__ qbin, uint32_t Q,
                                                                const uint32_t *__restrict__ bin_start,
                                                                uint32_t *__restrict__ bin_fill, const float *__restrict__ q,
                                                                uint32_t *__restrict__ perm, float4 *__restrict__ qsorted)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Q) return;
    const uint32_t b = qbin[t];
    const uint32_t pos = bin_start[b] + atomicAdd(&bin_fill[b], 1u);
    perm[pos] = t;
    qsorted[pos] = make_float4(q[3 * t], q[3 * t + 1], q[3 * t + 2], __uint_as_float(t));
}

// Counting sort of the batch on LDS histograms, one or two levels (replaces the global-atomic version
// above for large batches: 2 M scattered device-scope atomics ran at ~20 G/s = 100 us per 1 M
// queries).  key = bin >> key_shift (< 2^20); level 1 = key >> 10 (<= 1024
// buckets), level 2 = key & 1023 inside a bucket.  Global atomics are one per (block, non-empty
// bucket); everything else is LDS atomics and coalesced traffic.
constexpr int kSortBuckets = 1024;
constexpr int kSortItems = 8;            // queries per thread in the level-1 kernels
constexpr int kSortPerBlock = 1024 * kSortItems;      // most queries per block in the level-1 kernels (engine.hip picks 1024..8192 by batch size)

// A thread's kSortItems queries of the level-1 kernels.  VEC (the query array is 16-byte aligned): the thread owns chunks of 4 CONSECUTIVE
// queries and fetches each chunk as three float4 (48 contiguous bytes) instead of twelve dwords at a stride of 12 bytes: a quarter of the
// load instructions.  Item k of thread `tid` is query sort_item<VEC>(base, k, tid); an item is live when sort_item_live says so.
template <bool VEC>
__device__ __forceinline__ uint32_t sort_item(uint32_t base, int k, uint32_t tid)
{
    return VEC ? base + 4u * (tid + 1024u * (uint32_t)(k >> 2)) + (uint32_t)(k & 3) : base + (uint32_t)k * 1024u + tid;
}
template <bool VEC>
__device__ __forceinline__ bool sort_item_live(uint32_t base, int k, uint32_t tid, int items, uint32_t Q)
{
    if (VEC) return (tid + 1024u * (uint32_t)(k >> 2)) < 256u * (uint32_t)items && sort_item<true>(base, k, tid) < Q;
    return k < items && sort_item<false>(base, k, tid) < Q;
}
template <bool VEC>
__device__ __forceinline__ void sort_load_items(const float *__restrict__ q, uint32_t base, uint32_t tid, int items, uint32_t Q, float (&qv)[kSortItems][3])
{
    if (VEC) {
#pragma unroll
        for (int g = 0; g < kSortItems / 4; g++) {
            const uint32_t t0 = sort_item<true>(base, 4 * g, tid);
            if ((tid + 1024u * (uint32_t)g) < 256u * (uint32_t)items && t0 + 3u < Q) {       // base is a multiple of 1024: 3 * t0 floats = a multiple of 12
                const float4 *v = reinterpret_cast<const float4 *>(q + 3 * (size_t)t0);
                const float4 A = v[0], B = v[1], C = v[2];
                qv[4 * g][0] = A.x; qv[4 * g][1] = A.y; qv[4 * g][2] = A.z;
                qv[4 * g + 1][0] = A.w; qv[4 * g + 1][1] = B.x; qv[4 * g + 1][2] = B.y;
                qv[4 * g + 2][0] = B.z; qv[4 * g + 2][1] = B.w; qv[4 * g + 2][2] = C.x;
                qv[4 * g + 3][0] = C.y; qv[4 * g + 3][1] = C.z; qv[4 * g + 3][2] = C.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = 4 * g + j;
                    const uint32_t t = sort_item<true>(base, k, tid);
                    const bool ok = sort_item_live<true>(base, k, tid, items, Q);
                    qv[k][0] = ok ? q[3 * (size_t)t] : 0.0f; qv[k][1] = ok ? q[3 * (size_t)t + 1] : 0.0f; qv[k][2] = ok ? q[3 * (size_t)t + 2] : 0.0f;
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < kSortItems; k++) {
            const uint32_t t = sort_item<false>(base, k, tid);
            const bool ok = sort_item_live<false>(base, k, tid, items, Q);
            qv[k][0] = ok ? q[3 * t] : 0.0f; qv[k][1] = ok ? q[3 * t + 1] : 0.0f; qv[k][2] = ok ? q[3 * t + 2] : 0.0f;
        }
    }
}

//
// Three dependent launches per batch (hist -> scatter1 -> fine); at <= 256 K queries the sort is launch-latency-bound
// (~40 us for five dependent operations, measured), so the bucket scan lives inside scatter1 and the two counter
// arrays are re-zeroed without extra launches where possible: total1 is zero on entry (zeroed at allocation, then by the fine
// kernel or, in single-level mode, a memset behind scatter1), fill1 is zeroed here, before any scatter1 block can touch it.
template <bool VEC>
__global__ __launch_bounds__(1024) void qsort_hist_kernel(GridDesc G, BinDesc B, int key_shift, int lshift, const float *__restrict__ q,
                                                          uint32_t Q, uint32_t per_block, uint32_t *__restrict__ keys,
                                                          uint32_t *__restrict__ total1, uint32_t *__restrict__ fill1,
                                                          uint32_t *__restrict__ total1_next)
{
    __shared__ uint32_t h[kSortBuckets];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < kSortBuckets; i += 1024) {
            fill1[i] = 0;
            if (total1_next) total1_next[i] = 0;      // the next batch's totals (its last reader finished a batch ago): no memset on the path
        }
    for (int i = threadIdx.x; i < kSortBuckets; i += 1024) h[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * per_block;
    const int items = (int)(per_block >> 10);             // per_block is a multiple of 1024, at most kSortPerBlock
    // all of a thread's queries are requested before the first is used: the loop used to pay one memory round trip per item
    float qv[kSortItems][3];
    sort_load_items<VEC>(q, base, threadIdx.x, items, Q, qv);
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t t = sort_item<VEC>(base, k, threadIdx.x);
        if (sort_item_live<VEC>(base, k, threadIdx.x, items, Q)) {
            const uint32_t key = query_bin(G, B, qv[k][0], qv[k][1], qv[k][2]) >> key_shift;
            if (keys) keys[t] = key;                 // nullptr: the scatter pass recomputes the key from the query it reads anyway
            atomicAdd(&h[key >> lshift], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSortBuckets; i += 1024)
        if (h[i]) atomicAdd(&total1[i], h[i]);
}

// level 1 scatter: the query record {x, y, z, bitcast(id)} travels with its key, so the fine pass
// never gathers from the (randomly ordered) input again.  Every block scans the 1024 bucket totals itself
// (thread i = bucket i) instead of waiting for a one-block scan kernel; block 0 publishes the starts for the fine pass.
template <bool VEC>
__global__ __launch_bounds__(1024) void qsort_scatter1_kernel(GridDesc G, BinDesc B, int key_shift, const uint32_t *__restrict__ keys, const float *__restrict__ q,
                                                              uint32_t Q, uint32_t per_block, int lshift, const uint32_t *__restrict__ total1,
                                                              uint32_t *__restrict__ fill1, uint32_t *__restrict__ start1,
                                                              uint32_t *__restrict__ tmp_key, float4 *__restrict__ tmp_rec,
                                                              uint32_t *__restrict__ perm, uint32_t *__restrict__ inv, int final_level)
{
    static_assert(kSortBuckets == 1024, "one thread per bucket");
    __shared__ uint32_t h[kSortBuckets];
    __shared__ uint32_t basepos[kSortBuckets];
    __shared__ uint32_t s_wave[16];
    h[threadIdx.x] = 0;
    const uint32_t tot = total1[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off, kWave);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t start = inc - tot;
    for (int w = 0; w < wave; w++) start += s_wave[w];
    if (blockIdx.x == 0) {
        start1[threadIdx.x] = start;
        if (threadIdx.x == kSortBuckets - 1) start1[kSortBuckets] = start + tot;
    }
    const uint32_t base = blockIdx.x * per_block;
    const int items = (int)(per_block >> 10);
    // keys and queries of all of a thread's items are requested up front (one round trip for each array instead of one per item)
    uint32_t key[kSortItems];
    float qv[kSortItems][3];
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t t = sort_item<VEC>(base, k, threadIdx.x);
        const bool ok = sort_item_live<VEC>(base, k, threadIdx.x, items, Q);
        key[k] = ok ? (keys ? keys[t] : 0u) : 0xFFFFFFFFu;
    }
    sort_load_items<VEC>(q, base, threadIdx.x, items, Q, qv);
    if (!keys) {
#pragma unroll
        for (int k = 0; k < kSortItems; k++)
            if (key[k] != 0xFFFFFFFFu) key[k] = query_bin(G, B, qv[k][0], qv[k][1], qv[k][2]) >> key_shift;
    }
    // the histogram atomic's return value IS the query's rank inside (block, bucket): one LDS atomic per query, not two
    uint32_t rank[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; k++)
        if (key[k] != 0xFFFFFFFFu) rank[k] = atomicAdd(&h[key[k] >> lshift], 1u);
    __syncthreads();
    {
        const uint32_t mine = h[threadIdx.x];
        basepos[threadIdx.x] = mine ? start + atomicAdd(&fill1[threadIdx.x], mine) : 0u;   // this block's slice of the bucket
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        if (key[k] != 0xFFFFFFFFu) {
            const uint32_t t = sort_item<VEC>(base, k, threadIdx.x);
            const uint32_t pos = basepos[key[k] >> lshift] + rank[k];
            if (!final_level) tmp_key[pos] = key[k];  // a fine pass follows and needs the key
            else if (perm) perm[pos] = t;             // single-level mode: this IS the final order; perm only for the kernels that read it
            if (inv) inv[t] = pos;                    // where query t went (coalesced): results come back through it
            tmp_rec[pos] = make_float4(qv[k][0], qv[k][1], qv[k][2], __uint_as_float(t));
        }
    }
}

// one block per level-1 bucket: counting sort by the low 10 key bits, then emit perm / qsorted.  1024 threads: a bucket holds
// Q / (occupied level-1 buckets) records (~5000 at 1 M queries over 211 buckets), and the three passes are latency-bound
// (same-box A/B: 256 threads 0.2068 ms per step, 1024 threads 0.2008; spreading the keys over more level-1 buckets instead,
// key >> 8 = 844 buckets, gained nothing)
constexpr int kFineThreads = 1024;
__global__ __launch_bounds__(kFineThreads) void qsort_fine_kernel(const uint32_t *__restrict__ tmp_key, const float4 *__restrict__ tmp_rec,
                                                                  const uint32_t *__restrict__ start1, uint32_t *__restrict__ total1,
                                                                  uint32_t lmask, uint32_t *__restrict__ perm, float4 *__restrict__ qsorted,
                                                                  uint32_t *__restrict__ inv)
{
    static_assert(kSortBuckets == kFineThreads, "one histogram entry per thread");
    __shared__ uint32_t h[kSortBuckets];
    __shared__ uint32_t s_wave[kFineThreads / 64];
    const uint32_t s = start1[blockIdx.x], e = start1[blockIdx.x + 1];
    if (threadIdx.x == 0) total1[blockIdx.x] = 0;      // its last reader (scatter1) has finished: ready for the next batch
    if (s == e) return;
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = s + threadIdx.x; i < e; i += kFineThreads) atomicAdd(&h[tmp_key[i] & lmask], 1u);
    __syncthreads();
    // exclusive scan of h[1024] in place: one entry per thread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t v = h[threadIdx.x];
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off, kWave);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t run = inc - v;
    for (int w = 0; w < wave; w++) run += s_wave[w];
    h[threadIdx.x] = run;
    __syncthreads();
    for (uint32_t i = s + threadIdx.x; i < e; i += kFineThreads) {
        const float4 rec = tmp_rec[i];
        const uint32_t pos = s + atomicAdd(&h[tmp_key[i] & lmask], 1u);
        if (perm) perm[pos] = __float_as_uint(rec.w);
        qsorted[pos] = rec;
        if (inv) inv[__float_as_uint(rec.w)] = pos;
    }
}

// results of a sorted batch back into arrival order: out[t] = sorted[inv[t]] -- a gather from the 12 MB of sorted results
// (cache resident) with coalesced writes, instead of 2 x Q scattered partial-line writes from the search kernel
__global__ __launch_bounds__(256) void unpermute_results_kernel(const uint32_t *__restrict__ inv, const uint32_t *__restrict__ sidx,
                                                                const double *__restrict__ sd2, uint32_t Q,
                                                                uint32_t *__restrict__ out_idx, double *__restrict__ out_d2)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Q) return;
    const uint32_t p = inv[t];
    out_idx[t] = sidx[p];
    out_d2[t] = sd2[p];
}

// =====================================================================================
// 4. Cell-pruned kernels: one lane per query walks an expanding cube of cells.
//    Termination is exact: a point outside the scanned cube of cells is at least
//    `bound` away (distance to the cube's faces, minus a slack covering the fp32 cell
//    assignment rounding), so the search stops once best_d2 <= bound^2.
// =====================================================================================
struct WorkCounters { unsigned long long points, cells, nodes; };    // nodes: pyramid node visits (pyramid.hpp), 8 boxes of 32 bytes each
constexpr int kWorkSlots = 64;    // instrumented kernels spread their two counters over 64 slots (same-address atomics serialise)

// XCD-aware block order (cdna_hip_programming.md T1).  Workgroups are dealt round-robin over
// the 8 XCDs, each with a private 4 MiB L2.  With the batch binned in cell order, giving XCD k
// the k-th contiguous eighth of the blocks keeps the three cell planes a stretch of queries
// needs inside ONE L2 instead of spreading every plane over all eight.  Bijective for any grid
// size; placement is a speed matter only.
__device__ __forceinline__ uint32_t xcd_contiguous_block(uint32_t b, uint32_t nblocks)
{
    const uint32_t xcd = b & 7u, k = b >> 3;
    const uint32_t q = nblocks >> 3, r = nblocks & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// distance from q to the nearest face of the cube of cells [c-r, c+r]^3 that still has cells behind it
// (+inf when the cube covers the whole grid), minus the slack that covers the fp32 cell assignment
__device__ __forceinline__ double cube_bound(const GridDesc &G, int cx, int cy, int cz, int r, double qx, double qy, double qz)
{
    double bound = __builtin_huge_val();
    if (cx - r > 0) bound = fmin(bound, qx - (G.oxd + (double)(cx - r) * G.hd));
    if (cx + r < G.gx - 1) bound = fmin(bound, (G.oxd + (double)(cx + r + 1) * G.hd) - qx);
    if (cy - r > 0) bound = fmin(bound, qy - (G.oyd + (double)(cy - r) * G.hd));
    if (cy + r < G.gy - 1) bound = fmin(bound, (G.oyd + (double)(cy + r + 1) * G.hd) - qy);
    if (cz - r > 0) bound = fmin(bound, qz - (G.ozd + (double)(cz - r) * G.hd));
    if (cz + r < G.gz - 1) bound = fmin(bound, (G.ozd + (double)(cz + r + 1) * G.hd) - qz);
    return bound == __builtin_huge_val() ? bound : bound - G.hd * (1.0 / 256.0);
}

// Points [s, e) of the cell-sorted array against one query.  Four independent 16-byte loads are
// issued before the first compare (the tail repeats the last point: a repeated (d2, index) never
// changes the winner), so a short run costs one memory round trip instead of one per point.
__device__ __forceinline__ void scan_points(const float4 *__restrict__ pts, uint32_t s, uint32_t e, double qx, double qy,
                                            double qz, double &bd, uint32_t &bi)
{
    for (uint32_t p = s; p < e; p += 4) {
        const uint32_t last = e - 1;
        const float4 P0 = pts[p], P1 = pts[min(p + 1, last)], P2 = pts[min(p + 2, last)], P3 = pts[min(p + 3, last)];
        const float4 P[4] = { P0, P1, P2, P3 };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double d2 = dist2((double)P[k].x, (double)P[k].y, (double)P[k].z, qx, qy, qz);
            const uint32_t id = __float_as_uint(P[k].w);
            if (better(d2, id, bd, bi)) { bd = d2; bi = id; }
        }
    }
}

// fp32 screening of points [s, e): tracks the smallest and second-smallest fp32 distance and the
// array position of the smallest.  Tail slots of the 4-wide load group count as +inf (a repeated
// point would fake a tie).
__device__ __forceinline__ void screen_points(const float4 *__restrict__ pts, uint32_t s, uint32_t e, float qx, float qy,
                                              float qz, float &m1, float &m2, uint32_t &p1)
{
    for (uint32_t p = s; p < e; p += 4) {
        const uint32_t last = e - 1;
        const float4 P0 = pts[p], P1 = pts[min(p + 1, last)], P2 = pts[min(p + 2, last)], P3 = pts[min(p + 3, last)];
        const float4 P[4] = { P0, P1, P2, P3 };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float dx = P[k].x - qx, dy = P[k].y - qy, dz = P[k].z - qz;
            float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            if (k > 0) d = (p + k <= last) ? d : __builtin_huge_valf();
            const bool lt = d < m1;
            m2 = lt ? m1 : fminf(m2, d);
            p1 = lt ? p + (uint32_t)k : p1;
            m1 = fminf(m1, d);
        }
    }
}

template <bool COUNT>
__device__ __forceinline__ void scan_run(const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                         uint32_t lin0, uint32_t lin1, double qx, double qy, double qz, double &bd,
                                         uint32_t &bi, uint32_t &npts, uint32_t &nruns)
{
    const uint32_t s = cell_start[lin0], e = cell_start[lin1 + 1];
    if (COUNT) { npts += e - s; nruns += 1; }
    scan_points(pts, s, e, qx, qy, qz, bd, bi);
}

template <bool COUNT>
__global__ __launch_bounds__(256) void nn_grid_kernel(GridDesc G, const float4 *__restrict__ pts,
                                                      const uint32_t *__restrict__ cell_start,
                                                      const float *__restrict__ q, uint32_t Q, uint32_t index_base,
                                                      const uint32_t *__restrict__ perm,
                                                      uint32_t *__restrict__ out_idx, double *__restrict__ out_d2,
                                                      WorkCounters *__restrict__ work)
{
    const uint32_t slot = (perm ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t npts = 0, nruns = 0;
    if (slot < Q) {
        const uint32_t t = perm ? perm[slot] : slot;     // binned order in, original order out
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
            if (r == 1) {
                // first cube (3x3x3 cells = 9 contiguous x-runs; almost every query ends here): the 18
                // cell_start reads are independent, so issue them all before touching any point
                uint32_t rs[9], re[9];
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const int zz = cz + k / 3 - 1, yy = cy + k % 3 - 1;
                    const bool ok = zz >= 0 && zz < G.gz && yy >= 0 && yy < G.gy;    // rows outside the grid hold nothing
                    const uint32_t row = ok ? cell_lin(G, 0, yy, zz) : 0u;
                    const uint32_t a = cell_start[row + x0], b = cell_start[row + x1 + 1];
                    rs[k] = a;
                    re[k] = ok ? b : a;
                    if (COUNT) { npts += re[k] - rs[k]; nruns += ok ? 1u : 0u; }
                }
                // fp32 screening first: fp64 costs ~3x per point, and 53 points are looked at per query.
                // If the runner-up is farther than the fp32 error band the fp32 argmin IS the exact
                // winner (every other point p has d2(p) >= d32(p)/(1+e) > d2(best); e < 4e-7, band 2^-19)
                // and one exact evaluation finishes the query; otherwise (near-ties, duplicates) the
                // runs are re-scanned in the exact arithmetic so the (d2, index) order decides.
                float m1 = __builtin_huge_valf(), m2 = __builtin_huge_valf();
                uint32_t p1 = 0;
                // one z-plane (3 rows) at a time: the first 4 points of each row are requested together
                // (12 independent 16-byte loads in flight), rows longer than 4 continue 4 at a time.
                // The kernel is bound by dependent memory round trips, not by arithmetic.
#pragma unroll
                for (int g = 0; g < 3; g++) {
                    float4 P[3][4];
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const uint32_t a = rs[3 * g + j], b = re[3 * g + j];
                        const uint32_t last = b > a ? b - 1 : 0u;      // empty row: read slot 0, masked below
#pragma unroll
                        for (int k = 0; k < 4; k++) P[j][k] = pts[min(a + (uint32_t)k, last)];
                    }
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const uint32_t a = rs[3 * g + j], b = re[3 * g + j];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float dx = P[j][k].x - qxf, dy = P[j][k].y - qyf, dz = P[j][k].z - qzf;
                            float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                            d = (a + (uint32_t)k < b) ? d : __builtin_huge_valf();
                            const bool lt = d < m1;
                            m2 = lt ? m1 : fminf(m2, d);
                            p1 = lt ? a + (uint32_t)k : p1;
                            m1 = fminf(m1, d);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 3; j++)
                        if (re[3 * g + j] > rs[3 * g + j] + 4u) screen_points(pts, rs[3 * g + j] + 4u, re[3 * g + j], qxf, qyf, qzf, m1, m2, p1);
                }
                if (m1 < __builtin_huge_valf()) {
                    if (m2 > m1 * (1.0f + 0x1p-19f) + 0x1p-90f) {
                        const float4 P = pts[p1];
                        bd = dist2((double)P.x, (double)P.y, (double)P.z, qx, qy, qz);
                        bi = __float_as_uint(P.w);
                    } else {
#pragma unroll 1
                        for (int k = 0; k < 9; k++) scan_points(pts, rs[k], re[k], qx, qy, qz, bd, bi);
                    }
                }
            } else
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
        if ((threadIdx.x & 63) == 0) { WorkCounters *w = work + (blockIdx.x & (kWorkSlots - 1)); atomicAdd(&w->points, a); atomicAdd(&w->cells, b); }
    }
}

// -------------------------------------------------------------------------------------
// Cooperative form of the cell-pruned NN: EIGHT lanes per query (8 queries per wave).
//
// The lane-per-query kernel above is bound by the vector L1's address path, not by arithmetic or
// DRAM (measured: ~150 L1 accesses per query, texture-address unit busy 65 % of the kernel):
// a lane reading the 6 points of a run one after the other issues 6 separate 16-byte accesses to
// the SAME 128-byte line.  Here the 8 lanes of a group read 8 consecutive points of a run with one
// coalesced 128-byte access, the 9 rows' cell_start entries are fetched by 9 different lanes at
// once, and the three rows of a z-plane are requested before any is consumed.
// Arithmetic and results are identical to nn_grid_kernel (same screening rule, same exact fp64
// winner by (d2, index), same termination bound).
// -------------------------------------------------------------------------------------
constexpr int kCoop = 8;

// Cross-lane traffic inside a group of 8 lanes goes through DPP (data-parallel primitives: the operand of a VALU instruction is read
// from another lane of the same row) instead of ds_bpermute (an LDS instruction with its issue slot, ~50+ cycles of latency and an
// lgkmcnt wait): the folds below sit on every query's critical path.
constexpr int kDppXor1 = 0xB1;         // quad_perm [1, 0, 3, 2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2, 3, 0, 1]
constexpr int kDppHalfMirror = 0x141;  // lane i <-> 7 - i inside every 8 lanes: pairs the two quads of a group
constexpr int kDppQuadBcast0 = 0x00, kDppQuadBcast1 = 0x55, kDppQuadBcast2 = 0xAA, kDppQuadBcast3 = 0xFF;    // quad_perm [k, k, k, k]
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
#ifdef PCT_AB_NO_DPP        // A/B switch (scripts/ab_build.sh): the same data movement through ds_bpermute
    const int lane = (int)(threadIdx.x & 63u), l8 = lane & 7, q = lane & 3;
    const int src = CTRL == kDppXor1 ? (lane ^ 1) : CTRL == kDppXor2 ? (lane ^ 2) : CTRL == kDppHalfMirror ? ((lane & ~7) | (7 - l8)) :
                    (lane - q + (CTRL == kDppQuadBcast0 ? 0 : CTRL == kDppQuadBcast1 ? 1 : CTRL == kDppQuadBcast2 ? 2 : 3));
    return (uint32_t)__shfl((int)v, src, kWave);
#else
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
#endif
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return __longlong_as_double((long long)(((uint64_t)dpp_u32<CTRL>((uint32_t)(b >> 32)) << 32) | dpp_u32<CTRL>((uint32_t)b)));
}

// (measured on the dense batch kernel: the DPP form of the folds below costs registers -- 64 VGPRs + 16 bytes of scratch at 8 waves per
// SIMD -- and the kernel ran 0.131 instead of 0.122 ms; at the present 7 waves / 72 VGPRs DPP folds + quad broadcasts of the run bounds
// measure the same as ds_bpermute, 0.1161-0.1163 against 0.1166-0.1167 ms, as does reading a +inf pad record instead of clamping and
// masking the slots beyond a run, 0.1156-0.1171 (profiles/r03_ab_dpp.txt).  They stay on ds_bpermute; the pyramid walk uses the DPP
// forms: 1.06 against 1.09 ms.)
__device__ __forceinline__ void coop_argmin8(double &d, uint32_t &i)
{
#pragma unroll
    for (int off = 1; off < kCoop; off <<= 1) {
        const double od = __shfl_xor(d, off, kWave);
        const uint32_t oi = (uint32_t)__shfl_xor((int)i, off, kWave);
        if (better(od, oi, d, i)) { d = od; i = oi; }
    }
}

// exact scan of [s, e) shared by the 8 lanes of a group (lane `sub` takes s+sub, s+sub+8, ...)
__device__ __forceinline__ void coop_scan_exact(const float4 *__restrict__ pts, uint32_t s, uint32_t e, uint32_t sub, double qx,
                                                double qy, double qz, double &bd, uint32_t &bi)
{
    for (uint32_t p = s + sub; p < e; p += kCoop) {
        const float4 P = pts[p];
        const double d2 = dist2((double)P.x, (double)P.y, (double)P.z, qx, qy, qz);
        const uint32_t id = __float_as_uint(P.w);
        if (better(d2, id, bd, bi)) { bd = d2; bi = id; }
    }
}

// rows of the cube (full = true) or of the shell of Chebyshev radius r around cell (cx,cy,cz), shared by 8 lanes
template <bool COUNT>
__device__ __forceinline__ void coop_scan_cube_or_shell(const GridDesc &G, const float4 *__restrict__ pts,
                                                        const uint32_t *__restrict__ cell_start, int cx, int cy, int cz, int r,
                                                        bool full, uint32_t sub, double qx, double qy, double qz, double &bd,
                                                        uint32_t &bi, uint32_t &npts, uint32_t &nruns)
{
    const int x0 = max(cx - r, 0), x1 = min(cx + r, G.gx - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, G.gy - 1);
    const int z0 = max(cz - r, 0), z1 = min(cz + r, G.gz - 1);
    for (int zz = z0; zz <= z1; zz++) {
        const bool zface = (zz == cz - r) || (zz == cz + r);
        for (int yy = y0; yy <= y1; yy++) {
            const uint32_t row = cell_lin(G, 0, yy, zz);
            if (full || zface || yy == cy - r || yy == cy + r) {
                const uint32_t s = cell_start[row + x0], e = cell_start[row + x1 + 1];
                if (COUNT && sub == 0) { npts += e - s; nruns += 1; }
                coop_scan_exact(pts, s, e, sub, qx, qy, qz, bd, bi);
            } else {
                if (cx - r >= 0) {
                    const uint32_t s = cell_start[row + cx - r], e = cell_start[row + cx - r + 1];
                    if (COUNT && sub == 0) { npts += e - s; nruns += 1; }
                    coop_scan_exact(pts, s, e, sub, qx, qy, qz, bd, bi);
                }
                if (cx + r <= G.gx - 1) {
                    const uint32_t s = cell_start[row + cx + r], e = cell_start[row + cx + r + 1];
                    if (COUNT && sub == 0) { npts += e - s; nruns += 1; }
                    coop_scan_exact(pts, s, e, sub, qx, qy, qz, bd, bi);
                }
            }
        }
    }
}

// fp32 screening of NR x-runs by the 8 lanes of a group (min, runner-up, position), then the exact fp64 winner: if the
// runner-up lies outside the fp32 error band the minimum IS the exact winner and is evaluated once in fp64, otherwise the
// runs are rescanned in exact arithmetic.  All NR rows are requested before any is consumed (the kernel is bound by
// dependent memory round trips: one for the bounds, one for the points).  Leaves (bd, bi) = (+inf, none) for empty runs.
// OPEN = true (A/B switch PCT_AB_OPEN_STAGE0, NOT the default): the 8 * DEPTH slots behind a run's first record are read and
// screened whatever the run's length -- what lies behind a short run are the records of the following cells, real points of the cloud
// (or the +inf pad records behind the last one, gb_pad_kernel), and the nearest neighbour over a superset of the block is still a
// valid candidate.  It removes the clamp, the empty-run select and the validity masks (~4 of ~17 vector instructions per slot) -- and
// measured SLOWER on the headline step: 0.146 ms (72 VGPRs, 7 waves) / 0.132 ms (77 VGPRs, 6 waves) / 0.21 ms (64 VGPRs: 60 B of
// scratch) against 0.120 ms masked (profiles/r03_ab_open_stage0.txt).  The clamp is what keeps the lanes beyond a run on the run's last
// cache line; without it every run costs two full lines more often, and this kernel pays for lines before it pays for instructions.
template <int NR, int DEPTH, bool OPEN = false>
__device__ __forceinline__ void coop_screen_rows(const float4 *__restrict__ pts, const uint32_t (&rs)[NR], const uint32_t (&re)[NR],
                                                 uint32_t sub, float qxf, float qyf, float qzf, double qx, double qy, double qz,
                                                 double &bd, uint32_t &bi)
{
    float m1 = __builtin_huge_valf(), m2 = __builtin_huge_valf();
    uint32_t p1 = 0;
    // DEPTH points per lane and row (8 * DEPTH per row for the group) are requested up front
    float4 P[NR][DEPTH];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const uint32_t a = rs[k], b = re[k];
        const uint32_t last = b > a ? b - 1 : 0u;               // empty row: read slot 0, masked below
#pragma unroll
        for (int j = 0; j < DEPTH; j++) P[k][j] = pts[OPEN ? a + sub + kCoop * j : min(a + sub + kCoop * j, last)];
    }
#ifdef PCT_AB_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);                           // nothing moves across: every load above is issued before the first use below
#endif
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const uint32_t a = rs[k], b = re[k];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const uint32_t p = a + sub + kCoop * j;
            const float dx = P[k][j].x - qxf, dy = P[k][j].y - qyf, dz = P[k][j].z - qzf;
            float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            if (!OPEN) d = (p < b) ? d : __builtin_huge_valf();
            const bool lt = d < m1;
            m2 = __builtin_amdgcn_fmed3f(m1, m2, d);            // second smallest of {m1 <= m2, d}
            p1 = lt ? p : p1;
            m1 = fminf(m1, d);
        }
    }
    // rows longer than 8 * DEPTH points.  Unrolled over the rows when there are few of them: a rolled loop indexes rs[] / re[] with
    // its counter, i.e. a chain of selects per iteration (~10 vector instructions x 4 rows in every wave of the batch kernel:
    // 121.5 -> 117 us)
    auto long_row = [&](uint32_t a, uint32_t b) {
        for (uint32_t p = a + kCoop * DEPTH + sub; p < b; p += kCoop) {
            const float4 Pp = pts[p];
            const float dx = Pp.x - qxf, dy = Pp.y - qyf, dz = Pp.z - qzf;
            const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            const bool lt = d < m1;
            m2 = __builtin_amdgcn_fmed3f(m1, m2, d);
            p1 = lt ? p : p1;
            m1 = fminf(m1, d);
        }
    };
    if constexpr (NR <= 4) {
#pragma unroll
        for (int k = 0; k < NR; k++) long_row(rs[k], re[k]);
    } else {
#pragma unroll 1
        for (int k = 0; k < NR; k++) long_row(rs[k], re[k]);
    }
    // fold (smallest, runner-up, position) over the 8 lanes
#pragma unroll
    for (int off = 1; off < kCoop; off <<= 1) {
        const float o1 = __shfl_xor(m1, off, kWave), o2 = __shfl_xor(m2, off, kWave);
        const uint32_t op = (uint32_t)__shfl_xor((int)p1, off, kWave);
        const bool lt = o1 < m1;
        m2 = fminf(fminf(m2, o2), fmaxf(m1, o1));                // second smallest of the two sorted pairs
        p1 = lt ? op : p1;
        m1 = fminf(m1, o1);
    }
    bd = __builtin_huge_val();
    bi = kNoIndex;
    if (m1 < __builtin_huge_valf()) {
        if (m2 > m1 * (1.0f + 0x1p-19f) + 0x1p-90f) {            // unique within the fp32 error band: it is the exact winner
            const float4 W = pts[p1];
            bd = dist2((double)W.x, (double)W.y, (double)W.z, qx, qy, qz);
            bi = __float_as_uint(W.w);
        } else {                                                  // near-ties / duplicates: exact (d2, index) order decides
#pragma unroll 1
            for (int k = 0; k < NR; k++) coop_scan_exact(pts, rs[k], re[k], sub, qx, qy, qz, bd, bi);
            coop_argmin8(bd, bi);
        }
    }
}

// The search itself, shared by the batch kernel and the low-latency inflation kernel: all 8 lanes of
// a group call it with the same query and their own `sub`; every lane returns the same (bd, bi).
template <bool COUNT>
__device__ __forceinline__ void coop_nn_search(const GridDesc &G, const float4 *__restrict__ pts,
                                               const uint32_t *__restrict__ cell_start, float qxf, float qyf, float qzf,
                                               uint32_t sub, double &bd, uint32_t &bi, uint32_t &npts, uint32_t &nruns)
{
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    const int cx = cell_coord(qxf, G.ox, G.inv_h, G.gx);
    const int cy = cell_coord(qyf, G.oy, G.inv_h, G.gy);
    const int cz = cell_coord(qzf, G.oz, G.inv_h, G.gz);
    bd = __builtin_huge_val();
    bi = kNoIndex;
    if (G.octant_first) {   // ---- stage 0: the 2x2x2 block of cells on the query's side of its own cell = 4 x-runs ----
        // About 27/8 fewer points and 9/4 fewer rows than the 3x3x3 cube; it decides the query whenever the best point found is
        // closer than the block's nearest face that still has cells behind it (>= half a cell away by construction).  Which
        // side is taken only affects how often that happens, never the result: the bound below is computed from the block
        // actually scanned.
        const float fx = (qxf - G.ox) * G.inv_h - (float)cx, fy = (qyf - G.oy) * G.inv_h - (float)cy, fz = (qzf - G.oz) * G.inv_h - (float)cz;
        const int xa = max(fx < 0.5f ? cx - 1 : cx, 0), xb = min(fx < 0.5f ? cx : cx + 1, G.gx - 1);
        const int ya = max(fy < 0.5f ? cy - 1 : cy, 0), yb = min(fy < 0.5f ? cy : cy + 1, G.gy - 1);
        const int za = max(fz < 0.5f ? cz - 1 : cz, 0), zb = min(fz < 0.5f ? cz : cz + 1, G.gz - 1);
        uint32_t rs[4], re[4];
        {
            const int ri = (int)sub & 3;                              // lanes 4..7 repeat lanes 0..3 (same addresses: no extra access)
            const bool ok = !((ri >> 1) && zb == za) && !((ri & 1) && yb == ya);
            const uint32_t row = cell_lin(G, 0, (ri & 1) ? yb : ya, (ri >> 1) ? zb : za);
            const uint32_t a = cell_start[row + xa], b = cell_start[row + xb + 1];
            const uint32_t my_s = a, my_e = ok ? b : a;
            if (COUNT && sub < 4) { npts += my_e - my_s; nruns += ok ? 1u : 0u; }
#pragma unroll
            for (int k = 0; k < 4; k++) { rs[k] = (uint32_t)__shfl((int)my_s, k, kCoop); re[k] = (uint32_t)__shfl((int)my_e, k, kCoop); }
        }
        coop_screen_rows<4, 2>(pts, rs, re, sub, qxf, qyf, qzf, qx, qy, qz, bd, bi);
        double bound = __builtin_huge_val();
        if (xa > 0) bound = fmin(bound, qx - (G.oxd + (double)xa * G.hd));
        if (xb < G.gx - 1) bound = fmin(bound, (G.oxd + (double)(xb + 1) * G.hd) - qx);
        if (ya > 0) bound = fmin(bound, qy - (G.oyd + (double)ya * G.hd));
        if (yb < G.gy - 1) bound = fmin(bound, (G.oyd + (double)(yb + 1) * G.hd) - qy);
        if (za > 0) bound = fmin(bound, qz - (G.ozd + (double)za * G.hd));
        if (zb < G.gz - 1) bound = fmin(bound, (G.ozd + (double)(zb + 1) * G.hd) - qz);
        if (bound == __builtin_huge_val()) return;                    // the block covers the whole grid
        bound -= G.hd * (1.0 / 256.0);                                // same slack as cube_bound (fp32 cell assignment)
        if (bound > 0.0 && bd <= bound * bound) return;
    }
    {   // ---- first cube: 3x3x3 cells = 9 x-runs ----
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.gx - 1);
        // lane `sub` fetches row `sub`'s bounds, lane 0 also row 8
        uint32_t my_s = 0, my_e = 0, s8 = 0, e8 = 0;
        {
            const int zz = cz + (int)sub / 3 - 1, yy = cy + (int)sub % 3 - 1;
            const bool ok = zz >= 0 && zz < G.gz && yy >= 0 && yy < G.gy;
            const uint32_t row = ok ? cell_lin(G, 0, yy, zz) : 0u;
            const uint32_t a = cell_start[row + x0], b = cell_start[row + x1 + 1];
            my_s = a;
            my_e = ok ? b : a;
            const int z8 = cz + 1, y8 = cy + 1;
            const bool ok8 = z8 < G.gz && y8 < G.gy;
            const uint32_t row8 = ok8 ? cell_lin(G, 0, y8, z8) : 0u;
            const uint32_t a8 = cell_start[row8 + x0], b8 = cell_start[row8 + x1 + 1];   // same address in all 8 lanes: one access
            s8 = a8;
            e8 = ok8 ? b8 : a8;
            if (COUNT && sub == 0) {
                npts += e8 - s8; nruns += ok8 ? 1u : 0u;
            }
            if (COUNT) {
                // every lane adds its own row; summed over the wave at the end
                npts += my_e - my_s; nruns += ok ? 1u : 0u;
            }
        }
        uint32_t rs[9], re[9];
#pragma unroll
        for (int k = 0; k < 8; k++) { rs[k] = (uint32_t)__shfl((int)my_s, k, kCoop); re[k] = (uint32_t)__shfl((int)my_e, k, kCoop); }
        rs[8] = s8; re[8] = e8;

        coop_screen_rows<9, 1>(pts, rs, re, sub, qxf, qyf, qzf, qx, qy, qz, bd, bi);
    }
    for (int r = 1;; r++) {
        if (r > 1) {   // ---- shell r (rare on dense clouds): rows walked in the same order by the whole group ----
            coop_scan_cube_or_shell<COUNT>(G, pts, cell_start, cx, cy, cz, r, false, sub, qx, qy, qz, bd, bi, npts, nruns);
            coop_argmin8(bd, bi);
        }
        const double bound = cube_bound(G, cx, cy, cz, r, qx, qy, qz);
        if (bound == __builtin_huge_val() || (bound > 0.0 && bd <= bound * bound)) break;
    }
}

// ---- the same search with a WAVE-cooperative fallback (the batch kernel's default) -------------------------------------------
// Stage 0 decides ~99 % of the queries at 6 points per cell, but a wave holds 8 queries and used to run the 3x3x3 cube for all of
// them -- 8 lanes per query, the 9 rows of ~18 points taken 8 at a time -- as soon as ONE was undecided: 0.99^8 = 8 % of the waves,
// 15 % of the kernel's time (18 of 124 us on the 10 M-point cloud).  Here the whole wave turns to each undecided query in turn:
// 9 lanes fetch the rows' bounds, every lane takes one point of each row (all nine loads in flight), one 6-step fold over the 64
// lanes, the exact winner.  The queries that even the cube does not decide (~1e-4) finish with the 8-lane shell walk as before.

// Is a query still undecided after its 2x2x2 block [xa..xb] x [ya..yb] x [za..zb] gave best distance bd?  (f = the query's fractional
// position in its cell (cx, cy, cz).)
__device__ __forceinline__ bool block_leaves_undecided(const GridDesc &G, int cx, int cy, int cz, int xa, int xb, int ya, int yb, int za, int zb,
                                                       float fx, float fy, float fz, double qx, double qy, double qz, double bd)
{
    // Quick accept in fp32 (the six fp64 face distances below cost ~6 of the kernel's 120 us): distances to the block's faces in cell
    // units from the fractional position already at hand, the nearest one with cells behind it shortened by 1/64 cell -- far more
    // than the fp32 rounding of fx (< 2e-4 cells) plus the exact test's own 1/256 slack -- so whatever passes here passes the exact
    // test too; the ~2 % of queries in that 1/64-cell band, the undecided ones and queries outside the grid take the exact test.
    {
        const float inf = __builtin_huge_valf();
        float bc = inf;
        bc = fminf(bc, xa > 0 ? fx + (float)(cx - xa) : inf);
        bc = fminf(bc, xb < G.gx - 1 ? (float)(xb + 1 - cx) - fx : inf);
        bc = fminf(bc, ya > 0 ? fy + (float)(cy - ya) : inf);
        bc = fminf(bc, yb < G.gy - 1 ? (float)(yb + 1 - cy) - fy : inf);
        bc = fminf(bc, za > 0 ? fz + (float)(cz - za) : inf);
        bc = fminf(bc, zb < G.gz - 1 ? (float)(zb + 1 - cz) - fz : inf);
        // only for queries inside the grid (|f| small): the error bound above is for coordinates of at most ~1024 cells
        if (bc < inf && fmaxf(fmaxf(fabsf(fx), fabsf(fy)), fabsf(fz)) < 2.0f) {
            const double b = (double)((bc - 0.015625f) * (float)G.hd * 0.999999f);
            if (b > 0.0 && bd <= b * b) return false;
        }
    }
    double bound = __builtin_huge_val();
    if (xa > 0) bound = fmin(bound, qx - (G.oxd + (double)xa * G.hd));
    if (xb < G.gx - 1) bound = fmin(bound, (G.oxd + (double)(xb + 1) * G.hd) - qx);
    if (ya > 0) bound = fmin(bound, qy - (G.oyd + (double)ya * G.hd));
    if (yb < G.gy - 1) bound = fmin(bound, (G.oyd + (double)(yb + 1) * G.hd) - qy);
    if (za > 0) bound = fmin(bound, qz - (G.ozd + (double)za * G.hd));
    if (zb < G.gz - 1) bound = fmin(bound, (G.ozd + (double)(zb + 1) * G.hd) - qz);
    if (bound == __builtin_huge_val()) return false;              // the block covers the whole grid
    bound -= G.hd * (1.0 / 256.0);                                // same slack as cube_bound (fp32 cell assignment)
    return !(bound > 0.0 && bd <= bound * bound);
}

// stage 0 of coop_nn_search alone; returns true when the query is still undecided
template <bool COUNT>
__device__ __forceinline__ bool coop_stage0(const GridDesc &G, const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                            float qxf, float qyf, float qzf, uint32_t sub, double &bd, uint32_t &bi, uint32_t &npts,
                                            uint32_t &nruns)
{
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    const int cx = cell_coord(qxf, G.ox, G.inv_h, G.gx);
    const int cy = cell_coord(qyf, G.oy, G.inv_h, G.gy);
    const int cz = cell_coord(qzf, G.oz, G.inv_h, G.gz);
    const float fx = (qxf - G.ox) * G.inv_h - (float)cx, fy = (qyf - G.oy) * G.inv_h - (float)cy, fz = (qzf - G.oz) * G.inv_h - (float)cz;
    const int xa = max(fx < 0.5f ? cx - 1 : cx, 0), xb = min(fx < 0.5f ? cx : cx + 1, G.gx - 1);
    const int ya = max(fy < 0.5f ? cy - 1 : cy, 0), yb = min(fy < 0.5f ? cy : cy + 1, G.gy - 1);
    const int za = max(fz < 0.5f ? cz - 1 : cz, 0), zb = min(fz < 0.5f ? cz : cz + 1, G.gz - 1);
    uint32_t rs[4], re[4];
    if (G.blocks) {                                               // the corner's entry of the block table: one line, every lane reads it itself
        const uint32_t u = (uint32_t)(fx < 0.5f ? cx : cx + 1), v = (uint32_t)(fy < 0.5f ? cy : cy + 1), w = (uint32_t)(fz < 0.5f ? cz : cz + 1);
        const uint4 *ent = G.blocks + 2u * ((w * ((uint32_t)G.gy + 1u) + v) * ((uint32_t)G.gx + 1u) + u);
        const uint4 S = ent[0], E = ent[1];
        rs[0] = S.x; rs[1] = S.y; rs[2] = S.z; rs[3] = S.w;
        re[0] = E.x; re[1] = E.y; re[2] = E.z; re[3] = E.w;
        if (COUNT && sub < 4) {
            const int ri = (int)sub;
            const bool ok = !((ri >> 1) && zb == za) && !((ri & 1) && yb == ya);
            npts += re[ri] - rs[ri]; nruns += ok ? 1u : 0u;
        }
    } else {
        const int ri = (int)sub & 3;                              // lanes 4..7 repeat lanes 0..3 (same addresses: no extra access)
        const bool ok = !((ri >> 1) && zb == za) && !((ri & 1) && yb == ya);
        const uint32_t row = cell_lin(G, 0, (ri & 1) ? yb : ya, (ri >> 1) ? zb : za);
        const uint32_t a = cell_start[row + xa], b = cell_start[row + xb + 1];
        const uint32_t my_s = a, my_e = ok ? b : a;
        if (COUNT && sub < 4) { npts += my_e - my_s; nruns += ok ? 1u : 0u; }
#pragma unroll
        for (int k = 0; k < 4; k++) { rs[k] = (uint32_t)__shfl((int)my_s, k, kCoop); re[k] = (uint32_t)__shfl((int)my_e, k, kCoop); }
    }
#ifdef PCT_AB_OPEN_STAGE0          // measured slower: see coop_screen_rows
    coop_screen_rows<4, 2, true>(pts, rs, re, sub, qxf, qyf, qzf, qx, qy, qz, bd, bi);
#else
    coop_screen_rows<4, 2>(pts, rs, re, sub, qxf, qyf, qzf, qx, qy, qz, bd, bi);
#endif
    return block_leaves_undecided(G, cx, cy, cz, xa, xb, ya, yb, za, zb, fx, fy, fz, qx, qy, qz, bd);
}

// the 3x3x3 cube around the cell of a WAVE-UNIFORM query, searched by all 64 lanes; every lane returns the same exact (bd, bi)
// = the winner by (d2, index) among the cube's points, (+inf, none) for an empty cube
template <bool COUNT>
__device__ __forceinline__ void wave_cube_search(const GridDesc &G, const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                                 float qxf, float qyf, float qzf, double &bd, uint32_t &bi, uint32_t &npts, uint32_t &nruns)
{
    const uint32_t lane = threadIdx.x & 63;
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    const int cx = cell_coord(qxf, G.ox, G.inv_h, G.gx);
    const int cy = cell_coord(qyf, G.oy, G.inv_h, G.gy);
    const int cz = cell_coord(qzf, G.oz, G.inv_h, G.gz);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.gx - 1);
    uint32_t my_s, my_e;
    {
        const int l = lane < 9 ? (int)lane : 0;                       // lanes 9.. repeat lane 0's addresses
        const int zz = cz + l / 3 - 1, yy = cy + l % 3 - 1;
        const bool ok = zz >= 0 && zz < G.gz && yy >= 0 && yy < G.gy;
        const uint32_t row = ok ? cell_lin(G, 0, yy, zz) : 0u;
        const uint32_t a = cell_start[row + x0], b = cell_start[row + x1 + 1];
        my_s = a;
        my_e = ok ? b : a;
        if (COUNT && lane < 9) { npts += my_e - my_s; nruns += ok ? 1u : 0u; }
    }
    uint32_t rs[9], re[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {                                     // wave-uniform: scalar registers
        rs[k] = (uint32_t)__builtin_amdgcn_readlane((int)my_s, k);
        re[k] = (uint32_t)__builtin_amdgcn_readlane((int)my_e, k);
    }
    float m1 = __builtin_huge_valf(), m2 = __builtin_huge_valf();
    uint32_t p1 = 0;
    float4 P[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const uint32_t a = rs[k], b = re[k];
        P[k] = pts[min(a + lane, b > a ? b - 1 : 0u)];                // empty row / lane beyond the row: any valid slot, masked below
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const uint32_t p = rs[k] + lane;
        const float dx = P[k].x - qxf, dy = P[k].y - qyf, dz = P[k].z - qzf;
        float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        d = (p < re[k]) ? d : __builtin_huge_valf();
        const bool lt = d < m1;
        m2 = lt ? m1 : fminf(m2, d);
        p1 = lt ? p : p1;
        m1 = fminf(m1, d);
    }
#pragma unroll 1
    for (int k = 0; k < 9; k++)                                       // rows of more than 64 points
        for (uint32_t p = rs[k] + 64u + lane; p < re[k]; p += 64u) {
            const float4 Pp = pts[p];
            const float dx = Pp.x - qxf, dy = Pp.y - qyf, dz = Pp.z - qzf;
            const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            const bool lt = d < m1;
            m2 = lt ? m1 : fminf(m2, d);
            p1 = lt ? p : p1;
            m1 = fminf(m1, d);
        }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {                          // (smallest, runner-up, position) over the wave
        const float o1 = __shfl_xor(m1, off, kWave), o2 = __shfl_xor(m2, off, kWave);
        const uint32_t op = (uint32_t)__shfl_xor((int)p1, off, kWave);
        const bool lt = o1 < m1;
        m2 = fminf(fminf(m2, o2), lt ? m1 : o1);
        p1 = lt ? op : p1;
        m1 = fminf(m1, o1);
    }
    bd = __builtin_huge_val();
    bi = kNoIndex;
    if (m1 < __builtin_huge_valf()) {                                 // wave-uniform
        if (m2 > m1 * (1.0f + 0x1p-19f) + 0x1p-90f) {                // unique within the fp32 error band: it is the exact winner
            const float4 W = pts[p1];
            bd = dist2((double)W.x, (double)W.y, (double)W.z, qx, qy, qz);
            bi = __float_as_uint(W.w);
        } else {                                                      // near-ties / duplicates: exact (d2, index) order decides
#pragma unroll 1
            for (int k = 0; k < 9; k++)
                for (uint32_t p = rs[k] + lane; p < re[k]; p += 64u) {
                    const float4 Pp = pts[p];
                    const double d2 = dist2((double)Pp.x, (double)Pp.y, (double)Pp.z, qx, qy, qz);
                    const uint32_t id = __float_as_uint(Pp.w);
                    if (better(d2, id, bd, bi)) { bd = d2; bi = id; }
                }
            wave_argmin(bd, bi);
        }
    }
}

// after the cube: shells of growing radius until the termination bound holds (8 lanes per query, rare on dense clouds)
template <bool COUNT>
__device__ __forceinline__ void coop_finish_shells(const GridDesc &G, const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                                   float qxf, float qyf, float qzf, uint32_t sub, double &bd, uint32_t &bi, uint32_t &npts,
                                                   uint32_t &nruns)
{
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    const int cx = cell_coord(qxf, G.ox, G.inv_h, G.gx);
    const int cy = cell_coord(qyf, G.oy, G.inv_h, G.gy);
    const int cz = cell_coord(qzf, G.oz, G.inv_h, G.gz);
    for (int r = 1;; r++) {
        if (r > 1) {
            coop_scan_cube_or_shell<COUNT>(G, pts, cell_start, cx, cy, cz, r, false, sub, qx, qy, qz, bd, bi, npts, nruns);
            coop_argmin8(bd, bi);
        }
        const double bound = cube_bound(G, cx, cy, cz, r, qx, qy, qz);
        if (bound == __builtin_huge_val() || (bound > 0.0 && bd <= bound * bound)) break;
    }
}

// One query per group of 8 lanes, every lane of the wave in step (live = the group has a query): stage 0, then the whole wave on each
// undecided query's cube in turn, then the 8-lane shell walk for what even the cube leaves open.  Every lane of a group returns its
// query's exact (bd, bi).
template <bool COUNT>
__device__ __forceinline__ void coop_wave_search(const GridDesc &G, const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start, bool live,
                                                 float qxf, float qyf, float qzf, uint32_t sub, double &bd, uint32_t &bi, uint32_t &npts, uint32_t &nruns)
{
    bd = __builtin_huge_val();
    bi = kNoIndex;
    bool undecided = false;
    if (live) undecided = coop_stage0<COUNT>(G, pts, cell_start, qxf, qyf, qzf, sub, bd, bi, npts, nruns);
    unsigned long long todo = __builtin_amdgcn_ballot_w64(undecided && sub == 0);
    while (todo) {                                    // wave-uniform: one undecided query at a time, all 64 lanes on it
        const int g = __builtin_ctzll(todo);
        todo &= todo - 1;
        const float bx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qxf), g)),
                    by = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qyf), g)),
                    bz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qzf), g));
        double cbd;
        uint32_t cbi;
        wave_cube_search<COUNT>(G, pts, cell_start, bx, by, bz, cbd, cbi, npts, nruns);
        if (((threadIdx.x & 63) >> 3) == (uint32_t)(g >> 3)) { bd = cbd; bi = cbi; }
    }
    if (live && undecided) coop_finish_shells<COUNT>(G, pts, cell_start, qxf, qyf, qzf, sub, bd, bi, npts, nruns);
}

template <bool COUNT, bool WAVE = false>
// Occupancy target 7 waves per SIMD, as minimum AND maximum: with 8 allowed the scheduler keeps the kernel at 64 VGPRs by issuing the
// eight record loads of stage 0 two at a time (four dependent round trips); capped at 7 it takes 72 VGPRs and issues all eight before the
// first use.  Headline step 0.1198-0.1210 -> 0.1163-0.1179 ms (6 waves: 0.122, 5: 0.134; profiles/r03_ab_occupancy_cap.txt).  Raising
// only the minimum (round 2's "7 waves" experiment) never changed the code: the scheduler still aimed for 8.
#ifndef PCT_AB_COOP_WAVES
#define PCT_AB_COOP_WAVES 7
#endif
#ifndef PCT_AB_COOP_WAVES_MAX
#define PCT_AB_COOP_WAVES_MAX 7
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCT_AB_COOP_WAVES, PCT_AB_COOP_WAVES_MAX))) void nn_grid_coop_kernel(GridDesc G, const float4 *__restrict__ pts,
                                                           const uint32_t *__restrict__ cell_start,
                                                           const float *__restrict__ q, uint32_t Q, uint32_t index_base,
                                                           const float4 *__restrict__ qsorted,
                                                           uint32_t *__restrict__ out_idx, double *__restrict__ out_d2,
                                                           WorkCounters *__restrict__ work, int sorted_out)
{
    const uint32_t sub = threadIdx.x & (kCoop - 1);
    const uint32_t bslot = qsorted ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t slot = bslot * (256 / kCoop) + (threadIdx.x / kCoop);
    uint32_t npts = 0, nruns = 0;
    if (WAVE && G.octant_first) {                     // wave-cooperative fallback (see above); every lane of the wave stays in step
        const bool live = slot < Q;
        uint32_t t = slot;
        float qxf = 0.0f, qyf = 0.0f, qzf = 0.0f;
        if (live) {
            if (qsorted) {
                const float4 R = qsorted[slot];
                qxf = R.x; qyf = R.y; qzf = R.z; t = sorted_out ? slot : __float_as_uint(R.w);
            } else {
                qxf = q[3 * t]; qyf = q[3 * t + 1]; qzf = q[3 * t + 2];
            }
        }
        double bd;
        uint32_t bi;
        coop_wave_search<COUNT>(G, pts, cell_start, live, qxf, qyf, qzf, sub, bd, bi, npts, nruns);
        if (live && sub == 0) {
            out_idx[t] = (bi == kNoIndex) ? kNoIndex : bi + index_base;
            out_d2[t] = bd;
        }
    } else if (slot < Q) {                            // uniform within a group of 8 lanes
        uint32_t t = slot;
        float qxf, qyf, qzf;
        if (qsorted) {                                // binned batch: query and output slot in one record
            const float4 R = qsorted[slot];
            qxf = R.x; qyf = R.y; qzf = R.z; t = sorted_out ? slot : __float_as_uint(R.w);     // sorted_out: results stay in sorted order
        } else {
            qxf = q[3 * t]; qyf = q[3 * t + 1]; qzf = q[3 * t + 2];
        }
        double bd;
        uint32_t bi;
        coop_nn_search<COUNT>(G, pts, cell_start, qxf, qyf, qzf, sub, bd, bi, npts, nruns);
        if (sub == 0) {
            out_idx[t] = (bi == kNoIndex) ? kNoIndex : bi + index_base;
            out_d2[t] = bd;
        }
    }
    if (COUNT) {
        unsigned long long a = npts, b = nruns;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += (unsigned long long)__shfl_xor((long long)a, off, kWave);
            b += (unsigned long long)__shfl_xor((long long)b, off, kWave);
        }
        if ((threadIdx.x & 63) == 0) { WorkCounters *w = work + (blockIdx.x & (kWorkSlots - 1)); atomicAdd(&w->points, a); atomicAdd(&w->cells, b); }
    }
}

// Cooperative radius count: 8 lanes per query, the rows of the ball's bounding box: the run bounds of up to 16 rows fetched at once
// (two per lane), then the rows two at a time, every lane with four 16-byte loads per row in flight (32 points per row for the group)
// before the first compare; longer rows finish in a tail loop.  Every distance is the exact fp64 one (kdtree.c:273,
// d2 <= r*r inclusive).  The box is [q - r - pad, q + r + pad] with pad = h/100: a point within r of q can only sit in a
// cell of that range, because the fp32 cell assignment errs by < 4e-4 cells (same argument as cube_bound's h/256 slack).
template <bool COUNT>
__global__ __launch_bounds__(256) void count_grid_coop_kernel(GridDesc G, const float4 *__restrict__ pts,
                                                              const uint32_t *__restrict__ cell_start,
                                                              const float *__restrict__ q, const float *__restrict__ rad, uint32_t Q,
                                                              const float4 *__restrict__ qsorted, uint32_t *__restrict__ count,
                                                              WorkCounters *__restrict__ work)
{
    const uint32_t sub = threadIdx.x & (kCoop - 1);
    const uint32_t bslot = qsorted ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t slot = bslot * (256 / kCoop) + (threadIdx.x / kCoop);
    uint32_t npts = 0, nruns = 0;
    if (slot < Q) {                                   // uniform within a group of 8 lanes
        uint32_t t = slot;
        float qxf, qyf, qzf;
        if (qsorted) {
            const float4 R = qsorted[slot];
            qxf = R.x; qyf = R.y; qzf = R.z; t = __float_as_uint(R.w);
        } else {
            qxf = q[3 * t]; qyf = q[3 * t + 1]; qzf = q[3 * t + 2];
        }
        const float rf = rad[t];
        const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
        const double r2 = (double)rf * (double)rf;
        const float pad = fabsf(rf) + 0.01f * (1.0f / G.inv_h);   // r enters only squared (kdtree.c:273)
        const int x0 = cell_coord(qxf - pad, G.ox, G.inv_h, G.gx), x1 = cell_coord(qxf + pad, G.ox, G.inv_h, G.gx);
        const int y0 = cell_coord(qyf - pad, G.oy, G.inv_h, G.gy), y1 = cell_coord(qyf + pad, G.oy, G.inv_h, G.gy);
        const int z0 = cell_coord(qzf - pad, G.oz, G.inv_h, G.gz), z1 = cell_coord(qzf + pad, G.oz, G.inv_h, G.gz);
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        uint32_t c = 0;
        // Rows of the box: the run bounds of up to 16 rows in ONE trip (two per lane), then the rows two at a time with 32 points of each
        // in flight (a row of 3-4 cells holds ~20): a query's 9-16 rows cost 1 + rows/2 dependent round trips.  (The first form took the
        // rows four at a time with 16 points each and finished every row in a tail loop, one trip per row: ~18 trips; 1 M counts of r = 1 on
        // the 10 M-point cloud 0.51 -> 0.43 ms, profiles/r03_ab_count_rows.txt.)
        for (int base = 0; base < nrows; base += 16) {
            uint32_t bs[2], be[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k = base + 8 * h + (int)sub;
                const bool ok = k < nrows;
                const uint32_t row = ok ? cell_lin(G, 0, y0 + k % ny, z0 + k / ny) : 0u;
                const uint32_t a = cell_start[row + x0], b = cell_start[row + x1 + 1];
                bs[h] = a; be[h] = ok ? b : a;
                if (COUNT) { npts += be[h] - bs[h]; nruns += ok ? 1u : 0u; }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {                         // rows base + 2 i, base + 2 i + 1
                if (base + 2 * i >= nrows) break;                 // uniform within the group of 8 lanes
                uint32_t rs[2], re[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int r = 2 * i + j;                      // compile-time: which register, which lane
                    rs[j] = (uint32_t)__shfl((int)bs[r >> 3], r & 7, kCoop);
                    re[j] = (uint32_t)__shfl((int)be[r >> 3], r & 7, kCoop);
                }
                float4 P[2][4];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const uint32_t last = re[j] > rs[j] ? re[j] - 1 : 0u;
#pragma unroll
                    for (int d = 0; d < 4; d++) P[j][d] = pts[min(rs[j] + sub + kCoop * d, last)];
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const bool in = rs[j] + sub + kCoop * d < re[j];
                        c += (in && dist2((double)P[j][d].x, (double)P[j][d].y, (double)P[j][d].z, qx, qy, qz) <= r2) ? 1u : 0u;
                    }
                }
#pragma unroll 1
                for (int j = 0; j < 2; j++)
                    for (uint32_t p = rs[j] + 4 * kCoop + sub; p < re[j]; p += kCoop) {
                        const float4 Pp = pts[p];
                        c += dist2((double)Pp.x, (double)Pp.y, (double)Pp.z, qx, qy, qz) <= r2 ? 1u : 0u;
                    }
            }
        }
#pragma unroll
        for (int off = 1; off < kCoop; off <<= 1) c += (uint32_t)__shfl_xor((int)c, off, kWave);
        if (sub == 0) count[t] = c;
    }
    if (COUNT) {
        unsigned long long a = npts, b = nruns;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += (unsigned long long)__shfl_xor((long long)a, off, kWave);
            b += (unsigned long long)__shfl_xor((long long)b, off, kWave);
        }
        if ((threadIdx.x & 63) == 0) { WorkCounters *w = work + (blockIdx.x & (kWorkSlots - 1)); atomicAdd(&w->points, a); atomicAdd(&w->cells, b); }
    }
}

// radius count through the grid, one lane per query (PCT_GRID_COOP=0): every cell row overlapping the ball's bounding box.
template <bool COUNT>
__global__ __launch_bounds__(256) void count_grid_kernel(GridDesc G, const float4 *__restrict__ pts,
                                                         const uint32_t *__restrict__ cell_start,
                                                         const float *__restrict__ q, const float *__restrict__ rad,
                                                         uint32_t Q, const uint32_t *__restrict__ perm,
                                                         uint32_t *__restrict__ count, WorkCounters *__restrict__ work)
{
    const uint32_t slot = (perm ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t npts = 0, nruns = 0;
    if (slot < Q) {
        const uint32_t t = perm ? perm[slot] : slot;
        const float qxf = q[3 * t], qyf = q[3 * t + 1], qzf = q[3 * t + 2], rf = rad[t];
        const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
        const double r2 = (double)rf * (double)rf;
        uint32_t c = 0;
        {
            const float pad = fabsf(rf) + 0.01f * (1.0f / G.inv_h);   // r enters only squared (kdtree.c:273)
            int x0 = cell_coord(qxf - pad, G.ox, G.inv_h, G.gx), x1 = cell_coord(qxf + pad, G.ox, G.inv_h, G.gx);
            int y0 = cell_coord(qyf - pad, G.oy, G.inv_h, G.gy), y1 = cell_coord(qyf + pad, G.oy, G.inv_h, G.gy);
            int z0 = cell_coord(qzf - pad, G.oz, G.inv_h, G.gz), z1 = cell_coord(qzf + pad, G.oz, G.inv_h, G.gz);
            // (no extra cell of margin: pad's h/100 already covers the < 4e-4-cell error of the fp32 cell assignment)
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
        if ((threadIdx.x & 63) == 0) { WorkCounters *w = work + (blockIdx.x & (kWorkSlots - 1)); atomicAdd(&w->points, a); atomicAdd(&w->cells, b); }
    }
}

// =====================================================================================
// 4b. Low-latency ("express") kernels: ONE launch per call, query passed by value or read from
//     host-mapped memory, results written straight to host-mapped memory.  The planner's RRT*
//     loop issues single queries in sequence (corridor_finder.cpp:719-756); there the cost is
//     launch + copy latency, not throughput.
// =====================================================================================
struct ExpressOut { double d2; double radius; uint32_t idx; uint32_t count; };

// Completion word of an express launch.  hipStreamSynchronize costs ~20 us of host time however short the kernel
// (profiles/r01_corridor_rocprof_summary.txt); here the kernel itself tells the host it is done: the thread that wrote a block's
// results fences them to system scope and takes a ticket, the block holding the last ticket stores `value` (release, system
// scope) into a word of host-mapped memory the host spins on.  seq == nullptr: no signalling (the caller synchronises).
struct ExpressSignal { uint32_t *counter; uint32_t *seq; uint32_t value; };

// called by the ONE thread that stored the block's host-visible results, after the stores
__device__ __forceinline__ void express_done(const ExpressSignal &S)
{
    if (!S.seq) return;
    __threadfence_system();
    bool last = gridDim.x == 1;
    if (!last) {
        last = atomicAdd(S.counter, 1u) == gridDim.x - 1;
        if (last) { atomicExch(S.counter, 0u); __threadfence_system(); }
    }
    if (last) __hip_atomic_store(S.seq, S.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the same for kernels in which every thread stores host-visible data: all fence, the block meets, thread 0 signals
__device__ __forceinline__ void express_done_block(const ExpressSignal &S)
{
    if (!S.seq) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) express_done(S);
}

// Results of a host-buffer batch leave the card by this kernel instead of two DMA copies + a stream synchronise: it stores
// (index, d2) -- or counts -- into host-mapped memory, every storing thread fences at system scope, and the last block releases the
// sequence word the host spins on (express_done_block).  A 4096-query batch saves ~80 us of copy setup + synchronise this way.
// ... and the queries enter the same way: one pass over the host-mapped staging buffer into device memory (the batch's kernels
// read their queries many times; only this copy crosses the bus)
__global__ __launch_bounds__(256) void import_floats_kernel(const float *__restrict__ h_src, uint32_t n, float *__restrict__ dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = h_src[i];
}

__global__ __launch_bounds__(256) void export_results_kernel(const uint32_t *__restrict__ idx, const double *__restrict__ d2, uint32_t Q,
                                                             uint32_t *__restrict__ h_idx, double *__restrict__ h_d2, ExpressSignal sig)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Q) {
        h_idx[i] = idx[i];
        if (d2) h_d2[i] = d2[i];
    }
    express_done_block(sig);
}

// one block, exact fp64 brute force over a small cloud (the RRT* node set of the kd_* drop-in).  Besides the winner (lowest index
// among the minima) it reports HOW MANY points attain the minimum (out->count): the kd_* drop-in resolves an exact tie the way
// the reference's tree walk does (kdtree_gpu.cpp reference_tie_winner) and only then needs the tied set.
__device__ __forceinline__ void tie_merge(double &d, uint32_t &i, uint32_t &c, double od, uint32_t oi, uint32_t oc)
{
    if (od < d) { d = od; i = oi; c = oc; }
    else if (od == d) { c += oc; i = min(i, oi); }
}

__global__ __launch_bounds__(1024) void nn_small_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ z, uint32_t n, double qx, double qy, double qz,
                                                        uint32_t index_base, ExpressOut *__restrict__ out, ExpressSignal sig)
{
    double bd = __builtin_huge_val();
    uint32_t bi = kNoIndex, bc = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const double d2 = dist2((double)x[i], (double)y[i], (double)z[i], qx, qy, qz);
        if (d2 < bd) { bd = d2; bi = i; bc = 1; }
        else if (d2 == bd) bc++;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(bd, off, kWave);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off, kWave), oc = (uint32_t)__shfl_xor((int)bc, off, kWave);
        tie_merge(bd, bi, bc, od, oi, oc);
    }
    __shared__ double s_d[16];
    __shared__ uint32_t s_i[16], s_c[16];
    if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = bd; s_i[threadIdx.x >> 6] = bi; s_c[threadIdx.x >> 6] = bc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) tie_merge(bd, bi, bc, s_d[w], s_i[w], s_c[w]);
        out->d2 = bd;
        out->idx = (bi == kNoIndex) ? kNoIndex : bi + index_base;
        out->count = bc;
        express_done(sig);
    }
}

// one block: ids of all points with d2 <= r2 (arrival order; the host sorts), count in out->count
__global__ __launch_bounds__(1024) void radius_small_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, uint32_t n, double qx, double qy, double qz,
                                                            double r2, uint32_t index_base, uint32_t *__restrict__ ids, uint32_t cap,
                                                            ExpressOut *__restrict__ out, ExpressSignal sig)
{
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024)
        if (dist2((double)x[i], (double)y[i], (double)z[i], qx, qy, qz) <= r2) {
            const uint32_t pos = atomicAdd(&s_n, 1u);
            if (pos < cap) ids[pos] = i + index_base;
        }
    __syncthreads();
    if (threadIdx.x == 0) out->count = s_n;
    express_done_block(sig);
}

// =====================================================================================
// 5. Planner arithmetic around the NN: sphere inflation and the sampled Bezier check
// =====================================================================================
// batched forms for speculative RRT* expansion: one 256-thread block per query over a small cloud
__global__ __launch_bounds__(256) void nn_small_batch_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, uint32_t n,
                                                             const double *__restrict__ q, uint32_t index_base,
                                                             ExpressOut *__restrict__ out, ExpressSignal sig)
{
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    const double qx = q[3 * blockIdx.x], qy = q[3 * blockIdx.x + 1], qz = q[3 * blockIdx.x + 2];
    double bd = __builtin_huge_val();
    uint32_t bi = kNoIndex;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const double d2 = dist2((double)x[i], (double)y[i], (double)z[i], qx, qy, qz);
        if (d2 < bd) { bd = d2; bi = i; }
    }
    wave_argmin(bd, bi);
    if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = bd; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (better(s_d[w], s_i[w], bd, bi)) { bd = s_d[w]; bi = s_i[w]; }
        out[blockIdx.x].d2 = bd;
        out[blockIdx.x].idx = (bi == kNoIndex) ? kNoIndex : bi + index_base;
        express_done(sig);
    }
}

// ids[block * cap_per_query + k] = k-th hit (arrival order), out[block].count = number of hits (may exceed the cap)
__global__ __launch_bounds__(256) void radius_small_batch_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                 const float *__restrict__ z, uint32_t n,
                                                                 const double *__restrict__ q, const double *__restrict__ r,
                                                                 uint32_t index_base, uint32_t *__restrict__ ids,
                                                                 uint32_t cap_per_query, ExpressOut *__restrict__ out, ExpressSignal sig)
{
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const double qx = q[3 * blockIdx.x], qy = q[3 * blockIdx.x + 1], qz = q[3 * blockIdx.x + 2];
    const double r2 = r[blockIdx.x] * r[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        if (dist2((double)x[i], (double)y[i], (double)z[i], qx, qy, qz) <= r2) {
            const uint32_t pos = atomicAdd(&s_n, 1u);
            if (pos < cap_per_query) ids[(size_t)blockIdx.x * cap_per_query + pos] = i + index_base;
        }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x].count = s_n;
    express_done_block(sig);
}

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

// the same epilogue writing one {d2, radius, idx} record per point into host-mapped memory (small batches: the host reads the
// results after one stream sync instead of issuing three device-to-host copies)
__global__ __launch_bounds__(256) void inflate_epilogue_out_kernel(InflateParams P, uint32_t n, const unsigned char *__restrict__ skip,
                                                                   int cloud_empty, const uint32_t *__restrict__ idx,
                                                                   const double *__restrict__ d2, ExpressOut *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (skip[i] || cloud_empty) {
        out[i].radius = P.max_radius - P.search_margin;
        out[i].idx = kNoIndex;
        out[i].d2 = __builtin_huge_val();
        return;
    }
    const double r = sqrt(d2[i]) - P.search_margin;
    out[i].radius = r < P.max_radius ? r : P.max_radius;
    out[i].idx = idx[i];
    out[i].d2 = d2[i];
}

// Block-wide winner: every thread passes its (d2, index) and gets back the block's best.
__device__ __forceinline__ void block_argmin256(double &d, uint32_t &i, double *s_d, uint32_t *s_i)
{
    wave_argmin(d, i);
    __syncthreads();                                   // previous use of s_d / s_i is over
    if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = d; s_i[threadIdx.x >> 6] = i; }
    __syncthreads();
    d = s_d[0]; i = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; w++)
        if (better(s_d[w], s_i[w], d, i)) { d = s_d[w]; i = s_i[w]; }
}

// Low-latency fused inflation: ONE 256-thread block per planner point (early-out test, narrowing, cell
// search, radius), arguments and results in host-mapped memory.  A single query has no batch to hide
// latency behind, so the parallelism goes across the ROWS of the cube / shell being searched: the 32
// groups of 8 lanes each take every 32nd row, and the block folds its candidates after each shell.
// stop_d2: when only the radius is wanted the search may stop once everything unseen is farther than
// max_radius + search_margin (the radius is then max_radius whatever lies beyond); +inf = exact NN.
// Same arithmetic and the same termination bound as coop_nn_search.
// INFLATE = false: plain nearest neighbour of the (fp32-valued) points in qpts -- no early-out, no radius.
// The block-wide search itself: nearest obstacle point of the fp32-narrowed (px, py, pz); every thread returns the winner.
__device__ __forceinline__ void block_nn_search(const GridDesc &G0, const float4 *__restrict__ pts0, const uint32_t *__restrict__ cs0,
                                                const CoarseLevels &C, double px, double py, double pz, double stop_d2,
                                                double *s_d, uint32_t *s_i, double &bd, uint32_t &bi)
{
    const uint32_t sub = threadIdx.x & (kCoop - 1), grp = threadIdx.x / kCoop;   // 32 groups
    const float qxf = (float)px, qyf = (float)py, qzf = (float)pz;                    // searchPoint.x = search_Pt(0), :125-128
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    bd = __builtin_huge_val();
    bi = kNoIndex;
    // steps: cube r=1 of the fine level, cube r=1 of every coarser level, then shells r=2,3,.. of the coarsest level
    for (int step = 0;; step++) {
        const int lvl = min(step, C.n);                       // 0 = fine, 1..C.n = coarse level lvl-1
        const int r = step <= C.n ? 1 : step - C.n + 1;
        const GridDesc &G = lvl == 0 ? G0 : C.G[lvl - 1];
        const float4 *pts = lvl == 0 ? pts0 : C.pts[lvl - 1];
        const uint32_t *cell_start = lvl == 0 ? cs0 : C.cell_start[lvl - 1];
        const int cx = cell_coord(qxf, G.ox, G.inv_h, G.gx);
        const int cy = cell_coord(qyf, G.oy, G.inv_h, G.gy);
        const int cz = cell_coord(qzf, G.oz, G.inv_h, G.gz);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, G.gx - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, G.gy - 1);
        const int z0 = max(cz - r, 0), z1 = min(cz + r, G.gz - 1);
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        for (int k = (int)grp; k < nrows; k += 256 / kCoop) {
            const int zz = z0 + k / ny, yy = y0 + k % ny;
            const uint32_t row = cell_lin(G, 0, yy, zz);
            if (r == 1 || zz == cz - r || zz == cz + r || yy == cy - r || yy == cy + r) {
                coop_scan_exact(pts, cell_start[row + x0], cell_start[row + x1 + 1], sub, qx, qy, qz, bd, bi);
            } else {
                if (cx - r >= 0) coop_scan_exact(pts, cell_start[row + cx - r], cell_start[row + cx - r + 1], sub, qx, qy, qz, bd, bi);
                if (cx + r <= G.gx - 1) coop_scan_exact(pts, cell_start[row + cx + r], cell_start[row + cx + r + 1], sub, qx, qy, qz, bd, bi);
            }
        }
        block_argmin256(bd, bi, s_d, s_i);
        const double bound = cube_bound(G, cx, cy, cz, r, qx, qy, qz);
        if (bound == __builtin_huge_val()) break;             // the cube covers the whole box: every point has been seen
        if (bound > 0.0 && (bd <= bound * bound || bound * bound >= stop_d2)) break;
    }
}

template <bool INFLATE>
__global__ __launch_bounds__(256) void inflate_block_kernel(GridDesc G0, const float4 *__restrict__ pts0,
                                                            const uint32_t *__restrict__ cs0, CoarseLevels C, InflateParams P,
                                                            const double *__restrict__ qpts, double stop_d2, uint32_t index_base,
                                                            ExpressOut *__restrict__ out, ExpressSignal sig)
{
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    const uint32_t slot = blockIdx.x;
    const double px = qpts[3 * slot], py = qpts[3 * slot + 1], pz = qpts[3 * slot + 2];
    if (INFLATE) {
        const double dx = px - P.sx, dy = py - P.sy, dz = pz - P.sz;
        if (sqrt(dx * dx + dy * dy + dz * dz) > P.sample_range + P.max_radius) {      // corridor_finder.cpp:115-116
            if (threadIdx.x == 0) { out[slot].radius = P.max_radius - P.search_margin; out[slot].idx = kNoIndex; out[slot].d2 = __builtin_huge_val(); express_done(sig); }
            return;
        }
    }
    double bd;
    uint32_t bi;
    block_nn_search(G0, pts0, cs0, C, px, py, pz, stop_d2, s_d, s_i, bd, bi);
    if (threadIdx.x == 0) {
        if (INFLATE) {
            const double rr = sqrt(bd) - P.search_margin;
            out[slot].radius = rr < P.max_radius ? rr : P.max_radius;
        }
        out[slot].idx = (bi == kNoIndex) ? kNoIndex : bi + index_base;
        out[slot].d2 = bd;
        express_done(sig);
    }
}

// One RRT* iteration's three dependent queries in ONE launch (corridor_finder.cpp:385-410 genNewNode, :428-437
// findNearstVertex, :464 the treeRewire neighbourhood): a 256-thread block per sample
//   A. nearest tree node of the fp32-narrowed sample (kd_nearestf semantics, lowest index on ties) over the node set,
//   B. steer: centre = nearest + (sample - nearest) * (r_nearest / dist) when the sample lies outside the node's sphere,
//      then the sphere inflation of that centre against the obstacle cloud (radiusSearch :113-133, early-out included),
//   C. the nodes within 2 * float(radius) of the fp32-narrowed centre (candidates for kd_nearest_rangef).
// Sequentially these are three launches + three host round trips (~45-70 us per sample); fused they are one (~20 us),
// for one sample or for a speculative batch.  node_aux[4*i..] = {x, y, z, radius} of node i as the planner holds them
// (fp64 coordinates, float radius widened), in host-mapped memory like the node coordinates themselves.
struct ExpandOut { double cx, cy, cz, radius; uint32_t near_idx, count; };

__global__ __launch_bounds__(256) void rrt_expand_kernel(const float *__restrict__ nx, const float *__restrict__ ny,
                                                         const float *__restrict__ nz, uint32_t n_nodes,
                                                         const double *__restrict__ node_aux, const double *__restrict__ samples,
                                                         GridDesc G0, const float4 *__restrict__ pts0, const uint32_t *__restrict__ cs0,
                                                         CoarseLevels C, int obstacles_empty, InflateParams P, double stop_d2,
                                                         uint32_t *__restrict__ ids, uint32_t cap_per_query, ExpandOut *__restrict__ out,
                                                         ExpressSignal sig)
{
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_ids[256];                // the neighbourhood is gathered here and handed to the host by ONE wave (below)
    const uint32_t slot = blockIdx.x;
    const double sx = samples[3 * slot], sy = samples[3 * slot + 1], sz = samples[3 * slot + 2];
    if (threadIdx.x == 0) s_n = 0;
    // ---- A: nearest node -------------------------------------------------------------------------------------------
    double bd = __builtin_huge_val();
    uint32_t near = kNoIndex;
    {
        const double qx = (double)(float)sx, qy = (double)(float)sy, qz = (double)(float)sz;
        for (uint32_t i = threadIdx.x; i < n_nodes; i += 256) {
            const double d2 = dist2((double)nx[i], (double)ny[i], (double)nz[i], qx, qy, qz);
            if (d2 < bd) { bd = d2; near = i; }
        }
        block_argmin256(bd, near, s_d, s_i);
    }
    // ---- B: steer + inflate (every thread computes the same centre) ------------------------------------------------
    double cx = sx, cy = sy, cz = sz;
    if (near != kNoIndex) {
        const double ax = node_aux[4 * near], ay = node_aux[4 * near + 1], az = node_aux[4 * near + 2], ar = node_aux[4 * near + 3];
        const double dx = ax - sx, dy = ay - sy, dz = az - sz;
        const double dis = sqrt(dx * dx + dy * dy + dz * dz);                    // getDis(nearest->coord, pt_sample)
        if (dis > ar) {                                                          // :392-400
            const double steer_dis = ar / dis;
            cx = ax + (sx - ax) * steer_dis;
            cy = ay + (sy - ay) * steer_dis;
            cz = az + (sz - az) * steer_dis;
        }
    }
    double radius;
    {
        const double dx = cx - P.sx, dy = cy - P.sy, dz = cz - P.sz;
        if (obstacles_empty || sqrt(dx * dx + dy * dy + dz * dz) > P.sample_range + P.max_radius) {
            radius = P.max_radius - P.search_margin;                             // :115-116
        } else {
            double od;
            uint32_t oi;
            block_nn_search(G0, pts0, cs0, C, cx, cy, cz, stop_d2, s_d, s_i, od, oi);
            const double rr = sqrt(od) - P.search_margin;
            radius = rr < P.max_radius ? rr : P.max_radius;
        }
    }
    // ---- C: neighbourhood candidates ---------------------------------------------------------------------------------
    {
        const float rf = fmaxf((float)radius, 0.0f) * 2.0f;                      // range = radius * 2 on the float member (:462)
        const double r = (double)rf, r2 = r * r;
        const double qx = (double)(float)cx, qy = (double)(float)cy, qz = (double)(float)cz;
        __syncthreads();                                                         // s_n = 0 is visible
        for (uint32_t i = threadIdx.x; i < n_nodes; i += 256)
            if (dist2((double)nx[i], (double)ny[i], (double)nz[i], qx, qy, qz) <= r2) {
                const uint32_t pos = atomicAdd(&s_n, 1u);
                if (pos < 256u) s_ids[pos] = i;
                else if (pos < cap_per_query) ids[(size_t)slot * cap_per_query + pos] = i;     // a neighbourhood beyond 256 nodes: straight to the host
            }
        __syncthreads();
    }
    const uint32_t n_hits = s_n;
    if (n_hits > 256u) {                           // rare: many threads stored host-visible data, all of them fence
        for (uint32_t k = threadIdx.x; k < min(256u, cap_per_query); k += 256) ids[(size_t)slot * cap_per_query + k] = s_ids[k];
        if (threadIdx.x == 0) {
            out[slot].cx = cx; out[slot].cy = cy; out[slot].cz = cz; out[slot].radius = radius;
            out[slot].near_idx = near;
            out[slot].count = n_hits;
        }
        express_done_block(sig);
        return;
    }
    // the usual case: wave 0 alone writes everything the host reads, so one wave fences instead of four
    if (threadIdx.x < 64) {
        for (uint32_t k = threadIdx.x; k < min(n_hits, cap_per_query); k += 64) ids[(size_t)slot * cap_per_query + k] = s_ids[k];
        if (threadIdx.x == 0) {
            out[slot].cx = cx; out[slot].cy = cy; out[slot].cz = cz; out[slot].radius = radius;
            out[slot].near_idx = near;
            out[slot].count = n_hits;
        }
        if (sig.seq) __threadfence_system();         // the wave's stores (ids by up to 64 lanes) before lane 0 takes the ticket
        if (threadIdx.x == 0) express_done(sig);
    }
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
        for (int j = 0; j <= order; j++) binom[j] = bernstein_binom(order, j);
        for (int d = 0; d < 3; d++) {
            double acc = 0.0;
            for (int j = 0; j < m; j++) acc += binom[j] * c[d * m + j] * pow_uint_cr(u, j) * pow_uint_cr(1.0 - u, order - j);
            pos[3 * s + d] = acc * T;
        }
    }
}

// Low-latency form of the whole check on an indexed cloud: ONE launch, a 256-thread block per sample.  The host enumerates
// the sample times (the same sequential fp64 additions as above: IEEE adds are the same on both sides) and hands (segment, t)
// per sample; the block evaluates getPosFromBezier with one thread per Bernstein term (two pow calls each, the terms then
// summed by one thread in the reference's j order), applies the inflation early-out and runs the block-wide cell search.
// Arguments and results live in host-mapped memory; the host picks the first sample with a negative radius.
__global__ __launch_bounds__(256) void bezier_block_kernel(GridDesc G0, const float4 *__restrict__ pts0, const uint32_t *__restrict__ cs0,
                                                           CoarseLevels C, InflateParams P, const double *__restrict__ coef, int row_stride,
                                                           const double *__restrict__ seg_time, const uint32_t *__restrict__ orders,
                                                           const uint32_t *__restrict__ sample_seg, const double *__restrict__ sample_t,
                                                           double stop_d2, uint32_t index_base, ExpressOut *__restrict__ out,
                                                           double *__restrict__ pos_out, ExpressSignal sig)
{
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    __shared__ double s_term[3 * (kMaxBezierOrder + 1)];
    __shared__ double s_pos[3];
    const uint32_t slot = blockIdx.x;
    const int seg = (int)sample_seg[slot];
    const int order = (int)orders[seg], m = order + 1;
    const double T = seg_time[seg];
    const double u = sample_t[slot] / T;
    if ((int)threadIdx.x < 3 * m) {
        const int d = (int)threadIdx.x / m, j = (int)threadIdx.x % m;
        const double b = bernstein_binom(order, j);
        s_term[d * m + j] = b * coef[(size_t)seg * row_stride + d * m + j] * pow_uint_cr(u, j) * pow_uint_cr(1.0 - u, order - j);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double acc = 0.0;
        for (int j = 0; j < m; j++) acc += s_term[threadIdx.x * m + j];
        s_pos[threadIdx.x] = acc * T;
    }
    __syncthreads();
    const double px = s_pos[0], py = s_pos[1], pz = s_pos[2];
    // every host-visible store of the block comes from thread 0, so one thread fences and signals
    if (threadIdx.x == 0) { pos_out[3 * slot] = px; pos_out[3 * slot + 1] = py; pos_out[3 * slot + 2] = pz; }
    const double dx = px - P.sx, dy = py - P.sy, dz = pz - P.sz;
    if (sqrt(dx * dx + dy * dy + dz * dz) > P.sample_range + P.max_radius) {          // corridor_finder.cpp:115-116
        if (threadIdx.x == 0) { out[slot].radius = P.max_radius - P.search_margin; out[slot].idx = kNoIndex; out[slot].d2 = __builtin_huge_val(); express_done(sig); }
        return;
    }
    double bd;
    uint32_t bi;
    block_nn_search(G0, pts0, cs0, C, px, py, pz, stop_d2, s_d, s_i, bd, bi);
    if (threadIdx.x == 0) {
        const double rr = sqrt(bd) - P.search_margin;
        out[slot].radius = rr < P.max_radius ? rr : P.max_radius;
        out[slot].idx = (bi == kNoIndex) ? kNoIndex : bi + index_base;
        out[slot].d2 = bd;
        express_done(sig);
    }
}

// getPosFromBezier for host-enumerated samples (segment, t), one thread per sample, the arithmetic of bezier_samples_kernel;
// positions go to device memory for the inflation kernels and to host-mapped memory for the caller
__global__ __launch_bounds__(128) void bezier_eval_kernel(const double *__restrict__ coef, int row_stride, const double *__restrict__ seg_time,
                                                          const uint32_t *__restrict__ orders, const uint32_t *__restrict__ sample_seg,
                                                          const double *__restrict__ sample_t, int n, double *__restrict__ pos_dev,
                                                          double *__restrict__ pos_mapped)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int seg = (int)sample_seg[s];
    const int order = (int)orders[seg], m = order + 1;
    const double T = seg_time[seg];
    const double u = sample_t[s] / T;
    const double *c = coef + (size_t)seg * row_stride;
    double binom[kMaxBezierOrder + 1];
    for (int j = 0; j <= order; j++) binom[j] = bernstein_binom(order, j);
    for (int d = 0; d < 3; d++) {
        double acc = 0.0;
        for (int j = 0; j < m; j++) acc += binom[j] * c[d * m + j] * pow_uint_cr(u, j) * pow_uint_cr(1.0 - u, order - j);
        pos_dev[3 * s + d] = acc * T;
        pos_mapped[3 * s + d] = acc * T;
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
