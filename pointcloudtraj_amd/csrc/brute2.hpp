// brute2.hpp -- the brute-force filter in EXPANDED form: 3 fused multiply-adds per pair instead of 6 operations.
//
// nn_tile_candidates_kernel (kernels.hpp) screens every (point, query) pair with d32 = (px-qx)^2 + (py-qy)^2 + (pz-qz)^2 in
// packed fp32: 3 subtractions, 1 multiplication, 2 FMAs per pair -- the arithmetic floor of that form, and the kernel runs at
// ~60 % of it.  The algebraically equal  |p|^2 - 2 p.q + |q|^2  needs, per pair, only the three FMAs of
// t = |p|^2 - 2 p.q  (|p|^2 once per point group and query tile, |q|^2 folded into the threshold), i.e. about 3.75 instead of
// 6.75 lane-operations per pair with the min / compare tail.  What it costs is cancellation: t carries an ABSOLUTE error
// proportional to |p|^2 + |q|^2, not a relative one.  The filter stays exact because
//   * it is only a filter: a pair that passes is re-evaluated in the reference's fp64 arithmetic on the ORIGINAL coordinates
//     and competes by (d2, index); nothing is decided on t;
//   * the threshold is widened by a proven bound of that error, so no pair with true d2 <= bound is ever rejected:
//       coordinates are centred on the cloud's bounding-box centre c first (pc = fl(p - c), qc = fl(q - c), each coordinate off by
//       at most u|.|, u = 2^-24): the centred distance D~ obeys sqrt(D~) <= sqrt(D) + eps, eps = u (R + |qc|), R = the box's
//       half diagonal;
//       pp = fl(|pc|^2) and the FMA chain t = fl(pc.x*(-2qc.x) + fl(pc.y*(-2qc.y) + fl(pc.z*(-2qc.z) + pp))) err by at most
//       3u (3 R^2 + |qc|^2) in absolute terms (every partial sum is bounded by |pc|^2 + 2|pc||qc| <= 2|pc|^2 + |qc|^2);
//     hence  D <= B  implies  t <= (sqrt(B) + eps)^2 - |qc|^2 + 3.5u (3 R^2 + |qc|^2) =: thr'(q), evaluated in fp64 by
//     brute2_prep_kernel and rounded UP to fp32.  B is the fp32 upper bound of the sampling pass, widened as before.
// The band costs extra candidates (more pairs pass than with the relative-error form), so the host uses this kernel only while
// the band is small against the cloud's point spacing (engine.hip use_expanded_filter) and the direct form otherwise.
// Results are bit-identical either way (tests/test_gpu_soak.py, test_filter_adversarial_near_ties run both).
#pragma once
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace pct {

struct CentreDesc { float cx, cy, cz; double R2; };       // bounding-box centre (fp32) and squared half diagonal

// per query: {-2 qc.x, -2 qc.y, -2 qc.z, thr'} for the NN filter
__global__ __launch_bounds__(256) void brute2_prep_kernel(CentreDesc C, const float *__restrict__ qf, const uint32_t *__restrict__ bound_bits,
                                                          uint32_t Q, float4 *__restrict__ qprep)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const float qx = qf[3 * i] - C.cx, qy = qf[3 * i + 1] - C.cy, qz = qf[3 * i + 2] - C.cz;     // qc = fl(q - c), the same rounding the kernel's operands have
    const double Qc = (double)qx * (double)qx + (double)qy * (double)qy + (double)qz * (double)qz;
    const double u = 0x1p-24;
    const float bound = __uint_as_float(bound_bits[i]);
    float thr = __builtin_huge_valf();
    if (bound < 3.0e38f) {
        const double B = (double)bound * (1.0 + 0x1p-19) + 0x1p-90;          // true d2 of the sampled point that set the bound is <= B
        const double eps = u * (sqrt(C.R2) + sqrt(Qc)) * 1.0001;
        const double sb = sqrt(B) + eps;
        const double t = (sb * sb - Qc + 3.5 * u * (3.0 * C.R2 + Qc)) ;
        const double tw = t + fabs(t) * 1e-6 + 1e-30;                        // slack for the fp64 evaluation itself, then round UP to fp32
        float tf = (float)tw;
        if ((double)tf < tw) tf = __uint_as_float(__float_as_uint(tf) + (tf >= 0.0f ? 1u : 0xFFFFFFFFu));    // next float towards +inf
        if (tf == 0.0f && tw > 0.0) tf = 1e-45f;
        thr = tf;
    }
    qprep[i] = make_float4(-2.0f * qx, -2.0f * qy, -2.0f * qz, thr);
}

// smallest t = |pc|^2 - 2 pc.qc over the 4 (centred) points of g: 6 packed FMAs + min3 + min
__device__ __forceinline__ float group_min_t(const PointGroup &g, v2f pp01, v2f pp23, float ax, float ay, float az)
{
    const v2f ax2 = v2f{ ax, ax }, ay2 = v2f{ ay, ay }, az2 = v2f{ az, az };
    v2f t01 = __builtin_elementwise_fma(g.z01, az2, pp01), t23 = __builtin_elementwise_fma(g.z23, az2, pp23);
    t01 = __builtin_elementwise_fma(g.y01, ay2, t01);
    t23 = __builtin_elementwise_fma(g.y23, ay2, t23);
    t01 = __builtin_elementwise_fma(g.x01, ax2, t01);
    t23 = __builtin_elementwise_fma(g.x23, ax2, t23);
    return fminf(fminf(t01.x, t01.y), fminf(t23.x, t23.y));
}

__device__ __forceinline__ void group_pp(const PointGroup &g, v2f &pp01, v2f &pp23)
{
    pp01 = g.x01 * g.x01; pp23 = g.x23 * g.x23;
    pp01 = __builtin_elementwise_fma(g.y01, g.y01, pp01);
    pp23 = __builtin_elementwise_fma(g.y23, g.y23, pp23);
    pp01 = __builtin_elementwise_fma(g.z01, g.z01, pp01);
    pp23 = __builtin_elementwise_fma(g.z23, g.z23, pp23);
}

// Same structure as nn_tile_candidates_kernel: a block stages its chunk of the cloud in LDS once (here: CENTRED), walks its slice
// of the batch in tiles of 8 wave-uniform queries; a lane whose group passes a query's threshold evaluates it exactly (fp64, the
// original coordinates re-read from global memory) and appends (d2, index) to that query's candidate list.
// kGroupsPerIter = point groups per lane and loop iteration: independent LDS reads and FMA chains in flight
template <int kGroupsPerIter>
__global__ __launch_bounds__(256) void nn_tile_candidates2_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                  const float *__restrict__ z, uint32_t n, uint32_t chunk_groups,
                                                                  CentreDesc C, const float4 *__restrict__ qprep, const double *__restrict__ q64,
                                                                  int Q, int qslice, uint32_t *__restrict__ cand_count,
                                                                  double *__restrict__ cand_d2, uint32_t *__restrict__ cand_idx)
{
    extern __shared__ float4 s_pts[];                 // [3][chunk_groups], centred
    const uint32_t ngroups = n >> 2;
    const uint32_t g0 = blockIdx.x * chunk_groups;
    const uint32_t ng = min(chunk_groups, ngroups > g0 ? ngroups - g0 : 0u);
    float4 *sx = s_pts, *sy = s_pts + chunk_groups, *sz = s_pts + 2 * chunk_groups;
    for (uint32_t i = threadIdx.x; i < ng; i += 256) {
        float4 X = reinterpret_cast<const float4 *>(x)[g0 + i], Y = reinterpret_cast<const float4 *>(y)[g0 + i], Z = reinterpret_cast<const float4 *>(z)[g0 + i];
        X.x -= C.cx; X.y -= C.cx; X.z -= C.cx; X.w -= C.cx;
        Y.x -= C.cy; Y.y -= C.cy; Y.z -= C.cy; Y.w -= C.cy;
        Z.x -= C.cz; Z.y -= C.cz; Z.z -= C.cz; Z.w -= C.cz;
        sx[i] = X; sy[i] = Y; sz[i] = Z;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool tail_owner = (blockIdx.x == gridDim.x - 1) && threadIdx.x < (n & 3u);   // n % 4 leftover points

    const int q_end = min(Q, ((int)blockIdx.y + 1) * qslice);
    for (int q0 = (int)blockIdx.y * qslice; q0 < q_end; q0 += kTileQ) {
        const int qcount = min(kTileQ, q_end - q0);
        float ax[kTileQ], ay[kTileQ], az[kTileQ], thr[kTileQ];
#pragma unroll
        for (int j = 0; j < kTileQ; j++) {
            const float4 P = qprep[q0 + (j < qcount ? j : qcount - 1)];
            ax[j] = P.x; ay[j] = P.y; az[j] = P.z; thr[j] = P.w;
        }
        for (uint32_t i0 = threadIdx.x; i0 < ng; i0 += 256 * kGroupsPerIter) {
            PointGroup pg[kGroupsPerIter];
            v2f pp01[kGroupsPerIter], pp23[kGroupsPerIter];
#pragma unroll
            for (int u = 0; u < kGroupsPerIter; u++) {
                const uint32_t i = min(i0 + 256u * (uint32_t)u, ng - 1);           // past the end: repeat the last group (its hits are masked below)
                pg[u] = make_group(sx[i], sy[i], sz[i]);
                group_pp(pg[u], pp01[u], pp23[u]);
            }
            unsigned long long hit[kGroupsPerIter];
#pragma unroll
            for (int u = 0; u < kGroupsPerIter; u++) hit[u] = 0ull;
#pragma unroll
            for (int j = 0; j < kTileQ; j++) {
#pragma unroll
                for (int u = 0; u < kGroupsPerIter; u++)
                    hit[u] |= __builtin_amdgcn_ballot_w64(group_min_t(pg[u], pp01[u], pp23[u], ax[j], ay[j], az[j]) <= thr[j]);
            }
#pragma unroll
            for (int u = 0; u < kGroupsPerIter; u++) {
                const uint32_t i = i0 + 256u * (uint32_t)u;
                if (hit[u] != 0ull && i < ng && ((hit[u] >> lane) & 1ull)) {       // rare: this lane's group may hold a winner for some query of the tile
                    const uint32_t g = g0 + i;
                    const float4 X = reinterpret_cast<const float4 *>(x)[g], Y = reinterpret_cast<const float4 *>(y)[g], Z = reinterpret_cast<const float4 *>(z)[g];
                    const float xs[4] = { X.x, X.y, X.z, X.w }, ys[4] = { Y.x, Y.y, Y.z, Y.w }, zs[4] = { Z.x, Z.y, Z.z, Z.w };
#pragma unroll 1
                    for (int j = 0; j < qcount; j++) {
                        if (!(group_min_t(pg[u], pp01[u], pp23[u], ax[j], ay[j], az[j]) <= thr[j])) continue;
                        const int qi = q0 + j;
                        const double Qx = q64[3 * qi], Qy = q64[3 * qi + 1], Qz = q64[3 * qi + 2];
                        double bd = __builtin_huge_val();
                        uint32_t bi = kNoIndex;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const double d2 = dist2((double)xs[k], (double)ys[k], (double)zs[k], Qx, Qy, Qz);
                            if (d2 < bd) { bd = d2; bi = 4u * g + (uint32_t)k; }       // ids grow with k: lowest index on ties
                        }
                        const uint32_t slot = atomicAdd(&cand_count[qi], 1u);
                        if (slot < kCandCap) { cand_d2[(size_t)qi * kCandCap + slot] = bd; cand_idx[(size_t)qi * kCandCap + slot] = bi; }
                    }
                }
            }
        }
        if (tail_owner) {       // the n % 4 leftover points: always candidates (at most 3 per query)
            const uint32_t id = 4u * ngroups + threadIdx.x;
            const double px = (double)x[id], py = (double)y[id], pz = (double)z[id];
            for (int j = 0; j < qcount; j++) {
                const int qi = q0 + j;
                const double d2 = dist2(px, py, pz, q64[3 * qi], q64[3 * qi + 1], q64[3 * qi + 2]);
                const uint32_t slot = atomicAdd(&cand_count[qi], 1u);
                if (slot < kCandCap) { cand_d2[(size_t)qi * kCandCap + slot] = d2; cand_idx[(size_t)qi * kCandCap + slot] = id; }
            }
        }
    }
}

// The same filter with the block's points held in REGISTERS instead of LDS: every thread keeps kRegGroups groups (4 points each,
// centred, with their |pc|^2) for the whole launch and the query tiles stream past as scalars.  No LDS reads in the loop, |pc|^2
// computed once per launch instead of once per tile, and occupancy set by registers alone (the LDS form holds 3 blocks per CU).
// kRegGroups = groups per thread: a block covers 256 * kRegGroups groups (4 -> 4096 points)
template <bool COUNT, int kRegGroups>
__global__ __launch_bounds__(256) void tile_reg_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, uint32_t n,
                                                       CentreDesc C, const float4 *__restrict__ qprep, const double *__restrict__ q64,
                                                       const double *__restrict__ r2, int Q, int qslice, uint32_t *__restrict__ cand_count,
                                                       double *__restrict__ cand_d2, uint32_t *__restrict__ cand_idx, uint32_t *__restrict__ count)
{
    __shared__ uint32_t s_cnt[kTileQ];
    const uint32_t ngroups = n >> 2;
    const uint32_t g0 = blockIdx.x * (256u * kRegGroups);
    PointGroup pg[kRegGroups];
    v2f pp01[kRegGroups], pp23[kRegGroups];
    bool live[kRegGroups];
#pragma unroll
    for (int u = 0; u < kRegGroups; u++) {
        const uint32_t g = g0 + 256u * (uint32_t)u + threadIdx.x;
        live[u] = g < ngroups;
        const uint32_t gl = live[u] ? g : (ngroups ? ngroups - 1 : 0u);
        float4 X = ngroups ? reinterpret_cast<const float4 *>(x)[gl] : make_float4(0, 0, 0, 0), Y = ngroups ? reinterpret_cast<const float4 *>(y)[gl] : X,
               Z = ngroups ? reinterpret_cast<const float4 *>(z)[gl] : X;
        X.x -= C.cx; X.y -= C.cx; X.z -= C.cx; X.w -= C.cx;
        Y.x -= C.cy; Y.y -= C.cy; Y.z -= C.cy; Y.w -= C.cy;
        Z.x -= C.cz; Z.y -= C.cz; Z.z -= C.cz; Z.w -= C.cz;
        pg[u] = make_group(X, Y, Z);
        group_pp(pg[u], pp01[u], pp23[u]);
    }
    const int lane = threadIdx.x & 63;
    const bool tail_owner = (blockIdx.x == gridDim.x - 1) && threadIdx.x < (n & 3u);   // n % 4 leftover points

    const int q_end = min(Q, ((int)blockIdx.y + 1) * qslice);
    for (int q0 = (int)blockIdx.y * qslice; q0 < q_end; q0 += kTileQ) {
        const int qcount = min(kTileQ, q_end - q0);
        if (COUNT) {
            if (threadIdx.x < kTileQ) s_cnt[threadIdx.x] = 0;
            __syncthreads();
        }
        float ax[kTileQ], ay[kTileQ], az[kTileQ], thr[kTileQ];
#pragma unroll
        for (int j = 0; j < kTileQ; j++) {
            const float4 P = qprep[q0 + (j < qcount ? j : qcount - 1)];
            ax[j] = P.x; ay[j] = P.y; az[j] = P.z; thr[j] = P.w;
        }
        unsigned long long hit[kRegGroups];
#pragma unroll
        for (int u = 0; u < kRegGroups; u++) hit[u] = 0ull;
#pragma unroll
        for (int j = 0; j < kTileQ; j++) {
#pragma unroll
            for (int u = 0; u < kRegGroups; u++)
                hit[u] |= __builtin_amdgcn_ballot_w64(group_min_t(pg[u], pp01[u], pp23[u], ax[j], ay[j], az[j]) <= thr[j]);
        }
#pragma unroll
        for (int u = 0; u < kRegGroups; u++) {
            if (hit[u] != 0ull && live[u] && ((hit[u] >> lane) & 1ull)) {          // rare: exact evaluation of this lane's group
                const uint32_t g = g0 + 256u * (uint32_t)u + threadIdx.x;
                const float4 X = reinterpret_cast<const float4 *>(x)[g], Y = reinterpret_cast<const float4 *>(y)[g], Z = reinterpret_cast<const float4 *>(z)[g];
                const float xs[4] = { X.x, X.y, X.z, X.w }, ys[4] = { Y.x, Y.y, Y.z, Y.w }, zs[4] = { Z.x, Z.y, Z.z, Z.w };
#pragma unroll 1
                for (int j = 0; j < qcount; j++) {
                    if (!(group_min_t(pg[u], pp01[u], pp23[u], ax[j], ay[j], az[j]) <= thr[j])) continue;
                    const int qi = q0 + j;
                    const double Qx = q64[3 * qi], Qy = q64[3 * qi + 1], Qz = q64[3 * qi + 2];
                    if (COUNT) {
                        const double rr = r2[qi];
                        uint32_t cl = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) cl += dist2((double)xs[k], (double)ys[k], (double)zs[k], Qx, Qy, Qz) <= rr ? 1u : 0u;
                        if (cl) atomicAdd(&s_cnt[j], cl);
                    } else {
                        double bd = __builtin_huge_val();
                        uint32_t bi = kNoIndex;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const double d2 = dist2((double)xs[k], (double)ys[k], (double)zs[k], Qx, Qy, Qz);
                            if (d2 < bd) { bd = d2; bi = 4u * g + (uint32_t)k; }   // ids grow with k: lowest index on ties
                        }
                        const uint32_t slot = atomicAdd(&cand_count[qi], 1u);
                        if (slot < kCandCap) { cand_d2[(size_t)qi * kCandCap + slot] = bd; cand_idx[(size_t)qi * kCandCap + slot] = bi; }
                    }
                }
            }
        }
        if (tail_owner) {       // the n % 4 leftover points, exact
            const uint32_t id = 4u * ngroups + threadIdx.x;
            const double px = (double)x[id], py = (double)y[id], pz = (double)z[id];
            for (int j = 0; j < qcount; j++) {
                const int qi = q0 + j;
                const double d2 = dist2(px, py, pz, q64[3 * qi], q64[3 * qi + 1], q64[3 * qi + 2]);
                if (COUNT) {
                    if (d2 <= r2[qi]) atomicAdd(&s_cnt[j], 1u);
                } else {
                    const uint32_t slot = atomicAdd(&cand_count[qi], 1u);
                    if (slot < kCandCap) { cand_d2[(size_t)qi * kCandCap + slot] = d2; cand_idx[(size_t)qi * kCandCap + slot] = id; }
                }
            }
        }
        if (COUNT) {
            __syncthreads();
            if ((int)threadIdx.x < qcount && s_cnt[threadIdx.x]) atomicAdd(&count[q0 + threadIdx.x], s_cnt[threadIdx.x]);
            __syncthreads();
        }
    }
}

// per query of a radius count: {-2 qc.x, -2 qc.y, -2 qc.z, hi'} with hi' = the largest value t can take for a point inside the ball
// (d2 <= r2), by the same error bound as the NN threshold; r2 (fp64) alongside for the exact test
__global__ __launch_bounds__(256) void brute2_prep_count_kernel(CentreDesc C, const float *__restrict__ qf, const float *__restrict__ rad, uint32_t Q,
                                                                float4 *__restrict__ qprep, double *__restrict__ r2out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const float qx = qf[3 * i] - C.cx, qy = qf[3 * i + 1] - C.cy, qz = qf[3 * i + 2] - C.cz;
    const double Qc = (double)qx * (double)qx + (double)qy * (double)qy + (double)qz * (double)qz;
    const double u = 0x1p-24;
    const double r = (double)rad[i], B = r * r;                        // kdtree.c:273: d2 <= range * range
    r2out[i] = B;
    const double eps = u * (sqrt(C.R2) + sqrt(Qc)) * 1.0001;
    const double sb = fabs(r) + eps;
    const double t = sb * sb - Qc + 3.5 * u * (3.0 * C.R2 + Qc);
    const double tw = t + fabs(t) * 1e-6 + 1e-30;
    float tf = (float)tw;
    if ((double)tf < tw) tf = __uint_as_float(__float_as_uint(tf) + (tf >= 0.0f ? 1u : 0xFFFFFFFFu));
    if (tf == 0.0f && tw > 0.0) tf = 1e-45f;
    if (!(B >= 0.0)) tf = -__builtin_huge_valf();                      // NaN radius: nothing is inside
    qprep[i] = make_float4(-2.0f * qx, -2.0f * qy, -2.0f * qz, tf);
}

}  // namespace pct
