// pyramid.hpp -- bounding-box pyramid over the cell index: exact nearest neighbour on clouds with SPARSE OCCUPANCY.
//
// Why: the reference's real input is pillar SURFACES on a 0.1 m lattice (Planner/src/map_generator.cpp:16-125), not a volume
// filled with points.  On such a cloud the shell walk of the cell-pruned search (kernels.hpp coop_nn_search) visits every cell of
// every shell between a free-space query and the nearest surface -- hundreds of empty x-runs -- and then scans every point of every
// occupied cell the final shells touch, although a wall's points in a cell occupy a thin slab of it: 859-2154 points and 39-310
// runs per query on the seed-6 map against 49 / 4 on the uniform benchmark cloud (profiles/r02 notes, gpurun_out/cl.log).
//
// What: level 0 holds, per cell of the index, the exact bounding box of the cell's points (their own fp32 coordinates: min / max,
// no rounding) plus the cell's run in the cell-sorted array; level l + 1 merges 2 x 2 x 2 nodes of level l.  A query that the
// 2x2x2-block stage of the batch kernel leaves undecided walks the pyramid from the top, its 8 lanes owning the 8 children of the
// current node: each lane computes the squared distance from the query to its child's box (the lower bound LB of every point
// below it), the group descends into the nearest child with LB <= best, scans leaf cells exactly, and backs up when no child
// qualifies.  Empty space costs nothing (an empty node has LB = +inf), and a wall at distance d contributes only the few cells
// whose BOX -- not whose cell -- reaches into the ball: the candidates shrink from "every occupied cell the shells cover" to the
// cells around the foot point.
//
// Exactness (index-exact parity with kdtree.c:345-402's arithmetic, lowest index on ties): LB is evaluated in the SAME arithmetic
// as dist2 -- fp64 on float-widened operands, ((dx^2 + dy^2) + dz^2), no FMA -- from per-axis gaps max(lo - q, q - hi, 0).  For a
// point p inside the box every |p_k - q_k| as computed is >= the gap as computed (fl(a - q) is monotone in a), and fp64 products
// and sums of non-negative terms are monotone under round-to-nearest, so LB <= dist2(p, q) holds for the COMPUTED values, not just
// the real ones: a subtree is skipped only when LB > best, hence no point with d2 <= best -- neither a better one nor an equal one
// with a lower index -- is ever behind a skipped node.  No slack constants, no dependence on the fp32 cell assignment.
#pragma once

namespace pct {

constexpr int kPyrMaxLevels = 12;      // 1024 cells per axis -> 11 levels (1024, 512, ..., 2, 1)

// 32 bytes = two 16-byte loads.  Empty node: lo = +inf, hi = -inf, count = 0.
struct PyrNode {
    float lox, loy, loz, hix;
    float hiy, hiz;
    uint32_t start;        // level 0: first record of the cell in the cell-sorted array
    uint32_t count;        // points below the node (saturating)
};

// Layout: the 8 children of a node are CONTIGUOUS (256 bytes = two cache lines; a node visit reads exactly them, one child per
// lane).  Level l is stored as blocks of 8 indexed by the PARENT's linear index in the level-(l+1) grid, x fastest:
//   node (x, y, z) of level l  ->  off[l] + 8 * lin_{l+1}(x >> 1, y >> 1, z >> 1) + (x & 1 | (y & 1) << 1 | (z & 1) << 2)
// where the grid of level nlev (the virtual root) is 1 x 1 x 1.  Slots beyond a level's real extent hold empty nodes.
struct PyrDesc {
    int nlev;                          // levels 0 .. nlev-1; the top level has at most 2 nodes per axis
    uint32_t off[kPyrMaxLevels];       // first slot of level l in the node array
};

__device__ __host__ __forceinline__ int pyr_dim(int g, int l) { return (g + (1 << l) - 1) >> l; }

__device__ __forceinline__ size_t pyr_slot(const GridDesc &G, const PyrDesc &PD, int l, int x, int y, int z)
{
    const int pgx = pyr_dim(G.gx, l + 1), pgy = pyr_dim(G.gy, l + 1);
    return (size_t)PD.off[l] + 8u * (((size_t)(z >> 1) * pgy + (y >> 1)) * pgx + (x >> 1)) + (size_t)((x & 1) | ((y & 1) << 1) | ((z & 1) << 2));
}

// level 0: 8 lanes per slot fold the cell's points; grid = 8 * (nodes of level 1) slots
__global__ __launch_bounds__(256) void pyr_leaf_kernel(GridDesc G, PyrDesc PD, const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                                       PyrNode *__restrict__ nodes, uint32_t nslots)
{
    const uint32_t slot = blockIdx.x * 32u + (threadIdx.x >> 3), sub = threadIdx.x & 7u;
    const int pgx = pyr_dim(G.gx, 1), pgy = pyr_dim(G.gy, 1);
    const uint32_t parent = slot >> 3, ch = slot & 7u;
    const int x = 2 * (int)(parent % (uint32_t)pgx) + (int)(ch & 1u), y = 2 * (int)((parent / (uint32_t)pgx) % (uint32_t)pgy) + (int)((ch >> 1) & 1u),
              z = 2 * (int)(parent / ((uint32_t)pgx * (uint32_t)pgy)) + (int)(ch >> 2);
    const bool real = slot < nslots && x < G.gx && y < G.gy && z < G.gz;
    float lo[3] = { __builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf() };
    float hi[3] = { -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf() };
    uint32_t s = 0, e = 0;
    if (real) { const uint32_t cell = cell_lin(G, x, y, z); s = cell_start[cell]; e = cell_start[cell + 1]; }
    for (uint32_t p = s + sub; p < e; p += 8u) {
        const float4 P = pts[p];
        lo[0] = fminf(lo[0], P.x); hi[0] = fmaxf(hi[0], P.x);
        lo[1] = fminf(lo[1], P.y); hi[1] = fmaxf(hi[1], P.y);
        lo[2] = fminf(lo[2], P.z); hi[2] = fmaxf(hi[2], P.z);
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, kWave));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, kWave));
        }
    if (slot < nslots && sub == 0) {
        float4 *out = reinterpret_cast<float4 *>(nodes + PD.off[0] + slot);
        out[0] = make_float4(lo[0], lo[1], lo[2], hi[0]);
        out[1] = make_float4(hi[1], hi[2], __uint_as_float(s), __uint_as_float(e - s));
    }
}

// level l (>= 1): one thread per slot merges the 8 children (one contiguous block of level l - 1)
__global__ __launch_bounds__(256) void pyr_up_kernel(GridDesc G, PyrDesc PD, int l, PyrNode *__restrict__ nodes, uint32_t nslots)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    const int pgx = pyr_dim(G.gx, l + 1), pgy = pyr_dim(G.gy, l + 1);
    const int gx = pyr_dim(G.gx, l), gy = pyr_dim(G.gy, l), gz = pyr_dim(G.gz, l);
    const uint32_t parent = slot >> 3, ch = slot & 7u;
    const int x = 2 * (int)(parent % (uint32_t)pgx) + (int)(ch & 1u), y = 2 * (int)((parent / (uint32_t)pgx) % (uint32_t)pgy) + (int)((ch >> 1) & 1u),
              z = 2 * (int)(parent / ((uint32_t)pgx * (uint32_t)pgy)) + (int)(ch >> 2);
    float lo[3] = { __builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf() };
    float hi[3] = { -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf() };
    unsigned long long cnt = 0;
    if (x < gx && y < gy && z < gz) {
        const float4 *blk = reinterpret_cast<const float4 *>(nodes + PD.off[l - 1] + 8u * (((size_t)z * gy + y) * gx + x));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 a = blk[2 * k], b = blk[2 * k + 1];
            lo[0] = fminf(lo[0], a.x); lo[1] = fminf(lo[1], a.y); lo[2] = fminf(lo[2], a.z);
            hi[0] = fmaxf(hi[0], a.w); hi[1] = fmaxf(hi[1], b.x); hi[2] = fmaxf(hi[2], b.y);
            cnt += __float_as_uint(b.w);
        }
    }
    float4 *out = reinterpret_cast<float4 *>(nodes + PD.off[l] + slot);
    out[0] = make_float4(lo[0], lo[1], lo[2], hi[0]);
    out[1] = make_float4(hi[1], hi[2], __uint_as_float(0u), __uint_as_float(cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt));
}

// start-level hint per level-0 cell: the lowest level L in [1, nlev] whose node above the cell holds a point (L = nlev: the virtual
// root).  A query starts its walk there instead of at the root: the levels above are only visited on the way UP, and only while
// the region test below cannot close the search.
__global__ __launch_bounds__(256) void pyr_hint_kernel(GridDesc G, PyrDesc PD, const PyrNode *__restrict__ nodes, unsigned char *__restrict__ hint)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= G.ncells) return;
    const int x = (int)(c % (uint32_t)G.gx), y = (int)((c / (uint32_t)G.gx) % (uint32_t)G.gy), z = (int)(c / ((uint32_t)G.gx * (uint32_t)G.gy));
    int L = 1;
    for (; L < PD.nlev; L++)
        if (nodes[pyr_slot(G, PD, L, x >> L, y >> L, z >> L)].count) break;
    hint[c] = (unsigned char)L;
}

// ---- cross-lane helpers for a group of 8 lanes (DPP: kernels.hpp dpp_u32 and the kDpp* patterns) ----
// minimum over the 8 lanes of a group, in all 8 lanes
__device__ __forceinline__ uint32_t group8_min_u32(uint32_t v)
{
    v = min(v, dpp_u32<kDppXor1>(v));
    v = min(v, dpp_u32<kDppXor2>(v));
    v = min(v, dpp_u32<kDppHalfMirror>(v));
    return v;
}
__device__ __forceinline__ float group8_min_f32(float v)
{
    v = fminf(v, __uint_as_float(dpp_u32<kDppXor1>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_u32<kDppXor2>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_u32<kDppHalfMirror>(__float_as_uint(v))));
    return v;
}
// value of lane `src` (0..7, the same in all 8 lanes) of the group, in all 8 lanes
__device__ __forceinline__ uint32_t dpp_bcast8(uint32_t v, uint32_t src)
{
    return (uint32_t)__shfl((int)v, (int)src, 8);
}
template <int CTRL>
__device__ __forceinline__ void group8_better_step(double &d, uint32_t &i)
{
    const uint64_t b = (uint64_t)__double_as_longlong(d);
    const uint32_t lo = dpp_u32<CTRL>((uint32_t)b), hi = dpp_u32<CTRL>((uint32_t)(b >> 32)), oi = dpp_u32<CTRL>(i);
    const double od = __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
    if (better(od, oi, d, i)) { d = od; i = oi; }
}
// winner by (d2, index) over the 8 lanes of a group, in all 8 lanes
__device__ __forceinline__ void group8_argmin(double &d, uint32_t &i)
{
    group8_better_step<kDppXor1>(d, i);
    group8_better_step<kDppXor2>(d, i);
    group8_better_step<kDppHalfMirror>(d, i);
}

// fp32 upper bound of the current best squared distance, widened so that "lb32 > it" proves "every point behind lb32 is farther":
// fp32 evaluation of a box gap or a point distance errs by < 2^-21 relative (a subtraction, three squares, two sums)
__device__ __forceinline__ float pyr_best32(double bd) { return __double2float_ru(bd) * (1.0f + 0x1p-20f); }

// squared distance from q to the box, fp32: a lower bound of the exact value up to the 2^-21 relative error pyr_best32 covers
__device__ __forceinline__ float pyr_box_lb32(float lox, float loy, float loz, float hix, float hiy, float hiz, float qx, float qy, float qz)
{
    const float gx = fmaxf(fmaxf(lox - qx, qx - hix), 0.0f);
    const float gy = fmaxf(fmaxf(loy - qy, qy - hiy), 0.0f);
    const float gz = fmaxf(fmaxf(loz - qz, qz - hiz), 0.0f);
    return (gx * gx + gy * gy) + gz * gz;
}

// One level-0 cell's run [s, s + n), n > 0, against the query, by the 8 lanes of a group.  fp32 screening of every point (the
// first 16 requested before any is used), then the exact fp64 distance only for the points inside the fp32 error band of the
// run's minimum that can still reach the current best: the exact winner by (d2, index) is among them (same band as
// coop_screen_rows, kernels.hpp).  Updates (bd, bi, bd32) in all 8 lanes.
__device__ __forceinline__ void pyr_scan_leaf(const float4 *__restrict__ pts, uint32_t s, uint32_t n, uint32_t sub, float qxf, float qyf, float qzf,
                                              double qx, double qy, double qz, double &bd, uint32_t &bi, float &bd32)
{
    const uint32_t e = s + n, last = e - 1;
    const uint32_t p0 = s + sub, p1 = p0 + 8u;
    const float4 A = pts[min(p0, last)], B = pts[min(p1, last)];
    const float inf = __builtin_huge_valf();
    float dA, dB;
    {
        const float dx = A.x - qxf, dy = A.y - qyf, dz = A.z - qzf;
        dA = p0 < e ? __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) : inf;
    }
    {
        const float dx = B.x - qxf, dy = B.y - qyf, dz = B.z - qzf;
        dB = p1 < e ? __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) : inf;
    }
    float m = fminf(dA, dB);
    for (uint32_t p = p1 + 8u; p < e; p += 8u) {                      // cells of more than 16 points
        const float4 P = pts[p];
        const float dx = P.x - qxf, dy = P.y - qyf, dz = P.z - qzf;
        m = fminf(m, __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
    }
    m = group8_min_f32(m);
    const float T = fminf(m * (1.0f + 0x1p-19f) + 0x1p-90f, bd32);
    bool improved = false;
    if (dA <= T) {
        const double d2 = dist2((double)A.x, (double)A.y, (double)A.z, qx, qy, qz);
        const uint32_t id = __float_as_uint(A.w);
        if (better(d2, id, bd, bi)) { bd = d2; bi = id; improved = true; }
    }
    if (dB <= T) {
        const double d2 = dist2((double)B.x, (double)B.y, (double)B.z, qx, qy, qz);
        const uint32_t id = __float_as_uint(B.w);
        if (better(d2, id, bd, bi)) { bd = d2; bi = id; improved = true; }
    }
    for (uint32_t p = p1 + 8u; p < e; p += 8u) {
        const float4 P = pts[p];
        const float dx = P.x - qxf, dy = P.y - qyf, dz = P.z - qzf;
        if (__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) <= T) {
            const double d2 = dist2((double)P.x, (double)P.y, (double)P.z, qx, qy, qz);
            const uint32_t id = __float_as_uint(P.w);
            if (better(d2, id, bd, bi)) { bd = d2; bi = id; improved = true; }
        }
    }
    if (group8_min_u32(improved ? 0u : 1u) == 0u) {                   // some lane of the group has a new best: agree on it
        group8_argmin(bd, bi);
        bd32 = pyr_best32(bd);
    }
}

// The walk.  All 8 lanes of a group call it with the same query, their own `sub`, and the current best (bd, bi) -- +inf / none, or
// what the 2x2x2 block [bxa..bxb] x [bya..byb] x [bza..bzb] of level-0 cells gave (those cells are not scanned again; pass
// bxa > bxb for "nothing scanned").  It starts at the level-L0 node above the query's cell (cx, cy, cz) (pyr_hint_kernel), works
// through that subtree nearest child first, and climbs: before a parent is read, the REGION TEST -- every point not yet seen lies
// outside the cells the finished node covers, hence farther than the distance from the query to that region's faces (those with
// cells beyond them), less a slack that covers the fp32 cell assignment and this test's own fp32 arithmetic -- ends the search as
// soon as best <= that distance squared.  Node boxes are screened in fp32 (pyr_box_lb32 against pyr_best32: conservative by
// construction), points are decided in exact fp64: on return every lane holds the exact winner by (d2, index).
// s_off: the levels' slot offsets in LDS (a level is picked per GROUP: not uniform across the wave).
template <bool COUNT>
__device__ __forceinline__ void pyr_nn_search(const GridDesc &G, int nlev, const uint32_t *s_off, const PyrNode *__restrict__ nodes,
                                              const float4 *__restrict__ pts, float qxf, float qyf, float qzf, uint32_t sub,
                                              int cx, int cy, int cz, int L0,
                                              int bxa, int bxb, int bya, int byb, int bza, int bzb,
                                              double &bd, uint32_t &bi, uint32_t &npts, uint32_t &nruns, uint32_t &nnodes)
{
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    const float hf = (float)G.hd;                                     // exact: hd is a float widened
    const float qrx = qxf - G.ox, qry = qyf - G.oy, qrz = qzf - G.oz; // grid-relative: magnitudes of at most ~1024 cells inside the grid
    // the region test's fp32 arithmetic is only trusted near the grid (errors < 4e-4 cells); a query farther out than 2048 cells
    // climbs to the root, which costs a few visits and is exact by the box bounds alone
    const bool near_grid = fmaxf(fmaxf(fabsf(qrx), fabsf(qry)), fabsf(qrz)) < 2048.0f * hf;
    const int ci = (int)(sub & 1u), cj = (int)((sub >> 1) & 1u), ck = (int)(sub >> 2);
    int L = L0;                                 // current node's level; its children live at L - 1 (the virtual root sits at nlev)
    int X = cx >> L, Y = cy >> L, Z = cz >> L;
    int topL = L;                               // highest level whose pending bits are initialised
    uint32_t mypend = 0xFFFFFFFFu;              // bit L: my child of the current level-L node has not been taken yet
    bool fetch = true;
    float lb = __builtin_huge_valf(), bd32 = pyr_best32(bd);
    uint32_t cstart = 0, ccount = 0;
    for (;;) {
        const int cl = L - 1;
        if (fetch) {
            const int pgx = pyr_dim(G.gx, L), pgy = pyr_dim(G.gy, L);
            const float4 *n = reinterpret_cast<const float4 *>(nodes + s_off[cl] + 8u * (((size_t)Z * pgy + Y) * pgx + X) + sub);
            const float4 a = n[0], b = n[1];
            cstart = __float_as_uint(b.z);
            ccount = __float_as_uint(b.w);
            if (cl == 0) {                                            // a cell stage 0 has scanned already
                const int x = 2 * X + ci, y = 2 * Y + cj, z = 2 * Z + ck;
                if (x >= bxa && x <= bxb && y >= bya && y <= byb && z >= bza && z <= bzb) ccount = 0;
            }
            lb = pyr_box_lb32(a.x, a.y, a.z, a.w, b.x, b.y, qxf, qyf, qzf);      // empty node: +inf
            if (COUNT && sub == 0) nnodes += 1;
        }
        // nearest child that is still pending, holds points and can hold a point with d2 <= best: one 32-bit key per lane
        // (the fp32 bound's bit pattern orders like the value; its 3 low bits make room for the lane)
        const bool cand = ((mypend >> L) & 1u) && ccount != 0 && lb <= bd32;
        const uint32_t key = group8_min_u32(cand ? ((__float_as_uint(lb) & ~7u) | sub) : 0xFFFFFFFFu);
        if (key == 0xFFFFFFFFu) {                // nothing left below this node: region test, then up
            const int x0 = X << L, x1 = (X + 1) << L, y0 = Y << L, y1 = (Y + 1) << L, z0 = Z << L, z1 = (Z + 1) << L;
            const float inf = __builtin_huge_valf();
            float bound = inf;
            bound = fminf(bound, x0 > 0 ? qrx - (float)x0 * hf : inf);
            bound = fminf(bound, x1 < G.gx ? (float)x1 * hf - qrx : inf);
            bound = fminf(bound, y0 > 0 ? qry - (float)y0 * hf : inf);
            bound = fminf(bound, y1 < G.gy ? (float)y1 * hf - qry : inf);
            bound = fminf(bound, z0 > 0 ? qrz - (float)z0 * hf : inf);
            bound = fminf(bound, z1 < G.gz ? (float)z1 * hf - qrz : inf);
            if (bound == inf) break;                                  // the node covers the whole grid (always true at the virtual root)
            // slack: the fp32 cell assignment places a point at most 4e-4 cells across a face (cube_bound, kernels.hpp), the fp32
            // products and differences above err by < 3e-4 cells inside the grid; outside it (|q| large) the bound is negative anyway
            bound -= hf * (1.0f / 128.0f);
            if (near_grid && bound > 0.0f && bd32 <= bound * bound * (1.0f - 0x1p-20f)) break;
            const uint32_t from = (uint32_t)((X & 1) | ((Y & 1) << 1) | ((Z & 1) << 2));
            mypend |= 1u << L;                   // a later visit to this level starts with all children pending
            X >>= 1; Y >>= 1; Z >>= 1; L++;
            if (L > topL) {                      // first time this high: everything pending except the child just finished
                topL = L;
                mypend = (sub == from) ? (mypend & ~(1u << L)) : (mypend | (1u << L));
            }
            fetch = true;                        // the parent's children are not in registers (any more): read them (a cache hit on the way back)
            continue;
        }
        const uint32_t who = key & 7u;
        if (sub == who) mypend &= ~(1u << L);
        if (cl == 0) {                           // a level-0 cell: its points
            const uint32_t s = (uint32_t)__shfl((int)cstart, (int)who, 8), n = (uint32_t)__shfl((int)ccount, (int)who, 8);
            pyr_scan_leaf(pts, s, n, sub, qxf, qyf, qzf, qx, qy, qz, bd, bi, bd32);
            if (COUNT && sub == 0) { npts += n; nruns += 1; }
            fetch = false;                       // same node: every lane's (lb, count) is still valid
            continue;
        }
        X = 2 * X + (int)(who & 1u); Y = 2 * Y + (int)((who >> 1) & 1u); Z = 2 * Z + (int)(who >> 2);
        L--;
        mypend |= 1u << L;
        fetch = true;
    }
}

// ---- the fp32 walk (fast path) ----------------------------------------------------------------------------------------------
// The same walk with NO fp64 inside the loop: every point is screened in fp32 and each lane keeps (smallest distance, that point's
// coordinates and index, runner-up distance); exact duplicates of the current best (same coordinates) only lower its index and do
// not count as runner-up.  At the end the 8 lanes fold their triples.  If the runner-up lies outside the fp32 error band of the
// minimum (m2 > m1 (1 + 2^-19) + 2^-90, the band of coop_screen_rows), the minimum IS the exact winner by (d2, index) and its exact
// fp64 distance is evaluated once, from the coordinates in hand; otherwise (near-ties, ties between different points: rare off
// lattices) the query is left to the exact walk above.  Pruning uses the band too: a node is skipped only when its box bound
// exceeds the group's current minimum by more than the band, so every point inside the band is seen and counted as runner-up.
struct PyrBest { float m1, m2, bx, by, bz; uint32_t id; };

__device__ __forceinline__ void pyr_best_point(PyrBest &B, float d, float px, float py, float pz, uint32_t id)
{
    const bool same = px == B.bx && py == B.by && pz == B.bz;         // an exact duplicate of the current best (then d == m1)
    const bool lt = d < B.m1;
    B.m2 = lt ? B.m1 : (same ? B.m2 : fminf(B.m2, d));
    B.id = lt ? id : (same ? min(B.id, id) : B.id);
    B.bx = lt ? px : B.bx; B.by = lt ? py : B.by; B.bz = lt ? pz : B.bz;
    B.m1 = lt ? d : B.m1;
}

template <int CTRL>
__device__ __forceinline__ void pyr_best_fold_step(PyrBest &B)
{
    PyrBest O;
    O.m1 = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(B.m1))); O.m2 = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(B.m2)));
    O.bx = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(B.bx))); O.by = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(B.by)));
    O.bz = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(B.bz))); O.id = dpp_u32<CTRL>(B.id);
    const bool same = O.bx == B.bx && O.by == B.by && O.bz == B.bz;
    const bool lt = O.m1 < B.m1;
    // other strictly nearer: my best becomes a runner-up; same point: merge; otherwise the other's best is a runner-up of mine
    const float m2 = lt ? fminf(B.m1, O.m2) : (same ? fminf(B.m2, O.m2) : fminf(B.m2, fminf(O.m1, O.m2)));
    B.id = lt ? O.id : (same ? min(B.id, O.id) : B.id);
    B.bx = lt ? O.bx : B.bx; B.by = lt ? O.by : B.by; B.bz = lt ? O.bz : B.bz;
    B.m1 = lt ? O.m1 : B.m1;
    B.m2 = m2;
}

// returns true when (bd, bi) is the exact answer; false: undecided (bd / bi untouched), the exact walk has to run
template <bool COUNT>
__device__ __forceinline__ bool pyr_nn_search_fast(const GridDesc &G, int nlev, const uint32_t *s_off, const PyrNode *__restrict__ nodes,
                                                   const float4 *__restrict__ pts, float qxf, float qyf, float qzf, uint32_t sub,
                                                   int cx, int cy, int cz, int L0,
                                                   int bxa, int bxb, int bya, int byb, int bza, int bzb,
                                                   double &bd, uint32_t &bi, float &lim_out, uint32_t &npts, uint32_t &nruns, uint32_t &nnodes)
{
    const float inf = __builtin_huge_valf();
    const float hf = (float)G.hd;
    const float qrx = qxf - G.ox, qry = qyf - G.oy, qrz = qzf - G.oz;
    const bool near_grid = fmaxf(fmaxf(fabsf(qrx), fabsf(qry)), fabsf(qrz)) < 2048.0f * hf;
    const int ci = (int)(sub & 1u), cj = (int)((sub >> 1) & 1u), ck = (int)(sub >> 2);
    int L = L0, X = cx >> L, Y = cy >> L, Z = cz >> L, topL = L;
    uint32_t mypend = 0xFFFFFFFFu;
    bool fetch = true;
    // what stage 0 found (exact) enters as lane 0's best with unknown coordinates (NaN never compares equal): anything the walk
    // finds inside its band makes the query undecided
    PyrBest B;
    B.m1 = (sub == 0 && bi != kNoIndex) ? (float)bd : inf;
    B.m2 = inf; B.bx = B.by = B.bz = __builtin_nanf(""); B.id = bi;
    float gm = __uint_as_float((uint32_t)__shfl((int)__float_as_uint(B.m1), 0, 8));       // the group's smallest distance so far
    float lim = (gm * (1.0f + 0x1p-19f) + 0x1p-90f) * (1.0f + 0x1p-20f);                   // boxes / regions beyond this cannot matter
    float lb = inf;
    uint32_t cstart = 0, ccount = 0;
    for (;;) {
        const int cl = L - 1;
#ifdef PCT_AB_ITERSTAT
        if (COUNT && sub == 0) npts += 1;                                 // A/B instrumentation: iterations of the walk instead of points
#endif
        if (fetch) {
            const uint32_t pgx = (uint32_t)pyr_dim(G.gx, L), pgy = (uint32_t)pyr_dim(G.gy, L);
            const float4 *n = reinterpret_cast<const float4 *>(nodes + (s_off[cl] + 8u * (((uint32_t)Z * pgy + (uint32_t)Y) * pgx + (uint32_t)X) + sub));
            const float4 a = n[0], b = n[1];
            cstart = __float_as_uint(b.z);
            ccount = __float_as_uint(b.w);
            if (cl == 0) {
                const int x = 2 * X + ci, y = 2 * Y + cj, z = 2 * Z + ck;
                if (x >= bxa && x <= bxb && y >= bya && y <= byb && z >= bza && z <= bzb) ccount = 0;
            }
            lb = pyr_box_lb32(a.x, a.y, a.z, a.w, b.x, b.y, qxf, qyf, qzf);
            if (COUNT && sub == 0) nnodes += 1;
        }
        const bool cand = ((mypend >> L) & 1u) && ccount != 0 && lb <= lim;
        const uint32_t key = group8_min_u32(cand ? ((__float_as_uint(lb) & ~7u) | sub) : 0xFFFFFFFFu);
        if (key == 0xFFFFFFFFu) {
            const int x0 = X << L, x1 = (X + 1) << L, y0 = Y << L, y1 = (Y + 1) << L, z0 = Z << L, z1 = (Z + 1) << L;
            float bound = inf;
            bound = fminf(bound, x0 > 0 ? qrx - (float)x0 * hf : inf);
            bound = fminf(bound, x1 < G.gx ? (float)x1 * hf - qrx : inf);
            bound = fminf(bound, y0 > 0 ? qry - (float)y0 * hf : inf);
            bound = fminf(bound, y1 < G.gy ? (float)y1 * hf - qry : inf);
            bound = fminf(bound, z0 > 0 ? qrz - (float)z0 * hf : inf);
            bound = fminf(bound, z1 < G.gz ? (float)z1 * hf - qrz : inf);
            if (bound == inf) break;
            bound -= hf * (1.0f / 128.0f);
            if (near_grid && bound > 0.0f && lim <= bound * bound * (1.0f - 0x1p-20f)) break;
            const uint32_t from = (uint32_t)((X & 1) | ((Y & 1) << 1) | ((Z & 1) << 2));
            mypend |= 1u << L;
            X >>= 1; Y >>= 1; Z >>= 1; L++;
            if (L > topL) {
                topL = L;
                mypend = (sub == from) ? (mypend & ~(1u << L)) : (mypend | (1u << L));
            }
            fetch = true;
            continue;
        }
        const uint32_t who = key & 7u;
        if (sub == who) mypend &= ~(1u << L);
        if (cl == 0) {
            const uint32_t s = dpp_bcast8(cstart, who), n = dpp_bcast8(ccount, who);
            const uint32_t e = s + n, last = e - 1, p0 = s + sub, p1 = p0 + 8u;
            const float4 A = pts[min(p0, last)], Bp = pts[min(p1, last)];
            {
                const float dx = A.x - qxf, dy = A.y - qyf, dz = A.z - qzf;
                const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (p0 < e) pyr_best_point(B, d, A.x, A.y, A.z, __float_as_uint(A.w));
            }
            {
                const float dx = Bp.x - qxf, dy = Bp.y - qyf, dz = Bp.z - qzf;
                const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (p1 < e) pyr_best_point(B, d, Bp.x, Bp.y, Bp.z, __float_as_uint(Bp.w));
            }
            for (uint32_t p = p1 + 8u; p < e; p += 8u) {              // cells of more than 16 points
                const float4 P = pts[p];
                const float dx = P.x - qxf, dy = P.y - qyf, dz = P.z - qzf;
                pyr_best_point(B, __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)), P.x, P.y, P.z, __float_as_uint(P.w));
            }
            gm = group8_min_f32(B.m1);
            lim = (gm * (1.0f + 0x1p-19f) + 0x1p-90f) * (1.0f + 0x1p-20f);
#ifndef PCT_AB_ITERSTAT
            if (COUNT && sub == 0) { npts += n; nruns += 1; }
#endif
            fetch = false;
            continue;
        }
        X = 2 * X + (int)(who & 1u); Y = 2 * Y + (int)((who >> 1) & 1u); Z = 2 * Z + (int)(who >> 2);
        L--;
        mypend |= 1u << L;
        fetch = true;
    }
    pyr_best_fold_step<kDppXor1>(B);
    pyr_best_fold_step<kDppXor2>(B);
    pyr_best_fold_step<kDppHalfMirror>(B);
    // undecided: a second point inside the band (or nothing found at all).  Everything that can win or tie lies within lim_out of the
    // query: the exact walk that takes over starts with that bound instead of discovering it
    lim_out = (B.m1 * (1.0f + 0x1p-19f) + 0x1p-90f) * (1.0f + 0x1p-20f);
    if (!(B.m2 > B.m1 * (1.0f + 0x1p-19f) + 0x1p-90f)) return false;
    if (B.bx == B.bx) {                                               // the winner is a point the walk saw: its exact distance, once
        bd = dist2((double)B.bx, (double)B.by, (double)B.bz, (double)qxf, (double)qyf, (double)qzf);
        bi = B.id;
    }                                                                 // else: stage 0's exact (bd, bi) stands
    return true;
}

// The batch kernels for clouds that carry the pyramid: stage 0 of the cell-pruned search (the 2x2x2 block on the query's side of
// its cell, as coop_stage0 in kernels.hpp -- it decides nearly every query that sits inside a dense region; skipped when the block
// holds no point), then a walk for whatever it leaves undecided.  Same launch shape and arguments as nn_grid_coop_kernel: 8 lanes
// per query, 32 queries per block, XCD-contiguous block order over the sorted batch.
//   FAST = true : the fp32 walk; queries it cannot decide (near-ties) are appended to `todo` ({count, ticket, slots...})
//   FAST = false: the exact walk, for every query (todo == nullptr) or for the listed slots only (a fixed grid strides over the
//                 list; its last block to finish clears count and ticket for the next batch: no memset between batches, and a
//                 captured graph replays correctly)
template <bool COUNT, bool FAST>
__device__ __forceinline__ void pyr_answer(const GridDesc &G, const PyrDesc &PD, const uint32_t *s_off, const PyrNode *__restrict__ nodes,
                                           const unsigned char *__restrict__ hint, const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                           const float *__restrict__ q, uint32_t index_base, const float4 *__restrict__ qsorted, uint32_t slot,
                                           uint32_t sub, uint32_t *__restrict__ out_idx, double *__restrict__ out_d2, int sorted_out,
                                           uint32_t *__restrict__ todo, uint32_t &npts, uint32_t &nruns, uint32_t &nnodes)
{
    uint32_t t = slot;
    float qxf, qyf, qzf;
    if (qsorted) {
        const float4 R = qsorted[slot];
        qxf = R.x; qyf = R.y; qzf = R.z; t = sorted_out ? slot : __float_as_uint(R.w);
    } else {
        qxf = q[3 * t]; qyf = q[3 * t + 1]; qzf = q[3 * t + 2];
    }
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    const int cx = cell_coord(qxf, G.ox, G.inv_h, G.gx), cy = cell_coord(qyf, G.oy, G.inv_h, G.gy), cz = cell_coord(qzf, G.oz, G.inv_h, G.gz);
    const float fx = (qxf - G.ox) * G.inv_h - (float)cx, fy = (qyf - G.oy) * G.inv_h - (float)cy, fz = (qzf - G.oz) * G.inv_h - (float)cz;
    const int xa = max(fx < 0.5f ? cx - 1 : cx, 0), xb = min(fx < 0.5f ? cx : cx + 1, G.gx - 1);
    const int ya = max(fy < 0.5f ? cy - 1 : cy, 0), yb = min(fy < 0.5f ? cy : cy + 1, G.gy - 1);
    const int za = max(fz < 0.5f ? cz - 1 : cz, 0), zb = min(fz < 0.5f ? cz : cz + 1, G.gz - 1);
    uint32_t rs[4], re[4];
    const int L0 = (int)hint[cell_lin(G, cx, cy, cz)];             // requested together with the run bounds
    {
        const int ri = (int)sub & 3;
        const bool ok = !((ri >> 1) && zb == za) && !((ri & 1) && yb == ya);
        const uint32_t row = cell_lin(G, 0, (ri & 1) ? yb : ya, (ri >> 1) ? zb : za);
        const uint32_t a = cell_start[row + xa], b = cell_start[row + xb + 1];
        const uint32_t my_s = a, my_e = ok ? b : a;
#ifndef PCT_AB_ITERSTAT
        if (COUNT && sub < 4) { npts += my_e - my_s; nruns += ok ? 1u : 0u; }
#endif
        rs[0] = dpp_u32<kDppQuadBcast0>(my_s); rs[1] = dpp_u32<kDppQuadBcast1>(my_s); rs[2] = dpp_u32<kDppQuadBcast2>(my_s); rs[3] = dpp_u32<kDppQuadBcast3>(my_s);
        re[0] = dpp_u32<kDppQuadBcast0>(my_e); re[1] = dpp_u32<kDppQuadBcast1>(my_e); re[2] = dpp_u32<kDppQuadBcast2>(my_e); re[3] = dpp_u32<kDppQuadBcast3>(my_e);
    }
    double bd = __builtin_huge_val();
    uint32_t bi = kNoIndex;
    bool undecided = true;
    if ((re[0] - rs[0]) + (re[1] - rs[1]) + (re[2] - rs[2]) + (re[3] - rs[3]) != 0u) {        // free space: nothing to screen
        coop_screen_rows<4, 2>(pts, rs, re, sub, qxf, qyf, qzf, qx, qy, qz, bd, bi);
        double bound = __builtin_huge_val();
        if (xa > 0) bound = fmin(bound, qx - (G.oxd + (double)xa * G.hd));
        if (xb < G.gx - 1) bound = fmin(bound, (G.oxd + (double)(xb + 1) * G.hd) - qx);
        if (ya > 0) bound = fmin(bound, qy - (G.oyd + (double)ya * G.hd));
        if (yb < G.gy - 1) bound = fmin(bound, (G.oyd + (double)(yb + 1) * G.hd) - qy);
        if (za > 0) bound = fmin(bound, qz - (G.ozd + (double)za * G.hd));
        if (zb < G.gz - 1) bound = fmin(bound, (G.ozd + (double)(zb + 1) * G.hd) - qz);
        if (bound == __builtin_huge_val()) undecided = false;                        // the block covers the whole grid
        else {
            bound -= G.hd * (1.0 / 256.0);
            undecided = !(bound > 0.0 && bd <= bound * bound);
        }
    }
    if (undecided) {
        if (FAST) {
            float lim;
            if (!pyr_nn_search_fast<COUNT>(G, PD.nlev, s_off, nodes, pts, qxf, qyf, qzf, sub, cx, cy, cz, L0, xa, xb, ya, yb, za, zb, bd, bi, lim, npts, nruns, nnodes)) {
                // (measured: the exact walk right here instead of a list costs the kernel its registers: 1.15 against 1.18e9 q/s)
                if (sub == 0) {
                    const uint32_t e = atomicAdd(&todo[0], 1u);
                    todo[2u + 2u * e] = slot;
                    todo[3u + 2u * e] = __float_as_uint(lim);
                }
                return;
            }
        } else {
            pyr_nn_search<COUNT>(G, PD.nlev, s_off, nodes, pts, qxf, qyf, qzf, sub, cx, cy, cz, L0, xa, xb, ya, yb, za, zb, bd, bi, npts, nruns, nnodes);
        }
    }
    if (sub == 0) {
        out_idx[t] = (bi == kNoIndex) ? kNoIndex : bi + index_base;
        out_d2[t] = bd;
    }
}

template <bool COUNT>
__device__ __forceinline__ void pyr_commit_work(uint32_t npts, uint32_t nruns, uint32_t nnodes, WorkCounters *__restrict__ work)
{
    if (COUNT) {
        unsigned long long a = npts, b = nruns, n = nnodes;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += (unsigned long long)__shfl_xor((long long)a, off, kWave);
            b += (unsigned long long)__shfl_xor((long long)b, off, kWave);
            n += (unsigned long long)__shfl_xor((long long)n, off, kWave);
        }
        if ((threadIdx.x & 63) == 0) {
            WorkCounters *w = work + (blockIdx.x & (kWorkSlots - 1));
            atomicAdd(&w->points, a); atomicAdd(&w->cells, b); atomicAdd(&w->nodes, n);
        }
    }
}

#ifndef PCT_AB_PYR_WAVES
#define PCT_AB_PYR_WAVES 6      // minimum 6, maximum 8: the current text lands at 63 VGPRs = 8 waves; CAPPING it at 7 / 6 / 5 waves per SIMD gives
                                // 1.14 / 1.07 / 0.97e9 q/s against 1.19-1.20e9 on the 10 M pillar cloud (profiles/r03_ab_occupancy_cap.txt)
#endif
template <bool COUNT, bool FAST>
#ifndef PCT_AB_PYR_WAVES_MAX
#define PCT_AB_PYR_WAVES_MAX 8
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCT_AB_PYR_WAVES, PCT_AB_PYR_WAVES_MAX))) void nn_grid_pyr_kernel(GridDesc G, PyrDesc PD, const PyrNode *__restrict__ nodes, const unsigned char *__restrict__ hint,
                                                          const float4 *__restrict__ pts,
                                                          const uint32_t *__restrict__ cell_start, const float *__restrict__ q, uint32_t Q,
                                                          uint32_t index_base, const float4 *__restrict__ qsorted, uint32_t *__restrict__ out_idx,
                                                          double *__restrict__ out_d2, WorkCounters *__restrict__ work, int sorted_out,
                                                          uint32_t *__restrict__ todo)
{
    __shared__ uint32_t s_off[kPyrMaxLevels];
    if (threadIdx.x < (uint32_t)kPyrMaxLevels) s_off[threadIdx.x] = PD.off[threadIdx.x];
    __syncthreads();
    const uint32_t sub = threadIdx.x & (kCoop - 1);
    const uint32_t bslot = qsorted ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t slot = bslot * (256 / kCoop) + (threadIdx.x / kCoop);
    uint32_t npts = 0, nruns = 0, nnodes = 0;
    if (slot < Q)                                     // uniform within a group of 8 lanes
        pyr_answer<COUNT, FAST>(G, PD, s_off, nodes, hint, pts, cell_start, q, index_base, qsorted, slot, sub, out_idx, out_d2, sorted_out, todo, npts, nruns, nnodes);
#ifdef PCT_AB_ITERSTAT
    if (COUNT) {                                      // `runs` := 8 x the longest walk of the wave (what the wave pays), `points` := the walks' own lengths
        uint32_t m = npts;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, kWave));
        nruns = (threadIdx.x & 63) == 0 ? 8u * m : 0u;
    }
#endif
    pyr_commit_work<COUNT>(npts, nruns, nnodes, work);
}

// the exact walk for the slots the fast kernel listed: a fixed grid strides over the list
template <bool COUNT>
__global__ __launch_bounds__(256) void nn_grid_pyr_todo_kernel(GridDesc G, PyrDesc PD, const PyrNode *__restrict__ nodes, const unsigned char *__restrict__ hint,
                                                               const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                                               const float *__restrict__ q, uint32_t index_base, const float4 *__restrict__ qsorted,
                                                               uint32_t *__restrict__ out_idx, double *__restrict__ out_d2, WorkCounters *__restrict__ work,
                                                               int sorted_out, uint32_t *__restrict__ todo)
{
    __shared__ uint32_t s_off[kPyrMaxLevels];
    __shared__ uint32_t s_count;
    if (threadIdx.x < (uint32_t)kPyrMaxLevels) s_off[threadIdx.x] = PD.off[threadIdx.x];
    if (threadIdx.x == 0) s_count = __hip_atomic_load(&todo[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t count = s_count, sub = threadIdx.x & (kCoop - 1);
    uint32_t npts = 0, nruns = 0, nnodes = 0;
    for (uint32_t e = blockIdx.x * (256 / kCoop) + (threadIdx.x / kCoop); e < count; e += gridDim.x * (256 / kCoop)) {
        // the exact walk with the bound the fp32 walk established: everything that can win or tie is within `lim` of the query, so the
        // walk prunes from its first node on (no stage 0, nothing excluded: ~10 dependent trips instead of ~30 for these stragglers)
        const uint32_t slot = todo[2u + 2u * e];
        const float lim = __uint_as_float(todo[3u + 2u * e]);
        uint32_t t = slot;
        float qxf, qyf, qzf;
        if (qsorted) {
            const float4 R = qsorted[slot];
            qxf = R.x; qyf = R.y; qzf = R.z; t = sorted_out ? slot : __float_as_uint(R.w);
        } else {
            qxf = q[3 * t]; qyf = q[3 * t + 1]; qzf = q[3 * t + 2];
        }
        const int cx = cell_coord(qxf, G.ox, G.inv_h, G.gx), cy = cell_coord(qyf, G.oy, G.inv_h, G.gy), cz = cell_coord(qzf, G.oz, G.inv_h, G.gz);
        const int L0 = (int)hint[cell_lin(G, cx, cy, cz)];
        double bd = (double)lim;                       // +inf when the fp32 walk found nothing (non-finite query)
        uint32_t bi = kNoIndex;
        pyr_nn_search<COUNT>(G, PD.nlev, s_off, nodes, pts, qxf, qyf, qzf, sub, cx, cy, cz, L0, 1, 0, 1, 0, 1, 0, bd, bi, npts, nruns, nnodes);
        if (sub == 0) {
            out_idx[t] = (bi == kNoIndex) ? kNoIndex : bi + index_base;
            out_d2[t] = (bi == kNoIndex) ? __builtin_huge_val() : bd;
        }
    }
    pyr_commit_work<COUNT>(npts, nruns, nnodes, work);
    // the last block to finish leaves the list empty for the next batch (every block has read the count by then)
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&todo[1], 1u) == gridDim.x - 1u) {
        __hip_atomic_store(&todo[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&todo[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace pct
