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

struct PyrDesc {
    int nlev;                          // levels 0 .. nlev-1; the top level has at most 2 nodes per axis
    uint32_t off[kPyrMaxLevels];       // first node of level l in the node array
};

__device__ __host__ __forceinline__ int pyr_dim(int g, int l) { return (g + (1 << l) - 1) >> l; }

// level 0: 8 lanes per cell fold the cell's points
__global__ __launch_bounds__(256) void pyr_leaf_kernel(GridDesc G, const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                                       PyrNode *__restrict__ nodes, uint32_t *__restrict__ n_empty)
{
    const uint32_t cell = blockIdx.x * 32u + (threadIdx.x >> 3), sub = threadIdx.x & 7u;
    float lo[3] = { __builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf() };
    float hi[3] = { -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf() };
    uint32_t s = 0, e = 0;
    if (cell < G.ncells) { s = cell_start[cell]; e = cell_start[cell + 1]; }
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
    if (cell < G.ncells && sub == 0) {
        float4 *out = reinterpret_cast<float4 *>(nodes + cell);
        out[0] = make_float4(lo[0], lo[1], lo[2], hi[0]);
        out[1] = make_float4(hi[1], hi[2], __uint_as_float(s), __uint_as_float(e - s));
    }
    if (n_empty) {
        const unsigned long long em = __builtin_amdgcn_ballot_w64(cell < G.ncells && sub == 0 && e == s);
        if ((threadIdx.x & 63) == 0 && em) atomicAdd(n_empty, (uint32_t)__builtin_popcountll(em));
    }
}

// level l from level l - 1: one thread per node merges its (up to) 8 children
__global__ __launch_bounds__(256) void pyr_up_kernel(int gx, int gy, int gz, int cgx, int cgy, int cgz, const PyrNode *__restrict__ child,
                                                     PyrNode *__restrict__ parent)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)gx * (uint32_t)gy * (uint32_t)gz) return;
    const int x = (int)(i % (uint32_t)gx), y = (int)((i / (uint32_t)gx) % (uint32_t)gy), z = (int)(i / ((uint32_t)gx * (uint32_t)gy));
    float lo[3] = { __builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf() };
    float hi[3] = { -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf() };
    unsigned long long cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int cx = 2 * x + (k & 1), cy = 2 * y + ((k >> 1) & 1), cz = 2 * z + (k >> 2);
        if (cx < cgx && cy < cgy && cz < cgz) {
            const float4 *n = reinterpret_cast<const float4 *>(child + ((size_t)cz * cgy + cy) * cgx + cx);
            const float4 a = n[0], b = n[1];
            lo[0] = fminf(lo[0], a.x); lo[1] = fminf(lo[1], a.y); lo[2] = fminf(lo[2], a.z);
            hi[0] = fmaxf(hi[0], a.w); hi[1] = fmaxf(hi[1], b.x); hi[2] = fmaxf(hi[2], b.y);
            cnt += __float_as_uint(b.w);
        }
    }
    float4 *out = reinterpret_cast<float4 *>(parent + i);
    out[0] = make_float4(lo[0], lo[1], lo[2], hi[0]);
    out[1] = make_float4(hi[1], hi[2], __uint_as_float(0u), __uint_as_float(cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt));
}

// squared distance from q to the box, in dist2's arithmetic (see the header: a true lower bound of the COMPUTED distances)
__device__ __forceinline__ double pyr_box_lb(float lox, float loy, float loz, float hix, float hiy, float hiz, double qx, double qy, double qz)
{
    const double gx = fmax(fmax((double)lox - qx, qx - (double)hix), 0.0);
    const double gy = fmax(fmax((double)loy - qy, qy - (double)hiy), 0.0);
    const double gz = fmax(fmax((double)loz - qz, qz - (double)hiz), 0.0);
    double s = gx * gx;
    s = s + gy * gy;
    s = s + gz * gz;
    return s;
}

// The walk.  All 8 lanes of a group call it with the same query, their own `sub`, and the current best (bd, bi) -- +inf / none, or
// what the 2x2x2 block [bxa..bxb] x [bya..byb] x [bza..bzb] of level-0 cells gave (those cells are not scanned again; pass
// bxa > bxb for "nothing scanned").  On return every lane holds the exact winner by (d2, index).
// s_off: the levels' node offsets in LDS (a level is picked per GROUP, so an index into kernel-argument space would not be uniform).
template <bool COUNT>
__device__ __forceinline__ void pyr_nn_search(const GridDesc &G, int nlev, const uint32_t *s_off, const PyrNode *__restrict__ nodes,
                                              const float4 *__restrict__ pts, float qxf, float qyf, float qzf, uint32_t sub,
                                              int bxa, int bxb, int bya, int byb, int bza, int bzb,
                                              double &bd, uint32_t &bi, uint32_t &npts, uint32_t &nruns, uint32_t &nnodes)
{
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    const int ci = (int)(sub & 1u), cj = (int)((sub >> 1) & 1u), ck = (int)(sub >> 2);
    int L = nlev, X = 0, Y = 0, Z = 0;          // current node (level L; the virtual root sits at level nlev); its children at L - 1
    uint32_t mypend = 0xFFFFFFFFu;              // bit L: my child of the current level-L node has not been taken yet
    bool fetch = true;
    double lb = __builtin_huge_val();
    uint32_t cstart = 0, ccount = 0;
    for (;;) {
        const int cl = L - 1;
        if (fetch) {
            const int gx = pyr_dim(G.gx, cl), gy = pyr_dim(G.gy, cl), gz = pyr_dim(G.gz, cl);
            const int x = 2 * X + ci, y = 2 * Y + cj, z = 2 * Z + ck;
            lb = __builtin_huge_val();
            ccount = 0; cstart = 0;
            if (x < gx && y < gy && z < gz) {
                const float4 *n = reinterpret_cast<const float4 *>(nodes + s_off[cl] + ((size_t)z * gy + y) * gx + x);
                const float4 a = n[0], b = n[1];
                cstart = __float_as_uint(b.z);
                ccount = __float_as_uint(b.w);
                if (cl == 0 && x >= bxa && x <= bxb && y >= bya && y <= byb && z >= bza && z <= bzb) ccount = 0;      // scanned by stage 0
                if (ccount) lb = pyr_box_lb(a.x, a.y, a.z, a.w, b.x, b.y, qx, qy, qz);
            }
            if (COUNT && sub == 0) nnodes += 1;
        }
        // nearest child that is still pending, holds points and can hold a point with d2 <= best
        const bool cand = ((mypend >> L) & 1u) && ccount != 0 && lb <= bd;
        double m = cand ? lb : __builtin_huge_val();
        uint32_t who = cand ? sub : 8u;
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            const double om = __shfl_xor(m, off, kWave);
            const uint32_t ow = (uint32_t)__shfl_xor((int)who, off, kWave);
            if (om < m || (om == m && ow < who)) { m = om; who = ow; }
        }
        if (who >= 8u) {                         // nothing left below this node
            if (L == nlev) break;
            mypend |= 1u << L;                   // a later visit to this level starts with all children pending
            X >>= 1; Y >>= 1; Z >>= 1; L++;
            fetch = true;                        // the parent's children were overwritten on the way down: read them again (cache hit)
            continue;
        }
        if (sub == who) mypend &= ~(1u << L);
        if (cl == 0) {                           // a level-0 cell: exact scan by the 8 lanes
            const uint32_t s = (uint32_t)__shfl((int)cstart, (int)who, 8), n = (uint32_t)__shfl((int)ccount, (int)who, 8);
            coop_scan_exact(pts, s, s + n, sub, qx, qy, qz, bd, bi);
            coop_argmin8(bd, bi);
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

// The batch kernel for clouds that carry the pyramid: stage 0 of the cell-pruned search (the 2x2x2 block on the query's side of
// its cell, kernels.hpp coop_stage0 -- it decides nearly every query that sits inside a dense region), then the walk above for
// whatever it leaves undecided.  Same launch shape and arguments as nn_grid_coop_kernel: 8 lanes per query, 32 queries per block,
// XCD-contiguous block order over the sorted batch.
template <bool COUNT>
__global__ __launch_bounds__(256) void nn_grid_pyr_kernel(GridDesc G, PyrDesc PD, const PyrNode *__restrict__ nodes, const float4 *__restrict__ pts,
                                                          const uint32_t *__restrict__ cell_start, const float *__restrict__ q, uint32_t Q,
                                                          uint32_t index_base, const float4 *__restrict__ qsorted, uint32_t *__restrict__ out_idx,
                                                          double *__restrict__ out_d2, WorkCounters *__restrict__ work, int sorted_out)
{
    __shared__ uint32_t s_off[kPyrMaxLevels];
    if (threadIdx.x < (uint32_t)kPyrMaxLevels) s_off[threadIdx.x] = PD.off[threadIdx.x];
    __syncthreads();
    const uint32_t sub = threadIdx.x & (kCoop - 1);
    const uint32_t bslot = qsorted ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t slot = bslot * (256 / kCoop) + (threadIdx.x / kCoop);
    uint32_t npts = 0, nruns = 0, nnodes = 0;
    if (slot < Q) {                                   // uniform within a group of 8 lanes
        uint32_t t = slot;
        float qxf, qyf, qzf;
        if (qsorted) {
            const float4 R = qsorted[slot];
            qxf = R.x; qyf = R.y; qzf = R.z; t = sorted_out ? slot : __float_as_uint(R.w);
        } else {
            qxf = q[3 * t]; qyf = q[3 * t + 1]; qzf = q[3 * t + 2];
        }
        double bd = __builtin_huge_val();
        uint32_t bi = kNoIndex;
        int xa = 1, xb = 0, ya = 1, yb = 0, za = 1, zb = 0;                   // empty block = nothing scanned yet
        bool undecided = true;
        if (G.octant_first) {
            undecided = coop_stage0<COUNT>(G, pts, cell_start, qxf, qyf, qzf, sub, bd, bi, npts, nruns);
            // the block stage 0 scanned (same expressions as coop_stage0)
            const int cx = cell_coord(qxf, G.ox, G.inv_h, G.gx), cy = cell_coord(qyf, G.oy, G.inv_h, G.gy), cz = cell_coord(qzf, G.oz, G.inv_h, G.gz);
            const float fx = (qxf - G.ox) * G.inv_h - (float)cx, fy = (qyf - G.oy) * G.inv_h - (float)cy, fz = (qzf - G.oz) * G.inv_h - (float)cz;
            xa = max(fx < 0.5f ? cx - 1 : cx, 0); xb = min(fx < 0.5f ? cx : cx + 1, G.gx - 1);
            ya = max(fy < 0.5f ? cy - 1 : cy, 0); yb = min(fy < 0.5f ? cy : cy + 1, G.gy - 1);
            za = max(fz < 0.5f ? cz - 1 : cz, 0); zb = min(fz < 0.5f ? cz : cz + 1, G.gz - 1);
        }
        if (undecided)
            pyr_nn_search<COUNT>(G, PD.nlev, s_off, nodes, pts, qxf, qyf, qzf, sub, xa, xb, ya, yb, za, zb, bd, bi, npts, nruns, nnodes);
        if (sub == 0) {
            out_idx[t] = (bi == kNoIndex) ? kNoIndex : bi + index_base;
            out_d2[t] = bd;
        }
    }
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

}  // namespace pct
