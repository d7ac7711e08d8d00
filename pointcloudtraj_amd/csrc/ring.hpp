// ring.hpp -- incremental cell index for the ROLLING obstacle map (config C5) and the fused replan batch.
//
// The planner's cloud callback replaces the obstacle index on every sensor frame
// (safeRegionRrtStar::setInput, Planner/src/corridor_finder.cpp:93-99: a full FLANN rebuild) and then runs the
// collision check (sim_planning_demo.cpp:159-178).  For a rolling window (pct_cloud_append_aos: the newest frame overwrites
// the oldest ring slots) a rebuild per frame costs more than the queries it serves, so the window keeps an index that is
// UPDATED in place:
//
//   * cells are world-anchored: cell(p) = floor(p / h) per axis in fp64 (exact enough that the h/256 slack of the search
//     bound is never needed), folded into a power-of-two table by (c & (g - 1)) -- toroidal addressing, so the table follows
//     the drone without ever being re-centred.  Two world cells that fold onto one bucket merely share it: every distance is
//     computed from the stored coordinates, and the termination bound below only speaks about buckets NOT yet visited.
//   * a bucket is a short queue of kRingK {x, y, z, ring slot} records with monotonic head/tail counters; where[slot] remembers
//     the place a ring slot's record was filed.  Evicting a point = marking ITS record dead (no search) and letting the
//     bucket's head move past dead records; a rolling window evicts in arrival order, so heads keep up with the evictions.
//   * a bucket that is full sends the newcomer to one global overflow queue (same discipline).  Queries scan that queue
//     exhaustively; it is empty unless the cloud has more than kRingK points in one cell (bulk duplicates, surfaces sampled
//     far finer than the cell).
//
// Search = the same expanding-cube search as the cell-sorted index (kernels.hpp section 4), same fp64 arithmetic
// ((dx*dx + dy*dy) + dz*dz, no FMA), winner by (d2, ring slot): lowest slot on exact ties.
#pragma once
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace pct {

constexpr uint32_t kRingK = 32;                 // records per bucket the index starts with (RingDesc::K, a power of two; the host doubles it,
constexpr uint32_t kRingKMax = 256;             // up to this, when the overflow queue shows that the window's cells hold more: surfaces on a
                                                // lattice finer than the cell, the same points sensed again frame after frame)
constexpr int kRingCellClamp = 1000000000;      // |cell coordinate| limit (non-finite / absurd coordinates land on the limit)

struct RingDesc {
    double inv_h, h;
    int gx, gy, gz;          // bucket table dimensions, powers of two
    int lx, ly;              // log2(gx), log2(gy)
    uint32_t K;              // records per bucket (power of two)
    // overflow queue capacity - 1.  The queue holds one entry per live spilled point plus dead entries its head has not passed yet;
    // the head is blocked only by a live entry, and entries behind a live head that are already dead can only belong to the same
    // insert launch as that head entry (younger launches are evicted later), so its length never exceeds capacity + one launch
    // <= 2 x cloud capacity: the table is a power of two >= 2 x capacity + 16.  `status` (host-mapped, two words) still gets
    // {1, length} from a launch that would overrun it and {0, length} otherwise: the host rebuilds the index from the SoA arrays.
    uint32_t ovf_mask;
    uint32_t *status;
};

// device-resident state the append kernels maintain and the query kernels read: a captured hipGraph stays valid across appends
struct RingState {
    uint32_t ovf_head, ovf_tail;     // monotonic; live entries are [head, tail)
    uint32_t count;                  // valid points in the ring
    uint32_t ticket;                 // blocks of an evict launch that have finished marking (reset by the last one)
};

__device__ __forceinline__ int ring_cell_coord(double v, double inv_h, bool &wild)
{
    const double c = floor(v * inv_h);
    if (!(c > -(double)kRingCellClamp && c < (double)kRingCellClamp)) {     // also NaN
        wild = true;
        return c > 0.0 ? kRingCellClamp : -kRingCellClamp;
    }
    return (int)c;
}

__device__ __forceinline__ uint32_t ring_lin(const RingDesc &R, int cx, int cy, int cz)
{
    return ((uint32_t)(cz & (R.gz - 1)) << (R.lx + R.ly)) | ((uint32_t)(cy & (R.gy - 1)) << R.lx) | (uint32_t)(cx & (R.gx - 1));
}

__device__ __forceinline__ uint32_t ring_bucket_of(const RingDesc &R, float x, float y, float z)
{
    bool wild = false;
    const int cx = ring_cell_coord((double)x, R.inv_h, wild), cy = ring_cell_coord((double)y, R.inv_h, wild),
              cz = ring_cell_coord((double)z, R.inv_h, wild);
    return ring_lin(R, cx, cy, cz);
}

// where[slot]: the place a ring slot's record was filed -- bit 31 set = position in the overflow queue, clear = sequence number
// inside its bucket.  Eviction goes straight to the record (no search) and marks it dead; heads then advance past dead records.
constexpr uint32_t kRingInOvf = 0x80000000u;
constexpr uint32_t kRingDead = 0xFFFFFFFFu;      // a record's id word once its point has left the window
constexpr uint32_t kRingUnfiled = 0xFFFFFFFFu;   // where[] of a point the overflow queue had no room for (an error state the host repairs)

// file one point under its bucket (or the overflow queue when the bucket is full); heads do not move while this runs
__device__ __forceinline__ void ring_file(const RingDesc &R, float px, float py, float pz, uint32_t slot, uint2 *__restrict__ ht,
                                          float4 *__restrict__ slots, float4 *__restrict__ ovf, uint32_t *__restrict__ where,
                                          RingState *__restrict__ st)
{
    const uint32_t b = ring_bucket_of(R, px, py, pz);
    const float4 rec = make_float4(px, py, pz, __uint_as_float(slot));
    const uint32_t h = ht[b].x;
    uint32_t t = __hip_atomic_load(&ht[b].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (t - h >= R.K) break;                                    // full: overflow queue
        const uint32_t seen = atomicCAS(&ht[b].y, t, t + 1u);
        if (seen == t) {
            slots[(size_t)b * R.K + (t & (R.K - 1))] = rec;
            where[slot] = t & ~kRingInOvf;
            return;
        }
        t = seen;
    }
    const uint32_t pos = atomicAdd(&st->ovf_tail, 1u);
    if (pos - __hip_atomic_load(&st->ovf_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > R.ovf_mask) {
        // cannot happen (see RingDesc).  If it does: no live record is clobbered -- the point stays reachable from the SoA arrays
        // only, marked unfiled so that its eviction touches nothing, and the status word makes the host refile the whole window
        // before the next answer is handed out (ring_host.inc: ring_overrun_repair)
        __hip_atomic_store(&R.status[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        where[slot] = kRingUnfiled;
        return;
    }
    ovf[pos & R.ovf_mask] = rec;
    where[slot] = (pos & ~kRingInOvf) | kRingInOvf;
}

// Pass 1 of an append: one thread per ring slot about to be overwritten retires the point the slot holds -- its record is
// marked dead where it was filed, then the owning bucket's head moves past every dead record in front (several threads may try:
// the compare-and-swap lets each step happen once; a bucket holds at most 32 records).  The overflow queue's head is moved once
// per launch, by the last block (below).  No record is filed while this runs.
__global__ __launch_bounds__(256) void ring_evict_kernel(RingDesc R, const float *__restrict__ x, const float *__restrict__ y,
                                                         const float *__restrict__ z, uint32_t slot0, uint32_t n,
                                                         uint2 *__restrict__ ht, float4 *__restrict__ slots, float4 *__restrict__ ovf,
                                                         const uint32_t *__restrict__ where, RingState *__restrict__ st)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // the overflow queue is either empty for the whole launch (nothing is filed while it runs) or needs its head moved afterwards
    const uint32_t q_tail = st->ovf_tail;
    const bool queue_in_use = q_tail != st->ovf_head;
    if (i < n) {
        const uint32_t slot = slot0 + i;
        const uint32_t w = where[slot];
        if (w == kRingUnfiled) {
            // never filed (queue overrun): nothing to retire
        } else if (w & kRingInOvf) {
            uint32_t *idw = reinterpret_cast<uint32_t *>(&ovf[w & R.ovf_mask].w);
            __hip_atomic_store(idw, kRingDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const uint32_t b = ring_bucket_of(R, x[slot], y[slot], z[slot]);
            float4 *base = slots + (size_t)b * R.K;
            __hip_atomic_store(reinterpret_cast<uint32_t *>(&base[w & (R.K - 1)].w), kRingDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t tail = ht[b].y;
            uint32_t h = __hip_atomic_load(&ht[b].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (h != tail) {
                const uint32_t *hw = reinterpret_cast<const uint32_t *>(&base[h & (R.K - 1)].w);
                if (__hip_atomic_load(hw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kRingDead) break;
                const uint32_t seen = atomicCAS(&ht[b].x, h, h + 1u);
                h = seen == h ? h + 1u : seen;
            }
        }
    }
    if (!queue_in_use) return;
    // The queue's head is moved by ONE block, the last to finish marking (a ticket per block): 256 entries per step, the dead
    // prefix measured with ballots.  (Every evicting thread advancing the head itself, as the buckets do, serialises tens of
    // thousands of compare-and-swaps on one word once the queue is long: 59 s for a 150 k-entry queue in the fuzz test.)
    __shared__ uint32_t s_last, s_head;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = atomicAdd(&st->ticket, 1u) == gridDim.x - 1;
        if (last) { atomicExch(&st->ticket, 0u); __threadfence(); }
        s_last = last ? 1u : 0u;
        s_head = __hip_atomic_load(&st->ovf_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    uint32_t h = s_head;
    for (;;) {
        const uint32_t pos = h + threadIdx.x;
        bool dead = false;
        if ((int32_t)(q_tail - pos) > 0)
            dead = __hip_atomic_load(reinterpret_cast<const uint32_t *>(&ovf[pos & R.ovf_mask].w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kRingDead;
        // length of the all-dead prefix of this 256-entry window: first thread that is not dead
        __shared__ uint32_t s_first;
        if (threadIdx.x == 0) s_first = 256u;
        __syncthreads();
        if (!dead) atomicMin(&s_first, threadIdx.x);
        __syncthreads();
        const uint32_t adv = s_first;
        __syncthreads();
        h += adv;
        if (adv < 256u) break;
    }
    if (threadIdx.x == 0) __hip_atomic_store(&st->ovf_head, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Pass 2: store the new frame in the SoA arrays and file it.  `src` = the frame, array of structures (x,y,z at byte offsets
// 0,4,8 of each stride-byte record) in device-visible memory.
__global__ __launch_bounds__(256) void ring_insert_kernel(RingDesc R, float *__restrict__ x, float *__restrict__ y, float *__restrict__ z,
                                                          const unsigned char *__restrict__ src, uint32_t stride, uint32_t n,
                                                          uint32_t slot0, uint2 *__restrict__ ht, float4 *__restrict__ slots,
                                                          float4 *__restrict__ ovf, uint32_t *__restrict__ where, RingState *__restrict__ st,
                                                          uint32_t new_count, ExpressSignal sig)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        if (new_count != 0xFFFFFFFFu) st->count = new_count;           // the append's last launch publishes the window's new size
        // overflow-queue length as this launch found it (a heuristic input for the host: an index whose cells are far too small or
        // too large for the data spills most points and is re-sized from the current contents, ring_host.inc)
        const uint32_t len = __hip_atomic_load(&st->ovf_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - __hip_atomic_load(&st->ovf_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&R.status[1], len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // (the frame may sit in host-mapped memory and be read over the bus: fetching the block's records with 16-byte loads into LDS
    // first was measured and changed nothing -- 600 KB take ~35 us either way, the bus's rate for a transfer this small)
    if (i < n) {
        const uint32_t slot = slot0 + i;
        const float *p = reinterpret_cast<const float *>(src + (size_t)i * stride);
        const float px = p[0], py = p[1], pz = p[2];
        x[slot] = px; y[slot] = py; z[slot] = pz;
        ring_file(R, px, py, pz, slot, ht, slots, ovf, where, st);
    }
    express_done_block(sig);       // the append's last launch carries the completion word the host polls (sig.seq == nullptr: nothing)
}

// file the points already in the SoA arrays (slots [slot0, slot0 + n)) after the tables have been cleared: the index of a
// freshly uploaded cloud, or a rebuild with another cell size
__global__ __launch_bounds__(256) void ring_refile_kernel(RingDesc R, const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ z, uint32_t slot0, uint32_t n,
                                                          uint2 *__restrict__ ht, float4 *__restrict__ slots, float4 *__restrict__ ovf,
                                                          uint32_t *__restrict__ where, RingState *__restrict__ st)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = slot0 + i;
    ring_file(R, x[slot], y[slot], z[slot], slot, ht, slots, ovf, where, st);
}

__global__ void ring_set_count_kernel(RingState *st, uint32_t count) { st->count = count; }

// ---- block-wide search ---------------------------------------------------------------------------------------------------
// Everything a query kernel needs to search one cloud: either index kind behind one handle.
struct RingView {
    RingDesc R;
    const uint2 *ht;
    const float4 *slots;
    const float4 *ovf;
    const RingState *st;
};

// the records of one bucket against the query, exact fp64, four loads in flight; the bucket is shared by `lanes` threads (a power of
// two), thread `part` of them takes records 4 part .. 4 part + 3, then 4 lanes further on
__device__ __forceinline__ void ring_scan_bucket(const RingView &V, uint32_t b, double qx, double qy, double qz, double &bd, uint32_t &bi,
                                                 uint32_t part = 0, uint32_t lanes = 1)
{
    const uint2 m = V.ht[b];
    const uint32_t n = m.y - m.x;
    const float4 *base = V.slots + (size_t)b * V.R.K;
    for (uint32_t j = 4u * part; j < n; j += 4u * lanes) {
        float4 P[4];
#pragma unroll
        for (int k = 0; k < 4; k++) P[k] = base[(m.x + min(j + (uint32_t)k, n - 1)) & (V.R.K - 1)];     // tail repeats the last record
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double d2 = dist2((double)P[k].x, (double)P[k].y, (double)P[k].z, qx, qy, qz);
            const uint32_t id = __float_as_uint(P[k].w);
            if (id != kRingDead && better(d2, id, bd, bi)) { bd = d2; bi = id; }
        }
    }
}

// threads that share one bucket of a step with `total` buckets: all 256 threads busy when the buckets are few; with grown buckets
// (R.K > 32: the window's cells hold hundreds of records) also in the larger shells, where one thread per bucket would walk a fat
// bucket alone while its neighbours have nothing to read
__device__ __forceinline__ int ring_lanes_per_bucket(const RingDesc &R, int total)
{
    if (R.K > kRingK) return total <= 32 ? 8 : total <= 512 ? 4 : total <= 2048 ? 2 : 1;
    return total <= 32 ? 8 : total <= 64 ? 4 : total <= 128 ? 2 : 1;
}

// Nearest point of the fp32-narrowed (px, py, pz) over a ring-indexed cloud; one 256-thread block, ONE THREAD PER BUCKET of the
// cube / shell being examined (a bucket holds ~6 records: a thread reads its head/tail pair, then its records, four at a time),
// block-wide fold after every shell.  stop_d2 as in block_nn_search (kernels.hpp): when only the radius is wanted the search may
// stop once everything unseen is beyond max_radius + search_margin.
//
// Exactness: after the cube of world cells [c - r, c + r]^3 has been visited (every bucket a cube cell folds onto), a point not
// yet seen lies in an unvisited bucket, so its world cell differs from every visited one along some axis that is still "open"
// (2r + 1 < table size on that axis), and along that axis it is beyond the cube's face: farther than the face distance.  The
// search stops only when best <= (min open face distance - h/256)^2, or when no axis is open (every bucket has been visited).
__device__ __forceinline__ void ring_block_nn_search(const RingView &V, double px, double py, double pz, double stop_d2,
                                                     double *s_d, uint32_t *s_i, double &bd, uint32_t &bi)
{
    const RingDesc &R = V.R;
    const float qxf = (float)px, qyf = (float)py, qzf = (float)pz;                    // searchPoint.x = search_Pt(0), :125-128
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    bd = __builtin_huge_val();
    bi = kNoIndex;
    {   // overflow queue: exhaustive
        const uint32_t oh = V.st->ovf_head, on = V.st->ovf_tail - oh;
        for (uint32_t k0 = threadIdx.x; k0 < on; k0 += 4u * 256u) {          // four records per thread in flight (a long queue is latency, not work)
            float4 P[4];
#pragma unroll
            for (int u = 0; u < 4; u++) P[u] = V.ovf[(oh + min(k0 + 256u * (uint32_t)u, on - 1u)) & R.ovf_mask];    // beyond the end: the last entry again
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double d2 = dist2((double)P[u].x, (double)P[u].y, (double)P[u].z, qx, qy, qz);
                const uint32_t id = __float_as_uint(P[u].w);
                if (id != kRingDead && better(d2, id, bd, bi)) { bd = d2; bi = id; }
            }
        }
    }
    bool wild = false;
    const int cx = ring_cell_coord(qx, R.inv_h, wild), cy = ring_cell_coord(qy, R.inv_h, wild), cz = ring_cell_coord(qz, R.inv_h, wild);
    for (int r = 1;; r++) {
        // an axis is open while the cube does not yet wrap around the table on it (a query with an absurd coordinate has none)
        const bool ox = !wild && 2 * r + 1 < R.gx, oy = !wild && 2 * r + 1 < R.gy, oz = !wild && 2 * r + 1 < R.gz;
        const bool was_x = !wild && 2 * r - 1 < R.gx, was_y = !wild && 2 * r - 1 < R.gy, was_z = !wild && 2 * r - 1 < R.gz;
        // positions per axis: the cube's 2r+1 world cells while open, the whole table once closed
        const int nx = ox ? 2 * r + 1 : R.gx, ny = oy ? 2 * r + 1 : R.gy, nz = oz ? 2 * r + 1 : R.gz;
        const int x0 = ox ? cx - r : 0, y0 = oy ? cy - r : 0, z0 = oz ? cz - r : 0;
        if (r == 1 || (!ox && was_x) || (!oy && was_y) || (!oz && was_z)) {
            // first cube, or an axis has just closed (or the query is wild): visit the whole box; buckets seen before are
            // merely seen again (a repeated (d2, slot) never changes the winner)
            // few buckets (the first cube has 27): several threads share a bucket -- one thread per bucket leaves 229 of the block's
            // threads idle while 27 walk their records four at a time, which is what a tick costs once the buckets are fat (surfaces,
            // re-sensed points: 40-250 records per occupied bucket)
            const int total = nx * ny * nz;
            const int lanes = ring_lanes_per_bucket(R, total);
            for (int k = (int)threadIdx.x / lanes; k < total; k += 256 / lanes) {
                const int jx = k % nx, jy = (k / nx) % ny, jz = k / (nx * ny);
                ring_scan_bucket(V, ring_lin(R, x0 + jx, y0 + jy, z0 + jz), qx, qy, qz, bd, bi, threadIdx.x % (uint32_t)lanes, (uint32_t)lanes);
            }
        } else {
            // shell r: two z-faces, then two y-faces without the z-face rows, then two x-faces without either
            const int nyi = oy ? ny - 2 : ny, nzi = oz ? nz - 2 : nz;
            const int A = oz ? 2 * nx * ny : 0, B = oy ? 2 * nx * nzi : 0, Cc = ox ? 2 * nyi * nzi : 0;
            const int lanes = ring_lanes_per_bucket(R, A + B + Cc);
            for (int k = (int)threadIdx.x / lanes; k < A + B + Cc; k += 256 / lanes) {
                int jx, jy, jz;
                if (k < A) {
                    const int f = k / (nx * ny), rem = k % (nx * ny);
                    jz = f ? nz - 1 : 0; jy = rem / nx; jx = rem % nx;
                } else if (k < A + B) {
                    const int k2 = k - A, f = k2 / (nx * nzi), rem = k2 % (nx * nzi);
                    jy = f ? ny - 1 : 0; jz = (oz ? 1 : 0) + rem / nx; jx = rem % nx;
                } else {
                    const int k3 = k - A - B, f = k3 / (nyi * nzi), rem = k3 % (nyi * nzi);
                    jx = f ? nx - 1 : 0; jz = (oz ? 1 : 0) + rem / nyi; jy = (oy ? 1 : 0) + rem % nyi;
                }
                ring_scan_bucket(V, ring_lin(R, x0 + jx, y0 + jy, z0 + jz), qx, qy, qz, bd, bi, threadIdx.x % (uint32_t)lanes, (uint32_t)lanes);
            }
        }
        block_argmin256(bd, bi, s_d, s_i);
        double bound = __builtin_huge_val();
        if (ox) bound = fmin(bound, fmin(qx - (double)(cx - r) * R.h, (double)(cx + r + 1) * R.h - qx));
        if (oy) bound = fmin(bound, fmin(qy - (double)(cy - r) * R.h, (double)(cy + r + 1) * R.h - qy));
        if (oz) bound = fmin(bound, fmin(qz - (double)(cz - r) * R.h, (double)(cz + r + 1) * R.h - qz));
        if (bound == __builtin_huge_val()) break;                   // every bucket has been visited
        bound -= R.h * (1.0 / 256.0);
        if (bound > 0.0 && (bd <= bound * bound || bound * bound >= stop_d2)) break;
    }
}

// ---- the fused planner batch ----------------------------------------------------------------------------------------------
// One launch answers everything a replan tick asks of the obstacle cloud (sim_planning_demo.cpp:159-178 -> :729-781, and
// SafeRegionEvaluate's re-check loop corridor_finder.cpp:829-835): a 256-thread block per planner point, three kinds of blocks:
//   [0, n_nodes)                      corridor nodes: radiusSearch of the node centre (checkRadius, :656-659)
//   [n_nodes, +n_samples)             Bezier samples: getPosFromBezier (:715-727) of the host-enumerated (segment, t), then
//                                     radiusSearch; collision <=> radius < 0 (checkTrajPtCol, corridor_finder.cpp:412-416)
//   [.., +n_ctrl)                     control points: the raw control point j of segment i scaled by T_i (the point the
//                                     optimizer's cone constraint keeps inside sphere i, traj_optimizer.cpp:624-648, in world
//                                     units as traj_optimizer.cpp:739-751 stores it), same threshold test -- SURVEY 3.3's
//                                     build extension for config C5
// Blocks beyond the counts in the argument block exit at once, so one captured grid serves every tick.
struct ReplanHeader {
    InflateParams P;
    int32_t n_nodes, n_samples, n_ctrl, nseg;
    int32_t row_stride, want_nn, first_seg;
    uint32_t seq;            // echoed into the summary once every result is in the caller's buffer (the host may poll for it)
    // offsets (in doubles / in uint32s) of the variable parts inside the argument block
    uint32_t off_nodes, off_coef, off_segtime, off_sample_t;       // in doubles from the start of the f64 area
    uint32_t off_orders, off_sample_seg, off_ctrl_seg, off_ctrl_j; // in uint32s from the start of the u32 area
};

struct ReplanSummary { long long first_hit_sample, first_hit_ctrl; int32_t n_nodes, n_samples, n_ctrl; uint32_t seq; };
// device words the blocks of one batch meet on: a ticket counter and the running first-hit minima (reset by the last block)
// `launches` counts the launches that have drained COMPLETELY (every block of the captured grid, participating or not): launch
// number L (1-based) reads header slot L & 1, and the host writes tick L's header into that slot.  The host is released by the last
// PARTICIPATING block, so it may fill tick L+1's header (the other slot) while trailing blocks of launch L are still starting; slot
// L & 1 is not written again before tick L+2, whose fill comes after tick L+1's answer, i.e. after launch L has drained (stream
// order).  A launch's header is therefore immutable while the launch is alive.
struct ReplanMeet { uint32_t ticket; int32_t first_sample, first_ctrl; uint32_t launches; uint32_t done; uint32_t pad[3]; };
constexpr size_t kReplanHdrSlot = 256;          // bytes per header slot in the argument block

template <bool RING>
__global__ __launch_bounds__(256) void replan_block_kernel(RingView V, GridDesc G0, const float4 *__restrict__ pts0,
                                                           const uint32_t *__restrict__ cs0, CoarseLevels C, int static_count,
                                                           const ReplanHeader *__restrict__ hdr, const double *__restrict__ f64a,
                                                           const uint32_t *__restrict__ u32a, uint32_t index_base,
                                                           ExpressOut *__restrict__ out, double *__restrict__ pos_out,
                                                           ReplanMeet *__restrict__ meet, ReplanSummary *__restrict__ sum)
{
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    __shared__ double s_term[3 * (kMaxBezierOrder + 1)];
    __shared__ double s_pos[3];
    __shared__ ReplanHeader s_H;
    // ONE thread fetches the header (host-mapped memory by default: a bus read) -- the slot of this launch's parity, see ReplanMeet
    if (threadIdx.x == 0) {
        const uint32_t launch = __hip_atomic_load(&meet->launches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        s_H = *reinterpret_cast<const ReplanHeader *>(reinterpret_cast<const unsigned char *>(hdr) + (launch & 1u) * kReplanHdrSlot);
    }
    __syncthreads();
    const ReplanHeader H = s_H;
    const int slot = (int)blockIdx.x;
    const int total = H.n_nodes + H.n_samples + H.n_ctrl;
    // every block of the grid, trailing ones included, signs off; the last one closes the launch (next launch = other header slot)
    auto sign_off = [&]() {
        if (threadIdx.x == 0 && atomicAdd(&meet->done, 1u) == gridDim.x - 1u) {
            atomicExch(&meet->done, 0u);
            atomicAdd(&meet->launches, 1u);
        }
    };
    if (slot >= total) { sign_off(); return; }
    const InflateParams P = H.P;
    double px, py, pz;
    if (slot < H.n_nodes) {
        const double *nd = f64a + H.off_nodes + 3 * (size_t)slot;
        px = nd[0]; py = nd[1]; pz = nd[2];
    } else if (slot < H.n_nodes + H.n_samples) {
        const int s = slot - H.n_nodes;
        const int seg = (int)u32a[H.off_sample_seg + s];
        const int order = (int)u32a[H.off_orders + seg], m = order + 1;
        const double T = f64a[H.off_segtime + seg];
        const double u = f64a[H.off_sample_t + s] / T;
        if ((int)threadIdx.x < 3 * m) {
            const int d = (int)threadIdx.x / m, j = (int)threadIdx.x % m;
            const double b = bernstein_binom(order, j);
            s_term[d * m + j] = b * f64a[H.off_coef + (size_t)seg * H.row_stride + d * m + j] * pow_uint_cr(u, j) * pow_uint_cr(1.0 - u, order - j);
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double acc = 0.0;
            for (int j = 0; j < m; j++) acc += s_term[threadIdx.x * m + j];
            s_pos[threadIdx.x] = acc * T;
        }
        __syncthreads();
        px = s_pos[0]; py = s_pos[1]; pz = s_pos[2];
    } else {
        const int k = slot - H.n_nodes - H.n_samples;
        const int seg = (int)u32a[H.off_ctrl_seg + k], j = (int)u32a[H.off_ctrl_j + k];
        const int m = (int)u32a[H.off_orders + seg] + 1;
        const double T = f64a[H.off_segtime + seg];
        const double *c = f64a + H.off_coef + (size_t)seg * H.row_stride;
        px = c[j] * T; py = c[m + j] * T; pz = c[2 * m + j] * T;
    }
    const double dx = px - P.sx, dy = py - P.sy, dz = pz - P.sz;
    const bool empty = RING ? V.st->count == 0 : static_count == 0;
    double bd = __builtin_huge_val(), radius = P.max_radius - P.search_margin;
    uint32_t bi = kNoIndex;
    if (!(empty || sqrt(dx * dx + dy * dy + dz * dz) > P.sample_range + P.max_radius)) {        // corridor_finder.cpp:115-116
        const double reach = P.max_radius + P.search_margin;
        const double stop_d2 = H.want_nn ? __builtin_huge_val() : reach * reach;
        if (RING) ring_block_nn_search(V, px, py, pz, stop_d2, s_d, s_i, bd, bi);
        else block_nn_search(G0, pts0, cs0, C, px, py, pz, stop_d2, s_d, s_i, bd, bi);
        const double rr = sqrt(bd) - P.search_margin;
        radius = rr < P.max_radius ? rr : P.max_radius;
    }
    // Thread 0 hands the block's result to the caller's (host-visible) buffer, joins the first-hit minima and takes a ticket; the
    // block holding the last ticket writes the summary and, last of all, the sequence word the host polls (system-scope release).
    // No second kernel: a separate one-block "finish" launch cost 8.9 us of a 23 us batch (profiles/r02_c5_kernel_stats.csv).
    if (threadIdx.x == 0) {
        out[slot].radius = radius;
        out[slot].idx = (bi == kNoIndex) ? kNoIndex : bi + index_base;
        out[slot].d2 = bd;
        out[slot].count = 0;
        if (slot >= H.n_nodes) { double *po = pos_out + 3 * (size_t)(slot - H.n_nodes); po[0] = px; po[1] = py; po[2] = pz; }
        if (radius < 0.0) {
            if (slot >= H.n_nodes && slot < H.n_nodes + H.n_samples) atomicMin(&meet->first_sample, slot - H.n_nodes);
            else if (slot >= H.n_nodes + H.n_samples) atomicMin(&meet->first_ctrl, slot - H.n_nodes - H.n_samples);
        }
        __threadfence_system();
        if (atomicAdd(&meet->ticket, 1u) == (uint32_t)total - 1u) {
            __threadfence_system();
            const int fs = atomicExch(&meet->first_sample, 0x7FFFFFFF), fc = atomicExch(&meet->first_ctrl, 0x7FFFFFFF);
            atomicExch(&meet->ticket, 0u);
            sum->first_hit_sample = fs == 0x7FFFFFFF ? -1ll : (long long)fs;
            sum->first_hit_ctrl = fc == 0x7FFFFFFF ? -1ll : (long long)fc;
            sum->n_nodes = H.n_nodes; sum->n_samples = H.n_samples; sum->n_ctrl = H.n_ctrl;
            __threadfence_system();
            __hip_atomic_store(&sum->seq, H.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    sign_off();
}

// a batch with no planner point at all still has to answer: summary only (one thread; returns at once otherwise).  It runs behind
// the block kernel on the stream, whose last block has already counted the launch: the header slot is the count's own parity.
__global__ void replan_empty_kernel(const ReplanHeader *__restrict__ hdr, const ReplanMeet *__restrict__ meet, ReplanSummary *__restrict__ sum)
{
    const uint32_t launch = __hip_atomic_load(&meet->launches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hdr = reinterpret_cast<const ReplanHeader *>(reinterpret_cast<const unsigned char *>(hdr) + (launch & 1u) * kReplanHdrSlot);
    if (hdr->n_nodes + hdr->n_samples + hdr->n_ctrl != 0) return;
    sum->first_hit_sample = -1; sum->first_hit_ctrl = -1;
    sum->n_nodes = 0; sum->n_samples = 0; sum->n_ctrl = 0;
    __threadfence_system();
    __hip_atomic_store(&sum->seq, hdr->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// plain NN / inflation of a device batch over a ring-indexed cloud: a block per query (the rolling map serves the planner's
// small batches; large throughput batches belong to the cell-sorted index)
template <bool INFLATE>
__global__ __launch_bounds__(256) void ring_batch_kernel(RingView V, InflateParams P, const float *__restrict__ qf, const double *__restrict__ q64,
                                                         double stop_d2, uint32_t index_base, uint32_t *__restrict__ out_idx,
                                                         double *__restrict__ out_d2, double *__restrict__ out_radius,
                                                         ExpressOut *__restrict__ out_rec, ExpressSignal sig)
{
    __shared__ double s_d[4];
    __shared__ uint32_t s_i[4];
    const size_t slot = blockIdx.x;
    double px, py, pz;
    if (q64) { px = q64[3 * slot]; py = q64[3 * slot + 1]; pz = q64[3 * slot + 2]; }
    else { px = (double)qf[3 * slot]; py = (double)qf[3 * slot + 1]; pz = (double)qf[3 * slot + 2]; }
    double bd = __builtin_huge_val(), radius = 0.0;
    uint32_t bi = kNoIndex;
    bool skip = V.st->count == 0;
    if (INFLATE) {
        const double dx = px - P.sx, dy = py - P.sy, dz = pz - P.sz;
        skip = skip || sqrt(dx * dx + dy * dy + dz * dz) > P.sample_range + P.max_radius;
        radius = P.max_radius - P.search_margin;
    }
    if (!skip) {
        ring_block_nn_search(V, px, py, pz, stop_d2, s_d, s_i, bd, bi);
        if (INFLATE) {
            const double rr = sqrt(bd) - P.search_margin;
            radius = rr < P.max_radius ? rr : P.max_radius;
        }
    }
    if (threadIdx.x == 0) {
        const uint32_t gi = (bi == kNoIndex) ? kNoIndex : bi + index_base;
        if (out_idx) out_idx[slot] = gi;
        if (out_d2) out_d2[slot] = bd;
        if (out_radius) out_radius[slot] = radius;
        if (out_rec) { out_rec[slot].idx = gi; out_rec[slot].d2 = bd; out_rec[slot].radius = radius; out_rec[slot].count = 0; }
        express_done(sig);
    }
}

}  // namespace pct
