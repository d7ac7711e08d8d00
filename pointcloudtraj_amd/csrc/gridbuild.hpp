// gridbuild.hpp -- building the uniform-cell index as a two-level counting sort on LDS histograms.
//
// What it replaces: cell_histogram_kernel + cell_scatter_kernel (kernels.hpp) issue one device-scope atomic per point and pass;
// scattered device atomics execute at the memory side at ~20 G/s chip-wide, so the 100 M-point cloud of config C4 spent
// 3.7 + 5.7 ms in those two kernels (profiles/r02_c4_kernel_stats.csv) where its 4 GB of traffic need under 1 ms.
//
// Level 1: the cell range is cut into <= 16384 slabs of 2^s1 consecutive cells (cell order: x fastest, then y, then z).  Every
//          block histograms a contiguous chunk of the cloud over the slabs in LDS (gb_hist_kernel: one global atomic per block and
//          non-empty slab), then reserves its share of every slab with one more atomic and moves its points there as
//          {x, y, z, index} records (gb_scatter_kernel).
// Level 2: one block per slab histograms the slab's records over its 2^s1 cells in LDS, scans, writes that piece of cell_start
//          and drops every record into its cell (gb_cells_kernel) -- all writes of a block land in the slab's own window.
// No device-scope atomic per point anywhere; the order of the records inside a cell is arbitrary, which every consumer tolerates
// (winners are picked by (d2, index), counts are counts).  The reference rebuilds its kd-tree from the cloud on every sensor frame
// (corridor_finder.cpp:93-99); this is that step for the cell index.
#pragma once

namespace pct {

constexpr int kGbThreads = 1024;
constexpr int kGbMaxSlabs = 16384;       // LDS histogram of level 1 (dynamic LDS: 4 bytes per slab, 64 KiB at most: two blocks per CU)
constexpr int kGbMaxSlabCells = 8192;    // LDS histogram of level 2 (32 KiB)
constexpr int kGbMaxBlocks = 512;

struct GbDesc {
    int s1;                // slab = cell >> s1
    uint32_t nslabs;       // <= kGbMaxSlabs
    uint32_t chunk;        // points per block of level 1 (multiple of 4)
};

// Self-check of a build, accumulated by the kernels that write the index (no extra pass) and compared on the host at the build's
// closing synchronise: the record ids must be a permutation of 0..n-1 (sum and xor of the ids are necessary conditions that a
// duplicated / lost / stale record breaks), no record may sit outside its slab, and the number of empty cells comes for free.
// A mismatch makes pct_cloud_build_grid FAIL (PCT_ERR_INTERNAL) instead of handing out an index that answers wrongly.
struct GbCheck { unsigned long long sum_ids; uint32_t xor_ids, misplaced, empty_cells, pad; };
constexpr int kGbCheckSlots = 64;      // blocks add into slot blockIdx & 63 (thousands of blocks on one word serialise: +36 us at 10 M points); the host sums

// zeroes the build's counters (a kernel of this library on the build's stream, not a runtime memset: one thing less between the
// launches that is not ours)
// kGridPad inert records behind the last cell-sorted one: the open-ended screening of the dense batch kernel (kernels.hpp
// coop_screen_rows<.., OPEN>) reads up to 15 records past the start of any run, i.e. past the end of the array for the last cells.
// +inf coordinates give a distance of +inf to every finite query: never a minimum, never inside a radius.
constexpr uint32_t kGridPad = 16;
__global__ void gb_pad_kernel(float4 *__restrict__ tail)
{
    if (threadIdx.x < kGridPad) tail[threadIdx.x] = make_float4(__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf(), __uint_as_float(0xFFFFFFFFu));
}

__global__ __launch_bounds__(256) void gb_zero_kernel(uint32_t *__restrict__ a, uint32_t n, GbCheck *__restrict__ chk)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = 0u;
    if (i < (uint32_t)kGbCheckSlots && chk) { chk[i].sum_ids = 0ull; chk[i].xor_ids = 0u; chk[i].misplaced = 0u; chk[i].empty_cells = 0u; chk[i].pad = 0u; }
}

// fold a thread's share of the check over the block (LDS), one set of global atomics per block
__device__ __forceinline__ void gb_check_commit(unsigned long long sum, uint32_t xr, uint32_t bad, uint32_t empty, GbCheck *__restrict__ chk)
{
    __shared__ unsigned long long s_sum;
    __shared__ uint32_t s_xor, s_bad, s_empty;
    if (threadIdx.x == 0) { s_sum = 0ull; s_xor = 0u; s_bad = 0u; s_empty = 0u; }
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += (unsigned long long)__shfl_xor((long long)sum, off, kWave);
        xr ^= (uint32_t)__shfl_xor((int)xr, off, kWave);
        bad += (uint32_t)__shfl_xor((int)bad, off, kWave);
        empty += (uint32_t)__shfl_xor((int)empty, off, kWave);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_sum, sum); atomicXor(&s_xor, xr);
        if (bad) atomicAdd(&s_bad, bad);
        if (empty) atomicAdd(&s_empty, empty);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        GbCheck *slot = chk + (blockIdx.x & (kGbCheckSlots - 1));
        atomicAdd(&slot->sum_ids, s_sum); atomicXor(&slot->xor_ids, s_xor);
        if (s_bad) atomicAdd(&slot->misplaced, s_bad);
        if (s_empty) atomicAdd(&slot->empty_cells, s_empty);
    }
}

// the same check over a finished index by a pass of its own (the per-point-atomic build of small clouds / very fine cells)
__global__ __launch_bounds__(256) void gb_check_kernel(GridDesc G, const float4 *__restrict__ sorted, const uint32_t *__restrict__ cell_start,
                                                       uint32_t n, GbCheck *__restrict__ chk);

__device__ __forceinline__ uint32_t gb_cell(const GridDesc &G, float px, float py, float pz)
{
    return cell_lin(G, cell_coord(px, G.ox, G.inv_h, G.gx), cell_coord(py, G.oy, G.inv_h, G.gy), cell_coord(pz, G.oz, G.inv_h, G.gz));
}

__global__ __launch_bounds__(256) void gb_check_kernel(GridDesc G, const float4 *__restrict__ sorted, const uint32_t *__restrict__ cell_start,
                                                       uint32_t n, GbCheck *__restrict__ chk)
{
    unsigned long long sum = 0;
    uint32_t xr = 0, bad = 0, empty = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const float4 P = sorted[p];
        const uint32_t id = __float_as_uint(P.w), cell = gb_cell(G, P.x, P.y, P.z);
        sum += id; xr ^= id;
        bad += (cell_start[cell] <= p && p < cell_start[cell + 1]) ? 0u : 1u;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < G.ncells; i += stride) empty += cell_start[i + 1] == cell_start[i] ? 1u : 0u;
    gb_check_commit(sum, xr, bad, empty, chk);
}

// Full structural verification of a built index AS THE KERNELS SEE IT (through the caches, on the stream that serves the queries):
// every id below n and seen exactly once (bitmap), every record inside the run of the cell its coordinates map to, cell_start a
// non-decreasing prefix from 0 to n.  out = {ids out of range, duplicated ids, misplaced records, decreasing cell_start steps,
// cell_start[0], cell_start[ncells]}.  Test hook (pct_debug_verify_grid): when a host read-back of the index and this disagree,
// the fault is in the read-back path, not in the index.
__global__ __launch_bounds__(256) void grid_verify_kernel(GridDesc G, const float4 *__restrict__ sorted, const uint32_t *__restrict__ cell_start,
                                                          uint32_t n, uint32_t *__restrict__ bitmap, unsigned long long *__restrict__ out)
{
    uint32_t bad_id = 0, dup = 0, misplaced = 0, decreasing = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const float4 P = sorted[p];
        const uint32_t id = __float_as_uint(P.w), cell = gb_cell(G, P.x, P.y, P.z);
        if (id >= n) bad_id++;
        else if (atomicOr(&bitmap[id >> 5], 1u << (id & 31u)) & (1u << (id & 31u))) dup++;
        misplaced += (cell_start[cell] <= p && p < cell_start[cell + 1]) ? 0u : 1u;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < G.ncells; i += stride) decreasing += cell_start[i + 1] < cell_start[i] ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bad_id += (uint32_t)__shfl_xor((int)bad_id, off, kWave); dup += (uint32_t)__shfl_xor((int)dup, off, kWave);
        misplaced += (uint32_t)__shfl_xor((int)misplaced, off, kWave); decreasing += (uint32_t)__shfl_xor((int)decreasing, off, kWave);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad_id) atomicAdd(&out[0], (unsigned long long)bad_id);
        if (dup) atomicAdd(&out[1], (unsigned long long)dup);
        if (misplaced) atomicAdd(&out[2], (unsigned long long)misplaced);
        if (decreasing) atomicAdd(&out[3], (unsigned long long)decreasing);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[4] = cell_start[0]; out[5] = cell_start[G.ncells]; }
}

// level 1, pass A: table[block][slab] = points of the block's chunk in the slab; slab_total[slab] += the same
__global__ __launch_bounds__(kGbThreads) void gb_hist_kernel(GridDesc G, GbDesc D, const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, uint32_t n, uint32_t *__restrict__ table,
                                                             uint32_t *__restrict__ slab_total)
{
    extern __shared__ uint32_t h[];                                  // nslabs counters
    for (uint32_t i = threadIdx.x; i < D.nslabs; i += kGbThreads) h[i] = 0;
    __syncthreads();
    const uint32_t b0 = blockIdx.x * D.chunk, b1 = min(n, b0 + D.chunk);
    const uint32_t full = b0 + ((b1 - b0) & ~3u);                    // whole groups of 4 points: one 16-byte load per array
    for (uint32_t i = b0 + 4 * threadIdx.x; i < full; i += 4 * kGbThreads) {
        const float4 X = *reinterpret_cast<const float4 *>(x + i), Y = *reinterpret_cast<const float4 *>(y + i),
                     Z = *reinterpret_cast<const float4 *>(z + i);
        atomicAdd(&h[gb_cell(G, X.x, Y.x, Z.x) >> D.s1], 1u);
        atomicAdd(&h[gb_cell(G, X.y, Y.y, Z.y) >> D.s1], 1u);
        atomicAdd(&h[gb_cell(G, X.z, Y.z, Z.z) >> D.s1], 1u);
        atomicAdd(&h[gb_cell(G, X.w, Y.w, Z.w) >> D.s1], 1u);
    }
    for (uint32_t i = full + threadIdx.x; i < b1; i += kGbThreads) atomicAdd(&h[gb_cell(G, x[i], y[i], z[i]) >> D.s1], 1u);
    __syncthreads();
    uint32_t *row = table + (size_t)blockIdx.x * D.nslabs;
    for (uint32_t i = threadIdx.x; i < D.nslabs; i += kGbThreads) {
        const uint32_t v = h[i];
        row[i] = v;
        if (v) atomicAdd(&slab_total[i], v);
    }
}

// exclusive scan of a[0..n) in LDS by the whole block, n <= blockDim.x * 16 handled in rounds with a carry; a[n] = total
template <int THREADS>
__device__ __forceinline__ void gb_block_scan(uint32_t *a, uint32_t n, uint32_t *s_wave, uint32_t *s_carry)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) *s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? a[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off, kWave);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wave_off = *s_carry;
        for (int w = 0; w < wave; w++) wave_off += s_wave[w];
        if (i < n) a[i] = wave_off + inc - v;
        __syncthreads();
        if (threadIdx.x == THREADS - 1) *s_carry = wave_off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) a[n] = *s_carry;
    __syncthreads();
}

// level 1, pass B: every block scans the slab totals (4096 values: cheaper than a launch of its own), reserves its share of every
// slab it has points for, and moves its chunk.  Block 0 also publishes slab_start for level 2.
__global__ __launch_bounds__(kGbThreads) void gb_scatter_kernel(GridDesc G, GbDesc D, const float *__restrict__ x, const float *__restrict__ y,
                                                                const float *__restrict__ z, uint32_t n, const uint32_t *__restrict__ table,
                                                                const uint32_t *__restrict__ slab_total, uint32_t *__restrict__ slab_cursor,
                                                                uint32_t *__restrict__ slab_start, float4 *__restrict__ tmp)
{
    extern __shared__ uint32_t base[];                               // nslabs + 1 entries
    __shared__ uint32_t s_wave[kGbThreads / 64], s_carry;
    for (uint32_t i = threadIdx.x; i < D.nslabs; i += kGbThreads) base[i] = slab_total[i];
    __syncthreads();
    gb_block_scan<kGbThreads>(base, D.nslabs, s_wave, &s_carry);
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i <= D.nslabs; i += kGbThreads) slab_start[i] = base[i];
    const uint32_t *row = table + (size_t)blockIdx.x * D.nslabs;
    for (uint32_t i = threadIdx.x; i < D.nslabs; i += kGbThreads) {
        const uint32_t v = row[i];
        if (v) base[i] += atomicAdd(&slab_cursor[i], v);             // this block's first slot in slab i
    }
    __syncthreads();
    const uint32_t b0 = blockIdx.x * D.chunk, b1 = min(n, b0 + D.chunk);
    const uint32_t full = b0 + ((b1 - b0) & ~3u);
    for (uint32_t i = b0 + 4 * threadIdx.x; i < full; i += 4 * kGbThreads) {
        const float4 X = *reinterpret_cast<const float4 *>(x + i), Y = *reinterpret_cast<const float4 *>(y + i),
                     Z = *reinterpret_cast<const float4 *>(z + i);
        tmp[atomicAdd(&base[gb_cell(G, X.x, Y.x, Z.x) >> D.s1], 1u)] = make_float4(X.x, Y.x, Z.x, __uint_as_float(i));
        tmp[atomicAdd(&base[gb_cell(G, X.y, Y.y, Z.y) >> D.s1], 1u)] = make_float4(X.y, Y.y, Z.y, __uint_as_float(i + 1));
        tmp[atomicAdd(&base[gb_cell(G, X.z, Y.z, Z.z) >> D.s1], 1u)] = make_float4(X.z, Y.z, Z.z, __uint_as_float(i + 2));
        tmp[atomicAdd(&base[gb_cell(G, X.w, Y.w, Z.w) >> D.s1], 1u)] = make_float4(X.w, Y.w, Z.w, __uint_as_float(i + 3));
    }
    for (uint32_t i = full + threadIdx.x; i < b1; i += kGbThreads) {
        const float px = x[i], py = y[i], pz = z[i];
        tmp[atomicAdd(&base[gb_cell(G, px, py, pz) >> D.s1], 1u)] = make_float4(px, py, pz, __uint_as_float(i));
    }
}

// ---- level 1 in TWO passes for large clouds --------------------------------------------------------------------------------------
// One pass into thousands of slabs keeps thousands of 16-byte write streams open per block: the lines leave L2 half filled
// (WRITE_SIZE 2x the bytes, 1.7 TB/s of fabric traffic at 100 M points).  With a fan-out of at most 128 per pass the open lines of
// all blocks fit the L2s and every line is written once: pass A = the kernels above with slab := SUPER-slab (2^sb consecutive slabs),
// records into the (still unused) final array; pass B below = the same thing inside every super-slab's region, into `tmp`, in the
// slab order level 2 expects.  Block (x, S) of pass B handles chunk x of super-slab S's region.  Measured: 100 M points 4.29 -> 3.95 ms
// (pass A 0.21 + 0.99, pass B 0.31 + 1.16 ms against 0.23 + 2.7 ms for the single pass), 10 M points the same either way (0.39 ms:
// 22 + 58 + 27 + 97 us against 24 + 183) -- used from 4096 slabs (~16 M points) upwards.
struct Gb2Desc {
    int s1, sb;            // slab = cell >> s1, super-slab = slab >> sb
    uint32_t nslabs, nsuper, parts;    // parts = blocks per super-slab
};

__global__ __launch_bounds__(kGbThreads) void gb_hist2_kernel(GridDesc G, Gb2Desc D, const uint32_t *__restrict__ super_start,
                                                              const float4 *__restrict__ recs, uint32_t *__restrict__ table2,
                                                              uint32_t *__restrict__ slab_total)
{
    __shared__ uint32_t h[128];
    const uint32_t S = blockIdx.y, nsub = 1u << D.sb;
    if (threadIdx.x < nsub) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t r0 = super_start[S], r1 = super_start[S + 1];
    const uint32_t chunk = (r1 - r0 + D.parts - 1) / D.parts;
    const uint32_t b0 = min(r1, r0 + blockIdx.x * chunk), b1 = min(r1, b0 + chunk);
    for (uint32_t i = b0 + threadIdx.x; i < b1; i += kGbThreads) {   // (four records in flight per thread measured the same)
        const float4 P = recs[i];
        atomicAdd(&h[(gb_cell(G, P.x, P.y, P.z) >> D.s1) - (S << D.sb)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < nsub) {
        const uint32_t v = h[threadIdx.x];
        table2[((size_t)S * D.parts + blockIdx.x) * nsub + threadIdx.x] = v;
        const uint32_t slab = (S << D.sb) + threadIdx.x;
        if (v && slab < D.nslabs) atomicAdd(&slab_total[slab], v);
    }
}

__global__ __launch_bounds__(kGbThreads) void gb_scatter2_kernel(GridDesc G, Gb2Desc D, const uint32_t *__restrict__ super_start,
                                                                 const float4 *__restrict__ recs, const uint32_t *__restrict__ table2,
                                                                 const uint32_t *__restrict__ slab_total, uint32_t *__restrict__ slab_cursor,
                                                                 uint32_t *__restrict__ slab_start, uint32_t n, float4 *__restrict__ tmp)
{
    __shared__ uint32_t base[128 + 1];
    __shared__ uint32_t s_wave[kGbThreads / 64], s_carry;
    const uint32_t S = blockIdx.y, nsub = 1u << D.sb;
    const uint32_t r0 = super_start[S], r1 = super_start[S + 1];
    if (threadIdx.x < nsub) {
        const uint32_t slab = (S << D.sb) + threadIdx.x;
        base[threadIdx.x] = slab < D.nslabs ? slab_total[slab] : 0u;
    }
    __syncthreads();
    gb_block_scan<kGbThreads>(base, nsub, s_wave, &s_carry);         // exclusive scan of this super-slab's slab sizes
    if (blockIdx.x == 0) {                                           // publish the slabs' global starts for level 2
        if (threadIdx.x < nsub) {
            const uint32_t slab = (S << D.sb) + threadIdx.x;
            if (slab < D.nslabs) slab_start[slab] = r0 + base[threadIdx.x];
        }
        if (S == D.nsuper - 1 && threadIdx.x == 0) slab_start[D.nslabs] = n;
    }
    if (threadIdx.x < nsub) {
        const uint32_t slab = (S << D.sb) + threadIdx.x;
        const uint32_t v = table2[((size_t)S * D.parts + blockIdx.x) * nsub + threadIdx.x];
        base[threadIdx.x] += r0 + ((v && slab < D.nslabs) ? atomicAdd(&slab_cursor[slab], v) : 0u);   // this block's first slot in the slab
    }
    __syncthreads();
    const uint32_t chunk = (r1 - r0 + D.parts - 1) / D.parts;
    const uint32_t b0 = min(r1, r0 + blockIdx.x * chunk), b1 = min(r1, b0 + chunk);
    for (uint32_t i = b0 + threadIdx.x; i < b1; i += kGbThreads) {
        const float4 P = recs[i];
        tmp[atomicAdd(&base[(gb_cell(G, P.x, P.y, P.z) >> D.s1) - (S << D.sb)], 1u)] = P;
    }
}

// level 2: block b owns slab b = cells [b << s1, min(ncells, (b + 1) << s1)) = records tmp[slab_start[b], slab_start[b + 1]).
// A slab of at most stage_cap records (the normal case: the host sizes the slabs for ~1-6 k points) is read ONCE into registers,
// ranked through the LDS cell counters, placed in LDS at its sorted position and written out as one contiguous run -- whole
// lines, no second read.  A larger slab (clustered clouds) streams through twice and scatters inside its own window.
// Dynamic LDS: (2^s1 + 1) counters, padded to 16 bytes, then stage_cap float4 records.
constexpr int kGbStagePerThread = 8;                                 // stage_cap <= 8 records per thread
template <int kGbCellThreads>
__global__ __launch_bounds__(kGbCellThreads) void gb_cells_kernel(GridDesc G, GbDesc D, const uint32_t *__restrict__ slab_start,
                                                                  const float4 *__restrict__ tmp, uint32_t n, uint32_t stage_cap,
                                                                  uint32_t *__restrict__ cell_start, float4 *__restrict__ sorted,
                                                                  GbCheck *__restrict__ chk)
{
    unsigned long long k_sum = 0;                                    // self-check (GbCheck), accumulated from the records in hand
    uint32_t k_xor = 0, k_bad = 0, k_empty = 0;
    extern __shared__ uint32_t cnt[];
    __shared__ uint32_t s_wave[kGbCellThreads / 64], s_carry;
    float4 *stage = reinterpret_cast<float4 *>(cnt + ((((1u << D.s1) + 1) + 3) & ~3u));
    const uint32_t b = blockIdx.x;
    const uint32_t c0 = b << D.s1, c1 = min(G.ncells, c0 + (1u << D.s1)), m = c1 - c0;
    const uint32_t p0 = slab_start[b], p1 = slab_start[b + 1], np = p1 - p0;
    for (uint32_t i = threadIdx.x; i < m; i += kGbCellThreads) cnt[i] = 0;
    __syncthreads();
    if (np <= stage_cap) {
        float4 R[kGbStagePerThread];
        uint32_t cell[kGbStagePerThread];
#pragma unroll
        for (int k = 0; k < kGbStagePerThread; k++) {
            const uint32_t i = threadIdx.x + k * kGbCellThreads;
            if (i < np) {
                R[k] = tmp[p0 + i];
                cell[k] = gb_cell(G, R[k].x, R[k].y, R[k].z) - c0;
                const uint32_t id = __float_as_uint(R[k].w);
                k_sum += id; k_xor ^= id;
                if (cell[k] >= m) { k_bad++; cell[k] = 0; }               // a record that is not this slab's (cannot happen): counted, kept in range
                atomicAdd(&cnt[cell[k]], 1u);
            }
        }
        __syncthreads();
        gb_block_scan<kGbCellThreads>(cnt, m, s_wave, &s_carry);
        for (uint32_t i = threadIdx.x; i < m; i += kGbCellThreads) {
            cell_start[c0 + i] = p0 + cnt[i];
            k_empty += cnt[i + 1] == cnt[i] ? 1u : 0u;
        }
        if (c1 == G.ncells && threadIdx.x == 0) cell_start[G.ncells] = n;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kGbStagePerThread; k++) {
            const uint32_t i = threadIdx.x + k * kGbCellThreads;
            if (i < np) stage[atomicAdd(&cnt[cell[k]], 1u)] = R[k];
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < np; i += kGbCellThreads) sorted[p0 + i] = stage[i];
        gb_check_commit(k_sum, k_xor, k_bad, k_empty, chk);
        return;
    }
    for (uint32_t i = p0 + threadIdx.x; i < p1; i += kGbCellThreads) {
        const float4 P = tmp[i];
        uint32_t cl = gb_cell(G, P.x, P.y, P.z) - c0;
        const uint32_t id = __float_as_uint(P.w);
        k_sum += id; k_xor ^= id;
        if (cl >= m) { k_bad++; cl = 0; }
        atomicAdd(&cnt[cl], 1u);
    }
    __syncthreads();
    gb_block_scan<kGbCellThreads>(cnt, m, s_wave, &s_carry);
    for (uint32_t i = threadIdx.x; i < m; i += kGbCellThreads) {
        cell_start[c0 + i] = p0 + cnt[i];
        k_empty += cnt[i + 1] == cnt[i] ? 1u : 0u;
    }
    if (c1 == G.ncells && threadIdx.x == 0) cell_start[G.ncells] = n;
    __syncthreads();
    for (uint32_t i = p0 + threadIdx.x; i < p1; i += kGbCellThreads) {
        const float4 P = tmp[i];
        uint32_t cl = gb_cell(G, P.x, P.y, P.z) - c0;
        if (cl >= m) cl = 0;
        sorted[p0 + atomicAdd(&cnt[cl], 1u)] = P;
    }
    gb_check_commit(k_sum, k_xor, k_bad, k_empty, chk);
}

}  // namespace pct
