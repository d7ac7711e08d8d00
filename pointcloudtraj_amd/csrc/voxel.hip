// voxel.hip -- voxel de-duplication on the GPU (include/pct_voxel.h), part of libpct_engine.so.
//
// Restates voxel_map / voxel_value_map (reference Planner/src/voxel_map.cpp:5-76): a set of integer voxel coordinates
// plus the list of voxel centres in first-seen order.  The sequential container decides "first seen" by input order; a
// batch on the GPU decides it the same way with one atomicMin per point:
//
//   1. vox_insert_kernel    key = pack(round(x/res), round(y/res), round(z/res)); claim / find the key's slot in an
//                           open-addressing table (64-bit CAS, linear probing); atomicMin(slot.val, base + i).
//                           Voxels of earlier batches hold their final id (< base), so they are unaffected; a voxel new in
//                           this batch ends up holding base + (lowest input index that maps to it).
//   2. vox_rank_kernel      is_first[i] = (slot.val == base + i); exclusive rank of the flags inside each 1024-point
//                           tile, tile totals.
//   3. vox_tile_scan_kernel exclusive scan of the tile totals (one block), grand total = voxels added.
//   4. vox_commit_kernel    first occurrences write their voxel (id = base + tile offset + rank: input order) -- integer
//                           coordinates, float centres -- and replace the slot's value by the id;
//   5. vox_report_kernel    per-point outputs (is_new, voxel index), only when asked for.
//
// HBM layout per map: table keys u64[T] + vals u32[T] (T = power of two >= 2 x voxels), voxel store ix/iy/iz i32[V]
// (SoA) and x/y/z f32[V] (SoA, ready for pct_cloud_upload_soa_dev), per-batch scratch slot u32[n], rank u32[n],
// tile u32[n/1024].  Traffic per input point: 12-24 B in, ~12 B of table atomics, 8 B scratch; per new voxel 24 B out.
// The kernel is bound by scattered device-scope atomics (~2e10 /s chip-wide), not by HBM bandwidth.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/pct_voxel.h"
#include "engine_internal.hpp"

using pct_internal::fail;

namespace {

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr int kVoxBias = 1 << 20;                 // voxel coordinates in [-2^20, 2^20)
constexpr int kTile = 1024;                       // points per rank tile (256 threads x 4)
constexpr int kWave = 64;

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

struct VoxFlags { uint32_t out_of_range, table_full, total; };

__device__ __forceinline__ uint32_t vox_hash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;   // murmur3 finaliser
    return (uint32_t)k;
}

// voxel_map.cpp:5-16: (int) round(coordinate / res), fp64 division on the (widened) coordinate
template <typename T>
__device__ __forceinline__ bool vox_coords(const unsigned char *rec, double res, int &ix, int &iy, int &iz)
{
    const T *p = reinterpret_cast<const T *>(rec);
    const double rx = round((double)p[0] / res), ry = round((double)p[1] / res), rz = round((double)p[2] / res);
    const double lim = (double)kVoxBias;
    if (!(rx >= -lim && rx < lim && ry >= -lim && ry < lim && rz >= -lim && rz < lim)) return false;   // also NaN
    ix = (int)rx; iy = (int)ry; iz = (int)rz;
    return true;
}

__device__ __forceinline__ unsigned long long vox_pack(int ix, int iy, int iz)
{
    return (unsigned long long)(uint32_t)(ix + kVoxBias) | ((unsigned long long)(uint32_t)(iy + kVoxBias) << 21) |
           ((unsigned long long)(uint32_t)(iz + kVoxBias) << 42);
}

__global__ __launch_bounds__(256) void vox_table_init_kernel(unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t T)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += stride) { keys[i] = kEmptyKey; vals[i] = 0xFFFFFFFFu; }
}

__device__ __forceinline__ uint32_t vox_find_or_claim(unsigned long long *__restrict__ keys, uint32_t mask, unsigned long long key)
{
    uint32_t slot = vox_hash(key) & mask;
    for (uint32_t probes = 0; probes <= mask; probes++) {
        const unsigned long long seen = keys[slot];                       // most points hit an existing voxel: plain read first
        if (seen == key) return slot;
        if (seen == kEmptyKey) {
            const unsigned long long old = atomicCAS(&keys[slot], kEmptyKey, key);
            if (old == kEmptyKey || old == key) return slot;
        }
        slot = (slot + 1) & mask;
    }
    return kNoSlot;
}

template <typename T>
__global__ __launch_bounds__(256) void vox_insert_kernel(const unsigned char *__restrict__ pts, uint32_t n, uint32_t stride_bytes,
                                                         double res, uint32_t base, unsigned long long *__restrict__ keys,
                                                         uint32_t *__restrict__ vals, uint32_t mask,
                                                         uint32_t *__restrict__ pslot, VoxFlags *__restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int ix, iy, iz;
    uint32_t slot = kNoSlot;
    if (vox_coords<T>(pts + (size_t)i * stride_bytes, res, ix, iy, iz)) {
        slot = vox_find_or_claim(keys, mask, vox_pack(ix, iy, iz));
        if (slot == kNoSlot) flags->table_full = 1;
        // a voxel of an earlier batch holds its final id (< base, written before this launch): nothing to decide, and the
        // re-observation of a known map (the common case for accumulating sensors) then costs no atomic at all
        else if (vals[slot] >= base) atomicMin(&vals[slot], base + i);
    } else {
        flags->out_of_range = 1;
    }
    pslot[i] = slot;
}

// re-insert the voxels [0, n) of the store into a fresh (larger) table: val = id
__global__ __launch_bounds__(256) void vox_rehash_kernel(const int *__restrict__ vx, const int *__restrict__ vy, const int *__restrict__ vz,
                                                         uint32_t n, unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals,
                                                         uint32_t mask, VoxFlags *__restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = vox_find_or_claim(keys, mask, vox_pack(vx[i], vy[i], vz[i]));
    if (slot != kNoSlot) vals[slot] = i;
    else flags->table_full = 1;
}

__global__ __launch_bounds__(256) void vox_rank_kernel(const uint32_t *__restrict__ pslot, const uint32_t *__restrict__ vals, uint32_t n,
                                                       uint32_t base, uint32_t *__restrict__ rank, uint32_t *__restrict__ tile_sum)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t first = blockIdx.x * kTile + threadIdx.x * 4;
    uint32_t f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = first + k;
        const uint32_t slot = i < n ? pslot[i] : kNoSlot;
        f[k] = (slot != kNoSlot && vals[slot] == base + i) ? 1u : 0u;
    }
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
    uint32_t run = wave_off + inc - tsum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // rank of a first occurrence inside its tile; bit 31 marks "is a first occurrence"
        if (first + k < n) rank[first + k] = run | (f[k] << 31);
        run += f[k];
    }
    if (threadIdx.x == 255) tile_sum[blockIdx.x] = wave_off + inc;
}

// one block: exclusive scan of the tile totals in place, grand total -> flags->total
__global__ __launch_bounds__(256) void vox_tile_scan_kernel(uint32_t *__restrict__ tile_sum, uint32_t ntiles, VoxFlags *__restrict__ flags)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t b = 0; b < ntiles; b += 256) {
        const uint32_t i = b + threadIdx.x;
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
    if (threadIdx.x == 0) flags->total = s_carry;
}

template <typename T>
__global__ __launch_bounds__(256) void vox_commit_kernel(const unsigned char *__restrict__ pts, uint32_t n, uint32_t stride_bytes, double res,
                                                         uint32_t base, const uint32_t *__restrict__ pslot, const uint32_t *__restrict__ rank,
                                                         const uint32_t *__restrict__ tile_off, uint32_t *__restrict__ vals,
                                                         int *__restrict__ vx, int *__restrict__ vy, int *__restrict__ vz,
                                                         float *__restrict__ fx, float *__restrict__ fy, float *__restrict__ fz)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = rank[i];
    if (!(r >> 31)) return;
    const uint32_t id = base + tile_off[i / kTile] + (r & 0x7FFFFFFFu);
    int ix, iy, iz;
    vox_coords<T>(pts + (size_t)i * stride_bytes, res, ix, iy, iz);
    vals[pslot[i]] = id;
    vx[id] = ix; vy[id] = iy; vz[id] = iz;
    // voxel_map.cpp:31 -- emplace_back(x * res, y * res, z * res): int -> double product, narrowed by PointXYZ's float fields
    fx[id] = (float)((double)ix * res); fy[id] = (float)((double)iy * res); fz[id] = (float)((double)iz * res);
}

__global__ __launch_bounds__(256) void vox_report_kernel(const uint32_t *__restrict__ pslot, const uint32_t *__restrict__ rank,
                                                         const uint32_t *__restrict__ vals, uint32_t n, uint8_t *__restrict__ is_new,
                                                         int32_t *__restrict__ voxel_index)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (is_new) is_new[i] = (uint8_t)(rank[i] >> 31);
    if (voxel_index) voxel_index[i] = pslot[i] == kNoSlot ? -1 : (int32_t)vals[pslot[i]];
}

template <typename T>
int dev_alloc(T **p, size_t count)
{
    void *v = nullptr;
    hipError_t e = hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return fail(PCT_ERR_ALLOC, "hipMalloc(%zu bytes) -> %s", count * sizeof(T), hipGetErrorString(e));
    *p = static_cast<T *>(v);
    return PCT_OK;
}

template <typename T>
void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

inline int blocks_for(int64_t n, int per) { return (int)std::max<int64_t>(1, (n + per - 1) / per); }

}  // namespace

struct pct_voxel_map {
    double res = 0.0;
    int64_t size = 0;                 // voxels
    // hash table
    uint32_t T = 0;
    unsigned long long *keys = nullptr;
    uint32_t *vals = nullptr;
    // voxel store
    int64_t vcap = 0;
    int *vx = nullptr, *vy = nullptr, *vz = nullptr;
    float *fx = nullptr, *fy = nullptr, *fz = nullptr;
    // per-batch scratch
    int64_t ncap = 0;
    uint32_t *pslot = nullptr, *rank = nullptr, *tile = nullptr;
    unsigned char *stage = nullptr;
    size_t stage_bytes = 0;
    uint8_t *d_is_new = nullptr;
    int32_t *d_index = nullptr;
    VoxFlags *d_flags = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
};

namespace {

int table_alloc(pct_voxel_map *m, uint32_t T)
{
    dev_free(m->keys); dev_free(m->vals);
    m->T = 0;
    PCTCHK(dev_alloc(&m->keys, T));
    PCTCHK(dev_alloc(&m->vals, T));
    m->T = T;
    vox_table_init_kernel<<<std::min(blocks_for(T, 256), 4096), 256, 0, pct_internal::stream()>>>(m->keys, m->vals, T);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

uint32_t table_size_for(int64_t voxels)
{
    uint64_t T = 1024;
    while (T < (uint64_t)voxels * 2) T <<= 1;     // load factor <= 0.5
    return (uint32_t)std::min<uint64_t>(T, 1ull << 31);
}

// room for `extra` more voxels: grow the store (copy) and the table (rehash) when needed
int ensure_capacity(pct_voxel_map *m, int64_t extra)
{
    hipStream_t s = pct_internal::stream();
    const int64_t need = m->size + extra;
    if (need > (int64_t)0x7FFFFFFF) return fail(PCT_ERR_INVALID, "voxel map would exceed 2^31 voxels");
    if (need > m->vcap) {
        const int64_t cap = std::max<int64_t>(need, m->vcap * 2);
        int *nx = nullptr, *ny = nullptr, *nz = nullptr;
        float *gx = nullptr, *gy = nullptr, *gz = nullptr;
        PCTCHK(dev_alloc(&nx, cap)); PCTCHK(dev_alloc(&ny, cap)); PCTCHK(dev_alloc(&nz, cap));
        PCTCHK(dev_alloc(&gx, cap)); PCTCHK(dev_alloc(&gy, cap)); PCTCHK(dev_alloc(&gz, cap));
        if (m->size) {
            const size_t bi = sizeof(int) * m->size, bf = sizeof(float) * m->size;
            HIPCHK(hipMemcpyAsync(nx, m->vx, bi, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(ny, m->vy, bi, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(nz, m->vz, bi, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(gx, m->fx, bf, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(gy, m->fy, bf, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(gz, m->fz, bf, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
        }
        dev_free(m->vx); dev_free(m->vy); dev_free(m->vz); dev_free(m->fx); dev_free(m->fy); dev_free(m->fz);
        m->vx = nx; m->vy = ny; m->vz = nz; m->fx = gx; m->fy = gy; m->fz = gz;
        m->vcap = cap;
    }
    if ((uint64_t)need * 2 > m->T) {
        PCTCHK(table_alloc(m, table_size_for(need)));
        if (m->size) {
            vox_rehash_kernel<<<blocks_for(m->size, 256), 256, 0, s>>>(m->vx, m->vy, m->vz, (uint32_t)m->size, m->keys, m->vals, m->T - 1, m->d_flags);
            HIPCHK(hipGetLastError());
        }
    }
    return PCT_OK;
}

int ensure_scratch(pct_voxel_map *m, int64_t n)
{
    if (n <= m->ncap) return PCT_OK;
    const int64_t cap = std::max<int64_t>(n, m->ncap * 2);
    dev_free(m->pslot); dev_free(m->rank); dev_free(m->tile); dev_free(m->d_is_new); dev_free(m->d_index);
    m->ncap = 0;
    PCTCHK(dev_alloc(&m->pslot, cap));
    PCTCHK(dev_alloc(&m->rank, cap));
    PCTCHK(dev_alloc(&m->tile, (cap + kTile - 1) / kTile));
    m->ncap = cap;
    return PCT_OK;
}

template <typename T>
int add_device(pct_voxel_map *m, const unsigned char *d_pts, int64_t n, int64_t stride, int64_t *n_new, uint8_t *d_is_new, int32_t *d_index)
{
    hipStream_t s = pct_internal::stream();
    HIPCHK(hipMemsetAsync(m->d_flags, 0, sizeof(VoxFlags), s));
    PCTCHK(ensure_capacity(m, n));            // every point could open a voxel
    PCTCHK(ensure_scratch(m, n));
    const uint32_t base = (uint32_t)m->size, un = (uint32_t)n;
    const int nb = blocks_for(n, 256), ntiles = blocks_for(n, kTile);
    HIPCHK(hipEventRecord(m->ev0, s));
    vox_insert_kernel<T><<<nb, 256, 0, s>>>(d_pts, un, (uint32_t)stride, m->res, base, m->keys, m->vals, m->T - 1, m->pslot, m->d_flags);
    vox_rank_kernel<<<ntiles, 256, 0, s>>>(m->pslot, m->vals, un, base, m->rank, m->tile);
    vox_tile_scan_kernel<<<1, 256, 0, s>>>(m->tile, (uint32_t)ntiles, m->d_flags);
    vox_commit_kernel<T><<<nb, 256, 0, s>>>(d_pts, un, (uint32_t)stride, m->res, base, m->pslot, m->rank, m->tile, m->vals,
                                            m->vx, m->vy, m->vz, m->fx, m->fy, m->fz);
    if (d_is_new || d_index) vox_report_kernel<<<nb, 256, 0, s>>>(m->pslot, m->rank, m->vals, un, d_is_new, d_index);
    HIPCHK(hipEventRecord(m->ev1, s));
    HIPCHK(hipGetLastError());
    VoxFlags f{};
    HIPCHK(hipMemcpyAsync(&f, m->d_flags, sizeof f, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    m->timed = true;
    if (f.table_full) return fail(PCT_ERR_INVALID, "voxel table overflow (internal sizing error)");
    m->size += f.total;
    if (n_new) *n_new = f.total;
    if (f.out_of_range)
        return fail(PCT_ERR_INVALID, "voxel coordinates outside [-2^20, 2^20) (or NaN) in the batch: those points were skipped");
    return PCT_OK;
}

int check_args(pct_voxel_map *m, const void *pts, int64_t n, int64_t stride, int is_f64)
{
    if (!m) return fail(PCT_ERR_INVALID, "null voxel map");
    if (n < 0 || (n > 0 && !pts)) return fail(PCT_ERR_INVALID, "bad point array (n=%lld)", (long long)n);
    const int64_t elem = is_f64 ? 8 : 4;
    if (stride < 3 * elem || stride % elem) return fail(PCT_ERR_INVALID, "stride %lld does not hold three %s coordinates", (long long)stride, is_f64 ? "double" : "float");
    if (n > (int64_t)0x7FFFFFFF - m->size) return fail(PCT_ERR_INVALID, "batch of %lld points is too large (2^31 limit with %lld voxels held)", (long long)n, (long long)m->size);
    if ((uint64_t)n * (uint64_t)stride > 0xFFFFFFFFull * 16ull) return fail(PCT_ERR_INVALID, "batch too large");
    return PCT_OK;
}

}  // namespace

extern "C" {

int pct_voxel_map_create(double res, int64_t capacity_hint, pct_voxel_map **out)
{
    if (!out) return fail(PCT_ERR_INVALID, "null output pointer");
    *out = nullptr;
    if (!(res > 0.0) || !std::isfinite(res)) return fail(PCT_ERR_INVALID, "voxel resolution must be positive and finite");
    PCTCHK(pct_internal::require_init());
    pct_voxel_map *m = new pct_voxel_map();
    m->res = res;
    int st = dev_alloc(&m->d_flags, 1);
    if (st == PCT_OK && hipEventCreate(&m->ev0) != hipSuccess) st = fail(PCT_ERR_HIP, "hipEventCreate failed");
    if (st == PCT_OK && hipEventCreate(&m->ev1) != hipSuccess) st = fail(PCT_ERR_HIP, "hipEventCreate failed");
    if (st == PCT_OK) st = ensure_capacity(m, std::max<int64_t>(capacity_hint, 1024));
    if (st != PCT_OK) { pct_voxel_map_destroy(m); return st; }
    *out = m;
    return PCT_OK;
}

int pct_voxel_map_destroy(pct_voxel_map *m)
{
    if (!m) return PCT_OK;
    if (pct_internal::stream()) (void)hipStreamSynchronize(pct_internal::stream());
    dev_free(m->keys); dev_free(m->vals);
    dev_free(m->vx); dev_free(m->vy); dev_free(m->vz); dev_free(m->fx); dev_free(m->fy); dev_free(m->fz);
    dev_free(m->pslot); dev_free(m->rank); dev_free(m->tile); dev_free(m->stage); dev_free(m->d_is_new); dev_free(m->d_index);
    dev_free(m->d_flags);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    delete m;
    return PCT_OK;
}

int pct_voxel_map_clear(pct_voxel_map *m)
{
    if (!m) return fail(PCT_ERR_INVALID, "null voxel map");
    hipStream_t s = pct_internal::stream();
    vox_table_init_kernel<<<std::min(blocks_for(m->T, 256), 4096), 256, 0, s>>>(m->keys, m->vals, m->T);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    m->size = 0;
    return PCT_OK;
}

int pct_voxel_map_size(const pct_voxel_map *m, int64_t *n_voxels)
{
    if (!m || !n_voxels) return fail(PCT_ERR_INVALID, "null argument");
    *n_voxels = m->size;
    return PCT_OK;
}

int pct_voxel_map_add_dev(pct_voxel_map *m, const void *d_pts, int64_t n, int64_t stride_bytes, int is_f64, int64_t *n_new,
                          uint8_t *d_is_new, int32_t *d_voxel_index)
{
    if (n_new) *n_new = 0;
    PCTCHK(check_args(m, d_pts, n, stride_bytes, is_f64));
    if (n == 0) return PCT_OK;
    const unsigned char *p = static_cast<const unsigned char *>(d_pts);
    return is_f64 ? add_device<double>(m, p, n, stride_bytes, n_new, d_is_new, d_voxel_index)
                  : add_device<float>(m, p, n, stride_bytes, n_new, d_is_new, d_voxel_index);
}

int pct_voxel_map_add(pct_voxel_map *m, const void *pts, int64_t n, int64_t stride_bytes, int is_f64, int64_t *n_new, uint8_t *is_new,
                      int32_t *voxel_index)
{
    if (n_new) *n_new = 0;
    PCTCHK(check_args(m, pts, n, stride_bytes, is_f64));
    if (n == 0) return PCT_OK;
    hipStream_t s = pct_internal::stream();
    const size_t bytes = (size_t)n * (size_t)stride_bytes;
    if (bytes > m->stage_bytes) {
        dev_free(m->stage);
        m->stage_bytes = 0;
        PCTCHK(dev_alloc(&m->stage, bytes));
        m->stage_bytes = bytes;
    }
    PCTCHK(ensure_scratch(m, n));
    if (is_new && !m->d_is_new) PCTCHK(dev_alloc(&m->d_is_new, (size_t)m->ncap));
    if (voxel_index && !m->d_index) PCTCHK(dev_alloc(&m->d_index, (size_t)m->ncap));
    HIPCHK(hipMemcpyAsync(m->stage, pts, bytes, hipMemcpyHostToDevice, s));
    const int st = pct_voxel_map_add_dev(m, m->stage, n, stride_bytes, is_f64, n_new, is_new ? m->d_is_new : nullptr,
                                         voxel_index ? m->d_index : nullptr);
    if (st != PCT_OK && st != PCT_ERR_INVALID) return st;
    if (is_new) HIPCHK(hipMemcpyAsync(is_new, m->d_is_new, (size_t)n, hipMemcpyDeviceToHost, s));
    if (voxel_index) HIPCHK(hipMemcpyAsync(voxel_index, m->d_index, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return st;
}

static int check_range(const pct_voxel_map *m, int64_t first, int64_t count, const void *out)
{
    if (!m) return fail(PCT_ERR_INVALID, "null voxel map");
    if (first < 0 || count < 0 || first + count > m->size) return fail(PCT_ERR_INVALID, "voxel range [%lld, %lld) outside [0, %lld)", (long long)first, (long long)(first + count), (long long)m->size);
    if (count > 0 && !out) return fail(PCT_ERR_INVALID, "null output buffer");
    return PCT_OK;
}

int pct_voxel_map_get_keys(const pct_voxel_map *m, int64_t first, int64_t count, int32_t *out_xyz)
{
    PCTCHK(check_range(m, first, count, out_xyz));
    if (count == 0) return PCT_OK;
    std::vector<int32_t> tmp((size_t)count * 3);
    hipStream_t s = pct_internal::stream();
    HIPCHK(hipMemcpyAsync(tmp.data(), m->vx + first, sizeof(int) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(tmp.data() + count, m->vy + first, sizeof(int) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(tmp.data() + 2 * count, m->vz + first, sizeof(int) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < count; i++) {
        out_xyz[3 * i] = tmp[i]; out_xyz[3 * i + 1] = tmp[count + i]; out_xyz[3 * i + 2] = tmp[2 * count + i];
    }
    return PCT_OK;
}

int pct_voxel_map_get_f64(const pct_voxel_map *m, int64_t first, int64_t count, double *out)
{
    PCTCHK(check_range(m, first, count, out));
    if (count == 0) return PCT_OK;
    std::vector<int32_t> k((size_t)count * 3);
    PCTCHK(pct_voxel_map_get_keys(m, first, count, k.data()));
    // voxel_map.cpp:31 with a double container: x * res (int -> double product), the same IEEE operation on the host
    for (int64_t i = 0; i < 3 * count; i++) out[i] = (double)k[i] * m->res;
    return PCT_OK;
}

int pct_voxel_map_get_f32(const pct_voxel_map *m, int64_t first, int64_t count, float *out, int64_t stride_floats)
{
    PCTCHK(check_range(m, first, count, out));
    if (stride_floats < 3) return fail(PCT_ERR_INVALID, "stride_floats must be >= 3");
    if (count == 0) return PCT_OK;
    std::vector<float> tmp((size_t)count * 3);
    hipStream_t s = pct_internal::stream();
    HIPCHK(hipMemcpyAsync(tmp.data(), m->fx + first, sizeof(float) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(tmp.data() + count, m->fy + first, sizeof(float) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(tmp.data() + 2 * count, m->fz + first, sizeof(float) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < count; i++) {
        float *o = out + i * stride_floats;
        o[0] = tmp[i]; o[1] = tmp[count + i]; o[2] = tmp[2 * count + i];
    }
    return PCT_OK;
}

int pct_voxel_map_soa_dev(const pct_voxel_map *m, const float **d_x, const float **d_y, const float **d_z, int64_t *n)
{
    if (!m || !d_x || !d_y || !d_z || !n) return fail(PCT_ERR_INVALID, "null argument");
    *d_x = m->fx; *d_y = m->fy; *d_z = m->fz; *n = m->size;
    return PCT_OK;
}

int pct_voxel_map_last_ms(const pct_voxel_map *m, float *ms)
{
    if (!m || !ms) return fail(PCT_ERR_INVALID, "null argument");
    if (!m->timed) return fail(PCT_ERR_INVALID, "no batch has been added yet");
    HIPCHK(hipEventElapsedTime(ms, m->ev0, m->ev1));
    return PCT_OK;
}

}  // extern "C"
