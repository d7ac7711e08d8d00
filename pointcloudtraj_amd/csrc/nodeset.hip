// nodeset.hip -- node sets of any dimension with fp64 coordinates (include/pct_engine.h "node sets"), part of libpct_engine.so.
//
// The general form of the reference's kd_* API: kd_create(k) for any k and positions given as doubles
// (Utils/kdtree/src/kdtree.c:112-131, 167-209).  The planner itself only builds 3-D trees from floats (corridor_finder.cpp:27-29),
// which libkdtree.so keeps in a pct_cloud; a tree of another dimension, or one holding a coordinate that fp32 cannot represent,
// lives here.  Layout: one column per coordinate ([dim][capacity] doubles: a wave reads 64 consecutive doubles of one column),
// node number = insertion order.  Both questions are exhaustive scans -- one thread per node, the reference's distance
//     s = 0; for i < dim: s += (node[i] - q[i])^2          (kdtree.c:267-272, 379-382, 420-423; fp64, no contraction)
// -- because what the caller gets back is decided by exact fp64 comparisons of that sum; the minimum is folded through one
// atomicMin on the bit pattern (a sum of squares is never negative, so doubles order like their bits), a second pass picks the
// lowest node number at the minimum and counts the ties (kd_nearest's winner among ties is a property of the insertion tree and
// is replayed on the host by libkdtree.so, csrc/kdtree_gpu.cpp reference_tie_winner).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "../../include/pct_engine.h"
#include "engine_internal.hpp"

using pct_internal::fail;

struct pct_nodeset {
    int dim = 0;
    int64_t cap = 0, size = 0;
    double *cols = nullptr;               // [dim][cap]
    double *dist = nullptr;               // [cap]: the last query's distances
    double *d_q = nullptr;                // [dim]
    unsigned long long *d_res = nullptr;  // {minimum as bits, lowest node number at the minimum, ties, hits}
    uint32_t *d_ids = nullptr;            // [cap]: hit list of the last range query
    double *h_q = nullptr;                // pinned staging: the query
    unsigned long long *h_res = nullptr;  // pinned: d_res read back
    std::vector<double> stage;            // transposed rows of one append
};

namespace {

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

constexpr int kMaxDim = 1024;
constexpr unsigned long long kNoKey = ~0ull;

__device__ __forceinline__ double node_d2(const double *__restrict__ cols, int64_t cap, int dim, uint32_t i, const double *__restrict__ q)
{
    double s = 0.0;
    for (int j = 0; j < dim; j++) {
        const double d = cols[(int64_t)j * cap + i] - q[j];
        s = s + d * d;
    }
    return s;
}

__global__ __launch_bounds__(256) void nodeset_dist_kernel(const double *__restrict__ cols, int64_t cap, int dim, uint32_t n, const double *__restrict__ q,
                                                           double *__restrict__ dist, unsigned long long *__restrict__ res)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    unsigned long long key = kNoKey;
    if (i < n) {
        const double s = node_d2(cols, cap, dim, i, q);
        dist[i] = s;
        key = (unsigned long long)__double_as_longlong(s);
        if (s != s) key = kNoKey - 1;                        // NaN (a non-finite coordinate): never the minimum unless everything is
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = (unsigned long long)__shfl_xor((long long)key, off, 64);
        key = o < key ? o : key;
    }
    if ((threadIdx.x & 63) == 0 && key != kNoKey) atomicMin(&res[0], key);
}

__global__ __launch_bounds__(256) void nodeset_pick_kernel(const double *__restrict__ dist, uint32_t n, unsigned long long *__restrict__ res)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const double s = dist[i];
    unsigned long long key = (unsigned long long)__double_as_longlong(s);
    if (s != s) key = kNoKey - 1;
    if (key == res[0]) {
        atomicMin(&res[1], (unsigned long long)i);
        atomicAdd(&res[2], 1ull);
    }
}

__global__ __launch_bounds__(256) void nodeset_radius_kernel(const double *__restrict__ cols, int64_t cap, int dim, uint32_t n, const double *__restrict__ q,
                                                             double r2, uint32_t *__restrict__ ids, uint32_t ids_cap, unsigned long long *__restrict__ res)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (node_d2(cols, cap, dim, i, q) <= r2) {               // kdtree.c:273 (inclusive)
        const unsigned long long slot = atomicAdd(&res[3], 1ull);
        if (slot < ids_cap) ids[slot] = i;
    }
}

int reserve(pct_nodeset *s, int64_t want)
{
    if (want <= s->cap) return PCT_OK;
    int64_t cap = std::max<int64_t>(s->cap, 1024);
    while (cap < want) cap *= 2;
    if (cap >= 0xFFFFFFFFll) return fail(PCT_ERR_INVALID, "node numbers travel as 32 bits");
    hipStream_t st = pct_internal::stream();
    double *cols = nullptr, *dist = nullptr;
    uint32_t *ids = nullptr;
    if (hipMalloc(&cols, sizeof(double) * (size_t)cap * s->dim) != hipSuccess || hipMalloc(&dist, sizeof(double) * (size_t)cap) != hipSuccess ||
        hipMalloc(&ids, sizeof(uint32_t) * (size_t)cap) != hipSuccess) {
        if (cols) (void)hipFree(cols);
        if (dist) (void)hipFree(dist);
        return fail(PCT_ERR_ALLOC, "node set of %lld x %d doubles does not fit", (long long)cap, s->dim);
    }
    if (s->size > 0) {
        const hipError_t e = hipMemcpy2DAsync(cols, sizeof(double) * (size_t)cap, s->cols, sizeof(double) * (size_t)s->cap, sizeof(double) * (size_t)s->size,
                                              (size_t)s->dim, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            (void)hipFree(cols); (void)hipFree(dist); (void)hipFree(ids);
            return fail(PCT_ERR_HIP, "node set: moving the columns failed");
        }
    }
    if (s->cols) (void)hipFree(s->cols);
    if (s->dist) (void)hipFree(s->dist);
    if (s->d_ids) (void)hipFree(s->d_ids);
    s->cols = cols; s->dist = dist; s->d_ids = ids; s->cap = cap;
    return PCT_OK;
}

int send_query(pct_nodeset *s, const double *q, hipStream_t st)
{
    std::memcpy(s->h_q, q, sizeof(double) * (size_t)s->dim);
    s->h_res[0] = kNoKey; s->h_res[1] = kNoKey; s->h_res[2] = 0; s->h_res[3] = 0;
    HIPCHK(hipMemcpyAsync(s->d_q, s->h_q, sizeof(double) * (size_t)s->dim, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->d_res, s->h_res, sizeof(unsigned long long) * 4, hipMemcpyHostToDevice, st));
    return PCT_OK;
}

}  // namespace

extern "C" {

int pct_nodeset_create(int dim, int64_t capacity, pct_nodeset **out)
{
    if (!out || dim < 1 || dim > kMaxDim || capacity < 0) return fail(PCT_ERR_INVALID, "bad nodeset_create arguments (1 <= dim <= %d)", kMaxDim);
    PCTCHK(pct_internal::require_init());
    pct_nodeset *s = new (std::nothrow) pct_nodeset();
    if (!s) return fail(PCT_ERR_ALLOC, "host allocation failed");
    s->dim = dim;
    if (hipMalloc(&s->d_q, sizeof(double) * (size_t)dim) != hipSuccess || hipMalloc(&s->d_res, sizeof(unsigned long long) * 4) != hipSuccess ||
        hipHostMalloc(&s->h_q, sizeof(double) * (size_t)dim, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&s->h_res, sizeof(unsigned long long) * 4, hipHostMallocDefault) != hipSuccess) {
        pct_nodeset_destroy(s);
        return fail(PCT_ERR_ALLOC, "node set: device / pinned allocation failed");
    }
    const int st = reserve(s, std::max<int64_t>(capacity, 1));
    if (st != PCT_OK) { pct_nodeset_destroy(s); return st; }
    *out = s;
    return PCT_OK;
}

int pct_nodeset_destroy(pct_nodeset *s)
{
    if (!s) return PCT_OK;
    (void)hipStreamSynchronize(pct_internal::stream());
    for (void *p : { (void *)s->cols, (void *)s->dist, (void *)s->d_q, (void *)s->d_res, (void *)s->d_ids })
        if (p) (void)hipFree(p);
    if (s->h_q) (void)hipHostFree(s->h_q);
    if (s->h_res) (void)hipHostFree(s->h_res);
    delete s;
    return PCT_OK;
}

int pct_nodeset_clear(pct_nodeset *s)
{
    if (!s) return fail(PCT_ERR_INVALID, "null node set");
    s->size = 0;
    return PCT_OK;
}

int64_t pct_nodeset_size(const pct_nodeset *s) { return s ? s->size : 0; }
int pct_nodeset_dim(const pct_nodeset *s) { return s ? s->dim : 0; }

int pct_nodeset_append(pct_nodeset *s, const double *rows, int64_t n)
{
    if (!s || n < 0 || (n > 0 && !rows)) return fail(PCT_ERR_INVALID, "bad nodeset_append arguments");
    if (n == 0) return PCT_OK;
    PCTCHK(reserve(s, s->size + n));
    hipStream_t st = pct_internal::stream();
    try { s->stage.resize((size_t)n * s->dim); } catch (const std::bad_alloc &) { return fail(PCT_ERR_ALLOC, "host allocation failed"); }
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < s->dim; j++) s->stage[(size_t)j * n + i] = rows[(size_t)i * s->dim + j];
    HIPCHK(hipMemcpy2DAsync(s->cols + s->size, sizeof(double) * (size_t)s->cap, s->stage.data(), sizeof(double) * (size_t)n, sizeof(double) * (size_t)n,
                            (size_t)s->dim, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));                      // the staging vector is pageable and reused
    s->size += n;
    return PCT_OK;
}

int pct_nodeset_nearest(pct_nodeset *s, const double *q, uint32_t *idx, double *d2, uint32_t *ties)
{
    if (!s || !q || !idx || !d2) return fail(PCT_ERR_INVALID, "bad nodeset_nearest arguments");
    if (s->size == 0) {
        *idx = PCT_NO_INDEX; *d2 = std::numeric_limits<double>::infinity();
        if (ties) *ties = 0;
        return PCT_OK;
    }
    hipStream_t st = pct_internal::stream();
    PCTCHK(send_query(s, q, st));
    const uint32_t n = (uint32_t)s->size, blocks = (n + 255u) / 256u;
    nodeset_dist_kernel<<<blocks, 256, 0, st>>>(s->cols, s->cap, s->dim, n, s->d_q, s->dist, s->d_res);
    nodeset_pick_kernel<<<blocks, 256, 0, st>>>(s->dist, n, s->d_res);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s->h_res, s->d_res, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (s->h_res[1] >= (unsigned long long)n) return fail(PCT_ERR_INTERNAL, "node set: no node at the minimum");
    *idx = (uint32_t)s->h_res[1];
    if (s->h_res[0] == kNoKey - 1) *d2 = std::numeric_limits<double>::quiet_NaN();
    else std::memcpy(d2, &s->h_res[0], sizeof(double));
    if (ties) *ties = (uint32_t)std::min<unsigned long long>(s->h_res[2], 0xFFFFFFFFull);
    return PCT_OK;
}

int pct_nodeset_radius_indices_r2(pct_nodeset *s, const double *q, double r2, uint32_t *idx_out, int64_t cap, int64_t *n_out)
{
    if (!s || !q || !n_out || cap < 0 || (cap > 0 && !idx_out)) return fail(PCT_ERR_INVALID, "bad nodeset_radius_indices_r2 arguments");
    *n_out = 0;
    if (s->size == 0) return PCT_OK;
    hipStream_t st = pct_internal::stream();
    PCTCHK(send_query(s, q, st));
    const uint32_t n = (uint32_t)s->size, blocks = (n + 255u) / 256u;
    nodeset_radius_kernel<<<blocks, 256, 0, st>>>(s->cols, s->cap, s->dim, n, s->d_q, r2, s->d_ids, (uint32_t)std::min<int64_t>(s->cap, 0xFFFFFFFFll), s->d_res);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s->h_res, s->d_res, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int64_t hits = (int64_t)s->h_res[3];
    *n_out = hits;
    if (hits == 0 || cap == 0) return PCT_OK;
    std::vector<uint32_t> all;
    try { all.resize((size_t)hits); } catch (const std::bad_alloc &) { return fail(PCT_ERR_ALLOC, "host allocation failed"); }
    HIPCHK(hipMemcpy(all.data(), s->d_ids, sizeof(uint32_t) * (size_t)hits, hipMemcpyDeviceToHost));     // hits <= size <= cap of d_ids
    std::sort(all.begin(), all.end());                     // arrival order -> ascending node number
    std::memcpy(idx_out, all.data(), sizeof(uint32_t) * (size_t)std::min<int64_t>(hits, cap));
    return PCT_OK;
}

}  // extern "C"
