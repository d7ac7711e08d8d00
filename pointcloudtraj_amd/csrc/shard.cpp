// shard.cpp -- libpct_shard.so: the multi-GPU exchange step of include/pct_shard.h.  Host code only: the kernels are reached
// through libpct_engine.so's C ABI, the collectives are RCCL's (ncclAllReduce over xGMI), everything on the caller's stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/pct_shard.h"

static_assert(PCT_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rendezvous token size");

struct pct_shard {
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int rank = 0, world = 1;
    // exchange workspaces (device), grown on demand
    int64_t cap = 0;
    double *d_ld2 = nullptr;         // per-shard squared distances
    uint32_t *d_lidx = nullptr;      // per-shard global indices
    int32_t *d_cand = nullptr;       // masked indices offered to the second reduction
    uint32_t *d_lcount = nullptr;
    float *d_q = nullptr, *d_r = nullptr;   // host-buffer convenience path
    uint32_t *d_oidx = nullptr;
    double *d_od2 = nullptr;
};

namespace {

thread_local char g_serr[512] = "";

int sfail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_serr, sizeof g_serr, fmt, ap);
    va_end(ap);
    std::fprintf(stderr, "pct_shard: %s\n", g_serr);
    return code;
}

#define NCCLCHK(call)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) return sfail(PCT_ERR_HIP, "%s -> %s", #call, ncclGetErrorString(r_));             \
    } while (0)
#define HIPCHK(call)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return sfail(PCT_ERR_HIP, "%s -> %s", #call, hipGetErrorString(e_));               \
    } while (0)
#define PCTCHK(call)                                                                                           \
    do {                                                                                                       \
        int s_ = (call);                                                                                       \
        if (s_ != PCT_OK) return sfail(s_, "%s -> %s", #call, pct_last_error());                                \
    } while (0)

template <typename T>
int grow(T **p, int64_t n)
{
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    void *v = nullptr;
    if (hipMalloc(&v, sizeof(T) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess) return sfail(PCT_ERR_ALLOC, "hipMalloc of an exchange workspace failed");
    *p = static_cast<T *>(v);
    return PCT_OK;
}

int reserve(pct_shard *s, int64_t Q)
{
    if (Q <= s->cap) return PCT_OK;
    HIPCHK(hipDeviceSynchronize());            // a previous batch may still read the old workspaces
    s->cap = 0;
    const int64_t n = std::max<int64_t>(Q, 256);
    if (grow(&s->d_ld2, n) || grow(&s->d_lidx, n) || grow(&s->d_cand, n) || grow(&s->d_lcount, n) || grow(&s->d_q, 3 * n) || grow(&s->d_r, n) ||
        grow(&s->d_oidx, n) || grow(&s->d_od2, n))
        return PCT_ERR_ALLOC;
    s->cap = n;
    return PCT_OK;
}

}  // namespace

extern "C" {

int pct_shard_unique_id(void *id_out)
{
    if (!id_out) return sfail(PCT_ERR_INVALID, "null id buffer");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    std::memcpy(id_out, id.internal, PCT_SHARD_ID_BYTES);
    return PCT_OK;
}

int pct_shard_init(const void *id, int rank, int world, int device, pct_shard **out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return sfail(PCT_ERR_INVALID, "bad shard_init arguments");
    PCTCHK(pct_init(device));
    pct_shard *s = new (std::nothrow) pct_shard();
    if (!s) return sfail(PCT_ERR_ALLOC, "host allocation failed");
    s->rank = rank;
    s->world = world;
    s->own_comm = true;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, PCT_SHARD_ID_BYTES);
    const ncclResult_t r = ncclCommInitRank(&s->comm, world, uid, rank);
    if (r != ncclSuccess) { delete s; return sfail(PCT_ERR_HIP, "ncclCommInitRank -> %s", ncclGetErrorString(r)); }
    *out = s;
    return PCT_OK;
}

int pct_shard_init_comm(void *nccl_comm, int rank, int world, pct_shard **out)
{
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return sfail(PCT_ERR_INVALID, "bad shard_init_comm arguments");
    pct_shard *s = new (std::nothrow) pct_shard();
    if (!s) return sfail(PCT_ERR_ALLOC, "host allocation failed");
    s->comm = static_cast<ncclComm_t>(nccl_comm);
    s->rank = rank;
    s->world = world;
    *out = s;
    return PCT_OK;
}

int pct_shard_destroy(pct_shard *s)
{
    if (!s) return PCT_OK;
    (void)hipDeviceSynchronize();
    if (s->own_comm && s->comm) (void)ncclCommDestroy(s->comm);
    for (void *p : { (void *)s->d_ld2, (void *)s->d_lidx, (void *)s->d_cand, (void *)s->d_lcount, (void *)s->d_q, (void *)s->d_r, (void *)s->d_oidx, (void *)s->d_od2 })
        if (p) (void)hipFree(p);
    delete s;
    return PCT_OK;
}

int pct_shard_rank(const pct_shard *s) { return s ? s->rank : -1; }
int pct_shard_world(const pct_shard *s) { return s ? s->world : 0; }

int pct_shard_range(const pct_shard *s, int64_t n_total, int64_t *begin, int64_t *end)
{
    if (!s || n_total < 0 || !begin || !end) return sfail(PCT_ERR_INVALID, "bad shard_range arguments");
    *begin = (int64_t)(((__int128)s->rank * n_total) / s->world);
    *end = (int64_t)(((__int128)(s->rank + 1) * n_total) / s->world);
    return PCT_OK;
}

int pct_shard_cloud_create(pct_shard *s, int64_t n_total, pct_cloud **out)
{
    if (!out) return sfail(PCT_ERR_INVALID, "null output");
    if (n_total >= 0x7FFFFFFFll) return sfail(PCT_ERR_INVALID, "global indices travel as int32: the cloud must hold fewer than 2^31 - 1 points");
    int64_t b = 0, e = 0;
    PCTCHK(pct_shard_range(s, n_total, &b, &e));
    PCTCHK(pct_cloud_create(std::max<int64_t>(e - b, 1), out));
    const int st = pct_cloud_set_index_base(*out, b);
    if (st != PCT_OK) { pct_cloud_destroy(*out); *out = nullptr; return sfail(st, "pct_cloud_set_index_base -> %s", pct_last_error()); }
    return PCT_OK;
}

int pct_shard_nn_dev(pct_shard *s, pct_cloud *local, int algo, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, void *stream)
{
    if (!s || !local || Q < 0 || (Q > 0 && (!d_q || !d_idx || !d_d2))) return sfail(PCT_ERR_INVALID, "bad shard_nn_dev arguments");
    if (Q == 0) return PCT_OK;
    PCTCHK(reserve(s, Q));
    PCTCHK(pct_cloud_reserve_queries(local, Q));
    hipStream_t st = static_cast<hipStream_t>(stream);
    PCTCHK(pct_nn_batch_dev(local, algo, d_q, Q, s->d_lidx, s->d_ld2, stream));          // per-shard winners, global indices
    NCCLCHK(ncclAllReduce(s->d_ld2, d_d2, (size_t)Q, ncclDouble, ncclMin, s->comm, st));
    PCTCHK(pct_merge_mask_dev(s->d_ld2, d_d2, s->d_lidx, s->d_cand, Q, stream));         // offer the index only where this shard holds the minimum
    NCCLCHK(ncclAllReduce(s->d_cand, s->d_cand, (size_t)Q, ncclInt32, ncclMin, s->comm, st));
    PCTCHK(pct_merge_finish_dev(s->d_cand, d_idx, Q, stream));                            // INT32_MAX -> PCT_NO_INDEX
    return PCT_OK;
}

int pct_shard_radius_count_dev(pct_shard *s, pct_cloud *local, int algo, const float *d_q, const float *d_r, int64_t Q, uint32_t *d_count, void *stream)
{
    if (!s || !local || Q < 0 || (Q > 0 && (!d_q || !d_r || !d_count))) return sfail(PCT_ERR_INVALID, "bad shard_radius_count_dev arguments");
    if (Q == 0) return PCT_OK;
    PCTCHK(reserve(s, Q));
    PCTCHK(pct_cloud_reserve_queries(local, Q));
    PCTCHK(pct_radius_count_batch_dev(local, algo, d_q, d_r, Q, s->d_lcount, stream));
    NCCLCHK(ncclAllReduce(s->d_lcount, d_count, (size_t)Q, ncclUint32, ncclSum, s->comm, static_cast<hipStream_t>(stream)));
    return PCT_OK;
}

int pct_shard_nn(pct_shard *s, pct_cloud *local, int algo, const float *q, int64_t Q, uint32_t *idx, double *d2)
{
    if (!s || !local || Q < 0 || (Q > 0 && (!q || !idx || !d2))) return sfail(PCT_ERR_INVALID, "bad shard_nn arguments");
    if (Q == 0) return PCT_OK;
    PCTCHK(reserve(s, Q));
    HIPCHK(hipMemcpy(s->d_q, q, sizeof(float) * 3 * (size_t)Q, hipMemcpyHostToDevice));
    PCTCHK(pct_shard_nn_dev(s, local, algo, s->d_q, Q, s->d_oidx, s->d_od2, nullptr));
    HIPCHK(hipMemcpy(idx, s->d_oidx, sizeof(uint32_t) * (size_t)Q, hipMemcpyDeviceToHost));   // null-stream copies: ordered behind the batch
    HIPCHK(hipMemcpy(d2, s->d_od2, sizeof(double) * (size_t)Q, hipMemcpyDeviceToHost));
    return PCT_OK;
}

}  // extern "C"
