// shard.cpp -- libpct_shard.so: the multi-GPU exchange step of include/pct_shard.h.  Host code only: the kernels are reached
// through libpct_engine.so's C ABI, the collectives are RCCL's (ncclAllReduce over xGMI), everything on the caller's stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "../../include/pct_shard.h"

static_assert(PCT_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rendezvous token size");

struct pct_shard {
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    bool local = false;              // one of several ranks living in ONE process (pct_shard_local_world): no communicator, the
                                     // *_world entry points move the data between the ranks' buffers themselves
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

// =====================================================================================================================================
// Routed form: slab ownership.  Under index-range sharding every rank answers every query, and the cost of a cell-pruned query depends
// on the local point density, not on the shard's size -- W ranks do W times the work.  Here the cloud is re-distributed ONCE into W
// slabs of equal point count along its longest axis (+ a halo of a few point spacings on both sides, points kept in ascending
// global-index order so that "lowest local index" is "lowest global index"), every query is answered by the ONE rank that owns its
// slab, an answer is certified when the point found is strictly nearer than the edge of the owner's halo, and the owned answers are
// exchanged as 16-byte records (an all-gather of variable-sized slices: grouped ncclSend / ncclRecv) instead of two all-reduces over
// the whole batch.  Uncertified answers (d2 = -1 in the record: rare with a halo of 4 spacings) are answered by everybody in a second,
// small round merged with the all_reduce(min) pair of the index-range form.  Results are identical to the single cloud's.
// The phases are written once; what differs between "one process per GPU over RCCL" and "several ranks in this process" is who moves
// the bytes between the ranks (exchange_* below).
// =====================================================================================================================================
struct RouteAnswer { uint32_t query, gid; double d2; };
struct PointRec { float x, y, z; uint32_t gid; };
static_assert(sizeof(RouteAnswer) == 16 && sizeof(PointRec) == 16, "record sizes");
constexpr int kRouteMaxWorld = 64;
constexpr int kRouteBins = 4096;

struct pct_route {
    pct_shard *s = nullptr;
    int rank = 0, world = 1, axis = 0;
    double cuts[kRouteMaxWorld + 1] = {};
    double halo = 0.0;
    int64_t n_total = 0, n_slab = 0;
    pct_cloud *slab = nullptr;
    uint32_t *d_gid = nullptr;
    // batch workspaces
    int64_t cap = 0;
    uint32_t *d_counts = nullptr, *d_mine_ids = nullptr, *d_lidx = nullptr, *d_flag = nullptr, *d_flag_ids = nullptr, *d_fidx = nullptr;
    float *d_mine_q = nullptr, *d_fq = nullptr;
    double *d_ld2 = nullptr, *d_fd2 = nullptr, *d_fbest = nullptr;
    int32_t *d_fcand = nullptr;
    RouteAnswer *d_send = nullptr, *d_recv = nullptr;
    uint32_t h_counts[kRouteMaxWorld] = {};
    int64_t mine = 0, flagged = 0;
    uint64_t st_owned = 0, st_uncert = 0, st_batches = 0;
    // partitioned batches (every rank brings its own queries): send side sized by this rank's Q, receive side by what the others route here
    int64_t pq_cap = 0, pr_cap = 0, pf_cap = 0;
    unsigned char *d_owner = nullptr;
    uint32_t *d_pflag = nullptr, *d_pflag_ids = nullptr;
    uint32_t *d_pcnt = nullptr, *d_pmat = nullptr, *d_sq_slot = nullptr, *d_rq_slot = nullptr, *d_plidx = nullptr, *d_pfidx = nullptr;
    float *d_sq_xyz = nullptr, *d_rq_xyz = nullptr, *d_pfq_mine = nullptr, *d_pfq_all = nullptr;
    double *d_pld2 = nullptr, *d_pfd2 = nullptr, *d_pfbest = nullptr;
    int32_t *d_pfcand = nullptr;
    RouteAnswer *d_ans_out = nullptr, *d_ans_in = nullptr;
    uint32_t p_send[kRouteMaxWorld] = {};          // how many of my queries each rank owns
    int64_t p_recv = 0;                            // how many queries the others (and I) route to me
};

namespace {

int route_reserve(pct_route *r, int64_t Q)
{
    if (Q <= r->cap) return PCT_OK;
    HIPCHK(hipDeviceSynchronize());
    r->cap = 0;
    const int64_t n = std::max<int64_t>(Q, 256);
    if (grow(&r->d_counts, kRouteMaxWorld) || grow(&r->d_mine_ids, n) || grow(&r->d_lidx, n) || grow(&r->d_flag, 4) || grow(&r->d_flag_ids, n) || grow(&r->d_fidx, n) ||
        grow(&r->d_mine_q, 3 * n) || grow(&r->d_fq, 3 * n) || grow(&r->d_ld2, n) || grow(&r->d_fd2, n) || grow(&r->d_fbest, n) || grow(&r->d_fcand, n) ||
        grow(&r->d_send, n) || grow(&r->d_recv, n))
        return PCT_ERR_ALLOC;
    r->cap = n;
    return PCT_OK;
}

// ---- build, per rank: statistics of the local rows ----
struct LocalStats { double lo[3], hi[3], n; };

LocalStats local_stats(const unsigned char *pts, int64_t n, int64_t stride)
{
    LocalStats L;
    for (int k = 0; k < 3; k++) { L.lo[k] = std::numeric_limits<double>::infinity(); L.hi[k] = -std::numeric_limits<double>::infinity(); }
    L.n = (double)n;
    for (int64_t i = 0; i < n; i++) {
        const float *p = reinterpret_cast<const float *>(pts + i * stride);
        for (int k = 0; k < 3; k++) { L.lo[k] = std::min(L.lo[k], (double)p[k]); L.hi[k] = std::max(L.hi[k], (double)p[k]); }
    }
    return L;
}

struct Layout { int axis; double lo, hi_edge, spacing; int64_t n_total; };

Layout layout_from(const LocalStats &G)
{
    Layout Y{};
    Y.n_total = (int64_t)G.n;
    double ext[3], emax = 0;
    for (int k = 0; k < 3; k++) { ext[k] = std::max(G.hi[k] - G.lo[k], 1e-30); emax = std::max(emax, ext[k]); }
    Y.axis = 0;
    for (int k = 1; k < 3; k++) if (ext[k] > ext[Y.axis]) Y.axis = k;
    Y.lo = G.lo[Y.axis];
    Y.hi_edge = G.hi[Y.axis] + ext[Y.axis] * 1e-9;
    double vol = 1.0;
    for (int k = 0; k < 3; k++) vol *= std::max(ext[k], emax * 1e-3);
    Y.spacing = std::cbrt(vol / std::max<double>(G.n, 1.0));
    return Y;
}

void local_histogram(const unsigned char *pts, int64_t n, int64_t stride, const Layout &Y, double *h)
{
    for (int b = 0; b < kRouteBins; b++) h[b] = 0.0;
    const double w = (Y.hi_edge - Y.lo) / kRouteBins;
    for (int64_t i = 0; i < n; i++) {
        const double x = (double)reinterpret_cast<const float *>(pts + i * stride)[Y.axis];
        const int b = std::max(0, std::min(kRouteBins - 1, (int)std::floor((x - Y.lo) / w)));
        h[b] += 1.0;
    }
}

void cuts_from(const Layout &Y, const double *h, int world, double *cuts)
{
    const double w = (Y.hi_edge - Y.lo) / kRouteBins;
    std::vector<double> cum(kRouteBins);
    double acc = 0;
    for (int b = 0; b < kRouteBins; b++) { acc += h[b]; cum[b] = acc; }
    cuts[0] = -std::numeric_limits<double>::infinity();
    for (int k = 1; k < world; k++) {
        const double want = (double)Y.n_total * k / world;
        const int b = (int)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());        // first bin whose cumulative count reaches the share
        cuts[k] = Y.lo + w * std::min(b + 1, kRouteBins);
    }
    cuts[world] = std::numeric_limits<double>::infinity();
}

// the rows of this rank that slab k needs: its own interval widened by the halo, ascending local order
void select_for(const unsigned char *pts, int64_t n, int64_t stride, int64_t index_begin, int axis, double lo, double hi, std::vector<PointRec> &out)
{
    for (int64_t i = 0; i < n; i++) {
        const float *p = reinterpret_cast<const float *>(pts + i * stride);
        const double x = (double)p[axis];
        if (x >= lo && x < hi) out.push_back(PointRec{ p[0], p[1], p[2], (uint32_t)(index_begin + i) });
    }
}

// the slab as a pct_cloud + the table local index -> global index, from `n` records in device memory (ascending global index)
int route_load_slab(pct_route *r, const PointRec *d_recs, int64_t n)
{
    r->n_slab = n;
    PCTCHK(pct_cloud_create(std::max<int64_t>(n, 1), &r->slab));
    if (grow(&r->d_gid, std::max<int64_t>(n, 1))) return PCT_ERR_ALLOC;
    if (n) {
        PCTCHK(pct_cloud_upload_aos_dev(r->slab, d_recs, n, sizeof(PointRec)));
        HIPCHK(hipMemcpy2D(r->d_gid, sizeof(uint32_t), reinterpret_cast<const unsigned char *>(d_recs) + 12, sizeof(PointRec), sizeof(uint32_t), (size_t)n, hipMemcpyDeviceToDevice));
        PCTCHK(pct_cloud_build_grid(r->slab, 0.0f));
    }
    return PCT_OK;
}

// ---- batch phases (per rank) ----
int phase_owner(pct_route *r, const float *d_q, int64_t Q, hipStream_t st)
{
    PCTCHK(route_reserve(r, Q));
    PCTCHK(pct_route_owner_dev(r->cuts, r->world, r->axis, r->rank, d_q, Q, r->d_counts, r->d_mine_ids, r->d_mine_q, st));
    HIPCHK(hipMemcpyAsync(r->h_counts, r->d_counts, sizeof(uint32_t) * (size_t)r->world, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));              // the owned share sizes the next launch and the exchange (the same numbers on every rank)
    r->mine = r->h_counts[r->rank];
    return PCT_OK;
}

int phase_answer(pct_route *r, hipStream_t st)
{
    const int64_t m = r->mine;
    if (m == 0) return PCT_OK;
    if (r->n_slab > 0) {
        PCTCHK(pct_cloud_reserve_queries(r->slab, m));
        PCTCHK(pct_nn_batch_dev(r->slab, PCT_ALGO_AUTO, r->d_mine_q, m, r->d_lidx, r->d_ld2, st));
    } else {
        HIPCHK(hipMemsetAsync(r->d_lidx, 0xFF, sizeof(uint32_t) * (size_t)m, st));                 // an owner without points: every answer is "ask everybody"
        HIPCHK(hipMemsetAsync(r->d_ld2, 0, sizeof(double) * (size_t)m, st));
    }
    const double lo_edge = r->cuts[r->rank] - r->halo, hi_edge = r->cuts[r->rank + 1] + r->halo;     // +-inf at the outer slabs
    PCTCHK(pct_route_certify_dev(r->axis, lo_edge, hi_edge, r->d_mine_q, r->d_mine_ids, m, r->d_lidx, r->d_ld2, r->d_gid, r->d_send, st));
    return PCT_OK;
}

int phase_scatter(pct_route *r, int64_t Q, uint32_t *d_idx, double *d_d2, hipStream_t st)
{
    PCTCHK(pct_route_scatter_dev(r->d_recv, Q, d_idx, d_d2, r->d_flag, r->d_flag_ids, st));
    uint32_t f = 0;
    HIPCHK(hipMemcpyAsync(&f, r->d_flag, sizeof f, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    r->flagged = f;
    r->st_owned += (uint64_t)r->mine;
    r->st_batches++;
    if (f) {                                       // the same set on every rank, but listed in arrival order: sort it (it is small)
        std::vector<uint32_t> ids(f);
        HIPCHK(hipMemcpy(ids.data(), r->d_flag_ids, sizeof(uint32_t) * f, hipMemcpyDeviceToHost));
        std::sort(ids.begin(), ids.end());
        HIPCHK(hipMemcpy(r->d_flag_ids, ids.data(), sizeof(uint32_t) * f, hipMemcpyHostToDevice));
    }
    return PCT_OK;
}

int phase_second_local(pct_route *r, const float *d_q, hipStream_t st)
{
    const int64_t f = r->flagged;
    PCTCHK(pct_route_gather_queries_dev(d_q, r->d_flag_ids, f, r->d_fq, st));
    if (r->n_slab > 0) {
        PCTCHK(pct_cloud_reserve_queries(r->slab, f));
        PCTCHK(pct_nn_batch_dev(r->slab, PCT_ALGO_AUTO, r->d_fq, f, r->d_fidx, r->d_fd2, st));
        PCTCHK(pct_route_to_global_dev(r->d_fidx, f, r->d_gid, st));
    } else {
        HIPCHK(hipMemsetAsync(r->d_fidx, 0xFF, sizeof(uint32_t) * (size_t)f, st));
        std::vector<double> inf((size_t)f, std::numeric_limits<double>::infinity());
        HIPCHK(hipMemcpyAsync(r->d_fd2, inf.data(), sizeof(double) * (size_t)f, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return PCT_OK;
}

int phase_second_finish(pct_route *r, uint32_t *d_idx, double *d_d2, hipStream_t st)       // d_fbest / d_fcand hold the merged round
{
    const int64_t f = r->flagged;
    PCTCHK(pct_merge_finish_dev(r->d_fcand, r->d_fidx, f, st));
    PCTCHK(pct_route_put_back_dev(r->d_flag_ids, f, r->d_fidx, r->d_fbest, d_idx, d_d2, st));
    return PCT_OK;
}

// ---- partitioned batches: phases per rank (the exchanges between them are RCCL's or plain copies, see the entry points) ----
int part_reserve_send(pct_route *r, int64_t Q)
{
    if (Q <= r->pq_cap) return PCT_OK;
    HIPCHK(hipDeviceSynchronize());
    r->pq_cap = 0;
    const int64_t n = std::max<int64_t>(Q, 256);
    if (grow(&r->d_owner, n) || grow(&r->d_pcnt, 4 * kRouteMaxWorld) || grow(&r->d_pmat, (int64_t)kRouteMaxWorld * kRouteMaxWorld + kRouteMaxWorld) || grow(&r->d_sq_slot, n) ||
        grow(&r->d_sq_xyz, 3 * n) || grow(&r->d_ans_in, n) || grow(&r->d_pflag, 4) || grow(&r->d_pflag_ids, n) || grow(&r->d_pfq_mine, 3 * n))
        return PCT_ERR_ALLOC;
    r->pq_cap = n;
    return PCT_OK;
}
int part_reserve_recv(pct_route *r, int64_t n_recv)
{
    if (n_recv <= r->pr_cap) return PCT_OK;
    HIPCHK(hipDeviceSynchronize());
    r->pr_cap = 0;
    const int64_t n = std::max<int64_t>(n_recv + n_recv / 4, 256);
    if (grow(&r->d_rq_slot, n) || grow(&r->d_rq_xyz, 3 * n) || grow(&r->d_plidx, n) || grow(&r->d_pld2, n) || grow(&r->d_ans_out, n)) return PCT_ERR_ALLOC;
    r->pr_cap = n;
    return PCT_OK;
}
int part_reserve_flag(pct_route *r, int64_t F)
{
    if (F <= r->pf_cap) return PCT_OK;
    HIPCHK(hipDeviceSynchronize());
    r->pf_cap = 0;
    const int64_t n = std::max<int64_t>(2 * F, 256);
    if (grow(&r->d_pfq_all, 3 * n) || grow(&r->d_pfidx, n) || grow(&r->d_pfd2, n) || grow(&r->d_pfbest, n) || grow(&r->d_pfcand, n)) return PCT_ERR_ALLOC;
    r->pf_cap = n;
    return PCT_OK;
}

int part_phase_count(pct_route *r, const float *d_q, int64_t Q, hipStream_t st)
{
    PCTCHK(part_reserve_send(r, Q));
    PCTCHK(pct_route_owner_all_dev(r->cuts, r->world, r->axis, d_q, Q, r->d_pcnt, r->d_owner, st));
    HIPCHK(hipMemcpyAsync(r->p_send, r->d_pcnt, sizeof(uint32_t) * (size_t)r->world, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return PCT_OK;
}
// mat[src * W + dst] = queries of rank src owned by rank dst (known to every rank)
int part_phase_partition(pct_route *r, const float *d_q, int64_t Q, const uint32_t *mat, hipStream_t st)
{
    uint32_t off[kRouteMaxWorld];
    uint32_t acc = 0;
    for (int k = 0; k < r->world; k++) { off[k] = acc; acc += r->p_send[k]; }
    PCTCHK(pct_route_partition_dev(off, r->world, r->d_owner, d_q, Q, r->d_pcnt + kRouteMaxWorld, r->d_sq_xyz, r->d_sq_slot, st));
    int64_t n = 0;
    for (int src = 0; src < r->world; src++) n += mat[(size_t)src * r->world + r->rank];
    r->p_recv = n;
    return part_reserve_recv(r, n);
}
int part_phase_answer(pct_route *r, hipStream_t st)
{
    const int64_t m = r->p_recv;
    if (m == 0) return PCT_OK;
    if (r->n_slab > 0) {
        PCTCHK(pct_cloud_reserve_queries(r->slab, m));
        PCTCHK(pct_nn_batch_dev(r->slab, PCT_ALGO_AUTO, r->d_rq_xyz, m, r->d_plidx, r->d_pld2, st));
    } else {
        HIPCHK(hipMemsetAsync(r->d_plidx, 0xFF, sizeof(uint32_t) * (size_t)m, st));
        HIPCHK(hipMemsetAsync(r->d_pld2, 0, sizeof(double) * (size_t)m, st));
    }
    const double lo_edge = r->cuts[r->rank] - r->halo, hi_edge = r->cuts[r->rank + 1] + r->halo;
    PCTCHK(pct_route_certify_dev(r->axis, lo_edge, hi_edge, r->d_rq_xyz, r->d_rq_slot, m, r->d_plidx, r->d_pld2, r->d_gid, r->d_ans_out, st));
    r->st_owned += (uint64_t)m;
    return PCT_OK;
}
int part_phase_scatter(pct_route *r, int64_t Q, uint32_t *d_idx, double *d_d2, hipStream_t st)
{
    PCTCHK(pct_route_scatter_dev(r->d_ans_in, Q, d_idx, d_d2, r->d_pflag, r->d_pflag_ids, st));
    uint32_t f = 0;
    HIPCHK(hipMemcpyAsync(&f, r->d_pflag, sizeof f, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    r->flagged = f;
    r->st_batches++;
    r->st_uncert += f;
    if (f) {
        std::vector<uint32_t> ids(f);
        HIPCHK(hipMemcpy(ids.data(), r->d_pflag_ids, sizeof(uint32_t) * f, hipMemcpyDeviceToHost));
        std::sort(ids.begin(), ids.end());
        HIPCHK(hipMemcpy(r->d_pflag_ids, ids.data(), sizeof(uint32_t) * f, hipMemcpyHostToDevice));
    }
    return PCT_OK;
}
int part_phase_second_answer(pct_route *r, int64_t F, hipStream_t st)        // d_pfq_all holds every rank's uncertified queries, rank order
{
    if (r->n_slab > 0) {
        PCTCHK(pct_cloud_reserve_queries(r->slab, F));
        PCTCHK(pct_nn_batch_dev(r->slab, PCT_ALGO_AUTO, r->d_pfq_all, F, r->d_pfidx, r->d_pfd2, st));
        PCTCHK(pct_route_to_global_dev(r->d_pfidx, F, r->d_gid, st));
    } else {
        HIPCHK(hipMemsetAsync(r->d_pfidx, 0xFF, sizeof(uint32_t) * (size_t)F, st));
        std::vector<double> inf((size_t)F, std::numeric_limits<double>::infinity());
        HIPCHK(hipMemcpyAsync(r->d_pfd2, inf.data(), sizeof(double) * (size_t)F, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return PCT_OK;
}
int part_phase_second_finish(pct_route *r, int64_t F, int64_t my_off, uint32_t *d_idx, double *d_d2, hipStream_t st)      // d_pfbest / d_pfcand merged
{
    PCTCHK(pct_merge_finish_dev(r->d_pfcand, r->d_pfidx, F, st));
    PCTCHK(pct_route_put_back_dev(r->d_pflag_ids, r->flagged, r->d_pfidx + my_off, r->d_pfbest + my_off, d_idx, d_d2, st));
    return PCT_OK;
}

void route_free(pct_route *r)
{
    if (!r) return;
    (void)hipDeviceSynchronize();
    if (r->slab) (void)pct_cloud_destroy(r->slab);
    for (void *p : { (void *)r->d_pflag, (void *)r->d_pflag_ids, (void *)r->d_owner, (void *)r->d_pcnt, (void *)r->d_pmat, (void *)r->d_sq_slot, (void *)r->d_rq_slot, (void *)r->d_plidx, (void *)r->d_pfidx, (void *)r->d_sq_xyz,
                     (void *)r->d_rq_xyz, (void *)r->d_pfq_mine, (void *)r->d_pfq_all, (void *)r->d_pld2, (void *)r->d_pfd2, (void *)r->d_pfbest, (void *)r->d_pfcand,
                     (void *)r->d_ans_out, (void *)r->d_ans_in })
        if (p) (void)hipFree(p);
    for (void *p : { (void *)r->d_gid, (void *)r->d_counts, (void *)r->d_mine_ids, (void *)r->d_lidx, (void *)r->d_flag, (void *)r->d_flag_ids, (void *)r->d_fidx, (void *)r->d_mine_q,
                     (void *)r->d_fq, (void *)r->d_ld2, (void *)r->d_fd2, (void *)r->d_fbest, (void *)r->d_fcand, (void *)r->d_send, (void *)r->d_recv })
        if (p) (void)hipFree(p);
    delete r;
}

pct_route *route_new(pct_shard *s, const Layout &Y, const double *cuts, double halo_spacings)
{
    pct_route *r = new (std::nothrow) pct_route();
    if (!r) return nullptr;
    r->s = s; r->rank = s->rank; r->world = s->world; r->axis = Y.axis; r->n_total = Y.n_total;
    for (int k = 0; k <= s->world; k++) r->cuts[k] = cuts[k];
    r->halo = halo_spacings * Y.spacing;
    return r;
}

}  // namespace

extern "C" {

// ---- one process per GPU: the exchanges are RCCL's ----
int pct_shard_route_build(pct_shard *s, const void *local_points, int64_t n_local, int64_t stride_bytes, int64_t index_begin, double halo_spacings, pct_route **out)
{
    if (!s || s->local || !out || n_local < 0 || (n_local > 0 && !local_points) || stride_bytes < 12 || index_begin < 0 || !(halo_spacings >= 0))
        return sfail(PCT_ERR_INVALID, "bad route_build arguments");
    if (s->world > kRouteMaxWorld) return sfail(PCT_ERR_INVALID, "at most %d ranks", kRouteMaxWorld);
    const unsigned char *pts = static_cast<const unsigned char *>(local_points);
    const int W = s->world;
    hipStream_t st = nullptr;
    // global bounding box / count, then the histogram along the longest axis: three small all-reduces on one device buffer
    double *d_buf = nullptr;
    if (grow(&d_buf, kRouteBins + 16)) return PCT_ERR_ALLOC;
    struct Free { double *&p; ~Free() { if (p) (void)hipFree(p); } } free_buf{ d_buf };
    LocalStats L = local_stats(pts, n_local, stride_bytes), G = L;
    HIPCHK(hipMemcpy(d_buf, L.lo, sizeof(double) * 3, hipMemcpyHostToDevice));
    NCCLCHK(ncclAllReduce(d_buf, d_buf, 3, ncclDouble, ncclMin, s->comm, st));
    HIPCHK(hipMemcpy(G.lo, d_buf, sizeof(double) * 3, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(d_buf, L.hi, sizeof(double) * 3, hipMemcpyHostToDevice));
    NCCLCHK(ncclAllReduce(d_buf, d_buf, 3, ncclDouble, ncclMax, s->comm, st));
    HIPCHK(hipMemcpy(G.hi, d_buf, sizeof(double) * 3, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(d_buf, &L.n, sizeof(double), hipMemcpyHostToDevice));
    NCCLCHK(ncclAllReduce(d_buf, d_buf, 1, ncclDouble, ncclSum, s->comm, st));
    HIPCHK(hipMemcpy(&G.n, d_buf, sizeof(double), hipMemcpyDeviceToHost));
    if ((int64_t)G.n >= 0x7FFFFFFFll) return sfail(PCT_ERR_INVALID, "global indices travel as 32 bits");
    double cuts[kRouteMaxWorld + 1];
    Layout Y{};
    if (G.n > 0) {
        Y = layout_from(G);
        std::vector<double> h(kRouteBins);
        local_histogram(pts, n_local, stride_bytes, Y, h.data());
        HIPCHK(hipMemcpy(d_buf, h.data(), sizeof(double) * kRouteBins, hipMemcpyHostToDevice));
        NCCLCHK(ncclAllReduce(d_buf, d_buf, kRouteBins, ncclDouble, ncclSum, s->comm, st));
        HIPCHK(hipMemcpy(h.data(), d_buf, sizeof(double) * kRouteBins, hipMemcpyDeviceToHost));
        cuts_from(Y, h.data(), W, cuts);
    } else {
        cuts[0] = -std::numeric_limits<double>::infinity();
        for (int k = 1; k <= W; k++) cuts[k] = std::numeric_limits<double>::infinity();
    }
    pct_route *r = route_new(s, Y, cuts, halo_spacings);
    if (!r) return sfail(PCT_ERR_ALLOC, "host allocation failed");
    // what every slab needs of my rows; the share sizes travel by all-gather, the rows by grouped send / recv
    std::vector<std::vector<PointRec>> to((size_t)W);
    std::vector<uint32_t> send_n((size_t)W), all_n((size_t)W * W);
    for (int k = 0; k < W && G.n > 0; k++) select_for(pts, n_local, stride_bytes, index_begin, r->axis, r->cuts[k] - r->halo, r->cuts[k + 1] + r->halo, to[(size_t)k]);
    for (int k = 0; k < W; k++) send_n[(size_t)k] = (uint32_t)to[(size_t)k].size();
    uint32_t *d_n = nullptr;
    if (grow(&d_n, (int64_t)W * W + W)) { route_free(r); return PCT_ERR_ALLOC; }
    struct FreeN { uint32_t *&p; ~FreeN() { if (p) (void)hipFree(p); } } free_n{ d_n };
    int stt = PCT_OK;
    auto bail = [&](int code) { route_free(r); return code; };
    if (hipMemcpy(d_n + (size_t)W * W, send_n.data(), sizeof(uint32_t) * W, hipMemcpyHostToDevice) != hipSuccess) return bail(sfail(PCT_ERR_HIP, "hipMemcpy failed"));
    if (ncclAllGather(d_n + (size_t)W * W, d_n, (size_t)W, ncclUint32, s->comm, st) != ncclSuccess) return bail(sfail(PCT_ERR_HIP, "ncclAllGather failed"));
    if (hipMemcpy(all_n.data(), d_n, sizeof(uint32_t) * W * W, hipMemcpyDeviceToHost) != hipSuccess) return bail(sfail(PCT_ERR_HIP, "hipMemcpy failed"));
    int64_t n_send = 0, n_recv = 0;
    for (int k = 0; k < W; k++) { n_send += send_n[(size_t)k]; n_recv += all_n[(size_t)k * W + s->rank]; }     // all_n[src * W + dst]
    PointRec *d_out = nullptr, *d_in = nullptr;
    if (grow(&d_out, n_send) || grow(&d_in, n_recv)) { if (d_out) (void)hipFree(d_out); return bail(PCT_ERR_ALLOC); }
    struct FreeP { PointRec *&a, *&b; ~FreeP() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } } free_p{ d_out, d_in };
    {
        int64_t off = 0;
        for (int k = 0; k < W; k++) {
            if (!to[(size_t)k].empty() && hipMemcpy(d_out + off, to[(size_t)k].data(), sizeof(PointRec) * to[(size_t)k].size(), hipMemcpyHostToDevice) != hipSuccess)
                return bail(sfail(PCT_ERR_HIP, "hipMemcpy failed"));
            off += (int64_t)to[(size_t)k].size();
        }
    }
    ncclResult_t nr = ncclGroupStart();
    {
        int64_t so = 0, ro = 0;
        for (int k = 0; k < W && nr == ncclSuccess; k++) {                 // receive in rank order: ascending global index without a sort
            const int64_t ns = send_n[(size_t)k], nrv = all_n[(size_t)k * W + s->rank];
            if (k == s->rank) {
                if (ns && hipMemcpyAsync(d_in + ro, d_out + so, sizeof(PointRec) * (size_t)ns, hipMemcpyDeviceToDevice, st) != hipSuccess) nr = ncclInternalError;
            } else {
                if (ns) nr = ncclSend(d_out + so, sizeof(PointRec) * (size_t)ns, ncclChar, k, s->comm, st);
                if (nr == ncclSuccess && nrv) nr = ncclRecv(d_in + ro, sizeof(PointRec) * (size_t)nrv, ncclChar, k, s->comm, st);
            }
            so += ns; ro += nrv;
        }
    }
    const ncclResult_t ne = ncclGroupEnd();
    if (nr != ncclSuccess || ne != ncclSuccess) return bail(sfail(PCT_ERR_HIP, "slab exchange failed: %s", ncclGetErrorString(nr != ncclSuccess ? nr : ne)));
    if (hipStreamSynchronize(st) != hipSuccess) return bail(sfail(PCT_ERR_HIP, "slab exchange: stream failed"));
    stt = route_load_slab(r, d_in, n_recv);
    if (stt != PCT_OK) return bail(stt);
    *out = r;
    return PCT_OK;
}

int pct_shard_route_nn_dev(pct_route *r, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, void *stream)
{
    if (!r || !r->s || r->s->local || Q < 0 || (Q > 0 && (!d_q || !d_idx || !d_d2))) return sfail(PCT_ERR_INVALID, "bad route_nn_dev arguments");
    if (Q == 0) return PCT_OK;
    pct_shard *s = r->s;
    hipStream_t st = static_cast<hipStream_t>(stream);
    PCTCHK(phase_owner(r, d_q, Q, st));
    PCTCHK(phase_answer(r, st));
    // all-gather of the owned slices: every rank knows every share (the same queries give the same counts everywhere)
    NCCLCHK(ncclGroupStart());
    {
        int64_t ro = 0;
        ncclResult_t nr = ncclSuccess;
        for (int k = 0; k < r->world && nr == ncclSuccess; k++) {
            const int64_t nk = r->h_counts[k];
            if (k == r->rank) {
                if (nk && hipMemcpyAsync(r->d_recv + ro, r->d_send, sizeof(RouteAnswer) * (size_t)nk, hipMemcpyDeviceToDevice, st) != hipSuccess) nr = ncclInternalError;
            } else {
                if (r->mine) nr = ncclSend(r->d_send, sizeof(RouteAnswer) * (size_t)r->mine, ncclChar, k, s->comm, st);
                if (nr == ncclSuccess && nk) nr = ncclRecv(r->d_recv + ro, sizeof(RouteAnswer) * (size_t)nk, ncclChar, k, s->comm, st);
            }
            ro += nk;
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr != ncclSuccess || ne != ncclSuccess) return sfail(PCT_ERR_HIP, "answer exchange failed: %s", ncclGetErrorString(nr != ncclSuccess ? nr : ne));
    }
    PCTCHK(phase_scatter(r, Q, d_idx, d_d2, st));
    if (r->flagged) {                              // second round: everybody answers the uncertified queries, merged as in the index-range form
        r->st_uncert += (uint64_t)r->flagged;
        PCTCHK(phase_second_local(r, d_q, st));
        NCCLCHK(ncclAllReduce(r->d_fd2, r->d_fbest, (size_t)r->flagged, ncclDouble, ncclMin, s->comm, st));
        PCTCHK(pct_merge_mask_dev(r->d_fd2, r->d_fbest, r->d_fidx, r->d_fcand, r->flagged, st));
        NCCLCHK(ncclAllReduce(r->d_fcand, r->d_fcand, (size_t)r->flagged, ncclInt32, ncclMin, s->comm, st));
        PCTCHK(phase_second_finish(r, d_idx, d_d2, st));
    }
    return PCT_OK;
}

int pct_shard_route_stats(const pct_route *r, int64_t *slab_points, uint64_t *owned, uint64_t *uncertified, uint64_t *batches)
{
    if (!r) return sfail(PCT_ERR_INVALID, "null route");
    if (slab_points) *slab_points = r->n_slab;
    if (owned) *owned = r->st_owned;
    if (uncertified) *uncertified = r->st_uncert;
    if (batches) *batches = r->st_batches;
    return PCT_OK;
}

int pct_shard_route_destroy(pct_route *r) { route_free(r); return PCT_OK; }

// ---- several ranks in ONE process (a rehearsal of W ranks on the process's one device, or the building block of a single-process
// multi-slab deployment): the same phases, the bytes are moved between the ranks' buffers by plain copies ----
int pct_shard_local_world(int world, pct_shard **out)
{
    if (!out || world < 1 || world > kRouteMaxWorld) return sfail(PCT_ERR_INVALID, "bad local_world arguments");
    for (int k = 0; k < world; k++) {
        pct_shard *s = new (std::nothrow) pct_shard();
        if (!s) { for (int j = 0; j < k; j++) delete out[j]; return sfail(PCT_ERR_ALLOC, "host allocation failed"); }
        s->rank = k; s->world = world; s->local = true;
        out[k] = s;
    }
    return PCT_OK;
}

int pct_shard_route_build_world(pct_shard *const *ranks, int world, const void *const *local_points, const int64_t *n_local, int64_t stride_bytes,
                                const int64_t *index_begin, double halo_spacings, pct_route **out)
{
    if (!ranks || !local_points || !n_local || !index_begin || !out || world < 1 || world > kRouteMaxWorld || stride_bytes < 12 || !(halo_spacings >= 0))
        return sfail(PCT_ERR_INVALID, "bad route_build_world arguments");
    for (int k = 0; k < world; k++) if (!ranks[k] || !ranks[k]->local || ranks[k]->world != world || ranks[k]->rank != k) return sfail(PCT_ERR_INVALID, "ranks must come from pct_shard_local_world, in order");
    PCTCHK(pct_init(0) == PCT_OK ? PCT_OK : PCT_ERR_NO_DEVICE);
    LocalStats G{};
    for (int k = 0; k < 3; k++) { G.lo[k] = std::numeric_limits<double>::infinity(); G.hi[k] = -std::numeric_limits<double>::infinity(); }
    for (int k = 0; k < world; k++) {
        const LocalStats L = local_stats(static_cast<const unsigned char *>(local_points[k]), n_local[k], stride_bytes);
        for (int d = 0; d < 3; d++) { G.lo[d] = std::min(G.lo[d], L.lo[d]); G.hi[d] = std::max(G.hi[d], L.hi[d]); }
        G.n += L.n;
    }
    if ((int64_t)G.n >= 0x7FFFFFFFll) return sfail(PCT_ERR_INVALID, "global indices travel as 32 bits");
    double cuts[kRouteMaxWorld + 1];
    Layout Y{};
    if (G.n > 0) {
        Y = layout_from(G);
        std::vector<double> h(kRouteBins, 0.0), hk(kRouteBins);
        for (int k = 0; k < world; k++) {
            local_histogram(static_cast<const unsigned char *>(local_points[k]), n_local[k], stride_bytes, Y, hk.data());
            for (int b = 0; b < kRouteBins; b++) h[(size_t)b] += hk[(size_t)b];
        }
        cuts_from(Y, h.data(), world, cuts);
    } else {
        cuts[0] = -std::numeric_limits<double>::infinity();
        for (int k = 1; k <= world; k++) cuts[k] = std::numeric_limits<double>::infinity();
    }
    for (int k = 0; k < world; k++) out[k] = nullptr;
    for (int dst = 0; dst < world; dst++) {
        pct_route *r = route_new(ranks[dst], Y, cuts, halo_spacings);
        if (!r) return sfail(PCT_ERR_ALLOC, "host allocation failed");
        out[dst] = r;
        std::vector<PointRec> recs;                      // source ranks in order: ascending global index
        for (int src = 0; src < world && G.n > 0; src++)
            select_for(static_cast<const unsigned char *>(local_points[src]), n_local[src], stride_bytes, index_begin[src], r->axis, r->cuts[dst] - r->halo, r->cuts[dst + 1] + r->halo, recs);
        PointRec *d_in = nullptr;
        if (grow(&d_in, (int64_t)recs.size())) return PCT_ERR_ALLOC;
        int stt = PCT_OK;
        if (!recs.empty() && hipMemcpy(d_in, recs.data(), sizeof(PointRec) * recs.size(), hipMemcpyHostToDevice) != hipSuccess) stt = sfail(PCT_ERR_HIP, "hipMemcpy failed");
        if (stt == PCT_OK) stt = route_load_slab(r, d_in, (int64_t)recs.size());
        (void)hipFree(d_in);
        if (stt != PCT_OK) return stt;
    }
    return PCT_OK;
}

// every rank gets the whole answer in its own d_idx[k] / d_d2[k] (the arrays may be the same for all ranks: they receive identical data)
int pct_shard_route_nn_world(pct_route *const *routes, int world, const float *d_q, int64_t Q, uint32_t *const *d_idx, double *const *d_d2, void *stream)
{
    if (!routes || !d_idx || !d_d2 || world < 1 || Q < 0 || (Q > 0 && !d_q)) return sfail(PCT_ERR_INVALID, "bad route_nn_world arguments");
    if (Q == 0) return PCT_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int k = 0; k < world; k++) { if (!routes[k] || !routes[k]->s->local) return sfail(PCT_ERR_INVALID, "not a local world"); PCTCHK(phase_owner(routes[k], d_q, Q, st)); }
    for (int k = 0; k < world; k++) PCTCHK(phase_answer(routes[k], st));
    for (int dst = 0; dst < world; dst++) {              // the all-gather of the owned slices, by copies
        int64_t ro = 0;
        for (int src = 0; src < world; src++) {
            const int64_t n = routes[src]->mine;
            if (n) HIPCHK(hipMemcpyAsync(routes[dst]->d_recv + ro, routes[src]->d_send, sizeof(RouteAnswer) * (size_t)n, hipMemcpyDeviceToDevice, st));
            ro += n;
        }
        if (ro != Q) return sfail(PCT_ERR_INTERNAL, "owned shares add up to %lld of %lld queries", (long long)ro, (long long)Q);
    }
    for (int k = 0; k < world; k++) PCTCHK(phase_scatter(routes[k], Q, d_idx[k], d_d2[k], st));
    const int64_t f = routes[0]->flagged;
    for (int k = 1; k < world; k++) if (routes[k]->flagged != f) return sfail(PCT_ERR_INTERNAL, "ranks disagree on the uncertified set");
    if (f) {
        for (int k = 0; k < world; k++) { routes[k]->st_uncert += (uint64_t)f; PCTCHK(phase_second_local(routes[k], d_q, st)); }
        HIPCHK(hipStreamSynchronize(st));
        // all_reduce(min) on d2, then on the indices offered where the local d2 is the minimum: on the host for this small set
        std::vector<double> best((size_t)f, std::numeric_limits<double>::infinity()), d((size_t)f);
        std::vector<int32_t> cand((size_t)f, 0x7FFFFFFF);
        std::vector<uint32_t> ix((size_t)f);
        for (int k = 0; k < world; k++) {
            HIPCHK(hipMemcpy(d.data(), routes[k]->d_fd2, sizeof(double) * (size_t)f, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < f; i++) best[(size_t)i] = std::min(best[(size_t)i], d[(size_t)i]);
        }
        for (int k = 0; k < world; k++) {
            HIPCHK(hipMemcpy(d.data(), routes[k]->d_fd2, sizeof(double) * (size_t)f, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(ix.data(), routes[k]->d_fidx, sizeof(uint32_t) * (size_t)f, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < f; i++)
                if (d[(size_t)i] == best[(size_t)i] && ix[(size_t)i] != 0xFFFFFFFFu) cand[(size_t)i] = std::min(cand[(size_t)i], (int32_t)ix[(size_t)i]);
        }
        for (int k = 0; k < world; k++) {
            HIPCHK(hipMemcpy(routes[k]->d_fbest, best.data(), sizeof(double) * (size_t)f, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(routes[k]->d_fcand, cand.data(), sizeof(int32_t) * (size_t)f, hipMemcpyHostToDevice));
            PCTCHK(phase_second_finish(routes[k], d_idx[k], d_d2[k], st));
        }
    }
    return PCT_OK;
}


// ---- partitioned batches: every rank brings its OWN queries (weak scaling: the whole job answers W x Q queries per step) ----
// The queries travel to the rank that owns their slab and the answers travel back: two variable-sized all-to-alls of 16 bytes per query
// (grouped ncclSend / ncclRecv), nothing is replicated, nobody answers a query twice -- the form whose aggregate rate can grow with the
// number of cards.  Three host synchronises per batch (share sizes, uncertified count, and the counts matrix in between).
int pct_shard_route_nn_partitioned_dev(pct_route *r, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, void *stream)
{
    if (!r || !r->s || r->s->local || Q < 0 || (Q > 0 && (!d_q || !d_idx || !d_d2))) return sfail(PCT_ERR_INVALID, "bad route_nn_partitioned_dev arguments");
    pct_shard *s = r->s;
    const int W = r->world, me = r->rank;
    hipStream_t st = static_cast<hipStream_t>(stream);
    PCTCHK(part_phase_count(r, d_q, Q, st));
    // everybody learns everybody's shares
    std::vector<uint32_t> mat((size_t)W * W);
    HIPCHK(hipMemcpyAsync(r->d_pmat + (size_t)W * W, r->p_send, sizeof(uint32_t) * (size_t)W, hipMemcpyHostToDevice, st));
    NCCLCHK(ncclAllGather(r->d_pmat + (size_t)W * W, r->d_pmat, (size_t)W, ncclUint32, s->comm, st));
    HIPCHK(hipMemcpyAsync(mat.data(), r->d_pmat, sizeof(uint32_t) * (size_t)W * W, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    PCTCHK(part_phase_partition(r, d_q, Q, mat.data(), st));
    auto all_to_all = [&](auto *send, const uint32_t *send_n, auto *recv, auto recv_n, size_t elem_bytes) -> int {      // segment k of `send` -> rank k
        ncclResult_t nr = ncclGroupStart();
        int64_t so = 0, ro = 0;
        for (int k = 0; k < W && nr == ncclSuccess; k++) {
            const int64_t ns = send_n[k], nv = recv_n(k);
            if (k == me) {
                if (ns && hipMemcpyAsync(reinterpret_cast<unsigned char *>(recv) + ro * elem_bytes, reinterpret_cast<const unsigned char *>(send) + so * elem_bytes, (size_t)ns * elem_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                    nr = ncclInternalError;
            } else {
                if (ns) nr = ncclSend(reinterpret_cast<const unsigned char *>(send) + so * elem_bytes, (size_t)ns * elem_bytes, ncclChar, k, s->comm, st);
                if (nr == ncclSuccess && nv) nr = ncclRecv(reinterpret_cast<unsigned char *>(recv) + ro * elem_bytes, (size_t)nv * elem_bytes, ncclChar, k, s->comm, st);
            }
            so += ns; ro += nv;
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr != ncclSuccess || ne != ncclSuccess) return sfail(PCT_ERR_HIP, "all-to-all failed: %s", ncclGetErrorString(nr != ncclSuccess ? nr : ne));
        return PCT_OK;
    };
    const auto from = [&](int k) { return (int64_t)mat[(size_t)k * W + me]; };          // what rank k routes to me
    uint32_t back_n[kRouteMaxWorld];
    for (int k = 0; k < W; k++) back_n[k] = mat[(size_t)k * W + me];
    PCTCHK(all_to_all(r->d_sq_xyz, r->p_send, r->d_rq_xyz, from, 12));
    PCTCHK(all_to_all(r->d_sq_slot, r->p_send, r->d_rq_slot, from, 4));
    PCTCHK(part_phase_answer(r, st));
    PCTCHK(all_to_all(r->d_ans_out, back_n, r->d_ans_in, [&](int k) { return (int64_t)r->p_send[k]; }, sizeof(RouteAnswer)));     // the answers go home
    PCTCHK(part_phase_scatter(r, Q, d_idx, d_d2, st));
    // second round: all uncertified queries of all ranks, answered by everybody
    uint32_t fmine = (uint32_t)r->flagged;
    std::vector<uint32_t> fall((size_t)W);
    HIPCHK(hipMemcpyAsync(r->d_pmat + (size_t)W * W, &fmine, sizeof fmine, hipMemcpyHostToDevice, st));
    NCCLCHK(ncclAllGather(r->d_pmat + (size_t)W * W, r->d_pmat, 1, ncclUint32, s->comm, st));
    HIPCHK(hipMemcpyAsync(fall.data(), r->d_pmat, sizeof(uint32_t) * (size_t)W, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int64_t F = 0, my_off = 0;
    for (int k = 0; k < W; k++) { if (k == me) my_off = F; F += fall[(size_t)k]; }
    if (F == 0) return PCT_OK;
    PCTCHK(part_reserve_flag(r, F));
    PCTCHK(pct_route_gather_queries_dev(d_q, r->d_pflag_ids, r->flagged, r->d_pfq_mine, st));
    {
        ncclResult_t nr = ncclGroupStart();
        int64_t ro = 0;
        for (int k = 0; k < W && nr == ncclSuccess; k++) {
            const int64_t nk = fall[(size_t)k];
            if (k == me) {
                if (nk && hipMemcpyAsync(r->d_pfq_all + 3 * ro, r->d_pfq_mine, sizeof(float) * 3 * (size_t)nk, hipMemcpyDeviceToDevice, st) != hipSuccess) nr = ncclInternalError;
            } else {
                if (r->flagged) nr = ncclSend(r->d_pfq_mine, sizeof(float) * 3 * (size_t)r->flagged, ncclChar, k, s->comm, st);
                if (nr == ncclSuccess && nk) nr = ncclRecv(r->d_pfq_all + 3 * ro, sizeof(float) * 3 * (size_t)nk, ncclChar, k, s->comm, st);
            }
            ro += nk;
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr != ncclSuccess || ne != ncclSuccess) return sfail(PCT_ERR_HIP, "second-round gather failed: %s", ncclGetErrorString(nr != ncclSuccess ? nr : ne));
    }
    PCTCHK(part_phase_second_answer(r, F, st));
    NCCLCHK(ncclAllReduce(r->d_pfd2, r->d_pfbest, (size_t)F, ncclDouble, ncclMin, s->comm, st));
    PCTCHK(pct_merge_mask_dev(r->d_pfd2, r->d_pfbest, r->d_pfidx, r->d_pfcand, F, st));
    NCCLCHK(ncclAllReduce(r->d_pfcand, r->d_pfcand, (size_t)F, ncclInt32, ncclMin, s->comm, st));
    PCTCHK(part_phase_second_finish(r, F, my_off, d_idx, d_d2, st));
    return PCT_OK;
}

// the same with every rank in this process: d_q[k] / Q[k] = rank k's own batch, d_idx[k] / d_d2[k] = its answers
int pct_shard_route_nn_partitioned_world(pct_route *const *routes, int world, const float *const *d_q, const int64_t *Q, uint32_t *const *d_idx, double *const *d_d2, void *stream)
{
    if (!routes || !d_q || !Q || !d_idx || !d_d2 || world < 1 || world > kRouteMaxWorld) return sfail(PCT_ERR_INVALID, "bad route_nn_partitioned_world arguments");
    const int W = world;
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::vector<uint32_t> mat((size_t)W * W);
    for (int k = 0; k < W; k++) {
        if (!routes[k] || !routes[k]->s->local) return sfail(PCT_ERR_INVALID, "not a local world");
        PCTCHK(part_phase_count(routes[k], d_q[k], Q[k], st));
        for (int d = 0; d < W; d++) mat[(size_t)k * W + d] = routes[k]->p_send[d];
    }
    for (int k = 0; k < W; k++) PCTCHK(part_phase_partition(routes[k], d_q[k], Q[k], mat.data(), st));
    for (int dst = 0; dst < W; dst++) {                  // queries to their owners
        int64_t ro = 0;
        for (int src = 0; src < W; src++) {
            const int64_t n = mat[(size_t)src * W + dst];
            int64_t so = 0;
            for (int j = 0; j < dst; j++) so += mat[(size_t)src * W + j];
            if (n) {
                HIPCHK(hipMemcpyAsync(routes[dst]->d_rq_xyz + 3 * ro, routes[src]->d_sq_xyz + 3 * so, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(routes[dst]->d_rq_slot + ro, routes[src]->d_sq_slot + so, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
            }
            ro += n;
        }
    }
    for (int k = 0; k < W; k++) PCTCHK(part_phase_answer(routes[k], st));
    for (int org = 0; org < W; org++) {                  // answers home: segment of rank `own`'s output that came from `org`
        int64_t ro = 0;
        for (int own = 0; own < W; own++) {
            const int64_t n = mat[(size_t)org * W + own];
            int64_t so = 0;
            for (int j = 0; j < org; j++) so += mat[(size_t)j * W + own];
            if (n) HIPCHK(hipMemcpyAsync(routes[org]->d_ans_in + ro, routes[own]->d_ans_out + so, sizeof(RouteAnswer) * (size_t)n, hipMemcpyDeviceToDevice, st));
            ro += n;
        }
    }
    int64_t F = 0;
    std::vector<int64_t> foff((size_t)W);
    for (int k = 0; k < W; k++) { PCTCHK(part_phase_scatter(routes[k], Q[k], d_idx[k], d_d2[k], st)); foff[(size_t)k] = F; F += routes[k]->flagged; }
    if (F == 0) return PCT_OK;
    for (int k = 0; k < W; k++) {
        PCTCHK(part_reserve_flag(routes[k], F));
        PCTCHK(pct_route_gather_queries_dev(d_q[k], routes[k]->d_pflag_ids, routes[k]->flagged, routes[k]->d_pfq_mine, st));
    }
    for (int dst = 0; dst < W; dst++)
        for (int src = 0; src < W; src++)
            if (routes[src]->flagged)
                HIPCHK(hipMemcpyAsync(routes[dst]->d_pfq_all + 3 * foff[(size_t)src], routes[src]->d_pfq_mine, sizeof(float) * 3 * (size_t)routes[src]->flagged, hipMemcpyDeviceToDevice, st));
    for (int k = 0; k < W; k++) PCTCHK(part_phase_second_answer(routes[k], F, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<double> best((size_t)F, std::numeric_limits<double>::infinity()), d((size_t)F);
    std::vector<int32_t> cand((size_t)F, 0x7FFFFFFF);
    std::vector<uint32_t> ix((size_t)F);
    for (int k = 0; k < W; k++) {
        HIPCHK(hipMemcpy(d.data(), routes[k]->d_pfd2, sizeof(double) * (size_t)F, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < F; i++) best[(size_t)i] = std::min(best[(size_t)i], d[(size_t)i]);
    }
    for (int k = 0; k < W; k++) {
        HIPCHK(hipMemcpy(d.data(), routes[k]->d_pfd2, sizeof(double) * (size_t)F, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ix.data(), routes[k]->d_pfidx, sizeof(uint32_t) * (size_t)F, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < F; i++)
            if (d[(size_t)i] == best[(size_t)i] && ix[(size_t)i] != 0xFFFFFFFFu) cand[(size_t)i] = std::min(cand[(size_t)i], (int32_t)ix[(size_t)i]);
    }
    for (int k = 0; k < W; k++) {
        HIPCHK(hipMemcpy(routes[k]->d_pfbest, best.data(), sizeof(double) * (size_t)F, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(routes[k]->d_pfcand, cand.data(), sizeof(int32_t) * (size_t)F, hipMemcpyHostToDevice));
        PCTCHK(part_phase_second_finish(routes[k], F, foff[(size_t)k], d_idx[k], d_d2[k], st));
    }
    return PCT_OK;
}

}  // extern "C"
