// engine.hip -- host side of libpct_engine.so: the C ABI of include/pct_engine.h over the
// gfx950 kernels in kernels.hpp.  One process drives one GPU (pct_init selects it); all work
// is queued on one library-owned HIP stream unless a caller passes its own (the *_dev entry
// points), so copies and kernels of a batch stay ordered without host synchronisation.
//
// Data layout in HBM per cloud:
//   x[cap4], y[cap4], z[cap4]   fp32 SoA, insertion order (cap4 = capacity rounded up to 4)
//   sorted[n]                   float4 {x, y, z, bitcast(original index)} in cell order   (grid only)
//   cell_start[ncells + 1]      u32 exclusive prefix of per-cell counts                   (grid only)
//   query workspaces sized by pct_cloud_reserve_queries
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/pct_engine.h"
#include "engine_internal.hpp"
#include "kernels.hpp"
#include "gridbuild.hpp"
#include "pyramid.hpp"
#include "ring.hpp"
#include "brute2.hpp"

using namespace pct;

namespace {

thread_local char g_err[512] = "";
hipStream_t g_stream = nullptr;
int g_device = -1;

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

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

constexpr int kMaxParts = 2048;       // streaming kernel: at most 8 blocks of 256 per CU
constexpr int64_t kPartQueries = 16384;   // the partial buffers hold this many queries (400 MB); a 1 M-query reservation used to
                                          // take 26 GB of HBM for them
constexpr int kBezierCapMax = 4096;

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

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

constexpr int kExpressMaxQ = 1024;        // queries per express (block-per-query) launch
constexpr int64_t kMappedMaxQ = 65536;    // host-buffer batches up to this size travel through host-mapped memory, larger ones by DMA
constexpr int64_t kSmallNNMax = 16384;    // clouds up to this size answer single queries with one-block kernels
constexpr uint32_t kExpressIdsCap = 1u << 16;

template <typename T>
int mapped_alloc(T **host, T **dev, size_t count)
{
    void *h = nullptr, *d = nullptr;
    hipError_t e = hipHostMalloc(&h, std::max<size_t>(count, 1) * sizeof(T), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) { if (h) (void)hipHostFree(h); return fail(PCT_ERR_ALLOC, "hipHostMalloc(mapped, %zu bytes) -> %s", count * sizeof(T), hipGetErrorString(e)); }
    *host = static_cast<T *>(h);
    *dev = static_cast<T *>(d);
    return PCT_OK;
}

}  // namespace

struct pct_cloud {
    int64_t cap = 0, cap4 = 0, count = 0, ring_next = 0;
    int64_t index_base = 0;
    float *x = nullptr, *y = nullptr, *z = nullptr;
    float4 *gb_tmp = nullptr;                   // index build scratch (gridbuild.hpp): slab-ordered records, kept between builds
    size_t gb_tmp_cap = 0;
    uint32_t *gb_small = nullptr;               // per-block slab table + slab totals / cursors / starts
    size_t gb_small_cap = 0;
    // "small" clouds keep their coordinates in host-mapped memory (kernels read them over the bus; the host
    // appends with plain stores and no launch) -- the RRT* node sets of the kd_* drop-in
    bool host_mapped = false;
    float *hx = nullptr, *hy = nullptr, *hz = nullptr;
    // express path: host-mapped result / argument / id buffers
    ExpressOut *h_xout = nullptr, *d_xout = nullptr;
    double *h_xin = nullptr, *d_xin = nullptr, *h_xr = nullptr, *d_xr = nullptr;
    uint32_t *h_xids = nullptr, *d_xids = nullptr;
    uint32_t *h_xseq = nullptr, *d_xseq = nullptr, *d_xcounter = nullptr;     // completion word of the express launches (kernels.hpp ExpressSignal)
    // host-buffer batches of up to kMappedMaxQ queries: queries read from, results exported to, host-mapped memory (no DMA copies)
    float *h_mq = nullptr, *d_mq = nullptr;
    uint32_t *h_mi = nullptr, *d_mi = nullptr;
    double *h_md = nullptr, *d_md = nullptr;
    int64_t mcap = 0;
    unsigned char *h_frame = nullptr, *d_frame = nullptr;       // host-mapped staging of appended sensor frames (ring_append)
    size_t frame_cap = 0;
    unsigned char *h_astage = nullptr, *d_astage = nullptr;     // the library's own staging of copied frames
    size_t astage_cap = 0;
    hipEvent_t ev_mut = nullptr;                                // recorded on the library's stream behind an asynchronous mutation
    bool mut_pending = false;                                   // ... which a call on another stream has to order itself behind
    bool append_pending = false;                                // ring_append returned before its insert kernel finished (ring_host.inc)
    uint32_t xseq = 0;
    // fused RRT* expansion (small clouds = node sets): per-node {x, y, z, radius} as the planner holds them, and the results
    double *h_aux = nullptr, *d_aux = nullptr;
    ExpandOut *h_eout = nullptr, *d_eout = nullptr;
    double *h_bpos = nullptr, *d_bpos = nullptr;        // express Bezier check: sample positions
    unsigned char *d_stage = nullptr;
    size_t stage_bytes = 0;
    float *d_bbox = nullptr;                    // bounding-box partials of the index build (a buffer of their own, not the upload staging)
    GbCheck *d_gbcheck = nullptr, *h_gbcheck = nullptr;     // the build's self-check (gridbuild.hpp): device words + pinned read-back
    GbCheck last_check{};                       // as read back by the last build
    // bounding-box pyramid over the cell index (pyramid.hpp): built for clouds with sparse occupancy
    bool has_pyr = false;
    PyrDesc P{};
    PyrNode *pyr_nodes = nullptr;
    unsigned char *pyr_hint = nullptr;          // start level of the walk per level-0 cell
    size_t pyr_cap = 0, pyr_total = 0, pyr_hint_cap = 0;
    double empty_frac = 0.0;
    bool was_sparse = false;                    // the previous build found most cells empty: this one uses smaller cells
    // grid
    bool has_grid = false;
    GridDesc G{};
    uint32_t *cell_start = nullptr;
    size_t cells_cap = 0;
    float4 *sorted = nullptr;
    size_t sorted_cap = 0;
    uint4 *blocks = nullptr;                 // block table (kernels.hpp block_corner_kernel): 2 x uint4 per lattice corner
    size_t blocks_cap = 0;
    BinDesc B{};
    // optional coarser copies of the index (clouds with sparse occupancy), see kernels.hpp CoarseLevels
    CoarseLevels C{};
    uint32_t *coarse_cell_start[kMaxCoarse] = { nullptr, nullptr, nullptr };
    size_t coarse_cells_cap[kMaxCoarse] = { 0, 0, 0 };
    float4 *coarse_sorted[kMaxCoarse] = { nullptr, nullptr, nullptr };
    size_t coarse_sorted_cap[kMaxCoarse] = { 0, 0, 0 };
    uint32_t *bin_start = nullptr, *bin_fill = nullptr, *bin_tiles = nullptr;   // query binning (sized at grid build)
    size_t bins_cap = 0;
    uint32_t *d_qbin = nullptr, *d_perm = nullptr;                              // sized by reserve_queries
    float4 *d_qsorted = nullptr;
    uint32_t *d_inv = nullptr, *d_sres_idx = nullptr;                           // inverse permutation and sorted-order results
    double *d_sres_d2 = nullptr;
    float4 *d_sorttmp = nullptr;                                                // {x,y,z,id} records of the two-level sort
    uint32_t *d_sortkey = nullptr;                                              // their keys
    uint32_t *d_sort1 = nullptr;                                                // total1 | start1(+1) | fill1 | total1 (second set)
    int sort_phase = 0;                                                         // which set of totals the next batch adds into
    // query workspaces
    int64_t qcap = 0;
    float *d_q = nullptr, *d_r = nullptr;
    double *d_q64 = nullptr, *d_r2 = nullptr, *d_d2 = nullptr, *d_radius = nullptr, *d_pts64 = nullptr;
    uint32_t *d_idx = nullptr, *d_count = nullptr, *d_bound = nullptr;
    uint32_t *d_todo = nullptr;                                 // {count, ticket, slots...}: queries the fp32 pyramid walk leaves to the exact one
    unsigned char *d_skip = nullptr;
    double *d_part_d2 = nullptr;      // per-(query, block) partial minima of the streaming kernels: part_q x kMaxParts entries;
    uint32_t *d_part_idx = nullptr;   // larger batches go through them in slices of part_q queries
    int64_t part_q = 0;
    uint32_t *d_ovf = nullptr;                                 // [0] = number of overflowed candidate lists, [1..] = their queries
    uint32_t *d_cand_count = nullptr, *d_cand_idx = nullptr;   // candidate lists of the brute-force filter: part_q x kCandCap
    double *d_cand_d2 = nullptr;
    // order-preserving crop (lidar): tile counts and the compacted {index, d2, x, y, z} of the last crop
    uint32_t *crop_tile = nullptr, *crop_idx = nullptr;
    double *crop_d2 = nullptr;
    float *crop_x = nullptr, *crop_y = nullptr, *crop_z = nullptr;
    size_t crop_tiles_cap = 0, crop_cap = 0;
    // bezier
    double *d_coef = nullptr, *d_segtime = nullptr;
    int *d_orders = nullptr, *d_nsamples = nullptr;
    long long *d_first_hit = nullptr;
    size_t coef_cap = 0, seg_cap = 0;
    // measurement
    hipEvent_t ev0 = nullptr, ev1 = nullptr;    // around the whole batch (all kernels of one query call)
    hipEvent_t ev2 = nullptr, ev3 = nullptr;    // around the batch's dominant kernel only (aliases of the ring's current pair)
    // the last kDomRing batches' dominant-kernel event pairs, so a caller can time K back-to-back batches without a host
    // sync in between and read every launch's duration afterwards (bench.py's roofline figure)
    static constexpr int kDomRing = 64;
    hipEvent_t dom_ring[2 * kDomRing] = {};
    uint64_t dom_seq = 0;                       // completed (begin + end) pairs
    uint64_t dom_launch = 0;                    // launches of the sampled (index) path since pct_set_timing_stride
    int dom_stride = 1;                         // the index path records every dom_stride-th launch (pct_set_timing_stride)
    hipEvent_t last2 = nullptr, last3 = nullptr;    // the most recent COMPLETE pair (pct_last_kernel_ms)
    bool ev_valid = false, dom_valid = false;
    int timing_level = 1;                       // 0 = no events, 1 = dominant kernel only (default), 2 = + the whole batch
    WorkCounters *d_work = nullptr;
    bool count_work = false;
    bool host_work = false;        // last batch's work is known on the host (streaming kernel)
    uint64_t host_points = 0;
    bool capturing = false;
    // bumped whenever something a captured plan baked in goes away or changes meaning: workspace reallocation, a new point
    // count on a cloud without the ring index, grid build / drop, ring-index (re)configuration
    uint64_t generation = 1;
    // contents version (bumped by every upload / append) and the bounding box last computed for it (brute2.hpp's centred filter)
    uint64_t content_epoch = 1, bbox_epoch = 0;
    float bbox_lo[3] = { 0, 0, 0 }, bbox_hi[3] = { 0, 0, 0 };
    // rolling-map index (ring.hpp): bucket table that appends update in place
    bool ring_on = false, ring_ready = false;
    float ring_cell_req = 0.0f;
    uint32_t ring_K = 32;                    // records per bucket of the rolling-map index (ring.hpp kRingK .. kRingKMax; grows when the overflow queue fills)
    float ring_extent_req[3] = { 0.0f, 0.0f, 0.0f };
    RingDesc R{};
    size_t ring_cells = 0;
    uint2 *ring_ht = nullptr;
    float4 *ring_slots = nullptr, *ring_ovf = nullptr;
    uint32_t *ring_where = nullptr;
    RingState *ring_st = nullptr;
    uint32_t *h_ring_status = nullptr, *d_ring_status = nullptr;      // host-mapped {overrun flag, overflow-queue length}
    int64_t ring_cfg_count = 0;                                       // points in the window when the table was last sized
    int ring_appends_since_cfg = 0;
    struct ReplanCtx *rp = nullptr;              // lazily created context of the un-captured fused planner batch
};

struct ReplanCtx;
void replan_ctx_free(ReplanCtx *x);

struct pct_plan {
    int kind = 0;                                // 0 = NN batch, 1 = fused replan batch
    uint64_t generation = 0;                     // the cloud's generation the graph was captured at
    int algo = 0;
    ReplanCtx *rx = nullptr;
    double run_us[4] = { 0, 0, 0, 0 };
    pct_cloud *c = nullptr;
    int64_t Q = 0;
    float *h_q = nullptr;
    uint32_t *h_idx = nullptr;
    double *h_d2 = nullptr;
    float *d_q = nullptr;
    uint32_t *d_idx = nullptr;
    double *d_d2 = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

namespace {

bool poll_results()
{
    static const bool v = [] { const char *e = std::getenv("PCT_POLL_RESULTS"); return e ? std::atoi(e) != 0 : true; }();
    return v;
}

// completion word for the next express launch on this cloud (seq = nullptr when polling is off: express_wait then synchronises)
ExpressSignal next_signal(pct_cloud *c)
{
    ++c->xseq;
    return ExpressSignal{ c->d_xcounter, poll_results() ? c->d_xseq : nullptr, c->xseq };
}

// wait for the express launch that carried next_signal(): spin on the host-mapped word its last block stores (a stream
// synchronise costs ~20 us of host time more), falling back to the stream after ~2 s of spinning
int express_wait(pct_cloud *c)
{
    if (poll_results()) {
        const volatile uint32_t *seq = c->h_xseq;
        for (long spins = 0; spins < 200000000l; spins++) {
            if (*seq == c->xseq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return PCT_OK; }
            __builtin_ia32_pause();
        }
    }
    HIPCHK(hipStreamSynchronize(g_stream));
    return PCT_OK;
}

// host-mapped query / result buffers of the mid-size host-buffer batches (grow-only, power of two)
int ensure_mapped_io(pct_cloud *c, int64_t Q)
{
    if (Q <= c->mcap) return PCT_OK;
    int64_t cap = 4096;
    while (cap < Q) cap <<= 1;
    if (c->h_mq) (void)hipHostFree(c->h_mq);
    if (c->h_mi) (void)hipHostFree(c->h_mi);
    if (c->h_md) (void)hipHostFree(c->h_md);
    c->h_mq = nullptr; c->h_mi = nullptr; c->h_md = nullptr; c->mcap = 0;
    PCTCHK(mapped_alloc(&c->h_mq, &c->d_mq, (size_t)(4 * cap)));          // 3 floats per query + a radius
    PCTCHK(mapped_alloc(&c->h_mi, &c->d_mi, (size_t)cap));
    PCTCHK(mapped_alloc(&c->h_md, &c->d_md, (size_t)cap));
    c->mcap = cap;
    return PCT_OK;
}

bool mapped_io_on()
{
    static const bool on = [] { const char *e = std::getenv("PCT_MAPPED_IO"); return e ? std::atoi(e) != 0 : true; }();
    return on && poll_results();
}

// Rolling-map appends and index builds return once their launches are queued on the library's stream.  Calls on that stream are
// ordered behind them by construction; a *_dev call on the caller's own stream waits on this event (until it has been seen complete).
int note_mutation(pct_cloud *c)
{
    if (!c->ev_mut) HIPCHK(hipEventCreateWithFlags(&c->ev_mut, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->ev_mut, g_stream));
    c->mut_pending = true;
    return PCT_OK;
}

int order_after_mutations(pct_cloud *c, hipStream_t s)
{
    if (!c->mut_pending || s == g_stream) return PCT_OK;
    const hipError_t q = hipEventQuery(c->ev_mut);
    if (q == hipSuccess) { c->mut_pending = false; return PCT_OK; }
    if (q != hipErrorNotReady) return fail(PCT_ERR_HIP, "hipEventQuery -> %s", hipGetErrorString(q));
    HIPCHK(hipStreamWaitEvent(s, c->ev_mut, 0));
    return PCT_OK;
}

int require_init()
{
    if (g_device < 0) return pct_init(0);
    return PCT_OK;
}

int ensure_stage(pct_cloud *c, size_t bytes)
{
    if (bytes <= c->stage_bytes) return PCT_OK;
    dev_free(c->d_stage);
    c->stage_bytes = 0;
    PCTCHK(dev_alloc(&c->d_stage, bytes));
    c->stage_bytes = bytes;
    return PCT_OK;
}

// Close a build: read the self-check back behind the last launch, synchronise, compare.  The ids of the records must sum and xor
// to those of 0..n-1 and no record may sit outside its slab / cell; anything else is a wrong index and is reported, not served.
int finish_build(pct_cloud *c, const GridDesc &G, hipError_t e, hipStream_t s)
{
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_gbcheck, c->d_gbcheck, sizeof(GbCheck) * kGbCheckSlots, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(PCT_ERR_HIP, "grid build failed: %s", hipGetErrorString(e));
    GbCheck k{};
    for (int i = 0; i < kGbCheckSlots; i++) {
        k.sum_ids += c->h_gbcheck[i].sum_ids; k.xor_ids ^= c->h_gbcheck[i].xor_ids;
        k.misplaced += c->h_gbcheck[i].misplaced; k.empty_cells += c->h_gbcheck[i].empty_cells;
    }
    c->last_check = k;
    const uint64_t n = (uint64_t)c->count;
    const uint64_t want_sum = n * (n - 1) / 2;
    uint32_t want_xor = 0;                       // xor of 0 .. n-1
    switch ((n - 1) & 3u) { case 0: want_xor = (uint32_t)(n - 1); break; case 1: want_xor = 1u; break; case 2: want_xor = (uint32_t)n; break; default: want_xor = 0u; }
    if (k.sum_ids != want_sum || k.xor_ids != want_xor || k.misplaced != 0 || k.empty_cells > G.ncells)
        return fail(PCT_ERR_INTERNAL, "index build self-check failed: ids sum %llu (want %llu) xor %08x (want %08x), %u misplaced records, %u of %u cells empty",
                    (unsigned long long)k.sum_ids, (unsigned long long)want_sum, k.xor_ids, want_xor, k.misplaced, k.empty_cells, G.ncells);
    return PCT_OK;
}

// counting sort of the cloud into the cells of G: cell_start (ncells+1 prefix) and the float4 {x,y,z,index} copy in cell order
int sort_into_cells(pct_cloud *c, const GridDesc &G, uint32_t **cell_start, size_t *cells_cap, float4 **sorted, size_t *sorted_cap)
{
    hipStream_t s = g_stream;
    const int64_t n = c->count;
    const uint64_t ncells = G.ncells;
    if (ncells + 1 > *cells_cap) {
        dev_free(*cell_start);
        *cells_cap = 0;
        PCTCHK(dev_alloc(cell_start, ncells + 1));
        *cells_cap = ncells + 1;
    }
    if ((size_t)n > *sorted_cap) {
        dev_free(*sorted);
        *sorted_cap = 0;
        PCTCHK(dev_alloc(sorted, (size_t)n + kGridPad));         // + the inert records behind the last one (gb_pad_kernel)
        *sorted_cap = (size_t)n;
    }
#ifdef PCT_AB_OPEN_STAGE0
    gb_pad_kernel<<<1, 64, 0, s>>>(*sorted + n);                  // only the open-ended stage 0 reads behind the last record
#endif
    // ---- two-level counting sort on LDS histograms (gridbuild.hpp): no device-scope atomic per point ----
    static const bool lds_build = [] { const char *e = std::getenv("PCT_LDS_GRID_BUILD"); return e ? std::atoi(e) != 0 : true; }();
    // slabs of 2^s1 consecutive cells, sized for ~2-6 k points each (level 2 then holds a whole slab in LDS), at most kGbMaxSlabs;
    // more, smaller slabs make level 1 slower (more open write streams, more LDS per block) faster than they help level 2
    const uint64_t want_slabs = std::min<uint64_t>((uint64_t)kGbMaxSlabs, std::max<uint64_t>(64, (uint64_t)n / 2048));
    int s1 = 0;
    while (((ncells + (1ull << s1) - 1) >> s1) > want_slabs) s1++;
    if (lds_build && n >= 4096 && (1u << s1) <= (uint32_t)kGbMaxSlabCells && (uint64_t)n < 0xFFFFFFF0ull) {
        GbDesc D{};
        D.s1 = s1;
        D.nslabs = (uint32_t)((ncells + (1ull << s1) - 1) >> s1);
        const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(kGbMaxBlocks, (n + 16383) / 16384));
        D.chunk = (uint32_t)((((n + nblk - 1) / nblk) + 3) & ~3ll);
        const int blocks = (int)((n + D.chunk - 1) / D.chunk);
        // level-2 block size by the mean slab: a block holds up to 8 records per thread in LDS (larger slabs stream through twice)
        static const int stage_on = [] { const char *e = std::getenv("PCT_GB_STAGE"); return e ? std::atoi(e) : 1; }();
        const double mean_slab = (double)n / D.nslabs;
        const int cthreads = mean_slab <= 1400 ? 256 : mean_slab <= 3000 ? 512 : 1024;
        const size_t lds1 = sizeof(uint32_t) * ((size_t)D.nslabs + 1);
        const size_t lds_cnt = sizeof(uint32_t) * ((((size_t)1 << s1) + 1 + 3) & ~(size_t)3);
        const uint32_t stage_cap = stage_on ? (uint32_t)std::min<size_t>((size_t)kGbStagePerThread * cthreads, (150 * 1024 - lds_cnt) / sizeof(float4)) : 0u;
        const size_t lds2 = lds_cnt + sizeof(float4) * stage_cap;
        static bool attr = false;
        if (!attr) {      // more than the default 64 KiB of LDS per block (gfx950: 160 KiB per CU)
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gb_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gb_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gb_cells_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gb_cells_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(gb_cells_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr = true;
        }
        // scratch kept with the cloud (a rebuild per sensor frame must not pay hipMalloc / hipFree): records + table + slab counters
        if ((size_t)n > c->gb_tmp_cap) {
            dev_free(c->gb_tmp);
            c->gb_tmp_cap = 0;
            PCTCHK(dev_alloc(&c->gb_tmp, (size_t)n));
            c->gb_tmp_cap = (size_t)n;
        }
        const size_t small_need = (size_t)blocks * D.nslabs + 3 * (size_t)D.nslabs + 8;
        if (small_need > c->gb_small_cap) {
            dev_free(c->gb_small);
            c->gb_small_cap = 0;
            PCTCHK(dev_alloc(&c->gb_small, small_need));
            c->gb_small_cap = small_need;
        }
        // level 1 in two passes of fan-out <= 128 when there are many slabs (gridbuild.hpp): pass A sorts by super-slab into the final
        // array (unused until level 2 writes it), pass B by slab inside every super-slab's region into gb_tmp
        static const bool two_pass_on = [] { const char *e = std::getenv("PCT_GB_TWO_PASS"); return e ? std::atoi(e) != 0 : true; }();
        uint32_t two_pass_min = 4096;                 // read per build: the tests lower it to reach this path with small clouds
        if (const char *e = std::getenv("PCT_GB_TWO_PASS_MIN_SLABS")) two_pass_min = (uint32_t)std::max(2, std::atoi(e));
        if (two_pass_on && D.nslabs >= two_pass_min) {
            Gb2Desc DB{};
            DB.s1 = s1;
            int lg = 0;
            while ((1u << lg) < D.nslabs) lg++;
            DB.sb = std::min(7, (lg + 1) / 2);
            DB.nslabs = D.nslabs;
            DB.nsuper = (D.nslabs + (1u << DB.sb) - 1) >> DB.sb;
            DB.parts = std::max<uint32_t>(1, 512u / DB.nsuper);
            GbDesc DA = D;
            DA.s1 = s1 + DB.sb;
            DA.nslabs = DB.nsuper;
            const size_t nsub = (size_t)1 << DB.sb;
            const size_t need2 = (size_t)blocks * DA.nslabs + 3 * (size_t)DA.nslabs + 8 + (size_t)DB.nsuper * DB.parts * nsub + 3 * (size_t)D.nslabs + 8;
            if (need2 > c->gb_small_cap) {
                dev_free(c->gb_small);
                c->gb_small_cap = 0;
                PCTCHK(dev_alloc(&c->gb_small, need2));
                c->gb_small_cap = need2;
            }
            uint32_t *tableA = c->gb_small, *super_total = tableA + (size_t)blocks * DA.nslabs, *super_cursor = super_total + DA.nslabs,
                     *super_start = super_cursor + DA.nslabs, *table2 = super_start + DA.nslabs + 1,
                     *slab_total = table2 + (size_t)DB.nsuper * DB.parts * nsub, *slab_cursor = slab_total + D.nslabs, *slab_start = slab_cursor + D.nslabs;
            hipError_t e = hipSuccess;
            gb_zero_kernel<<<ceil_div(2 * (int64_t)DA.nslabs, 256), 256, 0, s>>>(super_total, 2 * DA.nslabs, nullptr);
            gb_zero_kernel<<<ceil_div(2 * (int64_t)D.nslabs, 256), 256, 0, s>>>(slab_total, 2 * D.nslabs, c->d_gbcheck);
            {
                const size_t ldsA = sizeof(uint32_t) * ((size_t)DA.nslabs + 1);
                gb_hist_kernel<<<blocks, kGbThreads, ldsA, s>>>(G, DA, c->x, c->y, c->z, (uint32_t)n, tableA, super_total);
                gb_scatter_kernel<<<blocks, kGbThreads, ldsA, s>>>(G, DA, c->x, c->y, c->z, (uint32_t)n, tableA, super_total, super_cursor, super_start, *sorted);
                const dim3 g2(DB.parts, DB.nsuper);
                gb_hist2_kernel<<<g2, kGbThreads, 0, s>>>(G, DB, super_start, *sorted, table2, slab_total);
                gb_scatter2_kernel<<<g2, kGbThreads, 0, s>>>(G, DB, super_start, *sorted, table2, slab_total, slab_cursor, slab_start, (uint32_t)n, c->gb_tmp);
                if (cthreads == 256) gb_cells_kernel<256><<<(int)D.nslabs, 256, lds2, s>>>(G, D, slab_start, c->gb_tmp, (uint32_t)n, stage_cap, *cell_start, *sorted, c->d_gbcheck);
                else if (cthreads == 512) gb_cells_kernel<512><<<(int)D.nslabs, 512, lds2, s>>>(G, D, slab_start, c->gb_tmp, (uint32_t)n, stage_cap, *cell_start, *sorted, c->d_gbcheck);
                else gb_cells_kernel<1024><<<(int)D.nslabs, 1024, lds2, s>>>(G, D, slab_start, c->gb_tmp, (uint32_t)n, stage_cap, *cell_start, *sorted, c->d_gbcheck);
                e = hipGetLastError();
            }
            return finish_build(c, G, e, s);
        }
        uint32_t *table = c->gb_small, *slab_total = table + (size_t)blocks * D.nslabs, *slab_cursor = slab_total + D.nslabs,
                 *slab_start = slab_cursor + D.nslabs;
        hipError_t e = hipSuccess;
        gb_zero_kernel<<<ceil_div(2 * (int64_t)D.nslabs, 256), 256, 0, s>>>(slab_total, 2 * D.nslabs, c->d_gbcheck);
        {
            gb_hist_kernel<<<blocks, kGbThreads, lds1, s>>>(G, D, c->x, c->y, c->z, (uint32_t)n, table, slab_total);
            gb_scatter_kernel<<<blocks, kGbThreads, lds1, s>>>(G, D, c->x, c->y, c->z, (uint32_t)n, table, slab_total, slab_cursor, slab_start, c->gb_tmp);
            if (cthreads == 256) gb_cells_kernel<256><<<(int)D.nslabs, 256, lds2, s>>>(G, D, slab_start, c->gb_tmp, (uint32_t)n, stage_cap, *cell_start, *sorted, c->d_gbcheck);
            else if (cthreads == 512) gb_cells_kernel<512><<<(int)D.nslabs, 512, lds2, s>>>(G, D, slab_start, c->gb_tmp, (uint32_t)n, stage_cap, *cell_start, *sorted, c->d_gbcheck);
            else gb_cells_kernel<1024><<<(int)D.nslabs, 1024, lds2, s>>>(G, D, slab_start, c->gb_tmp, (uint32_t)n, stage_cap, *cell_start, *sorted, c->d_gbcheck);
            e = hipGetLastError();
        }
        return finish_build(c, G, e, s);
    }
    // ---- small clouds / very fine user-given cells: one device atomic per point and pass ----
    uint32_t *d_cnt = nullptr, *d_pcell = nullptr, *d_tiles = nullptr;
    const uint32_t ntiles = (uint32_t)((ncells + kScanTile - 1) / kScanTile);
    int st = dev_alloc(&d_cnt, ncells);
    if (!st) st = dev_alloc(&d_pcell, (size_t)n);
    if (!st) st = dev_alloc(&d_tiles, ntiles);
    if (st) { dev_free(d_cnt); dev_free(d_pcell); dev_free(d_tiles); return st; }
    hipError_t e = hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * ncells, s);
    const int pblocks = (int)std::min<int64_t>(4096, (n + 255) / 256);
    if (e == hipSuccess) {
        cell_histogram_kernel<<<pblocks, 256, 0, s>>>(G, c->x, c->y, c->z, (uint32_t)n, d_cnt, d_pcell);
        scan_tiles_kernel<<<ntiles, 256, 0, s>>>(d_cnt, (uint32_t)ncells, *cell_start, d_tiles);
        scan_tile_sums_kernel<<<1, 256, 0, s>>>(d_tiles, ntiles);
        scan_add_kernel<<<ceil_div((int64_t)ncells, 256), 256, 0, s>>>(*cell_start, (uint32_t)ncells, d_tiles, (uint32_t)n);
        e = hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * ncells, s);
    }
    if (e == hipSuccess) {
        cell_scatter_kernel<<<pblocks, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)n, d_pcell, *cell_start, d_cnt, *sorted);
        gb_zero_kernel<<<1, 64, 0, s>>>(nullptr, 0, c->d_gbcheck);
        gb_check_kernel<<<(int)std::min<int64_t>(1024, (std::max<int64_t>(n, (int64_t)ncells) + 255) / 256), 256, 0, s>>>(G, *sorted, *cell_start, (uint32_t)n, c->d_gbcheck);
        e = hipGetLastError();
    }
    st = finish_build(c, G, e, s);
    dev_free(d_cnt); dev_free(d_pcell); dev_free(d_tiles);
    return st;
}

void drop_grid(pct_cloud *c)
{
    if (c->has_grid) c->generation++;
    c->has_grid = false;
    c->has_pyr = false;
}

// Bounding-box pyramid over the freshly built cell index (pyramid.hpp).  Built when the index is sparsely occupied -- surfaces,
// clusters, a window much larger than its contents: the clouds on which the shell walk pays for empty space -- or on request
// (PCT_PYRAMID=1 forces it on any cloud: the tests and the soak run the dense fixtures through the walk that way; =0 never).
int build_pyramid(pct_cloud *c, const GridDesc &G)
{
    c->has_pyr = false;
    c->empty_frac = G.ncells ? (double)c->last_check.empty_cells / (double)G.ncells : 0.0;
    int mode = -1;
    if (const char *e = std::getenv("PCT_PYRAMID")) mode = std::atoi(e);
    double min_empty = 0.25;                      // uniform cloud at 6 points per cell: e^-6 = 0.25 % of the cells are empty
    if (const char *e = std::getenv("PCT_PYRAMID_MIN_EMPTY")) min_empty = std::atof(e);
    if (mode == 0 || (mode < 0 && c->empty_frac < min_empty)) { c->was_sparse = false; return PCT_OK; }
    PyrDesc P{};
    int nlev = 1;
    while (pyr_dim(G.gx, nlev - 1) > 2 || pyr_dim(G.gy, nlev - 1) > 2 || pyr_dim(G.gz, nlev - 1) > 2) nlev++;
    if (nlev > kPyrMaxLevels) return fail(PCT_ERR_INTERNAL, "pyramid deeper than %d levels", kPyrMaxLevels);
    P.nlev = nlev;
    // level l = blocks of 8 children per node of level l + 1 (pyramid.hpp): 8 * |grid of level l + 1| slots
    size_t total = 0, nslots[kPyrMaxLevels] = {};
    for (int l = 0; l < nlev; l++) {
        nslots[l] = 8 * (size_t)pyr_dim(G.gx, l + 1) * pyr_dim(G.gy, l + 1) * pyr_dim(G.gz, l + 1);
        P.off[l] = (uint32_t)total;
        total += nslots[l];
    }
    if (total > 0xFFFFFFF0ull) return PCT_OK;      // cannot be addressed with 32-bit slot offsets: stay with the shell walk
    if (total > c->pyr_cap) {
        dev_free(c->pyr_nodes);
        c->pyr_cap = 0;
        PCTCHK(dev_alloc(&c->pyr_nodes, total));
        c->pyr_cap = total;
    }
    hipStream_t s = g_stream;
    pyr_leaf_kernel<<<ceil_div((int64_t)nslots[0], 32), 256, 0, s>>>(G, P, c->sorted, c->cell_start, c->pyr_nodes, (uint32_t)nslots[0]);
    for (int l = 1; l < nlev; l++)
        pyr_up_kernel<<<ceil_div((int64_t)nslots[l], 256), 256, 0, s>>>(G, P, l, c->pyr_nodes, (uint32_t)nslots[l]);
    if ((size_t)G.ncells > c->pyr_hint_cap) {
        dev_free(c->pyr_hint);
        c->pyr_hint_cap = 0;
        PCTCHK(dev_alloc(&c->pyr_hint, (size_t)G.ncells));
        c->pyr_hint_cap = (size_t)G.ncells;
    }
    pyr_hint_kernel<<<ceil_div((int64_t)G.ncells, 256), 256, 0, s>>>(G, P, c->pyr_nodes, c->pyr_hint);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    c->P = P;
    c->pyr_total = total;
    c->has_pyr = true;
    c->was_sparse = c->empty_frac > 0.6;
    return PCT_OK;
}

// host AoS -> device SoA slots [dst0, dst0+n)
int upload_range(pct_cloud *c, const void *pts, int64_t n, int64_t stride, int64_t dst0)
{
    if (n == 0) return PCT_OK;
    if (c->host_mapped) {   // plain host stores; every earlier kernel on this cloud has been waited for
        const unsigned char *src = static_cast<const unsigned char *>(pts);
        for (int64_t i = 0; i < n; i++) {
            const float *p = reinterpret_cast<const float *>(src + i * stride);
            c->hx[dst0 + i] = p[0]; c->hy[dst0 + i] = p[1]; c->hz[dst0 + i] = p[2];
        }
        return PCT_OK;
    }
    PCTCHK(ensure_stage(c, (size_t)n * stride + 64));
    HIPCHK(hipMemcpyAsync(c->d_stage, pts, (size_t)n * stride, hipMemcpyHostToDevice, g_stream));
    if (stride == 12 && (dst0 & 3) == 0 && n >= 4) {
        const uint32_t ng = (uint32_t)(n >> 2);
        deinterleave12_kernel<<<ceil_div(ng, 256), 256, 0, g_stream>>>(reinterpret_cast<const float4 *>(c->d_stage), ng,
                                                                        reinterpret_cast<float4 *>(c->x + dst0),
                                                                        reinterpret_cast<float4 *>(c->y + dst0),
                                                                        reinterpret_cast<float4 *>(c->z + dst0));
        const int64_t done = (int64_t)ng * 4;
        if (done < n)
            deinterleave_kernel<<<1, 256, 0, g_stream>>>(c->d_stage + done * 12, 12, (uint32_t)(n - done), c->x, c->y, c->z,
                                                         (uint32_t)(dst0 + done));
    } else {
        deinterleave_kernel<<<ceil_div(n, 256), 256, 0, g_stream>>>(c->d_stage, (uint32_t)stride, (uint32_t)n, c->x, c->y, c->z,
                                                                    (uint32_t)dst0);
    }
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

// Streaming kernels are grid-stride: at most 4 blocks of 256 threads per CU (16 waves per CU),
// so the whole grid is resident in ONE round whatever the kernel's register count -- a grid of
// 8 blocks per CU ran as 7 + 1 rounds at 66 VGPRs and cost almost 2x.
// Q: the batch's size where the caller knows it (0 = unknown).  One or two queries against a DRAM-resident cloud do better with half the
// waves: 512 blocks (2 waves per SIMD) read 100 M points at 0.78-0.80 of the 8 TB/s peak, 1024 at 0.74-0.77, 2048 at 0.72-0.73
// (scripts/probe_stream.py; Q = 4 and the Infinity-Cache-sized 10 M-point cloud are the other way round or level).
int stream_blocks(int64_t n, int64_t Q = 0)
{
    const char *e = std::getenv("PCT_STREAM_BLOCKS");      // tuning knob for scripts/probe.py, scripts/probe_stream.py
    const int cap = e ? std::max(1, std::min(kMaxParts, std::atoi(e))) : ((Q == 1 || Q == 2) && n >= (1ll << 25) ? 512 : 1024);
    const int64_t groups = std::max<int64_t>(n >> 2, 1);
    return (int)std::min<int64_t>(cap, (groups + 255) / 256);
}

template <int QT>
void launch_nn_stream(pct_cloud *c, const double *d_q64, int blocks, int q0, int qcount, hipStream_t s)
{
    nn_stream_kernel<QT><<<blocks, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, d_q64, q0, qcount, c->d_part_d2,
                                                 c->d_part_idx, blocks);
}

template <int QT>
void launch_count_stream(pct_cloud *c, int blocks, int q0, int qcount, uint32_t *d_count, hipStream_t s)
{
    count_stream_kernel<QT><<<blocks, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, c->d_q64, c->d_r2, q0, qcount, d_count);
}

int pick_tile(int64_t remaining)
{
    if (remaining >= 8) return 8;
    if (remaining > 2) return 4;
    if (remaining == 2) return 2;
    return 1;
}

void begin_timing(pct_cloud *c, hipStream_t s)
{
    c->ev_valid = false;
    if (c->capturing || c->timing_level < 2) return;     // an event pair costs ~5-9 us of a 190 us batch (same-box A/B)
    if (hipEventRecord(c->ev0, s) == hipSuccess) c->ev_valid = true;
}

void end_timing(pct_cloud *c, hipStream_t s)
{
    if (c->capturing || !c->ev_valid) return;
    if (hipEventRecord(c->ev1, s) != hipSuccess) c->ev_valid = false;
}

// events around the dominant kernel of the batch (what rocprofv3's per-kernel average also measures)
bool dom_ext_on()
{
    static const bool on = [] { const char *e = std::getenv("PCT_DOM_EXT_EVENTS"); return e ? std::atoi(e) != 0 : true; }();
    return on;
}

// ext = the caller launches the kernel with hipExtLaunchKernelGGL(start = ev2, stop = ev3) itself (and calls dom_done): nothing is
// recorded here; that path also honours the sampling stride
void dom_begin(pct_cloud *c, hipStream_t s, bool ext = false)
{
    c->dom_valid = false;
    if (c->capturing || c->timing_level < 1) return;
    if (ext && c->dom_stride > 1 && (c->dom_launch++ % (uint64_t)c->dom_stride) != 0) return;
    const int slot = (int)(c->dom_seq % pct_cloud::kDomRing);
    c->ev2 = c->dom_ring[2 * slot];
    c->ev3 = c->dom_ring[2 * slot + 1];
    if (ext && dom_ext_on()) { c->dom_valid = true; return; }
    if (hipEventRecord(c->ev2, s) == hipSuccess) c->dom_valid = true;
}

void dom_done(pct_cloud *c)
{
    c->dom_valid = false;
    c->dom_seq++;
    c->last2 = c->ev2;
    c->last3 = c->ev3;
}

void dom_end(pct_cloud *c, hipStream_t s)
{
    if (c->capturing || !c->dom_valid) return;
    if (hipEventRecord(c->ev3, s) != hipSuccess) c->dom_valid = false;
    else dom_done(c);
}

// streaming NN over the fp64 queries already in c->d_q64: one slice of at most c->part_q queries starting at qoff
int nn_stream_q64_slice(pct_cloud *c, int64_t qoff, int64_t Q, uint32_t *d_idx, double *d_d2, hipStream_t s)
{
    const double *d_q64 = c->d_q64 + 3 * qoff;
    d_idx += qoff;
    d_d2 += qoff;
    const int blocks = stream_blocks(c->count, Q);
    begin_timing(c, s);
    dom_begin(c, s);
    for (int64_t q0 = 0; q0 < Q;) {
        const int qt = pick_tile(Q - q0);
        const int qcount = (int)std::min<int64_t>(qt, Q - q0);
        switch (qt) {
        case 8: launch_nn_stream<8>(c, d_q64, blocks, (int)q0, qcount, s); break;
        case 4: launch_nn_stream<4>(c, d_q64, blocks, (int)q0, qcount, s); break;
        case 2: launch_nn_stream<2>(c, d_q64, blocks, (int)q0, qcount, s); break;
        default: launch_nn_stream<1>(c, d_q64, blocks, (int)q0, qcount, s); break;
        }
        q0 += qcount;
    }
    dom_end(c, s);
    nn_reduce_partials_kernel<<<(int)Q, 256, 0, s>>>(c->d_part_d2, c->d_part_idx, blocks, (uint32_t)c->index_base, d_idx, d_d2);
    end_timing(c, s);
    HIPCHK(hipGetLastError());
    c->host_work = true;                    // the streaming kernel examines every point for every query
    c->host_points = (uint64_t)Q * (uint64_t)c->count;
    return PCT_OK;
}

int nn_stream_q64(pct_cloud *c, int64_t Q, uint32_t *d_idx, double *d_d2, hipStream_t s)
{
    for (int64_t off = 0; off < Q; off += c->part_q) PCTCHK(nn_stream_q64_slice(c, off, std::min<int64_t>(c->part_q, Q - off), d_idx, d_d2, s));
    c->host_points = (uint64_t)Q * (uint64_t)c->count;
    return PCT_OK;
}

bool lds_sort_on()
{
    static const bool v = [] { const char *e = std::getenv("PCT_LDS_SORT"); return e ? std::atoi(e) != 0 : true; }();
    return v;
}

// counting sort of the batch by coarse cell -> c->d_perm (nullptr result = keep arrival order)
// need_perm / need_inv: the 8-lanes-per-query kernels read the sorted records only; the lane-per-query kernels read perm, the
// sorted-result gather reads inv -- 4 B per query of scattered (perm) or streamed (inv) writes that the default path does not pay
int bin_queries(pct_cloud *c, const float *d_q, int64_t Q, hipStream_t s, const uint32_t **perm_out, bool need_perm = true, bool need_inv = false)
{
    *perm_out = nullptr;
    int64_t min_q = 16384;
    if (const char *e = std::getenv("PCT_SORT_MIN_Q")) min_q = std::atoll(e);
    if (Q < min_q) return PCT_OK;
    const BinDesc &B = c->B;
    if (lds_sort_on()) {
        // two-level counting sort on LDS histograms (kernels.hpp)
        int key_shift = 0;
        while ((((uint64_t)B.nbins - 1) >> key_shift) >= (1ull << 20)) key_shift++;
        // two sets of bucket totals used in turn: a batch's histogram pass zeroes the set the NEXT batch will add into (a captured
        // graph replays one set, so it keeps the memset behind its scatter pass instead)
        static const bool pingpong_on = [] { const char *e = std::getenv("PCT_SORT_PINGPONG"); return e ? std::atoi(e) != 0 : true; }();
        const bool pingpong = pingpong_on && !c->capturing;
        uint32_t *total1 = c->d_sort1 + (c->sort_phase ? 3 * kSortBuckets + 8 : 0), *start1 = c->d_sort1 + kSortBuckets, *fill1 = c->d_sort1 + 2 * kSortBuckets + 4;
        uint32_t *total1_next = c->d_sort1 + (c->sort_phase ? 0 : 3 * kSortBuckets + 8);
        // PCT_SORT_LEVELS: 1 (default) = one counting pass into <= 1024 spatial buckets, queries left in arrival order inside a
        // bucket; 2 = a second pass orders every bucket by the remaining key bits.  With the block-first search the finer order
        // no longer pays for its pass (same-box: 0.198-0.204 ms per step with it, 0.168 without).
        static const int levels = [] { const char *e = std::getenv("PCT_SORT_LEVELS"); return e ? std::max(1, std::min(2, std::atoi(e))) : 1; }();
        int lshift = 10;                         // level-1 bucket = key >> 10 (keys < 2^20): finer buckets (key >> 8, >> 9) measured the same
        if (const char *e = std::getenv("PCT_SORT_LSHIFT")) {           // tuning, single-level mode: finer or coarser buckets (always <= 1024 of them)
            lshift = std::max(0, std::atoi(e));
            while (((((uint64_t)B.nbins - 1) >> key_shift) >> lshift) >= (uint64_t)kSortBuckets) lshift++;
        }
        static const int per_block_env = [] { const char *e = std::getenv("PCT_SORT_PER_BLOCK"); return e ? std::min(kSortPerBlock, std::max(1024, std::atoi(e) / 1024 * 1024)) : 0; }();
        // ~128 blocks: small batches want parallelism (64 K queries: 21 us at 1024 per block, 36 us at 8192), large ones
        // want long per-block bucket slices (1 M: 67 us at 8192, 87 us at 1024)
        const uint32_t per_block = per_block_env ? (uint32_t)per_block_env
                                                 : (uint32_t)std::min<int64_t>(kSortPerBlock, std::max<int64_t>(1024, (Q / 128 + 1023) / 1024 * 1024));
        const int nb = ceil_div(Q, (int64_t)per_block);
        // the key array costs 4 B per query written by one pass and read by the next: the scatter pass recomputes keys instead
        static const bool store_keys = [] { const char *e = std::getenv("PCT_SORT_STORE_KEYS"); return e ? std::atoi(e) != 0 : false; }();
        uint32_t *keys = store_keys ? c->d_qbin : nullptr;
        // the histogram pass only adds into the global totals, so it can use shorter slices than the scatter pass (whose writes want
        // long per-block runs); more blocks measured SLOWER though (PCT_SORT_HIST_DIV = 1 / 2 / 4 / 8: 0.165 / 0.167 / 0.169 / 0.179 ms
        // per step, same box), so the default keeps one slice size for both
        static const int hist_div = [] { const char *e = std::getenv("PCT_SORT_HIST_DIV"); return e ? std::max(1, std::min(8, std::atoi(e))) : 1; }();
        const uint32_t hist_per_block = std::max<uint32_t>(1024u, (per_block / (uint32_t)hist_div) / 1024u * 1024u);
        // 16-byte aligned query arrays are fetched four queries (three float4) at a time (kernels.hpp sort_load_items)
        static const bool vec_on = [] { const char *e = std::getenv("PCT_SORT_VEC"); return e ? std::atoi(e) != 0 : true; }();
        const bool vec = vec_on && (reinterpret_cast<uintptr_t>(d_q) & 15u) == 0;
        if (vec) qsort_hist_kernel<true><<<ceil_div(Q, (int64_t)hist_per_block), 1024, 0, s>>>(c->G, B, key_shift, lshift, d_q, (uint32_t)Q, hist_per_block, keys, total1, fill1,
                                                                                             pingpong ? total1_next : nullptr);
        else qsort_hist_kernel<false><<<ceil_div(Q, (int64_t)hist_per_block), 1024, 0, s>>>(c->G, B, key_shift, lshift, d_q, (uint32_t)Q, hist_per_block, keys, total1, fill1,
                                                                                           pingpong ? total1_next : nullptr);
        if (levels == 1) {
            if (vec) qsort_scatter1_kernel<true><<<nb, 1024, 0, s>>>(c->G, B, key_shift, keys, d_q, (uint32_t)Q, per_block, lshift, total1, fill1, start1, c->d_sortkey, c->d_qsorted, need_perm ? c->d_perm : nullptr,
                                                                    need_inv ? c->d_inv : nullptr, 1);
            else qsort_scatter1_kernel<false><<<nb, 1024, 0, s>>>(c->G, B, key_shift, keys, d_q, (uint32_t)Q, per_block, lshift, total1, fill1, start1, c->d_sortkey, c->d_qsorted, need_perm ? c->d_perm : nullptr,
                                                      need_inv ? c->d_inv : nullptr, 1);
            if (pingpong) c->sort_phase ^= 1;
            else HIPCHK(hipMemsetAsync(total1, 0, sizeof(uint32_t) * kSortBuckets, s));   // the fine pass would have re-zeroed it
        } else {
            qsort_scatter1_kernel<false><<<nb, 1024, 0, s>>>(c->G, B, key_shift, keys, d_q, (uint32_t)Q, per_block, lshift, total1, fill1, start1, c->d_sortkey, c->d_sorttmp, nullptr, nullptr, 0);
            qsort_fine_kernel<<<kSortBuckets, kFineThreads, 0, s>>>(c->d_sortkey, c->d_sorttmp, start1, total1, (1u << lshift) - 1u, need_perm ? c->d_perm : nullptr,
                                                                    c->d_qsorted, need_inv ? c->d_inv : nullptr);
        }
        HIPCHK(hipGetLastError());
        *perm_out = c->d_perm;
        return PCT_OK;
    }
    const uint32_t ntiles = (B.nbins + kScanTile - 1) / kScanTile;
    HIPCHK(hipMemsetAsync(c->bin_fill, 0, sizeof(uint32_t) * B.nbins, s));
    query_bin_count_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(c->G, B, d_q, (uint32_t)Q, c->bin_fill, c->d_qbin);
    scan_tiles_kernel<<<ntiles, 256, 0, s>>>(c->bin_fill, B.nbins, c->bin_start, c->bin_tiles);
    scan_tile_sums_kernel<<<1, 256, 0, s>>>(c->bin_tiles, ntiles);
    scan_add_kernel<<<ceil_div(B.nbins, 256), 256, 0, s>>>(c->bin_start, B.nbins, c->bin_tiles, (uint32_t)Q);
    HIPCHK(hipMemsetAsync(c->bin_fill, 0, sizeof(uint32_t) * B.nbins, s));
    query_bin_scatter_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(c->d_qbin, (uint32_t)Q, c->bin_start, c->bin_fill, d_q, c->d_perm, c->d_qsorted);
    HIPCHK(hipGetLastError());
    *perm_out = c->d_perm;
    return PCT_OK;
}

int g_filter_mode = -2;        // -2 = not read yet
int filter_mode()
{
    if (g_filter_mode == -2) { const char *e = std::getenv("PCT_TILE_EXPANDED"); g_filter_mode = e ? std::atoi(e) : -1; }
    return g_filter_mode;
}

int reg_groups()
{
    static const int v = [] { const char *e = std::getenv("PCT_TILE_REG_GROUPS"); const int g = e ? std::atoi(e) : 3; return (g == 2 || g == 4 || g == 6) ? g : 3; }();
    return v;
}

// bounding box of the cloud's current contents, cached per contents version (one reduction + one read-back when stale)
int cloud_bbox_cached(pct_cloud *c)
{
    if (c->bbox_epoch == c->content_epoch) return PCT_OK;
    hipStream_t s = g_stream;
    const int64_t n = c->count;
    const int bblocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
    float *d_part = c->d_bbox;                                         // 1024 x 6 floats, allocated with the cloud
    bbox_partial_kernel<<<bblocks, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)n, d_part);
    std::vector<float> part((size_t)bblocks * 6);
    hipError_t e = hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(float), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(PCT_ERR_HIP, "bbox reduction failed: %s", hipGetErrorString(e));
    for (int k = 0; k < 3; k++) { c->bbox_lo[k] = INFINITY; c->bbox_hi[k] = -INFINITY; }
    for (int b = 0; b < bblocks; b++)
        for (int k = 0; k < 3; k++) {
            c->bbox_lo[k] = std::min(c->bbox_lo[k], part[(size_t)b * 6 + k]);
            c->bbox_hi[k] = std::max(c->bbox_hi[k], part[(size_t)b * 6 + 3 + k]);
        }
    c->bbox_epoch = c->content_epoch;
    return PCT_OK;
}

// Should the brute-force filter take the expanded form (brute2.hpp)?  Only while its absolute error band, ~14 u R^2 (u = 2^-24,
// R = half diagonal of the bounding box), stays small against the squared point spacing of the cloud -- otherwise the band, not
// the sampled bound, decides how many pairs pass.  Fills the centre / R^2 the kernels need.
bool use_expanded_filter(pct_cloud *c, CentreDesc *out)
{
    const int mode = filter_mode();                              // 0 = never, 1 = whenever valid, -1 = auto
    if (mode == 0 || c->host_mapped || c->capturing || c->count < 4) return false;
    if (mode < 0 && c->count < 200000) return false;             // small clouds: the bounding-box pass would cost more than it saves
    if (cloud_bbox_cached(c) != PCT_OK) return false;
    double h[3], R2 = 0.0, vol = 8.0;
    float ctr[3];
    for (int k = 0; k < 3; k++) {
        if (!std::isfinite(c->bbox_lo[k]) || !std::isfinite(c->bbox_hi[k])) return false;
        ctr[k] = (float)(0.5 * ((double)c->bbox_lo[k] + (double)c->bbox_hi[k]));
        h[k] = std::max((double)c->bbox_hi[k] - (double)ctr[k], (double)ctr[k] - (double)c->bbox_lo[k]);
        R2 += h[k] * h[k];
        vol *= h[k];
    }
    R2 *= 1.000001;
    if (!std::isfinite(R2) || R2 > 1e30) return false;
    const double spacing2 = std::pow(vol / (double)c->count, 2.0 / 3.0);
    if (mode < 0 && !(14.0 * 0x1p-24 * R2 <= 0.25 * spacing2)) return false;
    out->cx = ctr[0]; out->cy = ctr[1]; out->cz = ctr[2];
    out->R2 = R2;
    return true;
}

constexpr int kMaxTileParts = 8192;      // tile kernel: at most this many point chunks per launch
constexpr uint32_t kChunkGroupsMax = 3072;   // 3 * 3072 * 16 B = 144 KiB of the CU's 160 KiB LDS

// Default brute-force path: packed-fp32 filter + exact fp64 recheck over LDS-staged chunks
// (kernels.hpp).  d_qf: the fp32 queries; c->d_q64 must already hold their widened copies.
int nn_stream_filtered_slice(pct_cloud *c, const float *d_qf, int64_t qoff, int64_t Q, uint32_t *d_idx, double *d_d2, hipStream_t s)
{
    d_qf += 3 * qoff;
    d_idx += qoff;
    d_d2 += qoff;
    const double *d_q64 = c->d_q64 + 3 * qoff;
    uint32_t *d_bound = c->d_bound + qoff;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_tile_filter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(3 * kChunkGroupsMax * sizeof(float4))));
        attr_set = true;
    }
    const int64_t ngroups = c->count >> 2;
    // chunk = 1024 groups (4096 points, 48 KiB LDS -> 3 blocks per CU); larger only for huge clouds
    uint32_t chunk = 1024;
    if (const char *e = std::getenv("PCT_TILE_CHUNK")) chunk = (uint32_t)std::max(256, std::min((int)kChunkGroupsMax, std::atoi(e)));
    if ((ngroups + chunk - 1) / chunk > kMaxTileParts) chunk = (uint32_t)std::min<int64_t>(kChunkGroupsMax, ((ngroups + kMaxTileParts - 1) / kMaxTileParts + 255) / 256 * 256);
    const int nblocks = (int)std::max<int64_t>(1, (ngroups + chunk - 1) / chunk);
    if (nblocks > kMaxTileParts) return fail(PCT_ERR_INVALID, "cloud of %lld points is too large for the brute-force path", (long long)c->count);
    // sample 1 chunk in 16 (everything for small clouds; sparser for huge ones so the partial buffer fits)
    const uint32_t stride = ngroups >= 256ll * kSampleStride * 8
                                ? (uint32_t)std::max<int64_t>(kSampleStride, (ngroups + 256ll * kMaxParts - 1) / (256ll * kMaxParts))
                                : 1u;
    const int64_t schunks = std::max<int64_t>(1, (ngroups + 256ll * stride - 1) / (256ll * stride));
    const int sblocks = (int)((schunks + kSampleGroups - 1) / kSampleGroups);
    const int64_t part_cap = c->part_q * kMaxParts;        // entries in d_part_d2 / d_part_idx
    begin_timing(c, s);
    // the sample partials borrow d_part_idx (u32 and float have the same size; [Q][sblocks], sblocks <= kMaxParts);
    // bound_reduce_kernel consumes them before the filter pass overwrites the buffer
    // slices of the batch in grid.y until ~2048 blocks are in flight (the kernels' tile loops are sequential)
    const auto slices_for = [&](int point_blocks, int *qslice) {
        const int tiles = (int)((Q + kTileQ - 1) / kTileQ);
        int slices = std::max(1, std::min(tiles, (2048 + point_blocks - 1) / point_blocks));
        *qslice = ((tiles + slices - 1) / slices) * kTileQ;
        return (int)((Q + *qslice - 1) / *qslice);
    };
    int sq = 0;
    const int sslices = slices_for(sblocks, &sq);
    nn_sample_bounds_kernel<<<dim3(sblocks, sslices), 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, stride, d_qf, (int)Q, sq,
                                                                   reinterpret_cast<float *>(c->d_part_idx), sblocks);
    bound_reduce_kernel<<<(int)Q, 256, 0, s>>>(reinterpret_cast<const float *>(c->d_part_idx), sblocks, d_bound);
    static const bool candidates = [] { const char *e = std::getenv("PCT_TILE_CANDIDATES"); return e ? std::atoi(e) != 0 : true; }();
    CentreDesc CD{};
    if (candidates && use_expanded_filter(c, &CD)) {
        // expanded form (brute2.hpp): 3 FMAs per pair on centred coordinates, thresholds widened by the proven error band
        static const int gpi = [] { const char *e = std::getenv("PCT_TILE_GROUPS"); return e ? std::atoi(e) : 2; }();
        static bool attr3 = false;
        if (!attr3) {
            const int lds = (int)(3 * kChunkGroupsMax * sizeof(float4));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_tile_candidates2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_tile_candidates2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_tile_candidates2_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr3 = true;
        }
        float4 *qprep = c->d_qsorted + qoff;                     // the query-sort records are idle on this path
        brute2_prep_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(CD, d_qf, d_bound, (uint32_t)Q, qprep);
        dom_begin(c, s);
        int cq = 0;
        const int cslices = slices_for(nblocks, &cq);
        const dim3 grid(nblocks, cslices);
        const size_t lds = 3 * (size_t)chunk * sizeof(float4);
        static const bool reg_points = [] { const char *e = std::getenv("PCT_TILE_REG"); return e ? std::atoi(e) != 0 : true; }();
        if (reg_points) {       // points held in registers (brute2.hpp tile_reg_kernel): a block covers 4096 points, no LDS
            const int rg = reg_groups();
            const int rblocks = (int)std::max<int64_t>(1, (ngroups + 256 * rg - 1) / (256 * rg));
            int rq = 0;
            const int rslices = slices_for(rblocks, &rq);
            const dim3 rgrid(rblocks, rslices);
            switch (rg) {
            case 2: tile_reg_kernel<false, 2><<<rgrid, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, CD, qprep, d_q64, nullptr, (int)Q, rq, c->d_cand_count, c->d_cand_d2, c->d_cand_idx, nullptr); break;
            case 4: tile_reg_kernel<false, 4><<<rgrid, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, CD, qprep, d_q64, nullptr, (int)Q, rq, c->d_cand_count, c->d_cand_d2, c->d_cand_idx, nullptr); break;
            case 6: tile_reg_kernel<false, 6><<<rgrid, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, CD, qprep, d_q64, nullptr, (int)Q, rq, c->d_cand_count, c->d_cand_d2, c->d_cand_idx, nullptr); break;
            default: tile_reg_kernel<false, 3><<<rgrid, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, CD, qprep, d_q64, nullptr, (int)Q, rq, c->d_cand_count, c->d_cand_d2, c->d_cand_idx, nullptr); break;
            }
        } else if (gpi == 1)
            nn_tile_candidates2_kernel<1><<<grid, 256, lds, s>>>(c->x, c->y, c->z, (uint32_t)c->count, chunk, CD, qprep, d_q64, (int)Q, cq, c->d_cand_count, c->d_cand_d2, c->d_cand_idx);
        else if (gpi == 4)
            nn_tile_candidates2_kernel<4><<<grid, 256, lds, s>>>(c->x, c->y, c->z, (uint32_t)c->count, chunk, CD, qprep, d_q64, (int)Q, cq, c->d_cand_count, c->d_cand_d2, c->d_cand_idx);
        else
            nn_tile_candidates2_kernel<2><<<grid, 256, lds, s>>>(c->x, c->y, c->z, (uint32_t)c->count, chunk, CD, qprep, d_q64, (int)Q, cq, c->d_cand_count, c->d_cand_d2, c->d_cand_idx);
        dom_end(c, s);
        HIPCHK(hipMemsetAsync(c->d_ovf, 0, sizeof(uint32_t), s));
        nn_reduce_candidates_kernel<<<(int)Q, 256, 0, s>>>(c->d_cand_count, c->d_cand_d2, c->d_cand_idx, (uint32_t)c->index_base, c->d_ovf, d_idx, d_d2);
        nn_overflow_scan_kernel<<<kOvfBlocks, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, d_q64, c->d_ovf, c->d_part_d2, c->d_part_idx);
        nn_overflow_fold_kernel<<<(int)Q, 256, 0, s>>>(c->d_ovf, c->d_part_d2, c->d_part_idx, kOvfBlocks, (uint32_t)c->index_base, d_idx, d_d2);
        end_timing(c, s);
        HIPCHK(hipGetLastError());
        c->host_work = true;
        c->host_points = (uint64_t)Q * (uint64_t)c->count;
        return PCT_OK;
    }
    if (candidates) {       // survivors of the bound go to per-query candidate lists: no per-tile block reductions, no partial arrays
        static bool attr2 = false;
        if (!attr2) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_tile_candidates_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(3 * kChunkGroupsMax * sizeof(float4))));
            attr2 = true;
        }
        dom_begin(c, s);
        int cq = 0;
        const int cslices = slices_for(nblocks, &cq);
        nn_tile_candidates_kernel<<<dim3(nblocks, cslices), 256, 3 * (size_t)chunk * sizeof(float4), s>>>(c->x, c->y, c->z, (uint32_t)c->count, chunk, d_qf,
                                                                                                          d_q64, d_bound, (int)Q, cq, c->d_cand_count,
                                                                                                          c->d_cand_d2, c->d_cand_idx);
        dom_end(c, s);
        HIPCHK(hipMemsetAsync(c->d_ovf, 0, sizeof(uint32_t), s));
        nn_reduce_candidates_kernel<<<(int)Q, 256, 0, s>>>(c->d_cand_count, c->d_cand_d2, c->d_cand_idx, (uint32_t)c->index_base, c->d_ovf, d_idx, d_d2);
        // overflowed lists (bulk exact ties): exact scan by the whole grid; both kernels return at once when there are none
        nn_overflow_scan_kernel<<<kOvfBlocks, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, d_q64, c->d_ovf, c->d_part_d2, c->d_part_idx);
        nn_overflow_fold_kernel<<<(int)Q, 256, 0, s>>>(c->d_ovf, c->d_part_d2, c->d_part_idx, kOvfBlocks, (uint32_t)c->index_base, d_idx, d_d2);
        end_timing(c, s);
        HIPCHK(hipGetLastError());
        c->host_work = true;
        c->host_points = (uint64_t)Q * (uint64_t)c->count;
        return PCT_OK;
    }
    const int64_t qb_max = std::max<int64_t>(kTileQ, part_cap / nblocks / kTileQ * kTileQ);
    for (int64_t qbase = 0; qbase < Q; qbase += qb_max) {
        const int qb = (int)std::min<int64_t>(qb_max, Q - qbase);
        if (qbase == 0) dom_begin(c, s);
        nn_tile_filter_kernel<<<nblocks, 256, 3 * (size_t)chunk * sizeof(float4), s>>>(c->x, c->y, c->z, (uint32_t)c->count, chunk, d_qf, d_q64,
                                                                                       d_bound, (int)qbase, qb, c->d_part_d2, c->d_part_idx, nblocks);
        if (qbase == 0) dom_end(c, s);
        nn_reduce_partials_kernel<<<qb, 256, 0, s>>>(c->d_part_d2, c->d_part_idx, nblocks, (uint32_t)c->index_base, d_idx + qbase, d_d2 + qbase);
    }
    end_timing(c, s);
    HIPCHK(hipGetLastError());
    c->host_work = true;
    c->host_points = (uint64_t)Q * (uint64_t)c->count;
    return PCT_OK;
}

int nn_stream_filtered(pct_cloud *c, const float *d_qf, int64_t Q, uint32_t *d_idx, double *d_d2, hipStream_t s)
{
    for (int64_t off = 0; off < Q; off += c->part_q)
        PCTCHK(nn_stream_filtered_slice(c, d_qf, off, std::min<int64_t>(c->part_q, Q - off), d_idx, d_d2, s));
    c->host_points = (uint64_t)Q * (uint64_t)c->count;
    return PCT_OK;
}

int ring_nn_dev(pct_cloud *c, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, hipStream_t s);

int nn_dev(pct_cloud *c, int algo, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, hipStream_t s)
{
    if (Q == 0) return PCT_OK;
    if (Q > c->qcap) return fail(PCT_ERR_INVALID, "batch of %lld exceeds reserved %lld (call pct_cloud_reserve_queries)", (long long)Q, (long long)c->qcap);
    if (c->count == 0) {
        fill_empty_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(d_idx, d_d2, (uint32_t)Q);
        HIPCHK(hipGetLastError());
        return PCT_OK;
    }
    if (c->ring_ready && (algo == PCT_ALGO_AUTO || algo == PCT_ALGO_GRID)) return ring_nn_dev(c, d_q, Q, d_idx, d_d2, s);   // rolling-map index
    if (algo == PCT_ALGO_AUTO) algo = c->has_grid ? PCT_ALGO_GRID : PCT_ALGO_STREAM;
    if (algo == PCT_ALGO_GRID) {
        if (!c->has_grid) return fail(PCT_ERR_INVALID, "PCT_ALGO_GRID without a grid (call pct_cloud_build_grid)");
        c->host_work = false;
        if (c->count_work) HIPCHK(hipMemsetAsync(c->d_work, 0, sizeof(WorkCounters) * kWorkSlots, s));
        begin_timing(c, s);
        const uint32_t *perm = nullptr;
        static const bool coop = [] { const char *e = std::getenv("PCT_GRID_COOP"); return e ? std::atoi(e) != 0 : true; }();
        static const bool sorted_writes_on = [] { const char *e = std::getenv("PCT_SORTED_WRITES"); return e ? std::atoi(e) != 0 : false; }();
        PCTCHK(bin_queries(c, d_q, Q, s, &perm, !coop, coop && sorted_writes_on));
        dom_begin(c, s, coop && !c->count_work);
        if (coop) {   // 8 lanes per query (default)
            const int blocks = ceil_div(Q, 256 / kCoop);
            // sorted batches: the kernel writes its results in sorted order (full lines) and one gather pass puts them back into
            // arrival order -- scattering 4 + 8 bytes per query from here cost 5.6x the result bytes in fabric writes
            static const bool sorted_writes = [] { const char *e = std::getenv("PCT_SORTED_WRITES"); return e ? std::atoi(e) != 0 : false; }();
            const bool so = perm != nullptr && sorted_writes && lds_sort_on();
            uint32_t *k_idx = so ? c->d_sres_idx : d_idx;
            double *k_d2 = so ? c->d_sres_d2 : d_d2;
            // wave-cooperative fallback for the queries the 2x2x2 block leaves undecided (kernels.hpp, default) or the 8-lane cube
            static const bool wave_cube = [] { const char *e = std::getenv("PCT_COOP_WAVE_CUBE"); return e ? std::atoi(e) != 0 : true; }();
            const float4 *recs = perm ? c->d_qsorted : nullptr;
            if (c->has_pyr) {     // sparse occupancy: stage 0, then the bounding-box pyramid instead of cube + shells (pyramid.hpp)
                // fp32 walk for everybody, then the exact walk for the (few) queries it lists as undecided; PCT_PYRAMID_EXACT=1:
                // the exact walk for everybody (tests)
                static const bool exact_only = [] { const char *e = std::getenv("PCT_PYRAMID_EXACT"); return e ? std::atoi(e) != 0 : false; }();
                const int so_i = so ? 1 : 0;
                if (exact_only) {
                    if (c->count_work)
                        nn_grid_pyr_kernel<true, false><<<blocks, 256, 0, s>>>(c->G, c->P, c->pyr_nodes, c->pyr_hint, c->sorted, c->cell_start, d_q, (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so_i, nullptr);
                    else
                        nn_grid_pyr_kernel<false, false><<<blocks, 256, 0, s>>>(c->G, c->P, c->pyr_nodes, c->pyr_hint, c->sorted, c->cell_start, d_q, (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so_i, nullptr);
                } else {
                    if (c->count_work)
                        nn_grid_pyr_kernel<true, true><<<blocks, 256, 0, s>>>(c->G, c->P, c->pyr_nodes, c->pyr_hint, c->sorted, c->cell_start, d_q, (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so_i, c->d_todo);
                    else if (c->dom_valid && dom_ext_on()) {
                        hipExtLaunchKernelGGL((nn_grid_pyr_kernel<false, true>), dim3(blocks), dim3(256), 0, s, c->ev2, c->ev3, 0, c->G, c->P, c->pyr_nodes, c->pyr_hint, c->sorted, c->cell_start, d_q,
                                              (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so_i, c->d_todo);
                        dom_done(c);
                    } else
                        nn_grid_pyr_kernel<false, true><<<blocks, 256, 0, s>>>(c->G, c->P, c->pyr_nodes, c->pyr_hint, c->sorted, c->cell_start, d_q, (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so_i, c->d_todo);
                    const int tblocks = (int)std::min<int64_t>(256, blocks);
                    if (c->count_work)
                        nn_grid_pyr_todo_kernel<true><<<tblocks, 256, 0, s>>>(c->G, c->P, c->pyr_nodes, c->pyr_hint, c->sorted, c->cell_start, d_q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so_i, c->d_todo);
                    else
                        nn_grid_pyr_todo_kernel<false><<<tblocks, 256, 0, s>>>(c->G, c->P, c->pyr_nodes, c->pyr_hint, c->sorted, c->cell_start, d_q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so_i, c->d_todo);
                }
            } else if (c->count_work) {
                if (wave_cube) nn_grid_coop_kernel<true, true><<<blocks, 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so ? 1 : 0);
                else nn_grid_coop_kernel<true, false><<<blocks, 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so ? 1 : 0);
            } else if (c->dom_valid && dom_ext_on()) {
                // the kernel's own begin / end timestamps (hipExtLaunchKernel): no marker packets on the stream -- the two
                // hipEventRecord calls of dom_begin / dom_end cost ~10 us of a 160 us step
                if (wave_cube)
                    hipExtLaunchKernelGGL((nn_grid_coop_kernel<false, true>), dim3(blocks), dim3(256), 0, s, c->ev2, c->ev3, 0, c->G, c->sorted, c->cell_start, d_q,
                                          (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so ? 1 : 0);
                else
                    hipExtLaunchKernelGGL((nn_grid_coop_kernel<false, false>), dim3(blocks), dim3(256), 0, s, c->ev2, c->ev3, 0, c->G, c->sorted, c->cell_start, d_q,
                                          (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so ? 1 : 0);
                dom_done(c);
            } else if (wave_cube)
                nn_grid_coop_kernel<false, true><<<blocks, 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so ? 1 : 0);
            else
                nn_grid_coop_kernel<false, false><<<blocks, 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, (uint32_t)Q, (uint32_t)c->index_base, recs, k_idx, k_d2, c->d_work, so ? 1 : 0);
            if (so) {
                dom_end(c, s);
                unpermute_results_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(c->d_inv, c->d_sres_idx, c->d_sres_d2, (uint32_t)Q, d_idx, d_d2);
                end_timing(c, s);
                HIPCHK(hipGetLastError());
                return PCT_OK;
            }
        } else if (c->count_work)
            nn_grid_kernel<true><<<ceil_div(Q, 256), 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, (uint32_t)Q,
                                                                   (uint32_t)c->index_base, perm, d_idx, d_d2, c->d_work);
        else
            nn_grid_kernel<false><<<ceil_div(Q, 256), 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, (uint32_t)Q,
                                                                    (uint32_t)c->index_base, perm, d_idx, d_d2, c->d_work);
        dom_end(c, s);
        end_timing(c, s);
        HIPCHK(hipGetLastError());
        return PCT_OK;
    }
    if (algo != PCT_ALGO_STREAM && algo != PCT_ALGO_STREAM_EXACT) return fail(PCT_ERR_INVALID, "unknown algo %d", algo);
    widen_queries_kernel<<<ceil_div(3 * Q, 256), 256, 0, s>>>(d_q, (uint32_t)(3 * Q), c->d_q64);
    // <= 4 queries: the all-fp64 kernel is already HBM-bound (50-62 % of peak); beyond that the
    // packed-fp32 filter wins
    static const int64_t exact_max_q = [] { const char *e = std::getenv("PCT_EXACT_MAX_Q"); return e ? std::atoll(e) : 4ll; }();
    if (algo == PCT_ALGO_STREAM_EXACT || Q <= exact_max_q) return nn_stream_q64(c, Q, d_idx, d_d2, s);
    return nn_stream_filtered(c, d_q, Q, d_idx, d_d2, s);
}

int count_dev(pct_cloud *c, int algo, const float *d_q, const float *d_r, int64_t Q, uint32_t *d_count, hipStream_t s)
{
    if (Q == 0) return PCT_OK;
    if (Q > c->qcap) return fail(PCT_ERR_INVALID, "batch of %lld exceeds reserved %lld", (long long)Q, (long long)c->qcap);
    HIPCHK(hipMemsetAsync(d_count, 0, sizeof(uint32_t) * Q, s));
    if (c->count == 0) return PCT_OK;
    if (algo == PCT_ALGO_AUTO) algo = c->has_grid ? PCT_ALGO_GRID : PCT_ALGO_STREAM;
    if (algo == PCT_ALGO_GRID) {
        if (!c->has_grid) return fail(PCT_ERR_INVALID, "PCT_ALGO_GRID without a grid");
        c->host_work = false;
        if (c->count_work) HIPCHK(hipMemsetAsync(c->d_work, 0, sizeof(WorkCounters) * kWorkSlots, s));
        begin_timing(c, s);
        const uint32_t *perm = nullptr;
        static const bool coop = [] { const char *e = std::getenv("PCT_GRID_COOP"); return e ? std::atoi(e) != 0 : true; }();
        PCTCHK(bin_queries(c, d_q, Q, s, &perm, !coop, false));
        dom_begin(c, s, coop && !c->count_work);
        if (coop) {   // 8 lanes per query (default)
            const int blocks = ceil_div(Q, 256 / kCoop);
            const float4 *qs = perm ? c->d_qsorted : nullptr;
            if (c->count_work)
                count_grid_coop_kernel<true><<<blocks, 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, d_r, (uint32_t)Q, qs, d_count, c->d_work);
            else
                count_grid_coop_kernel<false><<<blocks, 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, d_r, (uint32_t)Q, qs, d_count, c->d_work);
        } else if (c->count_work)
            count_grid_kernel<true><<<ceil_div(Q, 256), 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, d_r, (uint32_t)Q, perm, d_count, c->d_work);
        else
            count_grid_kernel<false><<<ceil_div(Q, 256), 256, 0, s>>>(c->G, c->sorted, c->cell_start, d_q, d_r, (uint32_t)Q, perm, d_count, c->d_work);
        dom_end(c, s);
        end_timing(c, s);
        HIPCHK(hipGetLastError());
        return PCT_OK;
    }
    if (algo != PCT_ALGO_STREAM) return fail(PCT_ERR_INVALID, "unknown algo %d", algo);
    widen_queries_kernel<<<ceil_div(3 * Q, 256), 256, 0, s>>>(d_q, (uint32_t)(3 * Q), c->d_q64);
    CentreDesc CD{};
    static const int64_t count_filter_min_q = [] { const char *e = std::getenv("PCT_COUNT_FILTER_MIN_Q"); return e ? std::atoll(e) : 16ll; }();
    if (Q >= count_filter_min_q && use_expanded_filter(c, &CD)) {
        // packed-fp32 filter in expanded form + exact fp64 test of whatever may lie inside the ball (brute2.hpp tile_reg_kernel<true>)
        const int64_t ngroups = c->count >> 2;
        const int rg = reg_groups();
        const int rblocks = (int)std::max<int64_t>(1, (ngroups + 256 * rg - 1) / (256 * rg));
        const int tiles = (int)((Q + kTileQ - 1) / kTileQ);
        const int slices = std::max(1, std::min(tiles, (2048 + rblocks - 1) / rblocks));
        const int rq = ((tiles + slices - 1) / slices) * kTileQ;
        const int rslices = (int)((Q + rq - 1) / rq);
        begin_timing(c, s);
        brute2_prep_count_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(CD, d_q, d_r, (uint32_t)Q, c->d_qsorted, c->d_r2);
        dom_begin(c, s);
        const dim3 rgrid(rblocks, rslices);
        switch (rg) {
        case 2: tile_reg_kernel<true, 2><<<rgrid, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, CD, c->d_qsorted, c->d_q64, c->d_r2, (int)Q, rq, nullptr, nullptr, nullptr, d_count); break;
        case 4: tile_reg_kernel<true, 4><<<rgrid, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, CD, c->d_qsorted, c->d_q64, c->d_r2, (int)Q, rq, nullptr, nullptr, nullptr, d_count); break;
        case 6: tile_reg_kernel<true, 6><<<rgrid, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, CD, c->d_qsorted, c->d_q64, c->d_r2, (int)Q, rq, nullptr, nullptr, nullptr, d_count); break;
        default: tile_reg_kernel<true, 3><<<rgrid, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)c->count, CD, c->d_qsorted, c->d_q64, c->d_r2, (int)Q, rq, nullptr, nullptr, nullptr, d_count); break;
        }
        dom_end(c, s);
        end_timing(c, s);
        HIPCHK(hipGetLastError());
        return PCT_OK;
    }
    square_radii_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(d_r, (uint32_t)Q, c->d_r2);
    const int blocks = stream_blocks(c->count);
    begin_timing(c, s);
    for (int64_t q0 = 0; q0 < Q;) {
        const int qt = pick_tile(Q - q0);
        const int qcount = (int)std::min<int64_t>(qt, Q - q0);
        switch (qt) {
        case 8: launch_count_stream<8>(c, blocks, (int)q0, qcount, d_count, s); break;
        case 4: launch_count_stream<4>(c, blocks, (int)q0, qcount, d_count, s); break;
        case 2: launch_count_stream<2>(c, blocks, (int)q0, qcount, d_count, s); break;
        default: launch_count_stream<1>(c, blocks, (int)q0, qcount, d_count, s); break;
        }
        q0 += qcount;
    }
    end_timing(c, s);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

InflateParams to_dev(const pct_inflate_params *p)
{
    return InflateParams{ p->start[0], p->start[1], p->start[2], p->sample_range, p->search_margin, p->max_radius };
}

// pts64 (device, Q x 3) -> radius/idx/d2 in the cloud's workspaces
// d_pts: the planner points (device-visible), default the staging buffer; d_out != nullptr: results as records in host-mapped memory
// o_radius / o_idx / o_d2: where the results go (default: the cloud's workspaces)
int inflate_dev(pct_cloud *c, const pct_inflate_params *p, int64_t Q, hipStream_t s, const double *d_pts = nullptr, ExpressOut *d_out = nullptr,
                double *o_radius = nullptr, uint32_t *o_idx = nullptr, double *o_d2 = nullptr)
{
    const InflateParams P = to_dev(p);
    if (!o_radius) o_radius = c->d_radius;
    if (!o_idx) o_idx = c->d_idx;
    if (!o_d2) o_d2 = c->d_d2;
    inflate_prologue_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(P, d_pts ? d_pts : c->d_pts64, (uint32_t)Q, c->d_q, c->d_skip);
    if (c->count > 0) PCTCHK(nn_dev(c, PCT_ALGO_AUTO, c->d_q, Q, o_idx, o_d2, s));
    if (d_out)
        inflate_epilogue_out_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(P, (uint32_t)Q, c->d_skip, c->count == 0 ? 1 : 0, o_idx, o_d2, d_out);
    else
        inflate_epilogue_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(P, (uint32_t)Q, c->d_skip, c->count == 0 ? 1 : 0, o_idx, o_d2, o_radius);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

}  // namespace

#include "ring_host.inc"

// ======================================================================================
//  C ABI
// ======================================================================================
namespace pct_internal {
hipStream_t stream() { return g_stream; }
int require_init() { return ::require_init(); }
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace pct_internal

extern "C" {

const char *pct_last_error(void) { return g_err; }


int pct_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pct_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(PCT_ERR_NO_DEVICE, "no HIP device (%s); this library has no host fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(PCT_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    if (g_stream && g_device != device) { (void)hipStreamDestroy(g_stream); g_stream = nullptr; }
    if (!g_stream) HIPCHK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_device = device;
    return PCT_OK;
}

int pct_sync(void)
{
    PCTCHK(require_init());
    HIPCHK(hipDeviceSynchronize());    // the library's stream and any caller stream handed to the *_dev entry points
    return PCT_OK;
}

static int cloud_create_impl(int64_t capacity, bool host_mapped, pct_cloud **out)
{
    if (!out || capacity < 0 || capacity > 0xFFFFFFF0ll) return fail(PCT_ERR_INVALID, "bad capacity");
    PCTCHK(require_init());
    pct_cloud *c = new (std::nothrow) pct_cloud();
    if (!c) return fail(PCT_ERR_ALLOC, "host allocation failed");
    c->cap = capacity;
    c->cap4 = (capacity + 3) & ~3ll;
    c->host_mapped = host_mapped;
    int s;
    if (host_mapped)
        s = mapped_alloc(&c->hx, &c->x, (size_t)c->cap4 + 4) || mapped_alloc(&c->hy, &c->y, (size_t)c->cap4 + 4) ||
            mapped_alloc(&c->hz, &c->z, (size_t)c->cap4 + 4);
    else
        s = dev_alloc(&c->x, (size_t)c->cap4 + 4) || dev_alloc(&c->y, (size_t)c->cap4 + 4) || dev_alloc(&c->z, (size_t)c->cap4 + 4);
    if (!s) s = dev_alloc(&c->d_work, kWorkSlots);
    if (!s) s = dev_alloc(&c->d_bbox, (size_t)1024 * 6);
    if (!s) s = dev_alloc(&c->d_gbcheck, kGbCheckSlots);
    if (!s && hipHostMalloc((void **)&c->h_gbcheck, sizeof(GbCheck) * kGbCheckSlots, hipHostMallocDefault) != hipSuccess) s = fail(PCT_ERR_ALLOC, "hipHostMalloc failed");
    if (!s) s = mapped_alloc(&c->h_xout, &c->d_xout, kExpressMaxQ);
    if (!s) s = mapped_alloc(&c->h_xin, &c->d_xin, 3 * kExpressMaxQ);
    if (!s) s = mapped_alloc(&c->h_xr, &c->d_xr, kExpressMaxQ);
    if (!s) s = mapped_alloc(&c->h_xids, &c->d_xids, kExpressIdsCap);
    if (!s) s = mapped_alloc(&c->h_xseq, &c->d_xseq, 16);
    if (!s) s = dev_alloc(&c->d_xcounter, 16);
    if (!s) { *c->h_xseq = 0; if (hipMemset(c->d_xcounter, 0, 16 * sizeof(uint32_t)) != hipSuccess) s = fail(PCT_ERR_HIP, "hipMemset failed"); }
    if (s) {
        pct_cloud_destroy(c);
        return PCT_ERR_ALLOC;
    }
    bool ring_ok = true;
    for (hipEvent_t &e : c->dom_ring) ring_ok = ring_ok && hipEventCreate(&e) == hipSuccess;
    c->ev2 = c->dom_ring[0];
    c->ev3 = c->dom_ring[1];
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || !ring_ok) {
        pct_cloud_destroy(c);
        return fail(PCT_ERR_HIP, "hipEventCreate failed");
    }
    *out = c;
    return PCT_OK;
}

int pct_cloud_create(int64_t capacity, pct_cloud **out) { return cloud_create_impl(capacity, false, out); }

int pct_cloud_create_small(int64_t capacity, pct_cloud **out)
{
    if (capacity > (1ll << 22)) return fail(PCT_ERR_INVALID, "small (host-mapped) clouds hold at most 4M points");
    return cloud_create_impl(capacity, true, out);
}

int pct_cloud_destroy(pct_cloud *c)
{
    if (!c) return PCT_OK;
    if (g_stream) (void)hipStreamSynchronize(g_stream);
    if (c->host_mapped) {
        if (c->hx) (void)hipHostFree(c->hx);
        if (c->hy) (void)hipHostFree(c->hy);
        if (c->hz) (void)hipHostFree(c->hz);
        c->x = c->y = c->z = nullptr;
    }
    if (c->h_xout) (void)hipHostFree(c->h_xout);
    if (c->h_xin) (void)hipHostFree(c->h_xin);
    if (c->h_xr) (void)hipHostFree(c->h_xr);
    if (c->h_xids) (void)hipHostFree(c->h_xids);
    if (c->h_xseq) (void)hipHostFree(c->h_xseq);
    if (c->ev_mut) (void)hipEventDestroy(c->ev_mut);
    if (c->h_frame) (void)hipHostFree(c->h_frame);
    if (c->h_astage) (void)hipHostFree(c->h_astage);
    if (c->h_mq) (void)hipHostFree(c->h_mq);
    if (c->h_mi) (void)hipHostFree(c->h_mi);
    if (c->h_md) (void)hipHostFree(c->h_md);
    dev_free(c->d_xcounter);
    if (c->h_aux) (void)hipHostFree(c->h_aux);
    if (c->h_eout) (void)hipHostFree(c->h_eout);
    if (c->h_bpos) (void)hipHostFree(c->h_bpos);
    dev_free(c->x); dev_free(c->y); dev_free(c->z); dev_free(c->d_stage); dev_free(c->gb_tmp); dev_free(c->gb_small);
    dev_free(c->blocks);
    dev_free(c->cell_start); dev_free(c->sorted); dev_free(c->bin_start); dev_free(c->bin_fill); dev_free(c->bin_tiles);
    for (int l = 0; l < kMaxCoarse; l++) { dev_free(c->coarse_cell_start[l]); dev_free(c->coarse_sorted[l]); }
    dev_free(c->d_qbin); dev_free(c->d_perm); dev_free(c->d_qsorted); dev_free(c->d_sorttmp); dev_free(c->d_sortkey); dev_free(c->d_sort1);
    dev_free(c->d_inv); dev_free(c->d_sres_idx); dev_free(c->d_sres_d2); dev_free(c->d_todo);
    dev_free(c->d_q); dev_free(c->d_r); dev_free(c->d_q64); dev_free(c->d_r2); dev_free(c->d_d2); dev_free(c->d_radius);
    dev_free(c->d_pts64); dev_free(c->d_idx); dev_free(c->d_count); dev_free(c->d_skip); dev_free(c->d_bound);
    dev_free(c->d_part_d2); dev_free(c->d_part_idx); dev_free(c->d_cand_count); dev_free(c->d_cand_d2); dev_free(c->d_cand_idx); dev_free(c->d_ovf);
    dev_free(c->d_coef); dev_free(c->d_segtime); dev_free(c->d_orders); dev_free(c->d_nsamples); dev_free(c->d_first_hit);
    dev_free(c->d_work);
    dev_free(c->d_bbox); dev_free(c->d_gbcheck); dev_free(c->pyr_nodes); dev_free(c->pyr_hint);
    if (c->h_gbcheck) (void)hipHostFree(c->h_gbcheck);
    dev_free(c->ring_ht); dev_free(c->ring_slots); dev_free(c->ring_ovf); dev_free(c->ring_where); dev_free(c->ring_st);
    if (c->h_ring_status) (void)hipHostFree(c->h_ring_status);
    replan_ctx_free(c->rp);
    dev_free(c->crop_tile); dev_free(c->crop_idx); dev_free(c->crop_d2); dev_free(c->crop_x); dev_free(c->crop_y); dev_free(c->crop_z);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->dom_ring) if (e) (void)hipEventDestroy(e);
    delete c;
    return PCT_OK;
}

int64_t pct_cloud_size(const pct_cloud *c) { return c ? c->count : 0; }
int64_t pct_cloud_capacity(const pct_cloud *c) { return c ? c->cap : 0; }

int pct_cloud_set_index_base(pct_cloud *c, int64_t base)
{
    if (!c || base < 0 || base + c->cap > 0xFFFFFFF0ll) return fail(PCT_ERR_INVALID, "bad index base");
    c->index_base = base;
    return PCT_OK;
}

int pct_cloud_upload_aos(pct_cloud *c, const void *pts, int64_t n, int64_t stride_bytes)
{
    if (!c || n < 0 || (n > 0 && !pts) || stride_bytes < 12 || (stride_bytes & 3)) return fail(PCT_ERR_INVALID, "bad upload arguments");
    if (n > c->cap) return fail(PCT_ERR_CAPACITY, "%lld points > capacity %lld", (long long)n, (long long)c->cap);
    drop_grid(c);
    PCTCHK(upload_range(c, pts, n, stride_bytes, 0));
    HIPCHK(hipStreamSynchronize(g_stream));    // the host buffer is the caller's again
    c->count = n;
    c->ring_next = n % std::max<int64_t>(c->cap, 1);
    return after_replace(c);
}

// sensor_msgs/PointCloud2 (rcvPointCloudCallBack, sim_planning_demo.cpp:159-167): a byte blob of `n` records of
// `point_step` bytes whose FLOAT32 fields x, y, z sit at arbitrary byte offsets.  Records are repacked on the host
// into 12-byte xyz (one pass; the message is pageable host memory anyway) and take the packed upload path.
int pct_cloud_upload_fields(pct_cloud *c, const void *data, int64_t n, int64_t point_step, int64_t off_x, int64_t off_y, int64_t off_z)
{
    if (!c || n < 0 || (n > 0 && !data) || point_step < 4 || off_x < 0 || off_y < 0 || off_z < 0 || off_x + 4 > point_step ||
        off_y + 4 > point_step || off_z + 4 > point_step)
        return fail(PCT_ERR_INVALID, "bad upload_fields arguments");
    if (off_x == 0 && off_y == 4 && off_z == 8 && (point_step & 3) == 0 && point_step >= 12) return pct_cloud_upload_aos(c, data, n, point_step);
    std::vector<float> packed;
    try { packed.resize((size_t)3 * n); } catch (const std::bad_alloc &) { return fail(PCT_ERR_ALLOC, "host allocation failed"); }
    const unsigned char *src = static_cast<const unsigned char *>(data);
    for (int64_t i = 0; i < n; i++) {
        std::memcpy(&packed[3 * i], src + i * point_step + off_x, 4);
        std::memcpy(&packed[3 * i + 1], src + i * point_step + off_y, 4);
        std::memcpy(&packed[3 * i + 2], src + i * point_step + off_z, 4);
    }
    return pct_cloud_upload_aos(c, packed.data(), n, 12);
}

int pct_cloud_upload_soa_dev(pct_cloud *c, const float *d_x, const float *d_y, const float *d_z, int64_t n)
{
    if (!c || n < 0 || (n > 0 && (!d_x || !d_y || !d_z))) return fail(PCT_ERR_INVALID, "bad upload arguments");
    if (n > c->cap) return fail(PCT_ERR_CAPACITY, "%lld points > capacity %lld", (long long)n, (long long)c->cap);
    drop_grid(c);
    if (n) {
        HIPCHK(hipMemcpyAsync(c->x, d_x, sizeof(float) * n, hipMemcpyDeviceToDevice, g_stream));
        HIPCHK(hipMemcpyAsync(c->y, d_y, sizeof(float) * n, hipMemcpyDeviceToDevice, g_stream));
        HIPCHK(hipMemcpyAsync(c->z, d_z, sizeof(float) * n, hipMemcpyDeviceToDevice, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
    }
    c->count = n;
    c->ring_next = n % std::max<int64_t>(c->cap, 1);
    return after_replace(c);
}

int pct_cloud_append_aos(pct_cloud *c, const void *pts, int64_t n, int64_t stride_bytes)
{
    if (!c || n < 0 || (n > 0 && !pts) || stride_bytes < 12 || (stride_bytes & 3)) return fail(PCT_ERR_INVALID, "bad append arguments");
    if (n > c->cap) return fail(PCT_ERR_CAPACITY, "appending %lld points to a ring of %lld", (long long)n, (long long)c->cap);
    if (n == 0) return PCT_OK;
    drop_grid(c);
    if (c->ring_ready) return ring_append(c, pts, n, stride_bytes);     // rolling-map index: updated in place
    const int64_t first = std::min(n, c->cap - c->ring_next);
    PCTCHK(upload_range(c, pts, first, stride_bytes, c->ring_next));
    HIPCHK(hipStreamSynchronize(g_stream));   // the staging buffer is reused by the wrapped part
    if (first < n) {
        PCTCHK(upload_range(c, static_cast<const unsigned char *>(pts) + first * stride_bytes, n - first, stride_bytes, 0));
        HIPCHK(hipStreamSynchronize(g_stream));
    }
    c->ring_next = (c->ring_next + n) % c->cap;
    c->count = std::min(c->cap, c->count + n);
    return after_replace(c);
}

int pct_cloud_reserve_queries(pct_cloud *c, int64_t Q)
{
    if (!c || Q < 0) return fail(PCT_ERR_INVALID, "bad reserve");
    if (Q <= c->qcap) return PCT_OK;
    if (c->capturing) return fail(PCT_ERR_INVALID, "cannot grow workspaces during graph capture");
    HIPCHK(hipStreamSynchronize(g_stream));
    const int64_t q = std::max<int64_t>(Q, 256);
    dev_free(c->d_q); dev_free(c->d_r); dev_free(c->d_q64); dev_free(c->d_r2); dev_free(c->d_d2); dev_free(c->d_radius);
    dev_free(c->d_pts64); dev_free(c->d_idx); dev_free(c->d_count); dev_free(c->d_skip); dev_free(c->d_bound);
    dev_free(c->d_part_d2); dev_free(c->d_part_idx); dev_free(c->d_cand_count); dev_free(c->d_cand_d2); dev_free(c->d_cand_idx); dev_free(c->d_ovf); dev_free(c->d_qbin); dev_free(c->d_perm); dev_free(c->d_qsorted); dev_free(c->d_sorttmp); dev_free(c->d_sortkey);
    dev_free(c->d_inv); dev_free(c->d_sres_idx); dev_free(c->d_sres_d2); dev_free(c->d_todo);
    c->qcap = 0;
    c->generation++;                    // captured plans hold these pointers
    PCTCHK(dev_alloc(&c->d_q, 3 * q));
    PCTCHK(dev_alloc(&c->d_r, q));
    PCTCHK(dev_alloc(&c->d_q64, 3 * q));
    PCTCHK(dev_alloc(&c->d_r2, q));
    PCTCHK(dev_alloc(&c->d_d2, q));
    PCTCHK(dev_alloc(&c->d_radius, q));
    PCTCHK(dev_alloc(&c->d_pts64, 3 * q));
    PCTCHK(dev_alloc(&c->d_idx, q));
    PCTCHK(dev_alloc(&c->d_count, q));
    PCTCHK(dev_alloc(&c->d_skip, q));
    PCTCHK(dev_alloc(&c->d_bound, q));
    PCTCHK(dev_alloc(&c->d_qbin, q));
    PCTCHK(dev_alloc(&c->d_perm, q));
    PCTCHK(dev_alloc(&c->d_qsorted, q));
    PCTCHK(dev_alloc(&c->d_sorttmp, q));
    PCTCHK(dev_alloc(&c->d_sortkey, q));
    PCTCHK(dev_alloc(&c->d_inv, q));
    PCTCHK(dev_alloc(&c->d_sres_idx, q));
    PCTCHK(dev_alloc(&c->d_sres_d2, q));
    PCTCHK(dev_alloc(&c->d_todo, 2 * q + 16));
    HIPCHK(hipMemset(c->d_todo, 0, sizeof(uint32_t) * 16));            // count and ticket: the exact-walk kernel leaves them zero after every batch
    if (!c->d_sort1) {
        PCTCHK(dev_alloc(&c->d_sort1, 4 * kSortBuckets + 8));
        HIPCHK(hipMemset(c->d_sort1, 0, sizeof(uint32_t) * (4 * kSortBuckets + 8)));   // the sort keeps both sets of totals zero between batches
    }
    c->part_q = std::min<int64_t>(q, kPartQueries);
    PCTCHK(dev_alloc(&c->d_part_d2, (size_t)c->part_q * kMaxParts));
    PCTCHK(dev_alloc(&c->d_part_idx, (size_t)c->part_q * kMaxParts));
    PCTCHK(dev_alloc(&c->d_ovf, (size_t)c->part_q + 1));
    PCTCHK(dev_alloc(&c->d_cand_count, (size_t)c->part_q));
    PCTCHK(dev_alloc(&c->d_cand_d2, (size_t)c->part_q * kCandCap));
    PCTCHK(dev_alloc(&c->d_cand_idx, (size_t)c->part_q * kCandCap));
    HIPCHK(hipMemset(c->d_cand_count, 0, sizeof(uint32_t) * (size_t)c->part_q));     // the reduce kernel keeps it zero between slices
    c->qcap = q;
    return PCT_OK;
}

// ---- grid ------------------------------------------------------------------------------
int pct_cloud_drop_grid(pct_cloud *c)
{
    if (!c) return fail(PCT_ERR_INVALID, "null cloud");
    drop_grid(c);
    return PCT_OK;
}

int pct_cloud_has_grid(const pct_cloud *c) { return c && c->has_grid ? 1 : 0; }

int pct_cloud_grid_info(const pct_cloud *c, int32_t dims[3], float *cell_size, float origin[3], int64_t *ncells)
{
    if (!c || !c->has_grid) return fail(PCT_ERR_INVALID, "no grid");
    if (dims) { dims[0] = c->G.gx; dims[1] = c->G.gy; dims[2] = c->G.gz; }
    if (cell_size) *cell_size = (float)c->G.hd;
    if (origin) { origin[0] = c->G.ox; origin[1] = c->G.oy; origin[2] = c->G.oz; }
    if (ncells) *ncells = c->G.ncells;
    return PCT_OK;
}

int pct_cloud_build_grid(pct_cloud *c, float cell_size)
{
    if (!c) return fail(PCT_ERR_INVALID, "null cloud");
    if (c->ring_on) return fail(PCT_ERR_INVALID, "this cloud keeps the rolling-map index (pct_cloud_ring_index); drop it before building the cell-sorted one");
    drop_grid(c);
    const int64_t n = c->count;
    if (n == 0) return fail(PCT_ERR_EMPTY, "cannot index an empty cloud");
    hipStream_t s = g_stream;

    // 1. bounding box
    const int bblocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
    PCTCHK(ensure_stage(c, sizeof(float) * (size_t)bblocks * 6));     // the upload staging buffer is idle here: no allocation per build
    float *d_part = reinterpret_cast<float *>(c->d_stage);
    bbox_partial_kernel<<<bblocks, 256, 0, s>>>(c->x, c->y, c->z, (uint32_t)n, d_part);
    std::vector<float> part((size_t)bblocks * 6);
    hipError_t e = hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(float), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(PCT_ERR_HIP, "bbox reduction failed: %s", hipGetErrorString(e));
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int b = 0; b < bblocks; b++)
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], part[(size_t)b * 6 + k]);
            hi[k] = std::max(hi[k], part[(size_t)b * 6 + 3 + k]);
        }
    for (int k = 0; k < 3; k++)
        if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) return fail(PCT_ERR_INVALID, "cloud holds non-finite coordinates");

    // 2. cell size: about 2 points per cell over the occupied bounding box, dims capped at 1024/axis
    double ext[3];
    for (int k = 0; k < 3; k++) ext[k] = std::max((double)hi[k] - (double)lo[k], 0.0);
    double h = cell_size;
    if (!(h > 0)) {
        // target points per cell: with the 2x2x2-block-first search (kernels.hpp coop_nn_search stage 0) larger cells win --
        // the block then decides 99 % of the queries (a wave needs all 8 of its queries decided to skip the cube);
        // same-box A/B on the 10 M uniform cloud: ppc 2 + cube first 0.169 ms, ppc 6 + block first 0.134 ms per 1 M queries
        double ppc = 6.0;
        // a cloud that carried the pyramid at its previous build (surfaces, clusters: most cells empty) is rebuilt with cells of
        // half the volume: the walk then scans 52 instead of 89 points per query for one more node visit (10 M pillar surfaces:
        // 0.99 -> 1.10e9 q/s, scripts/probe_pyr.py with PCT_GRID_PPC 3 / 4 / 6 / 9 / 12).  The per-frame rebuild of a sensor cloud
        // (corridor_finder.cpp:93-99) sees the same kind of cloud frame after frame, so the first build's verdict serves the next.
        if (c->was_sparse) ppc = 3.0;
        if (const char *e = std::getenv("PCT_GRID_PPC")) ppc = std::max(0.05, std::atof(e));
        const double diag = std::max({ ext[0], ext[1], ext[2], 1e-6 });
        double vol = 1.0;
        for (int k = 0; k < 3; k++) vol *= std::max(ext[k], diag * 1e-3);
        h = std::cbrt(vol * ppc / (double)n);
    }
    const double max_ext = std::max({ ext[0], ext[1], ext[2] });
    h = std::max(h, max_ext / 1023.0);
    if (!(h > 0)) h = 1.0;   // all points identical
    const float hf = (float)h;
    GridDesc G{};
    G.ox = lo[0]; G.oy = lo[1]; G.oz = lo[2];
    G.inv_h = 1.0f / hf;
    G.oxd = lo[0]; G.oyd = lo[1]; G.ozd = lo[2];
    G.hd = (double)hf;
    G.gx = std::max(1, std::min(1024, (int)std::floor(ext[0] / G.hd) + 1));
    G.gy = std::max(1, std::min(1024, (int)std::floor(ext[1] / G.hd) + 1));
    G.gz = std::max(1, std::min(1024, (int)std::floor(ext[2] / G.hd) + 1));
    const uint64_t ncells = (uint64_t)G.gx * G.gy * G.gz;
    if (ncells > 0x7FFFFFF0ull) return fail(PCT_ERR_INVALID, "grid of %llu cells is too large", (unsigned long long)ncells);
    G.ncells = (uint32_t)ncells;

    // 3. counting sort
    PCTCHK(sort_into_cells(c, G, &c->cell_start, &c->cells_cap, &c->sorted, &c->sorted_cap));
    PCTCHK(build_pyramid(c, G));
    // 4. sparse occupancy (points on surfaces): add coarser levels so free-space queries do not walk empty fine shells
    c->C.n = 0;
    {
        // OFF by default (threshold > 1): measured on the seed-6 pillar map the coarse cubes cut the cell rows visited per
        // free-space query from 307 to 17 but raise the points examined from 869 to 18 000 (a cube four times wider holds
        // sixteen times more SURFACE points), 3.5x slower overall -- walking fine shells is the better trade for surface
        // clouds.  Kept selectable (PCT_PYRAMID_EMPTY_FRAC=0.5) and covered by tests for volumetric sparse clouds.
        double sparse_at = 2.0;
        if (const char *ev = std::getenv("PCT_PYRAMID_EMPTY_FRAC")) sparse_at = std::atof(ev);
        uint32_t n_empty = 0;
        if (sparse_at < 1.0) {                 // the occupancy count (a launch + a host round trip) only when the levels can be asked for
            uint32_t *d_empty = nullptr;
            PCTCHK(dev_alloc(&d_empty, 1));
            hipError_t e2 = hipMemsetAsync(d_empty, 0, sizeof(uint32_t), s);
            count_empty_cells_kernel<<<(int)std::min<uint64_t>(1024, (ncells + 255) / 256), 256, 0, s>>>(c->cell_start, (uint32_t)ncells, d_empty);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(&n_empty, d_empty, sizeof n_empty, hipMemcpyDeviceToHost, s);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(s);
            dev_free(d_empty);
            if (e2 != hipSuccess) return fail(PCT_ERR_HIP, "occupancy count failed: %s", hipGetErrorString(e2));
        }
        if ((double)n_empty > sparse_at * (double)ncells && ncells > 512) {
            GridDesc L = G;
            for (int l = 0; l < kMaxCoarse && std::max({ L.gx, L.gy, L.gz }) > 3; l++) {
                L.hd *= 4.0;
                L.inv_h = (float)(1.0 / L.hd);
                L.gx = (L.gx + 3) / 4; L.gy = (L.gy + 3) / 4; L.gz = (L.gz + 3) / 4;
                L.ncells = (uint32_t)L.gx * (uint32_t)L.gy * (uint32_t)L.gz;
                PCTCHK(sort_into_cells(c, L, &c->coarse_cell_start[l], &c->coarse_cells_cap[l], &c->coarse_sorted[l], &c->coarse_sorted_cap[l]));
                c->C.G[l] = L;
                c->C.pts[l] = c->coarse_sorted[l];
                c->C.cell_start[l] = c->coarse_cell_start[l];
                c->C.n = l + 1;
            }
        }
    }
    G.octant_first = 1;
    if (const char *eo = std::getenv("PCT_OCTANT_FIRST")) G.octant_first = std::atoi(eo) != 0;
    // block table for stage 0 of the dense batch kernel (kernels.hpp block_corner_kernel).  OFF by default -- measured slower on the
    // headline step (0.126-0.131 ms against 0.120-0.121, profiles/r03_ab_block_table.txt): the 6.7 MB cell table it replaces is served
    // by the L2s, the 56 MB of corner entries are not, and one line that misses costs more than four that hit.  PCT_BLOCK_TABLE=1
    // (read at every build) keeps it selectable and tested.
    G.blocks = nullptr;
    {
        const char *et = std::getenv("PCT_BLOCK_TABLE");
        const bool table_on = et ? std::atoi(et) != 0 : false;
        const uint64_t ncorners = (uint64_t)(G.gx + 1) * (uint64_t)(G.gy + 1) * (uint64_t)(G.gz + 1);
        if (table_on && G.octant_first && !c->has_pyr && ncorners < 0x7FFFFFF0ull) {
            if ((size_t)(2 * ncorners) > c->blocks_cap) {
                dev_free(c->blocks);
                c->blocks_cap = 0;
                PCTCHK(dev_alloc(&c->blocks, (size_t)(2 * ncorners)));
                c->blocks_cap = (size_t)(2 * ncorners);
            }
            block_corner_kernel<<<ceil_div((int64_t)ncorners, 256), 256, 0, s>>>(G, c->cell_start, c->blocks, (uint32_t)ncorners);
            HIPCHK(hipGetLastError());
            PCTCHK(note_mutation(c));            // queued, not awaited: a *_dev call on another stream waits on the event
            G.blocks = c->blocks;
        }
    }
    c->G = G;
    // query bins: (2^shift)^3 cells each
    BinDesc B{};
    B.shift = 1;
    if (const char *eb = std::getenv("PCT_BIN_SHIFT")) B.shift = std::max(0, std::min(4, std::atoi(eb)));
    B.bx = ((G.gx - 1) >> B.shift) + 1;
    B.by = ((G.gy - 1) >> B.shift) + 1;
    B.bz = ((G.gz - 1) >> B.shift) + 1;
    B.strip = 16;
    if (const char *es = std::getenv("PCT_BIN_STRIP")) B.strip = std::max(1, std::atoi(es));
    B.strip = std::min(B.strip, B.by);
    B.nbins = (uint32_t)B.bx * (uint32_t)(((B.by + B.strip - 1) / B.strip) * B.strip) * (uint32_t)B.bz;
    if ((size_t)B.nbins + 1 > c->bins_cap) {
        dev_free(c->bin_start); dev_free(c->bin_fill); dev_free(c->bin_tiles);
        c->bins_cap = 0;
        PCTCHK(dev_alloc(&c->bin_start, (size_t)B.nbins + 1));
        PCTCHK(dev_alloc(&c->bin_fill, (size_t)B.nbins));
        PCTCHK(dev_alloc(&c->bin_tiles, (size_t)(B.nbins + kScanTile - 1) / kScanTile));
        c->bins_cap = (size_t)B.nbins + 1;
    }
    c->B = B;
    c->has_grid = true;
    c->generation++;
    return PCT_OK;
}

// ---- device-buffer entry points --------------------------------------------------------
int pct_nn_batch_dev(pct_cloud *c, int algo, const float *d_q, int64_t Q, uint32_t *d_idx, double *d_d2, void *stream)
{
    if (!c || Q < 0 || (Q > 0 && (!d_q || !d_idx || !d_d2))) return fail(PCT_ERR_INVALID, "bad nn_batch_dev arguments");
    PCTCHK(order_after_mutations(c, (hipStream_t)stream));
    return nn_dev(c, algo, d_q, Q, d_idx, d_d2, (hipStream_t)stream);   // NULL = HIP's null stream (torch's default stream)
}

int pct_radius_count_batch_dev(pct_cloud *c, int algo, const float *d_q, const float *d_r, int64_t Q, uint32_t *d_count, void *stream)
{
    if (!c || Q < 0 || (Q > 0 && (!d_q || !d_r || !d_count))) return fail(PCT_ERR_INVALID, "bad radius_count_batch_dev arguments");
    PCTCHK(order_after_mutations(c, (hipStream_t)stream));
    return count_dev(c, algo, d_q, d_r, Q, d_count, (hipStream_t)stream);
}

// ---- host-buffer entry points ----------------------------------------------------------
int pct_nn_batch_algo(pct_cloud *c, int algo, const float *q, int64_t Q, uint32_t *idx, double *d2)
{
    if (!c || Q < 0 || (Q > 0 && (!q || !idx || !d2))) return fail(PCT_ERR_INVALID, "bad nn_batch arguments");
    if (Q == 0) return PCT_OK;
    if (Q <= kExpressMaxQ && c->ring_ready && c->count > 0 && (algo == PCT_ALGO_AUTO || algo == PCT_ALGO_GRID)) {
        // small batch on the rolling map: one launch, a block per query, arguments/results in mapped memory
        for (int64_t i = 0; i < 3 * Q; i++) c->h_xin[i] = (double)q[i];
        ring_batch_kernel<false><<<(int)Q, 256, 0, g_stream>>>(ring_view(c), InflateParams{}, nullptr, c->d_xin, (double)INFINITY, (uint32_t)c->index_base,
                                                               nullptr, nullptr, nullptr, c->d_xout, next_signal(c));
        HIPCHK(hipGetLastError());
        PCTCHK(express_wait(c));
        for (int64_t i = 0; i < Q; i++) { idx[i] = c->h_xout[i].idx; d2[i] = c->h_xout[i].d2; }
        return PCT_OK;
    }
    if (Q <= kExpressMaxQ && c->has_grid && c->count > 0 && (algo == PCT_ALGO_AUTO || algo == PCT_ALGO_GRID)) {
        // small batch on an indexed cloud: one launch, a block per query, arguments/results in mapped memory
        for (int64_t i = 0; i < 3 * Q; i++) c->h_xin[i] = (double)q[i];
        inflate_block_kernel<false><<<(int)Q, 256, 0, g_stream>>>(c->G, c->sorted, c->cell_start, c->C, InflateParams{}, c->d_xin, (double)INFINITY,
                                                                   (uint32_t)c->index_base, c->d_xout, next_signal(c));
        HIPCHK(hipGetLastError());
        PCTCHK(express_wait(c));
        for (int64_t i = 0; i < Q; i++) { idx[i] = c->h_xout[i].idx; d2[i] = c->h_xout[i].d2; }
        return PCT_OK;
    }
    PCTCHK(pct_cloud_reserve_queries(c, Q));
    if (Q <= kMappedMaxQ && mapped_io_on()) {
        // queries imported from host-mapped memory by a kernel, results exported by a kernel, completion by polling:
        // no DMA copy, no stream synchronise (C2's 4096-query batch: 0.63 -> 0.5x ms)
        PCTCHK(ensure_mapped_io(c, Q));
        std::memcpy(c->h_mq, q, sizeof(float) * 3 * (size_t)Q);
        import_floats_kernel<<<ceil_div(3 * Q, 256), 256, 0, g_stream>>>(c->d_mq, (uint32_t)(3 * Q), c->d_q);
        PCTCHK(nn_dev(c, algo, c->d_q, Q, c->d_idx, c->d_d2, g_stream));
        export_results_kernel<<<ceil_div(Q, 256), 256, 0, g_stream>>>(c->d_idx, c->d_d2, (uint32_t)Q, c->d_mi, c->d_md, next_signal(c));
        HIPCHK(hipGetLastError());
        PCTCHK(express_wait(c));
        std::memcpy(idx, c->h_mi, sizeof(uint32_t) * (size_t)Q);
        std::memcpy(d2, c->h_md, sizeof(double) * (size_t)Q);
        if (c->count == 0) return fail(PCT_ERR_EMPTY, "nearest-neighbour query against an empty cloud");
        return PCT_OK;
    }
    HIPCHK(hipMemcpyAsync(c->d_q, q, sizeof(float) * 3 * Q, hipMemcpyHostToDevice, g_stream));
    PCTCHK(nn_dev(c, algo, c->d_q, Q, c->d_idx, c->d_d2, g_stream));
    HIPCHK(hipMemcpyAsync(idx, c->d_idx, sizeof(uint32_t) * Q, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemcpyAsync(d2, c->d_d2, sizeof(double) * Q, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    if (c->count == 0) return fail(PCT_ERR_EMPTY, "nearest-neighbour query against an empty cloud");
    return PCT_OK;
}

int pct_nn_batch(pct_cloud *c, const float *q, int64_t Q, uint32_t *idx, double *d2)
{
    return pct_nn_batch_algo(c, PCT_ALGO_AUTO, q, Q, idx, d2);
}

int pct_nn_batch_q64(pct_cloud *c, const double *q, int64_t Q, uint32_t *idx, double *d2)
{
    return pct_nn_batch_q64_ties(c, q, Q, idx, d2, nullptr);
}

int pct_nn_batch_q64_ties(pct_cloud *c, const double *q, int64_t Q, uint32_t *idx, double *d2, uint32_t *ties)
{
    if (!c || Q < 0 || (Q > 0 && (!q || !idx || !d2))) return fail(PCT_ERR_INVALID, "bad nn_batch_q64 arguments");
    if (Q == 0) return PCT_OK;
    if (ties) for (int64_t i = 0; i < Q; i++) ties[i] = 0;        // 0 = not counted on this path
    PCTCHK(pct_cloud_reserve_queries(c, Q));
    if (c->count == 0) {
        for (int64_t i = 0; i < Q; i++) { idx[i] = PCT_NO_INDEX; d2[i] = INFINITY; }
        return fail(PCT_ERR_EMPTY, "nearest-neighbour query against an empty cloud");
    }
    if (Q > 1 && Q <= kExpressMaxQ && c->count <= kSmallNNMax) {   // express batch: a block per query, everything in mapped memory
        std::memcpy(c->h_xin, q, sizeof(double) * 3 * Q);
        nn_small_batch_kernel<<<(int)Q, 256, 0, g_stream>>>(c->x, c->y, c->z, (uint32_t)c->count, c->d_xin, (uint32_t)c->index_base, c->d_xout, next_signal(c));
        HIPCHK(hipGetLastError());
        PCTCHK(express_wait(c));
        for (int64_t i = 0; i < Q; i++) { idx[i] = c->h_xout[i].idx; d2[i] = c->h_xout[i].d2; }
        return PCT_OK;
    }
    if (Q == 1 && c->count <= kSmallNNMax) {   // express: one one-block launch, query by value, result in mapped memory
        nn_small_kernel<<<1, 1024, 0, g_stream>>>(c->x, c->y, c->z, (uint32_t)c->count, q[0], q[1], q[2], (uint32_t)c->index_base, c->d_xout, next_signal(c));
        HIPCHK(hipGetLastError());
        PCTCHK(express_wait(c));
        idx[0] = c->h_xout[0].idx;
        d2[0] = c->h_xout[0].d2;
        if (ties) ties[0] = c->h_xout[0].count;
        return PCT_OK;
    }
    // The fp32 filter is only valid when the query coordinates themselves are fp32 values
    // (always the case for kd_nearestf); genuinely double queries take the all-fp64 kernel.
    bool f32_exact = true;
    for (int64_t i = 0; i < 3 * Q && f32_exact; i++) f32_exact = (double)(float)q[i] == q[i];
    HIPCHK(hipMemcpyAsync(c->d_q64, q, sizeof(double) * 3 * Q, hipMemcpyHostToDevice, g_stream));
    if (f32_exact && Q > 4) {
        std::vector<float> qf((size_t)3 * Q);
        for (int64_t i = 0; i < 3 * Q; i++) qf[i] = (float)q[i];
        HIPCHK(hipMemcpyAsync(c->d_q, qf.data(), sizeof(float) * 3 * Q, hipMemcpyHostToDevice, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));      // qf goes out of scope below
        PCTCHK(nn_stream_filtered(c, c->d_q, Q, c->d_idx, c->d_d2, g_stream));
    } else {
        PCTCHK(nn_stream_q64(c, Q, c->d_idx, c->d_d2, g_stream));
    }
    HIPCHK(hipMemcpyAsync(idx, c->d_idx, sizeof(uint32_t) * Q, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemcpyAsync(d2, c->d_d2, sizeof(double) * Q, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return PCT_OK;
}

int pct_radius_count_batch_algo(pct_cloud *c, int algo, const float *q, const float *r, int64_t Q, uint32_t *count)
{
    if (!c || Q < 0 || (Q > 0 && (!q || !r || !count))) return fail(PCT_ERR_INVALID, "bad radius_count arguments");
    if (Q == 0) return PCT_OK;
    PCTCHK(pct_cloud_reserve_queries(c, Q));
    if (Q <= kMappedMaxQ && mapped_io_on()) {          // as pct_nn_batch_algo: kernels move the arguments and the counts, the host polls
        PCTCHK(ensure_mapped_io(c, Q));
        std::memcpy(c->h_mq, q, sizeof(float) * 3 * (size_t)Q);
        std::memcpy(c->h_mq + 3 * Q, r, sizeof(float) * (size_t)Q);
        import_floats_kernel<<<ceil_div(3 * Q, 256), 256, 0, g_stream>>>(c->d_mq, (uint32_t)(3 * Q), c->d_q);
        import_floats_kernel<<<ceil_div(Q, 256), 256, 0, g_stream>>>(c->d_mq + 3 * Q, (uint32_t)Q, c->d_r);
        PCTCHK(count_dev(c, algo, c->d_q, c->d_r, Q, c->d_count, g_stream));
        export_results_kernel<<<ceil_div(Q, 256), 256, 0, g_stream>>>(c->d_count, nullptr, (uint32_t)Q, c->d_mi, nullptr, next_signal(c));
        HIPCHK(hipGetLastError());
        PCTCHK(express_wait(c));
        std::memcpy(count, c->h_mi, sizeof(uint32_t) * (size_t)Q);
        return PCT_OK;
    }
    HIPCHK(hipMemcpyAsync(c->d_q, q, sizeof(float) * 3 * Q, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(c->d_r, r, sizeof(float) * Q, hipMemcpyHostToDevice, g_stream));
    PCTCHK(count_dev(c, algo, c->d_q, c->d_r, Q, c->d_count, g_stream));
    HIPCHK(hipMemcpyAsync(count, c->d_count, sizeof(uint32_t) * Q, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return PCT_OK;
}

int pct_radius_count_batch(pct_cloud *c, const float *q, const float *r, int64_t Q, uint32_t *count)
{
    return pct_radius_count_batch_algo(c, PCT_ALGO_AUTO, q, r, Q, count);
}

// order-preserving compaction of the points within r of q into c->crop_* (kernels.hpp section 1b)
static int crop_device(pct_cloud *c, const double q[3], double rr, int64_t *total_out)
{
    const uint32_t n = (uint32_t)c->count;
    const uint32_t ntiles = (n + kCropTile - 1) / kCropTile;
    if ((size_t)ntiles + 1 > c->crop_tiles_cap) {
        dev_free(c->crop_tile);
        c->crop_tiles_cap = 0;
        PCTCHK(dev_alloc(&c->crop_tile, (size_t)ntiles + 1));
        c->crop_tiles_cap = (size_t)ntiles + 1;
    }
    if ((size_t)n > c->crop_cap) {
        dev_free(c->crop_idx); dev_free(c->crop_d2); dev_free(c->crop_x); dev_free(c->crop_y); dev_free(c->crop_z);
        c->crop_cap = 0;
        const size_t cap = std::max<size_t>((size_t)c->cap, n);
        PCTCHK(dev_alloc(&c->crop_idx, cap)); PCTCHK(dev_alloc(&c->crop_d2, cap));
        PCTCHK(dev_alloc(&c->crop_x, cap)); PCTCHK(dev_alloc(&c->crop_y, cap)); PCTCHK(dev_alloc(&c->crop_z, cap));
        c->crop_cap = cap;
    }
    begin_timing(c, g_stream);
    HIPCHK(hipMemsetAsync(c->crop_tile + ntiles, 0, sizeof(uint32_t), g_stream));
    crop_count_kernel<<<(int)ntiles, 256, 0, g_stream>>>(c->x, c->y, c->z, n, q[0], q[1], q[2], rr, c->crop_tile);
    scan_tile_sums_kernel<<<1, 256, 0, g_stream>>>(c->crop_tile, ntiles + 1);       // entry ntiles becomes the grand total
    crop_scatter_kernel<<<(int)ntiles, 256, 0, g_stream>>>(c->x, c->y, c->z, n, q[0], q[1], q[2], rr, (uint32_t)c->index_base, c->crop_tile,
                                                          (uint32_t)c->crop_cap, c->crop_idx, c->crop_d2, c->crop_x, c->crop_y, c->crop_z);
    end_timing(c, g_stream);
    HIPCHK(hipGetLastError());
    uint32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, c->crop_tile + ntiles, sizeof total, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    *total_out = total;
    return PCT_OK;
}

int pct_radius_indices(pct_cloud *c, const float q[3], float r, uint32_t *idx_out, int64_t cap, int64_t *n_out)
{
    if (!q) return fail(PCT_ERR_INVALID, "bad radius_indices arguments");
    const double qd[3] = { (double)q[0], (double)q[1], (double)q[2] };
    return pct_radius_indices_q64(c, qd, (double)r, idx_out, cap, n_out);
}

int pct_radius_indices_q64(pct_cloud *c, const double q[3], double r, uint32_t *idx_out, int64_t cap, int64_t *n_out)
{
    return pct_radius_indices_r2_q64(c, q, r * r, idx_out, cap, n_out);
}

int pct_radius_indices_r2_q64(pct_cloud *c, const double q[3], double r2, uint32_t *idx_out, int64_t cap, int64_t *n_out)
{
    if (!c || !q || cap < 0 || (cap > 0 && !idx_out) || !n_out) return fail(PCT_ERR_INVALID, "bad radius_indices arguments");
    *n_out = 0;
    if (c->count == 0) return PCT_OK;
    if (c->count <= 4 * kSmallNNMax && c->count <= (int64_t)kExpressIdsCap) {   // express: one launch, ids in mapped memory
        radius_small_kernel<<<1, 1024, 0, g_stream>>>(c->x, c->y, c->z, (uint32_t)c->count, q[0], q[1], q[2], r2, (uint32_t)c->index_base,
                                                       c->d_xids, kExpressIdsCap, c->d_xout, next_signal(c));
        HIPCHK(hipGetLastError());
        PCTCHK(express_wait(c));
        const int64_t total = c->h_xout[0].count, got = std::min<int64_t>(total, cap);
        std::copy(c->h_xids, c->h_xids + std::min<int64_t>(got, kExpressIdsCap), idx_out);
        std::sort(idx_out, idx_out + got);
        *n_out = total;
        return PCT_OK;
    }
    int64_t total = 0;
    PCTCHK(crop_device(c, q, r2, &total));                 // ascending index order, no host sort needed
    const int64_t got = std::min<int64_t>(total, cap);
    if (got > 0) {
        HIPCHK(hipMemcpyAsync(idx_out, c->crop_idx, sizeof(uint32_t) * got, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
    }
    *n_out = total;
    return PCT_OK;
}

// lidar crop with everything its consumer builds from it (camera_sensor.cpp:133-145, 398-401)
int pct_radius_crop(pct_cloud *c, const double q[3], double r, int sort_by_distance, int64_t cap, uint32_t *idx_out, double *d2_out,
                    float *xyz_out, int64_t *n_out)
{
    if (!c || !q || cap < 0 || !n_out) return fail(PCT_ERR_INVALID, "bad radius_crop arguments");
    *n_out = 0;
    if (c->count == 0) return PCT_OK;
    int64_t total = 0;
    PCTCHK(crop_device(c, q, r * r, &total));
    *n_out = total;
    const int64_t got = std::min<int64_t>(total, cap);
    if (got == 0) return PCT_OK;
    if (sort_by_distance && got < total) return fail(PCT_ERR_CAPACITY, "a distance-sorted crop needs room for all %lld hits (cap %lld)", (long long)total, (long long)cap);
    std::vector<uint32_t> hi((size_t)got);
    std::vector<double> hd((size_t)got);
    std::vector<float> hx, hy, hz;
    HIPCHK(hipMemcpyAsync(hi.data(), c->crop_idx, sizeof(uint32_t) * got, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemcpyAsync(hd.data(), c->crop_d2, sizeof(double) * got, hipMemcpyDeviceToHost, g_stream));
    if (xyz_out) {
        hx.resize((size_t)got); hy.resize((size_t)got); hz.resize((size_t)got);
        HIPCHK(hipMemcpyAsync(hx.data(), c->crop_x, sizeof(float) * got, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipMemcpyAsync(hy.data(), c->crop_y, sizeof(float) * got, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipMemcpyAsync(hz.data(), c->crop_z, sizeof(float) * got, hipMemcpyDeviceToHost, g_stream));
    }
    HIPCHK(hipStreamSynchronize(g_stream));
    std::vector<uint32_t> order((size_t)got);
    for (int64_t i = 0; i < got; i++) order[(size_t)i] = (uint32_t)i;
    if (sort_by_distance)     // PCL hands back radiusSearch results nearest first; ties keep ascending index (stable)
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hd[a] < hd[b]; });
    for (int64_t i = 0; i < got; i++) {
        const uint32_t k = order[(size_t)i];
        if (idx_out) idx_out[i] = hi[k];
        if (d2_out) d2_out[i] = hd[k];
        if (xyz_out) { xyz_out[3 * i] = hx[k]; xyz_out[3 * i + 1] = hy[k]; xyz_out[3 * i + 2] = hz[k]; }
    }
    return PCT_OK;
}

// dst = the points of src within r of q, in src's order, device to device (known_map_pcl of camera_sensor.cpp:398-401)
int pct_cloud_crop_to(pct_cloud *src, const double q[3], double r, pct_cloud *dst)
{
    if (!src || !dst || !q || src == dst) return fail(PCT_ERR_INVALID, "bad crop_to arguments");
    int64_t total = 0;
    if (src->count) PCTCHK(crop_device(src, q, r * r, &total));
    if (total > dst->cap) return fail(PCT_ERR_CAPACITY, "crop holds %lld points, destination capacity is %lld", (long long)total, (long long)dst->cap);
    return pct_cloud_upload_soa_dev(dst, src->crop_x, src->crop_y, src->crop_z, total);
}

// K range queries against a small cloud in one launch: ids_out[k * cap_per_query + j] (arrival order, not sorted),
// counts_out[k] = number of hits of query k (may exceed cap_per_query: then only the first cap_per_query are stored).
int pct_radius_indices_batch_q64(pct_cloud *c, const double *q, const double *r, int64_t K, uint32_t *ids_out, int64_t cap_per_query,
                                 int64_t *counts_out)
{
    if (!c || K < 0 || (K > 0 && (!q || !r || !ids_out || !counts_out)) || cap_per_query <= 0) return fail(PCT_ERR_INVALID, "bad radius_indices_batch arguments");
    if (K == 0) return PCT_OK;
    if (K > kExpressMaxQ || c->count > 4 * kSmallNNMax) return fail(PCT_ERR_INVALID, "batched range queries serve small clouds (<= 65536 points) and K <= 1024");
    for (int64_t k = 0; k < K; k++) counts_out[k] = 0;
    if (c->count == 0) return PCT_OK;
    const uint32_t cap = (uint32_t)std::min<int64_t>(cap_per_query, kExpressIdsCap / K);
    std::memcpy(c->h_xin, q, sizeof(double) * 3 * K);
    std::memcpy(c->h_xr, r, sizeof(double) * K);
    radius_small_batch_kernel<<<(int)K, 256, 0, g_stream>>>(c->x, c->y, c->z, (uint32_t)c->count, c->d_xin, c->d_xr, (uint32_t)c->index_base,
                                                            c->d_xids, cap, c->d_xout, next_signal(c));
    HIPCHK(hipGetLastError());
    PCTCHK(express_wait(c));
    for (int64_t k = 0; k < K; k++) {
        counts_out[k] = c->h_xout[k].count;
        const int64_t got = std::min<int64_t>(counts_out[k], cap);
        std::copy(c->h_xids + k * cap, c->h_xids + k * cap + got, ids_out + k * cap_per_query);
        if (counts_out[k] > cap) counts_out[k] = -counts_out[k];      // negative = truncated: the caller must re-ask that query alone
    }
    return PCT_OK;
}

int pct_inflate_batch(pct_cloud *c, const pct_inflate_params *p, const double *pts, int64_t Q, double *radius, uint32_t *idx, double *d2)
{
    if (!c || !p || Q < 0 || (Q > 0 && (!pts || !radius))) return fail(PCT_ERR_INVALID, "bad inflate arguments");
    if (Q == 0) return PCT_OK;
    if (c->ring_ready) {          // rolling map: a block per point over the bucket table; small batches through mapped memory
        const double reach = p->max_radius + p->search_margin;
        const double stop_d2 = (idx || d2) ? (double)INFINITY : reach * reach;
        if (Q <= kExpressMaxQ) {
            std::memcpy(c->h_xin, pts, sizeof(double) * 3 * Q);
            ring_batch_kernel<true><<<(int)Q, 256, 0, g_stream>>>(ring_view(c), to_dev(p), nullptr, c->d_xin, stop_d2, (uint32_t)c->index_base, nullptr, nullptr,
                                                                  nullptr, c->d_xout, next_signal(c));
            HIPCHK(hipGetLastError());
            PCTCHK(express_wait(c));
            for (int64_t i = 0; i < Q; i++) {
                radius[i] = c->h_xout[i].radius;
                if (idx) idx[i] = c->h_xout[i].idx;
                if (d2) d2[i] = c->h_xout[i].d2;
            }
            return PCT_OK;
        }
        PCTCHK(pct_cloud_reserve_queries(c, Q));
        HIPCHK(hipMemcpyAsync(c->d_pts64, pts, sizeof(double) * 3 * Q, hipMemcpyHostToDevice, g_stream));
        ring_batch_kernel<true><<<(int)Q, 256, 0, g_stream>>>(ring_view(c), to_dev(p), nullptr, c->d_pts64, stop_d2, (uint32_t)c->index_base, c->d_idx, c->d_d2,
                                                              c->d_radius, nullptr, ExpressSignal{});
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(radius, c->d_radius, sizeof(double) * Q, hipMemcpyDeviceToHost, g_stream));
        if (idx) HIPCHK(hipMemcpyAsync(idx, c->d_idx, sizeof(uint32_t) * Q, hipMemcpyDeviceToHost, g_stream));
        if (d2) HIPCHK(hipMemcpyAsync(d2, c->d_d2, sizeof(double) * Q, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        return PCT_OK;
    }
    if (Q <= kExpressMaxQ && c->has_grid && c->count > 0) {   // express: one fused launch (a block per point), arguments and results in mapped memory
        std::memcpy(c->h_xin, pts, sizeof(double) * 3 * Q);
        // idx / d2 not wanted: the search may stop once everything unseen is beyond max_radius + search_margin
        const double reach = p->max_radius + p->search_margin;
        const double stop_d2 = (idx || d2) ? (double)INFINITY : reach * reach;
        inflate_block_kernel<true><<<(int)Q, 256, 0, g_stream>>>(c->G, c->sorted, c->cell_start, c->C, to_dev(p), c->d_xin, stop_d2, (uint32_t)c->index_base, c->d_xout, next_signal(c));
        HIPCHK(hipGetLastError());
        PCTCHK(express_wait(c));
        for (int64_t i = 0; i < Q; i++) {
            radius[i] = c->h_xout[i].radius;
            if (idx) idx[i] = c->h_xout[i].idx;
            if (d2) d2[i] = c->h_xout[i].d2;
        }
        return PCT_OK;
    }
    PCTCHK(pct_cloud_reserve_queries(c, Q));
    if (Q <= kExpressMaxQ) {      // small batch on an un-indexed (e.g. rolling) cloud: brute-force kernels, arguments and results in mapped memory
        std::memcpy(c->h_xin, pts, sizeof(double) * 3 * Q);
        PCTCHK(inflate_dev(c, p, Q, g_stream, c->d_xin, c->d_xout));
        HIPCHK(hipStreamSynchronize(g_stream));
        for (int64_t i = 0; i < Q; i++) {
            radius[i] = c->h_xout[i].radius;
            if (idx) idx[i] = c->h_xout[i].idx;
            if (d2) d2[i] = c->h_xout[i].d2;
        }
        return PCT_OK;
    }
    HIPCHK(hipMemcpyAsync(c->d_pts64, pts, sizeof(double) * 3 * Q, hipMemcpyHostToDevice, g_stream));
    PCTCHK(inflate_dev(c, p, Q, g_stream));
    HIPCHK(hipMemcpyAsync(radius, c->d_radius, sizeof(double) * Q, hipMemcpyDeviceToHost, g_stream));
    if (idx) HIPCHK(hipMemcpyAsync(idx, c->d_idx, sizeof(uint32_t) * Q, hipMemcpyDeviceToHost, g_stream));
    if (d2) HIPCHK(hipMemcpyAsync(d2, c->d_d2, sizeof(double) * Q, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return PCT_OK;
}

// ---- fused RRT* expansion step (kernels.hpp rrt_expand_kernel) ----------------------------------------------------------
int pct_cloud_small_aux(pct_cloud *nodes, double **host_aux)
{
    if (!nodes || !host_aux) return fail(PCT_ERR_INVALID, "null argument");
    if (!nodes->host_mapped) return fail(PCT_ERR_INVALID, "per-node planner data lives beside small (host-mapped) clouds only");
    if (!nodes->h_aux) PCTCHK(mapped_alloc(&nodes->h_aux, &nodes->d_aux, (size_t)4 * (size_t)nodes->cap));
    *host_aux = nodes->h_aux;
    return PCT_OK;
}

int pct_rrt_expand_batch(pct_cloud *nodes, pct_cloud *obstacles, const pct_inflate_params *p, const double *samples, int64_t K,
                         int64_t cap_per_query, pct_expand_result *out, uint32_t *ids)
{
    if (!nodes || !obstacles || !p || K < 0 || (K > 0 && (!samples || !out || !ids)) || cap_per_query <= 0) return fail(PCT_ERR_INVALID, "bad expand arguments");
    if (K == 0) return PCT_OK;
    if (K > kExpressMaxQ) return fail(PCT_ERR_INVALID, "at most %d samples per expansion launch", kExpressMaxQ);
    if (!nodes->host_mapped || (nodes->count > 0 && !nodes->h_aux)) return fail(PCT_ERR_INVALID, "the node set must be a small cloud with per-node planner data (pct_cloud_small_aux)");
    if (obstacles->count > 0 && !obstacles->has_grid) return fail(PCT_ERR_INVALID, "the obstacle cloud needs its cell index (pct_cloud_build_grid)");
    if (!nodes->h_eout) PCTCHK(mapped_alloc(&nodes->h_eout, &nodes->d_eout, (size_t)kExpressMaxQ));
    const uint32_t cap = (uint32_t)std::min<int64_t>(cap_per_query, kExpressIdsCap / K);
    std::memcpy(nodes->h_xin, samples, sizeof(double) * 3 * K);
    const double reach = p->max_radius + p->search_margin;       // only the radius is wanted: stop once everything unseen is beyond it
    rrt_expand_kernel<<<(int)K, 256, 0, g_stream>>>(nodes->x, nodes->y, nodes->z, (uint32_t)nodes->count, nodes->d_aux, nodes->d_xin,
                                                    obstacles->G, obstacles->sorted, obstacles->cell_start, obstacles->C,
                                                    obstacles->count == 0 ? 1 : 0, to_dev(p), reach * reach, nodes->d_xids, cap, nodes->d_eout,
                                                    K <= 8 ? next_signal(nodes) : ExpressSignal{});
    HIPCHK(hipGetLastError());
    // one or a few samples: the completion word (1500 one-sample iterations 37 -> 32 ms); speculative batches of 16-256 blocks:
    // a system-scope fence per block costs more than the stream synchronise saves (2.85 vs 2.60 ms per 1500 iterations at K = 64)
    if (K <= 8) PCTCHK(express_wait(nodes));
    else HIPCHK(hipStreamSynchronize(g_stream));
    for (int64_t k = 0; k < K; k++) {
        const ExpandOut &e = nodes->h_eout[k];
        out[k].center[0] = e.cx; out[k].center[1] = e.cy; out[k].center[2] = e.cz;
        out[k].radius = e.radius;
        out[k].near_idx = e.near_idx == kNoIndex ? -1 : (int32_t)e.near_idx;
        const int64_t got = std::min<int64_t>(e.count, cap);
        out[k].count = e.count > cap ? -(int32_t)e.count : (int32_t)e.count;     // negative = list truncated: ask that range query alone
        std::copy(nodes->h_xids + k * cap, nodes->h_xids + k * cap + got, ids + k * cap_per_query);
    }
    return PCT_OK;
}

int pct_bezier_check(pct_cloud *c, const pct_bezier_traj *traj, const pct_inflate_params *p, double t_start, double stop_time, double dt,
                     int64_t *first_hit, int64_t *nsamples, int64_t cap, double *pos, double *radius, double *d2, uint32_t *idx)
{
    if (!c || !traj || !p || !first_hit || !nsamples || !traj->polycoef || !traj->seg_time || !traj->orders || traj->nseg <= 0 ||
        !(dt > 0) || cap <= 0)
        return fail(PCT_ERR_INVALID, "bad bezier_check arguments");
    if (cap > kBezierCapMax) cap = kBezierCapMax;
    for (int i = 0; i < traj->nseg; i++)
        if (traj->orders[i] < 0 || traj->orders[i] > kMaxBezierOrder || 3 * (traj->orders[i] + 1) > traj->row_stride)
            return fail(PCT_ERR_INVALID, "segment %d: order %d unsupported", i, traj->orders[i]);
    if (c->ring_ready) {          // rolling map: the fused planner batch with samples only
        pct_replan_out o{};
        o.sample_pos = pos; o.sample_radius = radius; o.sample_d2 = d2; o.sample_idx = idx;
        PCTCHK(replan_direct(c, p, nullptr, 0, traj, t_start, stop_time, dt, (idx || d2) ? 1 : 0, 0, (int)cap, &o));
        *nsamples = o.nsamples;
        *first_hit = o.first_hit_sample;
        return PCT_OK;
    }
    const size_t ncoef = (size_t)traj->nseg * traj->row_stride;
    if (ncoef + (size_t)traj->nseg <= 3 * (size_t)kExpressMaxQ - 64) {
        // express: the host enumerates the sample times (sim_planning_demo.cpp:729-771, the same sequential fp64 additions as
        // bezier_samples_kernel), then ONE launch evaluates, inflates and searches every sample (bezier_block_kernel)
        double *hd = c->h_xin;                                  // [coef | seg_time | sample_t]
        uint32_t *hu = c->h_xids;                               // [orders | sample_seg]
        std::memcpy(hd, traj->polycoef, sizeof(double) * ncoef);
        std::memcpy(hd + ncoef, traj->seg_time, sizeof(double) * traj->nseg);
        for (int i = 0; i < traj->nseg; i++) hu[i] = (uint32_t)traj->orders[i];
        double *ht = hd + ncoef + traj->nseg;
        uint32_t *hs = hu + traj->nseg;
        const int64_t room = std::min<int64_t>({ cap, (int64_t)kExpressMaxQ, (int64_t)(3 * (size_t)kExpressMaxQ - ncoef - (size_t)traj->nseg) });
        double t_s = t_start;
        int first_seg;
        for (first_seg = 0; first_seg < traj->nseg; ++first_seg) {
            if (t_s > traj->seg_time[first_seg] && first_seg + 1 < traj->nseg) t_s -= traj->seg_time[first_seg];
            else break;
        }
        int64_t n = 0;
        double t_accu = 0.0;
        for (int i = first_seg; i < traj->nseg; i++) {
            const double T = traj->seg_time[i];
            for (double t = (i == first_seg) ? t_s : 0.0; t < T; t += dt) {
                t_accu += dt;
                if (t_accu > stop_time) break;
                if (n < room) { ht[n] = t; hs[n] = (uint32_t)i; }
                n++;
            }
        }
        const int64_t m = std::min<int64_t>(n, room);
        const bool fits = n <= room || room == cap;             // more samples than one express launch holds: staged path below
        if (fits && m > 0) {
            if (!c->h_bpos) PCTCHK(mapped_alloc(&c->h_bpos, &c->d_bpos, (size_t)3 * kExpressMaxQ));
            if (c->has_grid && c->count > 0) {        // indexed cloud: everything in ONE launch
                const double reach = p->max_radius + p->search_margin;
                const double stop_d2 = (idx || d2) ? (double)INFINITY : reach * reach;
                bezier_block_kernel<<<(int)m, 256, 0, g_stream>>>(c->G, c->sorted, c->cell_start, c->C, to_dev(p), c->d_xin, (int)traj->row_stride,
                                                                   c->d_xin + ncoef, c->d_xids, c->d_xids + traj->nseg, c->d_xin + ncoef + traj->nseg,
                                                                   stop_d2, (uint32_t)c->index_base, c->d_xout, c->d_bpos, next_signal(c));
                HIPCHK(hipGetLastError());
                PCTCHK(express_wait(c));
            } else {                                      // un-indexed (rolling) cloud: evaluate, then the brute-force inflation; still no copies
                PCTCHK(pct_cloud_reserve_queries(c, m));
                bezier_eval_kernel<<<ceil_div(m, 128), 128, 0, g_stream>>>(c->d_xin, (int)traj->row_stride, c->d_xin + ncoef, c->d_xids, c->d_xids + traj->nseg,
                                                                           c->d_xin + ncoef + traj->nseg, (int)m, c->d_pts64, c->d_bpos);
                PCTCHK(inflate_dev(c, p, m, g_stream, c->d_pts64, c->d_xout));
                HIPCHK(hipStreamSynchronize(g_stream));
            }
        }
        if (fits) {
            int64_t fh = -1;
            for (int64_t i = 0; i < m; i++) {
                if (fh < 0 && c->h_xout[i].radius < 0.0) fh = i;
                if (radius) radius[i] = c->h_xout[i].radius;
                if (d2) d2[i] = c->h_xout[i].d2;
                if (idx) idx[i] = c->h_xout[i].idx;
            }
            if (pos && m) std::memcpy(pos, c->h_bpos, sizeof(double) * 3 * m);
            *nsamples = n;
            *first_hit = fh;
            return PCT_OK;
        }
    }
    PCTCHK(pct_cloud_reserve_queries(c, cap));
    if (ncoef > c->coef_cap) { dev_free(c->d_coef); c->coef_cap = 0; PCTCHK(dev_alloc(&c->d_coef, ncoef)); c->coef_cap = ncoef; }
    if ((size_t)traj->nseg > c->seg_cap) {
        dev_free(c->d_segtime); dev_free(c->d_orders); c->seg_cap = 0;
        PCTCHK(dev_alloc(&c->d_segtime, traj->nseg));
        PCTCHK(dev_alloc(&c->d_orders, traj->nseg));
        c->seg_cap = traj->nseg;
    }
    if (!c->d_nsamples) PCTCHK(dev_alloc(&c->d_nsamples, 1));
    if (!c->d_first_hit) PCTCHK(dev_alloc(&c->d_first_hit, 1));
    hipStream_t s = g_stream;
    HIPCHK(hipMemcpyAsync(c->d_coef, traj->polycoef, sizeof(double) * ncoef, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_segtime, traj->seg_time, sizeof(double) * traj->nseg, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_orders, traj->orders, sizeof(int) * traj->nseg, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->d_pts64, 0, sizeof(double) * 3 * cap, s));
    BezierDesc B{ c->d_coef, c->d_segtime, c->d_orders, (int)traj->row_stride, traj->nseg, t_start, stop_time, dt, (int)cap };
    const size_t smem = (size_t)cap * (sizeof(double) + sizeof(int));
    bezier_samples_kernel<<<1, 256, smem, s>>>(B, c->d_pts64, c->d_nsamples);
    PCTCHK(inflate_dev(c, p, cap, s));
    first_hit_kernel<<<1, 256, 0, s>>>(c->d_radius, c->d_nsamples, (int)cap, c->d_first_hit);
    HIPCHK(hipGetLastError());
    int ns = 0;
    long long fh = -1;
    HIPCHK(hipMemcpyAsync(&ns, c->d_nsamples, sizeof ns, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&fh, c->d_first_hit, sizeof fh, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int64_t m = std::min<int64_t>(ns, cap);
    if (pos && m) HIPCHK(hipMemcpy(pos, c->d_pts64, sizeof(double) * 3 * m, hipMemcpyDeviceToHost));
    if (radius && m) HIPCHK(hipMemcpy(radius, c->d_radius, sizeof(double) * m, hipMemcpyDeviceToHost));
    if (d2 && m) HIPCHK(hipMemcpy(d2, c->d_d2, sizeof(double) * m, hipMemcpyDeviceToHost));
    if (idx && m) HIPCHK(hipMemcpy(idx, c->d_idx, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
    *nsamples = ns;
    *first_hit = fh;
    return PCT_OK;
}

// ---- stream variants of the planner arithmetic: device buffers, asynchronous on the caller's stream ------------------------------
int pct_inflate_batch_dev(pct_cloud *c, const pct_inflate_params *p, const double *d_pts, int64_t Q, double *d_radius, uint32_t *d_idx,
                          double *d_d2, void *stream)
{
    if (!c || !p || Q < 0 || (Q > 0 && (!d_pts || !d_radius))) return fail(PCT_ERR_INVALID, "bad inflate_batch_dev arguments");
    if (Q == 0) return PCT_OK;
    if (Q > c->qcap) return fail(PCT_ERR_INVALID, "batch of %lld exceeds reserved %lld (call pct_cloud_reserve_queries)", (long long)Q, (long long)c->qcap);
    PCTCHK(order_after_mutations(c, (hipStream_t)stream));
    return inflate_dev(c, p, Q, (hipStream_t)stream, d_pts, nullptr, d_radius, d_idx, d_d2);
}

int pct_bezier_check_dev(pct_cloud *c, const pct_bezier_traj *traj, const pct_inflate_params *p, double t_start, double stop_time, double dt,
                         int64_t cap, double *d_pos, double *d_radius, double *d_d2, uint32_t *d_idx, long long *d_first_hit, int32_t *d_nsamples,
                         void *stream)
{
    if (!c || !traj || !p || !traj->polycoef || !traj->seg_time || !traj->orders || traj->nseg <= 0 || !(dt > 0) || cap <= 0 || !d_radius ||
        !d_first_hit || !d_nsamples)
        return fail(PCT_ERR_INVALID, "bad bezier_check_dev arguments");
    if (cap > kBezierCapMax || cap > c->qcap) return fail(PCT_ERR_INVALID, "cap %lld exceeds %lld (4096, and the reserved batch size)", (long long)cap,
                                                          (long long)std::min<int64_t>(kBezierCapMax, c->qcap));
    for (int i = 0; i < traj->nseg; i++)
        if (traj->orders[i] < 0 || traj->orders[i] > kMaxBezierOrder || 3 * (traj->orders[i] + 1) > traj->row_stride)
            return fail(PCT_ERR_INVALID, "segment %d: order %d unsupported", i, traj->orders[i]);
    const size_t ncoef = (size_t)traj->nseg * traj->row_stride;
    hipStream_t s = (hipStream_t)stream;
    PCTCHK(order_after_mutations(c, s));
    if (ncoef > c->coef_cap || (size_t)traj->nseg > c->seg_cap) {       // grow the coefficient buffers (earlier work may still read the old ones)
        HIPCHK(hipDeviceSynchronize());
        if (ncoef > c->coef_cap) { dev_free(c->d_coef); c->coef_cap = 0; PCTCHK(dev_alloc(&c->d_coef, ncoef)); c->coef_cap = ncoef; }
        if ((size_t)traj->nseg > c->seg_cap) {
            dev_free(c->d_segtime); dev_free(c->d_orders); c->seg_cap = 0;
            PCTCHK(dev_alloc(&c->d_segtime, traj->nseg));
            PCTCHK(dev_alloc(&c->d_orders, traj->nseg));
            c->seg_cap = traj->nseg;
        }
    }
    HIPCHK(hipMemcpyAsync(c->d_coef, traj->polycoef, sizeof(double) * ncoef, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_segtime, traj->seg_time, sizeof(double) * traj->nseg, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_orders, traj->orders, sizeof(int) * traj->nseg, hipMemcpyHostToDevice, s));
    double *pos = d_pos ? d_pos : c->d_pts64;
    HIPCHK(hipMemsetAsync(pos, 0, sizeof(double) * 3 * cap, s));
    BezierDesc B{ c->d_coef, c->d_segtime, c->d_orders, (int)traj->row_stride, traj->nseg, t_start, stop_time, dt, (int)cap };
    const size_t smem = (size_t)cap * (sizeof(double) + sizeof(int));
    bezier_samples_kernel<<<1, 256, smem, s>>>(B, pos, d_nsamples);
    PCTCHK(inflate_dev(c, p, cap, s, pos, nullptr, d_radius, d_idx, d_d2));
    first_hit_kernel<<<1, 256, 0, s>>>(d_radius, d_nsamples, (int)cap, d_first_hit);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

// ---- hipGraph plan -----------------------------------------------------------------------
static int plan_capture_nn(pct_plan *p)
{
    pct_cloud *c = p->c;
    if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
    PCTCHK(pct_cloud_reserve_queries(c, p->Q));
    HIPCHK(hipStreamSynchronize(g_stream));
    hipError_t e = hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return fail(PCT_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    c->capturing = true;
    (void)hipMemcpyAsync(p->d_q, p->h_q, sizeof(float) * 3 * p->Q, hipMemcpyHostToDevice, g_stream);
    const int st = nn_dev(c, p->algo, p->d_q, p->Q, p->d_idx, p->d_d2, g_stream);
    (void)hipMemcpyAsync(p->h_idx, p->d_idx, sizeof(uint32_t) * p->Q, hipMemcpyDeviceToHost, g_stream);
    (void)hipMemcpyAsync(p->h_d2, p->d_d2, sizeof(double) * p->Q, hipMemcpyDeviceToHost, g_stream);
    c->capturing = false;
    e = hipStreamEndCapture(g_stream, &p->graph);
    if (st != PCT_OK) return st;
    if (e != hipSuccess || !p->graph) return fail(PCT_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) return fail(PCT_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    p->generation = c->generation;
    return PCT_OK;
}

int pct_plan_create_nn(pct_cloud *c, int algo, int64_t Q, pct_plan **out)
{
    if (!c || !out || Q <= 0) return fail(PCT_ERR_INVALID, "bad plan arguments");
    PCTCHK(pct_cloud_reserve_queries(c, Q));
    pct_plan *p = new (std::nothrow) pct_plan();
    if (!p) return fail(PCT_ERR_ALLOC, "host allocation failed");
    p->c = c;
    p->Q = Q;
    p->algo = algo;
    hipError_t e = hipHostMalloc((void **)&p->h_q, sizeof(float) * 3 * Q, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->h_idx, sizeof(uint32_t) * Q, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->h_d2, sizeof(double) * Q, hipHostMallocDefault);
    int st = PCT_OK;
    if (e != hipSuccess) st = fail(PCT_ERR_ALLOC, "hipHostMalloc: %s", hipGetErrorString(e));
    if (!st) st = dev_alloc(&p->d_q, 3 * Q);
    if (!st) st = dev_alloc(&p->d_idx, Q);
    if (!st) st = dev_alloc(&p->d_d2, Q);
    if (!st) st = plan_capture_nn(p);
    if (st) { pct_plan_destroy(p); return st; }
    *out = p;
    return PCT_OK;
}

int pct_plan_run(pct_plan *p, const float *q, uint32_t *idx, double *d2)
{
    if (!p || p->kind != 0 || !q || !idx || !d2) return fail(PCT_ERR_INVALID, "bad plan_run arguments");
    // The captured kernels hold the cloud's point count, its index description and its workspace pointers.  When any of them
    // has changed since the capture (upload / append, grid build or drop, a larger batch elsewhere that reallocated the
    // workspaces) the cloud's generation has moved on and the graph is captured again before it runs.
    if (p->generation != p->c->generation) PCTCHK(plan_capture_nn(p));
    memcpy(p->h_q, q, sizeof(float) * 3 * p->Q);
    HIPCHK(hipGraphLaunch(p->exec, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    memcpy(idx, p->h_idx, sizeof(uint32_t) * p->Q);
    memcpy(d2, p->h_d2, sizeof(double) * p->Q);
    return PCT_OK;
}

int pct_plan_destroy(pct_plan *p)
{
    if (!p) return PCT_OK;
    if (g_stream) (void)hipStreamSynchronize(g_stream);
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    if (p->h_q) (void)hipHostFree(p->h_q);
    if (p->h_idx) (void)hipHostFree(p->h_idx);
    if (p->h_d2) (void)hipHostFree(p->h_d2);
    dev_free(p->d_q); dev_free(p->d_idx); dev_free(p->d_d2);
    replan_ctx_free(p->rx);
    delete p;
    return PCT_OK;
}

// ---- measurement -------------------------------------------------------------------------
int pct_last_kernel_ms(pct_cloud *c, float *ms)
{
    if (!c || !ms) return fail(PCT_ERR_INVALID, "bad arguments");
    if (!c->last3) return fail(PCT_ERR_INVALID, "no timed batch yet");
    HIPCHK(hipEventSynchronize(c->last3));
    HIPCHK(hipEventElapsedTime(ms, c->last2, c->last3));
    return PCT_OK;
}

// multi-GPU exchange helper (pointcloudtraj_amd/dist.py): cand[i] = idx_local[i] if this rank holds the global minimum, else INT32_MAX
int pct_merge_mask_dev(const double *d_d2_local, const double *d_d2_best, const uint32_t *d_idx_local, int32_t *d_cand, int64_t Q, void *stream)
{
    if (Q < 0 || (Q > 0 && (!d_d2_local || !d_d2_best || !d_idx_local || !d_cand))) return fail(PCT_ERR_INVALID, "bad merge_mask arguments");
    if (Q == 0) return PCT_OK;
    PCTCHK(require_init());
    merge_mask_kernel<<<ceil_div(Q, 256), 256, 0, (hipStream_t)stream>>>(d_d2_local, d_d2_best, d_idx_local, d_cand, (uint32_t)Q);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

int pct_merge_finish_dev(const int32_t *d_cand, uint32_t *d_idx, int64_t Q, void *stream)
{
    if (Q < 0 || (Q > 0 && (!d_cand || !d_idx))) return fail(PCT_ERR_INVALID, "bad merge_finish arguments");
    if (Q == 0) return PCT_OK;
    PCTCHK(require_init());
    merge_finish_kernel<<<ceil_div(Q, 256), 256, 0, (hipStream_t)stream>>>(d_cand, d_idx, (uint32_t)Q);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

// ---- device helpers of the routed multi-GPU form (include/pct_shard.h; host logic in csrc/shard.cpp) ----------------------------
int pct_cloud_upload_aos_dev(pct_cloud *c, const void *d_pts, int64_t n, int64_t stride_bytes)
{
    if (!c || n < 0 || (n > 0 && !d_pts) || stride_bytes < 12 || (stride_bytes & 3)) return fail(PCT_ERR_INVALID, "bad upload arguments");
    if (n > c->cap) return fail(PCT_ERR_CAPACITY, "%lld points > capacity %lld", (long long)n, (long long)c->cap);
    if (c->host_mapped) return fail(PCT_ERR_INVALID, "small (host-mapped) clouds are filled from host buffers");
    drop_grid(c);
    if (n) {
        deinterleave_kernel<<<ceil_div(n, 256), 256, 0, g_stream>>>(static_cast<const unsigned char *>(d_pts), (uint32_t)stride_bytes, (uint32_t)n, c->x, c->y, c->z, 0u);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(g_stream));
    }
    c->count = n;
    c->ring_next = n % std::max<int64_t>(c->cap, 1);
    return after_replace(c);
}

static int route_cuts(const double *cuts, int world, int axis, RouteCuts *C)
{
    if (!cuts || world < 1 || world > kRouteMaxWorld || axis < 0 || axis > 2) return fail(PCT_ERR_INVALID, "bad slab description (at most %d ranks)", kRouteMaxWorld);
    C->world = world; C->axis = axis;
    for (int k = 0; k <= world; k++) C->cut[k] = cuts[k];
    return PCT_OK;
}

int pct_route_owner_dev(const double *cuts, int world, int axis, int rank, const float *d_q, int64_t Q, uint32_t *d_counts, uint32_t *d_mine_ids,
                        float *d_mine_q, void *stream)
{
    if (Q < 0 || rank < 0 || rank >= world || (Q > 0 && (!d_q || !d_counts || !d_mine_ids || !d_mine_q))) return fail(PCT_ERR_INVALID, "bad route_owner arguments");
    RouteCuts C{};
    PCTCHK(route_cuts(cuts, world, axis, &C));
    PCTCHK(require_init());
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * (size_t)world, s));
    if (Q) route_owner_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(C, rank, d_q, (uint32_t)Q, d_counts, d_mine_ids, d_mine_q);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

int pct_route_owner_all_dev(const double *cuts, int world, int axis, const float *d_q, int64_t Q, uint32_t *d_counts, unsigned char *d_owner, void *stream)
{
    if (Q < 0 || (Q > 0 && (!d_q || !d_counts || !d_owner))) return fail(PCT_ERR_INVALID, "bad route_owner_all arguments");
    RouteCuts C{};
    PCTCHK(route_cuts(cuts, world, axis, &C));
    PCTCHK(require_init());
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * (size_t)world, s));
    if (Q) route_owner_all_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(C, d_q, (uint32_t)Q, d_counts, d_owner);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

int pct_route_partition_dev(const uint32_t *offsets, int world, const unsigned char *d_owner, const float *d_q, int64_t Q, uint32_t *d_cursors,
                            float *d_out_xyz, uint32_t *d_out_slot, void *stream)
{
    if (!offsets || world < 1 || world > kRouteMaxWorld || Q < 0 || (Q > 0 && (!d_owner || !d_q || !d_cursors || !d_out_xyz || !d_out_slot)))
        return fail(PCT_ERR_INVALID, "bad route_partition arguments");
    RouteOffsets O{};
    for (int k = 0; k < world; k++) O.v[k] = offsets[k];
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(d_cursors, 0, sizeof(uint32_t) * (size_t)world, s));
    if (Q) route_partition_kernel<<<ceil_div(Q, 256), 256, 0, s>>>(O, d_owner, d_q, (uint32_t)Q, d_cursors, d_out_xyz, d_out_slot);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

int pct_route_certify_dev(int axis, double lo_edge, double hi_edge, const float *d_mine_q, const uint32_t *d_mine_ids, int64_t m, const uint32_t *d_lidx,
                          const double *d_ld2, const uint32_t *d_gid, void *d_answers, void *stream)
{
    if (m < 0 || axis < 0 || axis > 2 || (m > 0 && (!d_mine_q || !d_mine_ids || !d_lidx || !d_ld2 || !d_gid || !d_answers))) return fail(PCT_ERR_INVALID, "bad route_certify arguments");
    if (m == 0) return PCT_OK;
    route_certify_kernel<<<ceil_div(m, 256), 256, 0, (hipStream_t)stream>>>(axis, lo_edge, hi_edge, d_mine_q, d_mine_ids, (uint32_t)m, d_lidx, d_ld2, d_gid,
                                                                             static_cast<RouteAnswer *>(d_answers));
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

int pct_route_scatter_dev(const void *d_answers, int64_t n, uint32_t *d_idx, double *d_d2, uint32_t *d_flag_count, uint32_t *d_flag_ids, void *stream)
{
    if (n < 0 || (n > 0 && (!d_answers || !d_idx || !d_d2 || !d_flag_count || !d_flag_ids))) return fail(PCT_ERR_INVALID, "bad route_scatter arguments");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(d_flag_count, 0, sizeof(uint32_t), s));
    if (n) route_scatter_kernel<<<ceil_div(n, 256), 256, 0, s>>>(static_cast<const RouteAnswer *>(d_answers), (uint32_t)n, d_idx, d_d2, d_flag_count, d_flag_ids);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

int pct_route_gather_queries_dev(const float *d_q, const uint32_t *d_ids, int64_t n, float *d_out, void *stream)
{
    if (n < 0 || (n > 0 && (!d_q || !d_ids || !d_out))) return fail(PCT_ERR_INVALID, "bad route_gather arguments");
    if (n) route_gather_queries_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(d_q, d_ids, (uint32_t)n, d_out);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

int pct_route_to_global_dev(uint32_t *d_lidx, int64_t n, const uint32_t *d_gid, void *stream)
{
    if (n < 0 || (n > 0 && (!d_lidx || !d_gid))) return fail(PCT_ERR_INVALID, "bad route_to_global arguments");
    if (n) route_to_global_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(d_lidx, (uint32_t)n, d_gid);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

int pct_route_put_back_dev(const uint32_t *d_ids, int64_t n, const uint32_t *d_idx, const double *d_d2, uint32_t *d_out_idx, double *d_out_d2, void *stream)
{
    if (n < 0 || (n > 0 && (!d_ids || !d_idx || !d_d2 || !d_out_idx || !d_out_d2))) return fail(PCT_ERR_INVALID, "bad route_put_back arguments");
    if (n) route_put_back_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(d_ids, (uint32_t)n, d_idx, d_d2, d_out_idx, d_out_d2);
    HIPCHK(hipGetLastError());
    return PCT_OK;
}

int pct_set_timing(pct_cloud *c, int level)
{
    if (!c || level < 0 || level > 2) return fail(PCT_ERR_INVALID, "timing level must be 0, 1 or 2");
    c->timing_level = level;
    return PCT_OK;
}

int pct_set_timing_stride(pct_cloud *c, int stride)
{
    if (!c || stride < 1) return fail(PCT_ERR_INVALID, "timing stride must be >= 1");
    c->dom_stride = stride;
    c->dom_launch = 0;
    return PCT_OK;
}

int pct_kernel_ms_samples(pct_cloud *c, uint64_t *count)
{
    if (!c || !count) return fail(PCT_ERR_INVALID, "bad arguments");
    *count = c->dom_seq;
    return PCT_OK;
}

int pct_kernel_ms_history(pct_cloud *c, float *ms, int cap, int *n)
{
    if (!c || !ms || !n || cap <= 0) return fail(PCT_ERR_INVALID, "bad arguments");
    const uint64_t have = std::min<uint64_t>(c->dom_seq, pct_cloud::kDomRing);
    const int take = (int)std::min<uint64_t>(have, (uint64_t)cap);
    for (int i = 0; i < take; i++) {                       // oldest of the `take` most recent batches first
        const int slot = (int)((c->dom_seq - take + i) % pct_cloud::kDomRing);
        HIPCHK(hipEventSynchronize(c->dom_ring[2 * slot + 1]));
        HIPCHK(hipEventElapsedTime(&ms[i], c->dom_ring[2 * slot], c->dom_ring[2 * slot + 1]));
    }
    *n = take;
    return PCT_OK;
}

int pct_last_batch_ms(pct_cloud *c, float *ms)
{
    if (!c || !ms) return fail(PCT_ERR_INVALID, "bad arguments");
    if (!c->ev_valid) return fail(PCT_ERR_INVALID, "no timed batch yet");
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return PCT_OK;
}

// test hook: the cell index as built (cell_start: ncells + 1 entries, records: 4 floats per point = x, y, z, bit-cast index)
int pct_debug_read_grid(pct_cloud *c, uint32_t *cell_start, float *records)
{
    if (!c || !c->has_grid) return fail(PCT_ERR_INVALID, "no cell index");
    HIPCHK(hipStreamSynchronize(g_stream));
    if (cell_start) HIPCHK(hipMemcpy(cell_start, c->cell_start, sizeof(uint32_t) * ((size_t)c->G.ncells + 1), hipMemcpyDeviceToHost));
    if (records) HIPCHK(hipMemcpy(records, c->sorted, sizeof(float4) * (size_t)c->count, hipMemcpyDeviceToHost));
    return PCT_OK;
}

int pct_debug_read_bounds(pct_cloud *c, float *out, int64_t Q)
{
    if (!c || !out || Q > c->qcap) return fail(PCT_ERR_INVALID, "bad arguments");
    HIPCHK(hipStreamSynchronize(g_stream));
    HIPCHK(hipMemcpy(out, c->d_bound, sizeof(float) * Q, hipMemcpyDeviceToHost));
    return PCT_OK;
}

int pct_debug_set_filter_mode(int mode)
{
    if (mode < -1 || mode > 1) return fail(PCT_ERR_INVALID, "filter mode must be -1 (automatic), 0 (direct form) or 1 (expanded form)");
    g_filter_mode = mode;
    return PCT_OK;
}

int pct_set_work_counters(pct_cloud *c, int enabled)
{
    if (!c) return fail(PCT_ERR_INVALID, "null cloud");
    c->count_work = enabled != 0;
    return PCT_OK;
}

int pct_last_work(pct_cloud *c, uint64_t *points_scanned, uint64_t *cells_scanned)
{
    if (!c) return fail(PCT_ERR_INVALID, "null cloud");
    WorkCounters w{};
    if (c->host_work) {
        w.points = c->host_points;
    } else {
        WorkCounters slots[kWorkSlots];
        HIPCHK(hipStreamSynchronize(g_stream));
        HIPCHK(hipMemcpy(slots, c->d_work, sizeof slots, hipMemcpyDeviceToHost));
        for (const WorkCounters &k : slots) { w.points += k.points; w.cells += k.cells; }
    }
    if (points_scanned) *points_scanned = w.points;
    if (cells_scanned) *cells_scanned = w.cells;
    return PCT_OK;
}

int pct_last_work_ex(pct_cloud *c, uint64_t out[3])
{
    if (!c || !out) return fail(PCT_ERR_INVALID, "bad arguments");
    out[0] = out[1] = out[2] = 0;
    if (c->host_work) { out[0] = c->host_points; return PCT_OK; }
    WorkCounters slots[kWorkSlots];
    HIPCHK(hipStreamSynchronize(g_stream));
    HIPCHK(hipMemcpy(slots, c->d_work, sizeof slots, hipMemcpyDeviceToHost));
    for (const WorkCounters &k : slots) { out[0] += k.points; out[1] += k.cells; out[2] += k.nodes; }
    return PCT_OK;
}

int pct_cloud_pyramid_info(const pct_cloud *c, int32_t *levels, int64_t *nodes, double *empty_fraction)
{
    if (!c || !c->has_grid) return fail(PCT_ERR_INVALID, "no cell index");
    if (levels) *levels = c->has_pyr ? c->P.nlev : 0;
    if (nodes) *nodes = c->has_pyr ? (int64_t)c->pyr_total : 0;
    if (empty_fraction) *empty_fraction = c->empty_frac;
    return PCT_OK;
}

int pct_debug_verify_grid(pct_cloud *c, uint64_t out[6])
{
    if (!c || !c->has_grid || !out) return fail(PCT_ERR_INVALID, "no cell index");
    const uint32_t n = (uint32_t)c->count;
    uint32_t *bitmap = nullptr;
    unsigned long long *d_out = nullptr;
    const size_t words = ((size_t)n + 31) / 32;
    int st = dev_alloc(&bitmap, words);
    if (!st) st = dev_alloc(&d_out, 6);
    hipError_t e = hipSuccess;
    if (!st) {
        e = hipMemsetAsync(bitmap, 0, sizeof(uint32_t) * std::max<size_t>(words, 1), g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long) * 6, g_stream);
        if (e == hipSuccess) {
            grid_verify_kernel<<<(int)std::min<int64_t>(2048, (std::max<int64_t>(n, (int64_t)c->G.ncells) + 255) / 256), 256, 0, g_stream>>>(c->G, c->sorted, c->cell_start, n, bitmap, d_out);
            e = hipGetLastError();
        }
        unsigned long long h[6] = { 0, 0, 0, 0, 0, 0 };
        if (e == hipSuccess) e = hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        for (int k = 0; k < 6; k++) out[k] = h[k];
    }
    dev_free(bitmap); dev_free(d_out);
    if (st) return st;
    if (e != hipSuccess) return fail(PCT_ERR_HIP, "verify_grid: %s", hipGetErrorString(e));
    return PCT_OK;
}

}  // extern "C"
