/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE.  Batch drivers around the PUBLIC kd_* API of
 * the reference library (oracle/_ref/libkdtree_ref.so, compiled unmodified from
 * /root/reference/Utils/kdtree/src/kdtree.c).  Contains no reference code: it only calls
 * the exported functions the way a client would, in C loops, so that fixtures and the
 * "reference" CPU baseline do not pay Python call overhead per query.
 *
 * Payload convention: data = (void*)(insertion index + 1), so NULL never aliases id 0.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>
#include "kdtree/kdtree.h"

int64_t refshim_insertf_batch(struct kdtree *t, const float *xyz, int64_t n, int64_t first_id)
{
    for (int64_t i = 0; i < n; i++)
        if (kd_insertf(t, xyz + 3 * i, (void *)(intptr_t)(first_id + i + 1))) return i;
    return n;
}

/* kd_nearestf + kd_res_item (position) per query; d2 recomputed from the returned fp64
 * position with the reference's own accumulation order. */
int refshim_nearestf_batch(struct kdtree *t, const float *q, int64_t nq, int32_t *idx, double *d2)
{
    for (int64_t i = 0; i < nq; i++) {
        struct kdres *r = kd_nearestf(t, q + 3 * i);
        if (!r) return -1;
        double p[3];
        void *d = kd_res_item(r, p);
        idx[i] = (int32_t)((intptr_t)d - 1);
        double s = 0;
        for (int k = 0; k < 3; k++) { double df = p[k] - (double)q[3 * i + k]; s += df * df; }
        d2[i] = s;
        kd_res_free(r);
    }
    return 0;
}

/* timing variant: the loop the reference's callers run (corridor_finder.cpp:428-437):
 * kd_nearestf -> kd_res_item_data -> kd_res_free.  Returns seconds, writes ids. */
double refshim_nearestf_timed(struct kdtree *t, const float *q, int64_t nq, int32_t *idx)
{
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int64_t i = 0; i < nq; i++) {
        struct kdres *r = kd_nearestf(t, q + 3 * i);
        idx[i] = (int32_t)((intptr_t)kd_res_item_data(r) - 1);
        kd_res_free(r);
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

/* kd_nearest_rangef for one query: ids in ITERATION order into out (capacity cap);
 * returns kd_res_size. */
int64_t refshim_rangef(struct kdtree *t, const float *q, float range, int32_t *out, int64_t cap)
{
    struct kdres *r = kd_nearest_rangef(t, q, range);
    if (!r) return -1;
    int64_t n = kd_res_size(r), k = 0;
    while (!kd_res_end(r)) {
        if (k < cap) out[k] = (int32_t)((intptr_t)kd_res_item_data(r) - 1);
        k++;
        kd_res_next(r);
    }
    kd_res_free(r);
    return (k == n) ? n : -2;
}

int refshim_range_countf_batch(struct kdtree *t, const float *q, const float *range, int64_t nq, int32_t *count)
{
    for (int64_t i = 0; i < nq; i++) {
        struct kdres *r = kd_nearest_rangef(t, q + 3 * i, range[i]);
        if (!r) return -1;
        count[i] = kd_res_size(r);
        kd_res_free(r);
    }
    return 0;
}

/* T host threads sharing one read-only tree, queries partitioned in contiguous slices ("T threads with
 * queries partitioned across threads on a shared read-only tree").  The *f query variants stage through a static buffer
 * (kdtree.c:345-361) and are not re-entrant, so each thread widens its query itself and calls kd_nearest (doubles), which
 * keeps no shared state.  Returns wall seconds for the whole batch. */
#include <pthread.h>
struct mt_job { struct kdtree *t; const float *q; int64_t begin, end; int32_t *idx; };

static void *mt_worker(void *arg)
{
    struct mt_job *j = (struct mt_job *)arg;
    for (int64_t i = j->begin; i < j->end; i++) {
        const double p[3] = { j->q[3 * i], j->q[3 * i + 1], j->q[3 * i + 2] };
        struct kdres *r = kd_nearest(j->t, p);
        j->idx[i] = (int32_t)((intptr_t)kd_res_item_data(r) - 1);
        kd_res_free(r);
    }
    return NULL;
}

double refshim_nearest_timed_mt(struct kdtree *t, const float *q, int64_t nq, int32_t *idx, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    struct mt_job job[256];
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int k = 0; k < threads; k++) {
        job[k].t = t; job[k].q = q; job[k].idx = idx;
        job[k].begin = nq * k / threads; job[k].end = nq * (k + 1) / threads;
        if (pthread_create(&th[k], NULL, mt_worker, &job[k])) return -1.0;
    }
    for (int k = 0; k < threads; k++) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
