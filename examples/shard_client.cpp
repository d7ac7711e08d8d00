// shard_client.cpp -- one rank of a multi-GPU planner process against libpct_shard.so (include/pct_shard.h), in C++:
//
//     shard_client <rank> <world> <rendezvous file> [total points = 8000000] [queries = 4096]
//
// Config C4 in miniature: ONE cloud of `total points` uniform points in [0,200)^3 (counter-based generator, so every rank
// produces exactly its own index range), rank r holds [r*N/W, (r+1)*N/W) on GPU r, the query batch is replicated, and
// pct_shard_nn answers it with per-shard kernels + ncclAllReduce(min) x 2.  Rank 0 writes the RCCL rendezvous token into the
// file (any channel would do: a ROS parameter, MPI, a socket), the other ranks wait for it.  Every rank then checks a sample of
// the merged answers against a plain host loop over the WHOLE cloud in the same fp64 arithmetic.  Exit code 0 = all matched.
// examples/run_shard_client.sh starts W ranks on W GPUs.  With W = 1 the collectives run on a one-rank communicator.
// The same batch then goes through the ROUTED form (pct_shard_route_build / pct_shard_route_nn_dev: slab ownership, owned answers
// exchanged as records) and must give the identical answers.
//
//     shard_client local <world> [total points] [queries] [halo spacings = 4]
//
// runs <world> ranks of the routed form inside this one process on one card (pct_shard_local_world: RCCL refuses two ranks on one
// device) against a single cloud holding everything: identical answers required, plus the sizes of the slabs and how many answers
// needed the second round.  A halo of 0 spacings forces that round for a good share of the batch.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pct_shard.h"

static uint64_t mix(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float coord(uint64_t seed, uint64_t k) { return 200.0f * ((float)(mix(seed, k) >> 40) * 0x1p-24f); }

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        const int st_ = (call);                                                                       \
        if (st_ != PCT_OK) { std::fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, #call, st_, pct_last_error()); return 2; } \
    } while (0)

#include <hip/hip_runtime_api.h>

static int run_local(int world, int64_t N, int64_t Q, double halo)
{
    const int rank = -1;
    CHECK(pct_init(0));
    std::vector<float> all((size_t)3 * N), q((size_t)3 * Q);
    for (size_t k = 0; k < all.size(); k++) all[k] = coord(6, k);
    for (size_t k = 0; k < q.size(); k++) q[k] = coord(7, k);
    // a few duplicated rows (exact ties across slabs must resolve to the lowest global index) and queries on top of points
    for (int64_t k = 0; k < std::min<int64_t>(N / 2, 2000); k++) for (int d = 0; d < 3; d++) all[(size_t)3 * (N - 1 - k) + d] = all[(size_t)3 * (7 * k % (N / 2)) + d];
    for (int64_t k = 0; k < std::min<int64_t>(Q / 4, 500); k++) for (int d = 0; d < 3; d++) q[(size_t)3 * k + d] = all[(size_t)3 * ((13 * k) % N) + d];
    // the reference answer: one cloud holding everything
    pct_cloud *whole = nullptr;
    CHECK(pct_cloud_create(N, &whole));
    CHECK(pct_cloud_upload_aos(whole, all.data(), N, 12));
    CHECK(pct_cloud_build_grid(whole, 0.0f));
    std::vector<uint32_t> want_i((size_t)Q), got_i((size_t)Q);
    std::vector<double> want_d((size_t)Q), got_d((size_t)Q);
    CHECK(pct_nn_batch_algo(whole, PCT_ALGO_GRID, q.data(), Q, want_i.data(), want_d.data()));
    // W ranks in this process: contiguous index ranges in, slabs out
    std::vector<pct_shard *> ranks((size_t)world);
    CHECK(pct_shard_local_world(world, ranks.data()));
    std::vector<const void *> lp((size_t)world);
    std::vector<int64_t> ln((size_t)world), lb((size_t)world);
    for (int r = 0; r < world; r++) {
        int64_t b = 0, e = 0;
        CHECK(pct_shard_range(ranks[(size_t)r], N, &b, &e));
        lp[(size_t)r] = all.data() + 3 * b; ln[(size_t)r] = e - b; lb[(size_t)r] = b;
    }
    std::vector<pct_route *> routes((size_t)world);
    CHECK(pct_shard_route_build_world(ranks.data(), world, lp.data(), ln.data(), 12, lb.data(), halo, routes.data()));
    float *d_q = nullptr; uint32_t *d_i = nullptr; double *d_d = nullptr;
    if (hipMalloc((void **)&d_q, sizeof(float) * 3 * Q) != hipSuccess || hipMalloc((void **)&d_i, sizeof(uint32_t) * Q * world) != hipSuccess ||
        hipMalloc((void **)&d_d, sizeof(double) * Q * world) != hipSuccess) return 2;
    if (hipMemcpy(d_q, q.data(), sizeof(float) * 3 * Q, hipMemcpyHostToDevice) != hipSuccess) return 2;
    std::vector<uint32_t *> oi((size_t)world);
    std::vector<double *> od((size_t)world);
    for (int r = 0; r < world; r++) { oi[(size_t)r] = d_i + (size_t)r * Q; od[(size_t)r] = d_d + (size_t)r * Q; }
    int bad = 0;
    for (int rep = 0; rep < 2; rep++) {            // twice: the workspaces and counters of a second batch
        CHECK(pct_shard_route_nn_world(routes.data(), world, d_q, Q, oi.data(), od.data(), nullptr));
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        for (int r = 0; r < world; r++) {
            if (hipMemcpy(got_i.data(), oi[(size_t)r], sizeof(uint32_t) * Q, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(got_d.data(), od[(size_t)r], sizeof(double) * Q, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            for (int64_t k = 0; k < Q; k++)
                if (got_i[(size_t)k] != want_i[(size_t)k] || got_d[(size_t)k] != want_d[(size_t)k]) {
                    if (bad++ < 8) std::fprintf(stderr, "rank %d query %lld: routed (%u, %.17g) single cloud (%u, %.17g)\n", r, (long long)k, got_i[(size_t)k], got_d[(size_t)k], want_i[(size_t)k], want_d[(size_t)k]);
                }
        }
    }
    // PARTITIONED batches: every rank brings its own queries (different seeds, different sizes) and gets its own answers
    {
        std::vector<std::vector<float>> pq((size_t)world);
        std::vector<const float *> dq((size_t)world);
        std::vector<int64_t> nq((size_t)world);
        std::vector<float *> to_free;
        for (int r = 0; r < world; r++) {
            nq[(size_t)r] = Q / 2 + 1000 * r;
            pq[(size_t)r].resize((size_t)3 * nq[(size_t)r]);
            for (size_t k = 0; k < pq[(size_t)r].size(); k++) pq[(size_t)r][k] = coord(100 + (uint64_t)r, k);
            for (int64_t k = 0; k < std::min<int64_t>(nq[(size_t)r] / 8, 300); k++) for (int d = 0; d < 3; d++) pq[(size_t)r][(size_t)3 * k + d] = all[(size_t)3 * ((17 * k + r) % N) + d];
            float *p = nullptr;
            if (hipMalloc((void **)&p, sizeof(float) * 3 * nq[(size_t)r]) != hipSuccess || hipMemcpy(p, pq[(size_t)r].data(), sizeof(float) * 3 * nq[(size_t)r], hipMemcpyHostToDevice) != hipSuccess) return 2;
            dq[(size_t)r] = p; to_free.push_back(p);
        }
        for (int rep = 0; rep < 2; rep++) {
            CHECK(pct_shard_route_nn_partitioned_world(routes.data(), world, dq.data(), nq.data(), oi.data(), od.data(), nullptr));
            if (hipDeviceSynchronize() != hipSuccess) return 2;
            for (int r = 0; r < world; r++) {
                const int64_t n = nq[(size_t)r];
                std::vector<uint32_t> wi((size_t)n), gi((size_t)n);
                std::vector<double> wd((size_t)n), gd((size_t)n);
                CHECK(pct_nn_batch_algo(whole, PCT_ALGO_GRID, pq[(size_t)r].data(), n, wi.data(), wd.data()));
                if (hipMemcpy(gi.data(), oi[(size_t)r], sizeof(uint32_t) * n, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(gd.data(), od[(size_t)r], sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) return 2;
                for (int64_t k = 0; k < n; k++)
                    if (gi[(size_t)k] != wi[(size_t)k] || gd[(size_t)k] != wd[(size_t)k]) {
                        if (bad++ < 8) std::fprintf(stderr, "partitioned: rank %d query %lld: routed (%u, %.17g) single cloud (%u, %.17g)\n", r, (long long)k, gi[(size_t)k], gd[(size_t)k], wi[(size_t)k], wd[(size_t)k]);
                    }
            }
        }
        for (float *p : to_free) (void)hipFree(p);
        std::printf("partitioned batches (every rank its own %lld+ queries): answers equal the single cloud's so far: %s\n", (long long)(Q / 2), bad ? "NO" : "yes");
    }
    uint64_t owned_sum = 0, uncert = 0;
    for (int r = 0; r < world; r++) {
        int64_t sp = 0; uint64_t ow = 0, un = 0, ba = 0;
        CHECK(pct_shard_route_stats(routes[(size_t)r], &sp, &ow, &un, &ba));
        std::printf("local rank %d/%d: slab of %lld points, owned %llu of %lld queries per batch, %llu uncertified per batch\n", r, world, (long long)sp,
                    (unsigned long long)(ow / ba), (long long)Q, (unsigned long long)(un / ba));
        owned_sum += ow / ba; uncert = un / ba;
    }
    (void)owned_sum;
    std::printf("routed form, %d ranks in one process, halo %.1f spacings: %lld queries, %llu in the second round, %d mismatches\n", world, halo, (long long)Q,
                (unsigned long long)uncert, bad);
    for (int r = 0; r < world; r++) { pct_shard_route_destroy(routes[(size_t)r]); pct_shard_destroy(ranks[(size_t)r]); }
    (void)hipFree(d_q); (void)hipFree(d_i); (void)hipFree(d_d);
    pct_cloud_destroy(whole);
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && std::strcmp(argv[1], "local") == 0)
        return run_local(std::atoi(argv[2]), argc > 3 ? std::atoll(argv[3]) : 2000000, argc > 4 ? std::atoll(argv[4]) : 65536, argc > 5 ? std::atof(argv[5]) : 4.0);
    if (argc < 4) { std::fprintf(stderr, "usage: %s rank world rendezvous_file [points] [queries]\n", argv[0]); return 64; }
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
    const std::string path = argv[3];
    const int64_t N = argc > 4 ? std::atoll(argv[4]) : 8000000, Q = argc > 5 ? std::atoll(argv[5]) : 4096;

    unsigned char id[PCT_SHARD_ID_BYTES];
    if (rank == 0) {
        CHECK(pct_shard_unique_id(id));
        const std::string tmp = path + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) { std::perror("rendezvous file"); return 2; }
        std::fclose(f);
        std::rename(tmp.c_str(), path.c_str());
    } else {
        FILE *f = nullptr;
        for (int tries = 0; tries < 6000 && !(f = std::fopen(path.c_str(), "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (!f || std::fread(id, 1, sizeof id, f) != sizeof id) { std::fprintf(stderr, "rank %d: no rendezvous token\n", rank); return 2; }
        std::fclose(f);
    }
    const int ndev = pct_device_count();
    if (ndev <= 0) { std::fprintf(stderr, "rank %d: no GPU\n", rank); return 2; }
    pct_shard *sh = nullptr;
    CHECK(pct_shard_init(id, rank, world, rank % ndev, &sh));

    int64_t b = 0, e = 0;
    CHECK(pct_shard_range(sh, N, &b, &e));
    std::vector<float> local((size_t)3 * (e - b));
    for (int64_t i = b; i < e; i++)
        for (int d = 0; d < 3; d++) local[(size_t)3 * (i - b) + d] = coord(6, (uint64_t)3 * i + d);
    pct_cloud *cloud = nullptr;
    CHECK(pct_shard_cloud_create(sh, N, &cloud));
    CHECK(pct_cloud_upload_aos(cloud, local.data(), e - b, 12));
    if (e > b) CHECK(pct_cloud_build_grid(cloud, 0.0f));

    std::vector<float> q((size_t)3 * Q);
    for (size_t k = 0; k < q.size(); k++) q[k] = coord(7, k);
    std::vector<uint32_t> idx((size_t)Q);
    std::vector<double> d2((size_t)Q);
    CHECK(pct_shard_nn(sh, cloud, PCT_ALGO_GRID, q.data(), Q, idx.data(), d2.data()));       // warm-up (first collective sets up the rings)
    const auto t0 = std::chrono::steady_clock::now();
    const int reps = 20;
    for (int r = 0; r < reps; r++) CHECK(pct_shard_nn(sh, cloud, PCT_ALGO_GRID, q.data(), Q, idx.data(), d2.data()));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::vector<uint32_t> idx_b((size_t)Q);
    std::vector<double> d2_b((size_t)Q);
    CHECK(pct_shard_nn(sh, cloud, PCT_ALGO_STREAM, q.data(), Q, idx_b.data(), d2_b.data()));  // brute-force kernels per shard, same exchange

    // check: a sample of the queries against the whole cloud on the host (every rank can regenerate it)
    int bad = 0;
    const int64_t nchk = std::min<int64_t>(Q, 24);
    for (int64_t k = 0; k < nchk; k++) {
        const double qx = q[3 * k], qy = q[3 * k + 1], qz = q[3 * k + 2];
        double best = INFINITY;
        uint32_t bi = PCT_NO_INDEX;
        for (int64_t i = 0; i < N; i++) {
            const double dx = (double)coord(6, (uint64_t)3 * i) - qx, dy = (double)coord(6, (uint64_t)3 * i + 1) - qy, dz = (double)coord(6, (uint64_t)3 * i + 2) - qz;
            double s = dx * dx; s = s + dy * dy; s = s + dz * dz;
            if (s < best) { best = s; bi = (uint32_t)i; }
        }
        if (best != d2[(size_t)k] || bi != idx[(size_t)k] || best != d2_b[(size_t)k] || bi != idx_b[(size_t)k]) {
            bad++;
            std::fprintf(stderr, "rank %d query %lld: host (%u, %.17g) grid (%u, %.17g) brute (%u, %.17g)\n", rank, (long long)k, bi, best, idx[(size_t)k],
                         d2[(size_t)k], idx_b[(size_t)k], d2_b[(size_t)k]);
        }
    }
    for (int64_t k = 0; k < Q; k++) if (idx[(size_t)k] != idx_b[(size_t)k] || d2[(size_t)k] != d2_b[(size_t)k]) bad++;
    // the routed form over the same communicator: slabs instead of index ranges, the owned answers exchanged as records
    {
        pct_route *route = nullptr;
        CHECK(pct_shard_route_build(sh, local.data(), e - b, 12, b, 4.0, &route));
        float *d_q = nullptr; uint32_t *d_i = nullptr; double *d_d = nullptr;
        if (hipMalloc((void **)&d_q, sizeof(float) * 3 * Q) != hipSuccess || hipMalloc((void **)&d_i, sizeof(uint32_t) * Q) != hipSuccess || hipMalloc((void **)&d_d, sizeof(double) * Q) != hipSuccess) return 2;
        if (hipMemcpy(d_q, q.data(), sizeof(float) * 3 * Q, hipMemcpyHostToDevice) != hipSuccess) return 2;
        CHECK(pct_shard_route_nn_dev(route, d_q, Q, d_i, d_d, nullptr));
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        const auto r0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) CHECK(pct_shard_route_nn_dev(route, d_q, Q, d_i, d_d, nullptr));
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        const double rms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count() / reps;
        std::vector<uint32_t> ri((size_t)Q);
        std::vector<double> rd((size_t)Q);
        if (hipMemcpy(ri.data(), d_i, sizeof(uint32_t) * Q, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(rd.data(), d_d, sizeof(double) * Q, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        int rbad = 0;
        for (int64_t k = 0; k < Q; k++) if (ri[(size_t)k] != idx[(size_t)k] || rd[(size_t)k] != d2[(size_t)k]) rbad++;
        int64_t sp = 0; uint64_t ow = 0, un = 0, ba = 0;
        CHECK(pct_shard_route_stats(route, &sp, &ow, &un, &ba));
        std::printf("rank %d/%d: routed form: slab of %lld points, %llu of %lld queries owned, %llu uncertified per batch, %.3f ms per batch (device buffers), %d mismatches vs index-range shards\n",
                    rank, world, (long long)sp, (unsigned long long)(ow / ba), (long long)Q, (unsigned long long)(un / ba), rms, rbad);
        bad += rbad;
        // partitioned batch: this rank's own queries (seed by rank), answers checked against the replicated form's for the same queries
        {
            std::vector<float> mq((size_t)3 * Q);
            for (size_t k = 0; k < mq.size(); k++) mq[k] = coord(200 + (uint64_t)rank, k);
            if (hipMemcpy(d_q, mq.data(), sizeof(float) * 3 * Q, hipMemcpyHostToDevice) != hipSuccess) return 2;
            CHECK(pct_shard_route_nn_partitioned_dev(route, d_q, Q, d_i, d_d, nullptr));
            if (hipDeviceSynchronize() != hipSuccess) return 2;
            std::vector<uint32_t> pi((size_t)Q), wi((size_t)Q);
            std::vector<double> pd((size_t)Q), wd((size_t)Q);
            if (hipMemcpy(pi.data(), d_i, sizeof(uint32_t) * Q, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(pd.data(), d_d, sizeof(double) * Q, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            CHECK(pct_shard_nn(sh, cloud, PCT_ALGO_GRID, mq.data(), Q, wi.data(), wd.data()));       // index-range shards: every rank must pass the same batch...
            int pbad = 0;
            if (world == 1) for (int64_t k = 0; k < Q; k++) if (pi[(size_t)k] != wi[(size_t)k] || pd[(size_t)k] != wd[(size_t)k]) pbad++;     // ... so only comparable with one rank
            std::printf("rank %d/%d: partitioned batch of %lld own queries: %d mismatches%s\n", rank, world, (long long)Q, pbad, world == 1 ? "" : " (not compared: ranks hold different batches)");
            bad += pbad;
        }
        pct_shard_route_destroy(route);
        (void)hipFree(d_q); (void)hipFree(d_i); (void)hipFree(d_d);
    }
    std::printf("rank %d/%d: shard [%lld, %lld) of %lld points, %lld queries merged in %.3f ms per batch (host buffers), %d mismatches\n", rank, world,
                (long long)b, (long long)e, (long long)N, (long long)Q, ms, bad);
    pct_cloud_destroy(cloud);
    pct_shard_destroy(sh);
    return bad ? 1 : 0;
}
