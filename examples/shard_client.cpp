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

int main(int argc, char **argv)
{
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
    std::printf("rank %d/%d: shard [%lld, %lld) of %lld points, %lld queries merged in %.3f ms per batch (host buffers), %d mismatches\n", rank, world,
                (long long)b, (long long)e, (long long)N, (long long)Q, ms, bad);
    pct_cloud_destroy(cloud);
    pct_shard_destroy(sh);
    return bad ? 1 : 0;
}
