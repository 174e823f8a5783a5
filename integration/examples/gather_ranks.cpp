// Host C++ over the C ABI for BASELINE configs[3]: pairs sharded over the GPUs of one node, one RCCL all-gather of the
// pose records (north_star: "host C++ calling hand-written HIP kernels ... with a trivial RCCL gather of poses").
// One process per GPU, no MPI: rank 0 creates the ncclUniqueId and publishes it through a file.
//
//   hipcc -O2 -o gather_ranks integration/examples/gather_ranks.cpp -Iinclude -Lmvslam_amd/lib -lmvslam_hip -lrccl \
//         -Wl,-rpath,$PWD/mvslam_amd/lib
//   for r in 0 1; do ./gather_ranks $r 2 /tmp/mvs_nccl_id & done; wait        # two GPUs
//   ./gather_ranks 0 1 /tmp/mvs_nccl_id                                       # one GPU: the same calls, world size 1
//
// Every rank runs `pairs` synthetic pairs (global indices rank * pairs ...), gathers all records and checks that its own
// block of the gathered array is its own result, and that every block carries the global indices of its owner.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "mvslam_hip.h"

#define CHECK(x)                                                                  \
    do {                                                                          \
        if (!(x)) {                                                               \
            std::fprintf(stderr, "rank %d: %s failed (line %d)\n", rank, #x, __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char **argv)
{
    const int rank = argc > 1 ? atoi(argv[1]) : 0, world = argc > 2 ? atoi(argv[2]) : 1;
    const char *id_file = argc > 3 ? argv[3] : "/tmp/mvs_nccl_id";
    const int pairs = argc > 4 ? atoi(argv[4]) : 16, n_kp = 400, H = 512;
    int n_dev = 0;
    CHECK(hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0);
    const int dev = rank % n_dev;
    CHECK(hipSetDevice(dev) == hipSuccess);

    ncclUniqueId id;
    if (rank == 0) {
        CHECK(ncclGetUniqueId(&id) == ncclSuccess);
        std::string tmp = std::string(id_file) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        CHECK(f && std::fwrite(&id, sizeof(id), 1, f) == 1);
        std::fclose(f);
        CHECK(std::rename(tmp.c_str(), id_file) == 0);
    } else {
        FILE *f = nullptr;
        for (int i = 0; i < 600 && !(f = std::fopen(id_file, "rb")); ++i)
            usleep(100000);
        CHECK(f && std::fread(&id, sizeof(id), 1, f) == 1);
        std::fclose(f);
    }
    ncclComm_t comm;
    CHECK(ncclCommInitRank(&comm, world, id, rank) == ncclSuccess);

    // synthetic pairs: one rigid scene per pair, the pair's global index keys the generator (and the RANSAC sampler)
    std::vector<uint8_t> d1((size_t)pairs * n_kp * 32), d2(d1.size());
    std::vector<float> k1((size_t)pairs * n_kp * 2), k2(k1.size());
    std::vector<int32_t> n(pairs, n_kp);
    std::vector<double> K((size_t)pairs * 9, 0.0);
    std::vector<int64_t> gidx(pairs);
    for (int p = 0; p < pairs; ++p) {
        gidx[p] = (int64_t)rank * pairs + p;
        std::mt19937_64 rng(1000 + gidx[p]);
        std::uniform_real_distribution<double> U(0.0, 1.0);
        std::normal_distribution<double> noise(0.0, 0.3);
        double *Kp = &K[(size_t)p * 9];
        Kp[0] = Kp[4] = 525; Kp[2] = 320; Kp[5] = 240; Kp[8] = 1;
        const double yaw = 0.02 + 0.001 * p, c = std::cos(yaw), s = std::sin(yaw);
        for (int i = 0; i < n_kp; ++i) {
            const double Z = 2 + 8 * U(rng), X = (U(rng) - 0.5) * Z, Y = (U(rng) - 0.5) * 0.8 * Z;
            const double X2 = c * X - s * Z - 0.3, Z2 = s * X + c * Z;
            const size_t o = ((size_t)p * n_kp + i);
            k1[2 * o] = (float)(525 * X / Z + 320 + noise(rng)); k1[2 * o + 1] = (float)(525 * Y / Z + 240 + noise(rng));
            k2[2 * o] = (float)(525 * X2 / Z2 + 320 + noise(rng)); k2[2 * o + 1] = (float)(525 * Y / Z2 + 240 + noise(rng));
            for (int b = 0; b < 32; ++b) {
                const uint8_t v = (uint8_t)(rng() & 0xff);
                d1[o * 32 + b] = v;
                d2[o * 32 + b] = (i % 10 < 8) ? v : (uint8_t)(rng() & 0xff);
            }
        }
    }
    mvs_ctx *ctx = nullptr;
    mvs_batch *b = nullptr;
    CHECK(mvs_ctx_create(dev, &ctx) == MVS_OK);
    CHECK(mvs_batch_create(ctx, pairs, n_kp, 32, &b) == MVS_OK);
    CHECK(mvs_batch_upload(b, 0, pairs, d1.data(), k1.data(), n.data(), d2.data(), k2.data(), n.data(), K.data(), gidx.data()) == MVS_OK);
    mvs_params prm;
    mvs_params_default(&prm);
    prm.num_hypotheses = H;
    prm.sampler = MVS_SAMPLER_PHILOX;
    prm.seed = 42;
    prm.max_error_sq = 1e-2;
    void *d_all = nullptr;
    const size_t rec = sizeof(mvs_pair_result), block = (size_t)pairs * rec;
    CHECK(hipMalloc(&d_all, block * world) == hipSuccess);
    CHECK(mvs_batch_run(b, &prm, pairs) == MVS_OK);
    CHECK(mvs_batch_gather_results(b, pairs, comm, d_all) == MVS_OK);     // enqueued behind the kernels, no sync needed
    CHECK(mvs_batch_sync(b) == MVS_OK);
    std::vector<mvs_pair_result> all((size_t)pairs * world), mine(pairs);
    CHECK(hipMemcpy(all.data(), d_all, block * world, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(mvs_batch_download(b, 0, pairs, mine.data(), nullptr, nullptr, nullptr, nullptr) == MVS_OK);
    CHECK(std::memcmp(&all[(size_t)rank * pairs], mine.data(), block) == 0);
    int valid = 0;
    for (size_t i = 0; i < all.size(); ++i)
        valid += all[i].valid;
    std::printf("rank %d / %d on device %d: gathered %zu records of %zu bytes through RCCL, %d valid, own block identical\n", rank,
                world, dev, all.size(), rec, valid);
    CHECK(valid >= (int)all.size() * 3 / 4);
    (void)hipFree(d_all);
    mvs_batch_destroy(b);
    mvs_ctx_destroy(ctx);
    ncclCommDestroy(comm);
    if (rank == 0)
        std::remove(id_file);
    return 0;
}
