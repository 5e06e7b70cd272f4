// The C ABI of libpmc_hip.so used from plain C++/HIP -- no Python, no PyTorch: device memory from
// hipMalloc, the library only launches kernels on the stream it is given.
//
//   hipcc -O2 -I include examples/cabi_demo.cpp -L pypmc_amd/lib -lpmc_hip -Wl,-rpath,$PWD/pypmc_amd/lib -o cabi_demo
//   ./cabi_demo [N]
//
// A 3-component Gaussian mixture in D = 4 (parameters fixed below), N samples on a lattice; prints
// log q(x_n) of the first samples, the importance-weight sums against a second mixture, and the
// Rao-Blackwell statistics N_k -- tests/test_gpu_cabi_demo.py compares them with the oracle.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pmc_hip.h"

#define HIP_OK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 2;                                                                  \
        }                                                                              \
    } while (0)
#define PMC_OK_(x)                                                                     \
    do {                                                                               \
        if ((x) != 0) {                                                                \
            std::fprintf(stderr, "%s: %s\n", #x, pmc_last_error());                    \
            return 3;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? std::atoll(argv[1]) : 1000;
    const int D = 4, K = 3;
    // diagonal covariances: precision = diag(1 / var); log_norm = -D/2 log 2pi - 1/2 sum log var
    const double mu[K * D] = {0, 0, 0, 0, 2, -1, 0.5, 1, -3, 2, 1, -1};
    const double var[K * D] = {1, 2, 0.5, 1, 0.3, 0.7, 1.1, 2.0, 1.5, 0.4, 0.9, 1.2};
    const double weight[K] = {0.5, 0.3, 0.2};
    std::vector<double> prec(K * D * D, 0.0), log_norm(K);
    for (int k = 0; k < K; ++k) {
        double ld = 0.0;
        for (int i = 0; i < D; ++i) {
            prec[(k * D + i) * D + i] = 1.0 / var[k * D + i];
            ld += std::log(var[k * D + i]);
        }
        log_norm[k] = -0.5 * D * std::log(2.0 * M_PI) - 0.5 * ld;
    }
    std::vector<double> x((size_t)N * D);
    for (int64_t n = 0; n < N; ++n)
        for (int i = 0; i < D; ++i) x[n * D + i] = std::sin(0.37 * (double)n + 1.3 * i) * 3.0;

    // host: parameter packs (proposal = the mixture, target = the same means with unit variances)
    const int64_t stride = pmc_pack_stride(D);
    std::vector<double> pack(K * stride), tpack(K * stride), tprec(K * D * D, 0.0), tln(K);
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < D; ++i) tprec[(k * D + i) * D + i] = 1.0;
        tln[k] = -0.5 * D * std::log(2.0 * M_PI);
    }
    PMC_OK_(pmc_pack_components(K, D, mu, prec.data(), log_norm.data(), nullptr, nullptr, nullptr, weight, nullptr,
                                pack.data()));
    PMC_OK_(pmc_pack_components(K, D, mu, tprec.data(), tln.data(), nullptr, nullptr, nullptr, weight, nullptr,
                                tpack.data()));

    // device buffers
    double *d_x, *d_pack, *d_tpack, *d_out, *d_w, *d_scalars, *d_u, *d_stats;
    void *d_ws;
    const int64_t ws_bytes = pmc_workspace_bytes(N, K, D), ulen = pmc_tile_buffer_len(N, K);
    const int64_t nstats = 8 + K * pmc_stats_stride(D) + 2 * K;
    HIP_OK(hipMalloc(&d_x, sizeof(double) * N * D));
    HIP_OK(hipMalloc(&d_pack, sizeof(double) * K * stride));
    HIP_OK(hipMalloc(&d_tpack, sizeof(double) * K * stride));
    HIP_OK(hipMalloc(&d_out, sizeof(double) * N));
    HIP_OK(hipMalloc(&d_w, sizeof(double) * N));
    HIP_OK(hipMalloc(&d_scalars, sizeof(double) * 8));
    HIP_OK(hipMalloc(&d_u, sizeof(double) * ulen));
    HIP_OK(hipMalloc(&d_stats, sizeof(double) * nstats));
    HIP_OK(hipMalloc(&d_ws, (size_t)ws_bytes));
    HIP_OK(hipMemcpy(d_x, x.data(), sizeof(double) * N * D, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_pack, pack.data(), sizeof(double) * K * stride, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_tpack, tpack.data(), sizeof(double) * K * stride, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_stats, 0, sizeof(double) * nstats));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    PMC_OK_(pmc_timing_enable(1));                        // HIP events around every hot kernel on `stream`
    // log q, importance weights against the target mixture + their sums, one pass over x
    PMC_OK_(pmc_importance_weights(d_x, N, D, d_pack, K, PMC_KIND_GAUSS, d_tpack, K, PMC_KIND_GAUSS, d_out, nullptr,
                                   d_w, nullptr, d_scalars, d_ws, stream));
    // Rao-Blackwell responsibilities weighted by the importance weights and N_k / sum u d / sum u d d^T in
    // one call (at D = 4 one fused kernel: pmc_estep_is_fused() says d_u could be NULL)
    PMC_OK_(pmc_estep(d_x, N, D, d_pack, K, PMC_KIND_GAUSS, PMC_RESP_PMC_RB, 0, d_w, nullptr, d_u, nullptr, nullptr,
                      d_stats + 8, d_stats, d_ws, stream));
    // the same iteration evaluating the proposal ONCE: the weighting pass keeps the Mahalanobis forms, the
    // update forms rho from them (no Mahalanobis forms) -- N_k must come out the same to rounding (at D = 4 the
    // call above is the one-kernel E-step, whose sums are ordered differently; from D = 8 on: bit for bit)
    double *d_tiles, *d_w2, *d_stats2, *d_scalars2;
    HIP_OK(hipMalloc(&d_tiles, sizeof(double) * pmc_maha_tiles_size(N, K)));
    HIP_OK(hipMalloc(&d_w2, sizeof(double) * N));
    HIP_OK(hipMalloc(&d_stats2, sizeof(double) * nstats));
    HIP_OK(hipMalloc(&d_scalars2, sizeof(double) * 8));
    PMC_OK_(pmc_importance_weights_keep(d_x, N, D, d_pack, K, PMC_KIND_GAUSS, d_tpack, K, PMC_KIND_GAUSS, nullptr,
                                        nullptr, d_w2, nullptr, d_scalars2, d_ws, d_tiles, stream));
    PMC_OK_(pmc_estep_from_tiles(d_x, N, D, d_pack, K, PMC_KIND_GAUSS, 0, d_w2, d_tiles, K, d_u, nullptr, d_stats2 + 8,
                                 d_stats2, d_ws, stream));
    HIP_OK(hipStreamSynchronize(stream));
    pmc_timing timings[8];
    int ntimings = 0;
    PMC_OK_(pmc_get_timings(timings, 8, &ntimings));

    std::vector<double> out(N), scalars(8), stats(nstats);
    HIP_OK(hipMemcpy(out.data(), d_out, sizeof(double) * N, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(scalars.data(), d_scalars, sizeof(double) * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(stats.data(), d_stats, sizeof(double) * nstats, hipMemcpyDeviceToHost));
    std::printf("abi %d arch ", pmc_abi_version());
    char arch[64];
    pmc_device_arch(0, arch, sizeof(arch));
    std::printf("%s N %lld\n", arch, (long long)N);
    for (int n = 0; n < 5 && n < N; ++n) std::printf("logq %d %.17g\n", n, out[n]);
    std::printf("sums %.17g %.17g %.17g\n", scalars[0], scalars[1], scalars[2]);
    for (int k = 0; k < K; ++k) std::printf("N_k %d %.17g\n", k, stats[8 + k * pmc_stats_stride(D)]);
    std::printf("fused %d\n", pmc_estep_is_fused(K, D, PMC_KIND_GAUSS, PMC_RESP_PMC_RB));
    std::vector<double> stats2(nstats);
    HIP_OK(hipMemcpy(stats2.data(), d_stats2, sizeof(double) * nstats, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int64_t i = 8; i < 8 + K * pmc_stats_stride(D); ++i) {
        const double scale = std::fabs(stats[i]) > 1e-300 ? std::fabs(stats[i]) : 1.0;
        worst = std::fmax(worst, std::fabs(stats2[i] - stats[i]) / scale);
    }
    std::printf("from_tiles worst relative difference %.3g\n", worst);
    for (int i = 0; i < ntimings && i < 8; ++i)
        std::printf("timing %s: %d launches, %.4f ms, %.3g flop, %.3g bytes\n", timings[i].name, timings[i].calls,
                    timings[i].ms, timings[i].flops, timings[i].bytes);

    // the cross-GPU exchange through the library (RCCL opened at run time): a communicator of this ONE rank -- with
    // more GPUs rank 0 hands `id` to the other processes -- sums the statistics buffer in place on `stream`
    char id[PMC_COMM_ID_BYTES];
    pmc_comm *comm = nullptr;
    PMC_OK_(pmc_comm_unique_id(id));
    PMC_OK_(pmc_comm_init(0, 1, id, 0, &comm));
    int rank = -1, world = -1;
    PMC_OK_(pmc_comm_rank(comm, &rank, &world));
    PMC_OK_(pmc_comm_allreduce_sum(comm, d_stats, nstats, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(stats2.data(), d_stats, sizeof(double) * nstats, hipMemcpyDeviceToHost));
    bool same = true;
    for (int64_t i = 0; i < nstats; ++i) same = same && stats2[i] == stats[i];
    std::printf("comm rank %d of %d allreduce %s\n", rank, world, same ? "identical" : "DIFFERENT");
    PMC_OK_(pmc_comm_destroy(comm));
    return 0;
}
