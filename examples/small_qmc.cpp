// examples/small_qmc.rs of the reference, through the C ABI (include/isingmc_hip.h) from compiled code:
// a 4-site ring with one ferromagnetic bond, transverse field 1, 1000 sweeps at beta = 1 — here for a batch of
// replicas at once, printing the energy estimate (exact diagonalisation: -4.0550 for this model).
//
//   g++ -std=c++17 -Iinclude examples/small_qmc.cpp -Lisingmontecarlo_amd -lisingmc_hip
//       -Wl,-rpath,$PWD/isingmontecarlo_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -o small_qmc && ./small_qmc
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "isingmc_hip.h"

int main(int argc, char **argv) {
    const uint32_t R = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 256;
    const uint32_t edges[] = {0, 1, 1, 2, 2, 3, 3, 0};
    const double J[] = {-1.0, 1.0, 1.0, 1.0};
    isingmc_config cfg{};
    cfg.struct_size = sizeof cfg;
    cfg.nreplicas = R; cfg.nvars = 4; cfg.nedges = 4;
    cfg.edges = edges; cfg.J = J;
    cfg.transverse = 1.0; cfg.longitudinal = 0.0;
    cfg.capacity = 256; cfg.cutoff0 = 3; // QmcIsingGraph::new_with_rng(edges, transverse, 0., 3, rng, None)
    cfg.seed = 2024; cfg.device = -1;
    isingmc_batch *b = nullptr;
    if (isingmc_create(&cfg, &b) != ISINGMC_OK) { std::fprintf(stderr, "create: %s\n", isingmc_last_error(nullptr)); return 1; }
    std::vector<double> beta(R, 1.0);
    // thermalise, then g.timesteps(1000, 1.0)
    if (isingmc_timesteps(b, 200, beta.data(), 0, 0) != ISINGMC_OK || isingmc_reset_accumulators(b) != ISINGMC_OK ||
        isingmc_timesteps(b, 1000, beta.data(), 1, 0) != ISINGMC_OK) {
        std::fprintf(stderr, "timesteps: %s\n", isingmc_last_error(b));
        return 1;
    }
    std::vector<uint64_t> acc(8 * (size_t)R);
    if (isingmc_get_accumulators(b, acc.data()) != ISINGMC_OK) return 1;
    std::vector<uint8_t> ok(R);
    if (isingmc_verify(b, ok.data()) != ISINGMC_OK) return 1;
    double mean = 0, sq = 0;
    for (uint32_t r = 0; r < R; ++r) {
        if (!ok[r]) { std::fprintf(stderr, "replica %u failed verification\n", r); return 2; }
        // QmcStepper::get_energy_for_average_n (qmc_ising.rs:805-809): E = -<n>/beta + offset
        const double e = -((double)acc[8 * r] / (double)acc[8 * r + 1]) / beta[r] + isingmc_get_offset(b);
        mean += e; sq += e * e;
    }
    mean /= R;
    const double sem = R > 1 ? std::sqrt((sq / R - mean * mean) / (R - 1)) : 0.0;
    std::printf("energy %.6f sem %.6f replicas %u\n", mean, sem, R);
    isingmc_destroy(b);
    return 0;
}
