/*
 * sse_oracle_batch.c — CPU ORACLE batch runner (test infrastructure, NOT product).
 * Runs independent replicas on host threads, one replica per thread at a time, the way the reference
 * fans graphs out over rayon threads (src/sse/parallel_tempering/tempering_container.rs:367-371,
 * :428-434).  Used by tests for convenience and by bench.py's cpu_baseline leg.
 */
#include "sse_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

int ora_batch_timesteps(ora_replica **reps, uint32_t nreplicas, uint64_t t, const double *betas,
                        uint32_t sampling_freq, uint32_t flags, int nthreads) {
    int rc = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(| : rc)
    for (int64_t i = 0; i < (int64_t)nreplicas; ++i)
        rc |= ora_timesteps(reps[i], t, betas[i], sampling_freq, flags);
    return rc;
}

int ora_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
