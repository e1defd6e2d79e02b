/* Sanitizer driver for the oracle (test infrastructure): every pass of the restatement on two small models under
 * AddressSanitizer / UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are not available on this pool).
 * Built and run by tests/test_oracle_cpu.py::test_oracle_is_clean_under_asan_ubsan. */
#include <stdio.h>
#include <stdlib.h>
#include "sse_oracle.h"
int main(void) {
    /* 3x3 periodic Villain-like lattice with a longitudinal field, and an open chain with random |J| */
    uint32_t ea[18], eb[18]; double J[18];
    int k = 0;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) { ea[k] = j * 3 + i; eb[k] = j * 3 + (i + 1) % 3; J[k] = -1.0; k++; }
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) { ea[k] = j * 3 + i; eb[k] = ((j + 1) % 3) * 3 + i; J[k] = (i % 2) ? -1.0 : 1.0; k++; }
    ora_model *m = ora_model_create(9, 18, ea, eb, J, 0.7, 0.2);
    int bad = 0;
    const uint32_t flagsets[] = {0u, ORA_FLAG_LOOP, ORA_FLAG_HEATBATH, ORA_FLAG_LOOP | ORA_FLAG_HEATBATH, ORA_FLAG_NO_CLUSTER | ORA_FLAG_LOOP};
    for (unsigned f = 0; f < sizeof flagsets / sizeof flagsets[0]; ++f) {
        ora_replica *r = ora_replica_create(m, 2048, 9, 99, f, NULL);
        if (ora_timesteps(r, 300, 1.5, 2, flagsets[f]) != 0 || !ora_verify(r)) bad = 1;
        int64_t a; uint64_t b, c;
        ora_itime_magnetization(r, &a, &b, &c);
        ora_replica_destroy(r);
    }
    ora_model *m2 = ora_model_create(9, 18, ea, eb, J, 0.7, 0.0);
    {
        ora_replica *r = ora_replica_create(m2, 2048, 9, 7, 0, NULL);
        if (ora_timesteps(r, 300, 1.5, 1, ORA_FLAG_RVB) != 0 || !ora_verify(r)) bad = 1;
        ora_replica_destroy(r);
    }
    { /* parallel tempering step over four replicas */
        ora_replica *slot[4];
        const double betas[4] = {0.5, 0.9, 1.4, 2.0};
        for (int t = 0; t < 4; ++t) slot[t] = ora_replica_create(m2, 2048, 9, 5, (uint32_t)t, NULL);
        for (int s = 0; s < 20; ++s) {
            for (int t = 0; t < 4; ++t) if (ora_timesteps(slot[t], 3, betas[t], 1, 0) != 0) bad = 1;
            ora_pt_step(slot, betas, 4, 5, 0, (uint64_t)s);
        }
        for (int t = 0; t < 4; ++t) { if (!ora_verify(slot[t])) bad = 1; ora_replica_destroy(slot[t]); }
    }
    ora_model_destroy(m); ora_model_destroy(m2);
    printf(bad ? "FAILED\n" : "clean\n");
    return bad;
}
