/* sse_oracle_internal.h — CPU ORACLE internals shared by its translation units (test infrastructure). */
#ifndef SSE_ORACLE_INTERNAL_H
#define SSE_ORACLE_INTERNAL_H
#include "sse_oracle.h"
#include "../include/sse_format.h"
#include <stdint.h>

struct ora_model {
    uint32_t nvars, nedges, nbonds;
    uint32_t *bond_a, *bond_b, *binfo;
    double *bweight;
    double *cumw; /* heat-bath cumulative max weights (heatbath.rs:16-35) */
    double wtot;
    double offset, gamma, h;
    double *mats; /* generic interactions (qmc_runner.rs:415-680): [nbonds][16] weights indexed in | out<<2, else NULL */
};

struct ora_replica {
    const ora_model *m;
    uint32_t cap, cutoff, n, replica;
    uint32_t *ops;
    uint8_t *state;
    uint64_t seed, epoch;
    uint64_t acc[8];
    /* scratch */
    uint32_t *parent, *cur;
    uint8_t *flip, *frozen, *touched;
};

static inline void draw(const ora_replica *r, uint32_t tag, uint32_t index, uint32_t out[4]) {
    uint32_t ctr[4] = {index, (uint32_t)r->epoch, r->replica,
                       (tag << 24) | (uint32_t)((r->epoch >> 32) & 0xFFFFFFu)};
    uint32_t key[2] = {(uint32_t)r->seed, (uint32_t)(r->seed >> 32)};
    ora_philox4x32_10(ctr, key, out);
}
/* Metropolis diagonal pass: one Philox call serves the slot pair (p, p^64): index = p with bit 6 cleared,
 * words (0,1) for the slot with bit 6 clear, words (2,3) for the other (include/sse_format.h). */
static inline void draw_diag(const ora_replica *r, uint32_t p, uint32_t out[2]) {
    uint32_t o[4];
    draw(r, SSE_TAG_DIAG, p & ~64u, o);
    if (p & 64u) { out[0] = o[2]; out[1] = o[3]; } else { out[0] = o[0]; out[1] = o[1]; }
}
static inline double u01(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }
static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }


double ora_bond_weight(const ora_model *m, uint32_t b, uint32_t in, uint32_t out);
#endif
