/*
 * sse_oracle.h — CPU ORACLE for the SSE sweep.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * It is a plain-C, sequential, one-replica-at-a-time restatement of the reference algorithms
 * (file:line citations at each function in sse_oracle.c).  It shares only the data FORMAT
 * (include/sse_format.h) with the HIP product path; every algorithm is written independently
 * and in the most obvious sequential form, so that agreement with the parallel kernels is evidence.
 *
 * Pinning status: the reference is Rust with un-vendored dependencies and no Rust
 * toolchain exists in this image, so oracle/_ref cannot be built.  The reference's own tests pin only
 * structural invariants (verify()) and a few helper known-answers; energies/magnetisations are pinned
 * here by exact diagonalisation fixtures (tests/golden/ed_*.json, generator tests/golden/make_ed_golden.py).
 * RNG-stream-level parity with the Rust path is impossible by construction (the reference threads a
 * sequential rand::Rng; this build uses counter-based Philox), i.e. "parity unpinned" at the bit level
 * against the reference, statistical (1σ) against exact results.
 */
#ifndef SSE_ORACLE_H
#define SSE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_model ora_model;
typedef struct ora_replica ora_replica;

/* update flags for ora_timestep(s) — same numeric values as include/isingmc_hip.h */
#define ORA_FLAG_LOOP 1u      /* one directed loop after the diagonal pass (Qmc::timestep) */
#define ORA_FLAG_NO_CLUSTER 2u /* skip the cluster step */
#define ORA_FLAG_HEATBATH 4u  /* heat-bath diagonal rule instead of Metropolis */
#define ORA_FLAG_RVB 8u       /* RVB sweep of (N+1)/2 attempts before the cluster step */

void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

ora_model *ora_model_create(uint32_t nvars, uint32_t nedges, const uint32_t *edge_a, const uint32_t *edge_b,
                            const double *J, double gamma, double h);
/* generic 1- and 2-variable interactions (qmc_runner.rs:415-680); mats[b][in | out<<2] */
ora_model *ora_model_create_generic(uint32_t nvars, uint32_t nbonds, const uint32_t *k, const uint32_t *var_a,
                                    const uint32_t *var_b, const double *mats, double offset);
/* Interaction::at / sym_under_ising (qmc_runner.rs:573-612,639-664): mat is [4^k], or [2^k] when diagonal != 0 */
double ora_interaction_at(uint32_t k, uint32_t diagonal, const double *mat, const uint8_t *inputs, const uint8_t *outputs);
int ora_interaction_sym_under_ising(uint32_t k, uint32_t diagonal, const double *mat);
void ora_model_destroy(ora_model *m);
uint32_t ora_model_nbonds(const ora_model *m);
double ora_model_offset(const ora_model *m);

/* init_state may be NULL: random spins from Philox(tag INIT). */
ora_replica *ora_replica_create(const ora_model *m, uint32_t capacity, uint32_t cutoff0, uint64_t seed,
                                uint32_t replica, const uint8_t *init_state);
void ora_replica_destroy(ora_replica *r);

void ora_diagonal_update(ora_replica *r, double beta);
void ora_heatbath_update(ora_replica *r, double beta);
uint32_t ora_cluster_update(ora_replica *r, double prob);
void ora_flip_free_spins(ora_replica *r);
uint32_t ora_loop_update(ora_replica *r); /* returns number of vertices visited */
/* RvbUpdater::rvb_update_with_ising_weight (qmc_traits/rvb.rs:88-290): `updates` attempts; returns #accepted */
uint32_t ora_rvb_update(ora_replica *r, uint32_t updates);
uint32_t ora_find_overlapping_starts(uint32_t p_start, uint32_t p_end, uint32_t cutoff, const uint32_t *fp, uint32_t L,
                                     uint32_t *out);
uint32_t ora_remove_doubles(uint32_t *v, uint32_t n);
int ora_timestep(ora_replica *r, double beta, uint32_t flags);
int ora_timesteps(ora_replica *r, uint64_t t, double beta, uint32_t sampling_freq, uint32_t flags);
int ora_verify(const ora_replica *r);
/* itime_fold (fast_ops.rs:1296-1315) of m, m^2, |m| over p = 0..cutoff-1 */
void ora_itime_magnetization(const ora_replica *r, int64_t *sum_m, uint64_t *sum_m2, uint64_t *sum_abs);

/* one parallel-tempering step of one chain; by_slot[t] = replica at temperature t (pointers are swapped) */
uint64_t ora_pt_step(ora_replica **by_slot, const double *betas, uint32_t ntemps, uint64_t seed, uint32_t chain,
                     uint64_t step);

uint32_t ora_get_n(const ora_replica *r);
uint32_t ora_get_cutoff(const ora_replica *r);
int ora_set_cutoff(ora_replica *r, uint32_t cutoff);
uint64_t ora_get_epoch(const ora_replica *r);
void ora_set_epoch(ora_replica *r, uint64_t epoch);
void ora_get_state(const ora_replica *r, uint8_t *out);
void ora_set_state(ora_replica *r, const uint8_t *in);
void ora_get_ops(const ora_replica *r, uint32_t *out /* cutoff words */);
int ora_set_ops(ora_replica *r, const uint32_t *words, uint32_t cutoff);
uint32_t ora_get_bond_count(const ora_replica *r, uint32_t bond);
/* acc[0]=sum n, acc[1]=nsamples, acc[2]=sum |2up-N|, acc[3]=sum (2up-N)^2, acc[4]=vertices visited by
 * off-diagonal passes, acc[5]=slots visited by diagonal passes, acc[6]=sum over samples of the number of transverse-bond ops */
void ora_get_accumulators(const ora_replica *r, uint64_t acc[8]);
void ora_reset_accumulators(ora_replica *r);

#ifdef __cplusplus
}
#endif
#endif
