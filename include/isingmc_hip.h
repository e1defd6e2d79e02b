/*
 * isingmc_hip.h — C ABI of the MI355X-native SSE sweep engine (libisingmc_hip.so).
 *
 * The reference (Renmusxd/IsingMonteCarlo, crate `qmc`) is 100 % safe Rust with no FFI; its extension
 * point is "implement the manager traits on your own container M and instantiate QmcIsingGraph<R,M> /
 * Qmc<R,M>" (src/sse/qmc_ising.rs:20-23,80-91; src/sse/qmc_runner.rs:21).  A GPU manager cannot serve the
 * per-slot closures of those traits across a device boundary, so the boundary sits at the sweep-level
 * default methods, batch-first (one handle = R independent replicas, as ParallelQmcTimeSteps treats
 * graphs: src/sse/parallel_tempering/tempering_container.rs:321-340).  Each entry point below names the
 * reference interface it replaces.  The Rust-side binding a maintainer would add is in INTEGRATION.md.
 *
 * Conventions: every function returns 0 on success, a negative ISINGMC_E* code otherwise; the message is
 * available from isingmc_last_error().  Array arguments are borrowed for the duration of the call; outputs
 * are caller-allocated.  A handle may be moved between host threads but used by one at a time (!Sync),
 * matching `&mut self` on every reference update.  There is NO CPU fallback: without a HIP device
 * isingmc_create fails with ISINGMC_ENODEVICE.
 */
#ifndef ISINGMC_HIP_H
#define ISINGMC_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "sse_format.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ISINGMC_OK 0
#define ISINGMC_EINVAL (-1)    /* bad argument (maps the reference's Result<(),String> errors) */
#define ISINGMC_ENODEVICE (-2) /* no usable HIP device / HIP runtime error */
#define ISINGMC_ECAPACITY (-3) /* cutoff would exceed the preallocated op-string capacity */
#define ISINGMC_EINTEGRITY (-4) /* device-side integrity failure (the reference would panic) */
#define ISINGMC_ENOTIMPL (-5)
#define ISINGMC_ELIMIT (-6)     /* a directed loop did not close within 64*cutoff+1024 vertices (the reference's loop is unbounded,
                                   directed_loop.rs:217-301); the replica's flag is cleared by isingmc_clear_errors */

/* update flags (isingmc_timesteps / isingmc_timestep) */
#define ISINGMC_FLAG_LOOP 1u       /* Qmc::set_do_loop_updates(true): one directed loop per step (qmc_runner.rs:268,366) */
#define ISINGMC_FLAG_NO_CLUSTER 2u /* skip the cluster step */
#define ISINGMC_FLAG_HEATBATH 4u   /* set_enable_heatbath(true) (qmc_ising.rs:444) / set_do_heatbath (qmc_runner.rs:258) */
#define ISINGMC_FLAG_RVB 8u        /* set_run_rvb(true) (qmc_ising.rs:435) */
#define ISINGMC_FLAG_PREP 0x10000u /* profiling label only: run the identical kernel under its "data preparation"
                                      symbol so that profilers separate input generation from the measured sweeps */

/* isingmc_config.flags */
#define ISINGMC_CFG_NO_LDS_TABLES 1u /* keep the bond table in HBM even when it would fit in LDS (testing) */
#define ISINGMC_CFG_PER_REPLICA_J 4u /* `J` holds nreplicas x nedges couplings (row r = replica r): independent disorder
                                        realisations on one graph. */
#define ISINGMC_CFG_GLOBAL_TABLES 8u /* keep the per-variable scan tables (spins, cut ranks) in a per-replica HBM scratch instead of
                                        LDS.  Chosen automatically for models whose tables exceed LDS (N >~ 10^4 variables, e.g. a
                                        32^3 lattice); this flag forces the path on any model (testing).  Needs
                                        ISINGMC_CFG_NO_LDS_TABLES or non-uniform couplings; slots_per_lane 1 or 4; no RVB updates. */
#define ISINGMC_CFG_NO_FAST_DIAG 16u /* run the diagonal pass through the general kernel even where the instruction-trimmed one
                                        (csrc/sse_fast.hip.h: uniform |J|, N <= 4096, 4 waves per replica) applies (testing / A-B timing) */
#define ISINGMC_CFG_FAST_LABEL 32u /* experimental: the trimmed diagonal kernel also labels the worldline segments and hands them to the
                                      cluster update of the same timestep, which then only runs the union-find (same results; on MI355X the
                                      diagonal launch loses more than the cluster update gains, see DESIGN.md, so it is off by default) */
#define ISINGMC_CFG_COMPACT 64u /* experimental: the trimmed diagonal kernel also writes the occupied slots as a dense list and the cluster
                                   update of the same timestep scans that list (n instead of M elements; same results; on MI355X the two
                                   launches trade 0.1 ms for 0.1 ms, see DESIGN.md, so it is off by default) */
#define ISINGMC_CFG_RVB_SERIAL_GROWTH 128u /* RVB sweeps grow the clusters of their attempts one at a time instead of a batch of them side by
                                            side on the waves of the workgroup (testing: the results are the same either way) */
#define ISINGMC_CFG_NO_LEAN_CLUSTER 256u /* run cluster updates through the general kernel even where the dedicated one (csrc/sse_cluster.hip.h:
                                           LDS edge tables, N <= 4095, default wave counts) applies (testing / A-B timing) */
#define ISINGMC_CFG_NO_DEFERRED_FLIPS 512u /* the dedicated cluster kernel rewrites the op-strings itself instead of leaving one flip byte per slot
                                             for the diagonal launch of the next timestep to apply (testing / A-B timing) */
#define ISINGMC_CFG_NO_PM_DECODE 1024u /* large disorder batches (per-replica coupling SIGNS, uniform |J| and fields, tables in HBM) decode
                                          bonds through the per-replica 16-byte records like any other model instead of the shared compact edge
                                          table + per-replica sign bits (testing / A-B timing) */
#define ISINGMC_CFG_RVB_FUSED 2048u /* run RVB sweeps through the fused kernel (growth and attempts in one launch, one replica per CU) even where
                                       the two-launch form applies (csrc/sse_rvb_split.hip.h); same results (testing / measurements) */
#define ISINGMC_CFG_FUSED_LAUNCH 2u  /* run whole timesteps inside one kernel launch instead of a diagonal-pass launch
                                        followed by an off-diagonal launch per timestep (same results, lower occupancy) */

typedef struct isingmc_batch isingmc_batch;

typedef struct isingmc_config {
    uint32_t struct_size;     /* = sizeof(isingmc_config) */
    uint32_t nreplicas;       /* R independent replicas resident on this device */
    uint32_t nvars;           /* N (reference: max edge index + 1, qmc_ising.rs:92) */
    uint32_t nedges;          /* E */
    const uint32_t *edges;    /* [2E]: a0,b0,a1,b1,...  (Vec<(Edge,f64)>, qmc_ising.rs:81) */
    const double *J;          /* [E]  couplings, J>0 antiferromagnetic (src/lib.rs:29); [R][E] with ISINGMC_CFG_PER_REPLICA_J */
    double transverse;        /* Gamma */
    double longitudinal;      /* h */
    uint32_t capacity;        /* op-string slots preallocated per replica (>= cutoff0) */
    uint32_t cutoff0;         /* initial cutoff (QmcIsingGraph::new_with_rng `cutoff`) */
    uint64_t seed;            /* Philox key; replaces the reference's `rng` argument */
    uint32_t replica_offset;  /* global index of local replica 0 (replica sharding over GPUs) */
    int32_t device;           /* HIP device ordinal, -1 = current device */
    const uint8_t *init_state; /* [R][N] 0/1 or NULL = random (make_random_spin_state, classical/graph.rs:451) */
    uint32_t waves_per_replica; /* 0 = auto (4); workgroup = this many wave64s cooperating on one replica: 1,4,6,8,16 */
    uint32_t slots_per_lane;    /* 0 = auto (4); op-string slots each lane holds per tile: 1,2,4 */
    uint32_t flags;             /* ISINGMC_CFG_* */
    uint32_t lds_uf_ids_limit;  /* 0 = as many cluster-segment ids as fit in LDS; smaller values force the HBM
                                   union-find path earlier (testing) */
    uint32_t waves_offdiag;     /* wave64s per replica for launches without a diagonal pass (directed loop, cluster,
                                 * free spins): 0 = same as waves_per_replica when that is given, otherwise decided
                                 * per launch (16 while the scan tables and the union-find fit in LDS); 1, 4, 6, 8, 16 */
    /* Generic interactions (qmc::sse::Qmc, qmc_runner.rs:94-156, Interaction :415-680).  When `interactions` is non-NULL
     * the batch is built from them and edges / J / transverse / longitudinal are ignored: bond b = interaction b.
     * Available updates: diagonal (Metropolis or heat-bath), directed loop, free spins, and cluster updates when every
     * interaction is symmetric under a global spin flip (Qmc::cluster_update, qmc_runner.rs:222-236; one-variable
     * interactions with four equal entries are the cluster edges).  RVB updates return ISINGMC_ENOTIMPL. */
    const struct isingmc_interaction *interactions;
    uint32_t ninteractions;
    double energy_offset;       /* added to -<n>/beta by isingmc_get_offset (sum of the offsets the caller absorbed,
                                   Qmc::make_interaction_and_offset) */
    /* Per-replica fields, only with ISINGMC_CFG_PER_REPLICA_J (every replica then runs on its own bond table): [R] transverse
     * fields / [R] longitudinal fields replacing `transverse` / `longitudinal`; NULL = the scalar for every replica.  The
     * longitudinal fields must be all zero (|h| <= DBL_EPSILON) or all non-zero: the bond numbering (qmc_ising.rs:228-246) is
     * common to a batch.  Parallel tempering between graphs whose Gamma and h differ (GraphWeights::relative_weight,
     * tempering_traits.rs:126-155) builds on this: the tables belong to the temperature slot. */
    const double *transverse_r;
    const double *longitudinal_r;
} isingmc_config;

/* One interaction: k = 1 or 2 variables and the 4^k matrix of the reference (Interaction::at, qmc_runner.rs:573-612):
 * index = outputs then inputs, first variable most significant, i.e. for k = 2  (out0 out1 in0 in1)  as a 4-bit number.
 * All entries must be >= 0 (they are sampling weights). */
typedef struct isingmc_interaction {
    uint32_t nvars;           /* 1 or 2 (more: ISINGMC_ENOTIMPL, the operator word holds two variables) */
    uint32_t vars[2];
    uint32_t diagonal_only;   /* 0: mat is the full [4^nvars] matrix (InteractionType::Full); 1: mat holds the [2^nvars] diagonal
                                 only (InteractionType::Diagonal, qmc_runner.rs:594-610), every off-diagonal weight is 0 */
    const double *mat;
} isingmc_interaction;

/* Interaction::at (qmc_runner.rs:573-612): the weight for inputs[nvars] / outputs[nvars] (bytes 0/1).  Host-side, no device
 * needed; isingmc_create tabulates every interaction through this function. */
int isingmc_interaction_at(const isingmc_interaction *it, const uint8_t *inputs, const uint8_t *outputs, double *out);
/* Interaction::sym_under_ising (qmc_runner.rs:639-664), restated with its index range as written there: a full matrix is
 * compared on indices [0, 2^nvars) against their bit-complements, a diagonal one on [0, 2^(nvars/2)).  (The engine itself
 * enables cluster updates only when EVERY entry equals its complement's, which implies this.) */
int isingmc_interaction_sym_under_ising(const isingmc_interaction *it, int *out);

/* QmcIsingGraph::new_with_rng (qmc_ising.rs:131-148) + OpContainerConstructor::new_with_bonds
 * (op_container.rs:110), for R replicas at once. */
int isingmc_create(const isingmc_config *cfg, isingmc_batch **out);
/* Drop */
void isingmc_destroy(isingmc_batch *b);
/* error text of the last failing call on this handle (b may be NULL: last isingmc_create failure) */
const char *isingmc_last_error(const isingmc_batch *b);

/* DiagonalUpdater::make_diagonal_update_with_rng_and_state_ref (qmc_traits/diagonal.rs:114-135), or with
 * ISINGMC_FLAG_HEATBATH HeatBathDiagonalUpdater::make_heatbath_diagonal_update_with_rng_and_state_ref
 * (qmc_traits/heatbath.rs:106-127).  beta[R].  Also applies cutoff = max(cutoff, n + n/2)
 * (qmc_ising.rs:269,786; qmc_runner.rs:197). */
int isingmc_diagonal_update(isingmc_batch *b, const double *beta, uint32_t flags);
/* ClusterUpdater::flip_each_cluster_rng (qmc_traits/cluster.rs:36-172) with the weight function of
 * qmc_ising.rs:759-775.  n_clusters[R] may be NULL. */
int isingmc_cluster_update(isingmc_batch *b, double prob, uint32_t *n_clusters);
/* LoopUpdater::make_loop_update_with_rng (qmc_traits/directed_loop.rs:103-171); lengths[R] may be NULL. */
int isingmc_loop_update(isingmc_batch *b, uint32_t *lengths);
/* RvbUpdater::rvb_update_with_ising_weight (qmc_traits/rvb.rs:88-290) as QmcIsingGraph::single_rvb_sweep drives it
 * (qmc_ising.rs:323-418): `updates` attempts per replica (0 = (N+1)/2); successes[R] may be NULL. */
int isingmc_rvb_update(isingmc_batch *b, uint32_t updates, uint32_t *successes);
/* qmc_ising.rs:780-784 / Qmc::flip_free_bits (qmc_runner.rs:241-255) */
int isingmc_flip_free_spins(isingmc_batch *b);
/* QmcStepper::timesteps_measure_with_self (qmc_traits/qmc_stepper.rs:133-162) over
 * QmcIsingGraph::timestep (qmc_ising.rs:644-795) / Qmc::timestep (qmc_runner.rs:363-377), fused on device:
 * t steps, sampling when (step+1) % sampling_freq == 0, accumulating into the per-replica accumulators. */
int isingmc_timesteps(isingmc_batch *b, uint64_t t, const double *beta, uint32_t sampling_freq, uint32_t flags);

/* accumulators [R][8] (u64): 0 sum n, 1 samples, 2 sum |2up-N|, 3 sum (2up-N)^2, 4 vertices visited by
 * off-diagonal passes, 5 slots visited by diagonal passes, 6 sum of transverse-op counts, 7 reserved.
 * Energy: QmcStepper::get_energy_for_average_n (qmc_ising.rs:805-809) = -(acc0/acc1)/beta + offset. */
int isingmc_get_accumulators(isingmc_batch *b, uint64_t *out);
int isingmc_reset_accumulators(isingmc_batch *b);
/* restore accumulators saved by isingmc_get_accumulators (checkpoint / resume): in[nrows][8] */
int isingmc_set_accumulators(isingmc_batch *b, const uint64_t *in);
/* Device-side error flags are sticky per replica (a replica that hit ISINGMC_ECAPACITY / ISINGMC_ELIMIT / an integrity
 * error is skipped by later launches).  This clears all of them; isingmc_import_ops clears the flag of the replica it
 * rewrites.  The reference panics instead (release = abort, Cargo.toml:43), so there is nothing to mirror. */
int isingmc_clear_errors(isingmc_batch *b);
/* QmcIsingGraph::get_offset (qmc_ising.rs:558) */
double isingmc_get_offset(const isingmc_batch *b);
/* the same per replica, out[R] (differs between replicas only with ISINGMC_CFG_PER_REPLICA_J) */
int isingmc_get_offsets(const isingmc_batch *b, double *out);
uint32_t isingmc_num_bonds(const isingmc_batch *b);

/* state_ref / clone_state / state_mut (qmc_ising.rs:497-509,797): out/in are [N] bytes 0/1 of replica r;
 * r == UINT32_MAX addresses all replicas, buffers are then [R][N]. */
int isingmc_get_state(isingmc_batch *b, uint32_t r, uint8_t *out);
int isingmc_set_state(isingmc_batch *b, uint32_t r, const uint8_t *in);
/* OpContainer::get_n / get_cutoff / set_cutoff (qmc_traits/op_container.rs:119-123), out[R] */
int isingmc_get_n(isingmc_batch *b, uint32_t *out);
int isingmc_get_cutoff(isingmc_batch *b, uint32_t *out);
int isingmc_set_cutoff(isingmc_batch *b, uint32_t r, uint32_t cutoff);
int isingmc_get_epoch(isingmc_batch *b, uint64_t *out);
/* restore the per-replica update counters (the Philox epoch): with isingmc_import_ops / isingmc_set_state /
 * isingmc_set_cutoffs this makes a resumed batch continue bit-exactly (checkpoint / resume; the reference's serde
 * support, qmc_ising.rs:1001-1087, serialises the same fields plus its RNG) */
int isingmc_set_epoch(isingmc_batch *b, const uint64_t *epochs);
/* OpContainer::itime_fold (fast_ops.rs:1296-1315; QmcStepper::imaginary_time_fold, qmc_ising.rs:815-821) for the
 * magnetisation m = sum_v (2 s_v - 1) of the propagated state: per replica the sums over p = 0..cutoff-1 of m, m^2
 * and |m| (divide by the cutoff for imaginary-time averages).  Arbitrary closures fold on the host over
 * isingmc_export_ops. */
int isingmc_itime_magnetization(isingmc_batch *b, int64_t *sum_m, uint64_t *sum_m2, uint64_t *sum_abs_m);
/* OpContainer::get_count (op_container.rs:129; fast_ops.rs:1281-1294) */
int isingmc_get_bond_count(isingmc_batch *b, uint32_t r, uint32_t bond, uint32_t *out);
/* get_pth for every p (op_container.rs:127): words[cutoff] in the sse_format.h encoding */
int isingmc_export_ops(isingmc_batch *b, uint32_t r, uint32_t *words, uint32_t nwords);
/* FastOps::new_from_ops (fast_ops.rs:80-174): install an op-string (words[nwords], slot p = index) */
int isingmc_import_ops(isingmc_batch *b, uint32_t r, const uint32_t *words, uint32_t nwords);
/* DebugOps::count_diagonal_and_off and count_constant_ops (qmc_debug.rs:10-41) for every replica: out[R][3] = diagonal ops,
 * off-diagonal ops (their sum is get_n), constant ops */
int isingmc_debug_counts(isingmc_batch *b, uint32_t *out);
/* Verify::verify (qmc_ising.rs:829-860; op_container.rs:137-159), ok[R] */
int isingmc_verify(isingmc_batch *b, uint8_t *ok);

/* ---- parallel tempering (src/sse/parallel_tempering/tempering_container.rs) -------------------------------
 * Configurations never move: a tempering swap exchanges the TEMPERATURE LABELS of two configurations (the
 * reference swaps (manager,state) between (graph,beta) slots, qmc_ising.rs:593-602 — the same thing seen
 * from the other side).  The per-replica beta[] argument of isingmc_timesteps carries the current labels. */
/* TemperingContainer::tempering_step decisions (:121-149, :241-302) for nchains independent chains of ntemps
 * temperatures sharing one Hamiltonian; host-side control logic exactly like the reference's container.
 *   betas[ntemps]; n_of_config[nchains*ntemps] = operator count of every configuration (global ids);
 *   config_at[ntemps*nchains] in/out: configuration id at slot t*nchains+chain; *nswaps += swaps done.
 * Philox tag PT: replica field = chain, epoch = step, index 0 = order coin, 1+t = pair (t,t+1). */
int isingmc_pt_decide(uint64_t seed, uint64_t step, uint32_t nchains, uint32_t ntemps, const double *betas,
                      const uint32_t *n_of_config, uint32_t *config_at, uint64_t *nswaps);
/* set_op_cutoff for every replica at once (tempering_container.rs:129-137): cutoffs[R], each only grows */
int isingmc_set_cutoffs(isingmc_batch *b, const uint32_t *cutoffs);
/* replica r accumulates into row rows[r] of an accumulator table with nrows rows (default: nrows = R, rows[r] = r);
 * isingmc_get_accumulators then returns [nrows][8].  Used to keep statistics per temperature slot. */
int isingmc_set_accumulator_rows(isingmc_batch *b, uint32_t nrows, const uint32_t *rows);

/* ---- native tempering step: temperature blocks sharded over ranks, neighbour exchange point to point --------------------
 * Rank g of `world` owns ntemps/world consecutive temperatures of every chain; its batch holds ntemps/world * nchains
 * replicas (replica r starts as slot (g*tper + r / nchains, r % nchains), isingmc_config.replica_offset = g * R).  Swaps inside
 * the block exchange temperature labels; at a block boundary the two ranks exchange the boundary walkers' operator counts
 * (tempering_container.rs:274-302: the swap test needs nothing else) and, when a swap is accepted, the two configurations
 * themselves, so that a rank always holds the configurations of its own temperatures.  Transport: RCCL ncclSend / ncclRecv on
 * device buffers inside one group call per exchange once isingmc_pt_attach_nccl succeeded, otherwise the host-staged
 * `transport` of the layout (one rank: none needed). */
typedef struct isingmc_pt_transport {
    void *ctx;
    /* blocking exchange with rank `peer`: send sbytes from sbuf and receive rbytes into rbuf (host memory); 0 on success */
    int (*sendrecv)(void *ctx, int peer, const void *sbuf, size_t sbytes, void *rbuf, size_t rbytes);
    /* element-wise maximum over all ranks, in place; 0 on success */
    int (*allreduce_max_u32)(void *ctx, uint32_t *buf, size_t count);
} isingmc_pt_transport;
typedef struct isingmc_pt_layout {
    uint32_t struct_size;     /* = sizeof(isingmc_pt_layout) */
    uint32_t ntemps, nchains; /* global */
    uint32_t rank, world;
    const double *betas;      /* [ntemps] */
    uint64_t seed;            /* Philox key of the swap decisions (the same on every rank) */
    const isingmc_pt_transport *transport; /* may be NULL when world == 1 */
} isingmc_pt_layout;
typedef struct isingmc_nccl_id { char internal[128]; } isingmc_nccl_id; /* = ncclUniqueId */
/* TemperingContainer::new + add_qmc_stepper (tempering_container.rs:40-75) for a batch that already holds the replicas.
 * With ISINGMC_CFG_PER_REPLICA_J the couplings belong to the temperature slot (row r of J = slot r of this rank) and swaps
 * use GraphWeights::relative_weight (tempering_traits.rs:126-155; the fields are common to the batch). */
int isingmc_pt_create(isingmc_batch *b, const isingmc_pt_layout *layout);
/* ncclGetUniqueId (rank 0, then broadcast by the launcher) and ncclCommInitRank for this batch's rank / world */
int isingmc_pt_nccl_unique_id(isingmc_nccl_id *out);
int isingmc_pt_attach_nccl(isingmc_batch *b, const isingmc_nccl_id *id);
/* TemperingContainer::tempering_step (tempering_container.rs:121-149).  *nswaps += swaps whose LOWER temperature this rank owns. */
int isingmc_pt_step(isingmc_batch *b, uint64_t *nswaps);
/* Where the swap decisions are taken.  A rank that owns every temperature of a batch with one Hamiltonian (world == 1, no
 * ISINGMC_CFG_PER_REPLICA_J) decides on the device by default: one small kernel per step equalises the cutoffs of each chain, draws the
 * same Philox numbers as the host path and swaps the labels in device memory; nothing is copied to the host unless the caller asks
 * (isingmc_pt_step with nswaps != NULL reads 8 bytes back; isingmc_pt_get_slots / get_state refresh the host mirrors).  on = 0
 * returns to the host path (same decisions: tested).  Multi-rank layouts and different Hamiltonians per temperature decide on the host. */
int isingmc_pt_set_device_decisions(isingmc_batch *b, int on);
int isingmc_pt_get_device_decisions(const isingmc_batch *b, int *on);
/* isingmc_timesteps at the temperatures of the current labels (TemperingContainer::timesteps, tempering_container.rs:100-119); with
 * device-side decisions the per-replica betas are read from device memory, no host array is involved */
int isingmc_pt_timesteps(isingmc_batch *b, uint64_t t, uint32_t sampling_freq, uint32_t flags);
/* current labels of the local replicas: global slot (t * nchains + chain), its beta, and the configuration's identity */
int isingmc_pt_get_slots(isingmc_batch *b, uint32_t *slot_of_replica, double *beta_of_replica, uint32_t *config_id_of_replica);

/* container-level save / load (the reference serialises the whole container, tempering_container.rs:683-792): labels, configuration
 * identities, step counter and swap count; the replicas themselves travel through export_ops / get_state / get_epoch */
int isingmc_pt_get_state(isingmc_batch *b, uint64_t *step, uint64_t *total_swaps);
int isingmc_pt_set_state(isingmc_batch *b, const uint32_t *slot_of_replica, const uint32_t *config_id_of_replica, uint64_t step, uint64_t total_swaps);

/* stream plumbing: use the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
int isingmc_set_stream(isingmc_batch *b, void *hip_stream);
int isingmc_synchronize(isingmc_batch *b);
/* HIP-event timing of the most recent isingmc_timesteps launch(es): total ms and number of kernel launches */
int isingmc_last_kernel_ms(isingmc_batch *b, float *ms, uint32_t *launches);
/* the same run split by kernel: ms[0]/launches[0] = diagonal-pass launches, ms[1]/launches[1] = all other launches */
int isingmc_last_pass_ms(isingmc_batch *b, float ms[2], uint32_t launches[2]);
/* ... and, of ms[1], the part spent in the RVB-sweep launches of split timesteps (0 when none ran) */
int isingmc_last_rvb_ms(isingmc_batch *b, float *ms, uint32_t *launches);
/* diagnostic builds (-DSSE_PHASE_TIMING) only: per-replica phase durations in 10-ns ticks, out[R][16]; zero otherwise */
int isingmc_debug_phase_ticks(isingmc_batch *b, uint64_t *out, int reset);
/* number of sweeps fused into one kernel launch by isingmc_timesteps (0 = all t steps in one launch) */
int isingmc_set_steps_per_launch(isingmc_batch *b, uint64_t steps);
/* build/launch configuration actually in use: out[0]=waves per replica, out[1]=dynamic LDS bytes,
 * out[2]=union-find ids that fit in LDS, out[3]=state words per replica, out[4]=slots per lane,
 * out[5]=1 if the edge table is staged in LDS, out[6]=bit 0: timesteps are issued as two launches (diagonal, rest); bit 1: per-variable
 * tables live in HBM (ISINGMC_CFG_GLOBAL_TABLES path); bit 2: the diagonal-pass launch is the trimmed kernel of sse_fast.hip.h; bit 3: ... and it labels the segments for the cluster
 * update of the same timestep; bit 4: ... or hands it the dense list of occupied slots; bit 5: the most recent cluster launch was the
 * dedicated kernel; bit 6: the most recent RVB sweep ran as growth + main launches (bits 16-23: waves per replica of that main
 * launch); bits 8-15: waves per replica of the most recent off-diagonal launch,
 * out[7]=dynamic LDS bytes of the diagonal-pass launch */
int isingmc_get_launch_info(const isingmc_batch *b, uint32_t out[8]);
/* Host-only: the chunk grid and op-string row stride isingmc_create derives for `capacity` slots and kernels of W (diagonal
 * launches) / up to Wmax (off-diagonal launches) wave64s per replica at K slots per lane: out = {chunk size, chunks, row stride in
 * words, tile = slots of the widest whole-tile access}.  Exposed so that the bounds every kernel relies on (whole-tile loads and
 * stores, the prefetch of an empty chunk range, cached-id rows) can be asserted on a CPU box. */
int isingmc_plan_geometry(uint32_t capacity, uint32_t W, uint32_t K, uint32_t Wmax, uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* ISINGMC_HIP_H */
