// sse_device.hip.h — gfx950 device code of the SSE sweep (workgroup-per-replica design).
//
// One workgroup of W wave64s owns one replica for a whole launch.  The op-string (one u32 per slot,
// include/sse_format.h) streams from HBM in tiles of W*64*K consecutive slots: wave w owns the contiguous
// range [w*64K, (w+1)*64K) of the tile and walks it in K sub-rounds of 64 slots (lane l, sub-round j holds
// slot w*64K + j*64 + l), so K coalesced 256-B loads per wave are in flight per tile.
// Everything the reference keeps in per-node linked lists (src/sse/fast_ops.rs:181-190: prev/next p,
// per-variable prev/next) is recomputed on chip by ORDERED SCANS:
//   * inside a sub-round: wave64 ballot + a serial loop over the (few) writer lanes with v_readlane;
//   * between sub-rounds of a wave: the wave updates its own copy of the per-variable table in LDS;
//   * across the W waves of a tile: W copies of the table; a writer in wave w updates the copies of waves
//     > w before the tile's readers run (one barrier) and the copies of waves < w after they are done
//     (XOR for spin bits, MAX for monotonically increasing segment ids), so copy[w] always equals
//     "the table as of the first slot of wave w in the current tile".
// The live operator count n (the reference reads s.get_n() per slot, qmc_traits/diagonal.rs:126) makes
// the diagonal rule a sequential recurrence n_{p+1} = n_p + d_p(n_p); a tile solves it exactly by
// fixed-point iteration with ballot/popcount prefix sums (unique fixed point = the sequential result).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sse_format.h"

namespace sse {

struct BondRec {       // 16 B, one dwordx4 load (general table, any N / E)
    uint32_t a_info;   // var a | (kind|pref) << 29
    uint32_t c;        // second var or SSE_NO_VAR
    double w;          // weight when satisfied: 2|J|, Gamma, 2|h|
};
#define SSE_INFO_SHIFT 29
#define SSE_VAR_MASK 0x1FFFFFFFu
// compact edge entry (staged in LDS when N <= 32768): a | c << 15 | prefers_aligned << 30
#define SSE_CE_VAR_MASK 0x7FFFu
#define SSE_CE_MAX_VARS 32768u
#define SSE_MAX_CHUNKS 128u

struct DevBatch {
    uint32_t R, N, E, Nb, cap, nwords;
    uint32_t stride;      // words between the op-strings (and segment-id rows) of consecutive replicas: cap rounded up to
                          // a whole number of tiles, so that full-tile loads and stores never leave the row; slots >= cutoff hold 0
    uint32_t *ops;        // [R][cap]
    uint32_t *state;      // [R][nwords] bit v of word v>>5
    uint32_t *n, *ntrans, *cutoff, *err, *aux;  // [R]
    uint64_t *epoch;      // [R]
    uint64_t *acc;        // [acc_rows][8]
    const uint32_t *acc_row; // [R] accumulator row of each replica
    const BondRec *bonds; // [Nb]
    const uint32_t *edges_compact; // [E] or null
    const uint32_t *pm_signs; // [rows][pm_words]: bit e = edge e prefers aligned spins (J < 0), one row per bond-table row: the "+-J"
                          // decode (MODE >= 3) takes a bond's variables from the shared compact edge table and only its sign from here
    uint32_t pm_words;    // words per row of pm_signs = ceil(E / 32)
    const double *edge_w; // [E] 2|J|
    const double *cumw;   // [Nb] heat-bath cumulative weights
    double wtot;
    double wJ, gamma, wh; // uniform 2|J| (if uniformJ), Gamma, 2|h|
    uint32_t uniformJ, hpos, has_long;
    uint32_t *segs;       // [R][cap] segment ids of every slot's two legs (lo | hi << 16), written by the cluster build
    uint32_t *segs2;      // [R][stride] second id of each slot when the ids need 32 bits (HBM union-find); nullptr until such a launch is planned
                          // and read by the apply pass of the LDS union-find path (spends spare HBM bandwidth to
                          // avoid recomputing the ordered scan)
    uint32_t *chunks;     // [R][2*SSE_MAX_CHUNKS]: per chunk of CH slots: occupied count, transverse-op count
    uint32_t CH, nchunks; // chunk size (multiple of 256 slots) and number of chunks covering cap
    uint32_t *uf_scratch; // [R][W*N+cap (+bit arrays)] union-find fallback in HBM
    // hand-over from the trimmed diagonal kernel (sse_fast.hip.h) to the cluster update that follows it in the same timestep:
    // the segment labelling rides on the diagonal pass (segs above is written there), the cluster update only unions
    uint32_t lite;        // 1 = the arrays below exist
    uint32_t *pairs;      // [R][stride]: segment-id pairs (lo | hi << 16) of the two-site ops, appended per wave: wave q of the
                          // diagonal launch owns [q*stride/4, (q+1)*stride/4)
    uint32_t *pcount;     // [R][4] pairs appended by each wave
    uint16_t *lastrank;   // [R][N] 1 + dense index of the last cut on each worldline (0 = none): the wrap-around joins
    uint32_t *touchbits;  // [R][nwords] variables that carry an op
    uint64_t *lite_epoch; // [R] the update counter at which segs / pairs describe the op-string (any other primitive in between
                          // moves the counter on and the cluster update falls back to its own scan)
    // ... or, instead, the dense list of the occupied slots in p order (the cluster update scans n instead of M elements)
    uint32_t *cops, *cpos; // [R][stride] op words / their slots; null = not allocated
    uint64_t *cops_epoch;  // [R] the update counter at which the list describes the op-string
    uint8_t *tbl;         // [R][tbl_stride] per-variable tables in HBM/L2 for models whose tables exceed LDS (MODE 2, see Tab)
    uint32_t tbl_stride;  // bytes per replica: Wmax*N*4 (scan records {rank, marker, touched} / spin bytes of the diagonal pass) + N, rounded up to 16
    uint32_t seed_lo, seed_hi, replica_offset;
    const uint32_t *rid;     // [R] or null: identity of the configuration held by each local replica = the `replica` word of its Philox
                             // counters (null: replica_offset + r).  Parallel tempering moves configurations between ranks at
                             // temperature-block boundaries; their random streams move with them
    const uint32_t *ham_row; // [R] or null: with per-replica couplings, the row of the bond tables a replica runs with (null: r).
                             // Tempering between different Hamiltonians: the row belongs to the temperature slot, not the configuration
    uint32_t lds_ufcap;   // ids that fit the LDS union-find arrays
    uint32_t lds_flipcap; // HBM union-find launches: ids whose flip BITS fit in LDS behind the fixed regions (0 = none): the apply pass
                          // then looks the two flips of every op up in LDS instead of in the parent array in HBM
    uint32_t lds_words;   // dynamic LDS words available to the workgroup
    const double *mats;   // generic interactions (Qmc, qmc_runner.rs:415-680): [Nb][16] weights indexed in | out<<2; NULL = Ising bonds
    uint32_t bond_stride; // 0, or Nb when every replica has its own bond table / cumulative weights (per-replica couplings)
    const double *wtot_r; // [R] per-replica total weight (bond_stride != 0)
    const uint32_t *adj_start, *adj; // [N+1], [2E] bonds_for_var (make_classical_bonds, qmc_ising.rs:421-432)
    // deferred cluster flips (sse_cluster.hip.h -> sse_fast.hip.h): instead of rewriting the op-string, the cluster update leaves one
    // byte per slot — the xor mask of the word's four state bits — and the diagonal pass of the next timestep applies it while it
    // streams the string anyway (the apply pass was bound by its 10 B/slot of memory traffic).  pend[r] = 1: replica r's string
    // in HBM is still the one BEFORE the flips; every other consumer of the strings goes through materialize_kernel first.
    uint8_t *flipb;       // [R][stride] or null
    uint32_t *pend;       // [R]
    uint32_t rvb_growers; // RVB: attempts grown side by side (0 = one at a time on wave 0)
    uint32_t *rvb_prod;   // [R][rvb_prod_cap][SSE_RVB_PROD_STRIDE] growth products of a sweep's attempts (sse_rvb_split.hip.h); null until an RVB sweep is planned
    uint32_t rvb_prod_cap; // attempts per replica that rvb_prod holds
    uint32_t rvb_prod_stride; // words per attempt in rvb_prod
    uint32_t dbg_flags;   // diagnostic builds only
    unsigned long long *dbg; // [R][16] phase durations in 10-ns ticks (diagnostic builds only, -DSSE_PHASE_TIMING)
};

#ifdef SSE_PHASE_TIMING
#define SSE_STAMP(slot) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); B.dbg[(size_t)r * 16 + (slot)] += t_ - dbg_t0; dbg_t0 = t_; } } while (0)
#define SSE_STAMP_INIT unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime()
#else
#define SSE_STAMP(slot) do { } while (0)
#define SSE_STAMP_INIT do { } while (0)
#endif

// which primitives a launch runs per step
#define SSE_DO_DIAG 1u
#define SSE_DO_LOOP 2u
#define SSE_DO_CLUSTER 4u
#define SSE_DO_FREE 8u
#define SSE_DO_GROW 16u
#define SSE_DO_HEATBATH 32u
#define SSE_DO_RVB 64u
#define SSE_DO_COMPACT 256u // trimmed diagonal launch only: write the dense op list for the cluster update of the same timestep
#define SSE_DO_LABEL 128u // trimmed diagonal launch only: label the segments for the cluster update of the same timestep

struct SweepArgs {
    const double *beta; // [R]
    uint64_t nsteps;
    uint64_t step0;     // index of the first step of this launch within the caller's timesteps() call
    uint32_t sampling_freq; // 0 = never sample
    uint32_t domask;
    double prob;
    uint32_t rvb_updates; // RVB attempts per step (0 = (N+1)/2, qmc_ising.rs:711)
    uint32_t *out_u32; // optional per-replica output (n_clusters / loop length / RVB successes) of the LAST step
    uint32_t only_flagged; // 1 = run only the replicas flagged in DevBatch::aux (left over by sse::cluster_kernel) and clear their flags
    uint32_t defer_flips;  // sse::cluster_kernel: leave the flips as one byte per slot (DevBatch::flipb) instead of applying them;
                           // sse::sweep_fast_kernel: apply the pending flip bytes of a replica while loading its string
};

// scalar add that the optimiser may not hoist or merge: the ten round keys are wave-uniform and loop-invariant,
// and hoisted out of the sweep loops they would occupy 20 scalar registers for the whole kernel (they were being
// spilled to vector lanes and read back with v_readlane on every use); one s_add per key and call is cheaper
__device__ __forceinline__ uint32_t philox_bump(uint32_t k, uint32_t w) {
    uint32_t r;
    // (readfirstlane: free when the key already sits in a scalar register; under scalar-register pressure the allocator may hold
    // the uniform key in a vector register, which the "s" constraint alone does not move back)
    const uint32_t ks = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    asm volatile("s_add_u32 %0, %1, %2" : "=s"(r) : "s"(ks), "s"(w) : "scc");
    return r;
}
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // one v_mad_u64_u32 per 32x32->64 product (hi and lo halves together) instead of v_mul_hi + v_mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // three-input xor in one instruction (gfx950 v_bitop3_b32, truth table 0x96)
        uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        if (i < 9) { k0 = philox_bump(k0, 0x9E3779B9u); k1 = philox_bump(k1, 0xBB67AE85u); }
    }
    return make_uint4(c0, c1, c2, c3);
}

struct Rng {
    uint32_t k0, k1, replica, epoch_lo, epoch_hi24;
#ifdef SSE_PHASE_TIMING
    uint32_t dbgx;
#endif
    __device__ __forceinline__ uint4 draw(uint32_t tag, uint32_t index) const {
#ifdef SSE_PHASE_TIMING // diagnostic builds: dbg bit 1 = cheap hash instead of Philox (timing attribution only)
        if (dbgx & 2u) {
            const uint32_t h = (index * 0x9E3779B9u) ^ (epoch_lo * 0x85EBCA6Bu) ^ (replica * 0xC2B2AE35u) ^ tag;
            return make_uint4(h * 0x27D4EB2Fu, h ^ (h >> 15), h * 0x165667B1u, h ^ (h << 13));
        }
#endif
        return philox4x32_10(index, epoch_lo, replica, (tag << 24) | epoch_hi24, k0, k1);
    }
};
__device__ __forceinline__ Rng make_rng(const DevBatch &B, uint32_t r, uint64_t epoch) {
    Rng g;
    g.k0 = B.seed_lo; g.k1 = B.seed_hi; g.replica = B.rid ? B.rid[r] : B.replica_offset + r;
    g.epoch_lo = (uint32_t)epoch; g.epoch_hi24 = (uint32_t)(epoch >> 32) & 0xFFFFFFu;
#ifdef SSE_PHASE_TIMING
    g.dbgx = B.dbg_flags;
#endif
    return g;
}
__device__ __forceinline__ double u01(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }

// decoded bond: variables, kind|pref<<2, weight when satisfied
struct Bd {
    uint32_t a, c, kp;
    double w;
};
__device__ __forceinline__ uint32_t bd_kind(const Bd &b) { return b.kp & SSE_BOND_KIND_MASK; }

// matrix element of the shifted bond operator (reference: src/sse/qmc_ising.rs:863-888); straight-line code
__device__ __forceinline__ double bond_weight(const Bd &b, uint32_t in, uint32_t out) {
    const uint32_t kind = b.kp & SSE_BOND_KIND_MASK, pref = (b.kp >> 2) & 1u;
    const uint32_t aligned = ((in ^ (in >> 1)) & 1u) ^ 1u;
    const uint32_t sat = (kind == SSE_BOND_TWO_SITE) ? (uint32_t)(aligned == pref) : (uint32_t)((in & 1u) == pref);
    const bool ok = (kind == SSE_BOND_TRANSVERSE) | ((in == out) & (sat != 0u));
    return ok ? b.w : 0.0;
}

// matrix element of bond b for any model: Interaction::at (qmc_runner.rs:573-612) when the batch carries weight
// matrices, the closed Ising form otherwise
__device__ __forceinline__ double op_weight(const DevBatch &B, uint32_t b, const Bd &d, uint32_t in, uint32_t out) {
    if (B.mats) return B.mats[(size_t)b * 16u + (in | (out << 2))];
    return bond_weight(d, in, out);
}

// ---------------------------------------------------------------------------------------------
// LDS carve (dynamic shared memory).  All sizes in u32 words.
// Every LDS access below indexes this one array directly so that the compiler always emits ds_*
// instructions (pointers held in structs decay to the generic address space and become flat_* ops,
// which are slower and make s_waitcnt on LDS data also wait for the in-flight HBM prefetches).
extern __shared__ __align__(16) uint32_t lds_raw[];
#define LDSW(off, i) lds_raw[(off) + (i)]
#define LDSI(off, i) (reinterpret_cast<int &>(lds_raw[(off) + (i)]))
#define LDSH(off, i) (reinterpret_cast<uint16_t *>(lds_raw)[2u * (off) + (i)]) // 16-bit element i of the array at word offset off
#define LDSB(off, i) (reinterpret_cast<uint8_t *>(lds_raw)[4u * (off) + (i)])  // 8-bit element i of the array at word offset off
// Tables that OTHER lanes of the same wave write between two reads of one lane need a wavefront-scope fence
// between the writes and the re-reads (the C++ memory model would otherwise let the compiler reuse the first
// value).  It emits no instruction: LDS operations of one wave execute in order.
// Diagnostic builds (-DSSE_PHASE_TIMING) can switch parts of a pass off at run time to attribute time; results are
// then wrong by construction.  Normal builds compile the switches away.
#ifdef SSE_PHASE_TIMING
#define SSE_DBG(B, bit) (((B).dbg_flags & (bit)) != 0u)
#else
#define SSE_DBG(B, bit) false
#endif
#define SSE_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define LDSHV(off, i) LDSH(off, i)
#define LDSWV(off, i) LDSW(off, i)

template <int W>
struct Lds {           // word offsets into lds_raw
    uint32_t o_state;  // [nwords]      p=0 spin state
    uint32_t o_touch;  // [nwords]      variables touched by any op
    uint32_t o_touch8; // [N] u8        the same as bytes while the cluster scan runs (plain byte stores instead of atomics)
    uint32_t o_tot;    // [2][W]        per-wave totals (double buffered by round parity)
    uint32_t o_chg;    // [2][W]
    uint32_t o_misc;   // [16]
    uint32_t o_chn;    // [SSE_MAX_CHUNKS] occupied slots per chunk
    uint32_t o_chtr;   // [SSE_MAX_CHUNKS] transverse ops per chunk
    uint32_t o_edges;  // [E]           compact edge table (CL mode only)
    uint32_t o_signs;  // [pm_words]    this replica's coupling signs (+-J decode only)
    uint32_t o_cur;    // [W][N] u16    per wave: rank+1 (within the wave's range) of the latest cut on each variable
    uint32_t o_cl;     // [W][N] u8     per wave: 1 + rank inside the current sub-round of a cut on the variable (0 = none)
    uint32_t o_frozen; // [ufwords]     bit per id: segment holds a longitudinal op
    uint32_t o_froot;  // [ufwords]     bit per id: root is frozen
    uint32_t o_parent; // [ufcap] u16 (the LDS union-find is only used when every id fits 16 bits)
    // diag_only: the launch runs the diagonal pass (+ directed loop) alone and its per-wave spin BYTES are the only per-variable
    // table (large models whose cluster tables live in HBM can still keep these in LDS)
    __device__ __forceinline__ void carve(uint32_t N, uint32_t nwords, uint32_t ufcap, uint32_t ledges, uint32_t has_long, bool tg = false,
                                          uint32_t pm_words = 0, bool diag_only = false) {
        uint32_t base = 0;
        o_state = base; base += nwords;
        o_touch = base; base += nwords;
        if (tg) N = 0; // MODE 2: the per-variable tables live in HBM (Tab<true>), only the bit arrays stay in LDS
        o_touch8 = base; base += diag_only ? 0u : (N + 3) / 4;
        o_tot = base; base += 2 * W;
        o_chg = base; base += 2 * W;
        o_misc = base; base += 16;
        o_chn = base; base += SSE_MAX_CHUNKS;
        o_chtr = base; base += SSE_MAX_CHUNKS;
        o_edges = base; base += ledges;
        o_signs = base; base += pm_words;
        o_cur = base; base += diag_only ? (W * N + 3) / 4 : (W * N + 1) / 2;
        o_cl = base; base += diag_only ? 0u : (W * N + 3) / 4;
        o_frozen = base; base += has_long ? (ufcap + 31) / 32 : 0u;
        o_froot = base; base += has_long ? (ufcap + 31) / 32 : 0u;
        o_parent = base;
    }
};
// Per-variable tables of the ordered scans (spin bytes of the diagonal pass, cut ranks / cut markers / touched bytes of the
// cluster scan).  TG = false: LDS (ds_* instructions, `reg` = word offset into lds_raw).  TG = true: a per-replica scratch in
// HBM, in practice served by L2 / Infinity Cache (`reg` = byte offset into g) — for models whose tables exceed the 160 KB of
// LDS (N >~ 10^4 variables, BASELINE configs[4] at 32^3).  A wave's own table is only touched by that wave between two
// barriers, and global accesses of one wave are ordered at wavefront scope without waits, so the code is the same in both
// modes; other waves' tables are only changed by atomics separated from their owners' accesses by a workgroup barrier.
template <bool TG>
struct Tab {
    uint8_t *g;
    uint32_t cur, cl, touch8;
    __device__ __forceinline__ uint32_t ld8(uint32_t reg, uint32_t i) const { if constexpr (TG) return g[reg + i]; else return LDSB(reg, i); }
    __device__ __forceinline__ void st8(uint32_t reg, uint32_t i, uint32_t v) const { if constexpr (TG) g[reg + i] = (uint8_t)v; else LDSB(reg, i) = (uint8_t)v; }
    __device__ __forceinline__ uint32_t ld16(uint32_t reg, uint32_t i) const { if constexpr (TG) return reinterpret_cast<const uint16_t *>(g + reg)[i]; else return LDSH(reg, i); }
    __device__ __forceinline__ void st16(uint32_t reg, uint32_t i, uint32_t v) const { if constexpr (TG) reinterpret_cast<uint16_t *>(g + reg)[i] = (uint16_t)v; else LDSH(reg, i) = (uint16_t)v; }
    __device__ __forceinline__ uint32_t ld32(uint32_t reg, uint32_t i) const { if constexpr (TG) return reinterpret_cast<const uint32_t *>(g + reg)[i]; else return LDSW(reg, i); }
    __device__ __forceinline__ void st32(uint32_t reg, uint32_t i, uint32_t v) const { if constexpr (TG) reinterpret_cast<uint32_t *>(g + reg)[i] = v; else LDSW(reg, i) = v; }
    // MODE 2 keeps the scan tables of a (wave, variable) pair in ONE 4-byte record {u16 rank, u8 marker, u8 touched}: a leg's
    // lookups and a cut's stores then hit one 64-B sector of HBM instead of three (that path is bound by random-sector traffic)
    __device__ __forceinline__ uint32_t rec_ld(uint32_t i) const { return reinterpret_cast<const uint32_t *>(g)[i]; }
    __device__ __forceinline__ void rec_st(uint32_t i, uint32_t v) const { reinterpret_cast<uint32_t *>(g)[i] = v; }
    __device__ __forceinline__ void rec_rank_st(uint32_t i, uint32_t v) const { reinterpret_cast<uint16_t *>(g)[2u * i] = (uint16_t)v; }
    __device__ __forceinline__ void rec_mark_st(uint32_t i, uint32_t v) const { g[4u * i + 2u] = (uint8_t)v; }
    __device__ __forceinline__ void rec_touch_st(uint32_t i) const { g[4u * i + 3u] = (uint8_t)1u; }
    __device__ __forceinline__ void xor32(uint32_t reg, uint32_t i, uint32_t bits) const {
        if constexpr (TG) __hip_atomic_fetch_xor(reinterpret_cast<uint32_t *>(g + reg) + i, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else atomicXor(&LDSW(reg, i), bits);
    }
};
template <bool TG, int W>
__device__ __forceinline__ Tab<TG> make_tab(const DevBatch &B, const Lds<W> &L, uint32_t r) {
    Tab<TG> T;
    if constexpr (TG) {
        T.g = B.tbl + (size_t)r * B.tbl_stride;
        T.cur = 0u; T.cl = (uint32_t)W * B.N * 2u; T.touch8 = (uint32_t)W * B.N * 4u; // byte offsets (N is a multiple of 4 in this mode); the cluster scan uses the records (rec_*) at offset 0
    } else {
        T.g = nullptr;
        T.cur = L.o_cur; T.cl = L.o_cl; T.touch8 = L.o_touch8;
    }
    return T;
}
// MODE of a kernel: how bonds are decoded and where the per-variable tables live
// (3 / 4: the "+-J" decode — every replica its own coupling SIGNS on a shared graph with uniform |J| and fields: a bond's variables
// come from the shared compact edge table in global memory (L2-resident), its sign from a per-replica bit array in LDS, its weight
// from three scalars; nothing per replica is fetched from HBM to decode an op.  3 = per-variable tables in LDS (diagonal launches
// only), 4 = in HBM like mode 2)
enum { SSE_MODE_GENERAL = 0, SSE_MODE_LDS_EDGES = 1, SSE_MODE_GLOBAL_TABLES = 2, SSE_MODE_PM_LDS_TABLES = 3, SSE_MODE_PM_GLOBAL_TABLES = 4 };

enum { MISC_NCLUST = 0, MISC_ANYFROZEN = 1, MISC_LOOP_A = 2, MISC_LOOP_B = 3, MISC_LOOP_C = 4, MISC_LOOP_D = 5 };

// The 16-byte bond record of bond b: loaded from this replica's table (general decode) or, for the +-J decode, put together from
// the shared compact edge table, the replica's sign bits in LDS and the uniform weights — everything downstream is the same code.
template <bool PM, int W>
__device__ __forceinline__ uint4 bond_rec(const DevBatch &B, const Lds<W> &L, uint32_t b) {
    if constexpr (PM) {
        const bool two = b < B.E;
        const uint32_t eb = two ? b : 0u;
        const uint32_t e = B.edges_compact[eb];
        const uint32_t sgn = (LDSW(L.o_signs, eb >> 5) >> (eb & 31u)) & 1u;
        const uint32_t s1 = b - B.E;
        const bool tr = s1 < B.N;
        const uint32_t a = two ? (e & SSE_CE_VAR_MASK) : (tr ? s1 : s1 - B.N);
        const uint32_t kp = two ? (SSE_BOND_TWO_SITE | (sgn << 2)) : (tr ? SSE_BOND_TRANSVERSE : (SSE_BOND_LONGITUDINAL | (B.hpos << 2)));
        const double w = two ? B.wJ : (tr ? B.gamma : B.wh);
        return make_uint4(a | (kp << SSE_INFO_SHIFT), two ? ((e >> 15) & SSE_CE_VAR_MASK) : SSE_NO_VAR, (uint32_t)__double2loint(w), (uint32_t)__double2hiint(w));
    } else {
        return *reinterpret_cast<const uint4 *>(B.bonds + b);
    }
}

template <bool CL, int W, bool PM = false>
__device__ __forceinline__ Bd decode_bond(const DevBatch &B, const Lds<W> &L, uint32_t b) {
    Bd d;
    if constexpr (CL) {
        // straight-line: one LDS read with a safe index, then selects
        const bool two = b < B.E;
        const uint32_t e = LDSW(L.o_edges, two ? b : 0u);
        const uint32_t s1 = b - B.E;
        const bool tr = s1 < B.N;
        d.a = two ? (e & SSE_CE_VAR_MASK) : (tr ? s1 : s1 - B.N);
        d.c = two ? ((e >> 15) & SSE_CE_VAR_MASK) : SSE_NO_VAR;
        d.kp = two ? (SSE_BOND_TWO_SITE | (((e >> 30) & 1u) << 2))
                   : (tr ? SSE_BOND_TRANSVERSE : (SSE_BOND_LONGITUDINAL | (B.hpos << 2)));
        // CL mode is only selected for uniform |J|: the three weights are scalars.  Select on their halves held in
        // scalar registers; written as a select of the doubles the compiler turns it into a per-lane LOAD from the
        // kernel-argument segment, whose s_waitcnt then also waits for the op-word prefetch (vmcnt is in order).
        const int jlo = __builtin_amdgcn_readfirstlane(__double2loint(B.wJ)), jhi = __builtin_amdgcn_readfirstlane(__double2hiint(B.wJ));
        const int glo = __builtin_amdgcn_readfirstlane(__double2loint(B.gamma)), ghi = __builtin_amdgcn_readfirstlane(__double2hiint(B.gamma));
        const int hlo = __builtin_amdgcn_readfirstlane(__double2loint(B.wh)), hhi = __builtin_amdgcn_readfirstlane(__double2hiint(B.wh));
        d.w = __hiloint2double(two ? jhi : (tr ? ghi : hhi), two ? jlo : (tr ? glo : hlo));
    } else {
        const uint4 q = bond_rec<PM, W>(B, L, b);
        d.a = q.x & SSE_VAR_MASK; d.c = q.y; d.kp = q.x >> SSE_INFO_SHIFT;
        d.w = __hiloint2double((int)q.w, (int)q.z);
    }
    return d;
}

// wave priority for the issue arbiter (s_setprio takes an immediate): x mod 4, wave-uniform
#ifndef SSE_ROTATE_PRIO
#define SSE_ROTATE_PRIO 2u // tiles of the trimmed diagonal kernel between two priority changes (power of two)
#endif
#ifndef SSE_GEN_ROTATE
#define SSE_GEN_ROTATE 2u  // the same for the tile loops of the general kernels
#endif
__device__ __forceinline__ void sse_set_prio(uint32_t x) {
    switch (x & 3u) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// wave64 ballot straight from the i1 condition (HIP's __ballot(int) goes through a 0/1 integer: v_cndmask + v_cmp_ne)
__device__ __forceinline__ uint64_t sse_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool sse_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }

// flip bit 0 of 8-bit element e of the table at word offset off (other waves' tables: 32-bit atomic on the word)
__device__ __forceinline__ void spin_table_flip(uint32_t off, uint32_t e) { atomicXor(&LDSW(off, e >> 2), 1u << ((e & 3u) * 8u)); }

// A wave-uniform double pinned into vector registers: selects between such values then cost two v_cndmask each,
// instead of copying scalar halves into vector registers at every use.
__device__ __forceinline__ double vgpr_copy(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x), vlo, vhi;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vlo) : "s"(lo));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vhi) : "s"(hi));
    return __hiloint2double(vhi, vlo);
}

// Row accesses as (uniform base pointer) + (32-bit byte offset): the offset is computed in 32 bits (rows are far below
// 2^30 words), which lets the compiler use the scalar-base + vector-offset addressing mode instead of 64-bit vector
// address arithmetic per access.
__device__ __forceinline__ uint32_t row_ld(const uint32_t *row, uint32_t idx) {
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(row) + (size_t)(idx * 4u));
}
__device__ __forceinline__ void row_st(uint32_t *row, uint32_t idx, uint32_t v) {
    *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(row) + (size_t)(idx * 4u)) = v;
}

__device__ __forceinline__ uint32_t vgpr_copy_u32(uint32_t x) {
    uint32_t v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
    return v;
}

// slot index of (tile, wave, sub-round j, lane)
template <int W, int K>
__device__ __forceinline__ uint32_t slot_of(uint32_t tile, int wave, int j, int lane) {
    return tile * (uint32_t)(W * 64 * K) + (uint32_t)(wave * 64 * K + j * 64 + lane);
}

// ---------------------------------------------------------------------------------------------
// Diagonal pass.  Reference: DiagonalUpdater::make_diagonal_update_with_rng_and_state_ref
// (qmc_traits/diagonal.rs:114-135) with metropolis_single_diagonal_update (:142-191), or the heat-bath
// rule (qmc_traits/heatbath.rs:149-209) when HB.
//
// The slot rule depends on the live operator count n, which makes the sweep sequential in p.  One tile of
// W*64*K slots is decided by a fixed-point iteration instead: every candidate slot evaluates its rule with
// n = (count at the tile start) + (net accepted candidates at earlier slots of the tile), starting from "none
// accepted", until no decision changes.  The fixed point is unique and equal to the sequential result (the
// decision of slot p only depends on decisions at slots < p), and it is reached in 2 rounds almost always
// because n moves by a few units inside a tile while the rule compares against M - n ~ 1e4..1e5.
//
// Per slot and round the work is: one int->f64 convert, one f64 multiply, two f64 compares (written straight to
// wave masks), the mask algebra on the scalar unit, and four mbcnt for the prefix counts.  All compares are the
// IEEE f64 expressions of oracle/sse_oracle.c (built with -ffp-contract=off on both sides).
template <int W, int K, bool CL, bool HB, bool TG, bool PM = false>
__device__ __forceinline__ void diagonal_pass(const DevBatch &B, const Lds<W> &L, uint32_t r, const Rng &rng, double beta, uint32_t M,
                              int &n_io, int &ntrans_io, uint32_t &gr) {
    constexpr int NT = W * 64;
    const Tab<TG> T = make_tab<TG, W>(B, L, r);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: keeps per-wave control flow uniform
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    const double beta_nb = beta * (double)B.Nb;
    const double hb_bw = beta * B.wtot;
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    // CL mode (uniform |J|): the three bond weights and beta*Nb times them, as vector-register constants
    double wJv = 0, wGv = 0, wHv = 0, nJv = 0, nGv = 0, nHv = 0;
    if constexpr (CL) {
        wJv = vgpr_copy(B.wJ); wGv = vgpr_copy(B.gamma); wHv = vgpr_copy(B.wh);
        nJv = vgpr_copy(beta_nb * B.wJ); nGv = vgpr_copy(beta_nb * B.gamma); nHv = vgpr_copy(beta_nb * B.wh);
    }
    // a few scalars every sub-round needs, pinned into vector registers: the pass is short of scalar registers (they
    // were being spilled to vector lanes and read back with v_readlane once per use) and has vector registers to spare
    const uint32_t pE = vgpr_copy_u32(B.E), pN = vgpr_copy_u32(B.N), pNb = vgpr_copy_u32(B.Nb), pM = vgpr_copy_u32(M);
    // Per-wave spin tables T_w[v] (u8, in the o_cur area the cluster scan uses later): bit 0 = spin of v at the
    // wave's current position; bits 1..7 = lane+1 of an off-diagonal op on v inside the sub-round being resolved.
    const uint32_t N = B.N, h_my = (uint32_t)wave * N;
    for (uint32_t i = tid; i < (uint32_t)W * N; i += NT) {
        const uint32_t v = i % N;
        T.st8(T.cur, i, (LDSW(L.o_state, v >> 5) >> (v & 31)) & 1u);
    }
    __syncthreads();

    const uint32_t ntiles = (M + NT * K - 1) / (NT * K);
    int n_start = n_io, ntrans = 0;

    // off-diagonal ops ("events") of a tile's words: which of the op's variables flip.  Ising bonds: only single-site ops
    // can be off-diagonal (bit 0 of in^out); generic two-variable interactions (never in CL mode) may flip either.
    auto event_of = [&](uint32_t wd, uint32_t &va, uint32_t &vc, bool &fc) -> bool {
        const uint32_t xb = sse_op_in(wd) ^ sse_op_out(wd);
        const bool fa = (xb & 1u) != 0u;
        fc = CL ? false : ((xb & 2u) != 0u);
        if constexpr (CL) { // only one-variable ops flip a spin here: their variable follows from the bond number alone
            const uint32_t s1 = sse_op_bond(wd) - B.E;
            va = fa ? (s1 < B.N ? s1 : s1 - B.N) : 0u;
            vc = va;
        } else {
            const Bd d = decode_bond<CL, W, PM>(B, L, (fa | fc) ? sse_op_bond(wd) : 0u);
            va = d.a; vc = d.c != SSE_NO_VAR ? d.c : d.a;
        }
        return fa;
    };
    // flip the spin of the event variables in the tables of waves [wlo, whi) (wave-uniform bounds)
    auto propagate = [&](const uint32_t (&var)[K], const bool (&ev)[K], int wlo, int whi) {
        if ((N & 3u) == 0u) { // tables start on word boundaries: word index and bit inside a table do not depend on the wave
            uint32_t widx[K], bit[K];
#pragma unroll
            for (int j = 0; j < K; ++j) { widx[j] = var[j] >> 2; bit[j] = 1u << ((var[j] & 3u) * 8u); }
            for (int w2 = wlo; w2 < whi; ++w2) {
                const uint32_t tbl = (uint32_t)w2 * (N >> 2); // word index of wave w2's table inside the region
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (ev[j]) T.xor32(T.cur, tbl + widx[j], bit[j]);
            }
        } else {
            for (int w2 = wlo; w2 < whi; ++w2) {
                const uint32_t base = (uint32_t)w2 * N;
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (ev[j]) T.xor32(T.cur, (base + var[j]) >> 2, 1u << (((base + var[j]) & 3u) * 8u));
            }
        }
    };

    // Loads and stores of whole tiles are unconditional and branch-free: rows are padded to a whole number of
    // tiles (DevBatch::stride) and slots >= M hold 0, so a partial last tile reads zeros and writes them back.
    // (Predicated loads put every access into its own basic block, and the compiler then waits for ALL
    // outstanding memory operations before the first use — i.e. for the prefetch it has just issued.)
    uint32_t wnext[K];
    // prologue: tile 0 words; their events go to the tables of later waves
#pragma unroll
    for (int j = 0; j < K; ++j) wnext[j] = row_ld(ops, slot_of<W, K>(0, wave, j, lane));
    {
        uint32_t var[K], var2[K]; bool ev[K], ev2[K];
#pragma unroll
        for (int j = 0; j < K; ++j) ev[j] = event_of(wnext[j], var[j], var2[j], ev2[j]);
        propagate(var, ev, wave + 1, W);
        if constexpr (!CL) propagate(var2, ev2, wave + 1, W);
    }
    __syncthreads();

    SSE_STAMP_INIT;
    for (uint32_t tile = 0; tile < ntiles; ++tile) {
#ifdef SSE_GEN_ROTATE
        sse_set_prio(tile / SSE_GEN_ROTATE + blockIdx.x);
#endif
        SSE_STAMP(11);
        uint32_t word[K];
#pragma unroll
        for (int j = 0; j < K; ++j) word[j] = wnext[j];
        {   // prefetch the next tile (after the last one: the same tile again, the values are not used)
            const uint32_t tn = tile + 1 < ntiles ? tile + 1 : tile;
#pragma unroll
            for (int j = 0; j < K; ++j) wnext[j] = row_ld(ops, slot_of<W, K>(tn, wave, j, lane));
        }

        // per slot, kept across the rounds:
        //   fa, fb : f64 operands of the rule (see the rounds below); fa = +inf when the slot is not a candidate
        //   cb     : M (insert candidate) or M + 1 (removal candidate), so that the rule's den is cb - n
        //   cw     : the op word to store when the candidate is accepted (new diagonal op, or 0 for a removal);
        //   keep   : the word to store otherwise (what the slot holds now)
        double fa[K], fb[K];
        uint32_t cb[K], cw[K], keep[K], evA[K], evC[K];
        bool isevj[K], isevc[K];
        uint64_t insm[K]; // insert candidates
        uint32_t trbits = 0; // bit j: the op at stake in sub-round j is a transverse-field op
        uint4 rnd = make_uint4(0, 0, 0, 0);
        // random numbers and bond of slot j (shared by the two loops below)
        auto draw_bond = [&](int j, uint32_t p, uint32_t wd, bool occ, bool is_empty, uint32_t &r0, uint32_t &r1) -> uint32_t {
            uint32_t r2 = 0;
            if (HB) {
                rnd = rng.draw(SSE_TAG_HEATBATH, p);
                r0 = rnd.x; r1 = rnd.y; r2 = rnd.z;
            } else {
                // slots p and p^64 share one Philox call (include/sse_format.h)
                if (K == 1 || (j & 1) == 0) rnd = rng.draw(SSE_TAG_DIAG, p & ~64u);
                const bool hi = (K == 1) ? ((p & 64u) != 0u) : ((j & 1) != 0);
                r0 = hi ? rnd.z : rnd.x; r1 = hi ? rnd.w : rnd.y;
            }
            uint32_t b;
            if (HB) {
                b = 0;
                if (occ) b = sse_op_bond(wd);
                else if (is_empty) {
                    const double c = u01(r2) * B.wtot;
                    uint32_t lo = 0, hi2 = B.Nb;
                    while (lo < hi2) { const uint32_t mid = lo + ((hi2 - lo) >> 1); if (B.cumw[mid] < c) lo = mid + 1; else hi2 = mid; }
                    b = lo < B.Nb ? lo : B.Nb - 1;
                }
            } else {
                b = occ ? sse_op_bond(wd) : __umulhi(r0, pNb);
            }
            return b;
        };
        // General bond table (non-uniform couplings, per-replica couplings, generic interactions): the 16-byte records live
        // in HBM/L2, so all K of a tile are requested before the first one is used (a load inside the sub-round loop would
        // expose its full latency K times per tile).
        uint32_t pre_b[K], pre_r0[K], pre_r1[K];
        uint4 pre_rec[K];
        if constexpr (!CL) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t p = slot_of<W, K>(tile, wave, j, lane);
                const uint32_t wd = word[j];
                pre_b[j] = draw_bond(j, p, wd, wd != 0u, (p < pM) & (wd == 0u), pre_r0[j], pre_r1[j]);
                pre_rec[j] = bond_rec<PM, W>(B, L, pre_b[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t p = slot_of<W, K>(tile, wave, j, lane);
            const uint32_t wd = word[j];
            const bool occ = wd != 0u;
            const uint32_t inb = sse_op_in(wd) & 1u, inc = (sse_op_in(wd) >> 1) & 1u;
            const uint32_t xbits = sse_op_in(wd) ^ sse_op_out(wd);
            const bool flipa = (xbits & 1u) != 0u;                        // the op flips its first variable
            const bool flipc = CL ? false : ((xbits & 2u) != 0u);          // ... its second (generic interactions only)
            const bool isev = flipa | flipc;
            const bool is_empty = (p < pM) & !occ;
            const bool is_diag = occ & !isev;
            uint32_t r0, r1, b;
            if constexpr (CL) b = draw_bond(j, p, wd, occ, is_empty, r0, r1);
            else { b = pre_b[j]; r0 = pre_r0[j]; r1 = pre_r1[j]; }
            // bond -> variables, kind, preferred alignment, weight w and beta*Nb*w
            uint32_t va, vc, pref;
            bool two, tr;
            double wbond, nbond;
            if constexpr (CL) {
                two = b < pE;
                const uint32_t e = LDSW(L.o_edges, two ? b : 0u);
                const uint32_t s1 = b - pE;  // wraps far above N for two-site bonds
                tr = s1 < pN;
                va = two ? (e & SSE_CE_VAR_MASK) : (tr ? s1 : s1 - pN);
                vc = two ? ((e >> 15) & SSE_CE_VAR_MASK) : va;
                pref = two ? ((e >> 30) & 1u) : B.hpos;
                wbond = two ? wJv : (tr ? wGv : wHv);
                nbond = two ? nJv : (tr ? nGv : nHv);
            } else {
                Bd d;
                { const uint4 q = pre_rec[j]; d.a = q.x & SSE_VAR_MASK; d.c = q.y; d.kp = q.x >> SSE_INFO_SHIFT; d.w = __hiloint2double((int)q.w, (int)q.z); }
                two = d.c != SSE_NO_VAR;
                tr = bd_kind(d) == SSE_BOND_TRANSVERSE;
                va = d.a; vc = two ? d.c : va;
                pref = (d.kp >> 2) & 1u;
                wbond = d.w;
                nbond = beta_nb * d.w;
            }
            const bool generic = !CL && B.mats != nullptr; // wave-uniform
            evA[j] = va; isevj[j] = flipa; evC[j] = vc; isevc[j] = flipc;
            trbits |= tr ? (1u << j) : 0u;
            // Spins at this slot = table value, corrected for the off-diagonal ops at EARLIER lanes of this
            // sub-round.  The op word itself carries the spin before (in) and after (out), so the event lanes
            // publish (lane+1, in) in the table, everybody reads, then they store the spin after their op.  Two
            // events on one variable inside a sub-round are rare; a serial loop over the event lanes handles them.
            const uint64_t ev0 = SSE_DBG(B, 16u) ? 0ull : sse_ballot(isev);
            if (ev0) {
                if (flipa) T.st8(T.cur, h_my + va, (((uint32_t)lane + 1u) << 1) | inb);
                if (flipc) T.st8(T.cur, h_my + vc, (((uint32_t)lane + 1u) << 1) | inc);
                SSE_WAVE_FENCE();
            }
            const uint32_t ea = T.ld8(T.cur, h_my + va), ec = T.ld8(T.cur, h_my + vc);
            uint32_t sa = ea & 1u, sc = ec & 1u;
            if (ev0) {
                const uint32_t La = ea >> 1, Lc = ec >> 1;
                const uint64_t dup = sse_ballot((flipa & (La != (uint32_t)lane + 1u)) | (flipc & (Lc != (uint32_t)lane + 1u)));
                if (!dup) {
                    sa ^= (uint32_t)((La - 1u) < (uint32_t)lane); // La == 0: no event on the variable
                    sc ^= (uint32_t)((Lc - 1u) < (uint32_t)lane);
                    SSE_WAVE_FENCE();
                    if (flipa) T.st8(T.cur, h_my + va, inb ^ 1u);
                    if (flipc) T.st8(T.cur, h_my + vc, inc ^ 1u);
                } else {
                    bool seen_a = false, seen_c = false;
                    uint64_t m = ev0;
                    while (m) {
                        const int Ls = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const bool later = lane > Ls;
                        // up to two flipped variables per event lane (different variables of one op: order irrelevant)
                        for (int which = 0; which < (CL ? 1 : 2); ++which) {
                            const uint32_t fL = __builtin_amdgcn_readlane(which ? (uint32_t)flipc : (uint32_t)flipa, Ls);
                            if (!fL) continue; // wave-uniform
                            const uint32_t vL = __builtin_amdgcn_readlane(which ? vc : va, Ls);
                            const uint32_t inL = __builtin_amdgcn_readlane(which ? inc : inb, Ls);
                            if (va == vL) { sa = later ? (inL ^ 1u) : (seen_a ? sa : inL); seen_a = true; }
                            if (vc == vL) { sc = later ? (inL ^ 1u) : (seen_c ? sc : inL); seen_c = true; }
                            if (lane == Ls) T.st8(T.cur, h_my + vL, inL ^ 1u); // in order: the last event wins
                        }
                    }
                    SSE_WAVE_FENCE();
                }
            }
            const uint32_t sub = sa | (two ? (sc << 1) : 0u);
            // Would a diagonal op on this bond have non-zero weight here (qmc_ising.rs:863-888)?  Two-site: the spins'
            // alignment equals the bond's preference; longitudinal: the spin equals the field's; transverse: always.
            // An op already in the string has its bond's weight (it was inserted with non-zero weight and the string
            // is consistent).
            const uint32_t agree = two ? ((sa ^ sc) ^ 1u) : sa;
            bool ok = tr | (agree == pref);
            double w_gen = 0.0;
            if constexpr (!CL) if (generic) {
                // Interaction::at (qmc_runner.rs:573-612): the weight of the op at stake — the diagonal op that would be
                // inserted (state sub) or the diagonal op in the slot (its own bits)
                const uint32_t st = is_empty ? sub : sse_op_in(wd);
                w_gen = B.mats[(size_t)b * 16u + (st | (st << 2))];
                nbond = beta_nb * w_gen;
                ok = true;
            }
            const double uacc = u01(HB ? r0 : r1);
            bool ins;
            if (HB) {
                // insert: u*(den + bW) < bW after the bond was chosen and kept with u1*maxw < w (heatbath.rs:163-193)
                const double w_ins = generic ? w_gen : (ok ? wbond : 0.0);
                ins = is_empty & (u01(r1) * wbond < w_ins);
                fa[j] = (ins | is_diag) ? uacc : inf;
                fb[j] = 0.0;
            } else {
                ins = is_empty & ok & (nbond > 0.0);
                // insert: u*den < num          (fa = u, fb = num)
                // remove: u*num < den          (fa = u*num)
                fa[j] = ins ? uacc : (is_diag ? uacc * nbond : inf);
                fb[j] = nbond;
            }
            insm[j] = sse_ballot(ins);
            cb[j] = pM + (ins ? 0u : 1u);
            cw[j] = ins ? sse_op_make(b, sub, sub) : 0u;
            keep[j] = wd;
        }

        SSE_STAMP(8);
        // ---- fixed point on n ----
        int npref[K];
#pragma unroll
        for (int j = 0; j < K; ++j) npref[j] = n_start;
        uint64_t acc[K], accp[K];
#pragma unroll
        for (int j = 0; j < K; ++j) accp[j] = 0ull;
        int tot_all = 0;
        bool first = true;
        for (;;) {
            int wtot = 0;
            bool changed = first;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const double t = (double)(int)(cb[j] - (uint32_t)npref[j]); // den of the rule
                uint64_t lt_ins, lt_rem;
                if (HB) {
                    const double lhs = fa[j] * (t + hb_bw);
                    lt_ins = sse_ballot(lhs < hb_bw);
                    lt_rem = sse_ballot(lhs < t);
                } else {
                    lt_ins = sse_ballot(fa[j] * t < fb[j]);
                    lt_rem = sse_ballot(fa[j] < t);
                }
                acc[j] = (lt_ins & insm[j]) | (lt_rem & ~insm[j]);
                changed |= acc[j] != accp[j];
                wtot += popc64(acc[j] & insm[j]) - popc64(acc[j] & ~insm[j]);
            }
            const int buf = gr & 1;
            if (lane == 0) { LDSI(L.o_tot, buf * W + wave) = wtot; LDSW(L.o_chg, buf * W + wave) = changed ? 1u : 0u; }
            __syncthreads();
            if (first && !SSE_DBG(B, 8u)) {
                // events of this tile -> tables of earlier waves (all readers of this tile are done);
                // events of the next tile -> tables of later waves (visible after the next barrier)
                propagate(evA, isevj, 0, wave);
                if constexpr (!CL) propagate(evC, isevc, 0, wave);
                if (tile + 1 < ntiles) {
                    uint32_t var[K], var2[K]; bool ev[K], ev2[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) ev[j] = event_of(wnext[j], var[j], var2[j], ev2[j]);
                    propagate(var, ev, wave + 1, W);
                    if constexpr (!CL) propagate(var2, ev2, wave + 1, W);
                }
            }
            // every lane reads the same words: move them to scalar registers so that the loop stays wave-uniform
            // (the compiler cannot see that an LDS value is the same in all lanes)
            int base = 0; tot_all = 0; uint32_t anychg = 0;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) {
                const int t = __builtin_amdgcn_readfirstlane(LDSI(L.o_tot, buf * W + w2));
                if (w2 < wave) base += t;
                tot_all += t;
                anychg |= (uint32_t)__builtin_amdgcn_readfirstlane((int)LDSW(L.o_chg, buf * W + w2));
            }
            gr++;
#ifdef SSE_PHASE_TIMING
            if (threadIdx.x == 0) B.dbg[(size_t)r * 16 + 13] += 1; // rounds
#endif
            if (SSE_DBG(B, 4u)) break;
            if (!first && !anychg) break;
            first = false;
            int run = n_start + base;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint64_t im = acc[j] & insm[j], rm = acc[j] & ~insm[j];
                npref[j] = run + popc64(im & lanemask_lt(lane)) - popc64(rm & lanemask_lt(lane));
                run += popc64(im) - popc64(rm);
                accp[j] = acc[j];
            }
        }
        SSE_STAMP(9);
#ifdef SSE_PHASE_TIMING
        if (threadIdx.x == 0) B.dbg[(size_t)r * 16 + 12] += 1; // tiles
#endif
        // ---- commit ----
        int dn = 0, dtr = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            row_st(ops, slot_of<W, K>(tile, wave, j, lane), ((acc[j] >> lane) & 1ull) ? cw[j] : keep[j]);
            const uint64_t im = acc[j] & insm[j], rm = acc[j] & ~insm[j];
            const uint64_t trm = sse_ballot((trbits >> j) & 1u);
            dn += popc64(im) - popc64(rm);
            dtr += popc64(im & trm) - popc64(rm & trm);
        }
        ntrans += dtr;
        if (lane == 0 && (dtr | dn)) { // a wave's 64*K slots of a tile lie inside one chunk (CH is a multiple of 256 >= 64*K)
            const uint32_t ch = slot_of<W, K>(tile, wave, 0, 0) / B.CH;
            if (dn) atomicAdd(&LDSW(L.o_chn, ch), (uint32_t)dn);
            if (dtr) atomicAdd(&LDSW(L.o_chtr, ch), (uint32_t)dtr);
        }
        n_start += tot_all;
    }
    // per-wave transverse deltas -> block total
    __syncthreads();
    if (lane == 0) LDSI(L.o_tot, wave) = ntrans;
    __syncthreads();
    int dt = 0;
#pragma unroll
    for (int w2 = 0; w2 < W; ++w2) dt += LDSI(L.o_tot, w2);
    __syncthreads();
    ntrans_io += dt;
    n_io = n_start;
}

// ---------------------------------------------------------------------------------------------
// Union-find storage: LDS (fast path) or the per-replica HBM scratch (when N + #cuts exceeds the LDS
// capacity).  The accessor keeps the address space static so that the LDS path compiles to ds_* ops.
template <bool G>
struct UFA {
    uint32_t *gparent, *gfrozen, *gfroot; // HBM arrays (G)
    uint32_t o_parent, o_frozen, o_froot; // lds_raw offsets (!G)
    __device__ __forceinline__ uint32_t get(uint32_t i) const { if constexpr (G) return gparent[i]; else return (uint32_t)LDSH(o_parent, i); }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { if constexpr (G) gparent[i] = v; else LDSH(o_parent, i) = (uint16_t)v; }
    __device__ __forceinline__ uint32_t cas(uint32_t i, uint32_t cmp, uint32_t v) const {
        if constexpr (G) return atomicCAS(&gparent[i], cmp, v);
        else {
            // 16-bit compare-and-swap through a 32-bit CAS on the containing word; a concurrent 16-bit store to
            // the other half only makes the CAS fail and retry with the value it returned
            const uint32_t widx = i >> 1, sh = (i & 1u) * 16u;
            uint32_t old = __hip_atomic_load(&LDSW(o_parent, widx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (;;) {
                const uint32_t cur = (old >> sh) & 0xFFFFu;
                if (cur != cmp) return cur;
                const uint32_t neww = (old & ~(0xFFFFu << sh)) | (v << sh);
                const uint32_t prev = atomicCAS(&LDSW(o_parent, widx), old, neww);
                if (prev == old) return cmp;
                old = prev;
            }
        }
    }
    __device__ __forceinline__ void frozen_or(uint32_t w, uint32_t bits) const { if constexpr (G) atomicOr(&gfrozen[w], bits); else atomicOr(&LDSW(o_frozen, w), bits); }
    __device__ __forceinline__ void froot_or(uint32_t w, uint32_t bits) const { if constexpr (G) atomicOr(&gfroot[w], bits); else atomicOr(&LDSW(o_froot, w), bits); }
    __device__ __forceinline__ uint32_t frozen_get(uint32_t w) const { if constexpr (G) return gfrozen[w]; else return LDSW(o_frozen, w); }
    __device__ __forceinline__ uint32_t froot_get(uint32_t w) const { if constexpr (G) return gfroot[w]; else return LDSW(o_froot, w); }
    __device__ __forceinline__ void bits_clear(uint32_t w) const {
        if constexpr (G) { gfrozen[w] = 0u; gfroot[w] = 0u; } else { LDSW(o_frozen, w) = 0u; LDSW(o_froot, w) = 0u; }
    }
};

// Lock-free union-find with smallest-id roots (canonical cluster labels).
template <bool G>
__device__ __forceinline__ uint32_t uf_find(const UFA<G> &uf, uint32_t x) {
    uint32_t p = uf.get(x);
    while (p != x) {
        const uint32_t g = uf.get(p);
        if (g != p) uf.set(x, g); // path halving; benign race (always an ancestor)
        x = p;
        p = g;
    }
    return x;
}
// Read-only find for the flatten phase: there the owner of id i overwrites parent[i] with the exact root, and a
// path-halving store from another thread's walk could land after it and put a non-root ancestor back.
template <bool G>
__device__ __forceinline__ uint32_t uf_find_ro(const UFA<G> &uf, uint32_t x) {
    uint32_t p = uf.get(x);
    while (p != x) { x = p; p = uf.get(x); }
    return x;
}
template <bool G>
__device__ __forceinline__ void uf_union(const UFA<G> &uf, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(uf, a);
        b = uf_find(uf, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        if (uf.cas(b, b, a) == b) return;
    }
}

// Union on trees that only the calling wave touches (cluster scan): plain stores; lanes that hook the same root in
// one instruction are detected by reading the parent back.
template <bool G>
__device__ __forceinline__ void uf_union_wave(const UFA<G> &uf, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(uf, a);
        b = uf_find(uf, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        uf.set(b, a);
        SSE_WAVE_FENCE();
        if (uf.get(b) == a) return;
    }
}

// Segment scan shared by cluster build and apply.  BARRIER-FREE: wave w owns a contiguous range of chunks of
// the op-string and scans it alone, in p order, with its own copy of the "latest cut per variable" table.
// Segment ids (min-root union-find => canonical label = smallest id of a cluster):
//   [0,N)                 P(0,v): the part of worldline v that contains p=0
//   [N, N+C)              N+k   : the segment opened by the k-th cut in p order (C = number of transverse ops);
//                                 dense ids come from the per-chunk transverse counts kept by the diagonal pass
//   [N+C, N+C+(W-1)N)     P(w,v): "whatever segment v is in when wave w's range begins" — artificial ids, larger
//                                 than every real id so they are never roots; joined to the real segments after
//                                 the scan (cluster_pass).
// COMPACT: the scan reads the dense list of occupied slots that the trimmed diagonal kernel of this timestep wrote (B.cops /
// B.cpos) instead of the padded op-string: the same p-ordered stream without its empty slots (a third of all slots at
// M = 1.5 n), every lane of every row useful.  Ranges are still bounded by chunks of the padded string — their occupied counts
// (o_chn) give the corresponding ranges of the list — and the segment ids still land at the ops' slots (segs[cpos]).
template <int W, int K, bool CL, bool APPLY, bool G, bool TG, bool COMPACT = false, bool PM = false>
__device__ __forceinline__ void cluster_scan(const DevBatch &B, const Lds<W> &L, uint32_t r, uint32_t M, const UFA<G> &uf,
                                             uint32_t C) {
    static_assert(!COMPACT || (!APPLY && !G && !TG), "the dense list feeds the build scan of the LDS union-find path");
    constexpr int NT = W * 64;
    const Tab<TG> T = make_tab<TG, W>(B, L, r);
    constexpr uint32_t TS = 64 * K; // slots per wave-tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: keeps per-wave control flow uniform
    const uint32_t N = B.N;
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    uint32_t *segs_row = B.segs + (size_t)r * B.stride;
    const uint32_t h_mycur = (uint32_t)wave * N; // element offset of this wave's tables inside o_cur / o_cl
    if constexpr (TG) { for (uint32_t i = tid; i < (uint32_t)W * N; i += NT) T.rec_st(i, 0u); }
    else {
        for (uint32_t i = tid; i < ((uint32_t)W * N + 1) / 2; i += NT) T.st32(T.cur, i, 0u);
        for (uint32_t i = tid; i < ((uint32_t)W * N + 3) / 4; i += NT) T.st32(T.cl, i, 0u);
    }
    __syncthreads();
    // this wave's chunk range and the dense id of its first cut
    const uint32_t used = (M + B.CH - 1) / B.CH;
    const uint32_t q = (used + W - 1) / W;
    const uint32_t c0 = min((uint32_t)wave * q, used), c1 = min(c0 + q, used);
    uint32_t cutbase = 0;
    for (uint32_t c = lane; c < c0; c += 64) cutbase += LDSW(L.o_chtr, c);
    for (int off = 32; off > 0; off >>= 1) cutbase += __shfl_xor(cutbase, off);
    cutbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)cutbase);
    uint32_t pbeg = c0 * B.CH, pend = min(c1 * B.CH, M);
    const uint32_t *src = ops;                      // the stream this wave scans: the padded string, or the dense list
    const uint32_t *cpos = nullptr;
    if constexpr (COMPACT) {
        uint32_t nb = 0, ne = 0;                    // ops in front of the range / up to its end
        for (uint32_t c = lane; c < c1; c += 64) { const uint32_t x = LDSW(L.o_chn, c); ne += x; nb += c < c0 ? x : 0u; }
        for (int off = 32; off > 0; off >>= 1) { nb += __shfl_xor(nb, off); ne += __shfl_xor(ne, off); }
        pbeg = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb); pend = (uint32_t)__builtin_amdgcn_readfirstlane((int)ne);
        src = B.cops + (size_t)r * B.stride; cpos = B.cpos + (size_t)r * B.stride;
    }
    const uint32_t my_placeholder_base = wave == 0 ? 0u : N + C + (uint32_t)(wave - 1) * N;
    const uint32_t idbase = N + cutbase - 1u; // id of the cut with local rank+1 == x is idbase + x
    if (lane == 0) LDSW(L.o_chg, wave) = idbase; // read back by cluster_pass when it joins the ranges
    uint32_t nlocal = 0;                      // cuts seen so far in this wave's range
    // branch-free prefetch (see diagonal_pass): ranges are whole tiles except at the end of the string, where the
    // padded row holds zeros; past the range end the last tile is simply read again
    uint32_t wnext[K], posn[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { wnext[j] = row_ld(src, pbeg + j * 64 + lane); if constexpr (COMPACT) posn[j] = row_ld(cpos, pbeg + j * 64 + lane); }
    for (uint32_t p0 = pbeg; p0 < pend; p0 += TS) {
#ifdef SSE_GEN_ROTATE
        sse_set_prio(p0 / TS / SSE_GEN_ROTATE + blockIdx.x);
#endif
        uint32_t word[K], pos[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { word[j] = (p0 + j * 64 + lane < pend) ? wnext[j] : 0u; pos[j] = COMPACT ? posn[j] : p0 + j * 64 + lane; }
        {
            const uint32_t pn0 = p0 + TS < pend ? p0 + TS : p0;
#pragma unroll
            for (int j = 0; j < K; ++j) { wnext[j] = row_ld(src, pn0 + j * 64 + lane); if constexpr (COMPACT) posn[j] = row_ld(cpos, pn0 + j * 64 + lane); }
        }
        uint32_t ua[K], uc[K]; // the tile's unions, issued together after the K sub-rounds (unions commute)
        bool utwo[K];
        uint4 pre_rec[K]; // general bond table: request the tile's K records together (see diagonal_pass)
        if constexpr (!CL) {
#pragma unroll
            for (int j = 0; j < K; ++j) pre_rec[j] = bond_rec<PM, W>(B, L, word[j] ? sse_op_bond(word[j]) : 0u);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            // straight-line, predicated code: every LDS read uses a safe index and is issued unconditionally
            const uint32_t wd = word[j];
            const bool nonempty = wd != 0u;
            Bd d;
            if constexpr (CL) d = decode_bond<CL, W>(B, L, nonempty ? sse_op_bond(wd) : 0u);
            else { const uint4 q = pre_rec[j]; d.a = q.x & SSE_VAR_MASK; d.c = q.y; d.kp = q.x >> SSE_INFO_SHIFT; d.w = 0.0; }
            const uint32_t va = d.a, kind = bd_kind(d);
            const bool two = nonempty & (d.c != SSE_NO_VAR);
            const uint32_t vc = two ? d.c : va;
            const bool iscut = nonempty & (kind == SSE_BOND_TRANSVERSE);
            const uint64_t cutmask = sse_ballot(nonempty) & sse_ballot(kind == SSE_BOND_TRANSVERSE); // = ballot(iscut), from the compare masks
            const uint32_t first = idbase + nlocal + 1u;
            const uint32_t kown = popc64(cutmask & lanemask_lt(lane)); // cuts of this sub-round at earlier lanes
            const uint32_t id_own = first + kown;
            // Ordered resolution inside the sub-round: a leg on variable x belongs to the segment of the latest
            // cut on x at an EARLIER slot.  The cut lanes publish 1 + (their rank inside the sub-round) in o_cl;
            // one round of LDS reads then gives every lane the latest cut before the sub-round (o_cur) and the
            // cut inside it (o_cl), which precedes the lane iff its rank is below the lane's own count of
            // earlier cuts.  Two cuts of one sub-round on the same variable are rare: a serial loop over the cut
            // lanes (ballot + v_readlane) resolves those.
            if (cutmask) {
                if (iscut) { if constexpr (TG) T.rec_mark_st(h_mycur + va, kown + 1u); else T.st8(T.cl, h_mycur + va, kown + 1u); }
                SSE_WAVE_FENCE();
            }
            uint32_t xa, xc, ma, mc;
            if constexpr (TG) {
                const uint32_t ra = T.rec_ld(h_mycur + va), rc = T.rec_ld(h_mycur + vc);
                xa = ra & 0xFFFFu; xc = rc & 0xFFFFu; ma = (ra >> 16) & 0xFFu; mc = (rc >> 16) & 0xFFu;
                if constexpr (!APPLY) { // touched flags: stored once per (wave, variable), not once per leg (every store dirties a sector)
                    if (nonempty & !(ra >> 24)) T.rec_touch_st(h_mycur + va);
                    if (nonempty & !(rc >> 24)) T.rec_touch_st(h_mycur + vc);
                }
            } else {
                xa = T.ld16(T.cur, h_mycur + va); xc = T.ld16(T.cur, h_mycur + vc);
                ma = T.ld8(T.cl, h_mycur + va); mc = T.ld8(T.cl, h_mycur + vc);
            }
            uint32_t seg_a = xa ? idbase + xa : my_placeholder_base + va;
            uint32_t seg_c = xc ? idbase + xc : my_placeholder_base + vc;
            if (cutmask) {
                const uint32_t myrank1 = id_own - idbase; // rank+1 of this lane's cut inside the wave's range
                const uint64_t dup = cutmask & sse_ballot(ma != kown + 1u);
                if (!dup) {
                    seg_a = ((ma - 1u) < kown) ? first + (ma - 1u) : seg_a; // ma == 0: no cut on the variable
                    seg_c = ((mc - 1u) < kown) ? first + (mc - 1u) : seg_c;
                    SSE_WAVE_FENCE();
                    if (iscut) { if constexpr (TG) T.rec_st(h_mycur + va, myrank1 | (1u << 24)); /* rank, marker 0, touched */ else { T.st16(T.cur, h_mycur + va, myrank1); T.st8(T.cl, h_mycur + va, 0u); } }
                } else {
                    bool lastcut = iscut; // no later cut lane of this sub-round is on the same variable
                    uint64_t m = cutmask;
                    uint32_t idL = first;
                    while (m) {
                        const int Ls = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const uint32_t vL = __builtin_amdgcn_readlane(va, Ls);
                        const bool later = lane > Ls, same_a = va == vL;
                        seg_a = (later & same_a) ? idL : seg_a;
                        seg_c = (later & (vc == vL)) ? idL : seg_c;
                        lastcut = lastcut & !((lane < Ls) & same_a);
                        idL++;
                    }
                    SSE_WAVE_FENCE();
                    if (iscut) { if constexpr (TG) T.rec_mark_st(h_mycur + va, 0u); else T.st8(T.cl, h_mycur + va, 0u); }
                    if (iscut & lastcut) { if constexpr (TG) T.rec_rank_st(h_mycur + va, myrank1); else T.st16(T.cur, h_mycur + va, myrank1); } // the last cut wins
                }
            }
            nlocal += popc64(cutmask);
            if (!APPLY) {
                if (iscut) uf.set(id_own, id_own);
                if constexpr (!TG) { if (nonempty & !SSE_DBG(B, 8u)) { T.st8(T.touch8, va, 1u); T.st8(T.touch8, vc, 1u); } } // (MODE 2: with the record lookups; diagnostic builds: bit 3 = time the scan without them)
                ua[j] = seg_a; uc[j] = seg_c;
                utwo[j] = two & !SSE_DBG(B, 1u); // diagnostic builds: bit 0 = time the scan without unions
                if (B.has_long) if (nonempty & (kind == SSE_BOND_LONGITUDINAL)) uf.frozen_or(seg_a >> 5, 1u << (seg_a & 31));
                if constexpr (!G) { // ids fit 16 bits on this path: remember them for the apply pass
                    const uint32_t hi = iscut ? id_own : (two ? seg_c : seg_a);
                    if constexpr (COMPACT) { if (nonempty) row_st(segs_row, pos[j], seg_a | (hi << 16)); } // (lanes past the range end hold no slot)
                    else row_st(segs_row, pos[j], seg_a | (hi << 16));
                } else if (B.segs2) { // 32-bit ids: two words per slot, so that the apply pass need not repeat the ordered scan
                    const uint32_t hi = iscut ? id_own : (two ? seg_c : seg_a);
                    row_st(segs_row, pos[j], seg_a);
                    row_st(B.segs2 + (size_t)r * B.stride, pos[j], hi);
                }
            } else {
                const uint32_t fa = uf.get(seg_a), fc = uf.get(seg_c), fo = uf.get(iscut ? id_own : seg_a);
                const uint32_t f2 = two ? (fc << 1) : 0u;
                const uint32_t in = sse_op_in(wd) ^ (fa | f2), out = sse_op_out(wd) ^ (fo | f2);
                const uint32_t neww = (wd & ~0xFu) | in | (out << SSE_OP_OUT_SHIFT);
                if (nonempty & (neww != wd)) ops[p0 + j * 64 + lane] = neww;
            }
        }
        if constexpr (!APPLY) {
            {
                // During the scan every wave only touches ids of its own range (its cuts and its placeholders), so
                // its trees are private until the ranges are joined: no atomics are needed, lanes of the wave that
                // hook the same root in one store instruction are sorted out by reading the parent back.  The K
                // unions of the tile go together, three overlapped rounds of table accesses for all of them: parents,
                // grandparents (root test + halving), read-back of the links.  All reads of a batch precede all its
                // stores, so every lane decides on the same snapshot; a link that another lane overwrote (same
                // root hooked twice) or a chain deeper than two falls back to the serial routine.  The same code serves
                // the parents in LDS and the 32-bit parents in HBM (one wave's accesses are ordered, as for the tables).
                uint32_t pa[K], pc[K], ga[K], gc[K], hi[K], lo[K];
                bool link[K], slow[K];
#pragma unroll
                for (int j = 0; j < K; ++j) { pa[j] = uf.get(utwo[j] ? ua[j] : 0u); pc[j] = uf.get(utwo[j] ? uc[j] : 0u); }
#pragma unroll
                for (int j = 0; j < K; ++j) { ga[j] = uf.get(pa[j]); gc[j] = uf.get(pc[j]); }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const bool differ = utwo[j] & (pa[j] != pc[j]); // same parent: one set already
                    const bool roots = (ga[j] == pa[j]) & (gc[j] == pc[j]);
                    link[j] = differ & roots;
                    slow[j] = differ & !roots;
                    lo[j] = pa[j] < pc[j] ? pa[j] : pc[j];
                    hi[j] = pa[j] < pc[j] ? pc[j] : pa[j];
                    // halving: an endpoint whose parent is not a root moves up (it is not a root itself then)
                    if (utwo[j] & (ga[j] != pa[j])) uf.set(ua[j], ga[j]);
                    if (utwo[j] & (gc[j] != pc[j])) uf.set(uc[j], gc[j]);
                    if (link[j]) uf.set(hi[j], lo[j]);
                }
                SSE_WAVE_FENCE();
                uint32_t chk[K];
#pragma unroll
                for (int j = 0; j < K; ++j) chk[j] = uf.get(link[j] ? hi[j] : 0u);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const bool redo = slow[j] | (link[j] & (chk[j] != lo[j]));
                    if (sse_any(redo)) { if (redo) uf_union_wave(uf, pa[j], pc[j]); }
                }
            }
        }
    }
    __syncthreads();
}

// Apply pass of the LDS union-find path (cluster.rs:139-167): every slot's segment ids were stored by the build
// scan, flip bits sit in the (flattened) parent table, so the slots can be rewritten in any order: plain strided
// streaming, no ordered scan.  Input bits flip with the incoming segment, output bits with the outgoing one.
template <int W, int K, bool CL, bool G = false, bool PM = false, bool LF = false>
__device__ __forceinline__ void cluster_apply_cached(const DevBatch &B, const Lds<W> &L, uint32_t r, uint32_t M, const UFA<G> &uf) {
    static_assert(!LF || G, "flip bits in LDS belong to the HBM union-find path");
    auto flip_of = [&](uint32_t id) -> uint32_t { // the flip of id: a bit in LDS (LF) or the (flattened, coin-overwritten) parent entry
        if constexpr (LF) return (LDSW(L.o_parent, id >> 5) >> (id & 31u)) & 1u; else return uf.get(id);
    };
    constexpr int NT = W * 64;
    const int tid = threadIdx.x;
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    const uint32_t *segs = B.segs + (size_t)r * B.stride;
    const uint32_t *segs2 = G ? B.segs2 + (size_t)r * B.stride : segs; // 32-bit ids: the second id of a slot has its own row
    // software-pipelined stream over whole tiles (branch-free loads and stores, see diagonal_pass): slots >= M are
    // empty and are written back unchanged
    constexpr uint32_t TS = (uint32_t)(K * NT);
    uint32_t wn[K], sn[K], tn[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { wn[j] = row_ld(ops, (uint32_t)(j * NT + tid)); sn[j] = row_ld(segs, (uint32_t)(j * NT + tid)); tn[j] = G ? row_ld(segs2, (uint32_t)(j * NT + tid)) : 0u; }
    for (uint32_t p0 = 0; p0 < M; p0 += TS) {
#ifdef SSE_GEN_ROTATE
        sse_set_prio(p0 / TS / SSE_GEN_ROTATE + blockIdx.x);
#endif
        uint32_t wd[K], sg[K], sh[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { wd[j] = wn[j]; sg[j] = sn[j]; sh[j] = tn[j]; }
        {
            const uint32_t pn0 = p0 + TS < M ? p0 + TS : p0;
#pragma unroll
            for (int j = 0; j < K; ++j) { wn[j] = row_ld(ops, pn0 + (uint32_t)(j * NT + tid)); sn[j] = row_ld(segs, pn0 + (uint32_t)(j * NT + tid)); if constexpr (G) tn[j] = row_ld(segs2, pn0 + (uint32_t)(j * NT + tid)); }
        }
        uint32_t second[K]; // general bond table: the second variable of the tile's K bonds, requested together
        if constexpr (!CL && !PM) {
#pragma unroll
            for (int j = 0; j < K; ++j) second[j] = reinterpret_cast<const uint32_t *>(B.bonds + (wd[j] ? sse_op_bond(wd[j]) : 0u))[1];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t w = wd[j];
            const bool nonempty = w != 0u;
            // segment ids of empty slots are stale: read a safe index
            const uint32_t fa = flip_of(nonempty ? (G ? sg[j] : (sg[j] & 0xFFFFu)) : 0u), fb = flip_of(nonempty ? (G ? sh[j] : (sg[j] >> 16)) : 0u);
            bool two;
            if constexpr (CL || PM) two = nonempty & (sse_op_bond(w) < B.E);
            else two = nonempty & (second[j] != SSE_NO_VAR);
            // two-site: both legs of variable a carry fa, both legs of variable c carry fb;
            // single-site: the input leg carries fa (incoming segment), the output leg fb (outgoing segment)
            const uint32_t in = sse_op_in(w) ^ (two ? (fa | (fb << 1)) : fa);
            const uint32_t out = sse_op_out(w) ^ (two ? (fa | (fb << 1)) : fb);
            const uint32_t neww = (w & ~0xFu) | in | (out << SSE_OP_OUT_SHIFT);
            row_st(ops, p0 + (uint32_t)(j * NT + tid), nonempty ? neww : 0u);
        }
    }
    __syncthreads();
}

// Cluster update.  Reference: ClusterUpdater::flip_each_cluster_rng (qmc_traits/cluster.rs:36-172) with the
// longitudinal weight function of qmc_ising.rs:759-775.  Returns the number of clusters.
template <int W, int K, bool CL, bool UF_GLOBAL, bool TG, bool LITE = false, bool COMPACT = false, bool PM = false>
__device__ __forceinline__ uint32_t cluster_pass(const DevBatch &B, const Lds<W> &L, uint32_t r, const Rng &rng, double prob,
                                                 uint32_t M, int n, int ntrans, uint32_t &gr, uint32_t &err) {
    static_assert(UF_GLOBAL || !TG, "tables in HBM imply the HBM union-find");
    static_assert(!LITE || (!UF_GLOBAL && !TG && CL), "the hand-over from the trimmed diagonal kernel uses the LDS union-find");
    constexpr int NT = W * 64;
    const Tab<TG> T = make_tab<TG, W>(B, L, r);
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t N = B.N, nwords = B.nwords;
    UFA<UF_GLOBAL> uf;
    {
        const size_t ids = (size_t)W * N + B.cap;
        uf.gparent = B.uf_scratch + (size_t)r * (ids + 2 * ((ids + 31) / 32));
        uf.gfrozen = uf.gparent + ids;
        uf.gfroot = uf.gfrozen + (ids + 31) / 32;
        uf.o_parent = L.o_parent; uf.o_frozen = L.o_frozen; uf.o_froot = L.o_froot;
    }
    for (uint32_t i = tid; i < nwords; i += NT) LDSW(L.o_touch, i) = 0u;
    for (uint32_t i = tid; i < (N + 3) / 4; i += NT) T.st32(T.touch8, i, 0u);
    if (tid == 0) { LDSW(L.o_misc, MISC_NCLUST) = 0u; LDSW(L.o_misc, MISC_ANYFROZEN) = 0u; }
    if (n == 0) { __syncthreads(); return 0u; } // cluster.rs:46-48
    SSE_STAMP_INIT;
    if constexpr (!LITE)
    { // the scan stores 16-bit cut ranks per wave range: every range must hold fewer than 65535 cuts
        const uint32_t used = (M + B.CH - 1) / B.CH, q = (used + W - 1) / W;
        uint32_t bad = 0;
        for (uint32_t w2 = 0; w2 < (uint32_t)W; ++w2) {
            uint32_t cuts = 0;
            for (uint32_t c = min(w2 * q, used); c < min(w2 * q + q, used); ++c) cuts += LDSW(L.o_chtr, c);
            bad |= (cuts >= 65535u);
        }
        if (bad) { err = 8u; return 0u; }
    }
    const uint32_t C = (uint32_t)ntrans;            // one id per cut (transverse op)
    const uint32_t S = LITE ? N + C : N + C + (uint32_t)(W - 1) * N; // + artificial range-boundary placeholders
    if constexpr (LITE) {
        // The diagonal pass of this timestep has already labelled every leg (B.segs) and listed the segment pairs that the
        // two-site ops join (B.pairs): what is left of the build is the union-find itself.  Ids: [0,N) initial segments,
        // N + k the segment opened by the k-th cut — no range placeholders, the labelling was done in one p-ordered stream.
        for (uint32_t i = tid; i < S; i += NT) uf.set(i, i);
        for (uint32_t i = tid; i < nwords; i += NT) LDSW(L.o_touch, i) = B.touchbits[(size_t)r * nwords + i];
        __syncthreads();
        const uint32_t *pairs = B.pairs + (size_t)r * B.stride;
        const uint32_t reg = B.stride / 4u;
        for (uint32_t q = 0; q < 4u; ++q) {
            const uint32_t cnt = B.pcount[(size_t)r * 4 + q];
            const uint32_t *pq = pairs + (size_t)q * reg;
            for (uint32_t i0 = 0; i0 < cnt; i0 += 4 * NT) { // four independent loads in flight per thread
                uint32_t pr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + (uint32_t)(u * NT + tid); pr[u] = i < cnt ? pq[i] : 0u; }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (pr[u]) uf_union(uf, pr[u] & 0xFFFFu, pr[u] >> 16); // (a pair never has two equal ids, so 0 = none)
            }
        }
        // worldlines are cyclic in imaginary time (cluster.rs:223-242): the segment behind the last cut on v is the one v starts in
        const uint16_t *lastrank = B.lastrank + (size_t)r * N;
        for (uint32_t v = tid; v < N; v += NT) { const uint32_t lr = lastrank[v]; if (lr) uf_union(uf, N + lr - 1u, v); }
        __syncthreads();
    } else {
    for (uint32_t i = tid; i < N; i += NT) uf.set(i, i);
    for (uint32_t i = tid; i < (uint32_t)(W - 1) * N; i += NT) uf.set(N + C + i, N + C + i);
    if (B.has_long) for (uint32_t i = tid; i < (S + 31) / 32; i += NT) uf.bits_clear(i);
    __syncthreads();
    // ---- build: label legs with segment ids, union through non-boundary ops ----
    SSE_STAMP(0);
    cluster_scan<W, K, CL, false, UF_GLOBAL, TG, COMPACT, PM>(B, L, r, M, uf, C);
    // touched bytes -> bits (read by the coins, the p=0 state update and the free-spin pass, all behind later barriers)
    for (uint32_t i = tid; i < nwords; i += NT) {
        uint32_t bits = 0;
        if constexpr (TG) { // the touched byte of every wave's record of the variable
            for (uint32_t k = 0; k < 32 && i * 32 + k < N; ++k) {
                uint32_t t = 0;
                for (uint32_t w2 = 0; w2 < (uint32_t)W; ++w2) t |= T.rec_ld(w2 * N + i * 32 + k) >> 24;
                bits |= (t & 1u) << k;
            }
        } else
        for (uint32_t k = 0; k < 8 && (i * 8 + k) < (N + 3) / 4; ++k) {
            const uint32_t w = T.ld32(T.touch8, i * 8 + k);
            bits |= ((w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u)) << (4 * k);
        }
        LDSW(L.o_touch, i) = bits;
    }
    SSE_STAMP(1);
    // join the ranges: the segment v is in when wave w's range ends continues into P(w+1,v); the last
    // range wraps around into P(0,v) (cluster.rs:223-242: worldlines are cyclic in imaginary time)
    for (uint32_t i = tid; i < (uint32_t)W * N; i += NT) {
        const uint32_t w2 = i / N, v = i - w2 * N;
        const uint32_t x = TG ? (T.rec_ld(i) & 0xFFFFu) : T.ld16(T.cur, i);
        const uint32_t last = x ? LDSW(L.o_chg, w2) + x : 0u;
        const uint32_t seg_end = last ? last : (w2 == 0 ? v : N + C + (w2 - 1) * N + v);
        const uint32_t nxt = (w2 + 1 == (uint32_t)W) ? v : N + C + w2 * N + v;
        if (seg_end != nxt) uf_union(uf, seg_end, nxt);
    }
    __syncthreads();
    } // !LITE
    SSE_STAMP(2);
    // ---- flatten: parent[i] := exact root (no union runs any more), frozen marks move to roots ----
    for (uint32_t i = tid; i < S; i += NT) {
        const uint32_t root = uf_find_ro(uf, i); // no halving stores here: only exact roots may be written
        uf.set(i, root);
        if (B.has_long && ((uf.frozen_get(i >> 5) >> (i & 31)) & 1u)) { uf.froot_or(root >> 5, 1u << (root & 31)); LDSW(L.o_misc, MISC_ANYFROZEN) = 1u; }
    }
    __syncthreads();
    SSE_STAMP(3);
    // ---- coins: one Philox draw per ROOT (= per cluster), then parent[i] := flip bit of its root, in place ----
    // LDS path: the roots are compacted into a list (wave ballot + one LDS counter) so that the draws run on dense
    // lanes; the list lives in the cut-marker tables and the flip bits in the cut-rank tables, both free after the
    // join.  Too many roots or ids for those tables (or no cut at all, or the HBM union-find): every id draws the
    // coin of its root itself — same results, more Philox calls.
    uint32_t myclusters = 0;
    const bool nocuts = (C == 0u);
    const uint32_t anyfrozen = LDSW(L.o_misc, MISC_ANYFROZEN);
    bool dense_done = false;
    // HBM union-find: the flip bit of every id also goes into LDS when the launch has room for S bits behind its fixed regions
    // (the region where the LDS union-find keeps its parents, unused on this path); the apply pass then needs no HBM lookups
    const bool lds_flips = UF_GLOBAL && B.segs2 != nullptr && S <= B.lds_flipcap; // (uniform)
    if (lds_flips) {
        for (uint32_t i = tid; i < (S + 31u) / 32u; i += NT) LDSW(L.o_parent, i) = 0u;
        __syncthreads();
    }
    if constexpr (!UF_GLOBAL) {
        const uint32_t list_cap = ((uint32_t)W * N + 3u) / 4u * 2u;  // u16 entries in the o_cl words
        const uint32_t bits_cap = ((uint32_t)W * N + 1u) / 2u * 32u; // bits in the o_cur words
        if (!nocuts && S <= bits_cap) {
            for (uint32_t i = tid; i < (S + 31u) / 32u; i += NT) LDSW(L.o_cur, i) = 0u;
            if (tid == 0) LDSW(L.o_misc, MISC_LOOP_A) = 0u;
            __syncthreads();
            for (uint32_t i0 = 0; i0 < S; i0 += NT) { // whole waves iterate together (ballot below)
                const uint32_t i = i0 + tid;
                const bool inr = i < S;
                const uint32_t root = inr ? uf.get(i) : 0xFFFFFFFFu;
                const bool isroot = inr & (root == i);
                if (isroot & (i < N + C)) {
                    const bool touched = (i >= N) || ((LDSW(L.o_touch, i >> 5) >> (i & 31)) & 1u);
                    if (touched) myclusters++;
                }
                const uint64_t m = sse_ballot(isroot);
                if (m) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&LDSW(L.o_misc, MISC_LOOP_A), (uint32_t)popc64(m));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    const uint32_t pos = base + popc64(m & lanemask_lt(lane));
                    if (isroot && pos < list_cap) LDSH(L.o_cl, pos) = (uint16_t)i;
                }
            }
            __syncthreads();
            const uint32_t nroots = LDSW(L.o_misc, MISC_LOOP_A);
            if (nroots <= list_cap) { // uniform: every thread read the same counter
                for (uint32_t k = tid; k < nroots; k += NT) {
                    const uint32_t root = LDSH(L.o_cl, k);
                    const uint4 o = rng.draw(SSE_TAG_CLUSTER, root);
                    const uint32_t isfrozen = B.has_long ? (uf.froot_get(root >> 5) >> (root & 31)) & 1u : 0u;
                    if (!isfrozen && u01(o.x) < prob) atomicOr(&LDSW(L.o_cur, root >> 5), 1u << (root & 31));
                }
                __syncthreads();
                for (uint32_t i = tid; i < S; i += NT) {
                    const uint32_t root = uf.get(i);
                    uf.set(i, (LDSW(L.o_cur, root >> 5) >> (root & 31)) & 1u);
                }
                dense_done = true;
            } else myclusters = 0; // counted again below
        }
    }
    if (!dense_done)
    for (uint32_t i = tid; i < S; i += NT) {
        const uint32_t root = uf.get(i);
        const uint32_t vi = i < N ? i : (i - N - C) % N; // variable of a placeholder id (nocuts: every id is one)
        const bool touched = (i >= N && !nocuts) || ((LDSW(L.o_touch, vi >> 5) >> (vi & 31)) & 1u);
        uint32_t f;
        if (nocuts) {
            // no cluster boundary anywhere: the whole graph is one cluster (cluster.rs:98-107), label 0
            const uint4 o = rng.draw(SSE_TAG_CLUSTER, 0u);
            f = (touched && !anyfrozen && u01(o.x) < prob) ? 1u : 0u;
        } else {
            if (root == i && touched && i < N + C) myclusters++;
            const uint4 o = rng.draw(SSE_TAG_CLUSTER, root);
            const uint32_t isfrozen = B.has_long ? (uf.froot_get(root >> 5) >> (root & 31)) & 1u : 0u;
            f = (!isfrozen && u01(o.x) < prob) ? 1u : 0u;
        }
        uf.set(i, f);
        if (lds_flips && f) atomicOr(&LDSW(L.o_parent, i >> 5), 1u << (i & 31u));
    }
    {
        uint32_t c = myclusters;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if (lane == 0 && c) atomicAdd(&LDSW(L.o_misc, MISC_NCLUST), c);
    }
    __syncthreads();
    SSE_STAMP(4);
    // ---- apply (cluster.rs:139-167) ----
    if constexpr (UF_GLOBAL) {
        if (lds_flips) cluster_apply_cached<W, K, CL, true, PM, true>(B, L, r, M, uf);
        else if (B.segs2) cluster_apply_cached<W, K, CL, true, PM>(B, L, r, M, uf); // both ids of every slot were stored by the build scan
        else cluster_scan<W, K, CL, true, UF_GLOBAL, TG, false, PM>(B, L, r, M, uf, C);  // (a replica that outgrew the LDS union-find before the host planned for it)
    } else cluster_apply_cached<W, K, CL, false, PM>(B, L, r, M, uf);
    SSE_STAMP(5);
    // p=0 state follows the placeholder segment of each touched variable
    for (uint32_t i = tid; i < nwords; i += NT) {
        uint32_t x = 0;
        const uint32_t t = LDSW(L.o_touch, i);
        for (uint32_t j = 0; j < 32 && i * 32 + j < N; ++j) x |= (uf.get(i * 32 + j) & 1u) << j;
        LDSW(L.o_state, i) ^= (x & t);
    }
    __syncthreads();
    return nocuts ? 1u : LDSW(L.o_misc, MISC_NCLUST);
}

// touched-variable scan for launches that flip free spins without a preceding cluster pass
template <int W, bool CL, bool PM = false>
__device__ __forceinline__ void touch_scan(const DevBatch &B, const Lds<W> &L, uint32_t r, uint32_t M) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x;
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    for (uint32_t i = tid; i < B.nwords; i += NT) LDSW(L.o_touch, i) = 0u;
    __syncthreads();
    for (uint32_t p0 = 0; p0 < M; p0 += 4 * NT) {
        uint32_t wd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t p = p0 + j * NT + tid; wd[j] = p < M ? ops[p] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!wd[j]) continue;
            const Bd d = decode_bond<CL, W, PM>(B, L, sse_op_bond(wd[j]));
            atomicOr(&LDSW(L.o_touch, d.a >> 5), 1u << (d.a & 31));
            if (d.c != SSE_NO_VAR) atomicOr(&LDSW(L.o_touch, d.c >> 5), 1u << (d.c & 31));
        }
    }
    __syncthreads();
}

// qmc_ising.rs:780-784 / qmc_runner.rs:241-255
template <int W>
__device__ __forceinline__ void free_spin_pass(const DevBatch &B, const Lds<W> &L, const Rng &rng) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < B.nwords; i += NT) {
        const uint32_t t = LDSW(L.o_touch, i);
        uint32_t s = LDSW(L.o_state, i);
        for (uint32_t j = 0; j < 32 && i * 32 + j < B.N; ++j)
            if (!((t >> j) & 1u)) {
                const uint4 o = rng.draw(SSE_TAG_FREE, i * 32 + j);
                s = (s & ~(1u << j)) | ((o.x >> 31) << j);
            }
        LDSW(L.o_state, i) = s;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Directed loop.  Reference: LoopUpdater::make_loop_update_with_rng (qmc_traits/directed_loop.rs:103-171)
// and loop_body (:217-301).  One loop per call; the walk itself is sequential (thread 0), the two things the
// reference does with linked lists are done cooperatively by the whole workgroup:
//   get_nth_p (:76-87, an O(n) list walk)        -> tile-wise ballot/popcount rank search
//   get_next/previous_p_for_rel_var (:51-54)     -> tile-wise search along the worldline direction
// Returns the number of vertices visited.
template <int W, bool CL, bool PM = false>
__device__ __forceinline__ uint32_t loop_pass(const DevBatch &B, const Lds<W> &L, uint32_t r, const Rng &rng, uint32_t M, int n, uint32_t &gr,
                              uint32_t &err) {
    constexpr int NT = W * 64;
    constexpr int U = 4; // independent loads in flight per thread during searches
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: keeps per-wave control flow uniform
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    if (n == 0) return 0u;
    const uint4 o0 = rng.draw(SSE_TAG_LOOP, 0u);
    const uint32_t nth = __umulhi(o0.x, (uint32_t)n);
    // ---- start vertex: the nth occupied slot in p order.  The per-chunk occupancy kept by the diagonal pass
    // names the chunk; only that chunk is scanned (sub-tiles of U*NT slots, wave-major inside) ----
    uint32_t cbase = 0, qbeg = 0, qend = M;
    {
        const uint32_t used = (M + B.CH - 1) / B.CH;
        for (uint32_t c = 0; c < used; ++c) {
            const uint32_t cn = LDSW(L.o_chn, c);
            if (nth < cbase + cn) { qbeg = c * B.CH; qend = min(qbeg + B.CH, M); break; }
            cbase += cn;
        }
    }
    if (tid == 0) LDSW(L.o_misc, MISC_LOOP_A) = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t q0 = qbeg; q0 < qend; q0 += U * NT) {
        uint32_t wd[U];
        uint64_t occ[U];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < U; ++j) { const uint32_t p = q0 + (uint32_t)(wave * 64 * U + j * 64 + lane); wd[j] = p < qend ? ops[p] : 0u; }
#pragma unroll
        for (int j = 0; j < U; ++j) { occ[j] = sse_ballot(wd[j] != 0u); cnt += popc64(occ[j]); }
        const int buf = gr & 1;
        if (lane == 0) LDSI(L.o_tot, buf * W + wave) = cnt;
        __syncthreads();
        gr++;
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < W; ++w2) { const uint32_t t = (uint32_t)LDSI(L.o_tot, buf * W + w2); if (w2 < wave) wbase += t; total += t; }
        uint32_t run = cbase + wbase;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t myrank = run + popc64(occ[j] & lanemask_lt(lane));
            if (wd[j] != 0u && myrank == nth) LDSW(L.o_misc, MISC_LOOP_A) = q0 + (uint32_t)(wave * 64 * U + j * 64 + lane);
            run += popc64(occ[j]);
        }
        cbase += total;
        if (cbase > nth) break;
    }
    __syncthreads();
    const uint32_t p0 = LDSW(L.o_misc, MISC_LOOP_A);
    if (p0 == 0xFFFFFFFFu) { err = 2u; return 0u; } // n inconsistent with the op-string
    // the word at the current vertex travels with the walk: the search below hands over the word it found together with the
    // distance (one 64-bit LDS minimum), so a vertex costs one round trip to the op-string instead of three
    const uint32_t o64 = (L.o_misc + 7u) & ~1u; // an 8-byte aligned pair inside o_misc[6, 9)
    uint32_t cur_word = ops[p0];
    uint32_t rel0, side0;
    {
        const Bd d0 = decode_bond<CL, W, PM>(B, L, sse_op_bond(cur_word));
        const uint32_t k0 = d0.c != SSE_NO_VAR ? 2u : 1u;
        rel0 = __umulhi(o0.y, k0);
        side0 = (o0.z >> 31) ? 0u : 1u; // gen() true -> Inputs (directed_loop.rs:153-157)
    }
    uint32_t p = p0, rel = rel0, side = side0, visited = 0;
    const uint32_t max_steps = 64u * M + 1024u; // the reference's loop is unbounded (directed_loop.rs:217-301); past this the replica reports ISINGMC_ELIMIT (clearable)
    bool finished = false;
    // A walk is one dependent chain (the longest of a batch's walks ends the launch), so what can be taken off the chain is: the
    // random numbers (64 steps' worth at a time, lane l of every wave holding step base + l: one Philox evaluation per 64 steps instead
    // of one per step on the single working lane), and the first search step's loads (both directions are requested before the
    // vertex update decides which one it will be: the loads fly while thread 0 computes).
    const bool spec = M > 2u * (uint32_t)(U * NT); // (the speculative rows then never reach the vertex itself, whose word is being rewritten)
    uint32_t ox = 0u;
    for (uint32_t step = 1; step <= max_steps; ++step) {
        if (((step - 1u) & 63u) == 0u) ox = rng.draw(SSE_TAG_LOOP, step + (uint32_t)lane).x;
        const uint32_t ux = (uint32_t)__builtin_amdgcn_readlane((int)ox, (int)((step - 1u) & 63u));
        uint32_t wf[U], wb[U];
        if (spec) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const uint32_t d = 1u + (uint32_t)(j * NT + tid);
                uint32_t qf = p + d, qb = p + M - d;
                if (qf >= M) qf -= M;
                if (qb >= M) qb -= M;
                wf[j] = ops[qf]; wb[j] = ops[qb];
            }
        }
        // ---- vertex update by thread 0 ----
        if (tid == 0) {
            const uint32_t word = cur_word;
            const Bd d = decode_bond<CL, W, PM>(B, L, sse_op_bond(word));
            const uint32_t k = d.c != SSE_NO_VAR ? 2u : 1u;
            uint32_t in_e = sse_op_in(word), out_e = sse_op_out(word);
            if (side == 0u) in_e ^= 1u << rel; else out_e ^= 1u << rel;
            double wl[4], total = 0.0;
            for (uint32_t leg = 0; leg < 2u * k; ++leg) {
                uint32_t i2 = in_e, o2 = out_e;
                if (leg < k) i2 ^= 1u << leg; else o2 ^= 1u << (leg - k);
                wl[leg] = op_weight(B, sse_op_bond(word), d, i2, o2);
                total += wl[leg];
            }
            double c = u01(ux) * total;
            uint32_t exit_leg = 2u * k - 1u;
            for (uint32_t leg = 0; leg < 2u * k; ++leg) {
                if (c < wl[leg]) { exit_leg = leg; break; }
                c -= wl[leg];
            }
            const uint32_t xside = exit_leg < k ? 0u : 1u, xrel = exit_leg < k ? exit_leg : exit_leg - k;
            if (xside == 0u) in_e ^= 1u << xrel; else out_e ^= 1u << xrel;
            ops[p] = (word & ~0xFu) | in_e | (out_e << SSE_OP_OUT_SHIFT);
            const bool closed = (p == p0 && xrel == rel0 && xside == side0);
            const uint32_t var = xrel == 0u ? d.a : d.c;
            LDSW(L.o_misc, MISC_LOOP_A) = closed ? 1u : 0u;
            LDSW(L.o_misc, MISC_LOOP_B) = var;
            LDSW(L.o_misc, MISC_LOOP_C) = xside | (xrel << 1) | ((((xside == 1u ? out_e : in_e) >> xrel) & 1u) << 2);
            LDSW(o64, 0) = 0xFFFFFFFFu; LDSW(o64, 1) = 0xFFFFFFFFu; // best (distance << 32 | word found there)
        }
        __syncthreads();
        visited++;
        if (LDSW(L.o_misc, MISC_LOOP_A)) { finished = true; break; }
        const uint32_t var = LDSW(L.o_misc, MISC_LOOP_B);
        const uint32_t info = LDSW(L.o_misc, MISC_LOOP_C);
        const uint32_t xside = info & 1u, newbit = (info >> 2) & 1u;
        const bool forward = xside == 1u;
        // ---- search the next op on worldline `var`, distance 1..M (distance M = the op itself) ----
        uint32_t found = 0xFFFFFFFFu, found_word = 0u;
        for (uint32_t d0 = 1; d0 <= M; d0 += U * NT) {
            uint32_t wd[U];
            if (spec && d0 == 1u) {
#pragma unroll
                for (int j = 0; j < U; ++j) wd[j] = forward ? wf[j] : wb[j];
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const uint32_t d = d0 + (uint32_t)(j * NT + tid);
                    wd[j] = 0u;
                    if (d <= M) {
                        uint32_t q = forward ? p + d : p + M - d;
                        if (q >= M) q -= M;
                        wd[j] = ops[q];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                bool match = false;
                if (wd[j]) {
                    const Bd d = decode_bond<CL, W, PM>(B, L, sse_op_bond(wd[j]));
                    match = d.a == var || d.c == var;
                }
                // (a worldline holds an op every few hundred slots: a handful of lanes per step get here)
                if (match) atomicMin(reinterpret_cast<unsigned long long *>(&LDSW(o64, 0)), ((unsigned long long)(d0 + (uint32_t)(j * NT + tid)) << 32) | wd[j]);
            }
            __syncthreads();
            found = LDSW(o64, 1);
            found_word = LDSW(o64, 0);
            __syncthreads();
            if (found != 0xFFFFFFFFu) break;
        }
        if (found == 0xFFFFFFFFu) { err = 2u; finished = true; break; }
        uint32_t q = forward ? p + found : p + M - found;
        const bool wrapped = forward ? (q >= M) : (found > p);
        if (q >= M) q -= M;
        const Bd dq = decode_bond<CL, W, PM>(B, L, sse_op_bond(found_word)); // (the op itself at distance M: the word as thread 0 just left it)
        const uint32_t nrel = dq.a == var ? 0u : 1u;
        if (wrapped && tid == 0) { // directed_loop.rs:276-288
            const uint32_t wi = var >> 5, bi = var & 31;
            LDSW(L.o_state, wi) = (LDSW(L.o_state, wi) & ~(1u << bi)) | (newbit << bi);
        }
        const uint32_t nside = xside ^ 1u;
        if (q == p0 && nrel == rel0 && nside == side0) { finished = true; break; } // :293
        p = q; rel = nrel; side = nside; cur_word = found_word;
    }
    if (!finished) err = 3u;
    __syncthreads();
    return visited;
}

} // namespace sse
#include "sse_rvb.hip.h"
#include "sse_fast.hip.h"
namespace sse {

// ---------------------------------------------------------------------------------------------
// One launch = nsteps timesteps of every replica.  Reference drivers: QmcIsingGraph::timestep
// (qmc_ising.rs:644-795), Qmc::timestep (qmc_runner.rs:363-377), measurement loop
// QmcStepper::timesteps_measure_with_self (qmc_traits/qmc_stepper.rs:133-162).
// PHASE only tags the symbol (0 = measured path, 1 = data preparation) so that profilers can tell the
// two apart; the code is identical.
// PASSES selects what is compiled in: SSE_PASSES_ALL = every pass (one launch runs whole timesteps), SSE_PASSES_DIAG =
// the diagonal pass alone.  The diagonal pass needs half the registers and a quarter of the LDS of the cluster
// pass, so as its own kernel it runs at twice the occupancy (4 waves per SIMD for W <= 4); the host then issues
// two launches per timestep (isingmc_hip.hip, run()).  n, cutoff, epoch, chunk counters travel through HBM.
// SSE_PASSES_OFFDIAG is the second of those launches with the diagonal and RVB code left out (fewer live scalars).
#ifndef SSE_MIN_WAVES_PER_SIMD
#define SSE_MIN_WAVES_PER_SIMD 1
#endif
enum { SSE_PASSES_ALL = 0, SSE_PASSES_DIAG = 1, SSE_PASSES_OFFDIAG = 2, SSE_PASSES_RVB = 3 }; // DIAG: diagonal pass + directed loop; OFFDIAG: cluster + free spins + sampling; RVB: the RVB sweep alone (its own register budget)
template <int W, int PASSES>
constexpr int sse_waves_per_simd() {
    if (PASSES == SSE_PASSES_DIAG) return W <= 4 ? 4 : (W <= 8 ? 2 : 1);
    return W == 8 ? SSE_MIN_WAVES_PER_SIMD : (W == 6 ? 3 : (W == 4 ? 2 : 1));
}
template <int W, int K, int MODE, int PHASE, int PASSES>
__global__ __launch_bounds__(W * 64, (sse_waves_per_simd<W, PASSES>())) void sweep_kernel(DevBatch B, SweepArgs A) {
    constexpr int NT = W * 64;
    constexpr bool CL = MODE == SSE_MODE_LDS_EDGES, TG = MODE == SSE_MODE_GLOBAL_TABLES || MODE == SSE_MODE_PM_GLOBAL_TABLES;
    constexpr bool PM = MODE == SSE_MODE_PM_LDS_TABLES || MODE == SSE_MODE_PM_GLOBAL_TABLES;
    static_assert(MODE != SSE_MODE_PM_LDS_TABLES || PASSES == SSE_PASSES_DIAG, "mode 3 is the diagonal launch of large +-J models");
    Lds<W> L;
    L.carve(B.N, B.nwords, B.lds_ufcap, CL ? B.E : 0u, B.has_long, TG, PM ? B.pm_words : 0u, MODE == SSE_MODE_PM_LDS_TABLES);
    const int tid = threadIdx.x;
    const uint32_t r = blockIdx.x;
    if (A.only_flagged && !B.aux[r]) return; // (uniform per workgroup; the flag is cleared at the end, behind the barriers below)
    if (B.bond_stride) { // per-replica couplings: this replica's tables (B is this workgroup's private copy)
        const uint32_t hr = B.ham_row ? B.ham_row[r] : r;
        B.bonds += (size_t)hr * B.bond_stride;
        B.cumw += (size_t)hr * B.bond_stride;
        B.wtot = B.wtot_r[hr];
        if constexpr (PM) for (uint32_t i = tid; i < B.pm_words; i += NT) LDSW(L.o_signs, i) = B.pm_signs[(size_t)hr * B.pm_words + i];
    }
    for (uint32_t i = tid; i < B.nwords; i += NT) LDSW(L.o_state, i) = B.state[(size_t)r * B.nwords + i];
    if constexpr (CL)
        for (uint32_t i = tid; i < B.E; i += NT) LDSW(L.o_edges, i) = B.edges_compact[i];
    for (uint32_t i = tid; i < 2 * SSE_MAX_CHUNKS; i += NT) LDSW(L.o_chn, i) = B.chunks[(size_t)r * 2 * SSE_MAX_CHUNKS + i];
    __syncthreads();
    int n = (int)B.n[r], ntrans = (int)B.ntrans[r];
    uint32_t M = B.cutoff[r], err = B.err[r], gr = 0, last_out = 0;
    uint64_t epoch = B.epoch[r];
    const double beta = A.beta ? A.beta[r] : 0.0;
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0;
    for (uint64_t step = 0; step < A.nsteps; ++step) {
        if (err) break;
        if constexpr (PASSES != SSE_PASSES_OFFDIAG && PASSES != SSE_PASSES_RVB)
        if (A.domask & SSE_DO_DIAG) {
            const Rng rng = make_rng(B, r, epoch);
            if (A.domask & SSE_DO_HEATBATH) diagonal_pass<W, K, CL, true, TG, PM>(B, L, r, rng, beta, M, n, ntrans, gr);
            else diagonal_pass<W, K, CL, false, TG, PM>(B, L, r, rng, beta, M, n, ntrans, gr);
            epoch++;
            a5 += M;
            if (A.domask & SSE_DO_GROW) { // qmc_ising.rs:786, qmc_runner.rs:197
                const uint32_t want = (uint32_t)n + (uint32_t)n / 2u;
                if (want > M) { if (want > B.cap) { err = 1u; break; } M = want; }
            }
        }
        if constexpr ((PASSES == SSE_PASSES_ALL || PASSES == SSE_PASSES_RVB) && !TG && !PM) // (RVB keeps its working set in LDS: refused by the host for MODE 2 models)
        if (A.domask & SSE_DO_RVB) { // qmc_ising.rs:705-752
            const uint32_t updates = A.rvb_updates ? A.rvb_updates : (B.N + 1u) / 2u;
            last_out = rvb_pass<W, CL>(B, L, r, epoch, M, updates, gr, err);
            epoch++;
            a4 += updates;
            if (err) break;
        }
        // the directed loop is one sequential walk: it runs in the small geometry of the diagonal launch
        if constexpr (PASSES != SSE_PASSES_OFFDIAG && PASSES != SSE_PASSES_RVB)
        if (A.domask & SSE_DO_LOOP) {
            const Rng rng = make_rng(B, r, epoch);
            last_out = loop_pass<W, CL, PM>(B, L, r, rng, M, n, gr, err);
            epoch++;
            a4 += last_out;
            if (err) break;
        }
        if constexpr (PASSES != SSE_PASSES_DIAG && PASSES != SSE_PASSES_RVB) {
        if (A.domask & SSE_DO_CLUSTER) {
            const Rng rng = make_rng(B, r, epoch);
            const uint32_t S_ids = (uint32_t)W * B.N + (uint32_t)ntrans;
            bool lite_done = false;
            if constexpr (CL && !TG) {
                // the trimmed diagonal kernel labelled the string for exactly this update (same update counter, nothing in between)
                const uint32_t S_lite = B.N + (uint32_t)ntrans;
                if (B.lite && B.lite_epoch[r] == epoch && S_lite <= B.lds_ufcap && S_lite <= 65535u) {
                    last_out = cluster_pass<W, K, CL, false, false, true>(B, L, r, rng, A.prob, M, n, ntrans, gr, err);
                    lite_done = true;
                }
            }
            if constexpr (CL && !TG) {
                // ... or wrote the dense list of occupied slots for it
                if (!lite_done && B.cops && B.cops_epoch[r] == epoch && S_ids <= B.lds_ufcap && S_ids <= 65535u) {
                    last_out = cluster_pass<W, K, CL, false, false, false, true>(B, L, r, rng, A.prob, M, n, ntrans, gr, err);
                    lite_done = true;
                }
            }
            if (lite_done) {}
            else if constexpr (TG) last_out = cluster_pass<W, K, CL, true, true, false, false, PM>(B, L, r, rng, A.prob, M, n, ntrans, gr, err);
            else if (S_ids <= B.lds_ufcap && S_ids <= 65535u) last_out = cluster_pass<W, K, CL, false, false>(B, L, r, rng, A.prob, M, n, ntrans, gr, err);
            else last_out = cluster_pass<W, K, CL, true, false>(B, L, r, rng, A.prob, M, n, ntrans, gr, err);
            epoch++;
            a4 += (uint64_t)n;
            if (err) break;
        }
        if (A.domask & SSE_DO_FREE) {
            const Rng rng = make_rng(B, r, epoch);
            if (!(A.domask & SSE_DO_CLUSTER)) touch_scan<W, CL, PM>(B, L, r, M);
            free_spin_pass<W>(B, L, rng);
            epoch++;
        }
        if (A.sampling_freq && (A.step0 + step + 1) % A.sampling_freq == 0) {
            if (tid == 0) LDSW(L.o_misc, MISC_LOOP_A) = 0u;
            __syncthreads();
            uint32_t up = 0;
            for (uint32_t i = tid; i < B.nwords; i += NT) up += __popc(LDSW(L.o_state, i));
            for (int off = 32; off > 0; off >>= 1) up += __shfl_down(up, off);
            if ((tid & 63) == 0 && up) atomicAdd(&LDSW(L.o_misc, MISC_LOOP_A), up);
            __syncthreads();
            const long long mag = 2ll * (long long)LDSW(L.o_misc, MISC_LOOP_A) - (long long)B.N;
            a0 += (uint64_t)n; a1 += 1; a2 += (uint64_t)(mag < 0 ? -mag : mag); a3 += (uint64_t)(mag * mag); a6 += (uint64_t)ntrans;
            __syncthreads();
        }
        } // PASSES != SSE_PASSES_DIAG
    }
    __syncthreads();
    // (the directed loop of the diagonal launch can flip p=0 spins too)
        for (uint32_t i = tid; i < B.nwords; i += NT) B.state[(size_t)r * B.nwords + i] = LDSW(L.o_state, i);
    for (uint32_t i = tid; i < 2 * SSE_MAX_CHUNKS; i += NT) B.chunks[(size_t)r * 2 * SSE_MAX_CHUNKS + i] = LDSW(L.o_chn, i);
    if (tid == 0) {
        B.n[r] = (uint32_t)n; B.ntrans[r] = (uint32_t)ntrans; B.cutoff[r] = M; B.err[r] = err; B.epoch[r] = epoch;
        if (A.out_u32) A.out_u32[r] = last_out;
        uint64_t *acc = B.acc + (size_t)B.acc_row[r] * 8;
        acc[0] += a0; acc[1] += a1; acc[2] += a2; acc[3] += a3; acc[4] += a4; acc[5] += a5; acc[6] += a6;
        if (A.only_flagged) B.aux[r] = 0u;
    }
}

struct LaunchCfg {
    uint32_t W, K, mode, phase, passes; // mode: SSE_MODE_*
    size_t lds_bytes;
    hipStream_t stream;
};
// one translation unit per W (sweep_w*.hip) defines these
hipError_t launch_sweep_w1(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A);
hipError_t launch_sweep_w4(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A);
hipError_t launch_sweep_w6(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A);
hipError_t launch_sweep_w8(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A);
hipError_t launch_sweep_w16(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A);
hipError_t launch_sweep_fast(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A); // sweep_fast.hip: sse_fast.hip.h, W = 4
hipError_t launch_cluster(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A);    // sweep_cluster.hip: sse_cluster.hip.h, W = 16
size_t cluster_fixed_words(uint32_t N, uint32_t nwords, uint32_t Nb);                    // LDS words of that kernel in front of its parent table
// sweep_rvb.hip (sse_rvb_split.hip.h): the RVB sweep as a growth launch (16 waves) and a main launch (c.W = 4, 8 or 16 waves)
hipError_t launch_rvb_grow(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A);
hipError_t launch_rvb_main(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A);
size_t rvb_split_grow_fixed_words(uint32_t N, uint32_t nwords, uint32_t ledges);           // LDS words of the growth launch in front of the constant-op table
size_t rvb_split_main_words(uint32_t W, uint32_t N, uint32_t nwords, uint32_t ledges, uint32_t E, uint32_t Nb); // LDS words of the main launch
size_t rvb_split_prod_stride(uint32_t Nb);                                                // words per attempt in DevBatch::rvb_prod; 0 = the model is too large for the two-launch form

template <int W, int K, int CL, int PHASE, int PASSES>
hipError_t launch_one(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_kernel<W, K, CL, PHASE, PASSES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sweep_kernel<W, K, CL, PHASE, PASSES>), dim3(B.R), dim3(W * 64), c.lds_bytes, c.stream, B, A);
    return hipGetLastError();
}
template <int W, int K, int CL>
hipError_t launch_k(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    if (c.passes == SSE_PASSES_DIAG) {
        if (c.phase && K == 4) return launch_one<W, K, CL, (K == 4 ? 1 : 0), SSE_PASSES_DIAG>(c, B, A);
        return launch_one<W, K, CL, 0, SSE_PASSES_DIAG>(c, B, A);
    }
    if (c.passes == SSE_PASSES_OFFDIAG) {
        if (c.phase && K == 4) return launch_one<W, K, CL, (K == 4 ? 1 : 0), SSE_PASSES_OFFDIAG>(c, B, A);
        return launch_one<W, K, CL, 0, SSE_PASSES_OFFDIAG>(c, B, A);
    }
    if (c.passes == SSE_PASSES_RVB) {
        if constexpr (CL != SSE_MODE_GLOBAL_TABLES && CL != SSE_MODE_PM_GLOBAL_TABLES) return launch_one<W, K, CL, 0, SSE_PASSES_RVB>(c, B, A);
        else return hipErrorInvalidValue;
    }
    if (c.phase && K == 4) return launch_one<W, K, CL, (K == 4 ? 1 : 0), SSE_PASSES_ALL>(c, B, A); // data-preparation symbol: default geometry only
    return launch_one<W, K, CL, 0, SSE_PASSES_ALL>(c, B, A);
}
template <int W>
hipError_t launch_w(const LaunchCfg &c, const DevBatch &B, const SweepArgs &A) {
    if (c.mode == SSE_MODE_PM_LDS_TABLES || c.mode == SSE_MODE_PM_GLOBAL_TABLES) { // +-J decode: the default geometry of large models only
        if constexpr (W == 4) {
            if (c.K != 4) return hipErrorInvalidValue;
            if (c.mode == SSE_MODE_PM_LDS_TABLES) {
                if (c.passes != SSE_PASSES_DIAG) return hipErrorInvalidValue;
                return c.phase ? launch_one<4, 4, SSE_MODE_PM_LDS_TABLES, 1, SSE_PASSES_DIAG>(c, B, A) : launch_one<4, 4, SSE_MODE_PM_LDS_TABLES, 0, SSE_PASSES_DIAG>(c, B, A);
            }
            return launch_k<4, 4, SSE_MODE_PM_GLOBAL_TABLES>(c, B, A);
        } else return hipErrorInvalidValue;
    }
    if (c.mode == SSE_MODE_GLOBAL_TABLES) { // tables in HBM: slots_per_lane 4 and 1 only
        if (c.K == 4) return launch_k<W, 4, SSE_MODE_GLOBAL_TABLES>(c, B, A);
        if (c.K == 1) return launch_k<W, 1, SSE_MODE_GLOBAL_TABLES>(c, B, A);
        return hipErrorInvalidValue;
    }
    const bool cl = c.mode == SSE_MODE_LDS_EDGES;
    if (c.K == 4 && cl) return launch_k<W, 4, SSE_MODE_LDS_EDGES>(c, B, A);
    if (c.K == 4 && !cl) return launch_k<W, 4, SSE_MODE_GENERAL>(c, B, A);
    if (c.K == 1 && cl) return launch_k<W, 1, SSE_MODE_LDS_EDGES>(c, B, A);
    if (c.K == 1 && !cl) return launch_k<W, 1, SSE_MODE_GENERAL>(c, B, A);
    if (c.K == 2 && cl) return launch_k<W, 2, SSE_MODE_LDS_EDGES>(c, B, A);
    if (c.K == 2 && !cl) return launch_k<W, 2, SSE_MODE_GENERAL>(c, B, A);
    return hipErrorInvalidValue;
}

} // namespace sse
