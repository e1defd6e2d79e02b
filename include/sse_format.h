/*
 * sse_format.h — data formats shared by the C-ABI, the HIP kernels and the CPU oracle.
 *
 * This header defines DATA LAYOUT ONLY (no algorithm): the 32-bit operator word, the bond
 * table entry, the Philox counter assignment and the stream tags.  Every implementation
 * (the HIP sources under isingmontecarlo_amd/csrc and the C sources under oracle) restates the algorithms independently.
 *
 * Reference types replaced (all paths relative to /root/reference):
 *   - src/sse/qmc_traits/op_container.rs:224-237  BasicOp{vars,bond,in_out,constant}
 *   - src/sse/fast_ops.rs:181-190                 FastOpNode (op + p links + per-var links)
 *   - src/sse/fast_ops.rs:35-49                   FastOpsTemplate{ops,n,p_ends,var_ends,bond_counters}
 * The ≈240 B array-of-structs node becomes ONE u32 per slot; links are never stored, they are
 * recomputed on chip by ordered scans.
 */
#ifndef SSE_FORMAT_H
#define SSE_FORMAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- operator word -------------------------------------------------------------------
 *   word == 0            : identity (empty slot)
 *   bits [31:4] = bond+1 : bond index into the bond table
 *   bit 0 = input  spin of relative var 0      bit 1 = input  spin of relative var 1
 *   bit 2 = output spin of relative var 0      bit 3 = output spin of relative var 1
 * Single-site bonds keep bits 1 and 3 zero.  An op is diagonal iff (bits[1:0] == bits[3:2]).
 */
#define SSE_OP_EMPTY 0u
#define SSE_OP_BOND_SHIFT 4
#define SSE_OP_IN_MASK 0x3u
#define SSE_OP_OUT_SHIFT 2
#define SSE_MAX_BONDS ((1u << 28) - 2u)

#if defined(__HIPCC__)
#define SSE_HD __host__ __device__
#else
#define SSE_HD
#endif

SSE_HD static inline uint32_t sse_op_make(uint32_t bond, uint32_t in_bits, uint32_t out_bits) {
    return ((bond + 1u) << SSE_OP_BOND_SHIFT) | (in_bits & 3u) | ((out_bits & 3u) << SSE_OP_OUT_SHIFT);
}
SSE_HD static inline uint32_t sse_op_bond(uint32_t w) { return (w >> SSE_OP_BOND_SHIFT) - 1u; }
SSE_HD static inline uint32_t sse_op_in(uint32_t w) { return w & 3u; }
SSE_HD static inline uint32_t sse_op_out(uint32_t w) { return (w >> SSE_OP_OUT_SHIFT) & 3u; }
SSE_HD static inline int sse_op_is_diagonal(uint32_t w) { return sse_op_in(w) == sse_op_out(w); }

/* ---- bond table ----------------------------------------------------------------------
 * Bond numbering follows src/sse/qmc_ising.rs:186-205,228-246:
 *   [0,E)        two-site Ising edges          (constant = false)
 *   [E,E+N)      transverse field, single site (constant = true : a cluster boundary)
 *   [E+N,E+2N)   longitudinal field, single site (constant = false), present iff |h| > DBL_EPSILON
 * A bond's diagonal weight is  (satisfied ? weight : 0)  with
 *   two-site     : satisfied = ((s_a == s_b) == prefers_aligned),  weight = 2|J|
 *   transverse   : always satisfied, every (in,out) pair,          weight = Gamma
 *   longitudinal : satisfied = (s == prefers_up),                  weight = 2|h|
 * (src/sse/qmc_ising.rs:863-888).
 */
#define SSE_BOND_TWO_SITE 0u
#define SSE_BOND_TRANSVERSE 1u
#define SSE_BOND_LONGITUDINAL 2u
#define SSE_BOND_KIND_MASK 0x3u
#define SSE_BOND_PREF_BIT 0x4u /* two-site: prefers aligned (J<0); longitudinal: prefers up (h>0) */
#define SSE_NO_VAR 0xFFFFFFFFu

/* ---- Philox4x32-10 counter assignment ------------------------------------------------
 *   key     = (seed_lo, seed_hi)
 *   counter = (index, epoch_lo, replica, (tag << 24) | (epoch_hi & 0xFFFFFF))
 * index is the op-string slot p (heat-bath diagonal pass), the slot pair (Metropolis diagonal pass, see
 * below), the canonical cluster label (cluster coin), the variable (free spins / initial state) or the
 * step number (directed loop).
 * Metropolis diagonal pass: slots p and p^64 share one call with index = p & ~64; the slot with bit 6
 * clear uses words (0: bond choice, 1: accept), the other words (2: bond choice, 3: accept).
 * Every primitive update consumes one epoch value and increments the replica's epoch.
 */
#define SSE_TAG_INIT 0u
#define SSE_TAG_DIAG 1u
#define SSE_TAG_CLUSTER 2u
#define SSE_TAG_FREE 3u
#define SSE_TAG_LOOP 4u
#define SSE_TAG_RVB 5u
#define SSE_TAG_PT 6u
#define SSE_TAG_HEATBATH 7u

#ifdef __cplusplus
}
#endif
#endif /* SSE_FORMAT_H */
