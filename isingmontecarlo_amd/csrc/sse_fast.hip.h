// sse_fast.hip.h — the diagonal sweep of the headline geometry, written for instruction count.
//
// Same algorithm, same Philox counters and bit-identical results as sse::diagonal_pass<4, K, CL = true, HB = false> (which stays the
// general implementation and the one the parity tests drive in every other geometry); reference: DiagonalUpdater::
// make_diagonal_update_with_rng_and_state_ref (qmc_traits/diagonal.rs:114-135) with metropolis_single_diagonal_update (:142-191).
// The general pass is VALU-bound at ~270 vector instructions per 64-slot row; this one spends ~110:
//   * ONE packed LDS table entry per bond (two-site, transverse and longitudinal alike) gives both variables, the truth table
//     "does a diagonal op on this bond have weight in spin state (sa, sc)", the state-bit mask of its op word and the weight
//     class — no selects on the bond kind anywhere;
//   * the W = 4 per-wave copies of the propagated spins are the four BYTES of one 32-bit entry per variable, so an
//     off-diagonal op reaches the copies of all later (or all earlier) waves with one ds_xor_b32 instead of a loop of W - 1
//     predicated atomics;
//   * candidates are told apart by wave masks only (insert / removal), never by f64 selects: every lane evaluates both
//     inequalities of the rule, the masks pick; the f64 expressions are exactly those of the general pass and of the oracle;
//   * no scalar-register spills: the kernel holds nothing but the diagonal pass (and the short directed loop behind it).
// Requirements (checked by the host, isingmc_hip.hip): uniform |J| (LDS edge tables), N <= 4096 variables, 4 waves per
// replica, Metropolis rule, two launches per timestep.
#pragma once

namespace sse {

#define SSE_FAST_MAX_VARS 4096u
#define LDS8(a) (reinterpret_cast<uint8_t *>(lds_raw)[(a)]) // byte at LDS byte address a
// packed bond entry: a [0,12) | c [12,24) (= a for one-variable bonds) | ok4 [24,28): bit (sa | sc << 1) set iff a diagonal op
// has non-zero weight in that state | submask [28,30): state bits an op word of this bond carries (3 or 1) | class [30,32)
#define SSE_FAST_CLASS_J 0u
#define SSE_FAST_CLASS_G 1u
#define SSE_FAST_CLASS_H 2u

// v_cndmask on a wave mask held in scalar registers: mask bit of the lane set ? a : b
__device__ __forceinline__ uint32_t sel64(uint64_t mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}

// bitfield insert: (a & mask) | (b & ~mask)
// (as an instruction: written in C the optimiser turns it back into compare + select, the very thing it is here to avoid)
__device__ __forceinline__ uint32_t bfi32(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}

struct FastLds {
    uint32_t o_nb;   // [4][2] f64 per class: beta*Nb*weight * 2^32 and beta*Nb*weight * 2^-32 (the 2^-32 of the uniform folded in)
    uint32_t o_tab;  // [Nb] packed bond entries
    uint32_t o_spin; // [N] u32: byte w = wave w's copy of the propagated spin (bit 0) + in-row event marker (bits 1..7)
    uint32_t o_dummy; // [64] one word per lane: target of the stores / atomics of lanes that have nothing to store
    uint32_t o_rank;  // [4][N] u32 (labelling only): row w = wave w's copy of "1 + dense index of the latest cut on the worldline"
                      // (bits 0..15, 0 = none yet) + in-row cut marker (bits 16..23).  Wave-major: a wave's random accesses
                      // spread over all 32 banks (variable-major would put every access of a wave on 8 of them)
    uint32_t o_tb;    // [N] u8 (labelling only): the variable carries an op
    uint32_t end;
};
// The tables start where the general layout keeps the compact edge table: the directed loop behind the diagonal pass is the
// only user of that table in this kernel and stages it when the tables below are dead.
template <int W>
__device__ __forceinline__ FastLds fast_carve(const Lds<W> &L, const DevBatch &B, bool label) {
    FastLds F;
    uint32_t base = (L.o_edges + 3u) & ~3u; // 16-byte aligned: the class constants are read as one b128
    F.o_nb = base; base += 16;
    F.o_tab = base; base += B.Nb;
    F.o_spin = base; base += B.N;
    F.o_dummy = base; base += 64;
    F.o_rank = base; base += label ? 4u * B.N : 0u;
    F.o_tb = base; base += label ? (B.N + 3u) / 4u : 0u;
    F.end = base;
    return F;
}

__device__ __forceinline__ uint32_t fast_entry(const DevBatch &B, uint32_t b, uint32_t ce) {
    if (b < B.E) {
        const uint32_t a = ce & SSE_CE_VAR_MASK, c = (ce >> 15) & SSE_CE_VAR_MASK, pref = (ce >> 30) & 1u;
        return a | (c << 12) | ((pref ? 0x9u : 0x6u) << 24) | (3u << 28) | (SSE_FAST_CLASS_J << 30);
    }
    const uint32_t s1 = b - B.E;
    if (s1 < B.N) return s1 | (s1 << 12) | (0xFu << 24) | (1u << 28) | (SSE_FAST_CLASS_G << 30);
    const uint32_t v = s1 - B.N; // longitudinal: sc == sa, so only states 00 and 11 occur
    return v | (v << 12) | ((B.hpos ? 0x8u : 0x1u) << 24) | (1u << 28) | (SSE_FAST_CLASS_H << 30);
}

// LABEL: also label every leg of the final string with its segment id for the cluster update that follows in the same
// timestep (cluster.rs:193-271 expand_whole_cluster's bookkeeping; the ordered-scan formulation of cluster_scan): B.segs per
// slot, the segment pairs joined by two-site ops (B.pairs), touched variables, last cut per worldline.  The labelling of
// tile t runs one tile late, behind the second barrier of tile t + 1: by then every wave knows how many cuts the earlier
// waves kept in tile t (dense cut numbering in p order), and has received their cuts in its copy of the rank table.
// COMPACT: also write the occupied slots of the final string as a dense list in p order (B.cops: op words, B.cpos: their
// slots) for the cluster update of the same timestep, whose ordered scan then runs over n instead of M elements with every lane
// useful.  An op's index in that list is the number of ops in front of it in the final string: ops in front of the tile (a running
// total), in the earlier waves' shares of the tile (their old occupancy, published once with the first round of the tile, plus
// their net accepted moves, which the rounds exchange anyway) and at earlier rows / lanes of this wave (wave masks).
template <int K, bool LABEL, bool COMPACT>
__device__ __forceinline__ void diagonal_fast(const DevBatch &B, const Lds<4> &L, const FastLds &F, uint32_t r, const Rng &rng, double beta,
                                              uint32_t M, int &n_io, int &ntrans_io, uint32_t &gr, uint32_t fbmask) {
    constexpr int W = 4, NT = W * 64;
    constexpr uint32_t TS = (uint32_t)(NT * K);
    static_assert(K == 2 || K == 4, "rows p and p + 64 share one Philox call");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    // pending cluster flips of this replica (fbmask = 0xFF) are applied to every word as it is loaded: one byte per slot, the xor
    // mask of its four state bits (0 for empty slots and beyond the cutoff); fbmask = 0: the string in HBM is current
    // (the byte load is unconditional — a branch would put it into a basic block of its own and cost the prefetch its counted
    // waits; without pending flips it reads bytes of the string itself and the mask drops them)
    const uint8_t *flipb = (fbmask ? B.flipb : reinterpret_cast<const uint8_t *>(B.ops)) + (size_t)r * B.stride;
    auto ld_word = [&](uint32_t idx) -> uint32_t {
        return row_ld(ops, idx) ^ ((uint32_t)*reinterpret_cast<const uint8_t *>(reinterpret_cast<const char *>(flipb) + (size_t)idx) & fbmask);
    };
    const uint32_t N = B.N, Nb = B.Nb, E = B.E;
    for (uint32_t i = tid; i < N; i += NT) LDSW(F.o_spin, i) = ((LDSW(L.o_state, i >> 5) >> (i & 31)) & 1u) * 0x01010101u;
    if constexpr (LABEL) {
        for (uint32_t i = tid; i < 4u * N; i += NT) LDSW(F.o_rank, i) = 0u;
        for (uint32_t i = tid; i < (N + 3u) / 4u; i += NT) LDSW(F.o_tb, i) = 0u;
    }
    __syncthreads();

    const uint32_t ntiles = (M + TS - 1) / TS;
    int n_start = n_io, ntrans = 0;
    // bytes of the waves after / before this one in a spin entry (wave-uniform)
    const uint32_t m_later = (uint32_t)((0x0101010100ull << (8 * wave)) & 0xFFFFFFFFull);
    const uint32_t m_earlier = 0x01010101u & ((1u << (8 * wave)) - 1u);
    const uint32_t spin_my = 4u * F.o_spin + (uint32_t)wave; // byte address of this wave's copy of variable 0
    const uint32_t markhi = (127u - (uint32_t)lane) << 1; // marker of an off-diagonal op of this lane (bits 1..7 of its variable's byte)

    // Predicated LDS stores and atomics are written branch-free: lanes that have nothing to do are pointed at a per-lane dummy
    // word instead (an address select costs two integer instructions; an exec-masked store costs a compare, a mask in two
    // scalar registers and an exec save / restore).
    const uint32_t dummy_w = F.o_dummy + (uint32_t)lane; // word index of this lane's dummy word
    // all-ones iff the op word is off-diagonal (CL: only transverse-field ops can be: bit 0 of in ^ out)
    auto evmask32 = [](uint32_t wd) -> uint32_t { return 0u - ((wd ^ (wd >> 2)) & 1u); };
    // flip the spin of the off-diagonal ops among K words in the copies selected by `mask`
    auto propagate = [&](const uint32_t (&wd)[K], uint32_t mask) {
        if (mask == 0u) return; // wave-uniform
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t v = F.o_spin + (wd[j] >> 4) - 1u - E;
            atomicXor(&lds_raw[bfi32(evmask32(wd[j]), v, dummy_w)], mask);
        }
    };

    uint32_t wnext[K];
#pragma unroll
    for (int j = 0; j < K; ++j) wnext[j] = ld_word((uint32_t)(wave * 64 * K + j * 64 + lane));
    propagate(wnext, m_later);
    __syncthreads();
    const uint32_t lane2 = 2u * (uint32_t)lane;
    const uint32_t vM = vgpr_copy_u32(M), vM1 = vgpr_copy_u32(M + 1u), vzero = vgpr_copy_u32(0u);

    // ---- labelling pipeline state (LABEL) ----
    uint32_t entp[K];         // previous tile: table entry of each slot's bond, with the truth-table bits replaced by "the final slot holds an op" (bit 24)
    uint32_t rk1[K];          // previous tile: 1 + dense index of this lane's cut (0 = not a cut)
    uint32_t cut2[K];         // the tile before: the same | variable << 16 (still owed to the copies of the earlier waves)
#pragma unroll
    for (int j = 0; j < K; ++j) { entp[j] = 0u; rk1[j] = 0u; cut2[j] = 0u; }
    uint32_t ccum = 0;        // COMPACT: ops of the final string in front of the current tile
    uint32_t *const cops_row = COMPACT ? B.cops + (size_t)r * B.stride : nullptr;
    uint32_t *const cpos_row = COMPACT ? B.cpos + (size_t)r * B.stride : nullptr;
    uint32_t cntp = 0;        // cuts this wave kept in the previous tile
    uint32_t kbase_prev = 0;  // dense index of this wave's first cut in the previous tile
    uint32_t cuts_before = 0; // cuts in all tiles before the previous one
    uint32_t pcount = 0;      // pairs appended by this wave
    uint32_t *const segs_row = B.segs + (size_t)r * B.stride;
    uint32_t *const pairs_w = B.pairs + (size_t)r * B.stride + (size_t)wave * (B.stride / 4u);
    // dense index of the first cut of this wave in the previous tile, from the counts of all waves (read behind a barrier)
    auto cut_ranks = [&](uint32_t kbase) {
        kbase_prev = kbase;
        uint32_t k0 = kbase;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint64_t cutm = sse_ballot((entp[j] & 0xC1000000u) == ((SSE_FAST_CLASS_G << 30) | (1u << 24))); // an op, on a transverse-field bond
            const uint32_t kown = __builtin_amdgcn_mbcnt_hi((uint32_t)(cutm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cutm, k0 + 1u));
            rk1[j] = sel64(cutm, kown, vzero);
            k0 += (uint32_t)popc64(cutm);
        }
    };
    // cuts -> the rank-table copies of the waves [wlo, whi): a max (ranks grow with p); only the cut lanes take part
    auto push_ranks = [&](const uint32_t (&va)[K], const uint32_t (&rk)[K], int wlo, int whi) {
        if (wlo >= whi) return; // wave-uniform
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (rk[j] != 0u)
                for (int w2 = wlo; w2 < whi; ++w2) atomicMax(&LDSW(F.o_rank, (uint32_t)w2 * N + va[j]), rk[j]);
    };
    // label the previous tile (index tl): rows in order, this wave's copy of the rank table carries the state between them.
    // Lane predicates are 32-bit all-ones / zero values here (selects are v_bfi, stores go to the lane's dummy word): the only
    // wave masks are the two that feed prefix counts.
    const uint32_t rank_my_b = 4u * (F.o_rank + (uint32_t)wave * N), dummy_b = 4u * dummy_w, tb_b = 4u * F.o_tb;
    const uint32_t vone = vgpr_copy_u32(1u);
    auto label_tile = [&](uint32_t tl) {
        const uint32_t pb = tl * TS + (uint32_t)(wave * 64 * K + lane);
        const uint32_t Nm1 = N - 1u;
        uint32_t k0 = kbase_prev; // dense index of the first cut of the row being labelled
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t e = entp[j], rk = rk1[j];
            const uint32_t va = e & 0xFFFu, vc = (e >> 12) & 0xFFFu;
            const uint32_t ba = rank_my_b + 4u * va, bc = rank_my_b + 4u * vc; // byte addresses of this wave's entries
            const uint64_t cutm = sse_ballot(rk != 0u);
            const uint32_t cut32 = sel64(cutm, ~vzero, vzero);
            const uint32_t ne32 = (uint32_t)((int32_t)(e << 7) >> 31); // bit 24 of the entry: the slot holds an op
            const uint32_t two32 = ne32 & (uint32_t)((int32_t)(e << 2) >> 31); // bit 29 of the entry: a two-site bond
            const uint32_t kown1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(cutm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cutm, 1u)); // 1 + cuts of this row at earlier lanes
            // in-row ordering (as in cluster_scan): the cut lanes publish 1 + their rank inside the row, everybody reads once
            LDS8(bfi32(cut32, ba + 2u, dummy_b)) = (uint8_t)kown1;
            SSE_WAVE_FENCE();
            const uint32_t ea = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(lds_raw) + ba);
            const uint32_t ec = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(lds_raw) + bc);
            const uint32_t xa = ea & 0xFFFFu, xc = ec & 0xFFFFu, ma = (ea >> 16) & 0xFFu, mc = (ec >> 16) & 0xFFu;
            uint32_t seg_a = xa ? Nm1 + xa : va, seg_c = xc ? Nm1 + xc : vc;
            const uint64_t dup = cutm & sse_ballot(ma != kown1);
            const uint32_t firstid = N + k0; // id of the row's first cut (wave-uniform)
            if (!dup) {
                seg_a = ((ma - 1u) < kown1 - 1u) ? firstid + (ma - 1u) : seg_a; // ma == 0: no cut on the variable in this row
                seg_c = ((mc - 1u) < kown1 - 1u) ? firstid + (mc - 1u) : seg_c;
                SSE_WAVE_FENCE();
                *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(lds_raw) + bfi32(cut32, ba, dummy_b)) = rk; // latest cut on the variable; clears the marker byte
            } else { // two cuts of this row on one variable (rare): lane order decides
                bool lastcut = (rk != 0u);
                uint64_t m = cutm;
                uint32_t idL = firstid;
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
                if (rk != 0u) LDS8(ba + 2u) = (uint8_t)0;
                if (lastcut) *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(lds_raw) + ba) = rk; // the last cut wins
                SSE_WAVE_FENCE();
            }
            LDS8(bfi32(ne32, tb_b + va, dummy_b)) = (uint8_t)vone;
            LDS8(bfi32(ne32, tb_b + vc, dummy_b)) = (uint8_t)vone;
            const uint32_t hi = bfi32(cut32, Nm1 + rk, bfi32(two32, seg_c, seg_a)); // cut: the segment it opens; two-site: the second leg's
            row_st(segs_row, pb + (uint32_t)(j * 64), seg_a | (hi << 16));
            const uint64_t twom = sse_ballot(two32 != 0u);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(twom >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)twom, pcount));
            if (two32) row_st(pairs_w, pos, seg_a | (seg_c << 16));
            pcount += (uint32_t)popc64(twom);
            cut2[j] = rk | ((va & cut32) << 16);
            k0 += (uint32_t)popc64(cutm);
        }
    };

    for (uint32_t tile = 0; tile < ntiles; ++tile) {
        // The issue arbiter serves the oldest wave first: of the four workgroups of a CU the first dispatched runs almost as if alone
        // (measured: the four end at 372 / 438 / 511 / 600 us on every CU) and the last one finishes the launch alone on its SIMDs,
        // bound by latency.  Rotating the waves' priorities lets the four progress side by side: they end within 60 us of each other
        // and the launch is 11 % shorter.
#if SSE_ROTATE_PRIO // (0: off — the arbiter's own order, for tools/wg_timeline.py)
        if ((tile & (SSE_ROTATE_PRIO - 1u)) == 0u) sse_set_prio((tile / SSE_ROTATE_PRIO) + blockIdx.x); // (the wave's slot on its SIMD, HW_ID.WAVE_ID, as the phase: no better)
#endif
        uint32_t word[K];
#pragma unroll
        for (int j = 0; j < K; ++j) word[j] = wnext[j];
        const uint32_t pbase = tile * TS + (uint32_t)(wave * 64 * K + lane);
        {
            const uint32_t pn = (tile + 1 < ntiles ? pbase + TS : pbase);
#pragma unroll
            for (int j = 0; j < K; ++j) wnext[j] = ld_word(pn + (uint32_t)(j * 64));
        }
        const bool partial = tile * TS + TS > M; // wave-uniform: only the last tile can hold slots >= M

        double un[K], nb[K];
        uint32_t cbv[K], neww[K], ent[K], bnd[K], rr1[K];
        uint64_t insm[K], remm[K], acc[K];
        // Lane predicates that have to survive the rounds are wave masks (insert candidates, removal candidates, accepted);
        // everything else is recomputed from the op word with integer arithmetic when needed — the pass is short of scalar
        // registers, and a spilled mask costs a v_readlane per half and use.
        // ---- phase 1, all rows at once (nothing here depends on the spin tables): random numbers, bond, packed table entry
        uint32_t occ_w = 0; // COMPACT: ops this wave's share of the tile holds before the update
        {
            uint4 rnd = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t wd = word[j];
                if ((j & 1) == 0) rnd = rng.draw(SSE_TAG_DIAG, pbase + (uint32_t)(j * 64)); // bit 6 of the slot index is clear on even rows
                const uint32_t r0 = (j & 1) ? rnd.z : rnd.x;
                rr1[j] = (j & 1) ? rnd.w : rnd.y;
                const uint64_t occm = sse_ballot(wd != 0u);
                if constexpr (COMPACT) occ_w += (uint32_t)popc64(occm);
                bnd[j] = sel64(occm, (wd >> 4) - 1u, __umulhi(r0, Nb));
                ent[j] = LDSW(F.o_tab, bnd[j]);
            }
        }
        // ---- phase 2, row after row: the propagated spins of the two variables.  In-row ordering of the off-diagonal ops: they
        // publish (127 - lane, spin before) in their variable's byte, everybody reads, they store the spin after.  A reader's
        // spin is flipped iff the publishing lane is below its own: 127 - L' < lane, i.e. bit 8 of (byte + 2 * lane) — plain
        // integer arithmetic, no compare.  (Rows without such an op run the same code: clean bytes have L' = 0, never a flip.)
        uint32_t sub0[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t e = ent[j], inb = word[j] & 1u;
            const uint32_t va = e & 0xFFFu, vc = (e >> 12) & 0xFFFu;
            const uint32_t adr_a = spin_my + 4u * va, adr_c = spin_my + 4u * vc;
            const uint32_t mark = markhi | inb;
            const uint32_t ev32 = evmask32(word[j]);
            const uint32_t adr_w = bfi32(ev32, adr_a, 4u * dummy_w); // where this lane stores: its variable's byte, or its dummy
            LDS8(adr_w) = (uint8_t)mark;
            SSE_WAVE_FENCE();
            const uint32_t ea = LDS8(adr_a), ec = LDS8(adr_c);
            uint32_t sa, sc;
            const uint64_t dup = sse_ballot(((ea ^ mark) & ev32) != 0u); // an off-diagonal op whose marker was overwritten
            if (!dup) {
                sa = (ea ^ ((ea + lane2) >> 8)) & 1u;
                sc = (ec ^ ((ec + lane2) >> 8)) & 1u;
                SSE_WAVE_FENCE();
                LDS8(adr_w) = (uint8_t)(inb ^ 1u);
            } else { // two off-diagonal ops of this row on one variable (rare): resolve in lane order
                sa = ea & 1u; sc = ec & 1u;
                bool seen_a = false, seen_c = false;
                uint64_t m = sse_ballot(ev32 != 0u);
                while (m) {
                    const int Ls = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const bool later = lane > Ls;
                    const uint32_t vL = __builtin_amdgcn_readlane(va, Ls), inL = __builtin_amdgcn_readlane(inb, Ls);
                    if (va == vL) { sa = later ? (inL ^ 1u) : (seen_a ? sa : inL); seen_a = true; }
                    if (vc == vL) { sc = later ? (inL ^ 1u) : (seen_c ? sc : inL); seen_c = true; }
                    if (lane == Ls) LDS8(adr_a) = (uint8_t)(inL ^ 1u); // in order: the last one wins
                }
                SSE_WAVE_FENCE();
            }
            sub0[j] = sa | (sc << 1);
        }
        // ---- phase 3, all rows: candidates and the operands of the rule
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t e = ent[j], wd = word[j];
            const uint32_t okbit = (e >> (24u + sub0[j])) & 1u; // a diagonal op on this bond has weight in this spin state
            const uint32_t sub = sub0[j] & (e >> 28) & 3u;
            const double2 nbp = *reinterpret_cast<const double2 *>(&lds_raw[F.o_nb + 4u * (e >> 30)]);
            const double u = (double)rr1[j]; // the uniform is u * 2^-32: the power of two sits in the two table constants (exact)
            insm[j] = sse_ballot(okbit > wd); // empty slot (word 0) and okbit 1
            if (partial) insm[j] &= sse_ballot(pbase + (uint32_t)(j * 64) < M);
            remm[j] = sse_ballot((wd & ~evmask32(wd)) != 0u); // occupied and diagonal
            // insert:  (u 2^-32) * den < num   <=>  u * den < num * 2^32  (u is converted again in every round: one instruction
            // against two registers per row held across the rounds)
            un[j] = u * nbp.y;  // remove:  (u 2^-32) * num < den   with  u * (num 2^-32) == (u 2^-32) * num  bit for bit
            nb[j] = nbp.x;
            cbv[j] = sel64(insm[j], vM, vM1);
            neww[j] = sel64(insm[j], ((bnd[j] + 1u) << 4) | sub | (sub << 2), vzero); // what an accepted candidate leaves in the slot
            acc[j] = 0ull;
        }

        // ---- fixed point on the live operator count (see diagonal_pass) ----
        int npref[K];
#pragma unroll
        for (int j = 0; j < K; ++j) npref[j] = n_start;
        int tot_all = 0, base = 0;
        uint32_t occbase = 0, occall = 0;
        bool first = true;
        for (;;) {
            int wtot = 0;
            bool changed = first;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const double t = (double)(int)(cbv[j] - (uint32_t)npref[j]);
                const uint64_t lt_ins = sse_ballot((double)rr1[j] * t < nb[j]);
                const uint64_t lt_rem = sse_ballot(un[j] < t);
                const uint64_t a = (lt_ins & insm[j]) | (lt_rem & remm[j]);
                changed |= a != acc[j];
                acc[j] = a;
                wtot += popc64(a & insm[j]) - popc64(a & remm[j]);
            }
            const int buf = gr & 1;
            // (LABEL: the first round also carries the number of cuts this wave kept in the previous tile, above bit 0)
            if (lane == 0) { LDSI(L.o_tot, buf * W + wave) = wtot; LDSW(L.o_chg, buf * W + wave) = (changed ? 1u : 0u) | (LABEL && first ? cntp << 1 : 0u) | (COMPACT && first ? occ_w << 1 : 0u); }
            __syncthreads();
            if (first) {
                // this tile's off-diagonal ops -> copies of the earlier waves (every reader of this tile is done); the next
                // tile's -> copies of the later waves (visible behind the next barrier, before anybody decodes that tile)
                propagate(word, m_earlier);
                if (tile + 1 < ntiles) propagate(wnext, m_later);
            }
            base = 0; tot_all = 0; uint32_t anychg = 0, cbase = 0, call = 0;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) {
                const int t = __builtin_amdgcn_readfirstlane(LDSI(L.o_tot, buf * W + w2));
                if (w2 < wave) base += t;
                tot_all += t;
                const uint32_t cw = (uint32_t)__builtin_amdgcn_readfirstlane((int)LDSW(L.o_chg, buf * W + w2));
                anychg |= cw & 1u;
                if (w2 < wave) cbase += cw >> 1;
                call += cw >> 1;
            }
            if constexpr (COMPACT) if (first) { occbase = cbase; occall = call; }
            if constexpr (LABEL) if (first) {
                // the previous tile's cuts get their dense numbers and go to the copies of the later waves (read behind the next
                // barrier); the cuts of the tile before that go to the copies of the earlier waves (which are done labelling it)
                if (tile >= 1u) {
                    cut_ranks(cuts_before + cbase);
                    cuts_before += call;
                    uint32_t va[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) va[j] = entp[j] & 0xFFFu;
                    push_ranks(va, rk1, wave + 1, W);
                }
                uint32_t va2[K], rk2[K];
#pragma unroll
                for (int j = 0; j < K; ++j) { va2[j] = cut2[j] >> 16; rk2[j] = cut2[j] & 0xFFFFu; }
                push_ranks(va2, rk2, 0, wave);
            }
            gr++;
            if (!first && !anychg) break;
            first = false;
            int run = n_start + base;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint64_t im = acc[j] & insm[j], rm = acc[j] & remm[j];
                const int ci = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, (uint32_t)run));
                const int cr = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));
                npref[j] = ci - cr;
                run += popc64(im) - popc64(rm);
            }
        }
        // ---- commit ----
        int dn = 0, dtr = 0;
        uint32_t fwn[K], cntn = 0u;
        uint32_t crun = ccum + occbase + (uint32_t)base; // COMPACT: index of this wave's first op of the tile in the dense list
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t fw = sel64(acc[j], neww[j], word[j]);
            row_st(ops, pbase + (uint32_t)(j * 64), fw);
            const uint64_t im = acc[j] & insm[j], rm = acc[j] & remm[j];
            if constexpr (COMPACT) {
                const uint64_t nemf = (sse_ballot(word[j] != 0u) & ~rm) | im; // the slot holds an op afterwards
                const uint32_t ci = __builtin_amdgcn_mbcnt_hi((uint32_t)(nemf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nemf, crun));
                if (fw != 0u) { row_st(cops_row, ci, fw); row_st(cpos_row, ci, pbase + (uint32_t)(j * 64)); }
                crun += (uint32_t)popc64(nemf);
            }
            const uint64_t trm = sse_ballot((ent[j] >> 30) == SSE_FAST_CLASS_G); // the bond at stake is a transverse-field bond
            dn += popc64(im) - popc64(rm);
            dtr += popc64(im & trm) - popc64(rm & trm);
            if constexpr (LABEL) {
                fwn[j] = fw;
                // cuts in the final string: transverse-field ops, diagonal or not = (kept ops | accepted inserts) on such bonds
                cntn += (uint32_t)popc64(((sse_ballot(word[j] != 0u) & ~rm) | im) & trm);
            }
        }
        if constexpr (LABEL) {
            // label the previous tile now: behind the last barrier of this tile's rounds, and with the round state dead
            if (tile >= 1u) label_tile(tile - 1u);
#pragma unroll
            for (int j = 0; j < K; ++j) entp[j] = (ent[j] & 0xF0FFFFFFu) | (min(fwn[j], 1u) << 24);
            cntp = cntn;
        }
        ntrans += dtr;
        if (lane == 0 && (dtr | dn)) { // the 64*K slots of a wave's share of a tile lie inside one chunk
            const uint32_t ch = (tile * TS + (uint32_t)(wave * 64 * K)) / B.CH;
            if (dn) atomicAdd(&LDSW(L.o_chn, ch), (uint32_t)dn);
            if (dtr) atomicAdd(&LDSW(L.o_chtr, ch), (uint32_t)dtr);
        }
        n_start += tot_all;
        if constexpr (COMPACT) ccum += occall + (uint32_t)tot_all;
    }
    if constexpr (LABEL) if (ntiles > 0u) {
        // drain the labelling pipeline: counts of the last tile, its cuts to the later waves, the owed ones to the earlier waves
        const int buf = gr & 1;
        if (lane == 0) LDSW(L.o_chg, buf * W + wave) = cntp << 1;
        __syncthreads();
        uint32_t cbase = 0;
#pragma unroll
        for (int w2 = 0; w2 < W; ++w2) {
            const uint32_t cw = (uint32_t)__builtin_amdgcn_readfirstlane((int)LDSW(L.o_chg, buf * W + w2));
            if (w2 < wave) cbase += cw >> 1;
        }
        gr++;
        cut_ranks(cuts_before + cbase);
        uint32_t va[K], va2[K], rk2[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { va[j] = entp[j] & 0xFFFu; va2[j] = cut2[j] >> 16; rk2[j] = cut2[j] & 0xFFFFu; }
        push_ranks(va, rk1, wave + 1, W);
        push_ranks(va2, rk2, 0, wave);
        __syncthreads();
        label_tile(ntiles - 1u);
        __syncthreads();
        // hand-over to the cluster update: last cut per worldline (all copies together have seen every cut), touched variables
        uint16_t *lastrank = B.lastrank + (size_t)r * N;
        for (uint32_t v = tid; v < N; v += NT) {
            const uint32_t a = LDSW(F.o_rank, v) & 0xFFFFu, b2 = LDSW(F.o_rank, N + v) & 0xFFFFu;
            const uint32_t c2 = LDSW(F.o_rank, 2u * N + v) & 0xFFFFu, d2 = LDSW(F.o_rank, 3u * N + v) & 0xFFFFu;
            lastrank[v] = (uint16_t)max(max(a, b2), max(c2, d2));
        }
        for (uint32_t i = tid; i < B.nwords; i += NT) {
            uint32_t bits = 0;
            for (uint32_t k = 0; k < 8 && (i * 8 + k) < (N + 3) / 4; ++k) {
                const uint32_t w = LDSW(F.o_tb, i * 8 + k);
                bits |= ((w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u)) << (4 * k);
            }
            B.touchbits[(size_t)r * B.nwords + i] = bits;
        }
        if (lane == 0) B.pcount[(size_t)r * 4 + wave] = pcount;
    }
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

// One launch = the diagonal sweep (and, if asked for, the directed loop behind it) of every replica: the first of the two
// launches of a timestep (isingmc_hip.hip run()), for the geometry above.  PHASE only tags the symbol (see sweep_kernel).
template <int K, int PHASE, bool LABEL, bool COMPACT>
__global__ __launch_bounds__(256, 4) void sweep_fast_kernel(DevBatch B, SweepArgs A) {
    static_assert(!(LABEL && COMPACT), "two alternative hand-overs to the cluster update");
    constexpr int W = 4, NT = W * 64;
    Lds<W> L;
    L.carve(B.N, B.nwords, B.lds_ufcap, B.E, B.has_long);
    const FastLds F = fast_carve<W>(L, B, LABEL);
    const int tid = threadIdx.x;
    const uint32_t r = blockIdx.x;
#ifdef SSE_WG_TIMELINE // diagnostic builds: when and where every workgroup ran (absolute 100-MHz ticks, HW_ID)
    if (tid == 0 && (B.dbg_flags & 32u)) { B.dbg[(size_t)r * 16 + 0] = __builtin_amdgcn_s_memrealtime(); B.dbg[(size_t)r * 16 + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); B.dbg[(size_t)r * 16 + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }
#endif
    const double beta = A.beta ? A.beta[r] : 0.0;
    for (uint32_t i = tid; i < B.nwords; i += NT) LDSW(L.o_state, i) = B.state[(size_t)r * B.nwords + i];
    for (uint32_t i = tid; i < B.Nb; i += NT) LDSW(F.o_tab, i) = fast_entry(B, i, i < B.E ? B.edges_compact[i] : 0u);
    for (uint32_t i = tid; i < 2 * SSE_MAX_CHUNKS; i += NT) LDSW(L.o_chn, i) = B.chunks[(size_t)r * 2 * SSE_MAX_CHUNKS + i];
    if (tid < 4) {
        const double beta_nb = beta * (double)B.Nb;
        const double w = tid == 0 ? B.wJ : (tid == 1 ? B.gamma : (tid == 2 ? B.wh : 0.0));
        const double num = beta_nb * w; // the general pass' nbond, then scaled by exact powers of two
        *reinterpret_cast<double *>(&lds_raw[F.o_nb + 4u * (uint32_t)tid]) = num * 4294967296.0;
        *reinterpret_cast<double *>(&lds_raw[F.o_nb + 4u * (uint32_t)tid + 2u]) = num * (1.0 / 4294967296.0);
    }
    __syncthreads();
    int n = (int)B.n[r], ntrans = (int)B.ntrans[r];
    uint32_t M = B.cutoff[r], err = B.err[r], gr = 0, last_out = 0;
    uint64_t epoch = B.epoch[r];
    uint64_t a4 = 0, a5 = 0;
    // pending cluster flips of this replica: applied by the first diagonal pass of the launch (the host passes defer_flips only
    // to launches that start with one)
    const uint32_t fbmask = (A.defer_flips && B.flipb && (A.domask & SSE_DO_DIAG) && !err && B.pend[r]) ? 0xFFu : 0u;
    for (uint64_t step = 0; step < A.nsteps; ++step) {
        if (err) break;
        if (A.domask & SSE_DO_DIAG) {
            const Rng rng = make_rng(B, r, epoch);
            diagonal_fast<K, LABEL, COMPACT>(B, L, F, r, rng, beta, M, n, ntrans, gr, step == 0 ? fbmask : 0u);
            epoch++;
            a5 += M;
            if (A.domask & SSE_DO_GROW) { // qmc_ising.rs:786, qmc_runner.rs:197
                const uint32_t want = (uint32_t)n + (uint32_t)n / 2u;
                if (want > M) { if (want > B.cap) { err = 1u; break; } M = want; }
            }
        }
        if (A.domask & SSE_DO_LOOP) {
            // the directed loop decodes through the compact edge table, which shares its LDS with the (now dead) diagonal tables
            __syncthreads();
            for (uint32_t i = tid; i < B.E; i += NT) LDSW(L.o_edges, i) = B.edges_compact[i];
            __syncthreads();
            const Rng rng = make_rng(B, r, epoch);
            last_out = loop_pass<W, true>(B, L, r, rng, M, n, gr, err);
            epoch++;
            a4 += last_out;
            if (err) break;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < B.nwords; i += NT) B.state[(size_t)r * B.nwords + i] = LDSW(L.o_state, i);
    for (uint32_t i = tid; i < 2 * SSE_MAX_CHUNKS; i += NT) B.chunks[(size_t)r * 2 * SSE_MAX_CHUNKS + i] = LDSW(L.o_chn, i);
    if (tid == 0) {
        B.n[r] = (uint32_t)n; B.ntrans[r] = (uint32_t)ntrans; B.cutoff[r] = M; B.err[r] = err; B.epoch[r] = epoch;
        if (fbmask && A.nsteps) B.pend[r] = 0u; // the diagonal pass rewrote the whole string with the flips applied
        if (A.out_u32) A.out_u32[r] = last_out;
        uint64_t *acc = B.acc + (size_t)B.acc_row[r] * 8;
        acc[4] += a4; acc[5] += a5;
        // the labelling describes the string as the very next update finds it (cluster ids must fit 16 bits)
        if constexpr (LABEL) B.lite_epoch[r] = (!err && A.nsteps == 1 && B.N + (uint32_t)ntrans <= 65535u) ? epoch : ~0ull;
        if constexpr (COMPACT) B.cops_epoch[r] = (!err && A.nsteps == 1) ? epoch : ~0ull; // the dense list describes the string as the very next update finds it
#ifdef SSE_WG_TIMELINE
        if (B.dbg_flags & 32u) B.dbg[(size_t)r * 16 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    }
}

} // namespace sse
