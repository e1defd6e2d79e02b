// sse_cluster.hip.h — the cluster update of the headline geometry as its own kernel, written for LDS traffic and
// instruction count (round 3).
//
// Same algorithm, same ids, same Philox counters and bit-identical results as sse::cluster_pass<16, K, CL = true, UF_GLOBAL = false>
// (sse_device.hip.h, which stays the general implementation: every other geometry, HBM union-find, HBM tables, generic
// interactions).  Reference: ClusterUpdater::flip_each_cluster_rng (qmc_traits/cluster.rs:36-172) + expand_whole_cluster
// (:193-271), the longitudinal weight function of qmc_ising.rs:759-775, then the free-spin step and the sampling of
// qmc_ising.rs:780-786 / qmc_stepper.rs:149-161.
//
// Why a second implementation.  Measured on MI355X (profiles/r02_*), the general off-diagonal kernel spends 179 vector + 112
// scalar + 35 LDS instructions per 64-slot row in its build scan, holds four cluster_pass variants in one symbol (128 VGPRs,
// 640 scalar spills, 72 B scratch) and keeps 8 B of cached ids per slot in HBM.  This kernel holds ONE path:
//   * one packed 32-bit LDS entry per bond, index = op word >> 4 (entry 0 = empty slot): both variables, "two-site", "cut"
//     (transverse-field op: the sign bit, so the cut mask is one signed compare), "longitudinal".  Empty slots and one-variable
//     bonds name the same variable twice / a dummy variable N, so that nothing below is predicated on the kind of the slot;
//   * per wave ONE packed 32-bit entry per variable: current segment id (16) | in-row cut marker (8) | touched (1).  A leg's
//     lookup is one ds_or_rtn_b32 (returns id + marker, sets the touched bit): two LDS round trips per row instead of the
//     four sub-dword reads + two touched-byte stores of the general scan (the scan is bound by LDS bank conflicts of random
//     full-wave accesses once the instruction count is down);
//   * the table holds the segment ID itself (placeholder ids are written at initialisation), not a rank to be converted;
//   * every slot keeps ONE 16-bit id in HBM (B.segs, two rows per dword): a two-site op's legs are in one cluster once its
//     union is done, a cut's own id is its dense number, recomputed by the apply pass from the row's cut mask.  HBM traffic of the
//     update: 4 + 2 (build) + 4 + 2 + 4 (apply) = 16 B per slot against 20 B before and 12 B algorithmic;
//   * the apply pass runs over the same wave -> range partition as the build: a wave only re-reads ids it stored itself.
// Replicas whose ids do not fit the LDS union-find of this launch (or that hold no op / no cut) are left untouched and flagged
// in B.aux; the host follows up with the general kernel restricted to flagged replicas (isingmc_hip.hip run()).
#pragma once

namespace sse {

#define SSE_CLW 16                  // waves per replica
#define SSE_CL_MAX_VARS 4095u       // variable fields of a bond entry are 13 bits, N itself names the dummy variable
// packed bond entry: a [0,13) | c [13,26) (= a for one-variable bonds, both = N for the empty slot) | two-site (bit 28) |
// longitudinal (bit 29) | cut (bit 31)
#define SSE_CLE_TWO (1u << 28)
#define SSE_CLE_LONG (1u << 29)
#define SSE_CLE_CUT (1u << 31)
#define SSE_CL_TOUCHED (1u << 24)

// LDS accesses of the hot loops by ABSOLUTE LDS byte address held in a 32-bit register (bases below include the address of
// lds_raw): written as pointer arithmetic on lds_raw every access costs an extra v_add of the symbol's (link-time) address
typedef __attribute__((address_space(3))) uint32_t sse_lds_u32;
typedef __attribute__((address_space(3))) uint16_t sse_lds_u16;
typedef __attribute__((address_space(3))) uint8_t sse_lds_u8;
#define LDS32B(a) (*(sse_lds_u32 *)(uintptr_t)(a))
#define LDS16B(a) (*(sse_lds_u16 *)(uintptr_t)(a))
#define LDS8B(a) (*(sse_lds_u8 *)(uintptr_t)(a))
__device__ __forceinline__ void lds_cas32(uint32_t addr, uint32_t expect, uint32_t val) { // ds_cmpst_b32, result unused
    (void)__hip_atomic_compare_exchange_strong((sse_lds_u32 *)(uintptr_t)addr, &expect, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct ClLds { // word offsets into lds_raw
    uint32_t o_tab;    // [Nb + 1]   packed bond entries, at word 0 (the index is the op word >> 4)
    uint32_t o_state;  // [nwords]
    uint32_t o_touch;  // [nwords]   bit v: some op acts on variable v
    uint32_t o_misc;   // [16]
    uint32_t o_chn;    // [SSE_MAX_CHUNKS]
    uint32_t o_chtr;   // [SSE_MAX_CHUNKS]
    uint32_t o_ent;    // [16][N + 1] per (wave, variable): id | marker << 16 | touched << 24; after the joins: root list + flip bits
    uint32_t o_frozen, o_froot; // [ufwords] (h != 0)
    uint32_t o_parent; // [ufcap] u16
    __device__ __forceinline__ void carve(uint32_t N, uint32_t nwords, uint32_t Nb, uint32_t ufcap, bool has_long) {
        uint32_t base = 0;
        o_tab = base; base += Nb + 1u;
        o_state = base; base += nwords;
        o_touch = base; base += nwords;
        o_misc = base; base += 16u;
        o_chn = base; base += SSE_MAX_CHUNKS;
        o_chtr = base; base += SSE_MAX_CHUNKS;
        o_ent = base; base += (uint32_t)SSE_CLW * (N + 1u);
        o_frozen = base; base += has_long ? (ufcap + 31u) / 32u : 0u;
        o_froot = base; base += has_long ? (ufcap + 31u) / 32u : 0u;
        o_parent = base;
    }
};
// words in front of the parent table (host: LDS planning)
static inline __host__ __device__ size_t cl_fixed_words(uint32_t N, uint32_t nwords, uint32_t Nb) {
    return (size_t)Nb + 1 + 2 * (size_t)nwords + 16 + 2 * SSE_MAX_CHUNKS + (size_t)SSE_CLW * (N + 1);
}

__device__ __forceinline__ uint32_t cl_entry(const DevBatch &B, uint32_t b, uint32_t ce) {
    if (b < B.E) return (ce & SSE_CE_VAR_MASK) | (((ce >> 15) & SSE_CE_VAR_MASK) << 13) | SSE_CLE_TWO;
    const uint32_t s1 = b - B.E;
    if (s1 < B.N) return s1 | (s1 << 13) | SSE_CLE_CUT;
    const uint32_t v = s1 - B.N;
    return v | (v << 13) | SSE_CLE_LONG;
}

// union on trees that only the calling wave touches (see uf_union_wave); returns the surviving root.  Both walks to the roots
// advance together (two independent LDS reads per step instead of two walks one after the other), and the two start nodes are
// re-pointed at their roots (they are not roots themselves when they differ from them)
__device__ __forceinline__ uint32_t cl_union_wave(const UFA<false> &uf, uint32_t a, uint32_t b) {
    for (;;) {
        uint32_t xa = a, xb = b, pa = uf.get(a), pb = uf.get(b);
        while ((pa != xa) | (pb != xb)) { xa = pa; xb = pb; pa = uf.get(xa); pb = uf.get(xb); } // (a root is its own parent: it stays put)
        if (a != xa) uf.set(a, xa);
        if (b != xb) uf.set(b, xb);
        if (xa == xb) return xa;
        const uint32_t lo = min(xa, xb), hi = max(xa, xb);
        uf.set(hi, lo);
        SSE_WAVE_FENCE();
        if (uf.get(hi) == lo) return lo;
        a = xa; b = xb;
    }
}

// One launch = the cluster update (+ free spins + sampling) of every replica: the second launch of a split timestep.
// PHASE only tags the symbol (see sweep_kernel).
#ifndef SSE_CL_FIND_LEVELS
#define SSE_CL_FIND_LEVELS 4 // hops of the straight-line find in front of a union
#endif
template <int K, bool HAS_LONG, int PHASE>
__global__ __launch_bounds__(SSE_CLW * 64, 4) void cluster_kernel(DevBatch B, SweepArgs A) {
    constexpr int W = SSE_CLW, NT = W * 64;
    constexpr uint32_t TS = 64u * K;
    static_assert(K == 2 || K == 4, "ids are stored two rows per dword");
    ClLds L;
    L.carve(B.N, B.nwords, B.Nb, B.lds_ufcap, HAS_LONG);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r = blockIdx.x;
    const uint32_t N = B.N, nwords = B.nwords;
    const int n = (int)B.n[r];
    const uint32_t C = B.ntrans[r], M = B.cutoff[r], err = B.err[r];
    if (err) return;                                  // sticky error: the general kernel would not touch the replica either
    const uint32_t S = N + C + (uint32_t)(W - 1) * N; // ids: initial segments, cuts, range-boundary placeholders
    if (n == 0 || C == 0u || S >= B.lds_ufcap || S >= 65535u) { // not this kernel's case: the general one follows up (id S itself is the
                                                               // null id of empty slots: a table entry of its own, its flip bit stays 0)
        if (tid == 0) B.aux[r] = 1u;
        return;
    }
    UFA<false> uf;
    uf.gparent = nullptr; uf.gfrozen = nullptr; uf.gfroot = nullptr;
    uf.o_parent = L.o_parent; uf.o_frozen = L.o_frozen; uf.o_froot = L.o_froot;
    uint64_t epoch = B.epoch[r];
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    uint32_t *ids = B.segs + (size_t)r * B.stride;    // 16-bit ids, rows 2k and 2k+1 share dword (p >> 1 rounded to the pair) + lane
    // deferred flips (A.defer_flips): the row masks live behind the ids (second half of the scratch row: 16 B per 64 slots)
    const bool DEFER = A.defer_flips != 0u && B.flipb != nullptr;
    uint4 *msk = reinterpret_cast<uint4 *>(ids + B.stride / 2u);

    SSE_STAMP_INIT;
    // ---- initialisation ----
    for (uint32_t i = tid; i < nwords; i += NT) { LDSW(L.o_state, i) = B.state[(size_t)r * nwords + i]; LDSW(L.o_touch, i) = 0u; }
    for (uint32_t i = tid; i <= B.Nb; i += NT)
        LDSW(L.o_tab, i) = i == 0u ? (N | (N << 13)) : cl_entry(B, i - 1u, i - 1u < B.E ? B.edges_compact[i - 1u] : 0u);
    for (uint32_t i = tid; i < 2 * SSE_MAX_CHUNKS; i += NT) LDSW(L.o_chn, i) = B.chunks[(size_t)r * 2 * SSE_MAX_CHUNKS + i];
    if (tid == 0) { LDSW(L.o_misc, MISC_NCLUST) = 0u; LDSW(L.o_misc, MISC_ANYFROZEN) = 0u; LDSW(L.o_misc, MISC_LOOP_A) = 0u; }
    // per-wave tables: the segment a variable is in when the wave's range begins = the placeholder id of (wave, variable)
    for (uint32_t i = tid; i < (uint32_t)W * (N + 1u); i += NT) {
        const uint32_t w2 = i / (N + 1u), v = i - w2 * (N + 1u);
        LDSW(L.o_ent, i) = v == N ? S : (w2 == 0u ? v : N + C + (w2 - 1u) * N + v); // (the dummy variable of empty slots carries the null id S)
    }
    for (uint32_t i = tid; i < S / 2u + 1u; i += NT) LDSW(L.o_parent, i) = (2u * i) | ((2u * i + 1u) << 16); // parent[i] = i for i <= S
    if constexpr (HAS_LONG) for (uint32_t i = tid; i < (S + 31u) / 32u; i += NT) uf.bits_clear(i);
    __syncthreads();

    // ---- this wave's range of chunks, the id of its first cut ----
    const uint32_t used = (M + B.CH - 1u) / B.CH;
    const uint32_t q = (used + W - 1u) / W;
    const uint32_t c0 = min((uint32_t)wave * q, used), c1 = min(c0 + q, used);
    uint32_t cutbase = 0;
    for (uint32_t c = lane; c < c0; c += 64) cutbase += LDSW(L.o_chtr, c);
    for (int off = 32; off > 0; off >>= 1) cutbase += __shfl_xor(cutbase, off);
    cutbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)cutbase);
    const uint32_t pbeg = c0 * B.CH, pend = min(c1 * B.CH, M); // whole tiles except at the end of the string, where the row holds zeros
    const uint32_t lds0 = (uint32_t)(uintptr_t)(sse_lds_u32 *)lds_raw; // LDS address of the dynamic region
    const uint32_t ent_b = lds0 + 4u * (L.o_ent + (uint32_t)wave * (N + 1u)); // LDS address of this wave's table
    const uint32_t par_b = lds0 + 4u * L.o_parent;

    SSE_STAMP(0);
    // ---- build: label every leg, union through the two-site ops (cluster.rs:193-271) ----
    // The id a table entry holds is a REPRESENTATIVE of the segment's cluster as this wave knows it, not necessarily the
    // segment's own id: after a union both legs' entries are rewritten with the surviving root (compare-and-store against the
    // value the lane saw, so that a later cut's fresh id is never overwritten).  A two-site op whose legs already carry the same
    // representative — most of them, clusters being long-lived — then costs no union-find access at all, and the others start
    // their finds one hop from a root.  Any member of the cluster serves the apply pass and the range joins equally well.
    {
        uint32_t cutnext = N + cutbase; // id of the next cut of this range
        uint32_t wnext[K], idpend[K / 2];
        uint4 mpend[K]; // deferred flips: the cut mask and the two-site mask of every row (the same in all lanes; lane 0 stores them)
#pragma unroll
        for (int j = 0; j < K; ++j) { wnext[j] = row_ld(ops, pbeg + (uint32_t)(j * 64 + lane)); mpend[j] = make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int j = 0; j < K / 2; ++j) idpend[j] = 0u;
        uint32_t pprev = pbeg;
        for (uint32_t p0 = pbeg; p0 < pend; p0 += TS) {
            uint32_t e[K];
            {
                uint32_t word[K];
#pragma unroll
                for (int j = 0; j < K; ++j) word[j] = wnext[j];
                // the ids of the previous tile are stored here, in front of the prefetch: a wave's memory operations complete in
                // order, so waiting for the prefetched words at the top of the next tile then waits for stores that have had a
                // whole tile to complete, not for stores issued a moment ago (first tile: a dummy store that the next one overwrites)
#pragma unroll
                for (int j = 0; j < K; j += 2) if (!SSE_DBG(B, 1u)) row_st(ids, (pprev >> 1) + (uint32_t)(j * 32 + lane), idpend[j / 2]);
                if (DEFER) { if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < K; ++j) msk[(pprev >> 6) + (uint32_t)j] = mpend[j];
                } }
                pprev = p0;
                const uint32_t pn0 = p0 + TS < pend ? p0 + TS : p0;
#pragma unroll
                for (int j = 0; j < K; ++j) wnext[j] = row_ld(ops, pn0 + (uint32_t)(j * 64 + lane));
#pragma unroll
                for (int j = 0; j < K; ++j) e[j] = LDS32B(lds0 + 4u * (word[j] >> 4)); // entry (word >> 4) of the table at LDS word 0
            }
            uint32_t ua[K], uc[K], adra[K], adrc[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t aa = ent_b + 4u * (e[j] & 0x1FFFu), ac = ent_b + 4u * ((e[j] >> 13) & 0x1FFFu);
                adra[j] = aa; adrc[j] = ac;
                const bool iscut = (int32_t)e[j] < 0;
                const uint64_t cutm = sse_ballot(iscut);
                const uint32_t kown = __builtin_amdgcn_mbcnt_hi((uint32_t)(cutm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cutm, 0u));
                // In-row ordering: the cut lanes publish 1 + their rank inside the row in the marker byte of their variable's entry,
                // every lane reads once; a cut on the leg's variable precedes the lane iff its rank is below the lane's own count.
                if (iscut) LDS8B(aa + 2u) = (uint8_t)(kown + 1u);
                SSE_WAVE_FENCE();
                const uint32_t ea = LDS32B(aa), ec = LDS32B(ac);
                LDS8B(aa + 3u) = (uint8_t)1; LDS8B(ac + 3u) = (uint8_t)1; // touched flag (byte 3): plain stores with an immediate offset, nothing waits for them
                const uint32_t qa = (ea >> 16) & 0xFFu, qc = (ec >> 16) & 0xFFu;
                uint32_t seg_a = ea & 0xFFFFu, seg_c = ec & 0xFFFFu;
                const uint64_t dup = cutm & sse_ballot(qa != kown + 1u);
                if (!dup) {
                    seg_a = ((qa - 1u) < kown) ? cutnext + (qa - 1u) : seg_a; // qa == 0: no cut on the variable in this row
                    seg_c = ((qc - 1u) < kown) ? cutnext + (qc - 1u) : seg_c;
                    SSE_WAVE_FENCE();
                    if (iscut) LDS32B(aa) = (cutnext + kown) | SSE_CL_TOUCHED; // the segment the cut opens; clears the marker
                } else { // two cuts of this row on one variable (rare): lane order decides
                    const uint32_t va = e[j] & 0x1FFFu, vc = (e[j] >> 13) & 0x1FFFu;
                    bool lastcut = iscut;
                    uint64_t m = cutm;
                    uint32_t idL = cutnext;
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
                    if (iscut) LDS8B(aa + 2u) = (uint8_t)0;
                    if (iscut & lastcut) LDS16B(aa) = (uint16_t)(cutnext + kown); // the last cut wins
                    SSE_WAVE_FENCE();
                }
                if constexpr (HAS_LONG)
                    if (e[j] & SSE_CLE_LONG) uf.frozen_or(seg_a >> 5, 1u << (seg_a & 31)); // qmc_ising.rs:759-775
                cutnext += (uint32_t)popc64(cutm);
                if (DEFER) { // (without a longitudinal field every op that is not a cut is a two-site op: the cut mask alone will do)
                    mpend[j].x = (uint32_t)cutm; mpend[j].y = (uint32_t)(cutm >> 32);
                    if constexpr (HAS_LONG) { const uint64_t twom = sse_ballot((e[j] & SSE_CLE_TWO) != 0u); mpend[j].z = (uint32_t)twom; mpend[j].w = (uint32_t)(twom >> 32); }
                }
                ua[j] = seg_a; uc[j] = seg_c; // (one-variable ops and empty slots: seg_c == seg_a, no union below)
            }
            // one 16-bit id per slot for the apply pass: the input leg's representative (a two-site op's other leg is in the same
            // cluster once its union below is done; a cut's own id is recomputed from the cut masks)
#pragma unroll
            for (int j = 0; j < K; j += 2) idpend[j / 2] = ua[j] | (ua[j + 1] << 16);
            // Unions: only the lanes whose two legs carry different representatives take part — a handful per tile once the
            // representatives have settled —, row by row, under their own exec mask (the union-find reads of a full wave at random
            // addresses would cost the LDS more than the whole scan).  Parents, grandparents (root test), link + read-back (two lanes
            // hooking one root); the trees a wave touches during the scan are private to it (own cuts, own placeholders): plain
            // stores, no atomics.  Anything deeper goes through the serial routine.
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const bool need = ua[j] != uc[j];
                if (!sse_any(need) || SSE_DBG(B, 2u)) continue; // wave-uniform
                if (need) {
                    // SSE_CL_FIND_LEVELS hops up from each representative, straight-line (a representative is a root or close to one: it was
                    // a root when it was written); nodes found below a root are re-pointed at it on the way (they are not roots, so
                    // no link of this batch can be undone by that)
                    const uint32_t pa = LDS16B(par_b + 2u * ua[j]), pc = LDS16B(par_b + 2u * uc[j]);
                    uint32_t ga = LDS16B(par_b + 2u * pa), gc = LDS16B(par_b + 2u * pc);
                    uint32_t ta = LDS16B(par_b + 2u * ga), tc = LDS16B(par_b + 2u * gc);
#pragma unroll
                    for (int lv = 3; lv < SSE_CL_FIND_LEVELS; ++lv) { // (further levels: the candidate root moves one hop up)
                        const uint32_t na = LDS16B(par_b + 2u * ta), nc = LDS16B(par_b + 2u * tc);
                        ga = ta; gc = tc; ta = na; tc = nc;
                    }
                    const bool found = (ta == ga) & (tc == gc); // ga / gc are roots (also when the chain is shorter: a root is its own parent)
                    const bool differ = ga != gc;
                    const bool link = found & differ;
                    uint32_t lo = min(ga, gc);
                    const uint32_t hi = max(ga, gc);
                    if (found & (pa != ga)) { LDS16B(par_b + 2u * ua[j]) = (uint16_t)ga; }
                    if (found & (pc != gc)) { LDS16B(par_b + 2u * uc[j]) = (uint16_t)gc; }
                    if (link) LDS16B(par_b + 2u * hi) = (uint16_t)lo;
                    SSE_WAVE_FENCE();
                    const uint32_t chk = LDS16B(par_b + 2u * hi);
#ifdef SSE_CL_UNION_COUNTERS // (a diagnostic build of its own: the counters cost the union block a quarter of its time)
                    if (wave == 3) { // why rows reach the serial routine
                        const uint64_t nf = sse_ballot(!found), cf = sse_ballot(found & link & (chk != lo)), nd = sse_ballot(true);
                        if (lane == (int)(__ffsll((long long)nd) - 1)) {
                            B.dbg[(size_t)r * 16 + 8] += 1; B.dbg[(size_t)r * 16 + 12] += (uint64_t)popc64(nd);
                            if (nf) B.dbg[(size_t)r * 16 + 9] += 1;
                            if (!nf && cf) B.dbg[(size_t)r * 16 + 10] += 1;
                            B.dbg[(size_t)r * 16 + 13] += (uint64_t)popc64(nf); B.dbg[(size_t)r * 16 + 14] += (uint64_t)popc64(cf);
                        }
                    }
#endif
                    if ((!found | (link & (chk != lo))) && !SSE_DBG(B, 16u)) lo = cl_union_wave(uf, pa, pc);
                    // both legs' entries now name the surviving root (or the common parent found one hop up)
                    lds_cas32(adra[j], ua[j] | SSE_CL_TOUCHED, lo | SSE_CL_TOUCHED);
                    lds_cas32(adrc[j], uc[j] | SSE_CL_TOUCHED, lo | SSE_CL_TOUCHED);
                }
            }
        }
        if (pbeg < pend) {
#pragma unroll
            for (int j = 0; j < K; j += 2) row_st(ids, (pprev >> 1) + (uint32_t)(j * 32 + lane), idpend[j / 2]);
            if (DEFER) { if (lane == 0) {
#pragma unroll
                for (int j = 0; j < K; ++j) msk[(pprev >> 6) + (uint32_t)j] = mpend[j];
            } }
        }
    }
    __syncthreads();
    SSE_STAMP(1);

    // ---- join the ranges (the segment a worldline is in when range w ends continues into the placeholder of range w + 1; the
    // last range wraps into [0, N): cluster.rs:223-242), collect the touched bits ----
    for (uint32_t v = tid; v < N; v += NT) {
        uint32_t t = 0;
#pragma unroll 4
        for (uint32_t w2 = 0; w2 < (uint32_t)W; ++w2) {
            const uint32_t x = LDSW(L.o_ent, w2 * (N + 1u) + v);
            t |= x;
            const uint32_t seg_end = x & 0xFFFFu;
            const uint32_t nxt = (w2 + 1u == (uint32_t)W) ? v : N + C + w2 * N + v;
            if (seg_end != nxt) uf_union(uf, seg_end, nxt);
        }
        if (t >> 24) atomicOr(&LDSW(L.o_touch, v >> 5), 1u << (v & 31));
    }
    __syncthreads();
    SSE_STAMP(2);
    // ---- flatten: parent[i] := exact root; frozen marks move to the roots ----
    for (uint32_t i = tid; i < S; i += NT) {
        const uint32_t root = uf_find_ro(uf, i);
        uf.set(i, root);
        if constexpr (HAS_LONG)
            if ((uf.frozen_get(i >> 5) >> (i & 31)) & 1u) { uf.froot_or(root >> 5, 1u << (root & 31)); LDSW(L.o_misc, MISC_ANYFROZEN) = 1u; }
    }
    // the per-wave tables are dead: their words now hold the flip bits (S bits) and, behind them, the list of roots (u16)
    const uint32_t o_bits = L.o_ent, o_list = L.o_ent + (S + 31u) / 32u;
    const uint32_t list_cap = 2u * ((uint32_t)W * (N + 1u) - (S + 31u) / 32u);
    for (uint32_t i = tid; i < (S + 31u) / 32u; i += NT) LDSW(o_bits, i) = 0u;
    __syncthreads();
    SSE_STAMP(3);
    // ---- coins: one Philox draw per root (= per cluster, keyed by the canonical label = smallest id), cluster.rs:111-137 ----
    const Rng rng = make_rng(B, r, epoch);
    uint32_t myclusters = 0;
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
            if (isroot && pos < list_cap) LDSH(o_list, pos) = (uint16_t)i;
        }
    }
    __syncthreads();
    const uint32_t nroots = LDSW(L.o_misc, MISC_LOOP_A);
    if (nroots <= list_cap) { // uniform: every thread read the same counter
        for (uint32_t k = tid; k < nroots; k += NT) {
            const uint32_t root = LDSH(o_list, k);
            const uint4 o = rng.draw(SSE_TAG_CLUSTER, root);
            const uint32_t isfrozen = HAS_LONG ? (uf.froot_get(root >> 5) >> (root & 31)) & 1u : 0u;
            if (!isfrozen && u01(o.x) < A.prob) atomicOr(&LDSW(o_bits, root >> 5), 1u << (root & 31));
        }
        __syncthreads();
        for (uint32_t i = tid; i < S; i += NT) {
            const uint32_t root = uf.get(i);
            uf.set(i, (LDSW(o_bits, root >> 5) >> (root & 31)) & 1u);
        }
    } else { // more roots than the list holds: every id draws the coin of its root itself (same results)
        for (uint32_t i = tid; i < S; i += NT) {
            const uint32_t root = uf.get(i);
            const uint4 o = rng.draw(SSE_TAG_CLUSTER, root);
            const uint32_t isfrozen = HAS_LONG ? (uf.froot_get(root >> 5) >> (root & 31)) & 1u : 0u;
            uf.set(i, (!isfrozen && u01(o.x) < A.prob) ? 1u : 0u);
        }
    }
    if (tid == 0) uf.set(S, 0u); // the null id never flips
    {
        uint32_t c = myclusters;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if (lane == 0 && c) atomicAdd(&LDSW(L.o_misc, MISC_NCLUST), c);
    }
    __syncthreads();
    SSE_STAMP(4);

    // ---- apply (cluster.rs:139-167): input bits flip with the incoming segment, output bits with the outgoing one ----
    if (DEFER) {
        // Deferred: the op-string is not touched.  Every slot gets the xor mask of its four state bits as one byte (0 for an empty
        // slot); the diagonal pass of the next timestep applies it while it streams the string (sse_fast.hip.h), any other reader
        // goes through materialize_kernel first.  Reads 2 B of ids + 16 B of row masks per row, writes 1 B per slot: a quarter
        // of the traffic of the in-place pass below.
        uint32_t cutnext = N + cutbase;
        uint8_t *fb = B.flipb + (size_t)r * B.stride;
        // The loop body is short, so the loads run D tiles ahead (one tile per memory round trip would leave the pass waiting):
        // per tile two dwords of ids per lane and the 64 bytes of row masks, one dword per lane 0..15 (read back with v_readlane)
        constexpr int D = 3;
        const uint32_t *mskw = reinterpret_cast<const uint32_t *>(msk);
        uint32_t qid[D][K / 2], qmk[D], bpend[K];
        auto issue = [&](int slot, uint32_t pt) { // loads of the tile at pt (clamped to the last tile of the range: values unused beyond it)
            const uint32_t pc = pt < pend ? pt : pbeg;
#pragma unroll
            for (int j = 0; j < K; j += 2) qid[slot][j / 2] = row_ld(ids, (pc >> 1) + (uint32_t)(j * 32 + lane));
            qmk[slot] = row_ld(mskw, (pc >> 6) * 4u + (uint32_t)(lane & 15));
        };
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d, pbeg + (uint32_t)d * TS);
#pragma unroll
        for (int j = 0; j < K; ++j) bpend[j] = 0u;
        uint32_t pprev = pbeg;
        bool first = true;
        for (uint32_t p0 = pbeg; p0 < pend; p0 += TS) {
            uint32_t id2[K / 2];
#pragma unroll
            for (int j = 0; j < K / 2; ++j) id2[j] = qid[0][j];
            const uint32_t mkv = qmk[0];
#pragma unroll
            for (int d = 0; d + 1 < D; ++d) {
#pragma unroll
                for (int j = 0; j < K / 2; ++j) qid[d][j] = qid[d + 1][j];
                qmk[d] = qmk[d + 1];
            }
            if (!first) {
#pragma unroll
                for (int j = 0; j < K; ++j) fb[pprev + (uint32_t)(j * 64 + lane)] = (uint8_t)bpend[j];
            }
            first = false;
            pprev = p0;
            issue(D - 1, p0 + (uint32_t)D * TS);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint64_t cutm = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mkv, 4 * j) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mkv, 4 * j + 1) << 32);
                const uint32_t own = __builtin_amdgcn_mbcnt_hi((uint32_t)(cutm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cutm, cutnext));
                const uint32_t idin = (j & 1) ? (id2[j / 2] >> 16) : (id2[j / 2] & 0xFFFFu);
                const uint32_t f1 = LDS16B(par_b + 2u * idin), f2 = LDS16B(par_b + 2u * own);
                uint32_t m_other = (0u - f1) & 0xFu; // two-site: all four bits follow the one cluster
                if constexpr (HAS_LONG) { // longitudinal op: both bits of its variable
                    const uint64_t twom = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mkv, 4 * j + 2) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mkv, 4 * j + 3) << 32);
                    m_other &= sel64(twom, vgpr_copy_u32(0xFu), vgpr_copy_u32(0x5u));
                }
                bpend[j] = sel64(cutm, f1 | (f2 << 2), m_other);
                cutnext += (uint32_t)popc64(cutm);
            }
        }
        if (pbeg < pend) {
#pragma unroll
            for (int j = 0; j < K; ++j) fb[pprev + (uint32_t)(j * 64 + lane)] = (uint8_t)bpend[j];
        }
    } else
    {
        uint32_t cutnext = N + cutbase;
        const uint32_t E1 = B.E + 1u;
        uint32_t wnext[K], inext[K / 2], wpend[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { wnext[j] = row_ld(ops, pbeg + (uint32_t)(j * 64 + lane)); wpend[j] = wnext[j]; }
#pragma unroll
        for (int j = 0; j < K; j += 2) inext[j / 2] = row_ld(ids, (pbeg >> 1) + (uint32_t)(j * 32 + lane));
        uint32_t pprev = pbeg;
        for (uint32_t p0 = pbeg; p0 < pend; p0 += TS) {
            uint32_t word[K], id2[K / 2];
#pragma unroll
            for (int j = 0; j < K; ++j) word[j] = wnext[j];
#pragma unroll
            for (int j = 0; j < K / 2; ++j) id2[j] = inext[j];
            {
                // the previous tile's words are stored in front of this tile's prefetch (see the build loop; first tile: its own
                // words, unchanged)
#pragma unroll
                for (int j = 0; j < K; ++j) row_st(ops, pprev + (uint32_t)(j * 64 + lane), wpend[j]);
                pprev = p0;
                const uint32_t pn0 = p0 + TS < pend ? p0 + TS : p0;
#pragma unroll
                for (int j = 0; j < K; ++j) wnext[j] = row_ld(ops, pn0 + (uint32_t)(j * 64 + lane));
#pragma unroll
                for (int j = 0; j < K; j += 2) inext[j / 2] = row_ld(ids, (pn0 >> 1) + (uint32_t)(j * 32 + lane));
            }
            // straight-line: both flip lookups are issued for every lane (lanes that hold no cut read a valid id, at most the first
            // id of the next range; empty slots carry the null id, whose flip is 0, so their word stays 0 without a test)
            uint32_t f1[K], f2[K];
            uint64_t cutm[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                cutm[j] = sse_ballot(((word[j] >> 4) - E1) < N); // bonds [E, E + N): transverse field
                const uint32_t own = __builtin_amdgcn_mbcnt_hi((uint32_t)(cutm[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cutm[j], cutnext));
                const uint32_t idin = (j & 1) ? (id2[j / 2] >> 16) : (id2[j / 2] & 0xFFFFu);
                f1[j] = LDS16B(par_b + 2u * idin);
                f2[j] = LDS16B(par_b + 2u * own); // the segment a cut opens
                cutnext += (uint32_t)popc64(cutm[j]);
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                // two-site: all four bits follow the one cluster; longitudinal: both bits of its variable; cut: input with the incoming,
                // output with the outgoing segment
                const uint32_t allbits = (((word[j] >> 4) - 1u) < B.E) ? 0xFu : 0x5u;
                const uint32_t m_other = (0u - f1[j]) & allbits;
                const uint32_t m_cut = f1[j] | (f2[j] << 2);
                wpend[j] = word[j] ^ sel64(cutm[j], m_cut, m_other);
            }
        }
        if (pbeg < pend) {
#pragma unroll
            for (int j = 0; j < K; ++j) row_st(ops, pprev + (uint32_t)(j * 64 + lane), wpend[j]);
        }
    }
    // p=0 state follows the initial segment of each touched variable
    for (uint32_t i = tid; i < nwords; i += NT) {
        uint32_t x = 0;
        const uint32_t t = LDSW(L.o_touch, i);
        for (uint32_t j = 0; j < 32 && i * 32 + j < N; ++j) x |= (uf.get(i * 32 + j) & 1u) << j;
        LDSW(L.o_state, i) ^= (x & t);
    }
    __syncthreads();
    SSE_STAMP(5);
    const uint32_t nclusters = LDSW(L.o_misc, MISC_NCLUST);
    epoch++;
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a6 = 0;
    // ---- free spins (qmc_ising.rs:780-784) ----
    if (A.domask & SSE_DO_FREE) {
        const Rng rngf = make_rng(B, r, epoch);
        for (uint32_t i = tid; i < nwords; i += NT) {
            const uint32_t t = LDSW(L.o_touch, i);
            uint32_t s = LDSW(L.o_state, i);
            for (uint32_t j = 0; j < 32 && i * 32 + j < N; ++j)
                if (!((t >> j) & 1u)) {
                    const uint4 o = rngf.draw(SSE_TAG_FREE, i * 32 + j);
                    s = (s & ~(1u << j)) | ((o.x >> 31) << j);
                }
            LDSW(L.o_state, i) = s;
        }
        __syncthreads();
        epoch++;
    }
    // ---- sampling (qmc_stepper.rs:149-161) ----
    if (A.sampling_freq && (A.step0 + 1) % A.sampling_freq == 0) {
        if (tid == 0) LDSW(L.o_misc, MISC_LOOP_A) = 0u;
        __syncthreads();
        uint32_t up = 0;
        for (uint32_t i = tid; i < nwords; i += NT) up += __popc(LDSW(L.o_state, i));
        for (int off = 32; off > 0; off >>= 1) up += __shfl_down(up, off);
        if (lane == 0 && up) atomicAdd(&LDSW(L.o_misc, MISC_LOOP_A), up);
        __syncthreads();
        const long long mag = 2ll * (long long)LDSW(L.o_misc, MISC_LOOP_A) - (long long)N;
        a0 = (uint64_t)n; a1 = 1; a2 = (uint64_t)(mag < 0 ? -mag : mag); a3 = (uint64_t)(mag * mag); a6 = (uint64_t)C;
    }
    for (uint32_t i = tid; i < nwords; i += NT) B.state[(size_t)r * nwords + i] = LDSW(L.o_state, i);
    if (tid == 0) {
        B.epoch[r] = epoch;
        if (DEFER) B.pend[r] = 1u; // (behind the barriers above: every wave's flip bytes are on their way; the kernel boundary orders them)
        if (A.out_u32) A.out_u32[r] = nclusters;
        uint64_t *acc = B.acc + (size_t)B.acc_row[r] * 8;
        acc[0] += a0; acc[1] += a1; acc[2] += a2; acc[3] += a3; acc[4] += (uint64_t)n; acc[6] += a6;
    }
}

} // namespace sse
