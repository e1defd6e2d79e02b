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
    uint32_t end;
};
template <int W>
__device__ __forceinline__ FastLds fast_carve(const Lds<W> &L, const DevBatch &B) {
    FastLds F;
    uint32_t base = (L.o_cur + 3u) & ~3u; // 16-byte aligned: the class constants are read as one b128
    F.o_nb = base; base += 16;
    F.o_tab = base; base += B.Nb;
    F.o_spin = base; base += B.N;
    F.o_dummy = base; base += 64;
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

template <int K>
__device__ __forceinline__ void diagonal_fast(const DevBatch &B, const Lds<4> &L, const FastLds &F, uint32_t r, const Rng &rng, double beta,
                                              uint32_t M, int &n_io, int &ntrans_io, uint32_t &gr) {
    constexpr int W = 4, NT = W * 64;
    constexpr uint32_t TS = (uint32_t)(NT * K);
    static_assert(K == 2 || K == 4, "rows p and p + 64 share one Philox call");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    const uint32_t N = B.N, Nb = B.Nb, E = B.E;
    for (uint32_t i = tid; i < N; i += NT) LDSW(F.o_spin, i) = ((LDSW(L.o_state, i >> 5) >> (i & 31)) & 1u) * 0x01010101u;
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
    for (int j = 0; j < K; ++j) wnext[j] = row_ld(ops, (uint32_t)(wave * 64 * K + j * 64 + lane));
    propagate(wnext, m_later);
    __syncthreads();
    const uint32_t lane2 = 2u * (uint32_t)lane;
    const uint32_t vM = vgpr_copy_u32(M), vM1 = vgpr_copy_u32(M + 1u), vzero = vgpr_copy_u32(0u);

    for (uint32_t tile = 0; tile < ntiles; ++tile) {
        uint32_t word[K];
#pragma unroll
        for (int j = 0; j < K; ++j) word[j] = wnext[j];
        const uint32_t pbase = tile * TS + (uint32_t)(wave * 64 * K + lane);
        {
            const uint32_t pn = (tile + 1 < ntiles ? pbase + TS : pbase);
#pragma unroll
            for (int j = 0; j < K; ++j) wnext[j] = row_ld(ops, pn + (uint32_t)(j * 64));
        }
        const bool partial = tile * TS + TS > M; // wave-uniform: only the last tile can hold slots >= M

        double ua[K], un[K], nb[K];
        uint32_t cbv[K], neww[K], ent[K], bnd[K], rr1[K];
        uint64_t insm[K], remm[K], acc[K];
        // Lane predicates that have to survive the rounds are wave masks (insert candidates, removal candidates, accepted);
        // everything else is recomputed from the op word with integer arithmetic when needed — the pass is short of scalar
        // registers, and a spilled mask costs a v_readlane per half and use.
        // ---- phase 1, all rows at once (nothing here depends on the spin tables): random numbers, bond, packed table entry
        {
            uint4 rnd = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t wd = word[j];
                if ((j & 1) == 0) rnd = rng.draw(SSE_TAG_DIAG, pbase + (uint32_t)(j * 64)); // bit 6 of the slot index is clear on even rows
                const uint32_t r0 = (j & 1) ? rnd.z : rnd.x;
                rr1[j] = (j & 1) ? rnd.w : rnd.y;
                bnd[j] = sel64(sse_ballot(wd != 0u), (wd >> 4) - 1u, __umulhi(r0, Nb));
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
            ua[j] = u;          // insert:  (u 2^-32) * den < num   <=>  u * den < num * 2^32
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
        int tot_all = 0;
        bool first = true;
        for (;;) {
            int wtot = 0;
            bool changed = first;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const double t = (double)(int)(cbv[j] - (uint32_t)npref[j]);
                const uint64_t lt_ins = sse_ballot(ua[j] * t < nb[j]);
                const uint64_t lt_rem = sse_ballot(un[j] < t);
                const uint64_t a = (lt_ins & insm[j]) | (lt_rem & remm[j]);
                changed |= a != acc[j];
                acc[j] = a;
                wtot += popc64(a & insm[j]) - popc64(a & remm[j]);
            }
            const int buf = gr & 1;
            if (lane == 0) { LDSI(L.o_tot, buf * W + wave) = wtot; LDSW(L.o_chg, buf * W + wave) = changed ? 1u : 0u; }
            __syncthreads();
            if (first) {
                // this tile's off-diagonal ops -> copies of the earlier waves (every reader of this tile is done); the next
                // tile's -> copies of the later waves (visible behind the next barrier, before anybody decodes that tile)
                propagate(word, m_earlier);
                if (tile + 1 < ntiles) propagate(wnext, m_later);
            }
            int base = 0; tot_all = 0; uint32_t anychg = 0;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) {
                const int t = __builtin_amdgcn_readfirstlane(LDSI(L.o_tot, buf * W + w2));
                if (w2 < wave) base += t;
                tot_all += t;
                anychg |= (uint32_t)__builtin_amdgcn_readfirstlane((int)LDSW(L.o_chg, buf * W + w2));
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
#pragma unroll
        for (int j = 0; j < K; ++j) {
            row_st(ops, pbase + (uint32_t)(j * 64), sel64(acc[j], neww[j], word[j]));
            const uint64_t im = acc[j] & insm[j], rm = acc[j] & remm[j];
            const uint64_t trm = sse_ballot((ent[j] >> 30) == SSE_FAST_CLASS_G); // the bond at stake is a transverse-field bond
            dn += popc64(im) - popc64(rm);
            dtr += popc64(im & trm) - popc64(rm & trm);
        }
        ntrans += dtr;
        if (lane == 0 && (dtr | dn)) { // the 64*K slots of a wave's share of a tile lie inside one chunk
            const uint32_t ch = (tile * TS + (uint32_t)(wave * 64 * K)) / B.CH;
            if (dn) atomicAdd(&LDSW(L.o_chn, ch), (uint32_t)dn);
            if (dtr) atomicAdd(&LDSW(L.o_chtr, ch), (uint32_t)dtr);
        }
        n_start += tot_all;
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
template <int K, int PHASE>
__global__ __launch_bounds__(256, 4) void sweep_fast_kernel(DevBatch B, SweepArgs A) {
    constexpr int W = 4, NT = W * 64;
    Lds<W> L;
    L.carve(B.N, B.nwords, B.lds_ufcap, B.E, B.has_long);
    const FastLds F = fast_carve<W>(L, B);
    const int tid = threadIdx.x;
    const uint32_t r = blockIdx.x;
    const double beta = A.beta ? A.beta[r] : 0.0;
    for (uint32_t i = tid; i < B.nwords; i += NT) LDSW(L.o_state, i) = B.state[(size_t)r * B.nwords + i];
    for (uint32_t i = tid; i < B.E; i += NT) LDSW(L.o_edges, i) = B.edges_compact[i]; // the directed loop decodes through this table
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
    for (uint64_t step = 0; step < A.nsteps; ++step) {
        if (err) break;
        if (A.domask & SSE_DO_DIAG) {
            const Rng rng = make_rng(B, r, epoch);
            diagonal_fast<K>(B, L, F, r, rng, beta, M, n, ntrans, gr);
            epoch++;
            a5 += M;
            if (A.domask & SSE_DO_GROW) { // qmc_ising.rs:786, qmc_runner.rs:197
                const uint32_t want = (uint32_t)n + (uint32_t)n / 2u;
                if (want > M) { if (want > B.cap) { err = 1u; break; } M = want; }
            }
        }
        if (A.domask & SSE_DO_LOOP) {
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
        if (A.out_u32) A.out_u32[r] = last_out;
        uint64_t *acc = B.acc + (size_t)B.acc_row[r] * 8;
        acc[4] += a4; acc[5] += a5;
    }
}

} // namespace sse
