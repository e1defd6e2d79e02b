// sse_device.hip.h — gfx950 device code of the SSE sweep (workgroup-per-replica design).
//
// One workgroup of W wave64s owns one replica for a whole launch.  The op-string (one u32 per slot,
// include/sse_format.h) streams from HBM in block-tiles of W*64 consecutive slots; everything that
// the reference keeps in per-node linked lists (src/sse/fast_ops.rs:181-190: prev/next p, per-variable
// prev/next) is recomputed on chip by ORDERED SCANS:
//   * in-wave: wave64 ballot + a serial loop over the (few) writer lanes with v_readlane;
//   * across the W waves of a tile: W copies of the per-variable table in LDS; a writer in wave w
//     updates the copies of waves > w before the readers run and the copies of waves <= w after
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

struct BondRec {       // 16 B, one dwordx4 load
    uint32_t a_info;   // var a | (kind|pref) << 29
    uint32_t c;        // second var or SSE_NO_VAR
    double w;          // weight when satisfied: 2|J|, Gamma, 2|h|
};
#define SSE_INFO_SHIFT 29
#define SSE_VAR_MASK 0x1FFFFFFFu

struct DevBatch {
    uint32_t R, N, E, Nb, cap, nwords;
    uint32_t *ops;        // [R][cap]
    uint32_t *state;      // [R][nwords] bit v of word v>>5
    uint32_t *n, *ntrans, *cutoff, *err, *aux;  // [R]
    uint64_t *epoch;      // [R]
    uint64_t *acc;        // [R][8]
    const BondRec *bonds; // [Nb]
    const double *cumw;   // [Nb] heat-bath cumulative weights
    double wtot;
    uint32_t *uf_scratch; // [R][N+cap] union-find fallback in HBM
    uint32_t seed_lo, seed_hi, replica_offset;
    uint32_t lds_ufcap;   // ids that fit the LDS union-find arrays
};

// which primitives a launch runs per step
#define SSE_DO_DIAG 1u
#define SSE_DO_LOOP 2u
#define SSE_DO_CLUSTER 4u
#define SSE_DO_FREE 8u
#define SSE_DO_GROW 16u
#define SSE_DO_HEATBATH 32u

struct SweepArgs {
    const double *beta; // [R]
    uint64_t nsteps;
    uint32_t sampling_freq; // 0 = never sample
    uint32_t domask;
    double prob;
    uint32_t *out_u32; // optional per-replica output (n_clusters / loop length) of the LAST step
};

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

struct Rng {
    uint32_t k0, k1, replica, epoch_lo, epoch_hi24;
    __device__ __forceinline__ uint4 draw(uint32_t tag, uint32_t index) const {
        return philox4x32_10(index, epoch_lo, replica, (tag << 24) | epoch_hi24, k0, k1);
    }
};
__device__ __forceinline__ Rng make_rng(const DevBatch &B, uint32_t r, uint64_t epoch) {
    Rng g;
    g.k0 = B.seed_lo; g.k1 = B.seed_hi; g.replica = B.replica_offset + r;
    g.epoch_lo = (uint32_t)epoch; g.epoch_hi24 = (uint32_t)(epoch >> 32) & 0xFFFFFFu;
    return g;
}
__device__ __forceinline__ double u01(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }

__device__ __forceinline__ uint32_t rec_var(const BondRec &b) { return b.a_info & SSE_VAR_MASK; }
__device__ __forceinline__ uint32_t rec_kind(const BondRec &b) { return (b.a_info >> SSE_INFO_SHIFT) & SSE_BOND_KIND_MASK; }
__device__ __forceinline__ uint32_t rec_pref(const BondRec &b) { return (b.a_info >> (SSE_INFO_SHIFT + 2)) & 1u; }

// matrix element of the shifted bond operator (reference: src/sse/qmc_ising.rs:863-888)
__device__ __forceinline__ double bond_weight(const BondRec &b, uint32_t in, uint32_t out) {
    const uint32_t kind = rec_kind(b), pref = rec_pref(b);
    if (kind == SSE_BOND_TRANSVERSE) return b.w;
    if (in != out) return 0.0;
    uint32_t sat = (kind == SSE_BOND_TWO_SITE) ? (uint32_t)(((in & 1u) == ((in >> 1) & 1u)) == (pref != 0u))
                                               : (uint32_t)((in & 1u) == pref);
    return sat ? b.w : 0.0;
}

__device__ __forceinline__ BondRec load_bond(const BondRec *tab, uint32_t b) {
    const uint4 q = *reinterpret_cast<const uint4 *>(tab + b);
    BondRec r;
    r.a_info = q.x; r.c = q.y;
    r.w = __hiloint2double((int)q.w, (int)q.z);
    return r;
}

// ---------------------------------------------------------------------------------------------
// LDS carve (dynamic shared memory).  All sizes in u32 words.
template <int W>
struct Lds {
    uint32_t *state;   // [nwords]      p=0 spin state
    uint32_t *scopy;   // [W][nwords]   spin-state copies (XOR scan)
    uint32_t *touch;   // [nwords]      variables touched by any op
    int *tot;          // [2][W]        per-wave totals (double buffered by round parity)
    uint32_t *chg;     // [2][W]
    uint32_t *misc;    // [16]
    uint32_t *cur;     // [W][N]        latest-cut copies (MAX scan)
    uint32_t *frozen;  // [ufwords]     bit per id: segment holds a longitudinal op
    uint32_t *froot;   // [ufwords]     bit per id: root is frozen
    uint32_t *parent;  // [ufcap]
    __device__ static size_t words(uint32_t N, uint32_t nwords, uint32_t ufcap) {
        return (size_t)nwords * (W + 2) + 4 * W + 16 + (size_t)W * N + 2 * ((ufcap + 31) / 32) + ufcap;
    }
    __device__ void carve(uint32_t *base, uint32_t N, uint32_t nwords, uint32_t ufcap) {
        state = base; base += nwords;
        scopy = base; base += W * nwords;
        touch = base; base += nwords;
        tot = (int *)base; base += 2 * W;
        chg = base; base += 2 * W;
        misc = base; base += 16;
        cur = base; base += (size_t)W * N;
        frozen = base; base += (ufcap + 31) / 32;
        froot = base; base += (ufcap + 31) / 32;
        parent = base;
    }
};
enum { MISC_NCLUST = 0, MISC_ANYFROZEN = 1, MISC_LOOP_A = 2, MISC_LOOP_B = 3, MISC_LOOP_C = 4, MISC_LOOP_D = 5 };

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }

// ---------------------------------------------------------------------------------------------
// Diagonal pass.  Reference: DiagonalUpdater::make_diagonal_update_with_rng_and_state_ref
// (qmc_traits/diagonal.rs:114-135) with metropolis_single_diagonal_update (:142-191), or the heat-bath
// rule (qmc_traits/heatbath.rs:149-209) when HB.
template <int W, bool HB>
__device__ void diagonal_pass(const DevBatch &B, Lds<W> &L, uint32_t r, const Rng &rng, double beta, uint32_t M,
                              int &n_io, int &ntrans_io, uint32_t &gr) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nwords = B.nwords;
    uint32_t *ops = B.ops + (size_t)r * B.cap;
    const double beta_nb = beta * (double)B.Nb;
    const double hb_bw = beta * B.wtot;
    const uint32_t tag = HB ? SSE_TAG_HEATBATH : SSE_TAG_DIAG;

    for (uint32_t i = tid; i < nwords * W; i += NT) L.scopy[i] = L.state[i % nwords];
    __syncthreads();

    const uint32_t nblk = (M + NT - 1) / NT;
    int n_start = n_io, ntrans = ntrans_io;

    // prologue: block 0 word + its bond record, its off-diagonal events into copies of later waves
    uint32_t wnext = 0;
    BondRec recnext;
    recnext.a_info = 0; recnext.c = SSE_NO_VAR; recnext.w = 0.0;
    {
        uint32_t p = tid;
        if (p < M) wnext = ops[p];
        if (wnext) recnext = load_bond(B.bonds, sse_op_bond(wnext));
        uint32_t x = sse_op_in(wnext) ^ sse_op_out(wnext);
        if (x & 1u) { uint32_t v = rec_var(recnext); for (int w2 = wave + 1; w2 < W; ++w2) atomicXor(&L.scopy[w2 * nwords + (v >> 5)], 1u << (v & 31)); }
        if (x & 2u) { uint32_t v = recnext.c;        for (int w2 = wave + 1; w2 < W; ++w2) atomicXor(&L.scopy[w2 * nwords + (v >> 5)], 1u << (v & 31)); }
    }
    __syncthreads();

    for (uint32_t blk = 0; blk < nblk; ++blk) {
        const uint32_t p = blk * NT + tid;
        const bool valid = p < M;
        const uint32_t word = wnext;
        BondRec rec = recnext;
        // prefetch the next tile
        wnext = 0;
        {
            uint32_t pn = p + NT;
            if (blk + 1 < nblk && pn < M) wnext = ops[pn];
        }
        const uint32_t xbits = sse_op_in(word) ^ sse_op_out(word);
        const bool is_empty = valid && word == 0u;
        const bool is_diag = word != 0u && xbits == 0u;

        const uint4 rnd = rng.draw(tag, p);
        uint32_t bsel = 0, sub = 0;
        double num = 0.0, u = 0.0;
        bool cand_ins = false, cand_rem = false;
        bool hb_ok2 = false;
        if (is_empty) {
            if (HB) {
                // stage 2 of the heat-bath rule does not depend on n: evaluate it once
                const double c = u01(rnd.z) * B.wtot;
                uint32_t lo = 0, hi = B.Nb;
                while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (B.cumw[mid] < c) lo = mid + 1; else hi = mid; }
                bsel = lo < B.Nb ? lo : B.Nb - 1;
            } else {
                bsel = __umulhi(rnd.x, B.Nb);
            }
            rec = load_bond(B.bonds, bsel);
        }
        // spin reads for insertion candidates: copy[wave] xor in-wave earlier events
        const uint32_t va = rec_var(rec), vc = rec.c;
        {
            const uint64_t ev0 = __ballot((xbits & 1u) != 0u);
            const uint64_t ev1 = __ballot((xbits & 2u) != 0u);
            uint32_t sa = 0, sc = 0;
            if (is_empty) {
                sa = (L.scopy[wave * nwords + (va >> 5)] >> (va & 31)) & 1u;
                if (vc != SSE_NO_VAR) sc = (L.scopy[wave * nwords + (vc >> 5)] >> (vc & 31)) & 1u;
            }
            uint64_t m = ev0;
            while (m) {
                const int Ls = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t vL = __builtin_amdgcn_readlane(va, Ls);
                if (lane > Ls) { sa ^= (uint32_t)(va == vL); sc ^= (uint32_t)(vc == vL); }
            }
            m = ev1;
            while (m) {
                const int Ls = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t vL = __builtin_amdgcn_readlane(vc, Ls);
                if (lane > Ls) { sa ^= (uint32_t)(va == vL); sc ^= (uint32_t)(vc == vL); }
            }
            sub = sa | (sc << 1);
        }
        if (is_empty) {
            const double w = bond_weight(rec, sub, sub);
            if (HB) {
                hb_ok2 = u01(rnd.y) * rec.w < w;
                cand_ins = hb_ok2;
                u = u01(rnd.x);
            } else {
                num = beta_nb * w;
                cand_ins = w > 0.0;
                u = u01(rnd.y);
            }
        } else if (is_diag) {
            cand_rem = true;
            if (HB) u = u01(rnd.x);
            else { num = beta_nb * bond_weight(rec, sse_op_in(word), sse_op_in(word)); u = u01(rnd.y); }
        }

        // ---- fixed point on n ----
        int npref = n_start;
        int dec = 0, dec_prev = 2;
        int tot_all = 0;
        bool first = true;
        for (;;) {
            dec = 0;
            if (cand_ins) {
                const double den = (double)((int)M - npref);
                if (HB) dec = (u * (den + hb_bw) < hb_bw) ? 1 : 0;
                else dec = (u * den < num) ? 1 : 0;
            } else if (cand_rem) {
                const double den = (double)((int)M - npref + 1);
                if (HB) dec = (u * (den + hb_bw) < den) ? -1 : 0;
                else dec = (u * num < den) ? -1 : 0;
            }
            const uint64_t im = __ballot(dec > 0), rm = __ballot(dec < 0);
            const uint64_t cm = __ballot(dec != dec_prev);
            const int buf = gr & 1;
            if (lane == 0) { L.tot[buf * W + wave] = popc64(im) - popc64(rm); L.chg[buf * W + wave] = cm != 0ull; }
            __syncthreads();
            if (first) {
                first = false;
                // events of this tile -> copies of waves <= mine (all readers of this tile are done)
                if (xbits & 1u) for (int w2 = 0; w2 <= wave; ++w2) atomicXor(&L.scopy[w2 * nwords + (va >> 5)], 1u << (va & 31));
                if (xbits & 2u) for (int w2 = 0; w2 <= wave; ++w2) atomicXor(&L.scopy[w2 * nwords + (vc >> 5)], 1u << (vc & 31));
                // events of the next tile -> copies of waves > mine (visible after the next barrier)
                recnext.a_info = 0; recnext.c = SSE_NO_VAR; recnext.w = 0.0;
                if (wnext) recnext = load_bond(B.bonds, sse_op_bond(wnext));
                const uint32_t xn = sse_op_in(wnext) ^ sse_op_out(wnext);
                if (xn & 1u) { uint32_t v = rec_var(recnext); for (int w2 = wave + 1; w2 < W; ++w2) atomicXor(&L.scopy[w2 * nwords + (v >> 5)], 1u << (v & 31)); }
                if (xn & 2u) { uint32_t v = recnext.c;        for (int w2 = wave + 1; w2 < W; ++w2) atomicXor(&L.scopy[w2 * nwords + (v >> 5)], 1u << (v & 31)); }
            }
            int base = 0; tot_all = 0; uint32_t anychg = 0;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) {
                const int t = L.tot[buf * W + w2];
                if (w2 < wave) base += t;
                tot_all += t;
                anychg |= L.chg[buf * W + w2];
            }
            gr++;
            if (dec_prev != 2 && !anychg) break;
            npref = n_start + base + popc64(im & lanemask_lt(lane)) - popc64(rm & lanemask_lt(lane));
            dec_prev = dec;
        }
        // ---- commit ----
        if (dec != 0) {
            const uint32_t neww = dec > 0 ? sse_op_make(bsel, sub, sub) : 0u;
            ops[p] = neww;
        }
        const bool tr = rec_kind(rec) == SSE_BOND_TRANSVERSE;
        ntrans += popc64(__ballot(dec > 0 && tr)) - popc64(__ballot(dec < 0 && tr));
        n_start += tot_all;
    }
    // per-wave transverse deltas -> block total
    __syncthreads();
    if (lane == 0) L.tot[wave] = ntrans - ntrans_io;
    __syncthreads();
    int dt = 0;
#pragma unroll
    for (int w2 = 0; w2 < W; ++w2) dt += L.tot[w2];
    __syncthreads();
    ntrans_io += dt;
    n_io = n_start;
}

// ---------------------------------------------------------------------------------------------
// Lock-free union-find with smallest-id roots (canonical cluster labels).
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
    uint32_t p = parent[x];
    while (p != x) {
        const uint32_t g = parent[p];
        if (g != p) parent[x] = g; // path halving; benign race (always an ancestor)
        x = p;
        p = g;
    }
    return x;
}
__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        if (atomicCAS(&parent[b], b, a) == b) return;
    }
}

// Segment scan shared by cluster build and apply.  For the tile's ops it yields, per lane, the segment ids
// of its legs (seg_a for var a, seg_c for var c, id_own for a cut's outgoing segment).
// Segment ids: [0,N) = worldline part containing p=0 (placeholder), N+k = segment opened by the k-th cut.
template <int W, bool APPLY>
__device__ void cluster_scan(const DevBatch &B, Lds<W> &L, uint32_t r, uint32_t M, uint32_t *parent,
                             uint32_t *frozen, uint32_t &gr, uint32_t &ncuts_out) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t N = B.N;
    uint32_t *ops = B.ops + (size_t)r * B.cap;
    for (uint32_t i = tid; i < (uint32_t)W * N; i += NT) L.cur[i] = 0u;
    __syncthreads();
    const uint32_t nblk = (M + NT - 1) / NT;
    uint32_t cutbase = 0;
    // deferred phase-2 writes of the previous tile
    bool pend = false; uint32_t pend_v = 0, pend_id = 0;
    uint32_t wnext = (tid < M) ? ops[tid] : 0u;
    for (uint32_t blk = 0; blk < nblk; ++blk) {
        const uint32_t p = blk * NT + tid;
        const uint32_t word = wnext;
        wnext = 0;
        if (blk + 1 < nblk && p + NT < M) wnext = ops[p + NT];
        BondRec rec; rec.a_info = 0; rec.c = SSE_NO_VAR; rec.w = 0.0;
        if (word) rec = load_bond(B.bonds, sse_op_bond(word));
        const uint32_t kind = rec_kind(rec);
        const uint32_t va = rec_var(rec), vc = rec.c;
        const bool nonempty = word != 0u;
        const bool iscut = nonempty && kind == SSE_BOND_TRANSVERSE;
        const uint64_t cutmask = __ballot(iscut);
        const int buf = gr & 1;
        if (lane == 0) L.tot[buf * W + wave] = popc64(cutmask);
        __syncthreads(); // (A)
        gr++;
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < W; ++w2) { const uint32_t t = (uint32_t)L.tot[buf * W + w2]; if (w2 < wave) wbase += t; total += t; }
        // previous tile's cuts become visible to waves <= their own (everyone finished reading that tile)
        if (pend) for (int w2 = 0; w2 <= wave; ++w2) atomicMax(&L.cur[w2 * N + pend_v], pend_id);
        const uint32_t first_id = N + cutbase + wbase;
        const uint32_t id_own = first_id + popc64(cutmask & lanemask_lt(lane));
        if (iscut) {
            if (!APPLY) parent[id_own] = id_own;
            for (int w2 = wave + 1; w2 < W; ++w2) atomicMax(&L.cur[w2 * N + va], id_own);
        }
        pend = iscut; pend_v = va; pend_id = id_own;
        __syncthreads(); // (B)
        uint32_t seg_a = 0, seg_c = 0;
        if (nonempty) {
            seg_a = L.cur[wave * N + va];
            if (vc != SSE_NO_VAR) seg_c = L.cur[wave * N + vc];
        }
        {
            uint64_t m = cutmask;
            uint32_t idL = first_id;
            while (m) {
                const int Ls = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t vL = __builtin_amdgcn_readlane(va, Ls);
                if (lane > Ls) { if (va == vL) seg_a = idL; if (vc == vL) seg_c = idL; }
                idL++;
            }
        }
        if (nonempty) {
            if (seg_a == 0u) seg_a = va;
            if (vc != SSE_NO_VAR && seg_c == 0u) seg_c = vc;
            if (!APPLY) {
                atomicOr(&L.touch[va >> 5], 1u << (va & 31));
                if (kind == SSE_BOND_TWO_SITE) {
                    atomicOr(&L.touch[vc >> 5], 1u << (vc & 31));
                    uf_union(parent, seg_a, seg_c);
                } else if (kind == SSE_BOND_LONGITUDINAL) {
                    atomicOr(&frozen[seg_a >> 5], 1u << (seg_a & 31));
                }
            } else {
                uint32_t in = sse_op_in(word), out = sse_op_out(word);
                const uint32_t fa = parent[seg_a];
                if (iscut) {
                    in ^= fa;
                    out ^= parent[id_own];
                } else if (kind == SSE_BOND_TWO_SITE) {
                    const uint32_t f2 = fa | (parent[seg_c] << 1);
                    in ^= f2; out ^= f2;
                } else {
                    in ^= fa; out ^= fa;
                }
                const uint32_t neww = (word & ~0xFu) | in | (out << SSE_OP_OUT_SHIFT);
                if (neww != word) ops[p] = neww;
            }
        }
        cutbase += total;
    }
    __syncthreads();
    if (pend) for (int w2 = 0; w2 <= wave; ++w2) atomicMax(&L.cur[w2 * N + pend_v], pend_id);
    __syncthreads();
    ncuts_out = cutbase;
}

// Cluster update.  Reference: ClusterUpdater::flip_each_cluster_rng (qmc_traits/cluster.rs:36-172) with the
// longitudinal weight function of qmc_ising.rs:759-775.  Returns the number of clusters.
template <int W, bool UF_GLOBAL>
__device__ uint32_t cluster_pass(const DevBatch &B, Lds<W> &L, uint32_t r, const Rng &rng, double prob, uint32_t M,
                                 int n, int ntrans, uint32_t &gr) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t N = B.N, nwords = B.nwords;
    uint32_t *parent, *frozen, *froot;
    if constexpr (UF_GLOBAL) {
        const size_t ids = (size_t)N + B.cap;
        parent = B.uf_scratch + (size_t)r * (ids + 2 * ((ids + 31) / 32));
        frozen = parent + ids;
        froot = frozen + (ids + 31) / 32;
    } else {
        parent = L.parent; frozen = L.frozen; froot = L.froot;
    }
    for (uint32_t i = tid; i < nwords; i += NT) L.touch[i] = 0u;
    if (tid == 0) { L.misc[MISC_NCLUST] = 0u; L.misc[MISC_ANYFROZEN] = 0u; }
    if (n == 0) { __syncthreads(); return 0u; } // cluster.rs:46-48
    const uint32_t S = N + (uint32_t)ntrans; // ids: N placeholders + one per cut (transverse op)
    for (uint32_t i = tid; i < N; i += NT) parent[i] = i;
    for (uint32_t i = tid; i < (S + 31) / 32; i += NT) { frozen[i] = 0u; froot[i] = 0u; }
    __syncthreads();
    // ---- build: label legs with segment ids, union through non-boundary ops ----
    uint32_t ncuts = 0;
    cluster_scan<W, false>(B, L, r, M, parent, frozen, gr, ncuts);
    // wrap-around: the part of worldline v before its first cut continues the segment of its last cut
    for (uint32_t v = tid; v < N; v += NT) { const uint32_t last = L.cur[v]; if (last) uf_union(parent, v, last); }
    __syncthreads();
    // ---- flatten: parent[i] := exact root (no union runs any more), frozen marks move to roots ----
    for (uint32_t i = tid; i < S; i += NT) {
        const uint32_t root = uf_find(parent, i);
        parent[i] = root;
        if ((frozen[i >> 5] >> (i & 31)) & 1u) { atomicOr(&froot[root >> 5], 1u << (root & 31)); L.misc[MISC_ANYFROZEN] = 1u; }
    }
    __syncthreads();
    // ---- coins: each thread reads only parent[i] of its own ids, so parent[i] := flip bit in place ----
    uint32_t myclusters = 0;
    const bool nocuts = (ncuts == 0u);
    const uint32_t anyfrozen = L.misc[MISC_ANYFROZEN];
    for (uint32_t i = tid; i < S; i += NT) {
        const uint32_t root = parent[i];
        const bool touched = i >= N || ((L.touch[i >> 5] >> (i & 31)) & 1u);
        uint32_t f;
        if (nocuts) {
            // no cluster boundary anywhere: the whole graph is one cluster (cluster.rs:98-107), label 0
            const uint4 o = rng.draw(SSE_TAG_CLUSTER, 0u);
            f = (touched && !anyfrozen && u01(o.x) < prob) ? 1u : 0u;
        } else {
            if (root == i && touched) myclusters++;
            const uint4 o = rng.draw(SSE_TAG_CLUSTER, root);
            const uint32_t isfrozen = (froot[root >> 5] >> (root & 31)) & 1u;
            f = (!isfrozen && u01(o.x) < prob) ? 1u : 0u;
        }
        parent[i] = f;
    }
    {
        uint32_t c = myclusters;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if (lane == 0 && c) atomicAdd(&L.misc[MISC_NCLUST], c);
    }
    __syncthreads();
    // ---- apply (cluster.rs:139-167) ----
    uint32_t ncuts2 = 0;
    cluster_scan<W, true>(B, L, r, M, parent, frozen, gr, ncuts2);
    // p=0 state follows the placeholder segment of each touched variable
    for (uint32_t i = tid; i < nwords; i += NT) {
        uint32_t x = 0;
        const uint32_t t = L.touch[i];
        for (uint32_t j = 0; j < 32 && i * 32 + j < N; ++j) x |= (parent[i * 32 + j] & 1u) << j;
        L.state[i] ^= (x & t);
    }
    __syncthreads();
    return nocuts ? 1u : L.misc[MISC_NCLUST];
}

// touched-variable scan for launches that flip free spins without a preceding cluster pass
template <int W>
__device__ void touch_scan(const DevBatch &B, Lds<W> &L, uint32_t r, uint32_t M) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x;
    const uint32_t *ops = B.ops + (size_t)r * B.cap;
    for (uint32_t i = tid; i < B.nwords; i += NT) L.touch[i] = 0u;
    __syncthreads();
    for (uint32_t p = tid; p < M; p += NT) {
        const uint32_t word = ops[p];
        if (!word) continue;
        const BondRec rec = load_bond(B.bonds, sse_op_bond(word));
        const uint32_t va = rec_var(rec);
        atomicOr(&L.touch[va >> 5], 1u << (va & 31));
        if (rec.c != SSE_NO_VAR) atomicOr(&L.touch[rec.c >> 5], 1u << (rec.c & 31));
    }
    __syncthreads();
}

// qmc_ising.rs:780-784 / qmc_runner.rs:241-255
template <int W>
__device__ void free_spin_pass(const DevBatch &B, Lds<W> &L, const Rng &rng) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < B.nwords; i += NT) {
        const uint32_t t = L.touch[i];
        uint32_t s = L.state[i];
        for (uint32_t j = 0; j < 32 && i * 32 + j < B.N; ++j)
            if (!((t >> j) & 1u)) {
                const uint4 o = rng.draw(SSE_TAG_FREE, i * 32 + j);
                s = (s & ~(1u << j)) | ((o.x >> 31) << j);
            }
        L.state[i] = s;
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
template <int W>
__device__ uint32_t loop_pass(const DevBatch &B, Lds<W> &L, uint32_t r, const Rng &rng, uint32_t M, int n, uint32_t &gr,
                              uint32_t &err) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *ops = B.ops + (size_t)r * B.cap;
    if (n == 0) return 0u;
    const uint4 o0 = rng.draw(SSE_TAG_LOOP, 0u);
    const uint32_t nth = __umulhi(o0.x, (uint32_t)n);
    // ---- start vertex: the nth occupied slot in p order ----
    const uint32_t nblk = (M + NT - 1) / NT;
    uint32_t cbase = 0;
    if (tid == 0) L.misc[MISC_LOOP_A] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t blk = 0; blk < nblk; ++blk) {
        const uint32_t p = blk * NT + tid;
        const uint32_t word = p < M ? ops[p] : 0u;
        const uint64_t occ = __ballot(word != 0u);
        const int buf = gr & 1;
        if (lane == 0) L.tot[buf * W + wave] = popc64(occ);
        __syncthreads();
        gr++;
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < W; ++w2) { const uint32_t t = (uint32_t)L.tot[buf * W + w2]; if (w2 < wave) wbase += t; total += t; }
        const uint32_t myrank = cbase + wbase + popc64(occ & lanemask_lt(lane));
        if (word != 0u && myrank == nth) L.misc[MISC_LOOP_A] = p;
        cbase += total;
        if (cbase > nth) break;
    }
    __syncthreads();
    const uint32_t p0 = L.misc[MISC_LOOP_A];
    if (p0 == 0xFFFFFFFFu) { err = 2u; return 0u; } // n inconsistent with the op-string
    uint32_t rel0, side0;
    {
        const BondRec rec0 = load_bond(B.bonds, sse_op_bond(ops[p0]));
        const uint32_t k0 = rec0.c != SSE_NO_VAR ? 2u : 1u;
        rel0 = __umulhi(o0.y, k0);
        side0 = (o0.z >> 31) ? 0u : 1u; // gen() true -> Inputs (directed_loop.rs:153-157)
    }
    uint32_t p = p0, rel = rel0, side = side0, visited = 0;
    const uint32_t max_steps = 8u * M + 64u;
    for (uint32_t step = 1; step <= max_steps; ++step) {
        // ---- vertex update by thread 0 ----
        if (tid == 0) {
            const uint32_t word = ops[p];
            const BondRec rec = load_bond(B.bonds, sse_op_bond(word));
            const uint32_t k = rec.c != SSE_NO_VAR ? 2u : 1u;
            uint32_t in_e = sse_op_in(word), out_e = sse_op_out(word);
            if (side == 0u) in_e ^= 1u << rel; else out_e ^= 1u << rel;
            double wl[4], total = 0.0;
            for (uint32_t leg = 0; leg < 2u * k; ++leg) {
                uint32_t i2 = in_e, o2 = out_e;
                if (leg < k) i2 ^= 1u << leg; else o2 ^= 1u << (leg - k);
                wl[leg] = bond_weight(rec, i2, o2);
                total += wl[leg];
            }
            const uint4 o = rng.draw(SSE_TAG_LOOP, step);
            double c = u01(o.x) * total;
            uint32_t exit_leg = 2u * k - 1u;
            for (uint32_t leg = 0; leg < 2u * k; ++leg) {
                if (c < wl[leg]) { exit_leg = leg; break; }
                c -= wl[leg];
            }
            const uint32_t xside = exit_leg < k ? 0u : 1u, xrel = exit_leg < k ? exit_leg : exit_leg - k;
            if (xside == 0u) in_e ^= 1u << xrel; else out_e ^= 1u << xrel;
            ops[p] = (word & ~0xFu) | in_e | (out_e << SSE_OP_OUT_SHIFT);
            const bool closed = (p == p0 && xrel == rel0 && xside == side0);
            const uint32_t var = xrel == 0u ? rec_var(rec) : rec.c;
            L.misc[MISC_LOOP_A] = closed ? 1u : 0u;
            L.misc[MISC_LOOP_B] = var;
            L.misc[MISC_LOOP_C] = xside | (xrel << 1) | ((((xside == 1u ? out_e : in_e) >> xrel) & 1u) << 2);
            L.misc[MISC_LOOP_D] = 0xFFFFFFFFu; // best distance
        }
        __syncthreads();
        visited++;
        if (L.misc[MISC_LOOP_A]) break;
        const uint32_t var = L.misc[MISC_LOOP_B];
        const uint32_t info = L.misc[MISC_LOOP_C];
        const uint32_t xside = info & 1u, newbit = (info >> 2) & 1u;
        const bool forward = xside == 1u;
        // ---- search the next op on worldline `var`, distance 1..M (distance M = the op itself) ----
        uint32_t found = 0xFFFFFFFFu;
        for (uint32_t d0 = 1; d0 <= M; d0 += NT) {
            const uint32_t d = d0 + tid;
            bool match = false;
            if (d <= M) {
                uint32_t q = forward ? p + d : p + M - d;
                if (q >= M) q -= M;
                const uint32_t word = ops[q];
                if (word) {
                    const BondRec rec = load_bond(B.bonds, sse_op_bond(word));
                    match = rec_var(rec) == var || rec.c == var;
                }
            }
            const uint64_t mm = __ballot(match);
            if (mm && lane == 0) atomicMin(&L.misc[MISC_LOOP_D], d0 + (uint32_t)(wave * 64) + (uint32_t)(__ffsll((long long)mm) - 1));
            __syncthreads();
            found = L.misc[MISC_LOOP_D];
            __syncthreads();
            if (found != 0xFFFFFFFFu) break;
        }
        if (found == 0xFFFFFFFFu) { err = 2u; break; }
        uint32_t q = forward ? p + found : p + M - found;
        bool wrapped = forward ? (q >= M) : (found > p);
        if (q >= M) q -= M;
        const BondRec recq = load_bond(B.bonds, sse_op_bond(ops[q]));
        const uint32_t nrel = rec_var(recq) == var ? 0u : 1u;
        if (wrapped && tid == 0) { // directed_loop.rs:276-288
            const uint32_t wi = var >> 5, bi = var & 31;
            L.state[wi] = (L.state[wi] & ~(1u << bi)) | (newbit << bi);
        }
        const uint32_t nside = xside ^ 1u;
        if (q == p0 && nrel == rel0 && nside == side0) break; // :293
        p = q; rel = nrel; side = nside;
    }
    __syncthreads();
    return visited;
}

// ---------------------------------------------------------------------------------------------
// One launch = nsteps timesteps of every replica.  Reference drivers: QmcIsingGraph::timestep
// (qmc_ising.rs:644-795), Qmc::timestep (qmc_runner.rs:363-377), measurement loop
// QmcStepper::timesteps_measure_with_self (qmc_traits/qmc_stepper.rs:133-162).
template <int W>
__global__ __launch_bounds__(W * 64) void sweep_kernel(DevBatch B, SweepArgs A) {
    extern __shared__ __align__(16) uint32_t lds_raw[];
    constexpr int NT = W * 64;
    Lds<W> L;
    L.carve(lds_raw, B.N, B.nwords, B.lds_ufcap);
    const int tid = threadIdx.x;
    const uint32_t r = blockIdx.x;
    for (uint32_t i = tid; i < B.nwords; i += NT) L.state[i] = B.state[(size_t)r * B.nwords + i];
    __syncthreads();
    int n = (int)B.n[r], ntrans = (int)B.ntrans[r];
    uint32_t M = B.cutoff[r], err = B.err[r], gr = 0, last_out = 0;
    uint64_t epoch = B.epoch[r];
    const double beta = A.beta ? A.beta[r] : 0.0;
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0;
    for (uint64_t step = 0; step < A.nsteps; ++step) {
        if (err) break;
        if (A.domask & SSE_DO_DIAG) {
            const Rng rng = make_rng(B, r, epoch);
            if (A.domask & SSE_DO_HEATBATH) diagonal_pass<W, true>(B, L, r, rng, beta, M, n, ntrans, gr);
            else diagonal_pass<W, false>(B, L, r, rng, beta, M, n, ntrans, gr);
            epoch++;
            a5 += M;
            if (A.domask & SSE_DO_GROW) { // qmc_ising.rs:786, qmc_runner.rs:197
                const uint32_t want = (uint32_t)n + (uint32_t)n / 2u;
                if (want > M) { if (want > B.cap) { err = 1u; break; } M = want; }
            }
        }
        if (A.domask & SSE_DO_LOOP) {
            const Rng rng = make_rng(B, r, epoch);
            last_out = loop_pass<W>(B, L, r, rng, M, n, gr, err);
            epoch++;
            a4 += last_out;
            if (err) break;
        }
        if (A.domask & SSE_DO_CLUSTER) {
            const Rng rng = make_rng(B, r, epoch);
            if (B.N + (uint32_t)ntrans <= B.lds_ufcap) last_out = cluster_pass<W, false>(B, L, r, rng, A.prob, M, n, ntrans, gr);
            else last_out = cluster_pass<W, true>(B, L, r, rng, A.prob, M, n, ntrans, gr);
            epoch++;
            a4 += (uint64_t)n;
        }
        if (A.domask & SSE_DO_FREE) {
            const Rng rng = make_rng(B, r, epoch);
            if (!(A.domask & SSE_DO_CLUSTER)) touch_scan<W>(B, L, r, M);
            free_spin_pass<W>(B, L, rng);
            epoch++;
        }
        if (A.sampling_freq && (step + 1) % A.sampling_freq == 0) {
            if (tid == 0) L.misc[MISC_LOOP_A] = 0u;
            __syncthreads();
            uint32_t up = 0;
            for (uint32_t i = tid; i < B.nwords; i += NT) up += __popc(L.state[i]);
            for (int off = 32; off > 0; off >>= 1) up += __shfl_down(up, off);
            if ((tid & 63) == 0 && up) atomicAdd(&L.misc[MISC_LOOP_A], up);
            __syncthreads();
            const long long mag = 2ll * (long long)L.misc[MISC_LOOP_A] - (long long)B.N;
            a0 += (uint64_t)n; a1 += 1; a2 += (uint64_t)(mag < 0 ? -mag : mag); a3 += (uint64_t)(mag * mag); a6 += (uint64_t)ntrans;
            __syncthreads();
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < B.nwords; i += NT) B.state[(size_t)r * B.nwords + i] = L.state[i];
    if (tid == 0) {
        B.n[r] = (uint32_t)n; B.ntrans[r] = (uint32_t)ntrans; B.cutoff[r] = M; B.err[r] = err; B.epoch[r] = epoch;
        if (A.out_u32) A.out_u32[r] = last_out;
        uint64_t *acc = B.acc + (size_t)r * 8;
        acc[0] += a0; acc[1] += a1; acc[2] += a2; acc[3] += a3; acc[4] += a4; acc[5] += a5; acc[6] += a6;
    }
}

// Verify::verify (qmc_ising.rs:829-860; op_container.rs:137-159) and bond counts, one thread per replica
// (debug API, not on the hot path).  ok[r] = 1 iff every op has non-zero weight, the propagated state matches
// every op's inputs, periodicity holds and the occupied-slot count equals n.
__global__ void verify_kernel(DevBatch B, uint32_t *scratch_state /*[R][nwords]*/, uint8_t *ok) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.R) return;
    uint32_t *s = scratch_state + (size_t)r * B.nwords;
    const uint32_t *s0 = B.state + (size_t)r * B.nwords;
    for (uint32_t i = 0; i < B.nwords; ++i) s[i] = s0[i];
    const uint32_t *ops = B.ops + (size_t)r * B.cap;
    const uint32_t M = B.cutoff[r];
    bool good = true;
    uint32_t count = 0, ntr = 0;
    for (uint32_t p = 0; p < B.cap; ++p) {
        const uint32_t w = ops[p];
        if (!w) continue;
        if (p >= M) { good = false; break; }
        count++;
        const uint32_t b = sse_op_bond(w);
        if (b >= B.Nb) { good = false; break; }
        const BondRec rec = load_bond(B.bonds, b);
        const uint32_t in = sse_op_in(w), out = sse_op_out(w);
        if (!(bond_weight(rec, in, out) > 2.220446049250313e-16)) good = false;
        if (rec_kind(rec) == SSE_BOND_TRANSVERSE) ntr++;
        const uint32_t a = rec_var(rec), c = rec.c;
        if (((s[a >> 5] >> (a & 31)) & 1u) != (in & 1u)) good = false;
        s[a >> 5] = (s[a >> 5] & ~(1u << (a & 31))) | ((out & 1u) << (a & 31));
        if (c != SSE_NO_VAR) {
            if (((s[c >> 5] >> (c & 31)) & 1u) != ((in >> 1) & 1u)) good = false;
            s[c >> 5] = (s[c >> 5] & ~(1u << (c & 31))) | (((out >> 1) & 1u) << (c & 31));
        } else if ((in | out) & 2u) good = false;
    }
    for (uint32_t i = 0; i < B.nwords; ++i) if (s[i] != s0[i]) good = false;
    if (count != B.n[r] || ntr != B.ntrans[r]) good = false;
    ok[r] = good ? 1 : 0;
}

} // namespace sse
