// sse_rvb_split.hip.h — the RVB sweep as two launches (included by sweep_rvb.hip after sse_device.hip.h).
//
// Reference: RvbUpdater::rvb_update_with_ising_weight (src/sse/qmc_traits/rvb.rs:88-290); the pieces are those of
// sse_rvb.hip.h.  The growth of an attempt (start, build_cluster, sub-variables, toggles, windows: rvb.rs:88-232,
// :1054-1123) reads the constant-op table, the adjacency and the attempt's own random numbers — never the op-string, which
// is the only thing an accepted attempt changes.  So the growths of ALL attempts of a sweep are independent of one another:
//
//   rvb_grow_kernel   one workgroup of 16 waves per replica builds the constant-op table in LDS (it takes half of it) and grows
//                     the sweep's attempts side by side, one per wave, wave w taking attempts w, w + P, ...; the products go to
//                     HBM (DevBatch::rvb_prod, one record per attempt);
//   rvb_main_kernel   runs the attempts in order (probability pass, accept, mutation: rvb_attempt).  Without the table its
//                     workgroup needs < 40 KB of LDS, so four replicas share a CU and hide each other's barriers, HBM round
//                     trips and single-wave replays — the fused kernel (rvb_pass) has one replica per CU and exposes them all.
//                     The record of the next attempt is fetched while the current one runs.
//
// Results are those of rvb_pass bit for bit (same functions, same draws).
#pragma once

namespace sse {

// record of one attempt in HBM: [8] header (GO_*), tog[ntog], togs[ntog], sub[nsub], sfl[nsub], wfrom[nwin], wuntil[nwin]
#define SSE_RVB_PROD_STRIDE (8u + 4u * SSE_RVB_MAXCL + 2u * SSE_RVB_MAXSUB + 2u * SSE_RVB_MAXWIN) // words per attempt (a large area's lists)
#define SSE_RVB_PROD_SMALL 400u // every product of a small growth area fits in this many words: fetched ahead, one word per thread
static_assert(8u + 4u * SSE_RVB_SLOT_CL + 2u * SSE_RVB_SLOT_SUB + 2u * SSE_RVB_SLOT_WIN <= SSE_RVB_PROD_SMALL, "small products are prefetched whole");
#define SSE_RVB_REGROW 0xFFFFFFFFu // header error word: the cluster outgrew its small area, the large one takes it

__device__ __forceinline__ uint32_t rvb_prod_words(uint32_t nsub, uint32_t nwin, uint32_t ntog) { return 8u + 2u * ntog + 2u * nsub + 2u * nwin; }

// LDS of the growth launch behind Lds::o_cur: what rvb_find_constants and rvb_grow touch, then the table
__host__ __device__ inline uint32_t rvb_grow_fixed_words(uint32_t N) {
    return 2u + (N + 1) + N + (N + 1) / 2 + 7 * SSE_RVB_SETCAP + 2 * SSE_RVB_MAXCL + 4 * SSE_RVB_MAXCL + 3 * SSE_RVB_MAXSUB + 2 * SSE_RVB_MAXWIN + 16 + 8;
}
template <int W>
__device__ __forceinline__ void rvb_carve_grow(RvbLds &R, const Lds<W> &L, const DevBatch &B) {
    uint32_t base = (L.o_cur + 1u) & ~1u;
    R.o_bfw = base; R.o_bnw = base + 2 * SSE_RVB_SETCAP; R.o_bfk = base + 4 * SSE_RVB_SETCAP; R.o_bfv = base + 5 * SSE_RVB_SETCAP; R.o_bnk = base + 6 * SSE_RVB_SETCAP;
    base += 7 * SSE_RVB_SETCAP;
    R.o_vstart = base; base += B.N + 1;
    R.o_zero = base; base += B.N;
    R.o_v2s = base; base += (B.N + 1) / 2;
    R.adj_lds = 0u; R.o_adjs = R.o_adj = 0u;
    R.o_clv = base; base += SSE_RVB_MAXCL;
    R.o_clf = base; base += SSE_RVB_MAXCL;
    R.o_tog = base; base += 2 * SSE_RVB_MAXCL;
    R.o_togs = base; base += 2 * SSE_RVB_MAXCL;
    R.o_sub = base; base += SSE_RVB_MAXSUB;
    R.o_sfl = base; base += SSE_RVB_MAXSUB;
    R.o_last = base; base += SSE_RVB_MAXSUB;
    R.o_wfrom = base; base += SSE_RVB_MAXWIN;
    R.o_wuntil = base; base += SSE_RVB_MAXWIN;
    R.o_ctl = base; base += 16;
    R.o_gout = base; base += 8;
    R.o_bk = R.o_bwb = R.o_bwa = R.o_glp = R.o_glw = R.o_gli = R.o_bix = 0u; // (probability pass / mutation only)
    R.o_cps = base;
    R.cps_cap = B.lds_words > base ? B.lds_words - base : 0u;
}

// LDS of the main launch: [nwords] state, [2W] totals, [E] compact edges (CL), then the scratch of rvb_attempt, two prefetched
// records and room for one large record
__host__ __device__ inline uint32_t rvb_main_words(uint32_t W, uint32_t N, uint32_t nwords, uint32_t ledges, uint32_t E) {
    return nwords + 2 * W + 16 + ledges + 2u + 4 * SSE_RVB_BONDCAP + (N + 1) / 2 + SSE_RVB_MAXSUB + SSE_RVB_BONDCAP + 3 * SSE_RVB_GCAP + (E + 1) / 2 + 16 +
           2 * SSE_RVB_PROD_SMALL + SSE_RVB_PROD_STRIDE + 8;
}
struct RvbMainLds { uint32_t o_pbuf, o_big; };
template <int W>
__device__ __forceinline__ void rvb_carve_main(Lds<W> &L, RvbLds &R, RvbMainLds &P, const DevBatch &B, uint32_t ledges) {
    uint32_t base = 0;
    L.o_state = base; base += B.nwords;
    L.o_tot = base; base += 2 * W;
    L.o_misc = base; base += 16;
    L.o_edges = base; base += ledges;
    L.o_touch = L.o_touch8 = L.o_chg = L.o_chn = L.o_chtr = L.o_signs = L.o_cur = L.o_cl = L.o_frozen = L.o_froot = L.o_parent = base; // (not used by this launch)
    base = (base + 1u) & ~1u;
    R.o_bwb = base; base += 2 * SSE_RVB_BONDCAP;
    R.o_bwa = base; base += 2 * SSE_RVB_BONDCAP;
    R.o_v2s = base; base += (B.N + 1) / 2;
    R.o_last = base; base += SSE_RVB_MAXSUB;
    R.o_bk = base; base += SSE_RVB_BONDCAP;
    R.o_glp = base; base += SSE_RVB_GCAP;
    R.o_glw = base; base += SSE_RVB_GCAP;
    R.o_gli = base; base += SSE_RVB_GCAP;
    R.o_bix = base; base += (B.E + 1) / 2;
    R.o_ctl = base; base += 16;
    P.o_pbuf = base; base += 2 * SSE_RVB_PROD_SMALL;
    P.o_big = base; base += SSE_RVB_PROD_STRIDE;
    R.adj_lds = 0u; R.o_adjs = R.o_adj = 0u;
    R.o_vstart = R.o_zero = R.o_cps = R.o_gout = R.o_clv = R.o_clf = 0u; R.cps_cap = 0u; // (growth only)
    R.o_bfw = R.o_bnw = R.o_bfk = R.o_bfv = R.o_bnk = 0u;
    R.o_sub = R.o_sfl = R.o_tog = R.o_togs = R.o_wfrom = R.o_wuntil = 0u; // set per attempt
}

// one wave copies the products of a finished growth to the attempt's record
__device__ __forceinline__ void rvb_store_product(uint32_t *dst, const GrowArea &A, bool small_area, int lane) {
    const uint32_t nsub = LDSW(A.o_out, GO_NSUB), nwin = LDSW(A.o_out, GO_NWIN), ntog = LDSW(A.o_out, GO_NTOG), gerr = LDSW(A.o_out, GO_ERR);
    if (lane < 8) dst[lane] = (lane == GO_ERR && gerr && small_area) ? SSE_RVB_REGROW : LDSW(A.o_out, lane);
    if (gerr) return;
    uint32_t off = 8u;
    for (uint32_t i = lane; i < ntog; i += 64u) { dst[off + i] = LDSW(A.o_tog, i); dst[off + ntog + i] = LDSW(A.o_togs, i); }
    off += 2u * ntog;
    for (uint32_t i = lane; i < nsub; i += 64u) { dst[off + i] = LDSW(A.o_sub, i); dst[off + nsub + i] = LDSW(A.o_sfl, i); }
    off += 2u * nsub;
    for (uint32_t i = lane; i < nwin; i += 64u) { dst[off + i] = LDSW(A.o_wfrom, i); dst[off + nwin + i] = LDSW(A.o_wuntil, i); }
}

template <bool CL>
__global__ __launch_bounds__(1024, 4) void rvb_grow_kernel(DevBatch B, SweepArgs A) {
    constexpr int W = 16, NT = W * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r = blockIdx.x;
    if (B.err[r]) return;
    Lds<W> L;
    L.carve(B.N, B.nwords, 0u, CL ? B.E : 0u, 0u);
    if (B.bond_stride) { const uint32_t hr = B.ham_row ? B.ham_row[r] : r; B.bonds += (size_t)hr * B.bond_stride; }
    if constexpr (CL)
        for (uint32_t i = tid; i < B.E; i += NT) LDSW(L.o_edges, i) = B.edges_compact[i];
    RvbLds R;
    rvb_carve_grow<W>(R, L, B);
    __syncthreads();
    const uint32_t M = B.cutoff[r];
    const uint64_t epoch = B.epoch[r];
    const uint32_t updates = A.rvb_updates ? A.rvb_updates : (B.N + 1u) / 2u;
    uint32_t *prod = B.rvb_prod + (size_t)r * B.rvb_prod_cap * SSE_RVB_PROD_STRIDE;
    SSE_STAMP_INIT; // diagnostic builds: 6 constants table, 7 growth in the small areas, 13 regrowth in the large one
    const uint32_t C = rvb_find_constants<W, CL>(B, L, R, r, M);
    SSE_STAMP(6);
    if (C == 0xFFFFFFFFu) { if (tid == 0) prod[GO_ERR] = 6u; return; } // the table does not fit in LDS: the main launch reports it at attempt 0
    const uint32_t nzero = LDSW(R.o_ctl, RC_NZERO);
    const uint32_t slots0 = (R.o_cps + C + 1u) & ~1u;
    uint32_t P = B.lds_words > slots0 ? (B.lds_words - slots0) / SSE_RVB_SLOT_WORDS : 0u;
    if (P > (uint32_t)W) P = (uint32_t)W;
    if (P > B.rvb_growers) P = B.rvb_growers;
    RvbDraw g;
    g.k0 = B.seed_lo; g.k1 = B.seed_hi; g.replica = B.rid ? B.rid[r] : B.replica_offset + r; g.epoch_lo = (uint32_t)epoch;
    if ((uint32_t)wave < P) {
        const GrowArea As = grow_area_small(slots0 + (uint32_t)wave * SSE_RVB_SLOT_WORDS);
        for (uint32_t a = (uint32_t)wave; a < updates; a += P) {
            g.attempt = a; g.k = 0;
            rvb_grow<W, CL, true>(B, L, R, As, g, C, nzero, M, lane);
            SSE_WAVE_FENCE();
            rvb_store_product(prod + (size_t)a * SSE_RVB_PROD_STRIDE, As, true, lane);
            SSE_WAVE_FENCE();
        }
    }
    __threadfence_block();
    __syncthreads();
    SSE_STAMP(7);
    if (wave == 0) { // clusters that outgrew a small area (or all of them when there is no room for small areas): the large one, one at a time
        const GrowArea big = grow_area_large(R);
        for (uint32_t base = 0; base < updates; base += 64u) {
            const uint32_t a = base + (uint32_t)lane;
            uint64_t m = sse_ballot(a < updates && (P == 0u || prod[(size_t)a * SSE_RVB_PROD_STRIDE + GO_ERR] == SSE_RVB_REGROW));
            while (m) {
                const uint32_t k = (uint32_t)__ffsll((long long)m) - 1u;
                m &= m - 1;
                g.attempt = base + k; g.k = 0;
                rvb_grow<W, CL, false>(B, L, R, big, g, C, nzero, M, lane);
                SSE_WAVE_FENCE();
                rvb_store_product(prod + (size_t)(base + k) * SSE_RVB_PROD_STRIDE, big, false, lane);
                SSE_WAVE_FENCE();
            }
        }
    }
    SSE_STAMP(13);
}

template <int W> constexpr int rvb_main_waves_per_simd() { return W <= 4 ? 4 : (W <= 8 ? 4 : 4); }

template <int W, bool CL>
__global__ __launch_bounds__(W * 64, (rvb_main_waves_per_simd<W>())) void rvb_main_kernel(DevBatch B, SweepArgs A) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x;
    const uint32_t r = blockIdx.x;
    uint32_t err = B.err[r];
    if (err) return;
    Lds<W> L;
    RvbLds R0;
    RvbMainLds PB;
    rvb_carve_main<W>(L, R0, PB, B, CL ? B.E : 0u);
    if (B.bond_stride) { const uint32_t hr = B.ham_row ? B.ham_row[r] : r; B.bonds += (size_t)hr * B.bond_stride; }
    for (uint32_t i = tid; i < B.nwords; i += NT) LDSW(L.o_state, i) = B.state[(size_t)r * B.nwords + i];
    if constexpr (CL)
        for (uint32_t i = tid; i < B.E; i += NT) LDSW(L.o_edges, i) = B.edges_compact[i];
    for (uint32_t i = tid; i < B.E; i += NT) LDSH(R0.o_bix, i) = (uint16_t)0xFFFFu;
    for (uint32_t v = tid; v < B.N; v += NT) LDSH(R0.o_v2s, v) = (uint16_t)0xFFFFu;
    if (tid == 0) { LDSW(R0.o_ctl, RC_ERR) = 0u; LDSW(R0.o_ctl, RC_SKIP) = 0u; LDSW(R0.o_ctl, RC_BROKE) = 0u; }
    const uint32_t M = B.cutoff[r];
    const uint64_t epoch = B.epoch[r];
    const uint32_t updates = A.rvb_updates ? A.rvb_updates : (B.N + 1u) / 2u;
    const uint32_t *prod = B.rvb_prod + (size_t)r * B.rvb_prod_cap * SSE_RVB_PROD_STRIDE;
    // record of attempt 0
    for (uint32_t i = tid; i < SSE_RVB_PROD_SMALL; i += NT) LDSW(PB.o_pbuf, i) = updates ? prod[i] : 0u;
    __syncthreads();
    uint32_t gr = 0, nsucc = 0;
    constexpr int NPRE = (SSE_RVB_PROD_SMALL + NT - 1) / NT;
    for (uint32_t attempt = 0; attempt < updates; ++attempt) {
        uint32_t buf = PB.o_pbuf + (attempt & 1u) * SSE_RVB_PROD_SMALL;
        uint32_t pre[NPRE]; // the next record's first words: requested now, parked in LDS at the end of this attempt
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const uint32_t i = (uint32_t)(j * NT + tid);
            pre[j] = (attempt + 1u < updates && i < SSE_RVB_PROD_SMALL) ? prod[(size_t)(attempt + 1u) * SSE_RVB_PROD_STRIDE + i] : 0u;
        }
        const uint32_t gerr = LDSW(buf, GO_ERR);
        if (gerr) { err = gerr; break; } // (uniform: every thread reads the same word)
        const uint32_t nsub = LDSW(buf, GO_NSUB), nwin = LDSW(buf, GO_NWIN), ntog = LDSW(buf, GO_NTOG);
        const uint32_t words = rvb_prod_words(nsub, nwin, ntog);
        if (words > SSE_RVB_PROD_SMALL) { // a large area's product: the whole record, now
            const uint32_t *src = prod + (size_t)attempt * SSE_RVB_PROD_STRIDE;
            for (uint32_t i = tid; i < words; i += NT) LDSW(PB.o_big, i) = src[i];
            __syncthreads();
            buf = PB.o_big;
        }
        RvbDraw g;
        g.k0 = B.seed_lo; g.k1 = B.seed_hi; g.replica = B.rid ? B.rid[r] : B.replica_offset + r; g.epoch_lo = (uint32_t)epoch; g.attempt = attempt;
        g.k = LDSW(buf, GO_K);
        RvbLds R = R0;
        R.o_tog = buf + 8u; R.o_togs = R.o_tog + ntog; R.o_sub = R.o_togs + ntog; R.o_sfl = R.o_sub + nsub; R.o_wfrom = R.o_sfl + nsub; R.o_wuntil = R.o_wfrom + nwin;
        const bool stop = rvb_attempt<W, CL>(B, L, R0, R, r, g, nsub, nwin, ntog, M, gr, nsucc);
        if (stop) break; // (uniform; RC_ERR holds the code)
        const uint32_t nbuf = PB.o_pbuf + ((attempt + 1u) & 1u) * SSE_RVB_PROD_SMALL;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const uint32_t i = (uint32_t)(j * NT + tid);
            if (i < SSE_RVB_PROD_SMALL) LDSW(nbuf, i) = pre[j];
        }
        __syncthreads();
    }
    __syncthreads();
    if (!err && LDSW(R0.o_ctl, RC_ERR)) err = LDSW(R0.o_ctl, RC_ERR);
    __syncthreads();
    for (uint32_t i = tid; i < B.nwords; i += NT) B.state[(size_t)r * B.nwords + i] = LDSW(L.o_state, i);
    if (tid == 0) {
        B.err[r] = err; // (as sweep_kernel: the update counter and the attempt count move on whether or not the sweep ended early)
        B.epoch[r] = epoch + 1u;
        if (A.out_u32) A.out_u32[r] = nsucc;
        uint64_t *acc = B.acc + (size_t)B.acc_row[r] * 8;
        acc[4] += updates;
    }
}

} // namespace sse
