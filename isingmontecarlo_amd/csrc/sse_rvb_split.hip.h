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

// record of one attempt in HBM: [8] header (GO_*; word GO_NADJ: edges listed below, or SSE_RVB_NOADJ), bm[bmw] bond map (bit b + 1: bond b
// touches a sub-variable; bmw = ceil((Nb + 1) / 32)), tog[ntog], togs[ntog], sub[nsub], sfl[nsub], wfrom[nwin], wuntil[nwin], then — when the
// whole record fits the part that is fetched ahead — sadj[nsub + 1], adjb[nadj]: the edges at each sub-variable (bonds_for_var), so
// that the boundary-bond updates of the replay find them in LDS.  DevBatch::rvb_prod_stride = SSE_RVB_PROD_STRIDE + bmw words per attempt.
#define SSE_RVB_PROD_STRIDE (8u + 4u * SSE_RVB_MAXCL + 2u * SSE_RVB_MAXSUB + 2u * SSE_RVB_MAXWIN) // header + a large area's lists
#define SSE_RVB_PROD_AHEAD 640u // words behind the bond map that are fetched ahead (header + lists of any small growth area: 400; + edges)
static_assert(8u + 4u * SSE_RVB_SLOT_CL + 2u * SSE_RVB_SLOT_SUB + 2u * SSE_RVB_SLOT_WIN <= SSE_RVB_PROD_AHEAD, "small products are prefetched whole");
static_assert(SSE_RVB_PROD_AHEAD <= SSE_RVB_PROD_STRIDE, "");
#define SSE_RVB_BM_MAX 256u // bond-map words the two-launch form supports (Nb <= 8192: the map is built in the dead weight arrays of a
                            // small growth area; larger models take the fused kernel)
static_assert(SSE_RVB_BM_MAX <= 4u * SSE_RVB_SLOT_SET, "");
#define SSE_RVB_REGROW 0xFFFFFFFFu // header error word: the cluster outgrew its small area, the large one takes it
#define SSE_RVB_NOADJ 0xFFFFFFFFu
enum { GO_NADJ = 5 };
__host__ __device__ inline uint32_t rvb_bm_words(uint32_t Nb) { return (Nb + 32u) >> 5; } // bits 1 .. Nb: bond b at bit b + 1
__host__ __device__ inline uint32_t rvb_gcap_main(uint32_t W) { return W <= 4u ? 64u * SSE_RVB_UG4 : 512u; } // = 64 * UG of rvb_attempt<W, CL, true>: a wave's share of a scan step
// LDS words of the record region of the main launch: two fetched-ahead parts side by side, or one whole large record over both
__host__ __device__ inline uint32_t rvb_region_words(uint32_t bmw) {
    const uint32_t two = 2u * (SSE_RVB_PROD_AHEAD + bmw), big = SSE_RVB_PROD_STRIDE + bmw;
    return two > big ? two : big;
}

__device__ __forceinline__ uint32_t rvb_prod_words(uint32_t bmw, uint32_t nsub, uint32_t nwin, uint32_t ntog) { return 8u + bmw + 2u * ntog + 2u * nsub + 2u * nwin; }

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
    R.o_bk = R.o_bwb = R.o_bwa = R.o_glp = R.o_glw = R.o_gli = R.o_bix = R.o_bm = R.o_sadj = R.o_adjb = 0u; R.gcap = 0u; // (probability pass / mutation only)
    R.o_cps = base;
    R.cps_cap = B.lds_words > base ? B.lds_words - base : 0u;
}

// LDS of the main launch: [nwords] state, [2W] totals, [E] compact edges (CL), then the scratch of rvb_attempt and the record
// region: two prefetched small records side by side, or one large record over both (the next record waits in registers meanwhile)
__host__ __device__ inline uint32_t rvb_main_words(uint32_t W, uint32_t N, uint32_t nwords, uint32_t ledges, uint32_t E, uint32_t Nb) {
    return nwords + 2 * W + 16 + ledges + 2u + 4 * SSE_RVB_BONDCAP + (N + 1) / 2 + SSE_RVB_MAXSUB + SSE_RVB_BONDCAP + 3 * rvb_gcap_main(W) + (E + 1) / 2 + 16 +
           rvb_region_words(rvb_bm_words(Nb)) + 8;
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
    R.gcap = rvb_gcap_main(W);
    R.o_glp = base; base += R.gcap;
    R.o_glw = base; base += R.gcap;
    R.o_gli = base; base += R.gcap;
    R.o_bix = base; base += (B.E + 1) / 2;
    R.o_ctl = base; base += 16;
    P.o_pbuf = base; P.o_big = base; base += rvb_region_words(rvb_bm_words(B.Nb));
    R.o_bm = R.o_sadj = R.o_adjb = 0u; // set per attempt
    R.adj_lds = 0u; R.o_adjs = R.o_adj = 0u;
    R.o_vstart = R.o_zero = R.o_cps = R.o_gout = R.o_clv = R.o_clf = 0u; R.cps_cap = 0u; // (growth only)
    R.o_bfw = R.o_bnw = R.o_bfk = R.o_bfv = R.o_bnk = 0u;
    R.o_sub = R.o_sfl = R.o_tog = R.o_togs = R.o_wfrom = R.o_wuntil = 0u; // set per attempt
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, int lane) {
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    return x;
}
// one wave copies the products of a finished growth to the attempt's record and adds the bond map (built in the area's dead
// candidate-weight arrays) and, if the record stays within the part that the main launch fetches ahead, the edges at every sub-variable
__device__ __forceinline__ void rvb_store_product(const DevBatch &B, uint32_t *dst, const GrowArea &A, bool small_area, int lane) {
    const uint32_t nsub = LDSW(A.o_out, GO_NSUB), nwin = LDSW(A.o_out, GO_NWIN), ntog = LDSW(A.o_out, GO_NTOG), gerr = LDSW(A.o_out, GO_ERR);
    if (lane < 5) dst[lane] = (lane == GO_ERR && gerr && small_area) ? SSE_RVB_REGROW : LDSW(A.o_out, lane);
    if (gerr) return;
    const uint32_t bmw = rvb_bm_words(B.Nb), o_scr = A.o_bfw;
    for (uint32_t i = lane; i < bmw; i += 64u) LDSW(o_scr, i) = 0u;
    uint32_t off = 8u + bmw;
    for (uint32_t i = lane; i < ntog; i += 64u) { dst[off + i] = LDSW(A.o_tog, i); dst[off + ntog + i] = LDSW(A.o_togs, i); }
    off += 2u * ntog;
    for (uint32_t i = lane; i < nsub; i += 64u) { dst[off + i] = LDSW(A.o_sub, i); dst[off + nsub + i] = LDSW(A.o_sfl, i); }
    off += 2u * nsub;
    for (uint32_t i = lane; i < nwin; i += 64u) { dst[off + i] = LDSW(A.o_wfrom, i); dst[off + nwin + i] = LDSW(A.o_wuntil, i); }
    off += 2u * nwin;
    SSE_WAVE_FENCE();
    // edges: degrees first (is there room?), then the lists; every bond also sets its bit
    uint32_t nadj = 0;
    for (uint32_t base = 0; base < nsub; base += 64u) {
        const uint32_t sidx = base + (uint32_t)lane;
        const uint32_t v = LDSW(A.o_sub, sidx < nsub ? sidx : 0u);
        const uint32_t deg = sidx < nsub ? B.adj_start[v + 1] - B.adj_start[v] : 0u;
        nadj += __builtin_amdgcn_readlane((int)wave_incl_scan(deg, lane), 63);
    }
    const bool with_adj = off + nsub + 1u + nadj <= SSE_RVB_PROD_AHEAD + bmw;
    const uint32_t o_sadj = off, o_adjb = off + nsub + 1u;
    uint32_t run = 0;
    for (uint32_t base = 0; base < nsub; base += 64u) {
        const uint32_t sidx = base + (uint32_t)lane;
        const bool in = sidx < nsub;
        const uint32_t v = LDSW(A.o_sub, in ? sidx : 0u);
        const uint32_t a0 = B.adj_start[v], deg = in ? B.adj_start[v + 1] - a0 : 0u;
        const uint32_t incl = wave_incl_scan(deg, lane), mine = run + incl - deg;
        if (in) {
            uint32_t b = B.E + v + 1u; // the variable's own transverse (and longitudinal) bond; bit index = bond + 1
            atomicOr(&LDSW(o_scr, b >> 5), 1u << (b & 31u));
            if (B.has_long) { b += B.N; atomicOr(&LDSW(o_scr, b >> 5), 1u << (b & 31u)); }
            if (with_adj) dst[o_sadj + sidx] = mine;
        }
        for (uint32_t k = 0; sse_any(k < deg); ++k) {
            if (k < deg) {
                const uint32_t b = B.adj[a0 + k];
                atomicOr(&LDSW(o_scr, (b + 1u) >> 5), 1u << ((b + 1u) & 31u));
                if (with_adj) dst[o_adjb + mine + k] = b;
            }
        }
        run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) { dst[GO_NADJ] = with_adj ? nadj : SSE_RVB_NOADJ; if (with_adj) dst[o_sadj + nsub] = nadj; }
    SSE_WAVE_FENCE();
    for (uint32_t i = lane; i < bmw; i += 64u) dst[8u + i] = LDSW(o_scr, i);
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
    if (tid == 0) LDSW(R.o_ctl, RC_NEXTA) = 0u;
    __syncthreads();
    const uint32_t M = B.cutoff[r];
    const uint64_t epoch = B.epoch[r];
    const uint32_t updates = A.rvb_updates ? A.rvb_updates : (B.N + 1u) / 2u;
    uint32_t *prod = B.rvb_prod + (size_t)r * B.rvb_prod_cap * B.rvb_prod_stride;
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
    if ((uint32_t)wave < P) { // a wave takes the next attempt nobody has taken yet (the growths differ in length by an order of magnitude)
        const GrowArea As = grow_area_small(slots0 + (uint32_t)wave * SSE_RVB_SLOT_WORDS);
        for (;;) {
            uint32_t a = 0;
            if (lane == 0) a = atomicAdd(&LDSW(R.o_ctl, RC_NEXTA), 1u);
            a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
            if (a >= updates) break;
            g.attempt = a; g.k = 0;
            rvb_grow<W, CL, true>(B, L, R, As, g, C, nzero, M, lane);
            SSE_WAVE_FENCE();
            rvb_store_product(B, prod + (size_t)a * B.rvb_prod_stride, As, true, lane);
            SSE_WAVE_FENCE();
        }
    }
    __threadfence_block();
    __syncthreads();
    SSE_STAMP(7);
    // clusters that outgrew a small area (or all of them when there is no room for small areas): large areas — the fixed one and as
    // many as the room of the small ones, now idle, holds — one wave each
    if (tid == 0) LDSW(R.o_ctl, RC_NEXTA) = 0u;
    __syncthreads();
    uint32_t NL = 1u + (P * SSE_RVB_SLOT_WORDS) / SSE_RVB_LARGE_WORDS;
    if (NL > (uint32_t)W) NL = (uint32_t)W;
    if ((uint32_t)wave < NL) {
        const GrowArea big = wave == 0 ? grow_area_large(R) : grow_area_large_at(slots0 + ((uint32_t)wave - 1u) * SSE_RVB_LARGE_WORDS);
        for (;;) { // the next group of 64 attempts nobody has looked at yet
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&LDSW(R.o_ctl, RC_NEXTA), 64u);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base >= updates) break;
            const uint32_t a = base + (uint32_t)lane;
            uint64_t m = sse_ballot(a < updates && (P == 0u || prod[(size_t)a * B.rvb_prod_stride + GO_ERR] == SSE_RVB_REGROW));
            while (m) {
                const uint32_t k = (uint32_t)__ffsll((long long)m) - 1u;
                m &= m - 1;
                g.attempt = base + k; g.k = 0;
                rvb_grow<W, CL, false>(B, L, R, big, g, C, nzero, M, lane);
                SSE_WAVE_FENCE();
                rvb_store_product(B, prod + (size_t)(base + k) * B.rvb_prod_stride, big, false, lane);
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
    const uint32_t *prod = B.rvb_prod + (size_t)r * B.rvb_prod_cap * B.rvb_prod_stride;
    const uint32_t bmw = rvb_bm_words(B.Nb), small = SSE_RVB_PROD_AHEAD + bmw, stride = B.rvb_prod_stride; // small: the part of a record that is fetched ahead
    // record of attempt 0
    for (uint32_t i = tid; i < small; i += NT) LDSW(PB.o_pbuf, i) = updates ? prod[i] : 0u;
    __syncthreads();
    uint32_t gr = 0, nsucc = 0;
    constexpr int NPRE = (SSE_RVB_PROD_AHEAD + SSE_RVB_BM_MAX + NT - 1) / NT;
    for (uint32_t attempt = 0; attempt < updates; ++attempt) {
        sse_set_prio(attempt / 4u + blockIdx.x); // the replicas of a CU take turns at the top issue priority (the arbiter would serve the oldest workgroup first: sse_fast.hip.h)
        uint32_t buf = PB.o_pbuf + (attempt & 1u) * small;
        uint32_t pre[NPRE]; // the next record's first words: requested now, parked in LDS at the end of this attempt
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const uint32_t i = (uint32_t)(j * NT + tid);
            pre[j] = (attempt + 1u < updates && i < small) ? prod[(size_t)(attempt + 1u) * stride + i] : 0u;
        }
        const uint32_t gerr = LDSW(buf, GO_ERR);
        if (gerr) { err = gerr; break; } // (uniform: every thread reads the same word)
        const uint32_t nsub = LDSW(buf, GO_NSUB), nwin = LDSW(buf, GO_NWIN), ntog = LDSW(buf, GO_NTOG);
        const uint32_t nadj = LDSW(buf, GO_NADJ); // (listed only when the whole record lies in the part fetched ahead)
        const uint32_t words = rvb_prod_words(bmw, nsub, nwin, ntog);
        if (words > small) { // a large area's product: the whole record, now, over both small ones (every thread has read the header)
            const uint32_t *src = prod + (size_t)attempt * stride;
            __syncthreads();
            for (uint32_t i = tid; i < words; i += NT) LDSW(PB.o_big, i) = src[i];
            __syncthreads();
            buf = PB.o_big;
        }
        RvbDraw g;
        g.k0 = B.seed_lo; g.k1 = B.seed_hi; g.replica = B.rid ? B.rid[r] : B.replica_offset + r; g.epoch_lo = (uint32_t)epoch; g.attempt = attempt;
        g.k = LDSW(buf, GO_K);
        RvbLds R = R0;
        R.o_bm = buf + 8u;
        R.o_tog = buf + 8u + bmw; R.o_togs = R.o_tog + ntog; R.o_sub = R.o_togs + ntog; R.o_sfl = R.o_sub + nsub; R.o_wfrom = R.o_sfl + nsub; R.o_wuntil = R.o_wfrom + nwin;
        if (nadj != SSE_RVB_NOADJ) { R.o_sadj = R.o_wuntil + nwin; R.o_adjb = R.o_sadj + nsub + 1u; } else R.o_sadj = R.o_adjb = 0u;
        const bool stop = rvb_attempt<W, CL, true>(B, L, R0, R, r, g, nsub, nwin, ntog, M, gr, nsucc);
        if (stop) break; // (uniform; RC_ERR holds the code)
        const uint32_t nbuf = PB.o_pbuf + ((attempt + 1u) & 1u) * small;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const uint32_t i = (uint32_t)(j * NT + tid);
            if (i < small) LDSW(nbuf, i) = pre[j];
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
