// sse_rvb.hip.h — resonating-valence-bond update on gfx950 (included by sse_device.hip.h).
//
// Reference: RvbUpdater::rvb_update_with_ising_weight (src/sse/qmc_traits/rvb.rs:88-290) with build_cluster
// (:1054-1123), find_overlapping_starts (:1125-1158), find_constants (:1160-1187), calculate_flip_prob
// (:649-946), calculate_mult (:1194-1220), mutate_graph (:294-615), WeightedBoundaryManager (:967-1052) and
// BondContainer (src/util/bondcontainer.rs).
//
// The update is a chain of `updates` dependent attempts, each touching a handful of variables inside a few
// imaginary-time windows.  The reference walks per-variable linked lists (fast_ops.rs:1027-1172, :669-774);
// here the workgroup GATHERS what a window needs straight from the op-string:
//   * find_constants        -> workgroup counting sort of the transverse-op positions by variable (LDS);
//   * ops touching the sub-variables inside a window -> cooperative scan with wave64 ballot + prefix-sum
//     compaction into an LDS list, in p order, in batches;
//   * the state of the sub-variables at a window start -> cooperative backward search for the last op on
//     each of them;
// and one lane then replays the (short) sequential rule over the gathered list.  All scratch lives in the LDS
// region that the cluster pass uses for its union-find (the two passes never overlap).
#pragma once

namespace sse {

#ifndef SSE_RVB_ADJ_LDS
#define SSE_RVB_ADJ_LDS 0 // 1: stage the adjacency lists in LDS (faster reads, fewer growth areas)
#endif
#define SSE_RVB_MAXSUB 512u    // sub-variables (cluster + boundary) of one attempt
#define SSE_RVB_MAXCL 72u      // cluster members (trailing_ones(u64)+1 <= 65)
#define SSE_RVB_SETCAP 192u    // candidates in each weighted boundary set
#define SSE_RVB_BONDCAP 288u   // boundary bonds tracked at once
#define SSE_RVB_GCAP 512u      // gathered ops per batch (any size >= one wave's share of a gather step works)
#define SSE_RVB_MAXWIN 80u     // time windows of one attempt
#ifndef SSE_RVB_UL4
#define SSE_RVB_UL4 8  // main launch of the two-launch form with 4 waves: slots per lane of a look-back step ...
#endif
#ifndef SSE_RVB_UG4
#define SSE_RVB_UG4 15 // ... and of a window step (its gathered-op lists hold 64 * UG ops)
#endif

struct RvbLds { // word offsets into lds_raw
    uint32_t o_vstart;  // [N+1]
    uint32_t o_zero;    // [N]   variables without constant ops
    uint32_t o_v2s;     // [N] u16: sub-variable index or 0xFFFF
    uint32_t o_adjs;    // [N+1] u16 adjacency starts (when staged)
    uint32_t o_adj;     // [2E] u16 adjacency (when staged)
    uint32_t o_sub;     // [MAXSUB] variable of each sub-variable (sorted)
    uint32_t o_sfl;     // [MAXSUB] bit0 cluster_starting_state, bit1 cluster_state, bit2 substate
    uint32_t o_last;    // [MAXSUB] backward-search result (p+1 of the last op before a window)
    uint32_t o_clv, o_clf; // [MAXCL]
    uint32_t o_tog;     // [2*MAXCL] toggle positions (sorted)
    uint32_t o_togs;    // [2*MAXCL] sub-variable of each toggle
    uint32_t o_wfrom, o_wuntil; // [MAXWIN]
    uint32_t o_bfk, o_bfv, o_bfw; // flips set: key (= flip index), var, weight (double = 2 words)
    uint32_t o_bnk, o_bnw;        // no-flips set: key (= var), weight
    uint32_t o_bk, o_bwb, o_bwa;  // boundary bonds: key, weight before, weight after
    uint32_t o_glp, o_glw, o_gli; // gathered ops: slot, word, sub-variables and kind (SSE_GI_*)
    uint32_t o_bix;     // [E] u16: entry of each edge in the boundary-bond set, 0xFFFF = absent (probability pass)
    uint32_t o_ctl;     // [16] control words shared between the sequential lane and the workgroup
    uint32_t o_gout;    // [8] results of a growth in the large area
    uint32_t o_cps;     // [cps_cap] constant-op positions grouped by variable
    uint32_t cps_cap;
    uint32_t adj_lds;
    uint32_t gcap;      // entries of each gathered-op list (>= one wave's share of a gather step)
    uint32_t o_bm;      // [ceil(Nb/32)] bit per bond: it touches a sub-variable of the current attempt (two-launch form only: BM scans)
    uint32_t o_sadj, o_adjb; // [nsub+1] ranges into [..] the edges at each sub-variable, when the attempt's record carries them (0 = bonds_for_var in HBM)
};
// gathered-op info word: sub-variable of the first / second leg (SSE_GI_NONE = not a sub-variable), bond kind, two-site bit
#define SSE_GI_NONE 0x3FFu
#define SSE_GI_KIND_SHIFT 20
#define SSE_GI_TWO (1u << 22)
enum { RC_NSUB = 0, RC_NWIN = 1, RC_ACCEPT = 2, RC_GLEN = 3, RC_NEXTP = 4, RC_ERR = 5, RC_NZERO = 6, RC_SKIP = 7, RC_BROKE = 10, RC_NEXTA = 11 };

__device__ __forceinline__ double &ldsd(uint32_t off, uint32_t i) { return reinterpret_cast<double *>(lds_raw)[(off >> 1) + i]; }

// words of RVB scratch in front of the constant-op table (mirrors rvb_carve; used by the host to size LDS)
__host__ __device__ inline uint32_t rvb_fixed_words(uint32_t N, uint32_t E) {
    const uint32_t adj = SSE_RVB_ADJ_LDS && (N < 65535u && E < 65535u) ? (N + 2) / 2 + E : 0u;
    static_assert(3 * SSE_RVB_GCAP >= 7 * SSE_RVB_SETCAP, "the candidate sets of the large growth area live in the gathered-op lists");
    return 2u + 4 * SSE_RVB_BONDCAP + (N + 1) + N + (N + 1) / 2 + adj + 3 * SSE_RVB_MAXSUB +
           6 * SSE_RVB_MAXCL + 2 * SSE_RVB_MAXWIN + SSE_RVB_BONDCAP + 3 * SSE_RVB_GCAP + (E + 1) / 2 + 16 + 8;
}

template <int W>
__device__ __forceinline__ void rvb_carve(RvbLds &R, const Lds<W> &L, const DevBatch &B) {
    uint32_t base = (L.o_cur + 1u) & ~1u; // even: doubles are 8-byte aligned
    R.o_bwb = base; base += 2 * SSE_RVB_BONDCAP;
    R.o_bwa = base; base += 2 * SSE_RVB_BONDCAP;
    R.o_vstart = base; base += B.N + 1;
    R.o_zero = base; base += B.N;
    R.o_v2s = base; base += (B.N + 1) / 2;
    R.adj_lds = SSE_RVB_ADJ_LDS && (B.N < 65535u && B.E < 65535u) ? 1u : 0u;
    R.o_adjs = base; base += R.adj_lds ? (B.N + 2) / 2 : 0u;
    R.o_adj = base; base += R.adj_lds ? B.E : 0u; // 2E u16
    R.o_sub = base; base += SSE_RVB_MAXSUB;
    R.o_sfl = base; base += SSE_RVB_MAXSUB;
    R.o_last = base; base += SSE_RVB_MAXSUB;
    R.o_clv = base; base += SSE_RVB_MAXCL;
    R.o_clf = base; base += SSE_RVB_MAXCL;
    R.o_tog = base; base += 2 * SSE_RVB_MAXCL;
    R.o_togs = base; base += 2 * SSE_RVB_MAXCL;
    R.o_wfrom = base; base += SSE_RVB_MAXWIN;
    R.o_wuntil = base; base += SSE_RVB_MAXWIN;
    R.o_bk = base; base += SSE_RVB_BONDCAP;
    base = (base + 1u) & ~1u;
    // the candidate sets of the large growth area (weights first: doubles) share the gathered-op lists: an attempt is grown
    // before its first window is fetched, and the lists of the attempt before it are done with by then
    R.o_bfw = base; R.o_bnw = base + 2 * SSE_RVB_SETCAP; R.o_bfk = base + 4 * SSE_RVB_SETCAP; R.o_bfv = base + 5 * SSE_RVB_SETCAP; R.o_bnk = base + 6 * SSE_RVB_SETCAP;
    R.gcap = SSE_RVB_GCAP; R.o_bm = 0u; R.o_sadj = R.o_adjb = 0u;
    R.o_glp = base; base += SSE_RVB_GCAP;
    R.o_glw = base; base += SSE_RVB_GCAP;
    R.o_gli = base; base += SSE_RVB_GCAP;
    R.o_bix = base; base += (B.E + 1) / 2;
    R.o_ctl = base; base += 16;
    R.o_gout = base; base += 8;
    R.o_cps = base;
    R.cps_cap = B.lds_words > base ? B.lds_words - base : 0u;
}

// ---- adjacency (bonds_for_var: make_classical_bonds, qmc_ising.rs:421-432) ----
__device__ __forceinline__ uint32_t adj_begin(const RvbLds &R, const DevBatch &B, uint32_t v) {
    return R.adj_lds ? (uint32_t)LDSH(R.o_adjs, v) : B.adj_start[v];
}
__device__ __forceinline__ uint32_t adj_at(const RvbLds &R, const DevBatch &B, uint32_t i) {
    return R.adj_lds ? (uint32_t)LDSH(R.o_adj, i) : B.adj[i];
}

// the edges at sub-variable sv (= variable v): from the attempt's record in LDS when it carries them, else from HBM
__device__ __forceinline__ void sadj_range(const RvbLds &R, const DevBatch &B, uint32_t v, uint32_t sv, uint32_t &i0, uint32_t &i1) {
    if (R.o_sadj) { i0 = LDSW(R.o_sadj, sv); i1 = LDSW(R.o_sadj, sv + 1u); }
    else { i0 = adj_begin(R, B, v); i1 = adj_begin(R, B, v + 1); }
}
__device__ __forceinline__ uint32_t sadj_at(const RvbLds &R, const DevBatch &B, uint32_t i) { return R.o_sadj ? LDSW(R.o_adjb, i) : adj_at(R, B, i); }

struct RvbDraw {
    uint32_t k0, k1, replica, epoch_lo, attempt, k;
    __device__ __forceinline__ uint4 next() {
        return philox4x32_10(k++, epoch_lo, replica, (SSE_TAG_RVB << 24) | (attempt & 0xFFFFFFu), k0, k1);
    }
};
// The same stream for a wave that runs the rule uniformly (rvb_grow): lane l evaluates draw number base + l, so one Philox evaluation
// serves the next 64 draws of the attempt (a growth takes 4 + 2 per member) instead of one — the wave's 64 lanes would all have
// computed the same number.
struct RvbDrawW {
    RvbDraw g;
    uint4 c;       // this lane's draw: number base + lane
    uint32_t base; // 0xFFFFFFFF: nothing cached
    __device__ __forceinline__ void init(const RvbDraw &g0) { g = g0; base = 0xFFFFFFFFu; c = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ uint4 next(int lane) {
        if (base == 0xFFFFFFFFu || g.k - base >= 64u) { // (uniform)
            base = g.k;
            c = philox4x32_10(base + (uint32_t)lane, g.epoch_lo, g.replica, (SSE_TAG_RVB << 24) | (g.attempt & 0xFFFFFFu), g.k0, g.k1);
        }
        const int i = (int)(g.k - base);
        g.k++;
        return make_uint4((uint32_t)__builtin_amdgcn_readlane((int)c.x, i), (uint32_t)__builtin_amdgcn_readlane((int)c.y, i),
                          (uint32_t)__builtin_amdgcn_readlane((int)c.z, i), (uint32_t)__builtin_amdgcn_readlane((int)c.w, i));
    }
};

// x^n by squaring: the same multiplication sequence as the oracle
__device__ __forceinline__ double powi_sq(double x, uint32_t n) {
    double r = 1.0;
    while (n) { if (n & 1u) r *= x; x *= x; n >>= 1; }
    return r;
}

// ---- wave-uniform execution -------------------------------------------------------------------------------------
// The sequential rule of an attempt is run by ONE WAVE whose 64 lanes all hold the same scalars (the compiler keeps them in
// SGPRs and branches on them with s_cbranch): the lanes differ only inside the searches, where lane i looks at entry i and a
// ballot gives the answer — a linear search costs one LDS round trip instead of one per entry.  Stores write the same value
// from every lane (one LDS write).
// (x mod m) for x < 2m: the index arithmetic of the circular lists without an integer division
__device__ __forceinline__ uint32_t wrap(uint32_t x, uint32_t m) { return x >= m ? x - m : x; }
__device__ __forceinline__ double readlane_f64(double x, uint32_t l) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, (int)l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), (int)l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// index of `key` among keys[0, n), -1 when absent
__device__ __forceinline__ int wv_find(uint32_t o_key, uint32_t n, uint32_t key, int lane) {
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const bool in = i < n;
        const uint32_t k = LDSW(o_key, in ? i : 0u);
        const uint64_t m = sse_ballot(in & (k == key));
        if (m) return (int)base + __ffsll((long long)m) - 1;
    }
    return -1;
}
// weighted candidate set of the boundary manager, run by a whole wave (same arithmetic, in the same order, as WSet)
struct WSetW {
    uint32_t o_key, o_var, o_w; // o_var == 0xFFFFFFFF: the variable is the key
    uint32_t n, cap;
    double total;
    __device__ __forceinline__ void init(uint32_t ok, uint32_t ov, uint32_t ow, uint32_t c) { o_key = ok; o_var = ov; o_w = ow; n = 0; cap = c; total = 0.0; }
    __device__ __forceinline__ uint32_t key_at(uint32_t i) const { return LDSW(o_key, i); }
    __device__ __forceinline__ uint32_t var_at(uint32_t i) const { return LDSW(o_var != 0xFFFFFFFFu ? o_var : o_key, i); }
    __device__ __forceinline__ void dump_vars(uint32_t o_dst, uint32_t at, int lane) const { // dst[at + i] = variable of entry i
        for (uint32_t base = 0; base < n; base += 64u) { const uint32_t i = base + (uint32_t)lane; if (i < n) LDSW(o_dst, at + i) = var_at(i); }
    }
    __device__ __forceinline__ bool add(uint32_t key, uint32_t var, double w, int lane) { // BondContainer::insert of (existing weight + w) (rvb.rs:1044-1045)
        const int i = wv_find(o_key, n, key, lane);
        if (i >= 0) {
            const double old = ldsd(o_w, i), neww = old + w;
            total += neww - old;
            ldsd(o_w, i) = neww;
            return true;
        }
        if (n >= cap) return false;
        LDSW(o_key, n) = key;
        if (o_var != 0xFFFFFFFFu) LDSW(o_var, n) = var;
        const double neww = 0.0 + w;
        ldsd(o_w, n) = neww;
        total += neww;
        n++;
        return true;
    }
    __device__ __forceinline__ void remove_at(uint32_t i) { // swap-remove (bondcontainer.rs:55-72)
        const uint32_t last = n - 1;
        const double w = ldsd(o_w, i), wl = ldsd(o_w, last);
        const uint32_t kl = LDSW(o_key, last), vl = o_var != 0xFFFFFFFFu ? LDSW(o_var, last) : 0u;
        LDSW(o_key, i) = kl;
        if (o_var != 0xFFFFFFFFu) LDSW(o_var, i) = vl;
        ldsd(o_w, i) = wl;
        n--;
        total -= w;
        if (total < 0.0) total = 0.0;
    }
    __device__ __forceinline__ uint32_t pick(double u, int lane) const { // bondcontainer.rs:29-45: the running difference is sequential, the weights arrive 64 at a time
        double p = u * total;
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + (uint32_t)lane;
            const double w = ldsd(o_w, i < n ? i : 0u);
            const uint32_t cnt = n - base < 64u ? n - base : 64u;
            for (uint32_t j = 0; j < cnt; ++j) {
                p -= readlane_f64(w, j);
                if (p <= 0.0) return base + j;
            }
        }
        return n - 1;
    }
};

// The same set for at most 64 entries with the keys (and variables) held in registers, entry i in lane i: a lookup is a
// compare and a ballot, no LDS round trip; only the weights stay in LDS (read when an entry is drawn or changed).
struct WSetR {
    uint32_t keyv, varv, lane_; // lane_: this lane's index (writes to "entry i" are selects on it)
    uint32_t o_w;
    bool has_var;
    uint32_t n, cap;
    double total;
    __device__ __forceinline__ void init(uint32_t, uint32_t ov, uint32_t ow, uint32_t c) { keyv = 0u; varv = 0u; lane_ = __lane_id(); has_var = ov != 0xFFFFFFFFu; o_w = ow; n = 0; cap = c < 64u ? c : 64u; total = 0.0; }
    __device__ __forceinline__ uint32_t key_at(uint32_t i) const { return (uint32_t)__builtin_amdgcn_readlane((int)keyv, (int)i); }
    __device__ __forceinline__ uint32_t var_at(uint32_t i) const { return (uint32_t)__builtin_amdgcn_readlane((int)(has_var ? varv : keyv), (int)i); }
    __device__ __forceinline__ void dump_vars(uint32_t o_dst, uint32_t at, int lane) const { if ((uint32_t)lane < n) LDSW(o_dst, at + (uint32_t)lane) = has_var ? varv : keyv; }
    __device__ __forceinline__ bool add(uint32_t key, uint32_t var, double w, int lane) {
        const uint64_t m = sse_ballot(((uint32_t)lane < n) & (keyv == key));
        if (m) {
            const uint32_t i = (uint32_t)__ffsll((long long)m) - 1u;
            const double old = ldsd(o_w, i), neww = old + w;
            total += neww - old;
            ldsd(o_w, i) = neww;
            return true;
        }
        if (n >= cap) return false;
        keyv = (lane_ == n ? key : keyv);
        if (has_var) varv = (lane_ == n ? var : varv);
        const double neww = 0.0 + w;
        ldsd(o_w, n) = neww;
        total += neww;
        n++;
        return true;
    }
    __device__ __forceinline__ void remove_at(uint32_t i) {
        const uint32_t last = n - 1;
        const double w = ldsd(o_w, i), wl = ldsd(o_w, last);
        { const uint32_t kl = (uint32_t)__builtin_amdgcn_readlane((int)keyv, (int)last); keyv = lane_ == i ? kl : keyv; }
        if (has_var) { const uint32_t vl = (uint32_t)__builtin_amdgcn_readlane((int)varv, (int)last); varv = lane_ == i ? vl : varv; }
        ldsd(o_w, i) = wl;
        n--;
        total -= w;
        if (total < 0.0) total = 0.0;
    }
    __device__ __forceinline__ uint32_t pick(double u, int lane) const {
        double p = u * total;
        const double w = ldsd(o_w, (uint32_t)lane < n ? (uint32_t)lane : 0u);
        for (uint32_t j = 0; j < n; ++j) {
            p -= readlane_f64(w, j);
            if (p <= 0.0) return j;
        }
        return n - 1;
    }
};
// members of the growing cluster (variable, flip index or SSE_NO_VAR), in LDS or in registers
struct ClusterL {
    uint32_t o_v, o_f, n, cap;
    __device__ __forceinline__ void init(uint32_t ov, uint32_t of, uint32_t c) { o_v = ov; o_f = of; n = 0; cap = c; }
    __device__ __forceinline__ void push(uint32_t v, uint32_t f) { LDSW(o_v, n) = v; LDSW(o_f, n) = f; n++; }
    __device__ __forceinline__ uint32_t v_at(uint32_t i) const { return LDSW(o_v, i); }
    __device__ __forceinline__ uint32_t f_at(uint32_t i) const { return LDSW(o_f, i); }
    __device__ __forceinline__ bool contains(uint32_t var, uint32_t pos, int lane) const { // popped flags of the boundary manager (:1038-1043)
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + (uint32_t)lane, ii = i < n ? i : 0u;
            const uint32_t cf = LDSW(o_f, ii), cv = LDSW(o_v, ii);
            const bool hit = pos != SSE_NO_VAR ? cf == pos : (cf == SSE_NO_VAR && cv == var);
            if (sse_ballot((i < n) & hit)) return true;
        }
        return false;
    }
    __device__ __forceinline__ void dump_vars(uint32_t o_dst, uint32_t at, int lane) const {
        for (uint32_t base = 0; base < n; base += 64u) { const uint32_t i = base + (uint32_t)lane; if (i < n) LDSW(o_dst, at + i) = LDSW(o_v, i); }
    }
};
struct ClusterR {
    uint32_t vv, fv, lane_, n, cap;
    __device__ __forceinline__ void init(uint32_t, uint32_t, uint32_t c) { vv = 0u; fv = 0u; lane_ = __lane_id(); n = 0; cap = c < 64u ? c : 64u; }
    __device__ __forceinline__ void push(uint32_t v, uint32_t f) { vv = (lane_ == n ? v : vv); fv = (lane_ == n ? f : fv); n++; }
    __device__ __forceinline__ uint32_t v_at(uint32_t i) const { return (uint32_t)__builtin_amdgcn_readlane((int)vv, (int)i); }
    __device__ __forceinline__ uint32_t f_at(uint32_t i) const { return (uint32_t)__builtin_amdgcn_readlane((int)fv, (int)i); }
    __device__ __forceinline__ bool contains(uint32_t var, uint32_t pos, int lane) const {
        const bool hit = pos != SSE_NO_VAR ? fv == pos : (fv == SSE_NO_VAR && vv == var);
        return sse_ballot(((uint32_t)lane < n) & hit) != 0ull;
    }
    __device__ __forceinline__ void dump_vars(uint32_t o_dst, uint32_t at, int lane) const { if ((uint32_t)lane < n) LDSW(o_dst, at + (uint32_t)lane) = vv; }
};
template <bool REG> struct GrowTypes { typedef WSetW Set; typedef ClusterL Cluster; };
template <> struct GrowTypes<true> { typedef WSetR Set; typedef ClusterR Cluster; };

// ---- boundary-bond sets: one key list, weights before / after the flip ----
struct BSet {
    uint32_t o_key, o_wb, o_wa;
    uint32_t n;
    double tb, ta;
    __device__ __forceinline__ int find(uint32_t key) const {
        for (uint32_t i = 0; i < n; ++i) if (LDSW(o_key, i) == key) return (int)i;
        return -1;
    }
    __device__ __forceinline__ bool insert(uint32_t key, double wb, double wa) {
        const int i = find(key);
        if (i >= 0) {
            tb += wb - ldsd(o_wb, i); ldsd(o_wb, i) = wb;
            ta += wa - ldsd(o_wa, i); ldsd(o_wa, i) = wa;
            return true;
        }
        if (n >= SSE_RVB_BONDCAP) return false;
        LDSW(o_key, n) = key; ldsd(o_wb, n) = wb; ldsd(o_wa, n) = wa;
        tb += wb; ta += wa;
        n++;
        return true;
    }
    __device__ __forceinline__ void remove(uint32_t key) {
        const int i = find(key);
        if (i < 0) return;
        const double wb = ldsd(o_wb, i), wa = ldsd(o_wa, i);
        const uint32_t last = n - 1;
        LDSW(o_key, i) = LDSW(o_key, last); ldsd(o_wb, i) = ldsd(o_wb, last); ldsd(o_wa, i) = ldsd(o_wa, last);
        n--;
        tb -= wb; if (tb < 0.0) tb = 0.0;
        ta -= wa; if (ta < 0.0) ta = 0.0;
    }
    __device__ __forceinline__ uint32_t pick_before(double u) const {
        double p = u * tb;
        uint32_t i = 0;
        while (i < n) {
            p -= ldsd(o_wb, i);
            if (p <= 0.0) break;
            i++;
        }
        return i < n ? i : n - 1;
    }
};

__device__ __forceinline__ uint32_t v2s_get(const RvbLds &R, uint32_t v) { return (uint32_t)LDSH(R.o_v2s, v); }

// two-site diagonal weight of edge b for spins (sa, sb) (qmc_ising.rs:382-385 -> :863-875)
template <bool CL, int W>
__device__ __forceinline__ double rvb_edge_w(const DevBatch &B, const Lds<W> &L, uint32_t b, uint32_t sa, uint32_t sb, Bd *out = nullptr) {
    const Bd d = decode_bond<CL, W>(B, L, b);
    if (out) *out = d;
    const uint32_t s = (sa & 1u) | ((sb & 1u) << 1);
    return bond_weight(d, s, s);
}

// the "Now update bonds" block of calculate_flip_prob (rvb.rs:901-934) and mutate_graph (:560-592) for one variable
template <bool CL, int W>
__device__ __forceinline__ bool rvb_update_bonds(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t v, BSet &bs, bool with_after) {
    const uint32_t sv = v2s_get(R, v);
    if (sv == 0xFFFFu) return true;
    uint32_t i0, i1;
    sadj_range(R, B, v, sv, i0, i1);
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t b = sadj_at(R, B, i);
        const Bd d = decode_bond<CL, W>(B, L, b);
        const uint32_t ov = d.a == v ? d.c : d.a;
        const uint32_t so = v2s_get(R, ov);
        if (so == 0xFFFFu) continue;
        const uint32_t fv = LDSW(R.o_sfl, sv), fo = LDSW(R.o_sfl, so);
        if (((fv ^ fo) & 2u) == 0u) {
            bs.remove(b);
        } else {
            const uint32_t sa = v2s_get(R, d.a), sb = v2s_get(R, d.c);
            uint32_t ba = (LDSW(R.o_sfl, sa) >> 2) & 1u, bb = (LDSW(R.o_sfl, sb) >> 2) & 1u;
            const uint32_t s0 = ba | (bb << 1);
            const double wbef = bond_weight(d, s0, s0);
            double waft = 0.0;
            if (with_after) { // ws_for_flip (:665-683): flip the variable that is inside the cluster
                const uint32_t flipsub = (fv & 2u) ? sv : so;
                if (flipsub == sa) ba ^= 1u; else bb ^= 1u;
                const uint32_t s1 = ba | (bb << 1);
                waft = bond_weight(d, s1, s1);
            }
            if (!bs.insert(b, wbef, waft)) return false;
        }
    }
    return true;
}

// set_initial_bonds (rvb.rs:617-645) / the initial fill of mutate_graph (:366-380)
template <bool CL, int W>
__device__ __forceinline__ bool rvb_initial_bonds(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t nsub, BSet &bs, bool with_after) {
    for (uint32_t s = 0; s < nsub; ++s) {
        if (!(LDSW(R.o_sfl, s) & 2u)) continue;
        const uint32_t v = LDSW(R.o_sub, s);
        uint32_t i0, i1;
        sadj_range(R, B, v, s, i0, i1);
        for (uint32_t i = i0; i < i1; ++i) {
            const uint32_t b = sadj_at(R, B, i);
            const Bd d = decode_bond<CL, W>(B, L, b);
            const uint32_t ov = d.a == v ? d.c : d.a;
            const uint32_t so = v2s_get(R, ov);
            if (so == 0xFFFFu || (LDSW(R.o_sfl, so) & 2u)) continue;
            const uint32_t sa = v2s_get(R, d.a), sb = v2s_get(R, d.c);
            uint32_t ba = (LDSW(R.o_sfl, sa) >> 2) & 1u, bb = (LDSW(R.o_sfl, sb) >> 2) & 1u;
            const uint32_t s0 = ba | (bb << 1);
            const double wbef = bond_weight(d, s0, s0);
            double waft = 0.0;
            if (with_after) {
                if (s == sa) ba ^= 1u; else bb ^= 1u;
                const uint32_t s1 = ba | (bb << 1);
                waft = bond_weight(d, s1, s1);
            }
            if (!bs.insert(b, wbef, waft)) return false;
        }
    }
    return true;
}

// boundary-bond set run by a whole wave (probability pass): the entries of BSet, found through the edge -> entry table
struct BSetW {
    uint32_t o_key, o_wb, o_wa, o_ix;
    uint32_t n;
    double tb, ta;
    __device__ __forceinline__ bool insert(uint32_t key, double wb, double wa) {
        const uint32_t i = (uint32_t)LDSH(o_ix, key);
        if (i != 0xFFFFu) {
            const double ob = ldsd(o_wb, i), oa = ldsd(o_wa, i);
            tb += wb - ob; ldsd(o_wb, i) = wb;
            ta += wa - oa; ldsd(o_wa, i) = wa;
            return true;
        }
        if (n >= SSE_RVB_BONDCAP) return false;
        LDSW(o_key, n) = key; ldsd(o_wb, n) = wb; ldsd(o_wa, n) = wa;
        LDSH(o_ix, key) = (uint16_t)n;
        tb += wb; ta += wa;
        n++;
        return true;
    }
    __device__ __forceinline__ void remove(uint32_t key) {
        const uint32_t i = (uint32_t)LDSH(o_ix, key);
        if (i == 0xFFFFu) return;
        const uint32_t last = n - 1;
        const double wb = ldsd(o_wb, i), wa = ldsd(o_wa, i), wbl = ldsd(o_wb, last), wal = ldsd(o_wa, last);
        const uint32_t kl = LDSW(o_key, last);
        LDSW(o_key, i) = kl; ldsd(o_wb, i) = wbl; ldsd(o_wa, i) = wal;
        LDSH(o_ix, kl) = (uint16_t)i;
        LDSH(o_ix, key) = (uint16_t)0xFFFFu; // after the line above: the removed entry may be the last one itself
        n--;
        tb -= wb; if (tb < 0.0) tb = 0.0;
        ta -= wa; if (ta < 0.0) ta = 0.0;
    }
    __device__ __forceinline__ void clear(int lane) { // leave the table all-absent for the next attempt
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + (uint32_t)lane;
            if (i < n) LDSH(o_ix, LDSW(o_key, i)) = (uint16_t)0xFFFFu;
        }
        n = 0; tb = 0.0; ta = 0.0;
    }
};

// rvb_update_bonds / one variable of rvb_initial_bonds by a whole wave: lane k fetches neighbour k of v and works out what
// happens to their bond; the set is then changed in neighbour order.  initial: only bonds from the cluster to outside it
// are inserted (set_initial_bonds, rvb.rs:617-645), nothing is removed.
template <bool CL, int W>
__device__ __forceinline__ bool rvb_update_bonds_w(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t v, BSetW &bs, bool initial, int lane) {
    const uint32_t sv = v2s_get(R, v);
    if (sv == 0xFFFFu) return true;
    const uint32_t fv = LDSW(R.o_sfl, sv);
    uint32_t i0, i1;
    sadj_range(R, B, v, sv, i0, i1);
    for (uint32_t base = i0; base < i1; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const bool in = i < i1;
        const uint32_t b = sadj_at(R, B, in ? i : i0);
        const Bd d = decode_bond<CL, W>(B, L, b);
        const uint32_t ov = d.a == v ? d.c : d.a;
        const uint32_t so = v2s_get(R, ov);
        const bool has = in & (so != 0xFFFFu);
        const uint32_t fo = LDSW(R.o_sfl, has ? so : 0u);
        const bool same = ((fv ^ fo) & 2u) == 0u;
        const uint32_t act = !has ? 0u : (initial ? ((fo & 2u) ? 0u : 2u) : (same ? 1u : 2u)); // 1 remove, 2 insert
        const uint32_t sa = d.a == v ? sv : so; // sub-variables of the bond's first / second site
        const uint32_t fa = d.a == v ? fv : fo, fb = d.a == v ? fo : fv;
        uint32_t ba = (fa >> 2) & 1u, bb = (fb >> 2) & 1u;
        const uint32_t s0 = ba | (bb << 1);
        const double wbef = bond_weight(d, s0, s0);
        const uint32_t flipsub = (fv & 2u) ? sv : so; // ws_for_flip (:665-683): flip the variable that is inside the cluster
        if (flipsub == sa) ba ^= 1u; else bb ^= 1u;
        const uint32_t s1 = ba | (bb << 1);
        const double waft = bond_weight(d, s1, s1);
        uint64_t m = sse_ballot(act != 0u);
        while (m) {
            const uint32_t k = (uint32_t)__ffsll((long long)m) - 1u;
            m &= m - 1;
            const uint32_t bk = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)k), ak = (uint32_t)__builtin_amdgcn_readlane((int)act, (int)k);
            if (ak == 1u) bs.remove(bk);
            else if (!bs.insert(bk, readlane_f64(wbef, k), readlane_f64(waft, k))) return false;
        }
    }
    return true;
}

// find_overlapping_starts (rvb.rs:1125-1158) over cps[fp0 .. fp0+Lf) by a whole wave: the position lists are sorted, so the first entry >= p_start is a count, and the
// run of overlapping segments ends at the first step whose test fails.  Calls f(index) for each overlapping segment, in order.
template <typename F>
__device__ __forceinline__ void rvb_overlaps_w(const RvbLds &R, uint32_t p_start, uint32_t p_end, uint32_t cutoff, uint32_t fp0, uint32_t Lf, int lane, F f) {
    uint32_t bin = 0;
    for (uint32_t base = 0; base < Lf; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const uint32_t x = LDSW(R.o_cps, fp0 + (i < Lf ? i : 0u));
        bin += (uint32_t)popc64(sse_ballot((i < Lf) & (x < p_start)));
    }
    const uint32_t prev = wrap(bin + Lf - 1, Lf);
    const uint32_t lowest = LDSW(R.o_cps, fp0 + prev);
    const uint32_t off_start = wrap(p_start + cutoff - lowest, cutoff), off_end = wrap(p_end + cutoff - lowest, cutoff); // (positions are below the cutoff)
    uint32_t count = Lf;
    for (uint32_t base = 0; base < Lf; base += 64u) {
        const uint32_t step = base + (uint32_t)lane;
        const bool in = step < Lf;
        const uint32_t ip = wrap(prev + (in ? step : 0u), Lf);
        const uint32_t p = LDSW(R.o_cps, fp0 + ip);
        const uint32_t next_p = LDSW(R.o_cps, fp0 + wrap(ip + 1, Lf));
        const uint32_t check_start = wrap(p + cutoff - lowest, cutoff), check_end = wrap(next_p + cutoff - lowest, cutoff);
        const bool has_overlap_start = check_start < off_start && off_start < check_end;
        const bool has_start_within = off_start < check_start && check_start < off_end;
        const bool eq = (p_start == p_end) || (check_start == check_end);
        const uint64_t stop = sse_ballot(in & !(eq || has_overlap_start || has_start_within));
        if (stop) { count = base + (uint32_t)__ffsll((long long)stop) - 1u; break; }
    }
    for (uint32_t step = 0; step < count; ++step) f(wrap(prev + step, Lf));
}

// ---------------------------------------------------------------------------------------------
// cooperative pieces

// find_constants (rvb.rs:1160-1187): counting sort of the transverse-op positions by variable.
// Returns C (number of constant ops) or 0xFFFFFFFF when the table does not fit in LDS.
template <int W, bool CL>
__device__ __forceinline__ uint32_t rvb_find_constants(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t r, uint32_t M) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x;
    const uint32_t N = B.N;
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    for (uint32_t v = tid; v < N; v += NT) { LDSW(R.o_zero, v) = 0u; LDSH(R.o_v2s, v) = (uint16_t)0xFFFFu; }
    if (R.adj_lds) {
        for (uint32_t v = tid; v <= N; v += NT) LDSH(R.o_adjs, v) = (uint16_t)B.adj_start[v];
        for (uint32_t i = tid; i < 2 * B.E; i += NT) LDSH(R.o_adj, i) = (uint16_t)B.adj[i];
    }
    __syncthreads();
    // per-variable counts (temporarily in o_zero)
    // (Ising bond tables hold the E edges first, then the N transverse bonds: bond E + v is variable v's — no decode needed here)
    for (uint32_t p = tid; p < M; p += NT) {
        const uint32_t wd = ops[p];
        const uint32_t v = (wd >> SSE_OP_BOND_SHIFT) - 1u - B.E; // (an empty slot wraps to a huge value)
        if (v < N) atomicAdd(&LDSW(R.o_zero, v), 1u);
    }
    __syncthreads();
    // exclusive prefix over the variables, a block of NT at a time (wave scans + the waves' totals through o_tot); the counts array
    // becomes the list of variables without constant ops, in increasing order: entry nz <= v is written only after the whole block
    // holding v has been read
    {
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        uint32_t run = 0, nz = 0;
        for (uint32_t v0 = 0; v0 < N; v0 += NT) {
            const uint32_t v = v0 + (uint32_t)tid;
            const uint32_t c = v < N ? LDSW(R.o_zero, v) : 0u;
            const bool z = (v < N) & (c == 0u);
            uint32_t inc = c;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if (lane >= o) inc += y; }
            const uint64_t zm = sse_ballot(z);
            if (lane == 63) LDSW(L.o_tot, wave) = inc;
            if (lane == 0) LDSW(L.o_tot, W + wave) = (uint32_t)popc64(zm);
            __syncthreads();
            uint32_t cbase = 0, zbase = 0, ctot = 0, ztot = 0;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) {
                const uint32_t a = LDSW(L.o_tot, w2), b = LDSW(L.o_tot, W + w2);
                if (w2 < wave) { cbase += a; zbase += b; }
                ctot += a; ztot += b;
            }
            if (v < N) LDSW(R.o_vstart, v) = run + cbase + inc - c;
            if (z) LDSW(R.o_zero, nz + zbase + (uint32_t)popc64(zm & lanemask_lt(lane))) = v;
            run += ctot; nz += ztot;
            __syncthreads();
        }
        if (tid == 0) { LDSW(R.o_vstart, N) = run; LDSW(R.o_ctl, RC_NZERO) = nz; }
    }
    __syncthreads();
    const uint32_t C = LDSW(R.o_vstart, N);
    if (C > R.cps_cap) return 0xFFFFFFFFu;
    // fill: vstart[v] doubles as the cursor and ends at the old vstart[v+1]; shift it back afterwards
    for (uint32_t p = tid; p < M; p += NT) {
        const uint32_t wd = ops[p];
        const uint32_t v = (wd >> SSE_OP_BOND_SHIFT) - 1u - B.E;
        if (v < N) LDSW(R.o_cps, atomicAdd(&LDSW(R.o_vstart, v), 1u)) = p;
    }
    __syncthreads();
    // vstart[v+1] := cursor[v], highest block first so that every element is read before it is overwritten
    for (int64_t v0 = (int64_t)((N - 1) / NT) * NT; v0 >= 0; v0 -= NT) {
        const uint32_t v = (uint32_t)v0 + tid;
        const uint32_t x = v < N ? LDSW(R.o_vstart, v) : 0u;
        __syncthreads();
        if (v < N) LDSW(R.o_vstart, v + 1) = x;
        __syncthreads();
    }
    if (tid == 0) LDSW(R.o_vstart, 0) = 0u;
    __syncthreads();
    for (uint32_t v = tid; v < N; v += NT) { // insertion sort: the lists hold ~ beta*Gamma entries each
        const uint32_t s = LDSW(R.o_vstart, v), e = LDSW(R.o_vstart, v + 1);
        for (uint32_t i = s + 1; i < e; ++i) {
            const uint32_t x = LDSW(R.o_cps, i);
            uint32_t j = i;
            while (j > s && LDSW(R.o_cps, j - 1) > x) { LDSW(R.o_cps, j) = LDSW(R.o_cps, j - 1); j--; }
            LDSW(R.o_cps, j) = x;
        }
    }
    __syncthreads();
    return C;
}

// ---- scans through the attempt's bond map (BM) ----
// The scans below visit every slot of a window and of the look-back span in front of it, and all but a few percent of the ops
// they see touch no sub-variable.  With a bit per BOND (set by the growth launch for every bond of every sub-variable) the test
// is one LDS read and a shift per slot; the decode and the variable -> sub-variable lookups are then done once per HIT, densely
// (64 hits per wave instruction), instead of once per slot.
// (the map is indexed by bond + 1 — the op word's bond field as it stands — and bit 0 stays clear: an empty slot needs no test of its own)
__device__ __forceinline__ bool bm_hit(const RvbLds &R, uint32_t wd) {
    const uint32_t q = wd >> SSE_OP_BOND_SHIFT;
    return ((LDSW(R.o_bm, q >> 5) >> (q & 31u)) & 1u) != 0u;
}
// Rows of a scan: slots p0 + 64 j, j < U, of slots [0, end) of a replica's op-string through a buffer resource — the hardware returns 0
// for a slot at or beyond `end` (an empty slot to every consumer), the row offsets are immediates: no vector instruction per row for
// address or bounds.  (p0 < 2^30)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ops_window(const uint32_t *ops, uint32_t end) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)ops, 0, (int)(end * 4u), 0x00020000);
}
template <int U>
__device__ __forceinline__ void load_rows(__amdgpu_buffer_rsrc_t rs, uint32_t p0, uint32_t (&w)[U]) {
    static_assert(U <= 16, "row offsets are 12-bit immediates");
    const int v = (int)(p0 * 4u);
#pragma unroll
    for (int j = 0; j < U; ++j) w[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, v + j * 256, 0, 0);
}

// info word of a gathered op (sub-variables of its legs, bond kind)
template <int W, bool CL>
__device__ __forceinline__ uint32_t rvb_info_word(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t wd) {
    const Bd d = decode_bond<CL, W>(B, L, wd ? sse_op_bond(wd) : 0u);
    const bool two = d.c != SSE_NO_VAR;
    const uint32_t sa = v2s_get(R, d.a), sc = v2s_get(R, two ? d.c : d.a);
    const bool ma = sa != 0xFFFFu, mc = two & (sc != 0xFFFFu);
    return (ma ? sa : SSE_GI_NONE) | ((mc ? sc : SSE_GI_NONE) << 10) | (bd_kind(d) << SSE_GI_KIND_SHIFT) | (two ? SSE_GI_TWO : 0u);
}
// last-op search: one op at slot p
template <int W, bool CL>
__device__ __forceinline__ void rvb_last_op(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t p, uint32_t wd) {
    const Bd d = decode_bond<CL, W>(B, L, sse_op_bond(wd));
    const uint32_t sa = v2s_get(R, d.a);
    if (sa != 0xFFFFu) atomicMax(&LDSW(R.o_last, sa), ((p + 1u) << 1) | (sse_op_out(wd) & 1u));
    if (d.c != SSE_NO_VAR) {
        const uint32_t sc = v2s_get(R, d.c);
        if (sc != 0xFFFFu) atomicMax(&LDSW(R.o_last, sc), ((p + 1u) << 1) | ((sse_op_out(wd) >> 1) & 1u));
    }
}
// last-op search over the U rows a lane holds (row j = slot lo + wave*64*U + j*64 + lane; slots at or beyond the span's end arrive as 0) through the bond map: the wave
// parks its hits in its own part of the (then idle) gathered-op lists and works them off densely; a wave with more hits than its
// part holds takes them lane by lane.  lists_free = false: the lists are in use (the rare longer look-back of rvb_fetch).
template <int W, bool CL, int U>
__device__ __forceinline__ void rvb_last_ops_bm(const DevBatch &B, const Lds<W> &L, const RvbLds &R, const uint32_t (&wl)[U], uint32_t lo, bool lists_free) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t p0 = lo + (uint32_t)(wave * 64 * U + lane);
    uint64_t hm[U];
    uint32_t nh = 0;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        hm[j] = sse_ballot(bm_hit(R, wl[j]));
        nh += (uint32_t)popc64(hm[j]);
    }
    const uint32_t seg = R.gcap / (uint32_t)W, s0 = (uint32_t)wave * seg;
    if (lists_free && nh <= seg) {
        uint32_t run = s0;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if ((hm[j] >> lane) & 1ull) {
                const uint32_t idx = run + popc64(hm[j] & lanemask_lt(lane));
                LDSW(R.o_glp, idx) = p0 + (uint32_t)(j * 64);
                LDSW(R.o_glw, idx) = wl[j];
            }
            run += (uint32_t)popc64(hm[j]);
        }
        SSE_WAVE_FENCE();
        for (uint32_t base = 0; base < nh; base += 64u) {
            const uint32_t i = base + (uint32_t)lane;
            if (i < nh) rvb_last_op<W, CL>(B, L, R, LDSW(R.o_glp, s0 + i), LDSW(R.o_glw, s0 + i));
        }
        SSE_WAVE_FENCE();
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j)
            if ((hm[j] >> lane) & 1ull) rvb_last_op<W, CL>(B, L, R, p0 + (uint32_t)(j * 64), wl[j]);
    }
}

// Gather the ops of slots [gp, until] that touch a sub-variable, in p order, into the LDS list (one batch).
// Returns through RC_GLEN / RC_NEXTP (slot to resume from, until+1 when the window is exhausted).  A step covers U*NT
// slots, wave-major (a wave's lanes hold U*64 consecutive slots), and requests the next step's words before it works on
// its own; a first step that alone overflows the list is cut at a wave boundary, so any list of >= U*64 entries is enough.
// the words of a gather's first step (requested early by the caller where it has something else to do meanwhile)
template <int W, int U>
__device__ __forceinline__ void rvb_gather_rows(const DevBatch &B, uint32_t r, uint32_t gp, uint32_t until, uint32_t M, uint32_t (&wd)[U]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    const uint32_t end = M == 0u ? 0u : (until < M ? until + 1u : M); // slots [gp, end)
    load_rows<U>(ops_window(ops, end), gp + (uint32_t)(wave * 64 * U + lane), wd);
}
template <int W, bool CL, int U = 8, bool BM = false>
__device__ __forceinline__ void rvb_gather(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t r, uint32_t gp, uint32_t until,
                                           uint32_t M, uint32_t &gr, uint32_t (&wd)[U]) {
    constexpr int NT = W * 64;
    static_assert(SSE_RVB_GCAP >= 64u * 8u, "the gathered-op list must hold one wave's share of a step (R.gcap >= 64 * U: the carve functions see to it)");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: keeps per-wave control flow uniform
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    const uint32_t last = until < M ? until : M - 1; // inclusive
    const __amdgpu_buffer_rsrc_t win = ops_window(ops, M == 0u ? 0u : last + 1u);
    uint32_t glen = 0, next = gp;
    while (M != 0u && next <= last) {
        uint32_t wn[U], info[U];
        uint64_t mm[U];
        int cnt = 0;
        load_rows<U>(win, next + (uint32_t)(U * NT) + (uint32_t)(wave * 64 * U + lane), wn);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if constexpr (BM) { info[j] = 0u; mm[j] = sse_ballot(bm_hit(R, wd[j])); } // (the info words follow once the list is complete)
            else {
                info[j] = rvb_info_word<W, CL>(B, L, R, wd[j]);
                mm[j] = sse_ballot((wd[j] != 0u) & (((info[j] & SSE_GI_NONE) != SSE_GI_NONE) | (((info[j] >> 10) & SSE_GI_NONE) != SSE_GI_NONE)));
            }
            cnt += popc64(mm[j]);
        }
        const int buf = gr & 1;
        if (lane == 0) LDSI(L.o_tot, buf * W + wave) = cnt;
        __syncthreads();
        gr++;
        uint32_t wbase = 0, total = 0, fit_waves = 0, fit_total = 0;
#pragma unroll
        for (int w2 = 0; w2 < W; ++w2) {
            const uint32_t t = (uint32_t)LDSI(L.o_tot, buf * W + w2);
            if (w2 < wave) wbase += t;
            total += t;
            if (fit_waves == (uint32_t)w2 && total <= R.gcap) { fit_waves = (uint32_t)w2 + 1u; fit_total = total; } // longest prefix of waves that fits an empty list
        }
        const bool whole = glen + total <= R.gcap;
        if (!whole && glen != 0u) break; // this step opens the next batch
        if (whole || (uint32_t)wave < fit_waves) {
            uint32_t run = glen + wbase;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if ((mm[j] >> lane) & 1ull) {
                    const uint32_t idx = run + popc64(mm[j] & lanemask_lt(lane));
                    LDSW(R.o_glp, idx) = next + (uint32_t)(wave * 64 * U + j * 64 + lane);
                    LDSW(R.o_glw, idx) = wd[j];
                    if constexpr (!BM) LDSW(R.o_gli, idx) = info[j];
                }
                run += popc64(mm[j]);
            }
        }
        if (!whole) { glen = fit_total; next += fit_waves * (uint32_t)(64 * U); break; }
        glen += total;
        next += U * NT;
#pragma unroll
        for (int j = 0; j < U; ++j) wd[j] = wn[j];
    }
    __syncthreads();
    if constexpr (BM)
        for (uint32_t i = tid; i < glen; i += NT) LDSW(R.o_gli, i) = rvb_info_word<W, CL>(B, L, R, LDSW(R.o_glw, i));
    if (tid == 0) { LDSW(R.o_ctl, RC_GLEN) = glen; LDSW(R.o_ctl, RC_NEXTP) = (M == 0u || next > last) ? last + 1 : next; }
    __syncthreads();
}

// Probability pass, one window: the spins of the sub-variables at its start (backward search for the last op on each, as
// rvb_state_at) and the first batch of its ops (as rvb_gather, plus the info word of each) from ONE round of loads — the
// look-back chunk and the first window chunk are requested together, then one barrier publishes both.  o_last must be all
// zero on entry and is left all zero.  RC_GLEN / RC_NEXTP as rvb_gather (a first chunk that overflows the list is left to
// rvb_gather's smaller steps: RC_GLEN 0, RC_NEXTP = from).
template <int W, bool CL, int UL = 4, int UG = 8, bool BM = false>
__device__ __forceinline__ void rvb_fetch(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t r, uint32_t from, uint32_t until,
                                          uint32_t M, uint32_t nsub, uint32_t &gr) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    const uint32_t last = until < M ? until : M - 1; // inclusive
    const uint32_t lo_b = from > (uint32_t)(UL * NT) ? from - (uint32_t)(UL * NT) : 0u; // look-back chunk [lo_b, from)
    SSE_STAMP_INIT; // diagnostic builds: 0 loads + matching, 1 compaction, 2 longer look-back, 3 states
    uint32_t wl[UL], wg[UG];
    load_rows<UL>(ops_window(ops, from), lo_b + (uint32_t)(wave * 64 * UL + lane), wl);
    load_rows<UG>(ops_window(ops, M == 0u ? 0u : last + 1u), from + (uint32_t)(wave * 64 * UG + lane), wg);
    if constexpr (BM) rvb_last_ops_bm<W, CL, UL>(B, L, R, wl, lo_b, true);
    else {
#pragma unroll
        for (int j = 0; j < UL; ++j)
            if (wl[j]) rvb_last_op<W, CL>(B, L, R, lo_b + (uint32_t)(wave * 64 * UL + j * 64 + lane), wl[j]);
    }
    uint32_t info[UG];
    uint64_t mm[UG];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < UG; ++j) {
        if constexpr (BM) { info[j] = 0u; mm[j] = sse_ballot(bm_hit(R, wg[j])); } // (the info words follow once the list is complete)
        else {
            info[j] = rvb_info_word<W, CL>(B, L, R, wg[j]);
            mm[j] = sse_ballot((wg[j] != 0u) & (((info[j] & SSE_GI_NONE) != SSE_GI_NONE) | (((info[j] >> 10) & SSE_GI_NONE) != SSE_GI_NONE)));
        }
        cnt += popc64(mm[j]);
    }
    const int buf = gr & 1;
    if (lane == 0) LDSI(L.o_tot, buf * W + wave) = cnt;
    __syncthreads();
    SSE_STAMP(0);
    gr++;
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < W; ++w2) { const uint32_t t = (uint32_t)LDSI(L.o_tot, buf * W + w2); if (w2 < wave) wbase += t; total += t; }
    const bool fits = total <= R.gcap;
    if (fits) {
        uint32_t run = wbase;
#pragma unroll
        for (int j = 0; j < UG; ++j) {
            if ((mm[j] >> lane) & 1ull) {
                const uint32_t idx = run + popc64(mm[j] & lanemask_lt(lane));
                LDSW(R.o_glp, idx) = from + (uint32_t)(wave * 64 * UG + j * 64 + lane);
                LDSW(R.o_glw, idx) = wg[j];
                if constexpr (!BM) LDSW(R.o_gli, idx) = info[j];
            }
            run += popc64(mm[j]);
        }
    }
    if (lo_b > 0u) { // sub-variables without an op in the look-back chunk: keep searching
        uint32_t missing = 0;
        for (uint32_t s2 = tid; s2 < nsub; s2 += NT) missing |= (LDSW(R.o_last, s2) == 0u);
        if (missing) LDSW(R.o_ctl, RC_SKIP) = 1u;
    }
    if (tid == 0) {
        const uint32_t next = from + (uint32_t)(UG * NT);
        LDSW(R.o_ctl, RC_GLEN) = fits ? total : 0u;
        LDSW(R.o_ctl, RC_NEXTP) = M == 0u ? last + 1 : (!fits ? from : (next > last ? last + 1 : next));
    }
    __syncthreads();
    if constexpr (BM)
        if (fits) for (uint32_t i = tid; i < total; i += NT) LDSW(R.o_gli, i) = rvb_info_word<W, CL>(B, L, R, LDSW(R.o_glw, i)); // (published by the barrier at the end)
    SSE_STAMP(1);
    uint32_t hi = lo_b;
    while (LDSW(R.o_ctl, RC_SKIP)) { // (rare) the chunks before, one barrier round each, as rvb_state_at
        __syncthreads();
        if (tid == 0) LDSW(R.o_ctl, RC_SKIP) = 0u;
        const uint32_t span = (uint32_t)(UL * NT);
        const uint32_t lo = hi > span ? hi - span : 0u;
        uint32_t wx[UL];
        load_rows<UL>(ops_window(ops, hi), lo + (uint32_t)(wave * 64 * UL + lane), wx);
        if constexpr (BM) rvb_last_ops_bm<W, CL, UL>(B, L, R, wx, lo, false);
        else {
#pragma unroll
            for (int j = 0; j < UL; ++j)
                if (wx[j]) rvb_last_op<W, CL>(B, L, R, lo + (uint32_t)(wave * 64 * UL + j * 64 + lane), wx[j]);
        }
        __syncthreads();
        hi = lo;
        if (hi > 0u) {
            uint32_t missing = 0;
            for (uint32_t s2 = tid; s2 < nsub; s2 += NT) missing |= (LDSW(R.o_last, s2) == 0u);
            if (missing) LDSW(R.o_ctl, RC_SKIP) = 1u;
        }
        __syncthreads();
    }
    SSE_STAMP(2);
    for (uint32_t s2 = tid; s2 < nsub; s2 += NT) {
        const uint32_t x = LDSW(R.o_last, s2);
        const uint32_t v = LDSW(R.o_sub, s2);
        const uint32_t bit = x ? (x & 1u) : ((LDSW(L.o_state, v >> 5) >> (v & 31)) & 1u);
        LDSW(R.o_sfl, s2) = (LDSW(R.o_sfl, s2) & 3u) | (bit << 2);
        LDSW(R.o_last, s2) = 0u;
    }
    __syncthreads();
    SSE_STAMP(3);
}

// Probability pass over one batch of gathered ops (calculate_flip_prob, rvb.rs:649-946), by a whole wave.  Only the ops that
// change something — off-diagonal ops and the cluster's own toggles — are visited one by one; the diagonal ops between two
// of them are counted (those sitting on a boundary bond) and checked (longitudinal ops inside the cluster) 64 at a time
// against the boundary set as it stands.  Returns true when the product has dropped to zero (`broke`).
template <int W, bool CL>
__device__ __forceinline__ bool rvb_replay_prob(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t glen, uint32_t ntog, uint32_t &next_tog,
                                                uint32_t &nb, double &mult, BSetW &bs, int lane) {
    const double EPS = 2.220446049250313e-16;
    for (uint32_t base = 0; base < glen; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const bool valid = i < glen;
        const uint32_t cnt = glen - base < 64u ? glen - base : 64u;
        const uint32_t p = LDSW(R.o_glp, valid ? i : 0u), wd = LDSW(R.o_glw, valid ? i : 0u), info = LDSW(R.o_gli, valid ? i : 0u);
        const uint32_t b = sse_op_bond(wd), in = sse_op_in(wd), out = sse_op_out(wd);
        const uint32_t sa = info & SSE_GI_NONE, sc = (info >> 10) & SSE_GI_NONE, kind = (info >> SSE_GI_KIND_SHIFT) & 3u;
        const bool two = (info & SSE_GI_TWO) != 0u, isedge = b < B.E, offdiag = in != out;
        bool isb = false; // p is one of the toggles still ahead (both lists ascend)
        for (uint32_t t = next_tog; t < ntog; ++t) isb |= p == LDSW(R.o_tog, t);
        const uint64_t evm0 = sse_ballot(valid & (offdiag | isb));
        uint32_t pos = 0;
        for (;;) {
            const bool inb = valid & isedge & ((uint32_t)LDSH(R.o_bix, (valid & isedge) ? b : 0u) != 0xFFFFu);
            const uint64_t inbm = sse_ballot(inb);
            const uint64_t ahead = ~lanemask_lt((int)pos) & (cnt < 64u ? lanemask_lt((int)cnt) : ~0ull);
            const uint64_t ev = evm0 & ~inbm & ahead;
            const uint32_t e = ev ? (uint32_t)__ffsll((long long)ev) - 1u : cnt;
            const uint64_t seg = ahead & (e < 64u ? lanemask_lt((int)e) : ~0ull);
            if (B.has_long) { // ising_ratio (qmc_ising.rs:722-735): a longitudinal op inside the cluster zeroes the product
                const bool a_in = (sa != SSE_GI_NONE) && (LDSW(R.o_sfl, sa != SSE_GI_NONE ? sa : 0u) & 2u);
                const bool c_in = (sc != SSE_GI_NONE) && (LDSW(R.o_sfl, sc != SSE_GI_NONE ? sc : 0u) & 2u);
                const bool all_in = a_in && (!two || c_in);
                const uint64_t lm = sse_ballot(valid & !inb & all_in & (kind == SSE_BOND_LONGITUDINAL)) & seg;
                if (lm) {
                    const uint32_t q = (uint32_t)__ffsll((long long)lm) - 1u;
                    nb += (uint32_t)popc64(inbm & ahead & lanemask_lt((int)q));
                    mult *= 0.0;
                    if (mult < EPS) return true;
                }
            }
            nb += (uint32_t)popc64(inbm & seg);
            if (!ev) break;
            // ---- the op at lane e changes the cluster or the spins ----
            const uint32_t sae = (uint32_t)__builtin_amdgcn_readlane((int)sa, (int)e), sce = (uint32_t)__builtin_amdgcn_readlane((int)sc, (int)e);
            const uint32_t oute = (uint32_t)__builtin_amdgcn_readlane((int)out, (int)e), infoe = (uint32_t)__builtin_amdgcn_readlane((int)info, (int)e);
            const bool isbe = (sse_ballot(isb) >> e) & 1ull, offe = (sse_ballot(offdiag) >> e) & 1ull;
            const bool twoe = (infoe & SSE_GI_TWO) != 0u;
            const uint32_t kinde = (infoe >> SSE_GI_KIND_SHIFT) & 3u;
            uint32_t fa = sae != SSE_GI_NONE ? LDSW(R.o_sfl, sae) : 0u, fc = sce != SSE_GI_NONE ? LDSW(R.o_sfl, sce) : 0u;
            const bool all_in = (sae != SSE_GI_NONE) && (fa & 2u) && (!twoe || ((sce != SSE_GI_NONE) && (fc & 2u)));
            if (isbe) { fa ^= 2u; next_tog++; }
            if (offe) { fa = (fa & 3u) | ((oute & 1u) << 2); fc = (fc & 3u) | (((oute >> 1) & 1u) << 2); }
            if (sae != SSE_GI_NONE) LDSW(R.o_sfl, sae) = fa;
            if (offe && sce != SSE_GI_NONE) LDSW(R.o_sfl, sce) = fc;
            SSE_WAVE_FENCE();
            if (all_in) {
                if (kinde == SSE_BOND_LONGITUDINAL) mult *= 0.0;
                if (mult < EPS) return true;
            }
            if (!(nb == 0 || fabs(bs.tb - bs.ta) < EPS)) mult *= powi_sq(bs.ta / bs.tb, nb);
            nb = 0;
            if (mult < EPS) return true;
            bool ok = true;
            if (sae != SSE_GI_NONE) ok = rvb_update_bonds_w<CL, W>(B, L, R, LDSW(R.o_sub, sae), bs, false, lane);
            if (ok && twoe && sce != SSE_GI_NONE) ok = rvb_update_bonds_w<CL, W>(B, L, R, LDSW(R.o_sub, sce), bs, false, lane);
            if (!ok) { LDSW(R.o_ctl, RC_ERR) = 7u; return true; }
            SSE_WAVE_FENCE();
            pos = e + 1u;
            if (pos >= cnt) break; // (also keeps the lane masks away from a shift by 64)
        }
    }
    return false;
}

// get_propagated_substate_with_hint (fast_ops.rs:1027-1172): the spin of every sub-variable just before slot `from`
// = the output bit of its last op in [0, from), else the p=0 state.  Cooperative backward search.
template <int W, bool CL, int U = 4, bool BM = false>
__device__ __forceinline__ void rvb_state_at(const DevBatch &B, const Lds<W> &L, const RvbLds &R, uint32_t r, uint32_t from, uint32_t nsub,
                                             bool flip_by_cluster) {
    constexpr int NT = W * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    for (uint32_t s = tid; s < nsub; s += NT) LDSW(R.o_last, s) = 0u;
    if (tid == 0) LDSW(R.o_ctl, RC_SKIP) = 0u;
    __syncthreads();
    uint32_t hi = from; // search slots [lo, hi)
    while (hi > 0) {
        const uint32_t span = (uint32_t)(U * NT);
        const uint32_t lo = hi > span ? hi - span : 0u;
        uint32_t wx[U];
        load_rows<U>(ops_window(ops, hi), lo + (uint32_t)(wave * 64 * U + lane), wx);
        if constexpr (BM) rvb_last_ops_bm<W, CL, U>(B, L, R, wx, lo, true); // (the gathered-op lists are idle: the gathers of this window come after)
        else {
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (wx[j]) rvb_last_op<W, CL>(B, L, R, lo + (uint32_t)(wave * 64 * U + j * 64 + lane), wx[j]);
        }
        __syncthreads();
        // all found?
        uint32_t missing = 0;
        for (uint32_t s = tid; s < nsub; s += NT) missing |= (LDSW(R.o_last, s) == 0u);
        if (missing) LDSW(R.o_ctl, RC_SKIP) = 1u;
        __syncthreads();
        const uint32_t any = LDSW(R.o_ctl, RC_SKIP);
        __syncthreads();
        if (tid == 0) LDSW(R.o_ctl, RC_SKIP) = 0u;
        hi = lo;
        if (!any) break;
    }
    __syncthreads();
    for (uint32_t s = tid; s < nsub; s += NT) {
        const uint32_t x = LDSW(R.o_last, s);
        const uint32_t v = LDSW(R.o_sub, s);
        uint32_t bit = x ? (x & 1u) : ((LDSW(L.o_state, v >> 5) >> (v & 31)) & 1u);
        uint32_t f = LDSW(R.o_sfl, s);
        if (flip_by_cluster) bit ^= (f >> 1) & 1u;
        LDSW(R.o_sfl, s) = (f & 3u) | (bit << 2);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// One attempt's growth products and scratch (word offsets into lds_raw).  The growth of an attempt reads only the
// constant-op table, the adjacency and its own random numbers — never the op-string — so the attempts of a batch grow
// side by side, one per wave, each into its own small area; an attempt whose cluster outgrows a small area is grown
// again in the large one when its turn comes.
struct GrowArea {
    uint32_t o_bfk, o_bfv, o_bfw, o_bnk, o_bnw, o_clv, o_clf, o_sub, o_sfl, o_tmp, o_tog, o_togs, o_wfrom, o_wuntil, o_out;
    uint32_t cap_set, cap_cl, cap_sub, cap_win;
};
enum { GO_NSUB = 0, GO_NWIN = 1, GO_NTOG = 2, GO_K = 3, GO_ERR = 4 };
#define SSE_RVB_SLOT_SET 64u // (<= 64: the keys of a small area live in registers)
#define SSE_RVB_SLOT_CL 16u
#define SSE_RVB_SLOT_SUB (SSE_RVB_SLOT_CL + 2u * SSE_RVB_SLOT_SET)
#define SSE_RVB_SLOT_WIN (SSE_RVB_SLOT_CL + 2u)
#define SSE_RVB_SLOT_WORDS (4u * SSE_RVB_SLOT_SET + 4u * SSE_RVB_SLOT_CL + 2u * SSE_RVB_SLOT_SUB + 2u * SSE_RVB_SLOT_WIN + 8u)
__device__ __forceinline__ GrowArea grow_area_small(uint32_t base) { // base even (doubles)
    GrowArea A;
    A.cap_set = SSE_RVB_SLOT_SET; A.cap_cl = SSE_RVB_SLOT_CL; A.cap_sub = SSE_RVB_SLOT_SUB; A.cap_win = SSE_RVB_SLOT_WIN;
    A.o_bfw = base; base += 2 * A.cap_set;
    A.o_bnw = base; base += 2 * A.cap_set;
    A.o_bfk = A.o_bfv = A.o_bnk = A.o_clv = A.o_clf = 0u; // in registers (rvb_grow<.., REG = true>)
    A.o_tog = base; base += 2 * A.cap_cl;
    A.o_togs = base; base += 2 * A.cap_cl;
    A.o_sub = base; base += A.cap_sub;
    A.o_sfl = base; base += A.cap_sub;
    static_assert(4u * SSE_RVB_SLOT_SET >= SSE_RVB_SLOT_SUB, "the sort scratch of a small area lives in its weight arrays");
    A.o_tmp = A.o_bfw; // (the candidate weights are dead once the cluster has its members: their room serves the sort)
    A.o_wfrom = base; base += A.cap_win;
    A.o_wuntil = base; base += A.cap_win;
    A.o_out = base;
    return A;
}
#define SSE_RVB_LARGE_WORDS (7u * SSE_RVB_SETCAP + 6u * SSE_RVB_MAXCL + 3u * SSE_RVB_MAXSUB + 2u * SSE_RVB_MAXWIN + 8u)
__device__ __forceinline__ GrowArea grow_area_large_at(uint32_t base) { // base even (doubles first); SSE_RVB_LARGE_WORDS words
    GrowArea A;
    A.cap_set = SSE_RVB_SETCAP; A.cap_cl = SSE_RVB_MAXCL; A.cap_sub = SSE_RVB_MAXSUB; A.cap_win = SSE_RVB_MAXWIN;
    A.o_bfw = base; A.o_bnw = base + 2 * SSE_RVB_SETCAP; A.o_bfk = base + 4 * SSE_RVB_SETCAP; A.o_bfv = base + 5 * SSE_RVB_SETCAP; A.o_bnk = base + 6 * SSE_RVB_SETCAP;
    base += 7 * SSE_RVB_SETCAP;
    A.o_clv = base; base += SSE_RVB_MAXCL;
    A.o_clf = base; base += SSE_RVB_MAXCL;
    A.o_tog = base; base += 2 * SSE_RVB_MAXCL;
    A.o_togs = base; base += 2 * SSE_RVB_MAXCL;
    A.o_sub = base; base += SSE_RVB_MAXSUB;
    A.o_sfl = base; base += SSE_RVB_MAXSUB;
    A.o_tmp = base; base += SSE_RVB_MAXSUB;
    A.o_wfrom = base; base += SSE_RVB_MAXWIN;
    A.o_wuntil = base; base += SSE_RVB_MAXWIN;
    A.o_out = base;
    return A;
}
__device__ __forceinline__ GrowArea grow_area_large(const RvbLds &R) {
    GrowArea A;
    A.cap_set = SSE_RVB_SETCAP; A.cap_cl = SSE_RVB_MAXCL; A.cap_sub = SSE_RVB_MAXSUB; A.cap_win = SSE_RVB_MAXWIN;
    A.o_bfw = R.o_bfw; A.o_bnw = R.o_bnw; A.o_bfk = R.o_bfk; A.o_bfv = R.o_bfv; A.o_bnk = R.o_bnk;
    A.o_clv = R.o_clv; A.o_clf = R.o_clf; A.o_tog = R.o_tog; A.o_togs = R.o_togs;
    A.o_sub = R.o_sub; A.o_sfl = R.o_sfl; A.o_tmp = R.o_last; A.o_wfrom = R.o_wfrom; A.o_wuntil = R.o_wuntil;
    A.o_out = R.o_gout;
    return A;
}

// start, cluster growth, sub-variables, windows of one attempt (rvb.rs:88-232, :1054-1123); run by a whole wave, uniform
template <int W, bool CL, bool REG>
__device__ __forceinline__ void rvb_grow(const DevBatch &B, const Lds<W> &L, const RvbLds &R, const GrowArea &A, RvbDraw g0, uint32_t C, uint32_t nzero,
                                         uint32_t M, int lane) {
    const uint32_t N = B.N;
    uint32_t lerr = 0;
    RvbDrawW g;
    g.init(g0);
    uint4 o = g.next(lane);
    const uint32_t choice = __umulhi(o.x, C + nzero);
    uint32_t v0, f0;
    if (choice < C) { // the variable whose range of the position table holds `choice`: 64-way search
        uint32_t lo = 0, hi = N; // vstart[lo] <= choice < vstart[hi]
        while (hi - lo > 1u) {
            const uint32_t step = (hi - lo + 63u) / 64u, idx = lo + (uint32_t)lane * step;
            const uint32_t x = LDSW(R.o_vstart, idx < hi ? idx : lo);
            const uint32_t cnt = (uint32_t)popc64(sse_ballot((idx < hi) & (x <= choice))); // a prefix of the lanes (lane 0 always)
            const uint32_t nhi = lo + cnt * step;
            lo += (cnt - 1u) * step;
            hi = nhi < hi ? nhi : hi;
        }
        v0 = lo; f0 = choice;
    } else { v0 = LDSW(R.o_zero, choice - C); f0 = SSE_NO_VAR; }
    o = g.next(lane);
    unsigned long long bits = (unsigned long long)o.x | ((unsigned long long)o.y << 32);
    uint32_t csize = 1;
    while ((bits & 1ull) && csize <= 64) { csize++; bits >>= 1; }
    typename GrowTypes<REG>::Set bf, bn; // REG: keys in registers (small areas, <= 64 entries), else in LDS
    typename GrowTypes<REG>::Cluster cl;
    bf.init(A.o_bfk, A.o_bfv, A.o_bfw, A.cap_set);
    bn.init(A.o_bnk, 0xFFFFFFFFu, A.o_bnw, A.cap_set);
    cl.init(A.o_clv, A.o_clf, A.cap_cl);
    auto push_adj = [&](uint32_t var, uint32_t pos, double w) {
        if (cl.contains(var, pos, lane)) return;
        const bool ok = pos != SSE_NO_VAR ? bf.add(pos, var, w, lane) : bn.add(var, var, w, lane);
        if (!ok) lerr = 7u;
    };
    push_adj(v0, f0, 1.0);
    uint32_t left = csize;
    while (left > 0 && (bf.n + bn.n) > 0 && !lerr) {
        o = g.next(lane);
        const double f_ratio = bf.total / (bf.total + bn.total);
        bool pick_flips = u01(o.x) < f_ratio;
        if (bf.n == 0) pick_flips = false;
        if (bn.n == 0) pick_flips = true;
        o = g.next(lane);
        uint32_t v, flip;
        if (pick_flips) { const uint32_t idx = bf.pick(u01(o.x), lane); v = bf.var_at(idx); flip = bf.key_at(idx); bf.remove_at(idx); }
        else { const uint32_t idx = bn.pick(u01(o.x), lane); v = bn.key_at(idx); flip = SSE_NO_VAR; bn.remove_at(idx); }
        if (cl.n >= cl.cap) { lerr = 7u; break; }
        cl.push(v, flip);
        const uint32_t vs = LDSW(R.o_vstart, v), vl = LDSW(R.o_vstart, v + 1) - vs;
        if (flip != SSE_NO_VAR) {
            const uint32_t rel = flip - vs;
            push_adj(v, wrap(rel + vl - 1, vl) + vs, 1.0);
            push_adj(v, wrap(rel + 1, vl) + vs, 1.0);
        }
        // the neighbours of v: lane k fetches bond k of its adjacency list and the other end's range of the position table
        // (one round of loads for all of them), then they are taken in list order
        const uint32_t i0 = adj_begin(R, B, v), i1 = adj_begin(R, B, v + 1);
        for (uint32_t nbase = i0; nbase < i1; nbase += 64u) {
            const uint32_t ni = nbase + (uint32_t)lane;
            const bool nin = ni < i1;
            const Bd dl = decode_bond<CL, W>(B, L, adj_at(R, B, nin ? ni : i0));
            const uint32_t ovl = dl.a == v ? dl.c : dl.a;
            const uint32_t osl = LDSW(R.o_vstart, ovl), oll = LDSW(R.o_vstart, ovl + 1) - osl;
            const double wl = dl.w * 0.5; // bond_mag = |J| (qmc_ising.rs:633-635)
            const uint32_t ncnt = i1 - nbase < 64u ? i1 - nbase : 64u;
            for (uint32_t k = 0; k < ncnt; ++k) {
                const uint32_t ov = (uint32_t)__builtin_amdgcn_readlane((int)ovl, (int)k), os = (uint32_t)__builtin_amdgcn_readlane((int)osl, (int)k);
                const uint32_t ol = (uint32_t)__builtin_amdgcn_readlane((int)oll, (int)k);
                const double weight = readlane_f64(wl, k);
                if (ol == 0) push_adj(ov, SSE_NO_VAR, weight);
                else if (flip != SSE_NO_VAR) {
                    const uint32_t rel = flip - vs;
                    const uint32_t finc = wrap(rel + 1, vl) + vs;
                    rvb_overlaps_w(R, LDSW(R.o_cps, flip), LDSW(R.o_cps, finc), M, os, ol, lane, [&](uint32_t ip) { push_adj(ov, ip + os, weight); });
                } else {
                    for (uint32_t pi = os; pi < os + ol; ++pi) push_adj(ov, pi, weight);
                }
            }
        }
        left--;
    }
    // ---- sub-variables: sorted union of cluster and remaining boundary variables (:155-172) ----
    // candidates -> o_last; an entry that repeats an earlier one is marked, the others rank themselves among the
    // unmarked ones (lane per candidate, the comparison partner is broadcast from LDS)
    const uint32_t ncl = cl.n;
    const uint32_t nc = lerr ? 0u : ncl + bf.n + bn.n;
    if (nc > A.cap_sub) lerr = 7u;
    if (!lerr) { cl.dump_vars(A.o_tmp, 0u, lane); bf.dump_vars(A.o_tmp, ncl, lane); bn.dump_vars(A.o_tmp, ncl + bf.n, lane); }
    SSE_WAVE_FENCE();
    uint32_t nsub = 0;
    if (!lerr) {
        const uint32_t DUP = 0x80000000u;
        for (uint32_t base = 0; base < nc; base += 64u) { // mark repeats (variables are < 2^31)
            const uint32_t i = base + (uint32_t)lane;
            const uint32_t vi = LDSW(A.o_tmp, i < nc ? i : 0u) & ~DUP;
            bool dup = false;
            for (uint32_t j = 0; j < base + 64u && j < nc; ++j) dup |= (j < i) & ((LDSW(A.o_tmp, j) & ~DUP) == vi);
            SSE_WAVE_FENCE();
            if ((i < nc) & dup) LDSW(A.o_tmp, i) = vi | DUP;
            SSE_WAVE_FENCE();
        }
        for (uint32_t base = 0; base < nc; base += 64u) {
            const uint32_t i = base + (uint32_t)lane;
            const uint32_t xi = LDSW(A.o_tmp, i < nc ? i : 0u);
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nc; ++j) rank += LDSW(A.o_tmp, j) < xi ? 1u : 0u; // marked entries compare high: never counted
            const bool keep = (i < nc) & !(xi & DUP);
            if (keep) { LDSW(A.o_sub, rank) = xi; LDSW(A.o_sfl, rank) = 0u; }
            nsub += (uint32_t)popc64(sse_ballot(keep));
        }
        SSE_WAVE_FENCE();
    }
    // ---- starting state and toggle positions (:174-196), sort, remove_doubles (:230-231) ----
    uint32_t ntog = 0;
    for (uint32_t i = 0; i < ncl && !lerr; ++i) {
        const uint32_t v = cl.v_at(i), fi = cl.f_at(i);
        const uint32_t sv = (uint32_t)wv_find(A.o_sub, nsub, v, lane); // the list is this attempt's own: the shared var -> sub table is filled when its turn comes
        if (fi != SSE_NO_VAR) {
            const uint32_t vs = LDSW(R.o_vstart, v), vl = LDSW(R.o_vstart, v + 1) - vs;
            uint32_t t0 = LDSW(R.o_cps, fi), t1;
            if (fi - vs + 1 >= vl) { LDSW(A.o_sfl, sv) |= 1u; t1 = LDSW(R.o_cps, vs); }
            else t1 = LDSW(R.o_cps, fi + 1);
            for (int q = 0; q < 2; ++q) { // sorted insert: the entries above x move up by one (lane per entry)
                const uint32_t x = q ? t1 : t0;
                uint32_t pos = 0;
                for (uint32_t base = 0; base < ntog; base += 64u) {
                    const uint32_t j = base + (uint32_t)lane;
                    pos += (uint32_t)popc64(sse_ballot((j < ntog) & (LDSW(A.o_tog, j < ntog ? j : 0u) <= x)));
                }
                for (uint32_t top = ntog; top > pos; ) { // highest block first: every entry is read before its slot is overwritten
                    const uint32_t lo = top - pos > 64u ? top - 64u : pos;
                    const uint32_t j = lo + (uint32_t)lane;
                    const uint32_t tp = LDSW(A.o_tog, j < top ? j : lo), ts = LDSW(A.o_togs, j < top ? j : lo);
                    SSE_WAVE_FENCE();
                    if (j < top) { LDSW(A.o_tog, j + 1) = tp; LDSW(A.o_togs, j + 1) = ts; }
                    SSE_WAVE_FENCE();
                    top = lo;
                }
                LDSW(A.o_tog, pos) = x; LDSW(A.o_togs, pos) = sv;
                SSE_WAVE_FENCE();
                ntog++;
            }
        } else LDSW(A.o_sfl, sv) |= 1u;
    }
    { // remove_doubles (util/vec_help.rs:4-24)
        uint32_t ii = 0, jj = 0;
        while (jj + 1 < ntog) {
            if (LDSW(A.o_tog, jj) == LDSW(A.o_tog, jj + 1)) jj += 2;
            else { LDSW(A.o_tog, ii) = LDSW(A.o_tog, jj); LDSW(A.o_togs, ii) = LDSW(A.o_togs, jj); ii++; jj++; }
        }
        if (jj < ntog) { LDSW(A.o_tog, ii) = LDSW(A.o_tog, jj); LDSW(A.o_togs, ii) = LDSW(A.o_togs, jj); ii++; jj++; }
        ntog = ii;
    }
    // ---- windows where the cluster is non-empty (mutate_graph :310-360); cluster_state := starting state ----
    uint32_t count = 0, nwin = 0, nuntil = 0;
    for (uint32_t base = 0; base < nsub; base += 64u) {
        const uint32_t sidx = base + (uint32_t)lane;
        const uint32_t f = sidx < nsub ? LDSW(A.o_sfl, sidx) & 1u : 0u;
        if (sidx < nsub) LDSW(A.o_sfl, sidx) = f | (f << 1);
        count += (uint32_t)popc64(sse_ballot(f != 0u));
    }
    SSE_WAVE_FENCE();
    if (count) LDSW(A.o_wfrom, nwin++) = 0u;
    for (uint32_t i = 0; i < ntog && !lerr; ++i) {
        const uint32_t p = LDSW(A.o_tog, i);
        if (count == 0) { if (nwin >= A.cap_win) { lerr = 7u; break; } LDSW(A.o_wfrom, nwin++) = p; }
        const uint32_t sv = LDSW(A.o_togs, i); // the toggle is a constant op of this cluster member
        const uint32_t f = LDSW(A.o_sfl, sv) ^ 2u;
        LDSW(A.o_sfl, sv) = f;
        if (f & 2u) count++; else count--;
        if (count == 0) LDSW(A.o_wuntil, nuntil++) = p;
    }
    if (count) LDSW(A.o_wuntil, nuntil++) = M;
    SSE_WAVE_FENCE();
    // restore cluster_state = starting state for the probability pass
    for (uint32_t base = 0; base < nsub; base += 64u) {
        const uint32_t sidx = base + (uint32_t)lane;
        if (sidx < nsub) { const uint32_t f = LDSW(A.o_sfl, sidx) & 1u; LDSW(A.o_sfl, sidx) = f | (f << 1); }
    }
    LDSW(A.o_out, GO_NSUB) = nsub;
    LDSW(A.o_out, GO_NWIN) = nwin;
    LDSW(A.o_out, GO_NTOG) = ntog;
    LDSW(A.o_out, GO_K) = g.g.k;
    LDSW(A.o_out, GO_ERR) = lerr;
}

// Phases B-D of one attempt (calculate_flip_prob, accept, mutate_graph) whose growth products — sub-variables with their starting
// flags, toggles, windows — are the lists R points at (R0: the replica's own scratch).  Returns true when the sweep has to stop.
// BIG: scan steps of >= 4096 slots whatever the wave count (the main launch of the two-launch form, whose lists hold 64 * UG ops)
template <int W, bool CL, bool BIG = false>
__device__ __forceinline__ bool rvb_attempt(const DevBatch &B, const Lds<W> &L, const RvbLds &R0, const RvbLds &R, uint32_t r, RvbDraw g,
                                            uint32_t nsub, uint32_t nwin, uint32_t ntog, uint32_t M, uint32_t &gr, uint32_t &nsucc) {
    constexpr int UG = BIG ? (W <= 4 ? SSE_RVB_UG4 : 8) : 8, UL = BIG ? (W <= 4 ? SSE_RVB_UL4 : (W <= 8 ? 8 : 4)) : 4; // (the lists of the main launch hold 64 * UG ops)
    constexpr bool BM = BIG; // ... and its records carry the bond map
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    SSE_STAMP_INIT;
    for (uint32_t s = tid; s < nsub; s += blockDim.x) { LDSH(R0.o_v2s, LDSW(R.o_sub, s)) = (uint16_t)s; LDSW(R0.o_last, s) = 0u; }
    __syncthreads();

    // ================= phase B: calculate_flip_prob (rvb.rs:649-946) over the windows =================
    double mult = 1.0;             // wave 0
    uint32_t nb = 0, next_tog = 0; // wave 0
    BSetW bw;
    bw.o_key = R.o_bk; bw.o_wb = R.o_bwb; bw.o_wa = R.o_bwa; bw.o_ix = R.o_bix; bw.n = 0; bw.tb = 0.0; bw.ta = 0.0;
    for (uint32_t wi = 0; wi < nwin; ++wi) {
        const uint32_t from = LDSW(R.o_wfrom, wi), until = LDSW(R.o_wuntil, wi);
        rvb_fetch<W, CL, UL, UG, BM>(B, L, R, r, from, until, M, nsub, gr);
        SSE_STAMP(8);
        if (wave == 0 && wi == 0 && from == 0) { // set_initial_bonds (rvb.rs:617-645): the cluster's bonds to the outside, in order
            bool ok = true;
            for (uint32_t base = 0; base < nsub && ok; base += 64u) {
                const uint32_t s2 = base + (uint32_t)lane;
                uint64_t m = sse_ballot(s2 < nsub && (LDSW(R.o_sfl, s2 < nsub ? s2 : 0u) & 2u));
                while (m && ok) {
                    const uint32_t k = (uint32_t)__ffsll((long long)m) - 1u;
                    m &= m - 1;
                    ok = rvb_update_bonds_w<CL, W>(B, L, R, LDSW(R.o_sub, base + k), bw, true, lane);
                }
            }
            if (!ok) LDSW(R.o_ctl, RC_ERR) = 7u;
        }
        bool done = false;
        for (;;) {
            const uint32_t glen = LDSW(R.o_ctl, RC_GLEN), gp = LDSW(R.o_ctl, RC_NEXTP);
            const bool more = !(M == 0u || gp > (until < M ? until : M - 1));
            uint32_t wrow[UG];
            if (BIG && more) rvb_gather_rows<W, UG>(B, r, gp, until, M, wrow); // on their way while wave 0 replays this batch
            if (wave == 0 && !LDSW(R.o_ctl, RC_ERR)) {
                if (rvb_replay_prob<W, CL>(B, L, R, glen, ntog, next_tog, nb, mult, bw, lane)) LDSW(R.o_ctl, RC_BROKE) = 1u;
            }
            __syncthreads(); // the lists are free again; everybody learns whether the product is already zero
            SSE_STAMP(10);
            if (LDSW(R.o_ctl, RC_BROKE) || LDSW(R.o_ctl, RC_ERR)) { done = true; break; }
            if (!more) break;
            if (!BIG) rvb_gather_rows<W, UG>(B, r, gp, until, M, wrow);
            rvb_gather<W, CL, UG, BM>(B, L, R, r, gp, until, M, gr, wrow);
            SSE_STAMP(9);
        }
        if (done) break;
    }
    // ================= phase C: accept (:241-246) =================
    if (wave == 0) {
        if (!(nb == 0 || fabs(bw.tb - bw.ta) < 2.220446049250313e-16)) mult *= powi_sq(bw.ta / bw.tb, nb);
        const uint4 o = g.next();
        const bool accept = (mult >= 1.0) || (u01(o.x) < mult);
        bw.clear(lane);
        LDSW(R.o_ctl, RC_ACCEPT) = accept ? 1u : 0u;
        LDSW(R.o_ctl, 9) = g.k;
        LDSW(R.o_ctl, RC_BROKE) = 0u;
    }
    __syncthreads();
    SSE_STAMP(11);
    if (LDSW(R.o_ctl, RC_ERR)) return true;
    g.k = LDSW(R.o_ctl, 9);
    if (LDSW(R.o_ctl, RC_ACCEPT)) {
        // ================= phase D: mutate_graph (:294-615) =================
        // cluster_state := starting state
        for (uint32_t s = tid; s < nsub; s += blockDim.x) { const uint32_t f = LDSW(R.o_sfl, s) & 1u; LDSW(R.o_sfl, s) = f | (f << 1); }
        __syncthreads();
        BSet bs;
        bs.o_key = R.o_bk; bs.o_wb = R.o_bwb; bs.o_wa = R.o_bwa; bs.n = 0; bs.tb = 0.0; bs.ta = 0.0;
        uint32_t nt2 = 0;
        for (uint32_t wi = 0; wi < nwin; ++wi) {
            const uint32_t from = LDSW(R.o_wfrom, wi), until = LDSW(R.o_wuntil, wi);
            rvb_state_at<W, CL, UL, BM>(B, L, R, r, from, nsub, true); // substate ^= cluster_state (:396-399, :315-318)
            SSE_STAMP(8);
            if (tid == 0 && wi == 0 && from == 0) {
                if (!rvb_initial_bonds<CL, W>(B, L, R, nsub, bs, false)) LDSW(R.o_ctl, RC_ERR) = 7u;
            }
            uint32_t gp = from;
            for (;;) {
                SSE_STAMP(12);
                uint32_t wrow[UG];
                rvb_gather_rows<W, UG>(B, r, gp, until, M, wrow);
                rvb_gather<W, CL, UG, BM>(B, L, R, r, gp, until, M, gr, wrow);
                SSE_STAMP(9);
                const uint32_t glen = LDSW(R.o_ctl, RC_GLEN);
                gp = LDSW(R.o_ctl, RC_NEXTP);
                if (tid == 0 && !LDSW(R.o_ctl, RC_ERR)) {
                    for (uint32_t i = 0; i < glen; ++i) {
                        const uint32_t p = LDSW(R.o_glp, i), wd = LDSW(R.o_glw, i);
                        const uint32_t b = sse_op_bond(wd), in = sse_op_in(wd), out = sse_op_out(wd);
                        const Bd d = decode_bond<CL, W>(B, L, b);
                        const uint32_t sa = v2s_get(R, d.a), sc = d.c != SSE_NO_VAR ? v2s_get(R, d.c) : 0xFFFFu;
                        const bool at_flip = nt2 < ntog && p == LDSW(R.o_tog, nt2);
                        if (b < B.E && bs.find(b) >= 0) {
                            // rotate the boundary op onto a boundary bond drawn by weight (:414-432)
                            const uint4 o = g.next();
                            const uint32_t nbnd = LDSW(bs.o_key, bs.pick_before(u01(o.x)));
                            const Bd nd = decode_bond<CL, W>(B, L, nbnd);
                            const uint32_t s2 = ((LDSW(R.o_sfl, v2s_get(R, nd.a)) >> 2) & 1u) | (((LDSW(R.o_sfl, v2s_get(R, nd.c)) >> 2) & 1u) << 1);
                            ops[p] = sse_op_make(nbnd, s2, s2);
                            continue;
                        }
                        if (at_flip) {
                            const uint32_t cs = (LDSW(R.o_sfl, sa) >> 1) & 1u;
                            const uint32_t nin = (in & 1u) ^ cs, nout = (out & 1u) ^ (cs ^ 1u);
                            ops[p] = sse_op_make(b, nin, nout);
                            LDSW(R.o_sfl, sa) = ((LDSW(R.o_sfl, sa) ^ 2u) & 3u) | (nout << 2);
                            nt2++;
                        } else {
                            const bool a_in = sa != 0xFFFFu && (LDSW(R.o_sfl, sa) & 2u);
                            const bool c_in = sc != 0xFFFFu && (LDSW(R.o_sfl, sc) & 2u);
                            if (a_in || c_in) {
                                const uint32_t mask = d.c != SSE_NO_VAR ? 3u : 1u;
                                const uint32_t nin = in ^ mask, nout = out ^ mask;
                                ops[p] = sse_op_make(b, nin, nout);
                                if (nin != nout) {
                                    if (sa != 0xFFFFu) LDSW(R.o_sfl, sa) = (LDSW(R.o_sfl, sa) & 3u) | ((nout & 1u) << 2);
                                    if (sc != 0xFFFFu) LDSW(R.o_sfl, sc) = (LDSW(R.o_sfl, sc) & 3u) | (((nout >> 1) & 1u) << 2);
                                }
                            } else if (in != out) {
                                if (sa != 0xFFFFu) LDSW(R.o_sfl, sa) = (LDSW(R.o_sfl, sa) & 3u) | ((out & 1u) << 2);
                                if (sc != 0xFFFFu) LDSW(R.o_sfl, sc) = (LDSW(R.o_sfl, sc) & 3u) | (((out >> 1) & 1u) << 2);
                            } else {
                                continue; // diagonal and untouched by the cluster (:513-514)
                            }
                        }
                        bool ok = rvb_update_bonds<CL, W>(B, L, R, d.a, bs, false);
                        if (ok && d.c != SSE_NO_VAR) ok = rvb_update_bonds<CL, W>(B, L, R, d.c, bs, false);
                        if (!ok) { LDSW(R.o_ctl, RC_ERR) = 7u; break; }
                    }
                }
                if (M == 0u || gp > (until < M ? until : M - 1)) break;
            }
        }
        __syncthreads();
        SSE_STAMP(12);
        // p=0 state of the sub-variables that start inside the cluster (:266-274)
        for (uint32_t s = tid; s < nsub; s += blockDim.x)
            if (LDSW(R.o_sfl, s) & 1u) { const uint32_t v = LDSW(R.o_sub, s); atomicXor(&LDSW(L.o_state, v >> 5), 1u << (v & 31)); }
        nsucc++;
    }
    __syncthreads();
    for (uint32_t s = tid; s < nsub; s += blockDim.x) LDSH(R.o_v2s, LDSW(R.o_sub, s)) = (uint16_t)0xFFFFu;
    __syncthreads();
    if (LDSW(R.o_ctl, RC_ERR)) return true;
    return false;
}

// ---------------------------------------------------------------------------------------------
template <int W, bool CL>
__device__ __forceinline__ uint32_t rvb_pass(const DevBatch &B, const Lds<W> &L, uint32_t r, uint64_t epoch, uint32_t M, uint32_t updates,
                                             uint32_t &gr, uint32_t &err) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    RvbLds R0;
    rvb_carve<W>(R0, L, B);
    if (tid == 0) { LDSW(R0.o_ctl, RC_ERR) = 0u; LDSW(R0.o_ctl, RC_SKIP) = 0u; LDSW(R0.o_ctl, RC_BROKE) = 0u; }
    for (uint32_t i = tid; i < B.E; i += blockDim.x) LDSH(R0.o_bix, i) = (uint16_t)0xFFFFu;
    __syncthreads();
    SSE_STAMP_INIT; // diagnostic builds: 6 constants table, 7 growth, 8 states, 9 gathers, 10 replay (probability), 11 accept, 12 replay (mutation)
    const uint32_t C = rvb_find_constants<W, CL>(B, L, R0, r, M);
    SSE_STAMP(6);
    if (C == 0xFFFFFFFFu) { err = 6u; return 0u; } // constant-op table does not fit in LDS
    const uint32_t nzero = LDSW(R0.o_ctl, RC_NZERO);
    uint32_t nsucc = 0;
    // small growth areas behind the used part of the constant-op table: as many as fit, at most one per wave
    const uint32_t slots0 = (R0.o_cps + C + 1u) & ~1u;
    uint32_t P = B.lds_words > slots0 ? (B.lds_words - slots0) / SSE_RVB_SLOT_WORDS : 0u;
    if (P > (uint32_t)W) P = (uint32_t)W;
    if (P > B.rvb_growers) P = B.rvb_growers;
    const uint32_t PB = P ? P : 1u; // attempts per batch
#ifdef SSE_PHASE_TIMING
    if (tid == 0) { B.dbg[(size_t)r * 16 + 15] = P; B.dbg[(size_t)r * 16 + 14] = B.lds_words; B.dbg[(size_t)r * 16 + 13] = C; B.dbg[(size_t)r * 16 + 5] = R0.o_cps; } // attempts grown side by side
#endif
    const GrowArea big = grow_area_large(R0);
    bool stop = false;

    for (uint32_t a0 = 0; a0 < updates && !stop; a0 += PB) {
    if (P && (uint32_t)wave < P && a0 + (uint32_t)wave < updates) {
        RvbDraw g;
        g.k0 = B.seed_lo; g.k1 = B.seed_hi; g.replica = B.rid ? B.rid[r] : B.replica_offset + r; g.epoch_lo = (uint32_t)epoch; g.attempt = a0 + (uint32_t)wave; g.k = 0;
        rvb_grow<W, CL, true>(B, L, R0, grow_area_small(slots0 + (uint32_t)wave * SSE_RVB_SLOT_WORDS), g, C, nzero, M, lane);
    }
    __syncthreads();
    SSE_STAMP(7);
    for (uint32_t aj = 0; aj < PB && a0 + aj < updates; ++aj) {
        const uint32_t attempt = a0 + aj;
        RvbDraw g;
        g.k0 = B.seed_lo; g.k1 = B.seed_hi; g.replica = B.rid ? B.rid[r] : B.replica_offset + r; g.epoch_lo = (uint32_t)epoch; g.attempt = attempt; g.k = 0;
        GrowArea A = P ? grow_area_small(slots0 + aj * SSE_RVB_SLOT_WORDS) : big;
        if (!P || LDSW(A.o_out, GO_ERR)) { // no room for small areas, or this cluster outgrew its own: the large area
            A = big;
            if (wave == 0) rvb_grow<W, CL, false>(B, L, R0, big, g, C, nzero, M, lane);
            __syncthreads();
            SSE_STAMP(7);
        }
        if (LDSW(A.o_out, GO_ERR)) { if (tid == 0) LDSW(R0.o_ctl, RC_ERR) = LDSW(A.o_out, GO_ERR); __syncthreads(); stop = true; break; }
        const uint32_t nsub = LDSW(A.o_out, GO_NSUB), nwin = LDSW(A.o_out, GO_NWIN), ntog = LDSW(A.o_out, GO_NTOG);
        g.k = LDSW(A.o_out, GO_K);
        RvbLds R = R0; // this attempt's lists
        R.o_sub = A.o_sub; R.o_sfl = A.o_sfl; R.o_tog = A.o_tog; R.o_togs = A.o_togs; R.o_wfrom = A.o_wfrom; R.o_wuntil = A.o_wuntil;
        if (rvb_attempt<W, CL>(B, L, R0, R, r, g, nsub, nwin, ntog, M, gr, nsucc)) { stop = true; break; }
    } // attempts of the batch, in order
    } // batches
    __syncthreads();
    if (LDSW(R0.o_ctl, RC_ERR)) err = LDSW(R0.o_ctl, RC_ERR);
    __syncthreads();
    return nsucc;
}

} // namespace sse
