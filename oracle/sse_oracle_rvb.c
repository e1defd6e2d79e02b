/*
 * sse_oracle_rvb.c — CPU ORACLE, resonating-valence-bond update (test infrastructure, NOT product).
 *
 * Sequential restatement of RvbUpdater::rvb_update_with_ising_weight (src/sse/qmc_traits/rvb.rs:88-290) with
 * build_cluster (:1054-1123), find_overlapping_starts (:1125-1158), find_constants (:1160-1187),
 * calculate_flip_prob (:649-946), calculate_mult (:1194-1220), mutate_graph (:294-615), the weighted boundary
 * manager (:967-1052) and BondContainer (src/util/bondcontainer.rs).  The reference walks per-variable linked
 * lists with a heap; here every "ops touching the sub-variables in p order" walk is a plain scan of the
 * op-string, which visits the same ops in the same order.
 *
 * Random numbers (Philox, tag RVB): counter = (k, epoch_lo, replica, tag<<24 | attempt) where k counts the
 * draws of one attempt: 0 start choice, 1 cluster size (64 bits = words 0,1), then two per cluster-growth pop
 * (set choice, weighted pick), one for the acceptance, one per rotated boundary op.
 */
#include "sse_oracle_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t key, v, p; double w; } wentry; /* p = flip index or SSE_NO_VAR */
typedef struct { wentry *e; uint32_t n, cap; int32_t *map; uint32_t mapn; double total; } wset;

static void wset_init(wset *s, uint32_t nkeys) {
    s->cap = 64; s->n = 0; s->total = 0.0;
    s->e = (wentry *)malloc(sizeof(wentry) * s->cap);
    s->mapn = nkeys;
    s->map = (int32_t *)malloc(sizeof(int32_t) * (nkeys ? nkeys : 1));
    for (uint32_t i = 0; i < nkeys; ++i) s->map[i] = -1;
}
static void wset_free(wset *s) { free(s->e); free(s->map); }
static int wset_contains(const wset *s, uint32_t key) { return key < s->mapn && s->map[key] >= 0; }
static double wset_weight(const wset *s, uint32_t key) { return wset_contains(s, key) ? s->e[s->map[key]].w : 0.0; }
/* bondcontainer.rs insert: update the weight in place if present, else append */
static void wset_insert(wset *s, uint32_t key, uint32_t v, uint32_t p, double w) {
    if (wset_contains(s, key)) {
        wentry *x = &s->e[s->map[key]];
        s->total += w - x->w;
        x->w = w;
        return;
    }
    if (s->n == s->cap) { s->cap *= 2; s->e = (wentry *)realloc(s->e, sizeof(wentry) * s->cap); }
    s->e[s->n] = (wentry){key, v, p, w};
    s->map[key] = (int32_t)s->n++;
    s->total += w;
}
/* bondcontainer.rs remove_index: swap with the last entry, pop, clamp the running total at 0 */
static void wset_remove(wset *s, uint32_t key) {
    if (!wset_contains(s, key)) return;
    uint32_t i = (uint32_t)s->map[key], last = s->n - 1;
    double w = s->e[i].w;
    s->e[i] = s->e[last];
    s->map[s->e[i].key] = (int32_t)i;
    s->map[key] = -1;
    s->n--;
    s->total -= w;
    if (s->total < 0.0) s->total = 0.0;
}
/* bondcontainer.rs get_random: p = u*total; walk the entries subtracting weights until p <= 0 */
static uint32_t wset_pick(const wset *s, double u) {
    double p = u * s->total;
    uint32_t i = 0;
    while (i < s->n) {
        p -= s->e[i].w;
        if (p <= 0.0) break;
        i++;
    }
    return i < s->n ? i : s->n - 1;
}

typedef struct {
    ora_replica *r;
    uint32_t attempt, k; /* draw counter inside the attempt */
} rvb_rng;
static void rvb_draw(rvb_rng *g, uint32_t o[4]) {
    const ora_replica *r = g->r;
    uint32_t ctr[4] = {g->k++, (uint32_t)r->epoch, r->replica, (SSE_TAG_RVB << 24) | (g->attempt & 0xFFFFFFu)};
    uint32_t key[2] = {(uint32_t)r->seed, (uint32_t)(r->seed >> 32)};
    ora_philox4x32_10(ctr, key, o);
}

/* x^n by squaring (the kernels use the same sequence of multiplications) */
static double powi_sq(double x, uint32_t n) {
    double r = 1.0;
    while (n) { if (n & 1u) r *= x; x *= x; n >>= 1; }
    return r;
}
/* rvb.rs:1194-1220 */
static double calc_mult(const wset *before, const wset *after, uint32_t n) {
    if (n == 0 || fabs(before->total - after->total) < DBL_EPSILON) return 1.0;
    return powi_sq(after->total / before->total, n);
}

/* two-site diagonal weight of edge b for spins (sa, sb): qmc_ising.rs:382-385 -> :863-875 */
static double edge_w(const ora_model *m, uint32_t b, uint32_t sa, uint32_t sb) {
    uint32_t s = (sa & 1u) | ((sb & 1u) << 1);
    return ora_bond_weight(m, b, s, s);
}

typedef struct {
    const ora_model *m;
    uint32_t nsub;
    const uint32_t *subvars;   /* sorted */
    const int32_t *var2sub;    /* [N] or -1 */
    const uint32_t *adj_start, *adj; /* bonds_for_var */
} subctx;

/* the "Now update bonds" block shared by calculate_flip_prob (:901-934) and mutate_graph (:560-592) */
static void update_bonds(const subctx *c, uint32_t v, const uint8_t *cstate, const uint8_t *substate, wset *before,
                         wset *after) {
    const ora_model *m = c->m;
    int32_t sv = c->var2sub[v];
    if (sv < 0) return;
    for (uint32_t i = c->adj_start[v]; i < c->adj_start[v + 1]; ++i) {
        uint32_t b = c->adj[i];
        uint32_t ov = m->bond_a[b] == v ? m->bond_b[b] : m->bond_a[b];
        int32_t so = c->var2sub[ov];
        if (so < 0) continue;
        if (cstate[sv] == cstate[so]) {
            wset_remove(before, b);
            if (after) wset_remove(after, b);
        } else {
            int32_t sa = c->var2sub[m->bond_a[b]], sb = c->var2sub[m->bond_b[b]];
            uint32_t ba = substate[sa], bb = substate[sb];
            double wbef = edge_w(m, b, ba, bb);
            wset_insert(before, b, 0, 0, wbef);
            if (after) {
                /* ws_for_flip (:665-683): flip the variable that is inside the cluster */
                int32_t flipsub = cstate[sv] ? sv : so;
                if (flipsub == sa) ba ^= 1u; else bb ^= 1u;
                wset_insert(after, b, 0, 0, edge_w(m, b, ba, bb));
            }
        }
    }
}

/* set_initial_bonds (:617-645) / the initial fill in mutate_graph (:366-380) */
static void initial_bonds(const subctx *c, const uint8_t *cstate, const uint8_t *substate, wset *before, wset *after) {
    const ora_model *m = c->m;
    for (uint32_t s = 0; s < c->nsub; ++s) {
        if (!cstate[s]) continue;
        uint32_t v = c->subvars[s];
        for (uint32_t i = c->adj_start[v]; i < c->adj_start[v + 1]; ++i) {
            uint32_t b = c->adj[i];
            uint32_t ov = m->bond_a[b] == v ? m->bond_b[b] : m->bond_a[b];
            int32_t so = c->var2sub[ov];
            if (so < 0 || cstate[so]) continue;
            int32_t sa = c->var2sub[m->bond_a[b]], sb = c->var2sub[m->bond_b[b]];
            uint32_t ba = substate[sa], bb = substate[sb];
            wset_insert(before, b, 0, 0, edge_w(m, b, ba, bb));
            if (after) {
                if ((int32_t)s == sa) ba ^= 1u; else bb ^= 1u;
                wset_insert(after, b, 0, 0, edge_w(m, b, ba, bb));
            }
        }
    }
}

/* find_overlapping_starts (:1125-1158): indices (relative to fp) of the segments of a neighbour that overlap the
 * imaginary-time interval [p_start, p_end) cyclically.  Writes them to out, returns the count. */
static uint32_t overlapping_starts(uint32_t p_start, uint32_t p_end, uint32_t cutoff, const uint32_t *fp, uint32_t L,
                                   uint32_t *out) {
    uint32_t bin = 0;
    while (bin < L && fp[bin] < p_start) bin++; /* insertion index; p_start never equals an entry */
    uint32_t prev = (bin + L - 1) % L;
    uint32_t lowest = fp[prev];
    uint32_t off_start = (p_start + cutoff - lowest) % cutoff, off_end = (p_end + cutoff - lowest) % cutoff;
    uint32_t cnt = 0;
    for (uint32_t step = 0; step < L; ++step) {
        uint32_t ip = (prev + step) % L;
        uint32_t p = fp[ip];
        uint32_t check_start = (p + cutoff - lowest) % cutoff;
        uint32_t next_p = fp[(ip + 1) % L];
        uint32_t check_end = (next_p + cutoff - lowest) % cutoff;
        int has_overlap_start = check_start < off_start && off_start < check_end;
        int has_start_within = off_start < check_start && check_start < off_end;
        int eq = (p_start == p_end) || (check_start == check_end);
        if (!(eq || has_overlap_start || has_start_within)) break;
        out[cnt++] = ip;
    }
    return cnt;
}

/* exported for the known-answer tests of the reference (rvb.rs:1228-1260) */
uint32_t ora_find_overlapping_starts(uint32_t p_start, uint32_t p_end, uint32_t cutoff, const uint32_t *fp, uint32_t L,
                                     uint32_t *out) {
    return overlapping_starts(p_start, p_end, cutoff, fp, L, out);
}

/* util/vec_help.rs:4-24 remove_doubles on a sorted array: cancel equal adjacent pairs */
uint32_t ora_remove_doubles(uint32_t *v, uint32_t n) {
    uint32_t ii = 0, jj = 0;
    while (jj + 1 < n) {
        if (v[jj] == v[jj + 1]) jj += 2;
        else v[ii++] = v[jj++];
    }
    if (jj < n) v[ii++] = v[jj++];
    return ii;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static int op_touches(const ora_model *m, uint32_t w, const int32_t *var2sub) {
    uint32_t b = sse_op_bond(w);
    if (var2sub[m->bond_a[b]] >= 0) return 1;
    return m->bond_b[b] != SSE_NO_VAR && var2sub[m->bond_b[b]] >= 0;
}

/* One RVB sweep of `updates` attempts.  Returns the number of accepted attempts. */
uint32_t ora_rvb_update(ora_replica *r, uint32_t updates) {
    const ora_model *m = r->m;
    const uint32_t N = m->nvars, M = r->cutoff, E = m->nedges;
    /* bonds_for_var: make_classical_bonds (qmc_ising.rs:421-432), edge order */
    uint32_t *adj_start = (uint32_t *)calloc(N + 2, sizeof(uint32_t));
    for (uint32_t e = 0; e < E; ++e) { adj_start[m->bond_a[e] + 1]++; adj_start[m->bond_b[e] + 1]++; }
    for (uint32_t v = 0; v < N; ++v) adj_start[v + 1] += adj_start[v];
    uint32_t *adj = (uint32_t *)malloc(sizeof(uint32_t) * (2 * E + 1));
    {
        uint32_t *fill = (uint32_t *)calloc(N, sizeof(uint32_t));
        for (uint32_t e = 0; e < E; ++e) {
            adj[adj_start[m->bond_a[e]] + fill[m->bond_a[e]]++] = e;
            adj[adj_start[m->bond_b[e]] + fill[m->bond_b[e]]++] = e;
        }
        free(fill);
    }
    /* find_constants (:1160-1187): per variable the positions of its constant (transverse) ops in p order */
    uint32_t *vstart = (uint32_t *)calloc(N + 1, sizeof(uint32_t)), *vlen = (uint32_t *)calloc(N, sizeof(uint32_t));
    uint32_t C = 0;
    for (uint32_t p = 0; p < M; ++p) {
        uint32_t w = r->ops[p];
        if (w && (m->binfo[sse_op_bond(w)] & SSE_BOND_KIND_MASK) == SSE_BOND_TRANSVERSE) { vlen[m->bond_a[sse_op_bond(w)]]++; C++; }
    }
    for (uint32_t v = 0; v < N; ++v) vstart[v + 1] = vstart[v] + vlen[v];
    uint32_t *cps = (uint32_t *)malloc(sizeof(uint32_t) * (C + 1));
    uint32_t *zero_vars = (uint32_t *)malloc(sizeof(uint32_t) * N), nzero = 0;
    {
        uint32_t *fill = (uint32_t *)calloc(N, sizeof(uint32_t));
        for (uint32_t p = 0; p < M; ++p) {
            uint32_t w = r->ops[p];
            if (w && (m->binfo[sse_op_bond(w)] & SSE_BOND_KIND_MASK) == SSE_BOND_TRANSVERSE) {
                uint32_t v = m->bond_a[sse_op_bond(w)];
                cps[vstart[v] + fill[v]++] = p;
            }
        }
        free(fill);
        for (uint32_t v = 0; v < N; ++v) if (vlen[v] == 0) zero_vars[nzero++] = v;
    }
    int32_t *var2sub = (int32_t *)malloc(sizeof(int32_t) * N);
    for (uint32_t v = 0; v < N; ++v) var2sub[v] = -1;
    uint8_t *popped_f = (uint8_t *)calloc(C + 1, 1), *popped_n = (uint8_t *)calloc(N, 1);
    uint32_t *ovl = (uint32_t *)malloc(sizeof(uint32_t) * (C + 1));
    uint32_t nsucc = 0;

    for (uint32_t attempt = 0; attempt < updates; ++attempt) {
        rvb_rng g = {r, attempt, 0};
        uint32_t o[4];
        /* ---- starting segment (:117-135) ---- */
        rvb_draw(&g, o);
        uint32_t choice = mulhi32(o[0], C + nzero);
        uint32_t v0, f0;
        if (choice < C) {
            uint32_t lo = 0, hi = N; /* the variable that owns entry `choice` */
            while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (vstart[mid] <= choice) lo = mid; else hi = mid; }
            while (vlen[lo] == 0) lo--; /* unreachable: kept for clarity */
            v0 = lo; f0 = choice;
        } else { v0 = zero_vars[choice - C]; f0 = SSE_NO_VAR; }
        /* ---- cluster size (:140, :1190-1192): trailing ones of 64 random bits, plus one ---- */
        rvb_draw(&g, o);
        uint64_t bits = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
        uint32_t csize = 1;
        while ((bits & 1ull) && csize <= 64) { csize++; bits >>= 1; }
        /* ---- build_cluster (:1054-1123) with the weighted boundary manager (:967-1052) ---- */
        wset bf, bn;
        wset_init(&bf, C + 1);
        wset_init(&bn, N);
        uint32_t *cl_v = (uint32_t *)malloc(sizeof(uint32_t) * (csize + 1)), *cl_f = (uint32_t *)malloc(sizeof(uint32_t) * (csize + 1));
        uint32_t ncl = 0;
        uint32_t *touched_f = (uint32_t *)malloc(sizeof(uint32_t) * (csize + 1)), *touched_n = (uint32_t *)malloc(sizeof(uint32_t) * (csize + 1));
        uint32_t ntf = 0, ntn = 0;
#define PUSH_ADJ(var, pos, wgt)                                                                                   \
    do {                                                                                                          \
        if ((pos) != SSE_NO_VAR) { if (!popped_f[pos]) wset_insert(&bf, (pos), (var), (pos), wset_weight(&bf, (pos)) + (wgt)); } \
        else if (!popped_n[var]) wset_insert(&bn, (var), (var), SSE_NO_VAR, wset_weight(&bn, (var)) + (wgt));   \
    } while (0)
        PUSH_ADJ(v0, f0, 1.0);
        uint32_t left = csize;
        while (left > 0 && (bf.n + bn.n) > 0) {
            /* pop_index (:1010-1026) */
            rvb_draw(&g, o);
            double f_ratio = bf.total / (bf.total + bn.total);
            int pick_flips = u01(o[0]) < f_ratio;
            if (bf.n == 0) pick_flips = 0;
            if (bn.n == 0) pick_flips = 1;
            wset *s = pick_flips ? &bf : &bn;
            rvb_draw(&g, o);
            uint32_t idx = wset_pick(s, u01(o[0]));
            uint32_t v = s->e[idx].v, flip = s->e[idx].p;
            if (pick_flips) { popped_f[flip] = 1; touched_f[ntf++] = flip; } else { popped_n[v] = 1; touched_n[ntn++] = v; }
            wset_remove(s, s->e[idx].key);
            cl_v[ncl] = v; cl_f[ncl] = flip; ncl++;
            if (flip != SSE_NO_VAR) {
                uint32_t rel = flip - vstart[v];
                uint32_t fdec = (rel + vlen[v] - 1) % vlen[v] + vstart[v], finc = (rel + 1) % vlen[v] + vstart[v];
                PUSH_ADJ(v, fdec, 1.0);
                PUSH_ADJ(v, finc, 1.0);
            }
            for (uint32_t i = adj_start[v]; i < adj_start[v + 1]; ++i) {
                uint32_t b = adj[i];
                double weight = fabs(m->bweight[b]) * 0.5; /* bond_mag = |J| (qmc_ising.rs:633-635) */
                uint32_t ov = m->bond_a[b] == v ? m->bond_b[b] : m->bond_a[b];
                if (vlen[ov] == 0) { PUSH_ADJ(ov, SSE_NO_VAR, weight); }
                else if (flip != SSE_NO_VAR) {
                    uint32_t rel = flip - vstart[v];
                    uint32_t finc = (rel + 1) % vlen[v] + vstart[v];
                    uint32_t cnt = overlapping_starts(cps[flip], cps[finc], M, cps + vstart[ov], vlen[ov], ovl);
                    for (uint32_t q = 0; q < cnt; ++q) { uint32_t fpz = ovl[q] + vstart[ov]; PUSH_ADJ(ov, fpz, weight); }
                } else {
                    for (uint32_t pi = vstart[ov]; pi < vstart[ov] + vlen[ov]; ++pi) PUSH_ADJ(ov, pi, weight);
                }
            }
            left--;
        }
#undef PUSH_ADJ
        /* ---- sub-variables = cluster + remaining boundary (:155-172) ---- */
        uint32_t nsv_cap = ncl + bf.n + bn.n;
        uint32_t *subvars = (uint32_t *)malloc(sizeof(uint32_t) * (nsv_cap + 1));
        uint32_t nsub = 0;
        for (uint32_t i = 0; i < ncl; ++i) subvars[nsub++] = cl_v[i];
        for (uint32_t i = 0; i < bf.n; ++i) subvars[nsub++] = bf.e[i].v;
        for (uint32_t i = 0; i < bn.n; ++i) subvars[nsub++] = bn.e[i].v;
        qsort(subvars, nsub, sizeof(uint32_t), cmp_u32);
        { uint32_t k2 = 0; for (uint32_t i = 0; i < nsub; ++i) if (i == 0 || subvars[i] != subvars[i - 1]) subvars[k2++] = subvars[i]; nsub = k2; }
        for (uint32_t s = 0; s < nsub; ++s) var2sub[subvars[s]] = (int32_t)s;
        for (uint32_t i = 0; i < ntf; ++i) popped_f[touched_f[i]] = 0;
        for (uint32_t i = 0; i < ntn; ++i) popped_n[touched_n[i]] = 0;
        /* cluster_starting_state and toggle positions (:174-196, :230-231) */
        uint8_t *cstart = (uint8_t *)calloc(nsub, 1);
        uint32_t *toggles = (uint32_t *)malloc(sizeof(uint32_t) * (2 * ncl + 1));
        uint32_t ntog = 0;
        for (uint32_t i = 0; i < ncl; ++i) {
            uint32_t v = cl_v[i], fi = cl_f[i];
            int32_t sv = var2sub[v];
            if (fi != SSE_NO_VAR) {
                uint32_t rel = fi - vstart[v];
                if (rel + 1 >= vlen[v]) { cstart[sv] = 1; toggles[ntog++] = cps[fi]; toggles[ntog++] = cps[vstart[v]]; }
                else { toggles[ntog++] = cps[fi]; toggles[ntog++] = cps[fi + 1]; }
            } else cstart[sv] = 1;
        }
        qsort(toggles, ntog, sizeof(uint32_t), cmp_u32);
        ntog = ora_remove_doubles(toggles, ntog);
        subctx ctx = {m, nsub, subvars, var2sub, adj_start, adj};

        /* ---- calculate_flip_prob (:649-946) ---- */
        uint8_t *substate = (uint8_t *)malloc(nsub), *cstate = (uint8_t *)malloc(nsub);
        for (uint32_t s = 0; s < nsub; ++s) { substate[s] = r->state[subvars[s]]; cstate[s] = cstart[s]; }
        uint32_t csz = 0;
        for (uint32_t s = 0; s < nsub; ++s) csz += cstate[s];
        double mult = 1.0;
        {
            wset before, after;
            wset_init(&before, E);
            wset_init(&after, E);
            uint32_t nb = 0, next = 0;
            if (csz) initial_bonds(&ctx, cstate, substate, &before, &after);
            int broke = 0;
            for (uint32_t p = 0; p < M && !broke; ++p) {
                uint32_t w = r->ops[p];
                if (!w || !op_touches(m, w, var2sub)) continue;
                uint32_t b = sse_op_bond(w), in = sse_op_in(w), out = sse_op_out(w);
                uint32_t va = m->bond_a[b], vc = m->bond_b[b];
                int32_t sa = var2sub[va], sc = (vc != SSE_NO_VAR) ? var2sub[vc] : -1;
                int is_bound = next < ntog && p == toggles[next];
                if (csz == 0 && !is_bound) {
                    if (next >= ntog) break; /* done with clusters (:727-732) */
                    /* skipping ahead: only the propagated sub-state moves (:756-759) */
                    if (sa >= 0) substate[sa] = (uint8_t)(out & 1u);
                    if (sc >= 0) substate[sc] = (uint8_t)((out >> 1) & 1u);
                    continue;
                }
                int offdiag = in != out;
                int all_in = (sa >= 0 && cstate[sa]) && (vc == SSE_NO_VAR || (sc >= 0 && cstate[sc]));
                if (wset_contains(&before, b)) {
                    nb++;
                } else {
                    if (is_bound) {
                        cstate[sa] ^= 1u;
                        if (cstate[sa]) csz++; else csz--;
                        next++;
                    }
                    if (offdiag) {
                        if (sa >= 0) substate[sa] = (uint8_t)(out & 1u);
                        if (sc >= 0) substate[sc] = (uint8_t)((out >> 1) & 1u);
                    }
                    if (all_in) {
                        /* ising_ratio (qmc_ising.rs:722-735): 0 for a longitudinal op, else 1 */
                        if ((m->binfo[b] & SSE_BOND_KIND_MASK) == SSE_BOND_LONGITUDINAL) mult *= 0.0;
                        if (mult < DBL_EPSILON) { broke = 1; break; }
                    }
                    if (offdiag || is_bound) {
                        mult *= calc_mult(&before, &after, nb);
                        nb = 0;
                        if (mult < DBL_EPSILON) { broke = 1; break; }
                        update_bonds(&ctx, va, cstate, substate, &before, &after);
                        if (vc != SSE_NO_VAR) update_bonds(&ctx, vc, cstate, substate, &before, &after);
                    }
                }
            }
            mult *= calc_mult(&before, &after, nb);
            wset_free(&before);
            wset_free(&after);
        }
        /* ---- accept (:241-246) ---- */
        rvb_draw(&g, o);
        int accept = (mult >= 1.0) || (u01(o[0]) < mult);
        if (accept) {
            /* ---- mutate_graph (:294-615) ---- */
            for (uint32_t s = 0; s < nsub; ++s) { substate[s] = r->state[subvars[s]]; cstate[s] = cstart[s]; }
            uint32_t count = 0;
            for (uint32_t s = 0; s < nsub; ++s) count += cstate[s];
            const int has_start = count != 0;
            /* windows where the cluster is non-empty (:310-360) */
            uint32_t *wfrom = (uint32_t *)malloc(sizeof(uint32_t) * (ntog + 2)), *wuntil = (uint32_t *)malloc(sizeof(uint32_t) * (ntog + 2));
            uint32_t nwin = 0, nuntil = 0;
            {
                uint8_t *cs2 = (uint8_t *)malloc(nsub);
                memcpy(cs2, cstate, nsub);
                uint32_t cnt = count;
                if (cnt) wfrom[nwin++] = 0;
                for (uint32_t i = 0; i < ntog; ++i) {
                    uint32_t p = toggles[i];
                    if (cnt == 0) wfrom[nwin++] = p;
                    int32_t sv = var2sub[m->bond_a[sse_op_bond(r->ops[p])]];
                    cs2[sv] ^= 1u;
                    if (cs2[sv]) cnt++; else cnt--;
                    if (cnt == 0) wuntil[nuntil++] = p;
                }
                if (cnt) wuntil[nuntil++] = M;
                free(cs2);
            }
            wset bonds;
            wset_init(&bonds, E);
            if (has_start) {
                for (uint32_t s = 0; s < nsub; ++s) substate[s] ^= cstate[s];
                initial_bonds(&ctx, cstate, substate, &bonds, NULL);
            }
            uint32_t next = 0;
            for (uint32_t wi = 0; wi < nwin; ++wi) {
                uint32_t from = wfrom[wi], until = wuntil[wi];
                /* get_propagated_substate_with_hint (fast_ops.rs:1027-1172): state of the sub-variables at `from` */
                if (from > 0) {
                    for (uint32_t p = 0; p < from; ++p) {
                        uint32_t w = r->ops[p];
                        if (!w) continue;
                        uint32_t b = sse_op_bond(w), out = sse_op_out(w);
                        int32_t sa = var2sub[m->bond_a[b]];
                        if (sa >= 0) substate[sa] = (uint8_t)(out & 1u);
                        if (m->bond_b[b] != SSE_NO_VAR) { int32_t sc = var2sub[m->bond_b[b]]; if (sc >= 0) substate[sc] = (uint8_t)((out >> 1) & 1u); }
                    }
                    for (uint32_t s = 0; s < nsub; ++s) substate[s] ^= cstate[s];
                }
                /* the window is inclusive: the closing cut at `until` is mutated too (fast_ops.rs:1222 `node_p > pend`) */
                for (uint32_t p = from; p <= until && p < M; ++p) {
                    uint32_t w = r->ops[p];
                    if (!w || !op_touches(m, w, var2sub)) continue;
                    uint32_t b = sse_op_bond(w), in = sse_op_in(w), out = sse_op_out(w);
                    uint32_t va = m->bond_a[b], vc = m->bond_b[b];
                    int32_t sa = var2sub[va], sc = (vc != SSE_NO_VAR) ? var2sub[vc] : -1;
                    int at_flip = next < ntog && p == toggles[next];
                    if (wset_contains(&bonds, b)) {
                        /* rotate the boundary op onto a boundary bond drawn by weight (:414-432) */
                        rvb_draw(&g, o);
                        uint32_t nbnd = bonds.e[wset_pick(&bonds, u01(o[0]))].key;
                        uint32_t s2 = substate[var2sub[m->bond_a[nbnd]]] | ((uint32_t)substate[var2sub[m->bond_b[nbnd]]] << 1);
                        r->ops[p] = sse_op_make(nbnd, s2, s2);
                        continue;
                    }
                    if (at_flip) {
                        uint32_t cs = cstate[sa];
                        uint32_t nin = (in & 1u) ^ cs, nout = (out & 1u) ^ (cs ^ 1u);
                        r->ops[p] = sse_op_make(b, nin, nout);
                        cstate[sa] ^= 1u;
                        substate[sa] = (uint8_t)nout;
                        next++;
                    } else {
                        int any_in = (sa >= 0 && cstate[sa]) || (sc >= 0 && cstate[sc]);
                        int diag = in == out;
                        if (any_in) {
                            uint32_t mask = (vc != SSE_NO_VAR) ? 3u : 1u;
                            uint32_t nin = in ^ mask, nout = out ^ mask;
                            r->ops[p] = sse_op_make(b, nin, nout);
                            if (nin != nout) {
                                if (sa >= 0) substate[sa] = (uint8_t)(nout & 1u);
                                if (sc >= 0) substate[sc] = (uint8_t)((nout >> 1) & 1u);
                            }
                        } else if (!diag) {
                            if (sa >= 0) substate[sa] = (uint8_t)(out & 1u);
                            if (sc >= 0) substate[sc] = (uint8_t)((out >> 1) & 1u);
                        } else {
                            continue; /* diagonal and untouched by the cluster: nothing to do, bonds unchanged (:513-514) */
                        }
                    }
                    update_bonds(&ctx, va, cstate, substate, &bonds, NULL);
                    if (vc != SSE_NO_VAR) update_bonds(&ctx, vc, cstate, substate, &bonds, NULL);
                }
            }
            wset_free(&bonds);
            free(wfrom); free(wuntil);
            if (has_start)
                for (uint32_t s = 0; s < nsub; ++s) r->state[subvars[s]] ^= cstart[s];
            nsucc++;
        }
        for (uint32_t s = 0; s < nsub; ++s) var2sub[subvars[s]] = -1;
        free(substate); free(cstate); free(cstart); free(toggles); free(subvars);
        free(cl_v); free(cl_f); free(touched_f); free(touched_n);
        wset_free(&bf); wset_free(&bn);
    }
    free(adj_start); free(adj); free(vstart); free(vlen); free(cps); free(zero_vars); free(var2sub);
    free(popped_f); free(popped_n); free(ovl);
    r->acc[4] += updates;
    r->epoch += 1;
    return nsucc;
}
