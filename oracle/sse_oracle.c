/*
 * sse_oracle.c — CPU ORACLE (test infrastructure, NOT product; see sse_oracle.h header).
 *
 * Sequential restatement of the SSE sweep of Renmusxd/IsingMonteCarlo in "Philox-indexed" form:
 * every random number is a pure function of (seed, replica, epoch, tag, index), so the result does
 * not depend on evaluation order and can be compared bit-for-bit with the parallel HIP kernels.
 * Citations are relative to /root/reference.
 */
#include "sse_oracle_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Philox4x32-10 --- */
/* Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11).   */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* --------------------------------------------------------------------------- model --- */
/* Bond numbering and offsets: src/sse/qmc_ising.rs:92-104 (nvars, offsets), :186-205 and :228-246
 * (bond ranges), weights :863-888. */
ora_model *ora_model_create(uint32_t nvars, uint32_t nedges, const uint32_t *edge_a, const uint32_t *edge_b,
                            const double *J, double gamma, double h) {
    ora_model *m = (ora_model *)calloc(1, sizeof(*m));
    int has_long = fabs(h) > DBL_EPSILON;
    m->nvars = nvars;
    m->nedges = nedges;
    m->nbonds = nedges + nvars + (has_long ? nvars : 0);
    m->gamma = gamma;
    m->h = h;
    m->bond_a = (uint32_t *)malloc(sizeof(uint32_t) * m->nbonds);
    m->bond_b = (uint32_t *)malloc(sizeof(uint32_t) * m->nbonds);
    m->binfo = (uint32_t *)malloc(sizeof(uint32_t) * m->nbonds);
    m->bweight = (double *)malloc(sizeof(double) * m->nbonds);
    m->cumw = (double *)malloc(sizeof(double) * m->nbonds);
    double off = 0.0;
    for (uint32_t e = 0; e < nedges; ++e) {
        m->bond_a[e] = edge_a[e];
        m->bond_b[e] = edge_b[e];
        m->binfo[e] = SSE_BOND_TWO_SITE | (J[e] < 0.0 ? SSE_BOND_PREF_BIT : 0u);
        m->bweight[e] = 2.0 * fabs(J[e]);
        off += fabs(J[e]);
    }
    for (uint32_t v = 0; v < nvars; ++v) {
        uint32_t b = nedges + v;
        m->bond_a[b] = v;
        m->bond_b[b] = SSE_NO_VAR;
        m->binfo[b] = SSE_BOND_TRANSVERSE;
        m->bweight[b] = gamma;
    }
    if (has_long)
        for (uint32_t v = 0; v < nvars; ++v) {
            uint32_t b = nedges + nvars + v;
            m->bond_a[b] = v;
            m->bond_b[b] = SSE_NO_VAR;
            m->binfo[b] = SSE_BOND_LONGITUDINAL | (h > 0.0 ? SSE_BOND_PREF_BIT : 0u);
            m->bweight[b] = 2.0 * fabs(h);
        }
    m->offset = off + (double)nvars * (gamma + fabs(h));
    double c = 0.0;
    for (uint32_t b = 0; b < m->nbonds; ++b) {
        c = (b == 0) ? m->bweight[0] : m->bweight[b] + c;
        m->cumw[b] = c;
    }
    m->wtot = c;
    return m;
}
/* Generic interactions (Qmc, qmc_runner.rs:94-156 make_interaction*, Interaction :415-680): bond b acts on k[b] = 1 or 2
 * variables with an arbitrary non-negative weight matrix, given here already in the op-word layout
 * mats[b][in | out << 2] (bit 0 = first variable, bit 1 = second).  Bond kinds only feed the transverse-op counters:
 * a one-variable bond whose four entries are equal is a cluster edge (is_valid_cluster_edge, cluster.rs:284-286). */
ora_model *ora_model_create_generic(uint32_t nvars, uint32_t nbonds, const uint32_t *k, const uint32_t *var_a,
                                    const uint32_t *var_b, const double *mats, double offset) {
    ora_model *m = (ora_model *)calloc(1, sizeof(*m));
    m->nvars = nvars;
    m->nedges = 0;
    m->nbonds = nbonds;
    m->bond_a = (uint32_t *)malloc(sizeof(uint32_t) * nbonds);
    m->bond_b = (uint32_t *)malloc(sizeof(uint32_t) * nbonds);
    m->binfo = (uint32_t *)malloc(sizeof(uint32_t) * nbonds);
    m->bweight = (double *)malloc(sizeof(double) * nbonds);
    m->cumw = (double *)malloc(sizeof(double) * nbonds);
    m->mats = (double *)malloc(sizeof(double) * 16 * nbonds);
    memcpy(m->mats, mats, sizeof(double) * 16 * nbonds);
    double c = 0.0;
    for (uint32_t b = 0; b < nbonds; ++b) {
        const double *mb = mats + 16 * (size_t)b;
        m->bond_a[b] = var_a[b];
        m->bond_b[b] = k[b] == 2 ? var_b[b] : SSE_NO_VAR;
        double maxw = 0.0; /* largest diagonal element (heatbath.rs:130-146 make_bond_weights) */
        for (uint32_t s = 0; s < (k[b] == 2 ? 4u : 2u); ++s) if (mb[s | (s << 2)] > maxw) maxw = mb[s | (s << 2)];
        m->bweight[b] = maxw;
        if (k[b] == 2) m->binfo[b] = SSE_BOND_TWO_SITE;
        else m->binfo[b] = (mb[0] == mb[1] && mb[0] == mb[4] && mb[0] == mb[5]) ? SSE_BOND_TRANSVERSE : SSE_BOND_LONGITUDINAL;
        c = (b == 0) ? maxw : maxw + c;
        m->cumw[b] = c;
    }
    m->wtot = c;
    m->offset = offset;
    return m;
}
/* Interaction::at (qmc_runner.rs:573-612) with index_from_state (:666-679): index = outputs then inputs, every bit
 * list MSB first; a Diagonal interaction (kind 1) holds only its 2^k diagonal and is 0 elsewhere (:594-610). */
double ora_interaction_at(uint32_t k, uint32_t diagonal, const double *mat, const uint8_t *inputs, const uint8_t *outputs) {
    size_t idx = 0;
    if (diagonal) {
        for (uint32_t i = 0; i < k; ++i) if ((inputs[i] != 0) != (outputs[i] != 0)) return 0.0;
        for (uint32_t i = 0; i < k; ++i) idx = (idx << 1) | (inputs[i] ? 1u : 0u);
        return mat[idx];
    }
    for (uint32_t i = 0; i < k; ++i) idx = (idx << 1) | (outputs[i] ? 1u : 0u);
    for (uint32_t i = 0; i < k; ++i) idx = (idx << 1) | (inputs[i] ? 1u : 0u);
    return mat[idx];
}
/* Interaction::sym_under_ising (qmc_runner.rs:639-664), index ranges exactly as written there */
int ora_interaction_sym_under_ising(uint32_t k, uint32_t diagonal, const double *mat) {
    const size_t mask = diagonal ? (((size_t)1 << k) - 1) : (((size_t)1 << (2 * k)) - 1);
    const size_t upto = diagonal ? ((size_t)1 << (k >> 1)) : ((size_t)1 << k);
    for (size_t i = 0; i < upto; ++i) {
        const double d = mat[i] - mat[(~i) & mask];
        if (!((d < 0 ? -d : d) < 2.220446049250313e-16)) return 0;
    }
    return 1;
}
void ora_model_destroy(ora_model *m) {
    if (!m) return;
    free(m->bond_a); free(m->bond_b); free(m->binfo); free(m->bweight); free(m->cumw); free(m->mats); free(m);
}
uint32_t ora_model_nbonds(const ora_model *m) { return m->nbonds; }
double ora_model_offset(const ora_model *m) { return m->offset; }

/* matrix element <out|H_b|in> of the shifted bond operator (qmc_ising.rs:863-888) */
double ora_bond_weight(const ora_model *m, uint32_t b, uint32_t in, uint32_t out) {
    if (m->mats) return m->mats[16 * (size_t)b + (in | (out << 2))]; /* Interaction::at (qmc_runner.rs:573-612) */
    uint32_t info = m->binfo[b];
    uint32_t pref = (info & SSE_BOND_PREF_BIT) ? 1u : 0u;
    switch (info & SSE_BOND_KIND_MASK) {
    case SSE_BOND_TWO_SITE: {
        if (in != out) return 0.0;
        uint32_t aligned = ((in & 1u) == ((in >> 1) & 1u)) ? 1u : 0u;
        return aligned == pref ? m->bweight[b] : 0.0;
    }
    case SSE_BOND_TRANSVERSE:
        return m->bweight[b];
    default:
        if (in != out) return 0.0;
        return (in & 1u) == pref ? m->bweight[b] : 0.0;
    }
}

/* ------------------------------------------------------------------------- replica --- */
ora_replica *ora_replica_create(const ora_model *m, uint32_t capacity, uint32_t cutoff0, uint64_t seed,
                                uint32_t replica, const uint8_t *init_state) {
    if (cutoff0 > capacity) return NULL;
    ora_replica *r = (ora_replica *)calloc(1, sizeof(*r));
    r->m = m;
    r->cap = capacity;
    r->cutoff = cutoff0;
    r->seed = seed;
    r->replica = replica;
    r->ops = (uint32_t *)calloc(capacity ? capacity : 1, sizeof(uint32_t));
    r->state = (uint8_t *)calloc(m->nvars, 1);
    r->parent = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)m->nvars + capacity));
    r->cur = (uint32_t *)malloc(sizeof(uint32_t) * m->nvars);
    r->flip = (uint8_t *)malloc((size_t)m->nvars + capacity);
    r->frozen = (uint8_t *)malloc((size_t)m->nvars + capacity);
    r->touched = (uint8_t *)malloc(m->nvars);
    if (init_state) {
        for (uint32_t v = 0; v < m->nvars; ++v) r->state[v] = init_state[v] ? 1 : 0;
    } else {
        /* classical/graph.rs:451-453 make_random_spin_state: one fair bit per variable */
        for (uint32_t v = 0; v < m->nvars; ++v) {
            uint32_t o[4];
            draw(r, SSE_TAG_INIT, v, o);
            r->state[v] = (uint8_t)(o[0] >> 31);
        }
    }
    return r;
}
void ora_replica_destroy(ora_replica *r) {
    if (!r) return;
    free(r->ops); free(r->state); free(r->parent); free(r->cur); free(r->flip); free(r->frozen);
    free(r->touched); free(r);
}

static uint32_t substate(const ora_replica *r, uint32_t b) {
    const ora_model *m = r->m;
    uint32_t s = r->state[m->bond_a[b]];
    if (m->bond_b[b] != SSE_NO_VAR) s |= (uint32_t)r->state[m->bond_b[b]] << 1;
    return s;
}
static void apply_outputs(ora_replica *r, uint32_t w) {
    const ora_model *m = r->m;
    uint32_t b = sse_op_bond(w), out = sse_op_out(w);
    r->state[m->bond_a[b]] = (uint8_t)(out & 1u);
    if (m->bond_b[b] != SSE_NO_VAR) r->state[m->bond_b[b]] = (uint8_t)((out >> 1) & 1u);
}

/* Metropolis diagonal sweep: qmc_traits/diagonal.rs:114-135 (sweep) and :142-191 (slot rule).
 *   empty   : b uniform in [0,Nb); insert iff  num > den || bernoulli(num/den),  den = M - n
 *   diagonal: remove iff den+1 > num || bernoulli((den+1)/num)
 *   offdiag : propagate the state.
 * bernoulli(x) with x>=1 always true, so both tests are written u*den < num / u*num < den+1 with one
 * uniform u in [0,1) (exact IEEE double multiply + compare; identical on CPU and GPU). */
void ora_diagonal_update(ora_replica *r, double beta) {
    const ora_model *m = r->m;
    const double beta_nb = beta * (double)m->nbonds;
    const uint32_t M = r->cutoff;
    for (uint32_t p = 0; p < M; ++p) {
        uint32_t w = r->ops[p];
        if (w == SSE_OP_EMPTY) {
            uint32_t o[2];
            draw_diag(r, p, o);
            uint32_t b = mulhi32(o[0], m->nbonds);
            uint32_t s = substate(r, b);
            double num = beta_nb * ora_bond_weight(m, b, s, s);
            double den = (double)(M - r->n);
            if (u01(o[1]) * den < num) {
                r->ops[p] = sse_op_make(b, s, s);
                r->n += 1;
            }
        } else if (sse_op_is_diagonal(w)) {
            uint32_t o[2];
            draw_diag(r, p, o);
            uint32_t b = sse_op_bond(w);
            double num = beta_nb * ora_bond_weight(m, b, sse_op_in(w), sse_op_in(w));
            double den = (double)(M - r->n + 1u);
            if (u01(o[1]) * num < den) {
                r->ops[p] = SSE_OP_EMPTY;
                r->n -= 1;
            }
        } else {
            apply_outputs(r, w);
        }
    }
    r->acc[5] += M;
    r->epoch += 1;
}

/* Heat-bath diagonal sweep: qmc_traits/heatbath.rs:149-209, bond table :8-61.
 *   empty   : bernoulli(bW/((M-n)+bW)); then p=u1, c=u2*W -> first bond with cum >= c; accept iff p*maxw < w
 *   diagonal: remove with bernoulli((M-n+1)/((M-n+1)+bW)). */
void ora_heatbath_update(ora_replica *r, double beta) {
    const ora_model *m = r->m;
    const uint32_t M = r->cutoff;
    const double bw = beta * m->wtot;
    for (uint32_t p = 0; p < M; ++p) {
        uint32_t w = r->ops[p];
        if (w == SSE_OP_EMPTY) {
            uint32_t o[4];
            draw(r, SSE_TAG_HEATBATH, p, o);
            double den = (double)(M - r->n) + bw;
            if (u01(o[0]) * den < bw) {
                double c = u01(o[2]) * m->wtot;
                uint32_t lo = 0, hi = m->nbonds; /* first index with cum >= c */
                while (lo < hi) {
                    uint32_t mid = lo + (hi - lo) / 2;
                    if (m->cumw[mid] < c) lo = mid + 1; else hi = mid;
                }
                uint32_t b = lo < m->nbonds ? lo : m->nbonds - 1;
                uint32_t s = substate(r, b);
                if (u01(o[1]) * m->bweight[b] < ora_bond_weight(m, b, s, s)) {
                    r->ops[p] = sse_op_make(b, s, s);
                    r->n += 1;
                }
            }
        } else if (sse_op_is_diagonal(w)) {
            uint32_t o[4];
            draw(r, SSE_TAG_HEATBATH, p, o);
            double num = (double)(M - r->n + 1u);
            double den = num + bw;
            if (u01(o[0]) * den < num) {
                r->ops[p] = SSE_OP_EMPTY;
                r->n -= 1;
            }
        } else {
            apply_outputs(r, w);
        }
    }
    r->acc[5] += M;
    r->epoch += 1;
}

/* ------------------------------------------------------------------- cluster update --- */
static uint32_t uf_find(uint32_t *parent, uint32_t x) {
    while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
    }
    return x;
}
static void uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) parent[b] = a; else parent[a] = b; /* root = smallest id: canonical label */
}

/* Cluster update: qmc_traits/cluster.rs:36-172.  Sides of ops are partitioned into clusters:
 *  - single-site constant ops (is_valid_cluster_edge :284-286) separate their input and output side;
 *  - every other op joins all of its legs (:205-211, :249-266);
 *  - consecutive legs on a worldline are joined, cyclically in imaginary time (:223-242).
 * Restated with worldline SEGMENTS: ids [0,N) = the part of worldline v containing p=0 (before its
 * first cut, joined cyclically to the part after its last cut); id N+k = the segment that starts at
 * the k-th cut in p order.  Canonical label of a cluster = its smallest segment id; the coin is
 * Philox(tag CLUSTER, index = label).  No cut anywhere => one cluster (:98-107), label 0.
 * Weighted variant (:111-135 with qmc_ising.rs:759-775): a cluster holding a longitudinal op gets
 * probability prob*0 and never flips. */
uint32_t ora_cluster_update(ora_replica *r, double prob) {
    const ora_model *m = r->m;
    const uint32_t N = m->nvars, M = r->cutoff;
    uint32_t nclusters = 0;
    if (r->n == 0) { /* cluster.rs:46-48 */
        r->epoch += 1;
        return 0;
    }
    uint32_t *parent = r->parent, *cur = r->cur;
    uint8_t *flip = r->flip, *frozen = r->frozen, *touched = r->touched;
    for (uint32_t v = 0; v < N; ++v) { parent[v] = v; cur[v] = v; touched[v] = 0; frozen[v] = 0; }
    uint32_t ncuts = 0;
    int any_long = 0;
    for (uint32_t p = 0; p < M; ++p) {
        uint32_t w = r->ops[p];
        if (w == SSE_OP_EMPTY) continue;
        uint32_t b = sse_op_bond(w), kind = m->binfo[b] & SSE_BOND_KIND_MASK;
        uint32_t a = m->bond_a[b];
        touched[a] = 1;
        if (kind == SSE_BOND_TRANSVERSE) {
            uint32_t id = N + ncuts++;
            parent[id] = id;
            frozen[id] = 0;
            cur[a] = id;
        } else if (kind == SSE_BOND_TWO_SITE) {
            uint32_t c = m->bond_b[b];
            touched[c] = 1;
            uf_union(parent, cur[a], cur[c]);
        } else if (!m->mats) { /* longitudinal-field op of the Ising model: its cluster never flips */
            frozen[cur[a]] = 1;
            any_long = 1;
        } /* generic Ising-symmetric interactions (Qmc::cluster_update, qmc_runner.rs:222-236): an interior op */
    }
    const uint32_t S = N + ncuts;
    if (ncuts == 0) {
        /* whole graph is one cluster */
        uint32_t o[4];
        draw(r, SSE_TAG_CLUSTER, 0, o);
        uint8_t f = (!any_long && u01(o[0]) < prob) ? 1 : 0;
        for (uint32_t v = 0; v < N; ++v) { parent[v] = v; flip[v] = touched[v] ? f : 0; }
        nclusters = 1;
    } else {
        for (uint32_t v = 0; v < N; ++v)
            if (cur[v] != v) uf_union(parent, v, cur[v]);
        /* propagate frozen to roots, then draw one coin per root */
        for (uint32_t i = 0; i < S; ++i)
            if (frozen[i]) frozen[uf_find(parent, i)] = 1;
        for (uint32_t i = 0; i < S; ++i) {
            uint32_t root = uf_find(parent, i);
            if (root == i && (i >= N || touched[i])) nclusters++;
            uint32_t o[4];
            draw(r, SSE_TAG_CLUSTER, root, o);
            flip[i] = (!frozen[root] && u01(o[0]) < prob) ? 1 : 0;
        }
    }
    /* apply (cluster.rs:139-167): toggle input bits by the flip of the incoming segment, output bits
     * by the outgoing one; p=0 state follows the segment that contains p=0. */
    for (uint32_t v = 0; v < N; ++v) cur[v] = v;
    uint32_t k = 0;
    for (uint32_t p = 0; p < M; ++p) {
        uint32_t w = r->ops[p];
        if (w == SSE_OP_EMPTY) continue;
        uint32_t b = sse_op_bond(w), kind = m->binfo[b] & SSE_BOND_KIND_MASK;
        uint32_t a = m->bond_a[b];
        uint32_t in = sse_op_in(w), out = sse_op_out(w);
        if (kind == SSE_BOND_TRANSVERSE) {
            uint32_t id = (ncuts ? N + k++ : a);
            in ^= flip[cur[a]];
            out ^= flip[id];
            cur[a] = id;
        } else if (kind == SSE_BOND_TWO_SITE) {
            uint32_t c = m->bond_b[b];
            uint32_t fa = flip[cur[a]], fc = flip[cur[c]];
            in ^= fa | (fc << 1);
            out ^= fa | (fc << 1);
        } else {
            uint32_t fa = flip[cur[a]];
            in ^= fa;
            out ^= fa;
        }
        r->ops[p] = sse_op_make(b, in, out);
    }
    for (uint32_t v = 0; v < N; ++v)
        if (touched[v]) r->state[v] ^= flip[v];
    r->acc[4] += r->n;
    r->epoch += 1;
    return nclusters;
}

/* qmc_ising.rs:780-784 / qmc_runner.rs:241-255: variables without any op get a fresh fair bit. */
void ora_flip_free_spins(ora_replica *r) {
    const ora_model *m = r->m;
    uint8_t *touched = r->touched;
    memset(touched, 0, m->nvars);
    for (uint32_t p = 0; p < r->cutoff; ++p) {
        uint32_t w = r->ops[p];
        if (w == SSE_OP_EMPTY) continue;
        uint32_t b = sse_op_bond(w);
        touched[m->bond_a[b]] = 1;
        if (m->bond_b[b] != SSE_NO_VAR) touched[m->bond_b[b]] = 1;
    }
    for (uint32_t v = 0; v < m->nvars; ++v)
        if (!touched[v]) {
            uint32_t o[4];
            draw(r, SSE_TAG_FREE, v, o);
            r->state[v] = (uint8_t)(o[0] >> 31);
        }
    r->epoch += 1;
}

/* -------------------------------------------------------------------- directed loop --- */
/* qmc_traits/directed_loop.rs:103-171 (start selection) and :217-301 (loop body).
 * Legs are numbered inputs 0..k-1 then outputs 0..k-1 (:233-240). */
static int next_on_worldline(const ora_replica *r, uint32_t p, uint32_t var, int forward, uint32_t *np,
                             uint32_t *nrel, int *wrapped) {
    const ora_model *m = r->m;
    const uint32_t M = r->cutoff;
    *wrapped = 0;
    uint32_t q = p;
    for (uint32_t step = 0; step < M; ++step) {
        if (forward) { q += 1; if (q == M) { q = 0; *wrapped = 1; } }
        else { if (q == 0) { q = M; *wrapped = 1; } q -= 1; }
        uint32_t w = r->ops[q];
        if (w == SSE_OP_EMPTY) continue;
        uint32_t b = sse_op_bond(w);
        if (m->bond_a[b] == var) { *np = q; *nrel = 0; return 1; }
        if (m->bond_b[b] == var) { *np = q; *nrel = 1; return 1; }
    }
    return 0;
}

uint32_t ora_loop_update(ora_replica *r) {
    const ora_model *m = r->m;
    const uint32_t M = r->cutoff;
    uint32_t visited = 0;
    if (r->n == 0) { r->epoch += 1; return 0; }
    uint32_t o[4];
    draw(r, SSE_TAG_LOOP, 0, o);
    uint32_t nth = mulhi32(o[0], r->n);
    uint32_t p0 = 0;
    for (uint32_t p = 0, c = 0; p < M; ++p)
        if (r->ops[p] != SSE_OP_EMPTY) { if (c == nth) { p0 = p; break; } c++; }
    uint32_t b0 = sse_op_bond(r->ops[p0]);
    uint32_t k0 = (m->bond_b[b0] != SSE_NO_VAR) ? 2u : 1u;
    uint32_t rel0 = mulhi32(o[1], k0);
    uint32_t side0 = (o[2] >> 31) ? 0u : 1u; /* gen() true -> Inputs(0) else Outputs(1) (:153-157) */
    uint32_t p = p0, rel = rel0, side = side0;
    for (uint32_t step = 1;; ++step) {
        uint32_t w = r->ops[p], b = sse_op_bond(w);
        uint32_t k = (m->bond_b[b] != SSE_NO_VAR) ? 2u : 1u;
        uint32_t in = sse_op_in(w), out = sse_op_out(w);
        /* toggle entrance */
        uint32_t in_e = in, out_e = out;
        if (side == 0) in_e ^= 1u << rel; else out_e ^= 1u << rel;
        double wl[4], total = 0.0;
        for (uint32_t leg = 0; leg < 2 * k; ++leg) {
            uint32_t i2 = in_e, o2 = out_e;
            if (leg < k) i2 ^= 1u << leg; else o2 ^= 1u << (leg - k);
            wl[leg] = ora_bond_weight(m, b, i2, o2);
            total += wl[leg];
        }
        draw(r, SSE_TAG_LOOP, step, o);
        double c = u01(o[0]) * total;
        uint32_t exit_leg = 2 * k - 1;
        for (uint32_t leg = 0; leg < 2 * k; ++leg) {
            if (c < wl[leg]) { exit_leg = leg; break; }
            c -= wl[leg];
        }
        uint32_t xside = exit_leg < k ? 0u : 1u, xrel = exit_leg < k ? exit_leg : exit_leg - k;
        if (xside == 0) in_e ^= 1u << xrel; else out_e ^= 1u << xrel;
        r->ops[p] = sse_op_make(b, in_e, out_e);
        visited++;
        if (p == p0 && xrel == rel0 && xside == side0) break; /* :266 */
        uint32_t var = xrel == 0 ? m->bond_a[b] : m->bond_b[b];
        uint32_t np, nrel;
        int wrapped;
        /* the op itself is the only one on the worldline => next is itself after a full wrap */
        if (!next_on_worldline(r, p, var, xside == 1, &np, &nrel, &wrapped)) {
            np = p; nrel = xrel; wrapped = 1;
        }
        if (wrapped) /* :276-288 */
            r->state[var] = (uint8_t)(((xside == 1 ? out_e : in_e) >> xrel) & 1u);
        uint32_t nside = xside ^ 1u;
        if (np == p0 && nrel == rel0 && nside == side0) break; /* :293 */
        p = np; rel = nrel; side = nside;
        if (step > (1u << 30)) break;
    }
    r->acc[4] += visited;
    r->epoch += 1;
    return visited;
}

/* ---------------------------------------------------------------------------- driver --- */
static void measure(ora_replica *r) {
    const ora_model *m = r->m;
    int64_t up = 0;
    for (uint32_t v = 0; v < m->nvars; ++v) up += r->state[v];
    int64_t mag = 2 * up - (int64_t)m->nvars;
    uint64_t ntrans = 0; /* transverse-bond ops: <sigma_x> = <ntrans>/(beta*Gamma*N) - 1 */
    for (uint32_t p = 0; p < r->cutoff; ++p)
        if (r->ops[p] != SSE_OP_EMPTY &&
            (m->binfo[sse_op_bond(r->ops[p])] & SSE_BOND_KIND_MASK) == SSE_BOND_TRANSVERSE)
            ntrans++;
    r->acc[0] += r->n;
    r->acc[1] += 1;
    r->acc[2] += (uint64_t)(mag < 0 ? -mag : mag);
    r->acc[3] += (uint64_t)(mag * mag);
    r->acc[6] += ntrans;
}

/* QmcIsingGraph::timestep (qmc_ising.rs:644-795) and Qmc::timestep (qmc_runner.rs:363-377):
 * diagonal -> [loop] -> cluster -> free spins; cutoff = max(cutoff, n + n/2) (qmc_ising.rs:786,
 * qmc_runner.rs:197). Returns nonzero if the cutoff would exceed the capacity. */
int ora_timestep(ora_replica *r, double beta, uint32_t flags) {
    if (flags & ORA_FLAG_HEATBATH) ora_heatbath_update(r, beta); else ora_diagonal_update(r, beta);
    uint32_t want = r->n + r->n / 2;
    if (want > r->cutoff) {
        if (want > r->cap) return 1;
        r->cutoff = want;
    }
    if (flags & ORA_FLAG_RVB) ora_rvb_update(r, (r->m->nvars + 1) / 2); /* qmc_ising.rs:705-752 */
    if (flags & ORA_FLAG_LOOP) ora_loop_update(r);
    if (!(flags & ORA_FLAG_NO_CLUSTER)) ora_cluster_update(r, 0.5);
    ora_flip_free_spins(r);
    return 0;
}

/* qmc_traits/qmc_stepper.rs:133-162: sample when (t+1) % freq == 0. */
int ora_timesteps(ora_replica *r, uint64_t t, double beta, uint32_t sampling_freq, uint32_t flags) {
    if (sampling_freq == 0) sampling_freq = 1;
    for (uint64_t i = 0; i < t; ++i) {
        int rc = ora_timestep(r, beta, flags);
        if (rc) return rc;
        if ((i + 1) % sampling_freq == 0) measure(r);
    }
    return 0;
}

/* Verify: qmc_ising.rs:829-860 (non-zero weights) + op_container.rs:137-159 (propagation, periodicity) */
int ora_verify(const ora_replica *r) {
    const ora_model *m = r->m;
    uint8_t *s = (uint8_t *)malloc(m->nvars);
    memcpy(s, r->state, m->nvars);
    int ok = 1;
    uint32_t count = 0;
    for (uint32_t p = 0; p < r->cutoff && ok; ++p) {
        uint32_t w = r->ops[p];
        if (w == SSE_OP_EMPTY) continue;
        count++;
        uint32_t b = sse_op_bond(w);
        if (b >= m->nbonds) { ok = 0; break; }
        uint32_t in = sse_op_in(w), out = sse_op_out(w);
        if (!(fabs(ora_bond_weight(m, b, in, out)) > DBL_EPSILON)) ok = 0;
        uint32_t a = m->bond_a[b], c = m->bond_b[b];
        if (s[a] != (in & 1u)) ok = 0;
        s[a] = (uint8_t)(out & 1u);
        if (c != SSE_NO_VAR) {
            if (s[c] != ((in >> 1) & 1u)) ok = 0;
            s[c] = (uint8_t)((out >> 1) & 1u);
        } else if ((in | out) & 2u) ok = 0;
    }
    if (ok && memcmp(s, r->state, m->nvars) != 0) ok = 0;
    if (ok && count != r->n) ok = 0;
    free(s);
    return ok;
}

/* Imaginary-time fold of the magnetisation (OpContainer::itime_fold, fast_ops.rs:1296-1315, with the closure
 * acc + f(m(state)) for f = m, m^2, |m|; m = sum_v (2 s_v - 1)): the fold visits the propagated state BEFORE the op
 * of every slot p = 0..cutoff-1. */
void ora_itime_magnetization(const ora_replica *r, int64_t *sum_m, uint64_t *sum_m2, uint64_t *sum_abs) {
    const ora_model *m = r->m;
    int64_t mag = 0;
    for (uint32_t v = 0; v < m->nvars; ++v) mag += r->state[v] ? 1 : -1;
    int64_t s1 = 0;
    uint64_t s2 = 0, sa = 0;
    for (uint32_t p = 0; p < r->cutoff; ++p) {
        s1 += mag; s2 += (uint64_t)(mag * mag); sa += (uint64_t)(mag < 0 ? -mag : mag);
        const uint32_t w = r->ops[p];
        if (w == SSE_OP_EMPTY) continue;
        const uint32_t in = sse_op_in(w), out = sse_op_out(w);
        mag += 2 * ((int64_t)(out & 1u) - (int64_t)(in & 1u));
        mag += 2 * ((int64_t)((out >> 1) & 1u) - (int64_t)((in >> 1) & 1u));
    }
    *sum_m = s1; *sum_m2 = s2; *sum_abs = sa;
}

/* ------------------------------------------------------------------------- accessors --- */
uint32_t ora_get_n(const ora_replica *r) { return r->n; }
uint32_t ora_get_cutoff(const ora_replica *r) { return r->cutoff; }
int ora_set_cutoff(ora_replica *r, uint32_t cutoff) {
    /* fast_ops.rs:1258-1262: the cutoff only ever grows */
    if (cutoff > r->cap) return 1;
    if (cutoff > r->cutoff) r->cutoff = cutoff;
    return 0;
}
uint64_t ora_get_epoch(const ora_replica *r) { return r->epoch; }
void ora_set_epoch(ora_replica *r, uint64_t epoch) { r->epoch = epoch; } /* tests: a configuration moved into another replica object */
void ora_get_state(const ora_replica *r, uint8_t *out) { memcpy(out, r->state, r->m->nvars); }
void ora_set_state(ora_replica *r, const uint8_t *in) {
    for (uint32_t v = 0; v < r->m->nvars; ++v) r->state[v] = in[v] ? 1 : 0;
}
void ora_get_ops(const ora_replica *r, uint32_t *out) { memcpy(out, r->ops, sizeof(uint32_t) * r->cutoff); }
int ora_set_ops(ora_replica *r, const uint32_t *words, uint32_t cutoff) {
    if (cutoff > r->cap) return 1;
    memset(r->ops, 0, sizeof(uint32_t) * r->cap);
    memcpy(r->ops, words, sizeof(uint32_t) * cutoff);
    if (cutoff > r->cutoff) r->cutoff = cutoff;
    uint32_t n = 0;
    for (uint32_t p = 0; p < r->cutoff; ++p) n += r->ops[p] != SSE_OP_EMPTY;
    r->n = n;
    return 0;
}
uint32_t ora_get_bond_count(const ora_replica *r, uint32_t bond) {
    uint32_t c = 0;
    for (uint32_t p = 0; p < r->cutoff; ++p)
        if (r->ops[p] != SSE_OP_EMPTY && sse_op_bond(r->ops[p]) == bond) c++;
    return c;
}
void ora_get_accumulators(const ora_replica *r, uint64_t acc[8]) { memcpy(acc, r->acc, sizeof(r->acc)); }
void ora_reset_accumulators(ora_replica *r) { memset(r->acc, 0, sizeof(r->acc)); }

/* ------------------------------------------------------------------ parallel tempering --- */
/* TemperingContainer::tempering_step (src/sse/parallel_tempering/tempering_container.rs:121-149) with
 * perform_swaps (:241-272) and swap_on_chunks (:274-302), for ONE chain of ntemps graphs that share a
 * Hamiltonian (rel_h_weight = 1, :285-287).  by_slot[t] is the replica currently at temperature t;
 * swapping graphs (qmc_ising.rs:593-602 swaps manager and state) = swapping the pointers.
 *   1. every graph's cutoff is raised to the chain maximum (:129-137);
 *   2. one coin decides whether pair set a = (0,1),(2,3).. or b = (1,2),(3,4).. goes first (:140-146);
 *   3. each pair draws one uniform u and swaps iff (beta_a/beta_b)^(n_b - n_a) > u (:255, :296-298).
 * Philox: tag PT, replica field = chain, epoch = step; index 0 = the order coin, index 1+t = the pair whose
 * lower slot is t.  Returns the number of swaps. */
/* f64::powi (tempering_container.rs:296) as Rust lowers it (compiler-rt __powidf2): repeated squaring, reciprocal for a
 * negative exponent */
static double powi_signed(double x, int64_t n) {
    uint64_t m = n < 0 ? (uint64_t)(-n) : (uint64_t)n;
    double r = 1.0;
    while (m) { if (m & 1u) r *= x; x *= x; m >>= 1; }
    return n < 0 ? 1.0 / r : r;
}
uint64_t ora_pt_step(ora_replica **by_slot, const double *betas, uint32_t ntemps, uint64_t seed, uint32_t chain,
                     uint64_t step) {
    if (ntemps <= 1) return 0;
    uint32_t maxcut = 0;
    for (uint32_t t = 0; t < ntemps; ++t) if (by_slot[t]->cutoff > maxcut) maxcut = by_slot[t]->cutoff;
    for (uint32_t t = 0; t < ntemps; ++t) ora_set_cutoff(by_slot[t], maxcut);
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {0, (uint32_t)step, chain, (SSE_TAG_PT << 24) | (uint32_t)((step >> 32) & 0xFFFFFFu)};
    uint32_t o[4];
    ora_philox4x32_10(ctr, key, o);
    const int a_first = (o[0] >> 31) != 0; /* gen_bool(0.5) */
    uint64_t swaps = 0;
    for (int phase = 0; phase < 2; ++phase) {
        const int set_a = (phase == 0) ? a_first : !a_first;
        /* make_first_subgraphs / make_second_subgraphs (:83-99) */
        for (uint32_t t = set_a ? 0u : 1u; t + 1 < ntemps; t += 2) {
            ctr[0] = 1u + t;
            ora_philox4x32_10(ctr, key, o);
            const double u = u01(o[0]);
            ora_replica *ga = by_slot[t], *gb = by_slot[t + 1];
            const double p_swap = powi_signed(betas[t] / betas[t + 1], (int64_t)gb->n - (int64_t)ga->n);
            if (p_swap > u) {
                by_slot[t] = gb;
                by_slot[t + 1] = ga;
                swaps++;
            }
        }
    }
    return swaps;
}
