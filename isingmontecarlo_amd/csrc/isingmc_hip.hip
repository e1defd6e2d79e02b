// isingmc_hip.hip — C ABI (include/isingmc_hip.h) over the gfx950 kernels of sse_device.hip.h.
// Host side only sequences launches and moves small control arrays; there is no CPU compute fallback.
#include "../../include/isingmc_hip.h"
#include "sse_device.hip.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace sse;

static thread_local std::string g_create_error;

struct isingmc_batch {
    DevBatch dev{};
    uint32_t W = 8, K = 4, mode = SSE_MODE_GENERAL; // mode: SSE_MODE_* (bond decode / where the per-variable tables live)
    uint32_t W_off = 0;                 // waves per replica of the off-diagonal launches; 0 = decide per launch (16 when its tables fit in LDS)
    uint32_t last_W_off = 0;
    uint64_t steps_per_launch = 0;
    uint32_t acc_rows = 0;
    uint32_t rvb_updates = 0;
    bool w8_ok = false;                 // an 8-wave off-diagonal geometry without LDS union-find fits (and the row stride allows it)
    uint32_t *d_acc_row = nullptr;
    size_t lds_bytes = 0, lds_bytes_rvb = 0, lds_fixed_words_ = 0, lds_total_words = 0;
    uint32_t max_ntrans = 0, uf_ids_limit = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    uint32_t last_launches = 0;
    bool fast_diag = false;             // the diagonal-pass launch uses sse_fast.hip.h (headline geometry: LDS edge tables, 4 waves, N <= 4096)
    size_t lds_bytes_fast = 0, lds_bytes_fast_label = 0;
    bool compact = false;               // ... and it writes the dense op list for the cluster update that follows in the same timestep
    bool lite = false;                  // ... or (experimental) labels the segments for the cluster update that follows in the same timestep
    bool fused_launch = false;          // ISINGMC_CFG_FUSED_LAUNCH: whole timesteps in one kernel (no diagonal-only launches)
    size_t lds_bytes_pm_diag = 0;       // +-J decode: LDS of the diagonal launch with its per-wave spin bytes in LDS (0 = they do not fit: mode 4 there too)
    bool lean_cluster = false;          // cluster (+ free spins + sampling) launches use sse_cluster.hip.h when their ids fit its LDS union-find
    bool last_lean = false;             // ... and the last such launch did
    bool defer = false;                 // ... leaving its flips as one byte per slot for the next (trimmed) diagonal launch to apply
    bool pending = false;               // some replicas' strings in HBM may still wait for their flip bytes (DevBatch::pend says which)
    const double *beta_dev = nullptr;   // isingmc_pt_timesteps: per-replica betas already on the device (used when the caller passes none)
    bool rvb_split = false;             // RVB sweeps run as a growth launch + a main launch (sse_rvb_split.hip.h) instead of the fused kernel
    uint32_t rvb_main_W = 4;            // waves per replica of that main launch
    bool last_rvb_split = false;        // ... and the last RVB sweep did
    std::vector<hipEvent_t> evpool;     // per-launch events of the split path (bounded, see run())
    float pass_ms[3] = {0.f, 0.f, 0.f}; // [0] diagonal-only launches, [1] all other launches of the last run, [2] of those: the RVB-sweep launches
    uint32_t pass_launches[3] = {0, 0, 0};
    double offset = 0.0;
    std::vector<double> offsets;        // per-replica energy offsets (ISINGMC_CFG_PER_REPLICA_J), else empty
    bool per_replica_J = false;
    bool generic = false;               // built from isingmc_interaction matrices
    bool generic_sym = false;           // ... all of them symmetric under a global spin flip (cluster updates allowed)
    std::vector<double> mats_host;      // [Nb][16] in | out<<2
    std::vector<BondRec> bonds_host;
    double *d_beta = nullptr;
    uint32_t *d_out = nullptr;
    uint32_t *d_vstate = nullptr;
    uint8_t *d_ok = nullptr;
    std::vector<void *> allocs;
    mutable std::string err;
    struct PtState *pt = nullptr;       // native parallel tempering (isingmc_pt_*), see the end of this file
    std::vector<uint32_t> ham_row_host; // [R] bond-table row of each local replica (tempering between different Hamiltonians), empty = identity
};

#define HIP_TRY(b, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (b)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                             \
            return ISINGMC_ENODEVICE;                                                                 \
        }                                                                                             \
    } while (0)

template <typename T>
static int dalloc(isingmc_batch *b, T **p, size_t count, bool zero = true) {
    void *q = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIP_TRY(b, hipMalloc(&q, bytes));
    b->allocs.push_back(q);
    if (zero) HIP_TRY(b, hipMemset(q, 0, bytes));
    *p = reinterpret_cast<T *>(q);
    return ISINGMC_OK;
}

__global__ void init_state_kernel(DevBatch B) {
    // classical/graph.rs:451-453 make_random_spin_state: one fair bit per variable (Philox tag INIT, epoch 0)
    const uint32_t r = blockIdx.x;
    const Rng rng = make_rng(B, r, 0ull);
    for (uint32_t i = threadIdx.x; i < B.nwords; i += blockDim.x) {
        uint32_t s = 0;
        for (uint32_t j = 0; j < 32 && i * 32 + j < B.N; ++j) s |= (rng.draw(SSE_TAG_INIT, i * 32 + j).x >> 31) << j;
        B.state[(size_t)r * B.nwords + i] = s;
    }
}


// Verify::verify (qmc_ising.rs:829-860; op_container.rs:137-159), one thread per replica (debug API, not on
// the hot path).  ok[r] = 1 iff every op has non-zero weight, the propagated state matches every op's
// inputs, periodicity holds, no op sits beyond the cutoff and the counters n / ntrans match the op-string.
__global__ void verify_kernel(DevBatch B, uint32_t *scratch_state /*[R][nwords]*/, uint8_t *ok) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.R) return;
    uint32_t *s = scratch_state + (size_t)r * B.nwords;
    const uint32_t *s0 = B.state + (size_t)r * B.nwords;
    for (uint32_t i = 0; i < B.nwords; ++i) s[i] = s0[i];
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    const uint32_t M = B.cutoff[r];
    bool good = true;
    uint32_t count = 0, ntr = 0, ccn = 0, cctr = 0;
    const uint32_t *chunks = B.chunks + (size_t)r * 2 * SSE_MAX_CHUNKS;
    for (uint32_t p = 0; p < B.cap; ++p) {
        if (p % B.CH == 0 && p) { // per-chunk counters kept by the diagonal pass must match the op-string
            const uint32_t c = p / B.CH - 1;
            if (chunks[c] != ccn || chunks[SSE_MAX_CHUNKS + c] != cctr) good = false;
            ccn = 0; cctr = 0;
        }
        const uint32_t w = ops[p];
        if (!w) continue;
        ccn++;
        if (p >= M) { good = false; break; }
        count++;
        const uint32_t b = sse_op_bond(w);
        if (b >= B.Nb) { good = false; break; }
        const BondRec rec = B.bonds[(size_t)(B.ham_row ? B.ham_row[r] : r) * B.bond_stride + b];
        Bd d;
        d.a = rec.a_info & SSE_VAR_MASK; d.c = rec.c; d.kp = rec.a_info >> SSE_INFO_SHIFT; d.w = rec.w;
        const uint32_t in = sse_op_in(w), out = sse_op_out(w);
        if (!(op_weight(B, b, d, in, out) > 2.220446049250313e-16)) good = false;
        if (bd_kind(d) == SSE_BOND_TRANSVERSE) { ntr++; cctr++; }
        const uint32_t a = d.a, c = d.c;
        if (((s[a >> 5] >> (a & 31)) & 1u) != (in & 1u)) good = false;
        s[a >> 5] = (s[a >> 5] & ~(1u << (a & 31))) | ((out & 1u) << (a & 31));
        if (c != SSE_NO_VAR) {
            if (((s[c >> 5] >> (c & 31)) & 1u) != ((in >> 1) & 1u)) good = false;
            s[c >> 5] = (s[c >> 5] & ~(1u << (c & 31))) | (((out >> 1) & 1u) << (c & 31));
        } else if ((in | out) & 2u) good = false;
    }
    if (good) { // last chunk
        const uint32_t c = (B.cap - 1) / B.CH;
        if (chunks[c] != ccn || chunks[SSE_MAX_CHUNKS + c] != cctr) good = false;
    }
    for (uint32_t i = 0; i < B.nwords; ++i) if (s[i] != s0[i]) good = false;
    if (count != B.n[r] || ntr != B.ntrans[r]) good = false;
    ok[r] = good ? 1 : 0;
}

// OpContainer::itime_fold (fast_ops.rs:1296-1315) for the magnetisation: sums over p = 0..cutoff-1 of m, m^2, |m| of
// the propagated state BEFORE slot p's op, m = sum_v (2 s_v - 1).  m only moves at off-diagonal ops (+-2 per flipped
// spin), so every thread takes a contiguous block of slots: block deltas -> exclusive prefix over the workgroup ->
// each thread replays its block from its starting m.  One workgroup of 256 threads per replica.
__global__ __launch_bounds__(256) void itime_magnetization_kernel(DevBatch B, long long *sum_m, unsigned long long *sum_m2,
                                                                  unsigned long long *sum_abs) {
    __shared__ long long sh[256];
    __shared__ long long red[3][4];
    const uint32_t r = blockIdx.x, tid = threadIdx.x;
    const uint32_t M = B.cutoff[r];
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    long long m0 = 0;
    for (uint32_t i = tid; i < B.nwords; i += 256) m0 += 2ll * __popc(B.state[(size_t)r * B.nwords + i]);
    for (int off = 32; off > 0; off >>= 1) m0 += __shfl_down(m0, off);
    if ((tid & 63) == 0) red[0][tid >> 6] = m0;
    const uint32_t per = (M + 255u) / 256u, p_lo = min(tid * per, M), p_hi = min(p_lo + per, M);
    auto delta = [](uint32_t w) -> long long {
        const uint32_t x = sse_op_in(w), y = sse_op_out(w);
        return 2ll * ((long long)(y & 1u) - (long long)(x & 1u) + (long long)((y >> 1) & 1u) - (long long)((x >> 1) & 1u));
    };
    long long d = 0;
    for (uint32_t p = p_lo; p < p_hi; ++p) d += delta(ops[p]);
    sh[tid] = d;
    __syncthreads();
    long long m = red[0][0] + red[0][1] + red[0][2] + red[0][3] - (long long)B.N; // popcount bits outside N are zero
    for (uint32_t t = 0; t < tid; ++t) m += sh[t];
    long long s1 = 0;
    unsigned long long s2 = 0, sa = 0;
    for (uint32_t p = p_lo; p < p_hi; ++p) {
        s1 += m; s2 += (unsigned long long)(m * m); sa += (unsigned long long)(m < 0 ? -m : m);
        m += delta(ops[p]);
    }
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); sa += __shfl_down(sa, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = (long long)s2; red[2][tid >> 6] = (long long)sa; }
    __syncthreads();
    if (tid == 0) {
        sum_m[r] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        sum_m2[r] = (unsigned long long)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        sum_abs[r] = (unsigned long long)(red[2][0] + red[2][1] + red[2][2] + red[2][3]);
    }
}

// DebugOps::count_diagonal_and_off / count_constant_ops (qmc_debug.rs:10-41): one workgroup per replica, out[r] = {diagonal,
// off-diagonal, constant} ops among the slots below the cutoff
__global__ __launch_bounds__(256) void debug_counts_kernel(DevBatch B, uint32_t *out) {
    __shared__ uint32_t red[3];
    const uint32_t r = blockIdx.x;
    if (threadIdx.x < 3) red[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    const uint32_t M = B.cutoff[r];
    const BondRec *bonds = B.bonds + (size_t)(B.bond_stride ? (B.ham_row ? B.ham_row[r] : r) : 0u) * B.bond_stride;
    uint32_t d = 0, o = 0, c = 0;
    for (uint32_t p = threadIdx.x; p < M; p += blockDim.x) {
        const uint32_t w = ops[p];
        if (!w) continue;
        if (sse_op_is_diagonal(w)) d++; else o++;
        if (((bonds[sse_op_bond(w)].a_info >> SSE_INFO_SHIFT) & SSE_BOND_KIND_MASK) == SSE_BOND_TRANSVERSE) c++; // BasicOp::constant
    }
    atomicAdd(&red[0], d); atomicAdd(&red[1], o); atomicAdd(&red[2], c);
    __syncthreads();
    if (threadIdx.x < 3) out[3 * r + threadIdx.x] = red[threadIdx.x];
}

// Deferred cluster flips (sse_cluster.hip.h) applied in place: ops[p] ^= flip byte, for the replicas whose flag is set.  Used by
// every consumer of the op-strings other than the trimmed diagonal kernel, which applies the bytes itself while it streams.
__global__ __launch_bounds__(1024) void materialize_kernel(DevBatch B) {
    const uint32_t r = blockIdx.x;
    if (!B.pend[r]) return; // (uniform per workgroup)
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    const uint8_t *fb = B.flipb + (size_t)r * B.stride;
    const uint32_t M = B.cutoff[r];
    for (uint32_t p = threadIdx.x; p < M; p += blockDim.x) { const uint32_t f = fb[p]; if (f) ops[p] ^= f; }
    __syncthreads();
    if (threadIdx.x == 0) B.pend[r] = 0u;
}
static int ensure_materialized(isingmc_batch *b) {
    if (!b->pending) return ISINGMC_OK;
    hipLaunchKernelGGL(materialize_kernel, dim3(b->dev.R), dim3(1024), 0, b->stream, b->dev);
    HIP_TRY(b, hipGetLastError());
    HIP_TRY(b, hipStreamSynchronize(b->stream)); // (callers read the strings with blocking copies or their own kernels on this stream; keep it simple)
    b->pending = false;
    return ISINGMC_OK;
}

static size_t lds_fixed_words(uint32_t W, uint32_t N, uint32_t nwords, uint32_t ledges, bool tg = false, uint32_t pm_words = 0) {
    // mirrors Lds<W>::carve up to and including o_cl: state, touched bits, touched bytes, round buffers, misc, chunk counters,
    // edge table, per-wave rank tables (u16) and marker tables (u8); with the tables in HBM (tg) only the bit arrays remain
    // (+ the coupling signs of the +-J decode)
    if (tg) return (size_t)nwords * 2 + 4 * W + 16 + 2 * SSE_MAX_CHUNKS + ledges + pm_words;
    return (size_t)nwords * 2 + ((size_t)N + 3) / 4 + 4 * W + 16 + 2 * SSE_MAX_CHUNKS + ledges + ((size_t)W * N + 1) / 2 + ((size_t)W * N + 3) / 4;
}
// dynamic LDS of the fast diagonal-pass launch (mirrors Lds<4>::carve up to o_cur, then FastLds: sse_fast.hip.h fast_carve)
static size_t fast_lds_bytes(uint32_t N, uint32_t nwords, uint32_t E, uint32_t Nb, bool label) {
    const size_t o_edges = (size_t)nwords * 2 + ((size_t)N + 3) / 4 + 4 * 4 + 16 + 2 * SSE_MAX_CHUNKS;
    size_t words = ((o_edges + 3) & ~(size_t)3) + 16 + Nb + N + 64 + (label ? 4 * (size_t)N + ((size_t)N + 3) / 4 : 0);
    if (words < o_edges + E) words = o_edges + E; // the directed loop behind the pass stages the compact edge table there
    return (4 * words + 7) & ~(size_t)7;
}
static bool is_tg(const isingmc_batch *b) { return b->mode == SSE_MODE_GLOBAL_TABLES || b->mode == SSE_MODE_PM_GLOBAL_TABLES; }
static bool is_pm(const isingmc_batch *b) { return b->mode == SSE_MODE_PM_GLOBAL_TABLES; }
// dynamic LDS of the diagonal-pass launch: the fixed regions up to the per-wave tables, which it uses as [W][N] spin bytes
static size_t diag_lds_bytes(const isingmc_batch *b) {
    const size_t words = is_tg(b) ? b->lds_fixed_words_ : b->lds_fixed_words_ - ((size_t)b->W * b->dev.N + 1) / 2;
    return (4 * words + 7) & ~(size_t)7;
}

// LDS footprint of the next launch.  The union-find of the cluster pass lives in LDS as 16-bit parents when all
// ids fit; its capacity follows the largest transverse-op count seen so far (+ headroom), so that the footprint
// stays small enough for two workgroups per CU whenever the model allows it.  Replicas that outgrow it use the HBM
// union-find for that sweep and the host enlarges the table before the next launch.
struct LdsPlan { uint32_t W, ufcap; size_t lds_bytes; bool all_ids_fit; };
static LdsPlan plan_lds(const isingmc_batch *b, uint32_t W) {
    const DevBatch &D = b->dev;
    const size_t fixed = lds_fixed_words(W, D.N, D.nwords, b->mode == SSE_MODE_LDS_EDGES ? D.E : 0u, is_tg(b), is_pm(b) ? D.pm_words : 0u);
    const size_t ids_max = (size_t)W * D.N + D.cap;
    const size_t want = (size_t)W * D.N + b->max_ntrans + b->max_ntrans / 16 + 384;
    size_t ids = want;
    if (b->uf_ids_limit) ids = b->uf_ids_limit;
    if (is_tg(b)) ids = 0; // tables in HBM: the union-find lives there too
    if (ids > 65535) ids = 65535;
    if (ids > ids_max) ids = ids_max;
    auto words = [&](size_t n) { return (n + 1) / 2 + (D.has_long ? 2 * ((n + 31) / 32) : 0); };
    while (ids > 0 && fixed + words(ids) > b->lds_total_words) ids -= (ids > 64 ? 64 : ids);
    LdsPlan p;
    p.W = W; p.ufcap = (uint32_t)ids;
    p.lds_bytes = (4 * (fixed + words(ids)) + 7) & ~(size_t)7;
    p.all_ids_fit = !is_tg(b) && fixed + 64 <= b->lds_total_words && ids >= (want < ids_max ? want : ids_max) && !b->uf_ids_limit;
    return p;
}
static void size_lds(isingmc_batch *b) {
    const LdsPlan p = plan_lds(b, b->W);
    b->dev.lds_ufcap = p.ufcap;
    b->lds_bytes = p.lds_bytes;
}

// LDS plan of the dedicated cluster kernel (sse_cluster.hip.h): 16 waves, packed per-wave tables, 16-bit parents for
// 16 N + (transverse ops seen so far + headroom) ids.  ok = false: the ids do not fit (the general kernel takes the launch).
struct LeanPlan { bool ok; uint32_t ufcap; size_t lds_bytes; };
static LeanPlan plan_lean(const isingmc_batch *b) {
    const DevBatch &D = b->dev;
    LeanPlan p{false, 0u, 0};
    if (!b->lean_cluster) return p;
    const size_t fixed = cluster_fixed_words(D.N, D.nwords, D.Nb);
    const size_t ids_max = (size_t)16 * D.N + D.cap;
    size_t want = (size_t)16 * D.N + b->max_ntrans + b->max_ntrans / 16 + 384;
    if (want > ids_max) want = ids_max;
    if (want > 65535) return p;
    const size_t words = fixed + (want + 1) / 2 + (D.has_long ? 2 * ((want + 31) / 32) : 0);
    if (words > b->lds_total_words) return p;
    p.ok = true; p.ufcap = (uint32_t)want; p.lds_bytes = (4 * words + 7) & ~(size_t)7;
    return p;
}

static int check_errors(isingmc_batch *b) {
    std::vector<uint32_t> err(b->dev.R), ntr(b->dev.R);
    HIP_TRY(b, hipMemcpyAsync(err.data(), b->dev.err, sizeof(uint32_t) * b->dev.R, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(b, hipMemcpyAsync(ntr.data(), b->dev.ntrans, sizeof(uint32_t) * b->dev.R, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    for (uint32_t r = 0; r < b->dev.R; ++r) if (ntr[r] > b->max_ntrans) b->max_ntrans = ntr[r];
    for (uint32_t r = 0; r < b->dev.R; ++r)
        if (err[r]) {
            char buf[160];
            if (err[r] == 1u) {
                snprintf(buf, sizeof buf, "replica %u: cutoff n + n/2 exceeds the op-string capacity %u", r, b->dev.cap);
                b->err = buf;
                return ISINGMC_ECAPACITY;
            }
            if (err[r] == 8u) {
                snprintf(buf, sizeof buf, "replica %u: more than 65534 transverse ops inside one wave's range of the cluster scan; raise waves_per_replica", r);
                b->err = buf;
                return ISINGMC_ECAPACITY;
            }
            if (err[r] == 6u || err[r] == 7u || err[r] == 5u) {
                snprintf(buf, sizeof buf, "replica %u: RVB working set exceeds the LDS scratch (code %u)", r, err[r]);
                b->err = buf;
                return ISINGMC_ECAPACITY;
            }
            if (err[r] == 3u) {
                snprintf(buf, sizeof buf, "replica %u: directed loop still open after 64*cutoff+1024 vertices (the reference has no bound; "
                                          "clear with isingmc_clear_errors and continue)", r);
                b->err = buf;
                return ISINGMC_ELIMIT;
            }
            snprintf(buf, sizeof buf, "replica %u: device integrity error %u", r, err[r]);
            b->err = buf;
            return ISINGMC_EINTEGRITY;
        }
    return ISINGMC_OK;
}

static int run(isingmc_batch *b, const double *beta, uint64_t nsteps, uint32_t freq, uint32_t domask, double prob,
               uint32_t *out_host) {
    if (!b) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    SweepArgs A{};
    A.beta = nullptr;
    if (beta) {
        for (uint32_t r = 0; r < b->dev.R; ++r)
            if (!(beta[r] >= 0.0) || !std::isfinite(beta[r])) { b->err = "beta must be finite and >= 0"; return ISINGMC_EINVAL; }
        HIP_TRY(b, hipMemcpyAsync(b->d_beta, beta, sizeof(double) * b->dev.R, hipMemcpyHostToDevice, b->stream));
        A.beta = b->d_beta;
    } else if (b->beta_dev) {
        A.beta = b->beta_dev;
    } else if (domask & SSE_DO_DIAG) {
        b->err = "beta is required for a diagonal update";
        return ISINGMC_EINVAL;
    }
    if ((domask & SSE_DO_RVB) && b->generic) { b->err = "RVB updates are Ising-specific: not available with generic interactions"; return ISINGMC_ENOTIMPL; }
    if ((domask & SSE_DO_CLUSTER) && b->generic && !b->generic_sym) { b->err = "Cannot perform cluster updates on graphs that break ising symmetry."; return ISINGMC_ENOTIMPL; } // qmc_runner.rs:224-226
    if ((domask & SSE_DO_RVB) && is_tg(b)) { b->err = "RVB updates keep their working set in LDS: not available for models whose per-variable tables live in HBM"; return ISINGMC_ENOTIMPL; }
    // Pending cluster flips: only a call whose first launch is the trimmed diagonal kernel may start on the un-flipped strings
    {
        const bool first_is_fast_diag = !b->fused_launch && (domask & SSE_DO_DIAG) && b->fast_diag && !(domask & SSE_DO_HEATBATH) && b->defer;
        if (b->pending && !first_is_fast_diag) { const int rcm = ensure_materialized(b); if (rcm) return rcm; }
    }
    A.sampling_freq = freq;
    A.domask = domask & 0xFFFFu;
    A.prob = prob;
    A.out_u32 = out_host ? b->d_out : nullptr;
    A.rvb_updates = b->rvb_updates;
    LaunchCfg lc{};
    lc.W = b->W; lc.K = b->K; lc.mode = b->mode; lc.phase = (domask >> 16) & 1u; lc.stream = b->stream;
    size_lds(b);
    if (!b->dev.segs2 && (domask & SSE_DO_CLUSTER) && !plan_lds(b, b->W_off ? b->W_off : b->W).all_ids_fit) {
        // the cluster ids of (some) replicas need the 32-bit union-find in HBM: room for the second id of every slot
        const int rc2 = dalloc(b, &b->dev.segs2, (size_t)b->dev.R * b->dev.stride, false);
        if (rc2) return rc2;
    }
    lc.lds_bytes = ((domask & SSE_DO_RVB) && b->lds_bytes_rvb > b->lds_bytes) ? b->lds_bytes_rvb : b->lds_bytes;
    b->dev.lds_words = (uint32_t)(lc.lds_bytes / 4);
    auto launch_dev = [&](const LaunchCfg &c, const DevBatch &dev, const SweepArgs &a) -> hipError_t {
        switch (c.W) {
        case 1: return launch_sweep_w1(c, dev, a);
        case 4: return launch_sweep_w4(c, dev, a);
        case 6: return launch_sweep_w6(c, dev, a);
        case 8: return launch_sweep_w8(c, dev, a);
        case 16: return launch_sweep_w16(c, dev, a);
        default: return hipErrorInvalidValue;
        }
    };
    auto launch = [&](const LaunchCfg &c, const SweepArgs &a) -> hipError_t { return launch_dev(c, b->dev, a); };
    auto fail_launch = [&](hipError_t e) { b->err = std::string("sweep launch: ") + hipGetErrorString(e); return ISINGMC_ENODEVICE; };
    uint32_t launches = 0;
    b->pass_ms[0] = b->pass_ms[1] = b->pass_ms[2] = 0.f;
    b->pass_launches[0] = b->pass_launches[1] = b->pass_launches[2] = 0;
    // passes of the first ("diagonal") launch of a split timestep: the diagonal pass and, unless an RVB sweep has to
    // come in between, the directed loop (one sequential walk: it gains nothing from the wider off-diagonal geometry)
    const uint32_t diag_bits = SSE_DO_DIAG | SSE_DO_HEATBATH | SSE_DO_GROW | ((A.domask & SSE_DO_RVB) ? 0u : SSE_DO_LOOP);
    const bool split = !b->fused_launch && (A.domask & SSE_DO_DIAG);
    size_t timed_steps = 0; // split path: steps whose launches carry events
    // launches without a diagonal or RVB pass use the kernel that leaves that code out
    DevBatch dev_off = b->dev;
    const bool loop_only = (A.domask & (SSE_DO_DIAG | SSE_DO_RVB | SSE_DO_CLUSTER | SSE_DO_FREE)) == 0 && (A.domask & SSE_DO_LOOP);
    // The off-diagonal kernel is latency-bound and small in registers: more waves per replica help as long as the
    // per-wave scan tables and the union-find of W*N + (transverse ops) ids still fit in LDS.  Decided from the largest
    // transverse-op count seen so far, and again every few timesteps of a long call (the count grows while a batch
    // equilibrates; replicas that outgrow the table only fall back to the slower HBM union-find, never fail).
    auto plan_offdiag = [&]() {
        uint32_t Wo = b->W_off ? b->W_off : b->W;
        bool hbm_uf = false;
        if (!b->W_off && b->W < 16) {
            const LdsPlan p16 = plan_lds(b, 16);
            if (p16.all_ids_fit) Wo = 16;
            else if (b->w8_ok && !b->uf_ids_limit && !plan_lds(b, b->W).all_ids_fit) {
                // the largest replicas need the 32-bit union-find in HBM whatever the geometry: spend the LDS on the scan
                // tables of 8 waves instead of on a 16-bit parent table that they cannot use (the HBM path is bound by
                // memory latency: twice the waves, twice the accesses in flight)
                Wo = 8; hbm_uf = true;
            }
        }
        LdsPlan po = plan_lds(b, Wo);
        if (hbm_uf) {
            const DevBatch &D = b->dev;
            po.ufcap = 0;
            po.lds_bytes = (4 * (lds_fixed_words(8, D.N, D.nwords, b->mode == SSE_MODE_LDS_EDGES ? D.E : 0u) + 64) + 7) & ~(size_t)7;
        }
        lc.W = Wo; lc.lds_bytes = po.lds_bytes;
        dev_off.lds_ufcap = po.ufcap;
        dev_off.lds_flipcap = 0u;
        if ((is_tg(b) || hbm_uf) && !(domask & SSE_DO_RVB)) {
            // HBM union-find launch: the LDS behind the fixed regions takes the flip bits of the ids (Wo * N + transverse ops seen so
            // far + headroom; a replica with more ids looks its flips up in HBM as before)
            const size_t used = po.lds_bytes / 4;
            const size_t want = ((size_t)Wo * b->dev.N + b->max_ntrans + b->max_ntrans / 16 + 384 + 31) / 32;
            const size_t avail = b->lds_total_words > used + 16 ? b->lds_total_words - used - 16 : 0;
            const size_t fw = want < avail ? want : avail;
            lc.lds_bytes = (4 * (used + fw) + 7) & ~(size_t)7;
            dev_off.lds_flipcap = (uint32_t)(32 * fw);
        }
        dev_off.lds_words = (uint32_t)(lc.lds_bytes / 4);
        b->last_W_off = Wo;
    };
    bool use_dev_off = false;
    const bool rvb_only = (A.domask & ~SSE_DO_GROW) == SSE_DO_RVB;
    if ((split || rvb_only) && (A.domask & SSE_DO_RVB) && !b->W_off && b->W < 16) {
        // RVB sweeps: the cooperative window scans of an attempt cover 4x more slots per step with 16 waves (the
        // sequential lane does not care); taken when the cluster tables of that geometry fit as well
        const LdsPlan p16 = plan_lds(b, 16);
        if (p16.all_ids_fit) {
            const DevBatch &D = b->dev;
            const size_t fixed16 = lds_fixed_words(16, D.N, D.nwords, b->mode == SSE_MODE_LDS_EDGES ? D.E : 0u);
            const size_t o_cur16 = fixed16 - ((size_t)16 * D.N + 1) / 2 - ((size_t)16 * D.N + 3) / 4;
            size_t want = 4 * (o_cur16 + 2 + rvb_fixed_words(D.N, D.E) + (size_t)D.cap);
            const size_t max_lds = b->lds_total_words * 4;
            if (want > max_lds) want = max_lds;
            want &= ~(size_t)7;
            lc.W = 16;
            lc.lds_bytes = want > p16.lds_bytes ? want : p16.lds_bytes;
            dev_off.lds_ufcap = p16.ufcap; dev_off.lds_words = (uint32_t)(lc.lds_bytes / 4);
            use_dev_off = true;
            b->last_W_off = 16;
        }
    }
    if (rvb_only) lc.passes = SSE_PASSES_RVB;   // the RVB sweep alone: its own kernel (no scratch spills, unlike the all-passes kernel)
    else if (loop_only) lc.passes = SSE_PASSES_DIAG; // a lone directed loop uses the small launch geometry too
    else if (!(A.domask & (SSE_DO_DIAG | SSE_DO_RVB | SSE_DO_LOOP)) || (split && !(A.domask & SSE_DO_RVB))) {
        lc.passes = SSE_PASSES_OFFDIAG;
        plan_offdiag();
    }
    // The cluster (+ free spins + sampling) launch of the headline geometry: the dedicated kernel, then the general one for the
    // replicas it flagged (ids beyond its LDS union-find, no op, no cut: a handful while a batch equilibrates, none afterwards;
    // that launch runs in the small diagonal geometry and its workgroups leave at once when their flag is clear).
    bool lean_now = false;
    LeanPlan lean{};
    auto plan_lean_now = [&]() { lean = plan_lean(b); lean_now = lean.ok && (b->K == 4 || b->K == 2) && !b->uf_ids_limit; b->last_lean = lean_now; };
    plan_lean_now();
    auto launch_lean = [&](const SweepArgs &a) -> hipError_t {
        LaunchCfg ll = lc;
        ll.W = 16; ll.K = b->K; ll.lds_bytes = lean.lds_bytes;
        DevBatch dv = b->dev;
        dv.lds_ufcap = lean.ufcap; dv.lds_words = (uint32_t)(lean.lds_bytes / 4);
        SweepArgs al = a;
        al.defer_flips = b->defer ? 1u : 0u;
        hipError_t e = launch_cluster(ll, dv, al);
        if (e != hipSuccess) return e;
        if (b->defer) b->pending = true;
        LaunchCfg lf = lc;
        lf.W = b->W; lf.passes = SSE_PASSES_OFFDIAG;
        const LdsPlan pf = plan_lds(b, b->W);
        DevBatch df = b->dev;
        lf.lds_bytes = pf.lds_bytes; df.lds_ufcap = pf.ufcap; df.lds_words = (uint32_t)(pf.lds_bytes / 4);
        SweepArgs af = a;
        af.only_flagged = 1u;
        return launch_dev(lf, df, af);
    };
    // RVB sweep: growth launch + main launch where that applies (sse_rvb_split.hip.h), else the fused kernel in geometry `lfused`
    auto launch_rvb = [&](const LaunchCfg &lfused, const DevBatch &dfused, const SweepArgs &a) -> hipError_t {
        const uint32_t updates = a.rvb_updates ? a.rvb_updates : (b->dev.N + 1u) / 2u;
        b->last_rvb_split = false;
        const size_t pstride = rvb_split_prod_stride(b->dev.Nb);
        if (b->rvb_split && a.nsteps == 1 && updates && pstride) {
            if (b->dev.rvb_prod_cap < updates) { // records of a sweep's attempts (grown on demand; no room -> the fused kernel)
                if (b->dev.rvb_prod) { (void)hipStreamSynchronize(b->stream); (void)hipFree(b->dev.rvb_prod); b->dev.rvb_prod = nullptr; b->dev.rvb_prod_cap = 0; }
                void *q = nullptr;
                if (hipMalloc(&q, (size_t)b->dev.R * updates * pstride * sizeof(uint32_t)) == hipSuccess) { b->dev.rvb_prod = (uint32_t *)q; b->dev.rvb_prod_cap = updates; b->dev.rvb_prod_stride = (uint32_t)pstride; }
                else { (void)hipGetLastError(); b->rvb_split = false; } // no room for the records: the fused kernel from now on
            }
            const DevBatch &D = b->dev;
            const uint32_t ledges = b->mode == SSE_MODE_LDS_EDGES ? D.E : 0u;
            const size_t max_lds = b->lds_total_words * 4;
            const size_t main_bytes = (4 * rvb_split_main_words(b->rvb_main_W, D.N, D.nwords, ledges, D.E, D.Nb) + 7) & ~(size_t)7;
            if (D.rvb_prod && main_bytes <= max_lds) {
                LaunchCfg lg = lfused;
                size_t want = 4 * (rvb_split_grow_fixed_words(D.N, D.nwords, ledges) + (size_t)D.cap + 16 * 640 + 2);
                if (want > max_lds) want = max_lds;
                lg.W = 16; lg.lds_bytes = want & ~(size_t)7;
                DevBatch dg = D;
                dg.lds_words = (uint32_t)(lg.lds_bytes / 4);
                hipError_t e = launch_rvb_grow(lg, dg, a);
                if (e != hipSuccess) return e;
                LaunchCfg lm = lfused;
                lm.W = b->rvb_main_W; lm.lds_bytes = main_bytes;
                DevBatch dm = D;
                dm.lds_words = (uint32_t)(main_bytes / 4);
                b->last_rvb_split = true;
                return launch_rvb_main(lm, dm, a);
            }
        }
        return launch_dev(lfused, dfused, a);
    };
    HIP_TRY(b, hipEventRecord(b->ev0, b->stream));
    if (!split) {
        const uint64_t per = b->steps_per_launch ? b->steps_per_launch : nsteps;
        for (uint64_t done = 0; done < nsteps; done += per) {
            A.step0 = done;
            A.nsteps = (nsteps - done < per) ? nsteps - done : per;
            const bool lean_here = lean_now && A.nsteps == 1 && (A.domask & SSE_DO_CLUSTER) && !(A.domask & ~(SSE_DO_CLUSTER | SSE_DO_FREE));
            const hipError_t e = lean_here ? launch_lean(A) : (rvb_only ? launch_rvb(lc, use_dev_off ? dev_off : b->dev, A) : launch_dev(lc, (use_dev_off || lc.passes == SSE_PASSES_OFFDIAG) ? dev_off : b->dev, A));
            if (e != hipSuccess) return fail_launch(e);
            launches++;
        }
        b->pass_launches[1] = launches;
    } else {
        // Two launches per timestep: the diagonal pass as its own kernel (twice the occupancy: it needs neither the
        // union-find LDS nor the registers of the cluster scan), then everything else.  Same Philox epochs, same
        // results as the fused launch; n / cutoff / chunk counters go through HBM in between (a few KB per replica).
        LaunchCfg ld = lc;
        ld.W = b->W;
        ld.passes = SSE_PASSES_DIAG;
        const bool use_fast = b->fast_diag && !(A.domask & SSE_DO_HEATBATH);
        // the cluster update of the same timestep takes the segment labelling from the diagonal launch (nothing but the
        // directed loop may sit in between: it changes no op's position or bond)
        const bool use_label = use_fast && b->lite && (A.domask & SSE_DO_CLUSTER) && !(A.domask & SSE_DO_RVB);
        const bool use_compact = use_fast && !use_label && b->compact && (A.domask & SSE_DO_CLUSTER) && !(A.domask & SSE_DO_RVB);
        // the diagonal launch needs the fixed regions up to the per-wave tables, which it uses as [W][N] bytes
        ld.lds_bytes = use_fast ? (use_label ? b->lds_bytes_fast_label : b->lds_bytes_fast) : diag_lds_bytes(b);
        if (is_pm(b) && b->lds_bytes_pm_diag) { ld.mode = SSE_MODE_PM_LDS_TABLES; ld.lds_bytes = b->lds_bytes_pm_diag; } // (the cluster tables stay in HBM)
        const uint32_t rest = A.domask & ~diag_bits;
        constexpr size_t MAX_TIMED = 256;
        const size_t want_ev = 4 * (size_t)(nsteps < MAX_TIMED ? nsteps : MAX_TIMED);
        while (b->evpool.size() < want_ev) { hipEvent_t ev; HIP_TRY(b, hipEventCreate(&ev)); b->evpool.push_back(ev); }
        constexpr uint64_t REPLAN_EVERY = 16;
        for (uint64_t done = 0; done < nsteps; ++done) {
            if (done && done % REPLAN_EVERY == 0 && lc.passes == SSE_PASSES_OFFDIAG) {
                int rcq = check_errors(b); // drains the stream, refreshes max_ntrans; an error ends the call here
                if (rcq) return rcq;
                plan_offdiag();
                plan_lean_now();
            }
            const bool timed = done < MAX_TIMED;
            SweepArgs a1 = A;
            a1.domask = (A.domask & diag_bits) | (use_label ? SSE_DO_LABEL : 0u) | (use_compact ? SSE_DO_COMPACT : 0u); a1.nsteps = 1; a1.step0 = done; a1.sampling_freq = 0; a1.out_u32 = nullptr;
            if (b->pending) { // (every step of a run after the first: the cluster update of the step before left flip bytes)
                if (use_fast && b->defer) { a1.defer_flips = 1u; b->pending = false; }
                else { const int rcm = ensure_materialized(b); if (rcm) return rcm; }
            }
            if (timed) HIP_TRY(b, hipEventRecord(b->evpool[4 * done], b->stream));
            hipError_t e = use_fast ? launch_sweep_fast(ld, b->dev, a1) : launch(ld, a1);
            if (e != hipSuccess) return fail_launch(e);
            launches++; b->pass_launches[0]++;
            if (timed) HIP_TRY(b, hipEventRecord(b->evpool[4 * done + 1], b->stream));
            const bool sample = freq && (done + 1) % freq == 0;
            uint32_t rest2 = rest;
            if ((rest & SSE_DO_RVB) && !(rest & SSE_DO_LOOP)) {
                // the RVB sweep as its own launch (register budget of its own: the all-passes kernel spills to scratch), then the
                // cluster / free-spin launch in its usual geometry
                SweepArgs ar = A;
                ar.domask = SSE_DO_RVB; ar.nsteps = 1; ar.step0 = done; ar.sampling_freq = 0; ar.out_u32 = nullptr;
                LaunchCfg lr = lc;
                lr.passes = SSE_PASSES_RVB;
                e = launch_rvb(lr, use_dev_off ? dev_off : b->dev, ar);
                if (e != hipSuccess) return fail_launch(e);
                launches++; b->pass_launches[1]++; b->pass_launches[2]++;
                rest2 = rest & ~SSE_DO_RVB;
            }
            if (timed) HIP_TRY(b, hipEventRecord(b->evpool[4 * done + 2], b->stream)); // (= the event above when no RVB sweep ran)
            if (rest2 || sample) {
                SweepArgs a2 = A;
                a2.domask = rest2; a2.nsteps = 1; a2.step0 = done;
                if (rest2 != rest) { // behind an RVB launch: the plain off-diagonal kernel and geometry
                    LaunchCfg lo = lc;
                    lo.passes = SSE_PASSES_OFFDIAG;
                    const LdsPlan po = plan_lds(b, lc.W);
                    DevBatch dv = use_dev_off ? dev_off : b->dev;
                    lo.lds_bytes = po.lds_bytes; dv.lds_ufcap = po.ufcap; dv.lds_flipcap = 0u; dv.lds_words = (uint32_t)(po.lds_bytes / 4);
                    if (lean_now && (rest2 & SSE_DO_CLUSTER) && !(rest2 & ~(SSE_DO_CLUSTER | SSE_DO_FREE))) e = launch_lean(a2);
                    else e = launch_dev(lo, dv, a2);
                } else if (lean_now && (rest2 & SSE_DO_CLUSTER) && !(rest2 & ~(SSE_DO_CLUSTER | SSE_DO_FREE))) {
                    e = launch_lean(a2);
                } else
                e = launch_dev(lc, (use_dev_off || lc.passes == SSE_PASSES_OFFDIAG) ? dev_off : b->dev, a2);
                if (e != hipSuccess) return fail_launch(e);
                launches++; b->pass_launches[1]++;
            }
            if (timed) { HIP_TRY(b, hipEventRecord(b->evpool[4 * done + 3], b->stream)); timed_steps++; }
        }
    }
    HIP_TRY(b, hipEventRecord(b->ev1, b->stream));
    int rc = check_errors(b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) { b->last_ms = ms; b->last_launches = launches; }
    if (!split) b->pass_ms[1] = b->last_ms;
    for (size_t i = 0; i < timed_steps; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, b->evpool[4 * i], b->evpool[4 * i + 1]) == hipSuccess) b->pass_ms[0] += t;
        if (hipEventElapsedTime(&t, b->evpool[4 * i + 1], b->evpool[4 * i + 3]) == hipSuccess) b->pass_ms[1] += t;
        if (hipEventElapsedTime(&t, b->evpool[4 * i + 1], b->evpool[4 * i + 2]) == hipSuccess) b->pass_ms[2] += t;
    }
    if (split && timed_steps && timed_steps < nsteps) { // scale the sampled steps up to the whole run
        const float f = (float)nsteps / (float)timed_steps;
        b->pass_ms[0] *= f; b->pass_ms[1] *= f; b->pass_ms[2] *= f;
    }
    if (rc) return rc;
    if (out_host) HIP_TRY(b, hipMemcpy(out_host, b->d_out, sizeof(uint32_t) * b->dev.R, hipMemcpyDeviceToHost));
    return ISINGMC_OK;
}

extern "C" {

int isingmc_interaction_at(const isingmc_interaction *it, const uint8_t *inputs, const uint8_t *outputs, double *out) {
    if (!it || !it->mat || !inputs || !outputs || !out || it->nvars == 0 || it->nvars > 2) return ISINGMC_EINVAL;
    // index_from_state (qmc_runner.rs:666-679): outputs then inputs, first variable most significant
    uint32_t iin = 0, iout = 0;
    for (uint32_t k = 0; k < it->nvars; ++k) { iin = (iin << 1) | (inputs[k] ? 1u : 0u); iout = (iout << 1) | (outputs[k] ? 1u : 0u); }
    if (it->diagonal_only) *out = (iin == iout) ? it->mat[iin] : 0.0;
    else *out = it->mat[(iout << it->nvars) | iin];
    return ISINGMC_OK;
}
int isingmc_interaction_sym_under_ising(const isingmc_interaction *it, int *out) {
    if (!it || !it->mat || !out || it->nvars == 0 || it->nvars > 2) return ISINGMC_EINVAL;
    const uint32_t n = it->nvars;
    const uint32_t mask = it->diagonal_only ? ((1u << n) - 1u) : ((1u << (2 * n)) - 1u);
    const uint32_t upto = it->diagonal_only ? (1u << (n >> 1)) : (1u << n);
    int sym = 1;
    for (uint32_t i = 0; i < upto; ++i)
        if (!(std::fabs(it->mat[i] - it->mat[(~i) & mask]) < DBL_EPSILON)) sym = 0;
    *out = sym;
    return ISINGMC_OK;
}

// Chunk grid of the per-chunk counters and the row stride of the op-string (and of every per-slot scratch row) for a batch whose
// kernels run with W waves (diagonal launches) and up to Wmax waves (off-diagonal launches) of K slots per lane.
//   CH      chunk size: <= SSE_MAX_CHUNKS chunks cover the capacity, CH a multiple of 256 (= a wave's tile at K = 4, two at K = 2)
//   stride  whole tiles of EITHER geometry (full-tile loads and stores never leave the row) and at least the chunk-rounded
//           capacity + 256: a cluster-scan wave whose chunk range is empty still prefetches one wave-tile at its range start
int isingmc_plan_geometry(uint32_t capacity, uint32_t W, uint32_t K, uint32_t Wmax, uint32_t out[4]) {
    if (!out || capacity == 0 || W == 0 || K == 0 || Wmax < W) return ISINGMC_EINVAL;
    const size_t CH = (((size_t)capacity + SSE_MAX_CHUNKS - 1) / SSE_MAX_CHUNKS + 255) / 256 * 256;
    const size_t nchunks = ((size_t)capacity + CH - 1) / CH;
    const size_t tile = (Wmax % W == 0 ? (size_t)Wmax : (size_t)Wmax * W) * 64 * K; // whole tiles of either launch geometry
    const size_t need1 = ((size_t)capacity + tile - 1) / tile * tile;
    const size_t need2 = ((size_t)capacity + CH - 1) / CH * CH + 256;
    const size_t need = need1 > need2 ? need1 : need2;
    const size_t stride = (need + tile - 1) / tile * tile;
    if (stride > 0xFFFFFFFFull / 4) return ISINGMC_EINVAL; // byte offsets inside a row are 32-bit (row_ld / row_st)
    out[0] = (uint32_t)CH; out[1] = (uint32_t)nchunks; out[2] = (uint32_t)stride; out[3] = (uint32_t)tile;
    return ISINGMC_OK;
}

int isingmc_create(const isingmc_config *cfg, isingmc_batch **out) {
    if (!cfg || !out || cfg->struct_size != sizeof(isingmc_config)) { g_create_error = "bad config pointer or struct_size"; return ISINGMC_EINVAL; }
    *out = nullptr;
    const bool generic = cfg->interactions != nullptr;
    if (generic) {
        if (cfg->nreplicas == 0 || cfg->nvars == 0 || cfg->ninteractions == 0) { g_create_error = "nreplicas, nvars, ninteractions must be > 0"; return ISINGMC_EINVAL; }
        if (cfg->flags & ISINGMC_CFG_PER_REPLICA_J) { g_create_error = "per-replica couplings are not available with generic interactions"; return ISINGMC_EINVAL; }
        for (uint32_t i = 0; i < cfg->ninteractions; ++i) {
            const isingmc_interaction &it = cfg->interactions[i];
            if (it.nvars > 2) { // qmc_runner.rs:415-680 allows any k; the 32-bit operator word holds two variables
                g_create_error = "interactions on more than two variables are not implemented (operator word = 2 in + 2 out bits)"; return ISINGMC_ENOTIMPL;
            }
            if ((it.nvars != 1 && it.nvars != 2) || !it.mat || it.vars[0] >= cfg->nvars || (it.nvars == 2 && (it.vars[1] >= cfg->nvars || it.vars[1] == it.vars[0]))) {
                g_create_error = "interaction must act on 1 or 2 distinct variables inside the model and carry a matrix"; return ISINGMC_EINVAL;
            }
            for (uint32_t k = 0; k < (it.diagonal_only ? (1u << it.nvars) : (1u << (2 * it.nvars))); ++k)
                if (!(it.mat[k] >= 0.0) || !std::isfinite(it.mat[k])) { g_create_error = "interaction matrix entries must be finite and >= 0"; return ISINGMC_EINVAL; }
        }
    } else
    if (cfg->nreplicas == 0 || cfg->nvars == 0 || (cfg->nedges != 0 && (!cfg->edges || !cfg->J))) { g_create_error = "nreplicas and nvars must be > 0 and edges/J non-null when nedges > 0"; return ISINGMC_EINVAL; }
    if (cfg->capacity == 0) { g_create_error = "capacity must be > 0"; return ISINGMC_EINVAL; }
    if (cfg->cutoff0 > cfg->capacity) { g_create_error = "cutoff0 exceeds capacity"; return ISINGMC_EINVAL; }
    if (cfg->nvars > SSE_VAR_MASK) { g_create_error = "too many variables"; return ISINGMC_EINVAL; }
    if (!generic && !(cfg->transverse >= 0.0)) { g_create_error = "transverse field must be >= 0"; return ISINGMC_EINVAL; }
    for (uint32_t e = 0; !generic && e < cfg->nedges; ++e)
        if (cfg->edges[2 * e] >= cfg->nvars || cfg->edges[2 * e + 1] >= cfg->nvars || cfg->edges[2 * e] == cfg->edges[2 * e + 1]) {
            g_create_error = "edge endpoint out of range";
            return ISINGMC_EINVAL;
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available (this library has no CPU fallback)"; return ISINGMC_ENODEVICE; }
    isingmc_batch *b = new isingmc_batch();
    int dev = cfg->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    b->device = dev;
    auto fail = [&](int rc) { g_create_error = b->err; isingmc_destroy(b); return rc; };
    if (hipSetDevice(dev) != hipSuccess) { b->err = "hipSetDevice failed"; return fail(ISINGMC_ENODEVICE); }

    DevBatch &D = b->dev;
    const bool perJ_cfg = (cfg->flags & ISINGMC_CFG_PER_REPLICA_J) != 0;
    if ((cfg->transverse_r || cfg->longitudinal_r) && (!perJ_cfg || generic)) { b->err = "per-replica fields need ISINGMC_CFG_PER_REPLICA_J (per-replica bond tables)"; return fail(ISINGMC_EINVAL); }
    auto gamma_of = [&](uint32_t row) { return cfg->transverse_r ? cfg->transverse_r[row] : cfg->transverse; };
    auto hfield_of = [&](uint32_t row) { return cfg->longitudinal_r ? cfg->longitudinal_r[row] : cfg->longitudinal; };
    const bool has_long = !generic && std::fabs(hfield_of(0)) > DBL_EPSILON; // qmc_ising.rs:230
    for (uint32_t r = 0; !generic && r < (perJ_cfg ? cfg->nreplicas : 1u); ++r) {
        if (!(gamma_of(r) >= 0.0) || !std::isfinite(gamma_of(r)) || !std::isfinite(hfield_of(r))) { b->err = "fields must be finite, transverse field >= 0"; return fail(ISINGMC_EINVAL); }
        if ((std::fabs(hfield_of(r)) > DBL_EPSILON) != has_long) { b->err = "longitudinal fields must be all zero or all non-zero within a batch"; return fail(ISINGMC_EINVAL); }
    }
    b->generic = generic;
    D.R = cfg->nreplicas; D.N = cfg->nvars; D.E = generic ? 0u : cfg->nedges;
    D.Nb = generic ? cfg->ninteractions : cfg->nedges + cfg->nvars + (has_long ? cfg->nvars : 0);
    if (D.Nb > SSE_MAX_BONDS) { b->err = "too many bonds"; return fail(ISINGMC_EINVAL); }
    D.cap = cfg->capacity; D.nwords = (cfg->nvars + 31) / 32;
    D.seed_lo = (uint32_t)cfg->seed; D.seed_hi = (uint32_t)(cfg->seed >> 32);
    D.replica_offset = cfg->replica_offset;

    // bond table (qmc_ising.rs:186-205,228-246; weights :863-888; offsets :97-99); one per replica when every
    // replica has its own couplings (disorder realisations, ISINGMC_CFG_PER_REPLICA_J: cfg->J is [R][E])
    const bool perJ = (cfg->flags & ISINGMC_CFG_PER_REPLICA_J) != 0;
    const uint32_t nH = perJ ? D.R : 1u;
    b->per_replica_J = perJ;
    std::vector<BondRec> &tab = b->bonds_host;
    tab.resize((size_t)nH * D.Nb);
    std::vector<double> cum((size_t)nH * D.Nb), wtots(nH);
    if (perJ) b->offsets.resize(nH);
    if (generic) {
        // bond b = interaction b.  Weights go to mats[b][in | out<<2] (bit 0 = first variable); the reference's index
        // is (out0 out1 in0 in1) with the first variable most significant (Interaction::index_from_state,
        // qmc_runner.rs:666-679).  Kinds only feed the transverse-op counters: a one-variable interaction with four
        // equal entries is a cluster edge (cluster.rs:284-286).
        b->mats_host.assign((size_t)D.Nb * 16, 0.0);
        double c = 0.0;
        for (uint32_t i = 0; i < D.Nb; ++i) {
            const isingmc_interaction &it = cfg->interactions[i];
            double *mb = b->mats_host.data() + (size_t)i * 16;
            for (uint32_t in = 0; in < (1u << it.nvars); ++in)      // device layout: bit 0 = first variable
                for (uint32_t out = 0; out < (1u << it.nvars); ++out) {
                    const uint8_t ib[2] = {(uint8_t)(in & 1u), (uint8_t)((in >> 1) & 1u)}, ob[2] = {(uint8_t)(out & 1u), (uint8_t)((out >> 1) & 1u)};
                    (void)isingmc_interaction_at(&it, ib, ob, &mb[in | (out << 2)]);
                }
            double maxw = 0.0; // heatbath.rs:130-146 make_bond_weights: largest diagonal element
            for (uint32_t st = 0; st < (it.nvars == 2 ? 4u : 2u); ++st) maxw = std::max(maxw, mb[st | (st << 2)]);
            const uint32_t kind = it.nvars == 2 ? SSE_BOND_TWO_SITE
                                  : ((mb[0] == mb[1] && mb[0] == mb[4] && mb[0] == mb[5]) ? SSE_BOND_TRANSVERSE : SSE_BOND_LONGITUDINAL);
            tab[i].a_info = it.vars[0] | (kind << SSE_INFO_SHIFT);
            tab[i].c = it.nvars == 2 ? it.vars[1] : SSE_NO_VAR;
            tab[i].w = maxw;
            c = (i == 0) ? maxw : maxw + c;
            cum[i] = c;
        }
        wtots[0] = c;
        b->offset = cfg->energy_offset;
        // Interaction::sym_under_ising (qmc_runner.rs:639-664): every weight equals the weight with all spins flipped
        b->generic_sym = true;
        for (uint32_t i = 0; i < D.Nb && b->generic_sym; ++i) {
            const double *mb = b->mats_host.data() + (size_t)i * 16;
            const uint32_t mask = cfg->interactions[i].nvars == 2 ? 0xFu : 0x5u;
            for (uint32_t idx = 0; idx < 16; ++idx)
                if ((idx & ~mask) == 0 && std::fabs(mb[idx] - mb[idx ^ mask]) >= DBL_EPSILON) { b->generic_sym = false; break; }
        }
    }
    for (uint32_t hI = 0; !generic && hI < nH; ++hI) {
        BondRec *t0 = tab.data() + (size_t)hI * D.Nb;
        const double *Jh = cfg->J + (size_t)hI * D.E;
        double off = 0.0;
        for (uint32_t e = 0; e < D.E; ++e) {
            const double J = Jh[e];
            t0[e].a_info = cfg->edges[2 * e] | ((SSE_BOND_TWO_SITE | (J < 0.0 ? SSE_BOND_PREF_BIT : 0u)) << SSE_INFO_SHIFT);
            t0[e].c = cfg->edges[2 * e + 1];
            t0[e].w = 2.0 * std::fabs(J);
            off += std::fabs(J);
        }
        const double gam = gamma_of(hI), hl = hfield_of(hI);
        for (uint32_t v = 0; v < D.N; ++v) {
            BondRec &t = t0[D.E + v];
            t.a_info = v | (SSE_BOND_TRANSVERSE << SSE_INFO_SHIFT); t.c = SSE_NO_VAR; t.w = gam;
        }
        if (has_long)
            for (uint32_t v = 0; v < D.N; ++v) {
                BondRec &t = t0[D.E + D.N + v];
                t.a_info = v | ((SSE_BOND_LONGITUDINAL | (hl > 0.0 ? SSE_BOND_PREF_BIT : 0u)) << SSE_INFO_SHIFT);
                t.c = SSE_NO_VAR; t.w = 2.0 * std::fabs(hl);
            }
        const double offset = off + (double)D.N * (gam + std::fabs(hl));
        if (hI == 0) b->offset = offset;
        if (perJ) b->offsets[hI] = offset;
        double c = 0.0;
        for (uint32_t i = 0; i < D.Nb; ++i) { c = (i == 0) ? t0[0].w : t0[i].w + c; cum[(size_t)hI * D.Nb + i] = c; }
        wtots[hI] = c;
    }
    D.wtot = wtots[0];
    D.bond_stride = perJ ? D.Nb : 0u;

    // launch geometry: W waves per replica, all of LDS for one workgroup
    int max_lds = 0;
    if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || max_lds <= 0) max_lds = 65536;
    const size_t total_words = (size_t)max_lds / 4;
    // default 4 waves per replica: with 16-bit union-find parents the footprint at the headline size stays below half
    // of the 160 KB LDS, so two workgroups share a CU and overlap each other's barriers (measured best on MI355X)
    uint32_t W = cfg->waves_per_replica ? cfg->waves_per_replica : 4;
    uint32_t K = cfg->slots_per_lane ? cfg->slots_per_lane : 4;
    if (W != 1 && W != 4 && W != 6 && W != 8 && W != 16) { b->err = "waves_per_replica must be 1, 4, 6, 8 or 16"; return fail(ISINGMC_EINVAL); }
    if (K != 1 && K != 2 && K != 4) { b->err = "slots_per_lane must be 1, 2 or 4"; return fail(ISINGMC_EINVAL); }
    // compact edge table staged in LDS when it is small enough (a|c<<15|pref<<30 needs N <= 32768)
    // uniform |J| lets the kernels keep the two-site weight in a scalar register
    D.uniformJ = 1u; D.wJ = tab[0].w;
    for (uint32_t hI = 0; hI < nH && D.uniformJ; ++hI)
        for (uint32_t e = 0; e < D.E; ++e) if (tab[(size_t)hI * D.Nb + e].w != tab[0].w) { D.uniformJ = 0u; break; }
    b->fused_launch = (cfg->flags & ISINGMC_CFG_FUSED_LAUNCH) != 0;
    D.rvb_growers = (cfg->flags & ISINGMC_CFG_RVB_SERIAL_GROWTH) ? 0u : 64u;
    const bool CL = !generic && !perJ && D.uniformJ && D.N <= SSE_CE_MAX_VARS && (size_t)D.E * 4 <= 48 * 1024 && !(cfg->flags & ISINGMC_CFG_NO_LDS_TABLES);
    const uint32_t ledges = CL ? D.E : 0u;
    // Per-variable scan tables: in LDS while W copies of them fit (with room for a union-find), otherwise in a per-replica
    // HBM scratch served by L2 / Infinity Cache (MODE 2; ISINGMC_CFG_GLOBAL_TABLES forces it on any model).
    bool TG = (cfg->flags & ISINGMC_CFG_GLOBAL_TABLES) != 0;
    if (!TG && lds_fixed_words(W, D.N, D.nwords, ledges) + 4096 > total_words) {
        if (cfg->waves_per_replica) { // explicit geometry: keep the LDS tables if a smaller W makes them fit (previous behaviour)
            while (W > 1 && lds_fixed_words(W, D.N, D.nwords, ledges) + 4096 > total_words) W = (W == 4) ? 1 : (W == 6 ? 4 : W >> 1);
            if (lds_fixed_words(W, D.N, D.nwords, ledges) + 64 > total_words) TG = true, W = cfg->waves_per_replica;
        } else TG = true;
    }
    if (TG && CL) { b->err = "ISINGMC_CFG_GLOBAL_TABLES needs the general bond table: combine it with ISINGMC_CFG_NO_LDS_TABLES"; return fail(ISINGMC_EINVAL); }
    if (TG && K == 2) K = 4;
    const size_t fixed = lds_fixed_words(W, D.N, D.nwords, ledges, TG, TG ? (D.E + 31u) / 32u : 0u); // (room for the +-J decode's signs, decided below)
    if (fixed + 64 > total_words) { b->err = "model too large: the spin-state bit arrays alone exceed LDS"; return fail(ISINGMC_ENOTIMPL); }
    // off-diagonal launches may use their own wave count (see run()): explicit, or decided per launch (then up to 16)
    uint32_t W_off = cfg->waves_offdiag;
    if (W_off != 0 && W_off != 1 && W_off != 4 && W_off != 6 && W_off != 8 && W_off != 16) { b->err = "waves_offdiag must be 0, 1, 4, 6, 8 or 16"; return fail(ISINGMC_EINVAL); }
    if (!W_off && cfg->waves_per_replica) W_off = W; // an explicit waves_per_replica pins both kinds of launch
    if (TG) W_off = W;                               // tables in HBM: one geometry for every launch
    if (W_off && lds_fixed_words(W_off, D.N, D.nwords, ledges, TG) + 64 > total_words) W_off = W;
    const bool w16_possible = !TG && lds_fixed_words(16, D.N, D.nwords, ledges) + 64 <= total_words;
    // (8 waves without an LDS union-find: the geometry of launches whose cluster ids need the 32-bit union-find in HBM anyway)
    const bool w8_possible = !TG && W < 8 && (K == 4 || K == 1) && lds_fixed_words(8, D.N, D.nwords, ledges) + 64 <= total_words;
    b->w8_ok = w8_possible && !W_off;
    const uint32_t Wmax = W_off ? (W_off > W ? W_off : W) : ((W < 16 && w16_possible) ? 16u : (w8_possible ? 8u : W));
    const size_t ids_max = (size_t)Wmax * D.N + D.cap;
    b->W = W; b->K = K; b->mode = TG ? SSE_MODE_GLOBAL_TABLES : (CL ? SSE_MODE_LDS_EDGES : SSE_MODE_GENERAL); b->W_off = W_off;
    // "+-J" decode for large disorder batches (BASELINE configs[4]): every replica its own coupling signs on one graph with uniform
    // |J|, Gamma, h.  The general decode fetches a 16-byte record per op and pass from a per-replica table of megabytes — one random
    // HBM sector each time, in a mode that is bound by exactly those; here the variables come from the shared compact edge table
    // (L2-resident), the sign from 12 KB of LDS.  Default geometry only.
    const bool PMJ = TG && perJ && D.uniformJ && !generic && W == 4 && K == 4 && D.N <= SSE_CE_MAX_VARS && !cfg->transverse_r && !cfg->longitudinal_r &&
                     !(cfg->flags & ISINGMC_CFG_NO_PM_DECODE);
    if (PMJ) {
        b->mode = SSE_MODE_PM_GLOBAL_TABLES;
        D.pm_words = (D.E + 31u) / 32u;
        // the diagonal launch keeps its per-wave spin bytes in LDS when W * N bytes fit next to the small arrays
        const size_t words = (size_t)D.nwords * 2 + 4 * W + 16 + 2 * SSE_MAX_CHUNKS + D.pm_words + ((size_t)W * D.N + 3) / 4;
        b->lds_bytes_pm_diag = (words + 64 <= total_words && !(cfg->flags & ISINGMC_CFG_GLOBAL_TABLES)) ? ((4 * words + 7) & ~(size_t)7) : 0;
    }
    { // chunk grid and row stride: one function (also exported for the CPU-side bound checks of tests/test_abi_cpu.py)
        uint32_t geo[4];
        if (isingmc_plan_geometry(D.cap, W, K, Wmax, geo) != ISINGMC_OK) { b->err = "capacity too large for the row stride"; return fail(ISINGMC_EINVAL); }
        D.CH = geo[0]; D.nchunks = geo[1]; D.stride = geo[2];
    }
    b->lds_fixed_words_ = fixed; b->lds_total_words = total_words; b->uf_ids_limit = cfg->lds_uf_ids_limit;
    b->lds_bytes_fast = fast_lds_bytes(D.N, D.nwords, D.E, D.Nb, false);
    b->fast_diag = CL && !TG && W == 4 && (K == 4 || K == 2) && D.N <= SSE_FAST_MAX_VARS && !b->fused_launch &&
                   !(cfg->flags & ISINGMC_CFG_NO_FAST_DIAG) && b->lds_bytes_fast <= 40 * 1024; // 4 workgroups per CU
    // ... and labels the segments for the cluster update of the same timestep (h = 0: no frozen segments to track)
    // the cluster update of that geometry has its own kernel too (16 waves, packed tables; sse_cluster.hip.h)
    b->lean_cluster = CL && !TG && !generic && D.N <= 4095u && !b->fused_launch && !cfg->waves_offdiag && !cfg->waves_per_replica &&
                      !(cfg->flags & ISINGMC_CFG_NO_LEAN_CLUSTER);
    b->lds_bytes_fast_label = fast_lds_bytes(D.N, D.nwords, D.E, D.Nb, true);
    // (opt-in as well: the diagonal launch pays for the two extra stores per op what the shorter scan gains)
    b->compact = b->fast_diag && (cfg->flags & ISINGMC_CFG_COMPACT) && !(cfg->flags & ISINGMC_CFG_FAST_LABEL);
    // (opt-in: measured on MI355X the labelling costs the diagonal launch more than it saves the cluster update)
    b->lite = b->fast_diag && !has_long && (cfg->flags & ISINGMC_CFG_FAST_LABEL) && b->lds_bytes_fast_label <= 40 * 1024;
    { // the RVB pass reuses everything from the scan tables on: launches that run it get enough LDS for its scratch
      // and constant-op table (other launches keep the smaller footprint, which decides workgroups per CU)
        const size_t o_cur = TG ? fixed : fixed - ((size_t)W * D.N + 1) / 2 - ((size_t)W * D.N + 3) / 4;
        const size_t want = 4 * (o_cur + 2 + rvb_fixed_words(D.N, D.E) + (size_t)D.cap);
        b->lds_bytes_rvb = (want < (size_t)max_lds ? want : (size_t)max_lds) & ~(size_t)7;
    }
    b->rvb_split = !generic && !TG && !is_pm(b) && !b->fused_launch && !(cfg->flags & ISINGMC_CFG_RVB_FUSED);
    { // waves of the RVB main launch: as many as keep about 16 waves on a CU (its LDS footprint decides how many replicas share one)
        const size_t w4 = 4 * rvb_split_main_words(4, D.N, D.nwords, CL ? D.E : 0u, D.E, D.Nb);
        const size_t per_cu = w4 ? (size_t)max_lds / w4 : 0;
        b->rvb_main_W = per_cu >= 4 ? 4u : (per_cu >= 2 ? 8u : 16u);
    }
    if (cfg->waves_per_replica == 4 || cfg->waves_per_replica == 8 || cfg->waves_per_replica == 16) b->rvb_main_W = cfg->waves_per_replica; // an explicit geometry is honoured here too
    size_lds(b);
    D.gamma = cfg->transverse; D.wh = 2.0 * std::fabs(cfg->longitudinal); D.hpos = cfg->longitudinal > 0.0 ? 1u : 0u;
    D.has_long = has_long ? 1u : 0u;

    int rc;
    if ((rc = dalloc(b, &D.ops, (size_t)D.R * D.stride))) return fail(rc);
    if ((rc = dalloc(b, &D.state, (size_t)D.R * D.nwords))) return fail(rc);
    if ((rc = dalloc(b, &D.n, D.R))) return fail(rc);
    if ((rc = dalloc(b, &D.ntrans, D.R))) return fail(rc);
    if ((rc = dalloc(b, &D.cutoff, D.R))) return fail(rc);
    if ((rc = dalloc(b, &D.err, D.R))) return fail(rc);
    if ((rc = dalloc(b, &D.aux, D.R))) return fail(rc);
    if ((rc = dalloc(b, &D.epoch, D.R))) return fail(rc);
    if ((rc = dalloc(b, &D.acc, (size_t)D.R * 8))) return fail(rc);
    b->acc_rows = D.R;
    if ((rc = dalloc(b, &b->d_acc_row, D.R))) return fail(rc);
    {
        std::vector<uint32_t> ident(D.R);
        for (uint32_t i = 0; i < D.R; ++i) ident[i] = i;
        if (hipMemcpy(b->d_acc_row, ident.data(), sizeof(uint32_t) * D.R, hipMemcpyHostToDevice) != hipSuccess) { b->err = "acc_row upload failed"; return fail(ISINGMC_ENODEVICE); }
        D.acc_row = b->d_acc_row;
    }
    if ((rc = dalloc(b, &D.chunks, (size_t)D.R * 2 * SSE_MAX_CHUNKS))) return fail(rc);
    if ((rc = dalloc(b, &D.segs, (size_t)D.R * D.stride, false))) return fail(rc);
    b->defer = b->lean_cluster && b->fast_diag && !(cfg->flags & ISINGMC_CFG_NO_DEFERRED_FLIPS);
    if (b->defer) { // flip bytes start (and stay, beyond every cutoff) zero
        if ((rc = dalloc(b, &D.flipb, (size_t)D.R * D.stride))) return fail(rc);
        if ((rc = dalloc(b, &D.pend, D.R))) return fail(rc);
    }
    if ((rc = dalloc(b, &D.dbg, (size_t)D.R * 16))) return fail(rc);
    BondRec *dbonds = nullptr; double *dcum = nullptr;
    if ((rc = dalloc(b, &dbonds, (size_t)nH * D.Nb, false))) return fail(rc);
    if ((rc = dalloc(b, &dcum, (size_t)nH * D.Nb, false))) return fail(rc);
    D.bonds = dbonds; D.cumw = dcum;
    if (perJ) {
        double *dwt = nullptr;
        if ((rc = dalloc(b, &dwt, nH, false))) return fail(rc);
        if (hipMemcpy(dwt, wtots.data(), sizeof(double) * nH, hipMemcpyHostToDevice) != hipSuccess) { b->err = "weight upload failed"; return fail(ISINGMC_ENODEVICE); }
        D.wtot_r = dwt;
    }
    {
        std::vector<double> ew(D.E);
        std::vector<uint32_t> ce(D.E, 0u);
        for (uint32_t e = 0; e < D.E; ++e) {
            ew[e] = tab[e].w;
            if (D.N <= SSE_CE_MAX_VARS)
                ce[e] = (tab[e].a_info & SSE_CE_VAR_MASK) | ((tab[e].c & SSE_CE_VAR_MASK) << 15) | (((tab[e].a_info >> (SSE_INFO_SHIFT + 2)) & 1u) << 30);
        }
        double *dew = nullptr; uint32_t *dce = nullptr;
        if ((rc = dalloc(b, &dew, D.E, false))) return fail(rc);
        if ((rc = dalloc(b, &dce, D.E, false))) return fail(rc);
        if (hipMemcpy(dew, ew.data(), sizeof(double) * D.E, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dce, ce.data(), sizeof(uint32_t) * D.E, hipMemcpyHostToDevice) != hipSuccess) { b->err = "edge table upload failed"; return fail(ISINGMC_ENODEVICE); }
        D.edge_w = dew; D.edges_compact = dce;
    }
    if (is_pm(b)) { // coupling signs of every bond-table row: bit e = prefers aligned (J < 0)
        std::vector<uint32_t> sg((size_t)nH * D.pm_words, 0u);
        for (uint32_t hI = 0; hI < nH; ++hI)
            for (uint32_t e = 0; e < D.E; ++e)
                if ((tab[(size_t)hI * D.Nb + e].a_info >> (SSE_INFO_SHIFT + 2)) & 1u) sg[(size_t)hI * D.pm_words + (e >> 5)] |= 1u << (e & 31);
        uint32_t *dsg = nullptr;
        if ((rc = dalloc(b, &dsg, sg.size(), false))) return fail(rc);
        if (hipMemcpy(dsg, sg.data(), 4 * sg.size(), hipMemcpyHostToDevice) != hipSuccess) { b->err = "sign upload failed"; return fail(ISINGMC_ENODEVICE); }
        D.pm_signs = dsg;
    }
    { // bonds_for_var (make_classical_bonds, qmc_ising.rs:421-432): edge order
        std::vector<uint32_t> as(D.N + 2, 0u), ad(2 * (size_t)D.E + 1), fill(D.N, 0u);
        for (uint32_t e = 0; e < D.E; ++e) { as[cfg->edges[2 * e] + 1]++; as[cfg->edges[2 * e + 1] + 1]++; }
        for (uint32_t v = 0; v < D.N; ++v) as[v + 1] += as[v];
        for (uint32_t e = 0; e < D.E; ++e) {
            const uint32_t a = cfg->edges[2 * e], c2 = cfg->edges[2 * e + 1];
            ad[as[a] + fill[a]++] = e;
            ad[as[c2] + fill[c2]++] = e;
        }
        uint32_t *das = nullptr, *dad = nullptr;
        if ((rc = dalloc(b, &das, D.N + 2, false))) return fail(rc);
        if ((rc = dalloc(b, &dad, 2 * (size_t)D.E + 1, false))) return fail(rc);
        if (hipMemcpy(das, as.data(), sizeof(uint32_t) * (D.N + 2), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dad, ad.data(), sizeof(uint32_t) * (2 * (size_t)D.E + 1), hipMemcpyHostToDevice) != hipSuccess) { b->err = "adjacency upload failed"; return fail(ISINGMC_ENODEVICE); }
        D.adj_start = das; D.adj = dad;
    }
    const size_t ufstride = ids_max + 2 * ((ids_max + 31) / 32);
    if ((rc = dalloc(b, &D.uf_scratch, (size_t)D.R * ufstride, false))) return fail(rc);
    if (b->compact) {
        if ((rc = dalloc(b, &D.cops, (size_t)D.R * D.stride, false))) return fail(rc);
        if ((rc = dalloc(b, &D.cpos, (size_t)D.R * D.stride, false))) return fail(rc);
        if ((rc = dalloc(b, &D.cops_epoch, D.R, false))) return fail(rc);
        if (hipMemset(D.cops_epoch, 0xFF, sizeof(uint64_t) * D.R) != hipSuccess) { b->err = "hipMemset failed"; return fail(ISINGMC_ENODEVICE); }
    }
    if (b->lite) {
        D.lite = 1u;
        if ((rc = dalloc(b, &D.pairs, (size_t)D.R * D.stride, false))) return fail(rc);
        if ((rc = dalloc(b, &D.pcount, (size_t)D.R * 4))) return fail(rc);
        if ((rc = dalloc(b, &D.lastrank, (size_t)D.R * D.N))) return fail(rc);
        if ((rc = dalloc(b, &D.touchbits, (size_t)D.R * D.nwords))) return fail(rc);
        if ((rc = dalloc(b, &D.lite_epoch, D.R, false))) return fail(rc);
        if (hipMemset(D.lite_epoch, 0xFF, sizeof(uint64_t) * D.R) != hipSuccess) { b->err = "hipMemset failed"; return fail(ISINGMC_ENODEVICE); }
    }
    if (TG) {
        D.tbl_stride = (uint32_t)((((size_t)Wmax * D.N * 4 + D.N) + 15) & ~(size_t)15); // 4-byte scan records per (wave, variable)
        if ((rc = dalloc(b, &D.tbl, (size_t)D.R * D.tbl_stride))) return fail(rc);
    }
    if ((rc = dalloc(b, &b->d_beta, D.R))) return fail(rc);
    if ((rc = dalloc(b, &b->d_out, D.R))) return fail(rc);
    if ((rc = dalloc(b, &b->d_vstate, (size_t)D.R * D.nwords))) return fail(rc);
    if ((rc = dalloc(b, &b->d_ok, D.R))) return fail(rc);
    if (hipMemcpy(dbonds, tab.data(), sizeof(BondRec) * (size_t)nH * D.Nb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dcum, cum.data(), sizeof(double) * (size_t)nH * D.Nb, hipMemcpyHostToDevice) != hipSuccess) { b->err = "table upload failed"; return fail(ISINGMC_ENODEVICE); }
    if (generic) {
        double *dm = nullptr;
        if ((rc = dalloc(b, &dm, b->mats_host.size(), false))) return fail(rc);
        if (hipMemcpy(dm, b->mats_host.data(), sizeof(double) * b->mats_host.size(), hipMemcpyHostToDevice) != hipSuccess) { b->err = "matrix upload failed"; return fail(ISINGMC_ENODEVICE); }
        D.mats = dm;
    }
    std::vector<uint32_t> cut(D.R, cfg->cutoff0);
    if (hipMemcpy(D.cutoff, cut.data(), sizeof(uint32_t) * D.R, hipMemcpyHostToDevice) != hipSuccess) { b->err = "cutoff upload failed"; return fail(ISINGMC_ENODEVICE); }
    if (hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess) { b->err = "hipEventCreate failed"; return fail(ISINGMC_ENODEVICE); }

    if (cfg->init_state) {
        rc = isingmc_set_state(b, UINT32_MAX, cfg->init_state);
        if (rc) return fail(rc);
    } else {
        hipLaunchKernelGGL(init_state_kernel, dim3(D.R), dim3(64), 0, b->stream, D);
        if (hipDeviceSynchronize() != hipSuccess) { b->err = "init_state_kernel failed"; return fail(ISINGMC_ENODEVICE); }
    }
    *out = b;
    return ISINGMC_OK;
}

static void pt_free(isingmc_batch *b);
void isingmc_destroy(isingmc_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    pt_free(b);
    for (void *p : b->allocs) (void)hipFree(p);
    if (b->dev.rvb_prod) (void)hipFree(b->dev.rvb_prod);
    for (hipEvent_t ev : b->evpool) (void)hipEventDestroy(ev);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    delete b;
}

const char *isingmc_last_error(const isingmc_batch *b) { return b ? b->err.c_str() : g_create_error.c_str(); }

int isingmc_diagonal_update(isingmc_batch *b, const double *beta, uint32_t flags) {
    uint32_t m = SSE_DO_DIAG | SSE_DO_GROW;
    if (flags & ISINGMC_FLAG_HEATBATH) m |= SSE_DO_HEATBATH;
    return run(b, beta, 1, 0, m, 0.5, nullptr);
}
int isingmc_cluster_update(isingmc_batch *b, double prob, uint32_t *n_clusters) {
    if (b && !(prob >= 0.0 && prob <= 1.0)) { b->err = "prob must be in [0,1]"; return ISINGMC_EINVAL; }
    std::vector<uint32_t> tmp;
    if (b && !n_clusters) { tmp.resize(b->dev.R); n_clusters = tmp.data(); }
    return run(b, nullptr, 1, 0, SSE_DO_CLUSTER, prob, n_clusters);
}
int isingmc_loop_update(isingmc_batch *b, uint32_t *lengths) {
    std::vector<uint32_t> tmp;
    if (b && !lengths) { tmp.resize(b->dev.R); lengths = tmp.data(); }
    return run(b, nullptr, 1, 0, SSE_DO_LOOP, 0.5, lengths);
}
int isingmc_rvb_update(isingmc_batch *b, uint32_t updates, uint32_t *successes) {
    if (!b) return ISINGMC_EINVAL;
    std::vector<uint32_t> tmp;
    if (!successes) { tmp.resize(b->dev.R); successes = tmp.data(); }
    b->rvb_updates = updates;
    const int rc = run(b, nullptr, 1, 0, SSE_DO_RVB, 0.5, successes);
    b->rvb_updates = 0;
    return rc;
}
int isingmc_flip_free_spins(isingmc_batch *b) { return run(b, nullptr, 1, 0, SSE_DO_FREE, 0.5, nullptr); }

int isingmc_timesteps(isingmc_batch *b, uint64_t t, const double *beta, uint32_t sampling_freq, uint32_t flags) {
    if (!b) return ISINGMC_EINVAL;
    b->rvb_updates = 0;
    uint32_t m = SSE_DO_DIAG | SSE_DO_GROW | SSE_DO_FREE;
    if (flags & ISINGMC_FLAG_HEATBATH) m |= SSE_DO_HEATBATH;
    if (flags & ISINGMC_FLAG_LOOP) m |= SSE_DO_LOOP;
    if (flags & ISINGMC_FLAG_RVB) m |= SSE_DO_RVB;
    if (!(flags & ISINGMC_FLAG_NO_CLUSTER)) m |= SSE_DO_CLUSTER;
    if (flags & ISINGMC_FLAG_PREP) m |= 0x10000u;
    if (sampling_freq == 0) sampling_freq = 1; // qmc_stepper.rs:147 unwrap_or(1)
    if (t == 0) return ISINGMC_OK;
    return run(b, beta, t, sampling_freq, m, 0.5, nullptr);
}

int isingmc_get_accumulators(isingmc_batch *b, uint64_t *out) {
    if (!b || !out) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipMemcpy(out, b->dev.acc, sizeof(uint64_t) * 8 * b->acc_rows, hipMemcpyDeviceToHost));
    return ISINGMC_OK;
}
int isingmc_set_accumulators(isingmc_batch *b, const uint64_t *in) {
    if (!b || !in) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    HIP_TRY(b, hipMemcpy(b->dev.acc, in, sizeof(uint64_t) * 8 * b->acc_rows, hipMemcpyHostToDevice));
    return ISINGMC_OK;
}
int isingmc_clear_errors(isingmc_batch *b) {
    if (!b) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    HIP_TRY(b, hipMemset(b->dev.err, 0, sizeof(uint32_t) * b->dev.R));
    b->err.clear();
    return ISINGMC_OK;
}
int isingmc_reset_accumulators(isingmc_batch *b) {
    if (!b) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipMemset(b->dev.acc, 0, sizeof(uint64_t) * 8 * b->acc_rows));
    return ISINGMC_OK;
}
int isingmc_set_accumulator_rows(isingmc_batch *b, uint32_t nrows, const uint32_t *rows) {
    if (!b || !rows || nrows == 0) return ISINGMC_EINVAL;
    for (uint32_t r = 0; r < b->dev.R; ++r) if (rows[r] >= nrows) { b->err = "accumulator row out of range"; return ISINGMC_EINVAL; }
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    if (nrows != b->acc_rows) {
        uint64_t *na = nullptr;
        int rc = dalloc(b, &na, (size_t)nrows * 8);
        if (rc) return rc;
        b->dev.acc = na; // the previous table stays in the allocation list until destroy
        b->acc_rows = nrows;
    }
    HIP_TRY(b, hipMemcpy(b->d_acc_row, rows, sizeof(uint32_t) * b->dev.R, hipMemcpyHostToDevice));
    return ISINGMC_OK;
}
int isingmc_set_cutoffs(isingmc_batch *b, const uint32_t *cutoffs) {
    if (!b || !cutoffs) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    std::vector<uint32_t> cur(b->dev.R);
    HIP_TRY(b, hipMemcpy(cur.data(), b->dev.cutoff, sizeof(uint32_t) * b->dev.R, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < b->dev.R; ++r) {
        if (cutoffs[r] > b->dev.cap) { b->err = "cutoff exceeds capacity"; return ISINGMC_ECAPACITY; }
        if (cutoffs[r] > cur[r]) cur[r] = cutoffs[r]; // fast_ops.rs:1258-1262: only grows
    }
    HIP_TRY(b, hipMemcpy(b->dev.cutoff, cur.data(), sizeof(uint32_t) * b->dev.R, hipMemcpyHostToDevice));
    return ISINGMC_OK;
}

// Host-side Philox4x32-10 for the tempering decisions (control logic, not the sweep path).
static void host_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// f64::powi as Rust lowers it (compiler-rt __powidf2): squaring sequence, reciprocal for negative exponents
static double host_powi(double x, int64_t n) {
    uint64_t m = n < 0 ? (uint64_t)(-n) : (uint64_t)n;
    double r = 1.0;
    while (m) { if (m & 1u) r *= x; x *= x; m >>= 1; }
    return n < 0 ? 1.0 / r : r;
}

int isingmc_pt_decide(uint64_t seed, uint64_t step, uint32_t nchains, uint32_t ntemps, const double *betas,
                      const uint32_t *n_of_config, uint32_t *config_at, uint64_t *nswaps) {
    if (!betas || !n_of_config || !config_at || nchains == 0 || ntemps == 0) return ISINGMC_EINVAL;
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint64_t swaps = 0;
    for (uint32_t chain = 0; chain < nchains && ntemps > 1; ++chain) {
        uint32_t ctr[4] = {0u, (uint32_t)step, chain, (SSE_TAG_PT << 24) | (uint32_t)((step >> 32) & 0xFFFFFFu)};
        uint32_t o[4];
        host_philox(ctr, key, o);
        const bool a_first = (o[0] >> 31) != 0u; // gen_bool(0.5) (tempering_container.rs:140)
        for (int phase = 0; phase < 2; ++phase) {
            const bool set_a = (phase == 0) ? a_first : !a_first;
            for (uint32_t t = set_a ? 0u : 1u; t + 1 < ntemps; t += 2) { // make_first/second_subgraphs (:83-99)
                ctr[0] = 1u + t;
                host_philox(ctr, key, o);
                const double u = (double)o[0] * (1.0 / 4294967296.0);
                uint32_t &ca = config_at[(size_t)t * nchains + chain], &cb = config_at[(size_t)(t + 1) * nchains + chain];
                const int64_t dn = (int64_t)n_of_config[cb] - (int64_t)n_of_config[ca];
                if (host_powi(betas[t] / betas[t + 1], dn) > u) { // swap_on_chunks (:296-298), equal Hamiltonians: f64::powi
                    const uint32_t tmp = ca; ca = cb; cb = tmp;
                    swaps++;
                }
            }
        }
    }
    if (nswaps) *nswaps += swaps;
    return ISINGMC_OK;
}
double isingmc_get_offset(const isingmc_batch *b) { return b ? b->offset : 0.0; }
int isingmc_get_offsets(const isingmc_batch *b, double *out) {
    if (!b || !out) return ISINGMC_EINVAL;
    for (uint32_t r = 0; r < b->dev.R; ++r) out[r] = b->per_replica_J ? b->offsets[b->ham_row_host.empty() ? r : b->ham_row_host[r]] : b->offset;
    return ISINGMC_OK;
}
uint32_t isingmc_num_bonds(const isingmc_batch *b) { return b ? b->dev.Nb : 0u; }

int isingmc_get_state(isingmc_batch *b, uint32_t r, uint8_t *out) {
    if (!b || !out || (r != UINT32_MAX && r >= b->dev.R)) { if (b) b->err = "bad replica index"; return ISINGMC_EINVAL; }
    HIP_TRY(b, hipSetDevice(b->device));
    const uint32_t r0 = r == UINT32_MAX ? 0 : r, cnt = r == UINT32_MAX ? b->dev.R : 1;
    std::vector<uint32_t> w((size_t)cnt * b->dev.nwords);
    HIP_TRY(b, hipMemcpy(w.data(), b->dev.state + (size_t)r0 * b->dev.nwords, w.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < cnt; ++i)
        for (uint32_t v = 0; v < b->dev.N; ++v) out[(size_t)i * b->dev.N + v] = (w[(size_t)i * b->dev.nwords + (v >> 5)] >> (v & 31)) & 1u;
    return ISINGMC_OK;
}
int isingmc_set_state(isingmc_batch *b, uint32_t r, const uint8_t *in) {
    if (!b || !in || (r != UINT32_MAX && r >= b->dev.R)) { if (b) b->err = "bad replica index"; return ISINGMC_EINVAL; }
    HIP_TRY(b, hipSetDevice(b->device));
    const uint32_t r0 = r == UINT32_MAX ? 0 : r, cnt = r == UINT32_MAX ? b->dev.R : 1;
    std::vector<uint32_t> w((size_t)cnt * b->dev.nwords, 0u);
    for (uint32_t i = 0; i < cnt; ++i)
        for (uint32_t v = 0; v < b->dev.N; ++v)
            if (in[(size_t)i * b->dev.N + v]) w[(size_t)i * b->dev.nwords + (v >> 5)] |= 1u << (v & 31);
    HIP_TRY(b, hipMemcpy(b->dev.state + (size_t)r0 * b->dev.nwords, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    return ISINGMC_OK;
}

static int get_u32(isingmc_batch *b, const uint32_t *src, uint32_t *out) {
    if (!b || !out) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipMemcpy(out, src, sizeof(uint32_t) * b->dev.R, hipMemcpyDeviceToHost));
    return ISINGMC_OK;
}
int isingmc_get_n(isingmc_batch *b, uint32_t *out) { return b ? get_u32(b, b->dev.n, out) : ISINGMC_EINVAL; }
int isingmc_get_cutoff(isingmc_batch *b, uint32_t *out) { return b ? get_u32(b, b->dev.cutoff, out) : ISINGMC_EINVAL; }
int isingmc_get_epoch(isingmc_batch *b, uint64_t *out) {
    if (!b || !out) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipMemcpy(out, b->dev.epoch, sizeof(uint64_t) * b->dev.R, hipMemcpyDeviceToHost));
    return ISINGMC_OK;
}
int isingmc_set_epoch(isingmc_batch *b, const uint64_t *epochs) {
    if (!b || !epochs) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipMemcpy(b->dev.epoch, epochs, sizeof(uint64_t) * b->dev.R, hipMemcpyHostToDevice));
    return ISINGMC_OK;
}
int isingmc_set_cutoff(isingmc_batch *b, uint32_t r, uint32_t cutoff) {
    if (!b || r >= b->dev.R) { if (b) b->err = "bad replica index"; return ISINGMC_EINVAL; }
    if (cutoff > b->dev.cap) { b->err = "cutoff exceeds capacity"; return ISINGMC_ECAPACITY; }
    HIP_TRY(b, hipSetDevice(b->device));
    uint32_t cur = 0;
    HIP_TRY(b, hipMemcpy(&cur, b->dev.cutoff + r, 4, hipMemcpyDeviceToHost));
    if (cutoff > cur) HIP_TRY(b, hipMemcpy(b->dev.cutoff + r, &cutoff, 4, hipMemcpyHostToDevice)); // fast_ops.rs:1258-1262: only grows
    return ISINGMC_OK;
}

int isingmc_export_ops(isingmc_batch *b, uint32_t r, uint32_t *words, uint32_t nwords) {
    if (!b || !words || r >= b->dev.R || nwords > b->dev.cap) { if (b) b->err = "bad arguments to export_ops"; return ISINGMC_EINVAL; }
    HIP_TRY(b, hipSetDevice(b->device));
    { const int rcm = ensure_materialized(b); if (rcm) return rcm; }
    HIP_TRY(b, hipMemcpy(words, b->dev.ops + (size_t)r * b->dev.stride, sizeof(uint32_t) * nwords, hipMemcpyDeviceToHost));
    return ISINGMC_OK;
}
int isingmc_import_ops(isingmc_batch *b, uint32_t r, const uint32_t *words, uint32_t nwords) {
    if (!b || (!words && nwords) || r >= b->dev.R) { if (b) b->err = "bad arguments to import_ops"; return ISINGMC_EINVAL; }
    if (nwords > b->dev.cap) { b->err = "op-string longer than capacity"; return ISINGMC_ECAPACITY; }
    { const int rcm = ensure_materialized(b); if (rcm) return rcm; }
    uint32_t n = 0, ntr = 0;
    const size_t hoff = b->per_replica_J ? (size_t)(b->ham_row_host.empty() ? r : b->ham_row_host[r]) * b->dev.Nb : 0; // this replica's bond table
    std::vector<uint32_t> chunks(2 * SSE_MAX_CHUNKS, 0u);
    for (uint32_t p = 0; p < nwords; ++p) {
        if (!words[p]) continue;
        const uint32_t bond = sse_op_bond(words[p]);
        if (bond >= b->dev.Nb) { b->err = "op refers to a bond outside the model"; return ISINGMC_EINVAL; }
        // Ising bonds: two-site and longitudinal ops have zero off-diagonal weight (qmc_ising.rs:863-888)
        const uint32_t kind = (b->bonds_host[hoff + bond].a_info >> SSE_INFO_SHIFT) & SSE_BOND_KIND_MASK;
        if (b->generic) {
            if (!(b->mats_host[(size_t)bond * 16 + (sse_op_in(words[p]) | (sse_op_out(words[p]) << 2))] > 0.0)) { b->err = "op with zero weight"; return ISINGMC_EINVAL; }
        } else
        if (kind != SSE_BOND_TRANSVERSE && sse_op_in(words[p]) != sse_op_out(words[p])) { b->err = "off-diagonal op on a diagonal-only bond (zero weight)"; return ISINGMC_EINVAL; }
        if (b->bonds_host[hoff + bond].c == SSE_NO_VAR && ((sse_op_in(words[p]) | sse_op_out(words[p])) & 2u)) { b->err = "single-site op with second-variable bits set"; return ISINGMC_EINVAL; }
        n++;
        chunks[p / b->dev.CH]++;
        if (((b->bonds_host[hoff + bond].a_info >> SSE_INFO_SHIFT) & SSE_BOND_KIND_MASK) == SSE_BOND_TRANSVERSE) { ntr++; chunks[SSE_MAX_CHUNKS + p / b->dev.CH]++; }
    }
    HIP_TRY(b, hipSetDevice(b->device));
    uint32_t *dst = b->dev.ops + (size_t)r * b->dev.stride;
    HIP_TRY(b, hipMemset(dst, 0, sizeof(uint32_t) * b->dev.stride));
    if (nwords) HIP_TRY(b, hipMemcpy(dst, words, sizeof(uint32_t) * nwords, hipMemcpyHostToDevice));
    uint32_t cur = 0;
    HIP_TRY(b, hipMemcpy(&cur, b->dev.cutoff + r, 4, hipMemcpyDeviceToHost));
    if (nwords > cur) HIP_TRY(b, hipMemcpy(b->dev.cutoff + r, &nwords, 4, hipMemcpyHostToDevice));
    HIP_TRY(b, hipMemcpy(b->dev.n + r, &n, 4, hipMemcpyHostToDevice));
    HIP_TRY(b, hipMemcpy(b->dev.ntrans + r, &ntr, 4, hipMemcpyHostToDevice));
    { const uint32_t zero = 0; HIP_TRY(b, hipMemcpy(b->dev.err + r, &zero, 4, hipMemcpyHostToDevice)); } // a fresh op-string: the replica's sticky error flag no longer applies
    if (ntr > b->max_ntrans) b->max_ntrans = ntr;
    HIP_TRY(b, hipMemcpy(b->dev.chunks + (size_t)r * 2 * SSE_MAX_CHUNKS, chunks.data(), sizeof(uint32_t) * chunks.size(), hipMemcpyHostToDevice));
    return ISINGMC_OK;
}
int isingmc_itime_magnetization(isingmc_batch *b, int64_t *sum_m, uint64_t *sum_m2, uint64_t *sum_abs_m) {
    if (!b || !sum_m || !sum_m2 || !sum_abs_m) { if (b) b->err = "bad arguments to itime_magnetization"; return ISINGMC_EINVAL; }
    HIP_TRY(b, hipSetDevice(b->device));
    { const int rcm = ensure_materialized(b); if (rcm) return rcm; }
    const uint32_t R = b->dev.R;
    long long *d1 = nullptr; unsigned long long *d2 = nullptr, *d3 = nullptr;
    HIP_TRY(b, hipMalloc(&d1, sizeof(long long) * R));
    if (hipMalloc(&d2, sizeof(unsigned long long) * R) != hipSuccess || hipMalloc(&d3, sizeof(unsigned long long) * R) != hipSuccess) {
        (void)hipFree(d1); if (d2) (void)hipFree(d2);
        b->err = "itime_magnetization: allocation failed"; return ISINGMC_ENODEVICE;
    }
    hipLaunchKernelGGL(itime_magnetization_kernel, dim3(R), dim3(256), 0, b->stream, b->dev, d1, d2, d3);
    hipError_t e = hipStreamSynchronize(b->stream);
    if (e == hipSuccess) e = hipMemcpy(sum_m, d1, sizeof(long long) * R, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(sum_m2, d2, sizeof(unsigned long long) * R, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(sum_abs_m, d3, sizeof(unsigned long long) * R, hipMemcpyDeviceToHost);
    (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(d3);
    if (e != hipSuccess) { b->err = std::string("itime_magnetization: ") + hipGetErrorString(e); return ISINGMC_ENODEVICE; }
    return ISINGMC_OK;
}
int isingmc_get_bond_count(isingmc_batch *b, uint32_t r, uint32_t bond, uint32_t *out) {
    if (!b || !out || r >= b->dev.R) { if (b) b->err = "bad arguments to get_bond_count"; return ISINGMC_EINVAL; }
    std::vector<uint32_t> w(b->dev.cap);
    int rc = isingmc_export_ops(b, r, w.data(), b->dev.cap);
    if (rc) return rc;
    uint32_t c = 0;
    for (uint32_t x : w) if (x && sse_op_bond(x) == bond) c++;
    *out = c;
    return ISINGMC_OK;
}

int isingmc_debug_counts(isingmc_batch *b, uint32_t *out) {
    if (!b || !out) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    { const int rcm = ensure_materialized(b); if (rcm) return rcm; }
    uint32_t *d = nullptr;
    HIP_TRY(b, hipMalloc((void **)&d, 12 * (size_t)b->dev.R));
    hipLaunchKernelGGL(debug_counts_kernel, dim3(b->dev.R), dim3(256), 0, b->stream, b->dev, d);
    hipError_t e = hipMemcpyAsync(out, d, 12 * (size_t)b->dev.R, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { b->err = std::string("debug_counts: ") + hipGetErrorString(e); return ISINGMC_ENODEVICE; }
    return ISINGMC_OK;
}

int isingmc_verify(isingmc_batch *b, uint8_t *ok) {
    if (!b || !ok) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    { const int rcm = ensure_materialized(b); if (rcm) return rcm; }
    hipLaunchKernelGGL(verify_kernel, dim3((b->dev.R + 63) / 64), dim3(64), 0, b->stream, b->dev, b->d_vstate, b->d_ok);
    HIP_TRY(b, hipGetLastError());
    HIP_TRY(b, hipMemcpyAsync(ok, b->d_ok, b->dev.R, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    return ISINGMC_OK;
}

int isingmc_set_stream(isingmc_batch *b, void *hip_stream) {
    if (!b) return ISINGMC_EINVAL;
    b->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return ISINGMC_OK;
}
int isingmc_debug_phase_ticks(isingmc_batch *b, uint64_t *out /*[R][16]*/, int reset) {
    if (!b || !out) return ISINGMC_EINVAL;
    if (reset >= 16) { b->dev.dbg_flags = (uint32_t)reset >> 4; reset &= 1; } // diagnostic builds: experiment flags
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipMemcpy(out, b->dev.dbg, sizeof(uint64_t) * 16 * b->dev.R, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(b, hipMemset(b->dev.dbg, 0, sizeof(uint64_t) * 16 * b->dev.R));
    return ISINGMC_OK;
}
int isingmc_set_steps_per_launch(isingmc_batch *b, uint64_t steps) {
    if (!b) return ISINGMC_EINVAL;
    b->steps_per_launch = steps;
    return ISINGMC_OK;
}
int isingmc_synchronize(isingmc_batch *b) {
    if (!b) return ISINGMC_EINVAL;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    return ISINGMC_OK;
}
int isingmc_last_kernel_ms(isingmc_batch *b, float *ms, uint32_t *launches) {
    if (!b) return ISINGMC_EINVAL;
    if (ms) *ms = b->last_ms;
    if (launches) *launches = b->last_launches;
    return ISINGMC_OK;
}
int isingmc_last_pass_ms(isingmc_batch *b, float ms[2], uint32_t launches[2]) {
    if (!b) return ISINGMC_EINVAL;
    if (ms) { ms[0] = b->pass_ms[0]; ms[1] = b->pass_ms[1]; }
    if (launches) { launches[0] = b->pass_launches[0]; launches[1] = b->pass_launches[1]; }
    return ISINGMC_OK;
}
int isingmc_last_rvb_ms(isingmc_batch *b, float *ms, uint32_t *launches) {
    if (!b) return ISINGMC_EINVAL;
    if (ms) *ms = b->pass_ms[2];
    if (launches) *launches = b->pass_launches[2];
    return ISINGMC_OK;
}
int isingmc_get_launch_info(const isingmc_batch *b, uint32_t out[8]) {
    if (!b || !out) return ISINGMC_EINVAL;
    out[0] = b->W; out[1] = (uint32_t)b->lds_bytes; out[2] = b->dev.lds_ufcap; out[3] = b->dev.nwords;
    out[4] = b->K; out[5] = b->mode == SSE_MODE_LDS_EDGES ? 1u : 0u; out[6] = (b->fused_launch ? 0u : 1u) | (b->last_W_off << 8) | (is_tg(b) ? 2u : 0u) | (b->fast_diag ? 4u : 0u) | (b->lite ? 8u : 0u) | (b->compact ? 16u : 0u) | (b->last_lean ? 32u : 0u) | (b->last_rvb_split ? 64u : 0u) | ((b->last_rvb_split ? b->rvb_main_W : 0u) << 16); out[7] = (uint32_t)diag_lds_bytes(b);
    return ISINGMC_OK;
}

} // extern "C"

// =====================================================================================================================
// Native parallel tempering (reference: TemperingContainer::tempering_step, parallel_tempering/tempering_container.rs:121-149,
// perform_swaps / swap_on_chunks :241-302, GraphWeights::relative_weight tempering_traits.rs:126-155).
//
// Sharding: rank g of G owns a contiguous block of ntemps/G temperatures for every chain ("walker").  Inside a block a swap
// exchanges temperature LABELS of two local replicas (slot_of); at a block boundary the two ranks exchange the boundary
// walkers' operator counts (4 B per chain and phase, ncclSend / ncclRecv in one group), both evaluate the same Philox-keyed
// decision, and an accepted swap moves the two configurations (op-string up to the cutoff, p=0 state, counters, Philox
// identity) through one more grouped send / receive.  Every rank therefore always holds exactly the configurations of its own
// temperature block, and no collective touches the sweep path.  The transport is RCCL point-to-point on device buffers when a
// communicator is attached (isingmc_pt_attach_nccl), otherwise the caller's host-staged sendrecv (tests: two ranks on one GPU).
#include <dlfcn.h>
#include <rccl/rccl.h> // types and enum values only (ncclUint32, ncclMax, ncclUniqueId): the library itself is dlopen()ed on first use
static_assert(sizeof(ncclUniqueId) == sizeof(isingmc_nccl_id), "isingmc_nccl_id must carry an ncclUniqueId");

struct PtState {
    uint32_t ntemps = 0, nchains = 0, rank = 0, world = 1, tper = 0;
    std::vector<double> betas;
    uint64_t seed = 0, step = 0, total_swaps = 0;
    std::vector<uint32_t> slot_of; // [R] global slot t*nchains + chain labelling local replica r
    std::vector<uint32_t> rid;     // [R] configuration identity (global id it was created with)
    uint32_t *d_rid = nullptr, *d_ham_row = nullptr;
    isingmc_pt_transport tr{};
    bool have_tr = false;
    // RCCL, loaded on first use
    void *nccl_lib = nullptr, *comm = nullptr;
    int (*p_send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*p_recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*p_gstart)() = nullptr;
    int (*p_gend)() = nullptr;
    int (*p_allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*p_init)(void **, int, isingmc_nccl_id, int) = nullptr;
    int (*p_destroy)(void *) = nullptr;
    // decisions on the device (single rank, one Hamiltonian for all temperatures): labels, betas per replica and the swap count live
    // in device memory; the host mirrors (slot_of) are refreshed on demand
    bool dev_decide = false, host_stale = false;
    uint32_t *d_slot_of = nullptr, *d_at = nullptr, *d_result = nullptr; // d_result: [0] swaps of the last step, [1] error flag
    double *d_betas = nullptr, *d_beta_r = nullptr;
    unsigned long long *d_total = nullptr; // total swaps since isingmc_pt_create / set_state
    uint32_t *d_items = nullptr; // [3 * nchains] accepted boundary swaps of one turn: (replica, word offset, cutoff)
    uint32_t *d_small = nullptr; // [4][nchains] staging of the boundary operator counts / cutoffs on the device (RCCL path)
    // different Hamiltonians per temperature (per-replica couplings): J rows of the neighbouring ranks' boundary slots
    bool hams_differ = false;
    std::vector<double> J_prev_last, J_next_first; // [nchains][E]
    uint32_t *d_counts = nullptr;                  // [R][Nb] bond counts (only when hams_differ)
    // configuration exchange
    uint32_t *d_pack_s = nullptr, *d_pack_r = nullptr;
    size_t pack_cap_words = 0;
    std::vector<uint32_t> h_pack_s, h_pack_r;
};

static void pt_free(isingmc_batch *b) {
    if (!b->pt) return;
    PtState *P = b->pt;
    if (P->comm && P->p_destroy) (void)P->p_destroy(P->comm);
    for (void *q : {(void *)P->d_rid, (void *)P->d_ham_row, (void *)P->d_small, (void *)P->d_counts, (void *)P->d_pack_s, (void *)P->d_pack_r, (void *)P->d_items,
                    (void *)P->d_slot_of, (void *)P->d_at, (void *)P->d_result, (void *)P->d_betas, (void *)P->d_beta_r, (void *)P->d_total})
        if (q) (void)hipFree(q);
    delete P;
    b->pt = nullptr;
}

// pack / unpack one configuration per workgroup: header (n, ntrans, cutoff, err, epoch lo/hi, rid, 0), p=0 state, chunk counters, op words
#define PT_HDR 8u
__global__ void pt_pack_kernel(DevBatch B, const uint32_t *rid, const uint32_t *items /*[nitems][3]: replica, word offset, cutoff*/, uint32_t *buf) {
    const uint32_t r = items[3 * blockIdx.x], off = items[3 * blockIdx.x + 1], cut = items[3 * blockIdx.x + 2];
    uint32_t *o = buf + off;
    if (threadIdx.x == 0) {
        o[0] = B.n[r]; o[1] = B.ntrans[r]; o[2] = B.cutoff[r]; o[3] = B.err[r];
        o[4] = (uint32_t)B.epoch[r]; o[5] = (uint32_t)(B.epoch[r] >> 32); o[6] = rid[r]; o[7] = 0u;
    }
    for (uint32_t i = threadIdx.x; i < B.nwords; i += blockDim.x) o[PT_HDR + i] = B.state[(size_t)r * B.nwords + i];
    for (uint32_t i = threadIdx.x; i < 2 * SSE_MAX_CHUNKS; i += blockDim.x) o[PT_HDR + B.nwords + i] = B.chunks[(size_t)r * 2 * SSE_MAX_CHUNKS + i];
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    for (uint32_t i = threadIdx.x; i < cut; i += blockDim.x) o[PT_HDR + B.nwords + 2 * SSE_MAX_CHUNKS + i] = ops[i];
}
__global__ void pt_unpack_kernel(DevBatch B, uint32_t *rid, const uint32_t *items, const uint32_t *buf) {
    const uint32_t r = items[3 * blockIdx.x], off = items[3 * blockIdx.x + 1], cut = items[3 * blockIdx.x + 2];
    const uint32_t *o = buf + off;
    if (threadIdx.x == 0) {
        B.n[r] = o[0]; B.ntrans[r] = o[1]; B.cutoff[r] = o[2]; B.err[r] = o[3];
        B.epoch[r] = (uint64_t)o[4] | ((uint64_t)o[5] << 32); rid[r] = o[6];
        if (B.lite) B.lite_epoch[r] = ~0ull;
        if (B.cops) B.cops_epoch[r] = ~0ull;
    }
    for (uint32_t i = threadIdx.x; i < B.nwords; i += blockDim.x) B.state[(size_t)r * B.nwords + i] = o[PT_HDR + i];
    for (uint32_t i = threadIdx.x; i < 2 * SSE_MAX_CHUNKS; i += blockDim.x) B.chunks[(size_t)r * 2 * SSE_MAX_CHUNKS + i] = o[PT_HDR + B.nwords + i];
    uint32_t *ops = B.ops + (size_t)r * B.stride;
    for (uint32_t i = threadIdx.x; i < cut; i += blockDim.x) ops[i] = o[PT_HDR + B.nwords + 2 * SSE_MAX_CHUNKS + i];
}
// OpContainer::get_count for every bond of every replica (op_container.rs:129): counts[r][bond]
__global__ void pt_bond_count_kernel(DevBatch B, uint32_t *counts) {
    const uint32_t r = blockIdx.x;
    const uint32_t *ops = B.ops + (size_t)r * B.stride;
    uint32_t *c = counts + (size_t)r * B.Nb;
    for (uint32_t i = threadIdx.x; i < B.Nb; i += blockDim.x) c[i] = 0u;
    __syncthreads();
    const uint32_t M = B.cutoff[r];
    for (uint32_t p = threadIdx.x; p < M; p += blockDim.x) { const uint32_t w = ops[p]; if (w) atomicAdd(&c[sse_op_bond(w)], 1u); }
}

struct PtDev {
    uint32_t *slot_of, *at, *result, *acc_row;
    const double *betas;
    double *beta_r;
    unsigned long long *total;
    uint32_t K, T, key0, key1;
    uint64_t step;
};
__device__ __forceinline__ double pt_dev_powi_signed(double x, long long n) { // (the multiplication sequence of pt_powi_signed)
    unsigned long long m = n < 0 ? (unsigned long long)(-n) : (unsigned long long)n;
    double r = 1.0;
    while (m) { if (m & 1ull) r *= x; x *= x; m >>= 1; }
    return n < 0 ? 1.0 / r : r;
}
// One tempering step of a rank that owns all temperatures (tempering_container.rs:121-149; the decisions of isingmc_pt_step's host
// path, same Philox counters): thread k takes chain k — equalise its cutoffs, draw the order coin, walk the two sets of pairs.
__global__ void pt_decide_kernel(DevBatch B, PtDev P) {
    __shared__ unsigned int s_swaps;
    if (threadIdx.x == 0) s_swaps = 0u;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < P.K; k += blockDim.x) {
        uint32_t maxcut = 0;
        for (uint32_t t = 0; t < P.T; ++t) { const uint32_t c = B.cutoff[P.at[t * P.K + k]]; maxcut = c > maxcut ? c : maxcut; }
        if (maxcut > B.cap) { P.result[1] = 1u; continue; }
        for (uint32_t t = 0; t < P.T; ++t) B.cutoff[P.at[t * P.K + k]] = maxcut;
        const uint32_t c3 = (SSE_TAG_PT << 24) | (uint32_t)((P.step >> 32) & 0xFFFFFFu);
        const bool a_first = (philox4x32_10(0u, (uint32_t)P.step, k, c3, P.key0, P.key1).x >> 31) != 0u;
        uint32_t swaps = 0;
        for (int phase = 0; phase < 2; ++phase) {
            const bool set_a = (phase == 0) ? a_first : !a_first;
            for (uint32_t t = 0; t + 1 < P.T; ++t) {
                if ((((t & 1u) == 0u) != set_a)) continue;
                const uint32_t la = t * P.K + k, lb = la + P.K;
                const uint32_t ra = P.at[la], rb = P.at[lb];
                const double u = (double)philox4x32_10(1u + t, (uint32_t)P.step, k, c3, P.key0, P.key1).x * (1.0 / 4294967296.0);
                const double p = pt_dev_powi_signed(P.betas[t] / P.betas[t + 1], (long long)B.n[rb] - (long long)B.n[ra]);
                if (p > u) { P.slot_of[ra] = lb; P.slot_of[rb] = la; P.at[la] = rb; P.at[lb] = ra; swaps++; }
            }
        }
        for (uint32_t t = 0; t < P.T; ++t) {
            const uint32_t r = P.at[t * P.K + k];
            P.beta_r[r] = P.betas[t];
            if (P.acc_row) P.acc_row[r] = t * P.K + k;
        }
        if (swaps) atomicAdd(&s_swaps, swaps);
    }
    __syncthreads();
    if (threadIdx.x == 0) { P.result[0] = s_swaps; *P.total += s_swaps; }
}

static double pt_powi(double x, uint32_t n) { // x^n by squaring (the oracle uses the same multiplication sequence)
    double r = 1.0;
    while (n) { if (n & 1u) r *= x; x *= x; n >>= 1; }
    return r;
}
// f64::powi for the swap test's temperature factor (tempering_container.rs:296): the same squaring sequence, the reciprocal
// for a negative exponent (compiler-rt __powidf2, which Rust's powi lowers to)
static double pt_powi_signed(double x, int64_t n) { return n < 0 ? 1.0 / pt_powi(x, (uint32_t)(-n)) : pt_powi(x, (uint32_t)n); }
// GraphWeights::relative_weight (tempering_traits.rs:126-155): the weight of a configuration under the Hamiltonian `to` relative
// to the one it lives in (`from`).  Rows are [E] couplings, then Gamma, then h: product over the edges of (J_to / J_from)^count in
// edge order, times (Gamma_to / Gamma_from)^(transverse ops), times (h_to / h_from)^(longitudinal ops) when h_from != 0
static double pt_relative_weight(const double *from, const double *to, const uint32_t *counts, uint32_t E, uint32_t N, bool has_long) {
    double w = 1.0;
    for (uint32_t e = 0; e < E; ++e) w *= pt_powi(to[e] / from[e], counts[e]);
    uint32_t tc = 0;
    for (uint32_t v = 0; v < N; ++v) tc += counts[E + v];
    w *= pt_powi(to[E] / from[E], tc);
    if (has_long && std::fabs(from[E + 1]) > DBL_EPSILON) {
        uint32_t lc = 0;
        for (uint32_t v = 0; v < N; ++v) lc += counts[E + N + v];
        w *= pt_powi(to[E + 1] / from[E + 1], lc);
    }
    return w;
}

// the Hamiltonian of bond-table row `row` as relative_weight needs it: J of every edge (weight 2|J|, "prefers aligned" = J < 0),
// Gamma (weight of the transverse bonds), h (weight 2|h|, "prefers up" = h > 0; 0 without longitudinal bonds)
static void pt_ham_row(const isingmc_batch *b, uint32_t row, double *out) {
    const uint32_t E = b->dev.E, N = b->dev.N, Nb = b->dev.Nb;
    const BondRec *t0 = b->bonds_host.data() + (size_t)row * Nb;
    for (uint32_t e = 0; e < E; ++e) out[e] = (((t0[e].a_info >> (SSE_INFO_SHIFT + 2)) & 1u) ? -0.5 : 0.5) * t0[e].w;
    out[E] = t0[E].w;
    out[E + 1] = b->dev.has_long ? (((t0[E + N].a_info >> (SSE_INFO_SHIFT + 2)) & 1u) ? 0.5 : -0.5) * t0[E + N].w : 0.0;
}

static int pt_exchange_small(isingmc_batch *b, int peer, const uint32_t *s, uint32_t *r, size_t count) {
    PtState *P = b->pt;
    if (peer < 0 || peer >= (int)P->world) return ISINGMC_OK;
    if (P->comm) { // RCCL point-to-point on device buffers, one group call
        uint32_t *ds = P->d_small, *dr = P->d_small + count;
        HIP_TRY(b, hipMemcpyAsync(ds, s, 4 * count, hipMemcpyHostToDevice, b->stream));
        if (P->p_gstart() || P->p_send(ds, count, (int)ncclUint32, peer, P->comm, b->stream) || P->p_recv(dr, count, (int)ncclUint32, peer, P->comm, b->stream) || P->p_gend()) {
            b->err = "RCCL send/recv failed"; return ISINGMC_ENODEVICE;
        }
        HIP_TRY(b, hipMemcpyAsync(r, dr, 4 * count, hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(b, hipStreamSynchronize(b->stream));
        return ISINGMC_OK;
    }
    if (!P->have_tr || P->tr.sendrecv(P->tr.ctx, peer, s, 4 * count, r, 4 * count)) { b->err = "tempering transport failed"; return ISINGMC_EINVAL; }
    return ISINGMC_OK;
}

extern "C" {

// device-side decisions: push the host's labels to the device / pull them back
static int pt_upload_labels(isingmc_batch *b) {
    PtState *P = b->pt;
    const uint32_t R = b->dev.R, K = P->nchains;
    std::vector<uint32_t> at(R);
    std::vector<double> br(R);
    for (uint32_t r = 0; r < R; ++r) { at[P->slot_of[r] - P->rank * R] = r; br[r] = P->betas[P->slot_of[r] / K]; }
    const unsigned long long tot = P->total_swaps;
    const uint32_t zero[2] = {0u, 0u};
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    HIP_TRY(b, hipMemcpy(P->d_slot_of, P->slot_of.data(), 4 * (size_t)R, hipMemcpyHostToDevice));
    HIP_TRY(b, hipMemcpy(P->d_at, at.data(), 4 * (size_t)R, hipMemcpyHostToDevice));
    HIP_TRY(b, hipMemcpy(P->d_beta_r, br.data(), 8 * (size_t)R, hipMemcpyHostToDevice));
    HIP_TRY(b, hipMemcpy(P->d_total, &tot, 8, hipMemcpyHostToDevice));
    HIP_TRY(b, hipMemcpy(P->d_result, zero, 8, hipMemcpyHostToDevice));
    P->host_stale = false;
    return ISINGMC_OK;
}
static int pt_sync_host(isingmc_batch *b) {
    PtState *P = b->pt;
    if (!P->dev_decide || !P->host_stale) return ISINGMC_OK;
    HIP_TRY(b, hipSetDevice(b->device));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    unsigned long long tot = 0;
    uint32_t res[2] = {0u, 0u};
    HIP_TRY(b, hipMemcpy(P->slot_of.data(), P->d_slot_of, 4 * (size_t)b->dev.R, hipMemcpyDeviceToHost));
    HIP_TRY(b, hipMemcpy(&tot, P->d_total, 8, hipMemcpyDeviceToHost));
    HIP_TRY(b, hipMemcpy(res, P->d_result, 8, hipMemcpyDeviceToHost));
    P->total_swaps = tot;
    P->host_stale = false;
    if (res[1]) { b->err = "cutoff exceeds capacity"; return ISINGMC_ECAPACITY; }
    return ISINGMC_OK;
}

int isingmc_pt_create(isingmc_batch *b, const isingmc_pt_layout *lay) {
    if (!b || !lay || lay->struct_size != sizeof(isingmc_pt_layout) || !lay->betas || lay->ntemps == 0 || lay->nchains == 0 || lay->world == 0 ||
        lay->rank >= lay->world || lay->ntemps % lay->world) { if (b) b->err = "bad tempering layout (temperatures must divide evenly over the ranks)"; return ISINGMC_EINVAL; }
    const uint32_t tper = lay->ntemps / lay->world;
    if ((size_t)tper * lay->nchains != b->dev.R) { b->err = "the batch must hold ntemps/world * nchains replicas"; return ISINGMC_EINVAL; }
    if (lay->world > 1 && !lay->transport) { b->err = "a transport (or isingmc_pt_attach_nccl) is needed for more than one rank"; return ISINGMC_EINVAL; }
    HIP_TRY(b, hipSetDevice(b->device));
    pt_free(b);
    PtState *P = new PtState();
    b->pt = P;
    P->ntemps = lay->ntemps; P->nchains = lay->nchains; P->rank = lay->rank; P->world = lay->world; P->tper = tper;
    P->betas.assign(lay->betas, lay->betas + lay->ntemps);
    P->seed = lay->seed;
    if (lay->transport) { P->tr = *lay->transport; P->have_tr = true; }
    const uint32_t R = b->dev.R;
    P->slot_of.resize(R); P->rid.resize(R);
    for (uint32_t r = 0; r < R; ++r) { P->slot_of[r] = P->rank * R + r; P->rid[r] = b->dev.replica_offset + r; }
    HIP_TRY(b, hipMalloc((void **)&P->d_rid, 4 * (size_t)R));
    HIP_TRY(b, hipMemcpy(P->d_rid, P->rid.data(), 4 * (size_t)R, hipMemcpyHostToDevice));
    b->dev.rid = P->d_rid;
    HIP_TRY(b, hipMalloc((void **)&P->d_small, 4 * 4 * (size_t)(P->nchains * 2 + 2)));
    HIP_TRY(b, hipMalloc((void **)&P->d_items, 4 * 3 * (size_t)P->nchains));
    P->pack_cap_words = (size_t)P->nchains * (PT_HDR + b->dev.nwords + 2 * SSE_MAX_CHUNKS + b->dev.cap);
    HIP_TRY(b, hipMalloc((void **)&P->d_pack_s, 4 * P->pack_cap_words));
    HIP_TRY(b, hipMalloc((void **)&P->d_pack_r, 4 * P->pack_cap_words));
    if (b->per_replica_J) {
        // different Hamiltonians between temperatures: the bond-table row belongs to the slot; neighbours' boundary rows once
        P->hams_differ = true;
        b->ham_row_host.resize(R);
        for (uint32_t r = 0; r < R; ++r) b->ham_row_host[r] = r;
        HIP_TRY(b, hipMalloc((void **)&P->d_ham_row, 4 * (size_t)R));
        HIP_TRY(b, hipMemcpy(P->d_ham_row, b->ham_row_host.data(), 4 * (size_t)R, hipMemcpyHostToDevice));
        b->dev.ham_row = P->d_ham_row;
        HIP_TRY(b, hipMalloc((void **)&P->d_counts, 4 * (size_t)R * b->dev.Nb));
        const uint32_t E = b->dev.E, K = P->nchains, HS = E + 2; // a Hamiltonian row: [E] couplings, Gamma, h
        auto Jrow = [&](uint32_t row, std::vector<double> &out, size_t at) { pt_ham_row(b, row, out.data() + at); };
        std::vector<double> first((size_t)K * HS), last((size_t)K * HS);
        for (uint32_t k = 0; k < K; ++k) { Jrow(k, first, (size_t)k * HS); Jrow((tper - 1) * K + k, last, (size_t)k * HS); }
        P->J_prev_last.assign((size_t)K * HS, 0.0); P->J_next_first.assign((size_t)K * HS, 0.0);
        if (P->world > 1) {
            const int prev = (int)P->rank - 1, next = (int)P->rank + 1;
            if (prev >= 0 && P->tr.sendrecv(P->tr.ctx, prev, first.data(), 8 * first.size(), P->J_prev_last.data(), 8 * first.size())) { b->err = "tempering transport failed"; return ISINGMC_EINVAL; }
            if (next < (int)P->world && P->tr.sendrecv(P->tr.ctx, next, last.data(), 8 * last.size(), P->J_next_first.data(), 8 * last.size())) { b->err = "tempering transport failed"; return ISINGMC_EINVAL; }
        }
    }
    if (P->world == 1 && !P->hams_differ) { // every pair is interior and weighs one Hamiltonian: the decisions run on the device
        HIP_TRY(b, hipMalloc((void **)&P->d_slot_of, 4 * (size_t)R));
        HIP_TRY(b, hipMalloc((void **)&P->d_at, 4 * (size_t)R));
        HIP_TRY(b, hipMalloc((void **)&P->d_result, 8));
        HIP_TRY(b, hipMalloc((void **)&P->d_betas, 8 * (size_t)P->ntemps));
        HIP_TRY(b, hipMalloc((void **)&P->d_beta_r, 8 * (size_t)R));
        HIP_TRY(b, hipMalloc((void **)&P->d_total, 8));
        HIP_TRY(b, hipMemcpy(P->d_betas, P->betas.data(), 8 * (size_t)P->ntemps, hipMemcpyHostToDevice));
        const int rcu = pt_upload_labels(b);
        if (rcu) return rcu;
        P->dev_decide = true;
    }
    return ISINGMC_OK;
}
int isingmc_pt_set_device_decisions(isingmc_batch *b, int on) {
    if (!b || !b->pt) { if (b) b->err = "isingmc_pt_create first"; return ISINGMC_EINVAL; }
    PtState *P = b->pt;
    HIP_TRY(b, hipSetDevice(b->device));
    if (on && !P->d_slot_of) { b->err = "device-side tempering decisions need a single rank and one Hamiltonian for all temperatures"; return ISINGMC_ENOTIMPL; }
    if (!on && P->dev_decide) { const int rc = pt_sync_host(b); if (rc) return rc; P->dev_decide = false; }
    else if (on && !P->dev_decide) { const int rc = pt_upload_labels(b); if (rc) return rc; P->dev_decide = true; }
    return ISINGMC_OK;
}
int isingmc_pt_get_device_decisions(const isingmc_batch *b, int *on) {
    if (!b || !b->pt || !on) return ISINGMC_EINVAL;
    *on = b->pt->dev_decide ? 1 : 0;
    return ISINGMC_OK;
}
// Sweeps at the temperatures of the current labels.  With device-side decisions the betas never visit the host.
int isingmc_pt_timesteps(isingmc_batch *b, uint64_t t, uint32_t sampling_freq, uint32_t flags) {
    if (!b || !b->pt) { if (b) b->err = "isingmc_pt_create first"; return ISINGMC_EINVAL; }
    PtState *P = b->pt;
    if (P->dev_decide) {
        b->beta_dev = P->d_beta_r;
        const int rc = isingmc_timesteps(b, t, nullptr, sampling_freq, flags);
        b->beta_dev = nullptr;
        return rc;
    }
    std::vector<double> br(b->dev.R);
    for (uint32_t r = 0; r < b->dev.R; ++r) br[r] = P->betas[P->slot_of[r] / P->nchains];
    return isingmc_timesteps(b, t, br.data(), sampling_freq, flags);
}

int isingmc_pt_nccl_unique_id(isingmc_nccl_id *out) {
    if (!out) return ISINGMC_EINVAL;
    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return ISINGMC_ENODEVICE;
    auto f = reinterpret_cast<int (*)(isingmc_nccl_id *)>(dlsym(lib, "ncclGetUniqueId"));
    return (f && f(out) == 0) ? ISINGMC_OK : ISINGMC_ENODEVICE;
}

int isingmc_pt_attach_nccl(isingmc_batch *b, const isingmc_nccl_id *id) {
    if (!b || !b->pt || !id) { if (b) b->err = "isingmc_pt_create first"; return ISINGMC_EINVAL; }
    PtState *P = b->pt;
    HIP_TRY(b, hipSetDevice(b->device));
    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD); // the copy the process already uses (e.g. torch's), if any
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { b->err = "librccl.so not found"; return ISINGMC_ENODEVICE; }
    P->nccl_lib = lib;
    P->p_send = reinterpret_cast<decltype(P->p_send)>(dlsym(lib, "ncclSend"));
    P->p_recv = reinterpret_cast<decltype(P->p_recv)>(dlsym(lib, "ncclRecv"));
    P->p_gstart = reinterpret_cast<decltype(P->p_gstart)>(dlsym(lib, "ncclGroupStart"));
    P->p_gend = reinterpret_cast<decltype(P->p_gend)>(dlsym(lib, "ncclGroupEnd"));
    P->p_allreduce = reinterpret_cast<decltype(P->p_allreduce)>(dlsym(lib, "ncclAllReduce"));
    P->p_init = reinterpret_cast<decltype(P->p_init)>(dlsym(lib, "ncclCommInitRank"));
    P->p_destroy = reinterpret_cast<decltype(P->p_destroy)>(dlsym(lib, "ncclCommDestroy"));
    if (!P->p_send || !P->p_recv || !P->p_gstart || !P->p_gend || !P->p_allreduce || !P->p_init) { b->err = "RCCL symbols missing"; return ISINGMC_ENODEVICE; }
    if (P->p_init(&P->comm, (int)P->world, *id, (int)P->rank) != 0) { P->comm = nullptr; b->err = "ncclCommInitRank failed"; return ISINGMC_ENODEVICE; }
    return ISINGMC_OK;
}

int isingmc_pt_get_slots(isingmc_batch *b, uint32_t *slot_of_replica, double *beta_of_replica, uint32_t *config_id_of_replica) {
    if (!b || !b->pt) { if (b) b->err = "isingmc_pt_create first"; return ISINGMC_EINVAL; }
    { const int rcs = pt_sync_host(b); if (rcs) return rcs; }
    const PtState *P = b->pt;
    for (uint32_t r = 0; r < b->dev.R; ++r) {
        if (slot_of_replica) slot_of_replica[r] = P->slot_of[r];
        if (beta_of_replica) beta_of_replica[r] = P->betas[P->slot_of[r] / P->nchains];
        if (config_id_of_replica) config_id_of_replica[r] = P->rid[r];
    }
    return ISINGMC_OK;
}

// Container-level save / load (the reference serialises the whole TemperingContainer, tempering_container.rs:683-792): the labels,
// the configurations' identities and the step counter; the replicas themselves go through the batch's own checkpoint.
int isingmc_pt_get_state(isingmc_batch *b, uint64_t *step, uint64_t *total_swaps) {
    if (!b || !b->pt) { if (b) b->err = "isingmc_pt_create first"; return ISINGMC_EINVAL; }
    { const int rcs = pt_sync_host(b); if (rcs) return rcs; }
    if (step) *step = b->pt->step;
    if (total_swaps) *total_swaps = b->pt->total_swaps;
    return ISINGMC_OK;
}
int isingmc_pt_set_state(isingmc_batch *b, const uint32_t *slot_of_replica, const uint32_t *config_id_of_replica, uint64_t step, uint64_t total_swaps) {
    if (!b || !b->pt || !slot_of_replica || !config_id_of_replica) { if (b) b->err = "isingmc_pt_create first"; return ISINGMC_EINVAL; }
    PtState *P = b->pt;
    const uint32_t R = b->dev.R, lo = P->rank * R;
    std::vector<uint8_t> seen(R, 0);
    for (uint32_t r = 0; r < R; ++r) {
        if (slot_of_replica[r] < lo || slot_of_replica[r] >= lo + R || seen[slot_of_replica[r] - lo]) { b->err = "slots must be a permutation of this rank's temperature block"; return ISINGMC_EINVAL; }
        seen[slot_of_replica[r] - lo] = 1;
    }
    HIP_TRY(b, hipSetDevice(b->device));
    for (uint32_t r = 0; r < R; ++r) { P->slot_of[r] = slot_of_replica[r]; P->rid[r] = config_id_of_replica[r]; }
    HIP_TRY(b, hipMemcpy(P->d_rid, P->rid.data(), 4 * (size_t)R, hipMemcpyHostToDevice));
    if (P->hams_differ) {
        for (uint32_t r = 0; r < R; ++r) b->ham_row_host[r] = P->slot_of[r] - lo;
        HIP_TRY(b, hipMemcpy(P->d_ham_row, b->ham_row_host.data(), 4 * (size_t)R, hipMemcpyHostToDevice));
    }
    P->step = step; P->total_swaps = total_swaps;
    if (P->dev_decide) return pt_upload_labels(b);
    return ISINGMC_OK;
}

// One tempering step of every chain (tempering_container.rs:121-149).  Adds the number of swaps this rank took part in
// as the LOWER temperature's owner (so that the sum over ranks counts every swap once) to *nswaps.
int isingmc_pt_step(isingmc_batch *b, uint64_t *nswaps) {
    if (!b || !b->pt) { if (b) b->err = "isingmc_pt_create first"; return ISINGMC_EINVAL; }
    PtState *P = b->pt;
    HIP_TRY(b, hipSetDevice(b->device));
    const uint32_t R = b->dev.R, K = P->nchains, T = P->ntemps, tper = P->tper, E = b->dev.E, Nb = b->dev.Nb;
    if (P->dev_decide) { // label swaps only: the op-strings (and any flip bytes still pending on them) are not touched
        if (T <= 1) { P->step++; return ISINGMC_OK; }
        PtDev D{};
        D.slot_of = P->d_slot_of; D.at = P->d_at; D.result = P->d_result; D.betas = P->d_betas; D.beta_r = P->d_beta_r; D.total = P->d_total;
        D.acc_row = b->acc_rows == T * K ? b->d_acc_row : nullptr; // per-slot accumulators (isingmc_set_accumulator_rows with the slots) follow the labels
        D.K = K; D.T = T; D.key0 = (uint32_t)P->seed; D.key1 = (uint32_t)(P->seed >> 32); D.step = P->step;
        hipLaunchKernelGGL(pt_decide_kernel, dim3(1), dim3(K < 256 ? ((K + 63) / 64) * 64 : 256), 0, b->stream, b->dev, D);
        HIP_TRY(b, hipGetLastError());
        P->step++;
        P->host_stale = true;
        if (nswaps) { // the caller wants this step's count: one small read-back
            uint32_t res[2] = {0u, 0u};
            HIP_TRY(b, hipStreamSynchronize(b->stream));
            HIP_TRY(b, hipMemcpy(res, P->d_result, 8, hipMemcpyDeviceToHost));
            if (res[1]) { b->err = "cutoff exceeds capacity"; return ISINGMC_ECAPACITY; }
            *nswaps += res[0];
        }
        return ISINGMC_OK;
    }
    { const int rcm = ensure_materialized(b); if (rcm) return rcm; }
    const uint32_t t_lo = P->rank * tper, t_hi = t_lo + tper; // my temperature block [t_lo, t_hi)
    const int prev = P->rank > 0 ? (int)P->rank - 1 : -1, next = P->rank + 1 < P->world ? (int)P->rank + 1 : -1;
    uint64_t swaps = 0;
    if (T <= 1) { P->step++; return ISINGMC_OK; }
    std::vector<uint32_t> n(R), cut(R);
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    HIP_TRY(b, hipMemcpy(n.data(), b->dev.n, 4 * (size_t)R, hipMemcpyDeviceToHost));
    HIP_TRY(b, hipMemcpy(cut.data(), b->dev.cutoff, 4 * (size_t)R, hipMemcpyDeviceToHost));
    // replica at a local slot
    std::vector<uint32_t> at(R);
    auto rebuild_at = [&]() { for (uint32_t r = 0; r < R; ++r) at[P->slot_of[r] - t_lo * K] = r; };
    rebuild_at();
    // ---- equalise the cutoffs of every chain over all temperatures (:129-137): max over the ranks ----
    std::vector<uint32_t> maxcut(K, 0u);
    for (uint32_t r = 0; r < R; ++r) { const uint32_t k = P->slot_of[r] % K; if (cut[r] > maxcut[k]) maxcut[k] = cut[r]; }
    if (P->world > 1) {
        if (P->comm) {
            HIP_TRY(b, hipMemcpyAsync(P->d_small, maxcut.data(), 4 * (size_t)K, hipMemcpyHostToDevice, b->stream));
            if (P->p_allreduce(P->d_small, P->d_small, K, (int)ncclUint32, (int)ncclMax, P->comm, b->stream)) { b->err = "ncclAllReduce failed"; return ISINGMC_ENODEVICE; }
            HIP_TRY(b, hipMemcpyAsync(maxcut.data(), P->d_small, 4 * (size_t)K, hipMemcpyDeviceToHost, b->stream));
            HIP_TRY(b, hipStreamSynchronize(b->stream));
        } else if (P->tr.allreduce_max_u32(P->tr.ctx, maxcut.data(), K)) { b->err = "tempering transport failed"; return ISINGMC_EINVAL; }
    }
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t k = P->slot_of[r] % K;
        if (maxcut[k] > b->dev.cap) { b->err = "cutoff exceeds capacity"; return ISINGMC_ECAPACITY; }
        cut[r] = maxcut[k];
    }
    HIP_TRY(b, hipMemcpy(b->dev.cutoff, cut.data(), 4 * (size_t)R, hipMemcpyHostToDevice));
    std::vector<uint32_t> counts; // bond counts of every local configuration (only when the Hamiltonians differ between temperatures)
    const uint32_t HS = E + 2;
    std::vector<double> Ja(HS), Jb(HS);
    // relative weight of local replica r (at local slot ls) towards the Hamiltonian of the slot above (+1) or below (-1)
    auto relw = [&](uint32_t r, uint32_t ls, int dir) -> double {
        if (!P->hams_differ) return 1.0;
        const uint32_t k = ls % K, tl = ls / K;
        pt_ham_row(b, ls, Ja.data());
        if (dir > 0) { if (tl + 1 < tper) pt_ham_row(b, ls + K, Jb.data()); else for (uint32_t e = 0; e < HS; ++e) Jb[e] = P->J_next_first[(size_t)k * HS + e]; }
        else { if (tl > 0) pt_ham_row(b, ls - K, Jb.data()); else for (uint32_t e = 0; e < HS; ++e) Jb[e] = P->J_prev_last[(size_t)k * HS + e]; }
        return pt_relative_weight(Ja.data(), Jb.data(), counts.data() + (size_t)r * Nb, E, b->dev.N, b->dev.has_long != 0u);
    };
    // order coin per chain (gen_bool(0.5), :140)
    const uint32_t key[2] = {(uint32_t)P->seed, (uint32_t)(P->seed >> 32)};
    std::vector<uint8_t> a_first(K);
    for (uint32_t k = 0; k < K; ++k) {
        const uint32_t ctr[4] = {0u, (uint32_t)P->step, k, (SSE_TAG_PT << 24) | (uint32_t)((P->step >> 32) & 0xFFFFFFu)};
        uint32_t o[4];
        host_philox(ctr, key, o);
        a_first[k] = (o[0] >> 31) != 0u;
    }
    auto decide = [&](uint32_t k, uint32_t t, uint32_t na, uint32_t nb2, double ra, double rb) -> bool {
        const uint32_t ctr[4] = {1u + t, (uint32_t)P->step, k, (SSE_TAG_PT << 24) | (uint32_t)((P->step >> 32) & 0xFFFFFFu)};
        uint32_t o[4];
        host_philox(ctr, key, o);
        const double u = (double)o[0] * (1.0 / 4294967296.0);
        double p = pt_powi_signed(P->betas[t] / P->betas[t + 1], (int64_t)nb2 - (int64_t)na); // swap_on_chunks (:296-298): powi
        if (P->hams_differ) p *= ra * rb;
        return p > u;
    };
    struct Wire { uint32_t n; uint32_t pad; double rel; };
    for (int phase = 0; phase < 2; ++phase) {
        if (P->hams_differ) { // (again in the second phase: a boundary swap of the first one replaced configurations)
            hipLaunchKernelGGL(pt_bond_count_kernel, dim3(R), dim3(256), 0, b->stream, b->dev, P->d_counts);
            counts.resize((size_t)R * Nb);
            HIP_TRY(b, hipMemcpyAsync(counts.data(), P->d_counts, 4 * counts.size(), hipMemcpyDeviceToHost, b->stream));
            HIP_TRY(b, hipStreamSynchronize(b->stream));
        }
        // boundary walkers of this phase: for chain k the pair (t, t+1) is in the phase's set iff (t even) == (set a)
        auto in_set = [&](uint32_t k, uint32_t t) { const bool set_a = (phase == 0) ? a_first[k] : !a_first[k]; return ((t & 1u) == 0u) == set_a; };
        // ---- exchange the operator counts (and relative weights) of the boundary walkers with both neighbours ----
        std::vector<Wire> s_prev(K), r_prev(K), s_next(K), r_next(K);
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t rf = at[k], rl = at[(tper - 1) * K + k];
            s_prev[k] = {n[rf], 0u, prev >= 0 ? relw(rf, k, -1) : 1.0};
            s_next[k] = {n[rl], 0u, next >= 0 ? relw(rl, (tper - 1) * K + k, +1) : 1.0};
        }
        static_assert(sizeof(Wire) == 16, "wire format");
        int rc;
        // (even ranks talk to their upper neighbour first: the host-staged transport is blocking)
        for (int turn = 0; turn < 2; ++turn) {
            const bool up = ((P->rank & 1u) == 0u) == (turn == 0);
            if (up) { if ((rc = pt_exchange_small(b, next, reinterpret_cast<const uint32_t *>(s_next.data()), reinterpret_cast<uint32_t *>(r_next.data()), 4 * (size_t)K))) return rc; }
            else if ((rc = pt_exchange_small(b, prev, reinterpret_cast<const uint32_t *>(s_prev.data()), reinterpret_cast<uint32_t *>(r_prev.data()), 4 * (size_t)K))) return rc;
        }
        // ---- decisions ----
        std::vector<uint32_t> items_next, items_prev; // accepted boundary swaps: (replica, word offset in the message, cutoff)
        size_t off_next = 0, off_prev = 0;
        for (uint32_t k = 0; k < K; ++k) {
            // interior pairs
            for (uint32_t t = t_lo; t + 1 < t_hi; ++t) {
                if (!in_set(k, t)) continue;
                const uint32_t la = (t - t_lo) * K + k, lb = la + K;
                const uint32_t ra_ = at[la], rb_ = at[lb];
                if (decide(k, t, n[ra_], n[rb_], relw(ra_, la, +1), relw(rb_, lb, -1))) {
                    std::swap(P->slot_of[ra_], P->slot_of[rb_]);
                    at[la] = rb_; at[lb] = ra_;
                    swaps++;
                }
            }
            const size_t words = PT_HDR + b->dev.nwords + 2 * SSE_MAX_CHUNKS + maxcut[k];
            // boundary pair with the next rank: (t_hi - 1, t_hi)
            if (next >= 0 && in_set(k, t_hi - 1)) {
                const uint32_t la = (tper - 1) * K + k, ra_ = at[la];
                if (decide(k, t_hi - 1, n[ra_], r_next[k].n, s_next[k].rel, r_next[k].rel)) {
                    items_next.insert(items_next.end(), {ra_, (uint32_t)off_next, maxcut[k]});
                    off_next += words;
                    n[ra_] = r_next[k].n;
                    swaps++; // counted by the owner of the lower temperature
                }
            }
            // boundary pair with the previous rank: (t_lo - 1, t_lo)
            if (prev >= 0 && in_set(k, t_lo - 1)) {
                const uint32_t rb_ = at[k];
                if (decide(k, t_lo - 1, r_prev[k].n, n[rb_], r_prev[k].rel, s_prev[k].rel)) {
                    items_prev.insert(items_prev.end(), {rb_, (uint32_t)off_prev, maxcut[k]});
                    off_prev += words;
                    n[rb_] = r_prev[k].n;
                }
            }
        }
        // ---- accepted boundary swaps: the two configurations change ranks ----
        for (int turn = 0; turn < 2; ++turn) {
            const bool up = ((P->rank & 1u) == 0u) == (turn == 0);
            const std::vector<uint32_t> &items = up ? items_next : items_prev;
            const size_t words = up ? off_next : off_prev;
            const int peer = up ? next : prev;
            if (peer < 0 || items.empty()) continue;
            const uint32_t nitems = (uint32_t)(items.size() / 3);
            uint32_t *d_it = P->d_items; // (at most one item per chain and turn)
            HIP_TRY(b, hipMemcpyAsync(d_it, items.data(), 4 * items.size(), hipMemcpyHostToDevice, b->stream));
            hipLaunchKernelGGL(pt_pack_kernel, dim3(nitems), dim3(256), 0, b->stream, b->dev, P->d_rid, d_it, P->d_pack_s);
            if (P->comm) {
                if (P->p_gstart() || P->p_send(P->d_pack_s, words, (int)ncclUint32, peer, P->comm, b->stream) || P->p_recv(P->d_pack_r, words, (int)ncclUint32, peer, P->comm, b->stream) || P->p_gend()) {
                    b->err = "RCCL send/recv failed"; return ISINGMC_ENODEVICE;
                }
            } else {
                P->h_pack_s.resize(words); P->h_pack_r.resize(words);
                HIP_TRY(b, hipMemcpyAsync(P->h_pack_s.data(), P->d_pack_s, 4 * words, hipMemcpyDeviceToHost, b->stream));
                HIP_TRY(b, hipStreamSynchronize(b->stream));
                if (P->tr.sendrecv(P->tr.ctx, peer, P->h_pack_s.data(), 4 * words, P->h_pack_r.data(), 4 * words)) { b->err = "tempering transport failed"; return ISINGMC_EINVAL; }
                HIP_TRY(b, hipMemcpyAsync(P->d_pack_r, P->h_pack_r.data(), 4 * words, hipMemcpyHostToDevice, b->stream));
            }
            hipLaunchKernelGGL(pt_unpack_kernel, dim3(nitems), dim3(256), 0, b->stream, b->dev, P->d_rid, d_it, P->d_pack_r);
            HIP_TRY(b, hipStreamSynchronize(b->stream));
        }
        if (!items_next.empty() || !items_prev.empty()) HIP_TRY(b, hipMemcpy(P->rid.data(), P->d_rid, 4 * (size_t)R, hipMemcpyDeviceToHost));
    }
    if (P->hams_differ) { // the bond-table row follows the slot
        for (uint32_t r = 0; r < R; ++r) b->ham_row_host[r] = P->slot_of[r] - t_lo * K;
        HIP_TRY(b, hipMemcpy(P->d_ham_row, b->ham_row_host.data(), 4 * (size_t)R, hipMemcpyHostToDevice));
    }
    P->step++;
    P->total_swaps += swaps;
    if (nswaps) *nswaps += swaps;
    return ISINGMC_OK;
}

} // extern "C"
