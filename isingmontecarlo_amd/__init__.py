"""isingmontecarlo_amd — MI355X-native SSE sweep for transverse-field Ising models.

Host-side mirror (Python, over the C ABI in include/isingmc_hip.h) of the reference crate's drivers for
the SSE hot path: `QmcIsingGraph` (src/sse/qmc_ising.rs) and `Qmc` (src/sse/qmc_runner.rs), batch-first:
one object = R independent replicas resident on one GPU.  All Monte-Carlo arithmetic runs in hand-written
gfx950 kernels (csrc/); there is no CPU fallback — constructing a graph without a HIP device raises.
"""
import ctypes as C
import os

import numpy as np

from . import _build

__all__ = ["QmcIsingGraph", "Qmc", "TemperingContainer", "NativeTemperingContainer", "IsingMcError", "load_library", "op_make", "op_fields",
           "interaction_at", "interaction_sym_under_ising",
           "FLAG_LOOP", "FLAG_NO_CLUSTER", "FLAG_HEATBATH", "FLAG_RVB", "FLAG_PREP", "CFG_NO_LDS_TABLES", "CFG_FUSED_LAUNCH", "CFG_PER_REPLICA_J", "CFG_GLOBAL_TABLES", "CFG_NO_FAST_DIAG", "CFG_FAST_LABEL", "CFG_COMPACT", "CFG_RVB_SERIAL_GROWTH", "CFG_NO_LEAN_CLUSTER", "CFG_NO_DEFERRED_FLIPS", "CFG_NO_PM_DECODE", "CFG_RVB_FUSED"]

FLAG_LOOP, FLAG_NO_CLUSTER, FLAG_HEATBATH, FLAG_RVB = 1, 2, 4, 8
FLAG_PREP = 0x10000
CFG_NO_LDS_TABLES = 1
CFG_PER_REPLICA_J = 4  # J is [nreplicas][nedges]: one disorder realisation per replica
CFG_GLOBAL_TABLES = 8  # per-variable scan tables in HBM instead of LDS (automatic for large models; forced by this flag)
CFG_NO_FAST_DIAG = 16  # general diagonal kernel even where the trimmed one applies (testing / A-B timing)
CFG_FAST_LABEL = 32  # experimental: segment labelling rides on the trimmed diagonal kernel (same results, currently slower)
CFG_RVB_SERIAL_GROWTH = 128  # RVB attempts grow their clusters one at a time (testing; same results)
CFG_COMPACT = 64  # experimental: cluster update scans the dense op list written by the trimmed diagonal kernel (same results, no net gain yet)
CFG_NO_LEAN_CLUSTER = 256  # general cluster kernel even where the dedicated one applies (testing / A-B timing)
CFG_NO_DEFERRED_FLIPS = 512  # the dedicated cluster kernel applies its flips itself instead of deferring them to the next diagonal launch (testing / A-B timing)
CFG_RVB_FUSED = 2048  # RVB sweeps through the fused kernel even where the growth + main launches apply (testing / A-B timing)
CFG_NO_PM_DECODE = 1024  # general bond records even where the +-J decode applies (testing / A-B timing)
CFG_FUSED_LAUNCH = 2  # whole timesteps in one kernel launch (default: diagonal launch + off-diagonal launch)
ALL = 0xFFFFFFFF

_ERRNAMES = {-1: "EINVAL", -2: "ENODEVICE", -3: "ECAPACITY", -4: "EINTEGRITY", -5: "ENOTIMPL", -6: "ELIMIT"}


class IsingMcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"isingmc {_ERRNAMES.get(code, code)}: {msg}")
        self.code = code


class _Interaction(C.Structure):
    """include/isingmc_hip.h: isingmc_interaction"""
    _fields_ = [("nvars", C.c_uint32), ("vars", C.c_uint32 * 2), ("diagonal_only", C.c_uint32), ("mat", C.POINTER(C.c_double))]


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("nreplicas", C.c_uint32), ("nvars", C.c_uint32),
                ("nedges", C.c_uint32), ("edges", C.POINTER(C.c_uint32)), ("J", C.POINTER(C.c_double)),
                ("transverse", C.c_double), ("longitudinal", C.c_double), ("capacity", C.c_uint32),
                ("cutoff0", C.c_uint32), ("seed", C.c_uint64), ("replica_offset", C.c_uint32),
                ("device", C.c_int32), ("init_state", C.POINTER(C.c_uint8)),
                ("waves_per_replica", C.c_uint32), ("slots_per_lane", C.c_uint32), ("flags", C.c_uint32),
                ("lds_uf_ids_limit", C.c_uint32), ("waves_offdiag", C.c_uint32),
                ("interactions", C.c_void_p), ("ninteractions", C.c_uint32), ("energy_offset", C.c_double),
                ("transverse_r", C.POINTER(C.c_double)), ("longitudinal_r", C.POINTER(C.c_double))]


_PT_SENDRECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
_PT_ALLMAX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t)


class _PtTransport(C.Structure):
    """include/isingmc_hip.h: isingmc_pt_transport"""
    _fields_ = [("ctx", C.c_void_p), ("sendrecv", _PT_SENDRECV), ("allreduce_max_u32", _PT_ALLMAX)]


class _PtLayout(C.Structure):
    """include/isingmc_hip.h: isingmc_pt_layout"""
    _fields_ = [("struct_size", C.c_uint32), ("ntemps", C.c_uint32), ("nchains", C.c_uint32), ("rank", C.c_uint32), ("world", C.c_uint32),
                ("betas", C.POINTER(C.c_double)), ("seed", C.c_uint64), ("transport", C.POINTER(_PtTransport))]


class _NcclId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


# every symbol include/isingmc_hip.h declares: name -> (restype, argtypes)
_vp, _u32, _u64, _f64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double
_P = C.POINTER
SYMBOLS = {
    "isingmc_create": (C.c_int, [_P(_Config), _P(_vp)]),
    "isingmc_destroy": (None, [_vp]),
    "isingmc_last_error": (C.c_char_p, [_vp]),
    "isingmc_diagonal_update": (C.c_int, [_vp, _P(_f64), _u32]),
    "isingmc_cluster_update": (C.c_int, [_vp, _f64, _P(_u32)]),
    "isingmc_loop_update": (C.c_int, [_vp, _P(_u32)]),
    "isingmc_rvb_update": (C.c_int, [_vp, _u32, _P(_u32)]),
    "isingmc_flip_free_spins": (C.c_int, [_vp]),
    "isingmc_timesteps": (C.c_int, [_vp, _u64, _P(_f64), _u32, _u32]),
    "isingmc_get_accumulators": (C.c_int, [_vp, _P(_u64)]),
    "isingmc_reset_accumulators": (C.c_int, [_vp]),
    "isingmc_set_accumulators": (C.c_int, [_vp, _P(_u64)]),
    "isingmc_clear_errors": (C.c_int, [_vp]),
    "isingmc_interaction_at": (C.c_int, [_P(_Interaction), _P(C.c_uint8), _P(C.c_uint8), _P(_f64)]),
    "isingmc_interaction_sym_under_ising": (C.c_int, [_P(_Interaction), _P(C.c_int)]),
    "isingmc_get_offset": (_f64, [_vp]),
    "isingmc_num_bonds": (_u32, [_vp]),
    "isingmc_get_state": (C.c_int, [_vp, _u32, _P(C.c_uint8)]),
    "isingmc_set_state": (C.c_int, [_vp, _u32, _P(C.c_uint8)]),
    "isingmc_get_n": (C.c_int, [_vp, _P(_u32)]),
    "isingmc_get_cutoff": (C.c_int, [_vp, _P(_u32)]),
    "isingmc_set_cutoff": (C.c_int, [_vp, _u32, _u32]),
    "isingmc_get_epoch": (C.c_int, [_vp, _P(_u64)]),
    "isingmc_get_bond_count": (C.c_int, [_vp, _u32, _u32, _P(_u32)]),
    "isingmc_export_ops": (C.c_int, [_vp, _u32, _P(_u32), _u32]),
    "isingmc_import_ops": (C.c_int, [_vp, _u32, _P(_u32), _u32]),
    "isingmc_verify": (C.c_int, [_vp, _P(C.c_uint8)]),
    "isingmc_debug_counts": (C.c_int, [_vp, _P(_u32)]),
    "isingmc_pt_decide": (C.c_int, [_u64, _u64, _u32, _u32, _P(_f64), _P(_u32), _P(_u32), _P(_u64)]),
    "isingmc_set_cutoffs": (C.c_int, [_vp, _P(_u32)]),
    "isingmc_pt_create": (C.c_int, [_vp, _P(_PtLayout)]),
    "isingmc_pt_nccl_unique_id": (C.c_int, [_P(_NcclId)]),
    "isingmc_pt_attach_nccl": (C.c_int, [_vp, _P(_NcclId)]),
    "isingmc_pt_step": (C.c_int, [_vp, _P(_u64)]),
    "isingmc_pt_get_slots": (C.c_int, [_vp, _P(_u32), _P(_f64), _P(_u32)]),
    "isingmc_pt_set_device_decisions": (C.c_int, [_vp, C.c_int]),
    "isingmc_pt_get_device_decisions": (C.c_int, [_vp, _P(C.c_int)]),
    "isingmc_pt_timesteps": (C.c_int, [_vp, _u64, _u32, _u32]),
    "isingmc_pt_get_state": (C.c_int, [_vp, _P(_u64), _P(_u64)]),
    "isingmc_pt_set_state": (C.c_int, [_vp, _P(_u32), _P(_u32), _u64, _u64]),
    "isingmc_set_accumulator_rows": (C.c_int, [_vp, _u32, _P(_u32)]),
    "isingmc_set_stream": (C.c_int, [_vp, _vp]),
    "isingmc_set_steps_per_launch": (C.c_int, [_vp, _u64]),
    "isingmc_debug_phase_ticks": (C.c_int, [_vp, _P(_u64), C.c_int]),
    "isingmc_synchronize": (C.c_int, [_vp]),
    "isingmc_last_kernel_ms": (C.c_int, [_vp, _P(C.c_float), _P(_u32)]),
    "isingmc_last_pass_ms": (C.c_int, [_vp, _P(C.c_float), _P(_u32)]),
    "isingmc_get_offsets": (C.c_int, [_vp, _P(C.c_double)]),
    "isingmc_set_epoch": (C.c_int, [_vp, _P(C.c_uint64)]),
    "isingmc_itime_magnetization": (C.c_int, [_vp, _P(C.c_int64), _P(C.c_uint64), _P(C.c_uint64)]),
    "isingmc_get_launch_info": (C.c_int, [_vp, _P(_u32)]),
    "isingmc_last_rvb_ms": (C.c_int, [_vp, _P(C.c_float), _P(_u32)]),
    "isingmc_plan_geometry": (C.c_int, [_u32, _u32, _u32, _u32, _P(_u32)]),
}

_LIB = None


def load_library(build=True):
    """dlopen the in-tree libisingmc_hip.so (building it first if stale) and bind every symbol."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    if build:
        path = _build.build()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -m isingmontecarlo_amd._build` (no CPU fallback exists)")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def op_make(bond, in_bits, out_bits):
    """Encode one operator word (include/sse_format.h)."""
    return ((bond + 1) << 4) | (in_bits & 3) | ((out_bits & 3) << 2)


def op_fields(word):
    """Decode an operator word -> (bond, in_bits, out_bits) or None for the identity."""
    if word == 0:
        return None
    return (int(word) >> 4) - 1, int(word) & 3, (int(word) >> 2) & 3


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _full_matrix(mat, k):
    """A [2^k] diagonal (InteractionType::Diagonal) expanded to the full [4^k] layout (index = outputs then inputs)."""
    m = np.asarray(mat, dtype=np.float64)
    if len(m) == 4 ** k:
        return m.copy()
    full = np.zeros(4 ** k)
    for s_ in range(2 ** k):
        full[(s_ << k) | s_] = m[s_]
    return full


def interaction_at(mat, inputs, outputs):
    """Interaction::at (qmc_runner.rs:573-612) through the C ABI (host-side helper, no device needed)."""
    lib = load_library()
    m = np.ascontiguousarray(np.asarray(mat, dtype=np.float64))
    k = len(inputs)
    it = _Interaction(nvars=k, diagonal_only=1 if len(m) == 2 ** k else 0, mat=_ptr(m, C.c_double))
    i = np.ascontiguousarray(np.asarray(inputs, dtype=np.uint8)); o = np.ascontiguousarray(np.asarray(outputs, dtype=np.uint8))
    out = C.c_double(0.0)
    rc = lib.isingmc_interaction_at(C.byref(it), _ptr(i, C.c_uint8), _ptr(o, C.c_uint8), C.byref(out))
    if rc:
        raise IsingMcError(rc, "bad interaction")
    return out.value


def interaction_sym_under_ising(mat, k):
    """Interaction::sym_under_ising (qmc_runner.rs:639-664) through the C ABI."""
    lib = load_library()
    m = np.ascontiguousarray(np.asarray(mat, dtype=np.float64))
    it = _Interaction(nvars=k, diagonal_only=1 if len(m) == 2 ** k else 0, mat=_ptr(m, C.c_double))
    out = C.c_int(0)
    rc = lib.isingmc_interaction_sym_under_ising(C.byref(it), C.byref(out))
    if rc:
        raise IsingMcError(rc, "bad interaction")
    return bool(out.value)


class QmcIsingGraph:
    """Batch of R transverse-field Ising SSE graphs on one GPU.

    Mirrors qmc::sse::QmcIsingGraph (src/sse/qmc_ising.rs): `edges` is a list of ((a, b), J) like the
    reference's Vec<(Edge, f64)>; `seed` replaces the `rng` argument (counter-based Philox on device).
    """

    def __init__(self, edges, transverse, longitudinal, cutoff, seed, state=None, nreplicas=1,
                 capacity=None, replica_offset=0, device=-1, waves_per_replica=0, slots_per_lane=0,
                 cfg_flags=0, lds_uf_ids_limit=0, waves_offdiag=0, couplings=None, nvars=None, transverse_r=None, longitudinal_r=None):
        """`couplings` (float64 [nreplicas][nedges]) gives every replica its own J values on the same graph (disorder
        realisations, BASELINE configs[4]); the J of `edges` is then ignored.  `nvars` overrides the reference's
        "largest edge index + 1" (qmc_ising.rs:92) and allows graphs without any edge (the reference's RVB fixtures,
        tests/check_rvb_crash.rs:68-110, run on isolated variables).  `transverse_r` / `longitudinal_r` (float64 [nreplicas]) give
        every replica its own fields (per-replica Hamiltonians, e.g. tempering between graphs that differ in Gamma and h:
        tempering_traits.rs:126-155); they imply per-replica bond tables like `couplings`."""
        lib = load_library()
        self._lib = lib
        self._h = None
        ed = np.ascontiguousarray(np.array([[a, b] for (a, b), _ in edges], dtype=np.uint32).reshape(-1, 2))
        js = np.ascontiguousarray(np.array([j for _, j in edges], dtype=np.float64))
        if len(ed) == 0 and nvars is None:
            raise IsingMcError(-1, "at least one edge (or an explicit nvars) is required")
        self.nvars = int(nvars) if nvars is not None else int(ed.max()) + 1  # qmc_ising.rs:92
        self.nreplicas = int(nreplicas)
        self.transverse, self.longitudinal = float(transverse), float(longitudinal)
        if couplings is not None:
            js = np.ascontiguousarray(np.asarray(couplings, dtype=np.float64))
            if js.shape != (int(nreplicas), len(ed)):
                raise IsingMcError(-1, "couplings must have shape [nreplicas][nedges]")
            cfg_flags = int(cfg_flags) | CFG_PER_REPLICA_J
        self.transverse_r = self.longitudinal_r = None
        if transverse_r is not None or longitudinal_r is not None:
            if couplings is None:  # per-replica tables with the same couplings everywhere
                js = np.ascontiguousarray(np.broadcast_to(js, (int(nreplicas), len(ed))).copy())
                cfg_flags = int(cfg_flags) | CFG_PER_REPLICA_J
            if transverse_r is not None:
                self.transverse_r = np.ascontiguousarray(np.broadcast_to(np.asarray(transverse_r, dtype=np.float64), (int(nreplicas),)).copy())
            if longitudinal_r is not None:
                self.longitudinal_r = np.ascontiguousarray(np.broadcast_to(np.asarray(longitudinal_r, dtype=np.float64), (int(nreplicas),)).copy())
        self.edges, self.J = ed, js
        if capacity is None:
            capacity = max(int(cutoff), 64)
        init = None
        if state is not None:
            st = np.asarray(state, dtype=np.uint8)
            if st.ndim == 1:
                st = np.broadcast_to(st, (self.nreplicas, self.nvars))
            init = np.ascontiguousarray(st)
            if init.shape != (self.nreplicas, self.nvars):
                raise IsingMcError(-1, "initial state has the wrong shape")
        cfg = _Config(struct_size=C.sizeof(_Config), nreplicas=self.nreplicas, nvars=self.nvars, nedges=len(ed),
                      edges=_ptr(ed, C.c_uint32), J=_ptr(js, C.c_double), transverse=self.transverse,
                      longitudinal=self.longitudinal, capacity=int(capacity), cutoff0=int(cutoff), seed=int(seed),
                      replica_offset=int(replica_offset), device=int(device),
                      init_state=_ptr(init, C.c_uint8) if init is not None else None,
                      waves_per_replica=int(waves_per_replica), slots_per_lane=int(slots_per_lane),
                      flags=int(cfg_flags), lds_uf_ids_limit=int(lds_uf_ids_limit), waves_offdiag=int(waves_offdiag),
                      transverse_r=_ptr(self.transverse_r, C.c_double) if self.transverse_r is not None else None,
                      longitudinal_r=_ptr(self.longitudinal_r, C.c_double) if self.longitudinal_r is not None else None)
        h = C.c_void_p()
        rc = lib.isingmc_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise IsingMcError(rc, lib.isingmc_last_error(None).decode())
        self._h = h
        self.capacity = int(capacity)
        self._flags = 0
        self._acc_rows = self.nreplicas

    # ---- lifetime ----
    def close(self):
        if getattr(self, "_h", None):
            self._lib.isingmc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise IsingMcError(rc, self._lib.isingmc_last_error(self._h).decode())

    def _betas(self, beta):
        b = np.ascontiguousarray(np.broadcast_to(np.asarray(beta, dtype=np.float64), (self.nreplicas,)))
        return b

    # ---- toggles (qmc_ising.rs:435,444) ----
    def set_enable_heatbath(self, enable):
        self._flags = (self._flags | FLAG_HEATBATH) if enable else (self._flags & ~FLAG_HEATBATH)

    def set_run_rvb(self, run_rvb):
        self._flags = (self._flags | FLAG_RVB) if run_rvb else (self._flags & ~FLAG_RVB)

    # ---- updates ----
    def timestep(self, beta):
        """QmcStepper::timestep (qmc_ising.rs:644): one sweep of every replica; returns the p=0 states."""
        self.timesteps(1, beta)
        return self.state_ref()

    def timesteps(self, t, beta, sampling_freq=1, flags=None):
        """QmcStepper::timesteps (qmc_stepper.rs:17-20): returns the energy estimate per replica."""
        f = self._flags if flags is None else flags
        self.reset_accumulators()
        b = self._betas(beta)
        self._check(self._lib.isingmc_timesteps(self._h, int(t), _ptr(b, C.c_double), int(sampling_freq), f))
        acc = self.accumulators()[:self.nreplicas]
        with np.errstate(divide="ignore", invalid="ignore"):
            avg_n = acc[:, 0] / acc[:, 1]
        return self.get_energy_for_average_n(avg_n, b)

    def run(self, t, beta, sampling_freq=1, flags=None):
        """timesteps without touching the accumulators (they keep summing)."""
        f = self._flags if flags is None else flags
        b = self._betas(beta)
        self._check(self._lib.isingmc_timesteps(self._h, int(t), _ptr(b, C.c_double), int(sampling_freq), f))

    def single_diagonal_step(self, beta):
        """qmc_ising.rs:208-272"""
        b = self._betas(beta)
        self._check(self._lib.isingmc_diagonal_update(self._h, _ptr(b, C.c_double), self._flags & FLAG_HEATBATH))

    def single_cluster_step(self, prob=0.5, flip_free=True):
        """qmc_ising.rs:275-320: cluster flips then free-spin randomisation; returns the cluster counts."""
        out = np.zeros(self.nreplicas, dtype=np.uint32)
        self._check(self._lib.isingmc_cluster_update(self._h, float(prob), _ptr(out, C.c_uint32)))
        if flip_free:
            self._check(self._lib.isingmc_flip_free_spins(self._h))
        return out

    def single_rvb_sweep(self, updates_in_sweep=None):
        """qmc_ising.rs:323-418: returns (successes per replica, attempts)."""
        upd = (self.nvars + 1) // 2 if updates_in_sweep is None else int(updates_in_sweep)
        out = np.zeros(self.nreplicas, dtype=np.uint32)
        self._check(self._lib.isingmc_rvb_update(self._h, upd, _ptr(out, C.c_uint32)))
        return out, upd

    def loop_update(self):
        out = np.zeros(self.nreplicas, dtype=np.uint32)
        self._check(self._lib.isingmc_loop_update(self._h, _ptr(out, C.c_uint32)))
        return out

    def flip_free_spins(self):
        self._check(self._lib.isingmc_flip_free_spins(self._h))

    # ---- observables / accessors ----
    def set_accumulator_rows(self, nrows, rows):
        """Replica r accumulates into row rows[r] of an [nrows][8] table (per-temperature statistics in PT)."""
        rw = np.ascontiguousarray(np.asarray(rows, dtype=np.uint32))
        self._check(self._lib.isingmc_set_accumulator_rows(self._h, int(nrows), _ptr(rw, C.c_uint32)))
        self._acc_rows = int(nrows)

    def set_cutoffs(self, cutoffs):
        c = np.ascontiguousarray(np.asarray(cutoffs, dtype=np.uint32))
        self._check(self._lib.isingmc_set_cutoffs(self._h, _ptr(c, C.c_uint32)))

    def accumulators(self):
        out = np.zeros((self._acc_rows, 8), dtype=np.uint64)
        self._check(self._lib.isingmc_get_accumulators(self._h, _ptr(out, C.c_uint64)))
        return out

    def reset_accumulators(self):
        self._check(self._lib.isingmc_reset_accumulators(self._h))

    def get_offset(self):
        return self._lib.isingmc_get_offset(self._h)

    def itime_magnetization(self):
        """imaginary_time_fold (qmc_ising.rs:815-821) of the magnetisation m = sum_v (2 s_v - 1): per replica the sums over
        p = 0..cutoff-1 of (m, m^2, |m|), computed on the device.  Divide by get_cutoff() for imaginary-time averages."""
        a = np.zeros(self.nreplicas, dtype=np.int64)
        b = np.zeros(self.nreplicas, dtype=np.uint64)
        c = np.zeros(self.nreplicas, dtype=np.uint64)
        self._check(self._lib.isingmc_itime_magnetization(self._h, _ptr(a, C.c_int64), _ptr(b, C.c_uint64), _ptr(c, C.c_uint64)))
        return a, b, c

    def imaginary_time_fold(self, fold_fn, init, r=0):
        """QmcStepper::imaginary_time_fold (qmc_ising.rs:815-821) with an arbitrary Python closure fold_fn(acc, state)
        for replica r: folds on the host over the exported op-string (fast_ops.rs:1296-1315)."""
        state = self.state_ref()[r].astype(bool).copy()
        ops = self.export_ops(r)
        e = self.edges
        generic = getattr(self, "interactions", None)
        acc = init
        for w in ops:
            acc = fold_fn(acc, state)
            if w:
                bond, _, out = op_fields(int(w))
                if generic is not None:
                    vs = generic[bond][1]
                    state[vs[0]] = bool(out & 1)
                    if len(vs) == 2:
                        state[vs[1]] = bool(out & 2)
                elif bond < len(e):
                    state[e[bond, 0]] = bool(out & 1); state[e[bond, 1]] = bool(out & 2)
                else:
                    state[(bond - len(e)) % self.nvars] = bool(out & 1)
        return acc

    def get_offsets(self):
        """Energy offset of every replica (they differ only with per-replica couplings)."""
        out = np.zeros(self.nreplicas, dtype=np.float64)
        self._check(self._lib.isingmc_get_offsets(self._h, _ptr(out, C.c_double)))
        return out

    def get_energy_for_average_n(self, average_n, beta):
        """qmc_ising.rs:805-809"""
        return -(np.asarray(average_n, dtype=np.float64) / beta) + self.get_offsets()

    def num_bonds(self):
        return self._lib.isingmc_num_bonds(self._h)

    def state_ref(self):
        out = np.zeros((self.nreplicas, self.nvars), dtype=np.uint8)
        self._check(self._lib.isingmc_get_state(self._h, ALL, _ptr(out, C.c_uint8)))
        return out

    clone_state = state_ref

    def set_state(self, state, r=None):
        st = np.ascontiguousarray(np.asarray(state, dtype=np.uint8))
        self._check(self._lib.isingmc_set_state(self._h, ALL if r is None else int(r), _ptr(st, C.c_uint8)))

    def _u32(self, fn):
        out = np.zeros(self.nreplicas, dtype=np.uint32)
        self._check(fn(self._h, _ptr(out, C.c_uint32)))
        return out

    def get_n(self):
        return self._u32(self._lib.isingmc_get_n)

    def get_cutoff(self):
        return self._u32(self._lib.isingmc_get_cutoff)

    def set_cutoff(self, cutoff, r=None):
        for i in (range(self.nreplicas) if r is None else [int(r)]):
            self._check(self._lib.isingmc_set_cutoff(self._h, i, int(cutoff)))

    def get_epoch(self):
        out = np.zeros(self.nreplicas, dtype=np.uint64)
        self._check(self._lib.isingmc_get_epoch(self._h, _ptr(out, C.c_uint64)))
        return out

    def get_bond_count(self, bond, r=0):
        out = C.c_uint32(0)
        self._check(self._lib.isingmc_get_bond_count(self._h, int(r), int(bond), C.byref(out)))
        return out.value

    def export_ops(self, r=0, nwords=None):
        if nwords is None:
            nwords = int(self.get_cutoff()[r])
        out = np.zeros(nwords, dtype=np.uint32)
        self._check(self._lib.isingmc_export_ops(self._h, int(r), _ptr(out, C.c_uint32), int(nwords)))
        return out

    def import_ops(self, words, r=0):
        w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
        self._check(self._lib.isingmc_import_ops(self._h, int(r), _ptr(w, C.c_uint32), len(w)))

    # ---- checkpoint / resume (the reference's serde feature, qmc_ising.rs:1001-1087) ----
    def save_checkpoint(self, path):
        """Write op-strings, p=0 states, cutoffs, update counters and accumulators of every replica to an .npz file.
        A batch built with the same model and seed and restored with load_checkpoint continues bit-exactly."""
        cut = self.get_cutoff()
        offs = np.zeros(self.nreplicas + 1, dtype=np.int64)
        offs[1:] = np.cumsum(cut.astype(np.int64))
        words = np.zeros(int(offs[-1]), dtype=np.uint32)
        for r in range(self.nreplicas):
            words[offs[r]:offs[r + 1]] = self.export_ops(r, int(cut[r]))
        generic = getattr(self, "interactions", None)
        gm = np.stack([m if len(m) == 16 else np.concatenate([m, np.zeros(12)]) for m, _ in generic]) if generic else np.zeros((0, 16))
        gv = np.array([[len(v), v[0], v[1] if len(v) == 2 else 0] for _, v in generic], dtype=np.uint32) if generic else np.zeros((0, 3), dtype=np.uint32)
        np.savez_compressed(path, format=np.array([2], dtype=np.uint32), nvars=np.array([self.nvars], dtype=np.uint32),
                            edges=self.edges, J=self.J, transverse=self.transverse, longitudinal=self.longitudinal,
                            cutoff=cut, offsets=offs, words=words, state=self.state_ref(), epoch=self.get_epoch(),
                            accumulators=self.accumulators(), interaction_mats=gm, interaction_vars=gv)

    def load_checkpoint(self, path):
        """Restore what save_checkpoint wrote into this batch (same graph and number of replicas required)."""
        z = np.load(path, allow_pickle=False)
        if int(z["nvars"][0]) != self.nvars or not np.array_equal(z["edges"], self.edges) or len(z["cutoff"]) != self.nreplicas:
            raise IsingMcError(-1, "checkpoint belongs to a different model or batch size")
        if not np.array_equal(z["J"], self.J) or float(z["transverse"]) != self.transverse or float(z["longitudinal"]) != self.longitudinal:
            raise IsingMcError(-1, "checkpoint belongs to a different Hamiltonian")
        generic = getattr(self, "interactions", None)
        if "interaction_vars" in z.files:  # generic Hamiltonians: the matrices and variables must agree too
            gv, gm = z["interaction_vars"], z["interaction_mats"]
            mine_v = np.array([[len(v), v[0], v[1] if len(v) == 2 else 0] for _, v in generic], dtype=np.uint32) if generic else np.zeros((0, 3), dtype=np.uint32)
            mine_m = np.stack([m if len(m) == 16 else np.concatenate([m, np.zeros(12)]) for m, _ in generic]) if generic else np.zeros((0, 16))
            if gv.shape != mine_v.shape or not np.array_equal(gv, mine_v) or not np.array_equal(gm, mine_m):
                raise IsingMcError(-1, "checkpoint belongs to a different set of interactions")
        elif generic:
            raise IsingMcError(-1, "checkpoint does not record its interactions (format 1): cannot be matched to a generic model")
        # the cutoff is part of the Philox trajectory (slots beyond it are never visited): a batch that already carries a
        # larger cutoff than the checkpoint cannot resume bit-exactly (cutoffs only grow, fast_ops.rs:1258-1262)
        if (self.get_cutoff() > z["cutoff"]).any():
            raise IsingMcError(-1, "this batch already has a larger cutoff than the checkpoint: build it with cutoff <= the saved one")
        offs, words = z["offsets"], z["words"]
        self.set_state(z["state"])
        for r in range(self.nreplicas):
            self.import_ops(words[offs[r]:offs[r + 1]], r)  # also clears the replica's device error flag
        self.set_cutoffs(z["cutoff"])
        ep = np.ascontiguousarray(z["epoch"].astype(np.uint64))
        self._check(self._lib.isingmc_set_epoch(self._h, _ptr(ep, C.c_uint64)))
        acc = np.ascontiguousarray(z["accumulators"].astype(np.uint64))
        if acc.shape == (self._acc_rows, 8):
            self._check(self._lib.isingmc_set_accumulators(self._h, _ptr(acc, C.c_uint64)))
        else:
            raise IsingMcError(-1, "checkpoint accumulators have a different row layout (set_accumulator_rows first)")

    def clear_errors(self):
        """Clear the sticky per-replica device error flags (ECAPACITY / ELIMIT / EINTEGRITY)."""
        self._check(self._lib.isingmc_clear_errors(self._h))

    def count_diagonal_and_off(self):
        """QmcDebug::count_diagonal_and_off (qmc_debug.rs:50-52): (diagonal, off-diagonal) op counts per replica."""
        c = self._debug_counts()
        return c[:, 0], c[:, 1]

    def count_constant_ops(self):
        """QmcDebug::count_constant_ops (qmc_debug.rs:54-56)."""
        return self._debug_counts()[:, 2]

    def _debug_counts(self):
        out = np.zeros((self.nreplicas, 3), dtype=np.uint32)
        self._check(self._lib.isingmc_debug_counts(self._h, _ptr(out, C.c_uint32)))
        return out

    def verify(self):
        out = np.zeros(self.nreplicas, dtype=np.uint8)
        self._check(self._lib.isingmc_verify(self._h, _ptr(out, C.c_uint8)))
        return out.astype(bool)

    def set_stream(self, stream_ptr):
        self._check(self._lib.isingmc_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def synchronize(self):
        self._check(self._lib.isingmc_synchronize(self._h))

    def last_kernel_ms(self):
        ms, n = C.c_float(0), C.c_uint32(0)
        self._check(self._lib.isingmc_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_pass_ms(self):
        """(ms, launches) of the last run split by kernel: index 0 = diagonal-pass launches, 1 = all other launches."""
        ms, n = (C.c_float * 2)(), (C.c_uint32 * 2)()
        self._check(self._lib.isingmc_last_pass_ms(self._h, ms, n))
        return (ms[0], ms[1]), (n[0], n[1])

    def last_rvb_ms(self):
        """(ms, launches) of the RVB-sweep launches of the last run (part of last_pass_ms()[0][1])."""
        ms, n = C.c_float(), C.c_uint32()
        self._check(self._lib.isingmc_last_rvb_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def launch_info(self):
        out = (C.c_uint32 * 8)()
        self._check(self._lib.isingmc_get_launch_info(self._h, out))
        return dict(waves_per_replica=out[0], lds_bytes=out[1], lds_uf_ids=out[2], state_words=out[3],
                    slots_per_lane=out[4], lds_edge_table=bool(out[5]), split_launches=bool(out[6] & 1), global_tables=bool(out[6] & 2), fast_diagonal=bool(out[6] & 4), fast_label=bool(out[6] & 8), compact_list=bool(out[6] & 16), lean_cluster=bool(out[6] & 32), rvb_split=bool(out[6] & 64), rvb_main_waves=(out[6] >> 16) & 0xFF,
                    waves_offdiag=(out[6] >> 8) & 0xFF,
                    lds_bytes_diagonal=out[7])

    def debug_phase_ticks(self, reset=True):
        out = np.zeros((self.nreplicas, 16), dtype=np.uint64)
        self._check(self._lib.isingmc_debug_phase_ticks(self._h, _ptr(out, C.c_uint64), 1 if reset else 0))
        return out

    def set_steps_per_launch(self, steps):
        self._check(self._lib.isingmc_set_steps_per_launch(self._h, int(steps)))

    def into_qmc(self, do_loop_updates=False):
        """IntoQmc::into_qmc (qmc_ising.rs:943-976): same container driven through Qmc::timestep."""
        q = Qmc.__new__(Qmc)
        q.__dict__ = dict(self.__dict__)  # its own dict: the handle moves to q, this object is left empty
        q._flags = (self._flags & FLAG_HEATBATH) | (FLAG_LOOP if do_loop_updates else 0)
        self._h = None
        return q


class Qmc(QmcIsingGraph):
    """qmc::sse::Qmc (src/sse/qmc_runner.rs) restricted to Ising interactions: timestep =
    diagonal -> [directed loop] -> cluster -> free spins (qmc_runner.rs:363-377)."""

    def set_do_loop_updates(self, do_loop_updates):
        self._flags = (self._flags | FLAG_LOOP) if do_loop_updates else (self._flags & ~FLAG_LOOP)

    def should_do_loop_update(self):
        return bool(self._flags & FLAG_LOOP)

    def set_do_heatbath(self, do_heatbath):
        self.set_enable_heatbath(do_heatbath)

    def diagonal_update(self, beta):
        self.single_diagonal_step(beta)

    def cluster_update(self):
        return self.single_cluster_step(flip_free=False)

    def flip_free_bits(self):
        self.flip_free_spins()


    # ---- generic interactions (qmc_runner.rs:94-156, Interaction :415-680) ----
    @staticmethod
    def interaction_and_offset(mat):
        """Interaction::new_offset (qmc_runner.rs:489-502): shift the diagonal so that its minimum is 0; returns the shifted
        matrix and the amount to subtract from the energy offset (Qmc::make_interaction_and_offset, :108-118)."""
        m = np.array(mat, dtype=np.float64).copy()
        tn = int(round(np.sqrt(len(m))))
        d = np.arange(tn) * (tn + 1)
        mn = float(m[d].min())
        m[d] -= mn
        return m, mn

    @classmethod
    def from_interactions(cls, nvars, interactions, cutoff, seed, energy_offset=0.0, state=None, nreplicas=1, capacity=None,
                          replica_offset=0, device=-1, waves_per_replica=0, slots_per_lane=0, do_loop_updates=True,
                          do_cluster_updates=None):
        """qmc::sse::Qmc with arbitrary one- and two-variable interactions: `interactions` = [(mat, vars), ...] with
        mat the reference's 4^k weight matrix (index = outputs then inputs, first variable most significant) and all
        entries >= 0.  timestep = diagonal update -> directed loop (if enabled) -> cluster update -> free spins
        (qmc_runner.rs:363-377); the cluster update runs by default exactly when the reference's would: the model has
        cluster edges (constant one-variable interactions) and does not break the Ising symmetry (:260-270)."""
        self = cls.__new__(cls)
        lib = load_library()
        self._lib, self._h = lib, None
        self.nvars, self.nreplicas = int(nvars), int(nreplicas)
        arr = (_Interaction * len(interactions))()
        keep = []
        for i, (mat, vs) in enumerate(interactions):
            m = np.ascontiguousarray(np.asarray(mat, dtype=np.float64))
            vs = [int(v) for v in vs]
            if len(vs) > 2:
                raise IsingMcError(-5, "interactions on more than two variables are not implemented")
            if len(vs) not in (1, 2) or len(m) not in (4 ** len(vs), 2 ** len(vs)):
                raise IsingMcError(-1, "interaction matrices must have 4^k (full) or 2^k (diagonal) entries for k = 1 or 2 variables")
            keep.append(m)
            arr[i].nvars = len(vs)
            arr[i].vars[0] = vs[0]
            arr[i].vars[1] = vs[1] if len(vs) == 2 else 0
            arr[i].diagonal_only = 1 if len(m) == 2 ** len(vs) else 0
            arr[i].mat = m.ctypes.data_as(C.POINTER(C.c_double))
        self.interactions = [(_full_matrix(m, len(v)), list(v)) for (m, v) in interactions]
        # host-side mirrors used by imaginary_time_fold: bond -> variables
        self.edges = np.zeros((0, 2), dtype=np.uint32)
        self.J = np.zeros(0)
        self.transverse = self.longitudinal = 0.0
        if capacity is None:
            capacity = max(int(cutoff), 64)
        init = None
        if state is not None:
            st = np.asarray(state, dtype=np.uint8)
            if st.ndim == 1:
                st = np.broadcast_to(st, (self.nreplicas, self.nvars))
            init = np.ascontiguousarray(st)
        cfg = _Config(struct_size=C.sizeof(_Config), nreplicas=self.nreplicas, nvars=self.nvars, nedges=0,
                      capacity=int(capacity), cutoff0=int(cutoff), seed=int(seed), replica_offset=int(replica_offset),
                      device=int(device), init_state=_ptr(init, C.c_uint8) if init is not None else None,
                      waves_per_replica=int(waves_per_replica), slots_per_lane=int(slots_per_lane),
                      interactions=C.cast(arr, C.c_void_p), ninteractions=len(interactions), energy_offset=float(energy_offset))
        h = C.c_void_p()
        rc = lib.isingmc_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise IsingMcError(rc, lib.isingmc_last_error(None).decode())
        self._h = h
        self.capacity = int(capacity)
        if do_cluster_updates is None:
            def sym(m, k):
                m = np.asarray(m); size = 4 ** k
                return all(abs(m[i] - m[(size - 1) ^ i]) < np.finfo(float).eps for i in range(size))
            full = self.interactions
            has_edges = any(len(v) == 1 and np.ptp(np.asarray(m)) < np.finfo(float).eps for m, v in full)
            do_cluster_updates = has_edges and all(sym(m, len(v)) for m, v in full)
        self._flags = (0 if do_cluster_updates else FLAG_NO_CLUSTER) | (FLAG_LOOP if do_loop_updates else 0)
        self._acc_rows = self.nreplicas
        return self


from .tempering import TemperingContainer, NativeTemperingContainer, pt_decide  # noqa: E402
