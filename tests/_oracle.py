"""ctypes binding of the CPU oracle (oracle/libsse_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(_ORACLE_DIR, "libsse_oracle.so")

FLAG_LOOP, FLAG_NO_CLUSTER, FLAG_HEATBATH, FLAG_RVB = 1, 2, 4, 8


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    u32, u64, f64, vp = C.c_uint32, C.c_uint64, C.c_double, C.c_void_p
    p = C.POINTER
    sig = {
        "ora_model_create": (vp, [u32, u32, p(u32), p(u32), p(f64), f64, f64]),
        "ora_model_destroy": (None, [vp]),
        "ora_model_nbonds": (u32, [vp]),
        "ora_model_offset": (f64, [vp]),
        "ora_replica_create": (vp, [vp, u32, u32, u64, u32, p(C.c_uint8)]),
        "ora_replica_destroy": (None, [vp]),
        "ora_diagonal_update": (None, [vp, f64]),
        "ora_heatbath_update": (None, [vp, f64]),
        "ora_cluster_update": (u32, [vp, f64]),
        "ora_flip_free_spins": (None, [vp]),
        "ora_loop_update": (u32, [vp]),
        "ora_timestep": (C.c_int, [vp, f64, u32]),
        "ora_timesteps": (C.c_int, [vp, u64, f64, u32, u32]),
        "ora_model_create_generic": (vp, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_double), C.c_double]),
        "ora_verify": (C.c_int, [vp]),
        "ora_itime_magnetization": (None, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "ora_get_n": (u32, [vp]),
        "ora_get_cutoff": (u32, [vp]),
        "ora_set_cutoff": (C.c_int, [vp, u32]),
        "ora_get_epoch": (u64, [vp]),
        "ora_set_epoch": (None, [vp, u64]),
        "ora_get_state": (None, [vp, p(C.c_uint8)]),
        "ora_set_state": (None, [vp, p(C.c_uint8)]),
        "ora_get_ops": (None, [vp, p(u32)]),
        "ora_set_ops": (C.c_int, [vp, p(u32), u32]),
        "ora_get_bond_count": (u32, [vp, u32]),
        "ora_get_accumulators": (None, [vp, p(u64)]),
        "ora_reset_accumulators": (None, [vp]),
        "ora_batch_timesteps": (C.c_int, [p(vp), u32, u64, p(f64), u32, u32, C.c_int]),
        "ora_max_threads": (C.c_int, []),
        "ora_philox4x32_10": (None, [p(u32), p(u32), p(u32)]),
        "ora_pt_step": (u64, [p(vp), p(f64), u32, u64, u32, u64]),
        "ora_rvb_update": (u32, [vp, u32]),
        "ora_find_overlapping_starts": (u32, [u32, u32, u32, p(u32), u32, p(u32)]),
        "ora_remove_doubles": (u32, [p(u32), u32]),
        "ora_interaction_at": (f64, [u32, u32, p(f64), p(C.c_uint8), p(C.c_uint8)]),
        "ora_interaction_sym_under_ising": (C.c_int, [u32, u32, p(f64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


LIB = None


def lib():
    global LIB
    if LIB is None:
        LIB = load()
    return LIB


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def op_make(bond, in_bits, out_bits):
    return ((bond + 1) << 4) | (in_bits & 3) | ((out_bits & 3) << 2)


def interaction_at(mat, inputs, outputs):
    m = np.ascontiguousarray(np.asarray(mat, dtype=np.float64))
    k = len(inputs)
    i = np.ascontiguousarray(np.asarray(inputs, dtype=np.uint8)); o = np.ascontiguousarray(np.asarray(outputs, dtype=np.uint8))
    return lib().ora_interaction_at(k, 1 if len(m) == 2 ** k else 0, _ptr(m, C.c_double), _ptr(i, C.c_uint8), _ptr(o, C.c_uint8))


def interaction_sym_under_ising(mat, k):
    m = np.ascontiguousarray(np.asarray(mat, dtype=np.float64))
    return bool(lib().ora_interaction_sym_under_ising(k, 1 if len(m) == 2 ** k else 0, _ptr(m, C.c_double)))


class Model:
    def __init__(self, nvars, edges, J, gamma, h=0.0):
        self.nvars = int(nvars)
        self.edges = np.ascontiguousarray(np.asarray(edges, dtype=np.uint32).reshape(-1, 2))
        self.J = np.ascontiguousarray(np.asarray(J, dtype=np.float64))
        self.gamma, self.h = float(gamma), float(h)
        ea = np.ascontiguousarray(self.edges[:, 0])
        eb = np.ascontiguousarray(self.edges[:, 1])
        self.ptr = lib().ora_model_create(self.nvars, len(self.J), _ptr(ea, C.c_uint32), _ptr(eb, C.c_uint32),
                                          _ptr(self.J, C.c_double), self.gamma, self.h)
        self.nbonds = lib().ora_model_nbonds(self.ptr)
        self.offset = lib().ora_model_offset(self.ptr)

    @classmethod
    def generic(cls, nvars, interactions, offset=0.0):
        """Arbitrary 1-/2-variable interactions [(mat, vars), ...] with the reference's matrix layout (index = outputs then
        inputs, first variable most significant, qmc_runner.rs:666-679); converted here to the oracle's in | out<<2."""
        self = cls.__new__(cls)
        self.nvars = int(nvars)
        nb = len(interactions)
        ks = np.zeros(nb, dtype=np.uint32); va = np.zeros(nb, dtype=np.uint32); vb = np.zeros(nb, dtype=np.uint32)
        mats = np.zeros((nb, 16), dtype=np.float64)
        for i, (mat, vs) in enumerate(interactions):
            mat = np.asarray(mat, dtype=np.float64)
            ks[i] = len(vs); va[i] = vs[0]; vb[i] = vs[1] if len(vs) == 2 else 0
            kk = len(vs)
            for i_ in range(1 << kk):  # oracle layout: bit 0 = first variable
                for o_ in range(1 << kk):
                    mats[i, i_ | (o_ << 2)] = interaction_at(mat, [(i_ >> j) & 1 for j in range(kk)], [(o_ >> j) & 1 for j in range(kk)])
        self.gamma = self.h = 0.0
        self.ptr = lib().ora_model_create_generic(self.nvars, nb, _ptr(ks, C.c_uint32), _ptr(va, C.c_uint32), _ptr(vb, C.c_uint32),
                                                  _ptr(np.ascontiguousarray(mats), C.c_double), float(offset))
        self.nbonds = nb
        self.offset = float(offset)
        return self

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().ora_model_destroy(self.ptr)
            self.ptr = None


class Replica:
    def __init__(self, model, capacity, cutoff0, seed, replica=0, init_state=None):
        self.model = model
        st = None
        if init_state is not None:
            arr = np.ascontiguousarray(np.asarray(init_state, dtype=np.uint8))
            st = _ptr(arr, C.c_uint8)
        self.ptr = lib().ora_replica_create(model.ptr, capacity, cutoff0, seed, replica, st)
        assert self.ptr, "oracle replica creation failed"

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().ora_replica_destroy(self.ptr)
            self.ptr = None

    def diagonal_update(self, beta):
        lib().ora_diagonal_update(self.ptr, beta)

    def heatbath_update(self, beta):
        lib().ora_heatbath_update(self.ptr, beta)

    def cluster_update(self, prob=0.5):
        return lib().ora_cluster_update(self.ptr, prob)

    def flip_free_spins(self):
        lib().ora_flip_free_spins(self.ptr)

    def loop_update(self):
        return lib().ora_loop_update(self.ptr)

    def rvb_update(self, updates):
        return lib().ora_rvb_update(self.ptr, updates)

    def timestep(self, beta, flags=0):
        return lib().ora_timestep(self.ptr, beta, flags)

    def timesteps(self, t, beta, freq=1, flags=0):
        rc = lib().ora_timesteps(self.ptr, t, beta, freq, flags)
        assert rc == 0, "oracle: cutoff exceeded capacity"

    def verify(self):
        return bool(lib().ora_verify(self.ptr))

    def itime_magnetization(self):
        a, b, c = C.c_int64(0), C.c_uint64(0), C.c_uint64(0)
        lib().ora_itime_magnetization(self.ptr, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    @property
    def n(self):
        return lib().ora_get_n(self.ptr)

    @property
    def cutoff(self):
        return lib().ora_get_cutoff(self.ptr)

    def set_cutoff(self, c):
        return lib().ora_set_cutoff(self.ptr, c)

    @property
    def epoch(self):
        return lib().ora_get_epoch(self.ptr)

    def set_epoch(self, e):
        lib().ora_set_epoch(self.ptr, int(e))

    def state(self):
        out = np.zeros(self.model.nvars, dtype=np.uint8)
        lib().ora_get_state(self.ptr, _ptr(out, C.c_uint8))
        return out

    def set_state(self, s):
        arr = np.ascontiguousarray(np.asarray(s, dtype=np.uint8))
        lib().ora_set_state(self.ptr, _ptr(arr, C.c_uint8))

    def ops(self):
        out = np.zeros(self.cutoff, dtype=np.uint32)
        lib().ora_get_ops(self.ptr, _ptr(out, C.c_uint32))
        return out

    def set_ops(self, words):
        arr = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
        rc = lib().ora_set_ops(self.ptr, _ptr(arr, C.c_uint32), len(arr))
        assert rc == 0

    def bond_count(self, b):
        return lib().ora_get_bond_count(self.ptr, b)

    def accumulators(self):
        out = np.zeros(8, dtype=np.uint64)
        lib().ora_get_accumulators(self.ptr, _ptr(out, C.c_uint64))
        return out

    def reset_accumulators(self):
        lib().ora_reset_accumulators(self.ptr)


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU boxes show every logical CPU
    of the host but grant a one-GPU job a quota of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def batch_timesteps(replicas, t, betas, freq=1, flags=0, nthreads=0):
    if nthreads == 0:
        nthreads = min(usable_cores(), len(replicas))
    arr = (C.c_void_p * len(replicas))(*[r.ptr for r in replicas])
    b = np.ascontiguousarray(np.asarray(betas, dtype=np.float64))
    rc = lib().ora_batch_timesteps(arr, len(replicas), t, _ptr(b, C.c_double), freq, flags, nthreads)
    assert rc == 0


def pt_step(by_slot, betas, seed, chain, step):
    """One tempering step of one chain; `by_slot` (list of Replica, index = temperature) is permuted in place."""
    arr = (C.c_void_p * len(by_slot))(*[r.ptr for r in by_slot])
    b = np.ascontiguousarray(np.asarray(betas, dtype=np.float64))
    swaps = lib().ora_pt_step(arr, _ptr(b, C.c_double), len(by_slot), seed, chain, step)
    lookup = {r.ptr: r for r in by_slot}
    by_slot[:] = [lookup[arr[i]] for i in range(len(by_slot))]
    return int(swaps)


def find_overlapping_starts(p_start, p_end, cutoff, flips):
    fp = np.ascontiguousarray(np.asarray(flips, dtype=np.uint32))
    out = np.zeros(len(fp), dtype=np.uint32)
    n = lib().ora_find_overlapping_starts(p_start, p_end, cutoff, _ptr(fp, C.c_uint32), len(fp), _ptr(out, C.c_uint32))
    return out[:n].tolist()


def remove_doubles(v):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.uint32))
    n = lib().ora_remove_doubles(_ptr(a, C.c_uint32), len(a)) if len(a) else 0
    return a[:n].tolist()
