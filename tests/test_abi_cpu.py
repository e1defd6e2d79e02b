"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/isingmc_hip.h declares, and fails loudly (no CPU fallback) when no HIP device exists."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "isingmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(isingmc_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    import isingmontecarlo_amd as im
    lib = im.load_library()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libisingmc_hip.so does not export {n}"
        assert n in im.SYMBOLS, f"python binding misses {n}"
    assert set(im.SYMBOLS) == set(names)


def test_config_struct_layout_matches_header():
    import isingmontecarlo_amd as im
    # 4 u32, 2 pointers, 2 doubles, 2 u32, u64, u32, i32, pointer, 5 u32 (+4 padding), pointer, u32 (+4), double, 2 pointers -> 144 bytes on LP64
    assert C.sizeof(im._Config) == 144 and C.sizeof(im._Interaction) == 24 and im._Interaction.mat.offset == 16
    assert im._Config.transverse_r.offset == 128 and im._Config.longitudinal_r.offset == 136
    assert im._Config.seed.offset == 56 and im._Config.init_state.offset == 72


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import isingmontecarlo_amd as im
    with pytest.raises(im.IsingMcError) as ei:
        im.QmcIsingGraph([((0, 1), 1.0)], 1.0, 0.0, 4, 1)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_create_rejects_bad_arguments():
    import isingmontecarlo_amd as im
    lib = im.load_library()
    h = C.c_void_p()
    cfg = im._Config(struct_size=4)  # wrong size
    assert lib.isingmc_create(C.byref(cfg), C.byref(h)) == -1
    assert b"struct_size" in lib.isingmc_last_error(None)
    ed = np.array([0, 5], dtype=np.uint32); js = np.array([1.0])
    cfg = im._Config(struct_size=C.sizeof(im._Config), nreplicas=1, nvars=2, nedges=1,
                     edges=ed.ctypes.data_as(C.POINTER(C.c_uint32)), J=js.ctypes.data_as(C.POINTER(C.c_double)),
                     transverse=1.0, capacity=8, cutoff0=4, device=-1)
    assert lib.isingmc_create(C.byref(cfg), C.byref(h)) == -1  # edge endpoint out of range
    cfg.nvars = 6; cfg.cutoff0 = 9
    assert lib.isingmc_create(C.byref(cfg), C.byref(h)) == -1  # cutoff0 > capacity


def test_create_rejects_bad_generic_interactions():
    """Argument checks of the generic-interaction path run before any device is touched."""
    import isingmontecarlo_amd as im
    with pytest.raises(im.IsingMcError) as ei:  # negative weight (Interaction::new, qmc_runner.rs:511-513)
        im.Qmc.from_interactions(2, [(np.array([1.0, -0.5, 0, 1.0]), (0,))], 4, 1)
    assert ei.value.code == -1 and ">= 0" in str(ei.value)
    with pytest.raises(im.IsingMcError):  # variable outside the model
        im.Qmc.from_interactions(2, [(np.ones(16), (0, 2))], 4, 1)
    with pytest.raises(im.IsingMcError):  # the same variable twice
        im.Qmc.from_interactions(2, [(np.ones(16), (1, 1))], 4, 1)
    with pytest.raises(im.IsingMcError):  # wrong matrix size for the number of variables
        im.Qmc.from_interactions(2, [(np.ones(4), (0, 1))], 4, 1)
    m, off = im.Qmc.interaction_and_offset([3.0, 1.0, 1.0, 2.0])  # Interaction::new_offset: diagonal minimum removed
    assert off == 2.0 and np.allclose(m, [1.0, 1.0, 1.0, 0.0])


def test_create_argument_checks_that_need_no_device():
    """capacity == 0 (would divide by zero in the chunk grid) and interactions on more than two variables."""
    import isingmontecarlo_amd as im
    lib = im.load_library()
    h = C.c_void_p()
    ed = np.array([0, 1], dtype=np.uint32); js = np.array([1.0])
    cfg = im._Config(struct_size=C.sizeof(im._Config), nreplicas=1, nvars=2, nedges=1,
                     edges=ed.ctypes.data_as(C.POINTER(C.c_uint32)), J=js.ctypes.data_as(C.POINTER(C.c_double)),
                     transverse=1.0, capacity=0, cutoff0=0, device=-1)
    assert lib.isingmc_create(C.byref(cfg), C.byref(h)) == -1 and b"capacity" in lib.isingmc_last_error(None)
    mat = np.ones(64)
    it = im._Interaction(nvars=3, mat=mat.ctypes.data_as(C.POINTER(C.c_double)))
    cfg = im._Config(struct_size=C.sizeof(im._Config), nreplicas=1, nvars=3, capacity=8, cutoff0=4, device=-1,
                     interactions=C.cast(C.pointer(it), C.c_void_p), ninteractions=1)
    assert lib.isingmc_create(C.byref(cfg), C.byref(h)) == -5  # ENOTIMPL, qmc_runner.rs:415-680 allows any k
    with pytest.raises(im.IsingMcError) as ei:
        im.Qmc.from_interactions(3, [(np.ones(64), (0, 1, 2))], 4, 1)
    assert ei.value.code == -5
    assert lib.isingmc_clear_errors(None) == -1 and lib.isingmc_set_accumulators(None, None) == -1


def test_into_qmc_moves_the_handle_without_a_device():
    """Advisor finding r1: into_qmc used to share one __dict__ and null the handle of BOTH objects."""
    import isingmontecarlo_amd as im
    g = im.QmcIsingGraph.__new__(im.QmcIsingGraph)
    g._lib, g._h, g._flags, g.nreplicas = None, 12345, im.FLAG_HEATBATH, 1
    q = g.into_qmc(do_loop_updates=True)
    assert q._h == 12345 and g._h is None and q._flags == (im.FLAG_HEATBATH | im.FLAG_LOOP)
    q._h = None  # nothing real to destroy


def test_null_handle_calls_do_not_crash():
    import isingmontecarlo_amd as im
    lib = im.load_library()
    assert lib.isingmc_timesteps(None, 1, None, 1, 0) == -1
    assert lib.isingmc_get_n(None, None) == -1
    assert lib.isingmc_num_bonds(None) == 0
    # the tempering entry points of round 3 (device-side decisions, sweeps at the labels' temperatures)
    import ctypes as C
    on = C.c_int(7)
    assert lib.isingmc_pt_timesteps(None, 1, 1, 0) == -1
    assert lib.isingmc_pt_set_device_decisions(None, 1) == -1
    assert lib.isingmc_pt_get_device_decisions(None, C.byref(on)) == -1 and on.value == 7
    assert lib.isingmc_pt_step(None, None) == -1
    lib.isingmc_destroy(None)


def test_op_word_helpers_round_trip():
    import isingmontecarlo_amd as im
    for bond in (0, 1, 4095, (1 << 28) - 3):
        for i in range(4):
            for o in range(4):
                w = im.op_make(bond, i, o)
                assert w != 0 and im.op_fields(w) == (bond, i, o)
    assert im.op_fields(0) is None


def test_lattice_builders_match_reference_ordering():
    import _lattices as lat
    e = lat.two_d_periodic(4)
    # benches/end_to_end.rs:12-30: 16 right bonds (J=-1) then 16 down bonds (+1 on even columns)
    assert len(e) == 32 and all(j == -1.0 for _, j in e[:16])
    assert e[0][0] == (0, 1) and e[16][0] == (0, 4) and e[16][1] == 1.0
    (a, b), j = e[16 + 4]  # i=1, j=0 -> odd column
    assert (a, b) == (1, 5) and j == -1.0
    assert lat.one_d_periodic(16)[-1] == ((15, 0), 1.0)


def test_fft_autocorrelation_matches_direct_sum():
    """fft_autocorrelation (autocorrelations.rs:99-133) against the defining circular sum."""
    from isingmontecarlo_amd.autocorrelations import fft_autocorrelation
    rng = np.random.default_rng(3)
    x = rng.normal(size=(37, 5)) + np.linspace(0, 2, 37)[:, None]
    got = fft_autocorrelation(x)
    y = x - x.mean(axis=0)
    y = y / np.sqrt((y * y).sum(axis=0))
    want = np.array([sum((y[:, i] * np.roll(y[:, i], -t)).sum() for i in range(5)) / 5 for t in range(37)])
    assert np.allclose(got, want, atol=1e-12) and abs(got[0] - 1.0) < 1e-12


def test_header_is_plain_c_and_matches_the_python_mirror(tmp_path):
    """include/isingmc_hip.h is the drop-in boundary: it must compile as C99 on its own and agree with the ctypes mirror."""
    import os, subprocess
    import isingmontecarlo_amd as im
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "isingmc_hip.h"\n#include <stdio.h>\nint main(void){ printf("%zu %zu\\n", sizeof(isingmc_config), sizeof(isingmc_interaction)); return 0; }\n')
    exe = str(tmp_path / "abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"), str(src), "-o", exe])
    a, b = (int(x) for x in subprocess.check_output([exe], text=True).split())
    assert a == C.sizeof(im._Config) and b == C.sizeof(im._Interaction)


def test_serde_dict_follows_the_reference_struct_definitions():
    """The dict of serde_format.to_serde walked against a schema written by hand from the Rust struct definitions (field names,
    nesting, Option / tuple / enum shapes as serde_json writes them): SerializeQmcGraph qmc_ising.rs:1010-1028, FastOpsTemplate
    fast_ops.rs:35-49, FastOpNodeTemplate :181-190, BasicOp op_container.rs:224-237, OpType :165-173, PRel directed_loop.rs:12-17,
    DefaultFastOpAllocator fast_op_alloc.rs:29-39, Allocator util/allocator.rs:31-38 (instances as its length).  No GPU: the
    graph is a stand-in that hands out a hand-built op-string."""
    import isingmontecarlo_amd as im
    from isingmontecarlo_amd.serde_format import to_serde

    class Stub:  # what to_serde asks of a graph
        nvars = 3
        edges = np.array([[0, 1], [1, 2]], dtype=np.uint32)
        J = np.array([1.0, -0.5])
        transverse, longitudinal = 0.7, 0.2
        _flags = 8  # run_rvb
        transverse_r = longitudinal_r = None
        def export_ops(self, r):  # two-site diagonal, empty, transverse off-diagonal on var 1, longitudinal diagonal on var 2
            return np.array([im.op_make(0, 0b10, 0b10), 0, im.op_make(2 + 1, 0, 1), im.op_make(2 + 3 + 2, 1, 1)], dtype=np.uint32)
        def state_ref(self):
            return np.array([[0, 1, 1]], dtype=np.uint8)
        def get_offsets(self):
            return np.array([4.2])

    d = to_serde(Stub(), 0, total_rvb_successes=5, rvb_clusters_counted=9)
    usize = lambda x: isinstance(x, int) and not isinstance(x, bool) and x >= 0
    f64 = lambda x: isinstance(x, float)
    boolean = lambda x: isinstance(x, bool)
    opt = lambda f: (lambda x: x is None or f(x))
    vec = lambda f: (lambda x: isinstance(x, list) and all(f(y) for y in x))
    tup = lambda *fs: (lambda x: isinstance(x, list) and len(x) == len(fs) and all(f(y) for f, y in zip(fs, x)))
    def struct(**fields):
        def check(x):
            assert isinstance(x, dict) and list(x.keys()) == list(fields.keys()), (list(x.keys()) if isinstance(x, dict) else x, list(fields.keys()))
            for k, f in fields.items():
                assert f(x[k]), (k, x[k])
            return True
        return check
    prel = struct(p=usize, relv=usize)
    optype = lambda x: isinstance(x, dict) and len(x) == 1 and (("Diagonal" in x and vec(boolean)(x["Diagonal"])) or
                                                                ("Offdiagonal" in x and tup(vec(boolean), vec(boolean))(x["Offdiagonal"])))
    basic_op = struct(vars=vec(usize), bond=usize, in_out=optype, constant=boolean)
    node = struct(op=basic_op, previous_p=opt(usize), next_p=opt(usize), previous_for_vars=vec(opt(prel)), next_for_vars=vec(opt(prel)))
    allocator = struct(instances=usize, gen_more=boolean)
    alloc = struct(usize_alloc=allocator, bool_alloc=allocator, opside_alloc=allocator, leg_alloc=allocator, option_usize_alloc=allocator,
                   f64_alloc=allocator, bond_container_alloc=allocator, bond_container_varpos_alloc=allocator, binary_heap_alloc=allocator)
    manager = struct(ops=vec(opt(node)), n=usize, p_ends=opt(tup(usize, usize)), var_ends=vec(opt(tup(prel, prel))),
                     bond_counters=vec(usize), alloc=alloc)
    graph = struct(edges=vec(tup(vec(usize), f64)), transverse=f64, longitudinal=f64, state=opt(vec(boolean)), cutoff=usize,
                   op_manager=opt(manager), total_energy_offset=f64, nvars=usize, run_rvb_steps=boolean,
                   classical_bonds=opt(vec(vec(usize))), total_rvb_successes=usize, rvb_clusters_counted=usize,
                   bond_weights=opt(struct(max_weight_and_cumulative=vec(tup(usize, f64, f64)))))
    assert graph(json.loads(json.dumps(d)))
    m = d["op_manager"]
    assert m["n"] == 3 and m["p_ends"] == [0, 3] and m["ops"][1] is None and m["bond_counters"] == [1, 0, 0, 1, 0, 0, 0, 1]
    assert m["ops"][0]["op"] == {"vars": [0, 1], "bond": 0, "in_out": {"Diagonal": [False, True]}, "constant": False}
    assert m["ops"][2]["op"] == {"vars": [1], "bond": 3, "in_out": {"Offdiagonal": [[False], [True]]}, "constant": True}
    assert m["ops"][2]["previous_for_vars"] == [{"p": 0, "relv": 1}] and m["ops"][0]["next_for_vars"] == [None, {"p": 2, "relv": 0}]
    assert m["var_ends"] == [[{"p": 0, "relv": 0}, {"p": 0, "relv": 0}], [{"p": 0, "relv": 1}, {"p": 2, "relv": 0}], [{"p": 3, "relv": 0}, {"p": 3, "relv": 0}]]
    assert m["alloc"]["usize_alloc"] == {"instances": 10, "gen_more": False} and m["alloc"]["binary_heap_alloc"]["instances"] == 1
    assert d["run_rvb_steps"] and d["classical_bonds"] == [[0], [0, 1], [1]] and d["total_rvb_successes"] == 5 and d["rvb_clusters_counted"] == 9


def test_row_stride_covers_every_whole_tile_access():
    """Bounds audit of the op-string rows (isingmc_plan_geometry, the function isingmc_create itself uses): for every launch geometry
    a batch can run with, the largest slot index any kernel forms stays inside the row.
      * diagonal / cached-apply passes stream whole tiles of W * 64 * K slots up to the cutoff (<= capacity);
      * a cluster-scan wave owns ceil(used / W) chunks and loads / stores wave-tiles of 64 * K slots from its range start — also
        when its range is empty (range start = end of the chunk-rounded string): that prefetch needs the + 256;
      * the dedicated cluster kernel keeps one 16-bit id per slot, two rows per dword, in the first half of a scratch row.
    (Round 2 saw a GPU memory fault in an uncommitted experiment on the 16-wave HBM-table path; the shipped index arithmetic
    is checked here for all geometries instead of trusting one passing run.)"""
    import isingmontecarlo_amd as im
    lib = im.load_library()
    out = (C.c_uint32 * 4)()
    rng = np.random.default_rng(5)
    caps = [1, 2, 63, 64, 255, 256, 257, 1023, 1024, 4095, 4096, 4097, 8192, 32768 + 1, 1 << 18, (1 << 18) + 777, 1 << 20, 3_000_001] + [int(x) for x in rng.integers(1, 1 << 22, 60)]
    for cap in caps:
        for W, Wmax in [(1, 1), (4, 4), (4, 8), (4, 16), (6, 6), (6, 16), (8, 8), (8, 16), (16, 16), (1, 16)]:
            for K in (1, 2, 4):
                assert lib.isingmc_plan_geometry(cap, W, K, Wmax, out) == 0
                CH, nchunks, stride, tile = (int(x) for x in out)
                assert CH % 256 == 0 and nchunks <= 128 and CH * nchunks >= cap
                for Wk in {W, Wmax, 8 if Wmax >= 8 and W < 8 else W}:  # (8 waves: the HBM union-find geometry of run())
                    if Wmax % Wk and Wk != W:
                        continue
                    ts = Wk * 64 * K
                    assert stride % ts == 0, (cap, W, Wmax, K, Wk)
                    # whole-tile streaming up to any cutoff M <= cap: last slot touched = roundup(M, ts) - 1
                    assert (cap + ts - 1) // ts * ts <= stride
                    # cluster scan: wave ranges in chunks, wave-tiles of 64 K slots from the range start, prefetch even when empty
                    for M in {cap, max(1, cap // 2), max(1, cap - 1)}:
                        used = (M + CH - 1) // CH
                        q = (used + Wk - 1) // Wk
                        for w in range(Wk):
                            c0 = min(w * q, used); c1 = min(c0 + q, used)
                            pbeg, pend = c0 * CH, min(c1 * CH, M)
                            last = pbeg + 64 * K - 1  # the unconditional prefetch at the range start
                            if pend > pbeg:
                                last = max(last, pbeg + ((pend - pbeg + 64 * K - 1) // (64 * K)) * 64 * K - 1)
                            assert last < stride, (cap, W, Wmax, K, Wk, M, w, last, stride)
                            assert (last >> 1) + 64 < stride  # 16-bit ids of the dedicated cluster kernel (K >= 2): dword (p0 >> 1) + 32 j + lane
