"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/isingmc_hip.h declares, and fails loudly (no CPU fallback) when no HIP device exists."""
import ctypes as C
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
    # 4 u32, 2 pointers, 2 doubles, 2 u32, u64, u32, i32, pointer, 5 u32 (+4 padding), pointer, u32 (+4), double -> 128 bytes on LP64
    assert C.sizeof(im._Config) == 128 and C.sizeof(im._Interaction) == 24 and im._Interaction.mat.offset == 16
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
