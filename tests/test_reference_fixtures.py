"""The fixtures the reference's own tests hold for the hot path, replayed through the CPU oracle, through the
C ABI's host helpers, and — marked gpu — through isingmc_import_ops and the HIP kernels, bit-exactly against the oracle.

Data: tests/golden/reference_fixtures.json (hand-built op-strings of tests/check_rvb_crash.rs:68-293 and
tests/check_loop_crash.rs:7-74, known answers of src/sse/qmc_runner.rs:785-959)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, "golden", "reference_fixtures.json")))


def word_of(op):
    """FastOp::diagonal / FastOp::offdiagonal(vars, bond, inputs, outputs, constant) -> operator word (include/sse_format.h)."""
    ib = sum(int(b) << k for k, b in enumerate(op["in"]))
    ob = sum(int(b) << k for k, b in enumerate(op["out"]))
    return ((op["bond"] + 1) << 4) | ib | (ob << 2)


def rvb_model_edges(fx):
    # EdgeNavigator of the fixture: bond_prefers_aligned -> J < 0 (src/lib.rs:29: J > 0 is antiferromagnetic), bond_mag -> |J|
    return [((e["vars"][0], e["vars"][1]), (-1.0 if e["prefers_aligned"] else 1.0) * e["mag"]) for e in fx["edges"]]


def loop_interactions(fx):
    """check_loop_crash.rs:21-28: weight 1 when inputs == outputs or inputs == reversed(outputs), else 0 (reference matrix layout)."""
    ints = []
    for it in fx["interactions"]:
        m = np.zeros(16)
        for i0 in range(2):
            for i1 in range(2):
                for o0 in range(2):
                    for o1 in range(2):
                        if (i0, i1) == (o0, o1) or (i0, i1) == (o1, o0):
                            m[(o0 << 3) | (o1 << 2) | (i0 << 1) | i1] = 1.0
        ints.append((m, tuple(it["vars"])))
    return ints


# ---------------------------------------------------------------- CPU: oracle + host helpers of the C ABI

@pytest.mark.parametrize("fx", FX["interaction_at"], ids=lambda f: f["ref"].split()[-1])
def test_interaction_at_known_answers(oracle, fx):
    import isingmontecarlo_amd as im
    for inputs, outputs, want in fx["cases"]:
        assert oracle.interaction_at(fx["mat"], inputs, outputs) == want
        assert im.interaction_at(fx["mat"], inputs, outputs) == want


@pytest.mark.parametrize("fx", FX["sym_under_ising"], ids=lambda f: f["ref"].split()[-1])
def test_sym_under_ising_known_answers(oracle, fx):
    import isingmontecarlo_amd as im
    assert oracle.interaction_sym_under_ising(fx["mat"], fx["n"]) == fx["expect"]
    assert im.interaction_sym_under_ising(fx["mat"], fx["n"]) == fx["expect"]


def test_diagonal_interaction_is_zero_off_the_diagonal(oracle):
    import isingmontecarlo_amd as im
    mat = [1.0, 6.0, 6.0, 1.0]  # InteractionType::Diagonal on two variables (qmc_runner.rs:594-610)
    for f in (oracle.interaction_at, im.interaction_at):
        assert f(mat, [0, 1], [0, 1]) == 6.0 and f(mat, [1, 1], [1, 1]) == 1.0
        assert f(mat, [0, 1], [1, 0]) == 0.0 and f(mat, [0, 0], [1, 1]) == 0.0


@pytest.mark.parametrize("fx", FX["rvb"], ids=lambda f: f["name"])
def test_rvb_fixture_through_the_oracle(oracle, fx):
    edges = rvb_model_edges(fx)
    m = oracle.Model(fx["nvars"], [list(e) for e, _ in edges], [j for _, j in edges], 1.0, 0.0)
    words = [word_of(o) for o in fx["ops"]]
    for seed in range(8):
        r = oracle.Replica(m, 64, len(words), seed, 0, fx["state"])
        r.set_ops(words)
        assert r.verify()
        for _ in range(fx["calls"]):
            r.rvb_update(fx["updates_per_call"])
            if fx["verify_every_call"]:
                assert r.verify()
        assert r.verify()
        assert r.n == len(words)


@pytest.mark.parametrize("fx", FX["loop"], ids=lambda f: f["name"])
def test_loop_fixture_through_the_oracle(oracle, fx):
    m = oracle.Model.generic(fx["nvars"], loop_interactions(fx))
    words = [word_of(o) for o in fx["ops"]]
    changed = False
    for seed in range(8):
        r = oracle.Replica(m, 64, len(words), seed, 0, fx["state"])
        r.set_ops(words)
        assert r.verify()
        for _ in range(fx["calls"]):
            r.loop_update()
        assert r.verify()
        changed |= not np.array_equal(r.ops(), np.array(words, dtype=np.uint32)) or r.state().any()
    assert changed  # with XX weights the loop does move through the two-site vertices


def test_verify_rejects_corrupted_strings(oracle):
    """Verify::verify (qmc_ising.rs:829-860) must say False on a broken string: a flipped input bit, a flipped p=0 spin."""
    import _lattices as lat
    edges = lat.one_d_periodic(6, -1.0)
    e, j = lat.split(edges)
    m = oracle.Model(6, e, j, 1.0, 0.0)
    r = oracle.Replica(m, 512, 6, 11, 0)
    r.timesteps(50, 2.0)
    assert r.verify() and r.n > 4
    words = r.ops().copy()
    p = int(np.flatnonzero(words)[1])
    bad = words.copy(); bad[p] ^= 1 | (1 << 2)  # in and out bit of the first variable: still diagonal, state no longer propagates
    r2 = oracle.Replica(m, 512, len(words), 11, 0, r.state())
    r2.set_ops(bad)
    assert not r2.verify()
    r3 = oracle.Replica(m, 512, len(words), 11, 0, r.state())
    r3.set_ops(words)
    assert r3.verify()
    st = r.state().copy(); st[int(np.flatnonzero(words)[0]) % 6] ^= 1
    touched = {e[(w >> 4) - 1][k] for w in words if w and (w >> 4) - 1 < len(e) for k in (0, 1)}
    v = min(touched)
    st = r.state().copy(); st[v] ^= 1
    r3.set_state(st)
    assert not r3.verify()


# ---------------------------------------------------------------- GPU: the same strings through isingmc_import_ops

@pytest.mark.gpu
@pytest.mark.parametrize("fx", FX["rvb"], ids=lambda f: f["name"])
def test_rvb_fixture_through_the_hip_path(oracle, fx):
    import isingmontecarlo_amd as im
    edges = rvb_model_edges(fx)
    R, seed = 6, 97531
    words = np.array([word_of(o) for o in fx["ops"]], dtype=np.uint32)
    g = im.QmcIsingGraph(edges, 1.0, 0.0, len(words), seed, state=fx["state"], nreplicas=R, capacity=64, nvars=fx["nvars"])
    m = oracle.Model(fx["nvars"], [list(e) for e, _ in edges], [j for _, j in edges], 1.0, 0.0)
    reps = [oracle.Replica(m, 64, len(words), seed, r, fx["state"]) for r in range(R)]
    for r in range(R):
        g.import_ops(words, r)
        reps[r].set_ops(words)
    assert g.verify().all()
    for call in range(fx["calls"]):
        succ, upd = g.single_rvb_sweep(fx["updates_per_call"])
        for r, rep in enumerate(reps):
            assert succ[r] == rep.rvb_update(upd), (fx["name"], call, r)
        if fx["verify_every_call"] or call % 10 == 9:
            assert g.verify().all()
    st, n = g.state_ref(), g.get_n()
    for r, rep in enumerate(reps):
        assert n[r] == rep.n == len(words)
        assert np.array_equal(st[r], rep.state())
        assert np.array_equal(g.export_ops(r), rep.ops())
        assert rep.verify()
    assert g.verify().all()


@pytest.mark.gpu
@pytest.mark.parametrize("fx", FX["loop"], ids=lambda f: f["name"])
def test_loop_fixture_through_the_hip_path(oracle, fx):
    import isingmontecarlo_amd as im
    ints = loop_interactions(fx)
    R, seed = 6, 86420
    words = np.array([word_of(o) for o in fx["ops"]], dtype=np.uint32)
    g = im.Qmc.from_interactions(fx["nvars"], ints, len(words), seed, state=fx["state"], nreplicas=R, capacity=64)
    m = oracle.Model.generic(fx["nvars"], ints)
    reps = [oracle.Replica(m, 64, len(words), seed, r, fx["state"]) for r in range(R)]
    for r in range(R):
        g.import_ops(words, r)
        reps[r].set_ops(words)
    for call in range(fx["calls"]):
        lens = g.loop_update()
        for r, rep in enumerate(reps):
            assert lens[r] == rep.loop_update(), (fx["name"], call, r)
    st = g.state_ref()
    for r, rep in enumerate(reps):
        assert np.array_equal(st[r], rep.state())
        assert np.array_equal(g.export_ops(r), rep.ops())
    assert g.verify().all() and all(rep.verify() for rep in reps)


@pytest.mark.gpu
def test_verify_rejects_corrupted_strings_on_the_device(oracle):
    """isingmc_verify and ora_verify must both say False on the same broken strings and True on the intact one."""
    import isingmontecarlo_amd as im
    import _lattices as lat
    edges = lat.one_d_periodic(6, -1.0)
    e, j = lat.split(edges)
    g = im.QmcIsingGraph(edges, 1.0, 0.0, 6, 11, nreplicas=4, capacity=512)
    g.run(50, 2.0)
    assert g.verify().all()
    m = oracle.Model(6, e, j, 1.0, 0.0)
    words = g.export_ops(0)
    st0 = g.state_ref()[0]
    occ = np.flatnonzero(words)
    # replica 1: a diagonal op whose bits no longer match the propagated state; replica 2: a flipped p=0 spin on a
    # variable that carries ops; replica 3: an op dropped without telling the counters... (import recounts, so: intact)
    bad = words.copy(); bad[occ[1]] ^= 1 | (1 << 2)
    g.import_ops(words, 1); g.set_state(st0, 1); g.import_ops(bad, 1)
    g.import_ops(words, 2)
    touched = {e[(int(w) >> 4) - 1][k] for w in words if w and (int(w) >> 4) - 1 < len(e) for k in (0, 1)}
    st = st0.copy(); st[min(touched)] ^= 1
    g.set_state(st, 2)
    g.import_ops(words, 3); g.set_state(st0, 3)
    ok = g.verify()
    assert ok[0] and not ok[1] and not ok[2] and ok[3], ok
    for words_r, st_r, want in ((words, st0, True), (bad, st0, False), (words, st, False)):
        rep = oracle.Replica(m, 512, len(words_r), 11, 0, st_r)
        rep.set_ops(words_r)
        assert rep.verify() == want


def cluster_fixture_words(fx, nedges):
    """The fixture's constant one-variable ops on the transverse-field bond of their variable (bond E + v); see cluster_note."""
    return np.array([((nedges + o["vars"][0] + 1) << 4) | int(o["in"][0]) | (int(o["out"][0]) << 2) for o in fx["ops"]], dtype=np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("lean", [True, False], ids=["dedicated_kernel", "general_kernel"])
@pytest.mark.parametrize("fx", FX["cluster"], ids=lambda f: f["name"])
def test_cluster_fixture_through_the_hip_path(oracle, fx, lean):
    """tests/cluster_test.rs:6-75 through isingmc_import_ops + isingmc_cluster_update, against the oracle: cluster counts, the
    flipped strings and states, for many seeds (the reference runs it with thread_rng and only expects no panic)."""
    import isingmontecarlo_amd as im
    nv = max(2, fx["nvars"])  # (a model needs an edge for its variable count here; the extra variable carries no op)
    edges = [((0, 1), 1.0)]
    R, seed = 16, 424242
    words = cluster_fixture_words(fx, len(edges))
    state = list(fx["state"]) + [0] * (nv - fx["nvars"])
    g = im.QmcIsingGraph(edges, 1.0, 0.0, len(words), seed, state=state, nreplicas=R, capacity=64, nvars=nv,
                         cfg_flags=0 if lean else im.CFG_NO_LEAN_CLUSTER)
    m = oracle.Model(nv, [[0, 1]], [1.0], 1.0, 0.0)
    reps = [oracle.Replica(m, 64, len(words), seed, r, state) for r in range(R)]
    for r in range(R):
        g.import_ops(words, r)
        reps[r].set_ops(words)
    assert g.verify().all()
    flipped = 0
    for call in range(6):
        nc = g.single_cluster_step(flip_free=False)
        for r, rep in enumerate(reps):
            assert rep.cluster_update(0.5) == fx["expect_clusters"] == nc[r], (fx["name"], call, r)
        st = g.state_ref()
        for r, rep in enumerate(reps):
            assert np.array_equal(st[r], rep.state()) and np.array_equal(g.export_ops(r), rep.ops())
            flipped += int(not np.array_equal(g.export_ops(r)[:len(words)], words))
        assert g.verify().all()
    assert flipped > 0  # the coins do flip segments
