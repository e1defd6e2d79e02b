"""GPU tests of the native tempering step (isingmc_pt_create / isingmc_pt_step): label swaps inside a temperature block,
configuration exchange between neighbouring ranks, relative Hamiltonian weights — against the oracle's graph-swapping
restatement of TemperingContainer::tempering_step (tempering_container.rs:121-149,241-302)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _lattices as lat

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

BETAS = np.array([0.5, 0.8, 1.1, 1.5, 2.0, 2.6])
K, SEED, STEPS, SWEEPS = 4, 2468, 15, 2


def oracle_reference(oracle):
    from test_tempering_cpu import reference_pt
    e, j = lat.split(lat.two_d_periodic(4))
    m = oracle.Model(16, e, j, 1.0, 0.0)
    return reference_pt(m, BETAS, K, SEED, 4096, 16, nsteps=STEPS, sweeps_per_step=SWEEPS)


@pytest.mark.parametrize("where", ["device", "device, nothing read back between steps", "host"])
def test_native_step_single_rank_matches_oracle(oracle, where):
    """A rank that owns all temperatures takes the swap decisions in a kernel (labels, betas per replica, accumulator rows and the
    swap count stay in device memory); the host path takes the same ones."""
    import isingmontecarlo_amd as im
    by_slot, swaps_ref = oracle_reference(oracle)
    g = im.QmcIsingGraph(lat.two_d_periodic(4), 1.0, 0.0, 16, SEED, nreplicas=len(BETAS) * K, capacity=4096)
    tc = im.NativeTemperingContainer(g, BETAS, K, SEED)
    assert tc.device_decisions  # the default for this layout
    if where == "host":
        tc.set_device_decisions(False)
    counted = 0
    for _ in range(STEPS):
        tc.timesteps(SWEEPS)
        sw = tc.tempering_step(count_swaps=where != "device, nothing read back between steps")
        counted += sw or 0
    assert tc.device_decisions == (where != "host")
    assert tc.get_total_swaps() == swaps_ref and swaps_ref > 0
    assert counted == (0 if where == "device, nothing read back between steps" else swaps_ref)
    st, n, cut = g.state_ref(), g.get_n(), g.get_cutoff()
    for r in range(g.nreplicas):
        t, k = divmod(int(tc.slot_of[r]), K)
        ref = by_slot[k][t]
        assert n[r] == ref.n and cut[r] == ref.cutoff
        assert np.array_equal(st[r], ref.state()) and np.array_equal(g.export_ops(r), ref.ops())
    assert tc.verify()
    acc = g.accumulators()
    assert acc.shape == (len(BETAS) * K, 8) and (acc[:, 1] == STEPS * SWEEPS).all()


@pytest.mark.timeout(600)
def test_native_step_two_ranks_exchange_configurations(oracle, tmp_path):
    """Two ranks (both on GPU 0), three temperatures each: swaps across the block boundary move whole configurations between
    the processes (op-string, state, counters, Philox identity).  The union of both ranks equals the oracle's single chain."""
    by_slot, swaps_ref = oracle_reference(oracle)
    out = str(tmp_path / "pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(HERE, "_pt_native_worker.py"), out]
    subprocess.check_call(cmd, env=env, cwd=os.path.dirname(HERE), timeout=500)
    moved = 0
    for rank in range(2):
        z = np.load(out + f".rank{rank}.npz")
        assert bool(z["ok"]) and int(z["swaps"]) == swaps_ref
        per = len(z["n"])
        for r in range(per):
            t, k = divmod(int(z["slot_of"][r]), K)
            assert rank * 3 <= t < rank * 3 + 3  # a rank only ever holds its own temperatures
            ref = by_slot[k][t]
            assert int(z["n"][r]) == ref.n and int(z["cutoff"][r]) == ref.cutoff
            assert np.array_equal(z["state"][r], ref.state())
            assert np.array_equal(z["ops"][r][:ref.cutoff], ref.ops())
            moved += int(z["config_of"][r]) // (3 * K) != rank  # configuration that started on the other rank
        assert (z["acc"][:, 1].reshape(len(BETAS), K)[rank * 3:rank * 3 + 3] == STEPS * SWEEPS).all()
    assert moved > 0


def _powi(x, n):
    r = 1.0
    while n:
        if n & 1:
            r *= x
        x *= x
        n >>= 1
    return r


def test_relative_weights_between_different_hamiltonians(oracle):
    """tempering_container.rs:830-858 test_bondstrength: nine graphs on one chain, couplings j * i / 10 (i = 1..9), the same beta:
    swaps are decided by GraphWeights::relative_weight alone (tempering_traits.rs:126-155).  Oracle side: configurations
    (op-string, state, update counter, Philox identity) are moved between replica objects built on the slots' models."""
    import isingmontecarlo_amd as im
    edges = [((0, 1), 1.0), ((1, 2), 1.0), ((2, 3), 1.0), ((3, 4), 1.0)]
    T, beta, gamma, seed, cap, steps = 9, 10.0, 0.1, 31415, 2048, 12
    J = np.array([[j * i / 10.0 for _, j in edges] for i in range(1, T + 1)])
    betas = np.full(T, beta)
    g = im.QmcIsingGraph(edges, gamma, 0.0, 10, seed, nreplicas=T, capacity=cap, couplings=J)
    tc = im.NativeTemperingContainer(g, betas, 1, seed)
    e = [list(ab) for ab, _ in edges]
    models = [oracle.Model(5, e, list(J[t]), gamma, 0.0) for t in range(T)]
    reps = [oracle.Replica(models[t], cap, 10, seed, t) for t in range(T)]
    ids = list(range(T))  # configuration identity at each slot
    key = (seed & 0xFFFFFFFF, seed >> 32)
    import ctypes as C
    def philox(idx, step):
        ctr = (C.c_uint32 * 4)(idx, step, 0, 6 << 24); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        oracle.lib().ora_philox4x32_10(ctr, k, o)
        return o[0]
    swaps_ref = 0
    for step in range(steps):
        tc.timesteps(20)
        for t in range(T):
            reps[t].timesteps(20, beta)
        tc.tempering_step()
        # ---- reference formulation with relative weights ----
        maxcut = max(r.cutoff for r in reps)
        for r in reps:
            assert r.set_cutoff(maxcut) == 0
        a_first = (philox(0, step) >> 31) != 0
        for phase in range(2):
            set_a = a_first if phase == 0 else not a_first
            for t in range(0 if set_a else 1, T - 1, 2):
                u = philox(1 + t, step) / 4294967296.0
                ga, gb = reps[t], reps[t + 1]
                rel_b = 1.0
                for b_ in range(len(edges)):
                    rel_b *= _powi(J[t + 1][b_] / J[t][b_], ga.bond_count(b_))
                rel_a = 1.0
                for b_ in range(len(edges)):
                    rel_a *= _powi(J[t][b_] / J[t + 1][b_], gb.bond_count(b_))
                p = (betas[t] / betas[t + 1]) ** float(gb.n - ga.n) * (rel_b * rel_a)
                if p > u:
                    swaps_ref += 1
                    na = oracle.Replica(models[t], cap, maxcut, seed, ids[t + 1], gb.state()); na.set_ops(gb.ops()); na.set_epoch(gb.epoch)
                    nb = oracle.Replica(models[t + 1], cap, maxcut, seed, ids[t], ga.state()); nb.set_ops(ga.ops()); nb.set_epoch(ga.epoch)
                    reps[t], reps[t + 1] = na, nb
                    ids[t], ids[t + 1] = ids[t + 1], ids[t]
    assert tc.get_total_swaps() == swaps_ref and swaps_ref > 0
    st, n = g.state_ref(), g.get_n()
    for r in range(T):
        t = int(tc.slot_of[r])
        assert int(tc.config_of[r]) == ids[t]
        assert n[r] == reps[t].n and np.array_equal(st[r], reps[t].state()) and np.array_equal(g.export_ops(r), reps[t].ops())
    assert tc.verify() and all(r.verify() for r in reps)


def test_container_save_and_load_resume_bit_exactly(oracle, tmp_path):
    """Container-level checkpoint (tempering_container.rs:683-792 serialises the whole container): a restored container
    continues exactly like the one that kept running — swaps, labels, configurations."""
    import isingmontecarlo_amd as im
    def make():
        g = im.QmcIsingGraph(lat.two_d_periodic(4), 1.0, 0.0, 16, SEED, nreplicas=len(BETAS) * K, capacity=4096)
        return g, im.NativeTemperingContainer(g, BETAS, K, SEED)
    g1, t1 = make()
    for _ in range(6):
        t1.timesteps(SWEEPS); t1.tempering_step()
    path = str(tmp_path / "ck")
    t1.save(path)
    g2, t2 = make()
    t2.load(path)
    for tc in (t1, t2):
        for _ in range(5):
            tc.timesteps(SWEEPS); tc.tempering_step()
    assert t1.get_total_swaps() == t2.get_total_swaps() > 0
    assert np.array_equal(t1.slot_of, t2.slot_of) and np.array_equal(t1.config_of, t2.config_of)
    assert np.array_equal(g1.get_n(), g2.get_n()) and np.array_equal(g1.state_ref(), g2.state_ref())
    for r in range(g1.nreplicas):
        assert np.array_equal(g1.export_ops(r), g2.export_ops(r))
    assert t2.verify()


def test_relative_weights_with_fields_varying_across_temperatures(oracle):
    """GraphWeights::relative_weight in full (tempering_traits.rs:126-155): couplings, transverse field AND longitudinal field
    differ between the graphs of one chain (the same beta everywhere, so swaps are decided by the three ratios alone:
    bond_ratio * transverse_ratio * longitudinal_ratio, in that order).  Oracle side as in the test above."""
    import isingmontecarlo_amd as im
    import ctypes as C
    edges = [((0, 1), -1.0), ((1, 2), 1.0), ((2, 3), -1.0), ((3, 0), -1.0), ((0, 2), 0.5)]
    T, beta, seed, cap, steps = 8, 3.0, 2718, 4096, 14
    E, N = len(edges), 4
    J = np.array([[j * (1.0 + 0.04 * t) for _, j in edges] for t in range(T)])
    gam = np.array([0.8 + 0.05 * t for t in range(T)])
    hl = np.array([0.30 - 0.02 * t for t in range(T)])
    betas = np.full(T, beta)
    g = im.QmcIsingGraph(edges, 1.0, 0.1, 8, seed, nreplicas=T, capacity=cap, couplings=J, transverse_r=gam, longitudinal_r=hl)
    tc = im.NativeTemperingContainer(g, betas, 1, seed)
    e = [list(ab) for ab, _ in edges]
    models = [oracle.Model(N, e, list(J[t]), float(gam[t]), float(hl[t])) for t in range(T)]
    reps = [oracle.Replica(models[t], cap, 8, seed, t) for t in range(T)]
    ids = list(range(T))
    key = (seed & 0xFFFFFFFF, seed >> 32)

    def philox(idx, step):
        ctr = (C.c_uint32 * 4)(idx, step, 0, 6 << 24); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        oracle.lib().ora_philox4x32_10(ctr, k, o)
        return o[0]

    def relative_weight(graph, t_from, t_to):
        w = 1.0
        for b_ in range(E):
            w *= _powi(J[t_to][b_] / J[t_from][b_], graph.bond_count(b_))
        w *= _powi(gam[t_to] / gam[t_from], sum(graph.bond_count(E + v) for v in range(N)))
        if abs(hl[t_from]) > np.finfo(float).eps:
            w *= _powi(hl[t_to] / hl[t_from], sum(graph.bond_count(E + N + v) for v in range(N)))
        return w

    swaps_ref = 0
    for step in range(steps):
        tc.timesteps(10)
        for t in range(T):
            reps[t].timesteps(10, beta)
        tc.tempering_step()
        maxcut = max(r.cutoff for r in reps)
        for r in reps:
            assert r.set_cutoff(maxcut) == 0
        a_first = (philox(0, step) >> 31) != 0
        for phase in range(2):
            set_a = a_first if phase == 0 else not a_first
            for t in range(0 if set_a else 1, T - 1, 2):
                u = philox(1 + t, step) / 4294967296.0
                ga, gb = reps[t], reps[t + 1]
                p = 1.0 * (relative_weight(ga, t, t + 1) * relative_weight(gb, t + 1, t))  # equal betas: (beta_a / beta_b)^dn == 1
                if p > u:
                    swaps_ref += 1
                    na = oracle.Replica(models[t], cap, maxcut, seed, ids[t + 1], gb.state()); na.set_ops(gb.ops()); na.set_epoch(gb.epoch)
                    nb = oracle.Replica(models[t + 1], cap, maxcut, seed, ids[t], ga.state()); nb.set_ops(ga.ops()); nb.set_epoch(ga.epoch)
                    reps[t], reps[t + 1] = na, nb
                    ids[t], ids[t + 1] = ids[t + 1], ids[t]
    assert tc.get_total_swaps() == swaps_ref and 0 < swaps_ref < steps * (T - 1)
    st, n = g.state_ref(), g.get_n()
    for r in range(T):
        t = int(tc.slot_of[r])
        assert int(tc.config_of[r]) == ids[t]
        assert n[r] == reps[t].n and np.array_equal(st[r], reps[t].state()) and np.array_equal(g.export_ops(r), reps[t].ops())
    assert tc.verify() and all(r.verify() for r in reps)
    off = g.get_offsets()
    for r in range(T):
        t = int(tc.slot_of[r])
        assert abs(off[r] - (np.abs(J[t]).sum() + N * (gam[t] + abs(hl[t])))) < 1e-12


@pytest.mark.timeout(900)
def test_rccl_transport_matches_the_host_staged_one():
    """The RCCL point-to-point transport (isingmc_pt_attach_nccl) against the host-staged one on the same problem, one rank per
    GPU (tools/rccl_selfcheck.py).  Needs two GPUs: skipped on the one-GPU boxes this project has been given so far — the RCCL
    branch is compiled and link-checked (test_abi_cpu.py) but has not executed anywhere yet."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(os.path.dirname(HERE), "tools", "rccl_selfcheck.py")]
    subprocess.check_call(cmd, env=env, cwd=os.path.dirname(HERE), timeout=800)
