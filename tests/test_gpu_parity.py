"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact on operator words,
operator counts, spin states and integer accumulators for the same Philox seed/counters."""
import numpy as np
import pytest

import _lattices as lat

pytestmark = pytest.mark.gpu


def make_pair(oracle, edges, gamma, h, cutoff, cap, seed, R, waves=0, state=None, k=0, cfg_flags=0, uf_limit=0):
    import isingmontecarlo_amd as im
    g = im.QmcIsingGraph(edges, gamma, h, cutoff, seed, state=state, nreplicas=R, capacity=cap,
                         waves_per_replica=waves, slots_per_lane=k, cfg_flags=cfg_flags, lds_uf_ids_limit=uf_limit)
    e, j = lat.split(edges)
    m = oracle.Model(g.nvars, e, j, gamma, h)
    reps = [oracle.Replica(m, cap, cutoff, seed, r, None if state is None else state) for r in range(R)]
    return g, m, reps


def assert_same(g, reps, what=""):
    n = g.get_n()
    cut = g.get_cutoff()
    st = g.state_ref()
    ep = g.get_epoch()
    for r, rep in enumerate(reps):
        assert n[r] == rep.n, f"{what}: n differs for replica {r}: {n[r]} vs {rep.n}"
        assert cut[r] == rep.cutoff, f"{what}: cutoff differs for replica {r}"
        assert ep[r] == rep.epoch, f"{what}: epoch differs for replica {r}"
        assert np.array_equal(st[r], rep.state()), f"{what}: state differs for replica {r}"
        ops = g.export_ops(r)
        assert np.array_equal(ops, rep.ops()), f"{what}: op words differ for replica {r}"


CASES = [
    ("ring8_afm", lat.one_d_periodic(8), 1.0, 0.0, 1.0),
    ("ring4_small_qmc", [((0, 1), -1.0), ((1, 2), 1.0), ((2, 3), 1.0), ((3, 0), 1.0)], 1.0, 0.0, 1.0),
    ("villain4", lat.two_d_periodic(4), 1.0, 0.0, 2.0),
    ("ferro6_long", lat.one_d_periodic(6, -1.0), 1.0, 0.3, 2.0),
    ("ferro8x8", lat.two_d_ferro(8), 1.0, 0.0, 4.0),
    ("ring5_randmag", [((0, 1), 0.7), ((1, 2), -1.3), ((2, 3), 1.9), ((3, 4), -0.6), ((4, 0), 1.1)], 1.2, -0.4, 1.5),
]


GEOMS = [(1, 1, 0), (1, 4, 0), (4, 2, 0), (4, 4, 1), (6, 2, 0), (6, 4, 1), (8, 1, 0), (8, 4, 0), (8, 4, 1), (16, 4, 0), (16, 2, 1)]


@pytest.mark.parametrize("waves", [1, 4, 8])
@pytest.mark.parametrize("name,edges,gamma,h,beta", CASES)
def test_init_state_matches(oracle, name, edges, gamma, h, beta, waves):
    g, m, reps = make_pair(oracle, edges, gamma, h, 16, 4096, 77, 3, waves)
    assert_same(g, reps, "init")


@pytest.mark.parametrize("waves,k,cfgf", GEOMS)
@pytest.mark.parametrize("name,edges,gamma,h,beta", CASES)
def test_primitives_step_by_step(oracle, name, edges, gamma, h, beta, waves, k, cfgf):
    R = 4
    g, m, reps = make_pair(oracle, edges, gamma, h, 16, 8192, 1234, R, waves, k=k, cfg_flags=cfgf)
    info = g.launch_info()
    uniform = len({abs(j) for _, j in edges}) == 1
    assert info["waves_per_replica"] == waves and info["slots_per_lane"] == k
    assert info["lds_edge_table"] == (cfgf == 0 and uniform)
    for it in range(12):
        g.single_diagonal_step(beta)
        for rep in reps:
            rep.diagonal_update(beta)
            want = rep.n + rep.n // 2
            if want > rep.cutoff:
                assert rep.set_cutoff(want) == 0
        assert_same(g, reps, f"{name} diag it={it}")
        nc = g.single_cluster_step(flip_free=False)
        for r, rep in enumerate(reps):
            assert nc[r] == rep.cluster_update(0.5), f"{name}: cluster count differs it={it} r={r}"
        assert_same(g, reps, f"{name} cluster it={it}")
        g.flip_free_spins()
        for rep in reps:
            rep.flip_free_spins()
        assert_same(g, reps, f"{name} free it={it}")
    assert g.verify().all()
    assert all(rep.verify() for rep in reps)


@pytest.mark.parametrize("flags", [0, 1, 4, 5, 2, 3])
@pytest.mark.parametrize("name,edges,gamma,h,beta", CASES)
def test_fused_timesteps(oracle, name, edges, gamma, h, beta, flags):
    R = 5
    g, m, reps = make_pair(oracle, edges, gamma, h, 8, 8192, 99, R)
    if flags & 1:
        g.set_steps_per_launch(7)  # sampling phase must carry across launches
    g.run(40, beta, sampling_freq=3, flags=flags)
    for rep in reps:
        rep.timesteps(40, beta, 3, flags)
    assert_same(g, reps, f"{name} flags={flags}")
    acc = g.accumulators()
    for r, rep in enumerate(reps):
        assert np.array_equal(acc[r, :7], rep.accumulators()[:7]), f"accumulators differ r={r}: {acc[r]} vs {rep.accumulators()}"
    assert g.verify().all()


@pytest.mark.parametrize("waves,k,cfgf", [(8, 4, 0), (1, 1, 1), (16, 2, 0)])
def test_loop_update_matches(oracle, waves, k, cfgf):
    edges = lat.two_d_periodic(4)
    R = 6
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 8, 4096, 5, R, waves=waves, k=k, cfg_flags=cfgf)
    g.run(20, 2.0)
    for rep in reps:
        rep.timesteps(20, 2.0, 1, 0)
    for it in range(30):
        lens = g.loop_update()
        for r, rep in enumerate(reps):
            assert lens[r] == rep.loop_update(), f"loop length differs it={it} r={r}"
        assert_same(g, reps, f"loop it={it}")
    assert g.verify().all()


# cfgf 2 = fused launches; 0 at the default geometry = trimmed diagonal kernel; 16 = general diagonal kernel;
# experimental hand-overs to the cluster update: 64 = dense op list, 32 = segment labelling
@pytest.mark.parametrize("waves,k,cfgf", [(8, 4, 0), (16, 4, 0), (8, 4, 1), (4, 4, 2), (8, 2, 3), (0, 0, 0), (0, 0, 64), (0, 0, 16), (4, 2, 0), (0, 0, 32)])
def test_medium_lattice_many_replicas(oracle, waves, k, cfgf):
    edges = lat.two_d_ferro(16)
    R = 16
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 256, 1 << 15, 2024, R, waves=waves, k=k, cfg_flags=cfgf)
    info = g.launch_info()
    if waves == 0:
        assert info["fast_diagonal"] == (cfgf != 16) and info["compact_list"] == (cfgf == 64) and info["fast_label"] == (cfgf == 32)
    g.run(30, 4.0)
    oracle.batch_timesteps(reps, 30, [4.0] * R)
    assert_same(g, reps, "16x16")
    g.run(10, 4.0, flags=1)  # with a directed loop between the diagonal launch and the cluster update
    oracle.batch_timesteps(reps, 10, [4.0] * R, 1, 1)
    assert_same(g, reps, "16x16 + loop")
    assert g.verify().all()


def test_many_replicas_stress(oracle):
    # wide batch at the bench geometry: every replica runs the concurrent union-find / flatten / coin phases with
    # its own interleaving, so an ordering bug between workgroup threads shows up as a parity break somewhere
    edges = lat.two_d_ferro(32)
    R = 1024  # BASELINE configs[1] at full size: 1024 replicas, 32x32, beta = 16
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 1024, 1 << 17, 77, R)
    g.run(30, 16.0, flags=1)  # the bench workload (with the directed loop), from the initial cutoff growth on
    oracle.batch_timesteps(reps, 30, [16.0] * R, 1, 1)
    assert_same(g, reps, "32x32 x1024")
    assert g.verify().all()


@pytest.mark.parametrize("waves,k", [(8, 4), (4, 1), (16, 4)])
def test_union_find_global_fallback(oracle, waves, k):
    # cap the LDS union-find so that N + (transverse ops) exceeds it: the HBM union-find path must agree
    edges = lat.two_d_ferro(12)
    R = 3
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 144, 1 << 14, 31, R, waves=waves, k=k, uf_limit=400)
    info = g.launch_info()
    assert info["lds_uf_ids"] == 400
    g.run(25, 4.0)
    oracle.batch_timesteps(reps, 25, [4.0] * R)
    assert_same(g, reps, "12x12 global union-find")
    # make sure this case really left the LDS path: N + (transverse ops) must exceed the LDS id capacity
    bonds = (reps[0].ops() >> 4).astype(np.int64) - 1
    ntrans = int(((bonds >= len(edges)) & (bonds < len(edges) + g.nvars)).sum())
    assert g.nvars + ntrans > info["lds_uf_ids"], (g.nvars, ntrans, info)


def test_capacity_error_is_loud(oracle):
    import isingmontecarlo_amd as im
    g = im.QmcIsingGraph(lat.two_d_ferro(8), 1.0, 0.0, 64, 1, nreplicas=2, capacity=128)
    with pytest.raises(im.IsingMcError) as ei:
        g.run(50, 8.0)
    assert ei.value.code == -3


def test_parallel_tempering_matches_oracle(oracle):
    """TemperingContainer over the GPU batch (labels swap, configurations stay) against the oracle's
    graph-swapping restatement of tempering_container.rs:121-149: same swaps, same per-slot configurations."""
    import isingmontecarlo_amd as im
    from test_tempering_cpu import reference_pt
    edges = lat.two_d_periodic(4)
    e, j = lat.split(edges)
    m = oracle.Model(16, e, j, 1.0, 0.0)
    betas = np.array([0.5, 0.8, 1.1, 1.5, 2.0, 2.6])
    K, seed = 4, 2468
    by_slot, swaps_ref = reference_pt(m, betas, K, seed, 4096, 16, nsteps=15, sweeps_per_step=2)
    g = im.QmcIsingGraph(edges, 1.0, 0.0, 16, seed, nreplicas=len(betas) * K, capacity=4096)
    tc = im.TemperingContainer(g, betas, K, seed)
    for _ in range(15):
        tc.timesteps(2)
        tc.tempering_step()
    assert tc.get_total_swaps() == swaps_ref and swaps_ref > 0
    st, n, cut = g.state_ref(), g.get_n(), g.get_cutoff()
    for k in range(K):
        for t in range(len(betas)):
            c = int(tc.config_at[t * K + k])
            ref = by_slot[k][t]
            assert n[c] == ref.n and cut[c] == ref.cutoff
            assert np.array_equal(st[c], ref.state())
            assert np.array_equal(g.export_ops(c), ref.ops())
    assert tc.verify()
    acc = tc.slot_accumulators()
    assert acc.shape == (len(betas) * K, 8) and (acc[:, 1] == 30).all()


RVB_CASES = [
    ("single_bond", [((0, 1), 1.0)], 1.0, 0.0, 1.0, 2),
    ("single_bond_long", [((0, 1), 1.0)], 1.0, 1.0, 1.0, 2),
    ("villain3", lat.two_d_periodic(3), 1.0, 0.0, 1.5, 9),
    ("villain4", lat.two_d_periodic(4), 1.0, 0.0, 2.0, 16),
    ("ferro8x8", lat.two_d_ferro(8), 1.0, 0.0, 3.0, 64),
    ("ferro8x8_long", lat.two_d_ferro(8), 1.0, 0.3, 3.0, 64),   # longitudinal ops inside clusters zero the weight ratio
    ("ferro16x16_b4", lat.two_d_ferro(16), 1.0, 0.0, 4.0, 256),  # thousands of attempts: clusters that outgrow a small growth area
]


# 128: attempts grown one at a time; 2048: the fused kernel instead of the growth + main launches (whose main launch runs with 8, 4, 4, 16 waves here)
@pytest.mark.parametrize("waves,k,cfgf", [(8, 4, 0), (1, 1, 1), (4, 2, 0), (0, 0, 128), (16, 4, 0), (0, 0, 2048), (8, 4, 2048 | 128)])
@pytest.mark.parametrize("name,edges,gamma,h,beta,cutoff", RVB_CASES, ids=[c[0] for c in RVB_CASES])
def test_rvb_update_matches_oracle(oracle, name, edges, gamma, h, beta, cutoff, waves, k, cfgf):
    """RvbUpdater::rvb_update (rvb.rs:88-290): attempt by attempt identical to the oracle (ops, state, successes)."""
    R = 4
    g, m, reps = make_pair(oracle, edges, gamma, h, cutoff, 8192, 1357, R, waves, k=k, cfg_flags=cfgf)
    for it in range(10):
        g.single_diagonal_step(beta)
        for rep in reps:
            rep.diagonal_update(beta)
            want = rep.n + rep.n // 2
            if want > rep.cutoff:
                assert rep.set_cutoff(want) == 0
        succ, upd = g.single_rvb_sweep()
        for r, rep in enumerate(reps):
            assert succ[r] == rep.rvb_update(upd), f"{name}: RVB successes differ it={it} r={r}"
        assert_same(g, reps, f"{name} rvb it={it}")
        g.single_cluster_step(flip_free=True)
        for rep in reps:
            rep.cluster_update(0.5)
            rep.flip_free_spins()
        assert_same(g, reps, f"{name} cluster it={it}")
    assert g.verify().all()
    info = g.launch_info()
    assert info["rvb_split"] == (not (cfgf & 2048)), info
    if info["rvb_split"]:
        assert info["rvb_main_waves"] == (waves if waves in (4, 8, 16) else 4), info


@pytest.mark.parametrize("cfgf", [0, 128, 2048])
def test_rvb_dense_windows(oracle, cfgf):
    """A small lattice at low temperature: most ops touch a sub-variable, so a window's first chunk overflows the gathered-op
    list (left to the smaller gather steps, which are cut at a wave boundary) and replay batches are full."""
    edges = lat.two_d_ferro(8)
    R, beta = 3, 12.0
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 6000, 1 << 14, 9753, R, cfg_flags=cfgf)
    for it in range(6):
        g.single_diagonal_step(beta)
        for rep in reps:
            rep.diagonal_update(beta)
            want = rep.n + rep.n // 2
            if want > rep.cutoff:
                assert rep.set_cutoff(want) == 0
        succ, upd = g.single_rvb_sweep()
        for r, rep in enumerate(reps):
            assert succ[r] == rep.rvb_update(upd), f"dense: RVB successes differ it={it} r={r}"
        assert_same(g, reps, f"dense rvb it={it}")
        g.single_cluster_step(flip_free=True)
        for rep in reps:
            rep.cluster_update(0.5)
            rep.flip_free_spins()
    assert g.get_n().min() > 2500
    assert g.verify().all()


def test_rvb_on_a_model_beyond_the_bond_map_takes_the_fused_kernel(oracle):
    """The two-launch RVB sweep builds a per-attempt bond map in 256 words of a growth area: models with more than 8192 bonds
    (here 64x64: 8192 edges + 4096 transverse bonds) run the fused kernel instead — same results, and launch_info says so."""
    edges = lat.two_d_ferro(64)
    R, beta = 2, 0.5
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 4096, 1 << 15, 4321, R)
    for it in range(3):
        g.single_diagonal_step(beta)
        for rep in reps:
            rep.diagonal_update(beta)
            want = rep.n + rep.n // 2
            if want > rep.cutoff:
                assert rep.set_cutoff(want) == 0
        succ, upd = g.single_rvb_sweep()
        for r, rep in enumerate(reps):
            assert succ[r] == rep.rvb_update(upd), (it, r)
        assert_same(g, reps, f"64x64 rvb it={it}")
    assert not g.launch_info()["rvb_split"]
    assert g.verify().all()


def test_rvb_fused_timesteps(oracle):
    edges = lat.two_d_periodic(4)
    R = 6
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 16, 8192, 8642, R)
    g.run(30, 2.0, sampling_freq=2, flags=8)
    for rep in reps:
        rep.timesteps(30, 2.0, 2, 8)
    assert_same(g, reps, "rvb fused")
    acc = g.accumulators()
    for r, rep in enumerate(reps):
        assert np.array_equal(acc[r, :7], rep.accumulators()[:7])
    assert g.verify().all()


def test_rvb_config2_full_size(oracle):
    """BASELINE configs[2] at full size (32x32, beta=16, ~10^5 slots): whole timesteps with the RVB sweep — batches of
    attempts grown side by side, 64-op replay batches, look-back and window chunks of thousands of slots — op for op."""
    edges = lat.two_d_ferro(32)
    R = 3
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 150000, 1 << 18, 2718, R)
    for it in range(2):
        g.run(4, 16.0, sampling_freq=1, flags=8)
        for rep in reps:
            rep.timesteps(4, 16.0, 1, 8)
        assert_same(g, reps, f"configs[2] rvb it={it}")
    assert g.get_n().min() > 50000
    acc = g.accumulators()
    for r, rep in enumerate(reps):
        assert np.array_equal(acc[r, :7], rep.accumulators()[:7])
    assert g.verify().all()


def test_large_lattice_64x64_runs_and_matches(oracle):
    """configs[3] geometry (64x64): edge table in LDS, union-find in HBM, many chunks per wave."""
    edges = lat.two_d_ferro(64)
    R = 2
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 4096, 1 << 17, 6464, R)
    info = g.launch_info()
    g.run(14, 2.0)
    oracle.batch_timesteps(reps, 14, [2.0] * R)
    assert_same(g, reps, "64x64")
    assert g.verify().all()
    assert 8 * 4096 + 2000 > info["lds_uf_ids"]  # W*N + cuts exceeds the LDS union-find: HBM path exercised


@pytest.mark.parametrize("waves,heatbath", [(0, False), (4, True), (1, False)])
def test_disorder_realisations_cubic_pm_j(oracle, waves, heatbath):
    """configs[4] in small: +-J random-bond cubic lattice with transverse and longitudinal field, one disorder
    realisation per replica in ONE batch (ISINGMC_CFG_PER_REPLICA_J); every replica is checked against an oracle
    replica built on its own couplings."""
    import isingmontecarlo_amd as im
    l, R = 4, 6
    edges = lat.cubic_periodic(l)
    rng = np.random.default_rng(20260)
    J = rng.choice([-1.0, 1.0], size=(R, len(edges)))
    gamma, h, beta, cutoff, cap, seed = 1.0, 0.1, 2.0, 64, 1 << 13, 4096
    g = im.QmcIsingGraph(edges, gamma, h, cutoff, seed, nreplicas=R, capacity=cap, couplings=J, waves_per_replica=waves)
    e = [ab for ab, _ in edges]
    reps = []
    for r in range(R):
        m = oracle.Model(g.nvars, e, list(J[r]), gamma, h)
        reps.append(oracle.Replica(m, cap, cutoff, seed, r, None))
    flags = im.FLAG_HEATBATH if heatbath else 0
    g.run(20, beta, flags=flags)
    oracle.batch_timesteps(reps, 20, [beta] * R, 1, flags)
    assert_same(g, reps, "cubic +-J")
    assert g.verify().all()
    # different realisations, different offsets only through sum |J| (equal here): energies differ through n
    assert np.allclose(g.get_offsets(), g.get_offset())
    assert len(set(int(x) for x in g.get_n())) > 1
    # RVB sweeps on per-replica couplings (qmc_ising.rs:705-752: with h != 0 every cluster that holds a longitudinal op is
    # frozen, and the reference relies on RVB moves for exactly this +-J + field workload): standalone sweeps and whole timesteps
    for it in range(3):
        succ, upd = g.single_rvb_sweep()
        for r, rep in enumerate(reps):
            assert succ[r] == rep.rvb_update(upd), (it, r)
        assert_same(g, reps, f"cubic +-J rvb sweep {it}")
    g.run(12, beta, flags=flags | im.FLAG_RVB)
    oracle.batch_timesteps(reps, 12, [beta] * R, 1, flags | im.FLAG_RVB)
    assert_same(g, reps, "cubic +-J timesteps with RVB")
    assert g.verify().all()


def test_itime_magnetization_fold(oracle):
    """imaginary_time_fold of m, m^2, |m| (fast_ops.rs:1296-1315) on the device against the oracle's sequential fold and the
    generic host-side fold."""
    edges = lat.two_d_ferro(8)
    R = 5
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 64, 1 << 13, 99, R)
    g.run(25, 3.0)
    oracle.batch_timesteps(reps, 25, [3.0] * R)
    assert_same(g, reps, "8x8 before fold")
    s1, s2, sa = g.itime_magnetization()
    for r, rep in enumerate(reps):
        assert (int(s1[r]), int(s2[r]), int(sa[r])) == rep.itime_magnetization(), f"replica {r}"
    mag = lambda st: 2 * int(st.sum()) - len(st)
    assert g.imaginary_time_fold(lambda acc, st: acc + mag(st) ** 2, 0, r=2) == int(s2[2])
    assert (np.abs(s1) <= sa.astype(np.int64)).all() and (sa > 0).any()


def _random_model(rng):
    """Random connected graph with random couplings (mixed signs, uniform or non-uniform magnitudes), fields and beta."""
    n = int(rng.integers(3, 40))
    edges = {}
    for v in range(1, n):  # spanning tree first: connected
        edges[(int(rng.integers(0, v)), v)] = 0.0
    for _ in range(int(rng.integers(0, 2 * n))):
        a, b = (int(x) for x in rng.integers(0, n, size=2))
        if a != b:
            edges[(min(a, b), max(a, b))] = 0.0
    uniform = bool(rng.integers(0, 2))
    out = []
    for (a, b) in sorted(edges):
        mag = 1.0 if uniform else float(rng.uniform(0.3, 2.0))
        out.append(((a, b), mag * (1.0 if rng.integers(0, 2) else -1.0)))
    gamma = float(rng.choice([0.4, 1.0, 1.7]))
    h = float(rng.choice([0.0, 0.0, 0.25, -0.6]))
    beta = float(rng.choice([0.5, 1.5, 3.0]))
    return out, gamma, h, beta


@pytest.mark.parametrize("seed", range(12))
def test_random_models_all_passes(oracle, seed):
    """Randomised sweep of the parameter space the fixed cases do not reach: random graphs, couplings, fields, beta, launch
    geometry, pass combination (Metropolis / heat-bath, with / without directed loop, fused / split launches)."""
    rng = np.random.default_rng(1000 + seed)
    edges, gamma, h, beta = _random_model(rng)
    waves = int(rng.choice([0, 1, 4, 8, 16]))
    k = int(rng.choice([0, 1, 2, 4]))
    cfgf = int(rng.choice([0, 1, 2, 3]))
    flags = int(rng.choice([0, 1, 4, 5]))
    R = int(rng.integers(1, 7))
    g, m, reps = make_pair(oracle, edges, gamma, h, 8, 1 << 13, 555 + seed, R, waves=waves, k=k, cfg_flags=cfgf)
    steps = int(rng.integers(10, 40))
    g.run(steps, beta, sampling_freq=2, flags=flags)
    for rep in reps:
        rep.timesteps(steps, beta, 2, flags)
    what = f"random model seed={seed} n={g.nvars} E={len(edges)} gamma={gamma} h={h} beta={beta} W={waves} K={k} cfg={cfgf} flags={flags}"
    assert_same(g, reps, what)
    acc = g.accumulators()
    for r, rep in enumerate(reps):
        assert np.array_equal(acc[r, :7], rep.accumulators()[:7]), what
    assert g.verify().all(), what


def test_checkpoint_resume_is_bit_exact(oracle, tmp_path):
    """save_checkpoint / load_checkpoint: a restored batch continues exactly like the original (and like the oracle)."""
    import isingmontecarlo_amd as im
    edges = lat.two_d_periodic(4)
    R = 4
    g, m, reps = make_pair(oracle, edges, 1.0, 0.2, 16, 1 << 12, 77, R)
    g.run(15, 2.0)
    path = str(tmp_path / "ckpt.npz")
    g.save_checkpoint(path)
    g.run(12, 2.0)
    g2 = im.QmcIsingGraph(edges, 1.0, 0.2, 16, 77, nreplicas=R, capacity=1 << 12)
    g2.load_checkpoint(path)
    assert g2.verify().all()
    g2.run(12, 2.0)
    oracle.batch_timesteps(reps, 27, [2.0] * R)
    assert_same(g, reps, "original")
    assert_same(g2, reps, "resumed")
    g3 = im.QmcIsingGraph(lat.two_d_periodic(4), 1.0, 0.0, 16, 77, nreplicas=R, capacity=1 << 12)
    with pytest.raises(im.IsingMcError):
        g3.load_checkpoint(path)


def test_cpp_example_through_the_c_abi(tmp_path):
    """examples/small_qmc.cpp (the reference's examples/small_qmc.rs from compiled code over include/isingmc_hip.h):
    builds with g++, runs on the GPU and lands on the exact-diagonalisation energy of the model."""
    import json, os, subprocess
    import isingmontecarlo_amd as im
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(im.load_library()._name)
    exe = str(tmp_path / "small_qmc")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "small_qmc.cpp"),
                           "-L" + libdir, "-lisingmc_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.check_output([exe, "512"], text=True).split()
    energy, sem = float(out[1]), float(out[3])
    golden = json.load(open(os.path.join(root, "tests", "golden", "ed_tfim.json")))
    exact = next(r["energy"] for c in golden if c["name"] == "small_qmc_ring4" for r in c["results"] if r["beta"] == 1.0)
    assert abs(energy - exact) < 5 * sem + 1e-3, (energy, sem, exact)


@pytest.mark.parametrize("cluster", [False, True])
@pytest.mark.parametrize("waves,k,heatbath,loop", [(0, 0, False, True), (4, 4, True, True), (1, 1, False, True), (8, 2, False, False), (16, 4, True, False)])
def test_generic_interactions_match_oracle(oracle, waves, k, heatbath, loop, cluster):
    """qmc::sse::Qmc with arbitrary one- and two-variable weight matrices (qmc_runner.rs:415-680): XXZ ring with hopping
    (two-variable off-diagonal ops) plus constant one-variable terms; diagonal (Metropolis / heat-bath) + directed loop +
    free spins against the oracle, bit for bit."""
    import isingmontecarlo_amd as im
    n, R, beta, cutoff, cap, seed = 7, 5, 1.5, 8, 1 << 12, 909
    ints = lat.xxz_ring_interactions(n)
    g = im.Qmc.from_interactions(n, ints, cutoff, seed, nreplicas=R, capacity=cap, waves_per_replica=waves, slots_per_lane=k,
                                 do_loop_updates=loop, do_cluster_updates=cluster)
    m = oracle.Model.generic(n, ints)
    reps = [oracle.Replica(m, cap, cutoff, seed, r, None) for r in range(R)]
    flags = (0 if cluster else im.FLAG_NO_CLUSTER) | (im.FLAG_LOOP if loop else 0) | (im.FLAG_HEATBATH if heatbath else 0)
    g.run(60, beta, sampling_freq=2, flags=flags)
    for rep in reps:
        rep.timesteps(60, beta, 2, flags)
    assert_same(g, reps, "generic interactions")
    acc = g.accumulators()
    for r, rep in enumerate(reps):
        assert np.array_equal(acc[r, :7], rep.accumulators()[:7])
    assert g.verify().all()
    if loop:  # two-variable off-diagonal ops (hopping) are really there
        assert any(((int(w) >> 4) - 1) < n and (int(w) & 3) != ((int(w) >> 2) & 3) for r in range(R) for w in g.export_ops(r) if w)
    # imaginary-time fold (device magnetisation sums vs the generic host-side fold over two-variable off-diagonal ops)
    s1, s2, sa = g.itime_magnetization()
    assert g.imaginary_time_fold(lambda a, st: a + (2 * int(st.sum()) - len(st)), 0, r=1) == int(s1[1])
    assert (int(s1[0]), int(s2[0]), int(sa[0])) == reps[0].itime_magnetization()
    # a model that breaks the Ising symmetry refuses cluster updates like the reference (qmc_runner.rs:224-226)
    broken = im.Qmc.from_interactions(2, [(np.array([1.0, 0, 0, 2.0]), (0,)), (np.full(4, 0.5), (1,))], 4, 1)
    with pytest.raises(im.IsingMcError):
        broken.single_cluster_step()


def test_variable_autocorrelation_runs_on_sampled_states(oracle):
    """QmcAutoCorrelations::calculate_variable_autocorrelation (autocorrelations.rs:37-50) over a batch: shape, r[0] = 1,
    and the sampled trajectory is the same one plain timesteps produce (checked through the oracle)."""
    from isingmontecarlo_amd.autocorrelations import variable_autocorrelation, spin_product_autocorrelation
    edges = lat.two_d_ferro(4)
    R = 3
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 16, 1 << 11, 5, R)
    ac = variable_autocorrelation(g, 64, 1.0, sampling_freq=2)
    assert ac.shape == (R, 32) and np.isfinite(ac).all() and (ac[:, 0] > 0.5).all() and (ac[:, 0] <= 1.0 + 1e-9).all()
    oracle.batch_timesteps(reps, 64, [1.0] * R)
    assert_same(g, reps, "after autocorrelation sampling")
    pc = spin_product_autocorrelation(g, 32, 1.0, [(0, 1), (2, 3, 4)], r=1)
    assert pc.shape == (32,) and np.isfinite(pc).all() and 0.0 <= pc[0] <= 1.0 + 1e-9


def test_variable_autocorrelation_values(oracle):
    """Values, not shapes: the autocorrelation of the states the batch samples equals the defining O(T^2) circular sum over the
    states the ORACLE samples on the same trajectory (autocorrelations.rs:99-133), through numpy's FFT and through hipFFT."""
    from isingmontecarlo_amd.autocorrelations import variable_autocorrelation, direct_autocorrelation
    edges = lat.two_d_ferro(4)
    R, T, freq = 3, 48, 2
    for device in (None, "cuda"):
        g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 16, 1 << 11, 77, R)
        ac = variable_autocorrelation(g, T * freq, 1.5, sampling_freq=freq, device=device)
        samples = [[] for _ in range(R)]
        for _ in range(T):
            oracle.batch_timesteps(reps, freq, [1.5] * R)
            for k, rep in enumerate(reps):
                samples[k].append(rep.state().astype(np.float64) * 2.0 - 1.0)
        assert_same(g, reps, "after autocorrelation sampling")
        for k in range(R):
            want = direct_autocorrelation(np.stack(samples[k]))
            assert ac[k].shape == want.shape and np.abs(ac[k] - want).max() < 1e-10, (device, k, np.abs(ac[k] - want).max())
            assert abs(want[0] - 1.0) < 1e-12 or (np.stack(samples[k]).std(axis=0) == 0).any()


def test_bond_counts_match_the_oracle(oracle):
    """OpContainer::get_count (op_container.rs:129; fast_ops.rs:1281-1294) for every bond, straight through isingmc_get_bond_count."""
    edges = lat.one_d_periodic(6, -1.0)
    R = 3
    g, m, reps = make_pair(oracle, edges, 0.8, 0.3, 6, 1 << 11, 2718, R)
    g.run(40, 2.5)
    oracle.batch_timesteps(reps, 40, [2.5] * R)
    assert g.num_bonds() == m.nbonds == 6 + 6 + 6
    n = g.get_n()
    for r, rep in enumerate(reps):
        counts = [g.get_bond_count(b, r) for b in range(m.nbonds)]
        assert counts == [rep.bond_count(b) for b in range(m.nbonds)]
        assert sum(counts) == n[r] and max(counts) > 0


def test_into_qmc_hands_the_batch_over(oracle):
    """IntoQmc::into_qmc (qmc_ising.rs:943-976; tests/convert_test.rs): the converted object owns the live device batch and
    continues the same Markov chain through Qmc::timestep (diagonal -> loop -> cluster -> free spins)."""
    import isingmontecarlo_amd as im
    edges = lat.one_d_periodic(8)
    R = 4
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 8, 1 << 11, 1234, R)
    g.run(10, 1.0)
    q = g.into_qmc(do_loop_updates=True)
    assert g._h is None and q._h is not None and isinstance(q, im.Qmc) and q.should_do_loop_update()
    q.run(10, 1.0)
    for rep in reps:
        rep.timesteps(10, 1.0, 1, 0)
        rep.timesteps(10, 1.0, 1, im.FLAG_LOOP)
    assert_same(q, reps, "into_qmc")
    assert q.verify().all()
    g.close()  # the emptied source object must not free the batch it handed over
    assert q.get_n().shape == (R,)


def test_checkpoint_restores_accumulators_and_refuses_larger_cutoffs(oracle, tmp_path):
    import isingmontecarlo_amd as im
    edges = lat.two_d_periodic(4)
    R = 3
    g = im.QmcIsingGraph(edges, 1.0, 0.0, 16, 5, nreplicas=R, capacity=1 << 12)
    g.run(20, 2.0, sampling_freq=2)
    path = str(tmp_path / "c.npz")
    g.save_checkpoint(path)
    g2 = im.QmcIsingGraph(edges, 1.0, 0.0, 16, 5, nreplicas=R, capacity=1 << 12)
    g2.load_checkpoint(path)
    assert np.array_equal(g2.accumulators(), g.accumulators()) and g.accumulators()[:, 1].all()
    g.run(7, 2.0, sampling_freq=2); g2.run(7, 2.0, sampling_freq=2)
    assert np.array_equal(g2.accumulators(), g.accumulators())
    big = im.QmcIsingGraph(edges, 1.0, 0.0, 4000, 5, nreplicas=R, capacity=1 << 12)  # larger cutoff than the saved one
    with pytest.raises(im.IsingMcError):
        big.load_checkpoint(path)


def test_error_flags_are_sticky_until_cleared(oracle):
    import isingmontecarlo_amd as im
    g = im.QmcIsingGraph(lat.two_d_ferro(4), 1.0, 0.0, 16, 3, nreplicas=2, capacity=96)
    with pytest.raises(im.IsingMcError) as ei:
        g.run(60, 6.0)
    assert ei.value.code == -3
    with pytest.raises(im.IsingMcError):  # sticky: the replica stays parked
        g.run(1, 0.1)
    g.clear_errors()
    g.run(1, 0.01)  # at a high temperature n + n/2 fits again


def test_more_than_two_variables_is_a_loud_enotimpl():
    """Interactions on k > 2 variables (qmc_runner.rs:415-680 allows any k) are refused loudly, never truncated."""
    import ctypes as C
    import isingmontecarlo_amd as im
    with pytest.raises(im.IsingMcError) as ei:
        im.Qmc.from_interactions(3, [(np.ones(64), (0, 1, 2))], 4, 1)
    assert ei.value.code == -5
    lib = im.load_library()
    mat = np.ones(64)
    it = im._Interaction(nvars=3, mat=mat.ctypes.data_as(C.POINTER(C.c_double)))
    cfg = im._Config(struct_size=C.sizeof(im._Config), nreplicas=1, nvars=3, capacity=8, cutoff0=4, device=-1,
                     interactions=C.cast(C.pointer(it), C.c_void_p), ninteractions=1)
    h = C.c_void_p()
    assert lib.isingmc_create(C.byref(cfg), C.byref(h)) == -5
    assert b"more than two variables" in lib.isingmc_last_error(None)


# ---- models whose per-variable tables live in HBM (ISINGMC_CFG_GLOBAL_TABLES; automatic for N >~ 10^4, BASELINE configs[4]) ----

@pytest.mark.parametrize("waves,k,flags", [(4, 4, 0), (1, 1, 0), (8, 4, 1), (4, 1, 4), (16, 4, 5), (6, 4, 0)])
@pytest.mark.parametrize("name,edges,gamma,h,beta", CASES, ids=[c[0] for c in CASES])
def test_global_tables_path_matches_oracle(oracle, name, edges, gamma, h, beta, waves, k, flags):
    """The HBM/L2-resident table path forced on small models: every pass combination against the oracle, bit for bit."""
    import isingmontecarlo_amd as im
    R = 4
    g, m, reps = make_pair(oracle, edges, gamma, h, 8, 8192, 4321, R, waves, k=k,
                           cfg_flags=im.CFG_GLOBAL_TABLES | im.CFG_NO_LDS_TABLES)
    info = g.launch_info()
    assert info["global_tables"] and not info["lds_edge_table"] and info["lds_uf_ids"] == 0
    for it in range(3):  # primitives one by one ...
        g.single_diagonal_step(beta)
        for rep in reps:
            rep.diagonal_update(beta)
            want = rep.n + rep.n // 2
            if want > rep.cutoff:
                assert rep.set_cutoff(want) == 0
        assert_same(g, reps, f"{name} TG diag it={it}")
        nc = g.single_cluster_step(flip_free=True)
        for r, rep in enumerate(reps):
            assert nc[r] == rep.cluster_update(0.5)
            rep.flip_free_spins()
        assert_same(g, reps, f"{name} TG cluster it={it}")
    g.run(30, beta, sampling_freq=2, flags=flags)  # ... and whole timesteps
    for rep in reps:
        rep.timesteps(30, beta, 2, flags)
    assert_same(g, reps, f"{name} TG timesteps flags={flags}")
    assert g.verify().all()
    with pytest.raises(im.IsingMcError) as ei:
        g.single_rvb_sweep()
    assert ei.value.code == -5


def test_global_tables_with_per_replica_couplings_and_generic_interactions(oracle):
    """configs[4] in small through the HBM-table path: +-J cubic lattice, one disorder realisation per replica, h != 0;
    and a generic-interaction model on the same path."""
    import isingmontecarlo_amd as im
    l, R = 4, 5
    edges = lat.cubic_periodic(l)
    rng = np.random.default_rng(99)
    J = rng.choice([-1.0, 1.0], size=(R, len(edges)))
    g = im.QmcIsingGraph(edges, 1.0, 0.1, 64, 777, nreplicas=R, capacity=1 << 13, couplings=J, cfg_flags=im.CFG_GLOBAL_TABLES)
    assert g.launch_info()["global_tables"]
    e = [ab for ab, _ in edges]
    reps = [oracle.Replica(oracle.Model(g.nvars, e, list(J[r]), 1.0, 0.1), 1 << 13, 64, 777, r, None) for r in range(R)]
    g.run(25, 2.0)
    oracle.batch_timesteps(reps, 25, [2.0] * R)
    assert_same(g, reps, "cubic +-J, tables in HBM")
    assert g.verify().all()
    # the same through heat-bath sweeps with a directed loop (the +-J decode serves every pass of the launch), and the general
    # 16-byte bond records on the same model for the A-B switch
    g.run(10, 2.0, flags=im.FLAG_HEATBATH | im.FLAG_LOOP)
    oracle.batch_timesteps(reps, 10, [2.0] * R, 1, im.FLAG_HEATBATH | im.FLAG_LOOP)
    assert_same(g, reps, "cubic +-J, tables in HBM, heat-bath + loop")
    g2 = im.QmcIsingGraph(edges, 1.0, 0.1, 64, 777, nreplicas=R, capacity=1 << 13, couplings=J, cfg_flags=im.CFG_GLOBAL_TABLES | im.CFG_NO_PM_DECODE)
    g2.run(35, 2.0)
    reps2 = [oracle.Replica(oracle.Model(g.nvars, e, list(J[r]), 1.0, 0.1), 1 << 13, 64, 777, r, None) for r in range(R)]
    oracle.batch_timesteps(reps2, 35, [2.0] * R)
    assert_same(g2, reps2, "cubic +-J, tables in HBM, general bond records")
    n = 7
    ints = lat.xxz_ring_interactions(n)
    # (generic models take the general bond table by construction; cluster updates on: the model is Ising-symmetric)
    q = im.Qmc.from_interactions(n, ints, 8, 31, nreplicas=3, capacity=1 << 12)
    assert not q.launch_info()["global_tables"]


def test_cubic_32_full_size_runs_on_the_global_tables_path(oracle):
    """BASELINE configs[4] at its stated lattice size: 32^3 = 32768 variables, 98304 edges, per-replica +-J, Gamma = 1,
    h = 0.1, beta = 4.  The per-variable tables (128 KB of spin bytes at 4 waves) do not fit LDS: the engine must pick the
    HBM-table path by itself.  A few replicas are checked bit for bit against the oracle over the cutoff-growth phase,
    and through the size-independent properties (verify, cutoff growth rule, energy sanity)."""
    import isingmontecarlo_amd as im
    l, R, beta, sweeps = 32, 4, 4.0, 12
    edges = lat.cubic_periodic(l)
    nsite = l ** 3
    rng = np.random.default_rng(32768)
    J = rng.choice([-1.0, 1.0], size=(R, len(edges)))
    cap = 1 << 21
    g = im.QmcIsingGraph(edges, 1.0, 0.1, nsite, 2026, nreplicas=R, capacity=cap, couplings=J)
    info = g.launch_info()
    assert g.nvars == nsite and info["global_tables"] and info["waves_per_replica"] == 4
    e = [ab for ab, _ in edges]
    reps = [oracle.Replica(oracle.Model(nsite, e, list(J[r]), 1.0, 0.1), cap, nsite, 2026, r, None) for r in range(R)]
    g.run(sweeps, beta)
    oracle.batch_timesteps(reps, sweeps, [beta] * R)
    assert_same(g, reps, "32^3 +-J")
    assert g.verify().all()
    n, cut = g.get_n(), g.get_cutoff()
    assert (cut >= n + n // 2).all() or (cut == cut.max()).all()  # qmc_ising.rs:786: cutoff = max(cutoff, n + n/2) after every sweep
    assert (n > 5 * nsite).all() and (n < 40 * nsite).all()       # still growing towards beta * (offset - E0) ~ 4 * 6 per site
    en = -(g.accumulators()[:, 0] / np.maximum(g.accumulators()[:, 1], 1)) / beta + g.get_offsets()
    assert np.isfinite(en).all() and (en / nsite < 4.1 + 1e-9).all() and (en / nsite > -3.0).all()  # offset/N = 3 + 1 + 0.1


@pytest.mark.parametrize("k,flags", [(0, 0), (0, 1), (2, 0), (4, 1)])
def test_segment_labelling_on_the_diagonal_kernel(oracle, k, flags):
    """ISINGMC_CFG_FAST_LABEL (experimental): the trimmed diagonal kernel labels the worldline segments one tile late and
    the cluster update of the same timestep only runs the union-find over the handed-over segment pairs.  Same Markov
    chain as every other path: bit-exact against the oracle, with and without a directed loop in between, and when a
    primitive is called out of band (the hand-over must then be ignored, not used stale)."""
    import isingmontecarlo_amd as im
    edges = lat.two_d_ferro(16)
    R, beta = 12, 4.0
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 256, 1 << 15, 4711, R, waves=4 if k else 0, k=k, cfg_flags=im.CFG_FAST_LABEL)
    info = g.launch_info()
    assert info["fast_diagonal"] and info["fast_label"]
    g.run(25, beta, sampling_freq=2, flags=flags)
    oracle.batch_timesteps(reps, 25, [beta] * R, 2, flags)
    assert_same(g, reps, "label hand-over")
    g.single_diagonal_step(beta)   # out of band: labels nothing that a later cluster update may use
    nc = g.single_cluster_step(flip_free=True)
    for r, rep in enumerate(reps):
        rep.diagonal_update(beta)
        want = rep.n + rep.n // 2
        if want > rep.cutoff:
            assert rep.set_cutoff(want) == 0
        assert nc[r] == rep.cluster_update(0.5)
        rep.flip_free_spins()
    g.run(6, beta, flags=flags)
    oracle.batch_timesteps(reps, 6, [beta] * R, 1, flags)
    assert_same(g, reps, "label hand-over after out-of-band primitives")
    acc = g.accumulators()
    for r, rep in enumerate(reps):
        assert np.array_equal(acc[r, :7], rep.accumulators()[:7])
    assert g.verify().all()


def test_debug_ops_counters_and_serde_layout(oracle):
    """DebugOps (qmc_debug.rs:10-41) on the device, and the checkpoint in the reference's serde field layout
    (qmc_ising.rs:1010-1028 and members) as a round trip through JSON."""
    import json
    import isingmontecarlo_amd as im
    from isingmontecarlo_amd.serde_format import to_serde, from_serde
    edges = lat.one_d_periodic(6, -1.0)
    R = 3
    g, m, reps = make_pair(oracle, edges, 0.9, 0.3, 6, 1 << 11, 99, R)
    g.run(40, 2.0)
    oracle.batch_timesteps(reps, 40, [2.0] * R)
    d, o = g.count_diagonal_and_off()
    c = g.count_constant_ops()
    n = g.get_n()
    for r, rep in enumerate(reps):
        w = rep.ops()
        occ = w[w != 0]
        nd = int(((occ & 3) == ((occ >> 2) & 3)).sum())
        bond = (occ >> 4).astype(np.int64) - 1
        assert (int(d[r]), int(o[r])) == (nd, len(occ) - nd) and int(d[r]) + int(o[r]) == n[r]
        assert int(c[r]) == int(((bond >= 6) & (bond < 12)).sum())
    ser = json.loads(json.dumps(to_serde(g, 1)))
    assert set(ser) == {"edges", "transverse", "longitudinal", "state", "cutoff", "op_manager", "total_energy_offset", "nvars",
                        "run_rvb_steps", "classical_bonds", "total_rvb_successes", "rvb_clusters_counted", "bond_weights"}
    man = ser["op_manager"]
    assert man["n"] == n[1] and sum(man["bond_counters"]) == n[1] and len(man["var_ends"]) == 6
    first, last = man["p_ends"]
    assert man["ops"][first]["previous_p"] is None and man["ops"][last]["next_p"] is None
    p, seen = first, 0
    while p is not None:  # the p-links visit every op once, in order
        seen += 1
        nxt = man["ops"][p]["next_p"]
        assert nxt is None or nxt > p
        p = nxt
    assert seen == n[1]
    g2 = im.QmcIsingGraph(edges, 0.9, 0.3, 6, 99, nreplicas=1, capacity=1 << 11)
    from_serde(g2, ser, 0)
    assert np.array_equal(g2.export_ops(0), g.export_ops(1)) and np.array_equal(g2.state_ref()[0], g.state_ref()[1])
    assert g2.verify().all()


def test_64x64_cold_end_runs_on_the_hbm_union_find_and_matches(oracle):
    """configs[3]'s cold end: 64x64 at beta = 16.  N + (transverse ops) exceeds both the LDS union-find and 16-bit ids
    (~85,000 cuts), so every sweep takes the 32-bit HBM union-find; ~5e5 slots per replica."""
    edges = lat.two_d_ferro(64)
    R = 2
    g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 4096, 1 << 20, 6416, R)
    g.run(26, 16.0)
    oracle.batch_timesteps(reps, 26, [16.0] * R)
    assert_same(g, reps, "64x64 beta=16")
    assert g.verify().all()
    bonds = (reps[0].ops() >> 4).astype(np.int64) - 1
    ntrans = int(((bonds >= len(edges)) & (bonds < len(edges) + g.nvars)).sum())
    assert g.nvars + ntrans > 65535 and reps[0].cutoff > 300000


# The dedicated cluster kernel (csrc/sse_cluster.hip.h) against the oracle, in the situations that take its different
# branches: h = 0 and h != 0 (frozen clusters), slots_per_lane 4 and 2, with a directed loop in front, single primitives, replicas
# it leaves to the general kernel (no cut / no op at high temperature, ids beyond the union-find planned before the first
# launch), and the same runs through the general kernel only (CFG_NO_LEAN_CLUSTER) for the A-B switch.
@pytest.mark.parametrize("k", [0, 2])
@pytest.mark.parametrize("nolean", [False, True, "inplace"])  # "inplace": the dedicated kernel without deferred flips
@pytest.mark.parametrize("name,edges,gamma,h,beta,cut0", [
    ("ferro16", lat.two_d_ferro(16), 1.0, 0.0, 4.0, 256),
    ("ferro8_long", lat.two_d_ferro(8), 1.0, 0.25, 3.0, 64),
    ("ring8_hot", lat.one_d_periodic(8), 0.4, 0.0, 0.3, 8),
    ("villain8_cold", lat.two_d_periodic(8), 1.0, 0.0, 8.0, 64),
])
def test_dedicated_cluster_kernel(oracle, name, edges, gamma, h, beta, cut0, nolean, k):
    import isingmontecarlo_amd as im
    R = 12
    cfgf = {False: 0, True: im.CFG_NO_LEAN_CLUSTER, "inplace": im.CFG_NO_DEFERRED_FLIPS}[nolean]
    g, m, reps = make_pair(oracle, edges, gamma, h, cut0, 1 << 15, 4711, R, k=k, cfg_flags=cfgf)
    g.run(12, beta, sampling_freq=2)
    oracle.batch_timesteps(reps, 12, [beta] * R, 2, 0)
    assert g.launch_info()["lean_cluster"] == (nolean is not True)
    assert_same(g, reps, f"{name} timesteps")
    for it in range(6):
        g.single_diagonal_step(beta)
        for rep in reps:
            rep.diagonal_update(beta)
            want = rep.n + rep.n // 2
            if want > rep.cutoff:
                assert rep.set_cutoff(want) == 0
        nc = g.single_cluster_step(flip_free=False)
        for r, rep in enumerate(reps):
            assert nc[r] == rep.cluster_update(0.5), f"{name}: cluster count differs it={it} r={r}"
        assert_same(g, reps, f"{name} cluster it={it}")
    g.run(10, beta, flags=1, sampling_freq=3)
    oracle.batch_timesteps(reps, 10, [beta] * R, 3, 1)
    assert_same(g, reps, f"{name} + loop")
    acc = g.accumulators()
    for r, rep in enumerate(reps):
        assert np.array_equal(acc[r, :7], rep.accumulators()[:7]), f"accumulators differ r={r}: {acc[r]} vs {rep.accumulators()}"
    assert g.verify().all()


def test_config0_sixteen_site_ring(oracle):
    """BASELINE configs[0] / benches/end_to_end.rs:45-60 (one_d): the 16-site periodic chain J = +1, Gamma = 1, h = 0, beta = 1,
    initial cutoff 16, 1000 warm-up timesteps then single timesteps — the reference's plumbing case, here as a parity case:
    every timestep's operator words, states and counters against the oracle, through the default (trimmed / dedicated) kernels
    and through the general ones."""
    import isingmontecarlo_amd as im
    edges = lat.one_d_periodic(16)
    for cfgf in (0, im.CFG_NO_FAST_DIAG | im.CFG_NO_LEAN_CLUSTER):
        g, m, reps = make_pair(oracle, edges, 1.0, 0.0, 16, 4096, 1234, 3, cfg_flags=cfgf)
        g.run(1000, 1.0)
        oracle.batch_timesteps(reps, 1000, [1.0] * 3)
        assert_same(g, reps, "one_d warm-up")
        for it in range(20):
            g.run(1, 1.0)
            for rep in reps:
                rep.timesteps(1, 1.0)
            assert_same(g, reps, f"one_d timestep {it}")
        acc = g.accumulators()
        for r, rep in enumerate(reps):
            assert np.array_equal(acc[r, :7], rep.accumulators()[:7])
        assert g.verify().all()
        # the energy estimator of the plumbing case is sane: E/N of the 16-site TFIM ring at beta = 1 lies between the classical
        # bound and the high-temperature value
        e = g.get_energy_for_average_n(acc[:, 0] / np.maximum(acc[:, 1], 1), 1.0) / 16.0
        assert (-1.6 < e).all() and (e < -0.3).all(), e
