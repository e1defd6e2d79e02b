#!/usr/bin/env python3
"""Does the +-J random-bond model with a longitudinal field (BASELINE configs[4]) need RVB sweeps?  With h != 0 every cluster
that holds a longitudinal op is frozen (qmc_ising.rs:759-775), so QmcIsingGraph::timestep without RVB moves the spins only
through free-spin flips and the clusters that happen to hold no longitudinal op; the reference offers RVB sweeps
(qmc_ising.rs:705-752) for exactly this case.  This tool measures it on the CPU oracle (the HIP path is bit-identical to it):
L^3 +-J, Gamma = 1, h = 0.1, one disorder realisation per replica, with and without RVB sweeps — integrated autocorrelation
times of the operator count n (energy estimator) and of the magnetisation per site, fraction of p=0 spins that ever flip, and
the acceptance of the RVB attempts.  usage: python tools/ergodicity_pmj.py [L=8] [beta=4] [sweeps=4000] [replicas=8]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
import _lattices as lat

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
beta = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
R = int(sys.argv[4]) if len(sys.argv) > 4 else 8
O.build()
edges = lat.cubic_periodic(L)
e, _ = lat.split(edges)
N = L ** 3
rng = np.random.default_rng(20261004)
J = rng.choice([-1.0, 1.0], size=(R, len(edges)))
cap = 1 << int(np.ceil(np.log2(2.0 * beta * (len(edges) * 1.3 + N * 1.2) + 4 * N)))


def tau_int(x):
    """integrated autocorrelation time with the usual self-consistent window (window = 6 tau)"""
    x = np.asarray(x, dtype=np.float64) - np.mean(x)
    if not x.any():
        return float("nan")
    f = np.fft.rfft(x, 2 * len(x))
    ac = np.fft.irfft(f * np.conj(f))[:len(x)]
    ac /= ac[0]
    tau = 0.5
    for w in range(1, len(x) // 4):
        tau += ac[w]
        if w >= 6 * tau:
            break
    return float(tau)


out = {}
for name, flags in (("no_rvb", 0), ("rvb", 8)):
    models = [O.Model(N, e, list(J[r]), 1.0, 0.1) for r in range(R)]
    reps = [O.Replica(models[r], cap, N, 4711, r) for r in range(R)]
    O.batch_timesteps(reps, T // 4, [beta] * R, 1, flags)  # equilibrate
    ns, ms, first = [], [], np.array([rep.state() for rep in reps])
    ever = np.zeros_like(first, dtype=bool)
    for t in range(T):
        O.batch_timesteps(reps, 1, [beta] * R, 1, flags)
        st = np.array([rep.state() for rep in reps])
        ever |= st != first
        ns.append([rep.n for rep in reps])
        ms.append((2.0 * st.sum(axis=1) - N) / N)
    ns, ms = np.array(ns, dtype=np.float64), np.array(ms)
    out[name] = {"tau_int_n_sweeps": float(np.nanmean([tau_int(ns[:, r]) for r in range(R)])),
                 "tau_int_m_sweeps": float(np.nanmean([tau_int(ms[:, r]) for r in range(R)])),
                 "fraction_of_p0_spins_that_flipped_at_least_once": float(ever.mean()),
                 "mean_abs_m": float(np.abs(ms).mean()), "mean_n": float(ns.mean()),
                 "energy_per_site": float(np.mean([-(ns[:, r].mean()) / beta + models[r].offset for r in range(R)]) / N)}
    print(name, out[name], flush=True)
print(json.dumps({"workload": f"{L}^3 +-J cubic, Gamma=1, h=0.1, beta={beta}, {R} disorder realisations x {T} measured sweeps (C oracle)", **out}))
