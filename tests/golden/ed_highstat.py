#!/usr/bin/env python3
"""High-statistics check of the oracle against exact diagonalisation (the run DESIGN.md quotes for the "within Monte-Carlo
error" claim; the test suite repeats it at 1/50 of the statistics with a 4-sigma gate).

For every system and temperature of tests/golden/ed_tfim.json: R replicas x SWEEPS measured sweeps (after SWEEPS/10 of
equilibration) with the plain QmcIsingGraph::timestep, deviation of energy, |m|, m^2 and <sigma_x> from the exact value in
units of the standard error over replicas.  Writes tests/golden/ed_highstat_r02.json (FLAGS = 0) or
tests/golden/ed_highstat_r03_flags<FLAGS>.json: FLAGS as isingmc_timesteps takes them — 1 = a directed loop per step
(Qmc::timestep with loop updates), 4 = heat-bath diagonal update, 8 = RVB sweeps (QmcIsingGraph::set_run_rvb: systems whose
couplings all have one magnitude only, qmc_ising.rs:435-447; the others are skipped).

usage: python tests/golden/ed_highstat.py [R] [SWEEPS] [FLAGS]      (defaults 64, 400000, 0; ~10 minutes on 8 cores)"""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import _oracle as O

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SWEEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
FLAGS = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ED = json.load(open(os.path.join(HERE, "ed_tfim.json")))
out, worst = [], 0.0
t0 = time.time()
for case in ED:
    if (FLAGS & 8) and len({abs(j) for j in case["J"]}) != 1:
        continue  # set_run_rvb refuses couplings of different magnitudes
    m = O.Model(case["nvars"], case["edges"], case["J"], case["gamma"], case["h"])
    for res in case["results"]:
        beta = res["beta"]
        reps = [O.Replica(m, 4096, case["nvars"], 777, r) for r in range(R)]
        O.batch_timesteps(reps, SWEEPS // 10, [beta] * R, 1, FLAGS)
        for r in reps:
            r.reset_accumulators()
        O.batch_timesteps(reps, SWEEPS, [beta] * R, 1, FLAGS)
        acc = np.array([r.accumulators() for r in reps], dtype=np.float64)
        n = case["nvars"]
        obs = {"energy": -(acc[:, 0] / acc[:, 1]) / beta + m.offset, "abs_m": acc[:, 2] / acc[:, 1] / n,
               "m2": acc[:, 3] / acc[:, 1] / n ** 2, "sx": acc[:, 6] / acc[:, 1] / (beta * case["gamma"] * n) - 1.0}
        row = {"system": case["name"], "beta": beta}
        for k, x in obs.items():
            mu, se = float(x.mean()), float(x.std(ddof=1) / np.sqrt(R))
            dev = (mu - res[k]) / se if se > 0 else 0.0
            row[k] = {"mc": mu, "stderr": se, "exact": res[k], "deviation_sigma": dev, "relative_stderr": se / abs(res[k]) if res[k] else None}
            worst = max(worst, abs(dev))
        out.append(row)
        print(row["system"], beta, {k: round(row[k]["deviation_sigma"], 2) for k in obs}, flush=True)
devs = np.array([row[k]["deviation_sigma"] for row in out for k in ("energy", "abs_m", "m2", "sx")])
summary = {"within_1_sigma": float((np.abs(devs) < 1).mean()), "within_2_sigma": float((np.abs(devs) < 2).mean()),
           "rms_sigma": float(np.sqrt((devs ** 2).mean())), "mean_sigma": float(devs.mean())}
json.dump({"replicas": R, "sweeps": SWEEPS, "seed": 777, "flags": FLAGS, "worst_abs_deviation_sigma": worst, "n_comparisons": 4 * len(out),
           "wall_s": time.time() - t0, "rows": out, "summary": summary},
          open(os.path.join(HERE, "ed_highstat_r02.json" if FLAGS == 0 else f"ed_highstat_r03_flags{FLAGS}.json"), "w"), indent=1)
print(summary)
print("worst |deviation| =", worst, "sigma over", 4 * len(out), "comparisons")
