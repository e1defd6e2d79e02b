#!/usr/bin/env python3
"""Per-pass timing of the sweep at equilibrium (HIP events through the C ABI). Dev tool, GPU only."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _lattices as lat
import isingmontecarlo_amd as im
if os.environ.get("ISINGMC_HIP_LIB"):  # timing-experiment builds (tools/experiment_build.py)
    im._build.LIB = os.environ["ISINGMC_HIP_LIB"]; im._build.build = lambda *a, **k: im._build.LIB

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=32); ap.add_argument("--beta", type=float, default=16.0)
ap.add_argument("--replicas", type=int, default=1024); ap.add_argument("--waves", type=int, default=0)
ap.add_argument("--equilibrate", type=int, default=60); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--woff", type=int, default=0); ap.add_argument("--k", type=int, default=0); ap.add_argument("--nolds", action="store_true")
a = ap.parse_args()
L, R, beta = a.L, a.replicas, a.beta
n_est = beta * 5.2 * L * L
cap = 1 << int(np.ceil(np.log2(2.0 * n_est + 4 * L * L)))
g = im.QmcIsingGraph(lat.two_d_ferro(L), 1.0, 0.0, L * L, 1234, nreplicas=R, capacity=cap, waves_per_replica=a.waves, slots_per_lane=a.k, cfg_flags=1 if a.nolds else 0, waves_offdiag=a.woff)
g.run(a.equilibrate, beta)
print("launch", g.launch_info(), "mean n", g.get_n().mean(), "mean M", g.get_cutoff().mean())
def t(fn, name):
    ms = []
    for _ in range(a.reps):
        fn(); ms.append(g.last_kernel_ms()[0])
    print(f"{name:28s} {np.median(ms):8.3f} ms  (min {min(ms):.3f})")
t(lambda: g.single_diagonal_step(beta), "diagonal")
t(lambda: g.single_cluster_step(flip_free=False), "cluster")
t(lambda: g.loop_update(), "loop")
t(lambda: g.flip_free_spins(), "free (with touch scan)")
t(lambda: g.run(1, beta), "timestep diag+cluster+free")
t(lambda: g.run(1, beta, flags=im.FLAG_LOOP), "timestep +loop")
t(lambda: g.run(1, beta, flags=im.FLAG_HEATBATH), "timestep heatbath")
if not os.environ.get("SKIP_RVB"):
    t(lambda: g.single_rvb_sweep(), "rvb sweep (N+1)/2 attempts")
    t(lambda: g.run(1, beta, flags=im.FLAG_RVB), "timestep with rvb")
t(lambda: g.run(10, beta), "10 timesteps fused")

tk = g.debug_phase_ticks()
if tk.any():
    g.debug_phase_ticks(reset=True)
    g.single_cluster_step(flip_free=False)
    tk = g.debug_phase_ticks().astype(float).mean(axis=0) * 10e-3  # us
    print("cluster phases (us per replica): init %.1f build %.1f join %.1f flatten %.1f coins %.1f apply %.1f" % tuple(tk[:6]))
    g.debug_phase_ticks(reset=True)
    g.single_diagonal_step(beta)
    raw = g.debug_phase_ticks().astype(float).mean(axis=0)
    print("diag phases (us per replica): compute %.1f rounds %.1f commit+loop %.1f ; tiles %.0f rounds %.0f" % (raw[8] * 10e-3, raw[9] * 10e-3, raw[11] * 10e-3, raw[12], raw[13]))

    # diagnostic experiment: cluster build scan without unions (configuration is NOT advanced afterwards)
    import ctypes as _C
    out = np.zeros((g.nreplicas, 16), dtype=np.uint64)
    g._lib.isingmc_debug_phase_ticks(g._h, out.ctypes.data_as(_C.POINTER(_C.c_uint64)), 16 + 1)
    g.single_cluster_step(flip_free=False)
    g._lib.isingmc_debug_phase_ticks(g._h, out.ctypes.data_as(_C.POINTER(_C.c_uint64)), 0)
    tk = out.astype(float).mean(axis=0) * 10e-3
    print("cluster phases WITHOUT unions (us): build %.1f apply %.1f" % (tk[1], tk[5]))
