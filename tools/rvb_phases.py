#!/usr/bin/env python3
"""Where the RVB sweep spends its time (diagnostic build: SSE_PHASE_TIMING=1 in the environment of the BUILD).  GPU only."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _lattices as lat
import isingmontecarlo_amd as im
if os.environ.get("ISINGMC_HIP_LIB"):  # timing-experiment builds (tools/experiment_build.py)
    im._build.LIB = os.environ["ISINGMC_HIP_LIB"]; im._build.build = lambda *a, **k: im._build.LIB
ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=32); ap.add_argument("--beta", type=float, default=16.0)
ap.add_argument("--replicas", type=int, default=1024); ap.add_argument("--equilibrate", type=int, default=60)
ap.add_argument("--cfg", type=lambda x: int(x, 0), default=0, help="ISINGMC_CFG_* flags (2048: the fused kernel)")
a = ap.parse_args()
L, R, beta = a.L, a.replicas, a.beta
cap = 1 << int(np.ceil(np.log2(2.0 * beta * 5.2 * L * L + 4 * L * L)))
g = im.QmcIsingGraph(lat.two_d_ferro(L), 1.0, 0.0, L * L, 1234, nreplicas=R, capacity=cap, cfg_flags=a.cfg)
g.run(a.equilibrate, beta)
g.run(3, beta, flags=im.FLAG_RVB)
ms = []
for _ in range(3):
    s, _ = g.single_rvb_sweep(); ms.append(g.last_kernel_ms()[0])
info = g.launch_info()
print("rvb sweep: %.2f ms (min %.2f), successes per replica %.1f of %d attempts; %s" % (np.median(ms), min(ms), np.mean(s), (L * L + 1) // 2,
      f"growth + main launches, main with {info['rvb_main_waves']} waves per replica" if info["rvb_split"] else "fused kernel"))
tk = g.debug_phase_ticks()
if tk.any():
    g.debug_phase_ticks(reset=True)
    g.single_rvb_sweep()
    t = g.debug_phase_ticks().astype(float).mean(axis=0) * 10e-3
    names = {6: "constants table", 7: "growth (small areas)", 13: "growth (large area)", 8: "states at window starts", 9: "gathers", 10: "replay: probability", 11: "accept", 12: "replay: mutation"}
    if not info["rvb_split"]: names.pop(13)
    print("inside the fetch, us: loads + matching %.0f, compaction %.0f, longer look-back %.0f, states %.0f" % tuple(t[:4]))
    print("per replica, us: " + ", ".join(f"{names[k]} {t[k]:.0f}" for k in sorted(names)) + f"; sum {sum(t[k] for k in names):.0f}; growers {t[15] / 10e-3:.0f}, LDS words {t[14] / 10e-3:.0f}, constants {t[13] / 10e-3:.0f}, table at {t[5] / 10e-3:.0f}")
