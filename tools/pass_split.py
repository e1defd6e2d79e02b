#!/usr/bin/env python3
"""Dev tool: per-launch split (diagonal launch / everything else) of whole timesteps at the headline geometry, with and without
the segment-labelling hand-over.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _lattices as lat
import isingmontecarlo_amd as im
if os.environ.get("ISINGMC_HIP_LIB"):  # timing-experiment builds (tools/experiment_build.py)
    im._build.LIB = os.environ["ISINGMC_HIP_LIB"]; im._build.build = lambda *a, **k: im._build.LIB
L, R, beta = 32, 1024, 16.0
cap = 1 << 18
for name, cf, kw in (("label+lite K=4", im.CFG_FAST_LABEL, {}), ("label+lite K=2", im.CFG_FAST_LABEL, dict(waves_per_replica=4, slots_per_lane=2, waves_offdiag=16)),
                     ("no label", 0, {}), ("dense list", im.CFG_COMPACT, {}), ("general diag", im.CFG_NO_FAST_DIAG, {})):
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    g = im.QmcIsingGraph(lat.two_d_ferro(L), 1.0, 0.0, L * L, 1234, nreplicas=R, capacity=cap, cfg_flags=cf, **kw)
    g.run(60, beta)
    for flags in (0, im.FLAG_LOOP):
        g.run(5, beta, flags=flags)
        g.run(20, beta, flags=flags)
        (d, o), (nd, no) = g.last_pass_ms()
        print(f"{name:14s} flags={flags}: diag {d/nd:.3f} ms  rest {o/no:.3f} ms  total {g.last_kernel_ms()[0]/20:.3f} ms  info {g.launch_info()['fast_label']}", flush=True)
    assert g.verify().all()
    del g
