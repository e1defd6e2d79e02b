#!/usr/bin/env python3
"""Dev experiment: diagonal-pass time vs workgroups per CU (LDS limited through lds_uf_ids_limit)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _lattices as lat
import isingmontecarlo_amd as im
L, R, beta = 32, 1024, 16.0
for k in (4, 2, 1):
    for lim in (0, 2000):
        g = im.QmcIsingGraph(lat.two_d_ferro(L), 1.0, 0.0, L * L, 1234, nreplicas=R, capacity=1 << 18,
                             waves_per_replica=8, slots_per_lane=k, lds_uf_ids_limit=lim)
        g.run(50, beta)
        ms = []
        for _ in range(5):
            g.single_diagonal_step(beta); ms.append(g.last_kernel_ms()[0])
        print(f"K={k} uf_limit={lim} lds={g.launch_info()['lds_bytes']} M={g.get_cutoff().mean():.0f} diag {np.median(ms):.3f} ms")
        g.close()
