#!/usr/bin/env python3
"""Long RVB parity run against the C oracle (GPU + the oracle built by __graft_entry__.build()): whole timesteps with RVB
sweeps on a 16x16 lattice, compared op for op every block — about 10^6 attempts, enough for the rare growth paths (clusters of
more than 16 members, candidate sets that outgrow a small growth area, long windows).  Dev tool; prints one line."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, _lattices as lat, _oracle, isingmontecarlo_amd as im
L, R, beta = 16, 8, 4.0
blocks, per = int(os.environ.get("SOAK_BLOCKS", "20")), int(os.environ.get("SOAK_SWEEPS", "50"))
edges = lat.two_d_ferro(L)
_oracle.build(); oracle = _oracle
e, j = lat.split(edges)
m = oracle.Model(L * L, e, j, 1.0, 0.0)
cap = 1 << 14
g = im.QmcIsingGraph(edges, 1.0, 0.0, 512, 777, nreplicas=R, capacity=cap)
reps = [oracle.Replica(m, cap, 512, 777, r, None) for r in range(R)]
t0 = time.time()
for blk in range(blocks):
    g.run(per, beta, sampling_freq=1, flags=im.FLAG_RVB)
    for rep in reps:
        rep.timesteps(per, beta, 1, im.FLAG_RVB)
    n = g.get_n()
    for r, rep in enumerate(reps):
        assert n[r] == rep.n and np.array_equal(g.state_ref()[r], rep.state()) and np.array_equal(g.export_ops(r), rep.ops()), (blk, r)
assert g.verify().all()
print("rvb soak ok: %d sweeps x %d replicas x %d attempts bit-exact against the oracle in %.1f s" % (blocks * per, R, (L * L + 1) // 2, time.time() - t0))
