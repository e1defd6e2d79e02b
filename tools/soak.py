import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, _lattices as lat, isingmontecarlo_amd as im
g = im.QmcIsingGraph(lat.two_d_ferro(32), 1.0, 0.0, 1024, 99, nreplicas=1024, capacity=1 << 18)
g.run(100, 16.0)
g.reset_accumulators()
t0 = time.time()
for blk in range(int(__import__("os").environ.get("SOAK_BLOCKS", "15"))):
    g.run(100, 16.0, flags=im.FLAG_LOOP)
    ok = g.verify()
    assert ok.all(), (blk, np.where(~ok)[0][:10])
acc = g.accumulators().astype(float)
e = -(acc[:, 0] / acc[:, 1]) / 16.0 + g.get_offset()
print("soak ok: %d sweeps x 1024 replicas in %.1f s, E/N = %.6f +- %.6f" % (100 * int(__import__("os").environ.get("SOAK_BLOCKS", "15")), time.time() - t0, e.mean() / 1024, e.std(ddof=1) / 32 / 1024))
