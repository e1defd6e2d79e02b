#!/usr/bin/env python3
"""Dev tool (GPU, diagnostic build only): time one pass with parts switched off through the dbg flags.
Every experiment starts from the same equilibrated state (fresh batch, deterministic equilibration).
usage: ISINGMC_HIP_LIB=<diagnostic lib> python3 tools/attribute.py [--pass diag|cluster] flag[,flag...] ..."""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _lattices as lat
import isingmontecarlo_amd as im
if os.environ.get("ISINGMC_HIP_LIB"):
    im._build.LIB = os.environ["ISINGMC_HIP_LIB"]; im._build.build = lambda *a, **k: im._build.LIB
ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=32); ap.add_argument("--beta", type=float, default=16.0)
ap.add_argument("--replicas", type=int, default=1024); ap.add_argument("--equilibrate", type=int, default=40)
ap.add_argument("--pass", dest="which", default="diag"); ap.add_argument("flags", nargs="*", default=["0"])
a = ap.parse_args()
L, R, beta = a.L, a.replicas, a.beta
cap = 1 << int(np.ceil(np.log2(2.0 * beta * 5.2 * L * L + 4 * L * L)))
for spec in a.flags:
    f = 0
    for x in spec.split(","): f |= int(x, 0)
    g = im.QmcIsingGraph(lat.two_d_ferro(L), 1.0, 0.0, L * L, 1234, nreplicas=R, capacity=cap)
    g.run(a.equilibrate, beta)
    out = np.zeros((R, 16), dtype=np.uint64)
    g._lib.isingmc_debug_phase_ticks(g._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), 16 * f + 1)
    if a.which == "diag": g.single_diagonal_step(beta)
    else: g.single_cluster_step(flip_free=False)
    ms = g.last_kernel_ms()[0]
    g._lib.isingmc_debug_phase_ticks(g._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), 0)
    tk = out.astype(float).mean(axis=0) * 10e-3
    if a.which == "diag":
        print(f"flags {f:#04x}: {ms:7.3f} ms   compute {tk[8]:.0f} rounds {tk[9]:.0f} commit {tk[11]:.0f} us; rounds/tile {out[:,13].sum()/max(1,out[:,12].sum()):.2f}")
    else:
        print(f"flags {f:#04x}: {ms:7.3f} ms   init %.0f build %.0f join %.0f flatten %.0f coins %.0f apply %.0f us" % tuple(tk[:6]))
        c = out.astype(float).mean(axis=0)
        if c[11] > 0: print("   wave 3 of a replica: %.0f tiles, %.0f rows with a union, %.0f shader cycles in the union blocks (%.0f of them in the serial routine); need lanes %.0f, successful write-backs %.0f, last (old<<32|expected) %x" % (c[11], c[8], c[9], c[10], c[12], c[13], int(out[0, 14])))
    del g
