#!/usr/bin/env python3
"""When and where every workgroup of the diagonal launch ran (diagnostic build: EXP_TU=sweep_fast python tools/experiment_build.py
timeline=-DSSE_WG_TIMELINE [timeline_norot=-DSSE_WG_TIMELINE,-DSSE_ROTATE_PRIO=0u]; ISINGMC_HIP_LIB=<that library>).  Start / end are
s_memrealtime stamps of thread 0, the place is HW_ID / XCC_ID.  Shows the order in which the four workgroups of a CU finish
(the issue arbiter serves the oldest wave first).  GPU only."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, _lattices as lat, isingmontecarlo_amd as im
if os.environ.get("ISINGMC_HIP_LIB"):
    im._build.LIB = os.environ["ISINGMC_HIP_LIB"]; im._build.build = lambda *a, **k: im._build.LIB
L, R, beta = 32, 1024, 16.0
cap = 1 << int(np.ceil(np.log2(2.0 * beta * 5.2 * L * L + 4 * L * L)))
g = im.QmcIsingGraph(lat.two_d_ferro(L), 1.0, 0.0, L * L, 1234, nreplicas=R, capacity=cap)
g.run(40, beta, flags=im.FLAG_LOOP)
out = np.zeros((R, 16), dtype=np.uint64)
for flags, what in ((im.FLAG_LOOP, "diagonal pass + directed loop"), (0, "diagonal pass alone")):
    g.run(1, beta, flags=flags)
    g._lib.isingmc_debug_phase_ticks(g._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), 16 * 32 + 1)
    g.run(1, beta, flags=flags)
    g._lib.isingmc_debug_phase_ticks(g._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), 0)
    st, en = out[:, 0].astype(np.int64), out[:, 1].astype(np.int64)
    if not st.any():
        sys.exit("no stamps: this library was not built with -DSSE_WG_TIMELINE")
    t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0
    hw, xcc = out[:, 2].astype(np.int64), out[:, 3].astype(np.int64) & 0xF
    key = xcc * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF)
    uk, cnt = np.unique(key, return_counts=True)
    ends = np.array([np.sort(en[key == k]) for k in uk if (key == k).sum() == 4])
    print(f"{what}: {R} workgroups on {len(uk)} CUs ({cnt.min()}-{cnt.max()} per CU), all started within {st.max():.1f} us; "
          f"ends {en.min():.0f} .. {en.max():.0f} us, mean residence {np.mean(en - st):.0f} us")
    print("   the four workgroups of a CU end at (mean over the CUs, us): " + " / ".join(f"{x:.0f}" for x in ends.mean(axis=0)) +
          f"; spread within a CU {np.mean(ends[:, -1] - ends[:, 0]):.0f} us")
