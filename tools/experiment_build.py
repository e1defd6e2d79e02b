#!/usr/bin/env python3
"""Dev tool: build timing-experiment variants of the library (results are WRONG by construction; never shipped).
usage: python tools/experiment_build.py NAME=-DMACRO[,-DMACRO2] ...   ->  isingmontecarlo_amd/csrc/build/exp/lib_NAME.so
Load one with ISINGMC_HIP_LIB=<path> (honoured by tools/ only, see tools/profile_passes.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "isingmontecarlo_amd", "csrc")
OUT = os.path.join(CSRC, "build", "exp")
os.makedirs(OUT, exist_ok=True)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
others = [os.path.join(CSRC, "build", f) for f in ["isingmc_hip.o", "sweep_w1.o", "sweep_w6.o", "sweep_w8.o", "sweep_w16.o"]]
procs = []
for spec in sys.argv[1:]:
    name, _, defs = spec.partition("=")
    obj = os.path.join(OUT, f"w4_{name}.o")
    cmd = ["hipcc"] + FLAGS + [d for d in defs.split(",") if d] + ["-c", "sweep_w4.hip", "-o", obj]
    procs.append((name, obj, subprocess.Popen(cmd, cwd=CSRC)))
for name, obj, pr in procs:
    assert pr.wait() == 0, name
    lib = os.path.join(OUT, f"lib_{name}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others)
    print(lib)
