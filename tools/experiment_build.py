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
TU = os.environ.get("EXP_TU", "sweep_w4")  # the translation unit rebuilt with the macros (the others come from the product build)
others = [os.path.join(CSRC, "build", f + ".o") for f in ["isingmc_hip", "sweep_fast", "sweep_cluster", "sweep_rvb", "sweep_w1", "sweep_w4", "sweep_w6", "sweep_w8", "sweep_w16"] if f != TU]
procs = []
for spec in sys.argv[1:]:
    name, _, defs = spec.partition("=")
    obj = os.path.join(OUT, f"{TU}_{name}.o")
    cmd = ["hipcc"] + FLAGS + [d for d in defs.split(",") if d] + ["-c", TU + ".hip", "-o", obj]
    procs.append((name, obj, subprocess.Popen(cmd, cwd=CSRC)))
for name, obj, pr in procs:
    assert pr.wait() == 0, name
    lib = os.path.join(OUT, f"lib_{name}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others + ["-ldl"])
    print(lib)
