"""In-tree build of the HIP extension (gfx950 only).  hipcc cross-compiles without a GPU."""
import glob
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libisingmc_hip.so")
STAMP = os.path.join(CSRC, "build", "build.sha256")
SOURCES = ["isingmc_hip.hip", "sweep_fast.hip", "sweep_cluster.hip", "sweep_rvb.hip", "sweep_w1.hip", "sweep_w4.hip", "sweep_w6.hip", "sweep_w8.hip", "sweep_w16.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
LINK_LIBS = ["-ldl"]  # RCCL (parallel-tempering neighbour exchange) is dlopen()ed on first use, not linked
if os.environ.get("SSE_MIN_WAVES"):  # experiment: force the register budget for N waves per SIMD (W = 8 kernels)
    FLAGS.append("-DSSE_MIN_WAVES_PER_SIMD=" + os.environ["SSE_MIN_WAVES"])
if os.environ.get("SSE_PHASE_TIMING"):  # diagnostic build: in-kernel phase stamps (never benchmarked)
    FLAGS.append("-DSSE_PHASE_TIMING")


def _inputs():
    """Every file the library is built from: all HIP sources and headers of csrc/ and the public headers."""
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                   glob.glob(os.path.join(INCLUDE, "*.h")))
    return files


def source_hash():
    """sha256 over the contents of every input file plus the compiler flags (so that flag changes from the environment,
    SSE_MIN_WAVES / SSE_PHASE_TIMING, force a rebuild too)."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + LINK_LIBS).encode())
    for p in _inputs():
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB):
        return True
    if not os.path.isdir(CSRC):  # a source-less deployment ships only the library
        return False
    try:
        with open(STAMP) as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def build(force=False, verbose=False):
    """Compile libisingmc_hip.so next to this file. Returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:  # one translation unit per wave count: compile them in parallel
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, cwd=CSRC)))
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + LINK_LIBS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
