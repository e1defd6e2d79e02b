"""In-tree build of the HIP extension (gfx950 only).  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libisingmc_hip.so")
SOURCES = ["isingmc_hip.hip", "sweep_w1.hip", "sweep_w4.hip", "sweep_w6.hip", "sweep_w8.hip", "sweep_w16.hip"]
HEADERS = ["sse_device.hip.h", os.path.join("..", "..", "include", "isingmc_hip.h"),
           os.path.join("..", "..", "include", "sse_format.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
if os.environ.get("SSE_MIN_WAVES"):  # experiment: force the register budget for N waves per SIMD (W = 8 kernels)
    FLAGS.append("-DSSE_MIN_WAVES_PER_SIMD=" + os.environ["SSE_MIN_WAVES"])
if os.environ.get("SSE_PHASE_TIMING"):  # diagnostic build: in-kernel phase stamps (never benchmarked)
    FLAGS.append("-DSSE_PHASE_TIMING")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


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
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
