#!/usr/bin/env python3
"""Copies the summaries produced by tools/collect_profiles.sh from gpurun_out/profiles_raw into profiles/ (timed kernels
only for the counter files) and rewrites profiles/r01_traffic.json."""
import csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW, OUT = os.path.join(ROOT, "gpurun_out", "profiles_raw"), os.path.join(ROOT, "profiles")
shutil.copy(glob.glob(os.path.join(RAW, "stats", "*", "*_kernel_stats.csv"))[0], os.path.join(OUT, "r01_kernel_stats.csv"))
for f in glob.glob(os.path.join(RAW, "r01_*.json")):
    shutil.copy(f, OUT)
timed = lambda k: "sweep_kernel" in k and ", 0, " in k
def keep(src_dir, out):
    src = sorted(glob.glob(os.path.join(RAW, src_dir, "*", "*_counter_collection.csv")))[-1]
    rows = list(csv.reader(open(src))); hdr = rows[0]; ki = hdr.index("Kernel_Name")
    csv.writer(open(os.path.join(OUT, out), "w")).writerows([hdr] + [r for r in rows[1:] if timed(r[ki])])
    return [r for r in rows[1:] if timed(r[ki])], hdr
f_rows, hdr = keep("pmc_fetch", "r01_pmc_fetch_size.csv")
w_rows, _ = keep("pmc_write", "r01_pmc_write_size.csv")
keep("pmc_sq1", "r01_pmc_sq_per_pass.csv"); keep("pmc_sq2", "r01_pmc_sq_lds_per_pass.csv")
ki, vi = hdr.index("Kernel_Name"), hdr.index("Counter_Value")
def mean_kb(rows, passes):
    v = [float(r[vi]) for r in rows if r[ki].split("(")[0].rstrip(">").rstrip().endswith(", %d" % passes)]
    return sum(v) / len(v)
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --no-cpu-baseline --steps 10 --warmup 2`; "
                     "FETCH_SIZE doubled for coalesced streaming reads (MI355X_MICROARCH.md, HBM section); bytes per launch",
           "workload": "configs[1] 32x32 beta=16 R=1024"}
for p in (1, 2):
    traffic["PASSES=%d" % p] = int((2 * mean_kb(f_rows, p) + mean_kb(w_rows, p)) * 1024)
json.dump(traffic, open(os.path.join(OUT, "r01_traffic.json"), "w"), indent=1)
print(traffic)
