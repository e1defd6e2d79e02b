#!/usr/bin/env python3
"""Copies the summaries produced by tools/collect_profiles.sh from gpurun_out/profiles_raw into profiles/ (timed kernels
only for the counter files) and rewrites profiles/<round>_traffic.json."""
import csv, glob, json, os, re, shutil, sys
R = os.environ.get("ROUND", "r03")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW, OUT = os.path.join(ROOT, "gpurun_out", "profiles_raw"), os.path.join(ROOT, "profiles")
# (gpurun merges every call's output: pick this round's stats file, the one that saw the trimmed diagonal kernel)
stats = [f for f in sorted(glob.glob(os.path.join(RAW, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime) if "sweep_fast_kernel" in open(f).read()]  # newest last
shutil.copy(stats[-1], os.path.join(OUT, R + "_kernel_stats.csv"))
stats_rvb = sorted(glob.glob(os.path.join(RAW, "stats_rvb", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if stats_rvb:
    shutil.copy(stats_rvb[-1], os.path.join(OUT, R + "_kernel_stats_config2_rvb.csv"))
for f in glob.glob(os.path.join(RAW, R + "_*.json")) + glob.glob(os.path.join(RAW, R + "_*.txt")):
    shutil.copy(f, OUT)
# RVB kernels: per-kernel counter means of the --pmc pass over tools/rvb_phases.py
pr = sorted(glob.glob(os.path.join(RAW, "pmc_rvb", "*", "*_counter_collection.csv")), key=os.path.getmtime)
if pr:
    import collections
    acc, dur = collections.defaultdict(list), collections.defaultdict(list)
    for row in csv.DictReader(open(pr[-1])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rvb_" not in k:
            continue
        acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
        dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    with open(os.path.join(OUT, R + "_pmc_rvb_kernels.txt"), "w") as f:
        f.write("# rocprofv3 --pmc ... -- python3 tools/rvb_phases.py (configs[2], 1024 replicas): means per launch of the two RVB kernels\n")
        for k in sorted(dur):
            f.write(f"{k}: {sum(dur[k]) / len(dur[k]) / 1e3:.1f} us per launch (n={len(dur[k])})\n")
        for (k, c), v in sorted(acc.items()):
            f.write(f"  {k:40s} {c:22s} {sum(v) / len(v):16.0f}\n")
# registers / spills / scratch of the kernels in the SHIPPED library (read from its code objects, not from a fresh compile)
import subprocess
with open(os.path.join(OUT, R + "_kernel_resources.txt"), "w") as f:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], stdout=f)
HAVE_CLUSTER_KERNEL = "cluster_kernel" in open(stats[-1]).read()
# the timed kernels: PHASE = 0 symbols (PHASE = 1 is the identical code under its data-preparation name)
def kind(k):
    k = k.split("(")[0]
    if "cluster_kernel" in k:  # sse::cluster_kernel<K, HAS_LONG, PHASE>
        return "offdiagonal" if k.rstrip().rstrip(">").rstrip().endswith(", 0") else None
    if HAVE_CLUSTER_KERNEL and re.match(r"(void )?sse::sweep_kernel<4, \d, 1, 0, 2>", k.strip()):
        return "offdiagonal_flagged"  # the general kernel behind sse::cluster_kernel (flagged replicas only): its own row, ADDED below
    if "sweep_fast_kernel" in k:
        return "diagonal" if ", 0, " in k else None
    if "sweep_kernel" in k and k.rstrip().rstrip(">").rstrip().endswith(", 0, 2"):
        return "offdiagonal"
    if "sweep_kernel" in k and k.rstrip().rstrip(">").rstrip().endswith(", 0, 1"):
        return "diagonal"
    return None
def keep(src_dir, out):
    src = [f for f in sorted(glob.glob(os.path.join(RAW, src_dir, "*", "*_counter_collection.csv")), key=os.path.getmtime) if "sweep_fast_kernel" in open(f).read()][-1]
    rows = list(csv.reader(open(src))); hdr = rows[0]; ki = hdr.index("Kernel_Name")
    sel = [r for r in rows[1:] if kind(r[ki])]
    csv.writer(open(os.path.join(OUT, out), "w")).writerows([hdr] + sel)
    return sel, hdr
f_rows, hdr = keep("pmc_fetch", R + "_pmc_fetch_size.csv")
w_rows, _ = keep("pmc_write", R + "_pmc_write_size.csv")
keep("pmc_sq1", R + "_pmc_sq_per_pass.csv"); keep("pmc_sq2", R + "_pmc_sq_lds_per_pass.csv")
ki, vi = hdr.index("Kernel_Name"), hdr.index("Counter_Value")
def mean_kb(rows, which):
    v = [float(r[vi]) for r in rows if kind(r[ki]) == which]
    return sum(v) / len(v)
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --no-cpu-baseline --steps 10 --warmup 2`; "
                     "FETCH_SIZE doubled (gfx950 tallies the 128-B requests of coalesced streaming reads at 64 B: MI355X_MICROARCH.md, HBM section; "
                     "self-calibrated on the diagonal kernel, whose reads are known to be 4 B/slot); bytes per launch",
           "workload": "configs[1] 32x32 beta=16 R=1024",
           "algorithmic_bytes_per_slot": {"diagonal": 8, "offdiagonal": 12}}
def mean_kb0(rows, which):
    v = [float(r[vi]) for r in rows if kind(r[ki]) == which]
    return sum(v) / len(v) if v else 0.0
for which in ("diagonal", "offdiagonal"):
    extra_f = mean_kb0(f_rows, which + "_flagged"); extra_w = mean_kb0(w_rows, which + "_flagged")  # (per step one launch of each)
    traffic[which] = int((2 * (mean_kb(f_rows, which) + extra_f) + mean_kb(w_rows, which) + extra_w) * 1024)
    traffic[which + "_fetch_KB_raw"] = mean_kb(f_rows, which) + extra_f; traffic[which + "_write_KB"] = mean_kb(w_rows, which) + extra_w
json.dump(traffic, open(os.path.join(OUT, R + "_traffic.json"), "w"), indent=1)
print(traffic)
