#!/usr/bin/env python3
"""Mean counter values per timed kernel (PHASE = 0 symbols) from rocprofv3 --pmc csv output directories (dev tool)."""
import csv, glob, os, sys, collections
for d in sorted(glob.glob(os.path.join(sys.argv[1], "*"))):
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "sse::" not in k or ", 1>" in k or ", 1, " in k.split("<", 1)[1][2:9] and "fast" in k:
                continue
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{os.path.basename(d):6s} {k:52s} {c:24s} {sum(v)/len(v):16.0f}  (n={len(v)})")
