"""List the loops (backward branches) of one kernel in a hipcc -S dump with instruction-class counts.
usage: python tools/isa_loops.py file.s kernel_substring"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and re.match(r"^\S+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))  # (kernels with early returns hold several s_endpgm)
body = lines[start:end + 1]
labels, ins = {}, []
for l in body:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t): labels[t.split(":")[0]] = len(ins)
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m: labels[m.group(1)] = len(ins); continue
    if t.endswith(":"): continue
    ins.append(t.split(";")[0].strip())
print("kernel instrs:", len(ins))
def cls(i):
    op = i.split()[0]
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    return "salu"
loops = []
for idx, i in enumerate(ins):
    m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", i)
    if m and m.group(1) in labels and labels[m.group(1)] <= idx:
        loops.append((labels[m.group(1)], idx, m.group(1)))
loops.sort()
for a, b, lab in loops:
    seg = ins[a:b + 1]
    c = {}
    for i in seg: c[cls(i)] = c.get(cls(i), 0) + 1
    tags = []
    txt = " ".join(seg)
    if "0xd2511f53" in txt.lower(): tags.append("philox")
    if "v_mul_f64" in txt: tags.append("f64")
    if "ds_cmpst" in txt or "ds_cmpswap" in txt: tags.append("cas")
    if "s_barrier" in txt: tags.append("bar")
    print(f"{lab:12s} [{a:6d},{b:6d}] n={b-a+1:5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + "  " + ",".join(tags))
