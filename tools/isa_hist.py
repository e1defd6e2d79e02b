"""Opcode histogram of an instruction-index range of one kernel in a hipcc -S dump (indices as printed by isa_loops.py).
usage: python tools/isa_hist.py file.s kernel_substring first last"""
import re, sys, collections
path, key, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and re.match(r"^\S+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
ins = []
for l in lines[start:end + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    ins.append(t.split(";")[0].strip())
c = collections.Counter(i.split()[0] for i in ins[a:b + 1])
print(" ".join(f"{v}x{k}" for k, v in c.most_common()))
