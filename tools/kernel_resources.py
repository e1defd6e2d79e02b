#!/usr/bin/env python3
"""Register / spill / scratch / LDS table of every kernel in the SHIPPED library (isingmontecarlo_amd/libisingmc_hip.so).

Reads the code objects out of the built .so itself — not a fresh compile of the sources — so the table describes the
binary that the tests and bench.py load: .hip_fatbin section -> split on the offload-bundle magic -> unbundle the gfx950
entry -> AMDGPU metadata note (msgpack rendered as YAML by llvm-readelf).  rocprofv3's kernel-trace columns
VGPR_Count / Scratch_Size must agree with these numbers.

usage: python tools/kernel_resources.py [--all] [path/to/lib.so] > profiles/rNN_kernel_resources.txt
       (default: only the kernels a bench / profile run times, i.e. PHASE = 0 symbols and the helper kernels; --all lists every symbol)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """gfx950 ELF images embedded in lib (one per translation unit)."""
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fatbin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(td, "stripped")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            piece = os.path.join(td, f"bundle{i}")
            with open(piece, "wb") as f:
                f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(td, f"co{i}.elf")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
                out.append(open(co, "rb").read())
    return out


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(elf_bytes)
        f.flush()
        txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], text=True)
    ks = []
    cur = None
    for line in txt.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:  # first key of a kernel record (keys are sorted)
            cur = {"agpr": int(m.group(1))}
            ks.append(cur)
            continue
        m = re.match(r"\s+\.(\w+):\s+(.*)$", line)
        if m and cur is not None:
            k, v = m.group(1), m.group(2).strip()
            if k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                     "group_segment_fixed_size", "max_flat_workgroup_size", "uses_dynamic_stack"):
                cur[k] = v.strip("'")
    return [k for k in ks if "name" in k]


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def main():
    args = [a for a in sys.argv[1:] if a != "--all"]
    show_all = "--all" in sys.argv[1:]
    lib = args[0] if args else os.path.join(ROOT, "isingmontecarlo_amd", "libisingmc_hip.so")
    rows = []
    for elf in code_objects(lib):
        rows += kernels_of(elf)
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["pretty"] = re.sub(r"\(sse::DevBatch.*$|\(.*$", "", n).replace("void ", "")
    rows.sort(key=lambda r: r["pretty"])
    st = os.stat(lib)
    import hashlib
    import time
    print(f"# {os.path.relpath(lib, ROOT)}: {st.st_size} bytes, built {time.strftime('%Y-%m-%d %H:%M:%S', time.gmtime(st.st_mtime))} UTC, "
          f"sha256 {hashlib.sha256(open(lib, 'rb').read()).hexdigest()[:16]}")
    print("# from the gfx950 code objects inside the library (tools/kernel_resources.py): AMDGPU metadata notes")
    print(f"# {'kernel':<78} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'vspill':>6} {'sspill':>6} {'scratch_B':>9} {'static_LDS_B':>12} {'threads':>7}")
    for r in rows:
        p = r["pretty"]
        if not show_all and is_prep_symbol(p):
            continue
        print(f"  {p:<78} {r.get('vgpr_count', '?'):>5} {r['agpr']:>5} {r.get('sgpr_count', '?'):>5} {r.get('vgpr_spill_count', '?'):>6} "
              f"{r.get('sgpr_spill_count', '?'):>6} {r.get('private_segment_fixed_size', '?'):>9} {r.get('group_segment_fixed_size', '?'):>12} "
              f"{r.get('max_flat_workgroup_size', '?'):>7}")


def is_prep_symbol(pretty):
    """PHASE = 1 symbols hold the identical code under the data-preparation name (template argument PHASE)."""
    m = re.match(r"sse::sweep_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)>", pretty)
    if m:
        return m.group(4) == "1"
    m = re.match(r"sse::sweep_fast_kernel<(\d+), (\d+),", pretty)
    if m:
        return m.group(2) == "1"
    m = re.match(r"sse::cluster_kernel<(\d+), (\w+), (\d+)>", pretty)
    if m:
        return m.group(3) == "1"
    return False


if __name__ == "__main__":
    main()
