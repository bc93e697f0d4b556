"""Registers, spills and LDS of the kernels in libmg_hip.so whose (demangled) name contains the given text, read from the
code object's notes.

    python tools/kernel_resources.py [text=jacobikc]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multigrid_dolfinx_amd", "libmg_hip.so")
want = sys.argv[1] if len(sys.argv) > 1 else "jacobikc"
with tempfile.TemporaryDirectory() as tmp:
    local = os.path.join(tmp, "lib.so")             # (the bundles are written next to the file)
    shutil.copy(so, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, check=True, capture_output=True)
    co = next(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
rows = []
for block in notes.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    val = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", block).group(1))
    rows.append((name, val("vgpr_count"), val("vgpr_spill_count"), val("sgpr_count"), val("private_segment_fixed_size"), val("group_segment_fixed_size")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
for (name, vgpr, spill, sgpr, scratch, lds), dn in zip(rows, names):
    if want in dn:
        print(f"{dn[:100]:100s} vgpr {vgpr:3d} spilled {spill:3d} sgpr {sgpr:3d} scratch {scratch:4d} B static lds {lds} B")
