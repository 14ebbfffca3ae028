#!/usr/bin/env python3
"""Register / LDS / scratch figures of the kernels in libasq_hip.so (or any .so / .o with gfx950 code objects) from the ELF notes.
    python tools/kernel_regs.py [pattern] [path]"""
import os, re, subprocess, sys, tempfile, shutil
LLVM = "/opt/rocm/lib/llvm/bin"
pat = sys.argv[1] if len(sys.argv) > 1 else ""
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "autosmoothquant_amd", "libasq_hip.so")
td = tempfile.mkdtemp()
try:
    so = os.path.join(td, "lib.so")
    shutil.copy(path, so)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=td)
    for o in sorted(p for p in os.listdir(td) if "gfx950" in p):
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(td, o)], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            cur = {"agpr_count": blk.split("\n", 1)[0].strip()}
            for line in blk.splitlines():
                m = re.match(r"\s*\.(\w+):\s+(\S+)", line)
                if m and m.group(1) not in cur:
                    cur[m.group(1)] = m.group(2)
            if cur.get("name") and cur.get("vgpr_count"):
                dem = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip() or cur["name"]
                if pat in dem or pat in cur["name"]:
                    print(f"vgpr {cur.get('vgpr_count'):>4} agpr {cur.get('agpr_count', '0'):>4} sgpr {cur.get('sgpr_count'):>4} scratch {cur.get('private_segment_fixed_size'):>4} spill {cur.get('vgpr_spill_count'):>3}  {dem[:170]}")
finally:
    shutil.rmtree(td, ignore_errors=True)
