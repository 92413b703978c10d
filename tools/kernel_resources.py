#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of the gfx950 code objects in kokoro_ruslan_amd/_obj (no GPU needed).

    python tools/kernel_resources.py [substring ...]

Reads the AMDGPU metadata notes of every object (clang-offload-bundler + llvm-readelf): VGPR / AGPR / SGPR counts, static LDS,
scratch and spill counts.  A kernel beside LDS-DMA must show 0 spilled VGPRs (DESIGN: scratch reloads queue behind the DMAs)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "kokoro_ruslan_amd", "_obj")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    with tempfile.TemporaryDirectory() as td:
        out, fat = os.path.join(td, "dev.o"), os.path.join(td, "fat.bin")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return []
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out):
            return []
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
    rows = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
        rows.append((g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                     g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
    return rows


def main():
    pats = sys.argv[1:]
    print(f"{'kernel':90s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s} {'vspill':>6s} {'sspill':>6s}")
    for f in sorted(os.listdir(OBJ)):
        if not f.endswith(".o"):
            continue
        for row in kernels(os.path.join(OBJ, f)):
            name = subprocess.run(["c++filt", row[0]], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "")
            if pats and not any(p in name for p in pats):
                continue
            print(f"{name[:90]:90s} {row[1]:>5s} {row[2]:>5s} {row[3]:>5s} {row[4]:>7s} {row[5]:>7s} {row[6]:>6s} {row[7]:>6s}")


if __name__ == "__main__":
    main()
