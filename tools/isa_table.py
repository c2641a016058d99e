#!/usr/bin/env python3
"""ISA resource table of every device kernel in the shipped library: registers, spills, scratch, LDS and the MFMA / LDS-DMA
instruction counts, read from the code objects embedded in posendf_amd/lib/libposendf_amd.so (no GPU needed).
usage: python tools/isa_table.py [library.so] > profiles/rNN/isa_resources.txt"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "posendf_amd", "lib", "libposendf_amd.so")
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=tmp)
        rows = []
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            co = os.path.join(tmp, name)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
            bodies = {m.group(1): m.group(2) for m in re.finditer(r"^[0-9a-f]+ <([\w.$]+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M)}
            for blk in notes.split("- .agpr_count:")[1:]:
                f = lambda key: (re.search(rf"\.{key}:\s+(\S+)", blk) or [None, "?"])[1]
                kname = f("name")
                body = bodies.get(kname, "")
                rows.append((kname, re.match(r"\s*(\d+)", blk).group(1), f("vgpr_count"), f("sgpr_count"), f("vgpr_spill_count"),
                             f("sgpr_spill_count"), f("private_segment_fixed_size"), f("group_segment_fixed_size"),
                             len(re.findall(r"v_mfma_f32_16x16x32[_a-z]*f16", body)), len(re.findall(r"v_mfma_f32_16x16x4[_a-z]*f32", body)),
                             body.count("global_load_lds_dwordx4"), body.count("\n")))
    print(f"# {os.path.relpath(lib, REPO)}: kernel resources from the embedded gfx950 code objects (llvm-readelf --notes, llvm-objdump -d)")
    print(f"{'kernel':46s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratchB':>8s} {'staticLDS':>9s} "
          f"{'mfma16x16x32f16':>15s} {'mfma16x16x4f32':>14s} {'lds_dma':>7s} {'instr':>7s}")
    for r in sorted(rows):
        print(f"{r[0]:46s} {r[2]:>5s} {r[1]:>5s} {r[3]:>5s} {r[4]:>6s} {r[5]:>6s} {r[6]:>8s} {r[7]:>9s} {r[8]:15d} {r[9]:14d} {r[10]:7d} {r[11]:7d}")
    print("# VGPR = total per lane (architectural + accumulation file, 512 available at one wave per SIMD); the fused kernels take "
          "their 160 KiB of LDS dynamically (staticLDS 0).")


if __name__ == "__main__":
    main()
