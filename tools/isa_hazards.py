#!/usr/bin/env python3
"""Static checks of the shipped gfx950 code objects for the two things hipcc cannot see inside inline asm:

1. wait states.  A VGPR written by a VALU instruction needs 2 wait states before an MFMA reads it as its A / B operand
   (cdna_hip_programming.md 5.7 item 2); hipcc pads its own instructions, never the inside of an asm statement.  The operand
   splits (`v_fma_mixlo_f16` / `v_fma_mixhi_f16`: csrc/pndf_kernel_split.hip split2, csrc/pndf_lbs.hip lbs_split2) and the asm
   `v_max_f32` of the activations write registers that may be MFMA operands: every such write must be followed by at least
   two wait states (any two instructions, or `s_nop 1`) before an MFMA that reads the register.
2. M0.  The LDS-DMA (`global_load_lds_dwordx4`) takes its LDS destination from M0, which the ring sets once per group of
   pieces (`s_mov_b32 m0, sN` right in front of the first piece); the following pieces rely on nothing else writing M0.
   Every instruction that writes M0 must therefore be one of those `s_mov_b32 m0` (followed within two instructions by a
   `global_load_lds`), and no `s_set_gpr_idx_*` / `s_movrel*` / `s_sendmsg` may appear in a kernel that issues LDS-DMA.

usage: python tools/isa_hazards.py [library.so]      exit status 1 and one line per finding when a check fails
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
WRITERS = ("v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_max_f32")
NEED = 2


def _regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def kernels(lib):
    """{kernel name: [(mnemonic, operand text), ...]} of every code object embedded in `lib`"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=tmp)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(tmp, name)], capture_output=True, text=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <([\w.$]+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M):
                ins = []
                for line in m.group(2).splitlines():
                    mm = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
                    if mm:
                        ins.append((mm.group(1), mm.group(2)))
                out[m.group(1)] = ins
    return out


def check_wait_states(ins):
    """findings [(index, writer, index of the MFMA, wait states)] and the smallest distance seen"""
    bad, closest, sites = [], None, 0
    for i, (op, args) in enumerate(ins):
        if not op.startswith(WRITERS):
            continue
        dst = _regs(args.split(",")[0])
        if not dst:
            continue
        sites += 1
        ws, j = 0, i + 1
        while j < len(ins) and ws < 8:
            o, a = ins[j]
            if o.startswith(("v_mfma", "v_smfma")):
                parts = re.split(r",\s*(?![^\[]*\])", a)
                srcs = set()
                for p in parts[1:3]:      # A and B (C = the accumulator takes part in the MFMA's own dependency check)
                    srcs |= _regs(p)
                if dst & srcs:
                    closest = ws if closest is None else min(closest, ws)
                    if ws < NEED:
                        bad.append((i, f"{op} {args}", j, ws))
                    break
                ws += 1
            elif o == "s_nop":
                ws += int(a.split()[0], 0) + 1
            elif o.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                break                       # control flow: what follows is another path (the padded compiler code)
            else:
                if dst & _regs(a.split(",")[0]) and o.startswith("v_") and not o.startswith(WRITERS[:2]):
                    break                   # the register was overwritten by another producer
                ws += 1
            j += 1
    return bad, closest, sites


def check_m0(ins):
    bad = []
    dma = [i for i, (o, _) in enumerate(ins) if o.startswith("global_load_lds")]
    if not dma:
        return bad, 0
    for i, (op, args) in enumerate(ins):
        first = args.split(",")[0].strip()
        if op.startswith(("s_set_gpr_idx", "s_movrel", "v_movrel", "s_sendmsg")):
            bad.append((i, f"{op} {args}: uses or changes M0 in a kernel that issues LDS-DMA"))
        elif first == "m0" and not op.startswith(("s_cmp", "s_bitcmp")):
            ours = op == "s_mov_b32" and any(o.startswith("global_load_lds") for o, _ in ins[i + 1:i + 3])
            if not ours:
                bad.append((i, f"{op} {args}: an M0 write that is not the ring's"))
    return bad, len(dma)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "posendf_amd", "lib", "libposendf_amd.so")
    ks = kernels(lib)
    failed = False
    print(f"# {os.path.relpath(lib, REPO)}: VALU-in-asm -> MFMA operand wait states (>= {NEED}) and M0 discipline of the LDS-DMA kernels")
    print(f"{'kernel':46s} {'asm-writer sites':>16s} {'closest MFMA read':>18s} {'lds_dma':>8s}  findings")
    for name in sorted(ks):
        ins = ks[name]
        bad_w, closest, sites = check_wait_states(ins)
        bad_m, ndma = check_m0(ins)
        if not sites and not ndma:
            continue
        print(f"{name:46s} {sites:16d} {('-' if closest is None else str(closest) + ' states'):>18s} {ndma:8d}  {len(bad_w) + len(bad_m)}")
        for i, what, j, ws in bad_w:
            failed = True
            print(f"  HAZARD {name}: instruction {i} `{what}` is read by the MFMA at {j} after {ws} wait state(s)")
        for i, what in bad_m:
            failed = True
            print(f"  M0 {name}: instruction {i} {what}")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
