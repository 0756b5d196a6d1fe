#!/usr/bin/env python3
"""Build-time guard for casmtr_amd/csrc/fine_vreg.hip: its value rows are loaded by inline-asm global_load_dword and become valid
only at the hand-written `s_waitcnt vmcnt(8)`.  The compiler does not know that, so nothing may read or copy those 32 registers
between the loads and the wait, and the kernel must not use scratch.  Compiles the file to gfx950 assembly and checks both
template instances.  Exit status 0 = ok."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(asm_text):
    problems = []
    names = re.findall(r"^(_Z22fine_level_vreg_kernelILb[01]EEv9FineVArgs):", asm_text, re.M)
    if len(names) != 2:
        return [f"expected 2 kernel instances, found {names}"]
    for name in names:
        body = asm_text.split(name + ":", 1)[1].split(".Lfunc_end", 1)[0]
        lines = body.split("\n")
        if any("scratch_" in l for l in lines):
            problems.append(f"{name}: scratch instructions present (spills)")
        loads = [i for i, l in enumerate(lines) if re.match(r"\s*global_load_dword v\d+, v\d+, s\[", l) and "ASMSTART" in lines[i - 1]]
        if len(loads) != 32:
            problems.append(f"{name}: expected 32 inline-asm value loads, found {len(loads)}")
            continue
        regs = {re.match(r"\s*global_load_dword (v\d+),", lines[i]).group(1) for i in loads}
        if len(regs) != 32:
            problems.append(f"{name}: value loads share destination registers")
        waits = [i for i, l in enumerate(lines) if "s_waitcnt vmcnt(8)" in l and i > loads[-1]]
        if len(waits) != 1:
            problems.append(f"{name}: expected exactly one vmcnt(8) wait after the loads, found {len(waits)}")
            continue
        for i in range(loads[-1] + 1, waits[0]):
            l = lines[i].split(";")[0]
            toks = set(re.findall(r"\bv\d+\b", l))
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", l):
                toks |= {f"v{k}" for k in range(int(a), int(b) + 1)}
            if toks & regs:
                problems.append(f"{name}: '{lines[i].strip()}' touches a value register before the wait")
    return problems


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "fv.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
               os.path.join(ROOT, "casmtr_amd", "csrc", "fine_vreg.hip"), "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, cwd=d)
        problems = check(open(out).read())
    for p in problems:
        print("fine_vreg ISA check:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
