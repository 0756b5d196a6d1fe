#!/usr/bin/env python3
"""Build-time guard for the LDS-DMA kernels with hand-counted waits (casmtr_amd/csrc/fine_quad.hip, cascade_quad.hip, coarse_tile.hip).  Their DMA chunks are
issued from inline asm that (a) writes M0 without saving it and (b) is waited for with hand-counted `s_waitcnt vmcnt(N)`.  Both are
only sound if the compiler
  * never touches M0 itself in these kernels (every M0 reference must sit inside an ;;#ASMSTART / ;;#ASMEND block), and
  * never spills (a scratch load / store between a DMA issue and its wait changes the vector-memory count the waits rely on).
fine_quad.hip additionally promises that its loop contains NO compiler-visible vector load and no compiler-generated `vmcnt` wait
(its front end stages the next item by DMA; an ordinary load would make the compiler drain the DMA ring with its own vmcnt(0)).
coarse_tile.hip: the instances used by the shipped configs (EMAX <= 11) must not spill; its query loads are waited for in the prologue.
linear_pc.hip (no DMA; the compiler counts its waits): its multiply waves prefetch weight fragments three k-stages ahead, which only works
while the compiler keeps COUNTING -- a conditional load or a conditionally defined register in or around the k-loop makes it fall back to
`s_waitcnt vmcnt(0)` right behind every load (seen twice while the kernel was written).  Checked: no spills, and between the first and
the last MFMA of every instance no vmcnt wait below 4.
Compiles the files to gfx950 assembly and checks every template instance.  Exit status 0 = ok."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = {"fine_quad.hip": (("fine_quad_kernel", 8),), "cascade_quad.hip": (("cascade_quad_kernel", 2),),
         "coarse_tile.hip": (("coarse_tile_kernel", 10),), "window_pair.hip": (("window_match_pair_kernel", 2),)}
NO_COMPILER_VMEM = {"fine_quad_kernel"}          # no vector load / vmcnt wait outside the inline-asm blocks
SPILL_EXEMPT = re.compile(r"coarse_tile_kernelILi16E")   # S > 704 keys: 144 VGPRs at 3 waves per SIMD, spill-free today but not promised


def check_claims(name, lines):
    """Dynamic item claiming (common.hpp work_claim_issue / work_claimed): the atomic's return value arrives asynchronously in a VGPR
    that the compiler believes valid at once.  Between the inline-asm `global_atomic_add vX, ... sc0` and the inline-asm
    `v_readfirstlane_b32 sY, vX` that consumes it, NO instruction may mention vX (a copy would read it before the data has arrived) and
    control may not leave the straight line (a label or a backward branch would mean the register is live around the loop).  The
    first `s_waitcnt vmcnt` in between is the hand-counted wait that covers the claim: it must be there."""
    problems = []
    in_asm = False
    i = 0
    while i < len(lines):
        l = lines[i]
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        m = re.match(r"\s*global_atomic_add (v\d+), v\d+, v\d+, s\[\d+:\d+\] sc0\s*$", l.split(";")[0]) if in_asm else None
        if m:
            reg = m.group(1)
            waited = False
            j = i + 1
            asm2 = True
            while j < len(lines):
                c = lines[j].split(";")[0]
                if "#ASMSTART" in lines[j]:
                    asm2 = True
                elif "#ASMEND" in lines[j]:
                    asm2 = False
                if re.match(rf"\s*v_readfirstlane_b32 s\d+, {reg}\s*$", c) and asm2:
                    break
                if re.search(r"s_waitcnt.*vmcnt", c):
                    waited = True
                if re.search(rf"\b{reg}\b", c) or re.search(rf"\bv\[(\d+):(\d+)\]", c) and any(
                        int(a) <= int(reg[1:]) <= int(b) for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", c)):
                    problems.append(f"{name}: {reg} (a claim in flight) is touched before its v_readfirstlane: {c.strip()}")
                    break
                if re.match(r"\s*s_endpgm", c):
                    problems.append(f"{name}: claim in {reg} is never collected")
                    break
                j += 1
            else:
                problems.append(f"{name}: claim in {reg} is never collected")
            if j < len(lines) and not waited and not any("touched" in p_ or "never" in p_ for p_ in problems):
                problems.append(f"{name}: no vmcnt wait between the claim in {reg} and its v_readfirstlane")
        i += 1
    return problems


def check(asm_text, kernel, n_expected):
    problems = []
    names = sorted(set(re.findall(rf"^(_Z\d+{kernel}\w+):", asm_text, re.M)))
    if len(names) != n_expected:
        return [f"{kernel}: expected {n_expected} instances, found {names}"]
    for name in names:
        body = asm_text.split(name + ":", 1)[1].split(".Lfunc_end", 1)[0]
        lines = body.split("\n")
        if any(re.match(r"\s*scratch_", l) for l in lines) and not SPILL_EXEMPT.search(name):
            problems.append(f"{name}: scratch instructions present (register spills)")
        in_asm = False
        ndma = 0
        exit_atomic = -100   # line of the last compiler-generated exit-counter atomic (common.hpp work_leave: outside the item loop, at a wave's exit)
        for li, l in enumerate(lines):
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            code = l.split(";")[0]
            if re.search(r"\bm0\b", code) and not in_asm:
                problems.append(f"{name}: compiler-generated M0 access: {code.strip()}")
            if "global_load_lds_dword" in code:
                ndma += 1
                if not in_asm:
                    problems.append(f"{name}: LDS-DMA outside inline asm")
            elif not in_asm and re.match(r"\s*global_atomic_add v\d+, v\d+, v\d+, s\[\d+:\d+\] offset:128 sc0", code):
                exit_atomic = li
            elif kernel in NO_COMPILER_VMEM and not in_asm and "vmcnt" in code and li - exit_atomic <= 8:
                pass   # the wait for the exit counter's return value
            elif kernel in NO_COMPILER_VMEM and not in_asm and (re.match(r"\s*(global|buffer|flat)_load", code) or "vmcnt" in code):
                problems.append(f"{name}: compiler-visible vector load / vmcnt wait in a hand-counted DMA kernel: {code.strip()}")
        if ndma == 0:
            problems.append(f"{name}: no LDS-DMA instructions found")
        problems += check_claims(name, lines)
        meta = asm_text.split(f".name:           {name}", 1)
        if len(meta) == 2:
            m = re.search(r"\.vgpr_spill_count:\s*(\d+)", meta[1])
            p = re.search(r"\.private_segment_fixed_size:\s*(\d+)", meta[1])
            if m and int(m.group(1)) != 0:
                problems.append(f"{name}: vgpr_spill_count {m.group(1)}")
            if p and int(p.group(1)) != 0:
                problems.append(f"{name}: private segment {p.group(1)} bytes")
    return problems


def check_prefetch(asm_text, kernel, n_expected, min_in_flight):
    problems = []
    names = sorted(set(re.findall(rf"^(_ZN?\w*{kernel}\w+):", asm_text, re.M)))
    if len(names) != n_expected:
        return [f"{kernel}: expected {n_expected} instances, found {len(names)}"]
    for name in names:
        lines = asm_text.split(name + ":", 1)[1].split(".Lfunc_end", 1)[0].split("\n")
        if any(re.match(r"\s*scratch_", l) for l in lines):
            problems.append(f"{name}: scratch instructions present (register spills)")
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        if not mf:
            problems.append(f"{name}: no MFMA found")
            continue
        for l in lines[mf[0]:mf[-1] + 1]:
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l.split(";")[0])
            if m and int(m.group(1)) < min_in_flight:
                problems.append(f"{name}: {l.strip()} inside the k-loop (the weight prefetch is being drained)")
    return problems


def main():
    bad = []
    with tempfile.TemporaryDirectory() as td:
        src, out = os.path.join(ROOT, "casmtr_amd", "csrc", "linear_pc.hip"), os.path.join(td, "linear_pc.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                               "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
        bad += check_prefetch(open(out).read(), "linear16p_kernel", 6, 4)
        for f, kernels in FILES.items():
            src = os.path.join(ROOT, "casmtr_amd", "csrc", f)
            out = os.path.join(td, f + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                                   "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
            for kernel, n in kernels:
                bad += check(open(out).read(), kernel, n)
    for b in bad:
        print("FAIL:", b)
    print("ok" if not bad else f"{len(bad)} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
