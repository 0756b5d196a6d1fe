#!/usr/bin/env python3
"""Static ISA summary of one gfx950 assembly file (hipcc -S --cuda-device-only): registers, spills, instruction mix per kernel,
compiler-generated M0 accesses (the LDS-DMA kernels write M0 from inline asm without saving it).  usage: isa_stats.py file.s [filter]"""
import re
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = {}
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', t, re.S):
    g = lambda k: re.search(rf'\.amdhsa_{k}\s+(\S+)', m.group(2)).group(1)
    meta[m.group(1)] = f"vgpr {g('next_free_vgpr')} sgpr {g('next_free_sgpr')}"
for n in sorted(meta):
    if flt not in n:
        continue
    body = t.split(n + ':', 1)[1].split('.Lfunc_end', 1)[0]
    ina, bad, cnt = False, 0, {}
    for l in body.split('\n'):
        if '#ASMSTART' in l:
            ina = True
        elif '#ASMEND' in l:
            ina = False
        code = l.split(';')[0]
        if re.search(r'\bm0\b', code) and not ina:
            bad += 1
        mm = re.match(r'\s+([a-z_0-9]+)', code)
        if mm:
            op = mm.group(1)
            k = ('mfma' if 'mfma' in op else 'scratch' if op.startswith('scratch') else 'lds' if op.startswith('ds_') else
                 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else
                 'vmem' if op.startswith(('global', 'buffer', 'flat')) else 'other')
            cnt[k] = cnt.get(k, 0) + 1
    print(n, meta[n], cnt, f"compiler M0 accesses: {bad}")
