"""Instruction mix of every loop (backward branch) of a disassembly written by tools/kernel_resources.py --dump:
    python tools/loop_mix.py /tmp/k.s"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
ins = []
for ln in lines:
    m = re.match(r"\s+(.*?)\s*// ([0-9A-F]+):", ln)
    if m:
        ins.append((int(m.group(2), 16), m.group(1)))
print(len(ins), 'instructions')
for i, (a, t) in enumerate(ins):
    m = re.match(r"s_c?branch\S*\s+(\d+)", t)
    if not m:
        continue
    off = int(m.group(1))
    if off > 32767:
        tgt = a + 4 + (off - 65536) * 4
        j0 = next(j for j, (a2, _) in enumerate(ins) if a2 >= tgt)
        body = [x for _, x in ins[j0:i + 1]]
        c = Counter(x.split()[0] for x in body)
        n = lambda f: sum(v for k, v in c.items() if f(k))
        print(f"loop [{j0}, {i}] {len(body)} instr: mfma {n(lambda k: k.startswith('v_mfma'))} accvgpr {n(lambda k: 'accvgpr' in k)} "
              f"valu {n(lambda k: k.startswith('v_') and not k.startswith('v_mfma'))} ds {n(lambda k: k.startswith('ds_'))} "
              f"vmem {n(lambda k: k.startswith(('global_', 'buffer_', 'scratch_')))} salu {n(lambda k: k.startswith('s_') and k not in ('s_waitcnt', 's_nop', 's_barrier'))} "
              f"waitcnt {c['s_waitcnt']} nop {c['s_nop']} barrier {c['s_barrier']}")
        if '-v' in sys.argv:
            print('   ', c.most_common(45))
