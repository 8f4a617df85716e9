"""Compact one-letter-per-instruction trace of a loop of a disassembly (tools/kernel_resources.py --dump):
    python tools/loop_trace.py /tmp/k.s <first> <last>      (indices as printed by tools/loop_mix.py)
m builtin MFMA (VGPR result), M MFMA into AGPRs, e transcendental, c convert, v other vector, a accvgpr move, d LDS read/write,
t transposing LDS read, G LDS-DMA, g other vector memory, W s_waitcnt, n s_nop, B barrier, s scalar."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
ins = []
for ln in lines:
    m = re.match(r"\s+(.*?)\s*// ([0-9A-F]+):", ln)
    if m:
        ins.append(m.group(1))
j0, j1 = int(sys.argv[2]), int(sys.argv[3])
out = []
for t in ins[j0:j1 + 1]:
    op = t.split()[0]
    if op.startswith('v_mfma'): c = 'M' if t.split()[1].startswith('a[') else 'm'
    elif op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')): c = 'e'
    elif op.startswith('v_cvt'): c = 'c'
    elif 'accvgpr' in op: c = 'a'
    elif op.startswith('v_'): c = 'v'
    elif op.startswith('ds_read_b64_tr'): c = 't'
    elif op.startswith('ds_'): c = 'd'
    elif op == 's_waitcnt': c = 'W'
    elif op == 's_nop': c = 'n'
    elif op == 's_barrier': c = 'B'
    elif op.startswith('global_load_lds') or (op.startswith('buffer_load') and ' lds' in t): c = 'G'
    elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): c = 'g'
    elif op.startswith('s_'): c = 's'
    else: c = '?'
    out.append(c)
s = ''.join(out)
for i in range(0, len(s), 120):
    print(s[i:i + 120])
