"""GEMM time per step and epilogue type from the kernel traces of tools/step_ab.sh:  python tools/ab_report.py gpurun_out/ab_default_5 ...
(epi0 bias, 1 bias+GELU, 2 bias+residual, 3 token table, 4 pixel shuffle, 5 plain, 6 x gelu', 7 pixel unshuffle; epiX = v3 on a
non-plain A operand; tail128 = the 128 x 128 kernel on the thin last round)."""
import csv,re,sys,glob
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        n=r['Name']
        if 'gemm' not in n: continue
        if 'v5' in n: m=re.search(r'v5IDF16bLi(\d)E',n); k='epi'+(m.group(1) if m else '1')
        elif 'v3' in n: m=re.search(r'v3IDF16bLi(\d)ELi(\d)E',n); k='epi'+(m.group(2) if m and m.group(1)=='0' else 'X')
        else: k='tail128'
        d[k]=d.get(k,0)+float(r['TotalDurationNs'])/3e6
    return d
for p in sys.argv[1:]:
    a=load(p+'/kernel_stats.csv')
    print(f"{p:40s}", ' '.join(f"{k}={v:5.2f}" for k,v in sorted(a.items())), 'sum', round(sum(a.values()),2))
