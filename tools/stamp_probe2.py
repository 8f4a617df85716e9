import sys, torch
sys.path.insert(0, '/root/repo')
from beach_seg_amd import ops
dev = torch.device("cuda:0")
for M, N, K in [(100352, 4096, 1024), (100352, 1024, 4096)]:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    for _ in range(6): ops.gemm_nt(a, w)
    print(f"plain M={M} N={N} K={K}", file=sys.stderr, flush=True)
    ops.gemm_nt(a, w)
