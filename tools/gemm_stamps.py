import sys, torch
sys.path.insert(0, '/root/repo')
from beach_seg_amd import ops
dev = torch.device("cuda:0")
for M, N, K, b in [(100352, 4096, 1024, False), (100352, 4096, 1024, True), (100352, 1024, 4096, False), (98304, 1024, 1024, False)]:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    bias = torch.randn(N, device=dev) if b else None
    print(f"M={M} N={N} K={K} bias/gelu={b}", flush=True)
    for _ in range(2): ops.gemm_nt(a, w, bias)
    torch.cuda.synchronize()
