"""`ops.gemm_nt` against a float64 matmul on shapes that are multiples of 256 (the ones the experimental v4 kernel takes):
BSG_GEMM=4 BSG_LIB=tools/diag/lib_v4.so python tools/gemm_check.py  (run on the GPU box)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from beach_seg_amd import ops
dev = torch.device("cuda:0")
for M, N, K in [(256, 256, 64), (256, 256, 128), (256, 256, 256), (512, 768, 1024), (2048, 1024, 4096)]:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).bfloat16()
    ref = a.double() @ w.double().t()
    out = ops.gemm_nt(a, w).double()
    d = (out - ref).abs()
    bad = d > 6e-3 * ref.abs().max()
    print(f"M={M} N={N} K={K}: rel {float(d.max()/ref.abs().max()):.3e} bad {int(bad.sum())} of {bad.numel()}", flush=True)
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("  bad rows", rows[:20].tolist(), "...", int(rows.numel()), " bad cols", cols[:20].tolist(), "...", int(cols.numel()))
