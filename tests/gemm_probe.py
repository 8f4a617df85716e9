"""GEMM micro-benchmark (GPU box): python tests/gemm_probe.py [M N K] ..."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from beach_seg_amd import ops
dev = torch.device("cuda:0")
shapes = [(100352, 4096, 1024), (100352, 1024, 4096), (100352, 3072, 1024), (100352, 1024, 1024), (100352, 1024, 2048), (100352, 1024, 3072), (100352, 4096, 2048), (65536, 1024, 1024), (65536, 4096, 1024)]
if len(sys.argv) > 3:
    shapes = [tuple(int(x) for x in sys.argv[1:4])]
for M, N, K in shapes:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    out = ops.gemm_nt(a, w)
    ref = (a[:256].float() @ w.float().t())
    err = ((out[:256].float() - ref).abs().max() / ref.abs().max()).item()
    for _ in range(3): ops.gemm_nt(a, w)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
    for _ in range(n): ops.gemm_nt(a, w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K}: {dt*1e3:.3f} ms  {2*M*N*K/dt/1e12:.1f} TFLOP/s  relerr {err:.2e}", flush=True)
