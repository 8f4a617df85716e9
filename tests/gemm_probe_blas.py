"""Calibration only (not product code): what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the step's shapes."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from beach_seg_amd import ops
dev = torch.device("cuda:0")
for M, N, K in [(100352, 4096, 1024), (100352, 1024, 4096), (100352, 3072, 1024), (100352, 1024, 1024), (100352, 1024, 3072), (100352, 16384, 4096)]:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    res = []
    for name, fn in (("ours", lambda: ops.gemm_nt(a, w)), ("blas", lambda: torch.matmul(a, w.t()))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
        for _ in range(n): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        res.append(f"{name} {dt*1e3:.3f} ms {2*M*N*K/dt/1e12:.0f} TF")
    print(f"M={M} N={N} K={K}: " + " | ".join(res), flush=True)
