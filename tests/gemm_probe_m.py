"""GEMM M-sweep (GPU box): python tests/gemm_probe_m.py N K M1 M2 ..."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from beach_seg_amd import ops
dev = torch.device("cuda:0")
N, K = int(sys.argv[1]), int(sys.argv[2])
for M in [int(x) for x in sys.argv[3:]]:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    for _ in range(3): ops.gemm_nt(a, w)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
    for _ in range(n): ops.gemm_nt(a, w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K}: {dt*1e3:.3f} ms  {2*M*N*K/dt/1e12:.1f} TFLOP/s  rounds {((M+255)//256)*((N+255)//256)/256:.2f}", flush=True)
