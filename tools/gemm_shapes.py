"""`ops.gemm_nt` timing on shapes given as M,N,K triples (run on the GPU box):  python tools/gemm_shapes.py 98304,1024,4096 ..."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from beach_seg_amd import ops
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    M, N, K = (int(v) for v in spec.split(","))
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    for _ in range(3): ops.gemm_nt(a, w)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
    for _ in range(n): ops.gemm_nt(a, w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K}: {dt*1e3:.3f} ms {2*M*N*K/dt/1e12:.0f} TF", flush=True)
