"""Three large GEMM shapes of the train step through `ops.gemm_nt` (run on the GPU box).  Used for same-box A/B of GEMM
kernel variants:  BSG_GEMM=4 BSG_LIB=tools/diag/lib_v4.so python tools/gemm_quick.py  (lib built with -DBSG_GEMM_V4)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from beach_seg_amd import ops
dev = torch.device("cuda:0")
for M, N, K in [(100352, 1024, 4096), (100352, 4096, 1024), (100352, 16384, 4096)]:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    for _ in range(3): ops.gemm_nt(a, w)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
    for _ in range(n): ops.gemm_nt(a, w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K}: {dt*1e3:.3f} ms {2*M*N*K/dt/1e12:.0f} TF", flush=True)
