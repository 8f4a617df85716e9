"""f32 exact vs f32 x3 (three f16 MFMAs on 22-bit operand splits) train step at ViT-L, B = 16: python tools/x3_probe.py"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from beach_seg_amd.engine import PromptTrainEngine
from beach_seg_amd.seggpt import SegGptNative
from beach_seg_amd.weights import SegGptGeometry, synth_state_dict
g = SegGptGeometry.vit_large(); dev = torch.device("cuda:0")
for x3 in ((True,) if len(sys.argv) > 2 else (False, True)):
    m = SegGptNative(synth_state_dict(g, seed=0, device=dev), g, device=dev, dtype=torch.float32, gemm_x3=x3)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    gen = torch.Generator(device=dev).manual_seed(7)
    rn = lambda: torch.randn(B, 3, 448, 448, device=dev, generator=gen)
    pix, lab, pmc = rn(), rn(), rn()
    yes = torch.ones(B, 1, 448, 448, dtype=torch.bool, device=dev)
    e = PromptTrainEngine(m, torch.rand(64, 3, 448, 448, device=dev, generator=gen), lr=1e-3)
    idx = torch.arange(B, device=dev)
    e.step(pix, lab, yes, idx, pmc); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): l = e.step(pix, lab, yes, idx, pmc)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    m.profile(True); e.step(pix, lab, yes, idx, pmc); torch.cuda.synchronize(); p = m.profile_read(); m.profile(False)
    print("x3" if x3 else "exact", f"{dt*1e3:.0f} ms/step = {B/dt:.1f} tiles/s loss {float(l):.6f}", {k: round(v[0], 1) for k, v in p.items()},
          {k: round(v[1] / (v[0] * 1e-3) / 1e12) for k, v in p.items() if v[0] > 0}, flush=True)
    del e, m; torch.cuda.empty_cache()
