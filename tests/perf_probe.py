"""Quick timing probe (GPU box): python tests/perf_probe.py B [f32|bf16] [train]"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from beach_seg_amd.seggpt import SegGptNative
from beach_seg_amd.weights import SegGptGeometry, synth_state_dict

B = int(sys.argv[1]); dt = torch.bfloat16 if sys.argv[2] == "bf16" else torch.float32
train = len(sys.argv) > 3
dev = torch.device("cuda:0")
g = SegGptGeometry.vit_large()
t0 = time.time(); sd = synth_state_dict(g, seed=0, device=dev); print("weights", time.time() - t0, flush=True)
m = SegGptNative(sd, g, device=dev, dtype=dt); del sd
x = lambda: torch.randn(B, 3, 448, 448, device=dev)
pix, prm, pm = x(), x().requires_grad_(train), x()
gp = torch.randn(B, 3, 896, 448, device=dev) * 1e-5
def step():
    out = m(pixel_values=pix, prompt_pixel_values=prm, prompt_masks=pm)
    if train: out.pred_masks.backward(gp)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.time(); n = 3
for _ in range(n): step()
torch.cuda.synchronize(); dt_ = (time.time() - t0) / n
fl = (3.2681e12 if train else 1.5897e12) * B
print(f"B={B} {sys.argv[2]} train={train}: {dt_*1e3:.1f} ms/step, {B/dt_:.2f} tiles/s, {fl/dt_/1e12:.1f} TFLOP/s", flush=True)
