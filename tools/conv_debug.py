"""Debug aid: dump the conv-related workspace regions of a tiny bf16 train step (run twice with / without
BSG_CONV_NO_RING=1 and compare):  python tools/conv_debug.py out.pt"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from beach_seg_amd import ops
from beach_seg_amd.seggpt import SegGptNative
from beach_seg_amd.weights import SegGptGeometry, synth_state_dict
from oracle import seggpt_oracle as O
from oracle.gen_inputs import synth_inputs

g = SegGptGeometry.tiny()
sd = synth_state_dict(g, seed=1)
B = 2
pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, B, 3)
pm = O.normalize(O.apply_mask_rgb(pal, pm_cls)); lab = O.normalize(O.apply_mask_rgb(pal, lb_cls)); yes = (lb_cls != 0)[:, None]
dev = "cuda:0"
m = SegGptNative(sd, g, device=dev, dtype=torch.bfloat16)
pred = m._run_forward(pix.to(dev), prm.to(dev), pm.to(dev), 0, train=True)
H, W = g.image_size
conv_out = m.workspace_region(B, True, "conv_out").view(torch.bfloat16)[: B * H * W * 64].reshape(B, H, W, 64).clone()
loss, gp = ops.loss_fwd_bwd(pred, lab.to(dev), yes.to(dev), 0.01, "reference", True)
first_row = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gpix = m._run_backward(gp, B, first_row=first_row)
dconv = m.workspace_region(B, True, "feat").view(torch.bfloat16)[: B * H * W * 64].reshape(B, H, W, 64).clone()
dfeat = m.workspace_region(B, True, "feat2").view(torch.bfloat16)[: B * H * W * 64].reshape(B, H, W, 64).clone()
torch.save({"pred": pred.cpu(), "conv_out": conv_out.cpu(), "dconv": dconv.cpu(), "dfeat": dfeat.cpu(), "gpix": gpix.cpu()}, sys.argv[1])
print("saved", sys.argv[1])
