"""Stage-by-stage comparison of the HIP path with the oracle (debugging aid; run on the GPU box):
    python tests/debug_stages.py [tiny|small] [f32|bf16]
Prints the relative error of every saved activation / gradient stage."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from beach_seg_amd import ops  # noqa: E402
from beach_seg_amd.seggpt import SegGptNative  # noqa: E402
from beach_seg_amd.weights import SegGptGeometry, synth_state_dict  # noqa: E402
from oracle import seggpt_oracle as O  # noqa: E402
from oracle.gen_inputs import synth_inputs  # noqa: E402


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def main():
    gname = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    dt = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.bfloat16
    g = getattr(SegGptGeometry, gname)()
    B = 2
    sd = synth_state_dict(g, seed=1)
    pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, B, 3)
    pm = O.normalize(O.apply_mask_rgb(pal, pm_cls))
    lab = O.normalize(O.apply_mask_rgb(pal, lb_cls))
    yes = (lb_cls != 0)[:, None]
    N, D, L = g.num_tokens, g.hidden_size, g.num_hidden_layers

    # oracle with intermediates
    p_ref = prm.clone().requires_grad_(True)
    img = torch.cat((p_ref, pix), 2)
    msk = torch.cat((pm, lab), 2)
    x = O.embeddings(sd, g, img, msk, O.default_bool_masked_pos(g))
    xs = [x]
    taps = []
    for i in range(L):
        x = O.layer(sd, i, g, x)
        if i == g.merge_index:
            x = (x[:B] + x[B:]) * 0.5
        xs.append(x)
        if i in g.intermediate_hidden_state_indices:
            taps.append(F.layer_norm(x, (D,), sd["model.encoder.layernorm.weight"], sd["model.encoder.layernorm.bias"], g.layer_norm_eps))
    pred_ref = O.decoder(sd, g, torch.cat(taps, -1))
    for t in xs + [pred_ref]:
        t.retain_grad()
    loss_ref = O.seggpt_loss(pred_ref, lab, yes, 0.01, "reference")
    loss_ref.backward()

    dev = torch.device("cuda:0")
    model = SegGptNative(sd, g, device=dev, dtype=dt)
    p = prm.to(dev).requires_grad_(True)
    out = model(pixel_values=pix.to(dev), prompt_pixel_values=p, prompt_masks=pm.to(dev), labels=lab.to(dev))
    torch.cuda.synchronize()

    def region(name, layer, dtype, shape):
        return model.workspace_region(B, True, name, layer).view(dtype)[: int(torch.tensor(shape).prod())].reshape(shape)

    for l in range(L + 1):
        S = xs[l].shape[0]
        print(f"x_in[{l}] rel err {rel(region('x_in', l, torch.float32, (S, N, D)), xs[l]):.3e}")
    print(f"pred rel err {rel(out.pred_masks, pred_ref):.3e}")
    loss = ops.seggpt_loss(out.pred_masks, lab.to(dev), yes.to(dev), 0.01, "reference")
    print(f"loss {loss.item():.6f} ref {loss_ref.item():.6f}")
    loss.backward()
    torch.cuda.synchronize()
    print(f"dpred rel err {rel(ops._LossFn.apply, pred_ref.grad) if False else 0:.1f} (see loss test)")
    print(f"grad rel err {rel(p.grad, p_ref.grad):.3e}  |grad| {p_ref.grad.abs().max().item():.3e}")
    masks = ops.decode_argmin(out.pred_masks.detach(), O.palette_norm(pal).to(dev))
    mref = O.decode_argmin(pred_ref.detach(), O.palette_norm(pal))
    print("mask mismatches", (masks.cpu() != mref).sum().item(), "of", mref.numel())


if __name__ == "__main__":
    main()
