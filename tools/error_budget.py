"""Per-stage error budget of the 16-bit dtype against the f32 parity mode (run on the GPU box):

    python tools/error_budget.py [geometry=vit_large] [B=1] [out.json]

Both models run the HIP path on the same weights and inputs; the f32 mode agrees with the reference vectors to ~1e-6
(`tests/test_gpu_parity.py`), so it stands in for the reference here (the CPU oracle needs ~12 s per ViT-L tile).
For every saved stage it prints  max|a - b| / max|b|  (the metric of the parity tests) and the rms-relative error,
for each variant of the 16-bit path (embed_split on / off), and counts decoded-mask mismatches with their palette margin.
"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from beach_seg_amd import ops  # noqa: E402
from beach_seg_amd.seggpt import SegGptNative  # noqa: E402
from beach_seg_amd.weights import SegGptGeometry, synth_state_dict  # noqa: E402
from oracle import seggpt_oracle as O  # noqa: E402
from oracle.gen_inputs import synth_inputs  # noqa: E402

DEV = torch.device("cuda:0")


def err(a, b):
    a, b = a.detach().float(), b.detach().float()
    d = (a - b)
    return [float(d.abs().max() / b.abs().max().clamp_min(1e-30)), float(d.norm() / b.norm().clamp_min(1e-30))]


def run(model, g, B, inputs, dt):
    pix, prm, pm, lab, yes, pn = inputs
    N, D, L = g.num_tokens, g.hidden_size, g.num_hidden_layers
    p = prm.clone().requires_grad_(True)
    out = model(pixel_values=pix, prompt_pixel_values=p, prompt_masks=pm, labels=lab)
    torch.cuda.synchronize()
    st = {}

    def region(name, layer, dtype, shape):
        n = 1
        for s in shape:
            n *= s
        return model.workspace_region(B, True, name, layer).view(dtype)[:n].reshape(shape).clone()

    for l in range(L + 1):
        S = 2 * B if l <= g.merge_index else B
        st[f"x_in[{l}]"] = region("x_in", l, torch.float32, (S, N, D))
        if l < L:
            st[f"x_mid[{l}]"] = region("x_mid", l, torch.float32, (S, N, D))
    st["pred"] = out.pred_masks.detach().clone()
    loss = ops.seggpt_loss(out.pred_masks, lab, yes, 0.01, "reference")
    loss.backward()
    torch.cuda.synchronize()
    st["grad_prompt"] = p.grad.clone()
    st["masks"] = ops.decode_argmin(out.pred_masks.detach(), pn)
    st["loss"] = loss.detach()
    return st


def main():
    gname = sys.argv[1] if len(sys.argv) > 1 else "vit_large"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    outp = sys.argv[3] if len(sys.argv) > 3 else None
    g = getattr(SegGptGeometry, gname)()
    sd = synth_state_dict(g, seed=0, device=DEV if gname == "vit_large" else "cpu")
    pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, B, 7)
    pm = O.normalize(O.apply_mask_rgb(pal, pm_cls))
    lab = O.normalize(O.apply_mask_rgb(pal, lb_cls))
    yes = (lb_cls != 0)[:, None]
    pn = O.palette_norm(pal)
    inputs = tuple(t.to(DEV) for t in (pix, prm, pm, lab, yes, pn))
    ref = run(SegGptNative(sd, g, device=DEV, dtype=torch.float32), g, B, inputs, torch.float32)
    report = {"geometry": gname, "batch": B, "metric": "[max|a-b|/max|b|, |a-b|_2/|b|_2] vs the f32 parity mode"}
    H = g.image_size[0] // 2
    x = ref["pred"][:, :, H:, :].permute(0, 2, 3, 1)
    d = ((x[:, :, :, None, :] - inputs[5][:, None, None, :, :]) ** 2).sum(-1)
    top2 = d.topk(2, dim=-1, largest=False).values
    margin = (top2[..., 1] - top2[..., 0])
    variants = [("f16_split", torch.float16, True), ("bf16_split", torch.bfloat16, True), ("bf16_nosplit", torch.bfloat16, False)]
    for name, dt, split in variants:
        model = SegGptNative(sd, g, device=DEV, dtype=dt, embed_split=split)
        got = run(model, g, B, inputs, dt)
        rep = {}
        for k in ref:
            if k in ("masks", "loss"):
                continue
            if k.startswith("x_") and not (k.endswith("[0]") or int(k[k.index("[") + 1:-1]) % 4 == 3 or k.startswith("x_in")):
                continue
            rep[k] = err(got[k], ref[k])
        bad = got["masks"] != ref["masks"]
        rep["loss_rel"] = float((got["loss"] - ref["loss"]).abs() / ref["loss"].abs())
        rep["mask_mismatches"] = int(bad.sum())
        rep["mask_pixels"] = int(bad.numel())
        rep["max_margin_of_a_mismatch"] = float(margin[bad].max()) if bad.any() else 0.0
        rep["pred_abs_err_max"] = float((got["pred"] - ref["pred"]).abs().max())
        report[name] = rep
        print(name, json.dumps({k: v for k, v in rep.items() if not k.startswith("x_")}), flush=True)
        for k, v in rep.items():
            if k.startswith("x_in"):
                print(f"  {k}: {v[0]:.3e} {v[1]:.3e}")
        del model
    if outp:
        Path(outp).write_text(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
