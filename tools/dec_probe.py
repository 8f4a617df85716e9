"""GPU probe: 16-bit error of the decoder tail (forward and backward stages) against the f32 parity mode, for decoder widths
64 and 128 on the tiny net -- `python tools/dec_probe.py`.  Debugging aid, not part of the product or the tests."""
import dataclasses
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from beach_seg_amd import ops  # noqa: E402
from beach_seg_amd.seggpt import SegGptNative  # noqa: E402
from beach_seg_amd.weights import SegGptGeometry, counter_noise, synth_state_dict  # noqa: E402

DEV = torch.device("cuda:0")


def inputs(g, B, seed):
    H, W = g.image_size[0] // 2, g.image_size[1]
    n = B * 3 * H * W
    mk = lambda s: counter_noise(n, seed * 10 + s).reshape(B, 3, H, W).to(DEV)
    return mk(1), mk(2), mk(3), mk(4)


def err(a, b):
    d = (a.float() - b.float())
    return f"{float(d.abs().max() / b.float().abs().max()):.2e}/{float(d.norm() / b.float().norm()):.2e}"


def run(g, sd, dt, B, seed):
    m = SegGptNative(sd, g, device=DEV, dtype=dt)
    pix, prm, pm, lab = inputs(g, B, seed)
    p = prm.clone().requires_grad_(True)
    out = m(pixel_values=pix, prompt_pixel_values=p, prompt_masks=pm)
    yes = torch.ones(B, 1, *lab.shape[2:], dtype=torch.bool, device=DEV)
    ops.seggpt_loss(out.pred_masks, lab, yes, 0.01, "reference").backward()
    torch.cuda.synchronize()
    ws = m._last_bwd[1]
    tdt = dt
    st = {"pred": out.pred_masks.detach().clone(), "grad": p.grad.clone()}
    import ctypes as C
    from beach_seg_amd import _native as N
    S = 1.0
    for name in ("gscale", "feat", "feat2", "dtaps", "conv_out"):
        off, nb = C.c_size_t(), C.c_size_t()
        N.check(m._lib.bsg_workspace_region(m._h, B, 1, name.encode(), -1, C.byref(off), C.byref(nb)))
        r = ws[off.value: off.value + nb.value]
        if name == "gscale":
            S = float(r.view(torch.float32)[0]) if dt == torch.float16 else 1.0
        else:
            st[name] = r.view(tdt).float().clone() / (1.0 if name == "conv_out" else S)
    st["S"] = S
    return st


def main():
    for dd, ws, iseed in ((64, 1, 3), (128, 4, 8), (64, 4, 8), (128, 1, 3)):
        g = dataclasses.replace(SegGptGeometry.tiny(), decoder_hidden_size=dd)
        sd = synth_state_dict(g, seed=ws)
        ref = run(g, sd, torch.float32, 2, iseed)
        for dt in (torch.float16, torch.bfloat16):
            got = run(g, sd, dt, 2, iseed)
            print(f"dec {dd} seeds ({ws},{iseed}) {str(dt)[6:]:9s} S={got['S']:.3g}: " + "  ".join(
                f"{k} {err(got[k], ref[k])}" for k in ("pred", "conv_out", "feat", "feat2", "dtaps", "grad")), flush=True)


if __name__ == "__main__":
    main()
