#!/usr/bin/env python3
"""Where does the 16-bit error of the HIP path come from, and what would each part cost to remove?  

A precision MODEL, not a measurement of the kernels: the network arithmetic in plain fp32 torch on the CPU (the same
formulas the kernels implement, `HF:modeling_seggpt.py`), with a mantissa rounding to 11 bits (IEEE half) or 8 bits
(bfloat16) injected at exactly the places where the HIP data flow of `beach_seg_amd/csrc/seggpt_api.hip` hands a value to
an MFMA as a 16-bit operand or stores it in 16 bits -- forward sites round the VALUE (straight-through gradient: the
backward then runs on the rounded value, as the kernels' saved activations do), backward sites round the GRADIENT.  The
rounding ignores the exponent range (the f16 dgrad chain runs on a power-of-two multiple of the gradient chosen on device,
`rowops.hpp`).  fp32 accumulation, the fp32 residual stream, fp32 softmax statistics / LayerNorm and the split-precision
patch embedding are exact in the model, as they are in the kernels.

For every site class ALONE, for all of them together and for the "what if we lifted X" subsets, it reports the error of
`pred_masks` and of the prompt-pixel gradient against the un-rounded run: max|a-b| / max|b| (the metric of
tests/test_gpu_parity.py) and rms(a-b) / rms(b).  Independent sources add in quadrature, so var_share = (rms_site / rms_all)^2.

    python tools/rounding_budget.py vit_large 0      # plain sigma = 0.02 weights   (~20 s per run on 8 cores)
    python tools/rounding_budget.py vit_large 4      # peaked attention, gain 4     (the vitl_peaked_e2e fixture)
    python tools/rounding_budget.py small 8          # 56 x 28 grid, 6 layers, gain 8 (the small_peaked_e2e fixture)
Writes profiles/r3_rounding_budget_<geometry>_<gain>.json.
"""
from __future__ import annotations

import json
import math
import sys
import time
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from beach_seg_amd.weights import SegGptGeometry, counter_noise, synth_state_dict  # noqa: E402

MANT_DROP = {"f16": 13, "bf16": 16}  # fp32 mantissa bits dropped (RNE): 10 / 7 explicit bits kept
DROP = 13
ON: set[str] = set()


def rnd(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even of the fp32 mantissa to the 16-bit format's width, any exponent."""
    i = x.contiguous().view(torch.int32)
    half = (1 << (DROP - 1)) - 1
    i = (i + half + ((i >> DROP) & 1)) & ~((1 << DROP) - 1)
    return i.view(torch.float32)


class _RoundST(torch.autograd.Function):  # value rounded, gradient passed through
    @staticmethod
    def forward(ctx, x):
        return rnd(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):  # value untouched, gradient rounded
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return rnd(g)


def rv(x, site):
    return _RoundST.apply(x) if site in ON else x


def rg(x, site):
    return _RoundGrad.apply(x) if site in ON else x


def at_rounded(f, x, site):
    """f(x) in value, but differentiated at round(x): the backward of the kernels re-evaluates f' from the 16-bit copy."""
    if site not in ON:
        return f(x)
    xr = x + (rnd(x.detach()) - x.detach())
    return f(xr) + (f(x.detach()) - f(xr.detach()))


class _GeluSavedDeriv(torch.autograd.Function):  # EPI_BIAS_GELU stores gelu'(h) in 16 bits for EPI_GELU_BWD
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return F.gelu(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        d = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        return g * (rnd(d) if "f_geluprime" in ON else d)


_wcache: dict = {}


def W(w, name):
    if "f_weights" not in ON:
        return w[name]
    if name not in _wcache:
        _wcache[name] = rnd(w[name])
    return _wcache[name]


def synth_inputs(g, B, seed):
    H, Wd = g.image_size[0] // 2, g.image_size[1]
    n = B * 3 * H * Wd
    pix = counter_noise(n, seed * 10 + 1).reshape(B, 3, H, Wd)
    prm = counter_noise(n, seed * 10 + 2).reshape(B, 3, H, Wd)
    blk = 16
    cls = lambda s: ((counter_noise(B * (H // blk) * (Wd // blk), s) * 1000).long().abs() % 4).reshape(
        B, H // blk, Wd // blk).repeat_interleave(blk, 1).repeat_interleave(blk, 2)
    pal = ((counter_noise(B * 4 * 3, seed * 10 + 5) * 1000).long().abs() % 256).to(torch.uint8).reshape(B, 4, 3)
    pal[:, 0] = 0
    return pix, prm, cls(seed * 10 + 3), cls(seed * 10 + 4), pal


def colour(pal, ids):
    mean = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
    rgb = pal[torch.arange(pal.shape[0])[:, None, None], ids.long()].permute(0, 3, 1, 2).float() / 255
    return (rgb - mean) / std


def patch_rows(img, p):
    B, C, H, Wd = img.shape
    return img.reshape(B, C, H // p, p, Wd // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // p) * (Wd // p), C * p * p)


def forward(w, g, pix, prm, pmask):
    """Value flow of forward_impl / backward_impl with the rounding sites marked."""
    e = "model.embeddings."
    D, N, (hp, wp), nh = g.hidden_size, g.num_tokens, g.grid, g.num_attention_heads
    img = torch.cat((prm, pix), 2)
    msk = torch.cat((pmask, pmask), 2)
    Wp = w[e + "patch_embeddings.projection.weight"].reshape(D, -1)  # split precision: exact
    xi = patch_rows(img, 16) @ Wp.t() + w[e + "patch_embeddings.projection.bias"]
    xm = patch_rows(msk, 16) @ Wp.t() + w[e + "patch_embeddings.projection.bias"]
    m = (torch.arange(N) >= N // 2).float().reshape(1, N, 1)
    xm = xm * (1 - m) + w[e + "mask_token"].reshape(1, 1, -1) * m
    pe = w[e + "position_embeddings"][:, 1:]
    n = int(round(math.sqrt(pe.shape[1])))
    if n != hp or n != wp:
        pe = F.interpolate(pe.reshape(1, n, n, -1).permute(0, 3, 1, 2), size=(hp, wp), mode="bicubic", align_corners=False).permute(0, 2, 3, 1)
    pos, ty = pe.reshape(N, D), w[e + "type_token_instance"].reshape(1, 1, -1)
    x = torch.cat((xi + w[e + "segment_token_input"].reshape(1, 1, -1) + pos + ty,
                   xm + w[e + "segment_token_prompt"].reshape(1, 1, -1) + pos + ty), 0)
    ih = torch.arange(hp)[:, None] - torch.arange(hp)[None, :] + hp - 1
    iw = torch.arange(wp)[:, None] - torch.arange(wp)[None, :] + wp - 1
    taps = []
    for i in range(g.num_hidden_layers):
        l = f"model.encoder.layers.{i}."
        S = x.shape[0]
        a = F.layer_norm(x, (D,), w[l + "layernorm_before.weight"], w[l + "layernorm_before.bias"], g.layer_norm_eps)
        a = rg(rv(a, "f_ln"), "b_dln")                                   # ln_out (T) | dn_a (T) into ln_bwd
        qkv = a @ W(w, l + "attention.qkv.weight").t() + w[l + "attention.qkv.bias"]
        qkv = rg(qkv, "b_dqkv").reshape(S, N, 3, nh, 64).permute(2, 0, 3, 1, 4)  # dqkv (T): A of the qkv^T dgrad
        q, k, v = rv(qkv[0], "f_qk"), rv(qkv[1], "f_qk"), rv(qkv[2], "f_v")      # qkv stored in T
        att = (q * 0.125) @ k.transpose(-2, -1)
        relc_h = rnd(w[l + "attention.rel_pos_h"]) if "f_weights" in ON else w[l + "attention.rel_pos_h"]
        relc_w = rnd(w[l + "attention.rel_pos_w"]) if "f_weights" in ON else w[l + "attention.rel_pos_w"]
        qg = q.reshape(S, nh, hp, wp, 64)
        att = att.reshape(S, nh, hp, wp, hp, wp) + torch.einsum("snhwc,hkc->snhwk", qg, relc_h[ih])[..., :, None] \
            + torch.einsum("snhwc,wkc->snhwk", qg, relc_w[iw])[..., None, :]
        att = rg(att.reshape(S, nh, N, N), "b_dS")                       # dS (T): operand of dQ / dK
        p = rv(torch.softmax(att, -1), "f_P")                            # P (T): operand of PV and of dV = P^T dO
        o = rg(rv((p @ v).permute(0, 2, 1, 3).reshape(S, N, D), "f_attn_o"), "b_dO")  # attn_o (T) | dO = dn_b (T)
        o = o @ W(w, l + "attention.proj.weight").t() + w[l + "attention.proj.bias"]
        x = x + rg(o, "b_dx")                                            # dx_t (T): A of the proj^T dgrad
        h = F.layer_norm(x, (D,), w[l + "layernorm_after.weight"], w[l + "layernorm_after.bias"], g.layer_norm_eps)
        h = rg(rv(h, "f_ln"), "b_dln")
        h = rg(h @ W(w, l + "mlp.lin1.weight").t() + w[l + "mlp.lin1.bias"], "b_dh")  # dh (T): A of the fc1^T dgrad
        h = rv(_GeluSavedDeriv.apply(h), "f_gelu")                       # h_act (T)
        x = x + rg(h @ W(w, l + "mlp.lin2.weight").t() + w[l + "mlp.lin2.bias"], "b_dx")
        if i == g.merge_index:
            B = x.shape[0] // 2
            x = (x[:B] + x[B:]) * 0.5
        if i in g.intermediate_hidden_state_indices:
            t = F.layer_norm(x, (D,), w["model.encoder.layernorm.weight"], w["model.encoder.layernorm.bias"], g.layer_norm_eps)
            taps.append(rg(rv(t, "f_ln"), "b_dtaps"))                    # taps (T) | dtaps (T) into ln_bwd
    feats = torch.cat(taps, -1)
    B, dd = feats.shape[0], g.decoder_hidden_size
    y = feats @ W(w, "decoder.decoder_embed.weight").t() + w["decoder.decoder_embed.bias"]
    y = rg(rv(y, "f_feat"), "b_dfeat")                                   # feat (T) | dfeat (T): A of the dec^T dgrad
    y = y.reshape(B, hp, wp, 16, 16, dd).permute(0, 5, 1, 3, 2, 4).reshape(B, dd, hp * 16, wp * 16)
    y = F.conv2d(y, W(w, "decoder.decoder_pred.conv.weight"), w["decoder.decoder_pred.conv.bias"], padding=1)
    y = rg(y, "b_dconv")                                                 # dconv (T): operand of the conv dgrad

    def head(c):  # LN(C) + GELU + 1x1: forward from the fp32 accumulators, backward re-evaluated from conv_out (T)
        z = F.layer_norm(c.permute(0, 2, 3, 1), (dd,), w["decoder.decoder_pred.layernorm.weight"],
                         w["decoder.decoder_pred.layernorm.bias"], g.layer_norm_eps).permute(0, 3, 1, 2)
        return F.conv2d(F.gelu(z), w["decoder.decoder_pred.head.weight"], w["decoder.decoder_pred.head.bias"])

    return at_rounded(head, y, "f_convout")


def loss_fn(pred, labels, yes, beta=0.01):
    H = labels.shape[2]
    l = F.smooth_l1_loss(pred[:, :, H:], labels, reduction="none", beta=beta)
    keep = yes.float().expand(-1, 3, -1, -1)
    return (l * keep).sum() / keep.sum()  # B = 1: the reference and the per-sample variant coincide


FWD = ["f_weights", "f_ln", "f_qk", "f_v", "f_P", "f_attn_o", "f_gelu", "f_geluprime", "f_feat", "f_convout"]
BWD = ["b_dconv", "b_dfeat", "b_dtaps", "b_dx", "b_dh", "b_dln", "b_dO", "b_dS", "b_dqkv"]
WHAT = {
    "f_weights": "Linear / conv / rel-pos weights rounded to T (static operands of every GEMM)",
    "f_ln": "LayerNorm outputs (A of qkv, fc1, decoder_embed)", "f_qk": "q, k stored in T (QK^T operands)",
    "f_v": "v stored in T (PV operand)", "f_P": "softmax probabilities as the PV / dV operand",
    "f_attn_o": "attention output (A of proj)", "f_gelu": "GELU output (A of fc2)",
    "f_geluprime": "gelu'(h) saved in T for the backward", "f_feat": "decoder feature map (operand of the 3x3 conv)",
    "f_convout": "conv output saved in T (head backward re-evaluates LN / GELU from it)",
    "b_dconv": "d conv_out (operand of the conv dgrad)", "b_dfeat": "d feat (A of the decoder_embed dgrad)",
    "b_dtaps": "d taps (into the tap LayerNorm backward)", "b_dx": "residual gradient copy dx_t (A of fc2^T / proj^T dgrads)",
    "b_dh": "d h_pre (A of the fc1^T dgrad)", "b_dln": "dn_a (GEMM output in T into ln_bwd)", "b_dO": "dO (attention backward operand)",
    "b_dS": "dS (operand of dQ / dK)", "b_dqkv": "dqkv (A of the qkv^T dgrad)",
}
# what a "lift" would cost in extra MFMA passes of the named GEMMs (a hi+lo split of ONE operand doubles that GEMM's K loop;
# both operands: x3), as a fraction of the train step's 3268 GF/tile -- see DESIGN.md section 2
SCENARIOS = {
    "all": FWD + BWD,
    "forward_sites_only": FWD, "backward_sites_only": BWD,
    "all_but_weights": [s for s in FWD + BWD if s != "f_weights"],
    "all_but_attention_operands": [s for s in FWD + BWD if s not in ("f_qk", "f_v", "f_P", "b_dO", "b_dS")],
    "all_but_qk": [s for s in FWD + BWD if s != "f_qk"],
    "all_but_decoder_tail": [s for s in FWD + BWD if s not in ("f_feat", "f_convout", "b_dconv", "b_dfeat", "b_dtaps")],
    "all_but_activation_A_operands": ["f_weights", "f_qk", "f_v", "f_P", "b_dS", "b_dO"],
}


def main():
    global DROP
    gname = sys.argv[1] if len(sys.argv) > 1 else "vit_large"
    gain = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    fmt = sys.argv[3] if len(sys.argv) > 3 else "f16"
    DROP = MANT_DROP[fmt]
    torch.set_num_threads(8)
    g = getattr(SegGptGeometry, gname)()
    w = synth_state_dict(g, seed=0 if gname == "vit_large" else 2)
    if gain:
        D = g.hidden_size
        for i in range(g.num_hidden_layers):
            l = f"model.encoder.layers.{i}.attention."
            w[l + "qkv.weight"][: 2 * D] *= gain
            w[l + "qkv.bias"][: 2 * D] *= gain
            w[l + "rel_pos_h"] *= gain
            w[l + "rel_pos_w"] *= gain
    pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, 1, 7 if gname == "vit_large" else 6)
    pm, lab, yes = colour(pal, pm_cls), colour(pal, lb_cls), (lb_cls != 0)[:, None]

    def run(sites):
        ON.clear()
        ON.update(sites)
        _wcache.clear()
        p = prm.clone().requires_grad_(True)
        pred = forward(w, g, pix, p, pm)
        (grad,) = torch.autograd.grad(loss_fn(pred, lab, yes), p)
        return pred.detach(), grad

    t0 = time.time()
    base = run([])
    print(f"[budget] {gname} gain {gain} {fmt}: base run {time.time() - t0:.0f} s", flush=True)

    def err(r):
        out = {}
        for name, a, b in (("pred", r[0], base[0]), ("grad", r[1], base[1])):
            d = (a - b).double()
            out[name + "_maxabs_rel"] = float(d.abs().max() / b.abs().max())
            out[name + "_rms_rel"] = float(d.pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt())
        return out

    res = {"_meta": {"geometry": gname, "peak_gain": gain, "format": fmt, "batch": 1,
                     "metric": "max|a-b|/max|b| and rms(a-b)/rms(b) against the un-rounded fp32 run of the same model",
                     "note": "precision MODEL of the HIP data flow (CPU, fp32 torch + mantissa rounding at the 16-bit sites); "
                             "the measured kernels are in tests/test_gpu_parity.py"}, "sites": {}, "scenarios": {}}
    for name, sites in SCENARIOS.items():
        res["scenarios"][name] = dict(err(run(sites)), sites=sites)
        print(f"[budget] scenario {name}: {res['scenarios'][name]}", flush=True)
    ra = res["scenarios"]["all"]
    for s in FWD + BWD:
        e = err(run([s]))
        e["pred_var_share"] = (e["pred_rms_rel"] / ra["pred_rms_rel"]) ** 2
        e["grad_var_share"] = (e["grad_rms_rel"] / ra["grad_rms_rel"]) ** 2
        e["what"] = WHAT[s]
        res["sites"][s] = e
        print(f"[budget] site {s}: pred rms {e['pred_rms_rel']:.2e} ({e['pred_var_share'] * 100:.0f} %)  grad rms "
              f"{e['grad_rms_rel']:.2e} ({e['grad_var_share'] * 100:.0f} %)", flush=True)
    out = ROOT / "profiles" / f"r3_rounding_budget_{gname}_{'peaked' if gain else 'plain'}_{fmt}.json"
    out.write_text(json.dumps(res, indent=1))
    print(f"[budget] wrote {out} ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
