"""Generates `tests/golden/*.npz` from the REFERENCE itself.  Runs only in the build container
(needs `/root/reference` and the installed `transformers` package); the GPU box never runs it.

What is executed:
  * the network: `transformers.SegGptForImageSegmentation` (the third-party module the reference
    calls at `src/util/ml_util.py:8`), random-init replaced by `beach_seg_amd.weights.synth_state_dict`
    because `BAAI/seggpt-vit-large` is not available offline;
  * the wrapper math: `SegGptLoss`, `PromptModel.process_pred_masks`, `PromptModel.create_palette`
    from `/root/reference/src/model.py`, `build_palette`, `torch_apply_mask_rgb`,
    `generate_random_rgb_palette` from `/root/reference/src/util/ml_util.py`, and `Accumulator.update`
    from `/root/reference/src/predict.py` -- imported with the absent third-party packages
    (lightning, torchvision, torchmetrics, kornia, shapely, rasterio, cv2, ...) stubbed in
    `sys.modules`; none of the stubbed symbols is on the arithmetic path.

Usage:  python oracle/gen_golden.py [--skip-vitl]
"""
from __future__ import annotations

import argparse
import dataclasses
import sys
import tempfile
import types
import zlib
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from beach_seg_amd.weights import SegGptGeometry, counter_noise, synth_state_dict  # noqa: E402
from oracle.gen_inputs import synth_inputs  # noqa: E402

GOLD = ROOT / "tests" / "golden"
PEAK_GAIN = 8.0  # q / k / rel-pos gain of the peaked-attention fixture (oracle/gen_inputs.peaked_state_dict)
VITL_PEAK_GAIN = 4.0  # the same for the 1024-wide net: logits scale with hidden_size * gain^2 (sigma = 0.02 weights)


def import_reference():
    import transformers  # noqa: F401  (must precede the stubs)

    class _LM(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return torch.device("cpu")

    lp = types.ModuleType("lightning.pytorch")
    lp.LightningModule = _LM
    lp.LightningDataModule = object
    lp.seed_everything = lambda *a, **k: None
    lightning = types.ModuleType("lightning")
    lightning.pytorch = lp
    sys.modules["lightning"] = lightning
    sys.modules["lightning.pytorch"] = lp
    for name in ["torchvision", "torchvision.utils", "torchmetrics", "torchmetrics.classification", "kornia",
                 "kornia.augmentation", "kornia.constants", "shapely", "shapely.geometry", "shapely.ops",
                 "rasterio", "rasterio.features", "rasterio.merge", "rasterio.warp", "rasterio.transform",
                 "rasterio.io", "rasterio.enums", "rasterio.crs", "rasterio.windows", "geopandas", "cv2",
                 "omegaconf", "dotenv", "skimage", "skimage.morphology", "skimage.measure", "skimage.graph",
                 "affine"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = MagicMock()
    sys.path.insert(0, "/root/reference")
    import src.model as ref_model
    import src.predict as ref_predict
    import src.util.ml_util as ref_ml

    return ref_model, ref_ml, ref_predict


def hf_model(g: SegGptGeometry, sd: dict):
    from transformers import SegGptConfig, SegGptForImageSegmentation

    m = SegGptForImageSegmentation(SegGptConfig(**g.to_hf_kwargs()))
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    for p in m.parameters():  # src/util/ml_util.py:9-10
        p.requires_grad_(False)
    return m.eval()  # src/util/ml_util.py:11 (torch.compile changes no arithmetic contract)




def attention_peak_stats(m, g, pix, prm, prompt_masks) -> dict:
    """Row-max logit and row-max probability of every attention block of the HF module (hooks on the softmax inputs are
    not exposed, so the logits are re-formed from each block's own qkv output the way HF:313-331 does)."""
    stats = {"max_logit": [], "mean_rowmax_prob": []}
    hooks = []

    def hook(mod, inp, out):
        x = inp[0]
        B_, H_, W_, _ = x.shape
        qkv = mod.qkv(x).reshape(B_, H_ * W_, 3, mod.num_attention_heads, -1).permute(2, 0, 3, 1, 4)
        q, k, _ = qkv.reshape(3, B_ * mod.num_attention_heads, H_ * W_, -1).unbind(0)
        attn = (q * mod.scale) @ k.transpose(-2, -1)
        attn = mod.add_decomposed_rel_pos(attn, q, mod.rel_pos_h, mod.rel_pos_w, (H_, W_), (H_, W_))
        stats["max_logit"].append(float((attn.max(-1).values - attn.mean(-1)).max()))
        stats["mean_rowmax_prob"].append(float(torch.softmax(attn.float(), -1).max(-1).values.mean()))

    for layer in m.model.encoder.layers:
        hooks.append(layer.attention.register_forward_hook(hook))
    with torch.no_grad():
        m(pixel_values=pix, prompt_pixel_values=prm, prompt_masks=prompt_masks, embedding_type="instance")
    for h in hooks:
        h.remove()
    return stats


def run_e2e(ref_model, ref_ml, g: SegGptGeometry, B: int, wseed: int, iseed: int, tag: str, full: bool, peaked: float = 0.0,
            every_layer: bool = False):
    if peaked:
        from oracle.gen_inputs import peaked_state_dict
        sd = peaked_state_dict(g, wseed, peaked)
    else:
        sd = synth_state_dict(g, seed=wseed)
    m = hf_model(g, sd)
    pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, B, iseed)
    bare = ref_model.PromptModel.__new__(ref_model.PromptModel)  # no __init__: needs hub weights
    torch.nn.Module.__init__(bare)
    bare.num_classes = 4
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    bare.normalize = lambda x: (x - mean) / std
    pal_norm = torch.stack([bare.normalize(p.view(4, 3, 1, 1).to(torch.float32) / 255).squeeze(-1).squeeze(-1)
                            for p in pal])  # src/model.py:221-229
    prompt_masks = bare.normalize(ref_ml.torch_apply_mask_rgb(pal, pm_cls[:, None]))  # src/model.py:209-210
    labels = bare.normalize(ref_ml.torch_apply_mask_rgb(pal, lb_cls[:, None]))  # src/model.py:238-239
    yes = (lb_cls != 0)[:, None]
    prm = prm.clone().requires_grad_(True)
    out = m(pixel_values=pix, labels=labels, prompt_pixel_values=prm, prompt_masks=prompt_masks,
            embedding_type="instance")  # src/model.py:245-251
    pred = out.pred_masks
    loss = ref_model.SegGptLoss(0.01)(pred, labels, yes)  # src/model.py:255
    (grad,) = torch.autograd.grad(loss, prm)
    masks = bare.process_pred_masks(pred.detach(), pal_norm)  # src/model.py:252
    with torch.no_grad():  # inference call without labels, src/model.py:139-144
        pred_nolab = m(pixel_values=pix, prompt_pixel_values=prm.detach(), prompt_masks=prompt_masks,
                       embedding_type="instance").pred_masks
    assert torch.equal(pred_nolab, pred.detach()), "labels must not influence pred_masks (HF:712-715)"
    loss_b1 = [float(ref_model.SegGptLoss(0.01)(pred[i:i + 1].detach(), labels[i:i + 1], yes[i:i + 1]))
               for i in range(B)]
    rec = dict(geometry=np.array(list(g.image_size) + [g.hidden_size, g.num_hidden_layers, g.num_attention_heads]),
               wseed=wseed, iseed=iseed, B=B, palette=pal.numpy(), pal_norm=pal_norm.numpy(),
               prompt_cls=pm_cls.numpy(), label_cls=lb_cls.numpy(), loss=float(loss),
               loss_b1=np.array(loss_b1), masks_crc=zlib.crc32(masks.to(torch.uint8).numpy().tobytes()))
    pred_np, grad_np = pred.detach().numpy(), grad.numpy()
    if full:
        rec.update(pixel_values=pix.numpy(), prompt_pixel_values=prm.detach().numpy(), pred=pred_np, grad=grad_np,
                   masks=masks.to(torch.uint8).numpy())
    else:  # inputs are regenerated from iseed by the tests; keep strided slices + norms
        st = 8
        rec.update(pred_slice=pred_np[:, :, ::st, ::st], grad_slice=grad_np[:, :, ::st, ::st],
                   masks_slice=masks.to(torch.uint8).numpy()[:, ::st, ::st], stride=st,
                   pred_l2=float(np.sqrt((pred_np.astype(np.float64) ** 2).sum())),
                   grad_l2=float(np.sqrt((grad_np.astype(np.float64) ** 2).sum())))
    if peaked:
        st_ = attention_peak_stats(m, g, pix, prm.detach(), prompt_masks)
        rec.update(peak_gain=peaked, max_logit_above_row_mean=np.array(st_["max_logit"]),
                   mean_rowmax_prob=np.array(st_["mean_rowmax_prob"]))
        print(tag, "max logit above row mean per layer", [round(v, 1) for v in st_["max_logit"]],
              "mean row-max prob", [round(v, 3) for v in st_["mean_rowmax_prob"]])
        assert max(st_["max_logit"]) > 20.0, "the peaked fixture must drive logits past 20"
        if every_layer:
            assert min(st_["max_logit"]) > 20.0, "every layer's row-max logit must clear its row mean by 20"
    np.savez_compressed(GOLD / f"{tag}.npz", **rec)
    print(tag, "loss", float(loss), "pred_l2", float(pred.norm()), "grad_l2", float(grad.norm()))
    return m, sd


def run_feature_ensemble(g: SegGptGeometry, wseed: int, iseed: int):
    """`feature_ensemble=True` with K prompts for one query (`src/predict_no_prompt.py:283-304` usage)."""
    sd = synth_state_dict(g, seed=wseed)
    m = hf_model(g, sd)
    K = 3
    pix, prm, pm_cls, _, pal = synth_inputs(g, K, iseed)
    pix = pix[:1].expand(K, -1, -1, -1).contiguous()
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    rgb = pal[torch.arange(K)[:, None, None], pm_cls.long()].permute(0, 3, 1, 2).float() / 255
    prompt_masks = (rgb - mean) / std
    with torch.no_grad():
        pred = m(pixel_values=pix, prompt_pixel_values=prm, prompt_masks=prompt_masks, embedding_type="instance",
                 feature_ensemble=True).pred_masks
    np.savez_compressed(GOLD / "tiny_feature_ensemble.npz", pixel_values=pix.numpy(),
                        prompt_pixel_values=prm.numpy(), prompt_masks=prompt_masks.numpy(), pred=pred.numpy(),
                        wseed=wseed)
    print("feature_ensemble pred_l2", float(pred.norm()))


def run_wrapper(ref_model, ref_ml):
    """G4: wrapper arithmetic on fixed inputs incl. the B>1 loss broadcast and a near-tie decode."""
    rec = {}
    rec["build_palette_3"] = np.array(ref_ml.build_palette(3))
    rec["build_palette_7"] = np.array(ref_ml.build_palette(7))
    torch.manual_seed(42)
    pal = ref_ml.generate_random_rgb_palette(4, 3, "cpu")
    rec["rand_palette_seed42"] = pal.numpy()
    mask = ((counter_noise(3 * 32 * 32, 77) * 1000).long().abs() % 4).reshape(3, 1, 32, 32).to(torch.uint8)
    rec["mask"] = mask.numpy()
    rec["apply_mask_rgb"] = ref_ml.torch_apply_mask_rgb(pal, mask).numpy()
    bare = ref_model.PromptModel.__new__(ref_model.PromptModel)
    torch.nn.Module.__init__(bare)
    bare.num_classes = 4
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    bare.normalize = lambda x: (x - mean) / std
    pal_norm = torch.stack([bare.normalize(p.view(4, 3, 1, 1).to(torch.float32) / 255).squeeze(-1).squeeze(-1)
                            for p in pal])
    rec["pal_norm"] = pal_norm.numpy()
    pred = counter_noise(3 * 3 * 64 * 32, 78).reshape(3, 3, 64, 32) * 1.5
    # near-ties: put some pixels (almost) on the bisector of two palette colours
    mid = 0.5 * (pal_norm[:, 1] + pal_norm[:, 2])
    pred[:, :, 40, 5] = mid
    pred[:, :, 41, 6] = mid + 1e-7
    pred[:, :, 42, 7] = pal_norm[:, 3]
    rec["decode_pred"] = pred.numpy()
    rec["decode_masks"] = bare.process_pred_masks(pred, pal_norm).numpy().astype(np.uint8)
    labels = counter_noise(3 * 3 * 32 * 32, 79).reshape(3, 3, 32, 32)
    yes = (mask != 0)
    rec["loss_labels"] = labels.numpy()
    for beta in (0.01, 0.5):
        fn = ref_model.SegGptLoss(beta)
        p = pred.clone().requires_grad_(True)
        l3 = fn(p, labels, yes)
        (g3,) = torch.autograd.grad(l3, p)
        rec[f"loss_B3_beta{beta}"] = float(l3)
        rec[f"loss_B3_grad_beta{beta}"] = g3.numpy()
        rec[f"loss_B1_beta{beta}"] = np.array([float(fn(pred[i:i + 1], labels[i:i + 1], yes[i:i + 1]))
                                               for i in range(3)])
    np.savez_compressed(GOLD / "wrapper.npz", **rec)
    print("wrapper ok", rec["loss_B3_beta0.01"], rec["loss_B1_beta0.01"])


def run_predict_glue(ref_predict):
    """G5: `Accumulator.update` clipping + vote arg-max on a small mosaic."""
    with tempfile.TemporaryDirectory() as td:
        acc = ref_predict.Accumulator((40, 50), Path(td), None, None, ("nodata", "sand", "water", "veg"))
        acc.initialize_current("d0")
        crops = [(-5, -3, 11, 13), (10, 10, 26, 26), (40, 30, 56, 46), (12, 8, 28, 24), (60, 60, 76, 76),
                 (-20, 5, -4, 21), (0, 0, 16, 16), (34, 24, 50, 40)]
        preds = []
        for i, c in enumerate(crops):
            pr = ((counter_noise(16 * 16, 900 + i) * 1000).long().abs() % 4).reshape(16, 16).numpy()
            preds.append(pr)
            oh = np.eye(4, dtype=np.uint8)[pr]
            acc.update("d0", c, oh, np.zeros((16, 16, 3), np.uint8), None)
        counter = acc.current_pred_counter.copy()
        final = np.argmax(counter, axis=2)
        acc.current_pred_counter = None  # keep __exit__/save_current (PNG/GeoTIFF IO) out of it
    np.savez_compressed(GOLD / "predict_glue.npz", crops=np.array(crops), preds=np.array(preds).astype(np.uint8),
                        counter=counter, final=final.astype(np.uint8))
    print("predict glue ok", counter.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-vitl", action="store_true")
    ap.add_argument("--only", default="", help="generate one case only: small_peaked | tiny_dec128 | vitl_peaked")
    ap.add_argument("--gain", type=float, default=0.0, help="peak gain override (vitl_peaked)")
    args = ap.parse_args()
    GOLD.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    ref_model, ref_ml, ref_predict = import_reference()
    if args.only == "small_peaked":
        run_e2e(ref_model, ref_ml, SegGptGeometry.small(), B=2, wseed=2, iseed=6, tag="small_peaked_e2e", full=False, peaked=PEAK_GAIN)
        return
    if args.only == "tiny_dec128":  # decoder_hidden_size = 128 (BASELINE config 5's decoder width) through the HF module
        run_e2e(ref_model, ref_ml, dataclasses.replace(SegGptGeometry.tiny(), decoder_hidden_size=128), B=2, wseed=4, iseed=8,
                tag="tiny_dec128_e2e", full=True)
        return
    if args.only == "vitl_peaked":  # full ViT-L with PEAKED attention (what a trained checkpoint produces), 24 layers deep
        run_e2e(ref_model, ref_ml, SegGptGeometry.vit_large(), B=1, wseed=0, iseed=7, tag="vitl_peaked_e2e", full=False,
                peaked=args.gain or VITL_PEAK_GAIN, every_layer=True)
        return
    run_wrapper(ref_model, ref_ml)
    run_predict_glue(ref_predict)
    run_e2e(ref_model, ref_ml, SegGptGeometry.tiny(), B=2, wseed=1, iseed=3, tag="tiny_e2e", full=True)
    run_feature_ensemble(SegGptGeometry.tiny(), wseed=1, iseed=4)
    run_e2e(ref_model, ref_ml, SegGptGeometry.small(), B=2, wseed=2, iseed=5, tag="small_e2e", full=False)
    run_e2e(ref_model, ref_ml, SegGptGeometry.small(), B=2, wseed=2, iseed=6, tag="small_peaked_e2e", full=False, peaked=PEAK_GAIN)
    if not args.skip_vitl:
        run_e2e(ref_model, ref_ml, SegGptGeometry.vit_large(), B=1, wseed=0, iseed=7, tag="vitl_e2e", full=False)
        run_e2e(ref_model, ref_ml, SegGptGeometry.vit_large(), B=1, wseed=0, iseed=7, tag="vitl_peaked_e2e", full=False,
                peaked=VITL_PEAK_GAIN, every_layer=True)
    run_e2e(ref_model, ref_ml, dataclasses.replace(SegGptGeometry.tiny(), decoder_hidden_size=128), B=2, wseed=4, iseed=8,
            tag="tiny_dec128_e2e", full=True)


if __name__ == "__main__":
    main()
