"""Pins the CPU oracle (`oracle/`) against vectors produced by the reference itself
(`oracle/gen_golden.py`: HF SegGPT + `/root/reference/src/model.py` wrapper math)."""
import zlib

import numpy as np
import pytest
import torch

from beach_seg_amd.weights import SegGptGeometry, synth_state_dict
from oracle import frontend_oracle as FO
from oracle import predict_oracle as PO
from oracle import seggpt_oracle as O
from oracle.gen_inputs import synth_inputs


def _e2e(g, rec, B):
    if "peak_gain" in rec.files:
        from oracle.gen_inputs import peaked_state_dict
        w = peaked_state_dict(g, int(rec["wseed"]), float(rec["peak_gain"]))
    else:
        w = synth_state_dict(g, seed=int(rec["wseed"]))
    pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, B, int(rec["iseed"]))
    assert np.array_equal(pal.numpy(), rec["palette"])
    pn = O.palette_norm(pal)
    np.testing.assert_allclose(pn.numpy(), rec["pal_norm"], rtol=0, atol=1e-6)
    pm = O.normalize(O.apply_mask_rgb(pal, pm_cls))
    lab = O.normalize(O.apply_mask_rgb(pal, lb_cls))
    prm = prm.clone().requires_grad_(True)
    pred = O.forward(w, g, pix, prm, pm, labels=lab)
    loss = O.seggpt_loss(pred, lab, (lb_cls != 0)[:, None], 0.01, "reference")
    (grad,) = torch.autograd.grad(loss, prm)
    masks = O.decode_argmin(pred.detach(), pn)
    return pred.detach(), float(loss), grad, masks, lab, lb_cls


def test_tiny_end_to_end(golden_dir):
    rec = np.load(golden_dir / "tiny_e2e.npz")
    g = SegGptGeometry.tiny()
    pred, loss, grad, masks, lab, lb_cls = _e2e(g, rec, 2)
    np.testing.assert_allclose(pred.numpy(), rec["pred"], rtol=1e-4, atol=2e-5)
    assert abs(loss - float(rec["loss"])) < 1e-5 * abs(float(rec["loss"]))
    gref = rec["grad"]
    assert np.abs(grad.numpy() - gref).max() < 1e-3 * np.abs(gref).max()
    assert np.array_equal(masks.numpy().astype(np.uint8), rec["masks"])
    for i in range(2):  # B=1 loss == per-sample masked mean (src/model.py:259 "Assume batch size 1")
        l1 = O.seggpt_loss(pred[i:i + 1], lab[i:i + 1], (lb_cls != 0)[i:i + 1, None], 0.01, "per_sample")
        assert abs(float(l1) - rec["loss_b1"][i]) < 1e-5


def test_tiny_decoder_width_128(golden_dir):
    """decoder_hidden_size = 128 (BASELINE config 5's decoder: `HF:configuration_seggpt.py:57-75`, head `HF:525-579`), vector
    produced by the HF module: pins the oracle's decoder for a width other than the checkpoint's 64."""
    import dataclasses

    rec = np.load(golden_dir / "tiny_dec128_e2e.npz")
    g = dataclasses.replace(SegGptGeometry.tiny(), decoder_hidden_size=128)
    pred, loss, grad, masks, _, _ = _e2e(g, rec, 2)
    np.testing.assert_allclose(pred.numpy(), rec["pred"], rtol=1e-4, atol=2e-5)
    assert abs(loss - float(rec["loss"])) < 1e-5 * abs(float(rec["loss"]))
    assert np.abs(grad.numpy() - rec["grad"]).max() < 1e-3 * np.abs(rec["grad"]).max()
    assert np.array_equal(masks.numpy().astype(np.uint8), rec["masks"])


def test_small_end_to_end(golden_dir):
    rec = np.load(golden_dir / "small_e2e.npz")
    g = SegGptGeometry.small()
    pred, loss, grad, masks, _, _ = _e2e(g, rec, 2)
    st = int(rec["stride"])
    np.testing.assert_allclose(pred.numpy()[:, :, ::st, ::st], rec["pred_slice"], rtol=1e-4, atol=2e-5)
    assert abs(loss - float(rec["loss"])) < 1e-5 * abs(float(rec["loss"]))
    gs = rec["grad_slice"]
    assert np.abs(grad.numpy()[:, :, ::st, ::st] - gs).max() < 1e-3 * np.abs(gs).max()
    assert abs(float(pred.double().norm()) - float(rec["pred_l2"])) < 1e-4 * float(rec["pred_l2"])
    assert abs(float(grad.double().norm()) - float(rec["grad_l2"])) < 1e-3 * float(rec["grad_l2"])
    m8 = masks.numpy().astype(np.uint8)
    assert np.array_equal(m8[:, ::st, ::st], rec["masks_slice"])
    assert zlib.crc32(m8.tobytes()) == int(rec["masks_crc"])


def test_small_peaked_attention_end_to_end(golden_dir):
    """q / k / rel-pos scaled x8: logits reach 24-28 above the row mean and the mean row-max probability is ~0.4, so
    the softmax is peaked the way a trained checkpoint's is (HF-generated vector, `oracle/gen_golden.py`)."""
    rec = np.load(golden_dir / "small_peaked_e2e.npz")
    assert rec["max_logit_above_row_mean"].min() > 20
    g = SegGptGeometry.small()
    pred, loss, grad, masks, _, _ = _e2e(g, rec, 2)
    st = int(rec["stride"])
    np.testing.assert_allclose(pred.numpy()[:, :, ::st, ::st], rec["pred_slice"], rtol=1e-4, atol=5e-5)
    assert abs(loss - float(rec["loss"])) < 1e-5 * abs(float(rec["loss"]))
    gs = rec["grad_slice"]
    assert np.abs(grad.numpy()[:, :, ::st, ::st] - gs).max() < 1e-3 * np.abs(gs).max()
    assert abs(float(grad.double().norm()) - float(rec["grad_l2"])) < 1e-3 * float(rec["grad_l2"])
    m8 = masks.numpy().astype(np.uint8)
    assert zlib.crc32(m8.tobytes()) == int(rec["masks_crc"])


def test_feature_ensemble(golden_dir):
    rec = np.load(golden_dir / "tiny_feature_ensemble.npz")
    g = SegGptGeometry.tiny()
    w = synth_state_dict(g, seed=int(rec["wseed"]))
    pred = O.forward(w, g, torch.from_numpy(rec["pixel_values"]), torch.from_numpy(rec["prompt_pixel_values"]),
                     torch.from_numpy(rec["prompt_masks"]), feature_ensemble=True)
    np.testing.assert_allclose(pred.numpy(), rec["pred"], rtol=1e-4, atol=2e-5)


def test_wrapper_math(golden_dir):
    rec = np.load(golden_dir / "wrapper.npz")
    assert np.array_equal(np.array(O.build_palette(3)), rec["build_palette_3"])
    assert np.array_equal(np.array(O.build_palette(7)), rec["build_palette_7"])
    pal = torch.from_numpy(rec["rand_palette_seed42"])
    mask = torch.from_numpy(rec["mask"])
    assert np.array_equal(O.apply_mask_rgb(pal, mask).numpy(), rec["apply_mask_rgb"])
    pn = O.palette_norm(pal)
    np.testing.assert_allclose(pn.numpy(), rec["pal_norm"], rtol=0, atol=1e-6)
    pred = torch.from_numpy(rec["decode_pred"])
    # decode from the reference's own normalised palette: must be bit-exact incl. the near-tie pixels
    assert np.array_equal(O.decode_argmin(pred, torch.from_numpy(rec["pal_norm"])).numpy().astype(np.uint8),
                          rec["decode_masks"])
    labels = torch.from_numpy(rec["loss_labels"])
    yes = mask != 0
    for beta in (0.01, 0.5):
        p = pred.clone().requires_grad_(True)
        l3 = O.seggpt_loss(p, labels, yes, beta, "reference")
        (g3,) = torch.autograd.grad(l3, p)
        assert abs(float(l3) - float(rec[f"loss_B3_beta{beta}"])) < 1e-6 * abs(float(l3))
        np.testing.assert_allclose(g3.numpy(), rec[f"loss_B3_grad_beta{beta}"], rtol=1e-5, atol=1e-9)
        for i in range(3):
            l1 = O.seggpt_loss(pred[i:i + 1], labels[i:i + 1], yes[i:i + 1], beta, "per_sample")
            assert abs(float(l1) - rec[f"loss_B1_beta{beta}"][i]) < 1e-6


def test_predict_glue(golden_dir):
    rec = np.load(golden_dir / "predict_glue.npz")
    counter = np.zeros((40, 50, 4), np.uint8)
    for crop, pr in zip(rec["crops"], rec["preds"]):
        PO.accumulate(counter, tuple(int(c) for c in crop), PO.one_hot(pr.astype(np.int64), 4))
    assert np.array_equal(counter, rec["counter"])
    assert np.array_equal(PO.vote_argmax(counter).astype(np.uint8), rec["final"])
    m = np.arange(448 * 448).reshape(448, 448) % 4
    r = PO.nearest_resize(m, 112)
    assert np.array_equal(r, m[::4, ::4])  # 448 -> 112: floor(i * 4)


@pytest.mark.skipif(not (pytest.importorskip("pathlib").Path(__file__).parent / "golden" / "vitl_e2e.npz").exists(),
                    reason="ViT-L fixture not generated")
def test_vitl_fixture_present(golden_dir):
    rec = np.load(golden_dir / "vitl_e2e.npz")
    assert rec["pred_slice"].shape == (1, 3, 112, 56)
    assert np.isfinite(rec["pred_slice"]).all() and float(rec["grad_l2"]) > 0


@pytest.mark.parametrize("fixture", ["vitl_e2e.npz", "vitl_peaked_e2e.npz"])
def test_vitl_oracle_vs_reference_vectors(golden_dir, fixture):
    """The oracle at the full reference geometry (ViT-L, 24 layers, 370.7 M parameters, B = 1: ~20 s on 8 cores) against
    the vectors the HF module produced: the plain sigma = 0.02 weights and the PEAKED-attention variant (q / k / rel-pos x 4:
    row-max logit 28-43 above the row mean in EVERY layer, mean row-max probability ~0.5 -- what a trained checkpoint's
    softmax looks like, `HF:modeling_seggpt.py:313-348` 24 layers deep)."""
    rec = np.load(golden_dir / fixture)
    if "peak_gain" in rec.files:
        assert float(rec["max_logit_above_row_mean"].min()) > 20.0 and len(rec["max_logit_above_row_mean"]) == 24
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    pred, loss, grad, masks, _, _ = _e2e(SegGptGeometry.vit_large(), rec, 1)
    st = int(rec["stride"])
    ps, gs = rec["pred_slice"], rec["grad_slice"]
    assert np.abs(pred.numpy()[:, :, ::st, ::st] - ps).max() < 1e-4 * np.abs(ps).max()
    assert np.abs(grad.numpy()[:, :, ::st, ::st] - gs).max() < 1e-3 * np.abs(gs).max()
    assert abs(loss - float(rec["loss"])) < 1e-5 * abs(float(rec["loss"]))
    assert abs(float(pred.double().norm()) - float(rec["pred_l2"])) < 1e-4 * float(rec["pred_l2"])
    assert abs(float(grad.double().norm()) - float(rec["grad_l2"])) < 1e-3 * float(rec["grad_l2"])
    m8 = masks.numpy().astype(np.uint8)
    mism = int((m8[:, ::st, ::st] != rec["masks_slice"]).sum())
    assert mism <= 1, mism  # two fp32 evaluation orders: at most a near-tie pixel may flip
    if mism == 0 and zlib.crc32(m8.tobytes()) != int(rec["masks_crc"]):
        print("[note] mask CRC differs from the HF vector although the strided slice agrees (near-tie pixel off the slice)")


def test_tile_front_end_vs_pillow_and_reference(golden_dir):
    """Front-end oracle (integer restatement of Pillow's 8-bit BICUBIC + the reference's padded_crop / tif_image) against
    Pillow's own output and the reference functions' output (fixture written by oracle/gen_golden_frontend.py), and the
    product's host-side table builder / mirrors against the same vectors."""
    from beach_seg_amd.data import padded_crop, pil_bicubic_tables, tif_image

    rec = np.load(golden_dir / "frontend_pil.npz")
    m = rec["mosaic"]
    for key, boxes, crop, S in (("out112", rec["boxes112"], 112, 448), ("out256", rec["boxes256"], 256, 448),
                                ("down256_96", rec["boxes256"], 256, 96)):
        u8, f = FO.tile_frontend(m, boxes, crop, S)
        assert np.array_equal(u8, rec[key]), key                     # bit-exact vs PIL
        want = (rec[key].astype(np.float32) / np.float32(255.0) - FO.IMAGENET_MEAN) / FO.IMAGENET_STD
        assert np.array_equal(f, want.transpose(0, 3, 1, 2))
        b1, k1 = pil_bicubic_tables(crop, S)
        b2, k2 = FO.pil_coeffs(crop, S)
        assert np.array_equal(b1, b2) and np.array_equal(k1, k2)
    assert np.array_equal(FO.tif_image_4band(rec["tif_bands"], rec["tif_nodata"]), rec["tif_rgb"])
    assert np.array_equal(tif_image(rec["tif_bands"], rec["tif_nodata"]), rec["tif_rgb"])
    for box, want in zip(rec["pc_boxes"], rec["pc_out"]):
        assert np.array_equal(padded_crop(rec["pc_src"], tuple(int(v) for v in box)), want)
        assert np.array_equal(FO.padded_crop(rec["pc_src"], box, 5) if box[0] >= 0 and box[1] >= 0 else want, want)


def test_hf_post_process_decode(golden_dir):
    """Oracle restatement of SegGptImageProcessor.post_process_semantic_segmentation (the decode of the reference's
    few-shot caller) against the transformers class itself (fixture from oracle/gen_golden_postprocess.py)."""
    from beach_seg_amd.ml_util import build_palette

    rec = np.load(golden_dir / "hf_postprocess.npz")
    pred, nl = torch.from_numpy(rec["pred"]), int(rec["num_labels"])
    assert PO.hf_build_palette(nl) == [tuple(int(v) for v in row) for row in rec["palette"]] == build_palette(nl)
    assert np.array_equal(torch.stack(PO.hf_post_process(pred, nl)).numpy(), rec["full"])
    assert np.array_equal(torch.stack(PO.hf_post_process(pred, nl, (12, 10))).numpy(), rec["small"])
    assert np.array_equal(torch.stack(PO.hf_post_process(pred.mean(0, keepdim=True), nl, (16, 16))).numpy(), rec["ens"])
