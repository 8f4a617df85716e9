"""Parity tests proper (`-m gpu`): the HIP path, called through the C ABI, against (a) the golden vectors the
reference itself produced (`tests/golden/`, see `oracle/gen_golden.py`) and (b) the CPU oracle on the same
seeded inputs.

Tolerances.  north_star: "within 1e-3 rel fp32 (argmax masks bit-exact)".  The bars below are ~1.5x what each (dtype,
fixture) MEASURES on the MI355X (printed by the tests as `[measured]`; round-3 table in DESIGN.md section 2): they document what
each arithmetic mode delivers, they are not derived from the tolerance.
  * dtype float32 (parity mode, exact-f32 MFMA): max-abs error <= 1e-4 of the tensor's max-abs (measured 1e-6 .. 4e-6; 2.4e-5 /
    9.1e-5 on ViT-L with peaked attention), loss to 1e-5 rel, decoded masks BIT-EXACT.  The one mode inside north_star's bar on
    every fixture.
  * "float32x3" = dtype float32 with gemm_x3 (every GEMM / attention MFMA as three f16 MFMAs on 22-bit operand splits, f32
    storage / softmax / LayerNorm): the same errors as exact f32 (1e-6 .. 3e-6; 2.4e-5 / 7.8e-5 on peaked ViT-L) at twice its
    rate; masks by the margin rule, 0 pixels differ on every fixture.
  * dtype float16 (IEEE-half MFMA operands at the bf16 MFMA rate, fp32 accumulate / residual stream / softmax / LN, the dgrad
    chain on a device-chosen power-of-two multiple of the gradient): 5e-4 .. 7e-4 on the small nets, 1.35e-3 / 1.43e-3 on plain
    ViT-L, 9.4e-3 / 3.2e-2 on ViT-L with PEAKED attention (row-max logit 28-43 above the row mean in all 24 layers).
  * dtype bfloat16 (throughput mode; the dtype BASELINE config 1 names): 4e-3 .. 6e-3 small, 8.7e-3 / 1.1e-2 plain ViT-L,
    7.2e-2 / 2.6e-1 peaked ViT-L.
    `tools/rounding_budget.py` (a CPU precision model of the kernels' data flow) reproduces these numbers from operand rounding
    alone and attributes them (`profiles/r3_rounding_budget_*.json`): weights 62-75 % of the variance on plain ViT-L, weights +
    LayerNorm outputs + q / k on the peaked one, amplified ~1e3 x by the peaked softmax -- the format, not a kernel defect.
    Masks: every pixel whose two nearest palette colours are further apart than the proven bound 2 * delta * ||p_j - p_k||_1
    (delta = the max-abs error the test has just asserted; the e^2 terms of the two squared distances cancel) must
    decode identically, and the number of differing pixels is asserted as a COUNT.
"""
import zlib

import numpy as np
import pytest
import torch

from beach_seg_amd import ops
from beach_seg_amd.weights import SegGptGeometry, synth_state_dict
from oracle import predict_oracle as PO
from oracle import seggpt_oracle as O
from oracle.gen_inputs import peaked_state_dict, synth_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relmax(a, b):
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


_models = {}
F32X3 = "float32x3"  # pseudo-dtype of the parametrisations: SegGptNative(dtype=float32, gemm_x3=True)


def geometry_of(gname):
    import dataclasses

    if gname == "tiny_dec128":  # the tiny net with BASELINE config 5's decoder width
        return dataclasses.replace(SegGptGeometry.tiny(), decoder_hidden_size=128)
    return getattr(SegGptGeometry, gname)()


def model_for(gname, wseed, dtype, peak_gain=0.0):
    from beach_seg_amd.seggpt import SegGptNative

    key = (gname, wseed, dtype, peak_gain)
    if key not in _models:
        _models.clear()  # one resident model at a time
        g = geometry_of(gname)
        if peak_gain:
            dev_sd = peaked_state_dict(g, wseed, peak_gain, device=DEV if gname == "vit_large" else "cpu")
        else:
            dev_sd = synth_state_dict(g, seed=wseed, device=DEV if gname == "vit_large" else "cpu")
        if dtype == F32X3:  # float32 storage / attention / LayerNorm, Linear GEMMs as three f16 MFMAs on 22-bit operand splits
            _models[key] = SegGptNative(dev_sd, g, device=DEV, dtype=torch.float32, gemm_x3=True)
        else:
            _models[key] = SegGptNative(dev_sd, g, device=DEV, dtype=dtype)
    return _models[key]


def run_case(gname, rec, B, dtype):
    g = geometry_of(gname)
    model = model_for(gname, int(rec["wseed"]), dtype, float(rec["peak_gain"]) if "peak_gain" in rec.files else 0.0)
    pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, B, int(rec["iseed"]))
    pm = O.normalize(O.apply_mask_rgb(pal, pm_cls))
    lab = O.normalize(O.apply_mask_rgb(pal, lb_cls))
    yes = (lb_cls != 0)[:, None]
    p = prm.to(DEV).requires_grad_(True)
    out = model(pixel_values=pix.to(DEV), labels=lab.to(DEV), prompt_pixel_values=p, prompt_masks=pm.to(DEV),
                embedding_type="instance")
    loss = ops.seggpt_loss(out.pred_masks, lab.to(DEV), yes.to(DEV), 0.01, "reference")
    loss.backward()
    pn = O.palette_norm(pal)
    masks = ops.decode_argmin(out.pred_masks.detach(), pn.to(DEV))
    torch.cuda.synchronize()
    return out.pred_masks.detach().cpu(), loss.item(), p.grad.cpu(), masks.cpu(), pn


def check_masks_by_margin(pred_ref_bottom, pn, masks, ref_masks, delta, max_mismatch, tag):
    """pred_ref_bottom (B,3,h,w): the reference prediction on the query half (or a strided slice of it); masks /
    ref_masks (B,h,w).  With |pred - pred_ref| <= delta everywhere, the gap between the squared distances to two palette
    colours moves by at most 2 * delta * ||p_j - p_k||_1, so pixels with a larger gap MUST decode identically."""
    x = torch.as_tensor(pred_ref_bottom).permute(0, 2, 3, 1)
    d = ((x[:, :, :, None, :] - pn[:, None, None, :, :]) ** 2).sum(-1)
    top2 = d.topk(2, dim=-1, largest=False).values
    l1 = (pn[:, :, None, :] - pn[:, None, :, :]).abs().sum(-1).amax(dim=(1, 2))  # (B,) worst palette pair
    bound = (2.0 * delta * l1)[:, None, None]
    safe = (top2[..., 1] - top2[..., 0]) > bound
    masks, ref_masks = torch.as_tensor(masks).long(), torch.as_tensor(ref_masks).long()
    bad = masks != ref_masks
    print(f"[measured] {tag}: {int(bad.sum())} of {bad.numel()} mask pixels differ; {float(safe.float().mean()) * 100:.2f} % of the "
          f"pixels are outside the margin bound {float(bound.max()):.3f}")
    assert not bool((bad & safe).any()), "a pixel outside the proven margin decoded differently"
    assert int(bad.sum()) <= max_mismatch, f"{int(bad.sum())} mask pixels differ (allowed {max_mismatch})"


# f32: the north_star bar.  bf16: ~1.5 x the measured error of each geometry (see the module docstring), `mm` = allowed
# number of differing mask pixels (of the pixels the fixture holds: all for tiny, every 8th row / column otherwise).
TOL = {
    torch.float32: {**{k: dict(t=1e-4, loss=1e-5, mm=0) for k in ("tiny", "tiny_dec128", "small", "small_peaked", "vit_large")},
                    # peaked ViT-L amplifies rounding ~1e3 x (see the 16-bit rows): two fp32 evaluation orders differ by 2.4e-5 / 9.1e-5
                    "vit_large_peaked": dict(t=1e-4, tg=3e-4, loss=1e-5, mm=0)},
    # float32 with every GEMM, attention and conv MFMA as three f16 MFMAs (22-bit operands; exact-f32 softmax / LayerNorm /
    # storage): bars 1e-4 like exact f32 (measured: see DESIGN.md section 2); NO mask pixel may differ (every fixture measures 0)
    F32X3: {**{k: dict(t=1e-4, loss=1e-5, mm=0) for k in ("tiny", "tiny_dec128", "small", "vit_large")},
            "small_peaked": dict(t=1e-4, tg=3e-4, loss=1e-5, mm=0),      # 1.4e-6 / 7.1e-5 (exact f32: 1.5e-6 / 3.1e-6)
            "vit_large_peaked": dict(t=2e-4, tg=6e-4, loss=1e-5, mm=0)},  # 2.4e-5 / 8.1e-5 (exact f32: 2.4e-5 / 9.1e-5)
    torch.float16: {   # IEEE half operands: ~8x less round-off than bf16; bars set from the measured values below
        "tiny": dict(t=1.2e-3, loss=1e-5, mm=4),          # measured pred 5.2e-4 / grad 7.4e-4, 1 of 8192 mask pixels
        "small": dict(t=1.2e-3, loss=1e-5, mm=2),         # 4.8e-4 / 6.7e-4, 0 of 6272
        "small_peaked": dict(t=2.5e-3, loss=1e-5, mm=2),  # 6.5e-4 / 1.63e-3, 0
        "vit_large": dict(t=2e-3, loss=1e-5, mm=2),       # 1.19e-3 / 1.32e-3 (rms-relative 0.9e-3 / 1.0e-3), 0 of 3136
        "tiny_dec128": dict(t=1.2e-3, tg=3.5e-3, loss=1e-5, mm=4),  # 6.8e-4 / 2.45e-3 (precision model of tools/rounding_budget.py: 5.0e-4 / 1.4e-3)
        # PEAKED attention 24 layers deep: measured 9.4e-3 / 3.2e-2 -- ten to thirty times north_star's 1e-3.  The precision model
        # (profiles/r3_rounding_budget_vit_large_peaked_f16.json) reproduces it (9.0e-3 / 3.1e-2) from operand rounding alone:
        # weights 37 %, LayerNorm outputs 27 %, q / k 24 % of the variance, all amplified through softmax rows whose logits
        # stand 28-43 above the row mean.  No 16-bit mode is inside the tolerance in this regime; float32 is (2.4e-5 / 9.1e-5).
        "vit_large_peaked": dict(t=1.5e-2, tg=5e-2, loss=1e-4, mm=32),
    },
    torch.bfloat16: {
        "tiny": dict(t=8e-3, loss=1e-3, mm=24),          # measured pred 4.1e-3 / grad 5.0e-3, 9 of 8192 mask pixels
        "small": dict(t=1e-2, loss=1e-3, mm=8),          # 5.0e-3 / 6.4e-3, 1 of 6272
        "small_peaked": dict(t=1.7e-2, loss=1e-3, mm=8),  # 5.4e-3 / 1.13e-2, 0 of 6272
        "vit_large": dict(t=1.6e-2, loss=1e-3, mm=8),    # 7.8e-3 / 1.03e-2, 0 of 3136
        "tiny_dec128": dict(t=8e-3, tg=1e-2, loss=1e-3, mm=24),          # 4.2e-3 / 6.7e-3, 0 of 8192
        "vit_large_peaked": dict(t=1.1e-1, tg=4e-1, loss=1e-3, mm=128),  # measured 7.2e-2 / 2.6e-1 (!): see the float16 comment
    },
}


DTYPES = [torch.float32, F32X3, torch.bfloat16, torch.float16]


@pytest.mark.parametrize("gname", ["tiny", "tiny_dec128"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_tiny_end_to_end_vs_reference_vectors(golden_dir, dtype, gname):
    """`tiny_dec128`: the same net with decoder_hidden_size = 128 (BASELINE config 5's decoder width: pixel-shuffle epilogue
    with 128-channel pixels, 3x3 conv 128 -> 128 in two K passes, LayerNorm(128) + GELU + 1x1 head, their backward), against
    the vector the HF module produced for that config (`HF:modeling_seggpt.py:525-579`)."""
    rec = np.load(golden_dir / f"{gname}_e2e.npz")
    pred, loss, grad, masks, pn = run_case(gname, rec, 2, dtype)
    tol = TOL[dtype][gname]
    e_pred, e_grad = relmax(pred, rec["pred"]), relmax(grad, rec["grad"])
    print(f"[measured] {gname} {dtype}: pred {e_pred:.2e} grad {e_grad:.2e} loss {abs(loss - float(rec['loss'])) / abs(float(rec['loss'])):.1e}")
    assert e_pred < tol["t"] and e_grad < tol.get("tg", tol["t"])
    assert abs(loss - float(rec["loss"])) < tol["loss"] * abs(float(rec["loss"]))
    ref_masks = torch.from_numpy(rec["masks"]).long()
    if dtype == torch.float32:
        assert torch.equal(masks, ref_masks)  # bit-exact
    else:
        H = pred.shape[2] // 2
        delta = tol["t"] * float(np.abs(rec["pred"]).max())
        check_masks_by_margin(rec["pred"][:, :, H:, :], pn, masks, ref_masks, delta, tol["mm"], f"tiny {dtype}")


def _sliced_case(golden_dir, fixture, gname, B, dtype):
    rec = np.load(golden_dir / fixture)
    key = gname + "_peaked" if "peaked" in fixture else gname
    pred, loss, grad, masks, pn = run_case(gname, rec, B, dtype)
    tol, st = TOL[dtype][key], int(rec["stride"])
    e_pred, e_grad = relmax(pred[:, :, ::st, ::st], rec["pred_slice"]), relmax(grad[:, :, ::st, ::st], rec["grad_slice"])
    e_pl2 = abs(float(pred.double().norm()) - float(rec["pred_l2"])) / float(rec["pred_l2"])
    e_gl2 = abs(float(grad.double().norm()) - float(rec["grad_l2"])) / float(rec["grad_l2"])
    print(f"[measured] {key} {dtype}: pred {e_pred:.2e} grad {e_grad:.2e} |pred|2 {e_pl2:.1e} |grad|2 {e_gl2:.1e} "
          f"loss {abs(loss - float(rec['loss'])) / abs(float(rec['loss'])):.1e}")
    assert e_pred < tol["t"] and e_grad < tol.get("tg", tol["t"])
    assert abs(loss - float(rec["loss"])) < tol["loss"] * abs(float(rec["loss"]))
    assert e_pl2 < tol["t"] and e_gl2 < 2 * tol.get("tg", tol["t"])
    m8 = masks.numpy().astype(np.uint8)
    if dtype == torch.float32:
        assert np.array_equal(m8[:, ::st, ::st], rec["masks_slice"])
        assert zlib.crc32(m8.tobytes()) == int(rec["masks_crc"])  # every pixel bit-exact
    else:
        H = pred.shape[2] // 2
        delta = tol["t"] * float(np.abs(rec["pred_slice"]).max())
        check_masks_by_margin(rec["pred_slice"][:, :, H // st:, :], pn, m8[:, ::st, ::st], rec["masks_slice"], delta, tol["mm"],
                              f"{key} {dtype}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_small_end_to_end_real_token_grid(golden_dir, dtype):
    """56 x 28 token grid (1568 tokens, the reference geometry: padded key slots, 13 query blocks, 28 key tiles)."""
    _sliced_case(golden_dir, "small_e2e.npz", "small", 2, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_small_peaked_attention_vs_reference_vectors(golden_dir, dtype):
    """The same grid with PEAKED attention (q / k / rel-pos x 8: logits 24-28 above the row mean, mean row-max
    probability ~0.4, as a trained checkpoint produces): drives the online-softmax rescale path and the exp2 range that
    the sigma = 0.02 fixtures (near-uniform rows) never reach.  Vector generated by the HF module itself."""
    _sliced_case(golden_dir, "small_peaked_e2e.npz", "small", 2, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_vit_large_vs_reference_vectors(golden_dir, dtype):
    """Full reference geometry (ViT-L, 24 layers, 370.7 M parameters), B=1."""
    _sliced_case(golden_dir, "vitl_e2e.npz", "vit_large", 1, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_vit_large_peaked_attention_vs_reference_vectors(golden_dir, dtype):
    """ViT-L with PEAKED attention, 24 layers deep (q / k / rel-pos x 4: the row-max logit clears the row mean by 28-43 in
    every layer, mean row-max probability ~0.5): the softmax regime of a trained checkpoint, which the sigma = 0.02 fixture
    (near-uniform rows) does not reach.  Vector generated by the HF module (`HF:modeling_seggpt.py:313-348`)."""
    _sliced_case(golden_dir, "vitl_peaked_e2e.npz", "vit_large", 1, dtype)


def test_against_oracle_semantic_embedding_and_no_labels():
    """Oracle comparison on a call pattern the fixtures do not hold: embedding_type='semantic', no labels."""
    g = SegGptGeometry.tiny()
    sd = synth_state_dict(g, seed=1)
    model = model_for("tiny", 1, torch.float32)
    pix, prm, pm_cls, _, pal = synth_inputs(g, 3, 11)
    pm = O.normalize(O.apply_mask_rgb(pal, pm_cls))
    ref = O.forward(sd, g, pix, prm, pm, embedding_type="semantic")
    with torch.no_grad():
        out = model(pixel_values=pix.to(DEV), prompt_pixel_values=prm.to(DEV), prompt_masks=pm.to(DEV),
                    embedding_type="semantic")
    assert relmax(out.pred_masks, ref) < 1e-4


def test_errors_match_the_hf_contract():
    g = SegGptGeometry.tiny()
    model = model_for("tiny", 1, torch.float32)
    x = torch.zeros(1, 3, 64, 64, device=DEV)
    with pytest.raises(ValueError, match="Embedding type should be either"):
        model(pixel_values=x, prompt_pixel_values=x, prompt_masks=x, embedding_type="panoptic")
    with pytest.raises(ValueError, match="doesn't match model"):
        bad = torch.zeros(1, 3, 32, 64, device=DEV)
        model(pixel_values=bad, prompt_pixel_values=bad, prompt_masks=bad)
    with pytest.raises(ValueError, match="channel dimension"):
        bad = torch.zeros(1, 4, 64, 64, device=DEV)
        model(pixel_values=bad, prompt_pixel_values=x, prompt_masks=x)


def test_two_live_forwards_then_two_backwards():
    """f1, f2, b1, b2 through the autograd boundary (two losses alive at once, different batch sizes): every backward
    must use the activations of ITS forward, as autograd through the HF module does."""
    g = SegGptGeometry.tiny()
    sd = synth_state_dict(g, seed=1)
    model = model_for("tiny", 1, torch.float32)

    def case(B, seed):
        pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, B, seed)
        pm = O.normalize(O.apply_mask_rgb(pal, pm_cls))
        lab = O.normalize(O.apply_mask_rgb(pal, lb_cls))
        return pix, prm, pm, lab, (lb_cls != 0)[:, None]

    cases = [case(2, 21), case(3, 22), case(2, 23)]
    refs = []
    for pix, prm, pm, lab, yes in cases:
        p = prm.clone().requires_grad_(True)
        loss = O.seggpt_loss(O.forward(sd, g, pix, p, pm), lab, yes, 0.01, "reference")
        refs.append(torch.autograd.grad(loss, p)[0])
    ps, losses = [], []
    for pix, prm, pm, lab, yes in cases:  # three forwards alive together (two share a batch size)
        p = prm.to(DEV).requires_grad_(True)
        out = model(pixel_values=pix.to(DEV), prompt_pixel_values=p, prompt_masks=pm.to(DEV))
        losses.append(ops.seggpt_loss(out.pred_masks, lab.to(DEV), yes.to(DEV), 0.01, "reference"))
        ps.append(p)
    assert len(model._leased) == 3
    for i in (0, 1, 2):
        losses[i].backward()
    for p, ref in zip(ps, refs):
        assert relmax(p.grad, ref) < 1e-4
    with torch.no_grad():  # an inference forward in between does not disturb a pending backward
        pix, prm, pm, lab, yes = cases[0]
        p = prm.to(DEV).requires_grad_(True)
        out = model(pixel_values=pix.to(DEV), prompt_pixel_values=p, prompt_masks=pm.to(DEV))
    p = prm.to(DEV).requires_grad_(True)
    out = model(pixel_values=pix.to(DEV), prompt_pixel_values=p, prompt_masks=pm.to(DEV))
    with torch.no_grad():
        model(pixel_values=pix.to(DEV), prompt_pixel_values=prm.to(DEV), prompt_masks=pm.to(DEV))
    ops.seggpt_loss(out.pred_masks, lab.to(DEV), yes.to(DEV), 0.01, "reference").backward()
    assert relmax(p.grad, refs[0]) < 1e-4
    del ps, losses, out, p
    import gc
    gc.collect()
    assert len(model._leased) == 0  # every workspace is given back once its graph is gone


# ------------------------------------------------------------------------------------ wrapper arithmetic
def test_loss_both_variants_vs_reference(golden_dir):
    rec = np.load(golden_dir / "wrapper.npz")
    pred = torch.from_numpy(rec["decode_pred"]).to(DEV)
    labels = torch.from_numpy(rec["loss_labels"]).to(DEV)
    yes = torch.from_numpy(rec["mask"] != 0).to(DEV)
    for beta in (0.01, 0.5):
        p = pred.clone().requires_grad_(True)
        l3 = ops.seggpt_loss(p, labels, yes, beta, "reference")  # B=3: the unsqueeze(1) broadcast (src/model.py:61)
        l3.backward()
        assert abs(l3.item() - float(rec[f"loss_B3_beta{beta}"])) < 2e-6 * abs(l3.item())
        np.testing.assert_allclose(p.grad.cpu().numpy(), rec[f"loss_B3_grad_beta{beta}"], rtol=2e-5, atol=1e-9)
        for i in range(3):
            l1 = ops.seggpt_loss(pred[i:i + 1], labels[i:i + 1], yes[i:i + 1], beta, "reference")
            assert abs(l1.item() - rec[f"loss_B1_beta{beta}"][i]) < 2e-6
        lp = ops.seggpt_loss(pred, labels, yes, beta, "per_sample")
        ref = O.seggpt_loss(pred.cpu(), labels.cpu(), yes.cpu(), beta, "per_sample")
        assert abs(lp.item() - ref.item()) < 2e-6 * abs(ref.item())
    # edge: nothing kept -> the reference divides 0/0
    l0 = ops.seggpt_loss(pred, labels, torch.zeros_like(yes), 0.01, "reference")
    assert torch.isnan(l0)


def test_mask_colourisation_kernel_bit_exact_and_loss_from_class_ids(golden_dir):
    """`torch_apply_mask_rgb` (`src/util/ml_util.py:114-132`) and `normalize` (`src/data.py:345`)
    as ONE HIP kernel, bit-exact against the reference's own output; and the loss taking class ids + normalised palette
    (`bsg_loss_fwd_bwd_ids`) bit-identical to the loss on the materialised label image (`src/model.py:238-239, 255`)."""
    from beach_seg_amd import ml_util

    rec = np.load(golden_dir / "wrapper.npz")
    pal, mask = torch.from_numpy(rec["rand_palette_seed42"]).to(DEV), torch.from_numpy(rec["mask"]).to(DEV)
    rgb = ml_util.torch_apply_mask_rgb(pal, mask)  # mean 0 / std 1 through the same kernel
    assert rgb.dtype == torch.float32 and np.array_equal(rgb.cpu().numpy(), rec["apply_mask_rgb"])
    assert np.array_equal(ml_util.torch_apply_mask_rgb(pal, mask[:, 0]).cpu().numpy(), rec["apply_mask_rgb"])  # (B,H,W) ids
    mean = np.array(ops.IMAGE_MEAN, np.float32).reshape(1, 3, 1, 1)
    std = np.array(ops.IMAGE_STD, np.float32).reshape(1, 3, 1, 1)
    want = (rec["apply_mask_rgb"] - mean) / std  # float32 numpy = the reference's CPU arithmetic
    got = ops.mask_rgb_norm(pal, mask)
    assert np.array_equal(got.cpu().numpy(), want)
    # ragged plane (h*w not a multiple of 4: scalar tail path) and K = 256
    g = torch.Generator().manual_seed(3)
    pal2 = torch.randint(0, 256, (2, 256, 3), generator=g, dtype=torch.uint8)
    ids2 = torch.randint(0, 256, (2, 7, 9), generator=g, dtype=torch.uint8)
    want2 = ((O.apply_mask_rgb(pal2, ids2).numpy() - mean) / std).astype(np.float32)
    assert np.array_equal(ops.mask_rgb_norm(pal2.to(DEV), ids2.to(DEV)).cpu().numpy(), want2)
    with pytest.raises(ValueError):
        ops.mask_rgb_norm(pal.float(), mask)
    # loss from class ids == loss on the materialised normalised label image, bit for bit (value and gradient)
    pred = torch.from_numpy(rec["decode_pred"]).to(DEV)
    pn = torch.from_numpy(rec["pal_norm"]).to(DEV)
    for variant in ("reference", "per_sample"):
        for beta in (0.01, 0.5):
            p1 = pred.clone().requires_grad_(True)
            l1 = ops.seggpt_loss(p1, got, mask != 0, beta, variant)
            l1.backward()
            p2 = pred.clone().requires_grad_(True)
            l2 = ops.seggpt_loss_ids(p2, mask, pn, beta, variant)
            l2.backward()
            assert torch.equal(l1, l2) and torch.equal(p1.grad, p2.grad), (variant, beta)
    ref = O.seggpt_loss(pred.cpu(), torch.from_numpy(want), torch.from_numpy(rec["mask"] != 0), 0.01, "reference")
    assert abs(l2.item() - ops.seggpt_loss_ids(pred, mask, pn, 0.5, "per_sample").item()) == 0
    assert abs(ops.seggpt_loss_ids(pred, mask, pn, 0.01, "reference").item() - ref.item()) < 2e-6 * abs(ref.item())


def test_decode_bit_exact_including_near_ties(golden_dir):
    rec = np.load(golden_dir / "wrapper.npz")
    pred = torch.from_numpy(rec["decode_pred"]).to(DEV)
    pn = torch.from_numpy(rec["pal_norm"]).to(DEV)
    m = ops.decode_argmin(pred, pn)
    assert m.dtype == torch.int64  # src/model.py:172
    assert np.array_equal(m.cpu().numpy().astype(np.uint8), rec["decode_masks"])
    assert np.array_equal(ops.decode_argmin(pred, pn, torch.uint8).cpu().numpy(), rec["decode_masks"])


def test_prompt_gather_scatter_and_adamw_vs_torch():
    torch.manual_seed(0)
    P, h, w, B = 5, 8, 16, 4
    params = torch.rand(P, 3, h, w)
    idx = torch.tensor([3, 0, 3, 4])
    mean = torch.tensor(O.IMAGE_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(O.IMAGE_STD).view(1, 3, 1, 1)
    # torch reference: list of Parameters, stack + Normalize, AdamW skips parameters without a gradient
    plist = [torch.nn.Parameter(params[i].clone()) for i in range(P)]
    opt = torch.optim.AdamW(plist, lr=1e-3)
    d_params = params.clone().to(DEV)
    m_, v_ = torch.zeros_like(d_params), torch.zeros_like(d_params)
    steps = [0] * P
    for it in range(3):
        gpix = torch.randn(B, 3, h, w)
        opt.zero_grad(set_to_none=True)
        x = (torch.stack([plist[i] for i in idx.tolist()]) - mean) / std
        (x * gpix).sum().backward()
        opt.step()
        got = ops.prompt_gather(d_params, idx.to(DEV))
        grads = torch.zeros_like(d_params)
        ops.prompt_grad_scatter(gpix.to(DEV), idx.to(DEV), grads)
        active = sorted(set(idx.tolist()))
        for a in active:
            steps[a] += 1
        ops.adamw_step(d_params.view(P, -1), grads.view(P, -1), m_.view(P, -1), v_.view(P, -1),
                       torch.tensor(active, device=DEV), [steps[a] for a in active], lr=1e-3)
        if it == 0:
            np.testing.assert_allclose(got.cpu().numpy(), ((params[idx] - mean) / std).numpy(), rtol=1e-6, atol=1e-6)
        ref = torch.stack([p.detach() for p in plist])
        np.testing.assert_allclose(d_params.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=2e-7)
        idx = torch.tensor([1, 1, 2, 0]) if it == 0 else torch.tensor([4, 3, 2, 2])


def test_predict_vote_glue_bit_exact(golden_dir):
    rec = np.load(golden_dir / "predict_glue.npz")
    counter = torch.zeros(40, 50, 4, dtype=torch.uint8, device=DEV)
    preds = torch.from_numpy(rec["preds"]).to(DEV)
    crops = torch.from_numpy(rec["crops"].astype(np.int32)).to(DEV)
    for i in range(len(crops)):  # overlapping crops go in separate launches (uint8 += is not atomic)
        ops.vote_paste(counter, preds[i:i + 1], crops[i:i + 1], 16)
    assert np.array_equal(counter.cpu().numpy(), rec["counter"])
    assert np.array_equal(ops.vote_argmax(counter).cpu().numpy(), rec["final"])
    # nearest resize 448 -> 112 fused into the paste (src/predict.py:259)
    big = (torch.arange(448 * 448, device=DEV).reshape(1, 448, 448) % 4).to(torch.uint8)
    c2 = torch.zeros(112, 112, 4, dtype=torch.uint8, device=DEV)
    ops.vote_paste(c2, big, torch.tensor([[0, 0, 112, 112]], dtype=torch.int32, device=DEV), 112)
    want = np.zeros((112, 112, 4), np.uint8)
    PO.accumulate(want, (0, 0, 112, 112), PO.one_hot(PO.nearest_resize(big[0].cpu().numpy().astype(np.int64), 112), 4))
    assert np.array_equal(c2.cpu().numpy(), want)
    # non-dyadic ratios: OpenCV's resizeNN index rule in DOUBLE (floor(dst * (1 / (dsize / ssize)))); a float32
    # evaluation differs on 7 rows / columns each for 448 -> 384 / 640 / 768
    for cs in (384, 640, 100):
        cN = torch.zeros(cs, cs, 4, dtype=torch.uint8, device=DEV)
        ops.vote_paste(cN, big, torch.tensor([[0, 0, cs, cs]], dtype=torch.int32, device=DEV), cs)
        wantN = np.zeros((cs, cs, 4), np.uint8)
        PO.accumulate(wantN, (0, 0, cs, cs), PO.one_hot(PO.nearest_resize(big[0].cpu().numpy().astype(np.int64), cs), 4))
        assert np.array_equal(cN.cpu().numpy(), wantN), cs
    # uint8 wrap at 256 votes
    c3 = torch.full((16, 16, 4), 255, dtype=torch.uint8, device=DEV)
    ops.vote_paste(c3, preds[:1], torch.tensor([[0, 0, 16, 16]], dtype=torch.int32, device=DEV), 16)
    assert int(c3.min()) == 0



def test_tile_front_end_bit_exact_vs_pillow(golden_dir):
    """bsg_tile_frontend: padded crop + Pillow-BICUBIC resize (u8, bit-exact vs PIL's own output) + /255 + Normalize
    (f32, bit-exact vs numpy float32 arithmetic), up-scaling 112 / 256 -> 448 and down-scaling 256 -> 96."""
    rec = np.load(golden_dir / "frontend_pil.npz")
    mosaic = torch.from_numpy(rec["mosaic"]).to(DEV)
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    for key, boxes, crop, S in (("out112", rec["boxes112"], 112, 448), ("out256", rec["boxes256"], 256, 448),
                                ("down256_96", rec["boxes256"], 256, 96)):
        out, u8 = ops.tile_frontend(mosaic, torch.from_numpy(boxes).to(DEV), crop, S, return_u8=True)
        assert np.array_equal(u8.cpu().numpy(), rec[key]), key
        want = ((rec[key].astype(np.float32) / np.float32(255.0) - mean) / std).transpose(0, 3, 1, 2)
        assert np.array_equal(out.cpu().numpy(), want), key
    with pytest.raises(ValueError):
        ops.tile_frontend(mosaic.float(), torch.from_numpy(rec["boxes112"]).to(DEV), 112, 448)



@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_results_do_not_depend_on_workspace_contents(golden_dir, dtype):
    """The ABI asks for a zeroed workspace; this checks how much actually depends on it: with every workspace byte set
    to 0xFF (NaN patterns in every dtype) before the call, forward and backward must give bit-identical results, on a
    grid whose token count is not a multiple of the dK/dV query tile (tail columns exist)."""
    rec = np.load(golden_dir / "tiny_e2e.npz")
    clean = run_case("tiny", rec, 2, dtype)
    model = model_for("tiny", int(rec["wseed"]), dtype)
    for key in list(model._ws):
        model._ws[key].fill_(0xFF)
    dirty = run_case("tiny", rec, 2, dtype)
    assert torch.equal(clean[0], dirty[0]) and clean[1] == dirty[1]
    assert torch.equal(clean[2], dirty[2]), float((clean[2] - dirty[2]).abs().max())



def test_hf_post_process_decode_bit_exact(golden_dir):
    """bsg_decode_hf / ops.post_process_semantic_segmentation against the transformers post-processor's own output:
    un-normalise, clip, palette arg-min at full size, with target sizes, and after the mean over prompts."""
    rec = np.load(golden_dir / "hf_postprocess.npz")
    pred, nl = torch.from_numpy(rec["pred"]).to(DEV), int(rec["num_labels"])
    full = ops.post_process_semantic_segmentation(pred, nl)
    assert np.array_equal(torch.stack(full).cpu().numpy(), rec["full"])
    small = ops.post_process_semantic_segmentation(pred, nl, [(12, 10)] * pred.shape[0])
    assert np.array_equal(torch.stack(small).cpu().numpy(), rec["small"])
    ens = ops.post_process_semantic_segmentation(pred.mean(dim=0).unsqueeze(0), nl, [(16, 16)])
    assert np.array_equal(torch.stack(ens).cpu().numpy(), rec["ens"])
    with pytest.raises(ValueError):
        ops.post_process_semantic_segmentation(pred, nl, [(12, 10)])


# ------------------------------------------------------------------- full-size, size-independent properties
def test_full_geometry_batch_invariance_and_linearity():
    """At the BASELINE geometry (ViT-L, bf16): a sample's prediction does not depend on its batch (bit-exact),
    and the dgrad is linear: backward(2 g) == 2 backward(g) bit-exactly (power-of-two scaling commutes with
    bf16/fp32 rounding)."""
    model = model_for("vit_large", 0, torch.bfloat16)
    g = SegGptGeometry.vit_large()
    B = 16
    gen = torch.Generator(device=DEV).manual_seed(7)
    mk = lambda: torch.randn(B, 3, 448, 448, device=DEV, generator=gen)
    pix, prm, pm = mk(), mk(), mk()
    with torch.no_grad():
        full = model(pixel_values=pix, prompt_pixel_values=prm, prompt_masks=pm).pred_masks
        for i in (0, B - 1):
            one = model(pixel_values=pix[i:i + 1], prompt_pixel_values=prm[i:i + 1], prompt_masks=pm[i:i + 1]).pred_masks
            assert torch.equal(one[0], full[i])
    assert torch.isfinite(full).all()
    p = prm[:2].clone().requires_grad_(True)
    out = model(pixel_values=pix[:2], prompt_pixel_values=p, prompt_masks=pm[:2]).pred_masks
    gp = torch.randn(out.shape, device=DEV, generator=gen) * 1e-4
    (g1,) = torch.autograd.grad(out, p, gp, retain_graph=True)
    (g2,) = torch.autograd.grad(out, p, 2 * gp)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert torch.equal(g2, 2 * g1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_feature_ensemble_vs_reference_vector(golden_dir, dtype):
    """`feature_ensemble=True` with K=3 prompts for one query (HF:414-423), against the
    vector HF itself produced."""
    rec = np.load(golden_dir / "tiny_feature_ensemble.npz")
    model = model_for("tiny", int(rec["wseed"]), dtype)
    with torch.no_grad():
        out = model(pixel_values=torch.from_numpy(rec["pixel_values"]).to(DEV),
                    prompt_pixel_values=torch.from_numpy(rec["prompt_pixel_values"]).to(DEV),
                    prompt_masks=torch.from_numpy(rec["prompt_masks"]).to(DEV), feature_ensemble=True)
    assert relmax(out.pred_masks, rec["pred"]) < TOL[dtype]["tiny"]["t"]
    if dtype == torch.float32:  # the whole few-shot crop step of src/predict_no_prompt.py:283-301 against the oracle's decode
        from beach_seg_amd.predict import ensemble_predict
        from oracle import predict_oracle as PO

        got = ensemble_predict(model, torch.from_numpy(rec["pixel_values"]).to(DEV),
                               torch.from_numpy(rec["prompt_pixel_values"]).to(DEV),
                               torch.from_numpy(rec["prompt_masks"]).to(DEV), num_classes=4, crop_size=16)
        want = PO.hf_post_process(torch.from_numpy(rec["pred"]).mean(0, keepdim=True), 3, (16, 16))[0]
        assert (got.cpu() == want).float().mean().item() > 0.99  # pred differs from HF's by ~1e-6: near-ties may flip


@pytest.mark.parametrize("dtype", [torch.float32, F32X3, torch.bfloat16, torch.float16])
def test_wide_grid_and_wide_encoder_vs_oracle(dtype):
    """BASELINE config 5 geometry (`SegGptGeometry.config5`): 64 x 32 token grid (1024 x 512 canvas: no padded key slots, Hp =
    64), a 2048-wide encoder with 32 heads and the 128-channel decoder; 2 layers so the CPU oracle finishes in seconds.  No reference checkpoint of this
    shape exists, so parity is against the oracle (itself pinned on the reference geometry).  bf16 is the dtype
    config 5 names (Hp = 64 with no padded key slots in the transposing-read attention path)."""
    import dataclasses

    from beach_seg_amd.seggpt import SegGptNative

    g = dataclasses.replace(SegGptGeometry.config5(), mlp_dim=4096, num_hidden_layers=2, merge_index=0,
                            intermediate_hidden_state_indices=(0, 1))
    assert (g.hidden_size, g.num_attention_heads, g.image_size, g.decoder_hidden_size) == (2048, 32, (1024, 512), 128)
    _models.clear()
    sd = synth_state_dict(g, seed=5)
    model = (SegGptNative(sd, g, device=DEV, dtype=torch.float32, gemm_x3=True) if dtype == F32X3 else
             SegGptNative(sd, g, device=DEV, dtype=dtype))
    pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, 1, 9)
    pm = O.normalize(O.apply_mask_rgb(pal, pm_cls))
    lab = O.normalize(O.apply_mask_rgb(pal, lb_cls))
    yes = (lb_cls != 0)[:, None]
    p_ref = prm.clone().requires_grad_(True)
    pred_ref = O.forward(sd, g, pix, p_ref, pm)
    loss_ref = O.seggpt_loss(pred_ref, lab, yes, 0.01, "reference")
    (grad_ref,) = torch.autograd.grad(loss_ref, p_ref)
    p = prm.to(DEV).requires_grad_(True)
    out = model(pixel_values=pix.to(DEV), prompt_pixel_values=p, prompt_masks=pm.to(DEV))
    loss = ops.seggpt_loss(out.pred_masks, lab.to(DEV), yes.to(DEV), 0.01, "reference")
    loss.backward()
    # bf16, 2 wide layers: measured 7.3e-3 / 7.3e-3 (decoder 64); f16 ~8x below
    t, tl = {torch.float32: (1e-4, 1e-5), F32X3: (1e-4, 1e-5), torch.bfloat16: (1.2e-2, 1e-3), torch.float16: (2e-3, 1e-5)}[dtype]
    print(f"[measured] config-5 class {dtype}: pred {relmax(out.pred_masks, pred_ref):.2e} grad {relmax(p.grad, grad_ref):.2e}")
    assert relmax(out.pred_masks, pred_ref) < t
    assert abs(loss.item() - loss_ref.item()) < tl * abs(loss_ref.item())
    assert relmax(p.grad, grad_ref) < t
    del model
