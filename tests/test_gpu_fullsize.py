"""Full-size parity (`-m gpu`): the configurations `bench.py` / `bench_predict.py` actually run, checked at their own
sizes through size-independent properties and against the oracle pipeline.

* BASELINE configs[1] (ViT-L, bf16, B=64 train step): the exact call sequence of `PromptTrainEngine.step` --
  `bsg_forward[_rows](save)` -> `bsg_loss_fwd_bwd` -> `bsg_backward_rows(first_row = H/2)` -- at B=64, where the persistent
  GEMM grid, the tail-split 128^2 launches and the row-windowed decoder backward are live.  A sample's prediction and
  prompt gradient must be BIT-identical to what the same sample gives alone (B=1), and the row-windowed backward must
  equal the plain one.
* BASELINE configs[3] (sliding-window predict): front-end -> hipGraph forward -> decode -> vote at ViT-L (f32 mode)
  against the oracle pipeline `frontend_oracle -> O.forward -> O.decode_argmin -> predict_oracle votes`.
"""
import numpy as np
import pytest
import torch

from beach_seg_amd import ml_util, ops
from beach_seg_amd.weights import SegGptGeometry, synth_state_dict
from oracle import frontend_oracle as FO
from oracle import predict_oracle as PO
from oracle import seggpt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _vitl(dtype):
    from test_gpu_parity import model_for

    return model_for("vit_large", 0, dtype)


def test_vit_large_b64_engine_call_sequence_matches_b1_bit_for_bit():
    model = _vitl(torch.bfloat16)
    g = SegGptGeometry.vit_large()
    B, Hh, W = 64, 448, 448
    gen = torch.Generator(device=DEV).manual_seed(7)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    pix, prm, pmask, label = rn(B, 3, Hh, W), rn(B, 3, Hh, W), rn(B, 3, Hh, W), rn(B, 3, Hh, W)
    yes = torch.ones(B, 1, Hh, W, dtype=torch.bool, device=DEV)
    # per-sample grad_pred (zero on the prompt half, as the reference loss produces it) so that the B x B coupling of
    # the reference loss does not enter the comparison
    gpred = torch.zeros(B, 3, 2 * Hh, W, device=DEV)
    gpred[:, :, Hh:, :] = rn(B, 3, Hh, W) * 1e-5
    pred = model._run_forward(pix, prm, pmask, 0, train=True)                      # bsg_forward(save_for_backward=1)
    loss, g_loss = ops.loss_fwd_bwd(pred, label, yes, 0.01, "reference", True)     # bsg_loss_fwd_bwd
    assert torch.isfinite(loss) and torch.isfinite(pred).all()
    assert float(g_loss[:, :, :Hh].abs().max()) == 0.0                             # the premise of first_row = H/2
    g_rows = model._run_backward(gpred, B, first_row=Hh)                           # bsg_backward_rows(first_row=448)
    g_full = model._run_backward(gpred, B, first_row=0)                            # bsg_backward on the same workspace
    assert torch.isfinite(g_rows).all() and float(g_rows.abs().max()) > 0
    assert torch.equal(g_rows, g_full), float((g_rows - g_full).abs().max())
    g_real = model._run_backward(g_loss, B, first_row=Hh)                          # the engine's own gradient: finite, non-zero
    assert torch.isfinite(g_real).all() and float(g_real.abs().max()) > 0
    # bsg_forward_rows(first_row = H/2), what the engine calls: the decoder over the query half (+ what the backward reads back);
    # the rows it writes hold the bits of the full forward, and the backward that follows gives the same gradient
    pred_w = model._run_forward(pix, prm, pmask, 0, train=True, first_row=Hh)
    assert torch.equal(pred_w[:, :, Hh:], pred[:, :, Hh:])
    assert float(pred_w[:, :, :Hh - 32].abs().max()) == 0.0                        # rows above the first computed 16-row tile: not written
    g_w = model._run_backward(gpred, B, first_row=Hh)
    assert torch.equal(g_w, g_rows), float((g_w - g_rows).abs().max())
    p1w = model._run_forward(pix[63:64], prm[63:64], pmask[63:64], 0, train=True, first_row=Hh)  # B = 1: the eight-wave GEMM with the row-window map
    assert torch.equal(p1w[0], pred_w[63])
    assert torch.equal(model._run_backward(gpred[63:64], 1, first_row=Hh)[0], g_rows[63])
    for i in (0, 37, 63):
        p1 = model._run_forward(pix[i:i + 1], prm[i:i + 1], pmask[i:i + 1], 0, train=True)
        assert torch.equal(p1[0], pred[i]), f"sample {i}: prediction depends on the batch"
        g1 = model._run_backward(gpred[i:i + 1], 1, first_row=Hh)
        assert torch.equal(g1[0], g_rows[i]), f"sample {i}: max diff {float((g1[0] - g_rows[i]).abs().max()):.3e}"


def test_vit_large_engine_step_updates_only_touched_prompts():
    """Two full `PromptTrainEngine.step`s at B=64 / P=8 (every prompt drawn 8 times: the scatter accumulates): the loss is
    finite, untouched state stays zero, and the step equals the hand-called sequence + `torch.optim.AdamW`."""
    from beach_seg_amd.engine import PromptTrainEngine

    model = _vitl(torch.bfloat16)
    B, P, Hh, W = 64, 8, 448, 448
    gen = torch.Generator(device=DEV).manual_seed(9)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    pix, pmask, label = rn(B, 3, Hh, W), rn(B, 3, Hh, W), rn(B, 3, Hh, W)
    yes = torch.ones(B, 1, Hh, W, dtype=torch.bool, device=DEV)
    P0 = torch.rand(P, 3, Hh, W, device=DEV, generator=gen)
    idx = torch.arange(B, device=DEV) % 4  # prompts 4..7 are never drawn
    eng = PromptTrainEngine(model, P0, lr=1e-3)
    l0 = eng.step(pix, label, yes, idx, pmask)
    # the same step by hand: gather -> forward -> loss -> backward_rows -> scatter -> torch AdamW on a ParameterList
    plist = [torch.nn.Parameter(P0[i].clone()) for i in range(P)]
    opt = torch.optim.AdamW(plist, lr=1e-3)
    prompts = ops.prompt_gather(P0, idx)
    pred = model._run_forward(pix, prompts, pmask, 0, train=True)
    loss, gp = ops.loss_fwd_bwd(pred, label, yes, 0.01, "reference", True)
    gpix = model._run_backward(gp, B, first_row=Hh)
    grads = torch.zeros_like(P0)
    ops.prompt_grad_scatter(gpix, idx, grads)
    for i in range(4):
        plist[i].grad = grads[i].clone()
    opt.step()
    assert float(l0) == float(loss)
    for i in range(P):
        assert torch.allclose(eng.params[i], plist[i].detach(), rtol=1e-5, atol=1e-6), i
    assert torch.equal(eng.params[4:], P0[4:]) and int(eng.steps[4:].sum()) == 0 and int(eng.steps[:4].min()) == 1
    l1 = eng.step(pix, label, yes, idx, pmask)
    assert torch.isfinite(l1) and float(l1) != float(l0)


def test_vit_large_predict_composition_vs_oracle_pipeline():
    """`bench_predict.py`'s composition on a small mosaic whose last window row / column sticks out (clipped paste):
    u8 mosaic -> device front-end (padded crop + PIL-BICUBIC + Normalize) -> hipGraph-replayed ViT-L forward (f32 mode) ->
    palette arg-min -> nearest 448->112 + one-hot vote -> arg-max, against the same pipeline on the CPU oracle."""
    from beach_seg_amd.predict import Accumulator, grid_crops

    geo = SegGptGeometry.vit_large()
    model = _vitl(torch.float32)
    mh, mw, crop, S, P = 200, 190, 112, 448, 2
    gen = torch.Generator().manual_seed(11)
    mosaic = (torch.rand(mh // 8, mw // 8 + 1, 3, generator=gen).repeat_interleave(8, 0).repeat_interleave(8, 1)[:mh, :mw] * 255).to(torch.uint8).contiguous()
    prompts = torch.rand(P, 3, S, S, generator=gen)
    pcls = torch.randint(0, 4, (P, S // 16, S // 16), generator=gen, dtype=torch.uint8).repeat_interleave(16, 1).repeat_interleave(16, 2)
    pal = torch.randint(0, 256, (4, 4, 3), generator=gen, dtype=torch.uint8)
    pal[:, 0] = 0
    crops = grid_crops(mh, mw, crop)
    n = crops.shape[0]
    assert n == 4
    idx = torch.arange(n) % P
    pn = O.palette_norm(pal)
    # ---- device pipeline
    img = ops.tile_frontend(mosaic.to(DEV), crops.to(DEV), crop, S)
    prm = ml_util.normalize(prompts[idx]).to(DEV)
    pmask = ml_util.normalize(ml_util.torch_apply_mask_rgb(pal.to(DEV), pcls[idx].to(DEV)))
    graphed = model.capture_forward(n)
    pred = graphed(img, prm, pmask)
    with torch.no_grad():
        eager = model(pixel_values=img, prompt_pixel_values=prm, prompt_masks=pmask).pred_masks
    assert torch.equal(pred, eager)
    masks = ops.decode_argmin(pred, pn.to(DEV), torch.uint8)
    acc = Accumulator((mh, mw), ("nodata", "sand", "water", "veg"), DEV)
    acc.update("d0", crops.to(DEV), masks, crop, disjoint=True)
    got = acc.result().cpu().numpy()
    # ---- oracle pipeline (CPU)
    sd = synth_state_dict(geo, seed=0)
    _, f = FO.tile_frontend(mosaic.numpy(), crops.numpy(), crop, S)
    assert np.array_equal(img.cpu().numpy(), f)
    o_prm = O.normalize(prompts[idx])
    o_pm = O.normalize(O.apply_mask_rgb(pal, pcls[idx]))
    with torch.no_grad():
        o_pred = O.forward(sd, geo, torch.from_numpy(f), o_prm, o_pm)
    o_masks = O.decode_argmin(o_pred, pn).numpy()
    counter = np.zeros((mh, mw, 4), np.uint8)
    for j in range(n):
        PO.accumulate(counter, tuple(int(v) for v in crops[j]), PO.one_hot(PO.nearest_resize(o_masks[j], crop), 4))
    want = PO.vote_argmax(counter).astype(np.uint8)
    err = float((pred.cpu() - o_pred).abs().max() / o_pred.abs().max())
    nbad = int((got != want).sum())
    print(f"[measured] predict composition ViT-L f32: pred rel err {err:.2e}, {nbad} of {want.size} final classes differ")
    assert err < 1e-4
    assert np.array_equal(got, want)  # f32 mode: bit-exact final classes


def test_config5_full_depth_batch_invariance_and_prompt_gradient():
    """BASELINE configs[4] at its OWN size (`SegGptGeometry.config5()`: canvas 1024 x 512, hidden 2048, 32 heads, mlp 8192,
    decoder 128, all 24 layers; bf16, B = 8, one GPU): the train-step call sequence of the engine.  Samples 0 and 7 must give
    bit-identical predictions and prompt gradients alone (B = 1) and in the batch, the gradient must be finite and non-zero,
    and the dgrad must be linear (power-of-two scaling commutes with every rounding of the chain).  Parity against the oracle
    is pinned at two layers (`test_wide_grid_and_wide_encoder_vs_oracle`); the full depth is too slow for the CPU."""
    from beach_seg_amd.seggpt import SegGptNative
    from test_gpu_parity import _models

    _models.clear()  # free the resident ViT-L of the tests above (one model at a time)
    torch.cuda.empty_cache()
    g = SegGptGeometry.config5()
    assert g.num_hidden_layers == 24 and g.hidden_size == 2048 and g.decoder_hidden_size == 128 and g.mlp_dim == 8192
    model = SegGptNative(synth_state_dict(g, seed=0, device=DEV), g, device=DEV, dtype=torch.bfloat16)
    B, Hh, W = 8, g.image_size[0] // 2, g.image_size[1]
    gen = torch.Generator(device=DEV).manual_seed(17)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    pix, prm, pmask = rn(B, 3, Hh, W), rn(B, 3, Hh, W), rn(B, 3, Hh, W)
    gpred = torch.zeros(B, 3, 2 * Hh, W, device=DEV)
    gpred[:, :, Hh:, :] = rn(B, 3, Hh, W) * 1e-5
    pred = model._run_forward(pix, prm, pmask, 0, train=True)
    assert torch.isfinite(pred).all() and float(pred.std()) > 0
    g_b = model._run_backward(gpred, B, first_row=Hh)
    assert torch.isfinite(g_b).all() and float(g_b.abs().max()) > 0
    g_2 = model._run_backward(2 * gpred, B, first_row=Hh)
    assert torch.equal(g_2, 2 * g_b)
    for i in (0, 7):
        p1 = model._run_forward(pix[i:i + 1], prm[i:i + 1], pmask[i:i + 1], 0, train=True)
        assert torch.equal(p1[0], pred[i]), f"sample {i}: prediction depends on the batch"
        g1 = model._run_backward(gpred[i:i + 1], 1, first_row=Hh)
        assert torch.equal(g1[0], g_b[i]), f"sample {i}: max diff {float((g1[0] - g_b[i]).abs().max()):.3e}"
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("gname,B", [("tiny", 3), ("vit_large", 8)])
def test_inference_forward_equals_train_forward_bit_for_bit(gname, B):
    """The inference forward runs fc1 with the gelu-only epilogue (`EPI_BIAS_GELU_FWD`: no derivative formed), the train forward
    with the one that also saves gelu' for the dgrad: the activations -- hence the predictions -- must be the same bits (tiny:
    the eight-wave / 128^2 GEMM path, ViT-L at B = 8: `gemm_nt_kernel_v5`)."""
    from test_gpu_parity import geometry_of, model_for

    model = model_for(gname, 0, torch.bfloat16)
    g = geometry_of(gname)
    Hh, W = g.image_size[0] // 2, g.image_size[1]
    gen = torch.Generator(device=DEV).manual_seed(23)
    pix, prm, pm = (torch.randn(B, 3, Hh, W, device=DEV, generator=gen) for _ in range(3))
    p_train = model._run_forward(pix, prm, pm, 0, train=True).clone()
    p_infer = model._run_forward(pix, prm, pm, 0, train=False)
    assert torch.isfinite(p_infer).all() and torch.equal(p_train, p_infer)
