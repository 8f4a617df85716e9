"""GPU tests of the host mirror: `PromptModel` steps (autograd path, torch.optim.AdamW over a ParameterList like
the reference) against the fused `PromptTrainEngine` step (HIP gather / loss / scatter / AdamW), and the predict
loop's device-side vote mosaic against the numpy restatement of `Accumulator`."""
import numpy as np
import pytest
import torch

from beach_seg_amd import ml_util, ops
from beach_seg_amd.config import BeachSegConfig
from beach_seg_amd.engine import PromptTrainEngine
from beach_seg_amd.model import PromptModel
from beach_seg_amd.predict import predict_mosaic
from beach_seg_amd.seggpt import SegGptNative
from beach_seg_amd.weights import SegGptGeometry, synth_state_dict
from oracle import predict_oracle as PO
from oracle import seggpt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _prompts(P, S, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [{"crop_idx": i, "date": "d", "image": torch.rand(3, S, S, generator=g).numpy(),
             "mask": torch.randint(0, 4, (S, S), generator=g, dtype=torch.uint8).numpy(),
             "nodata": np.zeros((S, S), bool)} for i in range(P)]


def test_training_step_autograd_path_equals_fused_engine_and_oracle():
    geo = SegGptGeometry.tiny()
    sd = synth_state_dict(geo, seed=1)
    net = SegGptNative(sd, geo, device=DEV, dtype=torch.float32)
    conf = BeachSegConfig(batch_size=2, checkpoint="synthetic:tiny", precision="32-true", inpt_size=64, crop_size=64)
    pm = PromptModel(conf, model=net)
    prompts = _prompts(3, 64)
    pm.create_trainable_params(prompts)
    opt = pm.configure_optimizers()["optimizer"]
    g = torch.Generator().manual_seed(5)
    batch = {"image": ml_util.normalize(torch.rand(2, 3, 64, 64, generator=g)).to(DEV),
             "mask": torch.randint(0, 4, (2, 1, 64, 64), generator=g, dtype=torch.uint8).to(DEV),
             "crop_idx": torch.tensor([0, 1])}
    # replay the RNG draws of training_step to feed the same palette / prompt choice to the other two paths
    pal_state, idx_state = pm.palette_g.get_state(), pm.g.get_state()
    loss = pm.fit_step(batch, opt)
    pm.palette_g.set_state(pal_state); pm.g.set_state(idx_state)
    pal, _ = pm.create_palette(2, train=True)
    idx = torch.randint(0, 3, (2,), generator=pm.g)
    label_color = ml_util.normalize(ml_util.torch_apply_mask_rgb(pal, batch["mask"]))
    pmask = torch.stack([torch.as_tensor(prompts[i]["mask"]) for i in idx.tolist()]).to(DEV)
    pmask_color = ml_util.normalize(ml_util.torch_apply_mask_rgb(pal, pmask))
    P0 = torch.stack([torch.as_tensor(p["image"]) for p in prompts])
    eng = PromptTrainEngine(net, P0, lr=opt.param_groups[0]["lr"], loss_beta=conf.loss_beta)
    loss2 = eng.step(batch["image"], label_color, batch["mask"] != 0, idx.to(DEV), pmask_color)
    torch.cuda.synchronize()
    assert abs(loss.item() - loss2.item()) < 1e-6 * abs(loss.item())
    got = torch.stack([p.detach() for p in pm.prompt_params_list])
    np.testing.assert_allclose(eng.params.cpu().numpy(), got.cpu().numpy(), rtol=1e-5, atol=1e-6)
    # and both equal the CPU oracle's loss on the same inputs
    xp = ml_util.normalize(P0[idx])
    pred = O.forward(sd, geo, batch["image"].cpu(), xp, pmask_color.cpu())
    lref = O.seggpt_loss(pred, label_color.cpu(), (batch["mask"] != 0).cpu(), conf.loss_beta, "reference")
    assert abs(loss.item() - lref.item()) < 1e-4 * abs(lref.item())
    # untouched prompt: no weight decay, no step (torch skips Parameters without grad)
    untouched = [i for i in range(3) if i not in idx.tolist()]
    for i in untouched:
        assert torch.equal(eng.params[i].cpu(), P0[i]) and int(eng.steps[i]) == 0


def test_validation_after_training_sees_the_updated_prompts():
    """`prepare_prompt`'s no-grad path gathers from a cached stack of the prompt Parameters; the cache must follow
    in-place updates (`optimizer.step()` keeps the list and the Parameter objects): after `fit_step`, `validation_step` /
    `forward` must run on the NEW prompt pixels, as the reference's `torch.stack` of the live Parameters does
    (`src/model.py:197`)."""
    geo = SegGptGeometry.tiny()
    net = SegGptNative(synth_state_dict(geo, seed=1), geo, device=DEV, dtype=torch.float32)
    conf = BeachSegConfig(batch_size=2, checkpoint="synthetic:tiny", precision="32-true", inpt_size=64, crop_size=64, lr=0.05)
    pm = PromptModel(conf, model=net)
    pm.create_trainable_params(_prompts(2, 64))
    opt = pm.configure_optimizers()["optimizer"]
    g = torch.Generator().manual_seed(5)
    batch = {"image": ml_util.normalize(torch.rand(2, 3, 64, 64, generator=g)).to(DEV),
             "mask": torch.randint(1, 4, (2, 1, 64, 64), generator=g, dtype=torch.uint8).to(DEV),
             "crop_idx": torch.tensor([0, 1])}
    pal, _ = pm.create_palette(2, train=False)
    with torch.no_grad():
        before, _ = pm.prepare_prompt(batch["crop_idx"], pal, train=False)  # fills the cache
    pm.fit_step(batch, opt)
    pm.fit_step(batch, opt)
    live = ml_util.normalize(torch.stack([p.detach() for p in pm.prompt_params_list]))
    with torch.no_grad():
        after, _ = pm.prepare_prompt(batch["crop_idx"], pal, train=False)
        after_dev, _ = pm.prepare_prompt(batch["crop_idx"].to(DEV), pal, train=False)  # the predict loop's device-index form
    assert not torch.equal(before["image"], after["image"])
    assert torch.equal(after["image"], live) and torch.equal(after_dev["image"], live)
    # an edit the version counters cannot see (.data assignment) needs the explicit hook
    pm.prompt_params_list[0].data = torch.zeros_like(pm.prompt_params_list[0].data)
    pm.invalidate_prompt_cache()
    with torch.no_grad():
        z, _ = pm.prepare_prompt(torch.tensor([0]), pal[:1], train=False)
    assert torch.equal(z["image"], ml_util.normalize(torch.zeros(1, 3, 64, 64, device=DEV)))
    # and a validation loss computed after training differs from the one before it
    pm2 = PromptModel(conf, model=net)
    pm2.create_trainable_params(_prompts(2, 64))
    l0 = pm2.validation_step(batch)
    pm2.palette_g.manual_seed(conf.seed + 1)
    opt2 = pm2.configure_optimizers()["optimizer"]
    pm2.fit_step(batch, opt2)
    pm2.palette_g.manual_seed(conf.seed + 1)
    l1 = pm2.validation_step(batch)
    assert torch.isfinite(l0) and torch.isfinite(l1) and l0.item() != l1.item()


def test_engine_step_from_class_ids_equals_step_on_materialised_images():
    """`PromptTrainEngine.step_ids` (colourisation kernel + id-based loss: SURVEY 8(a) row 4 fused into the step) against
    `step` on the explicitly colourised + normalised images: bit-identical loss and parameters."""
    geo = SegGptGeometry.tiny()
    net = SegGptNative(synth_state_dict(geo, seed=1), geo, device=DEV, dtype=torch.float32)
    g = torch.Generator().manual_seed(8)
    P0 = torch.rand(3, 3, 64, 64, generator=g)
    pix = ml_util.normalize(torch.rand(2, 3, 64, 64, generator=g)).to(DEV)
    lab_ids = torch.randint(0, 4, (2, 1, 64, 64), generator=g, dtype=torch.uint8).to(DEV)
    pm_ids = torch.randint(0, 4, (2, 64, 64), generator=g, dtype=torch.uint8).to(DEV)
    pal = torch.randint(0, 256, (2, 4, 3), generator=g, dtype=torch.uint8)
    pal[:, 0] = 0
    pn = PromptModel._palette_norm(pal).to(DEV)
    pal = pal.to(DEV)
    idx = torch.tensor([2, 0], device=DEV)
    e1 = PromptTrainEngine(net, P0, lr=1e-2)
    e2 = PromptTrainEngine(net, P0, lr=1e-2)
    for _ in range(2):
        l1 = e1.step(pix, ops.mask_rgb_norm(pal, lab_ids), lab_ids != 0, idx, ops.mask_rgb_norm(pal, pm_ids))
        l2 = e2.step_ids(pix, lab_ids, pal, pn, idx, pm_ids)
    torch.cuda.synchronize()
    assert torch.equal(l1, l2) and torch.equal(e1.params, e2.params)


def test_f16_overflow_guard_skips_the_step_and_backs_off():
    """The f16 dgrad chain runs on S * gradient in IEEE half; a non-finite prompt gradient must not reach AdamW.  Engine:
    the device-side flag zeroes the `touched` mask (parameters, moments and step counts untouched, `skipped_steps` + 1) and
    the next backward runs with 4x more headroom; autograd path: `last_backward_overflowed()` lets `fit_step` skip
    `optimizer.step()` the way GradScaler does."""
    geo = SegGptGeometry.tiny()
    net = SegGptNative(synth_state_dict(geo, seed=1), geo, device=DEV, dtype=torch.float16)
    g = torch.Generator().manual_seed(8)
    P0 = torch.rand(3, 3, 64, 64, generator=g)
    pix = ml_util.normalize(torch.rand(2, 3, 64, 64, generator=g)).to(DEV)
    lab = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
    pmc = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
    yes = torch.ones(2, 1, 64, 64, dtype=torch.bool, device=DEV)
    idx = torch.tensor([2, 0], device=DEV)
    eng = PromptTrainEngine(net, P0, lr=1e-2)
    eng.step(pix, lab, yes, idx, pmc)
    st = net.grad_overflow_state(2).clone()
    assert st.tolist()[:2] == [0, 0] and int(eng.skipped_steps) == 0
    before = (eng.params.clone(), eng.exp_avg.clone(), eng.steps.clone())
    bad = pix.clone()
    bad[0, 0, 3, 3] = float("inf")  # non-finite activations -> non-finite pred -> the INCOMING grad_pred is already non-finite
    eng.step(bad, lab, yes, idx, pmc)
    torch.cuda.synchronize()
    st = net.grad_overflow_state(2)
    # the step is dropped, but a bad batch is not the dgrad chain's doing: the back-off stays where it was
    assert st[0].item() == 1 and st[1].item() == 0 and st[3].item() == 0 and st[4].item() == 1 and st[5].item() == 1
    assert int(eng.skipped_steps) == 1
    assert torch.equal(eng.params, before[0]) and torch.equal(eng.exp_avg, before[1]) and torch.equal(eng.steps, before[2])
    eng.step(pix, lab, yes, idx, pmc)  # clean again: flag clears, the step is applied
    st = net.grad_overflow_state(2)
    assert st[0].item() == 0 and st[1].item() == 0 and st[4].item() == 0 and int(eng.skipped_steps) == 1
    assert not torch.equal(eng.params, before[0]) and torch.isfinite(eng.params).all()
    # a TRUE overflow of the half-precision chain on a finite gradient: push the scale target 2^12 up by hand (max |S * grad_pred|
    # lands at 2^20 > 65504) -> flag, step dropped, 4x more headroom for the next backward, counted as a dgrad overflow
    before = (eng.params.clone(), eng.exp_avg.clone(), eng.steps.clone())
    net.grad_overflow_state(2)[1] = -12
    eng.step(pix, lab, yes, idx, pmc)
    torch.cuda.synchronize()
    st = net.grad_overflow_state(2)
    assert st[0].item() == 1 and st[1].item() == -10 and st[3].item() == 1 and st[4].item() == 0 and st[5].item() == 1
    assert int(eng.skipped_steps) == 2 and torch.equal(eng.params, before[0]) and torch.equal(eng.steps, before[2])
    m = eng.overflow_metrics()
    assert m == {"skipped_steps": 2, "backoff_exp": -10, "dgrad_overflows": 1, "nonfinite_input_steps": 1}
    net.grad_overflow_state(2)[1] = 0
    # autograd boundary
    p = P0[:2].to(DEV).requires_grad_(True)
    out = net(pixel_values=bad, prompt_pixel_values=p, prompt_masks=pmc)
    ops.seggpt_loss(out.pred_masks, lab, yes, 0.01, "reference").backward()
    assert net.last_backward_overflowed()
    p.grad = None
    out = net(pixel_values=pix, prompt_pixel_values=p, prompt_masks=pmc)
    ops.seggpt_loss(out.pred_masks, lab, yes, 0.01, "reference").backward()
    assert not net.last_backward_overflowed() and torch.isfinite(p.grad).all()


def test_x3_mode_engine_step_tracks_exact_f32():
    """`gemm_x3` (float32 storage, every GEMM / attention MFMA as three f16 MFMAs on 22-bit operand splits, the dgrad chain
    on the device-chosen power-of-two multiple of the gradient): two optimiser steps of the fused engine against the exact-f32
    mode on the same inputs -- loss to 1e-6, parameters to fp32 round-off -- and the precision string "32-x3" selects it."""
    geo = SegGptGeometry.tiny()
    sd = synth_state_dict(geo, seed=1)
    g = torch.Generator().manual_seed(8)
    P0 = torch.rand(3, 3, 64, 64, generator=g)
    pix = ml_util.normalize(torch.rand(2, 3, 64, 64, generator=g)).to(DEV)
    lab = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
    pmc = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
    yes = torch.ones(2, 1, 64, 64, dtype=torch.bool, device=DEV)
    idx = torch.tensor([2, 0], device=DEV)
    res = []
    for x3 in (False, True):
        net = SegGptNative(sd, geo, device=DEV, dtype=torch.float32, gemm_x3=x3)
        eng = PromptTrainEngine(net, P0, lr=1e-2)
        losses = [eng.step(pix, lab, yes, idx, pmc) for _ in range(2)]
        torch.cuda.synchronize()
        assert int(eng.skipped_steps) == 0
        res.append((torch.stack(losses).cpu(), eng.params.cpu()))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-6, atol=0)
    assert float((res[0][1] - res[1][1]).abs().max()) < 2e-5  # two AdamW steps of lr 1e-2: sign-level agreement of the updates
    conf = BeachSegConfig(batch_size=2, checkpoint="synthetic:tiny", precision="32-x3", inpt_size=64, crop_size=64)
    pm = PromptModel(conf)
    assert pm.model.dtype == torch.float32 and pm.model.gemm_x3
    with pytest.raises(ValueError):
        SegGptNative(sd, geo, device=DEV, dtype=torch.bfloat16, gemm_x3=True)


@pytest.mark.parametrize("dtype,x3", [(torch.bfloat16, False), (torch.float16, False), (torch.float32, False), (torch.float32, True)])
def test_forward_rows_window_holds_the_bits_of_the_full_forward(dtype, x3):
    """`bsg_forward_rows(first_row = H/2)` -- the fused engine's forward: the decoder runs over the query half (plus the rows
    `bsg_backward_rows` reads back) -- against `bsg_forward` on the same inputs, every dtype: the written rows are bit-identical,
    the rows above the first computed tile are not touched, and the backward that follows gives the same prompt gradient."""
    geo = SegGptGeometry.tiny()
    net = SegGptNative(synth_state_dict(geo, seed=1), geo, device=DEV, dtype=dtype, gemm_x3=x3)
    H, W = geo.image_size
    g = torch.Generator().manual_seed(11)
    B = 3
    pix, prm, pm = (torch.randn(B, 3, H // 2, W, generator=g).to(DEV) for _ in range(3))
    gpred = torch.zeros(B, 3, H, W, device=DEV)
    gpred[:, :, H // 2:] = torch.randn(B, 3, H // 2, W, generator=g).to(DEV) * 1e-3
    full = net._run_forward(pix, prm, pm, 0, train=True)
    g_full = net._run_backward(gpred, B, first_row=H // 2)
    win = net._run_forward(pix, prm, pm, 0, train=True, first_row=H // 2)
    g_win = net._run_backward(gpred, B, first_row=H // 2)
    assert torch.equal(win[:, :, H // 2:], full[:, :, H // 2:])
    first_tile = max(0, 16 * ((H // 2 - 1) // 16) - 8) // 16 * 16
    assert first_tile > 0 and float(win[:, :, :first_tile].abs().max()) == 0.0
    assert torch.equal(win[:, :, first_tile:], full[:, :, first_tile:])
    assert torch.isfinite(g_win).all() and float(g_win.abs().max()) > 0 and torch.equal(g_win, g_full)
    with pytest.raises(RuntimeError):
        net._run_forward(pix, prm, pm, 0, train=True, first_row=H)
    # a backward over MORE rows than the forward computed would read decoder activations that were never written: refused
    with pytest.raises(RuntimeError, match="computed rows"):
        net._run_backward(gpred, B, first_row=0)
    net._run_forward(pix, prm, pm, 0, train=True)  # a full forward on the same workspace lifts it
    assert torch.equal(net._run_backward(gpred, B, first_row=0), g_full)


def test_validation_forward_and_predict_mosaic():
    geo = SegGptGeometry.tiny()
    net = SegGptNative(synth_state_dict(geo, seed=1), geo, device=DEV, dtype=torch.float32)
    conf = BeachSegConfig(batch_size=2, checkpoint="synthetic:tiny", precision="32-true", inpt_size=64, crop_size=16)
    pm = PromptModel(conf, model=net)
    pm.create_trainable_params(_prompts(4, 64, seed=3))
    g = torch.Generator().manual_seed(9)
    n, cs, mosaic = 6, 16, (40, 50)
    images = ml_util.normalize(torch.rand(n, 3, 64, 64, generator=g)).to(DEV)
    crop_idx = torch.tensor([0, 1, 2, 3, 0, 1])
    crops = torch.tensor([[0, 0, 16, 16], [16, 0, 32, 16], [40, 30, 56, 46], [-4, 10, 12, 26], [8, 8, 24, 24], [30, 20, 46, 36]],
                         dtype=torch.int32)
    state = pm.palette_g.get_state()
    out = predict_mosaic(pm, images, crop_idx, crops, mosaic, cs, batch_size=4)
    # numpy restatement of the same loop (src/predict.py:232-262) from the same decoded masks
    pm.palette_g.set_state(state)
    counter = np.zeros((*mosaic, 4), np.uint8)
    for s in range(0, n, 4):
        pred = pm({"image": images[s:s + 4], "crop_idx": crop_idx[s:s + 4]}).cpu().numpy()
        for j in range(pred.shape[0]):
            small = PO.nearest_resize(pred[j], cs)
            PO.accumulate(counter, tuple(int(v) for v in crops[s + j]), PO.one_hot(small, 4))
    assert np.array_equal(out.cpu().numpy(), PO.vote_argmax(counter).astype(np.uint8))
    assert out.shape == mosaic and out.dtype == torch.uint8
    # validation_step: prompt with the same crop_idx, eval aug, loss finite, F1 counters updated
    batch = {"image": images[:2], "mask": torch.randint(1, 4, (2, 1, 64, 64), dtype=torch.uint8), "crop_idx": torch.tensor([2, 3])}
    l = pm.validation_step(batch)
    assert torch.isfinite(l) and int(pm.val_metrics.state().sum()) > 0


def test_accumulator_finalises_the_previous_date():
    """`Accumulator.update` on a new date first saves the mosaic of the previous one (`src/predict.py:129-132`), and
    leaving the `with` block saves the last (`:90-91`)."""
    from beach_seg_amd.predict import Accumulator

    g = torch.Generator().manual_seed(4)
    masks = torch.randint(0, 4, (4, 32, 32), generator=g, dtype=torch.uint8).to(DEV)
    crops = torch.tensor([[0, 0, 16, 16], [16, 0, 32, 16], [8, 8, 24, 24], [0, 16, 16, 32]], dtype=torch.int32)
    seen = []
    with Accumulator((32, 32), ("nodata", "sand", "water", "veg"), DEV, on_finish=lambda d, m: seen.append(d)) as acc:
        acc.update("2020-01-01", crops[:2], masks[:2], 16, disjoint=True)
        acc.update("2020-01-01", crops[2:3], masks[2:3], 16)
        acc.update("2020-02-01", crops[3:], masks[3:], 16)
    assert seen == ["2020-01-01", "2020-02-01"] and [d for d, _ in acc.finished] == seen
    for date, sl in (("2020-01-01", slice(0, 3)), ("2020-02-01", slice(3, 4))):
        counter = np.zeros((32, 32, 4), np.uint8)
        for j in range(sl.start, sl.stop):
            PO.accumulate(counter, tuple(int(v) for v in crops[j]),
                          PO.one_hot(PO.nearest_resize(masks[j].cpu().numpy().astype(np.int64), 16), 4))
        got = dict(acc.finished)[date]
        assert np.array_equal(got.cpu().numpy(), PO.vote_argmax(counter).astype(np.uint8))


def test_load_model_from_checkpoint_files(tmp_path):
    """`load_model(path)` (`src/util/ml_util.py:7-13` with a LOCAL checkpoint: the hub name needs network): a
    `.safetensors` file and a `torch.save`d state dict (loaded with weights_only=True) give the same network as the
    in-memory state dict, and a missing file raises."""
    from safetensors.torch import save_file

    geo = SegGptGeometry.tiny()
    sd = synth_state_dict(geo, seed=4)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "ck.safetensors"))
    torch.save(sd, tmp_path / "ck.pt")
    g = torch.Generator().manual_seed(1)
    x = [torch.randn(2, 3, 64, 64, generator=g).to(DEV) for _ in range(3)]
    with torch.no_grad():
        want = SegGptNative(sd, geo, device=DEV, dtype=torch.float32)(pixel_values=x[0], prompt_pixel_values=x[1], prompt_masks=x[2]).pred_masks
        for name in ("ck.safetensors", "ck.pt"):
            net = ml_util.load_model(str(tmp_path / name), device=DEV, dtype=torch.float32, geometry=geo)
            got = net(pixel_values=x[0], prompt_pixel_values=x[1], prompt_masks=x[2]).pred_masks
            assert torch.equal(got, want), name
    with pytest.raises(FileNotFoundError):
        ml_util.load_model("BAAI/seggpt-vit-large", device=DEV)
    bad = dict(sd)
    bad.pop("decoder.decoder_pred.head.bias")
    with pytest.raises(KeyError):
        SegGptNative(bad, geo, device=DEV)


def test_hipgraph_replay_equals_eager_forward():
    """BASELINE config 4: the forward is capturable (no allocation / synchronisation inside the C ABI) and a graph
    replay reproduces the eager launch sequence bit for bit."""
    geo = SegGptGeometry.small()
    net = SegGptNative(synth_state_dict(geo, seed=2), geo, device=DEV, dtype=torch.bfloat16)
    g = torch.Generator(device=DEV).manual_seed(3)
    mk = lambda: torch.randn(2, 3, 448, 448, device=DEV, generator=g)
    graphed = net.capture_forward(2)
    for _ in range(2):
        pix, prm, pm = mk(), mk(), mk()
        with torch.no_grad():
            eager = net(pixel_values=pix, prompt_pixel_values=prm, prompt_masks=pm).pred_masks.clone()
        assert torch.equal(graphed(pix, prm, pm), eager)
