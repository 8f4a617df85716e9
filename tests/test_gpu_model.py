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
