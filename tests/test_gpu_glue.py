"""`-m gpu`: the device kernels either side of the network:
confusion matrix behind the F1 metric, `tif_image` on device, train-time augmentation with backward, the train
driver end to end, and the two-process data-parallel step."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from beach_seg_amd import ops
from beach_seg_amd.config import BeachSegConfig
from beach_seg_amd.data import sample_train_aug_params, tif_image
from oracle.train_aug_oracle import train_aug_reference
from oracle import frontend_oracle as FO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = Path(__file__).resolve().parents[1]


def test_confusion_kernel_vs_bincount():
    g = torch.Generator().manual_seed(0)
    n, K = 448 * 448 * 3 + 17, 4
    pred = torch.randint(0, K, (n,), generator=g)
    target = torch.randint(0, K, (n,), generator=g, dtype=torch.uint8)
    for ignore in (0, None, 2):
        for p in (pred.to(DEV), pred.to(torch.uint8).to(DEV)):
            cm = torch.zeros(K, K, dtype=torch.int64, device=DEV)
            ops.confusion_update(cm, p, target.to(DEV), ignore)
            ops.confusion_update(cm, p, target.to(DEV), ignore)  # accumulates
            t, q = target.long(), pred
            if ignore is not None:
                keep = t != ignore
                t, q = t[keep], q[keep]
            want = 2 * torch.bincount(t * K + q, minlength=K * K).reshape(K, K)
            assert torch.equal(cm.cpu(), want), ignore
    from beach_seg_amd.model import MulticlassF1
    m_dev, m_cpu = MulticlassF1(K, 0, DEV), MulticlassF1(K, 0, "cpu")
    m_dev.update(pred.to(DEV), target.to(DEV))
    m_cpu.update(pred, target)
    assert torch.equal(m_dev.confmat.cpu(), m_cpu.confmat) and abs(m_dev.compute() - m_cpu.compute()) < 1e-12


def test_tif_image_on_device(golden_dir):
    rec = np.load(golden_dir / "frontend_pil.npz")
    got = ops.tif_image(torch.from_numpy(rec["tif_bands"]).to(DEV), torch.from_numpy(rec["tif_nodata"]).to(DEV))
    assert np.array_equal(got.cpu().numpy(), rec["tif_rgb"])  # 4 bands: bit-exact vs the reference's own output
    g8 = ops.tif_image(torch.from_numpy(rec["tif8_bands"]).to(DEV), torch.from_numpy(rec["tif8_nodata"]).to(DEV)).cpu().numpy()
    d8 = np.abs(g8.astype(int) - rec["tif8_rgb"].astype(int))
    print(f"[measured] 8-band tif_image: {int((d8 > 0).sum())} of {d8.size} bytes differ, max {int(d8.max())}")
    assert d8.max() <= 1 and (d8 > 0).mean() < 1e-3  # log10: libm float vs correctly rounded double, then truncation
    # raw uint16 Dove tile -> RGB -> PIL-exact resized, normalised network input, never leaving the device
    from beach_seg_amd.data import synthetic_dove_bands
    raw = synthetic_dove_bands(1234, 256)
    rgb = ops.tif_image(torch.from_numpy(raw).to(DEV))
    assert np.array_equal(rgb.cpu().numpy(), tif_image(raw))
    boxes = np.array([[0, 0, 112, 112], [200, 200, 312, 312]], dtype=np.int32)
    x = ops.tile_frontend(rgb, torch.from_numpy(boxes).to(DEV), 112, 448)
    _, want = FO.tile_frontend(tif_image(raw), boxes, 112, 448)
    assert np.array_equal(x.cpu().numpy(), want)
    with pytest.raises(ValueError):
        ops.tif_image(torch.zeros(3, 8, 8, device=DEV))


def test_train_aug_kernel_forward_and_backward_vs_torch():
    conf = BeachSegConfig(vertical_flip=0.5, horizontal_flip=0.5, erasing_p=0.7, gauss_p=0.6, erasing_scale=(0.05, 0.2))
    g = torch.Generator().manual_seed(11)
    B, h, w = 6, 40, 56
    params, noise = sample_train_aug_params(B, h, w, conf, g)
    assert noise is not None and int((params[:, 3] > 0).sum()) > 0
    img = torch.rand(B, 3, h, w, generator=g)
    mask = torch.randint(0, 4, (B, 1, h, w), generator=g, dtype=torch.uint8)
    a = img.clone().requires_grad_(True)
    ref, mref = train_aug_reference(a, mask[:, 0], params, noise)
    gout = torch.randn(B, 3, h, w, generator=g)
    (ref * gout).sum().backward()
    b = img.to(DEV).requires_grad_(True)
    out, mo = ops.train_aug(b, mask.to(DEV), params.to(DEV), noise.to(DEV))
    (out * gout.to(DEV)).sum().backward()
    assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=1e-6, atol=1e-6)
    assert torch.equal(mo.cpu()[:, 0], mref)
    assert torch.allclose(b.grad.cpu(), a.grad, rtol=1e-6, atol=1e-7)
    out2, _ = ops.train_aug(img.to(DEV), None, params.to(DEV), None)  # no mask, no noise tensor
    ref2, _ = train_aug_reference(img, None, params, None)
    assert torch.allclose(out2.cpu(), ref2, rtol=1e-6, atol=1e-6)
    # erase_mask (flags bit 5): the erased box is class 0 in the mask as well
    pe, _ = sample_train_aug_params(B, h, w, conf, torch.Generator().manual_seed(11), erase_mask=True)
    assert torch.equal(pe[:, 1:], params[:, 1:]) and int((pe[:, 0] & 32 != 0).sum()) == int((params[:, 3] > 0).sum())
    _, me = ops.train_aug(img.to(DEV), mask.to(DEV), pe.to(DEV), noise.to(DEV))
    _, mref_e = train_aug_reference(img, mask[:, 0], pe, noise)
    assert torch.equal(me.cpu()[:, 0], mref_e) and not torch.equal(mref_e, mref)


def test_train_aug_color_jiggle_and_sharpness_vs_torch():
    """ColorJiggle + RandomSharpness inside the device augmentation chain (`bsg_train_aug` with colour parameters) against the
    torch statement of kornia's published formulas (`oracle/train_aug_oracle.py`; kornia itself is not installable here:
    parity unpinned), forward and the gradient wrt the prompt pixels, for every operation order class, factors on both
    sides of 1 (the clamps engage), a hue shift that wraps, sharpness factors inside and outside (0, 1), with and without
    the operations per sample."""
    g = torch.Generator().manual_seed(5)
    B, h, w = 8, 24, 36
    conf = BeachSegConfig(vertical_flip=0.5, horizontal_flip=0.5, erasing_p=0.5, gauss_p=0.5, erasing_scale=(0.05, 0.2),
                          brightness=0.3, contrast=0.4, saturation=0.6, hue=0.4, sharpness=1.0, sharpness_p=0.6)
    params, noise, color = sample_train_aug_params(B, h, w, conf, g, with_color=True)
    perms = [(0, 1, 2, 3), (3, 2, 1, 0), (2, 0, 3, 1), (1, 3, 0, 2)]
    for b in range(B):
        color[b, 5] = float(sum(o << (2 * k) for k, o in enumerate(perms[b % 4])))
    color[1, 4] = 1.7   # sharpening proper: the blend is clamped
    color[2, 4] = 0.0   # the blurred image itself
    params[1, 0] |= 8
    params[2, 0] |= 8
    params[3, 0] &= ~16  # no colour jiggle on this sample
    params[4, 0] &= ~8
    assert noise is not None
    img = torch.rand(B, 3, h, w, generator=g)
    img[5, :, :4] = img[5, :1, :4]  # grey pixels: zero chroma (the `deltac == 0` branch of rgb_to_hsv)
    a = img.clone().requires_grad_(True)
    ref, _ = train_aug_reference(a, None, params, noise, color=color)
    gout = torch.randn(B, 3, h, w, generator=g)
    (ref * gout).sum().backward()
    b_ = img.to(DEV).requires_grad_(True)
    out, _ = ops.train_aug(b_, None, params.to(DEV), noise.to(DEV), color=color.to(DEV))
    (out * gout.to(DEV)).sum().backward()
    assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=2e-5)
    # gradients: identical up to float rounding except where a clamp / hue sector boundary sits within rounding of the pixel
    ga, gb = a.grad, b_.grad.cpu()
    grey = torch.zeros(B, 3, h, w, dtype=torch.bool)
    grey[5, :, :4] = True  # at zero chroma the max / min picks are ties: autograd's choice of winner is not defined
    close = torch.isclose(gb, ga, rtol=1e-3, atol=1e-4) | grey
    assert float(close.float().mean()) > 0.999, float(close.float().mean())
    assert float((gb - ga)[~grey].norm() / ga[~grey].norm()) < 2e-2
    with pytest.raises(ValueError):
        ops.train_aug(b_, None, params.to(DEV), None, color=color[:, :5].to(DEV))


def test_train_driver_end_to_end(tmp_path):
    """`beach_seg_amd.train.main` (src/train.py:71-122): prompt_batch.pt before != after, conf.yaml, classes.txt, and
    epochs = conf.epochs * 5 (the `len(dict)` quirk of src/train.py:98)."""
    from beach_seg_amd import train
    from beach_seg_amd.data import BeachSegDataModule, synthetic_dove_scene

    argv = ["checkpoint=synthetic:tiny", "epochs=1", "batch_size=2", "inpt_size=64", "crop_size=64", "precision=32-true",
            f"model_training_root={tmp_path}", "project=run0", "lr=0.01", "erasing_p=0.5", "gauss_p=0.5"]
    conf = BeachSegConfig.from_dotlist(argv)
    dm = BeachSegDataModule(conf, scene=synthetic_dove_scene(size=128))  # 4 crops of 64 px -> 4 prompts
    seen = {}
    orig_save = torch.save

    def spy(obj, f, *a, **k):
        if str(f).endswith("prompt_batch.pt"):
            seen.setdefault("saves", []).append(torch.stack([t.clone() for t in obj["image"]]))
        return orig_save(obj, f, *a, **k)

    torch.save = spy
    try:
        res = train.main(argv, datamodule=dm)
    finally:
        torch.save = orig_save
    run = tmp_path / "run0"
    assert res["epochs"] == 5 and len(res["log"]) == 5  # epochs * len(prompt_batch dict) = 1 * 5
    assert (run / "classes.txt").read_text().split("\n") == list(conf.classes)
    assert "checkpoint: synthetic:tiny" in (run / "conf.yaml").read_text()
    before, after = seen["saves"]
    assert before.shape == (4, 3, 64, 64) and not torch.equal(before, after)
    saved = torch.load(run / "prompt_batch.pt", weights_only=True)
    assert set(saved) == {"crop_idx", "date", "mask", "nodata", "image"} and torch.equal(torch.stack(saved["image"]), after)
    assert all(np.isfinite(r["train/loss"]) and np.isfinite(r["val/loss"]) and 0 <= r["val/f1"] <= 1 for r in res["log"])
    # cosine schedule stepped per epoch with T_max = conf.epochs = 1 while max_epochs = 5 (the len(dict) quirk): it swings
    # between lr and min_lr every epoch, as torch's CosineAnnealingLR does past T_max
    assert res["log"][0]["lr"] > res["log"][1]["lr"] and abs(res["log"][0]["lr"] - res["log"][2]["lr"]) < 1e-12


def test_two_process_data_parallel_step_on_one_gpu(tmp_path):
    """Two FRESH processes (gloo, both on cuda:0) run two `PromptTrainEngine.step`s on different batches.  (a) both
    ranks end with bit-identical params / exp_avg / exp_avg_sq / steps; (b) they equal ONE process stepping on the
    concatenated batch (per-sample loss, all pixels valid: the mean of the two rank gradients IS the gradient of the
    concatenated batch), to 1e-6."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "dp_worker.py"), str(tmp_path / f"rank{r}.pt")],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(2))
    for k in ("params", "exp_avg", "exp_avg_sq", "steps"):
        assert torch.equal(r0[k], r1[k]), k
    assert not torch.equal(r0["loss"], r1["loss"])  # different data on the two ranks
    # one process, concatenated batch
    sys.path.insert(0, str(ROOT / "tests"))
    import dp_worker

    single = dp_worker.run(world=1, rank=0)
    for k in ("params", "exp_avg", "exp_avg_sq"):
        err = float((single[k] - r0[k]).abs().max() / single[k].abs().max().clamp_min(1e-30))
        print(f"[measured] DP world 2 vs single process, {k}: rel err {err:.2e}")
        assert err < 1e-6, k
    assert torch.equal(single["steps"], r0["steps"])
    assert abs(float(single["loss"].mean()) - float((r0["loss"] + r1["loss"]).mean() / 2)) < 1e-6 * abs(float(single["loss"].mean()))
    assert torch.equal(r0["confmat"], single["confmat"]) and abs(float(r0["mean_loss"]) - float(single["mean_loss"])) < 1e-6
    # sharded predict (SURVEY section 8 e, inference): both ranks hold the full mosaic, bit-identical to one process
    one = dp_worker.run_predict(world=1, rank=0)["mosaic"]
    assert torch.equal(r0["mosaic"], one) and torch.equal(r1["mosaic"], one) and len(one.unique()) > 1


def test_empty_and_degenerate_inputs():
    """Edge cases the reference's own code paths accept: an empty batch through the colourisation (torch's advanced indexing
    returns an empty tensor), a predict loop without windows (the mosaic stays class 0, `np.argmax` of all-zero votes), and a
    window list shorter than one batch with `use_graph` (falls back to the eager forward)."""
    from beach_seg_amd.config import BeachSegConfig
    from beach_seg_amd.model import PromptModel
    from beach_seg_amd.predict import predict_mosaic
    from beach_seg_amd.seggpt import SegGptNative
    from beach_seg_amd.weights import SegGptGeometry, synth_state_dict

    pal = torch.zeros(0, 4, 3, dtype=torch.uint8, device=DEV)
    out = ops.mask_rgb_norm(pal, torch.zeros(0, 8, 8, dtype=torch.uint8, device=DEV))
    assert out.shape == (0, 3, 8, 8) and out.dtype == torch.float32
    geo = SegGptGeometry.tiny()
    net = SegGptNative(synth_state_dict(geo, seed=1), geo, device=DEV, dtype=torch.float32)
    conf = BeachSegConfig(batch_size=2, checkpoint="synthetic:tiny", precision="32-true", inpt_size=64, crop_size=16)
    pm = PromptModel(conf, model=net)
    g = torch.Generator().manual_seed(2)
    pm.create_trainable_params([{"crop_idx": i, "date": "d", "image": torch.rand(3, 64, 64, generator=g).numpy(),
                                 "mask": torch.randint(0, 4, (64, 64), generator=g, dtype=torch.uint8).numpy(),
                                 "nodata": np.zeros((64, 64), bool)} for i in range(2)])
    empty = predict_mosaic(pm, torch.zeros(0, 3, 64, 64), torch.zeros(0, dtype=torch.long), torch.zeros(0, 4, dtype=torch.int32),
                           (24, 40), 16, batch_size=4)
    assert empty.shape == (24, 40) and int(empty.sum()) == 0
    imgs = torch.randn(3, 3, 64, 64, generator=g)
    crops = torch.tensor([[0, 0, 16, 16], [16, 0, 32, 16], [0, 8, 16, 24]], dtype=torch.int32)
    state = pm.palette_g.get_state()
    a = predict_mosaic(pm, imgs, torch.tensor([0, 1, 0]), crops, (24, 40), 16, batch_size=4, use_graph=True)  # 3 < 4: eager
    pm.palette_g.set_state(state)
    b = predict_mosaic(pm, imgs, torch.tensor([0, 1, 0]), crops, (24, 40), 16, batch_size=2)
    assert a.shape == (24, 40)
    assert torch.equal(a[:, 32:], torch.zeros_like(a[:, 32:]))  # no window there
    # different batch splits draw different palette sequences (the reference draws per batch), so only shapes / support match
    covered = torch.zeros(24, 40, dtype=torch.bool)
    for x0, y0, x1, y1 in crops.tolist():
        covered[y0:min(y1, 24), x0:min(x1, 40)] = True
    for m in (a, b):  # classes in range, nothing voted outside the union of the windows
        assert m.shape == (24, 40) and int(m.min()) >= 0 and int(m.max()) <= 3
        assert int(m.cpu()[~covered].abs().sum()) == 0
