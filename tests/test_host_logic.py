"""CPU tests of the host-side mirror of the reference interface (no GPU): config surface, palette / mask
colouring vs the reference-generated vectors, LR schedule vs torch's own schedulers, dataset dictionary, and the
N > 1 data-parallel reduction under gloo with world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from beach_seg_amd import ml_util
from beach_seg_amd.config import CLASSES, BeachSegConfig
from beach_seg_amd.data import BeachSegDataModule, padded_crop, tif_image
from beach_seg_amd.engine import reduce_prompt_grads, shard_batch
from beach_seg_amd.model import MulticlassF1, lr_at_epoch


def test_config_surface_matches_reference_fields():
    c = BeachSegConfig()
    assert c.classes == CLASSES == ("nodata", "sand", "water", "veg")
    assert (c.crop_size, c.inpt_size, c.loss_beta, c.lr, c.optimizer, c.scheduler) == (112, 448, 0.01, 1e-3, "adamw", "cosine")
    c2 = BeachSegConfig.from_dotlist(["batch_size=4", "lr=0.01", "precision=bf16-true", "scale=(0.5,1.0)", "debug=true"])
    assert (c2.batch_size, c2.lr, c2.precision, c2.scale, c2.debug) == (4, 0.01, "bf16-true", (0.5, 1.0), True)
    with pytest.raises(KeyError):
        BeachSegConfig.from_dotlist(["no_such_key=1"])


def test_palette_and_mask_colouring_vs_reference(golden_dir):
    rec = np.load(golden_dir / "wrapper.npz")
    assert np.array_equal(np.array(ml_util.build_palette(3)), rec["build_palette_3"])
    assert np.array_equal(np.array(ml_util.build_palette(7)), rec["build_palette_7"])
    pal = torch.from_numpy(rec["rand_palette_seed42"])
    # the colourisation itself is a HIP kernel (`bsg_mask_rgb_norm`, bit-exact in tests/test_gpu_parity.py): no CPU path
    from beach_seg_amd._native import NativeError
    from beach_seg_amd.model import PromptModel

    with pytest.raises(NativeError):
        ml_util.torch_apply_mask_rgb(pal, torch.from_numpy(rec["mask"]))
    # normalised palette: formed on the host in float32 with true divisions = the reference's CPU bits (src/model.py:221-229)
    assert np.array_equal(PromptModel._palette_norm(pal).numpy(), rec["pal_norm"])
    torch.manual_seed(42)  # the reference draws from the global RNG (src/util/ml_util.py:102-108)
    assert np.array_equal(ml_util.generate_random_rgb_palette(4, 3, "cpu").numpy(), rec["rand_palette_seed42"])
    x = torch.rand(2, 3, 4, 4)
    assert torch.allclose(ml_util.denormalize(ml_util.normalize(x)), x, atol=1e-6)


@pytest.mark.parametrize("warmup,epochs", [(0, 5), (2, 6)])
def test_lr_schedule_matches_torch_sequential_lr(warmup, epochs):
    """`configure_optimizers` (src/model.py:385-428) rebuilt with torch's own schedulers vs the closed form."""
    c = BeachSegConfig(batch_size=4, world_size=2, warmup_epochs=warmup, epochs=epochs)
    ratio = (4 * 2 / 1) ** 0.5
    lr, init_lr, min_lr = c.lr * ratio, c.init_lr * ratio, c.min_lr * ratio
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=lr)
    scheds, miles = [], []
    if warmup:
        scheds.append(torch.optim.lr_scheduler.LambdaLR(opt, [lambda e: ((lr - init_lr) * (e / warmup) + init_lr) / lr]))
        miles.append(warmup)
    scheds.append(torch.optim.lr_scheduler.CosineAnnealingLR(opt, epochs, min_lr))
    sched = torch.optim.lr_scheduler.SequentialLR(opt, scheds, miles)
    for e in range(epochs + warmup):
        assert abs(opt.param_groups[0]["lr"] - lr_at_epoch(c, e)) < 1e-9, e
        opt.step()
        sched.step()


def test_tile_front_end_and_dataset_dict():
    rng = np.random.default_rng(0)
    bands = rng.integers(200, 3200, size=(4, 64, 64), dtype=np.uint16)
    rgb = tif_image(bands)
    assert rgb.shape == (64, 64, 3) and rgb.dtype == np.uint8 and rgb.max() == 255
    c = padded_crop(np.arange(16).reshape(4, 4), (-1, 2, 3, 6), fill=99)
    assert c.shape == (4, 4) and c[0, 0] == 99 and c[0, 1] == 8 and c[2, 0] == 99 and c[1, 3] == 14
    conf = BeachSegConfig(crop_size=112, inpt_size=448)
    dm = BeachSegDataModule(conf)
    dm.setup("fit")
    item = dm.train_dataset[0]
    assert set(item) == {"crop_idx", "date", "image", "mask", "nodata"}  # src/data.py:118-124
    assert item["image"].shape == (3, 448, 448) and item["image"].dtype == np.float32
    assert 0.0 <= item["image"].min() and item["image"].max() <= 1.0
    assert item["mask"].shape == (448, 448) and item["mask"].dtype == np.uint8 and item["nodata"].dtype == bool
    assert len(dm.prompt_imgs) == len(dm.train_dataset)  # every train crop is a prompt (src/data.py:74-76)


def test_f1_counts():
    m = MulticlassF1(4, ignore_index=0)
    m.update(torch.tensor([1, 2, 2, 3, 1]), torch.tensor([1, 2, 3, 3, 0]))
    assert m.tp.tolist() == [0, 1, 1, 1] and m.fp.tolist() == [0, 0, 1, 0] and m.fn.tolist() == [0, 0, 0, 1]
    assert abs(m.compute() - (1.0 + 2 / 3 + 2 / 3) / 3) < 1e-9


def test_f1_macro_average_follows_torchmetrics_rules():
    """ignore_index drops the ignored TARGETS only: predicting the ignored class on a valid pixel is a false positive of
    that class, which then takes part in the macro mean with F1 = 0 (torchmetrics `_adjust_weights_safe_divide`: a class
    is left out only when tp + fp + fn == 0)."""
    m = MulticlassF1(4, ignore_index=0)
    m.update(torch.tensor([0, 2, 2, 3]), torch.tensor([1, 2, 3, 3]))
    assert m.confmat.tolist() == [[0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 1, 1]]
    # class 0: fp 1 -> F1 0 (counted); class 1: fn 1 -> 0; class 2: tp 1 fp 1 -> 2/3; class 3: tp 1 fn 1 -> 2/3
    assert abs(m.compute() - (0 + 0 + 2 / 3 + 2 / 3) / 4) < 1e-12
    m2 = MulticlassF1(4, ignore_index=None)
    m2.update(torch.tensor([0, 2]), torch.tensor([0, 2]))
    assert abs(m2.compute() - 1.0) < 1e-12  # classes 1 and 3 never appear: weight 0
    m.reset()
    assert int(m.confmat.sum()) == 0 and m.compute() == 0.0


def test_train_aug_parameters_and_reference_semantics():
    from beach_seg_amd.data import sample_train_aug_params
    from oracle.train_aug_oracle import train_aug_reference

    conf = BeachSegConfig(vertical_flip=0.5, horizontal_flip=0.5, erasing_p=0.6, gauss_p=0.5)
    g = torch.Generator().manual_seed(3)
    params, noise = sample_train_aug_params(16, 32, 48, conf, g)
    p2, n2 = sample_train_aug_params(16, 32, 48, conf, torch.Generator().manual_seed(3))
    assert torch.equal(params, p2) and torch.equal(noise, n2) and params.dtype == torch.int32 and params.shape == (16, 5)
    fl, ex, ey, ew, eh = params.unbind(1)
    assert int(fl.min()) >= 0 and int(fl.max()) <= 7 and 0 < int((fl & 1).sum()) < 16 and 0 < int(((fl >> 1) & 1).sum()) < 16
    on = ew > 0
    assert 0 < int(on.sum()) < 16 and bool(((ex + ew)[on] <= 48).all()) and bool(((ey + eh)[on] <= 32).all())
    area = (ew * eh)[on].float() / (32 * 48)
    assert float(area.min()) > 0.01 and float(area.max()) < 0.08  # erasing_scale (0.02, 0.05) up to rounding
    # semantics + gradient of the reference statement: flips undo themselves, the erased box gets no gradient
    img = torch.rand(2, 3, 8, 10, requires_grad=True)
    mask = torch.randint(0, 4, (2, 8, 10), dtype=torch.uint8)
    prm = torch.tensor([[1 | 4, 2, 1, 3, 2], [2, 0, 0, 0, 0]], dtype=torch.int32)
    nz = torch.randn(2, 3, 8, 10) * 0.1
    out, mo = train_aug_reference(img, mask, prm, nz)
    mean = torch.tensor((0.485, 0.456, 0.406)).view(3, 1, 1); std = torch.tensor((0.229, 0.224, 0.225)).view(3, 1, 1)
    x0 = img[0].detach().flip(-2).clone(); x0[:, 1:3, 2:5] = 0
    assert torch.allclose(out[0], (x0 + nz[0] - mean) / std, atol=1e-6) and torch.equal(mo[0], mask[0].flip(-2))
    assert torch.allclose(out[1], (img[1].detach().flip(-1) - mean) / std, atol=1e-6) and torch.equal(mo[1], mask[1].flip(-1))
    out.sum().backward()
    g0 = (torch.ones(3, 8, 10) / std); g0[:, 1:3, 2:5] = 0
    assert torch.allclose(img.grad[0], g0.flip(-2)) and torch.allclose(img.grad[1], (torch.ones(3, 8, 10) / std))
    # erase_mask (flags bit 5): the erased box becomes class 0 in the mask too; the sampler sets the bit on erased samples only
    prm2 = prm.clone(); prm2[0, 0] |= 32
    _, mo2 = train_aug_reference(img.detach(), mask, prm2, nz)
    want = mask[0].flip(-2).clone(); want[1:3, 2:5] = 0
    assert torch.equal(mo2[0], want) and torch.equal(mo2[1], mo[1])
    pe, _ = sample_train_aug_params(16, 32, 48, conf, torch.Generator().manual_seed(3), erase_mask=True)
    assert torch.equal(pe[:, 1:], params[:, 1:]) and torch.equal((pe[:, 0] & 32) != 0, params[:, 3] > 0) and torch.equal(pe[:, 0] & 31, params[:, 0])


def test_train_aug_color_parameters_and_statement():
    """The colour half of the sampler (ColorJiggle / RandomSharpness ranges of src/config.py:52-58 through kornia's documented
    generators) and the torch statement of the two operations: identity factors are the identity, HSV round-trips, a half-turn
    hue shift applied twice returns the image, sharpness factor 1 is the identity and factor 0 the blurred interior."""
    from beach_seg_amd.data import sample_train_aug_params
    from oracle.train_aug_oracle import _color_jiggle, _hsv_to_rgb, _rgb_to_hsv, _sharpness

    conf = BeachSegConfig(sharpness_p=0.5)
    params, noise, color = sample_train_aug_params(64, 16, 16, conf, torch.Generator().manual_seed(1), with_color=True)
    p2, _ = sample_train_aug_params(64, 16, 16, conf, torch.Generator().manual_seed(1))
    assert torch.equal(params[:, 1:], p2[:, 1:]) and torch.equal(params[:, 0] & 7, p2[:, 0])  # colour draws come last
    assert color.shape == (64, 6) and bool((params[:, 0] & 16).all()) and 0 < int(((params[:, 0] >> 3) & 1).sum()) < 64
    for k, a in enumerate((conf.brightness, conf.contrast, conf.saturation)):
        assert float(color[:, k].min()) >= 1 - a and float(color[:, k].max()) <= 1 + a
    assert float(color[:, 3].abs().max()) <= conf.hue and 0 <= float(color[:, 4].min()) and float(color[:, 4].max()) <= conf.sharpness
    code = int(color[0, 5])
    assert sorted((code >> (2 * k)) & 3 for k in range(4)) == [0, 1, 2, 3] and bool((color[:, 5] == code).all())
    x = torch.rand(3, 12, 14, generator=torch.Generator().manual_seed(2))
    assert torch.allclose(_hsv_to_rgb(_rgb_to_hsv(x)), x, atol=2e-6)
    ident = torch.tensor([1.0, 1.0, 1.0, 0.0, 1.0, float(0 | 1 << 2 | 2 << 4 | 3 << 6)])
    assert torch.allclose(_color_jiggle(x, ident), x, atol=2e-6)
    half = torch.tensor([1.0, 1.0, 1.0, 0.5, 1.0, float(3 | 3 << 2 | 0 << 4 | 1 << 6)])  # hue + 0.5 turn, twice
    assert torch.allclose(_color_jiggle(x, half), x, atol=5e-6)
    assert torch.equal(_sharpness(x, 1.0), x)
    blur = _sharpness(x, 0.0)
    assert torch.equal(blur[:, 0], x[:, 0]) and torch.equal(blur[:, :, -1], x[:, :, -1])
    want = (x[:, :3, :3].sum((1, 2)) + 4 * x[:, 1, 1]) / 13
    assert torch.allclose(blur[:, 1, 1], want, atol=1e-6)


def test_tif_image_eight_band_and_uint16(golden_dir):
    rec = np.load(golden_dir / "frontend_pil.npz")
    assert np.array_equal(tif_image(rec["tif8_bands"], rec["tif8_nodata"]), rec["tif8_rgb"])  # reference's own output
    u16 = np.random.default_rng(5).integers(200, 3200, size=(4, 20, 24), dtype=np.uint16)
    assert np.array_equal(tif_image(u16), tif_image(u16.astype(np.float32)))
    with pytest.raises(ValueError):
        tif_image(np.zeros((5, 4, 4), np.float32))


def test_shard_batch():
    assert list(shard_batch(8, 1, 4)) == [2, 3]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, n = 5, 7
    flat = torch.zeros(P * n + P)
    rows = [0, 3] if rank == 0 else [3, 4]  # rank-local prompts touched this step
    for r in rows:
        flat[r * n:(r + 1) * n] += (rank + 1) * (r + 1)
        flat[P * n + r] = 1.0
    reduce_prompt_grads(flat)
    # the second, tiny collective: logged loss + F1 state (src/model.py:316, 327 sync_dist=True)
    from beach_seg_amd.engine import reduce_metrics
    cm = torch.zeros(4, 4, dtype=torch.int64)
    cm[1, 1], cm[2, 3] = 3 + rank, 5
    mean_loss, cm_all = reduce_metrics(torch.tensor(2.0 * (rank + 1)), 2, cm)
    assert abs(float(mean_loss) - (2.0 + 4.0) / 4) < 1e-12 and cm_all[1, 1] == 7 and cm_all[2, 3] == 10 and int(cm_all.sum()) == 17
    # the inference-side collective (SURVEY section 8 e): per-rank u8 vote counters summed once, wrapping at 256 like `+= 1`
    from beach_seg_amd.predict import reduce_vote_counters
    votes = torch.zeros(3, 4, 2, dtype=torch.uint8)
    votes[0, 0, 0], votes[1, 2, 1], votes[2, 3, 0] = 200 + rank, 7 * rank, 255
    votes[0, 1, 1] = 200 if rank == 0 else 100  # neither rank wraps on its own; the SUM across ranks crosses 255 -> 44
    votes[0, 2, 0] = 255 if rank == 0 else 1    # exactly 256 votes -> 0, as the single process's uint8 `+= 1` gives
    reduce_vote_counters(votes)
    assert int(votes[0, 0, 0]) == (200 + 201) % 256 and int(votes[1, 2, 1]) == 7 and int(votes[2, 3, 0]) == (255 + 255) % 256
    assert int(votes[0, 1, 1]) == (200 + 100) % 256 == 44 and int(votes[0, 2, 0]) == 0
    assert int(votes.sum()) == 145 + 7 + 254 + 44
    q.put((rank, flat.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_reduction_gloo_world2():
    """N = 2 ranks, gloo on CPU: every rank ends with the SUM of the dense prompt-gradient rows and the union of
    the touched flags -- the one collective of the training step (engine.py)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    P, n = 5, 7
    want = torch.zeros(P * n + P)
    want[0:n] = 1 * 1
    want[3 * n:4 * n] = 1 * 4 + 2 * 4
    want[4 * n:5 * n] = 2 * 5
    want[P * n + 0], want[P * n + 3], want[P * n + 4] = 1, 2, 1
    assert torch.equal(outs[0], want) and torch.equal(outs[1], want)


def test_bench_two_rank_launch_contract_cpu_rehearsal():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank, 127.0.0.1), in the
    device-less rehearsal mode: rendezvous, the step's collective on a buffer of the engine's layout, barrier + max-over-ranks
    timing, and ONE JSON line from rank 0 carrying the N = 2 contract fields.  No kernel runs (N > 1 throughput stays
    unmeasured until the driver's multi-GPU node runs it)."""
    import json
    import os
    import subprocess
    import sys as _sys
    from pathlib import Path

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, BSG_BENCH_REHEARSE="cpu", MASTER_ADDR="127.0.0.1")
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--geometry", "tiny",
           "--prompts", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 128 and out["config"]["parallelism"] == "dp2"
    assert out["metric"] == "train tiles/sec" and out["unit"] == "tiles/s" and out["higher_is_better"] is True
    assert "rehearsal" in out and "NOT a measurement" in out["rehearsal"]


def test_sharded_accumulator_refuses_votes_after_the_rank_sum():
    """Sharded predict (`Accumulator(world > 1)`): once a date's counters hold the SUM over ranks, a further `update` for that
    date would add local votes on top of it on one rank only -- the mirror raises instead of diverging; a new date starts clean."""
    from beach_seg_amd.predict import Accumulator

    acc = Accumulator((4, 4), ("nodata", "sand", "water", "veg"), torch.device("cpu"), world=2)
    assert acc._reduced is False  # set in __init__: no AttributeError on the reduce paths before the first date
    with pytest.raises(AssertionError):
        acc.save_current()
    acc.initialize_current("d0")
    acc.reduce_votes()  # (no process group here: the reduction itself is a no-op, the state change is what is under test)
    with pytest.raises(RuntimeError, match="after its votes were reduced"):
        acc.update("d0", torch.zeros(0, 4, dtype=torch.int32), torch.zeros(0, 2, 2, dtype=torch.uint8), 2)
    acc.initialize_current("d1")
    assert acc._reduced is False


def test_bench_flop_counts_reference_and_executed():
    """`bench.py` prices utilisation on the FLOPs the fused step executes: the reference's count (SURVEY.md section 8 d: 1589.7 +
    1678.3 GF per ViT-L tile) minus the decoder / attention-backward rows that only feed the unread prompt half of the prediction
    (`bsg_forward_rows` / `bsg_backward_rows`, window formulas of `seggpt_api.hip`)."""
    import bench
    from beach_seg_amd.weights import SegGptGeometry

    g = SegGptGeometry.vit_large()
    fwd, bwd = bench.flops_per_tile(g)
    assert abs(fwd / 1e9 - 1589.7) < 0.1 and abs(bwd / 1e9 - 1678.3) < 0.1
    ex = bench.executed_flops_per_tile(g)
    dec = 2.0 * 1568 * 4096 * 16384
    conv, head, att = 2.0 * 896 * 448 * 9 * 64 * 64, 2.0 * 896 * 448 * 64 * 3, 4.0 * 1568 * 1568 * 1024
    want = fwd + bwd - dec * (25 + 27) / 56 - (conv + head) * 416 / 896 - conv * 432 / 896 - head * 424 / 896 \
        - att * (640 + 704) / 1568 - att * ((1568 - 896) / 1568 + 0.5)
    assert abs(ex - want) < 1e6 and 0.90 < ex / (fwd + bwd) < 0.95
    g5 = SegGptGeometry.config5()
    assert bench.executed_flops_per_tile(g5) < sum(bench.flops_per_tile(g5))
