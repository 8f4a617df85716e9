"""One rank of the two-process data-parallel test (`tests/test_gpu_glue.py`): tiny SegGPT in f32, two engine steps.
world 2: this rank's half of the global batch, gloo collectives (two ranks share cuda:0; RCCL refuses that);
world 1: the whole batch in one process (the reference the ranks must reproduce)."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def run(world: int, rank: int) -> dict:
    import torch.distributed as dist

    from beach_seg_amd import ops
    from beach_seg_amd.engine import PromptTrainEngine, reduce_metrics, shard_batch
    from beach_seg_amd.model import MulticlassF1
    from beach_seg_amd.seggpt import SegGptNative
    from beach_seg_amd.weights import SegGptGeometry, synth_state_dict

    dev = torch.device("cuda:0")
    g = SegGptGeometry.tiny()
    net = SegGptNative(synth_state_dict(g, seed=1), g, device=dev, dtype=torch.float32)
    gen = torch.Generator().manual_seed(123)
    GB, P, H, W = 4, 5, g.image_size[0] // 2, g.image_size[1]
    rn = lambda *s: torch.randn(*s, generator=gen)
    pix, label, pmask = rn(2, GB, 3, H, W), rn(2, GB, 3, H, W), rn(2, GB, 3, H, W)
    target = torch.randint(1, 4, (2, GB, H, W), generator=gen, dtype=torch.uint8)
    idx = torch.tensor([[0, 2, 2, 4], [1, 0, 3, 3]])
    # ranks start from DIFFERENT prompt values on purpose: the engine must broadcast rank 0's
    P0 = torch.rand(P, 3, H, W, generator=torch.Generator().manual_seed(7 + (rank if world > 1 else 0)))
    eng = PromptTrainEngine(net, P0, lr=1e-2, loss_variant="per_sample")
    sl = list(shard_batch(GB, rank, world))
    yes = torch.ones(len(sl), 1, H, W, dtype=torch.bool, device=dev)
    f1 = MulticlassF1(4, 0, dev)
    losses = []
    for step in range(2):
        losses.append(eng.step(pix[step, sl].to(dev), label[step, sl].to(dev), yes, idx[step, sl].to(dev), pmask[step, sl].to(dev)))
        t = target[step, sl].to(dev)
        f1.update(t.long() % 3 + 1, t)  # a deterministic stand-in for the decoded masks
    mean_loss, cm = reduce_metrics(torch.stack(losses).sum(), len(losses), f1.confmat)
    torch.cuda.synchronize()
    return {"params": eng.params.cpu(), "exp_avg": eng.exp_avg.cpu(), "exp_avg_sq": eng.exp_avg_sq.cpu(),
            "steps": eng.steps.cpu(), "loss": torch.stack(losses).cpu(), "mean_loss": mean_loss.cpu(), "confmat": cm.cpu()}


def run_predict(world: int, rank: int) -> dict:
    """Sharded sliding-window predict (`predict.predict_mosaic(rank=, world=)`): overlapping windows (stride < crop, so votes
    from different ranks land on the same pixels), a window count that does not divide by the world size.  Palettes are
    RANDOM per batch like the reference's (`src/model.py:134`): every rank draws the unsharded loop's sequence and window i
    keeps row i of it, so the sharded mosaic must equal the single process's bit for bit without pinning anything."""
    import numpy as np

    from beach_seg_amd.config import BeachSegConfig
    from beach_seg_amd.model import PromptModel
    from beach_seg_amd.predict import grid_crops, predict_mosaic
    from beach_seg_amd.seggpt import SegGptNative
    from beach_seg_amd.weights import SegGptGeometry, synth_state_dict

    dev = torch.device("cuda:0")
    geo = SegGptGeometry.tiny()
    net = SegGptNative(synth_state_dict(geo, seed=1), geo, device=dev, dtype=torch.float32)
    conf = BeachSegConfig(batch_size=4, checkpoint="synthetic:tiny", precision="32-true", inpt_size=64, crop_size=16)
    pm = PromptModel(conf, model=net)
    gen = torch.Generator().manual_seed(5)
    pm.create_trainable_params([{"crop_idx": i, "date": "d", "image": torch.rand(3, 64, 64, generator=gen).numpy(),
                                 "mask": torch.randint(0, 4, (64, 64), generator=gen, dtype=torch.uint8).numpy(),
                                 "nodata": np.zeros((64, 64), bool)} for i in range(3)])
    crops = grid_crops(40, 56, 16, stride=8)  # 5 x 7 = 35 windows
    n = crops.shape[0]
    images = torch.randn(n, 3, 64, 64, generator=gen)
    crop_idx = torch.randint(0, 3, (n,), generator=gen)
    mosaic = predict_mosaic(pm, images, crop_idx, crops, (40, 56), 16, batch_size=4, rank=rank, world=world)
    torch.cuda.synchronize()
    return {"mosaic": mosaic.cpu()}


if __name__ == "__main__":
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = run(world, rank)
    out.update(run_predict(world, rank))
    torch.save(out, sys.argv[1])
    dist.barrier()
    dist.destroy_process_group()
