#!/usr/bin/env python3
"""BASELINE config 4 (a parity-test configuration, NOT the headline bench line): inference-only sliding window
over a synthetic mosaic, 1 GPU, network forward replayed from a captured hipGraph, votes accumulated on device.

    python bench_predict.py [--size 8192] [--batch 64] [--crop 112]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench_predict.py   # N replicas

With N > 1 ranks every GPU holds a replica of the network, takes windows rank, rank + N, ... and the uint8 vote counters are
summed once at the end (`Accumulator.reduce_votes`, RCCL): SURVEY.md section 8 (e), inference.

The reference's loop (`src/predict.py:232-262`) runs batch_size 1 on the CPU; here the 5,476 windows of an
8192 x 8192 mosaic are cut (zero-padded at the edges), resized to 448 x 448 with Pillow-exact BICUBIC and normalised on
device (`ops.tile_frontend`), pushed through the ViT-L forward in batches, decoded, nearest-resized and voted without
leaving the GPU."""
import argparse, json, os, sys, time
from pathlib import Path
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the HIP runtime initialises (dmabuf IPC only on this pool: RCCL needs it)
import torch
import torch.nn.functional as F
sys.path.insert(0, str(Path(__file__).resolve().parent))
from beach_seg_amd import ml_util, ops  # noqa: F401
from beach_seg_amd.config import BeachSegConfig
from beach_seg_amd.model import PromptModel
from beach_seg_amd.predict import Accumulator, crops_are_disjoint, grid_crops


def run_predict(pm: PromptModel, size: int = 8192, batch: int = 64, crop: int = 112, prompts: int = 32, use_graph: bool = True,
                rank: int = 0, world: int = 1) -> dict:
    """The whole config-4 pipeline on `pm` (a `PromptModel` on the HIP network); returns the result record.  Timed region:
    palette draw + device front-end + forward + decode + votes + arg-max over every window of this rank."""
    dev = pm.device
    conf = pm.conf
    g = torch.Generator(device=dev).manual_seed(11)
    S = conf.inpt_size
    pm.create_trainable_params([{"crop_idx": i, "date": "d", "image": torch.rand(3, S, S, device=dev, generator=g),
                                 "mask": torch.randint(0, 4, (S, S), device=dev, generator=g, dtype=torch.uint8),
                                 "nodata": torch.zeros(S, S, dtype=torch.bool)} for i in range(prompts)])
    mosaic = (torch.rand(size // 64, size // 64, 3, device=dev, generator=g).repeat_interleave(64, 0).repeat_interleave(64, 1) * 255).to(torch.uint8)
    crops_all = grid_crops(size, size, crop)
    n_all = crops_all.shape[0]
    order = torch.arange(n_all)[rank::world]  # this rank's windows
    crops = crops_all[order]
    n = crops.shape[0]
    graphed = pm.model.capture_forward(batch) if use_graph else None
    acc = Accumulator((size, size), conf.classes, dev)
    acc.initialize_current("d0")
    # host data of the whole loop goes up once (per-batch uploads block the host behind the previous batch's forward)
    crops_dev = crops.to(dev)
    idx_dev = (order % prompts).to(dev)
    sizes = [min(batch, n - s) for s in range(0, n, batch)]
    disjoint = crops_are_disjoint(crops)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    palettes = pm.create_palettes(sizes, train=True)
    with torch.no_grad():
        for b, s in enumerate(range(0, n, batch)):
            cb = crops_dev[s:s + batch]
            img = ops.tile_frontend(mosaic, cb, crop, S)  # padded crop + PIL-BICUBIC + /255 + Normalize
            pal, pal_norm = palettes[b]
            pb, pmasks = pm.prepare_prompt(idx_dev[s:s + batch], pal, train=False)
            if graphed is not None and img.shape[0] == batch:
                out = graphed(img, pb["image"], pmasks)
            else:
                out = pm.model(pixel_values=img, prompt_pixel_values=pb["image"], prompt_masks=pmasks).pred_masks
            pred = pm.process_pred_masks(out, pal_norm)
            acc.update("d0", cb if disjoint else crops[s:s + batch], pred.to(torch.uint8), crop, disjoint=disjoint)
    if world > 1:
        acc.reduce_votes()
    result = acc.result()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return {"config": f"predict sliding window {size}x{size}, crop {crop}, batch {batch}, hipGraph={use_graph}",
            "n_gpus": world, "tiles": n_all, "seconds": round(dt, 3), "tiles_per_s": round(n_all / dt, 1),
            "fwd_ms_per_tile": round(dt / n_all * 1e3, 3), "inference_tflops": round(n_all / dt * 1.5897, 1),
            "classes_present": torch.unique(result).tolist()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--crop", type=int, default=112)
    ap.add_argument("--prompts", type=int, default=32)
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    conf = BeachSegConfig(checkpoint="synthetic:vit_large", precision="bf16-true", crop_size=a.crop, batch_size=a.batch)
    pm = PromptModel(conf, device=dev)
    rec = run_predict(pm, a.size, a.batch, a.crop, a.prompts, not a.no_graph, rank, world)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if rank:
            return
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
