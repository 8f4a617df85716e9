"""Thin train driver reproducing `/root/reference/src/train.py:71-77, 97-122`: data module + model wiring
(`post_init`, `create_trainable_params`), `prompt_batch.pt` written before AND after fit (same path, `:76-77, 121-122`),
`conf.yaml` (`:111`), `classes.txt` (`:118`), `max_epochs = conf.epochs * len(prompt_batch)` -- the dict's key count, 5:
quirk kept (`:98`).  What Lightning's `Trainer.fit` does per epoch is spelled out: train batches through
`on_after_batch_transfer` (= `train_aug`, `src/data.py:295-313`) -> `training_step` -> backward -> AdamW, then the
validation loop (eval `aug`, `validation_step`), the per-epoch LR schedule and the epoch-level F1 log.

    python -m beach_seg_amd.train checkpoint=synthetic:tiny epochs=1 batch_size=2 inpt_size=64 crop_size=64
"""
from __future__ import annotations

import dataclasses
import sys
from pathlib import Path

import torch

from .config import BeachSegConfig
from .data import BeachSegDataModule
from .model import PromptModel


def _to_device_batch(batch: dict, device) -> dict:
    return {"image": batch["image"].to(device), "mask": batch["mask"][:, None].to(device), "crop_idx": batch["crop_idx"]}


def main(argv: list[str], datamodule: BeachSegDataModule | None = None, limit_batches: int | None = None) -> dict:
    conf = BeachSegConfig.from_dotlist(argv)
    torch.manual_seed(conf.seed)
    run_dir = Path(conf.model_training_root) / conf.project
    run_dir.mkdir(parents=True, exist_ok=True)
    dm = datamodule or BeachSegDataModule(conf)
    dm.setup("fit")
    model = PromptModel(conf)
    model.post_init(dm)  # src/train.py:73
    model.create_trainable_params(dm.prompt_imgs)

    def save_prompts():  # src/train.py:75-77 / 121-122 (`handle_item`: tensors to the CPU)
        torch.save({k: ([p.detach().cpu() for p in v] if k == "image" else (v.cpu() if torch.is_tensor(v) else v))
                    for k, v in model.prompt_batch.items()}, run_dir / "prompt_batch.pt")

    save_prompts()
    import yaml
    (run_dir / "conf.yaml").write_text(yaml.safe_dump({k: (list(v) if isinstance(v, tuple) else str(v) if isinstance(v, Path) else v)
                                                       for k, v in dataclasses.asdict(conf).items()}))  # src/train.py:111
    opt = model.configure_optimizers()["optimizer"]
    max_epochs = conf.epochs * len(model.prompt_batch)  # src/train.py:98 (len(dict) == 5)
    log = []
    for epoch in range(max_epochs):
        tl = []
        for bi, batch in enumerate(dm.train_dataloader()):
            if limit_batches is not None and bi >= limit_batches:
                break
            tl.append(model.fit_step(dm.train_aug(_to_device_batch(batch, model.device)), opt))
        vl = []
        for bi, batch in enumerate(dm.val_dataloader()):
            if limit_batches is not None and bi >= limit_batches:
                break
            vl.append(model.validation_step(dm.aug(_to_device_batch(batch, model.device))))
        rec = {"epoch": epoch, "lr": opt.param_groups[0]["lr"], "train/loss": float(torch.stack(tl).mean()),
               "val/loss": float(torch.stack(vl).mean()), "train/f1": model.train_metrics.compute(),
               "val/f1": model.val_metrics.compute()}
        log.append(rec)
        print(" ".join(f"{k} {v:.5f}" if isinstance(v, float) else f"{k} {v}" for k, v in rec.items()), flush=True)
        model.train_metrics.reset()
        model.val_metrics.reset()
        model.on_epoch_end(opt)
    (run_dir / "classes.txt").write_text("\n".join(conf.classes))  # src/train.py:117-118
    save_prompts()
    return {"run_dir": run_dir, "epochs": max_epochs, "log": log}


if __name__ == "__main__":
    main(sys.argv[1:])
