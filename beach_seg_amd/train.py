"""Thin train driver reproducing `/root/reference/src/train.py:71-77, 97-107, 115-122`: data module + model
wiring, `prompt_batch.pt` before and after fit, epochs = `conf.epochs * len(prompt_batch)` (the dict's key count,
5 -- quirk kept, `:98`).      python -m beach_seg_amd.train checkpoint=synthetic:tiny epochs=1 batch_size=2
"""
from __future__ import annotations

import sys
from pathlib import Path

import torch

from .config import BeachSegConfig
from .data import BeachSegDataModule
from .model import PromptModel


def main(argv: list[str]) -> None:
    conf = BeachSegConfig.from_dotlist(argv)
    torch.manual_seed(conf.seed)
    run_dir = Path(conf.model_training_root) / conf.project
    run_dir.mkdir(parents=True, exist_ok=True)
    dm = BeachSegDataModule(conf)
    dm.setup("fit")
    model = PromptModel(conf)
    model.create_trainable_params(dm.prompt_imgs)
    save = lambda: torch.save({k: ([p.detach().cpu() for p in v] if k == "image" else v)
                               for k, v in model.prompt_batch.items()}, run_dir / "prompt_batch.pt")
    save()
    opt = model.configure_optimizers()["optimizer"]
    max_epochs = conf.epochs * len(model.prompt_batch)  # src/train.py:98 (len(dict) == 5)
    for epoch in range(max_epochs):
        for batch in dm.train_dataloader():
            b = dm.train_aug({"image": batch["image"].to(model.device), "mask": batch["mask"][:, None].to(model.device),
                              "crop_idx": batch["crop_idx"]})
            loss = model.fit_step(b, opt)
        model.on_epoch_end(opt)
        print(f"epoch {epoch}: train/loss {loss.item():.5f} train/f1 {model.train_metrics.compute():.4f}", flush=True)
        model.train_metrics.reset()
    (run_dir / "classes.txt").write_text("\n".join(conf.classes))
    save()


if __name__ == "__main__":
    main(sys.argv[1:])
