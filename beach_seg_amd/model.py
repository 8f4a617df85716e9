"""`PromptModel`: host-side mirror of `/root/reference/src/model.py:67-438` on the HIP path.  Same method
names, argument meaning and results for the compute the reference does per step: palette creation
(:215-231), prompt selection / colouring (:177-213), the network call (:139-144, :245-251, :282-288), arg-min
decode (:155-175), `SegGptLoss` (:40-64), AdamW + warm-up/cosine schedule (:385-428).  Out of scope (SURVEY.md
section 8): TensorBoard image grids (:310-383) and kornia's random augmentations (:195-224 of `src/data.py`;
only Normalize / CenterCrop-at-native-size are applied).

Lightning is not required: `training_step` returns the loss on the autograd graph exactly like the reference, and
`fit_step` does what Lightning does around it (`loss.backward()`, `optimizer.step()`, `zero_grad`).
"""
from __future__ import annotations

import math

import torch

from . import ml_util, ops
from .config import BeachSegConfig
from .seggpt import SegGptNative


class SegGptLoss(torch.nn.Module):
    """`src/model.py:40-64` on the fused HIP loss kernel; `variant="reference"` keeps the `unsqueeze(1)` broadcast."""

    def __init__(self, beta: float, variant: str = "reference"):
        super().__init__()
        self.beta, self.variant = beta, variant

    def forward(self, pred_masks, labels, yesdata):
        return ops.seggpt_loss(pred_masks, labels, yesdata, self.beta, self.variant)


class MulticlassF1:
    """Macro F1 over the non-ignored classes from tp/fp/fn counts (stand-in for
    `torchmetrics.MulticlassF1Score(num_classes, ignore_index=0)`, `src/model.py:85-93`; torchmetrics is not
    installed here, so this metric is "parity unpinned")."""

    def __init__(self, num_classes: int, ignore_index: int = 0, device="cpu"):
        self.n, self.ignore = num_classes, ignore_index
        self.tp = torch.zeros(num_classes, dtype=torch.long, device=device)
        self.fp = torch.zeros_like(self.tp)
        self.fn = torch.zeros_like(self.tp)

    def update(self, pred: torch.Tensor, target: torch.Tensor) -> None:
        keep = target != self.ignore
        p, t = pred[keep].long(), target[keep].long()
        for c in range(self.n):
            self.tp[c] += ((p == c) & (t == c)).sum()
            self.fp[c] += ((p == c) & (t != c)).sum()
            self.fn[c] += ((p != c) & (t == c)).sum()

    def state(self) -> torch.Tensor:  # (3, n) counts: what `sync_dist=True` all-reduces (src/model.py:316, 327)
        return torch.stack([self.tp, self.fp, self.fn])

    def compute(self, state: torch.Tensor | None = None) -> float:
        tp, fp, fn = (state if state is not None else self.state()).double()
        f1 = 2 * tp / (2 * tp + fp + fn).clamp_min(1)
        cls = [c for c in range(self.n) if c != self.ignore]
        return float(f1[cls].mean())

    def reset(self) -> None:
        self.tp.zero_(); self.fp.zero_(); self.fn.zero_()


def lr_at_epoch(conf: BeachSegConfig, epoch: int) -> float:
    """Closed form of `SequentialLR([LambdaLR(linear_warmup)], CosineAnnealingLR(T_max=epochs, eta_min))` stepped
    once per epoch (`src/model.py:385-428`), incl. the sqrt batch-size scaling (:386-393)."""
    ratio = (conf.batch_size * conf.world_size * conf.grad_accum_steps / conf.base_lr_batch_size) ** 0.5
    lr, init_lr, min_lr = conf.lr * ratio, conf.init_lr * ratio, conf.min_lr * ratio
    w = conf.warmup_epochs
    if w and epoch < w:
        return init_lr + (lr - init_lr) * epoch / w
    e = epoch - (w or 0)
    return min_lr + (lr - min_lr) * (1 + math.cos(math.pi * e / conf.epochs)) / 2


class PromptModel(torch.nn.Module):
    def __init__(self, conf: BeachSegConfig, model: SegGptNative | None = None, device="cuda:0"):
        super().__init__()
        self.conf = conf
        self.num_classes = len(conf.classes)
        self.nodata_idx = 0
        dtype = torch.float32 if conf.precision.startswith("32") else torch.bfloat16
        self.model = model if model is not None else ml_util.load_model(conf.checkpoint, device=device, dtype=dtype)
        self.train_metrics = MulticlassF1(self.num_classes, self.nodata_idx, self.model.device)
        self.val_metrics = MulticlassF1(self.num_classes, self.nodata_idx, self.model.device)
        self.g = torch.Generator(device="cpu")  # src/model.py:98-99 (the draw itself is tiny; kept on the host)
        self.g.manual_seed(conf.seed)
        self.palette_g = torch.Generator(device="cpu").manual_seed(conf.seed + 1)
        self.loss_fn = SegGptLoss(conf.loss_beta, conf.loss_variant)
        self.normalize, self.denormalize = ml_util.normalize, ml_util.denormalize
        self.current_epoch = 0

    @property
    def device(self) -> torch.device:
        return self.model.device

    # ---- src/model.py:115-130
    def create_trainable_params(self, prompt_imgs: list[dict]) -> None:
        dev = self.device
        self.prompt_batch = {
            "crop_idx": torch.tensor([p["crop_idx"] for p in prompt_imgs]),
            "date": [p["date"] for p in prompt_imgs],
            "mask": torch.stack([torch.as_tensor(p["mask"]) for p in prompt_imgs]).to(dev),
            "nodata": torch.stack([torch.as_tensor(p["nodata"]) for p in prompt_imgs]).to(dev),
        }
        params = [torch.nn.Parameter(torch.as_tensor(p["image"], dtype=torch.float32).to(dev).clone()) for p in prompt_imgs]
        self.prompt_params_list = torch.nn.ParameterList(params)
        self.prompt_batch["image"] = params

    # ---- src/model.py:215-231
    def create_palette(self, batch_size: int, train: bool) -> tuple[torch.Tensor, torch.Tensor]:
        if train:
            pal = ml_util.generate_random_rgb_palette(self.num_classes, batch_size, "cpu", self.palette_g).to(self.device)
        else:
            p = torch.tensor(ml_util.build_palette(self.num_classes - 1), dtype=torch.uint8)
            pal = torch.stack([p for _ in range(batch_size)]).to(self.device)
        mean = torch.tensor(ml_util.IMAGE_MEAN, device=self.device)
        std = torch.tensor(ml_util.IMAGE_STD, device=self.device)
        pal_norm = (pal.to(torch.float32) / 255 - mean) / std
        return pal, pal_norm

    # ---- src/model.py:177-213
    def prepare_prompt(self, batch_idxes, batch_palette: torch.Tensor, train: bool):
        idx = batch_idxes.flatten().tolist() if isinstance(batch_idxes, torch.Tensor) else (
            [batch_idxes] if isinstance(batch_idxes, int) else list(batch_idxes))
        image = torch.stack([self.prompt_batch["image"][i] for i in idx], dim=0)  # autograd-tracked stack
        mask = self.prompt_batch["mask"][torch.tensor(idx, device=self.device)]
        prompt_batch = {"image": self.normalize(image), "mask": mask,
                        "crop_idx": self.prompt_batch["crop_idx"][torch.tensor(idx)]}
        prompt_color = self.normalize(ml_util.torch_apply_mask_rgb(batch_palette, mask))
        return prompt_batch, prompt_color

    # ---- src/model.py:155-175
    def process_pred_masks(self, in_pred_masks: torch.Tensor, batch_palette_norm: torch.Tensor) -> torch.Tensor:
        if batch_palette_norm.shape[1:] != (4, 3):
            raise ValueError("process_pred_masks is defined for 4 palette entries x 3 channels (src/model.py:169)")
        return ops.decode_argmin(in_pred_masks.detach(), batch_palette_norm)

    # ---- src/model.py:132-147
    @torch.no_grad()
    def forward(self, batch_dict: dict) -> torch.Tensor:
        B = batch_dict["image"].shape[0]
        pal, pal_norm = self.create_palette(B, train=True)  # the reference draws a random palette here too
        prompt_batch, prompt_masks = self.prepare_prompt(batch_dict["crop_idx"], pal, train=False)
        out = self.model(pixel_values=batch_dict["image"].to(self.device), prompt_pixel_values=prompt_batch["image"],
                         prompt_masks=prompt_masks, embedding_type="instance")
        return self.process_pred_masks(out.pred_masks, pal_norm)

    def _step(self, batch: dict, prompt_idx, train: bool, metrics: MulticlassF1) -> torch.Tensor:
        B = batch["mask"].shape[0]
        mask = batch["mask"].to(self.device)
        pal, pal_norm = self.create_palette(B, train=True)
        color_mask_norm = self.normalize(ml_util.torch_apply_mask_rgb(pal, mask))
        prompt_batch, prompt_masks = self.prepare_prompt(prompt_idx, pal, train=train)
        out = self.model(pixel_values=batch["image"].to(self.device), labels=color_mask_norm,
                         prompt_pixel_values=prompt_batch["image"], prompt_masks=prompt_masks, embedding_type="instance")
        pred_masks = self.process_pred_masks(out.pred_masks, pal_norm)
        loss = self.loss_fn(out.pred_masks, color_mask_norm, mask != 0)
        metrics.update(pred_masks, mask.squeeze(1))
        return loss

    # ---- src/model.py:233-269
    def training_step(self, batch: dict, batch_idx: int = 0) -> torch.Tensor:
        B = batch["mask"].shape[0]
        prompt_idx = torch.randint(0, len(self.prompt_params_list), (B,), generator=self.g)
        return self._step(batch, prompt_idx, True, self.train_metrics)

    # ---- src/model.py:271-308
    @torch.no_grad()
    def validation_step(self, batch: dict, batch_idx: int = 0) -> torch.Tensor:
        return self._step(batch, batch["crop_idx"], False, self.val_metrics)

    # ---- src/model.py:385-428
    def configure_optimizers(self) -> dict:
        if self.conf.optimizer != "adamw":
            raise RuntimeError(f"Unexpected optimizer {self.conf.optimizer}")
        if self.conf.scheduler != "cosine":
            raise RuntimeError(f"Unexpected scheduler {self.conf.scheduler}")
        opt = torch.optim.AdamW(self.prompt_params_list.parameters(), lr=lr_at_epoch(self.conf, 0))
        return {"optimizer": opt, "lr_scheduler": {"interval": "epoch", "frequency": 1,
                                                   "lr_at_epoch": lambda e: lr_at_epoch(self.conf, e)}}

    def fit_step(self, batch: dict, optimizer: torch.optim.Optimizer) -> torch.Tensor:
        """What Lightning does around `training_step`: zero_grad(set_to_none) -> backward -> step."""
        optimizer.zero_grad(set_to_none=True)
        loss = self.training_step(batch)
        loss.backward()
        optimizer.step()
        return loss.detach()

    def on_epoch_end(self, optimizer: torch.optim.Optimizer) -> None:
        self.current_epoch += 1
        for gq in optimizer.param_groups:
            gq["lr"] = lr_at_epoch(self.conf, self.current_epoch)
