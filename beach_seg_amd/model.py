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

    def forward_ids(self, pred_masks, class_ids, palette_norm):
        """`forward(pred, normalize(torch_apply_mask_rgb(palette, class_ids)), class_ids != 0)` (`src/model.py:238-239,
        255`) with the colourisation folded into the loss kernel."""
        return ops.seggpt_loss_ids(pred_masks, class_ids, palette_norm, self.beta, self.variant)


class MulticlassF1:
    """`torchmetrics.MulticlassF1Score(num_classes, ignore_index=0)` (macro average; `src/model.py:85-93`).  State = the
    (K, K) confusion matrix over the pixels whose target is not ignored, updated by ONE HIP kernel per call
    (`bsg_confusion_update`, no host synchronisation); it is what `sync_dist=True` sums over ranks
    (`src/model.py:316, 327`; `engine.reduce_metrics`).  `compute` restates torchmetrics 1.x (`_multiclass_stat_scores_
    update` + `_fbeta_reduce` / `_adjust_weights_safe_divide`): tp = diag, fp = column sums - tp, fn = row sums - tp,
    F1_c = 2 tp / (2 tp + fp + fn) with 0/0 := 0, macro mean over the classes with tp + fp + fn > 0.  torchmetrics is not
    installed here, so the metric is "parity unpinned" beyond this restatement."""

    def __init__(self, num_classes: int, ignore_index: int | None = 0, device="cpu"):
        self.n, self.ignore = num_classes, ignore_index
        self.confmat = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=device)

    def update(self, pred: torch.Tensor, target: torch.Tensor) -> None:
        if self.confmat.is_cuda:
            ops.confusion_update(self.confmat, pred, target, self.ignore)
        else:  # host tensors (CPU unit tests of the formula): the same counts with bincount
            t, p = target.flatten().long(), pred.flatten().long()
            if self.ignore is not None:
                keep = t != self.ignore
                t, p = t[keep], p[keep]
            self.confmat += torch.bincount(t * self.n + p, minlength=self.n * self.n).reshape(self.n, self.n)

    def state(self) -> torch.Tensor:
        return self.confmat

    @property
    def tp(self):
        return self.confmat.diag()

    @property
    def fp(self):
        return self.confmat.sum(0) - self.confmat.diag()

    @property
    def fn(self):
        return self.confmat.sum(1) - self.confmat.diag()

    def compute(self, state: torch.Tensor | None = None) -> float:
        cm = (state if state is not None else self.confmat).double()
        tp = cm.diag()
        fp, fn = cm.sum(0) - tp, cm.sum(1) - tp
        den = 2 * tp + fp + fn
        f1 = torch.where(den > 0, 2 * tp / den.clamp_min(1), torch.zeros_like(den))
        w = ((tp + fp + fn) > 0).double()
        return float((w * f1).sum() / w.sum()) if float(w.sum()) > 0 else 0.0

    def reset(self) -> None:
        self.confmat.zero_()


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
        # Lightning precision strings (src/config.py:35): "32-true" -> exact-f32 kernels, "bf16-*" -> bf16, "16-*" -> IEEE half
        dtype = (torch.float32 if conf.precision.startswith("32") else
                 torch.float16 if conf.precision.startswith("16") else torch.bfloat16)
        if dtype != torch.float32 and model is None:
            import warnings
            # measured against the reference-generated vectors (tests/test_gpu_parity.py)
            warnings.warn(f"precision={conf.precision!r}: the 16-bit operand formats are OUTSIDE the reference's 1e-3 tolerance where "
                          "attention is peaked, as in a trained checkpoint (ViT-L peaked fixture: float16 9.4e-3 prediction / 3.2e-2 "
                          "prompt gradient, bfloat16 7.2e-2 / 2.6e-1).  Only '32-true' (exact f32) and '32-x3' (float32 storage, three f16 "
                          "MFMAs per product, 2.4x faster) are inside it on every fixture.", stacklevel=2)
        # "32-x3": float32 storage / softmax / LayerNorm with every GEMM, attention and conv MFMA as three f16 MFMAs (22-bit operands)
        self.model = model if model is not None else ml_util.load_model(conf.checkpoint, device=device, dtype=dtype,
                                                                        gemm_x3="x3" in conf.precision)
        self.train_metrics = MulticlassF1(self.num_classes, self.nodata_idx, self.model.device)
        self.val_metrics = MulticlassF1(self.num_classes, self.nodata_idx, self.model.device)
        self.g = torch.Generator(device="cpu")  # src/model.py:98-99 (the draw itself is tiny; kept on the host)
        self.g.manual_seed(conf.seed)
        self.palette_g = torch.Generator(device="cpu").manual_seed(conf.seed + 1)
        self.loss_fn = SegGptLoss(conf.loss_beta, conf.loss_variant)
        self.normalize, self.denormalize = ml_util.normalize, ml_util.denormalize
        self.train_aug = self.aug = None  # set by post_init (src/model.py:104-108); None = Normalize only
        self.current_epoch = 0
        self._prompt_stack = None

    # ---- src/model.py:104-113
    def post_init(self, datamodule) -> None:
        self.train_aug, self.aug = datamodule.train_aug, datamodule.aug
        self.normalize, self.denormalize = datamodule.normalize, datamodule.denormalize

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
        self.invalidate_prompt_cache()

    def invalidate_prompt_cache(self) -> None:
        """Drop the no-grad path's stacked copy of the prompt Parameters.  `prepare_prompt` also detects in-place updates
        through the Parameters' version counters; call this after an edit those cannot see (`p.data = ...`, an external
        engine writing the storage through a raw pointer)."""
        self._prompt_stack = None

    # ---- src/model.py:215-231
    def _palette_host(self, batch_size: int, train: bool) -> torch.Tensor:
        if train:
            return ml_util.generate_random_rgb_palette(self.num_classes, batch_size, "cpu", self.palette_g)
        p = torch.tensor(ml_util.build_palette(self.num_classes - 1), dtype=torch.uint8)
        return torch.stack([p for _ in range(batch_size)])

    @staticmethod
    def _palette_norm(pal: torch.Tensor) -> torch.Tensor:
        """Normalised palette on the HOST in float32, `(c / 255 - mean) / std` with true divisions -- the bits the
        reference's CPU run holds (`src/model.py:221-229`).  (On the GPU torch evaluates `x / 255` as `x * (1 / 255)`, one
        ulp off on some entries; the arg-min decode and the id-based loss compare against these values.)"""
        mean, std = torch.tensor(ml_util.IMAGE_MEAN), torch.tensor(ml_util.IMAGE_STD)
        return (pal.cpu().to(torch.float32) / 255 - mean) / std

    def create_palette(self, batch_size: int, train: bool) -> tuple[torch.Tensor, torch.Tensor]:
        pal = self._palette_host(batch_size, train)
        return pal.to(self.device), self._palette_norm(pal).to(self.device)

    def draw_palette_rows(self, batch_sizes: list[int], train: bool) -> torch.Tensor:
        """u8 (sum(batch_sizes), K, 3) on the HOST: one `create_palette` draw per batch, in batch order (the generator
        sequence of the reference's per-batch `create_palette(B, train=True)`, `src/model.py:134`)."""
        if not batch_sizes:
            return torch.zeros(0, self.num_classes, 3, dtype=torch.uint8)
        return torch.cat([self._palette_host(n, train) for n in batch_sizes])

    def split_palette_rows(self, rows: torch.Tensor, batch_sizes: list[int]) -> list[tuple[torch.Tensor, torch.Tensor]]:
        """Host palette rows -> per-batch (palette u8, normalised palette f32) device views, uploaded ONCE."""
        if not batch_sizes:
            return []
        norm = self._palette_norm(rows).to(self.device)
        pal = rows.to(self.device)
        out, s = [], 0
        for n in batch_sizes:
            out.append((pal[s:s + n], norm[s:s + n]))
            s += n
        return out

    def create_palettes(self, batch_sizes: list[int], train: bool) -> list[tuple[torch.Tensor, torch.Tensor]]:
        """The palettes of a whole predict loop, drawn on the host in batch order (the same generator sequence as one
        `create_palette` per batch) and uploaded ONCE: a per-batch upload is a blocking copy that keeps the host from
        queueing batch i + 1 while batch i runs."""
        return self.split_palette_rows(self.draw_palette_rows(batch_sizes, train), batch_sizes)

    # ---- src/model.py:177-213
    def prepare_prompt(self, batch_idxes, batch_palette: torch.Tensor, train: bool):
        on_device = isinstance(batch_idxes, torch.Tensor) and batch_idxes.is_cuda and not torch.is_grad_enabled()
        if on_device:  # predict loop: the indices already live on the device -- no host list, no upload, no sync
            idx, sel = None, batch_idxes.flatten().long()
        else:
            idx = batch_idxes.flatten().tolist() if isinstance(batch_idxes, torch.Tensor) else (
                [batch_idxes] if isinstance(batch_idxes, int) else list(batch_idxes))
            sel = torch.tensor(idx, device=self.device)
        if torch.is_grad_enabled():
            image = torch.stack([self.prompt_batch["image"][i] for i in idx], dim=0)  # autograd-tracked stack
        else:  # inference (predict loop): one gather from a cached stack instead of B Parameter reads
            # keyed on the list AND on every Parameter's version counter: optimizer.step() / .copy_() / .data edits bump
            # `_version` in place without changing the list's identity, and validation after training must see them
            params = self.prompt_batch["image"]
            key = (id(params), tuple(id(p) for p in params), tuple(p._version for p in params))
            if self._prompt_stack is None or self._prompt_stack[0] != key:
                self._prompt_stack = (key, torch.stack([p.detach() for p in params]))
            image = self._prompt_stack[1].index_select(0, sel)
        mask = self.prompt_batch["mask"][sel]
        aug = self.train_aug if train else self.aug
        if aug is not None:  # src/model.py:204-207: flips / erasing / noise / Normalize (train) or Normalize (eval)
            out = aug({"image": image, "mask": mask})
            image_n, mask = out["image"], out["mask"]
        else:
            image_n = self.normalize(image)
        crop_idx = (self.prompt_batch["crop_idx"].to(self.device)[sel] if on_device
                    else self.prompt_batch["crop_idx"][torch.tensor(idx)])
        prompt_batch = {"image": image_n, "mask": mask, "crop_idx": crop_idx}
        prompt_color = ops.mask_rgb_norm(batch_palette, mask)  # normalize(torch_apply_mask_rgb(...)), src/model.py:211-212
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
        prompt_batch, prompt_masks = self.prepare_prompt(prompt_idx, pal, train=train)
        # The reference colourises the label (`src/model.py:238-239`) and hands it to the network as `labels=`, where the
        # default bool_masked_pos drops it (HF:706-715); its only consumer is the loss, which here reads the class ids and
        # the normalised palette directly (`bsg_loss_fwd_bwd_ids`): no f32 label image exists on this path.
        out = self.model(pixel_values=batch["image"].to(self.device), labels=None,
                         prompt_pixel_values=prompt_batch["image"], prompt_masks=prompt_masks, embedding_type="instance")
        pred_masks = self.process_pred_masks(out.pred_masks, pal_norm)
        loss = self.loss_fn.forward_ids(out.pred_masks, mask, pal_norm)
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
        if getattr(self.model, "last_backward_overflowed", lambda: False)():  # f16 dgrad chain overflowed: skip, like GradScaler
            optimizer.zero_grad(set_to_none=True)
            return loss.detach()
        optimizer.step()
        self.invalidate_prompt_cache()
        return loss.detach()

    def on_epoch_end(self, optimizer: torch.optim.Optimizer) -> None:
        self.current_epoch += 1
        for gq in optimizer.param_groups:
            gq["lr"] = lr_at_epoch(self.conf, self.current_epoch)
