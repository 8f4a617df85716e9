"""Predict loop of `/root/reference/src/predict.py:232-262` with the vote mosaic kept on the GPU.

`Accumulator.update` (`:120-159`) = clip the crop window to the mosaic and add a one-hot vote into uint8
counters; `save_current` (`:100`) = arg-max over classes.  Here the nearest-neighbour down-size of the decoded
mask (`:259`), the one-hot (`:260`) and the paste run as one HIP kernel per batch of crops, the arg-max as another;
PNG / GeoTIFF export (`:93-112`) is out of scope.  Integer semantics (uint8 wrap at 256 votes, first-index
arg-max) match the numpy reference bit for bit (`tests/test_gpu_parity.py::test_predict_vote_glue_bit_exact`).
"""
from __future__ import annotations

import torch

from . import ops
from .model import PromptModel


class Accumulator:
    """Device-side vote mosaic with the life cycle of the reference class (`src/predict.py:55-159`): a change of date
    finalises the previous mosaic first (`save_current`, `:129-132`), and leaving the `with` block finalises the last
    one (`__exit__`, `:90-91`).  "Finalise" here = arg-max over the votes (`:100`) handed to `on_finish(date, mosaic)`
    and kept in `finished`; the PNG / GeoTIFF export of `:93-112` is the caller's (out of scope)."""

    def __init__(self, out_shape: tuple[int, int], classes: tuple[str, ...], device, on_finish=None, world: int = 1,
                 process_group=None):
        """`world` > 1 (sharded predict): every finalisation -- date change, `__exit__`, `save_current` -- first sums the
        ranks' vote counters, so each rank finalises the FULL mosaic of that date; all ranks must then see the same date
        sequence (a rank without windows for a date still calls `update` with zero crops, or `initialize_current`)."""
        self.out_shape, self.num_classes, self.classes, self.device = out_shape, len(classes), classes, device
        self.world, self.process_group = world, process_group
        self.current_date = None
        self.current_pred_counter = None
        self._reduced = False  # world > 1: the current counter already holds the SUM over ranks
        self.on_finish = on_finish
        self.finished: list[tuple[str, torch.Tensor]] = []

    def __enter__(self):
        assert self.current_pred_counter is None and self.current_date is None
        return self

    def __exit__(self, a, b, c):
        self.save_current()

    def save_current(self) -> torch.Tensor:
        assert self.current_pred_counter is not None and self.current_date is not None
        if self.world > 1 and not self._reduced:
            self.reduce_votes(self.process_group)
        pred = ops.vote_argmax(self.current_pred_counter)
        self.finished.append((self.current_date, pred))
        if self.on_finish is not None:
            self.on_finish(self.current_date, pred)
        return pred

    def initialize_current(self, date: str) -> None:
        self.current_date = date
        self._reduced = False
        self.current_pred_counter = torch.zeros((*self.out_shape, self.num_classes), dtype=torch.uint8, device=self.device)

    def update(self, date: str, crops: torch.Tensor, masks: torch.Tensor, crop_size: int,
               disjoint: bool = False) -> None:
        """crops i32 (n,4) (xmin,ymin,xmax,ymax); masks u8 (n,hin,win) decoded at network resolution.  `disjoint`:
        the caller guarantees that the windows of this call do not overlap each other (a regular grid with stride >=
        crop size): one launch, no host round trip.  Otherwise the windows are grouped on the host once per call."""
        if date != self.current_date:
            if self.current_pred_counter is not None:
                self.save_current()
            self.initialize_current(date)
        if self.world > 1 and self._reduced:
            # the counter holds the ranks' SUM: local votes on top of it would diverge between ranks and never be re-synchronised
            raise RuntimeError(f"Accumulator.update for date {date!r} after its votes were reduced across ranks "
                               "(result() / save_current() / reduce_votes()); finalise a date only after its last update")
        if disjoint:
            ops.vote_paste(self.current_pred_counter, masks, crops.to(self.device), crop_size)
            return
        # votes are plain uint8 read-modify-writes: windows that overlap go in separate launches
        for group in _non_overlapping_groups(crops):
            ops.vote_paste(self.current_pred_counter, masks[group], crops[group].to(self.device), crop_size)

    def reduce_votes(self, process_group=None) -> None:
        """Multi-GPU predict (SURVEY.md section 8 e: windows sharded by index, replicas of the network): sum the ranks'
        vote counters once, before the arg-max.  uint8 addition wraps at 256 on every backend, exactly like the single
        process's `+= 1` (`src/predict.py:150-159`), so the reduced mosaic is bit-identical to the unsharded one."""
        reduce_vote_counters(self.current_pred_counter, process_group)
        self._reduced = True

    def result(self) -> torch.Tensor:
        if self.world > 1 and not self._reduced:
            self.reduce_votes(self.process_group)
        return ops.vote_argmax(self.current_pred_counter)


def reduce_vote_counters(counter: torch.Tensor, process_group=None) -> torch.Tensor:
    """In-place SUM all-reduce of a u8 (H,W,K) vote counter over the ranks (modulo 256).  RCCL sums uint8 natively; under
    gloo a device tensor is staged through the host (gloo cannot reduce device memory of a ROCm build)."""
    import torch.distributed as dist

    if counter.dtype != torch.uint8:
        raise ValueError("vote counters are uint8")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return counter
    if counter.is_cuda and dist.get_backend(process_group) == "gloo":
        h = counter.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=process_group)
        counter.copy_(h)
    else:
        dist.all_reduce(counter, op=dist.ReduceOp.SUM, group=process_group)
    return counter


def crops_are_disjoint(crops: torch.Tensor) -> bool:
    """True when no two windows of the (n,4) list overlap: decided ONCE per mosaic on the host (sort by the grid
    origin; a regular grid with stride >= window side is disjoint by construction)."""
    b = crops.cpu().numpy().astype("int64")
    if len(b) < 2:
        return True
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    if (w != w[0]).any() or (h != h[0]).any():
        return len(_non_overlapping_groups(crops)) == 1
    # equal-size windows: they are disjoint iff all distinct origin pairs differ by >= the side in x or in y
    import numpy as np
    xs, ys = np.unique(b[:, 0]), np.unique(b[:, 1])
    on_lattice = len(b) == len(np.unique(b[:, :2], axis=0))
    if on_lattice and (len(xs) < 2 or np.diff(xs).min() >= w[0]) and (len(ys) < 2 or np.diff(ys).min() >= h[0]):
        return True
    return len(_non_overlapping_groups(crops)) == 1


def _non_overlapping_groups(crops: torch.Tensor) -> list[torch.Tensor]:
    boxes = crops.cpu().tolist()
    groups: list[list[int]] = []
    for i, (x0, y0, x1, y1) in enumerate(boxes):
        for g in groups:
            if all(x1 <= boxes[j][0] or boxes[j][2] <= x0 or y1 <= boxes[j][1] or boxes[j][3] <= y0 for j in g):
                g.append(i)
                break
        else:
            groups.append([i])
    return [torch.tensor(g, dtype=torch.long, device=crops.device) for g in groups]


@torch.no_grad()
def predict_mosaic(model: PromptModel, images: torch.Tensor, crop_idx: torch.Tensor, crops: torch.Tensor,
                   out_shape: tuple[int, int], crop_size: int, batch_size: int = 64, date: str = "d0",
                   use_graph: bool = False, rank: int = 0, world: int = 1, process_group=None) -> torch.Tensor:
    """Sliding-window inference (BASELINE config 4): images f32 (n,3,S,S) normalised, one prompt per crop_idx.
    Returns the u8 (H,W) class mosaic.  All-nodata crops are the caller's to skip (`src/predict.py:235`).
    `use_graph`: replay the network forward from one captured hipGraph (full batches only; the tail runs eagerly).
    `world` > 1: this rank takes windows rank, rank + world, ... (every rank is a replica of the network), the vote
    counters are summed once at the end and every rank returns the full mosaic.  The reference draws a RANDOM palette per
    batch (`src/model.py:134`); here every rank draws the palettes of the UNSHARDED loop (same generator, same batch
    sizes) and window i keeps row i of that sequence wherever it runs, so the sharded mosaic is bit-identical to the
    world = 1 run of the same `model.palette_g` state (identical seeds on all ranks assumed, as `PromptModel` sets them)."""
    acc = Accumulator(out_shape, model.conf.classes, model.device, world=world, process_group=process_group)
    disjoint = crops_are_disjoint(crops)  # decided once per mosaic: no host round trip per batch of windows
    n_global = images.shape[0]
    pal_rows = model.draw_palette_rows([min(batch_size, n_global - s) for s in range(0, n_global, batch_size)], train=True)
    if world > 1:
        images, crop_idx, crops = images[rank::world], crop_idx[rank::world], crops[rank::world]
        pal_rows = pal_rows[rank::world]
    acc.initialize_current(date)  # a rank without windows still joins the reduction with an all-zero counter
    # everything the loop needs from the host goes up ONCE: a per-batch upload blocks the host until the stream has
    # drained, i.e. until the previous batch's forward is done, and the GPU then idles through the host's share of a batch
    crops_dev = crops.to(model.device)
    idx_dev = crop_idx.to(model.device)
    n_total = images.shape[0]
    sizes = [min(batch_size, n_total - s) for s in range(0, n_total, batch_size)]
    palettes = model.split_palette_rows(pal_rows, sizes)  # uploaded once
    graphed = model.model.capture_forward(batch_size) if use_graph and n_total >= batch_size else None
    with torch.no_grad():
        for b, s in enumerate(range(0, n_total, batch_size)):
            sl = slice(s, s + batch_size)
            pal, pal_norm = palettes[b]
            prompt_batch, prompt_masks = model.prepare_prompt(idx_dev[sl], pal, train=False)
            img = images[sl].to(model.device)
            if graphed is not None and sizes[b] == batch_size:
                out = graphed(img, prompt_batch["image"], prompt_masks)
            else:
                out = model.model(pixel_values=img, prompt_pixel_values=prompt_batch["image"], prompt_masks=prompt_masks,
                                  embedding_type="instance").pred_masks
            pred = model.process_pred_masks(out, pal_norm)
            acc.update(date, crops_dev[sl] if disjoint else crops[sl], pred.to(torch.uint8), crop_size, disjoint=disjoint)
    return acc.result()  # world > 1: sums the ranks' counters first


def grid_crops(height: int, width: int, crop_size: int, stride: int | None = None) -> torch.Tensor:
    """Regular sliding-window grid (xmin,ymin,xmax,ymax); the last row / column may stick out of the mosaic and is
    clipped by the vote paste exactly like `Accumulator.update` (`src/predict.py:138-159`)."""
    stride = stride or crop_size
    xs = list(range(0, width, stride))
    ys = list(range(0, height, stride))
    return torch.tensor([[x, y, x + crop_size, y + crop_size] for y in ys for x in xs], dtype=torch.int32)


@torch.no_grad()
def ensemble_predict(net, pixel_values: torch.Tensor, prompt_pixel_values: torch.Tensor, prompt_masks: torch.Tensor,
                     num_classes: int, crop_size: int) -> torch.Tensor:
    """One crop of the few-shot loop of `/root/reference/src/predict_no_prompt.py:283-304`: the query repeated once per
    prompt (K, 3, S, S), `feature_ensemble=True` forward, mean of `pred_masks` over the K prompts, HF post-process decode
    to (crop_size, crop_size) with `num_labels = num_classes - 1`.  `net` is the object `load_model` returns
    (`SegGptNative`).  Returns the int64 class map on the device; nodata masking and the vote are the caller's
    (`:302-304`, `Accumulator.update`)."""
    out = net(pixel_values=pixel_values, prompt_pixel_values=prompt_pixel_values, prompt_masks=prompt_masks,
              embedding_type="instance", feature_ensemble=True)
    pred = out.pred_masks.mean(dim=0).unsqueeze(0)
    return ops.post_process_semantic_segmentation(pred, num_classes - 1, [(crop_size, crop_size)])[0]
