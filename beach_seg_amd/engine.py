"""Fused training step of the hot path, data-parallel over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI).  What is trained are the P prompt
images (`/root/reference/src/model.py:115-130`); the frozen network is replicated.  A step on one rank is
  gather prompts + Normalize -> SegGPT forward -> SegGptLoss -> dgrad to the prompt pixels -> scatter into the
  dense (P, 3*h*w) gradient buffer -> [all-reduce] -> AdamW on the touched prompts
which is `training_step` (`src/model.py:233-269`) + Lightning's `loss.backward()` + `optimizer.step()`
(`src/model.py:398`), minus the kornia random augmentations.  Everything is enqueued on
the current stream with no host synchronisation.

The path shards by sample: each rank takes its own B tiles (weak scaling).  The only exchange is ONE sum
all-reduce per step of the flat prompt-gradient buffer plus its P "touched" flags and one overflow flag (P x 602,112 fp32; 154 MB at
P = 64), the DDP-equivalent of the gradients the reference would reduce; gradients are averaged over ranks as
DDP does.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops
from .seggpt import SegGptNative


def reduce_prompt_grads(flat: torch.Tensor, process_group=None) -> None:
    """The ONE data-path collective of a training step: sum over ranks of [P x n gradient rows | P touched flags].
    `nccl` (= RCCL over xGMI) on GPUs; `gloo` in the CPU tests."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        if flat.is_cuda and dist.get_backend(process_group) == "gloo":  # one-GPU rehearsal / tests: gloo moves host memory
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=process_group)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)


def broadcast_from_rank0(t: torch.Tensor, process_group=None) -> None:
    """Rank 0's copy of `t` to every rank (gloo needs host memory: staged through the CPU there)."""
    if dist.get_backend(process_group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src=0, group=process_group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=0, group=process_group)


def reduce_metrics(loss_sum: torch.Tensor, n_steps: int, confmat: torch.Tensor, process_group=None
                   ) -> tuple[torch.Tensor, torch.Tensor]:
    """The second, tiny collective of SURVEY.md section 8(e): what Lightning's `sync_dist=True` does for the logged loss and
    the torchmetrics state (`src/model.py:316, 327`).  One SUM all-reduce of [loss sum, step count, K*K confusion counts]
    (float64: exact for counts < 2^53) -> (mean loss over all ranks' steps, global confusion matrix)."""
    K = confmat.shape[0]
    buf = torch.cat([loss_sum.detach().double().reshape(1), torch.tensor([float(n_steps)], dtype=torch.float64, device=loss_sum.device),
                     confmat.double().flatten()])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        if buf.is_cuda and dist.get_backend(process_group) == "gloo":
            h = buf.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=process_group)
            buf.copy_(h)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=process_group)
    return buf[0] / buf[1].clamp_min(1), buf[2:].round().long().reshape(K, K)


def shard_batch(global_batch: int, rank: int, world: int) -> range:
    """Contiguous slice of the global batch owned by `rank` (SURVEY.md section 8 e)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return range(rank * per, (rank + 1) * per)


class PromptTrainEngine:
    def __init__(self, model: SegGptNative, prompt_images: torch.Tensor, lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 1e-2, loss_beta: float = 0.01,
                 loss_variant: str = "reference", process_group=None):
        """prompt_images: f32 (P,3,h,w) in [0,1] (the `nn.Parameter`s of `src/model.py:121-126`)."""
        self.model = model
        dev = model.device
        self.params = prompt_images.detach().to(dev, torch.float32).contiguous().clone()
        P = self.params.shape[0]
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.steps = torch.zeros(P, dtype=torch.int64, device=dev)  # one `step` per Parameter, as torch.optim
        self.rows = torch.arange(P, dtype=torch.int32, device=dev)
        # trailing P floats of the reduce buffer carry the "touched" flags so that ONE collective moves both
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.loss_beta, self.loss_variant = loss_beta, loss_variant
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        if self.world > 1:  # replicas must start from the same parameters, as DDP guarantees at construction
            broadcast_from_rank0(self.params, process_group)
        n = self.params[0].numel()
        # ... and ONE more float: the f16 overflow flag of this rank's backward (any rank overflowing skips the step on all)
        self._flat = torch.zeros(P * n + P + 1, dtype=torch.float32, device=dev)
        self.grads = self._flat[: P * n].view_as(self.params)
        self._touched_f = self._flat[P * n: P * n + P]
        self._overflow_f = self._flat[P * n + P:]
        self.skipped_steps = torch.zeros((), dtype=torch.int64, device=dev)  # steps dropped by the overflow guard
        self._last_batch = 0

    def step(self, pixel_values: torch.Tensor, label_color: torch.Tensor, yesdata: torch.Tensor,
             prompt_idx: torch.Tensor, prompt_mask_color: torch.Tensor) -> torch.Tensor:
        """One optimiser step on this rank's batch; returns this rank's loss (device scalar, no sync)."""
        return self._step(pixel_values, prompt_idx, prompt_mask_color,
                          lambda pred: ops.loss_fwd_bwd(pred, label_color, yesdata, self.loss_beta, self.loss_variant, True))

    def step_ids(self, pixel_values: torch.Tensor, label_ids: torch.Tensor, palette: torch.Tensor,
                 palette_norm: torch.Tensor, prompt_idx: torch.Tensor, prompt_mask_ids: torch.Tensor) -> torch.Tensor:
        """The same step from what `training_step` actually holds (`src/model.py:233-255`): label class ids u8 (B,[1,]h,w),
        the batch palette u8 (B,K,3) + its normalised copy f32 (B,K,3), the class ids of the chosen prompts' masks.  The
        prompt mask is colourised + normalised by one kernel (`bsg_mask_rgb_norm`); the label image is never formed -- the
        loss kernel reads class id and palette (`bsg_loss_fwd_bwd_ids`), yesdata = (id != 0)."""
        prompt_mask_color = ops.mask_rgb_norm(palette, prompt_mask_ids)
        return self._step(pixel_values, prompt_idx, prompt_mask_color,
                          lambda pred: ops.loss_fwd_bwd_ids(pred, label_ids, palette_norm, self.loss_beta, self.loss_variant, True))

    def _step(self, pixel_values, prompt_idx, prompt_mask_color, loss_and_grad) -> torch.Tensor:
        m = self.model
        B = self._last_batch = pixel_values.shape[0]
        self._flat.zero_()
        prompts = ops.prompt_gather(self.params, prompt_idx)  # stack + Normalize
        half = m.geometry.image_size[0] // 2
        # SegGptLoss reads the query half of the prediction (`src/model.py:53-57`): the decoder runs over those rows only
        pred = m._run_forward(pixel_values, prompts, prompt_mask_color, 0, train=True, first_row=half)
        loss, gpred = loss_and_grad(pred)
        gpix = m._run_backward(gpred, B, first_row=half)  # ... and the loss gradient is zero on the prompt half
        ops.prompt_grad_scatter(gpix, prompt_idx, self.grads)
        self._touched_f.index_fill_(0, prompt_idx.long(), 1.0)
        if m.dtype == torch.float16 or getattr(m, "gemm_x3", False):  # device-side overflow guard of the scaled dgrad chain (no host sync)
            self._overflow_f.copy_(m.grad_overflow_state(B)[:1])
        reduce_prompt_grads(self._flat, self.pg)  # RCCL over xGMI when world > 1
        ok = self._overflow_f == 0  # an overflow on ANY rank drops the step on every rank: replicas stay identical
        self.skipped_steps += (~ok).sum()
        touched = ((self._touched_f > 0) & ok).to(torch.uint8)
        self.steps += touched.long()
        ops.adamw_step(self.params.view(self.params.shape[0], -1), self.grads.view(self.params.shape[0], -1),
                       self.exp_avg.view(self.params.shape[0], -1), self.exp_avg_sq.view(self.params.shape[0], -1),
                       self.rows, self.steps.clamp_min(1), self.lr, self.betas, self.eps, self.weight_decay,
                       grad_scale=1.0 / self.world, touched=touched)
        return loss.detach()

    def overflow_metrics(self) -> dict:
        """State of the f16 / x3 overflow guard for logging (ONE host synchronisation; call it per epoch, not per step):
        steps dropped on any rank, this rank's back-off exponent (the dgrad chain runs 2^-backoff below its target scale),
        true dgrad overflows, and backwards dropped because the incoming gradient itself was non-finite (bad batch)."""
        m = self.model
        out = {"skipped_steps": int(self.skipped_steps), "backoff_exp": 0, "dgrad_overflows": 0, "nonfinite_input_steps": 0}
        if (m.dtype == torch.float16 or getattr(m, "gemm_x3", False)) and self._last_batch and m._last_ws is not None:
            st = m.grad_overflow_state(self._last_batch, m._last_ws).tolist()
            out.update(backoff_exp=st[1], dgrad_overflows=st[3], nonfinite_input_steps=st[5])
        return out
