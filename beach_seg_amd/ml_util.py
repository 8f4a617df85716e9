"""Mirror of the on-path part of `/root/reference/src/util/ml_util.py`: `load_model` (:7-13), the processor
constants (:16-17), `build_palette` (:72-89), `generate_random_rgb_palette` (:99-111), `torch_apply_mask_rgb`
(:114-132).  Same names, argument meaning and return types; `load_model` returns the HIP-backed `SegGptNative`."""
from __future__ import annotations

from pathlib import Path

import torch

from .seggpt import SegGptNative
from .weights import SegGptGeometry, synth_state_dict

IMAGE_MEAN = (0.485, 0.456, 0.406)  # SegGptImageProcessor defaults read at src/data.py:192-193
IMAGE_STD = (0.229, 0.224, 0.225)


def load_state_dict(checkpoint: str, geometry: SegGptGeometry, device="cpu") -> dict:
    """`checkpoint`: "synthetic:<geometry>[:seed]" or a local file (.safetensors, or a torch file loaded with
    weights_only=True).  The hub name of the reference (`BAAI/seggpt-vit-large`) needs network access."""
    if checkpoint.startswith("synthetic"):
        parts = checkpoint.split(":")
        seed = int(parts[2]) if len(parts) > 2 else 0
        return synth_state_dict(geometry, seed=seed, device=device)
    p = Path(checkpoint)
    if not p.exists():
        raise FileNotFoundError(f"{checkpoint!r} is not a local file; hub downloads are not available here -- "
                                "pass a local state dict or 'synthetic:<geometry>[:seed]'")
    if p.suffix == ".safetensors":
        from safetensors.torch import load_file

        return load_file(str(p), device=str(device))
    return torch.load(str(p), map_location=device, weights_only=True)


def geometry_of(checkpoint: str) -> SegGptGeometry:
    if checkpoint.startswith("synthetic"):
        parts = checkpoint.split(":")
        return getattr(SegGptGeometry, parts[1] if len(parts) > 1 and parts[1] else "vit_large")()
    return SegGptGeometry.vit_large()


def load_model(checkpoint: str, device="cuda:0", dtype=torch.bfloat16, geometry: SegGptGeometry | None = None,
               gemm_x3: bool = False) -> SegGptNative:
    """`src/util/ml_util.py:7-13`: build the net, freeze it, eval mode.  (The reference's `torch.compile` has no
    counterpart: the kernels are already fused.)  `gemm_x3` (float32 only): the Linear GEMMs as three f16 MFMAs on 22-bit
    operand splits instead of exact-f32 MFMAs (`SegGptNative`)."""
    g = geometry or geometry_of(checkpoint)
    return SegGptNative(load_state_dict(checkpoint, g, device), g, device=device, dtype=dtype, gemm_x3=gemm_x3).eval()


def build_palette(num_labels: int) -> list[tuple[int, int, int]]:
    """`src/util/ml_util.py:72-89` (Painter's colour coding; class 0 -> black)."""
    base = int(num_labels ** (1 / 3)) + 1
    margin = 256 // base
    color_list = [(0, 0, 0)]
    for location in range(num_labels):
        r = 255 - (location // base**2) * margin
        g = 255 - ((location % base**2) // base) * margin
        b = 255 - (location % base) * margin
        color_list.append((r, g, b))
    return color_list


def generate_random_rgb_palette(num_labels: int, batch_size: int, device, generator=None) -> torch.Tensor:
    """`src/util/ml_util.py:99-111`: uint8 (B, N, 3), class 0 black.  The reference draws from the global RNG;
    `generator` makes the draw reproducible."""
    lut = torch.randint(0, 256, (batch_size, num_labels, 3), dtype=torch.uint8, device=device, generator=generator)
    lut[:, 0] = 0
    return lut


def torch_apply_mask_rgb(palette: torch.Tensor, input: torch.Tensor) -> torch.Tensor:
    """`src/util/ml_util.py:114-132`: class ids (B,1,H,W)/(B,H,W) -> f32 (B,3,H,W) in [0,1], on the HIP kernel
    (`bsg_mask_rgb_norm` with mean 0 / std 1: palette / 255 exactly).  The hot path uses the fused form
    `ops.mask_rgb_norm` (LUT gather + Normalize in one pass) or skips the image altogether (`ops.seggpt_loss_ids`)."""
    from . import ops

    return ops.mask_rgb_norm(palette, input, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0))


_CONST_CACHE: dict = {}


def mean_std(device, dtype=torch.float32) -> tuple[torch.Tensor, torch.Tensor]:
    """ImageNet mean / std as (1,3,1,1) tensors, cached per (device, dtype): building them with `torch.tensor(...,
    device=cuda)` is a pageable host-to-device copy that blocks the host until the stream has drained -- once per call
    in the predict loop, that serialised the host behind every batch's forward."""
    key = (str(device), dtype)
    if key not in _CONST_CACHE:
        _CONST_CACHE[key] = (torch.tensor(IMAGE_MEAN, dtype=dtype, device=device).view(1, 3, 1, 1),
                             torch.tensor(IMAGE_STD, dtype=dtype, device=device).view(1, 3, 1, 1))
    return _CONST_CACHE[key]


def normalize(x: torch.Tensor) -> torch.Tensor:
    """`BeachSegDataModule.normalize` (`src/data.py:345-346`)."""
    mean, std = mean_std(x.device, x.dtype)
    return (x - mean) / std


def denormalize(x: torch.Tensor) -> torch.Tensor:
    """`src/data.py:342-343`."""
    mean, std = mean_std(x.device, x.dtype)
    return x * std + mean
