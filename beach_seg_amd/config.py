"""`BeachSegConfig`: the reference's config surface (`/root/reference/src/config.py:7-98`), field for field, so a
saved `conf.yaml` / dotlist overrides keep working.  Only the defaults that pointed at the author's laptop
(`config.py:19-20`) are neutralised."""
from __future__ import annotations

import os
from dataclasses import dataclass, fields
from pathlib import Path

CLASSES = ("nodata", "sand", "water", "veg")  # src/config.py:7-12


@dataclass
class BeachSegConfig:
    project: str = "beach_seg"
    seed: int = 42
    data: Path = Path("data")
    model_training_root: Path = Path("results")
    classes: tuple[str, ...] = CLASSES
    devices: tuple[str, ...] = ("auto",)
    accelerator: str = "auto"
    deterministic: bool = False
    num_viz_images: int = 9
    viz_size: int = 224

    epochs: int = 1
    debug: bool = False
    world_size: int = 1
    grad_accum_steps: int = 1
    log_every_n_steps: int = 10
    precision: str = "32-true"  # "32-true" -> exact-f32 kernels, "bf16-true" / "bf16-mixed" -> bf16, "16-true" / "16-mixed" -> IEEE half
    workers: int = -1
    batch_size: int = 1

    checkpoint: str = "BAAI/seggpt-vit-large"  # or a local state-dict path, or "synthetic:<geometry>[:seed]"

    monitor_metric: str = "val/f1"
    monitor_mode: str = "max"

    crop_size: int = 112
    inpt_size: int = 448
    resample: str = "BICUBIC"  # PIL.Image.Resampling name (src/config.py:47)

    horizontal_flip: float = 0.5
    vertical_flip: float = 0.5
    hue: float = 0.1
    saturation: float = 0.1
    contrast: float = 0.1
    brightness: float = 0.1
    scale: tuple[float, float] = (0.4, 1.0)
    sharpness: float = 1.0
    sharpness_p: float = 0.2
    erasing_scale: tuple[float, float] = (0.02, 0.05)
    erasing_p: float = 0.1
    gauss_mean: float = 0.0
    gauss_std: float = 0.1
    gauss_p: float = 0.1
    channel_shift_limit: float = 0.01
    channel_shift_p: float = 0.2
    mosaic_p: float = 0.0
    jigsaw_grid: tuple[int, int] = (2, 2)
    jigsaw_p: float = 0.0

    lr: float = 1e-3
    loss_beta: float = 0.01
    base_lr_batch_size: int = 1
    warmup_epochs: int = 0
    init_lr: float = 5e-04
    min_lr: float = 5e-04
    optimizer: str = "adamw"
    scheduler: str = "cosine"
    ema_alpha = 0.99

    # --- additions of this build (not in the reference) ---
    loss_variant: str = "reference"  # "reference" keeps the B x B broadcast of src/model.py:61

    @classmethod
    def from_dotlist(cls, dotlist: list[str]) -> "BeachSegConfig":
        """`OmegaConf.from_cli()`-style `key=value` overrides (`src/train.py:31-36`)."""
        conf = cls()
        types = {f.name: f.type for f in fields(cls)}
        for item in dotlist:
            k, _, v = item.partition("=")
            if k not in types:
                raise KeyError(f"unknown config key {k!r}")
            cur = getattr(conf, k)
            if isinstance(cur, bool):
                val = v.lower() in ("1", "true", "yes")
            elif isinstance(cur, int):
                val = int(v)
            elif isinstance(cur, float):
                val = float(v)
            elif isinstance(cur, Path):
                val = Path(v)
            elif isinstance(cur, tuple):
                val = tuple(type(cur[0])(x) for x in v.strip("()[]").split(",") if x)
            else:
                val = v
            setattr(conf, k, val)
        return conf


def cpu_count() -> int:
    return os.cpu_count() or 0


def num_workers(conf: BeachSegConfig) -> int:
    """`src/config.py:81-91`."""
    per_gpu = cpu_count() // max(1, conf.world_size)
    return per_gpu if conf.workers == -1 else min(per_gpu, conf.workers)
