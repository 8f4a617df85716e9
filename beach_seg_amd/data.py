"""Tile-loader surface of `/root/reference/src/data.py` on synthetic / in-memory rasters.

Kept: the `BeachSegDataset.__getitem__` dictionary `{crop_idx, date, image f32(3,S,S) in [0,1], mask u8(S,S),
nodata bool(S,S)}` (`src/data.py:118-124`), the resize rules (`:93-113`: PIL bicubic for the image, nearest for
mask / nodata), the "hack" that marks valid pixels as class 1 when no label exists (`:115-116`), and the
DataModule attributes the model reads (`prompt_imgs`, `normalize`, `denormalize`, `aug`, `train_aug`).
Replaced: GeoTIFF / shapefile IO (rasterio, geopandas: absent here) by in-memory
arrays; the 4-band -> RGB collapse follows `tif_image` (`src/util/geo_util.py:449-470`).
"""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image

from . import ml_util
from .config import BeachSegConfig


def tif_image(bands: np.ndarray, nodata: np.ndarray | None = None) -> np.ndarray:
    """4-band (B,G,R,NIR) or 8-band surface-reflectance raster (C,H,W) -> uint8 (H,W,3): the display image of
    `src/util/geo_util.py:449-470` / `src/util/multichannel_img.py:7-29`, written channel-vectorised.
      4 bands: channels (band 4, band 3, mean(band 1, band 2)); ONE floor = their minimum over the valid pixels; values
               clipped to [floor, floor + 3000] and shifted to 0; each channel divided by its own maximum.
      8 bands: log10(1 + mean) of the band groups (6-8, 3-5, 1-2); each channel shifted by its minimum over the valid
               pixels and divided by its maximum.
    nodata pixels come out 0; the float image is truncated to uint8 after x255.  The reference reads rasters as float32
    (`geo_util.py:385`); integer rasters are promoted to float32.  Host twin of `ops.tif_image` (the device kernel)."""
    x = np.asarray(bands)
    if not np.issubdtype(x.dtype, np.floating):
        x = x.astype(np.float32)
    hole = np.zeros(x.shape[1:], dtype=bool) if nodata is None else np.asarray(nodata, dtype=bool)
    seen = ~hole
    if x.shape[0] == 4:
        ch = np.stack((x[3], x[2], x[:2].mean(axis=0)))                       # (3,H,W)
        floor = ch[:, seen].min()
        ch = np.clip(ch, floor, 3000 + floor) - floor
        ch = ch - ch[:, seen].min()                                             # exact +0 whenever a valid pixel exists
    elif x.shape[0] == 8:
        ch = np.stack([np.log10(1.0 + grp.mean(axis=0)) for grp in (x[5:], x[2:5], x[:2])])
        ch = ch - ch[:, seen].min(axis=1)[:, None, None]
    else:
        raise ValueError(f"expected 4 or 8 bands, got {x.shape[0]}")
    ch = ch / ch.max(axis=(1, 2), keepdims=True)
    ch[:, hole] = 0
    return (np.moveaxis(ch, 0, -1) * 255).astype(np.uint8)


def sample_train_aug_params(batch: int, h: int, w: int, config: BeachSegConfig, generator: torch.Generator | None = None,
                            with_color: bool = False, erase_mask: bool = False):
    """Random parameters of the train-time augmentation chain (`src/data.py:195-224`), drawn on the host from an explicit
    generator (kornia draws them internally; kornia is not installable here, so the DISTRIBUTIONS follow kornia's
    documented parameter generators and the draw order is this function's own: "parity unpinned"):
      vertical / horizontal flip ~ Bernoulli(p);  erasing ~ Bernoulli(erasing_p) with area fraction ~ U(erasing_scale),
      log-uniform aspect ratio in (0.3, 3.3), box placed uniformly inside the image;  noise ~ Bernoulli(gauss_p);
      with_color: ColorJiggle (p = 1) brightness / contrast / saturation factors ~ U(max(0, 1 - a), 1 + a), hue factor ~
      U(-hue, hue) (turns), ONE random order of the four operations per batch;  RandomSharpness ~ Bernoulli(sharpness_p)
      with factor ~ U(0, sharpness).
    Returns (params i32 (B,5) = [flags, ex0, ey0, ew, eh], noise f32 (B,3,h,w) or None) and, with_color, a third item
    color f32 (B,6) = [brightness, contrast, saturation, hue, sharpness factor, order code] (flags bits 3 / 4 set).
    `erase_mask` (flags bit 5): the erased box also becomes class 0 (nodata) in the MASK, which drops it from the loss
    (`mask != 0`, `src/model.py:255`) -- what recent kornia versions do when `AugmentationSequential(data_keys=None)` routes
    the mask through `RandomErasing` (`src/data.py:195-214`); off by default: which of the two the reference's unpinned
    `kornia` did cannot be checked offline ("parity unpinned"), so both are built."""
    import math

    g = generator
    u = lambda *s: torch.rand(*s, generator=g)
    flags = (u(batch) < config.vertical_flip).int() | ((u(batch) < config.horizontal_flip).int() << 1)
    do_noise = u(batch) < config.gauss_p
    flags |= do_noise.int() << 2
    erase = u(batch) < config.erasing_p
    area = (config.erasing_scale[0] + u(batch) * (config.erasing_scale[1] - config.erasing_scale[0])) * h * w
    ratio = torch.exp(math.log(0.3) + u(batch) * (math.log(3.3) - math.log(0.3)))
    ew = torch.sqrt(area * ratio).round().clamp(1, w).int()
    eh = torch.sqrt(area / ratio).round().clamp(1, h).int()
    ex = (u(batch) * (w - ew + 1).float()).floor().int()
    ey = (u(batch) * (h - eh + 1).float()).floor().int()
    zero = torch.zeros_like(ew)
    noise = None
    if bool(do_noise.any()):
        noise = torch.randn(batch, 3, h, w, generator=g) * config.gauss_std + config.gauss_mean
    color = None
    if with_color:
        def around_one(a):
            lo, hi = max(0.0, 1.0 - a), 1.0 + a
            return lo + u(batch) * (hi - lo)

        order = torch.randperm(4, generator=g)
        code = float(sum(int(o) << (2 * k) for k, o in enumerate(order)))
        sharp_on = u(batch) < config.sharpness_p
        color = torch.stack([around_one(config.brightness), around_one(config.contrast), around_one(config.saturation),
                             (u(batch) * 2 - 1) * config.hue, u(batch) * config.sharpness, torch.full((batch,), code)],
                            dim=1).float().contiguous()
        flags |= (sharp_on.int() << 3) | 16
    if erase_mask:
        flags |= erase.int() << 5
    params = torch.stack([flags, torch.where(erase, ex, zero), torch.where(erase, ey, zero), torch.where(erase, ew, zero),
                          torch.where(erase, eh, zero)], dim=1).to(torch.int32).contiguous()
    return (params, noise, color) if with_color else (params, noise)


def pil_bicubic_tables(in_size: int, out_size: int) -> tuple[np.ndarray, np.ndarray]:
    """Coefficient tables of Pillow's BICUBIC resize of an 8-bit image from `in_size` to `out_size` pixels along one
    axis (libImaging/Resample.c: `precompute_coeffs` with the a = -0.5 cubic, support 2 x max(scale, 1), then
    `normalize_coeffs_8bpc`: 22-bit fixed point, round half away from zero), for the device front-end
    (`ops.tile_frontend` / `bsg_tile_frontend`): bounds i32 (out, 2) = (first source index, tap count), coef i32
    (out, kmax).  The reference resizes with `Image.resize(..., resample=BICUBIC)` (`src/data.py:93-96`)."""
    import math

    def cubic(x: float) -> float:
        a = -0.5
        x = -x if x < 0 else x
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0

    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    kmax = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, kmax), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    one = float(1 << 22)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - xmin
        w = [cubic((x + xmin - center + 0.5) * (1.0 / fscale)) for x in range(n)]
        ww = sum(w)  # accumulated left to right like the C loop
        for x in range(n):
            v = w[x] / ww if ww != 0.0 else w[x]
            coef[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx] = (xmin, n)
    return bounds, coef


def padded_crop(arr: np.ndarray, crop: tuple[int, int, int, int], fill=0) -> np.ndarray:
    """`src/util/geo_util.py:316-341`: window (xmin,ymin,xmax,ymax) that may stick out of the raster."""
    xmin, ymin, xmax, ymax = crop
    h, w = arr.shape[:2]
    out = np.full((ymax - ymin, xmax - xmin) + arr.shape[2:], fill, dtype=arr.dtype)
    dy0, dy1, dx0, dx1 = max(ymin, 0), min(ymax, h), max(xmin, 0), min(xmax, w)
    if dy1 > dy0 and dx1 > dx0:
        out[dy0 - ymin: dy1 - ymin, dx0 - xmin: dx1 - xmin] = arr[dy0:dy1, dx0:dx1]
    return out


class BeachSegDataset(torch.utils.data.Dataset):
    """`src/data.py:37-127` over in-memory rasters: date -> (rgb u8 (H,W,3), nodata bool (H,W)), optional masks."""

    def __init__(self, date_merged_imgs: dict, date_masks: dict | None, crops: list[tuple[int, int, int, int]],
                 config: BeachSegConfig, create_prompts: bool = False):
        self.date_merged_imgs, self.date_masks = date_merged_imgs, date_masks or {}
        self.crops, self.config = crops, config
        self.imgs = [{"date": d, "crop_idx": i} for d in date_merged_imgs for i in range(len(crops))]
        if create_prompts:
            self.prompt_imgs = [self.get_crop(x) for x in self.imgs]  # src/data.py:74-76

    def __len__(self):
        return len(self.imgs)

    def get_crop(self, img_data: dict) -> dict:
        c = self.config
        date, crop_idx = img_data["date"], img_data["crop_idx"]
        crop = self.crops[crop_idx]
        img, nodata = self.date_merged_imgs[date]
        label = self.date_masks.get(date)
        crop_img = padded_crop(img, crop)
        crop_nodata = padded_crop(nodata, crop, fill=True)
        crop_label = padded_crop(label, crop) if label is not None else np.zeros(crop_img.shape[:2], np.uint8)
        S = c.inpt_size
        if S != c.crop_size:
            res = getattr(Image.Resampling, c.resample) if isinstance(c.resample, str) else c.resample
            crop_img = np.array(Image.fromarray(crop_img).resize((S, S), resample=res))
            crop_label = np.array(Image.fromarray(crop_label).resize((S, S), resample=Image.Resampling.NEAREST))
            crop_nodata = np.array(Image.fromarray(crop_nodata).resize((S, S), resample=Image.Resampling.NEAREST))
        crop_img = crop_img.astype(np.float32) / 255.0
        if not np.all(crop_nodata) and np.all(crop_label == 0):  # src/data.py:115-116
            crop_label = crop_label.copy()
            crop_label[~crop_nodata] = 1
        return {"crop_idx": crop_idx, "date": date, "image": crop_img.transpose(2, 0, 1).copy(),
                "mask": crop_label, "nodata": crop_nodata}

    def __getitem__(self, idx):
        return self.get_crop(self.imgs[idx])


def synthetic_dove_scene(seed: int = 1234, size: int = 256, n_dates: int = 1) -> tuple[dict, dict]:
    """uint16 4-band tiles U[200, 3200), block-structured labels 1..3."""
    rng = np.random.default_rng(seed)
    imgs, masks = {}, {}
    for d in range(n_dates):
        bands = rng.integers(200, 3200, size=(4, size, size), dtype=np.uint16)
        lab = rng.integers(1, 4, size=(size // 32, size // 32), dtype=np.uint8).repeat(32, 0).repeat(32, 1)
        date = f"2025-01-{d + 1:02d}"
        imgs[date] = (tif_image(bands), np.zeros((size, size), bool))
        masks[date] = lab
    return imgs, masks


def synthetic_dove_bands(seed: int = 1234, size: int = 256) -> np.ndarray:
    """The raw uint16 (4, size, size) tile of `synthetic_dove_scene` (first date): the device front-end's input."""
    return np.random.default_rng(seed).integers(200, 3200, size=(4, size, size), dtype=np.uint16)


class BeachSegDataModule:
    """The attributes `PromptModel` / the drivers read from `src/data.py:181-346`."""

    def __init__(self, config: BeachSegConfig, scene: tuple[dict, dict] | None = None,
                 crops: list[tuple[int, int, int, int]] | None = None):
        self.config = config
        self.mean, self.std = ml_util.IMAGE_MEAN, ml_util.IMAGE_STD
        self.normalize, self.denormalize = ml_util.normalize, ml_util.denormalize
        self.aug_generator = torch.Generator().manual_seed(config.seed + 2)
        self.erase_mask = False  # True: RandomErasing also zeroes its box in the mask (see sample_train_aug_params)
        self.scene = scene or synthetic_dove_scene()
        size = next(iter(self.scene[0].values()))[0].shape[0]
        cs = config.crop_size
        self.crops = crops or [(x, y, x + cs, y + cs) for y in range(0, size, cs) for x in range(0, size, cs)]

    def aug(self, batch: dict) -> dict:  # CenterCrop(inpt_size) is the identity at native size; Normalize
        out = dict(batch)
        out["image"] = self.normalize(batch["image"])
        return out

    def train_aug(self, batch: dict) -> dict:
        """`src/data.py:195-224` on the batch dict: flips (image and mask together), ColorJiggle, RandomSharpness,
        RandomErasing, Gaussian noise, Normalize, on device with backward to `batch["image"]` (`ops.train_aug`).  The two
        colour operations follow kornia's published formulas ("parity unpinned": kornia is not installable here), and so does
        the mask side of RandomErasing: by default the mask only follows the flips; `self.erase_mask = True` also sets the
        erased box to class 0 in the mask.  Random parameters come from `self.aug_generator`."""
        from . import ops

        img = batch["image"]
        B, _, h, w = img.shape
        params, noise, color = sample_train_aug_params(B, h, w, self.config, self.aug_generator, with_color=True,
                                                       erase_mask=self.erase_mask)
        mask = batch.get("mask")
        out, mo = ops.train_aug(img, mask, params.to(img.device), noise.to(img.device) if noise is not None else None,
                                self.mean, self.std, color=color.to(img.device))
        res = dict(batch)
        res["image"] = out
        if mo is not None:
            res["mask"] = mo
        return res

    def setup(self, stage: str) -> None:
        imgs, masks = self.scene
        if stage in ("fit", "train", "validate"):
            self.train_dataset = BeachSegDataset(imgs, masks, self.crops, self.config, create_prompts=True)
            self.val_dataset = self.train_dataset  # src/data.py:245-251: identical datasets
            self.prompt_imgs = self.train_dataset.prompt_imgs
        if stage == "predict":
            self.predict_dataset = BeachSegDataset(imgs, None, self.crops, self.config)

    def _loader(self, ds, batch_size, shuffle):
        return torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=0)

    def train_dataloader(self):
        return self._loader(self.train_dataset, self.config.batch_size, True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, self.config.batch_size, False)

    def predict_dataloader(self):
        return self._loader(self.predict_dataset, 1, False)  # src/data.py:287-293
