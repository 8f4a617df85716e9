"""CPU restatement of the tile front-end (TEST INFRASTRUCTURE ONLY -- imported by tests/ and by nothing on the product path).

The reference resizes every crop with Pillow (`/root/reference/src/data.py:93-96`: `Image.fromarray(crop_img).resize(
(inpt_size, inpt_size), resample=config.resample)`, `config.py:48` BICUBIC) after `padded_crop`
(`src/util/geo_util.py:316-341`, zero padding outside the mosaic), then `/255` (`data.py:96`) and the ImageNet
`Normalize` (`data.py:226-234`).  Pillow (third-party, in this container: 12.2) resamples 8-bit images in two INTEGER
passes, horizontal then vertical, with a uint8 intermediate image and 22-bit fixed-point coefficients
(libImaging/Resample.c: `precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`,
`ImagingResampleVertical_8bpc`); `F.interpolate(mode="bicubic")` (a = -0.75, float) is NOT interchangeable with it.
Pinned against Pillow itself: `tests/golden/frontend_pil.npz` (written by `oracle/gen_golden_frontend.py`).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size: int, out_size: int, support: float = 2.0, filt=bicubic_filter):
    """precompute_coeffs + normalize_coeffs_8bpc for the full-image box: (bounds i32 [out][2], kk i32 [out][ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        for x in range(xmax):
            w = filt((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            v = k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_u8(img: np.ndarray, out_size: int) -> np.ndarray:
    """img u8 (H, W, C) square -> u8 (out, out, C): Pillow's BICUBIC for 8-bit images, restated with integers."""
    h, w, c = img.shape
    bx, kx = pil_coeffs(w, out_size)
    by, ky = pil_coeffs(h, out_size)
    src = img.astype(np.int64)
    tmp = np.zeros((h, out_size, c), dtype=np.uint8)
    for xx in range(out_size):
        x0, n = bx[xx]
        acc = np.full((h, c), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for j in range(n):
            acc += src[:, x0 + j, :] * int(kx[xx, j])
        tmp[:, xx, :] = _clip8(acc)
    t64 = tmp.astype(np.int64)
    out = np.zeros((out_size, out_size, c), dtype=np.uint8)
    for yy in range(out_size):
        y0, n = by[yy]
        acc = np.full((out_size, c), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for j in range(n):
            acc += t64[y0 + j] * int(ky[yy, j])
        out[yy] = _clip8(acc)
    return out


def padded_crop(mosaic: np.ndarray, box, crop: int) -> np.ndarray:
    """`crop x crop` window at (xmin, ymin) of an (H, W, C) mosaic, zeros outside (geo_util.py:316-341)."""
    x0, y0 = int(box[0]), int(box[1])
    out = np.zeros((crop, crop, mosaic.shape[2]), dtype=mosaic.dtype)
    ys, xs = max(0, min(mosaic.shape[0], y0 + crop) - y0), max(0, min(mosaic.shape[1], x0 + crop) - x0)
    if ys > 0 and xs > 0 and x0 >= 0 and y0 >= 0:
        out[:ys, :xs] = mosaic[y0:y0 + ys, x0:x0 + xs]
    return out


IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def tile_frontend(mosaic: np.ndarray, boxes: np.ndarray, crop: int, out_size: int):
    """-> (u8 (n, S, S, 3), f32 (n, 3, S, S) normalised) exactly as data.py:88-96 + Normalize produce them."""
    u8 = np.stack([pil_resize_u8(padded_crop(mosaic, b, crop), out_size) for b in boxes])
    f = (u8.astype(np.float32) / np.float32(255.0) - IMAGENET_MEAN) / IMAGENET_STD
    return u8, np.ascontiguousarray(f.transpose(0, 3, 1, 2))


def tif_image_4band(data: np.ndarray, nodata: np.ndarray) -> np.ndarray:
    """`/root/reference/src/util/geo_util.py:449-470`, 4-band branch, restated step by step (float raster):
    R = band 4, G = band 3, B = mean(band 1, band 2); shift/clip to [0, 3000] above the valid minimum; per-channel
    divide by the channel max; nodata -> 0; x255 truncated to uint8.  Pinned by `frontend_pil.npz::tif_rgb`."""
    img = np.stack([data[3], data[2], data[:2].mean(axis=0).astype(data.dtype)])
    lo = img[:, ~nodata].min()
    img = np.clip(img, lo, 3000 + lo) - lo
    img = img - img[:, ~nodata].min()
    out = np.zeros(data.shape[1:] + (3,), dtype=np.uint8)
    for i in range(3):
        ch = img[i] / img[i].max()
        ch[nodata] = 0
        out[..., i] = (ch * 255).astype(np.uint8)
    return out
