"""TEST INFRASTRUCTURE ONLY -- plain-torch statement of the train-time augmentation chain of
`/root/reference/src/data.py:195-224` (kornia `AugmentationSequential`: RandomVerticalFlip, RandomHorizontalFlip,
ColorJiggle, RandomSharpness, RandomErasing, RandomGaussianNoise, Normalize) with EXPLICIT random parameters: the checker of
the HIP kernels `bsg_train_aug` / `bsg_train_aug_bwd` (forward values, and the gradient through torch autograd).  Imported
only by `tests/`; the product path (`beach_seg_amd.ops.train_aug`) never touches it.

PARITY UNPINNED for the two colour operations and for the mask side of RandomErasing: kornia is not installable in the build
container and the reference ships no fixtures, so these follow kornia's published definitions (`kornia.color.rgb_to_hsv` /
`hsv_to_rgb`, `kornia.enhance.adjust_brightness` / `adjust_contrast` / `adjust_saturation` / `adjust_hue` / `sharpness`,
`kornia.augmentation.ColorJiggle.apply_transform`) rather than outputs of the library itself.  Flips, erasing of the image,
noise and Normalize are elementary and exact.
"""
from __future__ import annotations

import torch

IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)


def _rgb_to_hsv(x: torch.Tensor) -> torch.Tensor:
    """kornia.color.rgb_to_hsv on one (3,H,W) image: h in [0, 2 pi), s, v."""
    import math

    mx, imax = x.max(0)
    mn = x.min(0)[0]
    dc = mx - mn
    s = dc / (mx + 1e-8)
    dc = torch.where(dc == 0, torch.ones_like(dc), dc)
    rc, gc, bc = (mx[None] - x).unbind(0)
    hs = torch.stack([bc - gc, (rc - bc) + 2.0 * dc, (gc - rc) + 4.0 * dc]) / dc[None]
    h = torch.gather(hs, 0, imax[None])[0]
    h = (h / 6.0) % 1.0
    return torch.stack([2.0 * math.pi * h, s, mx])


def _hsv_to_rgb(x: torch.Tensor) -> torch.Tensor:
    import math

    h, s, v = x[0] / (2 * math.pi), x[1], x[2]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    p, q, t = v * (1.0 - s), v * (1.0 - f * s), v * (1.0 - (1.0 - f) * s)
    hi = hi.long()
    idx = torch.stack([hi, hi + 6, hi + 12])
    table = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q))
    return torch.gather(table, 0, idx)


def _color_jiggle(x: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """kornia.augmentation.ColorJiggle.apply_transform on one image with explicit factors c = [brightness, contrast,
    saturation, hue, -, order code] (kornia.enhance.adjust_brightness(f - 1) / adjust_contrast / adjust_saturation /
    adjust_hue(2 pi f), applied in the coded order)."""
    import math

    order = int(c[5])
    for k in range(4):
        op = (order >> (2 * k)) & 3
        if op == 0:
            x = (x + (float(c[0]) - 1.0)).clamp(0, 1)
        elif op == 1:
            x = (x * float(c[1])).clamp(0, 1)
        else:
            hsv = _rgb_to_hsv(x)
            if op == 2:
                hsv = torch.stack([hsv[0], (hsv[1] * float(c[2])).clamp(0, 1), hsv[2]])
            else:
                hsv = torch.stack([torch.fmod(hsv[0] + float(c[3]) * 2 * math.pi, 2 * math.pi), hsv[1], hsv[2]])
            x = _hsv_to_rgb(hsv)
    return x


def _sharpness(x: torch.Tensor, factor: float) -> torch.Tensor:
    """kornia.enhance.sharpness on one (3,H,W) image: blend of the image with its clamped 3x3 [[1,1,1],[1,5,1],[1,1,1]]/13
    blur (interior only), `_blend_one` rules for the clamp."""
    import torch.nn.functional as F

    k = (torch.tensor([[1.0, 1.0, 1.0], [1.0, 5.0, 1.0], [1.0, 1.0, 1.0]], dtype=x.dtype, device=x.device) / 13).view(1, 1, 3, 3)
    d = F.conv2d(x[None], k.repeat(3, 1, 1, 1), groups=3)[0].clamp(0.0, 1.0)
    inner = F.pad(torch.ones_like(d), [1, 1, 1, 1])
    result = torch.where(inner == 1, F.pad(d, [1, 1, 1, 1]), x)
    if factor == 0.0:
        return result
    if factor == 1.0:
        return x
    res = result + (x - result) * factor
    return res if 0.0 < factor < 1.0 else res.clamp(0, 1)


def train_aug_reference(img: torch.Tensor, mask: torch.Tensor | None, params: torch.Tensor, noise: torch.Tensor | None,
                        mean=IMAGE_MEAN, std=IMAGE_STD, color: torch.Tensor | None = None):
    """What `ops.train_aug` must compute (any device, autograd through torch)."""
    out, mout = [], []
    for b in range(img.shape[0]):
        fl, ex, ey, ew, eh = (int(v) for v in params[b])
        x = img[b]
        m = mask[b] if mask is not None else None
        if fl & 1:
            x = x.flip(-2)
            m = m.flip(-2) if m is not None else None
        if fl & 2:
            x = x.flip(-1)
            m = m.flip(-1) if m is not None else None
        if color is not None and fl & 16:
            x = _color_jiggle(x, color[b])
        if color is not None and fl & 8:
            x = _sharpness(x, float(color[b, 4]))
        if ew > 0 and eh > 0:
            keep = torch.ones_like(x[0])
            keep[ey:ey + eh, ex:ex + ew] = 0
            x = x * keep
            if m is not None and fl & 32:  # erase_mask: the box becomes class 0 in the mask as well
                m = m.clone()
                m[ey:ey + eh, ex:ex + ew] = 0
        if fl & 4 and noise is not None:
            x = x + noise[b].to(x.device)
        out.append(x)
        mout.append(m)
    out = torch.stack(out)
    mean_t = torch.tensor(mean, dtype=out.dtype, device=out.device).view(1, 3, 1, 1)
    std_t = torch.tensor(std, dtype=out.dtype, device=out.device).view(1, 3, 1, 1)
    return (out - mean_t) / std_t, (torch.stack(mout) if mask is not None else None)
