"""ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the predict-loop glue
(`/root/reference/src/predict.py:120-159, 100, 259-260`): nearest-neighbour down-size of the decoded
mask, one-hot vote paste with clipping, final arg-max.  Integer work: the product path must match
it bit for bit."""
from __future__ import annotations

import numpy as np


def nearest_resize(mask: np.ndarray, size: int) -> np.ndarray:
    """`cv2.resize(pred, (size, size), interpolation=cv2.INTER_NEAREST)` (`src/predict.py:259`), OpenCV's
    resizeNN index rule in double precision: fx = dsize / (double) ssize, ifx = 1 / fx, source index =
    min(cvFloor(dst * ifx), ssize - 1) on both axes.  cv2 is not importable here ("parity unpinned" beyond this
    restatement of imgproc/resize.cpp); for dyadic ratios (448 -> 112) every formulation agrees."""
    h, w = mask.shape
    ys = np.minimum(np.floor(np.arange(size, dtype=np.float64) * (1.0 / (np.float64(size) / np.float64(h)))).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(size, dtype=np.float64) * (1.0 / (np.float64(size) / np.float64(w)))).astype(np.int64), w - 1)
    return mask[ys[:, None], xs[None, :]]


def one_hot(pred: np.ndarray, num_classes: int) -> np.ndarray:
    """`np.eye(num_classes, dtype=np.uint8)[pred]` (`src/predict.py:260`)."""
    return np.eye(num_classes, dtype=np.uint8)[pred]


def accumulate(counter: np.ndarray, crop: tuple[int, int, int, int], one_hot_pred: np.ndarray) -> bool:
    """`Accumulator.update` vote paste (`src/predict.py:138-157`): clip the crop window to the mosaic,
    add the matching source slice into the uint8 counters (wraps at 256 like numpy uint8)."""
    h, w = counter.shape[:2]
    xmin, ymin, xmax, ymax = crop
    dy0, dy1, dx0, dx1 = max(ymin, 0), min(ymax, h), max(xmin, 0), min(xmax, w)
    sy0, sx0 = dy0 - ymin, dx0 - xmin
    sy1, sx1 = sy0 + (dy1 - dy0), sx0 + (dx1 - dx0)
    if sy1 <= sy0 or sx1 <= sx0:
        return False
    counter[dy0:dy1, dx0:dx1] += one_hot_pred[sy0:sy1, sx0:sx1]
    return True


def vote_argmax(counter: np.ndarray) -> np.ndarray:
    """`np.argmax(self.current_pred_counter, axis=2)` (`src/predict.py:100`)."""
    return np.argmax(counter, axis=2)


def hf_build_palette(num_labels: int):
    """`HF:image_processing_seggpt.py` build_palette: class 0 black, then a base^3 lattice walked from white downwards."""
    base = int(num_labels ** (1 / 3)) + 1
    margin = 256 // base
    out = [(0, 0, 0)]
    for loc in range(num_labels):
        r, g, b = loc // base ** 2, (loc % base ** 2) // base, loc % base
        out.append((255 - r * margin, 255 - g * margin, 255 - b * margin))
    return out


def hf_post_process(pred_masks, num_labels: int, target_size=None):
    """`SegGptImageProcessor.post_process_semantic_segmentation` (`HF:image_processing_seggpt.py:300-332`), the decode the
    reference's second caller uses (`/root/reference/src/predict_no_prompt.py:297-303`): bottom half of the canvas ->
    un-normalise (x * std + mean) -> clip(x * 255, 0, 255) -> optional nearest resize -> arg-min of the squared distance
    to the integer palette.  pred_masks: torch f32 (B, 3, 2H, W) -> list of int64 (h, w)."""
    import torch

    std = torch.tensor([0.229, 0.224, 0.225])
    mean = torch.tensor([0.485, 0.456, 0.406])
    m = pred_masks[:, :, pred_masks.shape[2] // 2:, :]
    m = (m.permute(0, 2, 3, 1) * std + mean).permute(0, 3, 1, 2)
    m = torch.clip(m * 255, 0, 255)
    pal = torch.tensor(hf_build_palette(num_labels), dtype=torch.float).view(1, 1, num_labels + 1, 3)
    out = []
    for mask in m:
        if target_size is not None:
            mask = torch.nn.functional.interpolate(mask.unsqueeze(0), size=target_size, mode="nearest")[0]
        c, h, w = mask.shape
        d = torch.pow(mask.permute(1, 2, 0).view(h, w, 1, c) - pal, 2).sum(-1)
        out.append(d.argmin(-1))
    return out
