"""Writes tests/golden/hf_postprocess.npz: `SegGptImageProcessor.post_process_semantic_segmentation` (the transformers
class the reference's predict_no_prompt.py:297-303 calls) on synthetic pred_masks, with and without target sizes, plus the
mean-over-prompts step of predict_no_prompt.py:296.  Run in the build container only (needs transformers)."""
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch
from transformers import SegGptImageProcessor

root = Path(__file__).resolve().parents[1]
gen = torch.Generator().manual_seed(77)
proc = SegGptImageProcessor()
num_labels = 3
pal = torch.tensor(proc.get_palette(num_labels), dtype=torch.float)           # (4, 3) integers
mean, std = torch.tensor(proc.image_mean), torch.tensor(proc.image_std)
# pixels near palette colours (+ noise, some out of [0, 255] so the clip matters), in normalised units, K prompts of one query
K, H, W = 3, 48, 40
cls = torch.randint(0, 4, (K, H, W), generator=gen)
rgb = pal[cls] + 60.0 * torch.randn(K, H, W, 3, generator=gen)
bottom = ((rgb / 255.0 - mean) / std).permute(0, 3, 1, 2)
pred = torch.cat([torch.randn(K, 3, H, W, generator=gen), bottom], dim=2).contiguous()   # (K, 3, 2H, W)
full = proc.post_process_semantic_segmentation(SimpleNamespace(pred_masks=pred), None, num_labels=num_labels)
small = proc.post_process_semantic_segmentation(SimpleNamespace(pred_masks=pred), [(12, 10)] * K, num_labels=num_labels)
meanpred = pred.mean(dim=0).unsqueeze(0)                                         # predict_no_prompt.py:296
ens = proc.post_process_semantic_segmentation(SimpleNamespace(pred_masks=meanpred), [(16, 16)], num_labels=num_labels)
np.savez_compressed(root / "tests" / "golden" / "hf_postprocess.npz", pred=pred.numpy(), num_labels=num_labels,
                    palette=pal.numpy(), full=torch.stack(full).numpy(), small=torch.stack(small).numpy(),
                    ens=torch.stack(ens).numpy())
print("wrote hf_postprocess.npz", torch.stack(full).shape, torch.stack(small).shape, torch.stack(ens).shape)
