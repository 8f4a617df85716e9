"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Deterministic synthetic inputs shared by the fixture generator
(`oracle/gen_golden.py`) and the tests, so large inputs need not be committed."""
from __future__ import annotations

import torch

from beach_seg_amd.weights import SegGptGeometry, counter_noise


def synth_inputs(g: SegGptGeometry, B: int, seed: int):
    """Post-normalisation-like inputs: N(0,1)-ish images, palette-coloured masks, block labels."""
    H, W = g.image_size[0] // 2, g.image_size[1]
    n = B * 3 * H * W
    pix = counter_noise(n, seed * 10 + 1).reshape(B, 3, H, W)
    prm = counter_noise(n, seed * 10 + 2).reshape(B, 3, H, W)
    blk = 16
    cls = lambda s: ((counter_noise(B * (H // blk) * (W // blk), s) * 1000).long().abs() % 4).reshape(
        B, H // blk, W // blk).repeat_interleave(blk, 1).repeat_interleave(blk, 2)
    pm_cls, lb_cls = cls(seed * 10 + 3), cls(seed * 10 + 4)
    pal = ((counter_noise(B * 4 * 3, seed * 10 + 5) * 1000).long().abs() % 256).to(torch.uint8).reshape(B, 4, 3)
    pal[:, 0] = 0
    return pix, prm, pm_cls.to(torch.uint8), lb_cls.to(torch.uint8), pal


def peaked_state_dict(g: SegGptGeometry, seed: int, gain: float = 8.0, device="cpu"):
    """Synthetic weights whose attention is PEAKED, as a trained checkpoint's is: the q and k rows of every qkv
    projection (weight and bias) and both rel-pos tables are multiplied by `gain` (a power of two: the product is
    exact, so every box rebuilds the same bits).  With the plain sigma = 0.02 init the logits are O(0.5) and every
    softmax row is near-uniform, which never drives the online-softmax rescale nor the exp2 range."""
    from beach_seg_amd.weights import synth_state_dict

    sd = synth_state_dict(g, seed=seed, device=device)
    D = g.hidden_size
    for i in range(g.num_hidden_layers):
        l = f"model.encoder.layers.{i}.attention."
        sd[l + "qkv.weight"][: 2 * D] *= gain
        sd[l + "qkv.bias"][: 2 * D] *= gain
        sd[l + "rel_pos_h"] *= gain
        sd[l + "rel_pos_w"] *= gain
    return sd
