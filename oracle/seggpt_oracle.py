"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (`beach_seg_amd/`).

CPU restatement (plain torch fp32/fp64 ops, token-major) of the network the reference drives
through `self.model(...)` (`/root/reference/src/model.py:139-144, 245-251, 282-288`), i.e. the
third-party `transformers.models.seggpt.modeling_seggpt` (v5.15.0; the reference leaves
`transformers` unpinned, `environment.yml:33`).  Every function cites the HF lines it restates
(`HF:` = `transformers/models/seggpt/modeling_seggpt.py`).  Gradients come from torch autograd
over this restatement.

Pinning: the reference ships no tests or golden vectors, so this oracle is
pinned against outputs of the reference code itself, generated in the build container by
`oracle/gen_golden.py` (imports HF SegGPT + the reference's `src/model.py` wrapper math) and
committed under `tests/golden/`.  `tests/test_oracle_golden.py` checks this file against them.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def patch_rows(img: torch.Tensor, p: int) -> torch.Tensor:
    """(B,C,H,W) -> (B, Hp*Wp, C*p*p), k = c*p*p + i*p + j: the im2col of a k=s=p conv
    (`HF:108,120` Conv2d(3->D, k16, s16) then permute to NHWC)."""
    B, C, H, W = img.shape
    x = img.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (H // p) * (W // p), C * p * p)


def pos_embed_grid(w: dict, g) -> torch.Tensor:
    """`interpolate_pos_encoding` (`HF:145-161`): drop CLS, bicubic 14x14 -> Hp x Wp.  -> (N, D)"""
    hp, wp = g.grid
    pe = w["model.embeddings.position_embeddings"][:, 1:]
    n = int(round(math.sqrt(pe.shape[1])))
    if n != hp or n != wp:
        pe = F.interpolate(pe.reshape(1, n, n, -1).permute(0, 3, 1, 2), size=(hp, wp), mode="bicubic",
                           align_corners=False).permute(0, 2, 3, 1)
    return pe.reshape(hp * wp, -1)


def embeddings(w: dict, g, img_canvas, mask_canvas, bool_masked_pos, embedding_type="instance"):
    """`SegGptEmbeddings.forward` (`HF:163-206`) -> (2B, N, D): first B image stream, last B mask stream."""
    e = "model.embeddings."
    W = w[e + "patch_embeddings.projection.weight"].reshape(g.hidden_size, -1)
    b = w[e + "patch_embeddings.projection.bias"]
    xi = patch_rows(img_canvas, g.patch_size) @ W.t() + b
    xm = patch_rows(mask_canvas, g.patch_size) @ W.t() + b
    m = bool_masked_pos.to(xm.dtype).reshape(-1, g.num_tokens, 1)
    xm = xm * (1 - m) + w[e + "mask_token"].reshape(1, 1, -1) * m
    if embedding_type == "semantic":
        ty = w[e + "type_token_semantic"]
    elif embedding_type == "instance":
        ty = w[e + "type_token_instance"]
    else:
        raise ValueError(f"Embedding type should be either 'semantic' or 'instance', but got {embedding_type}")
    pos = pos_embed_grid(w, g)
    xi = xi + w[e + "segment_token_input"].reshape(1, 1, -1) + pos + ty.reshape(1, 1, -1)
    xm = xm + w[e + "segment_token_prompt"].reshape(1, 1, -1) + pos + ty.reshape(1, 1, -1)
    return torch.cat((xi, xm), 0)


def rel_tables(rel_pos: torch.Tensor, size: int) -> torch.Tensor:
    """`get_rel_pos` at q_size == k_size == size (`HF:236-266`): the linear resize to 2*size-1 is the
    identity, so R[q, k, :] = rel_pos[q - k + size - 1]."""
    assert rel_pos.shape[0] == 2 * size - 1
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + size - 1
    return rel_pos[idx]


def attention(w: dict, l: str, g, x: torch.Tensor) -> torch.Tensor:
    """`SegGptAttention.forward` (`HF:313-348`) incl. `add_decomposed_rel_pos` (`HF:268-311`).  x: (S,N,D)."""
    S, N, D = x.shape
    nh, hd = g.num_attention_heads, g.head_dim
    hp, wp = g.grid
    qkv = x @ w[l + "attention.qkv.weight"].t() + w[l + "attention.qkv.bias"]
    qkv = qkv.reshape(S, N, 3, nh, hd).permute(2, 0, 3, 1, 4)  # (3,S,nh,N,hd)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q * hd**-0.5) @ k.transpose(-2, -1)  # (S,nh,N,N)
    Rh = rel_tables(w[l + "attention.rel_pos_h"], hp)  # (hp,hp,hd)
    Rw = rel_tables(w[l + "attention.rel_pos_w"], wp)
    qg = q.reshape(S, nh, hp, wp, hd)  # rel-pos uses the UNSCALED q (HF:326-329)
    rel_h = torch.einsum("snhwc,hkc->snhwk", qg, Rh)
    rel_w = torch.einsum("snhwc,wkc->snhwk", qg, Rw)
    att = att.reshape(S, nh, hp, wp, hp, wp) + rel_h[..., :, None] + rel_w[..., None, :]
    att = torch.softmax(att.reshape(S, nh, N, N).float(), dim=-1).to(q.dtype)
    o = (att @ v).permute(0, 2, 1, 3).reshape(S, N, D)
    return o @ w[l + "attention.proj.weight"].t() + w[l + "attention.proj.bias"]


def layer(w: dict, i: int, g, x: torch.Tensor, feature_ensemble: bool = False) -> torch.Tensor:
    """`SegGptLayer.forward` (`HF:400-435`); DropPath is the identity in eval (`HF:379`)."""
    l = f"model.encoder.layers.{i}."
    eps = g.layer_norm_eps
    D = x.shape[-1]
    a = attention(w, l, g, F.layer_norm(x, (D,), w[l + "layernorm_before.weight"],
                                        w[l + "layernorm_before.bias"], eps))
    ensemble_cond = 2 if g.merge_index > i else 1
    if feature_ensemble and a.shape[0] // 2 >= ensemble_cond:  # HF:414-423
        half = a.shape[1] // 2
        prompt, inputs = a[:, :half], a[:, half:]
        if ensemble_cond == 2:
            num_prompts = a.shape[0] // 2
            inputs = inputs.reshape(2, num_prompts, -1)
            inputs = inputs.mean(dim=1, keepdim=True).expand_as(inputs).reshape(prompt.shape)
        else:
            inputs = inputs.mean(dim=0, keepdim=True).expand_as(inputs)
        a = torch.cat([prompt, inputs], dim=1)
    x = x + a
    h = F.layer_norm(x, (D,), w[l + "layernorm_after.weight"], w[l + "layernorm_after.bias"], eps)
    h = F.gelu(h @ w[l + "mlp.lin1.weight"].t() + w[l + "mlp.lin1.bias"])  # exact erf GELU (ACT2FN["gelu"])
    return x + h @ w[l + "mlp.lin2.weight"].t() + w[l + "mlp.lin2.bias"]


def encoder(w: dict, g, x: torch.Tensor, feature_ensemble: bool = False) -> list[torch.Tensor]:
    """`SegGptEncoder.forward` (`HF:447-495`): merge after block `merge_index`, LN-tapped features."""
    taps = []
    D = x.shape[-1]
    for i in range(g.num_hidden_layers):
        x = layer(w, i, g, x, feature_ensemble)
        if i == g.merge_index:
            B = x.shape[0] // 2
            x = (x[:B] + x[B:]) * 0.5
        if i in g.intermediate_hidden_state_indices:
            taps.append(F.layer_norm(x, (D,), w["model.encoder.layernorm.weight"],
                                     w["model.encoder.layernorm.bias"], g.layer_norm_eps))
    return taps


def decoder(w: dict, g, feats: torch.Tensor) -> torch.Tensor:
    """`SegGptDecoder.forward` + `SegGptDecoderHead` (`HF:525-579`).  feats (B,N,4D) -> (B,3,H,W)."""
    B = feats.shape[0]
    hp, wp = g.grid
    p, dd = g.patch_size, g.decoder_hidden_size
    y = feats @ w["decoder.decoder_embed.weight"].t() + w["decoder.decoder_embed.bias"]
    y = y.reshape(B, hp, wp, p, p, dd).permute(0, 5, 1, 3, 2, 4).reshape(B, dd, hp * p, wp * p)
    y = F.conv2d(y, w["decoder.decoder_pred.conv.weight"], w["decoder.decoder_pred.conv.bias"], padding=1)
    y = F.layer_norm(y.permute(0, 2, 3, 1), (dd,), w["decoder.decoder_pred.layernorm.weight"],
                     w["decoder.decoder_pred.layernorm.bias"], g.layer_norm_eps).permute(0, 3, 1, 2)
    y = F.gelu(y)
    return F.conv2d(y, w["decoder.decoder_pred.head.weight"], w["decoder.decoder_pred.head.bias"])


def default_bool_masked_pos(g) -> torch.Tensor:
    """`HF:902-909`: top half visible, bottom half masked."""
    n = g.num_tokens
    return torch.cat([torch.zeros(n // 2, dtype=torch.bool), torch.ones(n - n // 2, dtype=torch.bool)])[None]


def forward(w: dict, g, pixel_values, prompt_pixel_values, prompt_masks, labels=None,
            embedding_type="instance", bool_masked_pos=None, feature_ensemble=False) -> torch.Tensor:
    """`SegGptForImageSegmentation.forward` (`HF:831-951`) -> pred_masks (B,3,2H,W)."""
    if pixel_values.shape[1] != g.num_channels:
        raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in "
                         "the configuration.")
    img = torch.cat((prompt_pixel_values, pixel_values), dim=2)  # HF:705
    if tuple(img.shape[2:]) != tuple(g.image_size):
        raise ValueError(f"Input image size ({img.shape[2]}*{img.shape[3]}) doesn't match model "
                         f"({g.image_size[0]}*{g.image_size[1]}).")
    msk = torch.cat((prompt_masks, prompt_masks if labels is None else labels), dim=2)  # HF:706-710
    if bool_masked_pos is None:
        bool_masked_pos = default_bool_masked_pos(g)
    x = embeddings(w, g, img, msk, bool_masked_pos, embedding_type)
    taps = encoder(w, g, x, feature_ensemble)
    return decoder(w, g, torch.cat(taps, dim=-1))  # HF:925-926


# --------------------------------------------------------------------------------------------
# Reference wrapper math (`/root/reference/src/model.py`, `src/util/ml_util.py`, `src/predict.py`)
# --------------------------------------------------------------------------------------------

IMAGE_MEAN = (0.485, 0.456, 0.406)  # SegGptImageProcessor defaults (HF:image_processing_seggpt.py:76-79)
IMAGE_STD = (0.229, 0.224, 0.225)


def normalize(x: torch.Tensor) -> torch.Tensor:
    """`BeachSegDataModule.normalize` (`src/data.py:345-346`, kornia Normalize): (x - mean) / std."""
    mean = torch.tensor(IMAGE_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGE_STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def build_palette(num_labels: int) -> list[tuple[int, int, int]]:
    """`src/util/ml_util.py:72-89`."""
    base = int(num_labels ** (1 / 3)) + 1
    margin = 256 // base
    out = [(0, 0, 0)]
    for loc in range(num_labels):
        out.append((255 - (loc // base**2) * margin, 255 - ((loc % base**2) // base) * margin,
                    255 - (loc % base) * margin))
    return out


def apply_mask_rgb(palette: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """`torch_apply_mask_rgb` (`src/util/ml_util.py:114-132`): LUT gather -> (B,3,H,W) f32 in [0,1]."""
    if mask.ndim == 4:
        mask = mask[:, 0]
    B = mask.shape[0]
    rgb = palette[torch.arange(B)[:, None, None], mask.long()]
    return rgb.permute(0, 3, 1, 2).to(torch.float32) / 255.0


def palette_norm(palette: torch.Tensor) -> torch.Tensor:
    """`create_palette` normalised copy (`src/model.py:221-229`): (c/255 - mean)/std -> (B,K,3) f32."""
    mean = torch.tensor(IMAGE_MEAN, dtype=torch.float32)
    std = torch.tensor(IMAGE_STD, dtype=torch.float32)
    return (palette.to(torch.float32) / 255 - mean) / std


def seggpt_loss(pred: torch.Tensor, labels: torch.Tensor, yesdata: torch.Tensor, beta: float,
                variant: str = "reference") -> torch.Tensor:
    """`SegGptLoss.forward` (`src/model.py:45-64`).  `variant="reference"` reproduces the `unsqueeze(1)`
    broadcast at `:61` (sum over i,j of keep_i * loss_j / sum keep); `"per_sample"` is keep_j * loss_j."""
    B, C, H2, W = pred.shape
    H = H2 // 2
    blank = torch.zeros((B, C, H, W), dtype=pred.dtype)
    gt = torch.cat([blank, labels], dim=2)
    keep = torch.cat([blank, yesdata.expand(-1, C, -1, -1).to(pred.dtype)], dim=2)
    loss = F.smooth_l1_loss(pred, gt, reduction="none", beta=beta)
    if variant == "reference":
        loss = loss * keep.unsqueeze(1)
    elif variant == "per_sample":
        loss = loss * keep
    else:
        raise ValueError(variant)
    return loss.sum() / keep.sum()


def decode_argmin(pred: torch.Tensor, pal_norm: torch.Tensor) -> torch.Tensor:
    """`process_pred_masks` (`src/model.py:155-175`): bottom half, squared distance to the normalised
    palette colours, argmin (first index on ties) -> i64 (B,H,W)."""
    H = pred.shape[2] // 2
    x = pred[:, :, H:, :].permute(0, 2, 3, 1)  # (B,H,W,3)
    d = x[:, :, :, None, :] - pal_norm[:, None, None, :, :]
    return torch.pow(d, 2).sum(-1).argmin(-1)
