"""Network geometry + deterministic synthetic weights for the SegGPT hot path.

The reference loads `BAAI/seggpt-vit-large` from the hub (`src/util/ml_util.py:7-13`);
that checkpoint is not available offline, so tests and benchmarks use weights from a
counter-based generator that needs integer arithmetic and ONE fp32 multiply per value
-- the same bits on a CPU, on a GPU, and under any library version.  Tensor names and
shapes are exactly the HF `SegGptForImageSegmentation.state_dict()` ones
(`HF:modeling_seggpt.py:89-579`), so a real checkpoint loads through the same path.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field

import torch


@dataclass(frozen=True)
class SegGptGeometry:
    """Mirror of the `SegGptConfig` fields the hot path reads (`HF:configuration_seggpt.py:57-75`)."""

    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: tuple[int, int] = (896, 448)  # canvas: prompt stacked over query on H
    patch_size: int = 16
    num_channels: int = 3
    mlp_dim: int = 4096
    pretrain_image_size: int = 224
    decoder_hidden_size: int = 64
    merge_index: int = 2
    intermediate_hidden_state_indices: tuple[int, ...] = (5, 11, 17, 23)
    layer_norm_eps: float = 1e-6
    beta: float = 0.01

    @property
    def grid(self) -> tuple[int, int]:
        return (self.image_size[0] // self.patch_size, self.image_size[1] // self.patch_size)

    @property
    def num_tokens(self) -> int:
        return self.grid[0] * self.grid[1]

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def validate(self) -> None:
        hp, wp = self.grid
        if self.head_dim != 64:
            raise ValueError(f"HIP attention kernels are built for head_dim 64, got {self.head_dim}")
        if self.decoder_hidden_size not in (64, 128):
            raise ValueError("HIP decoder kernels are built for decoder_hidden_size 64 or 128")
        if self.patch_size != 16 or self.num_channels != 3:
            raise ValueError("patch_size must be 16 and num_channels 3")
        if hp % 2 or self.image_size[0] % 32:
            raise ValueError("canvas height must hold two equal halves of whole patches")
        if wp > 32 or wp % 4:
            raise ValueError("token-grid width must be a multiple of 4 and <= 32")
        if self.hidden_size % 64 or self.mlp_dim % 64:
            raise ValueError("hidden_size and mlp_dim must be multiples of 64")
        if self.merge_index > min(self.intermediate_hidden_state_indices):
            raise ValueError("merge_index must not exceed the first tap index")  # HF:configuration_seggpt.py:81-85

    @staticmethod
    def vit_large() -> "SegGptGeometry":
        return SegGptGeometry()

    @staticmethod
    def tiny() -> "SegGptGeometry":
        """Small net with the real head_dim / patch / decoder width: runs in the oracle in < 1 s."""
        return SegGptGeometry(
            hidden_size=128, num_hidden_layers=6, num_attention_heads=2, image_size=(128, 64),
            mlp_dim=512, pretrain_image_size=64, merge_index=2,
            intermediate_hidden_state_indices=(2, 3, 4, 5),
        )

    @staticmethod
    def small() -> "SegGptGeometry":
        """Mid-size net: real token grid 56x28 (1568 tokens) but 2 heads / 6 layers."""
        return SegGptGeometry(
            hidden_size=128, num_hidden_layers=6, num_attention_heads=2, image_size=(896, 448),
            mlp_dim=512, pretrain_image_size=224, merge_index=2,
            intermediate_hidden_state_indices=(2, 3, 4, 5),
        )

    @staticmethod
    def config5() -> "SegGptGeometry":
        """BASELINE.json configs[4] ("4x512x512 tiles, deeper encoder (2x channels)") as SURVEY.md section 8(d) maps it onto
        `SegGptConfig` (`HF:configuration_seggpt.py:57-75`): canvas 1024 x 512 (64 x 32 tokens), hidden 2048, 32 heads,
        mlp 4 x hidden, decoder width 128, 24 layers.  Random-init only: no checkpoint of this shape exists."""
        return SegGptGeometry(hidden_size=2048, num_attention_heads=32, image_size=(1024, 512), mlp_dim=8192,
                              decoder_hidden_size=128)

    def to_hf_kwargs(self) -> dict:
        return dict(
            hidden_size=self.hidden_size, num_hidden_layers=self.num_hidden_layers,
            num_attention_heads=self.num_attention_heads, image_size=list(self.image_size),
            patch_size=self.patch_size, num_channels=self.num_channels, mlp_dim=self.mlp_dim,
            pretrain_image_size=self.pretrain_image_size, decoder_hidden_size=self.decoder_hidden_size,
            merge_index=self.merge_index,
            intermediate_hidden_state_indices=list(self.intermediate_hidden_state_indices),
            layer_norm_eps=self.layer_norm_eps, beta=self.beta, drop_path_rate=0.0,
        )


def state_dict_shapes(g: SegGptGeometry) -> dict[str, tuple[int, ...]]:
    """Names and shapes of `SegGptForImageSegmentation.state_dict()` for geometry `g`."""
    D, hp, wp = g.hidden_size, *g.grid
    p, hd = g.patch_size, g.head_dim
    s: dict[str, tuple[int, ...]] = {}
    e = "model.embeddings."
    for tok in ("mask_token", "segment_token_input", "segment_token_prompt", "type_token_semantic",
                "type_token_instance"):
        s[e + tok] = (1, 1, 1, D)
    s[e + "position_embeddings"] = (1, (g.pretrain_image_size // p) ** 2 + 1, D)
    s[e + "patch_embeddings.projection.weight"] = (D, g.num_channels, p, p)
    s[e + "patch_embeddings.projection.bias"] = (D,)
    for i in range(g.num_hidden_layers):
        l = f"model.encoder.layers.{i}."
        s[l + "attention.rel_pos_h"] = (2 * hp - 1, hd)
        s[l + "attention.rel_pos_w"] = (2 * wp - 1, hd)
        s[l + "attention.qkv.weight"] = (3 * D, D)
        s[l + "attention.qkv.bias"] = (3 * D,)
        s[l + "attention.proj.weight"] = (D, D)
        s[l + "attention.proj.bias"] = (D,)
        s[l + "mlp.lin1.weight"] = (g.mlp_dim, D)
        s[l + "mlp.lin1.bias"] = (g.mlp_dim,)
        s[l + "mlp.lin2.weight"] = (D, g.mlp_dim)
        s[l + "mlp.lin2.bias"] = (D,)
        for ln in ("layernorm_before", "layernorm_after"):
            s[l + ln + ".weight"] = (D,)
            s[l + ln + ".bias"] = (D,)
    s["model.encoder.layernorm.weight"] = (D,)
    s["model.encoder.layernorm.bias"] = (D,)
    dd, nt = g.decoder_hidden_size, len(g.intermediate_hidden_state_indices)
    s["decoder.decoder_embed.weight"] = (p * p * dd, D * nt)
    s["decoder.decoder_embed.bias"] = (p * p * dd,)
    s["decoder.decoder_pred.conv.weight"] = (dd, dd, 3, 3)
    s["decoder.decoder_pred.conv.bias"] = (dd,)
    s["decoder.decoder_pred.layernorm.weight"] = (dd,)
    s["decoder.decoder_pred.layernorm.bias"] = (dd,)
    s["decoder.decoder_pred.head.weight"] = (3, dd, 1, 1)
    s["decoder.decoder_pred.head.bias"] = (3,)
    return s


def _to_i64(x: int) -> int:
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


_GOLDEN = _to_i64(0x9E3779B97F4A7C15)
_M1 = _to_i64(0xBF58476D1CE4E5B9)
_M2 = _to_i64(0x94D049BB133111EB)


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    return (x >> s) & ((1 << (64 - s)) - 1)


def counter_noise(numel: int, seed: int, device="cpu") -> torch.Tensor:
    """`numel` fp32 values, zero mean, unit variance, bounded (|x| <= 3.47): a splitmix64 hash of
    (seed, index) cut into four 16-bit lanes whose sum is an Irwin-Hall(4) variate."""
    idx = torch.arange(numel, dtype=torch.int64, device=device)
    z = idx * _GOLDEN + _to_i64(seed * 0xD1342543DE82EF95 + 0x2545F4914F6CDD1D)
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    z = z ^ _lsr(z, 31)
    s = (z & 0xFFFF) + (_lsr(z, 16) & 0xFFFF) + (_lsr(z, 32) & 0xFFFF) + _lsr(z, 48)
    # Var[U(0,65535) summed 4x] = 4 * (65536^2 - 1) / 12
    scale = 1.0 / (((65536.0**2 - 1.0) / 3.0) ** 0.5)
    return (s - 131070).to(torch.float32) * scale


def synth_state_dict(g: SegGptGeometry, seed: int = 0, device="cpu", hf_init: bool = False
                     ) -> dict[str, torch.Tensor]:
    """Deterministic fp32 state dict.  `hf_init=False` (default) also perturbs biases and LayerNorm
    affine parameters so that no term of the network is trivially zero in a parity test;
    `hf_init=True` follows `HF:modeling_seggpt.py:591-609` (zero bias, unit LayerNorm)."""
    out = {}
    std = 0.02
    for name, shape in state_dict_shapes(g).items():
        n = 1
        for d in shape:
            n *= d
        key = zlib.crc32(name.encode()) ^ (seed * 0x1000193)
        noise = counter_noise(n, key, device).reshape(shape)
        is_ln = "layernorm" in name
        if name.endswith(".bias"):
            t = torch.zeros(shape, device=device) if hf_init else noise * (0.05 if is_ln else std)
        elif is_ln:  # LayerNorm weight
            t = torch.ones(shape, device=device) if hf_init else 1.0 + 0.1 * noise
        else:
            t = noise * std
        out[name] = t.contiguous()
    return out
