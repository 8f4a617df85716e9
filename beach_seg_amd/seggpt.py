"""`SegGptNative`: drop-in for the object the reference gets from `load_model`
(`/root/reference/src/util/ml_util.py:7-13`, a frozen `transformers.SegGptForImageSegmentation`) at its call
sites `self.model(pixel_values=, prompt_pixel_values=, prompt_masks=, labels=, embedding_type=)`
(`src/model.py:139-144, 245-251, 282-288`).  Same keyword signature (`HF:modeling_seggpt.py:831-844`), same
output attribute (`out.pred_masks`, f32 (B,3,2H,W), on the autograd graph with a gradient path to
`prompt_pixel_values`), same `ValueError`s.  All arithmetic runs in the hand-written HIP kernels behind the
C ABI (`include/beach_seg_amd.h`); torch only owns device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from . import _native as N
from .weights import SegGptGeometry, state_dict_shapes


@dataclass
class SegGptImageSegmentationOutput:
    """Fields of `HF:modeling_seggpt.py:63-85`; the reference reads only `pred_masks`."""
    loss: Optional[torch.Tensor] = None
    pred_masks: Optional[torch.Tensor] = None
    hidden_states: None = None
    attentions: None = None


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def token_tables(sd: dict, g: SegGptGeometry) -> tuple[torch.Tensor, torch.Tensor]:
    """Constant folding of `SegGptEmbeddings.forward` (`HF:163-206`) under the default `bool_masked_pos`
    (`HF:902-909`): per (stream kind, token) the sum of conv bias (or mask_token where the mask stream is
    masked), segment token, bicubic-resized position embedding (`HF:145-161`) and type token.
    Returns (instance, semantic) tables, each f32 (2, N, D), computed on the CPU in fp32."""
    e = "model.embeddings."
    hp, wp = g.grid
    D = g.hidden_size
    cpu = {k: v.detach().float().cpu() for k, v in sd.items() if k.startswith(e)}
    pe = cpu[e + "position_embeddings"][:, 1:]
    n = int(round(pe.shape[1] ** 0.5))
    if n != hp or n != wp:
        pe = F.interpolate(pe.reshape(1, n, n, -1).permute(0, 3, 1, 2), size=(hp, wp), mode="bicubic",
                           align_corners=False).permute(0, 2, 3, 1)
    pos = pe.reshape(hp * wp, D)
    bias = cpu[e + "patch_embeddings.projection.bias"].reshape(1, D)
    masked = torch.arange(hp * wp) >= (hp * wp) // 2
    out = []
    for ty in ("type_token_instance", "type_token_semantic"):
        t = cpu[e + ty].reshape(1, D)
        img = bias + cpu[e + "segment_token_input"].reshape(1, D) + pos + t
        first = torch.where(masked[:, None], cpu[e + "mask_token"].reshape(1, D).expand(hp * wp, D),
                            bias.expand(hp * wp, D))
        msk = first + cpu[e + "segment_token_prompt"].reshape(1, D) + pos + t
        out.append(torch.stack([img, msk]).contiguous())
    return out[0], out[1]


def _rel_cat(rel_h: torch.Tensor, rel_w: torch.Tensor) -> torch.Tensor:
    """[LH + LW][64]: rel_pos_h rows at 0.., rel_pos_w rows at LH.., zeros elsewhere (LH/LW = 2Hp / 2Wp rounded up
    to 16): operand of the rel-pos table kernel; its transpose feeds the rel-pos gradient kernel."""
    rel_h, rel_w = rel_h.detach().float(), rel_w.detach().float()
    LH = (rel_h.shape[0] + 1 + 15) // 16 * 16
    LW = (rel_w.shape[0] + 1 + 15) // 16 * 16
    cat = rel_h.new_zeros(LH + LW, rel_h.shape[1])
    cat[: rel_h.shape[0]] = rel_h
    cat[LH: LH + rel_w.shape[0]] = rel_w
    return cat


def _split3(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[W_hi | W_lo | W_hi] along the contraction axis (W = hi + lo in `dtype`): partner of activations laid out
    [x_hi | x_hi | x_lo], so ONE low-precision GEMM with K tripled sums x_hi W_hi + x_hi W_lo + x_lo W_hi."""
    w = w.float()
    hi = w.to(dtype)
    lo = (w - hi.float()).to(dtype)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


X3_WEIGHT_SHIFT = 32.0  # bsg_config.gemm_x3: every Linear weight is stored x 2^5 (exact) so that f16(w) and f16(w - f16(w)) are normal


def build_weight_table(sd: dict, g: SegGptGeometry, dtype: torch.dtype, device, embed_split: bool = False,
                       gemm_x3: bool = False) -> list[torch.Tensor]:
    """Device tensors in the slot order documented in `include/beach_seg_amd.h`."""
    missing = [k for k in state_dict_shapes(g) if k not in sd]
    if missing:
        raise KeyError(f"state dict lacks {missing[:3]}... ({len(missing)} keys)")
    for k, shp in state_dict_shapes(g).items():
        if tuple(sd[k].shape) != shp:
            raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shp}")
    D = g.hidden_size

    def T(x):  # activation dtype
        return x.detach().to(device=device, dtype=dtype).contiguous()

    def f32(x):
        return x.detach().to(device=device, dtype=torch.float32).contiguous()

    def x3(w: torch.Tensor) -> torch.Tensor:  # `bsg_config.gemm_x3` weight format (ops.x3_weight), on the target device
        from .ops import x3_weight

        return x3_weight(w.to(device=device, dtype=torch.float32))

    def lin(name):  # (weight [out][in], weight^T [in][out])
        w = sd[name].detach().to(device=device, dtype=torch.float32)
        return (x3(w), x3(w.t())) if gemm_x3 else (T(w), T(w.t()))

    tab_i, tab_s = token_tables(sd, g)
    pw = sd["model.embeddings.patch_embeddings.projection.weight"].reshape(D, -1)
    cw = sd["decoder.decoder_pred.conv.weight"].detach().to(device=device, dtype=torch.float32)  # [co][ci][ky][kx]
    dd = g.decoder_hidden_size
    conv_w = cw.permute(0, 2, 3, 1).reshape(dd, 9, dd)  # [co][tap][ci]
    conv_wT = cw.flip(2, 3).permute(1, 2, 3, 0).reshape(dd, 9, dd)  # [ci][tap'][co], taps flipped (dgrad)
    dw, dwT = lin("decoder.decoder_embed.weight")
    pw32 = pw.detach().to(device=device, dtype=torch.float32)
    patch = ((_split3(pw32, dtype), _split3(pw32.t(), dtype)) if embed_split else
             (x3(pw32), x3(pw32.t())) if gemm_x3 else (T(pw32), T(pw32.t())))
    table = [
        patch[0], patch[1], f32(tab_i), f32(tab_s),
        f32(sd["model.encoder.layernorm.weight"]), f32(sd["model.encoder.layernorm.bias"]),
        dw, dwT, f32(sd["decoder.decoder_embed.bias"]), T(conv_w), T(conv_wT),
        f32(sd["decoder.decoder_pred.conv.bias"]), f32(sd["decoder.decoder_pred.layernorm.weight"]),
        f32(sd["decoder.decoder_pred.layernorm.bias"]),
        f32(sd["decoder.decoder_pred.head.weight"].reshape(3, dd)), f32(sd["decoder.decoder_pred.head.bias"]),
    ]
    assert len(table) == N.BSG_GLOBAL_WEIGHTS
    for i in range(g.num_hidden_layers):
        l = f"model.encoder.layers.{i}."
        qw, qwT = lin(l + "attention.qkv.weight")
        ow, owT = lin(l + "attention.proj.weight")
        w1, w1T = lin(l + "mlp.lin1.weight")
        w2, w2T = lin(l + "mlp.lin2.weight")
        table += [
            f32(sd[l + "layernorm_before.weight"]), f32(sd[l + "layernorm_before.bias"]),
            qw, qwT, f32(sd[l + "attention.qkv.bias"]), ow, owT, f32(sd[l + "attention.proj.bias"]),
            f32(sd[l + "layernorm_after.weight"]), f32(sd[l + "layernorm_after.bias"]),
            w1, w1T, f32(sd[l + "mlp.lin1.bias"]), w2, w2T, f32(sd[l + "mlp.lin2.bias"]),
            f32(sd[l + "attention.rel_pos_h"]), f32(sd[l + "attention.rel_pos_w"]),
            T(_rel_cat(sd[l + "attention.rel_pos_h"], sd[l + "attention.rel_pos_w"])),
            T(_rel_cat(sd[l + "attention.rel_pos_h"], sd[l + "attention.rel_pos_w"]).t()),
        ]
    return table


class _WsLease:
    """Exclusive hold of one saved-activation workspace by one autograd node: taken at forward, given back when the
    node (its `ctx`) is freed.  A second grad-enabled forward while the first graph is alive therefore gets a
    workspace of its own (f1, f2, b1, b2 works like it does through the HF module's autograd), and a backward
    can never read activations of another forward."""

    def __init__(self, model: "SegGptNative", ws: torch.Tensor):
        self.model, self.ws = model, ws
        model._leased.add(ws.data_ptr())

    def __del__(self):
        try:
            self.model._leased.discard(self.ws.data_ptr())
        except Exception:  # interpreter shutdown
            pass


class _SegGptFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: "SegGptNative", pixel_values, prompt_pixel_values, prompt_masks, emb: int, need_grad: bool):
        ws = model._free_train_workspace(pixel_values.shape[0]) if need_grad else None
        pred = model._run_forward(pixel_values.detach(), prompt_pixel_values.detach(), prompt_masks.detach(), emb,
                                  train=need_grad, ws=ws)
        ctx.model, ctx.batch = model, pixel_values.shape[0]
        ctx.lease = _WsLease(model, ws) if need_grad else None
        return pred

    @staticmethod
    def backward(ctx, grad_pred):
        if ctx.lease is None:
            raise RuntimeError("backward through a forward that saved no activations")
        g = ctx.model._run_backward(grad_pred.contiguous().float(), ctx.batch, ws=ctx.lease.ws)
        ctx.model._last_bwd = (ctx.batch, ctx.lease.ws)
        return None, None, g, None, None, None


class SegGptNative(torch.nn.Module):
    """Frozen SegGPT on the HIP kernels.  `dtype`: torch.float32 (parity mode, exact-f32 MFMA), torch.bfloat16 (the
    dtype BASELINE config 1 names) or torch.float16 (same MFMA rate as bf16 with 8x less operand round-off; the backward
    runs on a device-chosen power-of-two multiple of the gradient)."""

    def __init__(self, state_dict: dict, geometry: SegGptGeometry, device="cuda:0", dtype=torch.bfloat16,
                 embed_split: Optional[bool] = None, gemm_x3: bool = False):
        """`embed_split` (16-bit dtypes; default on): patch embedding and its dgrad as split-precision GEMMs, so pixels and
        the prompt-pixel gradient are not quantised to the MFMA operand type (`bsg_config.embed_split`)."""
        super().__init__()
        geometry.validate()
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("dtype must be torch.float32, torch.bfloat16 or torch.float16")
        self.geometry, self.dtype = geometry, dtype
        self._device = torch.device(device)
        self._lib = N.load()
        self.embed_split = (dtype != torch.float32) if embed_split is None else bool(embed_split)
        if self.embed_split and dtype == torch.float32:
            raise ValueError("embed_split applies to the 16-bit dtypes only")
        # gemm_x3 (float32 only): "float32 at three f16 MFMAs" -- exact-f32 storage / attention / LayerNorm / conv, the Linear
        # GEMMs on the f16 matrix cores with 22-bit operands (`bsg_config.gemm_x3`)
        self.gemm_x3 = bool(gemm_x3)
        if self.gemm_x3 and dtype != torch.float32:
            raise ValueError("gemm_x3 applies to dtype float32 only")
        self._table = build_weight_table(state_dict, geometry, dtype, self._device, self.embed_split, self.gemm_x3)
        cfg = N.BsgConfig()
        g = geometry
        cfg.hidden_size, cfg.num_layers, cfg.num_heads = g.hidden_size, g.num_hidden_layers, g.num_attention_heads
        cfg.canvas_h, cfg.canvas_w, cfg.patch_size, cfg.mlp_dim = g.image_size[0], g.image_size[1], g.patch_size, g.mlp_dim
        cfg.decoder_hidden, cfg.merge_index = g.decoder_hidden_size, g.merge_index
        cfg.num_taps = len(g.intermediate_hidden_state_indices)
        for i, t in enumerate(g.intermediate_hidden_state_indices):
            cfg.taps[i] = t
        cfg.layer_norm_eps = g.layer_norm_eps
        cfg.dtype = {torch.float32: N.BSG_DTYPE_F32, torch.bfloat16: N.BSG_DTYPE_BF16, torch.float16: N.BSG_DTYPE_F16}[dtype]
        cfg.embed_split = int(self.embed_split)
        cfg.gemm_x3 = int(self.gemm_x3)
        ptrs = (C.c_void_p * len(self._table))(*[t.data_ptr() for t in self._table])
        h = C.c_void_p()
        with torch.cuda.device(self._device):
            N.check(self._lib.bsg_create(C.byref(cfg), ptrs, len(self._table), C.byref(h)))
        self._h = h
        self._ws: dict[tuple[int, int], torch.Tensor] = {}
        self._train_ws: dict[int, list[torch.Tensor]] = {}  # per batch size: saved-activation workspaces
        self._leased: set[int] = set()                      # data_ptrs held by live autograd nodes (_WsLease)
        self._last_ws: Optional[torch.Tensor] = None
        self._last_bwd: Optional[tuple[int, torch.Tensor]] = None  # (batch, workspace) of the last autograd backward

    def __del__(self):
        try:
            h = self.__dict__.pop("_h", None)
            if h:
                self._lib.bsg_destroy(h)
        except Exception:  # interpreter shutdown
            pass

    # ---- surface the reference touches on the HF object
    @property
    def device(self) -> torch.device:  # src/model.py:98
        return self._device

    def eval(self):  # src/util/ml_util.py:11 -- weights are frozen and DropPath is the identity already
        return self

    def to(self, *a, **k):
        return self

    # ---- C ABI calls
    def workspace(self, batch: int, train: bool) -> torch.Tensor:
        key = (batch, int(train))
        if key not in self._ws:
            n = self._lib.bsg_workspace_bytes(self._h, batch, int(train))
            self._ws[key] = torch.zeros(n, dtype=torch.uint8, device=self._device)  # ABI: zero-initialised
            if train:
                self._train_ws.setdefault(batch, []).append(self._ws[key])
        return self._ws[key]

    def _free_train_workspace(self, batch: int) -> torch.Tensor:
        """A saved-activation workspace no live autograd node holds (the cached one first; a new one only when two
        grad-enabled forwards of this batch size are alive at once)."""
        self.workspace(batch, True)
        for ws in self._train_ws[batch]:
            if ws.data_ptr() not in self._leased:
                return ws
        n = self._lib.bsg_workspace_bytes(self._h, batch, 1)
        ws = torch.zeros(n, dtype=torch.uint8, device=self._device)
        self._train_ws[batch].append(ws)
        return ws

    def workspace_region(self, batch: int, train: bool, name: str, layer: int = -1) -> torch.Tensor:
        """uint8 view of a named region of the most recent workspace of that shape (test / debugging aid)."""
        off, nb = C.c_size_t(), C.c_size_t()
        N.check(self._lib.bsg_workspace_region(self._h, batch, int(train), name.encode(), layer, C.byref(off), C.byref(nb)))
        return self.workspace(batch, train)[off.value: off.value + nb.value]

    PROFILE_CATEGORIES = ("gemm", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "conv3x3")

    def profile(self, enable: bool) -> None:
        N.check(self._lib.bsg_profile_reset(self._h))
        N.check(self._lib.bsg_profile_enable(self._h, int(enable)))

    def profile_read(self) -> dict:
        """{category: (kernel ms, algorithmic flops, launches)} since `profile(True)`; waits for the events."""
        out = {}
        for i, name in enumerate(self.PROFILE_CATEGORIES):
            ms, fl, n = C.c_double(), C.c_double(), C.c_long()
            N.check(self._lib.bsg_profile_read(self._h, i, C.byref(ms), C.byref(fl), C.byref(n)))
            out[name] = (ms.value, fl.value, n.value)
        return out

    def _run_forward(self, pix, prm, pmask, emb: int, train: bool, ensemble: bool = False,
                     ws: Optional[torch.Tensor] = None, first_row: int = 0) -> torch.Tensor:
        """`ws` None: the fused-engine convention -- a free workspace of this shape, remembered as `_last_ws` for the
        `_run_backward` that follows immediately (no autograd node involved).  `first_row` > 0 (`bsg_forward_rows`): the
        caller reads the prediction on canvas rows >= first_row only; the rows above the first computed 16-row tile come
        back as zeros."""
        B = pix.shape[0]
        H, W = self.geometry.image_size
        pix, prm, pmask = (t.to(self._device, torch.float32).contiguous() for t in (pix, prm, pmask))
        pred = (torch.zeros if first_row else torch.empty)((B, 3, H, W), dtype=torch.float32, device=self._device)
        if ws is None:
            ws = self._free_train_workspace(B) if train else self.workspace(B, False)
        with torch.cuda.device(self._device):
            if ensemble:
                N.check(self._lib.bsg_forward_ensemble(self._h, _stream(), B, _ptr(pix), _ptr(prm), _ptr(pmask), emb,
                                                       _ptr(pred), _ptr(ws), ws.numel()))
            else:
                N.check(self._lib.bsg_forward_rows(self._h, _stream(), B, _ptr(pix), _ptr(prm), _ptr(pmask), emb, int(first_row),
                                                   _ptr(pred), _ptr(ws), ws.numel(), int(train)))
        self._last_ws = ws if train else None
        return pred

    def _run_backward(self, grad_pred: torch.Tensor, B: int, first_row: int = 0,
                      ws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`first_row` > 0: the caller guarantees grad_pred == 0 on canvas rows < first_row (bsg_backward_rows).
        `ws`: the workspace the matching forward saved into (autograd path); None = the engine's last forward."""
        if ws is None:
            ws = self._last_ws
        if ws is None:
            raise RuntimeError("backward without a forward that saved activations")
        if ws.numel() < self._lib.bsg_workspace_bytes(self._h, B, 1):
            raise RuntimeError(f"backward with batch {B} on a workspace saved by a smaller forward")
        H, W = self.geometry.image_size
        g = torch.empty((B, 3, H // 2, W), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            N.check(self._lib.bsg_backward_rows(self._h, _stream(), B, _ptr(grad_pred), int(first_row), _ptr(g), _ptr(ws),
                                                ws.numel()))
        return g

    def grad_overflow_state(self, batch: int, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """int32 device view [overflow flag of the last backward, back-off exponent, clean backwards, dgrad overflows so far,
        last input was non-finite, backwards dropped for a non-finite input, pending forward flag] of the f16 / x3 overflow guard
        (`include/beach_seg_amd.h`, region "gscale"); all zero for the other dtypes.  `ws`: the workspace of that backward
        (default: the engine's last one).  No synchronisation: act on it on the stream."""
        ws = ws if ws is not None else (self._last_ws if self._last_ws is not None else self.workspace(batch, True))
        off, nb = C.c_size_t(), C.c_size_t()
        N.check(self._lib.bsg_workspace_region(self._h, batch, 1, b"gscale", -1, C.byref(off), C.byref(nb)))
        return ws[off.value + 64: off.value + 92].view(torch.int32)

    def last_backward_overflowed(self) -> bool:
        """f16 only: did the last autograd backward produce a non-finite prompt gradient?  (Host synchronisation, like
        `GradScaler.step`: the caller skips `optimizer.step()` when True.)"""
        if not (self.dtype == torch.float16 or self.gemm_x3) or self._last_bwd is None:
            return False
        return bool(int(self.grad_overflow_state(*self._last_bwd)[0]))

    def capture_forward(self, batch: int, embedding_type: str = "instance") -> "GraphedForward":
        """Capture one inference forward (no autograd) of `batch` samples into a hipGraph.  The C ABI enqueues
        everything on the caller's stream without allocation or synchronisation, so the whole launch sequence
        (~350 kernels) replays as one graph launch: BASELINE config 4 (sliding-window predict)."""
        return GraphedForward(self, batch, embedding_type)

    def forward(self, pixel_values, prompt_pixel_values, prompt_masks, bool_masked_pos=None, feature_ensemble=None,
                embedding_type=None, labels=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, **kwargs) -> SegGptImageSegmentationOutput:
        g = self.geometry
        if bool_masked_pos is not None:
            raise NotImplementedError("only the default bool_masked_pos (bottom half masked, HF:902-909) is built")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("attention maps / hidden states are never materialised by the fused kernels")
        if pixel_values.shape[1] != g.num_channels:  # HF:112-115
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in "
                             "the configuration.")
        Hh, W = g.image_size[0] // 2, g.image_size[1]
        for t in (pixel_values, prompt_pixel_values, prompt_masks) + ((labels,) if labels is not None else ()):
            if tuple(t.shape[2:]) != (Hh, W) or t.shape[0] != pixel_values.shape[0] or t.shape[1] != 3:
                raise ValueError(f"Input image size ({2 * t.shape[2]}*{t.shape[3]}) doesn't match model "
                                 f"({g.image_size[0]}*{g.image_size[1]}).")  # HF:116-119
        embedding_type = "instance" if embedding_type is None else embedding_type
        if embedding_type not in ("instance", "semantic"):  # HF:199
            raise ValueError(f"Embedding type should be either 'semantic' or 'instance', but got {embedding_type}")
        emb = 0 if embedding_type == "instance" else 1
        need_grad = torch.is_grad_enabled() and prompt_pixel_values.requires_grad
        if feature_ensemble:  # few-shot inference of src/predict_no_prompt.py:283-304 (HF:414-423)
            if need_grad:
                raise NotImplementedError("feature_ensemble is an inference path: no backward is built for it")
            return SegGptImageSegmentationOutput(loss=None, pred_masks=self._run_forward(
                pixel_values.detach(), prompt_pixel_values.detach(), prompt_masks.detach(), emb, train=False, ensemble=True))
        pred = _SegGptFn.apply(self, pixel_values, prompt_pixel_values, prompt_masks, emb, need_grad)
        # `labels` never reach the network under the default mask (HF:706-715); HF's own `loss` output is
        # unused by the reference (src/model.py:292), so it is not computed here.
        return SegGptImageSegmentationOutput(loss=None, pred_masks=pred)


class GraphedForward:
    """hipGraph replay of `SegGptNative` inference at a fixed batch size: copy inputs into the static buffers,
    replay, read `pred` (f32 (B,3,H,W), overwritten by the next call)."""

    def __init__(self, model: SegGptNative, batch: int, embedding_type: str = "instance"):
        if embedding_type not in ("instance", "semantic"):
            raise ValueError(f"Embedding type should be either 'semantic' or 'instance', but got {embedding_type}")
        self.model, self.batch = model, batch
        emb = 0 if embedding_type == "instance" else 1
        dev = model.device
        Hh, W = model.geometry.image_size[0] // 2, model.geometry.image_size[1]
        self.pix = torch.zeros(batch, 3, Hh, W, device=dev)
        self.prm = torch.zeros_like(self.pix)
        self.pmask = torch.zeros_like(self.pix)
        model.workspace(batch, False)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up outside capture: one-time kernel attribute calls happen here
            for _ in range(2):
                model._run_forward(self.pix, self.prm, self.pmask, emb, train=False)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.pred = model._run_forward(self.pix, self.prm, self.pmask, emb, train=False)

    @torch.no_grad()
    def __call__(self, pixel_values, prompt_pixel_values, prompt_masks) -> torch.Tensor:
        self.pix.copy_(pixel_values)
        self.prm.copy_(prompt_pixel_values)
        self.pmask.copy_(prompt_masks)
        self.graph.replay()
        return self.pred
