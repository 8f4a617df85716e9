"""Torch-facing wrappers of the wrapper-math entry points of the C ABI: the reference's loss
(`/root/reference/src/model.py:40-64`), palette arg-min decode (`src/model.py:155-175`), prompt gather /
gradient scatter (`src/model.py:177-213`), AdamW (`src/model.py:398`) and predict-loop voting
(`src/predict.py:100, 120-159, 259-260`).  Tensors must live on the GPU; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N

IMAGE_MEAN = (0.485, 0.456, 0.406)  # SegGptImageProcessor defaults, consumed at src/data.py:192-193
IMAGE_STD = (0.229, 0.224, 0.225)
VARIANTS = {"reference": 0, "per_sample": 1}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise N.NativeError("beach_seg_amd.ops run on the MI355X only: tensor is on the CPU")


def loss_fwd_bwd(pred, labels, yes, beta: float, variant: str, want_grad: bool):
    """One fused kernel pass: (loss scalar, d loss / d pred or None).  No autograd involved."""
    lib = N.load()
    B, _, H2, W = pred.shape
    h = H2 // 2
    pred_c = pred.detach().float().contiguous()
    labels_c = labels.detach().float().contiguous()
    yes_c = yes.reshape(B, h, W).to(torch.uint8).contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred_c) if want_grad else None
    scratch = torch.empty(lib.bsg_loss_scratch_bytes(h, W), dtype=torch.uint8, device=pred.device)
    with torch.cuda.device(pred.device):
        N.check(lib.bsg_loss_fwd_bwd(_stream(), B, h, W, _ptr(pred_c), _ptr(labels_c), _ptr(yes_c), float(beta),
                                     VARIANTS[variant], _ptr(loss), _ptr(grad), _ptr(scratch), scratch.numel()))
    return loss[0], grad


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, labels, yes, beta, variant):
        loss, ctx.grad = loss_fwd_bwd(pred, labels, yes, beta, variant, pred.requires_grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None, None


def seggpt_loss(pred, labels, yesdata, beta: float, variant: str = "reference") -> torch.Tensor:
    """`SegGptLoss.forward` (`src/model.py:45-64`).  pred (B,3,2H,W), labels (B,3,H,W), yesdata bool (B,1,H,W).
    `variant="reference"` keeps the `unsqueeze(1)` batch broadcast of `:61`."""
    _need_gpu(pred, labels, yesdata)
    if variant not in VARIANTS:
        raise ValueError(f"variant must be one of {list(VARIANTS)}")
    return _LossFn.apply(pred, labels, yesdata, beta, variant)


def loss_fwd_bwd_ids(pred, class_ids, palette_norm, beta: float, variant: str, want_grad: bool):
    """`loss_fwd_bwd` with the label image left un-materialised (`bsg_loss_fwd_bwd_ids`): class_ids u8 (B,h,w) /
    (B,1,h,w), palette_norm f32 (B,K,3); label pixel = palette_norm[b][id], yesdata = (id != 0)."""
    lib = N.load()
    B, _, H2, W = pred.shape
    h, K = H2 // 2, palette_norm.shape[1]
    pred_c = pred.detach().float().contiguous()
    ids = class_ids.reshape(B, h, W).to(torch.uint8).contiguous()
    pal = palette_norm.detach().float().contiguous()
    if tuple(pal.shape) != (B, K, 3):
        raise ValueError(f"palette_norm must be (B, K, 3), got {tuple(pal.shape)}")
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred_c) if want_grad else None
    scratch = torch.empty(lib.bsg_loss_scratch_bytes(h, W), dtype=torch.uint8, device=pred.device)
    with torch.cuda.device(pred.device):
        N.check(lib.bsg_loss_fwd_bwd_ids(_stream(), B, h, W, K, _ptr(pred_c), _ptr(ids), _ptr(pal), float(beta),
                                         VARIANTS[variant], _ptr(loss), _ptr(grad), _ptr(scratch), scratch.numel()))
    return loss[0], grad


class _LossIdsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, class_ids, palette_norm, beta, variant):
        loss, ctx.grad = loss_fwd_bwd_ids(pred, class_ids, palette_norm, beta, variant, pred.requires_grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None, None


def seggpt_loss_ids(pred, class_ids, palette_norm, beta: float, variant: str = "reference") -> torch.Tensor:
    """`SegGptLoss.forward(pred, Normalize(torch_apply_mask_rgb(palette, mask)), mask != 0)` (`src/model.py:238-239, 255`)
    without materialising the f32 label image: the kernel reads the class id and looks the colour up."""
    _need_gpu(pred, class_ids, palette_norm)
    if variant not in VARIANTS:
        raise ValueError(f"variant must be one of {list(VARIANTS)}")
    return _LossIdsFn.apply(pred, class_ids, palette_norm, beta, variant)


def mask_rgb_norm(palette: torch.Tensor, class_ids: torch.Tensor, mean=IMAGE_MEAN, std=IMAGE_STD) -> torch.Tensor:
    """`normalize(torch_apply_mask_rgb(palette, input))` (`src/util/ml_util.py:114-132` + `src/data.py:345`; call sites
    `src/model.py:211-212, 238-239`) in ONE HIP kernel: palette u8 (B,K,3), class ids u8 (B,1,H,W) / (B,H,W) -> f32
    (B,3,H,W), bit-exact against the reference's float32 CPU arithmetic.  mean=(0,0,0), std=(1,1,1): the un-normalised
    [0,1] image of `torch_apply_mask_rgb` alone."""
    _need_gpu(palette, class_ids)
    if palette.dtype != torch.uint8 or palette.dim() != 3 or palette.shape[2] != 3:
        raise ValueError("palette must be uint8 (B, K, 3)")
    ids = class_ids[:, 0] if class_ids.dim() == 4 else class_ids
    if ids.dim() != 3 or ids.shape[0] != palette.shape[0]:
        raise ValueError(f"class ids must be (B,1,H,W) or (B,H,W) with B = {palette.shape[0]}, got {tuple(class_ids.shape)}")
    lib = N.load()
    B, h, w = ids.shape
    ids = ids.to(torch.uint8).contiguous()
    out = torch.empty((B, 3, h, w), dtype=torch.float32, device=ids.device)
    if out.numel() == 0:  # empty batch / empty plane: torch's gather returns the empty tensor, nothing to launch
        return out
    with torch.cuda.device(ids.device):
        N.check(lib.bsg_mask_rgb_norm(_stream(), B, h, w, palette.shape[1], _ptr(ids), _ptr(palette.contiguous()), _f3(mean),
                                      _f3(std), _ptr(out)))
    return out


def decode_argmin(pred: torch.Tensor, palette_norm: torch.Tensor, out_dtype=torch.int64) -> torch.Tensor:
    """`process_pred_masks` (`src/model.py:155-175`) -> (B,H,W) int64 (reference dtype) or uint8."""
    _need_gpu(pred, palette_norm)
    lib = N.load()
    B, _, H2, W = pred.shape
    h, K = H2 // 2, palette_norm.shape[1]
    pred_c = pred.detach().float().contiguous()
    pal = palette_norm.detach().float().contiguous()
    out = torch.empty((B, h, W), dtype=out_dtype, device=pred.device)
    with torch.cuda.device(pred.device):
        N.check(lib.bsg_decode_argmin(_stream(), B, h, W, K, _ptr(pred_c), _ptr(pal),
                                      _ptr(out) if out_dtype == torch.int64 else None,
                                      _ptr(out) if out_dtype == torch.uint8 else None))
    return out


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def prompt_gather(params: torch.Tensor, idx: torch.Tensor, mean=IMAGE_MEAN, std=IMAGE_STD) -> torch.Tensor:
    """stack(params[idx]) then Normalize: (P,3,h,w) f32, idx (B,) -> (B,3,h,w) f32."""
    _need_gpu(params, idx)
    lib = N.load()
    B, (_, _, h, w) = idx.numel(), params.shape
    out = torch.empty((B, 3, h, w), dtype=torch.float32, device=params.device)
    idx32 = idx.to(torch.int32).contiguous()
    with torch.cuda.device(params.device):
        N.check(lib.bsg_prompt_gather(_stream(), B, h, w, _ptr(params), _ptr(idx32), _f3(mean), _f3(std), _ptr(out)))
    return out


def prompt_grad_scatter(grad_pixels: torch.Tensor, idx: torch.Tensor, grad_params: torch.Tensor, std=IMAGE_STD) -> None:
    """grad_params[idx[b]] += grad_pixels[b] / std  (duplicates accumulate, as autograd through stack does)."""
    _need_gpu(grad_pixels, idx, grad_params)
    lib = N.load()
    B, _, h, w = grad_pixels.shape
    idx32 = idx.to(torch.int32).contiguous()
    with torch.cuda.device(grad_params.device):
        N.check(lib.bsg_prompt_grad_scatter(_stream(), B, h, w, _ptr(grad_pixels.contiguous()), _ptr(idx32), _f3(std),
                                            _ptr(grad_params)))


def adamw_step(params, grads, exp_avg, exp_avg_sq, active: torch.Tensor, steps, lr: float,
               betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, grad_scale: float = 1.0,
               touched: torch.Tensor | None = None) -> None:
    """torch.optim.AdamW step on rows `active` of the flat (P, n) buffers; `steps[a]` = that row's step count
    after this update (list of ints, or a device tensor: no host sync).  Bias corrections are formed in double,
    as torch does.  `touched` (u8 per ROW) skips rows without a gradient."""
    _need_gpu(params, grads, exp_avg, exp_avg_sq, active)
    lib = N.load()
    n_active = active.numel()
    t = torch.as_tensor(steps, dtype=torch.float64, device=params.device)
    ss = (lr / (1.0 - betas[0] ** t)).to(torch.float32)
    b2 = ((1.0 - betas[1] ** t) ** 0.5).to(torch.float32)
    act = active.to(torch.int32).contiguous()
    row = params[0].numel()
    with torch.cuda.device(params.device):
        N.check(lib.bsg_adamw_step(_stream(), n_active, row, _ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq),
                                   _ptr(act), _ptr(touched), _ptr(ss), _ptr(b2), lr, betas[0], betas[1], eps,
                                   weight_decay, grad_scale))


def vote_paste(counter: torch.Tensor, masks: torch.Tensor, crops: torch.Tensor, crop_size: int) -> None:
    """`cv2.resize(INTER_NEAREST)` + `np.eye(K)[pred]` + `Accumulator.update` for a set of NON-overlapping crops.
    counter u8 (mh,mw,K); masks u8 (n,hin,win); crops i32 (n,4) = (xmin,ymin,xmax,ymax)."""
    _need_gpu(counter, masks, crops)
    lib = N.load()
    n, hin, win = masks.shape
    mh, mw, K = counter.shape
    with torch.cuda.device(counter.device):
        N.check(lib.bsg_vote_paste(_stream(), n, _ptr(masks.contiguous()), hin, win, crop_size,
                                   _ptr(crops.to(torch.int32).contiguous()), _ptr(counter), mh, mw, K))


def vote_argmax(counter: torch.Tensor) -> torch.Tensor:
    """`np.argmax(counter, axis=2)` (`src/predict.py:100`) -> u8 (mh,mw)."""
    _need_gpu(counter)
    lib = N.load()
    mh, mw, K = counter.shape
    out = torch.empty((mh, mw), dtype=torch.uint8, device=counter.device)
    with torch.cuda.device(counter.device):
        N.check(lib.bsg_vote_argmax(_stream(), _ptr(counter), mh * mw, K, _ptr(out)))
    return out


_FRONTEND_TABLES: dict = {}


def tile_frontend(mosaic: torch.Tensor, crops: torch.Tensor, crop_size: int, out_size: int,
                  mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), return_u8: bool = False):
    """Device tile front-end (`src/data.py:88-96`): u8 (H,W,3) mosaic + i32 (n,4) windows (xmin,ymin,..; side =
    crop_size, zero padding outside) -> f32 (n,3,S,S) = Normalize(PIL-BICUBIC resize / 255), bit-for-bit what
    `Image.resize(resample=BICUBIC)` + `/255` + `(x-mean)/std` give on the host.  `return_u8` also returns the resized
    bytes u8 (n,S,S,3)."""
    import ctypes as C

    from .data import pil_bicubic_tables
    _need_gpu(mosaic, crops)
    if mosaic.dtype != torch.uint8 or mosaic.dim() != 3 or mosaic.shape[2] != 3 or not mosaic.is_contiguous():
        raise ValueError("mosaic must be a contiguous uint8 (H, W, 3) tensor")
    if crops.dtype != torch.int32 or crops.dim() != 2 or crops.shape[1] != 4:
        raise ValueError("crops must be int32 (n, 4)")
    lib = N.load()
    key = (crop_size, out_size, mosaic.device)
    if key not in _FRONTEND_TABLES:
        b, k = pil_bicubic_tables(crop_size, out_size)
        _FRONTEND_TABLES[key] = (torch.from_numpy(b).to(mosaic.device), torch.from_numpy(k).to(mosaic.device), k.shape[1])
    bounds, coef, kmax = _FRONTEND_TABLES[key]
    n = crops.shape[0]
    out = torch.empty((n, 3, out_size, out_size), dtype=torch.float32, device=mosaic.device)
    u8 = torch.empty((n, out_size, out_size, 3), dtype=torch.uint8, device=mosaic.device) if return_u8 else None
    f3 = C.c_float * 3
    crops = crops.contiguous()
    with torch.cuda.device(mosaic.device):
        N.check(lib.bsg_tile_frontend(_stream(), _ptr(mosaic), mosaic.shape[0], mosaic.shape[1], n, _ptr(crops), crop_size,
                                      out_size, _ptr(coef), _ptr(bounds), kmax, f3(*mean), f3(*std), _ptr(out),
                                      _ptr(u8) if u8 is not None else None))
    return (out, u8) if return_u8 else out


def post_process_semantic_segmentation(pred_masks: torch.Tensor, num_labels: int, target_sizes=None,
                                       mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> list[torch.Tensor]:
    """`SegGptImageProcessor.post_process_semantic_segmentation(outputs, target_sizes, num_labels)`
    (`HF:image_processing_seggpt.py:300-332`) as the reference's few-shot caller uses it
    (`src/predict_no_prompt.py:297-303`): pred_masks f32 (B,3,2H,W) -> list of B int64 maps, (H,W) or the target sizes.
    The decode runs in one HIP kernel at full resolution; the nearest resize (F.interpolate's index rule
    floor(dst * in / out)) commutes with the per-pixel arg-min and is applied to the class map."""
    import ctypes as C

    from .ml_util import build_palette
    _need_gpu(pred_masks)
    B, ch, H2, W = pred_masks.shape
    if ch != 3 or H2 % 2:
        raise ValueError("pred_masks must be (B, 3, 2H, W)")
    if target_sizes is not None and len(target_sizes) != B:
        raise ValueError("Make sure that you pass in as many target sizes as the batch dimension of the logits")
    lib = N.load()
    pred = pred_masks.detach().float().contiguous()
    pal = torch.tensor(build_palette(num_labels), dtype=torch.float32, device=pred.device)
    out = torch.empty((B, H2 // 2, W), dtype=torch.uint8, device=pred.device)
    f3 = C.c_float * 3
    with torch.cuda.device(pred.device):
        N.check(lib.bsg_decode_hf(_stream(), B, H2 // 2, W, num_labels + 1, _ptr(pred), _ptr(pal), f3(*mean), f3(*std),
                                  _ptr(out)))
    res = []
    for i in range(B):
        m = out[i]
        if target_sizes is not None:
            th, tw = target_sizes[i]
            ys = (torch.arange(th, device=m.device) * (m.shape[0] / th)).floor().long().clamp_(max=m.shape[0] - 1)
            xs = (torch.arange(tw, device=m.device) * (m.shape[1] / tw)).floor().long().clamp_(max=m.shape[1] - 1)
            m = m[ys][:, xs]
        res.append(m.long())
    return res


_DTYPE_CODE = {torch.float32: N.BSG_DTYPE_F32, torch.bfloat16: N.BSG_DTYPE_BF16, torch.float16: N.BSG_DTYPE_F16}


def x3_weight(w: torch.Tensor) -> torch.Tensor:
    """f32 (N, K) -> the pre-split weight format of `bsg_config.gemm_x3`: w x 2^5, every 16-byte chunk of a row holding
    [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] (f16) of its four values, viewed as float32 (N, K)."""
    w = (w.detach().float() * 32.0).contiguous()
    hi = w.half()
    lo = (w - hi.float()).half()
    n, k = w.shape
    return torch.cat([hi.view(n, k // 4, 4), lo.view(n, k // 4, 4)], dim=2).contiguous().view(torch.float32).view(n, k)


def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, x3: bool = False) -> torch.Tensor:
    """out = a @ w.T (+ bias) on the hand-written MFMA GEMM: a (M,K), w (N,K), both bf16, both f16 or both f32.  `x3` (f32
    only): three f16 MFMAs on 22-bit operand splits instead of exact-f32 MFMAs (the weight is pre-split here)."""
    _need_gpu(a, w, bias)
    lib = N.load()
    if a.dtype != w.dtype or a.dtype not in _DTYPE_CODE:
        raise ValueError("a and w must both be float32, both bfloat16 or both float16")
    if x3 and a.dtype != torch.float32:
        raise ValueError("x3 applies to float32 operands")
    M, K = a.shape
    out = torch.empty((M, w.shape[0]), dtype=a.dtype, device=a.device)
    wk = x3_weight(w) if x3 else w.contiguous()
    with torch.cuda.device(a.device):
        N.check(lib.bsg_op_gemm(_stream(), 3 if x3 else _DTYPE_CODE[a.dtype], M, w.shape[0], K, _ptr(a.contiguous()),
                                _ptr(wk), _ptr(bias), _ptr(out)))
    return out


def gemm_nt_epilogue(epilogue: str, a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None,
                     aux: torch.Tensor | None = None, save_grad: bool = True):
    """The NT GEMM with one of the encoder blocks' fused epilogues (bf16 / f16 operands): "gelu" -> (gelu(a w^T + bias),
    gelu'(...) or None); "residual" -> a w^T + bias + aux in float32 (a new tensor); "gelu_bwd" -> (a w^T) * aux."""
    _need_gpu(a, w, bias, aux)
    lib = N.load()
    if a.dtype != w.dtype or a.dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("a and w must both be bfloat16 or both float16")
    code = {"gelu": 1, "residual": 2, "gelu_bwd": 3}[epilogue]
    M, K = a.shape
    n = w.shape[0]
    out2 = None
    if code == 2:
        out = aux.to(torch.float32).clone().contiguous()  # in place on the copy: the way the residual stream is updated
        auxp = out
    else:
        out = torch.empty((M, n), dtype=a.dtype, device=a.device)
        auxp = aux.contiguous() if aux is not None else None
        if code == 1 and save_grad:
            out2 = torch.empty_like(out)
    with torch.cuda.device(a.device):
        N.check(lib.bsg_op_gemm_epilogue(_stream(), _DTYPE_CODE[a.dtype], code, M, n, K, _ptr(a.contiguous()), _ptr(w.contiguous()),
                                         _ptr(bias), _ptr(auxp), _ptr(out), _ptr(out2)))
    return (out, out2) if code == 1 else out


def attention_scratch(S: int, nh: int, hp: int, device) -> torch.Tensor:
    return torch.zeros(N.load().bsg_op_attention_scratch_bytes(S, nh, hp), dtype=torch.uint8, device=device)


def attention(which: int, qkv: torch.Tensor, rel_cat: torch.Tensor, S: int, nh: int, hp: int, wp: int, out: torch.Tensor,
              lse2: torch.Tensor, scratch: torch.Tensor, rel_catT: torch.Tensor | None = None,
              dout: torch.Tensor | None = None, dqkv: torch.Tensor | None = None,
              windows: tuple[int, int, int, int] | None = None) -> None:
    """The fused attention kernels on their own (`bsg_op_attention`): bit 0 forward, bit 1 dQ, bit 2 dK/dV (bits 3-5: its A/B
    variants, see the header).  bf16 or f16 tensors:
    qkv (S*N, 3*nh*64), rel_cat ([LH+LW], 64) / rel_catT, dout / out (S*N, nh*64), dqkv like qkv; lse2 f32 (S, nh, hp*32).
    `windows` = (dq_begin, dq_end, q_begin, key_rows): the row windows of `bsg_backward_rows` for this call (header)."""
    _need_gpu(qkv, rel_cat, out, lse2, scratch)
    lib = N.load()
    with torch.cuda.device(qkv.device):
        if qkv.dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("the stand-alone attention entry takes bfloat16 or float16 tensors")
        if windows is not None:
            N.check(lib.bsg_op_attention_windows(*(int(v) for v in windows)))
        N.check(lib.bsg_op_attention(_stream(), _DTYPE_CODE[qkv.dtype], which, S, nh, hp, wp, _ptr(qkv), _ptr(rel_cat), _ptr(rel_catT), _ptr(dout),
                                     _ptr(out), _ptr(lse2), _ptr(dqkv), _ptr(scratch), scratch.numel()))


def tif_image(bands: torch.Tensor, nodata: torch.Tensor | None = None) -> torch.Tensor:
    """Device `tif_image` (`src/util/geo_util.py:449-470`): bands f32 or u16 (C,H,W), C in {4, 8}; nodata bool/u8 (H,W) or
    None -> u8 (H,W,3), the mosaic `tile_frontend` consumes.  4 bands: bit-exact against the reference's own output."""
    _need_gpu(bands, nodata)
    if bands.dim() != 3 or bands.shape[0] not in (4, 8):
        raise ValueError(f"expected a (4|8, H, W) raster, got {tuple(bands.shape)}")
    if bands.dtype == torch.float32:
        dt = 0
    elif bands.dtype == torch.uint16:
        dt = 1
    else:
        raise ValueError("bands must be float32 (what the reference reads) or uint16")
    lib = N.load()
    Cb, H, W = bands.shape
    nd = nodata.to(torch.uint8).contiguous() if nodata is not None else None
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=bands.device)
    scratch = torch.empty(32, dtype=torch.uint8, device=bands.device)
    with torch.cuda.device(bands.device):
        N.check(lib.bsg_tif_image(_stream(), Cb, H, W, dt, _ptr(bands.contiguous()), _ptr(nd), _ptr(out), _ptr(scratch)))
    return out


class _TrainAugFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, mask, params, noise, mean, std, color):
        lib = N.load()
        B, _, h, w = img.shape
        img_c = img.detach().float().contiguous()
        out = torch.empty_like(img_c)
        mask_c = mask.to(torch.uint8).contiguous() if mask is not None else None
        mask_out = torch.empty_like(mask_c) if mask_c is not None else None
        scratch = torch.empty_like(img_c) if color is not None else None
        with torch.cuda.device(img.device):
            N.check(lib.bsg_train_aug(_stream(), B, h, w, _ptr(img_c), _ptr(mask_c), _ptr(params), _ptr(color), _ptr(noise),
                                      _f3(mean), _f3(std), _ptr(out), _ptr(mask_out), _ptr(scratch)))
        ctx.params, ctx.std, ctx.shape, ctx.color = params, std, (B, h, w), color
        ctx.img = img_c if color is not None else None  # the colour chain's Jacobian is evaluated at the input pixels
        ctx.mark_non_differentiable(*([mask_out] if mask_out is not None else []))
        return (out, mask_out) if mask_out is not None else (out, None)

    @staticmethod
    def backward(ctx, gout, _gmask=None):
        lib = N.load()
        B, h, w = ctx.shape
        g = gout.contiguous().float()
        gin = torch.empty_like(g)
        scratch = torch.empty((2,) + tuple(g.shape), dtype=torch.float32, device=g.device) if ctx.color is not None else None
        with torch.cuda.device(g.device):
            N.check(lib.bsg_train_aug_bwd(_stream(), B, h, w, _ptr(g), _ptr(ctx.img), _ptr(ctx.params), _ptr(ctx.color),
                                          _f3(ctx.std), _ptr(gin), _ptr(scratch)))
        return gin, None, None, None, None, None, None


def train_aug(img: torch.Tensor, mask: torch.Tensor | None, params: torch.Tensor, noise: torch.Tensor | None = None,
              mean=IMAGE_MEAN, std=IMAGE_STD, color: torch.Tensor | None = None):
    """The train-time augmentation chain of `src/data.py:195-224` with explicit random parameters (`bsg_train_aug`):
    img f32 (B,3,h,w) in [0,1] (autograd-tracked: the stacked prompt Parameters), mask u8 (B,h,w) / (B,1,h,w) or None,
    params i32 (B,5) and color f32 (B,6) or None from `data.sample_train_aug_params`, noise f32 (B,3,h,w) or None ->
    (normalised image, mask).  color = [brightness, contrast, saturation, hue, sharpness factor, order code] switches on
    ColorJiggle (params flag bit 4) and RandomSharpness (bit 3): kornia's published formulas, parity unpinned."""
    _need_gpu(img, mask, params, noise, color)
    if params.dtype != torch.int32 or tuple(params.shape) != (img.shape[0], 5):
        raise ValueError("params must be int32 (B, 5)")
    if color is not None and (color.dtype != torch.float32 or tuple(color.shape) != (img.shape[0], 6)):
        raise ValueError("color must be float32 (B, 6)")
    m = mask
    if m is not None and m.dim() == 4:
        m = m[:, 0]
    out, mo = _TrainAugFn.apply(img, m, params.contiguous(), noise.contiguous() if noise is not None else None, tuple(mean),
                                tuple(std), color.contiguous() if color is not None else None)
    if mo is not None and mask.dim() == 4:
        mo = mo[:, None]
    return out, mo


def confusion_update(confmat: torch.Tensor, pred: torch.Tensor, target: torch.Tensor, ignore_index: int | None) -> None:
    """confmat i64 (K,K) += counts of (target, pred) pairs with target != ignore_index: one kernel, no host sync
    (`MulticlassF1Score.update`, `src/model.py:256, 295`)."""
    _need_gpu(confmat, pred, target)
    if confmat.dtype != torch.int64 or confmat.dim() != 2 or confmat.shape[0] != confmat.shape[1]:
        raise ValueError("confmat must be int64 (K, K)")
    lib = N.load()
    t = target.to(torch.uint8).contiguous()
    p = pred.contiguous()
    if p.dtype not in (torch.int64, torch.uint8):
        p = p.to(torch.int64)
    if p.numel() != t.numel():
        raise ValueError("pred and target must have the same number of pixels")
    with torch.cuda.device(confmat.device):
        N.check(lib.bsg_confusion_update(_stream(), t.numel(), confmat.shape[0], -1 if ignore_index is None else ignore_index,
                                         _ptr(p) if p.dtype == torch.int64 else None,
                                         _ptr(p) if p.dtype == torch.uint8 else None, _ptr(t), _ptr(confmat)))
