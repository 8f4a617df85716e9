"""`-m gpu`: unit numerics of the hand-written HIP kernels on their own, through the C ABI (`bsg_op_gemm`,
`bsg_op_attention`), against a plain PyTorch fp32 reference of the same op on the same (already rounded) operands."""
import pytest
import torch

from beach_seg_amd import ops
from beach_seg_amd.seggpt import _rel_cat

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1000, 192, 128), (777, 1024, 1024), (2 * 1568, 3072, 1024), (4099, 320, 256),
                                   (16 * 1568, 1024, 256), (25000, 1024, 192), (16 * 1568, 768, 128), (8 * 1568, 2048, 512)])
def test_gemm_nt_vs_torch(dtype, M, N, K):
    """Every GEMM variant: v5 (16-bit dtypes, M and N multiples of 256, an even number of K tiles: (256, 256, 256)-class shapes,
    (16 * 1568, 1024, 256) = 392 tiles in two persistent rounds, (16 * 1568, 768, 128) = only the two DMA-less last K tiles,
    (8 * 1568, 2048, 512)), v3 (256 x 256 persistent: f32, ragged or odd-K-tile shapes), v2 (N <= 192), ragged M / N edges, bias
    epilogue; (16 * 1568, 1024, 256), (25000, 1024, 192) and (16 * 1568, 768, 128) have a ragged second round of 256-row tiles,
    a ragged last row tile and a ragged last column tile.  v5 against v3 bit for bit, every epilogue, forward and backward:
    `test_gpu_fullsize.py::test_vit_large_b64_engine_call_sequence_matches_b1_bit_for_bit` (B = 64 runs v5, B = 1 v3)."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = (torch.rand(M, K, device=DEV, generator=g) * 2 - 1).to(dtype)
    w = (torch.rand(N, K, device=DEV, generator=g) * 2 - 1).to(dtype)
    b = torch.randn(N, device=DEV, generator=g)
    ref = a.double() @ w.double().t()
    out = ops.gemm_nt(a, w)
    outb = ops.gemm_nt(a, w, b)
    # 16-bit: output rounding 2^-9 (bf16) / 2^-12 (f16) of a value up to max|ref|; operands are exact in every dtype
    tol = {torch.float32: 1e-5, torch.bfloat16: 6e-3, torch.float16: 8e-4}[dtype]
    assert rel(out, ref) < tol and rel(outb, ref + b.double()) < tol
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(512, 768, 256), (8 * 1568, 1024, 128), (2048, 1024, 1024), (64 * 1536, 1024, 1024), (777, 1024, 256), (1000, 320, 128)])
def test_gemm_fused_epilogues_vs_torch(dtype, M, N, K):
    """fc1 (bias + erf GELU, with the saved derivative), proj / fc2 (bias + fp32 residual, in place) and dfc2 (times the saved
    gelu') epilogues against float64 torch, on shapes that take `gemm_nt_kernel_v5` (the first four: M, N multiples of 256, 2 /
    2 / 16 / 16 K tiles, one of them two persistent rounds, one the six full rounds of the train step's N = 1024 GEMMs) and on ragged ones that take v3.  The two kernels run the same epilogue
    arithmetic, so the same tolerances hold: output rounding of the 16-bit results, ~1e-6 for the fp32 residual."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = (torch.rand(M, K, device=DEV, generator=g) * 2 - 1).to(dtype)
    w = ((torch.rand(N, K, device=DEV, generator=g) * 2 - 1) * (3.0 / K ** 0.5)).to(dtype)  # pre-activations of a few units
    b = torch.randn(N, device=DEV, generator=g)
    acc = a.double() @ w.double().t()
    x = acc + b.double()
    eps = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    # bias + GELU and its derivative
    y, dy = ops.gemm_nt_epilogue("gelu", a, w, b)
    cdf = 0.5 * (1 + torch.erf(x / 2 ** 0.5))
    pdf = torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    assert rel(y, x * cdf) < 1.5 * eps and rel(dy, cdf + x * pdf) < 1.5 * eps
    y1, none = ops.gemm_nt_epilogue("gelu", a, w, b, save_grad=False)
    assert none is None and torch.equal(y1, y)
    # bias + fp32 residual
    r = torch.randn(M, N, device=DEV, generator=g) * 3
    out = ops.gemm_nt_epilogue("residual", a, w, b, aux=r)
    assert out.dtype == torch.float32 and rel(out, x + r.double()) < 2e-6
    # gradient times the saved derivative
    h = (torch.rand(M, N, device=DEV, generator=g) * 1.2 - 0.1).to(dtype)
    d = ops.gemm_nt_epilogue("gelu_bwd", a, w, aux=h)
    assert rel(d, acc * h.double()) < 1.5 * eps
    assert all(torch.isfinite(t.float()).all() for t in (y, dy, out, d))
    # run to run: the same launch gives the same bits (a variant of v5 whose epilogue spilled accumulators to scratch did not)
    y2, dy2 = ops.gemm_nt_epilogue("gelu", a, w, b)
    assert torch.equal(y2, y) and torch.equal(dy2, dy)
    assert torch.equal(ops.gemm_nt_epilogue("residual", a, w, b, aux=r), out)
    assert torch.equal(ops.gemm_nt_epilogue("gelu_bwd", a, w, aux=h), d)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (777, 1024, 1024), (2 * 1568, 3072, 1024), (4099, 320, 256), (25000, 1024, 192),
                                   (64 * 1568, 1024, 256)])  # the last: 6.125 rounds -- full rounds on the 256 x 256 kernel + the tail kernel's rows
def test_gemm_x3_vs_exact(M, N, K):
    """`gemm_nt_kernel_v3<float, ..., X3>`: float32 operands as hi = f16(x), lo = f16(x - hi), three f16 MFMAs per exact-f32
    group (bsg_config.gemm_x3), weights pre-split by the host: against a float64 product -- 22-bit operands leave ~2^-21 of
    the row norm, i.e. within a small factor of the exact-f32 kernel's own error -- on full, ragged-M, ragged-N and multi-round
    shapes, with and without the bias epilogue."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.rand(M, K, device=DEV, generator=g) * 2 - 1
    w = (torch.rand(N, K, device=DEV, generator=g) * 2 - 1) * 0.05  # weight-like magnitudes: lo parts need the 2^5 pre-scale
    b = torch.randn(N, device=DEV, generator=g)
    ref = a.double() @ w.double().t()
    exact, x3, x3b = ops.gemm_nt(a, w), ops.gemm_nt(a, w, x3=True), ops.gemm_nt(a, w, b, x3=True)
    e_exact, e_x3 = rel(exact, ref), rel(x3, ref)
    print(f"[measured] gemm x3 {M}x{N}x{K}: exact-f32 kernel {e_exact:.1e}, x3 {e_x3:.1e}")
    assert e_x3 < 2e-6 and rel(x3b, ref + b.double()) < 2e-6 and e_exact < 2e-6
    assert torch.isfinite(x3).all()
    if M == 64 * 1568:  # the tail rows (the last 2,048) on their own: the same accuracy as the rows of the full rounds
        t0 = M - 2048
        assert rel(x3[t0:], ref[t0:]) < 2e-6 and rel(exact[t0:], ref[t0:]) < 2e-6 and rel(x3[:t0], ref[:t0]) < 2e-6


def _attention_reference(qkv, rel_h, rel_w, dout, S, nh, hp, wp):
    N, D = hp * wp, nh * 64
    x = qkv.float().reshape(S, N, 3, nh, 64).permute(2, 0, 3, 1, 4)
    q, k, v = (t.clone().requires_grad_(True) for t in (x[0], x[1], x[2]))
    dev = qkv.device
    ih = torch.arange(hp, device=dev)[:, None] - torch.arange(hp, device=dev)[None, :] + hp - 1
    iw = torch.arange(wp, device=dev)[:, None] - torch.arange(wp, device=dev)[None, :] + wp - 1
    qg = q.reshape(S, nh, hp, wp, 64)
    relh = torch.einsum("snhwc,hkc->snhwk", qg, rel_h[ih])  # unscaled q (HF:268-311)
    relw = torch.einsum("snhwc,wkc->snhwk", qg, rel_w[iw])
    att = (q * 0.125) @ k.transpose(-2, -1)
    att = (att.reshape(S, nh, hp, wp, hp, wp) + relh[..., :, None] + relw[..., None, :]).reshape(S, nh, N, N)
    o = (torch.softmax(att, -1) @ v).permute(0, 2, 1, 3).reshape(S * N, D)
    o.backward(dout.float())
    rows = lambda t: t.permute(0, 2, 1, 3).reshape(S * N, D)
    return o.detach(), rows(q.grad), rows(k.grad), rows(v.grad), float((att.max(-1).values - att.mean(-1)).max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hp,wp,S,nh,gain", [(8, 8, 3, 2, 1.0), (56, 28, 1, 2, 1.0), (56, 28, 1, 2, 6.0), (64, 32, 1, 1, 3.0)])
def test_attention_kernels_vs_torch(hp, wp, S, nh, gain, dtype):
    """Forward, dQ, dK / dV with the decomposed rel-pos bias against autograd through the plain formula; `gain` scales q, k
    and the rel-pos tables so that the logits reach tens (peaked rows: the online-softmax rescale path, exp2 range)."""
    N, D = hp * wp, nh * 64
    g = torch.Generator(device=DEV).manual_seed(hp * 100 + wp)
    qkv = torch.randn(S * N, 3 * D, device=DEV, generator=g) * 0.8
    qkv[:, : 2 * D] *= gain
    qkv = qkv.to(dtype)
    # f16 holds 6e-5 .. 65504 at full precision: its backward runs on a gradient of order 1 (the product path scales the
    # dgrad chain by a power of two chosen on device, rowops.hpp), bf16 takes the raw 1e-3
    dout = (torch.randn(S * N, D, device=DEV, generator=g) * (1e-3 if dtype == torch.bfloat16 else 1.0)).to(dtype)
    rel_h = (torch.randn(2 * hp - 1, 64, device=DEV, generator=g) * 0.2 * gain).to(dtype).float()
    rel_w = (torch.randn(2 * wp - 1, 64, device=DEV, generator=g) * 0.2 * gain).to(dtype).float()
    rc = _rel_cat(rel_h, rel_w).to(dtype).contiguous()
    out = torch.empty(S * N, D, device=DEV, dtype=dtype)
    lse2 = torch.zeros(S, nh, hp * 32, device=DEV)
    dqkv = torch.zeros_like(qkv)
    ops.attention(7, qkv, rc, S, nh, hp, wp, out, lse2, ops.attention_scratch(S, nh, hp, DEV), rc.t().contiguous(), dout, dqkv)
    o, gq, gk, gv, peak = _attention_reference(qkv, rel_h, rel_w, dout, S, nh, hp, wp)
    errs = (rel(out, o), rel(dqkv[:, :D], gq), rel(dqkv[:, D:2 * D], gk), rel(dqkv[:, 2 * D:], gv))
    print(f"[measured] attention {dtype} {hp}x{wp} gain {gain}: max logit above row mean {peak:.1f}; fwd {errs[0]:.2e} dq {errs[1]:.2e} "
          f"dk {errs[2]:.2e} dv {errs[3]:.2e}")
    if gain > 1:
        assert peak > 20
    # P / dS / output rounding of the operand type (2^-9 bf16, 2^-12 f16) against an fp32 reference on the same operands
    assert max(errs) < (1.5e-2 if dtype == torch.bfloat16 else 2.5e-3)
    assert torch.isfinite(dqkv.float()).all() and torch.isfinite(lse2).all()
    # the dK / dV kernel variants (one wave per SIMD with 3 / 2 key rows per wave = the default on these grids, eight waves,
    # two four-wave workgroups) run the same arithmetic in the same order on the tables the dQ launch left: bit-identical
    scratch = ops.attention_scratch(S, nh, hp, DEV)
    ops.attention(3, qkv, rc, S, nh, hp, wp, out, lse2, scratch, rc.t().contiguous(), dout, dqkv)
    want = dqkv[:, D:].clone()
    for bit in (8, 16, 32):
        dqkv[:, D:].zero_()
        ops.attention(bit, qkv, rc, S, nh, hp, wp, out, lse2, scratch, rc.t().contiguous(), dout, dqkv)
        if bit == 8:
            ref = dqkv[:, D:].clone()
            assert rel(ref[:, :D], gk) < 1.5e-2 and rel(ref[:, D:], gv) < 1.5e-2
        assert torch.equal(dqkv[:, D:], ref), f"dK / dV variant bit {bit} differs from the one-wave-per-SIMD kernel"
    del want


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hp,wp,S,nh", [(56, 28, 2, 2), (64, 32, 1, 1)])
def test_attention_backward_row_windows_are_exact(hp, wp, S, nh, dtype):
    """The row windows of `bsg_backward_rows` at kernel level (`bsg_op_attention_windows`), against the same kernels without
    them, bit for bit.  (1) Top tap's block: dO is exactly zero on the first token rows -- dQ workgroups wholly below
    `dq_begin` return zero rows (which is what the full kernel computes there), dK / dV streams the queries from `q_begin` on
    and gets the same sums.  (2) Block 0: dq for the queries below `dq_end`, dK / dV for the first Hp / 2 key rows only -- the
    wanted rows carry the same bits, the others are zero (dq) or not written (dk, dv)."""
    N, D = hp * wp, nh * 64
    g = torch.Generator(device=DEV).manual_seed(hp + wp)
    qkv = (torch.randn(S * N, 3 * D, device=DEV, generator=g) * 0.8).to(dtype)
    dout = (torch.randn(S * N, D, device=DEV, generator=g) * (1e-3 if dtype == torch.bfloat16 else 1.0)).to(dtype)
    rel_h = torch.randn(2 * hp - 1, 64, device=DEV, generator=g) * 0.2
    rel_w = torch.randn(2 * wp - 1, 64, device=DEV, generator=g) * 0.2
    rc = _rel_cat(rel_h, rel_w).to(dtype).contiguous()
    rcT = rc.t().contiguous()
    out = torch.empty(S * N, D, device=DEV, dtype=dtype)
    lse2 = torch.zeros(S, nh, hp * 32, device=DEV)
    scratch = ops.attention_scratch(S, nh, hp, DEV)
    run = lambda which, do, dqkv, win=None: ops.attention(which, qkv, rc, S, nh, hp, wp, out, lse2, scratch, rcT, do, dqkv, windows=win)
    run(1, dout, torch.zeros_like(qkv))
    # (1) zero dO on the token rows above row 27 / 31 of the grid
    zero_tok = (hp // 2 - 1) * wp
    dz = dout.clone().reshape(S, N, D)
    dz[:, :zero_tok] = 0
    dz = dz.reshape(S * N, D)
    full, win = torch.zeros_like(qkv), torch.full_like(qkv, 7.0)
    run(6, dz, full)
    run(6, dz, win, (zero_tok // 128 * 128, 0, zero_tok // 64 * 64, 0))
    assert torch.equal(win, full)
    assert float(full.reshape(S, N, 3 * D)[:, :zero_tok, :D].abs().max()) == 0.0  # the premise: dq of a query with zero dO is zero
    # (2) the prompt half only
    q_end, key_rows = min(N, (N // 2 + 127) // 128 * 128), hp // 2
    full = torch.zeros_like(qkv)
    run(6, dout, full)
    win = torch.full_like(qkv, 7.0)
    run(6, dout, win, (0, q_end, 0, key_rows))
    f3, w3 = full.reshape(S, N, 3 * D), win.reshape(S, N, 3 * D)
    assert torch.equal(w3[:, :q_end, :D], f3[:, :q_end, :D]) and float(w3[:, q_end:, :D].abs().max()) == 0.0
    assert torch.equal(w3[:, :key_rows * wp, D:], f3[:, :key_rows * wp, D:])
    assert bool((w3[:, key_rows * wp:, D:] == 7.0).all())  # not written
    assert torch.isfinite(win.float()).all()
    # the eight-wave kernel (what the f32 / x3 modes and other token grids run; bit 4) and the 2 x 4-wave form (bit 5) take the same
    # windows; their key window is rounded up to the row group of a workgroup
    for bit, grp in ((16, 8), (32, 4)):
        rows = (key_rows + grp - 1) // grp * grp * wp
        full_v, win_v = torch.zeros_like(qkv), torch.full_like(qkv, 7.0)
        run(2 | bit, dout, full_v)
        run(2 | bit, dout, win_v, (0, q_end, 0, key_rows))
        fv, wv = full_v.reshape(S, N, 3 * D), win_v.reshape(S, N, 3 * D)
        assert torch.equal(wv[:, :rows, D:], fv[:, :rows, D:]) and bool((wv[:, rows:, D:] == 7.0).all()), bit
        full_v, win_v = torch.zeros_like(qkv), torch.full_like(qkv, 7.0)
        run(2 | bit, dz, full_v)
        run(2 | bit, dz, win_v, (zero_tok // 128 * 128, 0, zero_tok // 64 * 64, 0))
        assert torch.equal(win_v, full_v), bit
