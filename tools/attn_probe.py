"""Attention micro-benchmark + check against a plain torch fp32 reference of the same op (GPU box):
    python tools/attn_probe.py [S=64] [check]
Times attn_fwd / attn_bwd_dq / attn_bwd_dkv of the library BSG_AB_LIB points at (tools/_ab.py; default: the in-tree build) on the
ViT-L geometry (16 heads, 56 x 28 tokens), random N(0,1)-scaled q/k/v as LayerNorm+QKV would produce them."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tools._ab import pick_lib
pick_lib()
from beach_seg_amd import ops
from beach_seg_amd.seggpt import _rel_cat

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
check = len(sys.argv) > 2
nh, hp, wp = 16, 56, 28
N, D = hp * wp, nh * 64
g = torch.Generator(device=dev).manual_seed(3)
qkv = (torch.randn(S * N, 3 * D, device=dev, generator=g) * 0.8).bfloat16()
dout = (torch.randn(S * N, D, device=dev, generator=g) * 1e-3).bfloat16()
rel_h = torch.randn(2 * hp - 1, 64, device=dev, generator=g) * 0.3
rel_w = torch.randn(2 * wp - 1, 64, device=dev, generator=g) * 0.3
rc = _rel_cat(rel_h, rel_w).bfloat16().contiguous()
rcT = rc.t().contiguous()
out = torch.empty(S * N, D, device=dev, dtype=torch.bfloat16)
lse2 = torch.zeros(S, nh, hp * 32, device=dev)
dqkv = torch.zeros_like(qkv)
scratch = ops.attention_scratch(S, nh, hp, dev)
call = lambda w: ops.attention(w, qkv, rc, S, nh, hp, wp, out, lse2, scratch, rcT, dout, dqkv)
call(7)
torch.cuda.synchronize()
res = {}
for name, w, fl in (("fwd", 1, 4.0), ("dq", 2, 4.0), ("dkv", 4, 4.0), ("dkv_1wave", 8, 4.0), ("dkv_8wave", 16, 4.0), ("dkv_2x4wave", 32, 4.0)):  # algorithmic: fwd 4 N^2 d, bwd 8 N^2 d over its two kernels
    for _ in range(2):
        call(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        call(w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    res[name] = ms
    print(f"{name}: {ms:.3f} ms  {fl * S * nh * N * N * 64 / ms / 1e9:.0f} TFLOP/s algorithmic", flush=True)
print("RESULT", " ".join(f"{k}={v:.3f}" for k, v in res.items()), flush=True)

# the one-wave-per-SIMD dK / dV kernel against the eight-wave one on the same tables: same arithmetic, same order
call(2 | 16)
ref_kv = dqkv[:, D:].clone()
dqkv[:, D:].zero_()
call(8)
torch.cuda.synchronize()
print("one-wave-per-SIMD dkv vs eight-wave dkv: max abs diff", float((dqkv[:, D:].float() - ref_kv.float()).abs().max()), "bit-identical", torch.equal(dqkv[:, D:], ref_kv), flush=True)

dqkv[:, D:].zero_()
call(4)
torch.cuda.synchronize()
print("default dkv vs eight-wave dkv: bit-identical", torch.equal(dqkv[:, D:], ref_kv), flush=True)

if check:
    s_chk = min(S, 2)
    x = qkv[: s_chk * N].float().reshape(s_chk, N, 3, nh, 64).permute(2, 0, 3, 1, 4)  # (3,S,nh,N,64)
    q, k, v = (t.clone().requires_grad_(True) for t in (x[0], x[1], x[2]))
    rh, rw = rel_h.bfloat16().float(), rel_w.bfloat16().float()
    ih = torch.arange(hp, device=dev)[:, None] - torch.arange(hp, device=dev)[None, :] + hp - 1
    iw = torch.arange(wp, device=dev)[:, None] - torch.arange(wp, device=dev)[None, :] + wp - 1
    Rh, Rw = rh[ih], rw[iw]  # (hp,hp,64), (wp,wp,64)
    qg = q.reshape(s_chk, nh, hp, wp, 64)
    relh = torch.einsum("snhwc,hkc->snhwk", qg, Rh)
    relw = torch.einsum("snhwc,wkc->snhwk", qg, Rw)
    att = (q * 0.125) @ k.transpose(-2, -1)
    att = (att.reshape(s_chk, nh, hp, wp, hp, wp) + relh[..., :, None] + relw[..., None, :]).reshape(s_chk, nh, N, N)
    o = torch.softmax(att, -1) @ v  # (S,nh,N,64)
    o_rows = o.permute(0, 2, 1, 3).reshape(s_chk * N, D)
    do = dout[: s_chk * N].float()
    o_rows.backward(do)
    rel = lambda a, b: float((a.float() - b).abs().max() / b.abs().max())
    print("check fwd rel err", rel(out[: s_chk * N], o_rows.detach()))
    gq = q.grad.permute(0, 2, 1, 3).reshape(s_chk * N, D)
    gk = k.grad.permute(0, 2, 1, 3).reshape(s_chk * N, D)
    gv = v.grad.permute(0, 2, 1, 3).reshape(s_chk * N, D)
    print("check dq rel err", rel(dqkv[: s_chk * N, :D], gq), "dk", rel(dqkv[: s_chk * N, D:2 * D], gk), "dv",
          rel(dqkv[: s_chk * N, 2 * D:], gv))
