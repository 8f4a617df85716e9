#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: train tiles/sec on synthetic tiles (BASELINE.json).

A "step" is one pass of the hot path over one batch per GPU: gather+normalise the learnable prompts -> SegGPT
ViT-L forward (bf16 MFMA, fp32 accumulate) -> reference SegGptLoss -> dgrad-only backward to the prompt pixels
-> [RCCL all-reduce of the prompt-gradient buffer, N > 1] -> AdamW on the prompts
(= `training_step` + `loss.backward()` + `optimizer.step()` of /root/reference/src/model.py:233-269, :398).
Workload = BASELINE.json configs[1] "bf16 batch=64 on 1 x MI355X, fwd+bwd, synthetic tiles"; with N GPUs every
rank takes its own 64 tiles (configs[2]: global batch 512 at N=8), so scaling is weak.  Inputs are resident in
HBM before the timed region starts.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line with `roofline` (GEMM kernels: algorithmic flops / HIP-event kernel time, measured
live on the launch stream during the timed steps) and, at N=1, `cpu_baseline` (the CPU oracle timed on the host
cores on one tile of the same geometry).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

TRAIN_FLOPS_PER_TILE = 3.2681e12  # forward 1589.7 GF + minimum dgrad 1678.3 GF
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(geometry, threads: int) -> dict:
    """The oracle (CPU restatement of the reference arithmetic, fp32 eager torch) on ONE synthetic tile:
    forward + reference loss + backward to the prompt pixels.  ~10-30 s of host work."""
    import torch

    from beach_seg_amd.weights import synth_state_dict
    from oracle import seggpt_oracle as O
    from oracle.gen_inputs import synth_inputs

    torch.set_num_threads(threads)
    sd = synth_state_dict(geometry, seed=0)
    pix, prm, pm_cls, lb_cls, pal = synth_inputs(geometry, 1, 7)
    pm = O.normalize(O.apply_mask_rgb(pal, pm_cls))
    lab = O.normalize(O.apply_mask_rgb(pal, lb_cls))
    p = prm.clone().requires_grad_(True)
    t0 = time.perf_counter()
    pred = O.forward(sd, geometry, pix, p, pm, labels=lab)
    loss = O.seggpt_loss(pred, lab, (lb_cls != 0)[:, None], 0.01, "reference")
    torch.autograd.grad(loss, p)
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "tiles/s", "cores": threads, "kind": "port",
            "sample": "1 tile (B=1), one fwd + SegGptLoss + bwd step of the ViT-L oracle, fp32 eager torch, "
                      f"{dt:.1f} s, no warm-up"}


def gemm_hbm_traffic_per_launch() -> float | None:
    """HBM bytes per GEMM launch (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 correction) from the committed rocprofv3 --pmc
    summary of this same command (`profiles/r1_step8_pmc_summary.json`, produced by `tools/pmc_summary.py`): PMC
    passes cannot run inside the timed process, so the figure is measured offline and reported here."""
    f = ROOT / "profiles" / "r1_step8_pmc_summary.json"
    if not f.exists():
        return None
    n = b = 0.0
    for k, v in json.loads(f.read_text()).items():
        if "gemm_nt_kernel" in k:
            n += v["launches"]
            b += v["launches"] * (v["hbm_fetch_MB_per_launch"] + v["hbm_write_MB_per_launch"]) * 1e6
    return round(b / n) if n else None


def log(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """Cores this process may actually use (cgroup / affinity share), capped at the 16 a 1-GPU box grants."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="tiles per GPU per step")
    ap.add_argument("--prompts", type=int, default=64, help="number of learnable prompt images P")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--geometry", default="vit_large")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loss-variant", default="reference", choices=["reference", "per_sample"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from beach_seg_amd.engine import PromptTrainEngine
    from beach_seg_amd.seggpt import SegGptNative
    from beach_seg_amd.weights import SegGptGeometry, synth_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    # BSG_BENCH_REHEARSE=1: rehearsal of the N > 1 control flow on a ONE-GPU box (every rank on cuda:0, gloo instead of
    # RCCL, which refuses two ranks on one device); the reported number is meaningless then.
    rehearse = bool(os.environ.get("BSG_BENCH_REHEARSE"))
    dev = torch.device("cuda:0" if rehearse else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL

    g = getattr(SegGptGeometry, args.geometry)()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    log(f"rank {rank}/{world}: building {args.geometry} ({args.dtype}) on {dev}")
    # BSG_BENCH_CPU_WEIGHTS: build the (bit-identical) synthetic weights on the host -- under rocprofv3 --pmc the
    # thousands of tiny generator kernels would otherwise dominate the profiling run
    wdev = torch.device("cpu") if os.environ.get("BSG_BENCH_CPU_WEIGHTS") else dev
    model = SegGptNative(synth_state_dict(g, seed=0, device=wdev), g, device=dev, dtype=dtype)
    log("model ready")
    B, P = args.batch, args.prompts
    Hh, W = g.image_size[0] // 2, g.image_size[1]
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    rn = lambda *s: torch.randn(*s, device=dev, generator=gen)
    pix, label_color, prompt_mask_color = rn(B, 3, Hh, W), rn(B, 3, Hh, W), rn(B, 3, Hh, W)
    yes = torch.ones(B, 1, Hh, W, dtype=torch.bool, device=dev)
    engine = PromptTrainEngine(model, torch.rand(P, 3, Hh, W, device=dev, generator=gen), lr=1e-3,
                               loss_variant=args.loss_variant)
    idx = (torch.arange(B, device=dev) + rank * B) % P

    def step():
        return engine.step(pix, label_color, yes, idx, prompt_mask_color)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    log("warm-up done")
    model.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof = model.profile_read()
    model.profile(False)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    log(f"timed region done: {dt / args.steps * 1e3:.1f} ms/step")
    if not torch.isfinite(loss) and not os.environ.get("BSG_DIAG_ALLOW_NONFINITE"):  # timing-only ablation builds only
        raise SystemExit("non-finite loss")

    if rank == 0:
        tiles = B * world * args.steps
        ms, fl, n = prof["gemm"]
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
        out = {
            "metric": "train tiles/sec", "value": round(tiles / dt, 3), "unit": "tiles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"SegGPT {args.geometry} 896x448 canvas, batch {B} tiles/GPU of 3x448x448 "
                                   f"(synthetic 4-band tile -> 3-ch), fwd + SegGptLoss({args.loss_variant}) + dgrad to "
                                   f"{P} prompt images + AdamW" + (", RCCL grad all-reduce" if world > 1 else ""),
                       "global_batch": B * world, "prompts": P, "parallelism": f"dp{world}",
                       "train_flops_per_tile": TRAIN_FLOPS_PER_TILE},
            "whole_step_tflops_per_gpu": round(tiles / world / dt * TRAIN_FLOPS_PER_TILE / 1e12, 1),
            "whole_step_frac_of_mfma_peak": round(tiles / world / dt * TRAIN_FLOPS_PER_TILE / 1e12 / peak, 4),
            "roofline": {"kernel": "gemm_nt_kernel (all epilogues)", "bound": "mfma", "achieved": round(achieved, 1),
                         "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": gemm_hbm_traffic_per_launch() if args.batch == 64 and args.dtype == "bf16" else None,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, offline pass)",
                         "avg_launch_ms": round(ms / max(n, 1), 4), "launches": n,
                         "algorithmic_gflop_per_launch": round(fl / max(n, 1) / 1e9, 2)},
            "kernel_time_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in prof.items()},
            "kernel_tflops": {k: round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 else 0.0 for k, v in prof.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle (one tile) ...")
            out["cpu_baseline"] = cpu_baseline(g, host_threads())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
