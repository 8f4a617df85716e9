"""ORACLE -- TEST INFRASTRUCTURE ONLY.  BASELINE.md section 3 calibration, build container only (needs `transformers`):
times the reference network itself (HF `SegGptForImageSegmentation`, the module `/root/reference/src/util/ml_util.py:8`
loads) against this repository's CPU restatement (`oracle/seggpt_oracle.py`, the thing `bench.py`'s `cpu_baseline` leg
times on the GPU box, where the HF package and /root/reference do not exist) on the same ViT-L tile, same threads:
forward + reference loss + backward to the prompt pixels, B=1, one warm-up + N timed steps each, interleaved.

    python oracle/calibrate_cpu.py [threads=8] [timed=2]   ->  prints one JSON line (ratio = restatement / HF time)
"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from beach_seg_amd.weights import SegGptGeometry, synth_state_dict  # noqa: E402
from oracle import seggpt_oracle as O  # noqa: E402
from oracle.gen_golden import hf_model  # noqa: E402
from oracle.gen_inputs import synth_inputs  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    timed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    torch.set_num_threads(threads)
    g = SegGptGeometry.vit_large()
    sd = synth_state_dict(g, seed=0)
    m = hf_model(g, sd)
    pix, prm, pm_cls, lb_cls, pal = synth_inputs(g, 1, 7)
    pm = O.normalize(O.apply_mask_rgb(pal, pm_cls))
    lab = O.normalize(O.apply_mask_rgb(pal, lb_cls))
    yes = (lb_cls != 0)[:, None]

    def hf_step():
        p = prm.clone().requires_grad_(True)
        pred = m(pixel_values=pix, labels=lab, prompt_pixel_values=p, prompt_masks=pm, embedding_type="instance").pred_masks
        loss = O.seggpt_loss(pred, lab, yes, 0.01, "reference")
        (gr,) = torch.autograd.grad(loss, p)
        return float(loss), gr

    def our_step():
        p = prm.clone().requires_grad_(True)
        pred = O.forward(sd, g, pix, p, pm, labels=lab)
        loss = O.seggpt_loss(pred, lab, yes, 0.01, "reference")
        (gr,) = torch.autograd.grad(loss, p)
        return float(loss), gr

    l1, g1 = hf_step()
    l2, g2 = our_step()
    t_hf, t_our = [], []
    for _ in range(timed):
        t0 = time.perf_counter(); hf_step(); t_hf.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); our_step(); t_our.append(time.perf_counter() - t0)
    a, b = sum(t_hf) / timed, sum(t_our) / timed
    print(json.dumps({"threads": threads, "timed_steps": timed, "hf_s_per_tile": round(a, 2), "restatement_s_per_tile": round(b, 2),
                      "ratio_restatement_over_hf": round(b / a, 3), "loss_rel_diff": abs(l1 - l2) / abs(l1),
                      "grad_rel_diff": float((g1 - g2).abs().max() / g1.abs().max())}))


if __name__ == "__main__":
    main()
