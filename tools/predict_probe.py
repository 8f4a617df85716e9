import sys, time, torch
sys.path.insert(0, '/root/repo')
from beach_seg_amd import ops
from beach_seg_amd.config import BeachSegConfig
from beach_seg_amd.model import PromptModel
from beach_seg_amd.predict import Accumulator, grid_crops
dev = torch.device("cuda:0")
conf = BeachSegConfig(checkpoint="synthetic:vit_large", precision="bf16-true", crop_size=112, batch_size=64)
pm = PromptModel(conf, device=dev)
g = torch.Generator(device=dev).manual_seed(11)
S = 448
pm.create_trainable_params([{"crop_idx": i, "date": "d", "image": torch.rand(3, S, S, device=dev, generator=g),
                             "mask": torch.randint(0, 4, (S, S), device=dev, generator=g, dtype=torch.uint8),
                             "nodata": torch.zeros(S, S, dtype=torch.bool)} for i in range(32)])
size = 2048
mosaic = (torch.rand(size // 64, size // 64, 3, device=dev, generator=g).repeat_interleave(64, 0).repeat_interleave(64, 1) * 255).to(torch.uint8)
crops = grid_crops(size, size, 112)
graphed = pm.model.capture_forward(64)
acc = Accumulator((size, size), conf.classes, dev)
names = ["frontend", "palette+prompt", "forward", "decode", "vote"]
tot = [0.0] * 5
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for rep in range(3):
    for s in range(0, 64 * 4, 64):
        cb = crops[s:s + 64]
        e0 = ev(); img = ops.tile_frontend(mosaic, cb.to(dev), 112, S)
        e1 = ev(); idx = torch.arange(s, s + 64) % 32
        pal, pal_norm = pm.create_palette(64, train=True); pb, pmasks = pm.prepare_prompt(idx, pal, train=False)
        e2 = ev(); out = graphed(img, pb["image"], pmasks)
        e3 = ev(); pred = pm.process_pred_masks(out, pal_norm)
        e4 = ev(); acc.update("d0", cb, pred.to(torch.uint8), 112)
        e5 = ev(); torch.cuda.synchronize()
        if rep:
            for i, (a, b) in enumerate([(e0, e1), (e1, e2), (e2, e3), (e3, e4), (e4, e5)]): tot[i] += a.elapsed_time(b)
n = 2 * 4
print({k: round(v / n, 2) for k, v in zip(names, tot)}, "ms per batch of 64")
t0 = time.perf_counter()
for s in range(0, 64 * 4, 64):
    idx = torch.arange(s, s + 64) % 32
    pal, pal_norm = pm.create_palette(64, train=True); pb, pmasks = pm.prepare_prompt(idx, pal, train=False)
torch.cuda.synchronize(); print("host palette+prompt per batch ms", (time.perf_counter() - t0) / 4 * 1e3)
