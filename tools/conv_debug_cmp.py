import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    x, y = a[k].float(), b[k].float()
    d = (x - y).abs()
    print(k, "max diff", float(d.max()), "ref max", float(y.abs().max()), "nonfinite", int((~torch.isfinite(x)).sum()))
    if d.max() > 1e-3 * y.abs().max() and x.dim() == 4 and x.shape[-1] == 64:
        bad = (d > 1e-3 * y.abs().max()).any(-1)  # (B,H,W)
        ys = bad.any(-1).nonzero()[:40].tolist(); xs = bad.any(1).nonzero()[:40].tolist()
        print("  bad rows (b,y):", ys[:24]); print("  bad cols (b,x):", xs[:24])
        cb = (d > 1e-3 * y.abs().max()).any(0).any(0).any(0).nonzero().flatten().tolist(); print("  bad channels:", cb)
