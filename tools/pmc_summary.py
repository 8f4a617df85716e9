"""Summarise rocprofv3 --pmc CSVs (run on the GPU box): per kernel family the per-launch HBM bytes
(FETCH_SIZE x2 on gfx950 for wide coalesced reads, WRITE_SIZE), MFMA / LDS busy fractions and effective clock.
usage: python tools/pmc_summary.py <dir_fetch> <dir_write> <dir_sq> <dir_grbm> <out.json> [<dir_issue>]
The optional sixth argument is a pass over the instruction-issue counters (SQ_ACTIVE_INST_VALU / LDS / VMEM / SCA per wave
cycle, VALU and LDS instructions per MFMA): what the waves of a kernel spend their issue slots on."""
import collections, csv, glob, hashlib, json, pathlib, re, sys


def csrc_sha():  # same hash as bench.py: ties this summary to the kernel sources it was measured on
    h = hashlib.sha256()
    for f in sorted((pathlib.Path(__file__).resolve().parents[1] / "beach_seg_amd" / "csrc").glob("*")):
        if f.suffix in (".hpp", ".hip"):
            h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def load(d):
    fs = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur, seen = collections.Counter(), collections.defaultdict(float), set()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"])
            if k.startswith("void at::") or "rocclr" in k:
                k = "torch/other"
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); cnt[k] += 1
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, cnt, dur


ia = load(sys.argv[6])[0] if len(sys.argv) > 6 else None
fa, fc, fd = load(sys.argv[1]); wa, wc, wd = load(sys.argv[2]); sa, sc, sd = load(sys.argv[3]); ga, gc, gd = load(sys.argv[4])
out = {}
for k in sd:
    n = max(sc[k], 1); busy = sa[k]["SQ_BUSY_CYCLES"] or 1; w = sa[k]["SQ_WAVE_CYCLES"] or 1
    out[k] = {
        "launches": sc[k], "avg_us": sd[k] / n / 1e3,
        "hbm_fetch_MB_per_launch": fa[k]["FETCH_SIZE"] * 1024 * 2 / max(fc[k], 1) / 1e6,
        "hbm_write_MB_per_launch": wa[k]["WRITE_SIZE"] * 1024 / max(wc[k], 1) / 1e6,
        "mfma_busy_frac": sa[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (busy * 32), "lds_busy_frac": sa[k]["SQ_LDS_IDX_ACTIVE"] / (busy * 8),
        "lds_conflict_per_active": sa[k]["SQ_LDS_BANK_CONFLICT"] / max(sa[k]["SQ_LDS_IDX_ACTIVE"], 1),
        "wait_any": sa[k]["SQ_WAIT_ANY"] / w, "wait_inst_any": sa[k]["SQ_WAIT_INST_ANY"] / w, "active_inst_any": sa[k]["SQ_ACTIVE_INST_ANY"] / w,
        "clock_GHz": ga[k]["GRBM_GUI_ACTIVE"] / 8 / max(gd[k], 1),
    }
    o = out[k]
    if ia is not None and k in ia:
        iw = ia[k]["SQ_WAVE_CYCLES"] or 1
        mf = ia[k]["SQ_INSTS_MFMA"]
        o.update({"active_inst_valu": ia[k]["SQ_ACTIVE_INST_VALU"] / iw, "active_inst_lds": ia[k]["SQ_ACTIVE_INST_LDS"] / iw,
                  "active_inst_vmem": ia[k]["SQ_ACTIVE_INST_VMEM"] / iw, "active_inst_sca": ia[k]["SQ_ACTIVE_INST_SCA"] / iw,
                  "valu_insts_per_mfma": (ia[k]["SQ_INSTS_VALU"] - mf) / mf if mf else None,
                  "lds_insts_per_mfma": ia[k]["SQ_INSTS_LDS"] / mf if mf else None})
    o["hbm_GBps"] = (o["hbm_fetch_MB_per_launch"] + o["hbm_write_MB_per_launch"]) * 1e6 / (o["avg_us"] * 1e3) if o["avg_us"] else 0
out["_meta"] = {"csrc_sha": csrc_sha(), "steps": 3,  # tools/profile_run.sh profiles `bench.py --steps 2 --warmup 1`
                "note": "FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE x1; one rocprofv3 --pmc pass per counter group"}
json.dump(out, open(sys.argv[5], "w"), indent=1)
print("wrote", sys.argv[5], len(out), "kernels")
