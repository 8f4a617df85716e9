"""ISA-level guard of the COMPILED library (CPU; needs llvm-objdump / llvm-readelf from the ROCm image).

Some kernels rest on properties hipcc does not know about, so they are checked in the machine code, at BUILD time
(`__graft_entry__.build()` raises when a check fails: a rebuilt library cannot ship unchecked) and again by
`tests/test_kernel_isa.py` (which FAILS, not skips, without the tools):

* `gemm_nt_kernel_v5` issues its MFMAs through inline asm so that the 256 accumulators stay in the accumulator half of the
  register file.  hipcc then inserts no wait states between an MFMA and a read of its result: nothing may touch an accumulator
  between an output tile's first and last MFMA (an accumulator copy or spill in that range reads MFMA results early -- seen
  once, with a C = 0 first-tile variant, as garbage in one GEMM shape); the K loop's counted waits need a loop free of scratch
  traffic and of compiler-inserted `vmcnt(0)`; its LDS-DMA statements write `m0` without declaring it (the clobber costs ~50
  instructions per kernel), so nothing else in the kernel may use `m0`.
* the x3 attention kernels hold ~120 split registers per unrolled iteration unless a scheduling fence follows every MFMA
  triple (without it hipcc hoists 16 iterations of reads + splits and spills 480 B per lane in dQ): no scratch at all.
* the 16-bit attention kernels run at two / three waves per SIMD by register budget: no spills, VGPR counts inside the budget.
* `attn_bwd_dkv4_kernel` (one wave per SIMD) pins 64 R accumulators in the accumulator file through asm MFMAs: see check_kv4.
"""
from __future__ import annotations

import re
import struct
import subprocess
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
OBJDUMP, READELF = LLVM / "llvm-objdump", LLVM / "llvm-readelf"


class IsaGuardError(RuntimeError):
    pass


def device_code_objects(lib: Path, outdir: Path) -> list:
    """Every gfx950 code object of the fat binary (one offload bundle per translation unit), extracted into `outdir`."""
    blob = Path(lib).read_bytes()
    outs, at = [], blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if at < 0:
        raise IsaGuardError(f"{lib}: not an uncompressed clang offload bundle (build without --offload-compress)")
    while at >= 0:
        n = struct.unpack_from("<Q", blob, at + 24)[0]
        off = at + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode()
            off += tl
            if "gfx950" in triple:
                out = Path(outdir) / f"dev{len(outs)}.co"
                out.write_bytes(blob[at + o:at + o + sz])
                outs.append(out)
        at = blob.find(b"__CLANG_OFFLOAD_BUNDLE__", at + 24)
    if not outs:
        raise IsaGuardError(f"{lib}: no gfx950 code object")
    return outs


def kernels(text: str) -> dict:
    """{kernel head line: [(address, instruction text, branch target address or None)]} from `llvm-objdump -d` output."""
    out = {}
    for k in re.split(r"\n(?=[0-9a-f]{16} <)", text):
        head, _, rest = k.partition("\n")
        m0 = re.match(r"([0-9a-f]{16}) <", head)
        if not m0:
            continue
        base = int(m0.group(1), 16)
        ins = []
        for ln in rest.split("\n"):
            m = re.match(r"\s+(.*?)\s*// ([0-9A-F]+): [0-9A-F ]+(?:<[^>]*\+0x([0-9a-f]+)>)?", ln)
            if m:
                ins.append((int(m.group(2), 16), m.group(1).strip(), base + int(m.group(3), 16) if m.group(3) else None))
        out[head] = ins
    return out


def kernel_metadata(co: Path) -> dict:
    """{mangled kernel name: {vgpr, agpr, sgpr, scratch, spill}} from the code object's notes."""
    notes = subprocess.run([str(READELF), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
    out = {}
    for blk in notes.split("  - .agpr_count:")[1:]:
        f = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "0"])[1]
        out[f("name")] = {"agpr": int(blk.split("\n")[0].strip()), "vgpr": int(f("vgpr_count")), "sgpr": int(f("sgpr_count")),
                          "scratch": int(f("private_segment_fixed_size")), "spill": int(f("vgpr_spill_count"))}
    return out


def _need(cond, msg):
    if not cond:
        raise IsaGuardError(msg)


def check_gemm_v5(kern: dict) -> int:
    seen = 0
    for head, ins in kern.items():
        if "gemm_nt_kernel_v5" not in head:
            continue
        seen += 1
        txt = [t for _, t, _ in ins]
        _need(sum(t.startswith("v_mfma") for t in txt) >= 512, f"{head}: MFMA count")  # steady pair + the two DMA-less tiles
        # (1) straight-line hazard: no access to an accumulator within 12 instructions behind the MFMA that writes it (a
        #     16x16x32 MFMA needs 8 passes before its result may be read) unless the settling `s_nop 15` pair stands between
        for i, t in enumerate(txt):
            if not t.startswith("v_accvgpr_"):
                continue
            regs = {int(r) for r in re.findall(r"\ba(\d+)\b", t)}
            for j in range(i - 1, max(i - 13, -1), -1):
                if txt[j].startswith("s_nop 15"):
                    break
                m = re.match(r"v_mfma\S* a\[(\d+):(\d+)\]", txt[j])
                _need(not (m and regs & set(range(int(m.group(1)), int(m.group(2)) + 1))),
                      f"{head}: `{t}` {i - j} instructions behind `{txt[j]}`")
        # (2) the steady-state K loop = the smallest backward-branch region holding exactly 256 MFMAs (two K tiles)
        loops = []
        for i, (addr, t, tgt) in enumerate(ins):
            if t.startswith(("s_cbranch", "s_branch")) and tgt is not None and tgt < addr:
                j0 = next(j for j, (a2, _, _) in enumerate(ins) if a2 >= tgt)
                if sum(x.startswith("v_mfma") for x in txt[j0:i + 1]) == 256:
                    loops.append((i - j0, j0, i))
        _need(loops, f"{head}: K loop not found")
        _, j0, i1 = min(loops)
        loop = txt[j0:i1 + 1]
        _need(not [t for t in loop if t.startswith("v_accvgpr_")], f"{head}: accumulator access inside the K loop")
        _need(not [t for t in loop if t.startswith("scratch_")], f"{head}: scratch traffic inside the K loop")
        _need(not [t for t in loop if t.startswith("s_waitcnt") and "vmcnt(0)" in t], f"{head}: vmcnt(0) inside the K loop")
        _need(sum(t.startswith("buffer_load_dwordx4") and "lds" in t for t in loop) == 32, f"{head}: LDS-DMA pieces per two K tiles")
        # (3) m0 only through the LDS-DMA set-up
        other = [t for t in txt if re.search(r"\bm0\b", t) and not t.startswith("s_mov_b32 m0,")]
        _need(not other, f"{head}: m0 used outside the LDS-DMA set-up: {other[:3]}")
    _need(seen >= 10, f"only {seen} v5 kernels found")  # 2 dtypes x the epilogues of gemm_v5_pick
    return seen


# attention kernels: VGPR budget by waves per SIMD (guide, register files: <= 168 -> 3 waves, <= 256 -> 2, <= 512 -> 1)
ATTN_BUDGET = (
    (r"attn_fwd_kernelIDF16", 168), (r"attn_bwd_dq_kernelIDF16", 256), (r"attn_bwd_dkv_kernelIDF16", 256),
    (r"attn_fwd_kernelIf", 256), (r"attn_bwd_dq_kernelIf", 512), (r"attn_bwd_dkv_kernelIf", 256),
)


def check_attention(kern: dict, meta: dict) -> int:
    seen = 0
    for name, md in meta.items():
        if "attn_" not in name:
            continue
        seen += 1
        _need(md["spill"] == 0 and md["scratch"] == 0, f"{name}: spills ({md['spill']} VGPRs, {md['scratch']} B of scratch)")
        for rx, budget in ATTN_BUDGET:
            if re.search(rx, name):
                _need(md["vgpr"] <= budget, f"{name}: {md['vgpr']} VGPRs, budget {budget}")
    for head, ins in kern.items():
        if "attn_" in head:
            _need(not [t for _, t, _ in ins if t.startswith("scratch_")], f"{head}: scratch traffic")
    _need(seen >= 9, f"only {seen} attention kernels found")
    return seen


def check_kv4(kern: dict, meta: dict) -> int:
    """attention_kv4.hpp: dK^T / dV^T live in the accumulator file through asm MFMAs, S / dP through VGPR-form builtin MFMAs
    (its translation unit is compiled with -amdgpu-mfma-vgpr-form).  Machine-code properties: no spills; every MFMA of the
    query loop either writes arch VGPRs (builtin) or accumulates in place in the accumulator file (asm); no accumulator-file
    register an MFMA writes is read or written by anything else between the first and the last of those MFMAs (a copy there
    would read MFMA results early: hipcc pads nothing behind an asm statement) -- the v_accvgpr moves hipcc uses to park other
    values in the spare accumulator-file registers are fine."""
    seen = 0
    for name, md in meta.items():
        if "attn_bwd_dkv4" in name:
            _need(md["spill"] == 0 and md["scratch"] == 0, f"{name}: spills")
    for head, ins in kern.items():
        if "attn_bwd_dkv4" not in head:
            continue
        seen += 1
        txt = [t for _, t, _ in ins]
        _need(not [t for t in txt if t.startswith("scratch_")], f"{head}: scratch traffic")
        # everything between the first and the last accumulator-file MFMA in program order (the prologue's zero-fill stands
        # before it, the settling nops and the stores behind it)
        idx = [i for i, t in enumerate(txt) if re.match(r"v_mfma\S* a\[", t)]
        _need(len(idx) >= 16, f"{head}: only {len(idx)} accumulator-file MFMAs")
        region = txt[idx[0]:idx[-1] + 1]
        acc_regs = set()
        for t in region:
            m2 = re.match(r"v_mfma\S* a\[(\d+):(\d+)\]", t)
            if m2:
                acc_regs |= set(range(int(m2.group(1)), int(m2.group(2)) + 1))
                m3 = re.search(r"a\[(\d+):(\d+)\]\s*$", t.split("//")[0].strip())
                _need(m3 and (m3.group(1), m3.group(2)) == (m2.group(1), m2.group(2)), f"{head}: accumulator-file MFMA that is not in place: `{t}`")
        _need(len(acc_regs) >= 128, f"{head}: only {len(acc_regs)} pinned accumulator registers")  # 64 per key row, two or three rows
        for t in region:
            if "accvgpr" in t:
                regs = {int(r) for r in re.findall(r"\ba(\d+)\b", t)}
                _need(not (regs & acc_regs), f"{head}: `{t}` touches a pinned accumulator between the MFMAs")
    _need(seen >= 4, f"only {seen} kv4 kernels found")  # 2 dtypes x (three, two rows per wave)
    return seen


def check_library(lib: Path) -> dict:
    """Run every check on `lib`; raises IsaGuardError on the first violation.  Returns counts for the log."""
    for tool in (OBJDUMP, READELF):
        if not tool.exists():
            raise IsaGuardError(f"{tool} is missing: the ISA guard is a hard requirement of the build")
    kern, meta = {}, {}
    with tempfile.TemporaryDirectory() as d:
        for co in device_code_objects(Path(lib), Path(d)):
            text = subprocess.run([str(OBJDUMP), "-d", str(co)], capture_output=True, text=True, check=True).stdout
            kern.update(kernels(text))
            meta.update(kernel_metadata(co))
    return {"gemm_v5": check_gemm_v5(kern), "attention": check_attention(kern, meta), "attention_kv4": check_kv4(kern, meta)}
