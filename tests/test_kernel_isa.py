"""ISA-level guard for `gemm_nt_kernel_v5` (CPU; needs the built library and llvm-objdump from the ROCm image).

v5 issues its MFMAs through inline asm so that the 256 accumulators stay in the accumulator half of the register file.  The price:
hipcc does not know those statements are MFMAs, so it inserts no wait states between one and a read of its result.  The kernel is
written so that nothing touches an accumulator between an output tile's first and last MFMA; this test checks the COMPILED code
for that property (an accumulator copy or spill in that range would read MFMA results early -- seen once, with a C = 0 first-tile
variant, as garbage in one GEMM shape) and for the two things the K loop's counted waits rely on: no scratch traffic and no
compiler-inserted `vmcnt(0)` inside the steady-state loop."""
import re
import shutil
import struct
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "beach_seg_amd" / "libbsg_hip.so"
OBJDUMP = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _device_code_object(tmp_path):
    blob = LIB.read_bytes()
    at = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if at < 0:
        pytest.skip("fat binary is not an uncompressed clang offload bundle")
    n = struct.unpack_from("<Q", blob, at + 24)[0]
    off = at + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", blob, off)
        off += 24
        triple = blob[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            out = tmp_path / "dev.co"
            out.write_bytes(blob[at + o:at + o + sz])
            return out
    pytest.skip("no gfx950 code object in the library")


def _kernels(text):
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


@pytest.mark.skipif(not LIB.exists() or not Path(OBJDUMP).exists(), reason="library not built or llvm-objdump missing")
def test_gemm_v5_accumulators_untouched_between_mfmas(tmp_path):
    co = _device_code_object(tmp_path)
    text = subprocess.run([OBJDUMP, "-d", str(co)], capture_output=True, text=True, check=True).stdout
    seen = 0
    for head, ins in _kernels(text).items():
        if "gemm_nt_kernel_v5" not in head:
            continue
        seen += 1
        txt = [t for _, t, _ in ins]
        assert sum(t.startswith("v_mfma") for t in txt) >= 512, head  # steady pair + the two DMA-less tiles
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
                assert not (m and regs & set(range(int(m.group(1)), int(m.group(2)) + 1))), \
                    f"{head}: `{t}` {i - j} instructions behind `{txt[j]}`"
        # (2) the steady-state K loop = the smallest backward-branch region holding exactly 256 MFMAs (two K tiles)
        loops = []
        for i, (addr, t, tgt) in enumerate(ins):
            if t.startswith(("s_cbranch", "s_branch")) and tgt is not None and tgt < addr:
                j0 = next(j for j, (a2, _, _) in enumerate(ins) if a2 >= tgt)
                if sum(x.startswith("v_mfma") for x in txt[j0:i + 1]) == 256:
                    loops.append((i - j0, j0, i))
        assert loops, f"{head}: K loop not found"
        _, j0, i1 = min(loops)
        loop = txt[j0:i1 + 1]
        assert not [t for t in loop if t.startswith("v_accvgpr_")], f"{head}: accumulator access inside the K loop"
        assert not [t for t in loop if t.startswith("scratch_")], f"{head}: scratch traffic inside the K loop"
        assert not [t for t in loop if t.startswith("s_waitcnt") and "vmcnt(0)" in t], f"{head}: vmcnt(0) inside the K loop"
        assert sum(t.startswith("buffer_load_dwordx4") and "lds" in t for t in loop) == 32, f"{head}: LDS-DMA pieces per two K tiles"
        # (3) the DMA asm statements set m0 without declaring it (the clobber costs ~50 instructions per kernel): nothing else in
        #     the kernel may use m0
        other = [t for t in txt if re.search(r"\bm0\b", t) and not t.startswith("s_mov_b32 m0,")]
        assert not other, f"{head}: m0 used outside the LDS-DMA set-up: {other[:3]}"
    assert seen >= 10, f"only {seen} v5 kernels found"  # 2 dtypes x the epilogues of gemm_v5_pick
