"""Per-kernel resources of the built library (CPU; needs llvm-readelf from the ROCm image):
    python tools/kernel_resources.py [regex] [--lib path] [--dump out.s]
prints VGPR / AGPR / SGPR / LDS / scratch of every gfx950 kernel whose demangled name matches; --dump writes the disassembly
of the matching kernels."""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")


sys.path.insert(0, str(ROOT))
from beach_seg_amd.isa_guard import device_code_objects  # noqa: E402


def main():
    args = sys.argv[1:]
    lib = ROOT / "beach_seg_amd" / "libbsg_hip.so"
    dump = None
    if "--lib" in args:
        i = args.index("--lib"); lib = Path(args[i + 1]); del args[i:i + 2]
    if "--dump" in args:
        i = args.index("--dump"); dump = Path(args[i + 1]); del args[i:i + 2]
    rx = re.compile(args[0] if args else ".")
    with tempfile.TemporaryDirectory() as d:
        keep = []
        for co in device_code_objects(lib, Path(d)):
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                f = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
                agpr = blk.split("\n")[0].strip()
                name = f("name")
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                if not (rx.search(dem) or rx.search(name)):
                    continue
                short = re.sub(r"\(.*", "", dem)
                print(f"vgpr {f('vgpr_count'):>4} (agpr {agpr:>3}) sgpr {f('sgpr_count'):>3} lds {f('group_segment_fixed_size'):>6} "
                      f"scratch {f('private_segment_fixed_size'):>4} spill {f('vgpr_spill_count'):>3}  {short}")
            if dump:
                text = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--demangle", str(co)], capture_output=True, text=True).stdout
                keep += [k for k in re.split(r"\n(?=[0-9a-f]{16} <)", text) if rx.search(k.split("\n", 1)[0])]
        if dump:
            dump.write_text("\n".join(keep))
            print(f"wrote {len(keep)} kernels to {dump}")


if __name__ == "__main__":
    main()
