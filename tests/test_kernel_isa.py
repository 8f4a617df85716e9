"""ISA-level guard of the compiled library (CPU; `beach_seg_amd/isa_guard.py` holds the checks and says why each exists).
The same function runs inside `__graft_entry__.build()`; here it FAILS (no skip) when the library or the LLVM tools are
missing, so a tree whose kernels were rebuilt without the guard cannot pass the CPU suite."""
from pathlib import Path

from beach_seg_amd import isa_guard

ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "beach_seg_amd" / "libbsg_hip.so"


def test_compiled_kernels_keep_the_properties_hipcc_cannot_see():
    assert LIB.exists(), f"{LIB} is not built: run `python __graft_entry__.py`"
    seen = isa_guard.check_library(LIB)
    assert seen["gemm_v5"] >= 10 and seen["attention"] >= 9 and seen["attention_kv4"] >= 4, seen
