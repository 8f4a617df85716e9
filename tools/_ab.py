"""A/B helper for the probes in tools/: `BSG_AB_LIB=<path to another build of libbsg_hip.so>` binds that build (e.g. one made by
tools/build_at.sh <rev> <out.so>) instead of the in-tree library.  Read HERE, in tooling -- the package itself reads no
environment variable for this."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pick_lib() -> None:
    p = os.environ.get("BSG_AB_LIB")
    if p:
        from beach_seg_amd import _native
        _native.use_library(p)
        print(f"[ab] library: {p}", file=sys.stderr, flush=True)
