#!/bin/bash
# tools/build_at.sh <git-rev> <out.so>: compile the library as of <git-rev> (for same-box A/B runs: bench.py --lib <out.so>, BSG_AB_LIB=<out.so> for the probes in tools/)
set -e
rev=$1; out=$2; d=$(mktemp -d)
git archive "$rev" beach_seg_amd/csrc include | tar -x -C "$d"
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -shared -fPIC -I"$d/include" -o "$out" "$d/beach_seg_amd/csrc/seggpt_api.hip"
rm -rf "$d"
