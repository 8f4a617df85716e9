#!/bin/bash
# tools/build_at.sh <git-rev> <out.so>: compile the library as of <git-rev> (for same-box A/B runs: bench.py --lib <out.so>, BSG_AB_LIB=<out.so> for the probes in tools/)
set -e
rev=$1; out=$2; d=$(mktemp -d)
git archive "$rev" beach_seg_amd/csrc include | tar -x -C "$d"
flags="-O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -I$d/include"
/opt/rocm/bin/hipcc $flags -c -o "$d/api.o" "$d/beach_seg_amd/csrc/seggpt_api.hip" &
objs="$d/api.o"
if [ -f "$d/beach_seg_amd/csrc/attention_kv4.hip" ]; then  # round 4 on: second translation unit (own -mllvm flag, __graft_entry__.build)
  /opt/rocm/bin/hipcc $flags -mllvm -amdgpu-mfma-vgpr-form -c -o "$d/kv4.o" "$d/beach_seg_amd/csrc/attention_kv4.hip" &
  objs="$objs $d/kv4.o"
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs
rm -rf "$d"
