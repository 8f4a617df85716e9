#!/bin/bash
# tools/step_ab.sh <arm>...: rocprofv3 kernel-trace A/B of the B=64 train step inside ONE gpurun call (boxes differ by 2-3 %).
# arm = lib:BSG_GEMM[:GROUP_M[:STAGGER]]; lib = "default" (the in-tree library) or the stem of tools/diag/<lib>.so (e.g. built by
# tools/build_at.sh <rev> tools/diag/<lib>.so).  Report: python tools/ab_report.py gpurun_out/ab_<lib>_<ver>... (GEMM ms per step and epilogue)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for arm in "$@"; do
  IFS=: read lib ver gm sg <<< "$arm"
  out=gpurun_out/ab_${lib}_$ver${gm:+_g$gm}${sg:+_s$sg}; mkdir -p $out
  if [ -n "$sg" ]; then export BSG_GEMM_STAGGER=$sg; else unset BSG_GEMM_STAGGER; fi
  export BSG_GEMM=$ver
  if [ -n "$gm" ]; then export BSG_GEMM_GROUP_M=$gm; else unset BSG_GEMM_GROUP_M; fi
  if [ "$lib" != "default" ]; then export BSG_LIB=tools/diag/$lib.so; else unset BSG_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
  cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
  rm -rf $out/trace
  echo "$arm: $(tail -1 $out/trace.log | grep -o '"ms_per_step": [0-9.]*')"
done
