#!/bin/bash
# tools/step_ab.sh <arm>...: rocprofv3 kernel-trace A/B of the B=64 train step inside ONE gpurun call (boxes differ by 2-3 %).
# arm = name:lib[:BSG_GEMM]; lib = "default" (the in-tree library) or the path of another build (tools/build_at.sh <rev> <out.so>).
# Report: python tools/ab_report.py gpurun_out/ab_<name>... (GEMM ms per step and epilogue); per-kernel totals in kernel_stats.csv
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for arm in "$@"; do
  IFS=: read name lib ver <<< "$arm"
  out=gpurun_out/ab_$name; mkdir -p $out
  if [ -n "$ver" ]; then export BSG_GEMM=$ver; else unset BSG_GEMM; fi
  libarg=""; if [ "$lib" != "default" ]; then libarg="--lib $lib"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline $libarg > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
  cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
  rm -rf $out/trace
  echo "$arm: $(tail -1 $out/trace.log | grep -o '"ms_per_step": [0-9.]*')"
done
