#!/bin/bash
# tools/profile_run.sh <tag>: rocprofv3 evidence of `bench.py` (B=64 bf16 train step) on the GPU box:
#   1. --kernel-trace --stats (per-kernel durations)            -> gpurun_out/<tag>/kernel_stats.csv
#   2. five separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_* busy+wait / GRBM_GUI_ACTIVE / SQ_* instruction issue),
#      summarised by tools/pmc_summary.py
#                                                                 -> gpurun_out/<tag>/pmc_summary.json
# PMC passes build the weights on the host (BSG_BENCH_CPU_WEIGHTS=1) so that the generator's thousands of tiny kernels do
# not dominate the counter run, and only include this library's kernels.
set -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
echo "[profile] kernel trace done"
export BSG_BENCH_CPU_WEIGHTS=1
INC='gemm_nt|attn_|conv3x3|ln_fwd|ln_bwd|head_bwd|patchify|merge_halves|loss_|adamw|prompt_|cast_rows|split_rows|absmax|grad_scale'
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "grbm:GRBM_GUI_ACTIVE" "issue:SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --pmc $ctr --kernel-include-regex "$INC" --output-format csv -d $out/pmc_$name -- $B > $out/pmc_$name.log 2>&1 || { tail -5 $out/pmc_$name.log; exit 1; }
  echo "[profile] pmc pass $name done"
done
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_grbm $out/pmc_summary.json $out/pmc_issue
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_grbm $out/pmc_issue $out/trace
