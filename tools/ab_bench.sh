#!/bin/bash
# usage: tools/ab_bench.sh "NAME=ENV1=a ENV2=b" ...   (each arg: label=env assignments; runs bench.py w/o cpu baseline)
# prints label, ms/step and the per-kernel profile of each variant; outputs under gpurun_out/ab/
mkdir -p gpurun_out/ab
for spec in "$@"; do
  label="${spec%%=*}"; envs="${spec#*=}"
  [ "$envs" = "$spec" ] && envs=""
  env $envs timeout -k 10 150 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ab/$label.json 2> gpurun_out/ab/$label.err || exit 1
  python - "$label" <<'PY'
import json,sys
l=sys.argv[1]
d=json.loads(open(f'gpurun_out/ab/{l}.json').read().strip().splitlines()[-1])
print(l, d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_time_ms_per_step'].items()}, flush=True)
PY
done
