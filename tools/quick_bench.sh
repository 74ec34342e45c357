#!/bin/bash
# quick A/B on the GPU box: bench.py (3 steps, no CPU baseline) under each "NAME:ENV=VAL,ENV=VAL" argument
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  [ "$envs" = "$spec" ] && envs=""
  env $(echo "$envs" | tr ',' ' ') timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_$name.json" 2> "$out/bench_$name.err"
  echo "$name exit $?"
  python - "$out/bench_$name.json" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(" ms_per_step %.1f consensus %.1f overlap %.1f kernel_ms %s" % (d["ms_per_step"], d["consensus_ms_per_step"], d["overlap"]["ms_per_step"], d.get("kernel_ms")))
except Exception as e: print(" parse failed", e)
P
done
