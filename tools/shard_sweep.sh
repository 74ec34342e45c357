#!/bin/bash
# What one rank of N does on config 2 (bench.py --seed-files N --shard 0 on one GPU) under different context / sub-batch settings.
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
i=0
for spec in "$@"; do
  i=$((i+1))
  n=${spec%%:*}; envs=${spec#*:}
  env $envs timeout 300 python bench.py --seed-files $n --shard 0 --steps 6 --warmup 2 --no-cpu-baseline > "$out/run_$i.json" 2> "$out/run_$i.err"
  python - "$out/run_$i.json" "$spec" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-60s %7.1f ms/step  cns %6.1f ovl %5.1f sort %4.1f piles %d" % (sys.argv[2], d["ms_per_step"], d["consensus_ms_per_step"], d["overlap"]["ms_per_step"], d["overlap"]["sort"]["ms_per_step"] + d["overlap"]["pile_assembly_ms_per_step"], d["config"]["piles_rank0"]))
except Exception as e:
    print("%-60s FAILED %s" % (sys.argv[2], e))
P
done | tee "$out/sweep.txt"
