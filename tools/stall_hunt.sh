#!/bin/bash
# Looks for the multi-hundred-millisecond stalls of single steps: N quick benches per environment, every step's time listed.
#   tools/stall_hunt.sh <tag> <runs> "<ENV=..,ENV2=..>" ["<other env>" ...]
tag=$1; runs=$2; shift 2
out=gpurun_out/$tag; mkdir -p "$out"
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' ')
  for i in $(seq 1 "$runs"); do
    env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline > "$out/run.json" 2> "$out/run.err"
    python - "$out/run.json" "$spec" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["step_ms"]["list"]; med = sorted(s)[len(s) // 2]
    print("%-40s ms/step %6.1f median %6.1f max %7.1f  stalls(>1.5x median): %s  allocs %s" % (sys.argv[2], d["ms_per_step"], med, max(s), [x for x in s if x > 1.5 * med], d["allocations"]["in_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
  done
done | tee "$out/stalls.txt"
