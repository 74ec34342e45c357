#!/bin/bash
# Runs quick benches with NDGPU_TRACE=1 NDGPU_DEBUG_ALLOC=1 until one holds a stalled step (or <runs> are through); keeps that run's stderr.
tag=$1; runs=$2; shift 2
out=gpurun_out/$tag; mkdir -p "$out"
for i in $(seq 1 "$runs"); do
  env "$@" NDGPU_TRACE=1 NDGPU_DEBUG_ALLOC=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline > "$out/run.json" 2> "$out/run.err"
  python - "$out/run.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d["step_ms"]["list"]; med = sorted(s)[len(s) // 2]
st = [x for x in s if x > 1.5 * med]
print("steps", s, "allocs", d["allocations"])
sys.exit(3 if st else 0)
P
  if [ $? -eq 3 ]; then cp "$out/run.err" "$out/stalled_$i.err"; cp "$out/run.json" "$out/stalled_$i.json"; echo "stalled run $i kept"; fi
done
