#!/bin/bash
# A/B sweep of runtime knobs on the default bench (config 2): one line per setting.  usage: tools/sweep.sh <tag> "ENV=.. ENV=.." "..." ...
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout 300 python bench.py --steps ${SWEEP_STEPS:-5} --warmup 2 --no-cpu-baseline > "$out/run_$i.json" 2> "$out/run_$i.err"
  python - "$out/run_$i.json" "$envs" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms"]; s = d["steps"]
    o = d["overlap"].get("kernel_ms_last_index", {})
    print("%-70s %7.1f ms/step  cns %6.1f ovl %5.1f | fwd %5.0f tb %5.0f links %5.0f score %5.0f lq %5.0f (event ms per step) | ovl: hits %.1f chain %.1f seed %.1f" % (sys.argv[2], d["ms_per_step"], d["consensus_ms_per_step"], d["overlap"]["ms_per_step"], k["forward_ms"]/s, k["traceback_ms"]/s, k["links_ms"]/s, k["score_ms"]/s, k["lq_ms"]/s, o.get("hits_ms", 0), o.get("chain_ms", 0), o.get("seed_ms", 0)))
except Exception as e:
    print("%-70s FAILED %s" % (sys.argv[2], e))
P
done | tee "$out/sweep.txt"
