#!/bin/bash
# product / candidate / product, the default bench without the CPU legs: tools/ab_candidate.sh <tag> <candidate> [steps]
out=gpurun_out/$1; cand=$2; steps=${3:-10}; mkdir -p "$out"
timeout 100 python bench.py --steps $steps --warmup 3 --no-cpu-baseline > "$out/p1.json" 2> "$out/p1.err"
timeout 100 python tools/kernel_candidate.py bench $cand --steps $steps --warmup 3 --no-cpu-baseline > "$out/c.json" 2> "$out/c.err"
timeout 100 python bench.py --steps $steps --warmup 3 --no-cpu-baseline > "$out/p2.json" 2> "$out/p2.err"
python - "$out/p1.json" "$out/c.json" "$out/p2.json" <<'P'
import json, sys
for p in sys.argv[1:]:
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(" %s: ms_per_step %.1f cns %.1f ovl %.1f | %s" % (p.split("/")[-1], d["ms_per_step"], d["consensus_ms_per_step"], d["overlap"]["ms_per_step"], {k[:-3]: round(v / d["steps"]) for k, v in d["kernel_ms"].items()}))
    except Exception as e:
        print(" %s: %r" % (p, e))
P
