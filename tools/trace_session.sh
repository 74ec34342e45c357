#!/bin/bash
# Host/device phase trace of a bench run + isolated (one context) kernel times.
set -u
tag=${1:-trace}
out=gpurun_out/$tag
mkdir -p "$out"
NDGPU_TRACE=1 NDGPU_PROF=1 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_trace.json" 2> "$out/bench_trace.err"
echo "trace exit $?"
NDGPU_CONTEXTS=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_ctx1.json" 2> "$out/bench_ctx1.err"
echo "ctx1 exit $?"
python - "$out/bench_ctx1.json" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(" ctx1 ms_per_step %.1f kernel_ms %s" % (d["ms_per_step"], d.get("kernel_ms")))
P
