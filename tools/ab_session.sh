#!/bin/bash
# A/B of the alignment-kernel forms inside the pipeline: bench line + rocprofv3 kernel stats per variant.
#   gpurun --timeout 1500 -- 'bash tools/ab_session.sh r03_ab'
set -u
tag=${1:-ab}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_$name.json" 2> "$out/bench_$name.err"
  echo "$name exit $?"; python - "$out/bench_$name.json" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(" ms_per_step %.1f value %.3g kernel_ms %s parity %s" % (d["ms_per_step"], d["value"], d.get("kernel_ms"), d.get("parity",{}).get("mismatch")))
except Exception as e: print(" parse failed", e)
P
  (cd /tmp && env "$@" timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/stats_$name" -o s -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/$out/prof_$name.log" 2>&1)
  python tools/rocprof_summary.py "$(ls "$out"/stats_$name/*results.db | head -1)" > "$out/kernel_stats_$name.txt" 2>> "$out/prof_$name.log"; head -14 "$out/kernel_stats_$name.txt"
  rm -rf "$out/stats_$name"
}
run default NDGPU_X=0
run pair NDGPU_K7=pair
run wave NDGPU_K8A=wave
run both NDGPU_K7=pair NDGPU_K8A=wave
