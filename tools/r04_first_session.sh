#!/bin/bash
# First GPU session of the next round: what round 3 finished without GPU minutes, then the open config-3 question.
#   1. the GPU tests of the late paths (-c, ksw_ll, out-of-core sort): tests/test_zzz_gpu_cigar.py
#   2. bench.py default (config 2) -- its line now carries `allocations` and `overlap.pool_calls`
#   3. tools/measure_modes.py (adds `--step 1 -c` next to the reference binary)
#   4. config 3 in full, one warm-up + one step: are buffers (re)allocated inside the timed step?  
# usage (on the GPU box, from the repo root):  bash tools/r04_first_session.sh <tag> [stages...]     stages: late bench modes c3
set -u
tag=${1:-r04a}; shift || true
stages=${*:-late bench}
out=gpurun_out/$tag
mkdir -p "$out"
for s in $stages; do
  case $s in
    late)  timeout 1500 python -m pytest tests/test_zzz_gpu_cigar.py -x -q > "$out/pytest_late.log" 2>&1; echo "late exit $?"; tail -5 "$out/pytest_late.log" ;;
    bench) timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_config2.json" 2> "$out/bench_config2.err"; echo "bench exit $?"
           python - "$out/bench_config2.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(" ms_per_step %.1f value %.1f M  allocations %s  pool_calls %s" % (d["ms_per_step"], d["value"] / 1e6, d["allocations"], d["overlap"]["pool_calls"]))
P
           ;;
    modes) timeout 2400 python tools/measure_modes.py "$out/overlap_modes.json" > "$out/modes.log" 2>&1; echo "modes exit $?"; tail -8 "$out/modes.log" ;;
    c3)    timeout 1500 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > "$out/bench_config3.json" 2> "$out/bench_config3.err"; echo "c3 exit $?"
           python - "$out/bench_config3.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(" ms_per_step %.0f consensus %.0f overlap %.0f allocations %s pool_calls %s kernel_ms %s" % (d["ms_per_step"], d["consensus_ms_per_step"],
      d["overlap"]["ms_per_step"], d["allocations"], d["overlap"]["pool_calls"], d["kernel_ms"]))
P
           ;;
    *) echo "unknown stage $s" ;;
  esac
done
