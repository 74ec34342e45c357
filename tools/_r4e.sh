mkdir -p gpurun_out/r4e
run() { name=$1; shift
  ( while true; do rocm-smi --showmeminfo vram --csv 2>/dev/null | tail -n +2 | head -1; sleep 2; done ) > gpurun_out/r4e/$name.vram & SM=$!
  NDGPU_TRACE=1 timeout 600 python bench.py "$@" > gpurun_out/r4e/$name.json 2> gpurun_out/r4e/$name.err; rc=$?
  kill $SM 2>/dev/null
  echo "$name rc=$rc peak_vram_bytes=$(cut -d, -f3 gpurun_out/r4e/$name.vram | sort -n | tail -1) oom_msgs=$(grep -c 'out of device memory' gpurun_out/r4e/$name.err) released=$(grep -c 'released' gpurun_out/r4e/$name.err)"
  python -c "
import json,sys; d=json.loads(open('gpurun_out/r4e/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'Mb/s', round(d['ms_per_step']), 'ms/step cns', round(d['consensus_ms_per_step']), 'ovl', round(d['overlap']['ms_per_step']), 'piles', d['config']['piles_rank0'], 'parity', (d.get('parity') or {}).get('piles'), (d.get('parity') or {}).get('mismatch'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))" 2>&1 | tail -2
}
run c5 --config 5 --seed-files 8 --steps 2 --warmup 1 --cpu-sample 64
