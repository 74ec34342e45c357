mkdir -p gpurun_out/r2s
NDGPU_TRACE=1 NDGPU_PROF=1 NDGPU_CONTEXTS=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2s/c1.json 2> gpurun_out/r2s/c1.err
NDGPU_TRACE=1 NDGPU_PROF=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2s/c8.json 2> gpurun_out/r2s/c8.err
