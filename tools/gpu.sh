#!/bin/bash
# build the product libraries, then run a command on the GPU box:  tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -m nextdenovo_amd.build > /tmp/ndgpu_build.log 2>&1 || { tail -20 /tmp/ndgpu_build.log; exit 1; }
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
