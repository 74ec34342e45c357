#!/bin/bash
# Everything that can be checked without a GPU, in the order a change should pass it:
#   1. the product builds for gfx950 (hipcc cross-compiles), the oracle and -- where /root/reference exists -- oracle/_ref build;
#   2. pytest -m "not gpu": oracles vs golden vectors / the compiled reference, host logic, ABI, the kernels' logic under the
#      lane-accurate interpreter (tests/simt);
#   3. optional (slower): the interpreter tests once more with the lanes of a wavefront run highest first and a random wavefront
#      schedule, and sanitizer sweeps (UBSan, ASan) of both interpreted libraries.
# usage: tools/check.sh [quick|full [fuzz]]
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build(); print('build ok')"
python -m pytest tests -x -q -m "not gpu"
[ "${1:-quick}" = full ] || exit 0
SIMT_LANES_DESCENDING=1 SIMT_SCHEDULE=11 python -m pytest tests/test_simt_kernels.py tests/test_simt_overlap.py tests/test_simt_ksw2.py -x -q
for san in undefined address; do
  pre=/usr/lib/gcc/x86_64-linux-gnu/11/lib$([ $san = undefined ] && echo ubsan || echo asan).so
  SIMT_SANITIZE=$san UBSAN_OPTIONS=halt_on_error=1 ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 LD_PRELOAD=$pre \
    python -m pytest tests/test_simt_kernels.py tests/test_simt_overlap.py tests/test_simt_ksw2.py -x -q \
      -k "not forced and not out_of and not step2_mode0 and not drawn_at_random and not every_failed"   # (child processes and exception paths do not mix with a preloaded sanitizer)
done
# 4. optional (`tools/check.sh fuzz`, after `full`): short campaigns of the fuzzers against the compiled reference programs (oracle/_ref) --
#    the oracles, then the device paths under the interpreter
[ "${2:-}" = fuzz ] || exit 0
python tools/fuzz_overlap.py plain 1 10 && python tools/fuzz_overlap.py step2 1 6 && python tools/fuzz_overlap.py sort 1 6 && python tools/fuzz_cigar.py oracle 1 6
NDGPU_SIMT=1 python tools/fuzz_overlap.py cli 1 6 && NDGPU_SIMT=1 python tools/fuzz_overlap.py dump 1 10 && NDGPU_SIMT=1 python tools/fuzz_cigar.py device 1 3
NDGPU_SIMT=1 python tools/fuzz_consensus.py 1 8 && NDGPU_SIMT=1 python tools/fuzz_stage.py 1 2
python tools/fuzz_inflate.py 1 60 && python tools/fuzz_overlap_options.py 1 8
