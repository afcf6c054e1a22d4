#!/bin/bash
# Diagnostic (run in the build container, then tools/diag_ablate.sh gemm on the GPU box): builds of the library with
# parts of the 256x256x64 GEMM main loop removed (results wrong, timing only) into lib/exp/ - where does the in-loop
# time go?  The instrumented kernel is a diagnostic COPY (tools/diag_src/gemm256_ablate.hip: the round-2 kernel with
# its MAVLM_GEMM_ABLATE_* blocks); the product kernel carries none.   usage: tools/diag_gemm_ablate.sh
set -e
cd "$(dirname "$0")/.."
SRC=memory-augmented-vlm_amd/csrc
OUT=memory-augmented-vlm_amd/lib/exp
mkdir -p $OUT /tmp/gab
VARIANTS=${VARIANTS:-"base READS DMA MFMA BAR"}
for v in $VARIANTS; do
  D=""; [ $v != base ] && D="-DMAVLM_GEMM_ABLATE_$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -I$SRC -c -o /tmp/gab/gemm256_$v.o tools/diag_src/gemm256_ablate.hip &
done
wait
for v in $VARIANTS; do
  objs=$(ls memory-augmented-vlm_amd/lib/obj/*.o | grep -v "/gemm256.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/gemm_$v.so $objs /tmp/gab/gemm256_$v.o
done
ls -la $OUT
