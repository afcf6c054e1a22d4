#!/bin/bash
# HBM traffic of the two dominant kernels at the bench shapes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (they do not fit one pass on gfx950), plus a kernel-trace pass for durations.  usage: tools/pmc_traffic.sh <outdir>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
OUT=$ROOT/gpurun_out/${1:-pmc_traffic}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
for op in attn gemm; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${op}_fetch -- python3 $ROOT/tools/bench_ops.py $op 3 > $OUT/${op}_fetch.log 2>&1 || echo "fetch $op failed"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${op}_write -- python3 $ROOT/tools/bench_ops.py $op 3 > $OUT/${op}_write.log 2>&1 || echo "write $op failed"
done
find $OUT -name "*counter_collection.csv" | wc -l
