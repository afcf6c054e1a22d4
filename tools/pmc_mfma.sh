#!/bin/bash
# MFMA utilisation of the matrix kernels: busy cycles of the matrix pipe (SQ_VALU_MFMA_BUSY_CYCLES, summed over SIMDs)
# against GRBM_GUI_ACTIVE (summed over the 8 XCDs).  usage: tools/pmc_mfma.sh <outdir-under-gpurun_out>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
OUT=$ROOT/gpurun_out/${1:-pmc_mfma}
cd /tmp && export TMPDIR=/tmp ATTN_ONLY3=1
mkdir -p $OUT
for op in attn colsum gemm; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$op -- python3 $ROOT/tools/bench_ops.py $op 2 > $OUT/$op.log 2>&1 || echo "$op failed"
done
find $OUT -name "*counter_collection.csv" | wc -l
