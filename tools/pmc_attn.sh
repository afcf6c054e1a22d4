#!/bin/bash
# PMC passes for one op of tools/bench_ops.py (separate passes: SQ has 8 slots, FETCH_SIZE/WRITE_SIZE do not fit together)
# usage: tools/pmc_attn.sh <op> <outdir-under-gpurun_out>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
OP=${1:-attn}; OUT=$ROOT/gpurun_out/${2:-pmc}
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $ROOT/tools/bench_ops.py $OP 3 > $OUT.$1.log 2>&1 || echo "pass $1 failed"; }
mkdir -p $OUT
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" &&
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM" &&
run tcc "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" &&
run fetch "FETCH_SIZE" &&
run write "WRITE_SIZE"
find $OUT -name "*counter_collection.csv" | head
