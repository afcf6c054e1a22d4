#!/bin/bash
# Round profile on the GPU box.  usage: tools/profile_round.sh <tag>    (outputs under gpurun_out/prof_<tag>/)
#   stats1 : rocprofv3 --kernel-trace --stats of bench.py with ONE video in flight (the configuration of bench.py's
#            instrumented pass: the attention kernel's average duration here is what roofline.avg_launch_ms must agree with)
#   stats2 : the same with the default two videos in flight (per-kernel averages include overlap with the other stream)
#   mfma   : --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES on tools/bench_ops.py (attn / colsum / gemm)
#   traffic: --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes on tools/bench_ops.py attn
# Counter passes carry only --kernel-trace (no --stats / sys / runtime traces), as the pool requires.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
TAG=${1:-r02}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp ATTN_ONLY3=1 MAVLM_BENCH_M8=0
B="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --repeats 3 --min-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- $B --videos-in-flight 1 > $OUT/stats1.json 2> $OUT/stats1.err || echo "stats1 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- $B > $OUT/stats2.json 2> $OUT/stats2.err || echo "stats2 failed"
for op in attn colsum gemm; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/mfma_$op -- python3 $ROOT/tools/bench_ops.py $op 2 > $OUT/mfma_$op.log 2>&1 || echo "mfma $op failed"
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/bench_ops.py attn 3 > $OUT/fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/tools/bench_ops.py attn 3 > $OUT/write.log 2>&1 || echo "write failed"
# keep what the summary needs (the merge back is capped at 64 MiB): stats + counter CSVs, no raw traces of the bench runs
find $OUT -name "*kernel_trace.csv" -path "*stats*" -delete
find $OUT -type f | head -50
du -sh $OUT
cd $ROOT && python3 tools/summarize_profiles.py $TAG
