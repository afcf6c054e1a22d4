#!/bin/bash
# Round profile on the GPU box, round 3 form, reused for round 4: EVERY pass runs bench.py itself (the judged command), so per-kernel averages,
# MFMA-busy shares and HBM bytes are those of the bench's real launch mix (row batches, frame-score variant, merges).
# usage: tools/profile_round3.sh <tag>    (outputs under gpurun_out/prof_<tag>/; then tools/summarize_profiles_r03.py <tag>)
#   stats1 : rocprofv3 --kernel-trace --stats, ONE stream (a row batch of 2 videos in flight) = the configuration of bench.py's
#            instrumented pass: the attention kernel's average duration here is what roofline.avg_launch_ms must agree with
#   stats2 : the same with the default two streams (per-kernel averages include overlap with the other stream)
#   mfma   : --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES, one stream
#   fetch / write : --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes, one stream
# Counter passes carry only --kernel-trace (no --stats / sys / runtime traces), as the pool requires.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp MAVLM_BENCH_M8=0
B="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --repeats 2 --min-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- $B --videos-in-flight 1 > $OUT/stats1.json 2> $OUT/stats1.err || echo "stats1 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- $B > $OUT/stats2.json 2> $OUT/stats2.err || echo "stats2 failed"
P="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --repeats 1 --min-seconds 0 --videos-in-flight 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/mfma -- $P > $OUT/mfma.json 2> $OUT/mfma.err || echo "mfma failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $P > $OUT/fetch.json 2> $OUT/fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $P > $OUT/write.json 2> $OUT/write.err || echo "write failed"
# keep what the summary needs (the merge back is capped at 64 MiB): stats + counter CSVs, no raw traces
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -size +20M -delete
du -sh $OUT
cd "$ROOT" && python3 tools/summarize_profiles_r03.py $TAG
