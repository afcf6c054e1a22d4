#!/bin/bash
# Diagnostic: rocprofv3 kernel stats of the attention schedules side by side (tools/diag_streamk.py).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_sk -- python3 $ROOT/tools/diag_streamk.py > $ROOT/gpurun_out/prof_sk.log 2>&1
cat $ROOT/gpurun_out/prof_sk/*/*kernel_stats.csv | cut -c1-200 | head -12
