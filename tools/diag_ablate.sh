#!/bin/bash
# Diagnostic: time alternative builds of the library (lib/exp/*.so, built by hand from a scratch copy of csrc/ with -D
# flags) with tools/bench_ops.py, interleaved over two rounds on one device.  usage: tools/diag_ablate.sh [op]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
cd "$ROOT" || exit 1
OP=${1:-attn}
for round in 1 2; do
  for so in memory-augmented-vlm_amd/lib/exp/*.so; do
    echo "== $(basename $so) round $round"
    MAVLM_LIB=$ROOT/$so ATTN_ONLY3=1 python tools/bench_ops.py $OP 20 2>&1 | grep -E "attn R|colsum"
  done
done
