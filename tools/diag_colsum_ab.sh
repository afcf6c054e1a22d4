#!/bin/bash
# Diagnostic: column-sum pass, library builds side by side on one device (lib/exp/*.so against the in-tree build),
# interleaved over two rounds.  usage: tools/diag_colsum_ab.sh
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
cd "$ROOT" || exit 1
for round in 1 2; do
  for so in memory-augmented-vlm_amd/lib/exp/*.so memory-augmented-vlm_amd/lib/libmavlm.so; do
    echo "== $(basename $so) round $round"
    MAVLM_LIB=$ROOT/$so COLSUM_QUICK=1 python tools/diag_colsum_fill.py 2>&1 | grep "S="
  done
done
