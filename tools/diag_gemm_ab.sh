#!/bin/bash
# usage: tools/diag_gemm_ab.sh   (on the GPU box; after tools/diag_gemm_ablate.sh in the build container)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
cd "$ROOT" || exit 1
for so in memory-augmented-vlm_amd/lib/exp/gemm_*.so; do
  echo "== $(basename $so)"
  MAVLM_LIB=$ROOT/$so python tools/diag_gemm_loop.py 2>&1 | grep "rows"
done
