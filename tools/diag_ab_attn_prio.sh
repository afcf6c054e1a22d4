#!/bin/bash
# Diagnostic: two builds of the library side by side on one device, interleaved (rule 24) - the in-tree build against
# lib/exp/*.so (built by hand with other -D flags; used for the static wave priority of the 8-wave attention workgroups).
# usage (GPU box): tools/diag_ab_attn_prio.sh
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/tests" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
cd "$ROOT" || exit 1
for round in 1 2; do
  for so in memory-augmented-vlm_amd/lib/libmavlm.so memory-augmented-vlm_amd/lib/exp/*.so; do
    [ -f "$so" ] || continue
    echo "== $(basename $so) round $round"
    MAVLM_LIB=$ROOT/$so STEPS=20 python tools/diag_batch_modes.py 64 1x2 2>&1 | grep -E "frames/s|attention_fwd"
  done
done
