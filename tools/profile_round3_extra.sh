#!/bin/bash
# Round-3 evidence beside tools/profile_round3.sh: (streams x row batch) tables for the three configs, the wide-head forward
# (both kernel forms; timing + PMC) and the GEMM PMC table.  usage (GPU box): bash tools/profile_round3_extra.sh
# Output: gpurun_out/r03_extra/*.txt (copied into profiles/r03_* by hand).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r03_extra"
mkdir -p "$OUT"
cd "$ROOT" || exit 1
{
  echo "# tools/diag_batch_modes.py, one MI355X: streams x row batch, same process, interleaved blocks"
  echo "## M = 64, D = 1024 (headline shape)"
  timeout -k 10 300 python tools/diag_batch_modes.py 64 2x1 1x2 2x2 2x4 || exit 1
  echo "## M = 8, D = 1024 (checkpoint shape)"
  timeout -k 10 300 python tools/diag_batch_modes.py 8 2x1 1x8 2x8 2x4 || exit 1
  echo "## M = 8, D = 3584, 256-frame videos (BASELINE.json configs[2], OneVision-7B width; wide-head attention batched over grid z)"
  HIDDEN=3584 FRAMES=256 timeout -k 10 400 python tools/diag_batch_modes.py 8 1x1 2x1 1x4 2x4 || exit 1
} > "$OUT/batch_modes.txt" 2>&1 || { tail -5 "$OUT/batch_modes.txt"; exit 1; }
echo "batch modes done"
{
  echo "# tools/diag_wide_groups.py: head_dim-448 forward, [1] = 16-query waves (rounds 1-2), [2] = 32-query waves on 32x32x16 (round 3)"
  timeout -k 10 300 python tools/diag_wide_groups.py || exit 1
  echo "# tools/bench_ops.py wide"
  timeout -k 10 300 python tools/bench_ops.py wide || exit 1
  echo "# HIDDEN=3584 MEM_TOKENS=8 tools/bench_train.py 3 (one training step at the OneVision-7B width, 64 frames)"
  HIDDEN=3584 MEM_TOKENS=8 timeout -k 10 300 python tools/bench_train.py 3 || exit 1
} > "$OUT/wide_head.txt" 2>&1 || { tail -5 "$OUT/wide_head.txt"; exit 1; }
echo "wide head done"
MODES=1,2 timeout -k 10 600 bash tools/pmc_wide.sh > "$OUT/wide_head_pmc.txt" 2>&1 && echo "wide pmc done"
timeout -k 10 600 bash tools/pmc_gemm.sh > "$OUT/gemm_pmc.txt" 2>&1 && echo "gemm pmc done"
