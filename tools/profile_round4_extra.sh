#!/bin/bash
# Round-4 evidence beside tools/profile_round4.sh.  usage (GPU box): bash tools/profile_round4_extra.sh
# Output: gpurun_out/r04_extra/*.txt (copied into profiles/r04_* by hand).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r04_extra"
mkdir -p "$OUT"
cd "$ROOT" || exit 1
{
  echo "# tools/diag_vs_hipblaslt.py: this library vs the vendor libraries, same box, interleaved (the product calls neither)"
  timeout -k 10 300 python tools/diag_vs_hipblaslt.py || exit 1
} > "$OUT/vs_vendor_libraries.txt" 2>&1 || { tail -5 "$OUT/vs_vendor_libraries.txt"; exit 1; }
echo "vendor done"
{
  echo "# tools/diag_gemm_shapes.py: every GEMM shape / epilogue of the path (automatic kernel choice) vs hipBLASLt (bias only); the Residual block"
  timeout -k 10 300 python tools/diag_gemm_shapes.py || exit 1
  echo "# tools/diag_gemm128.py: the 128x256 two-workgroups-per-CU kernel (hook 129) against the automatic choice and hipBLASLt"
  timeout -k 10 300 python tools/diag_gemm128.py || exit 1
  echo "# tools/diag_gemm128_small.py: the same at the small row counts of a single video with 8 memory tokens"
  timeout -k 10 300 python tools/diag_gemm128_small.py || exit 1
  echo "# tools/diag_gemm_r4.py: tile order of the persistent kernel, GELU vs ReLU epilogue"
  timeout -k 10 300 python tools/diag_gemm_r4.py || exit 1
} > "$OUT/gemm_shapes.txt" 2>&1 || { tail -5 "$OUT/gemm_shapes.txt"; exit 1; }
echo "gemm shapes done"
{
  echo "# tools/diag_attn_order.py: XCD-affine unit order of the stream-K attention against the position order (time)"
  timeout -k 10 300 python tools/diag_attn_order.py || exit 1
  echo "# tools/pmc_attn_order.sh: FETCH_SIZE x2 / WRITE_SIZE per launch under the two orders"
  timeout -k 10 400 bash tools/pmc_attn_order.sh || exit 1
} > "$OUT/attn_unit_order.txt" 2>&1 || { tail -5 "$OUT/attn_unit_order.txt"; exit 1; }
echo "attention order done"
{
  echo "# tools/diag_batch_modes.py, one MI355X: streams x row batch, same process, interleaved blocks"
  echo "## M = 64, D = 1024 (headline shape)"
  timeout -k 10 300 python tools/diag_batch_modes.py 64 2x1 1x2 2x2 2x4 || exit 1
  echo "## M = 8, D = 1024 (checkpoint shape)"
  timeout -k 10 300 python tools/diag_batch_modes.py 8 2x1 1x8 2x8 2x4 || exit 1
  echo "## M = 8, D = 3584, 256-frame videos (BASELINE.json configs[2], OneVision-7B width)"
  HIDDEN=3584 FRAMES=256 STEPS=3 timeout -k 10 400 python tools/diag_batch_modes.py 8 1x1 1x4 2x4 || exit 1
  echo "# HIDDEN=3584 MEM_TOKENS=8 tools/bench_train.py 3 (one training step at the OneVision-7B width, 64 frames)"
  HIDDEN=3584 MEM_TOKENS=8 timeout -k 10 300 python tools/bench_train.py 3 || exit 1
  echo "# tools/bench_train.py 3 (one training step at the bench shape: M = 64, D = 1024)"
  timeout -k 10 300 python tools/bench_train.py 3 || exit 1
} > "$OUT/batch_modes.txt" 2>&1 || { tail -5 "$OUT/batch_modes.txt"; exit 1; }
echo "batch modes done"
timeout -k 10 600 bash tools/pmc_gemm.sh > "$OUT/gemm_pmc.txt" 2>&1 && echo "gemm pmc done"
timeout -k 10 400 bash tools/pmc_gemm_ab.sh > "$OUT/gemm_pmc_ab.txt" 2>&1 && echo "gemm pmc a/b done"
timeout -k 10 400 bash tools/profile_ov7b.sh > "$OUT/ov7b.txt" 2>&1 && echo "ov7b done"
