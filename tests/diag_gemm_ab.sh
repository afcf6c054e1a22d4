#!/bin/bash
# usage: tests/diag_gemm_ab.sh   (on the GPU box; after tests/diag_gemm_ablate.sh in the build container)
cd $GRAFT_REPO_ROOT
for so in memory-augmented-vlm_amd/lib/exp/gemm_*.so; do
  echo "== $(basename $so)"
  MAVLM_LIB=$GRAFT_REPO_ROOT/$so python tests/diag_gemm_loop.py 2>&1 | grep "rows"
done
