#!/bin/bash
# Diagnostic: time ablated builds of attn_fwd3 (lib/exp/libmavlm_abl{0..3}.so, built by hand with -DABL=n from a scratch
# copy of csrc/: 1 = no fma before the exp, 2 = + no running max, 3 = + no row sums; results are WRONG, only the time counts)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for a in 0 1 2 3; do
    echo "== ABL=$a round $round"
    MAVLM_LIB=$GRAFT_REPO_ROOT/memory-augmented-vlm_amd/lib/exp/libmavlm_abl$a.so ATTN_ONLY3=1 python tests/bench_ops.py attn 20 2>&1 | grep "attn R"
  done
done
