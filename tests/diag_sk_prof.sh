cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sk -- python3 $GRAFT_REPO_ROOT/tests/diag_streamk.py > $GRAFT_REPO_ROOT/gpurun_out/prof_sk.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof_sk/*/*kernel_stats.csv | cut -c1-200 | head -12
