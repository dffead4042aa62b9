R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export PYTHONPATH=$R TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_norm -- python $R/scripts/r6/probe_norm.py > $O/c39.log 2>&1
T=$(find /tmp/p_norm -name "*kernel_trace.csv" | head -1)
python $R/scripts/r6/agg_trace.py $T gn_ > $O/c39_grid.txt 2>&1
cat $O/c39_grid.txt
