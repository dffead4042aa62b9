R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $R/bench.py --config c4 --steps 6 --warmup 3 --no-cpu-baseline > $O/c43_trace.log 2>&1
T=$(find /tmp/p_c4 -name "*kernel_trace.csv" | head -1)
python $R/scripts/r6/agg_trace.py $T > $O/c43_agg.txt 2>&1
