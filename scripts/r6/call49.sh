R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2 -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $O/c49_trace.log 2>&1
T=$(find /tmp/p_c2 -name "*kernel_trace.csv" | head -1)
python $R/scripts/r6/agg_trace.py $T > $O/c49_agg.txt 2>&1
cd $R
IPOKE_CONV_LOG=1 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-secondary > $O/c49_c2.out 2> $O/c49_c2_log.txt
python scripts/conv_log_summary.py $O/c49_c2_log.txt CONV 4 40 > $O/c49_convs.txt
