R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export IPOKE_BENCH_NO_PIPELINE=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python $R/bench.py --config c5 --steps 6 --warmup 3 --no-cpu-baseline > $O/c41_trace.log 2>&1
T=$(find /tmp/p_c5 -name "*kernel_trace.csv" | head -1)
python $R/scripts/r6/agg_trace.py $T > $O/c41_agg.txt 2>&1
python $R/scripts/trace_steady.py $T gru_fused_fwd 4 > $O/c41_steady.txt 2>&1
head -60 $O/c41_steady.txt
