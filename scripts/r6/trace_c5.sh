R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python $R/bench.py --config c5 --quick --steps 6 --warmup 3 --no-cpu-baseline > $O/c5t_run.log 2>&1
T=$(find /tmp/p_c5 -name "*kernel_trace.csv" | head -1)
python $R/scripts/trace_steady.py $T gru_fused_fwd 4 > $O/c5t_steady.txt 2>&1
python $R/scripts/trace_gaps.py $T > $O/c5t_gaps.txt 2>&1
