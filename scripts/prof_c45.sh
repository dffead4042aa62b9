cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s5prof
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python $R/bench.py --config c5 --steps 5 --warmup 3 --no-cpu-baseline > $O/c5_trace_run.log 2>&1
cp $(find /tmp/p_c5 -name "*kernel_stats.csv" | head -1) $O/c5_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $R/bench.py --config c4 --steps 5 --warmup 3 --no-cpu-baseline > $O/c4_trace_run.log 2>&1
cp $(find /tmp/p_c4 -name "*kernel_stats.csv" | head -1) $O/c4_kernel_stats.csv
ls $O
