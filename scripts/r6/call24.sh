R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
IPOKE_WGRAD_LOG=1 python bench.py --config c4 --steps 2 --warmup 2 --no-cpu-baseline > $O/c24_c4.out 2> $O/c24_c4_log.txt
python scripts/conv_log_summary.py $O/c24_c4_log.txt WGRAD 5 22
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $R/bench.py --config c4 --steps 5 --warmup 3 --no-cpu-baseline > $O/c24_trace.log 2>&1
cp $(find /tmp/p_c4 -name "*kernel_stats.csv" | head -1) $O/c24_c4_kernel_stats.csv
