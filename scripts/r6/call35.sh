R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp
export TMPDIR=/tmp
for v in 0 1; do
IPOKE_NO_RES_POST=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4_$v -- python $R/bench.py --config c4 --steps 8 --warmup 3 --no-cpu-baseline > $O/c35_trace_$v.log 2>&1
T=$(find /tmp/p_c4_$v -name "*kernel_trace.csv" | head -1)
python $R/scripts/trace_steady.py $T reparam_kernel 6 > $O/c35_c4_steady_$v.txt 2>&1
done
