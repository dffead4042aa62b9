R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $R/bench.py --config c4 --steps 8 --warmup 3 --no-cpu-baseline > $O/c26_trace.log 2>&1
T=$(find /tmp/p_c4 -name "*kernel_trace.csv" | head -1)
python $R/scripts/trace_steady.py $T reparam_kernel 6 > $O/c26_c4_steady.txt 2>&1
python $R/scripts/trace_gaps.py $T > $O/c26_c4_gaps.txt 2>&1
python $R/scripts/trace_overlap.py $T > $O/c26_c4_overlap.txt 2>&1
head -50 $O/c26_c4_steady.txt
