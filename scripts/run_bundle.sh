cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
for s in 4 2; do
  rm -rf /tmp/p_c2
  IPOKE_UNIT_SPLIT=$s rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/r05/c2_trace_run_s$s.log 2>&1
  python $R/scripts/trace_steady.py $(find /tmp/p_c2 -name "*kernel_trace.csv" | head -1) flow_nll 6 > $R/gpurun_out/r05/steady_s$s.txt 2>&1
  python $R/scripts/trace_overlap.py $(find /tmp/p_c2 -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/r05/overlap_s$s.txt 2>&1
done
head -30 $R/gpurun_out/r05/steady_s4.txt
