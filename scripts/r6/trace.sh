# usage: bash scripts/r6/trace.sh TAG [ENV=..]...   -- c2 kernel trace -> steady / gaps / overlap tables under gpurun_out/r06
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p_$TAG
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $O/${TAG}_trace_run.log 2>&1
T=$(find /tmp/p_$TAG -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/p_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
python $R/scripts/trace_gaps.py $T > $O/${TAG}_trace_gaps.txt 2>&1
python $R/scripts/trace_overlap.py $T > $O/${TAG}_overlap.txt 2>&1
python $R/scripts/trace_steady.py $T flow_nll 6 > $O/${TAG}_steady.txt 2>&1
python $R/scripts/trace_by_grid.py $T tn_glds > $O/${TAG}_tn_by_grid.txt 2>&1
python $R/scripts/trace_by_grid.py $T lat8 >> $O/${TAG}_tn_by_grid.txt 2>&1
python $R/scripts/trace_prev.py $T conv3x3_k64 > $O/${TAG}_k64_prev.txt 2>&1
python $R/scripts/trace_prev.py $T "igemm_nt_glds_kernel<bool _Accum, int, E, 4, 5, 2, 3, 1, true, 2>" > $O/${TAG}_conv2_prev.txt 2>&1
