set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02f
mkdir -p $O
# c2 kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/c2_trace_run.log 2>&1
cp $(find /tmp/p_c2 -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
# PMC passes (separate runs)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_$c.log 2>&1
  python $R/scripts/pmc_summary.py $(find /tmp/p_$c -name "*counter_collection.csv" | head -1) $O/r02_bench_pmc_$c.json igemm_nt_glds igemm_tn_glds conv3x3_s8 macow_unit adam_amsgrad
done
# c5 / c4 traces
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python $R/bench.py --config c5 --steps 5 --warmup 3 --no-cpu-baseline > $O/c5_trace_run.log 2>&1
cp $(find /tmp/p_c5 -name "*kernel_stats.csv" | head -1) $O/r02_c5_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $R/bench.py --config c4 --steps 5 --warmup 3 --no-cpu-baseline > $O/c4_trace_run.log 2>&1
cp $(find /tmp/p_c4 -name "*kernel_stats.csv" | head -1) $O/r02_c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4gan -- python $R/bench.py --config c4gan --steps 3 --warmup 1 --no-cpu-baseline > $O/c4gan_trace_run.log 2>&1
cp $(find /tmp/p_c4gan -name "*kernel_stats.csv" | head -1) $O/r02_c4gan_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fvd -- python $R/bench.py --config fvd --steps 5 --warmup 3 --no-cpu-baseline > $O/fvd_trace_run.log 2>&1
cp $(find /tmp/p_fvd -name "*kernel_stats.csv" | head -1) $O/r02_fvd_kernel_stats.csv
# un-profiled bench lines
python $R/bench.py --config fvd --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_fvd_bench_line.json
python $R/bench.py --config fvd --fvd-dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/r02_fvd_bench_line.json
python $R/bench.py --config c4gan --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_c4gan_bench_line.json
python $R/bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_c5_bench_line.json
python $R/bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_c4_bench_line.json
python $R/bench.py 2>/dev/null | tail -1 > $O/r02_bench_line.json
ls -la $O
