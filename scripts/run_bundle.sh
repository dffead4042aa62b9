bash scripts/profile_round.sh r03b > /dev/null 2>&1
ls gpurun_out/r03b | head -30
cat gpurun_out/r03b/r03b_bench_line.json | cut -c1-600
