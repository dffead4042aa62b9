bash scripts/profile_round.sh r03c > /dev/null 2>&1
ls gpurun_out/r03c | head -30
