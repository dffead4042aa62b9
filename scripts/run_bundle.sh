#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
for c in c4 c5; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -- python $R/bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline > $O/${c}_stats_run.log 2>&1
cp $(find /tmp/p_$c -name "*kernel_stats.csv" | head -1) $O/${c}_kernel_stats_now.csv
done
