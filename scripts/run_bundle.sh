#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_a -- python $R/bench.py --config c5 --steps 6 --warmup 3 --no-cpu-baseline > $O/a.log 2>&1
cp $(find /tmp/p_a -name "*kernel_stats.csv" | head -1) $O/c5_stats_fused.csv
IPOKE_NO_FUSED_RESIDUAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b -- python $R/bench.py --config c5 --steps 6 --warmup 3 --no-cpu-baseline > $O/b.log 2>&1
cp $(find /tmp/p_b -name "*kernel_stats.csv" | head -1) $O/c5_stats_sep.csv
