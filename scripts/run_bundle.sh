#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
date
timeout -s INT 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5p -- python -X faulthandler $R/bench.py --config c5 --steps 5 --warmup 3 --no-cpu-baseline > $O/c5_pipe_trace_run.log 2>&1
echo "c5 pipelined trace rc=$?"; date
grep -v "^E2026\|^W2026\|^I2026" $O/c5_pipe_trace_run.log | tail -30
