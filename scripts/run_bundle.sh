#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_train_mode_gpu.py tests/test_vae_train_gpu.py -x -q 2>&1 | tail -3
run() { python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('loss'))" || tail -5 gpurun_out/err.log; }
echo "== new"; run
echo "== new"; run
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python /root/repo/bench.py --config c4 --steps 6 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/c4_stats_run.log 2>&1
f=$(find /tmp/p_c4 -name "*kernel_stats.csv" | head -1); cp $f /root/repo/gpurun_out/c4_kernel_stats_now.csv
