#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_train_mode_gpu.py tests/test_vae_train_gpu.py -x -q 2>&1 | tail -4
run() { python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('loss'))" || tail -5 gpurun_out/err.log; }
echo "== folded"; run
echo "== unfolded"; IPOKE_STEM_TRAIN_UNFOLDED=1 run
echo "== folded"; run
