#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_vae_bwd_units_gpu.py -x -q -s -k "fused_gru" 2>&1 | grep "fused GRU\|gradients\|passed\|failed"
run() { python bench.py --config $1 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('loss'))" || tail -5 gpurun_out/err.log; }
echo "== c4 fused"; run c4
echo "== c4 unfused"; IPOKE_GRU_FUSED=0 run c4
echo "== c4 fused"; run c4
python -m pytest tests/test_train_mode_gpu.py tests/test_vae_train_gpu.py tests/test_vae_gpu.py -x -q 2>&1 | tail -3
