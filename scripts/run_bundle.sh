J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"), d.get("loss"))'
for v in 0 1; do echo -n "c4 NO_CT_PHASES=$v "; IPOKE_NO_CT_PHASES=$v python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"; done
for v in 0 1; do echo -n "c4gan NO_CT_PHASES=$v "; IPOKE_NO_CT_PHASES=$v python bench.py --config c4gan --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$J"; done
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_train_mode_gpu.py tests/test_vae_bwd_units_gpu.py tests/test_vae_train_gpu.py -m gpu -x -q 2>&1 | tail -3
