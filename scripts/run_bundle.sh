J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"))'
python scripts/probe_enc.py 2>/dev/null | tail -1
IPOKE_HALO16=0 python scripts/probe_enc.py 2>/dev/null | tail -1
for i in 1 2; do
echo -n "new  "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"
echo -n "old  "; IPOKE_HALO16=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"
done
echo -n "c4 new  "; python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"
echo -n "c4 old  "; IPOKE_HALO16=0 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"
echo -n "c4gan new  "; python bench.py --config c4gan --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$J"
echo -n "c4gan old  "; IPOKE_HALO16=0 python bench.py --config c4gan --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$J"
timeout 1200 python -m pytest tests/test_conv_halo_gpu.py tests/test_vae_gpu.py tests/test_bench_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
