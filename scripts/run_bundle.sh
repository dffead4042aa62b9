J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"))'
timeout 900 python -m pytest tests/test_conv_halo_gpu.py tests/test_flow_gpu.py tests/test_units_gpu.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do
echo -n "new  "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"
echo -n "prev "; IPOKE_LIB_PATH=$PWD/scripts/exp/libipoke_prev.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"
done
echo -n "c5 new  "; python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"
echo -n "c5 prev "; IPOKE_LIB_PATH=$PWD/scripts/exp/libipoke_prev.so python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"
