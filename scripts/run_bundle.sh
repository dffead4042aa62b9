J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"), [k["avg_launch_us"] for k in d["roofline_other_kernels"]])'
timeout 900 python -m pytest tests/test_macow_unit_gpu.py tests/test_flow_gpu.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do for v in 0 1; do
echo -n "NO_UNIT_ZC=$v  "; if [ $v = 1 ]; then export IPOKE_NO_UNIT_ZC=1; else unset IPOKE_NO_UNIT_ZC; fi; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"
done; done
