J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"), d["roofline"]["avg_launch_us"], d["roofline"].get("isolated_avg_launch_us"))'
for v in 1 0; do echo -n "BD=$v "; IPOKE_NT_BD=$v python scripts/probe_gemm.py 20 2>/dev/null | tail -1; done
for v in 1 0; do echo -n "BD=$v B=40 "; IPOKE_NT_BD=$v python scripts/probe_gemm.py 40 2>/dev/null | tail -1; done
timeout 900 python -m pytest tests/test_flow_gpu.py tests/test_units_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do for v in 1 0; do
echo -n "BD=$v  "; IPOKE_NT_BD=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"
done; done
