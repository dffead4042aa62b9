python -m pytest tests/test_macow_unit_gpu.py tests/test_flow_gpu.py -m gpu -q 2>&1 | tail -3
for C in 64 32 8; do python scripts/probe_unit.py $C 20 2>&1 | grep -v "^$" | tail -4; done
for v in "" "" ""; do echo "== $v"; env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"ms_per_step\"], d[\"roofline\"][\"avg_launch_us\"], d[\"roofline\"][\"isolated_avg_launch_us\"], [ (k[\"kernel\"][:14], k[\"avg_launch_us\"]) for k in d[\"roofline_other_kernels\"]])"; done
python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['ms_per_step'])"
