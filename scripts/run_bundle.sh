J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"))'
for v in 0 1 0 1; do
echo -n "NO_ENC_FORK=$v  "; IPOKE_NO_ENC_FORK=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"
done
python scripts/probe_enc.py 2>/dev/null | tail -1
IPOKE_NO_ENC_FORK=1 python scripts/probe_enc.py 2>/dev/null | tail -1
timeout 900 python -m pytest tests/test_full_gpu.py tests/test_second_stage_options_gpu.py -m gpu -x -q 2>&1 | tail -3
