J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"))'
for r in 1 2; do for v in 12 16 20 14; do echo -n "PIECES=$v "; IPOKE_PIECES=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"; done; done
