J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"))'
for i in 1 2; do
echo -n "new  "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"
echo -n "old  "; IPOKE_GN_FUSED=0 IPOKE_NO_STEM_FOLD=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$J"
done
echo -n "c4 new  "; python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"
echo -n "c4 old  "; IPOKE_GN_FUSED=0 IPOKE_NO_STEM_FOLD=1 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"
echo -n "c5 new  "; python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"
echo -n "c5 old  "; IPOKE_GN_FUSED=0 IPOKE_NO_STEM_FOLD=1 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$J"
