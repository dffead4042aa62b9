cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4o
mkdir -p $O
for v in 512 2048 1024 512 2048; do IPOKE_ROWSCALE_ROWS=$v python $R/bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 rows=$v', d['ms_per_step'], d['loss'])" >> $O/ab.txt; done
B="python $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-secondary"
for v in 128 160 192 224 128 160 192; do IPOKE_NATIVE_ADAM_BLOCKS=$v $B 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 adam_blocks=$v', d['ms_per_step'], d['ms_per_step_median'])" >> $O/ab.txt; done
cat $O/ab.txt
