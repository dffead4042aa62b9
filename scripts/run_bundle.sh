set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4m
mkdir -p $O
cd $R && timeout 900 python -m pytest tests/test_flow_gpu.py tests/test_full_gpu.py -q -x 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
cd /tmp
B="python $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-secondary"
for i in 1 2 3; do $B 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'], d['ms_per_step_median'], d['loss'], d['roofline']['traffic'])" >> $O/ab.txt; done
cat $O/ab.txt
