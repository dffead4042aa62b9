set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4h
mkdir -p $O
cd $R && timeout 900 python -m pytest tests/test_units_gpu.py tests/test_flow_gpu.py tests/test_full_gpu.py tests/test_second_stage_options_gpu.py -q -x 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
timeout 600 python -m pytest "tests/test_bench_configs_gpu.py::test_full_size_flow" -q -x -s 2>&1 | grep -E "passed|failed|Error|error|B=20|B=40" | tail -20 > $O/tests2.txt; cat $O/tests2.txt
cd /tmp
B="python $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-secondary"
for v in "IPOKE_C2_STRAIGHT=0" "IPOKE_C2_STRAIGHT=1" "IPOKE_C2_STRAIGHT=0" "IPOKE_C2_STRAIGHT=1"; do env $v $B 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['ms_per_step'], d['ms_per_step_median'], d['loss'], 'conv2 insitu', r['avg_launch_us'], 'iso', r.get('isolated_avg_launch_us'))" >> $O/ab.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; done
cat $O/ab.txt
