R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
for e in "A=1" "IPOKE_K64=0" "A=2" "IPOKE_K64=0"; do
  env $e python bench.py --config c5 --quick --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', '$e', d['ms_per_step'])"
done
python -m pytest tests/test_bench_configs_gpu.py -x -q -m gpu -k "full_size_flow and bf16" 2>&1 | tail -2
python -m pytest tests/test_full_gpu.py -x -q -m gpu -k "reproducible and bf16" 2>&1 | tail -2
