R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_wgrad_batched_gpu.py -x -q -m gpu -k "halo_staged" -s 2>&1 | grep -v "^$" | tail -12
for e in "A=1" "IPOKE_WGRAD_HALO=0" "A=2" "IPOKE_WGRAD_HALO=0"; do
  env $e python bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', '$e', d['ms_per_step'], d['loss'])"
done
