R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_wgrad_batched_gpu.py tests/test_vae_bwd_units_gpu.py tests/test_c4_dispatch_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in 1 0 1 0; do
  IPOKE_REDUCE_VEC=$v python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c25_$v.err | tail -1 > $O/c25_$v.json
  python -c "import json;d=json.load(open('$O/c25_$v.json'));print('REDUCE_VEC=$v',d['ms_per_step'],d.get('ms_per_step_median'),d.get('loss'))"
done
