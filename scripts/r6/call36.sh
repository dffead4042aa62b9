R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_vae_bwd_units_gpu.py tests/test_c4_dispatch_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1 0 1; do
  IPOKE_NO_RES_POST=$v python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c36_$v.err | tail -1 > $O/c36_$v.json
  python -c "import json;d=json.load(open('$O/c36_$v.json'));print('NO_RES_POST=$v',d['ms_per_step'],d.get('loss'))" || tail -5 $O/c36_$v.err
done
