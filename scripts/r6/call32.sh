R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_encoder_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests -m gpu -x -q -k "decoder or sample or first_stage" 2>&1 | tail -4
for v in 0 1 0 1; do
  IPOKE_NO_RES_POST=$v python bench.py --config c5 --no-cpu-baseline 2>$O/c32_$v.err | tail -1 > $O/c32_$v.json
  python -c "import json;d=json.load(open('$O/c32_$v.json'));print('NO_RES_POST=$v',d['ms_per_step'],d.get('hipgraph'))" || tail -5 $O/c32_$v.err
done
