R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_encoder_kernels_gpu.py -m gpu -x -q -s -k "hands or residual" 2>&1 | tail -12
python -m pytest tests -m gpu -x -q -k "decoder or sample or first_stage" 2>&1 | tail -4
for v in 0 1 0 1; do
  IPOKE_NO_NEXT_STATS=$v python bench.py --config c5 --no-cpu-baseline 2>$O/c33_$v.err | tail -1 > $O/c33_$v.json
  python -c "import json;d=json.load(open('$O/c33_$v.json'));h=d.get('hipgraph');print('NO_NEXT_STATS=$v',d['ms_per_step'],h['pipelined_ms_per_step'],h['full_graph_ms_per_step'])" || tail -5 $O/c33_$v.err
done
