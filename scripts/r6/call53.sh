R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_encoder_kernels_gpu.py -m gpu -x -q -k "phase" 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q -k "decoder or sample" 2>&1 | tail -3
for v in 0 1 0 1; do
  IPOKE_NO_PHASE_PAD=$v timeout 300 python bench.py --config c5 --no-cpu-baseline 2>$O/c53_$v.err | tail -1 > $O/c53_$v.json
  python -c "import json;d=json.load(open('$O/c53_$v.json'));print('c5 NO_PHASE_PAD=$v',d['ms_per_step'])" || tail -5 $O/c53_$v.err
done
