R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp PYTHONPATH=$R
python -m pytest tests/test_c4_dispatch_gpu.py tests/test_encoder_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests -m gpu -x -q -k "decoder or sample or first_stage" 2>&1 | tail -3
python scripts/r6/probe_norm.py 2>&1 | grep "^[0-9]"
for v in 1 0 1 0; do
  IPOKE_GN_APPLY_FRAMES=$v python bench.py --config c5 --no-cpu-baseline 2>$O/c42_$v.err | tail -1 > $O/c42_$v.json
  python -c "import json;d=json.load(open('$O/c42_$v.json'));print('c5 APPLY_FRAMES=$v',d['ms_per_step'])" || tail -5 $O/c42_$v.err
  IPOKE_GN_APPLY_FRAMES=$v python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c42_$v.err | tail -1 > $O/c42_$v.json
  python -c "import json;d=json.load(open('$O/c42_$v.json'));print('c4 APPLY_FRAMES=$v',d['ms_per_step'])" || tail -5 $O/c42_$v.err
done
