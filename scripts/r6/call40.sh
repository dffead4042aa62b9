R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_encoder_kernels_gpu.py -m gpu -x -q 2>&1 | tail -2
for lib in prev hip prev hip; do
  IPOKE_LIB_PATH=$R/ipoke_amd/libipoke_$lib.so python bench.py --config c5 --no-cpu-baseline 2>$O/c40_$lib.err | tail -1 > $O/c40_$lib.json
  python -c "import json;d=json.load(open('$O/c40_$lib.json'));print('c5 $lib',d['ms_per_step'])" || tail -5 $O/c40_$lib.err
  IPOKE_LIB_PATH=$R/ipoke_amd/libipoke_$lib.so python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c40_$lib.err | tail -1 > $O/c40_$lib.json
  python -c "import json;d=json.load(open('$O/c40_$lib.json'));print('c4 $lib',d['ms_per_step'])" || tail -5 $O/c40_$lib.err
done
