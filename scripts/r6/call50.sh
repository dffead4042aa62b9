R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_encoder_kernels_gpu.py -m gpu -x -q -k "four_tap or phases" 2>&1 | tail -4
timeout 900 python -m pytest tests -m gpu -x -q -k "decoder or sample or first_stage or c4_dispatch or train_mode" 2>&1 | tail -3
for v in 1 0 1 0; do
  IPOKE_HALO16_PHASE=$v timeout 300 python bench.py --config c5 --no-cpu-baseline 2>$O/c50_$v.err | tail -1 > $O/c50_$v.json
  python -c "import json;d=json.load(open('$O/c50_$v.json'));print('c5 HALO16_PHASE=$v',d['ms_per_step'])" || tail -5 $O/c50_$v.err
  IPOKE_HALO16_PHASE=$v timeout 300 python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c50_$v.err | tail -1 > $O/c50_$v.json
  python -c "import json;d=json.load(open('$O/c50_$v.json'));print('c4 HALO16_PHASE=$v',d['ms_per_step'],d['loss'])" || tail -5 $O/c50_$v.err
done
