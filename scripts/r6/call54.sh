R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "first_stage or train_mode or c4_dispatch or phases" 2>&1 | tail -3
for v in 0 1 0 1; do
  IPOKE_NO_PHASE_PAD=$v timeout 300 python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c54_$v.err | tail -1 > $O/c54_$v.json
  python -c "import json;d=json.load(open('$O/c54_$v.json'));print('c4 NO_PHASE_PAD=$v',d['ms_per_step'],d['loss'])" || tail -5 $O/c54_$v.err
done
