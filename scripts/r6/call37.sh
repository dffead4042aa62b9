R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
for v in 64 128 32 64 128; do
  IPOKE_NORM_BWD_POS=$v python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c37_$v.err | tail -1 > $O/c37_$v.json
  python -c "import json;d=json.load(open('$O/c37_$v.json'));print('NORM_BWD_POS=$v',d['ms_per_step'],d.get('loss'))" || tail -5 $O/c37_$v.err
done
