R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
for v in 8 6 8 6; do
  IPOKE_HALO16_PHASE_FILL=$v timeout 300 python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c52_$v.err | tail -1 > $O/c52_$v.json
  python -c "import json;d=json.load(open('$O/c52_$v.json'));print('c4 FILL=$v',d['ms_per_step'])" || tail -5 $O/c52_$v.err
  IPOKE_HALO16_PHASE_FILL=$v timeout 300 python bench.py --config c5 --no-cpu-baseline 2>$O/c52_$v.err | tail -1 > $O/c52_$v.json
  python -c "import json;d=json.load(open('$O/c52_$v.json'));print('c5 FILL=$v',d['ms_per_step'])" || tail -5 $O/c52_$v.err
done
