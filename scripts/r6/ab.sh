# usage: bash scripts/r6/ab.sh TAG "ENV1=.. ENV2=.." ["..." ...]   -- quick c2 A/B lines inside one GPU call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
TAG=$1; shift
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 8 2>$O/${TAG}_$i.err | tail -1 > $O/${TAG}_$i.json
  python - <<PY
import json
try:
    d=json.load(open("$O/${TAG}_$i.json")); print("$e", d["ms_per_step"], d["ms_per_step_median"], d.get("handoff_timeouts"), d["loss"])
except Exception as ex:
    print("$e FAILED", ex); print(open("$O/${TAG}_$i.err").read()[-600:])
PY
done
