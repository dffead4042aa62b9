R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
for c in "--config c3" "--config c2 --dtype f32 --steps 8 --warmup 3"; do
  python bench.py $c --no-secondary --no-cpu-baseline 2>$O/c13.err | tail -1 > $O/c13.json
  python - <<PY
import json
d=json.load(open("$O/c13.json")); print("$c", d["ms_per_step"], d["value"], d.get("handoff_timeouts"), d["loss"], d["roofline"]["frac"], [ (k["kernel"][:40], k["avg_launch_us"]) for k in d.get("roofline_other_kernels",[])])
PY
done
