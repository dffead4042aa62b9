R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests -q -m gpu -x > $O/suite.log 2>&1; tail -15 $O/suite.log
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/suite_c5.json
python bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/suite_c4.json
python - <<PY
import json
for c in ("c5","c4"):
    d=json.load(open("$O/suite_%s.json"%c)); print(c, d["ms_per_step"], d.get("hipgraph",{}).get("full_graph_ms_per_step"), d.get("roofline_note"), (d.get("roofline") or {}).get("kernel"))
PY
