# round 6, GPU call 1: new unit tests, c2 A/B of the deterministic accumulation and of the deferred optimizer pieces
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_units_gpu.py -x -q -m gpu -k "split_k_accumulation or skinny" > $O/c1_units.log 2>&1; tail -3 $O/c1_units.log
b() { tag=$1; shift; env "$@" python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 8 2>$O/c1_$tag.err | tail -1 > $O/c1_$tag.json; python - <<PY
import json; d=json.load(open("$O/c1_$tag.json")); print("$tag", d["ms_per_step"], d["ms_per_step_median"], d.get("handoff_timeouts"), d["loss"])
PY
}
b base A=1
b atomics IPOKE_DGRAD_ATOMICS=1
b defer4 IPOKE_ADAM_DEFER=4
b defer8 IPOKE_ADAM_DEFER=8
b defer12 IPOKE_ADAM_DEFER=12
b base2 A=1
timeout 900 python -m pytest tests/test_full_gpu.py -x -q -m gpu -k "reproducible" -s > $O/c1_repro.log 2>&1; tail -8 $O/c1_repro.log
