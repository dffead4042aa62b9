set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
b() { tag=$1; shift; env "$@" python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 8 2>$O/c4_$tag.err | tail -1 > $O/c4_$tag.json; python - <<PY
import json; d=json.load(open("$O/c4_$tag.json")); print("$tag", d["ms_per_step"], d["ms_per_step_median"], d.get("handoff_timeouts"), d["loss"])
PY
}
b base A=1
b p2d_fwd IPOKE_PREFETCH_2D=fwd
b p2d_bwd IPOKE_PREFETCH_2D=bwd
b base2 A=1
b p2d_fwd2 IPOKE_PREFETCH_2D=fwd
b p2d_bwd2 IPOKE_PREFETCH_2D=bwd
IPOKE_PREFETCH_2D=fwd python -m pytest tests/test_full_gpu.py -x -q -m gpu -k "prefetch" > $O/c4_prefetch_test.log 2>&1; tail -3 $O/c4_prefetch_test.log
