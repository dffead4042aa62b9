set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
b() { tag=$1; shift; env "$@" python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 8 2>$O/c3_$tag.err | tail -1 > $O/c3_$tag.json; python - <<PY
import json; d=json.load(open("$O/c3_$tag.json")); print("$tag", d["ms_per_step"], d["ms_per_step_median"], d.get("handoff_timeouts"), d["loss"])
PY
}
b base A=1
b at_start IPOKE_PREFETCH_AT=start
b noprefetch IPOKE_NO_PREFETCH=1
b chain IPOKE_PREFETCH_STREAM=chain
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $O/c3_trace_run.log 2>&1
T=$(find /tmp/p_c2 -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/p_c2 -name "*kernel_stats.csv" | head -1) $O/r06a_bench_kernel_stats.csv
python $R/scripts/trace_gaps.py $T > $O/r06a_bench_trace_gaps.txt 2>&1
python $R/scripts/trace_overlap.py $T > $O/r06a_bench_overlap.txt 2>&1
python $R/scripts/trace_steady.py $T flow_nll 6 > $O/r06a_bench_steady.txt 2>&1
python $R/scripts/trace_step_tail.py $T > $O/r06a_step_tail.txt 2>&1
