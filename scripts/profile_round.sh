# Round profile: kernel traces (+ stats, queue gaps, chain / side-stream overlap), PMC passes (separate runs), bench lines.
# Usage on the GPU box:  bash scripts/profile_round.sh r03
set -x
RN=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RN}
mkdir -p $O
# c2 kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $O/c2_trace_run.log 2>&1
cp $(find /tmp/p_c2 -name "*kernel_stats.csv" | head -1) $O/${RN}_bench_kernel_stats.csv
python $R/scripts/trace_gaps.py $(find /tmp/p_c2 -name "*kernel_trace.csv" | head -1) > $O/${RN}_bench_trace_gaps.txt 2>&1
python $R/scripts/trace_overlap.py $(find /tmp/p_c2 -name "*kernel_trace.csv" | head -1) > $O/${RN}_bench_overlap.txt 2>&1
python $R/scripts/trace_steady.py $(find /tmp/p_c2 -name "*kernel_trace.csv" | head -1) flow_nll 6 > $O/${RN}_bench_steady.txt 2>&1
# PMC passes (separate runs, counters only)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-secondary > $O/pmc_$c.log 2>&1
  python $R/scripts/pmc_summary.py $(find /tmp/p_$c -name "*counter_collection.csv" | head -1) $O/${RN}_bench_pmc_$c.json igemm_nt_glds igemm_nn_glds igemm_tn_glds conv3x3_s8 macow_unit adam_amsgrad adam_cast relayout
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p_sq -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-secondary > $O/pmc_sq.log 2>&1
python $R/scripts/pmc_summary.py $(find /tmp/p_sq -name "*counter_collection.csv" | head -1) $O/${RN}_bench_pmc_SQ.json igemm_nt_glds igemm_tn_glds conv3x3_s8 macow_unit adam_amsgrad
# c5 / c4 traces (the two-batches-in-flight loop of the c5 line is skipped under the profiler)
export IPOKE_BENCH_NO_PIPELINE=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python $R/bench.py --config c5 --steps 5 --warmup 3 --no-cpu-baseline > $O/c5_trace_run.log 2>&1
cp $(find /tmp/p_c5 -name "*kernel_stats.csv" | head -1) $O/${RN}_c5_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $R/bench.py --config c4 --steps 5 --warmup 3 --no-cpu-baseline > $O/c4_trace_run.log 2>&1
cp $(find /tmp/p_c4 -name "*kernel_stats.csv" | head -1) $O/${RN}_c4_kernel_stats.csv
# un-profiled bench lines
unset IPOKE_BENCH_NO_PIPELINE
python $R/bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${RN}_c5_bench_line.json
python $R/bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${RN}_c4_bench_line.json
python $R/bench.py --config c3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/${RN}_c3_bench_line.json
python $R/bench.py 2>/dev/null | tail -1 > $O/${RN}_bench_line.json
ls -la $O
# the counter files must describe the kernels of THIS tree (bench.py drops roofline.traffic otherwise)
python - <<PY || { echo "STALE PMC FILES: re-run the PMC passes"; exit 1; }
import hashlib, json, sys
h = hashlib.sha256(open("$R/ipoke_amd/csrc/gemm.hip", "rb").read()).hexdigest()[:16]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = json.load(open("$O/${RN}_bench_pmc_%s.json" % c))
    assert rows and all(r.get("gemm_hip_sha16") == h for r in rows), c
PY
