# SQ counters of the train step (separate --pmc pass, kernel-trace only): matrix-core busy cycles, LDS stalls / bank conflicts per kernel
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02f
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_VALU_MFMA_BUSY_CYCLES\|SQ_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAIT_INST_LDS\|SQ_LDS_BANK_CONFLICT\|SQ_ACTIVE_INST_LDS\|SQ_INSTS_VALU_MFMA_MOPS_BF16\|SQ_INST_CYCLES_VMEM\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_ANY" | sort -u > $O/sq_counters_available.txt
cat $O/sq_counters_available.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p_sq -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_sq.log 2>&1
python $R/scripts/pmc_summary.py $(find /tmp/p_sq -name "*counter_collection.csv" | head -1) $O/r02_bench_pmc_SQ.json igemm_nt_glds igemm_tn_glds conv3x3_s8 macow_unit adam_amsgrad
tail -3 $O/pmc_sq.log
