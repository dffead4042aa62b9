R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
IPOKE_WGRAD_LOG=1 IPOKE_CONV_LOG=1 python bench.py --config c4 --steps 2 --warmup 2 --no-cpu-baseline > $O/c15_c4.out 2> $O/c15_c4_log.txt
python scripts/conv_log_summary.py $O/c15_c4_log.txt > $O/c15_c4_summary.txt 2>&1; head -70 $O/c15_c4_summary.txt
