R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
IPOKE_CONV_LOG=1 python bench.py --config c4 --steps 2 --warmup 2 --no-cpu-baseline > $O/c44_c4.out 2> $O/c44_c4_log.txt
python scripts/conv_log_summary.py $O/c44_c4_log.txt CONV 4 32
