R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
for sk in 0 1 2; do
echo "== IPOKE_NT_SKINNY=$sk"
IPOKE_NT_SKINNY=$sk IPOKE_CONV_LOG=1 python bench.py --config c4 --steps 2 --warmup 2 --no-cpu-baseline > $O/c47.out 2> $O/c47_log_$sk.txt
python scripts/conv_log_summary.py $O/c47_log_$sk.txt CONV 4 80 | grep "kern=1 " | grep -E "Nout=(64|32|16|8) " 
done
