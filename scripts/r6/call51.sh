R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
for v in 1 0; do
echo "== IPOKE_HALO16_PHASE=$v"
IPOKE_HALO16_PHASE=$v IPOKE_CONV_LOG=1 python bench.py --config c4 --steps 2 --warmup 2 --no-cpu-baseline > $O/c51.out 2> $O/c51_log_$v.txt
python scripts/conv_log_summary.py $O/c51_log_$v.txt CONV 4 90 | grep "k=1x2x2"
done
