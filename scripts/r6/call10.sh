R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_wgrad_batched_gpu.py -x -q -m gpu > $O/c10_wgrad.log 2>&1; tail -12 $O/c10_wgrad.log
bash scripts/r6/ab.sh c10 "A=1" "IPOKE_WGRAD_LAT8=0" "A=2" "IPOKE_WGRAD_LAT8=0"
