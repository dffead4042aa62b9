R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_wgrad_batched_gpu.py -x -q -m gpu > $O/c7_wgrad.log 2>&1; tail -4 $O/c7_wgrad.log
python -m pytest tests/test_full_gpu.py -x -q -m gpu -k "epilogue_trains" > $O/c7_epi.log 2>&1; tail -6 $O/c7_epi.log
bash scripts/r6/ab.sh c7 "A=1" "IPOKE_WGRAD_ADAM=0" "A=2" "IPOKE_WGRAD_ADAM=0" "IPOKE_WGRAD_ADAM=2"
