R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "stream_budget" -s > $O/c9_streams.log 2>&1; grep -n "pairwise\|c2 step\|passed\|failed\|Warning" $O/c9_streams.log | head
bash scripts/r6/ab.sh c9 "A=1" "A=2"
