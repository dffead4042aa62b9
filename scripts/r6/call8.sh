R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "stream_budget" -s > $O/c8_streams.log 2>&1; tail -6 $O/c8_streams.log
