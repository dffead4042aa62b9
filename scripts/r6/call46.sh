R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_units_gpu.py -m gpu -x -q -s -k "one_chunk" 2>&1 | tail -8
