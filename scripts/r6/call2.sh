set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_units_gpu.py -x -q -m gpu -k "split_k_accumulation or skinny" > $O/c2_units.log 2>&1; tail -3 $O/c2_units.log
timeout 1200 python -m pytest tests/test_full_gpu.py -x -q -m gpu -k "reproducible" -s > $O/c2_repro.log 2>&1; tail -12 $O/c2_repro.log
