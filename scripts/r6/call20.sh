R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_units_gpu.py -x -q -m gpu -k "narrow_input or conv_forward_vs_torch" -s 2>&1 | grep -v "^$" | tail -14
bash scripts/r6/ab.sh c20 "A=1" "IPOKE_K64=0" "A=2" "IPOKE_K64=0"
