R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_units_gpu.py -x -q -m gpu -k "column_slice" 2>&1 | tail -3
fails=0
for i in $(seq 1 16); do
  python -m pytest tests/test_train_mode_gpu.py tests/test_vae_bwd_units_gpu.py tests/test_c4_dispatch_gpu.py -x -q -m gpu -k "reproducible or parity_phases or dispatch" > $O/flaky2_$i.log 2>&1 || fails=$((fails+1))
done
echo "failures in 16 runs: $fails"
