R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
for mode in 1 0; do
  for i in 1 2 3 4 5 6 7 8; do
    IPOKE_WGRAD_HALO=$mode python -m pytest tests/test_train_mode_gpu.py tests/test_vae_bwd_units_gpu.py tests/test_c4_dispatch_gpu.py -x -q -m gpu -k "reproducible or parity_phases or dispatch" > $O/flaky_${mode}_$i.log 2>&1
    echo "halo=$mode run $i: $(tail -1 $O/flaky_${mode}_$i.log)"
    grep -E "^E |^FAILED" $O/flaky_${mode}_$i.log | head -6
  done
done
