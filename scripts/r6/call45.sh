R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_units_gpu.py -m gpu -x -q -s -k "one_chunk" 2>&1 | tail -8
python -m pytest tests -m gpu -x -q -k "first_stage or c4_dispatch or train_mode" 2>&1 | tail -3
for v in 1 0 1 0; do
  IPOKE_K8=$v python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c45_$v.err | tail -1 > $O/c45_$v.json
  python -c "import json;d=json.load(open('$O/c45_$v.json'));print('c4 K8=$v',d['ms_per_step'],d['loss'])" || tail -5 $O/c45_$v.err
done
IPOKE_CONV_LOG=1 python bench.py --config c4 --steps 2 --warmup 2 --no-cpu-baseline > $O/c45_c4.out 2> $O/c45_c4_log.txt
python scripts/conv_log_summary.py $O/c45_c4_log.txt CONV 4 40 | grep "Kc=8 "
