R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -m pytest tests/test_c4_dispatch_gpu.py -m gpu -x -q -k "modulation or row_scale" 2>&1 | tail -3
for v in 1 0 1 0; do
  IPOKE_NORM_FRAMES_SUM=$v python bench.py --config c4 --steps 20 --warmup 6 --no-cpu-baseline 2>$O/c30_$v.err | tail -1 > $O/c30_$v.json
  python -c "import json;d=json.load(open('$O/c30_$v.json'));print('FRAMES_SUM=$v',d['ms_per_step'],d.get('loss'))" || tail -5 $O/c30_$v.err
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $R/bench.py --config c4 --steps 8 --warmup 3 --no-cpu-baseline > $O/c30_trace.log 2>&1
T=$(find /tmp/p_c4 -name "*kernel_trace.csv" | head -1)
python $R/scripts/trace_steady.py $T reparam_kernel 6 > $O/c30_c4_steady.txt 2>&1
grep -E "gn_bwd|sum_frames|steps of" $O/c30_c4_steady.txt
