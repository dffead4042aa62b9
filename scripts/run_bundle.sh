set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4g
mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-secondary"
for v in "X=0" "GPU_MAX_HW_QUEUES=8" "IPOKE_COND_EARLY=1" "IPOKE_ENC_GRAPH=1 GPU_MAX_HW_QUEUES=8" "IPOKE_ENC_GRAPH=1 IPOKE_ENC_GRAPH_SIDE=0" "IPOKE_COND_EARLY=1 GPU_MAX_HW_QUEUES=8" "X=0" "IPOKE_COND_EARLY=1"; do env $v $B 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['ms_per_step_median'], d['loss'])" >> $O/ab.txt; tail -2 $O/err.txt | grep -v amdgpu.ids; done
cat $O/ab.txt
