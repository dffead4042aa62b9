set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4j
mkdir -p $O
cd $R && timeout 900 python -m pytest tests/test_vae_bwd_units_gpu.py tests/test_train_mode_gpu.py tests/test_vae_gpu.py tests/test_vae_train_gpu.py tests/test_gan_step_gpu.py tests/test_full_gpu.py "tests/test_bench_configs_gpu.py::test_forward_sample_128_z64" "tests/test_bench_configs_gpu.py::test_sample_graph_replay_is_bit_identical" "tests/test_bench_configs_gpu.py::test_gru_and_spade_decoder_128_z64" -q -x 2>&1 | tail -25 > $O/tests.txt; cat $O/tests.txt
cd /tmp
for v in 0 1; do IPOKE_GRU_PYTHON=$v python $R/bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 gru_python=$v', d['ms_per_step'], d['loss'])" >> $O/ab.txt; tail -2 $O/err.txt | grep -v amdgpu; done
for v in 0 1; do IPOKE_GRU_PYTHON=$v python $R/bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 gru_python=$v', d['ms_per_step'], d['hipgraph']['full_graph_ms_per_step'])" >> $O/ab.txt; tail -2 $O/err.txt | grep -v amdgpu; done
cat $O/ab.txt
