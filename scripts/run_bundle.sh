mkdir -p gpurun_out/fin
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -22 > gpurun_out/fin/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/fin/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/fin/bench.err | tail -1 > gpurun_out/fin/bench_line.json
cat gpurun_out/fin/tests.txt gpurun_out/fin/smoke.txt; tail -c 400 gpurun_out/fin/bench_line.json
