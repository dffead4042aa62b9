mkdir -p gpurun_out/fin
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/fin/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/fin/smoke.txt
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/fin/r03_c5_bench_line.json
python bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/fin/r03_c4_bench_line.json
python bench.py --config c4gan --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/fin/r03_c4gan_bench_line.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/fin/r03_bench_line.json
cat gpurun_out/fin/tests.txt gpurun_out/fin/smoke.txt
