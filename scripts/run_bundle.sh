mkdir -p gpurun_out/fin
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/fin/tests.txt
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/fin/r03_c5_bench_line.json
python bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/fin/r03_c4_bench_line.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/fin/r03_bench_line.json
cat gpurun_out/fin/tests.txt
