set -x
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > gpurun_out/final_smoke.txt
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cat gpurun_out/final_tests.txt gpurun_out/final_smoke.txt
tail -c 3000 gpurun_out/final_bench.json
