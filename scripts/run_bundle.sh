mkdir -p gpurun_out/fin
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/fin/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/fin/smoke.txt
cat gpurun_out/fin/tests.txt gpurun_out/fin/smoke.txt
