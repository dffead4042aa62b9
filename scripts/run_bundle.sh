mkdir -p gpurun_out/r4c
timeout 1400 python -m pytest tests -m gpu -x -q --durations=30 > gpurun_out/r4c/tests.txt 2>&1
tail -45 gpurun_out/r4c/tests.txt
