#!/bin/bash
# scratch: the command bundle of the last gpurun call (rewritten per call)
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
