#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q 2>&1 | tail -5 ) 2>&1 | tail -9
