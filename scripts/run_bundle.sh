#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 120 scripts/exp/ldpath_probe 2>&1 | tee gpurun_out/ldpath.txt
