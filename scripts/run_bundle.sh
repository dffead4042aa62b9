#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "unit or mcf or engine" 2>&1 | tail -3
python scripts/probe_unit.py 64 20 2>&1 | grep "fused\|next to"
for i in 1 2; do
for v in scripts/exp/libipoke_nt2.so ipoke_amd/libipoke_hip.so; do
  echo "== $v"; IPOKE_LIB_PATH=$PWD/$v python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"
done; done
