#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q 2>&1 | tail -5 ) 2>&1 | tail -9
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
