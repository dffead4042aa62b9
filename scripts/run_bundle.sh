#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for tag in on off; do
  if [ $tag = off ]; then export IPOKE_NO_AN_EXT=1; fi
  rm -rf /tmp/prof_$tag
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o c5 -- python /root/repo/bench.py --config c5 --steps 20 --warmup 5 > /dev/null 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag $f"
  grep -E "extract_cols|actnorm_inv|affine_inv|unit_inv" $f | cut -c1-160
done
