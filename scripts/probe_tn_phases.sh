#!/bin/bash
# Developer probe: builds the ablated variants of the LDS-DMA weight-gradient kernel (-DIPOKE_TN_ABL=1..4, gemm.hip) into
# scripts/exp/libtn_abl<N>.so (run HERE, hipcc cross-compiles); on the GPU box:  bash scripts/probe_tn_phases.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = run ]; then
  for n in 0 1 2 3 4; do
    lib=scripts/exp/libtn_abl$n.so; [ $n = 0 ] && lib=ipoke_amd/libipoke_hip.so
    echo -n "ABL=$n  "; IPOKE_LIB_PATH=$PWD/$lib python scripts/probe_tn.py 20
  done
  exit 0
fi
make -C ipoke_amd/csrc -j8 >/dev/null
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIPOKE_TN_ABL=$n -c ipoke_amd/csrc/gemm.hip -o /tmp/gemm_abl$n.o &
done
wait
for n in 1 2 3 4; do
  objs=$(ls ipoke_amd/csrc/build/*.o | grep -v /gemm.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_abl$n.o -o scripts/exp/libtn_abl$n.so
done
ls -la scripts/exp/*.so
