#!/bin/bash
# QTIP matvec: the shipped library against the variant with the phase-stamp sites compiled in (tools/build_variant.sh qstamps qtip.hip -DGQ_STAMPS=1)
# -- the only difference between the round-4 and round-5 matvec kernels -- alternating, same box
mkdir -p gpurun_out
for r in 1 2; do
  for v in shipped qstamps; do
    if [ $v = shipped ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
    echo "== $v round $r"
    MV_ONLY=1 python3 tools/bench_qtip_mv.py 2>&1 | grep kernel
  done
done
unset GQ_LIB_PATH
