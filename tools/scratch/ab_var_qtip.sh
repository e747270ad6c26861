#!/bin/bash
# A/B on one box: the current library against guidedquant_amd/abl_$1, QTIP decode (Llama-2-7b shape and power-of-two MLP) + bare matvec
V=$1
for r in 1 2; do for v in base $V; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/qtip_decode_bench.py 11008 32 2>&1 | tail -1 | cut -c1-140
  python tools/qtip_decode_bench.py 8192 32 2>&1 | tail -1 | cut -c1-140
  MV_ONLY=1 MV_SHAPE=11008x4096 python tools/bench_qtip_mv.py 2>&1 | grep -i matvec | tail -1 | cut -c1-160
done; done
