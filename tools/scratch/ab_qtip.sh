#!/bin/bash
# A/B on one box: QTIP decode (Llama-2-7b shape) of the current library against guidedquant_amd/abl_$1, alternating, every run under a timeout
V=$1
for r in 1 2 3; do for v in base $V; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v: $(timeout 120 python tools/qtip_decode_bench.py 11008 32 2>&1 | tail -1 | grep -o 'tokens_per_sec": [0-9.]*')"
done; done
