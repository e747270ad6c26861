#!/bin/bash
for v in shipped dqr1 dqr3; do
  if [ $v = shipped ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  CHECK=0 BITS=${BITS:-3,4} timeout 300 python3 tools/r6/dq_check.py 2>&1 | grep -E "w1w3|w2" 
done
