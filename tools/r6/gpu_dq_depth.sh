#!/bin/bash
for v in shipped dqd2 dqd3; do
  if [ $v = shipped ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  if [ $v != shipped ]; then CHECK=1 BITS=2,3,4 timeout 300 python3 tools/r6/dq_check.py 2>&1 | grep -E "^ok|Error|error|assert" | tr '\n' ' '; echo; fi
  CHECK=0 BITS=${BITS:-2,3,4} timeout 300 python3 tools/r6/dq_check.py 2>&1 | grep -E "w1w3|w2|wqkv"
done
