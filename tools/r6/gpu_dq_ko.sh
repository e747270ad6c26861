#!/bin/bash
for v in shipped dqko1 dqko2 dqko3 dqko4 dqko7; do
  if [ $v = shipped ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  CHECK=0 BITS=2,3,4 timeout 300 python3 tools/r6/dq_check.py 2>&1 | grep -E "w1w3" | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); r = d['plain/graph-form us']; print('  bits', d['bits'], 'dq plain', r['dq'][0], r['dq2'][0])"
done
