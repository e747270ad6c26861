#!/bin/bash
# QTIP matvec: scheduling barriers at the (compiled-out) stamp sites of the band engine, one site at a time and all five
for v in shipped qstamps qsb31 qsb1 qsb2 qsb4 qsb8 qsb16 shipped qstamps; do
  if [ $v = shipped ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  MV_ONLY=1 python3 tools/bench_qtip_mv.py 2>&1 | grep kernel | python3 -c "
import sys, json
print('   '.join('%dx%d %.2f' % (d['M'], d['K'], d['us']) for d in map(json.loads, sys.stdin)))"
done
unset GQ_LIB_PATH
