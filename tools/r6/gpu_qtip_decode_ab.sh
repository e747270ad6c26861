#!/bin/bash
# QTIP decode (bench.py --backend qtip --quick) and the bare matvec: shipped library / every stamp site of qtip.hip compiled in / only the band engine's
for r in 1 2; do
for v in shipped qstamps qeng; do
  if [ $v = shipped ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v round $r"
  python3 bench.py --backend qtip --quick --steps 300 --warmup 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  decode', d['value'], 'tokens/s; matvec', d['roofline']['avg_launch_us'], 'us')"
  MV_ONLY=1 python3 tools/bench_qtip_mv.py 2>&1 | grep kernel | python3 -c "
import sys, json
print('  matvec ' + '   '.join('%dx%d %.2f' % (d['M'], d['K'], d['us']) for d in map(json.loads, sys.stdin)))"
done
done
unset GQ_LIB_PATH
