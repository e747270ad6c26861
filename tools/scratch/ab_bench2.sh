#!/bin/bash
# A/B on one box: decode tokens/s (bench.py --quick, 2-bit AP) and the QTIP decode of the current library against guidedquant_amd/abl_$1
V=$1
for r in 1 2 3; do for v in base $V; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v: ap $(python bench.py --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')  qtip $(python tools/qtip_decode_bench.py 11008 32 2>&1 | tail -1 | grep -o 'tokens_per_sec": [0-9.]*')"
done; done
