#!/bin/bash
# A/B on one box: exact-mode decode tokens/s (bench.py --mode exact --quick) of the current library against guidedquant_amd/abl_$1
V=$1
for r in 1 2 3; do for v in base $V; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== exact $v: $(python bench.py --mode exact --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')"
done; done
