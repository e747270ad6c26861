#!/bin/bash
# A/B on one box: bench.py --quick (decode tokens/s) of the current library against guidedquant_amd/abl_$1, alternating, 3 rounds
V=$1
for r in 1 2 3; do for v in base $V; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== bench $v: $(python bench.py --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | cut -c1-70)"
done; done
