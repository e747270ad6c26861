#!/bin/bash
# A/B on one box: the current library against guidedquant_amd/abl_$1, shared-image launches (wqkv norm, w1w3 norm_pairs) + bench.py --quick
V=$1
for r in 1 2 3; do for v in base $V; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/bench_ap.py --bits 2 --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-120
  python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-120
  python tools/bench_ap.py --bits 3 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-120
  python tools/bench_ap.py --bits 4 --shapes w1w3 w2 2>&1 | grep shape | cut -c1-120
done; done
for v in base $V base $V; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== bench $v"; python bench.py --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | cut -c1-100
done
