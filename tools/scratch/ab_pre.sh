#!/bin/bash
# A/B on one box: the current library against guidedquant_amd/abl_pre (ap_plane.hip of an earlier commit), per-shape launches
for r in 1 2; do for v in base pre; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/bench_ap.py --bits 3 --shapes wqkv wo w1w3 w2 2>&1 | cut -c1-120 | grep shape
  python tools/bench_ap.py --bits 2 --shapes w2 wo 2>&1 | cut -c1-120 | grep shape
done; done
