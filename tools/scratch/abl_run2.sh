#!/bin/bash
# like abl_run.sh, shared-image launches only (wqkv norm, w1w3 norm_pairs), two rounds
for r in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/bench_ap.py --bits 2 --shapes wqkv --launch norm | cut -c1-120
  python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs | cut -c1-120
done; done
