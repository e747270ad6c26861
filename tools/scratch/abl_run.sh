#!/bin/bash
# side-by-side timing of the library variants under guidedquant_amd/abl_*/ (GPU box): the decode step's four launch forms at 2 bits
for v in "$@"; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/bench_ap.py --bits ${ABL_BITS:-2} --shapes wqkv --launch norm | cut -c1-120
  python tools/bench_ap.py --bits ${ABL_BITS:-2} --shapes w1w3 --launch norm_pairs | cut -c1-120
  python tools/bench_ap.py --bits ${ABL_BITS:-2} --shapes wo w2 --launch resid | cut -c1-120
done
