#!/bin/bash
# A/B of one environment knob on the decode step's four launch forms: tools/env_ab.sh NAME VAL_A VAL_B [rounds]
n=$1; a=$2; b=$3; r=${4:-3}
for i in $(seq $r); do for v in $a $b; do
  echo "== $n=$v"
  export $n=$v
  python tools/bench_ap.py --bits ${ABL_BITS:-2} --shapes wqkv --launch norm | cut -c1-120
  python tools/bench_ap.py --bits ${ABL_BITS:-2} --shapes w1w3 --launch norm_pairs | cut -c1-120
  python tools/bench_ap.py --bits ${ABL_BITS:-2} --shapes wo w2 --launch resid | cut -c1-120
done; done
