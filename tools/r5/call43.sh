#!/bin/bash
out=gpurun_out/r5_call43.txt; mkdir -p gpurun_out; : > $out
q() { python bench.py "$@" 2>/dev/null | tail -1 | cut -c40-75; }
{
echo "### exact mode: default grid rule vs GQ_AP_BPC=3 / 2 (decode, then per launch)"
for r in 1 2; do
echo "default     $(q --mode exact --quick --steps 200 --warmup 40)"
echo "GQ_AP_BPC=3 $(GQ_AP_BPC=3 q --mode exact --quick --steps 200 --warmup 40)"
echo "GQ_AP_BPC=2 $(GQ_AP_BPC=2 q --mode exact --quick --steps 200 --warmup 40)"
done
export GQ_AP_EXACT=1
for b in 2 3 4; do for cfg in "" "GQ_AP_BPC=3" "GQ_AP_BPC=2"; do
  a=$(env $cfg python tools/bench_ap.py --bits $b --shapes wqkv --launch norm 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  w=$(env $cfg python tools/bench_ap.py --bits $b --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  o=$(env $cfg python tools/bench_ap.py --bits $b --shapes wo w2 --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/' | tr '\n' ' ')
  echo "exact bits $b [$cfg]: wqkv $a  w1w3 $w  wo/w2 $o"
done; done
} >> $out 2>&1
