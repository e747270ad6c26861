#!/bin/bash
out=gpurun_out/r5_call39.txt; mkdir -p gpurun_out; : > $out
{
echo "### shared-image plane kernel at 3 / 4 bits: K split of a row group over items (GQ_PL_LOG2CS) and ring slots per wave (GQ_PL_S); us per launch"
for b in 3 4; do
for cfg in "" "GQ_PL_LOG2CS=0" "GQ_PL_LOG2CS=1" "GQ_PL_LOG2CS=2" "GQ_PL_S=1" "GQ_PL_RAWX=0" "GQ_PL_HIMG=0"; do
  a=$(env $cfg python tools/bench_ap.py --bits $b --shapes wqkv --launch norm 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  w=$(env $cfg python tools/bench_ap.py --bits $b --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  d=$(env $cfg GQ_PL_LOCAL=0 python tools/bench_ap.py --bits $b --shapes w2 --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  echo "bits $b [$cfg]: wqkv $a  w1w3 $w  w2 (shared-image kernel) $d"
done; done
} >> $out 2>&1
