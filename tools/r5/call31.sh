#!/bin/bash
out=gpurun_out/r5_call31.txt; mkdir -p gpurun_out; : > $out
{
echo "### stream kernel variants: base, consumer poll interval (sleep3 / sleep8 x 64 cycles), main loop aligned to 64 / 256 bytes"
for r in 1 2; do for v in base sleep3 sleep8 align6 align8; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  a=$(python tools/bench_ap.py --bits 2 --shapes wqkv --launch qkv_rope 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  b=$(python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  c=$(python tools/bench_ap.py --bits 2 --shapes w1w3 --launch plain 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  echo "$v: wqkv $a  w1w3 $b  w1w3 plain $c  $(python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
done; done
} >> $out 2>&1
