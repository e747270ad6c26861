#!/bin/bash
out=gpurun_out/r5_call32.txt; mkdir -p gpurun_out; : > $out
{
echo "### stream kernel: builders at one priority (base) vs the later-started builder of a SIMD first (bprio)"
for r in 1 2 3; do for v in base bprio; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  a=$(python tools/bench_ap.py --bits 2 --shapes wqkv --launch qkv_rope 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  b=$(python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  echo "$v: wqkv $a  w1w3 $b  $(python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
done; done
} >> $out 2>&1
