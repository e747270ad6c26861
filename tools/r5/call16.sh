#!/bin/bash
out=gpurun_out/r5_call16.txt; mkdir -p gpurun_out; : > $out
{
echo "### attention: cached rows requested ahead of the position (GQ_ATTN_SPEC), bench --quick --steps 300"
for r in 1 2; do for v in base spec64 spec96 spec128; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v $(python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c40-75)"
done; done
unset GQ_LIB_PATH
} >> $out 2>&1
