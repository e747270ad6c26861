#!/bin/bash
out=gpurun_out/r5_call21.txt; mkdir -p gpurun_out; : > $out
{
echo "### attention: the product launch (32 blocks, one head each) vs the whole-group form in N identical copies (probe build)"
for S in 33 65 101 129; do
echo "-- cache length $S: product"; ROPED=1 python tools/bench_attn.py $S 1 2>&1 | tail -4
for c in 32 64; do echo "-- cache length $S: probe copies=$c (grid 8 x $c, 8 waves)"; GQ_LIB_PATH=$PWD/guidedquant_amd/abl_attnprobe/libgq_hip.so GQ_ATTN_PROBE_COPIES=$c ROPED=1 python tools/bench_attn.py $S 1 2>&1 | tail -4; done
done
echo "### w1w3 / wqkv consumer: x as 4 fp32 partial vectors + residual (8 more 16-byte loads per builder lane)"
for v in base part4; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  for r in 1 2; do
  python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-150
  python tools/bench_ap.py --bits 2 --shapes wqkv --launch qkv_rope 2>&1 | grep shape | cut -c1-150
  done
done
unset GQ_LIB_PATH
echo "### icache ubench"; ./tools/ubench/icache 2>&1 | tail -20
} >> $out 2>&1
