#!/bin/bash
out=gpurun_out/r5_call24.txt; mkdir -p gpurun_out; : > $out
{
echo "### GQ_ST_EPI (0 run-time flags, 1 RoPE instance for wqkv, 2 + pair instance for w1w3), bench --quick --steps 300"
for r in 1 2 3; do for e in 0 1 2; do
echo "GQ_ST_EPI=$e $(GQ_ST_EPI=$e python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
done; done
echo "### dispatch at 3 / 4 bits: default vs stream kernel for every shape it serves (GQ_ST=3) vs plane kernels for every shape (GQ_PL_MIN_MWEIGHTS=1)"
for b in 3 4; do
for cfg in "" "GQ_ST=3" "GQ_PL_MIN_MWEIGHTS=1" "GQ_PL_MIN_MWEIGHTS=1 GQ_ST=3"; do
  echo "== bits $b [$cfg]"
  env $cfg python tools/bench_ap.py --bits $b --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-150
  env $cfg python tools/bench_ap.py --bits $b --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-150
  env $cfg python tools/bench_ap.py --bits $b --shapes wo w2 --launch resid 2>&1 | grep shape | cut -c1-150
done; done
} >> $out 2>&1
