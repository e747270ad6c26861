#!/bin/bash
out=gpurun_out/r5_call47.txt; mkdir -p gpurun_out; : > $out
{
echo "### local-image kernel (wo / w2, 2 bits): request order PL_XFIRST x priority scheme PL_PRIO; shipped = x2p4.  us per launch, then decode"
for v in base x0p0 x0p1 x0p4 x1p0 x1p1 x1p4 x2p0 x2p1 base; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  o=$(python tools/bench_ap.py --bits 2 --shapes wo w2 --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/' | tr '\n' ' ')
  o3=$(python tools/bench_ap.py --bits 3 --shapes wo w2 --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/' | tr '\n' ' ')
  echo "$v: 2-bit wo/w2 $o 3-bit $o3 $(python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
done
} >> $out 2>&1
