#!/bin/bash
out=gpurun_out/r5_call52.txt; mkdir -p gpurun_out; : > $out
{
echo "### local-image kernel at 3 / 4 bits: request order x priority scheme (shipped = x2p4 = base); wo / w2 us per launch, decode"
for r in 1 2; do for v in base x1p4 x0p4 x1p0; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  o3=$(python tools/bench_ap.py --bits 3 --shapes wo w2 --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/' | tr '\n' ' ')
  o4=$(python tools/bench_ap.py --bits 4 --shapes w2 --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/' | tr '\n' ' ')
  echo "$v: 3-bit wo/w2 $o3 4-bit w2 $o4 decode3 $(python bench.py --bits 3 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-60) decode4 $(python bench.py --bits 4 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-60)"
done; done
} >> $out 2>&1
