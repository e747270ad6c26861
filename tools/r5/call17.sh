#!/bin/bash
out=gpurun_out/r5_call17.txt; mkdir -p gpurun_out; : > $out
{
echo "### exact kernel, RMSNorm launch: block size x blocks per CU (wqkv 3 / 4 bit, default mode dispatch; us)"
for t in 0 256 512; do for b in 1 2 3; do
  echo "== T=$t BPC=$b  $(GQ_AP_T=$t GQ_AP_BPC=$b python tools/bench_ap.py --bits 3 4 --shapes wqkv --launch norm 2>&1 | grep shape | sed 's/.*"bits": \([0-9]\).*"us": \([0-9.]*\).*/\1bit \2/' | tr '\n' ' ')"
done; done
echo "### the same for the plain launch (wqkv 3 / 4 bit) and exact mode 2-bit norm (wqkv, w1w3)"
for t in 0 512; do for b in 1 2 3; do
  echo "== T=$t BPC=$b plain $(GQ_AP_T=$t GQ_AP_BPC=$b python tools/bench_ap.py --bits 3 4 --shapes wqkv 2>&1 | grep shape | sed 's/.*"bits": \([0-9]\).*"us": \([0-9.]*\).*/\1bit \2/' | tr '\n' ' ')  exact2 norm $(GQ_AP_EXACT=1 GQ_AP_T=$t GQ_AP_BPC=$b python tools/bench_ap.py --bits 2 --shapes wqkv w1w3 --launch norm 2>&1 | grep shape | sed 's/.*"shape": "\([a-z0-9]*\)".*"us": \([0-9.]*\).*/\1 \2/' | tr '\n' ' ')"
done; done
} >> $out 2>&1
