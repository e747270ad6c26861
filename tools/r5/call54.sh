#!/bin/bash
out=gpurun_out/r5_call54.txt; mkdir -p gpurun_out; : > $out
{
echo "### 4 bits, wo on the exact kernel: ring depth x blocks per CU x block size; us per launch (resid)"
for d in 1 2 3; do for b in 1 2 3; do for t in 0 256 512; do
  echo "GQ_AP_D=$d GQ_AP_BPC=$b GQ_AP_T=$t: $(GQ_AP_D=$d GQ_AP_BPC=$b GQ_AP_T=$t python tools/bench_ap.py --bits 4 --shapes wo --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')"
done; done; done
echo "default: $(python tools/bench_ap.py --bits 4 --shapes wo --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')"
} >> $out 2>&1
