#!/bin/bash
out=gpurun_out/r5_call44.txt; mkdir -p gpurun_out; : > $out
export TMPDIR=/tmp
R=$PWD
cd /tmp
{
for cfg in "X=0" "GQ_AP_BPC=3"; do
  rm -rf /tmp/px; echo "######## exact mode [$cfg]"
  env $cfg rocprofv3 --kernel-trace --stats -d /tmp/px -o t -- python $R/bench.py --mode exact --quick --steps 100 --warmup 50 > /tmp/px.log 2>&1
  for f in $(find /tmp/px -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-150 | head -9; done
  tail -1 /tmp/px.log | cut -c40-75
done
echo "### 4 bits with the local-image kernel for w2 (new default) -- decode"
cd $R; python bench.py --bits 4 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75
echo "### tests"; timeout 1200 python -m pytest tests/test_ap_fused_gpu.py tests/test_decode_default_gpu.py tests/test_ap_gemv_gpu.py -q -m gpu 2>&1 | tail -3
} >> $R/$out 2>&1
