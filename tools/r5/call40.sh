#!/bin/bash
out=gpurun_out/r5_call40.txt; mkdir -p gpurun_out; : > $out
{
echo "### w1w3 at 3 / 4 bits (shared-image plane kernel): ring slots per wave x image helpers x K split; us per launch (two readings)"
for b in 3 4; do
for s in 1 2; do for h in 0 1; do for cs in 0 1; do
  cfg="GQ_PL_S=$s GQ_PL_HIMG=$h GQ_PL_LOG2CS=$cs"
  w1=$(env $cfg python tools/bench_ap.py --bits $b --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  w2=$(env $cfg python tools/bench_ap.py --bits $b --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  echo "bits $b [$cfg]: w1w3 $w1 $w2"
done; done; done; done
echo "### 70B shapes at 2 bits on the shared-image kernel (wqkv 10240x8192 norm, w1w3 57344x8192 norm_pairs): GQ_PL_S"
for cfg in "" "GQ_PL_S=1"; do
  echo "[$cfg] $(env $cfg python bench.py --model meta-llama/Llama-3.3-70B-Instruct --quick --steps 100 --warmup 20 2>/dev/null | tail -1 | cut -c40-75)"
done
} >> $out 2>&1
