#!/bin/bash
out=gpurun_out/r5_call33.txt; mkdir -p gpurun_out; : > $out
{
echo "### stream kernel: unit image by two dense reads + DPP (new) vs eight masked reads (masked)"
timeout 900 python -m pytest tests/test_ap_stream_gpu.py -q -m gpu -x 2>&1 | tail -3
for r in 1 2 3; do for v in masked new; do
  if [ "$v" = new ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  a=$(python tools/bench_ap.py --bits 2 --shapes wqkv --launch qkv_rope 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  b=$(python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  echo "$v: wqkv $a  w1w3 $b  $(python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
done; done
unset GQ_LIB_PATH
echo "### GPU suite"; timeout 2000 python -m pytest tests -q -m gpu 2>&1 | tail -4
} >> $out 2>&1
