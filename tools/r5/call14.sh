#!/bin/bash
out=gpurun_out/r5_call14.txt; mkdir -p gpurun_out; : > $out
{
for v in base notau oldexact; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"; timeout 600 python -m pytest tests/test_decode_default_gpu.py -q -m gpu -s -k "all_32_layers" 2>&1 | grep -v amdgpu | tail -12 | cut -c1-1500
done
} >> $out 2>&1
