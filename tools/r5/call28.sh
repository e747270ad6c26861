#!/bin/bash
out=gpurun_out/r5_call28.txt; mkdir -p gpurun_out; : > $out
{
echo "### Llama-3.3-70B 2-bit on one GPU: plane-kernel SPEC instances on / off"
for sp in 0 1 0 1; do
echo "GQ_PL_SPEC=$sp $(GQ_PL_SPEC=$sp python bench.py --model meta-llama/Llama-3.3-70B-Instruct --quick --steps 100 --warmup 20 2>/dev/null | tail -1 | cut -c40-75)"
done
echo "### tests touching the plane kernels"; timeout 1500 python -m pytest tests -q -m gpu -k "plane or fused or stream or decode_default or rows" 2>&1 | tail -3
} >> $out 2>&1
