#!/bin/bash
out=gpurun_out/r5_call56.txt; mkdir -p gpurun_out; : > $out
q() { python bench.py "$@" 2>/dev/null | tail -1 | cut -c40-75; }
{
echo "### lm_head rows per block (GQ_DENSE_RPB; default rule gives 176 = 729 blocks): 2-bit decode"
for v in 0 96 112 128 144 160 176 192 208 224 256 320 512 0; do echo "GQ_DENSE_RPB=$v $(GQ_DENSE_RPB=$v q --quick --steps 300 --warmup 60)"; done
} >> $out 2>&1
