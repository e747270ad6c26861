#!/bin/bash
out=gpurun_out/r5_call41.txt; mkdir -p gpurun_out; : > $out
{
echo "### one ring slot per wave (default now) vs GQ_PL_S1=0: decode tokens/s"
for b in 3 4; do for s1 in 0 1 0 1; do
echo "bits=$b GQ_PL_S1=$s1 $(GQ_PL_S1=$s1 python bench.py --bits $b --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75)"
done; done
echo "### full GPU suite"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4
} >> $out 2>&1
