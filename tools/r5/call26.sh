#!/bin/bash
out=gpurun_out/r5_call26.txt; mkdir -p gpurun_out; : > $out
{
echo "### full GPU suite"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8
} >> $out 2>&1
