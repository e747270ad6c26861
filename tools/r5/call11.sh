#!/bin/bash
out=gpurun_out/r5_call11.txt; mkdir -p gpurun_out; : > $out
{
echo "### full GPU suite"; timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -15
echo "### handover debug"; timeout 300 python tools/r5/ho_debug.py 2>&1 | grep -v amdgpu | tail -20
} >> $out 2>&1
