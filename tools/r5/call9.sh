#!/bin/bash
out=gpurun_out/r5_call9.txt; mkdir -p gpurun_out; : > $out
{
echo "### full GPU suite (verbose tail)"; timeout 1800 python -X faulthandler -m pytest tests/ -v -m gpu -x 2>&1 | grep -v "^  File\|PASSED" | tail -60
} >> $out 2>&1
