#!/bin/bash
out=gpurun_out/r5_call36.txt; mkdir -p gpurun_out; : > $out
{ timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -5; } >> $out 2>&1
