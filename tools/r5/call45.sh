#!/bin/bash
out=gpurun_out/r5_call45.txt; mkdir -p gpurun_out; : > $out
q() { python bench.py "$@" 2>/dev/null | tail -1 | cut -c40-75; }
{
echo "### after the env-cache fix"
echo "exact 2-bit  $(q --mode exact --quick --steps 200 --warmup 40)"
echo "exact 3-bit  $(q --mode exact --bits 3 --quick --steps 200 --warmup 40)"
echo "exact 4-bit  $(q --mode exact --bits 4 --quick --steps 200 --warmup 40)"
echo "qtip         $(q --backend qtip --quick --steps 200 --warmup 40)"
echo "default 2    $(q --quick --steps 300 --warmup 60)"
echo "default 3    $(q --bits 3 --quick --steps 200 --warmup 40)"
echo "default 4    $(q --bits 4 --quick --steps 200 --warmup 40)"
echo "### full GPU suite"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4
} >> $out 2>&1
