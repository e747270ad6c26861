#!/bin/bash
out=gpurun_out/r5_call53.txt; mkdir -p gpurun_out; : > $out
q() { python bench.py "$@" 2>/dev/null | tail -1 | cut -c40-75; }
{
echo "### request order of the local-image kernel per bit width (2 bits: 2, 3 / 4 bits: 0)"
echo "2-bit $(q --quick --steps 300 --warmup 60)"
echo "3-bit $(q --bits 3 --quick --steps 200 --warmup 40)"
echo "4-bit $(q --bits 4 --quick --steps 200 --warmup 40)"
echo "### full GPU suite"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4
} >> $out 2>&1
