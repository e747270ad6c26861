#!/bin/bash
out=gpurun_out/r5_call57_$(date +%s).txt; mkdir -p gpurun_out; : > $out
{
echo "### final build, driver form (--steps 20 --warmup 5) and 400 steps, one box"
for r in 1 2 3; do echo "driver form $(python bench.py --gpus 1 --steps 20 --warmup 5 --quick 2>/dev/null | tail -1 | cut -c40-75)"; done
echo "400 steps   $(python bench.py --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | cut -c40-75)"
echo "3-bit       $(python bench.py --bits 3 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75)"
echo "4-bit       $(python bench.py --bits 4 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75)"
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -1
} >> $out 2>&1
