#!/bin/bash
out=gpurun_out/r5_call15.txt; mkdir -p gpurun_out; : > $out
{
echo "### driver form vs longer warm-up / more steps (same box)"
for r in 1 2; do
python bench.py --quick --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-120
python bench.py --quick --steps 20 --warmup 200 2>/dev/null | tail -1 | cut -c1-125
python bench.py --quick --steps 100 --warmup 5 2>/dev/null | tail -1 | cut -c1-125
python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-125
done
echo "### tests"; timeout 900 python -m pytest tests/test_decode_default_gpu.py tests/test_decode_gpu.py tests/test_hf_routes_gpu.py -q -m gpu 2>&1 | tail -4
} >> $out 2>&1
