#!/bin/bash
out=gpurun_out/r5_call8.txt; mkdir -p gpurun_out; : > $out
{
echo "### local kernel: image build knocked out (GQ_PL_XFLAGS=64) vs full"
for r in 1 2; do
python tools/bench_ap.py --bits 2 --shapes wo w2 --launch resid 2>&1 | grep shape | cut -c1-150
GQ_PL_XFLAGS=64 python tools/bench_ap.py --bits 2 --shapes wo w2 --launch resid 2>&1 | grep shape | cut -c1-150
done
echo "### full GPU suite"; timeout 1800 python -m pytest tests/ -q -m gpu -x 2>&1 | grep -v "^  File" | tail -30
} >> $out 2>&1
