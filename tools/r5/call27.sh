#!/bin/bash
out=gpurun_out/r5_call27.txt; mkdir -p gpurun_out; : > $out
{
python -m pytest "tests/test_decode_default_gpu.py::test_default_mode_decode_at_8b_widths_matches_torch_forward" -q -m gpu 2>&1 | grep -E "AssertionError|assert |passed|failed" | head -20
echo "--- with wqkv back on the exact kernel (GQ_PL_MIN_MWEIGHTS=32 restores the old threshold for every width incl. 2-bit... so only bits 3/4 are comparable)"
GQ_PL_MIN_MWEIGHTS=32 python -m pytest "tests/test_decode_default_gpu.py::test_default_mode_decode_at_8b_widths_matches_torch_forward" -q -m gpu 2>&1 | grep -E "AssertionError|assert |passed|failed" | head -20
} >> $out 2>&1
